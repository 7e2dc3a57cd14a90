#!/bin/bash
# whole-step A/B: split-bf16 filter gradients (default) against RCV_MFMA_FP32=1, same box, interleaved
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in bf3 fp32; do
    if [ $v = fp32 ]; then export RCV_MFMA_FP32=1; else unset RCV_MFMA_FP32; fi
    for wl in robo_unet_640x480_bs32 unet_640x480_bs32; do
      timeout -k 10 300 python bench.py --workload $wl --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['config']['workload'], 'ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'img/s', d['value'])
"
    done
  done
done
