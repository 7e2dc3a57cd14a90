#!/bin/bash
# whole-step hipGraph replay against the eager two-stream schedule, re-measured with the split-bf16 kernels
cd $GRAFT_REPO_ROOT
for wl in robo_unet_640x480_bs32 robo_unet_160x120_bs64; do
  for g in 0 1; do
    timeout -k 10 300 python bench.py --workload $wl --graph $g --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl graph=$g ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'))
"
  done
done
