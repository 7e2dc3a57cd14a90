#!/bin/bash
# whole-step A/B of the stride-2 split-bf16 conv (conv2_bf3) against conv_dma on the same layers, interleaved on one box
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in s2fwd nos2; do
    if [ $v = nos2 ]; then export RCV_NO_BF3S2=1; unset RCV_BF3S2_FWD; else unset RCV_NO_BF3S2; export RCV_BF3S2_FWD=1; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', 'ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'))
"
  done
done
