#!/bin/bash
# conv_bf3 tile size inside the step (experiments build librcv_X.so, RCV_BF3_WN): 320-pixel tiles (one round of 256 workgroups on the
# 128-channel layers) against 160-pixel tiles (two rounds of 512)
cd $GRAFT_REPO_ROOT
export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/librcv_X.so
for r in 1 2 3; do
  for wn in 10 5; do
    RCV_BF3_WN=$wn timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('WN=$wn', 'ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'))
"
  done
done
