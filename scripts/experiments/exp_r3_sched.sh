#!/bin/bash
# schedule knobs re-measured with the split-bf16 kernels in the step (their durations changed what overlaps with what)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'))
"; }
for r in 1 2; do
  run default
  RCV_DGRAD_FIRST=1 run dgrad_first
  RCV_SIDE_STREAM=all run side_all
  RCV_SIDE_STREAM=off run side_off
  RCV_REDUCE_BATCH=6 run reduce6
done
