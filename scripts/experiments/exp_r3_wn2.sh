#!/bin/bash
# conv_bf3 tile size on the smaller configs (experiments build librcv_X.so): the planner default against forced 320- / 160-pixel tiles
# (160x120: 1.58 default = 1.58 forced 160 < 1.65 forced 320; 320x240: 2.38-2.39 all three)
cd $GRAFT_REPO_ROOT
export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/librcv_X.so
for wl in robo_unet_160x120_bs64 robo_unet_320x240_bs32; do
for r in 1 2; do
  for wn in def 10 5; do
    if [ $wn = def ]; then unset RCV_BF3_WN; else export RCV_BF3_WN=$wn; fi
    timeout -k 10 300 python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$wl WN=$wn', 'ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'))
"
  done
done
done
