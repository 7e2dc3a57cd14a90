#!/bin/bash
# priority of the library's second stream (filter gradients): default vs lowest vs highest (experiments build)
cd $GRAFT_REPO_ROOT
export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/librcv_X.so
for r in 1 2; do
for p in none -1 1; do
  for w in robo_unet_640x480_bs32 robo_unet_160x120_bs64; do
    if [ $p = none ]; then unset RCV_SIDE_PRIO; else export RCV_SIDE_PRIO=$p; fi
    timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio=$p', d['config']['workload'], d['ms_per_step'], d['config'].get('second_stream'), d['config'].get('backward_ms_all_reduce_off'))"
  done
done; done
