#!/bin/bash
# round 3: rehearsal of the N > 1 path with 2 and 4 ranks (the box allows 6 GPU processes, launcher included) sharing the box's one GPU over gloo (RCCL refuses several ranks per device).
# Functional only -- the timings mean nothing.  The loss after the timed steps must not depend on how many ranks ran (same seed per rank
# index, gradients averaged): printed for comparison with the single-rank line.
cd $GRAFT_REPO_ROOT
export RCV_DIST_BACKEND=gloo
python bench.py --workload robo_unet_160x120_bs64 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline | cut -c1-120,380-520
for n in 2 4; do
  timeout -k 10 300 python bench.py --gpus $n --workload robo_unet_160x120_bs64 --batch 8 --steps 4 --warmup 2 --no-roofline | cut -c1-120,380-520
done
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 2 --no-roofline | cut -c1-140
