#!/bin/bash
# A/B of two library builds on the whole step: per-family kernel times (classifier rows) from bench.py --breakdown
cd $GRAFT_REPO_ROOT
O=gpurun_out/cls.log
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_blocks.py tests/test_gpu_dp.py -m gpu -x -q -p no:cacheprovider > gpurun_out/cls_tests.log 2>&1
echo "tests exit=$?" >> $O; tail -1 gpurun_out/cls_tests.log >> $O
for r in 1 2; do
for L in librcv_A.so librcv.so; do
  echo "== $L" >> $O
  RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/$L timeout -k 10 300 python bench.py --steps 20 --warmup 5 --breakdown --no-cpu-baseline "$@" 2>&1 >/dev/null | grep -E "^cls_|^adam|^wgrad_first|^conv_first" >> $O
  RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/$L timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $O
done; done
cat $O
