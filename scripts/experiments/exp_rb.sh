#!/bin/bash
# batched filter-gradient reductions (engine.REDUCE_BATCH) on/off: step time of four workloads, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/rb.log
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_dp.py -m gpu -x -q -p no:cacheprovider > gpurun_out/rb_tests.log 2>&1
echo "tests exit=$?" >> $O; tail -1 gpurun_out/rb_tests.log >> $O
for r in 1 2; do
for rb in 1 6; do
for w in robo_unet_640x480_bs32 robo_unet_160x120_bs64 unet_640x480_bs32 robo_unet_320x240_bs32; do
  RCV_REDUCE_BATCH=$rb timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rb=$rb', d['config']['workload'], d['ms_per_step'], d['config'].get('second_stream'))" >> $O
done; done; done
cat $O
