#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/side.log
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_dp.py tests/test_a_dist_gpu.py -m gpu -x -q -p no:cacheprovider > gpurun_out/side_tests.log 2>&1
echo "tests exit=$?" >> $O; tail -1 gpurun_out/side_tests.log >> $O
for w in robo_unet_640x480_bs32 robo_unet_160x120_bs64 unet_640x480_bs32 robo_unet_320x240_bs32; do
  timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'], d['config'].get('second_stream'), d['config'].get('backward_ms_all_reduce_off'))" >> $O
done
cat $O
