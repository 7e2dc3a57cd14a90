#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_dp.py tests/test_gpu_net.py -m gpu -x -q -p no:cacheprovider -k "adam or trainer or metrics or prune or captured" > gpurun_out/adam_tests.log 2>&1
echo "tests exit=$?"; tail -1 gpurun_out/adam_tests.log
cd /tmp && export TMPDIR=/tmp
for w in robo_unet_640x480_bs32 robo_unet_160x120_bs64; do
  rm -rf /tmp/pa; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  grep -h "adam_l1" /tmp/pa/*/*kernel_stats.csv | cut -c1-40,240-330
done
