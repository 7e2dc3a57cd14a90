#!/bin/bash
# usage (GPU box): scripts/profile_config.sh TAG WORKLOAD  -> gpurun_out/prof_TAG/stats (kernel trace + stats of one bench.py workload)
TAG=$1; W=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RCV_NO_SIDE_STREAM=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats exit=$?"
