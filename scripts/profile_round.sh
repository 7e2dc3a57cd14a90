#!/bin/bash
# usage (GPU box): scripts/profile_round.sh TAG   -> gpurun_out/prof_TAG/{stats,fetch,write}
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# per-kernel numbers are taken with the backward pass on ONE stream (RCV_NO_SIDE_STREAM): with the filter gradients overlapped on the
# side stream a kernel's duration includes the time it shares the chip, which is not what bench.py's per-op HIP events report
export RCV_NO_SIDE_STREAM=1
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats exit=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD --no-roofline > /dev/null 2> $OUT/fetch.err
echo "fetch exit=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD --no-roofline > /dev/null 2> $OUT/write.err
echo "write exit=$?"
find $OUT -name "*.csv" | head -20
