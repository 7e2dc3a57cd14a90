#!/bin/bash
# usage (GPU box): scripts/bench_rows.sh TAG [bench args]  -> gpurun_out/bench_TAG.json / .err (per-kernel table + per-op rows)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
RCV_BENCH_ROWS=1 timeout -k 10 500 python bench.py --steps 20 --warmup 5 --breakdown --no-cpu-baseline "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench exit=$?"; cat gpurun_out/bench_$TAG.json; grep -v amdgpu.ids gpurun_out/bench_$TAG.err | head -30
