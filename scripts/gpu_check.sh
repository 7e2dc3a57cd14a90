#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/gpu_check.sh TAG [bench args...]
# runs the GPU parity tests, then bench.py with the per-kernel breakdown; everything lands in gpurun_out/
TAG=${1:-x}; shift
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -q --timeout 240 -p no:cacheprovider > gpurun_out/test_$TAG.log 2>&1
echo "tests exit=$?"; tail -3 gpurun_out/test_$TAG.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --breakdown --no-cpu-baseline "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench exit=$?"; cat gpurun_out/bench_$TAG.json; grep -v amdgpu.ids gpurun_out/bench_$TAG.err | head -24
