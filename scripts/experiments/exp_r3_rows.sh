#!/bin/bash
# per-kernel table of the headline step (HIP-event timed op by op)
cd $GRAFT_REPO_ROOT
RCV_BENCH_ROWS=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --breakdown > gpurun_out/rows.json 2> gpurun_out/rows.err
echo "exit=$?"; grep -v "^[BF] " gpurun_out/rows.err | tail -45
