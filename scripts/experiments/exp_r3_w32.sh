#!/bin/bash
# conv_bf3 on v_mfma_f32_32x32x16_bf16 (320-pixel tile): accuracy tests, op timings, whole step
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -s -k "test_conv_kernels_vs_fp64" > gpurun_out/w32_tests.log 2>&1
echo "tests exit=$?"; grep -E "passed|failed|Error|error" gpurun_out/w32_tests.log | tail -5; grep "conv_bf3<64,320>" gpurun_out/w32_tests.log | tail -10
run() { timeout -k 10 120 python scripts/bench_op.py "$@" --reps 50 2>&1 | tail -1 | sed -E 's/ N32 / /; s/mode=//; s/stats=//; s/merged=[0-9] tile=- //'; }
run conv 32 30 40 128 128 --mode affine --stats fwd --wino 3
run conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc --wino 3
run conv 32 60 80 64 64 --mode affine --stats fwd --wino 3
run conv 32 60 80 64 64 --mode grad_enc --stats bwd_enc --wino 3
for r in 1 2; do
timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('step ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'))
"
done
