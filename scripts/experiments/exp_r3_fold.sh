#!/bin/bash
# eval-mode BatchNorm folding (EVAL_FOLD_BN): parity tests of the inference graphs, then LabelProp / PB_FCN latency with and without it
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_pbfcn.py tests/test_gpu_blocks.py -x -q -m gpu -p no:cacheprovider -k "labelprop or LabelProp or eval or infer or golden or pbfcn or PB" > gpurun_out/fold_tests.log 2>&1
echo "tests exit=$?"; tail -5 gpurun_out/fold_tests.log
for v in fold nofold fold nofold; do
  if [ $v = nofold ]; then export RCV_NO_EVAL_FOLD=1; else unset RCV_NO_EVAL_FOLD; fi
  for wl in labelprop_160x120_b2 labelprop_160x120_b64; do
    timeout -k 10 200 python bench.py --workload $wl --steps 400 --warmup 50 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['config']['workload'], 'ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'))
"
  done
done
