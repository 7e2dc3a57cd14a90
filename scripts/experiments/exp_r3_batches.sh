#!/bin/bash
# other per-GPU batch sizes through bench.py (plans change: some split-bf16 kernels leave the plan below 24 images): sanity + throughput
cd $GRAFT_REPO_ROOT
for b in 8 16 24 48; do
  timeout -k 10 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('batch $b', d['value'], 'img/s', d['ms_per_step'], 'ms loss', d['config']['loss_after'])
"
done
