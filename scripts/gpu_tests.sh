#!/bin/bash
# usage (GPU box): scripts/gpu_tests.sh TAG  -> gpurun_out/test_TAG.log (the -m gpu suite as the driver runs it) + smoke
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -p no:cacheprovider -s > gpurun_out/test_$TAG.log 2>&1
echo "tests exit=$?"; grep -E "passed|failed|error" gpurun_out/test_$TAG.log | tail -5; grep -E "mask pixels differ" gpurun_out/test_$TAG.log | sort | uniq -c | head -30
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
echo "smoke exit=$?"; tail -2 gpurun_out/smoke_$TAG.log
