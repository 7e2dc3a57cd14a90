#!/bin/bash
# usage (GPU box): scripts/profile_all.sh TAG
#   headline workload: kernel trace + stats, FETCH_SIZE pass, WRITE_SIZE pass (separate --pmc runs, MI355X_MICROARCH.md section HBM)
#   configs 2, 3, 5:   kernel trace + stats, FETCH_SIZE, WRITE_SIZE
#   dominant kernel:   SQ counters of one 128->128 launch (scripts/pmc_op.sh)
# everything lands in gpurun_out/prof_TAG*/ ; scripts/summarize_profile.py copies the summaries into profiles/
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
# per-kernel numbers are taken with the backward pass on ONE stream: with the filter gradients overlapped on the side stream a
# kernel's duration includes the time it shares the chip, which is not what bench.py's per-op HIP events report
export RCV_NO_SIDE_STREAM=1
prof() {   # name, bench args...
  local name=$1; shift
  local OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$name
  mkdir -p $OUT
  local CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline $*"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_stats.json 2> $OUT/stats.err
  echo "$name stats exit=$?"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD --no-roofline > /dev/null 2> $OUT/fetch.err
  echo "$name fetch exit=$?"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD --no-roofline > /dev/null 2> $OUT/write.err
  echo "$name write exit=$?"
}
prof ${TAG}
prof ${TAG}_robo_unet_160x120_bs64 --workload robo_unet_160x120_bs64
prof ${TAG}_unet_640x480_bs32 --workload unet_640x480_bs32
prof ${TAG}_labelprop_160x120_b2 --workload labelprop_160x120_b2 --steps 50
prof ${TAG}_labelprop_160x120_b64 --workload labelprop_160x120_b64 --steps 50
unset RCV_NO_SIDE_STREAM
cd $GRAFT_REPO_ROOT && bash scripts/pmc_op.sh ${TAG}_conv128 conv 32 30 40 128 128 --mode affine --stats fwd > /dev/null 2>&1
echo "pmc exit=$?"
cd $GRAFT_REPO_ROOT && bash scripts/pmc_op.sh ${TAG}_wino128 conv 32 30 40 128 128 --mode affine --stats fwd --wino 1 > /dev/null 2>&1
echo "pmc wino exit=$?"
cd $GRAFT_REPO_ROOT && bash scripts/pmc_op.sh ${TAG}_bf3conv128 conv 32 30 40 128 128 --mode affine --stats fwd --wino 3 > /dev/null 2>&1
echo "pmc bf3 exit=$?"
