#!/bin/bash
# round 3: the 32 -> 64 stride-2 data gradient (two-tensor load, conv_dma<1,5,4,1>): time and HBM fetch per tile shape / chunk width
cd $GRAFT_REPO_ROOT
export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/librcv_X.so
ARGS="conv 32 120 160 32 64 --stride 2 --mode grad_dec --stats bwd_dec"
run() {   # name, env...
  local name=$1; shift
  env "$@" python scripts/bench_op.py $ARGS | sed "s|^|$name |"
  (cd /tmp && export TMPDIR=/tmp && env "$@" timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_dg_$name -- python3 $GRAFT_REPO_ROOT/scripts/bench_op.py $ARGS --reps 2 > /dev/null 2>&1)
  python3 - $GRAFT_REPO_ROOT/gpurun_out/pmc_dg_$name $name <<'PY'
import csv, glob, sys
tot = n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_dma" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print("%s FETCH_SIZE per launch: raw %.0f KB -> %.1f MB (x2 gfx950 correction)" % (sys.argv[2], tot / max(n, 1), 2 * tot / max(n, 1) / 1024))
PY
}
run default A=1
run t4_2x20 RCV_CONV_TILE=4,2,20
run t4_1x40 RCV_CONV_TILE=4,1,40
run t4_1x80 RCV_CONV_TILE=4,1,80
run t4_2x40 RCV_CONV_TILE=4,2,40
run t4_4x20 RCV_CONV_TILE=4,4,20
