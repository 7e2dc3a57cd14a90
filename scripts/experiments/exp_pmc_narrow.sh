#!/bin/bash
# narrow conv kernel: dynamic instruction mix of one launch (vector ALU / MFMA / scalar / LDS / memory instructions per wave)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_narrow
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local tag=$1; shift
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM \
    --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/scripts/bench_op.py "$@" --reps 2 > $OUT/$tag.log 2> $OUT/$tag.err
  echo "$tag exit=$?"
}
run f16 conv 32 240 320 16 16 --mode affine --stats fwd
run b16 conv 32 240 320 16 16 --mode grad_enc --stats bwd_enc
run f8 conv 32 480 640 8 8 --mode affine --stats fwd
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("f16", "b16", "f8"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "convs_mfma" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(tag, {k: round(v / max(n[k], 1)) for k, v in sorted(acc.items())})
PY
