#!/bin/bash
# wave-specialised filter gradient (128->128): dynamic instruction mix, whole kernel / without staging / without MFMAs
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_wgmix
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local tag=$1; shift
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM \
    --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/scripts/bench_op.py "$@" --reps 2 > $OUT/$tag.log 2> $OUT/$tag.err
  echo "$tag exit=$?"
}
run full wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc
run nostage wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc --flags 1048576
run nomfma wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc --flags 2097152
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("full", "nostage", "nomfma"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wgrad_mfma" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(tag, {k: round(v / max(n[k], 1)) for k, v in sorted(acc.items())})
PY
