#!/bin/bash
# usage (GPU box): scripts/pmc_op.sh TAG <bench_op.py args...>  -> SQ counters of ONE op, summed per kernel (gpurun_out/pmc_TAG.txt)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/scripts/bench_op.py "$@" --reps 3 > $OUT/a.log 2> $OUT/a.err
echo "pass a exit=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 \
  --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/scripts/bench_op.py "$@" --reps 3 > $OUT/b.log 2> $OUT/b.err
echo "pass b exit=$?"
# (the split-bf16 kernels issue bf16 MFMAs; a pass of its own: if the counter name is unknown to this rocprofv3 only this pass fails)
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/c -- python3 $GRAFT_REPO_ROOT/scripts/bench_op.py "$@" --reps 3 > $OUT/c.log 2> $OUT/c.err
echo "pass c exit=$?"
python3 - $OUT <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.txt
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("a", "b", "c"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "conv" not in k and "wgrad" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            print("   %-32s %16.0f  per launch %14.0f" % (c, v, v / max(n[(k, c)], 1)))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.txt
