#!/bin/bash
# interleaved A/B of two builds on the Winograd shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/abw.log
: > $O
timeout -k 10 300 python -m pytest tests/test_gpu_blocks.py -m gpu -x -q -p no:cacheprovider -k "winograd" > gpurun_out/abw_tests.log 2>&1
echo "tests exit=$?" >> $O; tail -1 gpurun_out/abw_tests.log >> $O
A=robocupvision_amd/librcv_A.so; B=robocupvision_amd/librcv.so
bash scripts/ab.sh $A $B 3 -- conv 32 30 40 128 128 --mode affine --stats fwd --wino 1 >> $O
bash scripts/ab.sh $A $B 3 -- conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc --resid 1 --wino 1 >> $O
bash scripts/ab.sh $A $B 2 -- conv 32 60 80 64 64 --mode affine --stats fwd --wino 1 >> $O
python - $O <<'PY'
import sys, re, collections
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    m = re.match(r"(\S+) (\S+)\s+(.*?) tile=.*: ([0-9.]+) ms", line)
    if m: d[(m.group(3), m.group(1))].append(float(m.group(4)))
    elif "tests" in line or "passed" in line or "failed" in line: print(line.strip())
for k in sorted(set(k[0] for k in d)):
    a, b = d.get((k, "librcv_A.so"), []), d.get((k, "librcv.so"), [])
    if a and b: print("%-70s A %.4f  B %.4f  B/A %.3f" % (k[:70], sum(a)/len(a), sum(b)/len(b), (sum(b)/len(b))/(sum(a)/len(a))))
PY
