#!/bin/bash
# interleaved A/B of robocupvision_amd/librcv_A.so (reference build) and librcv.so on the dominant conv shapes; parity tests first
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab.log
: > $O
timeout -k 10 500 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_net.py -m gpu -x -q -p no:cacheprovider > gpurun_out/ab_tests.log 2>&1
echo "tests exit=$?" >> $O; tail -1 gpurun_out/ab_tests.log >> $O
A=robocupvision_amd/librcv_A.so; B=robocupvision_amd/librcv.so
bash scripts/ab.sh $A $B 3 -- conv 32 30 40 128 128 --mode affine --stats fwd >> $O
bash scripts/ab.sh $A $B 3 -- conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc --resid 1 >> $O
bash scripts/ab.sh $A $B 2 -- conv 32 60 80 64 64 --mode affine --stats fwd >> $O
bash scripts/ab.sh $A $B 2 -- conv 32 60 80 64 128 --stride 2 --mode affine --stats fwd >> $O
bash scripts/ab.sh $A $B 2 -- conv 32 120 160 32 64 --stride 2 --mode grad_dec --stats bwd_enc --resid 1 >> $O
bash scripts/ab.sh $A $B 2 -- tconv 32 30 40 128 64 --mode grad_enc --stats bwd_enc --resid 1 >> $O
python - $O <<'PY'
import sys, re, collections
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    m = re.match(r"(\S+) (\S+)\s+(.*?) tile=.*: ([0-9.]+) ms", line)
    if m: d[(m.group(3), m.group(1))].append(float(m.group(4)))
    elif "tests" in line or "passed" in line or "failed" in line: print(line.strip())
keys = sorted(set(k[0] for k in d))
for k in keys:
    a, b = d.get((k, "librcv_A.so"), []), d.get((k, "librcv.so"), [])
    if a and b: print("%-70s A %.4f  B %.4f  B/A %.3f" % (k[:70], sum(a)/len(a), sum(b)/len(b), (sum(b)/len(b))/(sum(a)/len(a))))
PY
