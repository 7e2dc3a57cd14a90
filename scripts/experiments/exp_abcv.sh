#!/bin/bash
# interleaved A/B of two builds on the narrow convolution shapes of the headline step (+ parity tests with build B)
cd $GRAFT_REPO_ROOT
O=gpurun_out/abcv.log
: > $O
timeout -k 10 400 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_net.py -m gpu -x -q -p no:cacheprovider > gpurun_out/abcv_tests.log 2>&1
echo "tests exit=$?" >> $O; tail -1 gpurun_out/abcv_tests.log >> $O
A=robocupvision_amd/${ALIB:-librcv_A.so}; B=robocupvision_amd/librcv.so
R=${1:-2}
bash scripts/ab.sh $A $B $R -- conv 32 240 320 16 16 --mode affine --stats fwd >> $O
bash scripts/ab.sh $A $B $R -- conv 32 240 320 16 16 --mode grad_enc --stats bwd_enc --resid 1 >> $O
bash scripts/ab.sh $A $B $R -- conv 32 120 160 32 32 --mode grad_enc --stats bwd_enc --resid 1 >> $O
bash scripts/ab.sh $A $B $R -- conv 32 480 640 8 16 --stride 2 --mode affine --stats fwd >> $O
bash scripts/ab.sh $A $B $R -- tconv 32 240 320 16 8 --merged 1 --mode affine --stats fwd >> $O
bash scripts/ab.sh $A $B $R -- tconv 32 240 320 16 8 --merged 1 --mode grad_enc --stats bwd_enc --resid 1 >> $O
bash scripts/ab.sh $A $B $R -- tconv 32 120 160 32 16 --merged 1 --mode affine --stats fwd >> $O
bash scripts/ab.sh $A $B $R -- tconv 32 120 160 32 16 --merged 1 --mode grad_enc --stats bwd_enc --resid 1 >> $O
bash scripts/ab.sh $A $B $R -- tconv 32 120 160 32 16 --merged 1 --mode grad_dec --stats bwd_dec --resid 1 >> $O
python - $O <<'PY'
import sys, re, collections
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    m = re.match(r"(\S+) (\S+)\s+(.*?) tile=.*: ([0-9.]+) ms", line)
    if m: d[(m.group(3), m.group(1))].append(float(m.group(4)))
    elif "tests" in line or "passed" in line or "failed" in line: print(line.strip())
for k in sorted(set(k[0] for k in d)):
    a = [v for (kk, l), vv in d.items() if kk == k and l != "librcv.so" for v in vv]; b = d.get((k, "librcv.so"), [])
    if a and b: print("%-84s A %.4f  B %.4f  B/A %.3f" % (k[:84], sum(a)/len(a), sum(b)/len(b), (sum(b)/len(b))/(sum(a)/len(a))))
PY
