#!/bin/bash
# interleaved A/B of two builds on the wave-specialised filter-gradient shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/abw2.log
: > $O
A=robocupvision_amd/${ALIB:-librcv_A.so}; B=robocupvision_amd/librcv.so
R=${1:-3}
bash scripts/ab.sh $A $B $R -- wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc >> $O
bash scripts/ab.sh $A $B $R -- wgrad 32 60 80 64 64 --mode affine --mode2 grad_enc >> $O
bash scripts/ab.sh $A $B $R -- wgrad 32 60 80 64 128 --stride 2 --mode affine --mode2 grad_enc >> $O
bash scripts/ab.sh $A $B $R -- wgrad 32 120 160 32 32 --mode affine --mode2 grad_enc >> $O
bash scripts/ab.sh $A $B $R -- wgrad 32 120 160 32 64 --stride 2 --mode affine --mode2 grad_enc >> $O
python - $O <<'PY'
import sys, re, collections
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    m = re.match(r"(\S+) (\S+)\s+(.*?) tile=.*: ([0-9.]+) ms", line)
    if m: d[(m.group(3), m.group(1))].append(float(m.group(4)))
for k in sorted(set(k[0] for k in d)):
    a, b = [v for (kk, l), vv in d.items() if kk == k and l != "librcv.so" for v in vv], d.get((k, "librcv.so"), [])
    if a and b: print("%-84s A %.4f  B %.4f  B/A %.3f" % (k[:84], sum(a)/len(a), sum(b)/len(b), (sum(b)/len(b))/(sum(a)/len(a))))
PY
