#!/bin/bash
# interleaved A/B of two builds on the narrow (shared-role) filter-gradient shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/abwn.log
: > $O
A=robocupvision_amd/${ALIB:-librcv_A.so}; B=robocupvision_amd/librcv.so
R=${1:-3}
bash scripts/ab.sh $A $B $R -- wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc >> $O
bash scripts/ab.sh $A $B $R -- wgrad 32 240 320 16 32 --stride 2 --mode affine --mode2 grad_enc >> $O
bash scripts/ab.sh $A $B $R -- wgrad 32 480 640 8 16 --stride 2 --mode affine --mode2 grad_enc >> $O
bash scripts/ab.sh $A $B $R -- wgrad 32 480 640 8 8 --mode affine --mode2 grad_enc >> $O
bash scripts/ab.sh $A $B $R -- wgrad 32 120 160 32 16 --mode affine --mode2 grad_enc >> $O
python - $O <<'PY'
import sys, re, collections
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    m = re.match(r"(\S+) (\S+)\s+(.*?) tile=.*: ([0-9.]+) ms", line)
    if m: d[(m.group(3), m.group(1), m.group(2))].append(float(m.group(4)))
keys = sorted(set((k[0]) for k in d))
for k in keys:
    a = [v for (kk, l, lab), vv in d.items() if kk == k and l != "librcv.so" for v in vv]
    b = [v for (kk, l, lab), vv in d.items() if kk == k and l == "librcv.so" for v in vv]
    lab = [lab for (kk, l, lab) in d if kk == k][0]
    if a and b: print("%-26s %-70s A %.4f  B %.4f  B/A %.3f" % (lab, k[:70], sum(a)/len(a), sum(b)/len(b), (sum(b)/len(b))/(sum(a)/len(a))))
PY
