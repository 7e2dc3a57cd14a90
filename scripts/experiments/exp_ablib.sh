#!/bin/bash
# whole-step A/B of two library builds (robocupvision_amd/librcv_A.so vs librcv.so) on one box: LabelProp B=64, headline, U-Net, 160x120
cd $GRAFT_REPO_ROOT
O=gpurun_out/ablib.log
: > $O
for r in 1 2; do
for L in librcv_A.so librcv.so; do
  for w in ${WORKLOADS:-labelprop_160x120_b64 robo_unet_640x480_bs32 unet_640x480_bs32 robo_unet_160x120_bs64}; do
    steps=20; if [[ $w == labelprop* ]]; then steps=300; fi
    RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/$L timeout -k 10 300 python bench.py --workload $w --steps $steps --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config']['workload'], d['ms_per_step'])" >> $O
  done
done; done
sort $O | awk '{k=$1" "$2; s[k]+=$3; n[k]++} END {for (k in s) printf "%-50s %.4f\n", k, s[k]/n[k]}' | sort -k2
