B="python scripts/bench_op.py"
for args in "wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc" "wgrad 32 60 80 64 64 --mode affine --mode2 grad_enc" "wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc" "wgrad 32 480 640 3 8 --mode nchw --mode2 grad_enc"; do
  echo "== $args"
  $B $args | awk -F" : " "{print \$2}"
  $B $args --flags $((1<<20)) | awk -F" : " "{print \$2}"
  $B $args --flags $((1<<21)) | awk -F" : " "{print \$2}"
  RCV_WGRAD_OCC=1 $B $args | awk -F" : " "{print \$2}"
done
