B="python scripts/bench_op.py"
for args in "conv 32 240 320 16 16 --mode affine --stats fwd" "conv 32 240 320 16 16 --mode grad_enc --stats bwd_enc --resid 1" "conv 32 480 640 8 16 --stride 2 --mode affine --stats fwd" "tconv 32 240 320 16 8 --mode grad_enc --stats bwd_enc --resid 1 --merged 1" "conv 32 480 640 3 8 --mode nchw --stats fwd" "conv 32 240 320 16 32 --stride 2 --mode affine --stats fwd"; do
  echo "== $args"
  $B $args | awk -F" : " "{print \$2}"
  $B $args --flags $((1<<20)) | awk -F" : " "{print \$2}"
  $B $args --flags $((1<<21)) | awk -F" : " "{print \$2}"
  RCV_CONVS_OCC=3 $B $args | awk -F" : " "{print \$2}"
done
