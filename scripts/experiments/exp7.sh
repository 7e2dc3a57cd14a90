B="python scripts/bench_op.py"
for args in "conv 32 30 40 128 128 --mode affine --stats fwd" "conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc" "conv 32 60 80 64 64 --mode affine --stats fwd"; do
  echo "== $args"
  $B $args | awk -F" : " "{print \$2}"
  $B $args --flags $((1<<20)) | awk -F" : " "{print \$2}"
  $B $args --flags $((1<<21)) | awk -F" : " "{print \$2}"
  $B $args --flags $((3<<20)) | awk -F" : " "{print \$2}"
done
