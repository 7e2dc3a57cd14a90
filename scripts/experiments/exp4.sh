B="python scripts/bench_op.py"
echo "== 128->128 30x40: dma vs staged"
$B conv 32 30 40 128 128 --mode affine --stats fwd
RCV_NO_DMA=1 $B conv 32 30 40 128 128 --mode affine --stats fwd
$B conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc
RCV_NO_DMA=1 $B conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc
echo "== 64->64 60x80"
$B conv 32 60 80 64 64 --mode affine --stats fwd
RCV_NO_DMA=1 $B conv 32 60 80 64 64 --mode affine --stats fwd
echo "== 64->128 s2, tconv 128->64 (phase)"
$B conv 32 60 80 64 128 --stride 2 --mode affine --stats fwd
RCV_NO_DMA=1 $B conv 32 60 80 64 128 --stride 2 --mode affine --stats fwd
$B tconv 32 30 40 128 64 --mode affine --stats fwd
RCV_NO_DMA=1 $B tconv 32 30 40 128 64 --mode affine --stats fwd
