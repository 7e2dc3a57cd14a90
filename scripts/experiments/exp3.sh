B="python scripts/bench_op.py"
echo "== wgrad L0 (3->8, nchw G, grad_enc P)"
$B wgrad 32 480 640 3 8 --mode nchw --mode2 grad_enc
RCV_NO_FOLD=1 $B wgrad 32 480 640 3 8 --mode nchw --mode2 grad_enc
echo "== wgrad L1a (8->16 s2)"
$B wgrad 32 480 640 8 16 --stride 2 --mode affine --mode2 grad_enc
echo "== wgrad Up3 (G = d_up 8ch 480x640 grad_dec, P = in 16ch)"
$B wgrad 32 480 640 8 16 --stride 2 --mode grad_dec --mode2 plain
echo "== wgrad 16->16 240x320"
$B wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc
echo "== wgrad 128->128 30x40"
$B wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc
RCV_WGRAD_OCC=1 $B wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc
echo "== wgrad 64->64 60x80, 32->32 120x160, 64->128 s2"
$B wgrad 32 60 80 64 64 --mode affine --mode2 grad_enc
$B wgrad 32 120 160 32 32 --mode affine --mode2 grad_enc
$B wgrad 32 60 80 64 128 --stride 2 --mode affine --mode2 grad_enc
