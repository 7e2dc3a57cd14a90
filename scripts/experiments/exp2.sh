B="python scripts/bench_op.py"
echo "== 16->16 240x320 fwd: narrow vs general"
$B conv 32 240 320 16 16 --mode affine --stats fwd
RCV_NO_NARROW=1 $B conv 32 240 320 16 16 --mode affine --stats fwd
RCV_CONVS_OCC=3 $B conv 32 240 320 16 16 --mode affine --stats fwd
RCV_CONVS_TILE=2,160 $B conv 32 240 320 16 16 --mode affine --stats fwd
RCV_CONVS_TILE=4,80 $B conv 32 240 320 16 16 --mode affine --stats fwd
echo "== 16->16 dgrad (grad_enc, resid, bwd_enc)"
$B conv 32 240 320 16 16 --mode grad_enc --stats bwd_enc --resid 1
RCV_NO_NARROW=1 $B conv 32 240 320 16 16 --mode grad_enc --stats bwd_enc --resid 1
echo "== L0 fwd"
$B conv 32 480 640 3 8 --mode nchw --stats fwd
echo "== L1a fwd 8->16 s2"
$B conv 32 480 640 8 16 --stride 2 --mode affine --stats fwd
RCV_NO_NARROW=1 $B conv 32 480 640 8 16 --stride 2 --mode affine --stats fwd
echo "== tconv merged 16->8 (Up3 fwd / L1a dgrad)"
$B tconv 32 240 320 16 8 --mode plain --stats fwd --merged 1
$B tconv 32 240 320 16 8 --mode grad_enc --stats bwd_enc --resid 1 --merged 1
RCV_NO_NARROW=1 $B tconv 32 240 320 16 8 --mode grad_enc --stats bwd_enc --resid 1 --merged 1
echo "== Up3 dgrad: gather s2 8->16"
$B conv 32 480 640 8 16 --stride 2 --mode grad_dec --stats bwd_dec
echo "== tconv merged 32->16 (Up2 fwd), general merged kernel"
$B tconv 32 120 160 32 16 --mode plain --stats fwd --merged 1
$B tconv 32 120 160 32 16 --mode plain --stats fwd --merged 0
