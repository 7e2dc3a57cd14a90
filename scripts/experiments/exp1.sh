set -e
B="python scripts/bench_op.py"
echo "== 128->128 30x40 fwd (affine, stats fwd): full / nostage / nomfma"
$B conv 32 30 40 128 128 --mode affine --stats fwd
$B conv 32 30 40 128 128 --mode affine --stats fwd --flags $((1<<20))
$B conv 32 30 40 128 128 --mode affine --stats fwd --flags $((1<<21))
echo "== tile variants for 128ch"
RCV_CONV_TILE=0,4,20 $B conv 32 30 40 128 128 --mode affine --stats fwd
RCV_CONV_TILE=4,2,40 $B conv 32 30 40 128 128 --mode affine --stats fwd
RCV_CONV_TILE=1,4,40 $B conv 32 30 40 128 128 --mode affine --stats fwd
echo "== 16->16 240x320 fwd: full / nostage / nomfma"
$B conv 32 240 320 16 16 --mode affine --stats fwd
$B conv 32 240 320 16 16 --mode affine --stats fwd --flags $((1<<20))
$B conv 32 240 320 16 16 --mode affine --stats fwd --flags $((1<<21))
RCV_CONV_TILE=3,2,160 $B conv 32 240 320 16 16 --mode affine --stats fwd
RCV_CONV_TILE=3,4,80 $B conv 32 240 320 16 16 --mode affine --stats fwd
RCV_CONV_TILE=3,10,32 $B conv 32 240 320 16 16 --mode affine --stats fwd
RCV_CONV_TILE=6,2,80 $B conv 32 240 320 16 16 --mode affine --stats fwd
RCV_CONV_TILE=6,5,32 $B conv 32 240 320 16 16 --mode affine --stats fwd
echo "== tconv 16->8 240x320 -> 480x640: phase vs merged"
$B tconv 32 240 320 16 8 --mode plain --stats fwd --merged 0
$B tconv 32 240 320 16 8 --mode plain --stats fwd --merged 1
RCV_CONV_TILE=2,2,160 $B tconv 32 240 320 16 8 --mode plain --stats fwd --merged 1
RCV_CONV_TILE=5,2,80 $B tconv 32 240 320 16 8 --mode plain --stats fwd --merged 1
RCV_CONV_TILE=5,5,32 $B tconv 32 240 320 16 8 --mode plain --stats fwd --merged 1
echo "== L0 fwd 3->8 480x640"
$B conv 32 480 640 3 8 --mode nchw --stats fwd
RCV_CONV_TILE=3,2,160 $B conv 32 480 640 3 8 --mode nchw --stats fwd
RCV_CONV_TILE=6,5,32 $B conv 32 480 640 3 8 --mode nchw --stats fwd
