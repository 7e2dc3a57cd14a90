#!/bin/bash
# usage (build container): scripts/build_variant.sh NAME [MAKEVAR=1 ...]  -> robocupvision_amd/librcv_NAME.so from a scratch copy of csrc
# (e.g. EXPERIMENTS=1: the RCV_* environment knobs compiled in).  Used with RCV_LIBRARY=... for A/B runs; never shipped.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/rcv_build_$NAME
mkdir -p $B/robocupvision_amd/csrc $B/include
for f in $ROOT/robocupvision_amd/csrc/*.hip $ROOT/robocupvision_amd/csrc/*.h $ROOT/robocupvision_amd/csrc/Makefile; do cmp -s $f $B/robocupvision_amd/csrc/$(basename $f) || cp $f $B/robocupvision_amd/csrc/; done
for f in $ROOT/include/*.h; do cmp -s $f $B/include/$(basename $f) || cp $f $B/include/; done
make -C $B/robocupvision_amd/csrc -j8 "$@" > $B/build.log 2>&1 || { tail -30 $B/build.log; exit 1; }
cp $B/robocupvision_amd/librcv.so $ROOT/robocupvision_amd/librcv_$NAME.so
echo "built robocupvision_amd/librcv_$NAME.so"
