#!/bin/bash
# Builds an experimental copy of the library with extra compile-time options:
#   tools/build_variant.sh <name> "<extra -D flags>" ["<hipcc-only flags, e.g. -mllvm ...>"]
#     -> spectroplot-js_amd/lib/variants/<name>.so  (SP_EXPERIMENT_KNOBS=1 SP_LIB_VARIANT=<name> loads it)
# Flags that mention SP_STAMPS or SP_ABL_ build from a copy of csrc/ with tools/experiments/frames_instrumentation.patch applied
# (per-wave clock stamps for tools/stamps.py; cost-attribution switches that remove a piece of the kernel - results invalid).
set -e
NAME=$1; EXTRA=$2; HIPEXTRA=$3
REPO=$(cd $(dirname $0)/.. && pwd)
ROOT=$REPO/spectroplot-js_amd
SRC=$ROOT
case "$EXTRA" in
  *SP_STAMPS*|*SP_ABL_*|*SP_PROTO_*|*SP_EXPERIMENT_KNOBS*)
    SRC=/tmp/sp_src_$NAME
    rm -rf $SRC && mkdir -p $SRC/spectroplot-js_amd $SRC/include
    cp -r $ROOT/csrc $ROOT/Makefile $SRC/spectroplot-js_amd/ && cp $REPO/include/*.h $SRC/include/
    (cd $SRC && patch -s -p1 < $REPO/tools/experiments/frames_instrumentation.patch)
    SRC=$SRC/spectroplot-js_amd ;;
esac
mkdir -p $ROOT/lib/variants
make -s -j8 -C $SRC BUILD=/tmp/sp_variant_$NAME OUT=$ROOT/lib/variants/$NAME.so EXTRA="$EXTRA" HIPEXTRA="$HIPEXTRA"
echo built $NAME
