#!/bin/bash
# Builds an experimental copy of the library with extra compile-time options:
#   tools/build_variant.sh <name> "<extra -D flags>" ["<hipcc-only flags, e.g. -mllvm ...>"]
#     -> spectroplot-js_amd/lib/variants/<name>.so  (SP_EXPERIMENT_KNOBS=1 SP_LIB_VARIANT=<name> loads it)
set -e
NAME=$1; EXTRA=$2; HIPEXTRA=$3
ROOT=$(cd $(dirname $0)/.. && pwd)/spectroplot-js_amd
make -s -j8 -C $ROOT BUILD=/tmp/sp_variant_$NAME OUT=lib/variants/$NAME.so EXTRA="$EXTRA" HIPEXTRA="$HIPEXTRA"
echo built $NAME
