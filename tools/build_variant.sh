#!/bin/bash
# Builds an experimental copy of the library with extra compile-time options:
#   tools/build_variant.sh <name> "<extra hipcc flags>"   -> spectroplot-js_amd/lib/variants/<name>.so  (SP_LIB_VARIANT=<name> loads it)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd $(dirname $0)/.. && pwd)/spectroplot-js_amd
B=/tmp/sp_variant_$NAME; mkdir -p $B $ROOT/lib/variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -I$ROOT/../include $EXTRA"
/opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 -c -o $B/sp_api.o $ROOT/csrc/sp_api.hip
g++ $FLAGS -c -o $B/sp_host.o $ROOT/csrc/sp_host.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/lib/variants/$NAME.so $B/sp_api.o $B/sp_host.o
echo built $NAME
