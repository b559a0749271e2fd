#!/bin/bash
# Build a VARIANT of libdlka_hip.so for an in-process A/B (scripts/ab_stack_knobs.py, pseudo-knob _lib=alt_lib/libdlka_NAME.so):
#   scripts/build_variant.sh NAME "EXTRA HIPCC FLAGS" [file.hip ...]
# The listed sources (default: all) are compiled with the extra flags into deformablelka_amd/csrc/_build_NAME/, every other object is taken from
# the default build (deformablelka_amd/csrc/_build/, `make -C deformablelka_amd/csrc` first).  alt_lib/ is git-ignored.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); cd $R/deformablelka_amd/csrc
NAME=$1; FLAGS=$2; shift 2
FILES=${@:-$(ls *.hip)}
B=_build_$NAME; mkdir -p $B $R/alt_lib
CXX="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$R/include -I. -Wall -Wno-unused-function"
pids=()
for f in $FILES; do
  extra=""; [ $f = cl_dwconv_lds.hip ] && extra="-fno-slp-vectorize"
  ( $CXX $extra $FLAGS -c $f -o $B/${f%.hip}.o ) & pids+=($!)
  while [ $(jobs -r | wc -l) -ge ${JOBS:-6} ]; do sleep 1; done
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for f in *.hip; do o=${f%.hip}.o; if [ -f $B/$o ] && echo " $FILES " | grep -q " $f "; then OBJS="$OBJS $B/$o"; else OBJS="$OBJS _build/$o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/alt_lib/libdlka_$NAME.so $OBJS
ls -la $R/alt_lib/libdlka_$NAME.so
