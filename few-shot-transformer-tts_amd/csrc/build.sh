#!/bin/bash
# Build libb2s_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   build.sh            only what changed
#   build.sh --clean    recompile every source (what __graft_entry__.build() runs)
#   build.sh --lab      measurement build with -DB2S_LAB (B2S_LAB_* environment switches that change results: skip the encoder, drop the
#                       exchange ordering) -> ../../tools/bin/libb2s_hip_lab.so, used through B2S_LIB_PATH; never the product library
set -e
cd "$(dirname "$0")"
OUT=../libb2s_hip.so
OBJ=obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
if [ "$1" = "--lab" ]; then OBJ=obj_lab; mkdir -p ../../tools/bin; OUT=../../tools/bin/libb2s_hip_lab.so; FLAGS="$FLAGS -DB2S_LAB"; fi
if [ "$1" = "--clean" ]; then rm -rf $OBJ; fi
mkdir -p $OBJ
SRCS="gemm gemm_glds gemm_glds256 gemm_skinny attention attention32 rowops engine capi_ops decode decode_fused enc_fused"
pids=()
for f in $SRCS; do
  if [ ! -f $OBJ/$f.o ] || [ $f.hip -nt $OBJ/$f.o ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer $OBJ/$f.o)" ] || [ ../../include/b2s_hip.h -nt $OBJ/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o $OBJ/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
objs=""; for f in $SRCS; do objs="$objs $OBJ/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT
echo "built $OUT"
