#!/bin/bash
# Build libb2s_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libb2s_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
# --clean: recompile every source (what __graft_entry__.build() runs); default: only what changed
if [ "$1" = "--clean" ]; then rm -rf obj; fi
mkdir -p obj
pids=()
for f in gemm gemm_glds gemm_glds256 gemm_skinny attention rowops engine capi_ops decode decode_fused enc_fused; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer obj/$f.o)" ] || [ ../../include/b2s_hip.h -nt obj/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC obj/gemm.o obj/gemm_glds.o obj/gemm_glds256.o obj/gemm_skinny.o obj/attention.o obj/rowops.o obj/engine.o obj/capi_ops.o obj/decode.o obj/decode_fused.o obj/enc_fused.o -o $OUT
echo "built $OUT"
