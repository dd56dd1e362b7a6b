#!/bin/bash
# Build libfod_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
mkdir -p lib build
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-comment"
pids=()
# attention: MFMA results stay in VGPRs (no AGPR round trip): the softmax reads every accumulator element, and
# v_accvgpr_read was a fifth of the loop's vector instructions; the kernels also fit one more wave per SIMD.
declare -A EXTRA=([attention]="-mllvm -amdgpu-mfma-vgpr-form=1" [attention_fp8]="-mllvm -amdgpu-mfma-vgpr-form=1" [gemm_nt_big]="-Wno-inline-asm" [gemm_tn_big]="-Wno-inline-asm" [bottleneck_fused]="-Wno-inline-asm")
for f in gemm_nt gemm_nt_big stem_pool linear_norm gemm_tn gemm_tn_big attention attention_fp8 bottleneck_fused elementwise loss optim lap_dev; do
  hipcc $FLAGS ${EXTRA[$f]} -c csrc/$f.hip -o build/$f.o &
  pids+=($!)
done
hipcc $FLAGS -c csrc/lap.cpp -o build/lap.o &
pids+=($!)
hipcc $FLAGS -c csrc/api.cpp -o build/api.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libfod_hip.so build/*.o -lpthread
# fast-call CPython wrappers of the same entry points (generated from the binding's signature table)
python3 ../tools/gen_fastcall.py > /dev/null
gcc -O2 -shared -fPIC -I"$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')" \
    build/fastcall.c -o lib/_fodfast.so -Llib -lfod_hip -Wl,-rpath,'$ORIGIN'
echo "built $(pwd)/lib/libfod_hip.so"
