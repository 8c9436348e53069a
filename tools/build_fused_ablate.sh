#!/bin/bash
# builds the timing-ablation variants of the fused BN-backward dgrad epilogue (conv_glds_impl.h, GLDS_FUSED_ABLATE bits: 1 no x loads,
# 2 no per-element pass, 4 no reduction -- WRONG results by construction); run HERE before gpurun, the .so files travel with the snapshot
set -e
cd "$(dirname "$0")/../fedfr_amd/csrc"
make > /dev/null
for ab in ${@:-1 2 4 7}; do
  mkdir -p build_ab$ab && cp build/*.o build_ab$ab/
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form=1 -DGLDS_FUSED_ABLATE=$ab -c conv_glds8_fused_w14.hip -o build_ab$ab/conv_glds8_fused_w14.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfedfr_hip_ab$ab.so build_ab$ab/*.o
  rm -rf build_ab$ab
done
ls -la ../libfedfr_hip_ab*.so
