#!/bin/bash
# builds the timing-ablation variants of csrc/wgrad9p.hip that tools/ab_w9p_ablate.sh runs (run HERE before gpurun: the .so files travel with the
# snapshot; they are git-ignored).  W9P_ABLATE bits: 1 no in-loop DMA, 2 no fragment reads, 4 no MFMA, 16 no slab stores -- results are WRONG by construction
set -e
cd "$(dirname "$0")/../fedfr_amd/csrc"
make > /dev/null
for ab in ${@:-3 7 16}; do
  mkdir -p build_ab$ab && cp build/*.o build_ab$ab/
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=on -DW9P_ABLATE=$ab -c wgrad9p.hip -o build_ab$ab/wgrad9p.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfedfr_hip_ab$ab.so build_ab$ab/*.o
  rm -rf build_ab$ab
done
ls -la ../libfedfr_hip_ab*.so
