#!/bin/bash
# Timing-ablation / A-B builds of ONE translation unit of the library (run HERE before gpurun: the .so files travel with the snapshot; they
# are git-ignored).  Usage: tools/build_ablate.sh <file.hip> <MACRO> <value> [<value> ...]   ->  fedfr_amd/libfedfr_hip_ab<value>.so
# (results of ablation builds are WRONG by construction; A-B builds select a variant).  Replaces the per-kernel build_*_ablate.sh scripts.
set -e
src=$1; macro=$2; shift 2
cd "$(dirname "$0")/../fedfr_amd/csrc"
make > /dev/null
obj=${src%.hip}.o
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=on"
case $src in conv_c64p.hip) ;; wgrad9p.hip) flags="$flags -mllvm -pragma-unroll-threshold=200000" ;; *) flags="$flags -mllvm -amdgpu-mfma-vgpr-form=1" ;; esac
for ab in "$@"; do
  mkdir -p build_ab$ab && cp build/*.o build_ab$ab/
  hipcc $flags -D$macro=$ab -c $src -o build_ab$ab/$obj
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfedfr_hip_ab$ab.so build_ab$ab/*.o
  rm -rf build_ab$ab
done
ls -la ../libfedfr_hip_ab*.so
