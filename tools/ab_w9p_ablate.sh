# (build the variants first: tools/build_w9p_ablate.sh)
# timing ablations of wgrad9p.hip (WRONG results by construction): 1 no in-loop DMA, 2 no fragment reads, 4 no MFMA, 16 no slab stores
for lib in libfedfr_hip.so libfedfr_hip_ab1.so libfedfr_hip_ab2.so libfedfr_hip_ab3.so libfedfr_hip_ab4.so libfedfr_hip_ab7.so libfedfr_hip_ab16.so; do
  [ -f fedfr_amd/$lib ] || continue
  echo "== $lib"; FEDFR_HIP_LIB_NAME=$lib python tools/conv_bench.py 30 "s3_256x256@14" wpair wgrad9p=1 2>/dev/null | grep -E "wpair"
done
