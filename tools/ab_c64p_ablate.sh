# timing ablations of conv_c64p.hip (WRONG results by construction): per-variant microbench of the 112x112 / 56x56 64-channel layers
for lib in libfedfr_hip.so libfedfr_hip_ab1.so libfedfr_hip_ab2.so libfedfr_hip_ab4.so libfedfr_hip_ab7.so; do
  echo "== $lib"; FEDFR_HIP_LIB_NAME=$lib python tools/conv_bench.py 20 "s1_64x64@" fwd,dgrad 2>/dev/null | grep -E "s1_64x64@112 |s1_64x64@56 "
done
