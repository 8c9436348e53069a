#!/bin/bash
# HBM traffic of the conv kernels (separate --pmc passes as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
# cannot share a pass).  usage: tools/pmc_traffic.sh <shape-filter> <which>
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
SHAPE=${1:-s3_256x256@14}; WHICH=${2:-fwd}
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o f -- python3 tools/conv_bench.py 5 "$SHAPE" "$WHICH" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o w -- python3 tools/conv_bench.py 5 "$SHAPE" "$WHICH" > /dev/null 2>&1
python3 tools/source_stamp.py
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_traffic/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:64]
        if "gemm" in k or "conv3x3" in k or "reduce_slabs" in k or "glds" in k or "wgrad9" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    f = d.get("FETCH_SIZE", [0]); w = d.get("WRITE_SIZE", [0])
    fk, wk = sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)
    # units: KiB; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM) -> corrected value = 2x
    print("%-66s launches %3d  FETCH_SIZE %9.0f KiB (x2 corrected: %7.1f MB)  WRITE_SIZE %9.0f KiB (%7.1f MB)" % (k, len(f), fk, 2 * fk * 1024 / 1e6, wk, wk * 1024 / 1e6))
PY
