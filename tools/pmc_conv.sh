#!/bin/bash
# PMC passes (SQ: 8 counters per pass) on one conv shape.  usage: tools/pmc_conv.sh <shape-filter> <which> [options...]
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
SHAPE=${1:-s3_256x256@14}; WHICH=${2:-fwd}; shift 2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o p1 -- python3 tools/conv_bench.py 5 "$SHAPE" "$WHICH" "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT -o p2 -- python3 tools/conv_bench.py 5 "$SHAPE" "$WHICH" "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA --output-format csv -d $OUT -o p3 -- python3 tools/conv_bench.py 5 "$SHAPE" "$WHICH" "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "gpurun_out/pmc")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc/p*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "gemm" not in k and "conv3x3" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VMEM"): cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    n = max(cnt[(k, "SQ_WAVE_CYCLES")], 1)
    print("==", k, "dispatches", n)
    wc = d.get("SQ_WAVE_CYCLES", 1)
    for name in sorted(d):
        print("   %-28s %14.0f  per-dispatch %12.0f   /WAVE_CYCLES %.3f" % (name, d[name], d[name] / n, d[name] / wc))
PY
