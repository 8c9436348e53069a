#!/bin/bash
# SQ counters (MFMA busy, LDS waits / conflicts, instruction mix) on the MFMA kernels the step actually runs, one shape at a time through
# tools/conv_bench.py.  Three --pmc passes of 8 SQ counters each (never combined with a trace domain other than --kernel-trace).
# usage: tools/pmc_sq.sh <tag> <shape-filter> <which> [option=value ...]      -> gpurun_out/pmc_sq/<tag>.txt
#   which: fwd | dgrad | fdgrad | wgrad | wpair (see tools/conv_bench.py)
export TMPDIR=/tmp
R=$PWD
TAG=${1:?tag}; SHAPE=${2:?shape}; WHICH=${3:?which}; shift 3
OUT=$R/gpurun_out/pmc_sq/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT -o p$i -- python3 $R/tools/conv_bench.py 5 "$SHAPE" "$WHICH" "$@" > $OUT/run$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/run$i.log; }
done
cd $R
{ python3 tools/source_stamp.py; OUT=$OUT TAG=$TAG SHAPE="$SHAPE" WHICH="$WHICH" OPTS="$*" python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["OUT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob(out + "/**/p*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:96]
        if not any(s in k for s in ("gemm", "conv3x3", "wgrad9", "reduce_slabs")): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for f in glob.glob(out + "/**/p1_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:96]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# tools/pmc_sq.sh %s %s %s %s" % (os.environ["TAG"], os.environ["SHAPE"], os.environ["WHICH"], os.environ["OPTS"]))
print("# counters are summed over all SEs / XCDs per dispatch; MfmaBusy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (both per-SQ sums, gfx94x formula),")
print("# launch durations under --pmc are inflated (counter collection serialises dispatches): quote durations from the --stats runs")
for k, d in sorted(agg.items()):
    n = max(cnt[(k, "SQ_WAVE_CYCLES")], 1)
    print("==", k, " dispatches", n, " avg_us(pmc pass 1) %.1f" % (sum(dur.get(k, [0])) / max(len(dur.get(k, [0])), 1)))
    wc = d.get("SQ_WAVE_CYCLES", 1.0); busy = d.get("SQ_BUSY_CYCLES", 1.0)
    for name in sorted(d):
        m = max(cnt[(k, name)], 1)
        print("   %-28s per-dispatch %14.0f   /WAVE_CYCLES %.4f   /BUSY_CYCLES %.4f" % (name, d[name] / m, d[name] / wc * n / m, d[name] / busy * n / m))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        print("   -> MfmaBusy (MFMA_BUSY / BUSY_CYCLES) %.3f ; LDS-wait share of wave cycles %.3f ; any-wait share %.3f" %
              (d["SQ_VALU_MFMA_BUSY_CYCLES"] / busy, d.get("SQ_WAIT_INST_LDS", 0) / max(cnt[(k, "SQ_WAIT_INST_LDS")], 1) / (wc / n), d.get("SQ_WAIT_INST_ANY", 0) / wc))
PY
} | tee $R/gpurun_out/pmc_sq/$TAG.txt
