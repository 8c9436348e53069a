#!/bin/bash
# SQ counters of EVERY kernel of the training step (single-stream pass: under --pmc dispatches are serialised anyway), summed per kernel name:
# wave cycles, wait shares, MFMA busy, LDS conflicts, waves.  Three --pmc passes of 8 SQ counters each (--kernel-trace only beside them).
# usage: tools/pmc_sq_step.sh <tag> [name-regex]        -> gpurun_out/<tag>_pmc_sq_step.txt   (BENCH_ARGS="--arch iresnet50" ...)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r06}; RX=${2:-.}
OUT=$R/gpurun_out/pmc_sq_step; rm -rf $OUT; mkdir -p $OUT
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  FEDFR_DUAL_STREAM=0 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT -o p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile $BENCH_ARGS > $OUT/run$i.json 2> $OUT/run$i.err || { echo "pass $i failed"; tail -5 $OUT/run$i.err; }
  echo "pass $i done"
done
cd $R
{ python3 tools/source_stamp.py; OUT=$OUT TAG=$TAG RX="$RX" python3 - <<'PY'
import csv, glob, collections, os, re
out = os.environ["OUT"]; rx = re.compile(os.environ["RX"])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
short = lambda n: re.sub(r"^void ", "", n)[:110]
for f in glob.glob(out + "/**/p*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not rx.search(k): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for f in glob.glob(out + "/**/p1_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# tools/pmc_sq_step.sh %s %s : SQ counters per kernel over a 3-step single-stream run of bench.py (per-dispatch averages)" % (os.environ["TAG"], os.environ["RX"]))
print("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES per SQ (summed over SEs/XCDs);")
print("# mfma_busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (4 x SQ_WAVE_CYCLES / SQ_WAVES): the share of a wave's lifetime during which its SIMD's matrix")
print("# pipe is busy (bench.py's definition); waves_per_simd = SQ_WAVES / 1024; durations under --pmc are inflated: quote durations from --stats runs")
rows = []
for k, d in agg.items():
    n = max(cnt[(k, "SQ_WAVE_CYCLES")], 1)
    tot_us = sum(dur.get(k, [0]))
    rows.append((tot_us, k, d, n))
for tot_us, k, d, n in sorted(rows, reverse=True):
    per = lambda name: d.get(name, 0.0) / max(cnt[(k, name)], 1)
    wc = max(per("SQ_WAVE_CYCLES"), 1.0); busy = max(per("SQ_BUSY_CYCLES"), 1.0)
    print("== %s\n   dispatches %d  avg_us(pmc pass 1) %.1f  waves/dispatch %.0f" % (k, n, tot_us / max(len(dur.get(k, [0])), 1), per("SQ_WAVES")))
    print("   share of wave cycles: wait_any %.3f  wait_inst_any %.3f  active_any %.3f (valu %.3f lds %.3f vmem %.3f sca %.3f)  lds_issue_wait %.3f" % (
        per("SQ_WAIT_ANY") / wc, per("SQ_WAIT_INST_ANY") / wc, per("SQ_ACTIVE_INST_ANY") / wc, per("SQ_ACTIVE_INST_VALU") / wc,
        per("SQ_ACTIVE_INST_LDS") / wc, per("SQ_ACTIVE_INST_VMEM") / wc, per("SQ_ACTIVE_INST_SCA") / wc, per("SQ_WAIT_INST_LDS") / wc))
    print("   mfma_busy %.3f  waves_per_simd %.2f   insts/dispatch: valu %.0f mfma %.0f lds %.0f vmem_rd %.0f vmem_wr %.0f salu %.0f smem %.0f   lds bank conflict / idx active %.3f (%.0f / %.0f)" % (
        (per("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0) / max(4.0 * per("SQ_WAVE_CYCLES") / max(per("SQ_WAVES"), 1.0), 1.0), per("SQ_WAVES") / 1024.0, per("SQ_INSTS_VALU"), per("SQ_INSTS_MFMA"), per("SQ_INSTS_LDS"), per("SQ_INSTS_VMEM_RD"),
        per("SQ_INSTS_VMEM_WR"), per("SQ_INSTS_SALU"), per("SQ_INSTS_SMEM"), per("SQ_LDS_BANK_CONFLICT") / max(per("SQ_LDS_IDX_ACTIVE"), 1.0),
        per("SQ_LDS_BANK_CONFLICT"), per("SQ_LDS_IDX_ACTIVE")))
PY
} > $R/gpurun_out/${TAG}_pmc_sq_step.txt
tail -3 $R/gpurun_out/${TAG}_pmc_sq_step.txt
