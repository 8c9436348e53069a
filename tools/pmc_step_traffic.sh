#!/bin/bash
# HBM traffic of a WHOLE training step (the default bench step): separate --pmc passes for FETCH_SIZE and WRITE_SIZE (they cannot share a
# pass; gfx950 FETCH_SIZE x2 correction as MI355X_MICROARCH.md prescribes), per-kernel-family sums per step.  usage: tools/pmc_step_traffic.sh <tag>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r03}; OUT=$R/gpurun_out/pmc_step; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o f -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $OUT/f.json 2> $OUT/f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o w -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $OUT/w.json 2> $OUT/w.err
cd $R
python3 tools/source_stamp.py
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
fam = lambda n: ("BatchNorm passes" if any(k in n for k in ("bn_apply", "bn_bwd", "bn_finalize", "colsum_stage", "bn1d")) else
                 "conv / dgrad (MFMA)" if any(k in n for k in ("conv3x3", "gemm_nt", "stem_fwd")) else
                 "weight gradients + slab reductions" if any(k in n for k in ("wgrad", "gemm_tn", "reduce_slabs", "stem_wgrad")) else
                 "SGD / shadows" if any(k in n for k in ("sgd_kernel", "shadow", "cast")) else "other")
tot = collections.defaultdict(lambda: [0.0, 0.0, 0]); steps = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
for f in glob.glob("gpurun_out/pmc_step/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n, c, v = r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
        if "stem_fwd_kernel" in n: steps[c] += 1
        t = tot[fam(n)]
        if c == "FETCH_SIZE": t[0] += 2 * v * 1024; t[2] += 1
        elif c == "WRITE_SIZE": t[1] += v * 1024
ns = max(steps["FETCH_SIZE"], 1), max(steps["WRITE_SIZE"], 1)
print("# HBM traffic per training step (iresnet100, B = 128, default execution), rocprofv3 --pmc, %d / %d steps in the FETCH / WRITE pass" % ns)
print("# (all steps of the run incl. warm-up and the concurrent-clients leg are averaged; FETCH_SIZE x2-corrected)")
gf = gw = 0.0
for k, (f, w, l) in sorted(tot.items(), key=lambda kv: -(kv[1][0] / ns[0] + kv[1][1] / ns[1])):
    print("%-40s launches/step %5.0f   fetched %7.2f GB   written %7.2f GB" % (k, l / ns[0], f / ns[0] / 1e9, w / ns[1] / 1e9))
    gf += f / ns[0]; gw += w / ns[1]
print("%-40s                        fetched %7.2f GB   written %7.2f GB   total %.2f GB" % ("whole step", gf / 1e9, gw / 1e9, (gf + gw) / 1e9))
PY
