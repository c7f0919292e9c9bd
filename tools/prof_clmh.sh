#!/bin/bash
# usage: tools/prof_clmh.sh <tag> [bench_clbeam.py mh arguments]   (on the GPU box, from the repo root)
# rocprofv3 kernel stats and SQ counters of the fused binnedCl Metropolis-Hastings kernel
set -e
tag=$1; shift
args=${@:-64 32 0.0 0.0 4096 300 1 0}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tests/bench_clbeam.py mh $args > $out/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc1 -- python3 tests/bench_clbeam.py mh $args > $out/pmc1.log 2>&1 || true
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -- python3 tests/bench_clbeam.py mh $args > $out/pmc2.log 2>&1 || true
python3 - "$out" <<'PY'
import csv, glob, sys
root = sys.argv[1]
for tag in ("pmc1", "pmc2"):
    fs = glob.glob(root + "/%s/*/*counter_collection.csv" % tag)
    if not fs:
        print(tag, "no counter file"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "k_cl_mh" in r["Kernel_Name"]]
    if not rows:
        print(tag, "kernel not found"); continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    acc = {}
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    print(tag, rows[0]["Kernel_Name"][:60], {k: "%.5g" % v for k, v in sorted(acc.items())})
fs = glob.glob(root + "/trace/*/*kernel_stats.csv")
if fs:
    for r in csv.DictReader(open(fs[0])):
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
