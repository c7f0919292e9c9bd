#!/bin/bash
# usage: tools/prof_clbeam.sh <tag>   (on the GPU box, from the repo root): rocprofv3 kernel stats and
# counters of the Cl (C4) and beam (C5) likelihood kernels
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cl_trace -- python3 tests/bench_clbeam.py cl > $out/cl_trace.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/beam_trace -- python3 tests/bench_clbeam.py beam > $out/beam_trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/cl_pmc -- python3 tests/bench_clbeam.py cl > $out/cl_pmc.log 2>&1 || true
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cl_fetch -- python3 tests/bench_clbeam.py cl > $out/cl_fetch.log 2>&1 || true
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $out/beam_pmc -- python3 tests/bench_clbeam.py beam > $out/beam_pmc.log 2>&1 || true
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/beam_fetch -- python3 tests/bench_clbeam.py beam > $out/beam_fetch.log 2>&1 || true
find $out -name "*.csv" | head -40
