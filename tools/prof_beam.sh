#!/bin/bash
# usage: tools/prof_beam.sh <tag>   (GPU box, repo root): rocprofv3 kernel stats and SQ counters of the beam
# likelihood kernel on config C5 (tests/bench_clbeam.py beam); summary printed by tools/summ_beam.py
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/beam_trace -- python3 tests/bench_clbeam.py beam > $out/beam_trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $out/beam_pmc -- python3 tests/bench_clbeam.py beam > $out/beam_pmc.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/beam_pmc2 -- python3 tests/bench_clbeam.py beam > $out/beam_pmc2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/beam_fetch -- python3 tests/bench_clbeam.py beam > $out/beam_fetch.log 2>&1
python3 tools/summ_beam.py $out
