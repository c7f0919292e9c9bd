#!/bin/bash
# PMC counters + kernel trace of the extended fused SED sampler (tools/time_ext.py, C2 shape).
# usage (GPU box, repo root): bash tools/prof_ext.sh <tag> [lib]
set -e
tag=$1
[ -n "$2" ] && export MI355MCMC_LIB=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_ext_$tag
mkdir -p $out
cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/time_ext.py > $out/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -- python3 tools/time_ext.py > $out/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VMEM --output-format csv -d $out/pmc_sq2 -- python3 tools/time_ext.py > $out/pmc_sq2.log 2>&1
python3 tools/summ_ext.py $out
