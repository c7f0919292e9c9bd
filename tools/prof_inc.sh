#!/bin/bash
# usage: tools/prof_inc.sh <tag>   (run on the GPU box from the repo root)
# SQ counters of the opt-in incremental sampler: bench.py's last k_mh_run* launch is k_mh_run_inc.
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --ess-steps 0 --adapt 100 --incremental-steps 1 > $out/pmc_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $out/pmc_sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --ess-steps 0 --adapt 100 --incremental-steps 1 > $out/pmc_sq2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $out/pmc_sq3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --ess-steps 0 --adapt 100 --incremental-steps 1 > $out/pmc_sq3.log 2>&1 || true
python3 tools/parse_pmc.py $out > $out/summary.txt 2>&1 || true
cat $out/summary.txt
