#!/bin/bash
# usage: tools/prof_bench.sh <tag>   (run on the GPU box from the repo root)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --ess-steps 0 --corr-steps 0 --c3-slice-steps 0 > $out/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --ess-steps 0 --corr-steps 0 --c3-slice-steps 0 --adapt 100 > $out/pmc_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --ess-steps 0 --corr-steps 0 --c3-slice-steps 0 --adapt 100 > $out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --ess-steps 0 --corr-steps 0 --c3-slice-steps 0 --adapt 100 > $out/pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $out/pmc_sq2 -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --ess-steps 0 --corr-steps 0 --c3-slice-steps 0 --adapt 100 > $out/pmc_sq2.log 2>&1 || true
find $out -name "*.csv" | head -50
