#!/bin/bash
# usage: tools/prof_cov65k.sh <lib> <tag>: kernel trace of the 65 536-chain Cl sampler with co-moments
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/cov65k_$2
mkdir -p $out
cd $R
export MI355MCMC_LIB=$1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tests/bench_clbeam.py mh 64 32 0.0 0.0 65536 100 1 1 > $out/log.txt 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -n 1)
head -n 4 $f | cut -c1-300
