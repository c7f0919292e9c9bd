#!/bin/bash
# A/B timing of several builds of the library on the SAME box (boxes differ by ~1 %): abuild/lib<V>.so
# usage (GPU box, repo root): bash tools/ab_bench.sh <rounds> A B C ...
n=$1; shift
for i in $(seq $n); do
  for v in "$@"; do
    echo -n "$v: "
    MI355MCMC_LIB=$PWD/abuild/lib$v.so timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ess-steps 0 --corr-steps 0 --incremental-steps 0 2>&1 | grep "kernel ms" || exit 1
  done
done
