#!/bin/bash
# A/B timing of the extended fused SED sampler (correlated proposals) over several builds of the library on
# the SAME box: abuild/lib<V>.so.  usage (GPU box, repo root): bash tools/ab_ext.sh <rounds> A B C ...
n=$1; shift
for i in $(seq $n); do
  for v in "$@"; do
    echo "== $v"
    MI355MCMC_LIB=$PWD/abuild/lib$v.so timeout -k 10 200 python tools/time_ext.py 2>&1 | grep "correlated" || exit 1
  done
done
