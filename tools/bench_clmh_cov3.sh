#!/bin/bash
# usage: tools/bench_clmh_cov3.sh (GPU box): the 11 x 11 grid, fused wave kernel (MI355_CLMH_WAVE=1) against
# the step-wise engine (=0), with and without co-moments
for n in 512 1024 2048 4096; do
  for cov in 0 1; do
    for w in 0 1; do
      r=$(MI355_CLMH_WAVE=$w timeout -k 10 120 python tests/bench_clbeam.py mh 64 32 0.05 0.02 $n 60 1 $cov 2>/dev/null | grep -o "\"workload\": \"[^\"]*\"\|\"us_per_step\": [0-9.]*" | paste - -)
      echo "121-grid chains=$n cov=$cov wave=$w $r"
    done
  done
done
