#!/bin/bash
# usage: tools/bench_clmh_cov.sh   (GPU box, repo root): us per step of the fused Cl samplers with and
# without the running co-moments, block kernel (MI355_CLMH_WAVE=0) against wave kernel (=1)
for a in "64 32 0.0 0.0 4096" "64 32 0.0 0.0 16384" "64 32 0.05 0.0 4096" "64 32 0.05 0.0 16384"; do
  for cov in 0 1; do
    for w in 0 1; do
      r=$(MI355_CLMH_WAVE=$w timeout -k 10 120 python tests/bench_clbeam.py mh $a 100 1 $cov 2>/dev/null | grep -o "\"us_per_step\": [0-9.]*")
      echo "$a cov=$cov wave=$w $r"
    done
  done
done
