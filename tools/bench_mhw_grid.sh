#!/bin/bash
# fused binnedCl sampler with a marginalisation grid against the launch-per-step engine, over chain counts
# (where the wave kernel stops paying: cl_mhw_plan's policy; GPU box, repo root)
for a in "64 32 0.05 0.0 16384 100" "64 32 0.05 0.0 65536 100" "64 32 0.05 0.02 8192 50" "64 32 0.05 0.02 2048 50"; do
  for f in 1 0; do
    timeout -k 10 120 python tests/bench_clbeam.py mh $a $f 0 | cut -c1-215 || exit 1
  done
done
