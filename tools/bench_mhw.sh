#!/bin/bash
# fused binnedCl sampler: the shapes VERDICT r2 item 2 names (GPU box, repo root)
for a in "64 32 0.0 0.0 4096 300 1 0" "64 32 0.0 0.0 65536 300 1 0" "64 32 0.05 0.0 4096 300 1 0" "64 32 0.05 0.0 4096 300 0 0" "64 32 0.0 0.0 4096 300 1 1" "64 32 0.05 0.02 4096 100 1 0" "64 32 0.05 0.02 4096 100 0 0" "128 32 0.0 0.0 4096 300 1 0" "256 64 0.0 0.0 4096 300 1 0"; do
  timeout -k 10 120 python tests/bench_clbeam.py mh $a | cut -c1-215
done
