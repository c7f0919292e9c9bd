#!/bin/bash
# wave-per-16-chains kernel against the block-per-16-chains kernel (MI355_CLMH_WAVE=0) over chain counts
for n in 2048 4096 8192 16384 32768 65536; do
  for w in 1 0; do
    echo -n "chains $n wave=$w: "
    MI355_CLMH_WAVE=$w timeout -k 10 120 python tests/bench_clbeam.py mh 64 32 0.0 0.0 $n 300 1 0 | grep -o '"us_per_step": [0-9.]*'
  done
done
for n in 4096 16384 65536; do
  for w in 1 0; do
    echo -n "calib11 chains $n wave=$w: "
    MI355_CLMH_WAVE=$w timeout -k 10 120 python tests/bench_clbeam.py mh 64 32 0.05 0.0 $n 200 1 0 | grep -o '"us_per_step": [0-9.]*'
  done
done
