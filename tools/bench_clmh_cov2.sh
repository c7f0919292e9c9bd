#!/bin/bash
# usage: tools/bench_clmh_cov2.sh (GPU box): block against wave kernel with co-moments over chain counts
for n in 8192 12288 16384 24576 32768; do
  for w in 0 1; do
    r=$(MI355_CLMH_WAVE=$w timeout -k 10 120 python tests/bench_clbeam.py mh 64 32 0.0 0.0 $n 100 1 1 2>/dev/null | grep -o "\"us_per_step\": [0-9.]*")
    echo "chains=$n cov=1 wave=$w $r"
  done
done
for n in 2048 4096; do
  for w in 0 1; do
    r=$(MI355_CLMH_WAVE=$w timeout -k 10 120 python tests/bench_clbeam.py mh 64 32 0.05 0.02 $n 100 1 1 2>/dev/null | grep -o "\"workload\": \"[^\"]*\"\|\"us_per_step\": [0-9.]*" | paste - -)
    echo "121-grid chains=$n cov=1 wave=$w $r"
  done
done
