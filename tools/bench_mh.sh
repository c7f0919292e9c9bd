set -e
for f in 1 0; do timeout -k 10 120 python tests/bench_clbeam.py mh 64 32 0.0 0.0 4096 300 $f 0; done
for f in 1 0; do timeout -k 10 120 python tests/bench_clbeam.py mh 64 32 0.0 0.0 4096 300 $f 1; done
for f in 1 0; do timeout -k 10 120 python tests/bench_clbeam.py mh 256 64 0.0 0.0 4096 300 $f 0; done
for f in 1 0; do timeout -k 10 120 python tests/bench_clbeam.py mh 64 32 0.05 0.0 4096 300 $f 0; done
timeout -k 10 120 python tests/bench_clbeam.py mh 64 32 0.0 0.0 65536 300 1 0
timeout -k 10 120 python tests/bench_clbeam.py cl
