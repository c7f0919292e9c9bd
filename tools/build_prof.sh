#!/bin/bash
# phase-timing build of the library (k_cl_mh_run prints cycles per phase): build/prof/libmi355mcmc_prof.so
# use: MI355MCMC_LIB=$PWD/build/prof/libmi355mcmc_prof.so python tests/bench_clbeam.py mh ...
set -e
cd "$(dirname "$0")/../mcmc_amd/csrc"
mkdir -p ../../build/prof
for f in api cl_api beam_api gmh_api; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off \
    -Wno-unused-function -I../../include -DMI355_CLMH_PROF $EXTRA -c $f.hip -o ../../build/prof/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared ../../build/prof/*.o -o ../../build/prof/libmi355mcmc_prof.so
