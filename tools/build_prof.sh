#!/bin/bash
# phase-timing build of the library (k_cl_mh_run prints cycles per phase): abuild/prof/libmi355mcmc_prof.so,
# through the library's own Makefile (same per-file flags as the shipped build).
# use: MI355MCMC_LIB=$PWD/abuild/prof/libmi355mcmc_prof.so python tests/bench_clbeam.py mh ...
set -e
cd "$(dirname "$0")/../mcmc_amd/csrc"
make -j4 OBJDIR=../../abuild/prof OUT=../../abuild/prof/libmi355mcmc_prof.so EXTRA="-DMI355_CLMH_PROF $EXTRA"
