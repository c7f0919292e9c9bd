#!/bin/bash
# A variant build of the library for A/B timing on one box (tools/ab_bench.sh, tools/ab_ext.sh):
# abuild/lib<tag>.so with the given extra compiler flags.  usage: tools/build_variant.sh <tag> [flags...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../mcmc_amd/csrc"
make -j4 OBJDIR=../../abuild/obj_$tag OUT=../../abuild/lib$tag.so EXTRA="$*"
