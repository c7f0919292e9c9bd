#!/bin/bash
# timing ablations of the fused SED kernel (build/libabl<k>.so built with -DMI355_ABL=k):
# 1 no moments, 2 no RNG, 3 no band loop, 4 no accept exp.  Not product code.
for k in 0 1 2 3 4; do
  if [ $k = 0 ]; then unset MI355MCMC_LIB; else export MI355MCMC_LIB=$PWD/build/libabl$k.so; fi
  echo -n "ABL=$k "
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --ess-steps 0 --adapt 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'])" || exit 1
done
