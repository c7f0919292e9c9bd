#!/usr/bin/env python3
"""Summarise a tools_prof.sh output directory: kernel stats + PMC counters of the MH kernel."""
import csv, glob, collections, sys
d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else 'k_mh_run'
for f in glob.glob(d + '/trace/vm/*_kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:6]:
        print('%-60s calls=%s avg_ms=%.3f pct=%s' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e6, r['Percentage']))
for sub in ['pmc_sq', 'pmc_sq2', 'pmc_fetch', 'pmc_write']:
    fs = glob.glob(d + '/' + sub + '/vm/*_counter_collection.csv')
    if not fs:
        continue
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(fs[0])):
        if kern in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = r
    # drop the first (adapt) launch: different step count
    print(sub, {k: '%.4g' % (sum(v[1:]) / max(1, len(v[1:]))) for k, v in agg.items()})
    if meta:
        print('   vgpr', meta['VGPR_Count'], 'agpr', meta['Accum_VGPR_Count'], 'sgpr', meta['SGPR_Count'], 'lds', meta['LDS_Block_Size'])
