#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_kernel.sh: summarize_pmc.py gpurun_out/pmc_<tag>"""
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_\w+)', r['Kernel_Name'])
        if not m:
            continue
        k = m.group(1)
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        meta[k] = dict(grid=r['Grid_Size'], vgpr=r['VGPR_Count'], sgpr=r['SGPR_Count'], lds=r['LDS_Block_Size'])
        acc[k]['dur_ns'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in acc.items():
    if k.startswith('k_n') or k.startswith('k_b'):
        continue
    print(k, meta[k])
    print('   ', {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
