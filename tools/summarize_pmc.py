#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_kernel.sh:
   summarize_pmc.py gpurun_out/pmc_<tag> [--json]     (--json: one document, for profiles/)"""
import collections, csv, glob, json, re, sys
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
out = {}
for k, v in acc.items():
    out[k] = {"launch": meta[k], "launches_per_pass": len(v['dur_ns']) // max(1, len(glob.glob(sys.argv[1] + '/*/*counter_collection.csv'))),
              "mean_per_launch": {c: round(sum(x) / len(x)) for c, x in sorted(v.items())}}
if "--json" in sys.argv:
    print(json.dumps({"source": "rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc (tools/pmc_kernel.sh), "
                                "one pass per counter group; means over all launches of a kernel in a pass; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* "
                                "are quad-cycles summed over waves, dur_ns is the kernel's duration under the profiler (kernels serialised)",
                      "kernels": out}, indent=1))
else:
    for k, e in out.items():
        if k.startswith('k_n') or k.startswith('k_b'):
            continue
        print(k, e["launch"])
        print('   ', e["mean_per_launch"])
