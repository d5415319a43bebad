#!/usr/bin/env python3
"""known bytes / counted bytes per calibration kernel (tools/calib.hip hbm) + the VALU issue costs -> one JSON document."""
import collections, csv, glob, json, os, re, sys
root = sys.argv[1]
known = None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for l in open(os.path.join(root, f"hbm_{c}.log")):
        if l.startswith('{"hbm_calibration_bytes"'):
            known = json.loads(l)["hbm_calibration_bytes"]
out = {"source": "tools/collect_calibration.sh on 1x MI355X: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE over tools/calib.hip "
                 "(1 GiB streaming buffers, 4x the Infinity Cache); counters are KiB per dispatch, medians of 3 dispatches",
       "hbm": {}, "valu": json.load(open(os.path.join(root, "valu.json")))}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            acc[name].append(float(r["Counter_Value"]) * 1024.0)
    for name, v in acc.items():
        if name not in known:
            continue
        med = sorted(v)[len(v) // 2]
        reads = name.startswith("cal_read")
        relevant = (c == "FETCH_SIZE") == reads or name == "cal_atomic8"
        e = out["hbm"].setdefault(name, {"known_bytes": known[name]})
        e[c + "_counted"] = round(med)
        if relevant and med > 0:
            e[c + "_factor_known_over_counted"] = round(known[name] / med, 4)
if os.path.exists(os.path.join(root, "atomics.json")):      # 64-bit atomic min: rate against the shape of the 512 bytes a wave touches
    out["atomics"] = json.load(open(os.path.join(root, "atomics.json")))
print(json.dumps(out, indent=1))
