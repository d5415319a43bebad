#!/usr/bin/env python3
"""The GPU timeline of a frame from a `rocprofv3 --kernel-trace --output-format csv` run: per kernel of the frame's chain its mean
duration and the mean idle time in front of it (previous kernel's end -> this kernel's start), over the last frames of the trace.
python tools/trace_timeline.py <dir with *_kernel_trace.csv> [frames]"""
import csv, glob, os, re, sys
d = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        m = re.search(r"\bk_\w+", r["Kernel_Name"])
        name = m.group(0) if m else r["Kernel_Name"].split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# a frame starts at k_put_views (or k_clear_cull / k_clear when the views come by copy)
first = "k_put_views" if any(n.startswith("k_put_views") for _, _, n in rows) else "k_clear"
starts = [i for i, r in enumerate(rows) if r[2].startswith(first)]
starts = starts[-frames - 1:]
acc, cnt, order = {}, {}, []
span = []
for a, b in zip(starts[:-1], starts[1:]):
    fr = rows[a:b]
    span.append(rows[b][0] - fr[0][0])
    seen = {}
    for j, (s, e, n) in enumerate(fr):
        k = seen.get(n, 0)
        seen[n] = k + 1
        key = f"{n}#{k}"
        if key not in acc:
            acc[key] = [0, 0]
            cnt[key] = 0
            order.append(key)
        prev_end = rows[a + j - 1][1] if a + j > 0 else s
        acc[key][0] += e - s
        acc[key][1] += s - prev_end
        cnt[key] += 1
print(f"{path}: {len(span)} frames, start-to-start {sum(span) / len(span) / 1e3:.1f} us")
tot_d = tot_g = 0.0
for key in order:
    dur, gap = acc[key][0] / cnt[key] / 1e3, acc[key][1] / cnt[key] / 1e3
    tot_d += dur * cnt[key] / len(span)
    tot_g += gap * cnt[key] / len(span)
    print(f"  {key:44s} x{cnt[key]:4d}  idle before {gap:7.2f} us   runs {dur:7.2f} us")
print(f"  sum per frame: kernels {tot_d:.1f} us + idle {tot_g:.1f} us")
