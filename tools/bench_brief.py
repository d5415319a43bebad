#!/usr/bin/env python3
"""One line per bench run: tools/bench_brief.py LABEL < bench-output (prints ms_per_step, kernel_ms and the counters)."""
import json, sys
d = json.loads(sys.stdin.read().strip().split("\n")[-1])
c = d.get("counters", {})
print(sys.argv[1] if len(sys.argv) > 1 else "", d["ms_per_step"], {k: round(v, 3) for k, v in d.get("kernel_ms", {}).items()},
      {k: c[k] for k in ("near_blocks", "far_tested", "far_survived", "big_items", "rare_items") if k in c})
