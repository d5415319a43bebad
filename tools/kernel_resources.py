#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in topo_kernels.hip (code-object metadata of a device-only compile).
   tools/kernel_resources.py [extra hipcc flags...]"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "topo-renderer_amd", "csrc", "topo_kernels.hip")
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --cuda-device-only -Wno-pass-failed".split()
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, *sys.argv[1:], "-c", "-o", "/tmp/kr_dev.o", src])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=/tmp/kr_dev.o",
                       "--targets=hip-amdgcn-amd-amdhsa--gfx950", "--output=/tmp/kr_gfx950.o"])
notes = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", "/tmp/kr_gfx950.o"], text=True)
cur = {}
rows = []
for l in notes.split("\n"):
    m = re.match(r"\s+\.(\w+):\s+(.*)", l)
    if not m:
        m2 = re.match(r"\s+- \.(\w+):\s+(.*)", l)
        if m2 and m2.group(1) in ("agpr_count", "args"):
            if cur.get("name"):
                rows.append(cur)
            cur = {}
            m = m2
        else:
            continue
    cur[m.group(1)] = m.group(2).strip()
if cur.get("name"):
    rows.append(cur)
seen = set()
for r in rows:
    n = re.search(r"(k_\w+)", r.get("name", ""))
    if not n or r["name"] in seen:
        continue
    seen.add(r["name"])
    tmpl = re.search(r"ILi(\d+)E", r["name"])
    print(f"{n.group(1) + ('<' + tmpl.group(1) + '>' if tmpl else ''):28s} vgpr {r.get('vgpr_count','?'):>4s} sgpr {r.get('sgpr_count','?'):>4s} "
          f"lds {r.get('group_segment_fixed_size','?'):>6s} scratch {r.get('private_segment_fixed_size','?'):>5s} spill v{r.get('vgpr_spill_count','0')} s{r.get('sgpr_spill_count','0')}")
