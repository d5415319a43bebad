#!/usr/bin/env python3
"""Static VALU/SALU/memory instruction counts of one kernel, attributed to source lines.
   tools/isa_by_line.py <kernel-name-substring> [min_count]   (needs hipcc; writes scratch files under /tmp)"""
import collections, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "topo-renderer_amd", "csrc", "topo_kernels.hip")
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -gline-tables-only --cuda-device-only -Wno-pass-failed".split()
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", "-o", "/tmp/tk_dev.o", src])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=/tmp/tk_dev.o",
                       "--targets=hip-amdgcn-amd-amdhsa--gfx950", "--output=/tmp/tk_gfx950.o"])
dis = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "-l", "/tmp/tk_gfx950.o"], text=True)
want, minc = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
inside, line = False, "?"
cnt = collections.defaultdict(lambda: collections.Counter())
for l in dis.split("\n"):
    m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
    if m:
        inside = want in m.group(1)
        continue
    if not inside:
        continue
    m = re.match(r"^; (/.*):(\d+)", l)
    if m:
        line = os.path.basename(m.group(1)) + ":" + m.group(2)
        continue
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if m:
        op = m.group(1)
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
        cnt[line][kind] += 1
tot = collections.Counter()
for c in cnt.values():
    tot.update(c)
print("total", dict(tot))
srcs = {}
for k, c in sorted(cnt.items(), key=lambda kv: -kv[1]["valu"]):
    if c["valu"] < minc:
        break
    f, n = k.split(":")
    path = os.path.join(R, "topo-renderer_amd", "csrc", f)
    if f not in srcs and os.path.exists(path):
        srcs[f] = open(path).read().split("\n")
    text = srcs[f][int(n) - 1].strip()[:110] if f in srcs else ""
    print(f"{k:24s} valu {c['valu']:4d} salu {c['salu']:3d} lds {c['lds']:3d} vmem {c['vmem']:3d} | {text}")
