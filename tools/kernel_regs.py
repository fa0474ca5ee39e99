#!/usr/bin/env python3
"""tools/kernel_regs.py <file.hip.cpp> [-DFLAG ...] — compile one kernel source for gfx950 and print, per kernel, VGPRs, spilled
VGPRs, scratch bytes and the waves per SIMD the register count allows (512 VGPRs per SIMD lane, granularity 8)."""
import re, subprocess, sys, tempfile
from pathlib import Path
LLVM = Path("/opt/rocm/lib/llvm/bin")
CSRC = Path(__file__).resolve().parents[1] / "swiftvideo_amd" / "csrc"
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w".split()
src, extra = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    d = Path(d)
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-x", "hip", "-c", str(CSRC / src), "-o", str(d / "o.o")], check=True)
    subprocess.run([LLVM / "llvm-objcopy", f"--dump-section=.hip_fatbin={d/'f.fatbin'}", d / "o.o"], check=True)
    subprocess.run([LLVM / "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={d/'f.fatbin'}", f"--output={d/'k.co'}"], check=True)
    notes = subprocess.run([LLVM / "llvm-readelf", "--notes", d / "k.co"], check=True, capture_output=True, text=True).stdout
cur, rows = None, []
for line in notes.splitlines():
    m = re.match(r"\s*\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = int(v)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void chv::", "")
    v = r.get("vgpr_count", 0)
    print(f"{n:60s} vgpr {v:4d} spill {r.get('vgpr_spill_count', 0):3d} scratch {r.get('private_segment_fixed_size', 0):4d} waves/SIMD {min(8, 512 // max(8, (v + 7) // 8 * 8))}")
