#!/usr/bin/env python3
"""tools/obj_regs.py <stem> [--asm out.s] — per kernel of the BUILT object swiftvideo_amd/csrc/<stem>.hip.o: VGPRs, SGPRs, spills, scratch,
LDS, code bytes and the waves per SIMD the register count allows (no recompilation; tools/kernel_regs.py compiles a source with extra flags)."""
import re, subprocess, sys, tempfile
from pathlib import Path
LLVM = Path("/opt/rocm/lib/llvm/bin")
CSRC = Path(__file__).resolve().parents[1] / "swiftvideo_amd" / "csrc"
stem = sys.argv[1]
asm_out = sys.argv[sys.argv.index("--asm") + 1] if "--asm" in sys.argv else None
with tempfile.TemporaryDirectory() as d:
    d = Path(d)
    subprocess.run([LLVM / "llvm-objcopy", f"--dump-section=.hip_fatbin={d/'f.fatbin'}", CSRC / f"{stem}.hip.o", d / "copy.o"], check=True)   # (an output operand: otherwise the input is rewritten in place)
    subprocess.run([LLVM / "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={d/'f.fatbin'}", f"--output={d/'k.co'}"], check=True)
    notes = subprocess.run([LLVM / "llvm-readelf", "--notes", d / "k.co"], check=True, capture_output=True, text=True).stdout
    syms = subprocess.run([LLVM / "llvm-readelf", "-s", "--wide", d / "k.co"], check=True, capture_output=True, text=True).stdout
    if asm_out:
        Path(asm_out).write_text(subprocess.run([LLVM / "llvm-objdump", "-d", d / "k.co"], check=True, capture_output=True, text=True).stdout)
sizes = {}
for line in syms.splitlines():
    f = line.split()
    if len(f) >= 8 and f[3] == "FUNC":
        sizes[f[7]] = int(f[2])
cur, rows = None, []
for line in notes.splitlines():
    m = re.match(r"\s*\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", line)
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
    print(f"{n:58s} vgpr {v:4d} sgpr {r.get('sgpr_count', 0):4d} vspill {r.get('vgpr_spill_count', 0):3d} sspill {r.get('sgpr_spill_count', 0):3d} "
          f"scratch {r.get('private_segment_fixed_size', 0):4d} code {sizes.get(r['name'], 0):6d} waves/SIMD {min(8, 512 // max(8, (v + 7) // 8 * 8))}")
