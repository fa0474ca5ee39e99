#!/usr/bin/env python3
"""tools/kernel_ops.py <stem> <name-substring> [prefix,...] — opcode counts of the kernels of swiftvideo_amd/csrc/<stem>.hip.o whose mangled
name contains the substring (static counts, not executed ones): memory and wait instructions by default."""
import re, subprocess, sys, tempfile, pathlib
from collections import Counter
ROOT = pathlib.Path(__file__).resolve().parent.parent
LLVM = pathlib.Path("/opt/rocm/lib/llvm/bin")
def main():
    stem, pat = sys.argv[1], sys.argv[2]
    pre = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("global_", "s_load", "scratch", "s_waitcnt", "buffer", "s_buffer", "ds_")
    d = pathlib.Path(tempfile.mkdtemp())
    obj = ROOT / "swiftvideo_amd" / "csrc" / f"{stem}.hip.o"
    subprocess.run([LLVM / "llvm-objcopy", f"--dump-section=.hip_fatbin={d/'f'}", obj, d / "copy.o"], check=True)
    subprocess.run([LLVM / "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={d/'f'}", f"--output={d/'co'}"], check=True)
    s = subprocess.run([LLVM / "llvm-objdump", "-d", "--no-show-raw-insn", d / "co"], check=True, capture_output=True, text=True).stdout
    if len(sys.argv) > 4: open(sys.argv[4], "w").write(s)
    parts = re.split(r"\n[0-9a-f]+ <(_Z[^>]*)>:\n", s)
    for name, body in zip(parts[1::2], parts[2::2]):
        if pat not in name: continue
        body = re.split(r"\n[0-9a-f]+ <_Z", body)[0]
        ins = [l.split()[0] for l in body.splitlines() if l.strip() and not l.strip().endswith(":")]
        c = Counter(ins)
        print(name[:70], len(ins), dict(sorted((k, v) for k, v in c.items() if k.startswith(pre))))
main()
