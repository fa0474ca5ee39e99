#!/usr/bin/env python3
"""tools/isa_cost.py <object.o> <kernel-name-regex> [--min N] — static VALU cost of a gfx950 kernel, per basic block.

The wave kernels are VALU-throughput bound (tools/ubench_tput.cpp: a SIMD retires one wave64 f32 mul/fma/add, v_and/xor/or,
v_add_u32, shift every ~1.1 ns, every other VALU instruction every ~1.8 ns, v_ashr_pk_u8_i32 every 3.45 ns, and the measured
launch time of tick_bgra_wave is the sum of these over the executed instructions), so the instruction mix of the big
straight-line blocks (the unrolled row loops) predicts the run time.  Prints, per basic block with at least --min VALU
instructions: VALU by class, SALU, LDS, VMEM and the block's VALU time in ns per wave."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = "/opt/rocm/lib/llvm/bin"
FAST = re.compile(r"^v_(mul_f32|fma_f32|fmac_f32|add_f32|sub_f32|subrev_f32|and_b32|or_b32|xor_b32|add_u32|sub_u32|subrev_u32|"
                  r"add_co_u32|addc_co_u32|sub_co_u32|subb_co_u32|lshlrev_b32|lshrrev_b32|ashrrev_i32|mov_b32|cndmask_b32|"
                  r"max_f32|min_f32|not_b32|cmp_\w+|cmpx_\w+|accvgpr_\w+|nop)(_e32|_e64|_dpp|_sdwa)?$")
QUARTER = re.compile(r"^v_(ashr_pk_u8_i32|ashr_pk_i8_i32|rcp_\w+|rsq_\w+|sqrt_\w+|exp_\w+|log_\w+|sin_\w+|cos_\w+|mul_lo_u32|mul_hi_u32|mul_hi_i32|mad_u64_u32|mad_i64_i32|pk_mul_f32|pk_fma_f32|pk_add_f32)")
NS = {"F": 1.1, "S": 1.8, "Q": 3.45}


def classify(op):
    if QUARTER.match(op):
        return "Q"
    if FAST.match(op) and not op.endswith("_sdwa"):
        return "F"
    return "S"


def disassemble(obj):
    tmp = Path(tempfile.mkdtemp())
    # (with no output operand llvm-objcopy rewrites its INPUT in place: the object would look newer than the library it was linked into)
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={tmp}/f.fatbin", obj, f"{tmp}/copy.o"])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={tmp}/f.fatbin", f"--output={tmp}/k.co"])
    return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--demangle", f"{tmp}/k.co"], text=True)


def main():
    obj, pat = sys.argv[1], re.compile(sys.argv[2])
    min_valu = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 40
    text = disassemble(obj)
    cur, blocks = None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            if re.match(r"^L\d+$|^\.L", name) or name.startswith("BB"):
                if cur is not None:
                    blocks.append({"label": name, "ins": []})
                continue
            cur = name if pat.search(name) else None
            if cur is not None:
                print(f"== {name}")
                blocks = [{"label": "entry", "ins": []}]
                kernels.append((name, blocks))
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*(//.*)?$", line)
        if not m:
            continue
        op = m.group(1)
        blocks[-1]["ins"].append(op)
        if op.startswith("s_cbranch") or op.startswith("s_branch") or op == "s_endpgm":
            blocks.append({"label": "+", "ins": []})


kernels = []
if __name__ == "__main__":
    main()
    for name, blocks in kernels:
        tot = {"F": 0, "S": 0, "Q": 0}
        print(f"{'block':>8} {'valu':>5} {'F':>4} {'S':>4} {'Q':>3} {'salu':>5} {'lds':>4} {'vmem':>4} {'smem':>4} {'ns':>7}")
        for i, b in enumerate(blocks):
            c = {"F": 0, "S": 0, "Q": 0}
            salu = lds = vmem = smem = 0
            for op in b["ins"]:
                if op.startswith("v_"):
                    c[classify(op)] += 1
                elif op.startswith("ds_"):
                    lds += 1
                elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                    vmem += 1
                elif op.startswith(("s_load", "s_buffer_load")):
                    smem += 1
                elif op.startswith("s_"):
                    salu += 1
            for k in c:
                tot[k] += c[k]
            n = sum(c.values())
            if n >= int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else n >= 40:
                ns = sum(c[k] * NS[k] for k in c)
                print(f"{i:>8} {n:>5} {c['F']:>4} {c['S']:>4} {c['Q']:>3} {salu:>5} {lds:>4} {vmem:>4} {smem:>4} {ns:>7.1f}")
                if "--hist" in sys.argv and i == int(sys.argv[sys.argv.index("--hist") + 1]):      # opcode histogram of one block
                    h = {}
                    for op in b["ins"]:
                        key = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
                        h[key] = h.get(key, 0) + 1
                    for op, k in sorted(h.items(), key=lambda kv: -kv[1]):
                        print(f"           {k:>5}  {op}" + (f"  [{classify(op)}]" if op.startswith("v_") else ""))
        print(f"   total static VALU: {sum(tot.values())} (F {tot['F']}, S {tot['S']}, Q {tot['Q']})")
