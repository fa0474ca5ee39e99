"""What the compiled gfx950 code objects must look like for the occupancy and memory-path assumptions of DESIGN.md §5 to
hold: no FLAT accesses in the kernels (FLAT counts in lgkmcnt as well as vmcnt and stalls every LDS wait behind the
global prefetch), the register budgets that give the documented waves per SIMD, spills only where they were measured and
accepted.  Reads the object files `make -C swiftvideo_amd/csrc` / `__graft_entry__.build()` leave in-tree (skipped when
they are not there); no GPU needed."""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "swiftvideo_amd" / "csrc"
LLVM = Path("/opt/rocm/lib/llvm/bin")
OBJECTS = ["kernels_general", "kernels_fast", "kernels_wave", "kernels_wave_cached", "kernels_wave_yuv", "kernels_wave_yuv_cached", "kernels_lanczos", "kernels_stream", "kernels_stream_yuv"]
VALIDATED_HIPCC_MAJOR_MINOR = "7.2"       # the hipcc the hand-scheduled kernels were validated with (GPU suite + tools/check_inflight.py)


def _code_object(tmp_path, stem):
    obj = CSRC / f"{stem}.hip.o"
    if not obj.exists() or not (LLVM / "llvm-objcopy").exists():
        pytest.skip(f"{obj.name} not built here")
    fat, co = tmp_path / f"{stem}.fatbin", tmp_path / f"{stem}.co"
    # (an output operand: without one llvm-objcopy rewrites the object in place and the next `make` relinks the library)
    subprocess.run([LLVM / "llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, tmp_path / f"{stem}.copy.o"], check=True)
    subprocess.run([LLVM / "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True)
    return co


def _kernels(co):
    """{demangled-ish name: {vgpr, vgpr_spill, sgpr_spill, lds}} from the code object's metadata note."""
    notes = subprocess.run([LLVM / "llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "name":
            cur = out.setdefault(v, {})
        elif cur is not None:
            cur[k] = int(v)
    return out


@pytest.mark.parametrize("stem", OBJECTS)
def test_no_flat_accesses(tmp_path, stem):
    co = _code_object(tmp_path, stem)
    asm = subprocess.run([LLVM / "llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    flat = [l.strip() for l in asm.splitlines() if re.search(r"\bflat_(load|store|atomic)", l)]
    assert not flat, f"{stem}: FLAT accesses (use gld/gst, pixel_math.hip.h): {flat[:3]}"
    assert re.search(r"\bglobal_(load|store)", asm), "no global accesses found: disassembly did not work"


def _find(kernels, fragment):
    hits = {n: k for n, k in kernels.items() if fragment in n}
    assert hits, f"no kernel named *{fragment}* in {sorted(kernels)[:4]}..."
    return hits


def test_nv12_tiled_kernels_keep_six_waves(tmp_path):
    every = _kernels(_code_object(tmp_path, "kernels_fast"))
    # clear x planar x (2+1 | 3+2 prefetch vectors) x (16 | 32 rows), each as the pointer kernel and as its by-value twin for transient
    # launches (tick_yuv_bgra_tiled_one: lone ticks, latency-bound — a few scalars in scratch are accepted there), + canvas_clear_bgra
    assert len(every) == 33
    twins = _find(every, "tick_yuv_bgra_tiled_one")
    assert len(twins) == 16
    for name, m in twins.items():
        assert m["vgpr_count"] <= 128 and m["vgpr_spill_count"] <= 12, (name, m)
    k = {n: m for n, m in every.items() if "tick_yuv_bgra_tiled" in n and n not in twins}
    assert len(k) == 16
    # NV12 (planar = Lb0E): <= 80 VGPRs = 6 waves per SIMD; the 32-row (3, 2) instantiation (the cfg2 bench kernel) may
    # spill the two registers it was measured with, nothing else spills
    for name, m in _find(k, "ELb0ELi").items():
        rows32, big = name.endswith("Li32EEEvPKNS_5DTickEPKNS_6DLayerEiiiiiiii"), "Li3ELi2E" in name
        limit = 80 if (rows32 or not big) else 96
        assert m["vgpr_count"] <= limit, (name, m)
        assert m["vgpr_spill_count"] <= (2 if (rows32 and big) else 0), (name, m)
    # planar sources: 5 waves (2 + 1) / 4 waves (3 + 2), no spills
    for name, m in _find(k, "ELb1ELi").items():
        assert m["vgpr_count"] <= (128 if "Li3ELi2E" in name else 102) and m["vgpr_spill_count"] == 0, (name, m)


def test_lanczos_exact_tap_kernels_do_not_spill(tmp_path):
    k = _kernels(_code_object(tmp_path, "kernels_lanczos"))
    for frag in ("lanczos3_bgraILi12ELb1ELb1ELi32ELi16E", "lanczos3_bgraILi6ELb1ELb1ELi32ELi16E"):
        for name, m in _find(k, frag).items():
            assert m["vgpr_count"] <= 128 and m["vgpr_spill_count"] == 0, (name, m)   # 4 blocks of 4 waves per CU (LDS-limited)


def test_lanczos_strip_kernel_keeps_its_prefetch_in_flight(tmp_path):
    """lanczos3_strip2 (2:1 reductions, one wave per strip): 4 waves per SIMD without scratch, and the prefetched source rows are
    awaited with `s_waitcnt vmcnt(3)` (three younger loads stay in flight).  Left to the compiler every row waited with vmcnt(0) —
    gfx950 has one counter for loads and stores and completes them out of order with respect to each other, and the loop stores its
    output rows — which cost a third of the kernel's time (profiles/r03_notes.md section 8)."""
    co = _code_object(tmp_path, "kernels_lanczos")
    for name, m in _find(_kernels(co), "lanczos3_strip2").items():
        assert m["vgpr_count"] <= 128 and m["vgpr_spill_count"] == 0 and m.get("private_segment_fixed_size", 0) == 0, (name, m)
    asm = subprocess.run([LLVM / "llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    bodies = re.split(r"\n[0-9a-f]+ <(_ZN3chv15lanczos3_strip2[^>]*)>:\n", asm)
    assert len(bodies) >= 5, "two instantiations expected"
    for name, body in zip(bodies[1::2], bodies[2::2]):
        body = re.split(r"\n[0-9a-f]+ <_Z", body)[0]
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)
        assert waits.count("3") >= 12 and waits.count("0") <= 3, (name, waits)
        assert len(re.findall(r"global_load_dwordx4", body)) >= 16, name


def test_stream_kernel_owns_m0_and_keeps_six_waves(tmp_path):
    """tick_bgra_stream fills its LDS rings with global_load_lds_dwordx4, whose LDS address is M0: set by hand (s_mov_b32 m0) right in
    front of every such load, and nothing else in the object may touch M0 (the compiler is not told).  Four layers: <= 80 VGPRs
    (6 waves per SIMD; the rings allow 6), nothing in scratch."""
    co = _code_object(tmp_path, "kernels_stream")
    for name, m in _find(_kernels(co), "tick_bgra_stream").items():
        assert m["vgpr_count"] <= 80 and m["vgpr_spill_count"] == 0 and m.get("private_segment_fixed_size", 0) == 0, (name, m)
    asm = subprocess.run([LLVM / "llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    lines = [l.split("//")[0].strip() for l in asm.splitlines()]
    lines = [l for l in lines if l and not l.endswith(":")]
    dma = [i for i, l in enumerate(lines) if l.startswith("global_load_lds_dwordx4")]
    assert len(dma) >= 8
    for i in dma:
        assert any(l.startswith("s_mov_b32 m0") for l in lines[max(0, i - 3):i]), lines[max(0, i - 3):i + 1]
    others = [l for l in lines if re.search(r"\bm0\b", l) and not l.startswith("s_mov_b32 m0")]
    assert not others, others[:3]


def test_stream_kernel_keeps_its_shortened_row(tmp_path):
    """The headline instantiation (four NV12 layers, absorbed colour matrix): what round 6 took OUT of a row must stay out.  Per layer and row:
    12 taps + 3 blend inputs through v_fma_mix_f32 (no v_cvt_f32_ubyte2 in front of the blend), one integer add (the green offset: red and blue
    are absorbed into the conversion biases), no mask and no subtraction of the rounding constant, the row's store addressed from a scalar base.
    Counted on the whole kernel (its row loop is not unrolled): the plain-matrix twin carries the three adds per layer."""
    co = _code_object(tmp_path, "kernels_stream")
    asm = subprocess.run([LLVM / "llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    bodies = re.split(r"\n[0-9a-f]+ <(_ZN3chv16tick_bgra_streamILi4ELb0ELb[01]E[^>]*)>:\n", asm)
    found = {}
    for name, body in zip(bodies[1::2], bodies[2::2]):
        body = re.split(r"\n[0-9a-f]+ <_Z", body)[0]
        ops = [l.split("//")[0].split()[0] for l in body.splitlines() if l.strip() and not l.strip().endswith(":")]
        found["ELb1EEE" in name[:40] or name.startswith("_ZN3chv16tick_bgra_streamILi4ELb0ELb1E")] = ops
    assert set(found) == {True, False}
    ab, plain = found[True], found[False]
    assert ab.count("v_fma_mix_f32") == 60 and plain.count("v_fma_mix_f32") == 48
    assert ab.count("v_cvt_f32_ubyte2_e32") == 0 and plain.count("v_cvt_f32_ubyte2_e32") == 12
    assert ab.count("v_mad_i32_i24") == plain.count("v_mad_i32_i24") == 16
    assert plain.count("v_add_u32_e32") - ab.count("v_add_u32_e32") == 8          # two channel offsets x four layers
    assert ab.count("v_readfirstlane_b32") <= 12                                     # (one per row: the packed row entry; the rest is set-up)
    for ops in (ab, plain):
        assert ops.count("v_lshl_add_u64") <= 5 and ops.count("global_store_dword") == 2       # (no 64-bit vector add per row in front of the store)


def _lds_dma_contract(co, min_loads):
    asm = subprocess.run([LLVM / "llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    lines = [l.split("//")[0].strip() for l in asm.splitlines()]
    lines = [l for l in lines if l and not l.endswith(":")]
    dma = [i for i, l in enumerate(lines) if l.startswith("global_load_lds_dwordx4")]
    assert len(dma) >= min_loads
    for i in dma:
        assert any(l.startswith("s_mov_b32 m0") for l in lines[max(0, i - 3):i]), lines[max(0, i - 3):i + 1]
    others = [l for l in lines if re.search(r"\bm0\b", l) and not l.startswith("s_mov_b32 m0")]
    assert not others, others[:3]


def test_yuv_stream_kernel_owns_m0_and_keeps_its_waves(tmp_path):
    """tick_yuv_stream: the same M0 contract as tick_bgra_stream.  Registers (what the measurements of profiles/r04_notes.md were taken with):
    one video layer slot — 80 VGPRs (6 waves per SIMD) with at most a few registers in scratch (88 registers and 5 waves measured 3 % slower); the
    encoder-side integer instantiation: 80 registers as well (its trip loop carries eight texel values down the lane; six waves with a handful of registers
    in scratch measured 0.5575 ms on encode_nv12 against 0.583 at five waves with none and 0.591 at four — same call); two video layers 96; the mixer instantiations (video + RGB overlays, three or four
    layer slots) 128 — their rings leave the LDS to four waves per SIMD anyway."""
    co = _code_object(tmp_path, "kernels_stream_yuv")
    kernels = _find(_kernels(co), "tick_yuv_stream")
    assert len(kernels) >= 20
    for name, m in kernels.items():
        nl, kinds = [int(x) for x in re.search(r"ILi\d+ELi(\d+)ELi(\d+)E", name).groups()]
        scratch = m.get("private_segment_fixed_size", 0)
        if kinds in (1, 2, 8) and nl == 1:
            if "tick_yuv_stream_one" in name:
                # a lone tick's twin: never more than five waves, and with that NOTHING in scratch (the encoder-side frame's twin kept seven
                # registers there at six: a kernel with a scratch segment is dispatched ~1 us later — profiles/r06_notes.md section 15)
                assert m["vgpr_count"] <= 96 and scratch == 0, (name, m)
            else:
                assert m["vgpr_count"] <= 80 and scratch <= 48, (name, m)
        elif kinds in (1, 2):
            assert m["vgpr_count"] <= 96, (name, m)
        else:
            assert m["vgpr_count"] <= 128, (name, m)
    _lds_dma_contract(co, 16)


def test_compiler_is_the_one_the_hand_scheduled_kernels_were_validated_with(built):
    """The streaming and Lanczos strip kernels rely on behaviours of gfx950 AND of the compiler that nothing in the language promises: loads
    return in order and share `vmcnt` with stores, hipcc never touches M0 by itself, inline-asm loads stay where they are written.  They were
    checked on the built code (this file, tools/check_inflight.py) and on the GPU with one hipcc; chv_build_flags() carries the version the
    library was built with, and another major.minor has to be re-validated (run the GPU suite and tools/check_inflight.py, then move the pin)."""
    from swiftvideo_amd import chipvideo as cv
    flags = dict(f.split("=", 1) for f in cv.build_flags().split(";") if "=" in f and ":" not in f.split("=", 1)[0])
    assert "hipcc" in flags and "clang" in flags, cv.build_flags()
    assert flags["hipcc"].startswith(VALIDATED_HIPCC_MAJOR_MINOR + "."), (
        f"libchipvideo.so was built with hipcc {flags['hipcc']}; the hand-scheduled kernels were validated with {VALIDATED_HIPCC_MAJOR_MINOR}.x — "
        "re-run `pytest -m gpu` and tools/check_inflight.py with this compiler, then update VALIDATED_HIPCC_MAJOR_MINOR")


def test_hand_awaited_loads_are_not_touched_while_in_flight(tmp_path):
    """The strip Lanczos kernels issue their row loads from inline asm and wait for them with a hand-written s_waitcnt: the compiler does
    not know the destination registers are still being written.  Two earlier versions of lanczos3_strip<T> let it copy such registers at
    control-flow joins (garbage pixels, then a memory fault); tools/check_inflight.py walks the built code and rejects any instruction
    that names a prefetch quad between its load and its wait, any spill of one, any reload of one still in flight."""
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    import check_inflight
    co = _code_object(tmp_path, "kernels_lanczos")
    asm = subprocess.run([LLVM / "llvm-objdump", "-d", "--no-show-raw-insn", "--demangle", co], check=True, capture_output=True, text=True).stdout
    seen, bad = check_inflight.check(asm, "lanczos3_strip")
    assert seen == 11, seen                       # lanczos3_strip2<odd|even> + lanczos3_strip<6 .. 22>
    assert not bad, bad[:5]


@pytest.mark.parametrize("stem", ["kernels_wave", "kernels_wave_yuv", "kernels_wave_cached", "kernels_wave_yuv_cached"])
def test_wave_kernels_keep_six_waves_and_scalar_descriptor_reads(tmp_path, stem):
    """One wave per strip (profiles/HISTORY.md 5.1): <= 80 VGPRs = 6 waves per SIMD (5 measured 8 % slower on the 4 x NV12
    pipeline), at most the two spills of the per-pixel fallback; and the tick / layer descriptors — uniform, read-only — come
    through the scalar unit.  They stopped doing so twice while these kernels were written: once through a fence over all
    memory, once through an `asm volatile` (touch_regs), either of which makes the compiler treat later descriptor reads as
    possibly clobbered and fetch every uniform with a per-lane global_load_dword (90 vector loads per wave, +35 % run time)."""
    co = _code_object(tmp_path, stem)
    for name, m in _find(_kernels(co), "wave").items():
        tall = "tick_bgra_waveILi16E" in name             # BGRA canvas, 16-row strips: 96 VGPRs = 5 waves, no spills
        # (the instantiation that also carries the per-pixel code for rotated layers — KINDS bit 3 — gets one wave less instead of
        # scratch traffic: the kernels are VALU-bound, profiles/r03_notes.md, and a fifth / sixth wave buys ~3 %)
        with_general = "tick_bgra_wave" in name and re.search(r"ELi15ELb[01]EEEv", name) is not None
        if "tick_yuv_wave" in name:
            # 4:2:0 canvases: the own-format instantiations (KINDS 1 / 2) at 6 waves, a register or two in scratch at most
            # (16-row strips: 0.41 vs 0.43 ms on y420p_main at 5); the ones that also carry the RGB-overlay rows at 5 waves with
            # NOTHING in scratch (at 6 they spilled five or six registers: 1.5x write traffic and no faster)
            own = re.search(r"ELi(8|16)ELi[12]ELb[01]EEEv", name) is not None
            # (since the lone tick's descriptors travel as the kernels' last argument — WaveOne, wave_common.hip.h — the two chosen pointers
            # cannot be reloaded from the kernarg segment at will: the uncleared 8-row instantiations with the per-pixel code keep ONE
            # register in scratch)
            per_pixel_uncleared = re.search(r"ELb0ELi8ELi15ELb0EEEv", name) is not None
            assert m["vgpr_count"] <= (80 if own else 96), (name, m)
            assert m["private_segment_fixed_size"] <= (64 if own else 8 if per_pixel_uncleared else 0), (name, m)
            assert name.endswith("NS_7WaveOneE"), name              # every instantiation takes a lone tick's descriptors by value
        elif "tick_bgra_wave_one" in name:
            # the lone tick's twins (descriptors as the first kernel argument): five waves, and NOTHING in scratch — a kernel with a scratch segment
            # is dispatched 2 us later, which is what the twins exist to avoid (at six waves they kept 2 - 14 registers there)
            assert m["vgpr_count"] <= 96 and m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (name, m)
        else:
            assert m["vgpr_count"] <= ((128 if with_general else 96) if tall else (96 if with_general else 80)), (name, m)
            assert m["vgpr_spill_count"] <= (0 if tall else 2), (name, m)
    asm = subprocess.run([LLVM / "llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    scalar = len(re.findall(r"\bs_load_dword", asm))
    vector1 = len(re.findall(r"\bglobal_load_dword\s", asm))        # single-dword vector loads: what a uniform read degrades to
    assert scalar >= 300 and vector1 * 4 <= scalar, f"{stem}: {scalar} scalar loads, {vector1} single-dword vector loads"
    # (descriptor pointers chosen between a batch's arrays and the kernarg segment must stay global / constant pointers: a generic one reads with flat_load)
    assert not re.search(r"\bflat_load", asm), f"{stem}: flat loads"


def test_wave_kernel_dma_staging_owns_m0_and_leaves_descriptor_reads_scalar(tmp_path):
    """tick_bgra_wave<.., KINDS = 4> (launches of RGB layers only: cfg3, cfg5) fills interior rectangles by global_load_lds_dwordx4 — the M0
    contract of the streaming kernels — from asm statements that are NOT volatile and clobber no memory: a first version that was volatile made
    every descriptor read after it a per-lane load (75 instead of 16 global_load_dword in this instantiation, cfg3 1.35 -> 2.41 ms on the GPU,
    profiles/r05_notes.md section 9).  Per instantiation, not over the object: the aggregate test above did not notice."""
    seen = 0
    bodies = []
    for stem in ("kernels_wave", "kernels_wave_cached"):          # (the instantiations that compute their geometry, and the ones that read it from tables)
        co = _code_object(tmp_path, stem)
        if stem == "kernels_wave":
            _lds_dma_contract(co, 4)                                # (RGB-only launches never take tables: KINDS = 4 exists in the first object only)
        asm = subprocess.run([LLVM / "llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
        bodies += re.split(r"\n[0-9a-f]+ <(_ZN3chv14tick_bgra_wave[^>]*)>:\n", asm)[1:]
    bodies = [None] + bodies
    for name, body in zip(bodies[1::2], bodies[2::2]):
        body = re.split(r"\n[0-9a-f]+ <_Z", body)[0]
        rgb_only = re.search(r"ELi4ELb[01]EEEv", name) is not None
        assert ("global_load_lds_dwordx4" in body) == rgb_only, name
        if rgb_only:
            seen += 1
            scalar, vector1 = len(re.findall(r"\bs_load_dword", body)), len(re.findall(r"\bglobal_load_dword\s", body))
            assert scalar >= 80 and vector1 <= 40, (name, scalar, vector1)       # (CLEAR = false: one canvas load per row on top of the 16 of the per-pixel path)
            assert "s_waitcnt vmcnt(0)" in body
    assert seen == 4
