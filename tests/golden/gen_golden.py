#!/usr/bin/env python3
"""Generate tests/golden/vectors.npz — golden input/output vectors for the picture kernels.

Run in the build container (where /root/reference exists):  python tests/golden/gen_golden.py

For every kernel x scenario of tests/scenarios.py the file stores the literal 59-float
ImageUniforms blob, the seeds of the inputs (bytes are regenerated with splitmix64,
tests/util.py) and the full expected output planes.  Expected outputs come from the
oracle (oracle/ref_kernels.c); for the ten kernels the reference implements in OpenCL C
they are first checked byte-for-byte against the reference's own kernel source compiled
for x86-64 (oracle/_ref/libclref.so, built by `make -C oracle clref` from
/root/reference/Sources/SwiftVideo/kernels.cl.swift) — the generator refuses to write
vectors that the two disagree on.  PARITY UNPINNED all the same: the sampler behind those
kernels is this repository's restatement of the OpenCL 1.2 specification (oracle/ref_kernels.h).
"""
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import scenarios as S  # noqa: E402
import util  # noqa: E402
from oracle import oracle as O  # noqa: E402


def case_seed(kernel, scenario):
    return (zlib.crc32(f"{kernel}/{scenario}".encode()) & 0xFFFF) + 1


def main():
    have_ref = O.clref() is not None
    out = {}
    index = []
    n_ref = 0
    for kernel in S.LAYER_KERNELS_REF + S.LAYER_KERNELS_OWN + ["img_bgra_bgra"]:
        _, s, d = kernel.split("_")[:3]
        for name, (cw, ch, iw, ih, kw) in S.SCENARIOS.items():
            if kernel == "img_bgra_bgra":
                u = util.full_canvas_uniforms((cw, ch), (iw, ih))   # Metal kernel reads only the sizes
            else:
                u = S.uniforms_for(name)
            seed = case_seed(kernel, name)
            src = util.alloc_image(s, iw, ih, seed=seed)
            canvas = util.alloc_image(d, cw, ch, seed=seed + 1000)
            exp = util.copy_image(canvas)
            assert O.run_kernel(kernel, exp, src, u) == 0
            if have_ref and kernel in S.LAYER_KERNELS_REF:
                chk = util.copy_image(canvas)
                assert O.run_clref(kernel, chk, src, u) == 0
                for a, b in zip(exp, chk):
                    assert np.array_equal(a, b), f"oracle != compiled reference kernel for {kernel}/{name}"
                n_ref += 1
            key = f"{kernel}/{name}"
            index.append(key)
            out[key + "/uniforms"] = u
            out[key + "/meta"] = np.array([cw, ch, iw, ih, seed], dtype=np.int64)
            for i, p in enumerate(exp):
                out[key + f"/out{i}"] = np.ascontiguousarray(p)
    # clear kernels
    for kernel in S.CLEAR_KERNELS:
        d = kernel.split("_")[2]
        for (cw, ch) in ((64, 36), (7, 5)):
            canvas = util.alloc_image(d, cw, ch, seed=5)
            assert O.run_kernel(kernel, canvas) == 0
            if have_ref:
                chk = util.alloc_image(d, cw, ch, seed=5)
                assert O.run_clref(kernel, chk) == 0
                for a, b in zip(canvas, chk):
                    assert np.array_equal(a, b)
                n_ref += 1
            key = f"{kernel}/{cw}x{ch}"
            index.append(key)
            out[key + "/meta"] = np.array([cw, ch, 0, 0, 5], dtype=np.int64)
            for i, p in enumerate(canvas):
                out[key + f"/out{i}"] = np.ascontiguousarray(p)
    # Lanczos-3
    for (iw, ih, ow, oh) in ((64, 36, 32, 18), (40, 30, 64, 48), (33, 17, 20, 11)):
        src = util.alloc_image("bgra", iw, ih, seed=900 + iw)
        dst = util.alloc_image("bgra", ow, oh)
        assert O.lanczos_bgra(dst[0], src[0]) == 0
        key = f"lanczos3/{iw}x{ih}_{ow}x{oh}"
        index.append(key)
        out[key + "/meta"] = np.array([ow, oh, iw, ih, 900 + iw], dtype=np.int64)
        out[key + "/out0"] = dst[0]
    out["index"] = np.array(index)
    dest = Path(__file__).resolve().parent / "vectors.npz"
    np.savez_compressed(dest, **out)
    print(f"wrote {dest}: {len(index)} cases, {n_ref} cross-checked against the compiled reference kernels "
          f"(libclref {'present' if have_ref else 'ABSENT'})")


if __name__ == "__main__":
    main()
