#!/usr/bin/env python3
"""Generate tests/golden/idle_vectors.npz — golden input / output vectors for the two kernels of `enum ComputeKernel` that no caller of the
reference dispatches (compute.swift:67,70): snd_s16i_s16i (kernels.cl.swift:534-562) and me_fullsearch (kernels.metal:129-267).

Run in the build container (where /root/reference exists):  python tests/golden/gen_idle_golden.py

Expected outputs come from the oracle (oracle/ref_kernels.c); snd_s16i_s16i is first checked sample for sample against the reference's own
kernel string compiled for x86-64 (oracle/_ref/libclref.so) — the generator refuses to write vectors the two disagree on; me_fullsearch
against the independent Python statement of the Metal source in tests/test_idle_kernels.py.  Inputs are stored literally (they are small)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import test_idle_kernels as T  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    out, index = {}, []
    have_ref = O.clref() is not None and hasattr(O.clref(), "clref_run_snd")
    for name, out0, ins, gains, fades in T.snd_cases():
        u = O.snd_uniforms(gains, fades)
        exp = out0.copy()
        assert O.snd_s16i_s16i(exp, ins, u) == 0
        if have_ref:
            chk = out0.copy()
            assert O.clref_snd_s16i_s16i(chk, ins, u) == 0 and np.array_equal(chk, exp), f"oracle != compiled reference kernel for snd/{name}"
        key = f"snd_s16i_s16i/{name}"
        index.append(key)
        out[key + "/uniforms"] = u
        out[key + "/out0"] = out0
        out[key + "/expected"] = exp
        out[key + "/n_inputs"] = np.array([len(ins)], dtype=np.int64)
        for i, a in enumerate(ins):
            out[key + f"/in{i}"] = a
    for i, (seed, w, h, block, window, smooth) in enumerate(T.ME_CASES):
        ref, cur = T.me_frames(seed, w, h, smooth)
        exp = np.zeros((-(-h // block[1]), -(-w // block[0]), 4), dtype=np.uint8)
        assert O.me_fullsearch(exp, ref, cur, block, window) == 0
        py = T.me_python(ref, cur, block, window, (w, h))
        assert np.array_equal(exp[: py.shape[0], : py.shape[1]], py), f"oracle != Python statement for me/{i}"
        key = f"me_fullsearch/{i}"
        index.append(key)
        out[key + "/uniforms"] = np.array([block[0], block[1], window[0], window[1], w, h], dtype=np.int32)
        out[key + "/ref"] = ref
        out[key + "/cur"] = cur
        out[key + "/expected"] = exp
    out["index"] = np.array(index)
    path = Path(__file__).resolve().parent / "idle_vectors.npz"
    np.savez_compressed(path, **out)
    print(f"{len(index)} cases -> {path} ({path.stat().st_size} bytes), snd cross-checked against the compiled reference kernel: {have_ref}")


if __name__ == "__main__":
    main()
