#!/usr/bin/env python3
"""Extract the OpenCL-C kernel strings of the reference at BUILD time.

Reads /root/reference/Sources/SwiftVideo/kernels.cl.swift where it lies and
writes one `.cl` translation unit per `OpenCLKernel` case (prelude
`kOpenCLKernelMatrixFuncs` prepended, exactly as buildComputeKernel does,
compute.cl.swift:153-170) into the output directory (oracle/_ref/cl, which is
git-ignored).  No reference text is ever committed to this repository.

TEST INFRASTRUCTURE, NOT PRODUCT.
"""
import re
import sys
from pathlib import Path


def extract(swift_path: Path):
    text = swift_path.read_text()
    m = re.search(r'let kOpenCLKernelMatrixFuncs =\s*"""\n(.*?)"""', text, re.S)
    if not m:
        raise SystemExit("prelude not found")
    prelude = m.group(1)
    kernels = {}
    for km in re.finditer(r'case (\w+) =\s*"""\n(.*?)"""', text, re.S):
        kernels[km.group(1)] = km.group(2)
    return prelude, kernels


def main():
    src = Path(sys.argv[1])
    out = Path(sys.argv[2])
    out.mkdir(parents=True, exist_ok=True)
    prelude, kernels = extract(src)
    for name, body in kernels.items():
        (out / f"{name}.cl").write_text(prelude + "\n" + body)
    print(" ".join(sorted(kernels)))


if __name__ == "__main__":
    main()
