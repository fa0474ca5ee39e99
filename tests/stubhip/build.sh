#!/bin/bash
# tests/stubhip/build.sh <address|thread> <out> — chipvideo.cpp (the host side of libchipvideo) + the stand-in runtime + the stress program, for the CPU, under a sanitizer
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$HERE/../.."
SAN="-fsanitize=$1"; [ "$1" = address ] && SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer $SAN -pthread -I"$HERE" -I"$ROOT/include" -I"$ROOT/swiftvideo_amd/csrc" \
    -D__clang_major__=0 -D__clang_minor__=0 -DCHV_ARCH=\"gfx950\" -DCHV_HIPCC_VERSION=\"stub\" -ffp-contract=off -w \
    "$ROOT/swiftvideo_amd/csrc/chipvideo.cpp" "$HERE/stub_runtime.cpp" "$HERE/stub_launchers.cpp" "$HERE/abi_stress.cpp" -o "$2"
