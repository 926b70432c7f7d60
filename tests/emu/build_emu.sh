#!/bin/bash
# TEST INFRASTRUCTURE: builds the CPU lane-emulated copy of the kernels (tests/emu/build/libantmmf_emu.so).
# Safe under pytest-xdist: one builder at a time (flock), skipped when the library is newer than every source, installed atomically.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CXX="${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
mkdir -p "$ROOT/tests/emu/build"
SRC="$ROOT/ant-multi-modal-framework_amd/csrc"
OUT="$ROOT/tests/emu/build/libantmmf_emu.so"
exec 9>"$ROOT/tests/emu/build/.build.lock"
flock 9
if [ -f "$OUT" ] && [ -z "$(find "$SRC" "$ROOT/tests/emu/hip_emu.h" -newer "$OUT" \( -name '*.hip' -o -name '*.h' \) -print -quit)" ]; then exit 0; fi
"$CXX" -std=c++20 -O1 -pthread -fPIC -shared -DANTMMF_EMULATE -Wno-unused-value -I"$ROOT/tests/emu" -I"$SRC" -x c++ \
  "$SRC/abi.hip" "$SRC/layernorm.hip" "$SRC/elementwise.hip" "$SRC/gemm.hip" "$SRC/attention.hip" "$SRC/loss.hip" "$SRC/tpmcl.hip" "$SRC/resize.hip" "$SRC/frames.hip" \
  -o "$OUT.tmp.$$"
mv -f "$OUT.tmp.$$" "$OUT"
