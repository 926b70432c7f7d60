#!/bin/bash
# TEST INFRASTRUCTURE: builds the CPU lane-emulated copy of the kernels (tests/emu/build/libantmmf_emu.so).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CXX="${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
mkdir -p "$ROOT/tests/emu/build"
SRC="$ROOT/ant-multi-modal-framework_amd/csrc"
"$CXX" -std=c++20 -O1 -pthread -fPIC -shared -DANTMMF_EMULATE -Wno-unused-value -I"$ROOT/tests/emu" -I"$SRC" -x c++ \
  "$SRC/abi.hip" "$SRC/layernorm.hip" "$SRC/elementwise.hip" "$SRC/gemm.hip" "$SRC/attention.hip" "$SRC/loss.hip" "$SRC/resize.hip" \
  -o "$ROOT/tests/emu/build/libantmmf_emu.so"
