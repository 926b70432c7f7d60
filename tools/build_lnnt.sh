#!/bin/bash
# A/B libraries for `tools/gpu_r6.sh lnnt`: the product objects with layernorm.hip recompiled under -DANTMMF_ROW_NT=1 | 2 | 3 (non-temporal loads / stores / both in the row
# kernels' 16-B accesses, csrc/common.h row_ld16 / row_st16) -> lib/libantmmf_hip_rownt{1,2,3}.so.  Needs `make -C csrc` first (the other objects are the product's).
set -e
cd "$(dirname "$0")/../ant-multi-modal-framework_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for v in 1 2 3; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DANTMMF_ROW_NT=$v -c layernorm.hip -o ../lib/obj/layernorm_rownt$v.o 2> /dev/null
  objs=$(ls ../lib/obj/*.o | grep -v "layernorm" | tr '\n' ' ')
  $HIPCC --offload-arch=gfx950 -shared -fPIC $objs ../lib/obj/layernorm_rownt$v.o -o ../lib/libantmmf_hip_rownt$v.so
done
ls -la ../lib/*.so
