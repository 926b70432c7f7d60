#!/bin/bash
for lds in 0 20000 40000 60000; do
  echo "--- dynamic LDS $lds"
  ANTMMF_LN_LDS=$lds timeout 300 python tools/ln_bench.py blk 2>&1 | grep "ln_fwd.*4096.act=gelu\|ln_fwd.image.1024.act=None" | cut -c1-140
  ANTMMF_LN_LDS=$lds ANTMMF_LN_WAVE4096=1 timeout 300 python tools/ln_bench.py wave 2>&1 | grep "ln_fwd.*4096.act=gelu" | cut -c1-140
done | tee gpurun_out/r3r_ln_occupancy.txt
