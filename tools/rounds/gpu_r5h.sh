#!/bin/bash
# round 5: the d-wide LayerNorm forward with two ADJACENT rows per wave (product default) vs one row per wave (lab: ANTMMF_LN_FWD_ADJ=0), the LayerNorm tests, a step bench
TAG=${1:-r5h}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -k "layernorm or real_width_vs_oracle or m2_towers" 2>&1 | tail -3
LAB=$PWD/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so
for a in 0 1 0 1; do echo "--- ANTMMF_LN_FWD_ADJ=$a"; ANTMMF_HIP_LIB=$LAB ANTMMF_LN_FWD_ADJ=$a timeout 300 python tools/ln_bench.py adj$a 2>/dev/null | grep "ln_fwd.*1024" | sed "s/^/adj $a /" | tee -a gpurun_out/${TAG}_ln_fwd_adjacent_rows_ab.jsonl | cut -c1-150; done
echo "=== bench"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_l14.json 2> gpurun_out/${TAG}_bench_l14.err; tail -2 gpurun_out/${TAG}_bench_l14.err; cut -c1-260 gpurun_out/${TAG}_bench_l14.json
