#!/bin/bash
# round 5, fourth call: rows per wave of the wave-per-row LayerNorm forward (lab knob ANTMMF_LN_FWD_RPW) on the step's four shapes; the DMAE loss-contract test with its report
TAG=${1:-r5d}
mkdir -p gpurun_out; export TMPDIR=/tmp
export ANTMMF_REAL_WIDTH_OUT=$PWD/gpurun_out/${TAG}_contracts.jsonl
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -k "loss_contract" 2>&1 | tail -3
LAB=$PWD/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so
for r in 1 2 4 1 2; do echo "--- ANTMMF_LN_FWD_RPW=$r"; ANTMMF_HIP_LIB=$LAB ANTMMF_LN_FWD_RPW=$r timeout 300 python tools/ln_bench.py rpw$r 2>/dev/null | grep "ln_fwd\|copy" | tee -a gpurun_out/${TAG}_ln_fwd_rows_per_wave_ab.jsonl | cut -c1-140; done
