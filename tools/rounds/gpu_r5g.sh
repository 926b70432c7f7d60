#!/bin/bash
# round 5: the stage-2 loss-contract test (six seeded batches vs the oracle)
mkdir -p gpurun_out; export TMPDIR=/tmp
export ANTMMF_REAL_WIDTH_OUT=$PWD/gpurun_out/r5g_contracts.jsonl
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -q --timeout 600 -k "loss_contract" 2>&1 | tail -3
cat $ANTMMF_REAL_WIDTH_OUT | cut -c1-600
