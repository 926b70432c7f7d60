#!/bin/bash
# round 5, third call: the two tests that failed in r5b (stale lab library; dmae12 gates), and what HBM gives a row kernel (tools/stream_probe.hip)
TAG=${1:-r5c}
mkdir -p gpurun_out; export TMPDIR=/tmp
export ANTMMF_REAL_WIDTH_OUT=$PWD/gpurun_out/${TAG}_real_width.jsonl
timeout 900 python -m pytest tests -m gpu -q --timeout 900 -k "tail_round_cells or (real_width and dmae12) or loss_contract or no_switches" 2>&1 | tail -5
echo "=== stream probe"; timeout 300 tools/stream_probe 2>&1 | tee gpurun_out/${TAG}_stream_probe.jsonl
echo "=== stream probe, text tower rows"; timeout 300 tools/stream_probe 78848 2>&1 | tee gpurun_out/${TAG}_stream_probe_text.jsonl | grep -E "copy U=4 grid=8192|ln R"
