#!/bin/bash
# Round-6 GPU experiments, one sub-command per question (run through gpurun from the repo root):
#   overlap   two CU-masked streams: a GEMM of the step on G CUs beside a row / attention kernel on the other 256 - G (tools/overlap_probe.py)
#   sustain   K-loop forms judged on sustained wall-clock TF + clock: the harness loops (tools/gemm_ablate sustain) and the product loop with / without stores (tools/gemm_sustain.py)
#   lnnt      the step with non-temporal loads / stores in the row kernels (A/B libraries built by `tools/build_lnnt.sh`), bench + rocprofv3 kernel stats each
#   step      default bench + rocprofv3 kernel stats of the same command (the per-round record)
# usage: gpu_r6.sh <sub-command> [TAG] [args]
CMD=$1; TAG=${2:-r6}
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
LIBDIR=$ROOT/ant-multi-modal-framework_amd/lib
LAB=$LIBDIR/libantmmf_hip_lab.so

prof_bench() {   # prof_bench NAME [env assignments...] : rocprofv3 kernel stats of the default bench command -> gpurun_out/${TAG}_${NAME}_kernel_stats.csv
  local name=$1; shift
  (cd /tmp && env "$@" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_prof_$name -o prof -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_prof_$name.log 2>&1)
  local f=$(find gpurun_out/${TAG}_prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_${name}_kernel_stats.csv
  local t=$(find gpurun_out/${TAG}_prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python tools/kernel_trace_split.py "$t" > gpurun_out/${TAG}_${name}_row_kernels_by_grid.txt
  [ -n "$t" ] && python tools/kernel_trace_split.py "$t" --seq "ln_bwd_kernel<unsigned short, 2, false, 0, true, true>" > gpurun_out/${TAG}_${name}_renorm_bwd_neighbours.txt
  find gpurun_out/${TAG}_prof_$name -type f ! -name "*stats*" -delete 2>/dev/null
  python tools/kernel_families.py gpurun_out/${TAG}_${name}_kernel_stats.csv 8 | tee gpurun_out/${TAG}_${name}_families.txt
}

case $CMD in
overlap)
  for G in ${3:-256 224 192 160 128}; do
    echo "=== G=$G"
    R=$((256-G)); [ $R -lt 8 ] && R=256
    if [ $G -eq 256 ]; then
      ANTMMF_HIP_LIB=$LAB timeout 600 python tools/overlap_probe.py $G $TAG 2>&1 | grep '^{' | cut -c1-420
    else
      ANTMMF_HIP_LIB=$LAB ANTMMF_GEMM_PERSIST_WGS=$G ANTMMF_WGRAD_WGS=$G ANTMMF_ATTN_PERSIST_WGS=$R ANTMMF_ROW_CUS=$R timeout 600 python tools/overlap_probe.py $G $TAG 2>&1 | grep '^{\|Error\|error' | cut -c1-420
    fi
  done ;;
sustain)
  SECS=${3:-10}
  timeout 600 tools/gemm_ablate sustain $SECS | tee gpurun_out/${TAG}_gemm_sustain_harness.jsonl | grep -v '"t_s"'
  ANTMMF_HIP_LIB=$LAB timeout 600 python tools/gemm_sustain.py $SECS $TAG | grep -v '"t_s"' ;;
lnnt)
  for v in base 1 2 3; do
    lib=$LIBDIR/libantmmf_hip.so; [ $v != base ] && lib=$LIBDIR/libantmmf_hip_rownt$v.so
    echo "=== row kernels nt bits = $v ($lib)"
    ANTMMF_HIP_LIB=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_l14_rownt_$v.json 2> gpurun_out/${TAG}_bench_l14_rownt_$v.err
    python -c "import json,sys; d=json.loads(open('gpurun_out/${TAG}_bench_l14_rownt_$v.json').read().strip().splitlines()[-1]); print('pairs/s', d['value'], 'ms', d['ms_per_step'], 'gemm ms', d['roofline']['gemm_ms_per_step'], 'clock', d['roofline']['gemm_clock_mhz'])"
    prof_bench l14_rownt_$v ANTMMF_HIP_LIB=$lib
  done ;;
step)
  timeout 900 python bench.py --gemm-table gpurun_out/${TAG}_gemm_table_l14.txt > gpurun_out/${TAG}_bench_l14.json 2> gpurun_out/${TAG}_bench_l14.err; tail -2 gpurun_out/${TAG}_bench_l14.err; cat gpurun_out/${TAG}_bench_l14.json
  prof_bench bench_l14 ;;
lnbench)   # isolated row kernels: product library against the non-temporal-load build
  for v in base 1; do
    lib=$LIBDIR/libantmmf_hip.so; [ $v != base ] && lib=$LIBDIR/libantmmf_hip_rownt$v.so
    ANTMMF_HIP_LIB=$lib timeout 300 python tools/ln_bench.py rownt_$v 2>&1 | grep '^{' | cut -c1-200
  done | tee gpurun_out/${TAG}_ln_bench_rownt_ab.jsonl ;;
rowpmc)    # HBM-side bytes of the row / attention kernels inside the step and in an isolated loop (separate --pmc passes, kernel trace only)
  for what in step loop; do
    P=$ROOT/gpurun_out/${TAG}_rowpmc_$what; mkdir -p $P
    for c in FETCH_SIZE WRITE_SIZE; do
      if [ $what = step ]; then cmd="python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline"; else cmd="python $ROOT/tools/ln_bench.py pmc"; fi
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/$c -o p -- $cmd > $P.$c.log 2>&1); echo "$what $c rc=$?"
    done
    python tools/row_pmc.py $P | tee gpurun_out/${TAG}_row_kernels_pmc_$what.txt
    find $P -type f -delete 2>/dev/null
  done ;;
dmaehead)  # level-3 loss deviation split into the head's and the towers' part at real width, temporal transformer on the fused bf16 layer vs the fp32 stream
  for mode in 1 0; do
    for c in dmae12 vtp8t dmae12tpm; do
      echo "--- $c ANTMMF_DMAE_BF16_STREAM=$mode"
      ANTMMF_DMAE_BF16_STREAM=$mode ANTMMF_REAL_WIDTH_REPORT_ONLY=1 timeout 900 python tests/real_width_case.py $c cuda:0 2>/dev/null | grep REALWIDTH | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[len('REALWIDTH '):]); d['bf16_stream'] = $mode
    print(json.dumps({k: d[k] for k in ('case', 'bf16_stream', 'loss1_rel', 'loss3_rel', 'loss3_head_part', 'loss3_tower_part', 'l3_simi_head_part_max_abs', 'l3_simi_max_abs', 'l3_simi_ref_absmax', 'margin', 'failed_gates')}), 'global_cos', d['grads']['global_cos'])
"
    done
  done | tee gpurun_out/${TAG}_dmae_head_tower_split.txt ;;
asserts)   # does a device-side assert fire on this wheel?  (contrastive._assert_equal_batch relies on torch._assert_async)
  timeout 120 python -c "
import torch
try:
    torch._assert_async(torch.zeros((), device='cuda').bool(), 'probe')
    torch.cuda.synchronize()
    print('torch._assert_async(False) did NOT fire on this wheel')
except Exception as e:
    print('torch._assert_async(False) fired:', type(e).__name__, str(e)[:200])
" 2>&1 | tail -3 | tee gpurun_out/${TAG}_assert_async_probe.txt ;;
*) echo "unknown sub-command $CMD"; exit 2 ;;
esac
