#!/bin/bash
# Round 6 (second session): q / k / v bias gradients out of the attention backward's token sums.
#   gpu_r6b.sh sums     attention tests, isolated bench (with / without the sums, the column-sum passes they replace), step A/B on l14 and vtp8
mkdir -p gpurun_out; export TMPDIR=/tmp
case "$1" in
sums)
  python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attention" --timeout 900 2>&1 | tail -5
  python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "layer or m2 or bert or univl_stage1" --timeout 900 2>&1 | tail -5
  python tools/attn_bench.py r6b_sums 10
  ATTN_BENCH_SHAPES="1024x12x197,512x12x197" python tools/attn_bench.py r6b_sums_b16 10
  for rep in 1 2; do
    for fl in 1 0; do
      echo "=== l14 ATTN_BWD_SUMS=$fl (rep $rep)"
      timeout 600 python tools/bench_flag.py ATTN_BWD_SUMS=$fl -- --no-cpu-baseline --steps 8 --warmup 3 2> gpurun_out/r6b_l14_sums$fl.err | tee gpurun_out/r6b_bench_l14_sums${fl}_rep$rep.json | cut -c1-260
    done
  done
  for fl in 1 0; do
    echo "=== rocprofv3 stats, l14 ATTN_BWD_SUMS=$fl"
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6b_prof_sums$fl -o prof -- python $GRAFT_REPO_ROOT/tools/bench_flag.py ATTN_BWD_SUMS=$fl -- --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r6b_prof_sums$fl.log 2>&1)
    f=$(find gpurun_out/r6b_prof_sums$fl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r6b_l14_sums${fl}_kernel_stats.csv && python tools/kernel_families.py "$f" | tee gpurun_out/r6b_l14_sums${fl}_families.txt
    find gpurun_out/r6b_prof_sums$fl -type f ! -name "*stats*" -delete 2>/dev/null
  done
  for fl in 1 0; do
    echo "=== vtp8 ATTN_BWD_SUMS=$fl"
    timeout 600 python tools/bench_flag.py ATTN_BWD_SUMS=$fl -- --workload vtp8 --no-cpu-baseline --steps 5 --warmup 2 2> gpurun_out/r6b_vtp8_sums$fl.err | tee gpurun_out/r6b_bench_vtp8_sums$fl.json | cut -c1-260
  done
  ;;
fwdabl)
  # timing-only ablations of the 257-token attention forward (lab library): what each part costs, same box, same process order A B A B
  export ANTMMF_HIP_LIB=$PWD/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so
  for rep in 1 2; do
    for abl in 0 1 2 4 8 16 32 24 25 38 27 63; do
      ANTMMF_ATTN_FWD_ABL=$abl ATTN_BENCH_SHAPES="1024x16x257" python tools/attn_bench.py fwdabl_$abl 20 2>/dev/null | grep "attention.fwd" | sed "s/^/abl=$abl rep=$rep /"
    done
  done | tee gpurun_out/r6b_attn_fwd_ablations.txt
  ;;
stagger)
  # attention forward with the first-wave stagger (lab knob ANTMMF_ATTN_FWD_STAGGER = number of 4-us sleeps of the upper-LDS-half workgroups), A B A B
  export ANTMMF_HIP_LIB=$PWD/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so
  for rep in 1 2; do
    for st in 0 1 2 3 4 6; do
      ANTMMF_ATTN_FWD_STAGGER=$st ATTN_BENCH_SHAPES="1024x16x257,1024x16x77,1024x12x197" python tools/attn_bench.py stagger_$st 20 2>/dev/null | grep "attention.fwd" | sed "s/^/stagger=$st rep=$rep /"
    done
  done | tee gpurun_out/r6b_attn_fwd_stagger.txt
  ;;
twoout)
  ANTMMF_HIP_LIB=$PWD/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so python tools/gemm_two_output_bench.py 10 2>/dev/null | tee gpurun_out/r6b_gemm_two_output_store_ablation.txt
  ;;
keepln2)
  # EXPERIMENT: keep the second LayerNorm's output for backward (4 instead of 5 streams in its backward kernel, + 2 B per token-channel): step A/B at 896 pairs (room for the 13.7 GiB), A B A B
  for rep in 1 2; do
    for fl in 1 0; do
      echo "=== l14 batch 896 KEEP_LN2_OUT=$fl (rep $rep)"
      timeout 600 python tools/bench_flag.py KEEP_LN2_OUT=$fl -- --batch 896 --no-cpu-baseline --steps 8 --warmup 3 2> gpurun_out/r6b_keepln2_$fl.err | tee gpurun_out/r6b_bench_l14_b896_keepln2_${fl}_rep$rep.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config'].get('peak_hbm_gib'), d['config'].get('reserved_hbm_gib'))"
    done
  done
  for fl in 1 0; do
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6b_prof_keepln2_$fl -o prof -- python $GRAFT_REPO_ROOT/tools/bench_flag.py KEEP_LN2_OUT=$fl -- --batch 896 --no-cpu-baseline --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r6b_prof_keepln2_$fl.log 2>&1)
    f=$(find gpurun_out/r6b_prof_keepln2_$fl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/kernel_families.py "$f" | tee gpurun_out/r6b_l14_b896_keepln2_${fl}_families.txt | head -12
    find gpurun_out/r6b_prof_keepln2_$fl -type f ! -name "*stats*" -delete 2>/dev/null
  done
  ;;
esac
