#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, per-function GPU tests in separate processes (a GPU fault in one
# test must not hide the others), then a short bench.  Logs go to gpurun_out/.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
LOG=gpurun_out/check.log
: > $LOG
echo "== rocminfo" >> $LOG; (rocminfo | grep -E "Marketing Name|gfx9" | head -4; nproc; free -g | head -2) >> $LOG 2>&1
echo "== entry" >> $LOG
timeout 600 python __graft_entry__.py >> $LOG 2>&1; echo "entry exit $?" >> $LOG
FUNCS_K="test_fps_bit_exact test_fps_cooperative_bit_exact test_knn_bit_exact test_knn_all_points_identical test_three_nn test_group_gather_and_patch_l1 test_gemm_shapes test_gemm_asymmetric_identity test_gemm_epilogues_and_views test_gemm_bf16x6 test_gemm_f16x3 test_gemm_swiglu_epilogue test_layernorm test_swiglu_ln test_group_max_pos_fourier_addbcast_interp test_flash_attention test_attention_small test_invalid_arguments_raise test_border_farthest_and_error_regions"
for t in $FUNCS_K; do
  echo "== kernels::$t" >> $LOG
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "$t" 2>&1 | tail -40 >> $LOG
done
FUNCS_E="test_evaluation_harness test_forward_eval_protocol test_cfg3_large_scene_tokenizer_and_run test_cfg5_giant_five_click_loop test_batch_pipeline_matches_predict_masks test_against_reference_golden test_against_oracle test_properties_full_size test_predictor_click_loop test_demo_server_segment_route test_out_of_range_coordinates_raise"
for t in $FUNCS_E; do
  echo "== e2e::$t" >> $LOG
  timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s --tb=short -p no:cacheprovider -k "$t" 2>&1 | tail -40 >> $LOG
done
echo "== bench" >> $LOG
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> $LOG
cat gpurun_out/bench.json >> $LOG; tail -5 gpurun_out/bench.err >> $LOG
for p in bf16x6 f32; do timeout 900 python bench.py --precision $p --no-cpu-baseline > gpurun_out/bench_$p.json 2>> gpurun_out/bench.err; cat gpurun_out/bench_$p.json >> $LOG; done
grep -E "passed|failed|error|exit" $LOG | tail -40
