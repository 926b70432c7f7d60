/* antmmf_hip_lab.h -- what libantmmf_hip_lab.so (`make -C ant-multi-modal-framework_amd/csrc lab`, -DANTMMF_GEMM_ABLATIONS) exports ON TOP of include/antmmf_hip.h.
 * Measurement / experiment surface: the product library has none of it and the product path never loads the lab library
 * (antmmf/hip/_lib.py loads it only when ANTMMF_HIP_LIB names it: tests, tools/gemm_bench, tools/gpu_*.sh).
 *
 *  - int antmmf_debug_set_gemm_variant(int bits)   process-global A/B switch of the GEMM dispatch (same bits as the ANTMMF_GEMM_VARIANT environment variable;
 *                                                  csrc/gemm.hip, the launch site documents them): round-1 ring kernels, burst vs rolling epilogue, store
 *                                                  layouts, tail split on / off, and -- timing only, wrong data by construction -- the store ablations.
 *  - environment: ANTMMF_GEMM_FORCE_TILE / _PERSIST / _PERSIST_WGS / _CONT / _RASTER, ANTMMF_WGRAD_WGS, ANTMMF_ATTN_VARIANT (attn_fwd32_kernel).
 *  - the sub-LN fold below: a complete, parity-tested alternative for the M2 feed-forward that measured neutral (rounds 3 and 4) and is therefore
 *    not part of the product; functional.set_ffn_fold(True) needs this library.
 */
#ifndef ANTMMF_HIP_LAB_H
#define ANTMMF_HIP_LAB_H
#include "antmmf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

int antmmf_debug_set_gemm_variant(int bits);
/* launches of the persistent NT kernel whose tail round ran as cells (tests: "the cells really ran") */
long antmmf_debug_gemm_cell_launches(void);
/* attention backward calls served by the one-kernel path attn_bwd_fused64_kernel (tests: "that kernel really ran") */
long antmmf_debug_attn_fused_launches(void);

/* ---- M2 feed-forward with the sub-LayerNorm folded into its GEMMs.  Reference prj/M2_Encoder/vlmo/torchscale/component/
 * feedforward_network.py:117-128: x -> fc1 -> gelu -> ffn_layernorm (over the 4d-wide row) -> fc2 (+ the residual of encoder.py:176-199).
 * With z = act(fc1(x)), (mu_i, rstd_i) the row statistics of z, W2g[j][k] = bf16(W2[j][k] gamma_k), c_j = sum_k W2g[j][k], b2f = b2 + W2 beta:
 *   y_ij = rstd_i (z W2g^T)_ij - rstd_i mu_i c_j + b2f_j + res_ij        -- exactly fc2(LayerNorm(z)) + res; no 4d-wide LayerNorm pass
 * and in backward the LayerNorm's two row means are dot products over d-wide tensors (antmmf_ffn_bwd_rows), so the 4d-wide tensors are touched by
 * GEMM epilogues only.  All matrices bf16 row-major with row strides in elements (multiples of 8); statistics / partial sums fp32.
 * workspace: device scratch for the per-tile partial sums of the large-shape kernel (fc1: n_ff / 64 * tokens * 8 bytes; dgrad: tokens / 128 * n_ff * 4);
 * NULL or too small -> the statistics / column sums are taken by a separate small pass over the stored output instead. */
/* prepare (once per optimizer step): W2 fp32 [n_out][n_ff] -> W2g bf16, c [n_out], b2f [n_out] (b2 nullable) */
int antmmf_ffn_prepare_w2(const float* W2, const float* gamma, const float* beta, const float* b2, void* W2g, float* c, float* b2f, int n_out,
                          int n_ff, antmmf_stream_t stream);
/* Z = act(X W1^T + b1), DACT = act'(X W1^T + b1) (same row stride ldz), stats[i] = (mean, 1/sqrt(var + eps)) of the ROUNDED row Z[i] */
int antmmf_ffn_fc1_fwd(const void* X, const void* W1, const float* b1, void* Z, void* DACT, float* stats, int tokens, int n_ff, int n_in,
                       int64_t ldx, int64_t ldw, int64_t ldz, int act, float eps, float* workspace, int64_t workspace_bytes, antmmf_stream_t stream);
int antmmf_ffn_fc2_fwd(const void* Z, const void* W2g, const float* colsum_w2g, const float* b2f, const float* stats, const void* RES, void* Y,
                       int tokens, int n_out, int n_ff, int64_t ldz, int64_t ldw, int64_t ldres, int64_t ldy, antmmf_stream_t stream);
/* backward row pass: rowv4[i] = (mu, rstd, m1, m2) with m1 = dy_i . c / n_ff, m2 = dy_i . (y_i - b2f - res_i) / n_ff;  dYs = bf16(rstd_i dy_i) (fc2's
 * wgrad operand);  s_col[j] += sum_i rstd_i mu_i dy_ij;  cs_col[j] += sum_i dy_ij (nullable).  n_out <= 2048. */
int antmmf_ffn_bwd_rows(const void* dY, const void* Y, const void* RES, const float* b2f, const float* colsum_w2g, const float* stats, float* rowv4,
                        void* dYs, float* s_col, float* cs_col, int tokens, int n_out, int n_ff, int64_t lddy, int64_t ldy, int64_t ldres,
                        int64_t lddys, antmmf_stream_t stream);
/* dU = DACT * (rstd_i (dY W2gT^T - m1_i) - zhat_ik rstd_i m2_i), zhat = (Z - mu_i) rstd_i;  W2gT = W2g transposed [n_ff][n_out];  db1[k] += sum_i dU_ik (nullable) */
int antmmf_ffn_fc2_dgrad(const void* dY, const void* W2gT, const void* Z, const void* DACT, const float* rowv4, void* dU, float* db1, int tokens,
                         int n_ff, int n_out, int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddu, float* workspace, int64_t workspace_bytes,
                         antmmf_stream_t stream);
/* Gm = dYs^T Z (fp32 [n_out][n_ff], from antmmf_gemm_wgrad_bf16 into a zeroed buffer):  dW2 += gamma_k (Gm - s_j) + beta_k cs_j;
 * dgamma_k += sum_j W2[j][k] (Gm[j][k] - s_j);  dbeta_k += sum_j W2[j][k] cs_j   (dgamma / dbeta nullable) */
int antmmf_ffn_wgrad_post(const float* Gm, const float* W2, const float* gamma, const float* beta, const float* s, const float* cs, float* dW2,
                          float* dgamma, float* dbeta, int n_out, int n_ff, antmmf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
