/* antmmf_hip.h -- C ABI of libantmmf_hip.so: the MI355X (gfx950) kernels behind AntMMF's
 * contrastive image/video-text training step.
 *
 * The reference (alipay/Ant-Multi-Modal-Framework) has no FFI and no native kernels: every operator
 * on this path is a stock torch op.  Its plugin surface is the Python registries
 * (antmmf.modules / antmmf.models) plus the op-replacement seam antmmf/utils/optim_utils.py:24-33,59-93
 * (`replace_speedup_op`).  This header is the boundary a maintainer binds from that seam; each entry
 * names the reference arithmetic it replaces (file:line, paths relative to the reference checkout).
 * INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *  - the caller owns every buffer (device memory; kernels never allocate, free or synchronise);
 *  - row-major; "ld*" are row strides in ELEMENTS; all pointers 16-byte aligned;
 *  - dtype tags: ANTMMF_F32 = 0, ANTMMF_BF16 = 1 (bf16 = upper 16 bits of an IEEE fp32, RNE);
 *  - every function enqueues on `stream` (a hipStream_t; pass torch's current stream) and returns
 *    0, or a negative errno-style code (-22 bad argument, -5 launch failure); nothing throws;
 *  - re-entrant; one process per GPU.  The entry points declared here keep no state between calls (beyond per-kernel launch attributes set once),
 *    and libantmmf_hip.so reads NO environment variable and has no dispatch switch: every call takes the one measured-best kernel for its shape.
 *    Two read-only probes are exported and deliberately not declared here (antmmf_debug_gemm_k64_launches: a launch counter the tests read;
 *    antmmf_debug_gemm_clock: shader / wall clock ticks of the last persistent-GEMM launch, read by bench.py) -- measurement, not ABI.
 *    The A/B switches (ANTMMF_GEMM_VARIANT, antmmf_debug_set_gemm_variant, forced tile sizes), the timing-only ablation kernels, the experiment kernels that
 *    measured slower and the optional sub-LN fold live in a SEPARATE library, libantmmf_hip_lab.so (`make lab`; include/antmmf_hip_lab.h): tests and tools
 *    load it explicitly, the product path never does.
 */
#ifndef ANTMMF_HIP_H
#define ANTMMF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANTMMF_F32 0
#define ANTMMF_BF16 1
#define ANTMMF_ACT_NONE 0
#define ANTMMF_ACT_GELU_ERF 1   /* modeling_bert.py:31-37; torchscale feedforward_network.py:80-86,120 */
#define ANTMMF_ACT_QUICK_GELU 2 /* clip/model.py:222-224 */
#define ANTMMF_ACT_RELU 3
/* OR-ed into the `act` argument of antmmf_gemm_bf16 (the activation id is act & 0xff):
 *   AUX_GRAD:  `aux` receives act'(pre-activation) instead of the pre-activation (forward of an FFN whose activation output is kept);
 *   GATE_GRAD: `gate` already holds act'(...) -- the epilogue multiplies by it as is (the matching dgrad: no transcendental in the epilogue). */
#define ANTMMF_ACT_AUX_GRAD 0x100
#define ANTMMF_ACT_GATE_GRAD 0x200

typedef void* antmmf_stream_t; /* hipStream_t */

/* 1 = real gfx950 library, 0 = the CPU lane emulator used by tests/emu. */
int antmmf_backend(void);
/* ABI version of this header (bumped on any signature change). */
int antmmf_abi_version(void);

/* ---- LayerNorm: y = (x - mean) * rstd * gamma + beta, statistics in fp32.
 * Replaces clip/model.py:213-219 (fp32-upcast LN), modeling_bert.py:63 (eps 1e-12),
 * torchscale LayerNorm (encoder.py:34,77; multihead_attention.py:51-55; feedforward_network.py:109).
 * cols % 8 == 0, cols <= 4096.  mean/rstd ([rows] fp32) may be NULL in fwd. */
int antmmf_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         int64_t rows, int cols, float eps, int dtype, antmmf_stream_t stream);
/* dx = LN'(dy) (+ dres if non-NULL); dgamma/dbeta (fp32, may be NULL) are ACCUMULATED (+=). */
int antmmf_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                         const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int cols, int dtype,
                         antmmf_stream_t stream);

/* ---- activation fused in front of the LayerNorm: y = LN(act(x)); bwd returns d/dx through both.  Replaces the
 * torchscale FFN's gelu -> ffn_layernorm pair (feedforward_network.py:117-128).  act = ANTMMF_ACT_*. */
int antmmf_act_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                             int64_t rows, int cols, float eps, int act, int dtype, antmmf_stream_t stream);
/* `dxsum` (nullable, fp32 [cols], accumulated): column sums of the returned dx = the bias gradient of the Linear that
 * produced x (fc1 for the fused gelu + ffn_layernorm pair; the attention out-projection for the LayerNorm behind it, with
 * the residual gradient in `dres`) -- saves a separate full read of dx.
 * `partials` (nullable fp32 scratch of >= 1024 * 3 * cols elements) lets the wide-row variant (cols > 1024) write
 * per-workgroup column sums and reduce them in a second launch instead of issuing ~8 M fp32 atomics per call. */
int antmmf_act_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                             const void* dres, void* dx, float* dgamma, float* dbeta, float* dxsum, int64_t rows, int cols,
                             int act, int dtype, float* partials, int64_t partial_elems, antmmf_stream_t stream);
/* Plain LayerNorm backward that also re-emits y = LN(x) (same rounding as antmmf_layernorm_fwd): the backward pass of a transformer
 * layer needs LN(x) again as the wgrad operand of the Linear behind it (encoder.py:34,77 recomputes nothing -- torch keeps every
 * LayerNorm output alive; this build keeps none and gets it back for one extra write). */
int antmmf_layernorm_bwd_renorm(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, const void* dres, void* dx, void* y, float* dgamma, float* dbeta, float* dxsum,
                                int64_t rows, int cols, int dtype, antmmf_stream_t stream);

/* ---- activations (n % 8 == 0): g = act(u);  du = dg * act'(u). */
int antmmf_act_fwd(const void* u, void* g, int64_t n, int act, int dtype, antmmf_stream_t stream);
int antmmf_act_bwd(const void* dg, const void* u, void* du, int64_t n, int act, int dtype, antmmf_stream_t stream);

/* ---- row L2 normalise: y = x / max(||x||, eps).  Replaces F.normalize (univl_video_base.py:114,158)
 * and x / x.norm() (vlmo_module.py:346-349; eps = 0).  inv_norm [rows] fp32 is saved for bwd. */
int antmmf_l2norm_fwd(const void* x, void* y, float* inv_norm, int64_t rows, int cols, float eps, int in_dtype,
                      int out_dtype, antmmf_stream_t stream);
int antmmf_l2norm_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, int64_t rows, int cols,
                      int in_dtype, int out_dtype, antmmf_stream_t stream);

/* ---- out[c] += sum_r x[r][c]   (bias / positional-embedding gradients; cols % 8 == 0, ld % 8 == 0). */
int antmmf_colsum(const void* x, float* out, int64_t rows, int cols, int64_t ld, int dtype, antmmf_stream_t stream);
/* ---- out[c][r] = in[r][c] for a bf16 matrix. */
int antmmf_transpose_bf16(const void* in, void* out, int rows, int cols, antmmf_stream_t stream);
/* All transposed weight copies of a step in one launch.  table (device memory): 5 int64 per matrix = element offset into in_base, element
 * offset into out_base, rows, cols, index of the matrix's first 64 x 64 tile; total_tiles = sum of ceil(rows/64) * ceil(cols/64). */
int antmmf_transpose_bf16_batched(const void* in_base, void* out_base, const int64_t* table, int n_mats, int64_t total_tiles,
                                  antmmf_stream_t stream);
/* ---- flat fp32 -> bf16 cast. */
int antmmf_cast_f32_bf16(const float* in, void* out, int64_t n, antmmf_stream_t stream);
/* ---- hi / lo split of an fp32 GEMM operand: hi = bf16(x), lo = bf16(x - float(hi)), both [rows_pad, cols_pad] dense and zero outside x [rows, cols] (row stride ldx);
 * cols_pad % 8 == 0.  hi.hi + hi.lo + lo.hi on the bf16 MFMA GEMM reproduces the fp32 product of the reference's similarity matmuls (univl_video_ret.py:357-387,
 * dmae_utils.py:85-131) to fp32 accuracy; this entry makes the split one pass (bit-identical to cast / subtract / cast). */
int antmmf_split_hi_lo_bf16(const float* x, int64_t ldx, int rows, int cols, void* hi, void* lo, int rows_pad, int cols_pad, antmmf_stream_t stream);

/* ---- patch extraction for a stride == kernel conv (nn.Conv2d(3, W, P, P): clip/model.py:289-295,310-312;
 * VisionEmbedding.proj: torchscale/component/embedding.py:49,69), with the optional input affine
 * (x - shift) * scale (M2 img_norm, transforms/utils.py:48).  img [B,C,H,W] (dtype) -> out [B*(H/P)*(W/P), kpad]
 * bf16, inner order (c, py, px); columns >= C*P*P are zero (kpad % 8 == 0). */
int antmmf_patchify(const void* img, void* out, int B, int C, int H, int W, int P, int kpad, float shift, float scale,
                    int dtype, antmmf_stream_t stream);
/* ---- x[b,0] = cls + pos[0];  x[b,1+p] = patch[b,p] (+ bias) + pos[1+p]   (clip/model.py:313-323; embedding.py:80-83,92-110).
 * patch [B*G, d] bf16, cls [d] / pos [(G+1), d] (nullable) / bias [d] (nullable) fp32, out [B, G+1, d] bf16. */
int antmmf_assemble_tokens(const void* patch, const float* cls, const float* pos, const float* bias, void* out,
                           int64_t B, int G, int d, antmmf_stream_t stream);
/* ---- dpatch[b,p] = dx[b,1+p]  (backward of the assembly; bf16). */
int antmmf_split_tokens(const void* dx, void* dpatch, int64_t B, int G, int d, antmmf_stream_t stream);

/* ---- embedding lookup: out[r] = word[ids[r]] (+ pos[pos_offset + r % seq]) (+ type[type_ids ? type_ids[r] : 0]),
 * rows flagged in zero_rows are zeroed (torchscale encoder.py:440).  Tables fp32, out bf16, d % 8 == 0.
 * Replaces BertEmbeddings (clip_text_encoder.py:36-60) and TextEmbedding/PositionalEmbedding (embedding.py:86-110). */
int antmmf_embed_gather(const int64_t* ids, const float* word, const float* pos, const float* type,
                        const int64_t* type_ids, const unsigned char* zero_rows, void* out, int64_t rows, int seq,
                        int d, int pos_offset, antmmf_stream_t stream);
/* ---- dtable[idx ? idx[r] : offset + r % seq] += dx[r]  (fp32 atomics; rows flagged in skip_rows are skipped). */
int antmmf_embed_scatter_add(const void* dx, const int64_t* idx, const unsigned char* skip_rows, float* dtable,
                             int64_t rows, int seq, int d, int offset, antmmf_stream_t stream);
/* The word-table gradient without atomics: sorted_idx = the token ids in ascending order (rows to skip carry an id >= n_table), src_row = the row of dx each entry came
 * from (the sort's permutation).  One wave per run of equal ids sums the run and adds it to its table row: a single writer per row. */
int antmmf_embed_scatter_add_sorted(const void* dx, const int64_t* sorted_idx, const int64_t* src_row, float* dtable, int64_t rows, int64_t n_table, int d,
                                    antmmf_stream_t stream);

/* ---- fused AdamW over a flat fp32 arena segment (torch.optim.AdamW semantics, decoupled decay);
 * g is multiplied by grad_scale first; `shadow` (bf16 compute copy, nullable) is rewritten. */
int antmmf_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int step, float grad_scale,
                      antmmf_stream_t stream);
/* The same step with one more gradient factor read from DEVICE memory (`dev_scale`, one fp32, nullable): the clipping coefficient
 * min(1, max_norm / (norm + 1e-6)) of antmmf/utils/general.py:47-75 is computed on the device and never visits the host. */
int antmmf_adamw_step_scaled(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float grad_scale,
                             const float* dev_scale, antmmf_stream_t stream);
/* ---- *out += sum x[i]^2  (gradient-norm clipping, antmmf/utils/general.py:47-56). */
int antmmf_sumsq(const float* x, float* out, int64_t n, antmmf_stream_t stream);

/* ---- bf16 MFMA GEMM:  C[i][j] = epi( alpha * sum_r P[i][r] Q[j][r] ),  i < I, j < J, r < R.
 * P is [I][R] (p_rmajor = 0) or [R][I] (p_rmajor = 1); Q likewise.  Supported: (0,0) forward Y = X W^T,
 * (0,1) dgrad dX = dY W, (1,1) wgrad dW = dY^T X.  epi: + bias[j] (fp32) -> store aux (pre-activation,
 * bf16) -> act, or * act'(gate[i][j]) when gate is given -> + residual[i][j] (bf16) -> store C
 * (bf16 or fp32; accumulate = 1 adds into an fp32 C; split_k > 1 needs fp32 + accumulate and no epilogue).
 * J % 4 == 0, ld* % 8 == 0 (ldc/ldr/ldaux/ldgate % 4 == 0), R % 8 == 0 for an r-contiguous operand.
 * Replaces the cuBLAS GEMMs behind nn.MultiheadAttention in/out proj (clip/model.py:231,251), BERT
 * query/key/value/dense (modeling_bert.py:120-122,182-186,221-238), M2 q/k/v/out_proj + fc1/fc2
 * (multihead_attention.py:46-50,91-93; feedforward_network.py:117-128), the patch-embed conv, the
 * projection heads (clip/model.py:330-333; heads.py:17-24) and the similarity matrix
 * (univl_video_ret.py:208-213; clip/model.py:442-444). */
int antmmf_gemm_bf16(const void* P, const void* Q, void* C, int I, int J, int R, int64_t ldp, int64_t ldq,
                     int64_t ldc, int p_rmajor, int q_rmajor, int c_dtype, float alpha, const float* bias, int act,
                     const void* residual, int64_t ldr, void* aux, int64_t ldaux, const void* gate, int64_t ldgate,
                     int accumulate, int split_k, antmmf_stream_t stream);
/* The dgrad through an activation whose derivative the forward stored, C = (P Q^T) * gate (antmmf_gemm_bf16 with gate + ANTMMF_ACT_GATE_GRAD), that also returns the column
 * sums of C as 256 partial rows: colsum_part [256][J] fp32, ZERO-FILLED by the caller; the bias gradient of the Linear in front of the activation -- c_fc of CLIP's MLP
 * (clip/model.py:222-241), BertIntermediate.dense (clip/modeling_bert.py:210-238): db = column sums of dU = (dY W2) * act' -- is the column sum of those 256 rows
 * (antmmf_colsum) instead of autograd's sum over the I rows of the 4d-wide dU.  Served by the rolling-epilogue kernel only (I, J multiples of 256, J <= 4096, R a multiple
 * of 64 >= 192, >= 512 tiles): antmmf_gemm_bf16_gated_colsum_ok returns 1 for those, the call ANTMMF_EINVAL otherwise. */
int antmmf_gemm_bf16_gated_colsum_ok(int I, int J, int R, int64_t ldc, int64_t ldgate);
int antmmf_gemm_bf16_gated_colsum(const void* P, const void* Q, void* C, int I, int J, int R, int64_t ldp, int64_t ldq, int64_t ldc, const void* gate, int64_t ldgate,
                                  float* colsum_part, antmmf_stream_t stream);
/* The same GEMM with a caller-owned DEVICE scratch buffer (fp32): used by the r-major / r-major layout (the weight-gradient kernels' token split: partial tiles ->
 * workspace, summed in split order by a second launch; NULL -> fp32 atomics).  The all-r-contiguous layouts need none: the persistent kernel finishes the leftover
 * tiles of its tile walk inside the same launch (round 4's K-split through this scratch is gone).  NULL / 0 is always valid. */
int antmmf_gemm_bf16_ws(const void* P, const void* Q, void* C, int I, int J, int R, int64_t ldp, int64_t ldq,
                        int64_t ldc, int p_rmajor, int q_rmajor, int c_dtype, float alpha, const float* bias, int act,
                        const void* residual, int64_t ldr, void* aux, int64_t ldaux, const void* gate, int64_t ldgate,
                        int accumulate, int split_k, float* workspace, int64_t workspace_bytes, antmmf_stream_t stream);

/* ---- wgrad: dW[n_out][k_in] += dY[tokens][n_out]^T X[tokens][k_in] (fp32 accumulate; both operands token-major bf16).
 * Large 256-aligned problems run a 4-stage LDS-DMA ring with hardware transpose reads, split over the token range;
 * the per-split partial sums go to the caller-owned fp32 `workspace` (>= 32 * n_out * k_in * 4 bytes always suffices;
 * NULL -> fp32 atomics) and are reduced into dW by a second launch on the same stream.  Replaces the cuBLAS wgrad
 * GEMMs autograd issues for every nn.Linear / in_proj on the path. */
int antmmf_gemm_wgrad_bf16(const void* dY, const void* X, float* dW, int64_t tokens, int n_out, int k_in, int64_t ld_dy,
                           int64_t ld_x, int64_t ld_dw, int split_k_hint, float* workspace, int64_t workspace_bytes,
                           antmmf_stream_t stream);

/* The same for n_seg (<= 4) weights of seg_rows x k_in whose gradient buffers dW[0 .. n_seg) are unrelated addresses (host array of device pointers) while their dY
 * columns are adjacent, dY[tokens][n_seg * seg_rows]: the separate q / k / v projections of BertSelfAttention (modeling_bert.py:140-146) and of torchscale's
 * MultiheadAttention (multihead_attention.py:66-71), whose dQ | dK | dV this build keeps packed -- one wgrad GEMM instead of three, the row segments scattered by its
 * reduce launch.  Shapes that do not take the workspace path run as n_seg ordinary calls; same workspace rule as above with n_out = n_seg * seg_rows. */
int antmmf_gemm_wgrad_bf16_seg(const void* dY, const void* X, float* const* dW, int n_seg, int seg_rows, int64_t tokens, int k_in,
                               int64_t ld_dy, int64_t ld_x, int64_t ld_dw, int split_k_hint, float* workspace, int64_t workspace_bytes,
                               antmmf_stream_t stream);

/* ---- fused multi-head attention, head_dim = 64, bf16, Nk <= 288 (whole key row in LDS):
 *   O[b,q,h,:] = softmax_k( scale * <Q[b,q,h,:], K[b,k,h,:]> + key_bias[b,k] ) V[b,k,h,:]
 * element (b, n, h, e) of Q lives at q + (b*Nq + n)*ldq + h*64 + e (K, V with Nk / ldk / ldv; O with ldo), so a packed
 * [B, N, 3, heads, 64] projection output is addressed with ld = 3*heads*64 and no copy.  key_bias [B, Nk] fp32 (nullable):
 * BERT's additive -10000 (modeling_bert.py:144-161; clip_text_encoder.py:97-100), torchscale's -inf key padding
 * (multihead_attention.py:130-142); CLIP's ViT passes none (clip/model.py:245-251).  Co-attention (vilbert.py:360-400)
 * is two calls with the streams swapped.  lse [B, heads, Nq] fp32 is saved for the backward. */
int antmmf_attention_fwd(const void* q, const void* k, const void* v, const float* key_bias, void* o, float* lse,
                         int B, int heads, int Nq, int Nk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                         float scale, float dropout_p, uint64_t dropout_seed, antmmf_stream_t stream);
/* dq/dk/dv use the same addressing as q/k/v (lddq, lddk, lddv); `o` and `lse` are the forward outputs.  Bit-reproducible (no atomics on any path).  Which kernels run is
 * the library's business: head size 64 with 33 ... 272 keys and no dropout goes to one persistent kernel (8 tensor passes over HBM), everything else to a dQ and a
 * dK / dV kernel (13 passes); the results agree to the rounding of the bf16 outputs.  q, k, v, d_o, o must be 16-byte aligned with row strides that are multiples of 8
 * elements, dq, dk, dv 8-byte aligned with strides that are multiples of 4 (checked on entry: ANTMMF_EINVAL otherwise). */
int antmmf_attention_bwd(const void* q, const void* k, const void* v, const float* key_bias, const void* o,
                         const float* lse, const void* d_o, void* dq, void* dk, void* dv, int B, int heads, int Nq,
                         int Nk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq,
                         int64_t lddk, int64_t lddv, float scale, float dropout_p, uint64_t dropout_seed,
                         antmmf_stream_t stream);
/* Key importance (head size 64): out[b][k] += weight * sum over heads and queries of the (post-dropout) attention probabilities, rebuilt from q, k and the
 * forward's lse -- the reduction the reference applies to the attention MAPS of `output_attentions=True` (words_importance,
 * prj/base_vtp/roi_univl/univl/model/univl_video_base.py:131-143; maps from modeling_bert.py:144-172).  out [B, Nk] fp32 is accumulated into (atomics). */
int antmmf_attention_key_importance(const void* q, const void* k, const float* key_bias, const float* lse, float* out, int B, int heads,
                                    int Nq, int Nk, int64_t ldq, int64_t ldk, float scale, float dropout_p, uint64_t dropout_seed,
                                    float weight, antmmf_stream_t stream);
/* The same pair with an explicit head size: 64 or 128 (ViLBERT's bi_hidden_size 1024 / 8 heads, antmmf/models/vilbert.py:326-339); element
 * (b, n, h, e) lives at base + (b*N + n)*ld + h*head_dim + e.  The two entry points above are head_dim = 64. */
int antmmf_attention_fwd_hd(const void* q, const void* k, const void* v, const float* key_bias, void* o, float* lse,
                            int B, int heads, int head_dim, int Nq, int Nk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                            float scale, float dropout_p, uint64_t dropout_seed, antmmf_stream_t stream);
int antmmf_attention_bwd_hd(const void* q, const void* k, const void* v, const float* key_bias, const void* o,
                            const float* lse, const void* d_o, void* dq, void* dk, void* dv, int B, int heads, int head_dim, int Nq,
                            int Nk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq,
                            int64_t lddk, int64_t lddv, float scale, float dropout_p, uint64_t dropout_seed,
                            antmmf_stream_t stream);

/* antmmf_attention_bwd (head size 64, no dropout) that also returns, per batch item, the sums over the tokens of dQ | dK | dV (fp32, of the unrounded gradients):
 *   sums[(b * 3 + {0: q, 1: k, 2: v}) * heads * 64 + h * 64 + e] = sum_n dX[b, n, h, e]            ([B, 3 * heads * 64], caller-owned; every element is written,
 *   the dV third only with want_dv != 0: torchscale's value-bias gradient comes out of the inner LayerNorm's backward, sum_k dV[k] = sum_q dO[q] per head)
 * The bias gradients of the q / k / v projections (nn.MultiheadAttention in_proj_bias clip/model.py:222-251; BertSelfAttention query / key / value
 * modeling_bert.py:140-146; torchscale q_proj / k_proj / v_proj multihead_attention.py:66-71) are the column sums of that B-row matrix -- autograd's sum over the
 * B * N rows of dQ | dK | dV (one pass over each tensor per layer) is not needed any more.  Served by the one-kernel backward only: antmmf_attention_bwd_sums_ok
 * returns 1 for the shapes it takes (head size 64, no dropout, 33 ... 272 keys), antmmf_attention_bwd_sums returns ANTMMF_EINVAL for any other --
 * the caller then runs antmmf_attention_bwd and sums the columns itself (antmmf_colsum). */
int antmmf_attention_bwd_sums_ok(int head_dim, int Nq, int Nk, float dropout_p);
int antmmf_attention_bwd_sums(const void* q, const void* k, const void* v, const float* key_bias, const void* o,
                              const float* lse, const void* d_o, void* dq, void* dk, void* dv, float* sums, int want_dv, int B, int heads,
                              int Nq, int Nk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq,
                              int64_t lddk, int64_t lddv, float scale, antmmf_stream_t stream);

/* dropout_p > 0: attention-probability dropout (BertSelfAttention, modeling_bert.py:157) with a counter-based mask
 * keep(seed, ((b * heads + h) * Nq + q) * Nk + k) that forward and backward regenerate identically (nothing is stored);
 * pass the same (dropout_p, dropout_seed) to both calls. */

/* ---- row-sharded MIL-NCE (get_mil_nce_loss, univl_video_ret.py:146-197) on fp32 similarity slabs:
 *   Rm[i][c] = <text_i, clip_c> (c over all Wr = B_g*n clips),  Cm[i][t] = <centre clip of video_i, text_t> (Wc = B_g),
 *   denom_i = LSE({log n + Cm[i][t]} U {Rm[i][c] : c / n != gi}),  loss_rows[i] = denom_i - (log n + Cm[i][gi]),
 *   gi = row_offset + i. */
int antmmf_milnce_fwd(const float* Rm, const float* Cm, int B, int Wr, int Wc, int n_pair, int row_offset,
                      float* loss_rows, float* denom, antmmf_stream_t stream);
/* dRm / dCm (out_dtype) for upstream per-row coefficients coef[i] = d loss / d loss_rows[i]. */
int antmmf_milnce_bwd(const float* Rm, const float* Cm, const float* denom, const float* coef, int B, int Wr, int Wc,
                      int n_pair, int row_offset, void* dRm, void* dCm, int out_dtype, antmmf_stream_t stream);
/* ---- softmax cross-entropy of s * x[i][:] against column row_offset + i, s = scale_mul * (log_scale ? exp(*log_scale) : 1).
 * InfoNCE over logit_scale.exp() * img @ txt.T (m2_encoder.py:92-95; clip/model.py:442-444); DMAE CrossEn (dmae_utils.py:528-537). */
int antmmf_softmax_ce_fwd(const float* x, int B, int W, int row_offset, const float* log_scale, float scale_mul,
                          float* loss_rows, float* lse, antmmf_stream_t stream);
/* dx = coef * s * (softmax - onehot);  *dscale += coef * sum_c (softmax - onehot) * x  (= d loss / d s; nullable). */
int antmmf_softmax_ce_bwd(const float* x, const float* lse, const float* coef, int B, int W, int row_offset,
                          const float* log_scale, float scale_mul, void* dx, float* dscale, int out_dtype,
                          antmmf_stream_t stream);

/* ---- MoCo loss rows (MocoUtils.moco_loss, prj/base_vtp/roi_univl/univl/model/moco_utils.py:71-81):
 *   loss_rows[i] = LSE({pos[i][:]} U {neg[i][:]}) / T - LSE(pos[i][:] / T);  pos [R, Np] fp32, neg [R, K] fp32 (q . queue), inv_t = 1/T. */
int antmmf_moco_fwd(const float* pos, const float* neg, int R, int Np, int K, float inv_t, float* loss_rows, float* lse_all,
                    float* lse_pos, antmmf_stream_t stream);
/* dpos (fp32) and dneg (out_dtype) for upstream per-row coefficients coef[i]. */
int antmmf_moco_bwd(const float* pos, const float* neg, const float* lse_all, const float* lse_pos, const float* coef, int R,
                    int Np, int K, float inv_t, float* dpos, void* dneg, int out_dtype, antmmf_stream_t stream);
/* ---- NegNCE rows (prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:539-563) on a row slab S [B, W] (row i = global row
 * row_offset + i; diag[j] = S_jj): pos_rows[i] = -log p_ii, neg_sum / neg_cnt = sum and count of -log(1 - p_ij) over the
 * margin-violating off-diagonal entries, p = clamp(softmax(scale * S_i), 1e-6, 1 - 1e-6).  CrossEn (:528-537) is
 * antmmf_softmax_ce_* with scale_mul = 100. */
int antmmf_negnce_fwd(const float* S, const float* diag, int B, int W, int row_offset, float scale, float margin,
                      float* pos_rows, float* neg_sum, float* neg_cnt, float* lse, antmmf_stream_t stream);
/* dS (out_dtype) for loss = coef[0] * sum_i pos_i + coef[1] * sum neg_ij; coef is a 2-float DEVICE array. */
int antmmf_negnce_bwd(const float* S, const float* diag, const float* lse, const float* coef, int B, int W, int row_offset,
                      float scale, float margin, void* dS, int out_dtype, antmmf_stream_t stream);
/* ---- DMAE weighted token-wise interaction, reduction over one S = text x video^T slab (DmaeUtils._get_wti_similarity,
 * prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:85-131).  S [A*T, B*V] fp32 (row a*T+t, column b*V+v), masks tmask [A,T] /
 * vmask [B,V] (1 = real), optional second-best-frame tables f2f [B,V] (max masked frame-frame similarity) and z2_of [B,V] (its
 * arg-max) -- both or neither.  Outputs t2v [A,B,T], v2t [A,B,V] and the arg-max indices z1 [A,B,T], tmax [A,B,V].  V <= 32. */
int antmmf_wti_reduce_fwd(const float* S, int A, int T, int B, int V, const float* tmask, const float* vmask, const float* f2f,
                          const int* z2_of, float* t2v, float* v2t, int* z1, int* tmax, antmmf_stream_t stream);
/* dS (out_dtype, fully written) from dt2v / dv2t; df2f [B,V] (nullable) is accumulated with atomics (caller zeroes it). */
int antmmf_wti_reduce_bwd(const float* S, int A, int T, int B, int V, const float* tmask, const float* vmask, const float* f2f,
                          const int* z2_of, const int* z1, const int* tmax, const float* dt2v, const float* dv2t, void* dS,
                          float* df2f, int out_dtype, antmmf_stream_t stream);
/* ---- DMAE stage-3 head, small fp32 row kernels (prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:147-165,411-470; tpmcl_utils.py:101-121).
 * token weights: out[n, t] = softmax over t of (feat[n, t, :] . w + bias[0]), tokens with mask[n, t] < 0.5 excluded (weight 0);
 * replaces Linear(D, 1) + masked_fill + softmax.  feat [N, T, D], mask [N, T] (nullable), T <= 128. */
int antmmf_token_weight_fwd(const float* feat, const float* w, const float* bias, const float* mask, float* out, int N, int T, int D,
                            antmmf_stream_t stream);
/* backward: dfeat [N, T, D] (nullable, fully written), dw [D] and dbias [1] (nullable) ACCUMULATED; p = the forward output; scratch: fp32
 * workspace of >= 512 * (D + 1) floats for the per-workgroup partial sums (deterministic, no atomics).  D <= 1024. */
int antmmf_token_weight_bwd(const float* feat, const float* w, const float* p, const float* dout, float* dfeat, float* dw, float* dbias,
                            float* scratch, long scratch_floats, int N, int T, int D, antmmf_stream_t stream);
/* aligned-pair token products (einsum 'ctd,cvd->ctv' with one text token; einsum 'abd,ab->ad'):
 *   pair_dots  out[c, v]    = x[c, :] . y[c, v, :]          pair_wsum  out[c, :] = sum_v w[c, v] y[c, v, :]
 *   pair_outer out[c, v, :] = w[c, v] x[c, :]               (each is the gradient of the others; V <= 128 for wsum / outer) */
int antmmf_pair_dots(const float* x, const float* y, float* out, int C, int V, int D, antmmf_stream_t stream);
int antmmf_pair_wsum(const float* w, const float* y, float* out, int C, int V, int D, antmmf_stream_t stream);
int antmmf_pair_outer(const float* w, const float* x, float* out, int C, int V, int D, antmmf_stream_t stream);
/* TokenImportanceSelector's keep mask (tpmcl_utils.py:101-121): keep[r, t] = 0 where the cumulative weight of the tokens in descending
 * order, token t included, is < thresh; 1 elsewhere.  T <= 64. */
int antmmf_tis_keep(const float* w, float thresh, float* keep, int R, int T, antmmf_stream_t stream);
/* ---- retrieval evaluation (antmmf/modules/metrics/global_retrieval_recall.py:13-89): rank[i] = min over the ground-truth columns
 * gt_idx[gt_off[i] .. gt_off[i+1]) of #{ j : S[i][j] > S[i][g] } (0 = first).  S fp32 [rows, cols] with row stride ld. */
int antmmf_rank_rows(const float* S, int64_t ld, int rows, int cols, const int* gt_off, const int* gt_idx, int* rank,
                     antmmf_stream_t stream);
/* ---- hidden-state dropout of the BERT blocks (modeling_bert.py:175-186,227-238): y = x * keep / (1 - p) (+ residual), keep from the
 * same counter-based mask family (index = element offset).  The backward is the same call on dy with residual = NULL. n % 8 == 0. */
int antmmf_dropout_add(const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed, int dtype,
                       antmmf_stream_t stream);
/* ---- momentum update of a MoCo key tower laid out flat: k = m k + (1 - m) q, k_shadow_bf16 (nullable) = bf16(k).
 * Replaces momentum_update_key_encoder's per-parameter loop (moco_utils.py:55-69). */
int antmmf_ema_update(float* k, const float* q, void* k_shadow_bf16, int64_t n, float m, antmmf_stream_t stream);

/* ---- input pipeline: Pillow-exact antialiased bicubic resize of a batch of RAGGED 8-bit interleaved images,
 * replacing the per-image CPU `square_transform(size)` = Resize((size, size), BICUBIC) + ToTensor
 * (prj/M2_Encoder/vlmo/transforms/square_transform.py:8-14, called from prj/M2_Encoder/m2_encoder.py:61-68; arithmetic =
 * Pillow src/libImaging/Resample.c, 8bpc path).  src (16-byte aligned): images packed back to back ([h_i, w_i, channels] uint8,
 * channels 1 / 3 / 4), src_bytes its size.
 * desc [n_images, 10] int64 (DEVICE): src byte offset, h, w, tmp byte offset, horizontal coeff offset, taps kx (0 = pass skipped),
 * horizontal bounds offset, vertical coeff offset, taps ky (0 = skipped), vertical bounds offset; coeffs = 22-bit fixed-point
 * int32 tables (horizontal: tap-major [kx rounded up to a multiple of 4, out_w], zero-padded; vertical: output-major [out_h, ky]) and bounds [out, 2] = (first tap, tap
 * count), both computed by the host in double precision exactly as Pillow's precompute_coeffs does.  tmp: scratch for the [h_i, out_w, channels] intermediates.  out: uint8 [n, out_h, out_w, C]
 * (out_f32 = 0, byte-identical to PIL) or float32 [n, C, out_h, out_w] = u8 / 255 (out_f32 = 1, ToTensor).  max_h / max_w: the
 * largest input height / width in the batch (max_w * channels <= 160 KiB: one row is staged in LDS). */
int antmmf_resize_bicubic_u8(const void* src, int64_t src_bytes, const int64_t* desc, int n_images, int max_h, int max_w,
                             int channels, int out_h, int out_w, const int32_t* coeffs, const int32_t* bounds, void* tmp, void* out,
                             int out_f32, antmmf_stream_t stream);

/* ---- input pipeline: the video-frame transform in front of the visual tower, for all frames of one video at once:
 * uint8 -> float32 (CustomTransforms.__call__, antmmf/datasets/processors/image_processors.py:520-547) -> bilinear resize to
 * (out_h, out_w) (ImageLongsideScaleAndPad, antmmf/utils/image_ops.py:127-223 = torchvision F.resize on a tensor =
 * interpolate(mode="bilinear", align_corners=False)) -> GroupNormalize (image_ops.py:72-108: / 255 if the resized maximum is > 1,
 * - mean[c], / std[c]) -> written into a (zero-initialised) padded canvas, the collate step of NestedTensor.from_tensor_list
 * (antmmf/structures/nested_tensor.py:51-63).  src: uint8 frames with BYTE strides (sn, sc, sh, sw) of [frame, channel, row, column]
 * ([n, C, h, w] and the decoder's [n, h, w, C] are both strides).  out: float32, element (f, c, y, x) at out[f on + c oc + y oh + x].
 * mean / std: device float[C], or both NULL (resize only: plain float32 frames).  div255: 1 / 0 fixed; -1 = evaluate the reference's
 * "resized.max() > 1" on the device into max_scratch (device int; one extra read-only pass, no host round trip). */
int antmmf_frames_bilinear_norm(const void* src, int n, int channels, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                float* out, int out_h, int out_w, int64_t on, int64_t oc, int64_t oh, const float* mean, const float* std,
                                int div255, int* max_scratch, antmmf_stream_t stream);
/* The same transform with torchvision >= 0.17's tensor default antialias=True: torch.nn.functional.interpolate(..., antialias=True), ATen's separable
 * triangle filter (horizontal pass into `temp`, DEVICE float[n * channels * h * out_w], then the vertical pass + GroupNormalize). */
int antmmf_frames_bilinear_aa_norm(const void* src, int n, int channels, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* temp,
                                   float* out, int out_h, int out_w, int64_t on, int64_t oc, int64_t oh, const float* mean, const float* stdv,
                                   int div255, int* max_scratch, antmmf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ANTMMF_HIP_H */
