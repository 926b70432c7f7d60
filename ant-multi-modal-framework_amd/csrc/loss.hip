// loss.hip -- row-sharded contrastive losses on fp32 similarity slabs (gfx950).
//
// Each rank owns B pairs and holds two slabs of the global similarity matrix, produced by the MFMA
// GEMM directly in fp32 and never materialised as the reference's [B_g*n]^2 kron/cat/mask temporaries:
//     Rm[i][c] = <text_i, clip_c>            i local (global index row_offset + i), c over ALL B_g*n clips
//     Cm[i][t] = <centre clip of video_i, text_t>                                     t over ALL B_g texts
// MIL-NCE (reference: get_mil_nce_loss, prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:146-197;
// closed form SURVEY.md 8a L1, restated per local row and checked against oracle/losses.py):
//     denom_i = LSE( { log n + Cm[i][t] }_t  U  { Rm[i][c] : c / n != gi } ),   l_i = denom_i - (log n + Cm[i][gi])
// Symmetric InfoNCE / CrossEn (reference logits: prj/M2_Encoder/m2_encoder.py:92-95, clip/model.py:442-444;
// CrossEn prj/dmae_vtp/.../dmae_utils.py:528-537) is softmax-CE over a scaled row with target column gi.
//
// HBM-bound: one workgroup per row, online (max, sum) per thread over 16-B loads, wave butterflies + one LDS hop.
// Algorithmic bytes / row: fwd 4*(Wr + Wc); bwd 4*(Wr + Wc) read + out-dtype*(Wr + Wc) written.
#include "common.h"

struct MS { float m, s; };
__device__ __forceinline__ void ms_add(MS& a, float x) {
    if (x == -INFINITY) return;
    if (x > a.m) { a.s = a.s * __expf(a.m - x) + 1.0f; a.m = x; } else a.s += __expf(x - a.m);
}
__device__ __forceinline__ MS ms_merge(MS a, MS b) {
    if (b.m == -INFINITY) return a;
    if (a.m == -INFINITY) return b;
    MS r; r.m = fmaxf(a.m, b.m); r.s = a.s * __expf(a.m - r.m) + b.s * __expf(b.m - r.m); return r;
}
__device__ __forceinline__ MS block_ms(MS v, MS* sh) {  // 256 threads; result valid in every thread
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { MS t; t.m = __shfl_xor(v.m, o, 64); t.s = __shfl_xor(v.s, o, 64); v = ms_merge(v, t); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return ms_merge(ms_merge(sh[0], sh[1]), ms_merge(sh[2], sh[3]));
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void milnce_fwd_kernel(const float* __restrict__ Rm, const float* __restrict__ Cm, int Wr, int Wc, int n_pair,
                                                         int row_offset, float* __restrict__ loss_rows, float* __restrict__ denom) {
    __shared__ MS sh[4];
    const int i = blockIdx.x, gi = row_offset + i;
    const float logn = __logf((float)n_pair);
    MS acc; acc.m = -INFINITY; acc.s = 0.f;
    const float* r = Rm + (long)i * Wr;
    const float* cm = Cm + (long)i * Wc;
    if (((Wr | Wc) & 3) == 0) {
        for (int c = threadIdx.x * 4; c < Wr; c += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(r + c);
            if (c / n_pair != gi) ms_add(acc, v.x);
            if ((c + 1) / n_pair != gi) ms_add(acc, v.y);
            if ((c + 2) / n_pair != gi) ms_add(acc, v.z);
            if ((c + 3) / n_pair != gi) ms_add(acc, v.w);
        }
        for (int t = threadIdx.x * 4; t < Wc; t += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(cm + t);
            ms_add(acc, v.x + logn); ms_add(acc, v.y + logn); ms_add(acc, v.z + logn); ms_add(acc, v.w + logn);
        }
    } else {
        for (int c = threadIdx.x; c < Wr; c += 256) if (c / n_pair != gi) ms_add(acc, r[c]);
        for (int t = threadIdx.x; t < Wc; t += 256) ms_add(acc, cm[t] + logn);
    }
    const MS tot = block_ms(acc, sh);
    if (threadIdx.x == 0) {
        const float d = tot.m + __logf(tot.s);
        denom[i] = d;
        loss_rows[i] = d - (logn + cm[gi]);
    }
}
template <typename TO>
__global__ __launch_bounds__(256) void milnce_bwd_kernel(const float* __restrict__ Rm, const float* __restrict__ Cm, const float* __restrict__ denom,
                                                         const float* __restrict__ coef, int Wr, int Wc, int n_pair, int row_offset,
                                                         TO* __restrict__ dRm, TO* __restrict__ dCm) {
    const int i = blockIdx.x, gi = row_offset + i;
    const float d = denom[i], k = coef[i], logn = __logf((float)n_pair);
    if (((Wr | Wc) & 3) == 0 && sizeof(TO) == 4) {
        for (int c = threadIdx.x * 4; c < Wr; c += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(Rm + (long)i * Wr + c);
            float4 o;
            o.x = (c / n_pair != gi) ? k * __expf(v.x - d) : 0.f; o.y = ((c + 1) / n_pair != gi) ? k * __expf(v.y - d) : 0.f;
            o.z = ((c + 2) / n_pair != gi) ? k * __expf(v.z - d) : 0.f; o.w = ((c + 3) / n_pair != gi) ? k * __expf(v.w - d) : 0.f;
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(dRm) + (long)i * Wr + c) = o;
        }
        for (int t = threadIdx.x * 4; t < Wc; t += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(Cm + (long)i * Wc + t);
            float4 o;
            o.x = k * (__expf(v.x + logn - d) - (t == gi ? 1.f : 0.f)); o.y = k * (__expf(v.y + logn - d) - (t + 1 == gi ? 1.f : 0.f));
            o.z = k * (__expf(v.z + logn - d) - (t + 2 == gi ? 1.f : 0.f)); o.w = k * (__expf(v.w + logn - d) - (t + 3 == gi ? 1.f : 0.f));
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(dCm) + (long)i * Wc + t) = o;
        }
        return;
    }
    for (int c = threadIdx.x; c < Wr; c += 256)
        st1<TO>(dRm + (long)i * Wr + c, (c / n_pair != gi) ? k * __expf(Rm[(long)i * Wr + c] - d) : 0.f);
    for (int t = threadIdx.x; t < Wc; t += 256)
        st1<TO>(dCm + (long)i * Wc + t, k * (__expf(Cm[(long)i * Wc + t] + logn - d) - (t == gi ? 1.f : 0.f)));
}

// softmax cross-entropy of scale * x[i][:] against column row_offset + i
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float* __restrict__ x, int W, int row_offset, const float* __restrict__ log_scale,
                                                             float scale_mul, float* __restrict__ loss_rows, float* __restrict__ lse) {
    __shared__ MS sh[4];
    const int i = blockIdx.x, gi = row_offset + i;
    const float sc = scale_mul * (log_scale ? __expf(*log_scale) : 1.0f);
    MS acc; acc.m = -INFINITY; acc.s = 0.f;
    const float* r = x + (long)i * W;
    if ((W & 3) == 0) {   // 16-B loads (rows start 16-B aligned then): a [1024 x 8192] slab of the global-batch loss streams at HBM rate
        for (int c = threadIdx.x * 4; c < W; c += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(r + c);
            ms_add(acc, sc * v.x); ms_add(acc, sc * v.y); ms_add(acc, sc * v.z); ms_add(acc, sc * v.w);
        }
    } else {
        for (int c = threadIdx.x; c < W; c += 256) ms_add(acc, sc * r[c]);
    }
    const MS tot = block_ms(acc, sh);
    if (threadIdx.x == 0) {
        const float l = tot.m + __logf(tot.s);
        lse[i] = l;
        loss_rows[i] = l - sc * r[gi];
    }
}
// dx = coef * sc * (softmax - onehot);  *dscale += coef * sum_c (softmax - onehot) * x      (d loss / d sc)
template <typename TO>
__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ lse, const float* __restrict__ coef,
                                                             int W, int row_offset, const float* __restrict__ log_scale, float scale_mul,
                                                             TO* __restrict__ dx, float* __restrict__ dscale) {
    __shared__ float sh[4];
    const int i = blockIdx.x, gi = row_offset + i;
    const float sc = scale_mul * (log_scale ? __expf(*log_scale) : 1.0f);
    const float l = lse[i], k = coef[i];
    float ds = 0.f;
    auto one = [&](int c, float xv) -> float {
        if (xv == -INFINITY) return 0.f;   // masked column (padding rows of a ragged global batch): no probability, no gradient
        const float p = __expf(sc * xv - l) - (c == gi ? 1.f : 0.f);
        ds += p * xv;
        return k * sc * p;
    };
    if ((W & 3) == 0 && sizeof(TO) == 4) {   // fp32 in / fp32 out, 16-B accesses
        for (int c = threadIdx.x * 4; c < W; c += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(x + (long)i * W + c);
            float4 o; o.x = one(c, v.x); o.y = one(c + 1, v.y); o.z = one(c + 2, v.z); o.w = one(c + 3, v.w);
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(dx) + (long)i * W + c) = o;
        }
    } else {
        for (int c = threadIdx.x; c < W; c += 256) st1<TO>(dx + (long)i * W + c, one(c, x[(long)i * W + c]));
    }
    if (dscale) {
        const float tot = block_sum(ds, sh);
        if (threadIdx.x == 0) atomicAdd(dscale, k * tot);
    }
}

// MoCo loss rows (reference: MocoUtils.moco_loss, prj/base_vtp/roi_univl/univl/model/moco_utils.py:71-81):
//     l_i = LSE( {pos[i][:]} U {neg[i][:]} ) / T  -  LSE( pos[i][:] / T )         pos [R, Np] (Np = 1 or n clips), neg [R, K] (queue)
__global__ __launch_bounds__(256) void moco_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg, int Np, int K, float inv_t,
                                                       float* __restrict__ loss_rows, float* __restrict__ lse_all, float* __restrict__ lse_pos) {
    __shared__ MS sh[4];
    const int i = blockIdx.x;
    MS ap; ap.m = -INFINITY; ap.s = 0.f;
    for (int c = threadIdx.x; c < Np; c += 256) ms_add(ap, pos[(long)i * Np + c] * inv_t);
    const MS tp = block_ms(ap, sh);
    MS an; an.m = -INFINITY; an.s = 0.f;
    for (int c = threadIdx.x; c < K; c += 256) ms_add(an, neg[(long)i * K + c] * inv_t);
    const MS tn = block_ms(an, sh);
    if (threadIdx.x == 0) {
        const MS ta = ms_merge(tp, tn);
        const float lp = tp.m + __logf(tp.s), la = ta.m + __logf(ta.s);
        lse_pos[i] = lp; lse_all[i] = la; loss_rows[i] = la - lp;
    }
}
// dpos = coef/T (softmax_all - softmax_pos) (fp32);  dneg = coef/T softmax_all (out dtype)
template <typename TO>
__global__ __launch_bounds__(256) void moco_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ neg, const float* __restrict__ lse_all,
                                                       const float* __restrict__ lse_pos, const float* __restrict__ coef, int Np, int K, float inv_t,
                                                       float* __restrict__ dpos, TO* __restrict__ dneg) {
    const int i = blockIdx.x;
    const float la = lse_all[i], lp = lse_pos[i], k = coef[i] * inv_t;
    for (int c = threadIdx.x; c < Np; c += 256) {
        const float z = pos[(long)i * Np + c] * inv_t;
        dpos[(long)i * Np + c] = k * (__expf(z - la) - __expf(z - lp));
    }
    for (int c = threadIdx.x; c < K; c += 256) st1<TO>(dneg + (long)i * K + c, k * __expf(neg[(long)i * K + c] * inv_t - la));
}

extern "C" int antmmf_moco_fwd(const float* pos, const float* neg, int R, int Np, int K, float inv_t, float* loss_rows, float* lse_all,
                               float* lse_pos, hipStream_t s) {
    if (!pos || !neg || !loss_rows || !lse_all || !lse_pos || R < 0 || Np <= 0 || K <= 0 || !(inv_t > 0.f)) return ANTMMF_EINVAL;
    if (!R) return ANTMMF_OK;
    hipLaunchKernelGGL(moco_fwd_kernel, dim3(R), dim3(256), 0, s, pos, neg, Np, K, inv_t, loss_rows, lse_all, lse_pos);
    return antmmf_check_launch();
}
extern "C" int antmmf_moco_bwd(const float* pos, const float* neg, const float* lse_all, const float* lse_pos, const float* coef, int R, int Np,
                               int K, float inv_t, float* dpos, void* dneg, int out_dtype, hipStream_t s) {
    if (!pos || !neg || !lse_all || !lse_pos || !coef || !dpos || !dneg || R < 0 || Np <= 0 || K <= 0) return ANTMMF_EINVAL;
    if (!R) return ANTMMF_OK;
    if (out_dtype == ANTMMF_BF16) hipLaunchKernelGGL(moco_bwd_kernel<bf16_t>, dim3(R), dim3(256), 0, s, pos, neg, lse_all, lse_pos, coef, Np, K, inv_t, dpos, (bf16_t*)dneg);
    else if (out_dtype == ANTMMF_F32) hipLaunchKernelGGL(moco_bwd_kernel<float>, dim3(R), dim3(256), 0, s, pos, neg, lse_all, lse_pos, coef, Np, K, inv_t, dpos, (float*)dneg);
    else return ANTMMF_EINVAL;
    return antmmf_check_launch();
}

// NegNCE (reference: prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:539-563) on a row slab S [B, W] of the similarity matrix
// (row i is global row gi = row_offset + i; diag[j] = S_jj for every column j):
//   p = clamp(softmax(scale * S_i,:), 1e-6, 1 - 1e-6);  positives -log p_i,gi;  negatives -log(1 - p_ij) over the off-diagonal
//   (i, j) that violate the margin against either diagonal: relu(m + S_ij - S_ii) + relu(m + S_ij - S_jj) > 0.
// Forward emits per-row  -log p_ii,  sum and count of the selected negatives, and the row LSE; the caller forms
//   c_pos * mean(pos) + c_neg * sum(neg) / count   (global count: one all-reduce in the multi-GPU case).
__global__ __launch_bounds__(256) void negnce_fwd_kernel(const float* __restrict__ S, const float* __restrict__ diag, int W, int row_offset, float scale,
                                                         float margin, float* __restrict__ pos_rows, float* __restrict__ neg_sum, float* __restrict__ neg_cnt,
                                                         float* __restrict__ lse) {
    __shared__ MS sh[4];
    __shared__ float shf[4];
    const int i = blockIdx.x, gi = row_offset + i;
    const float* r = S + (long)i * W;
    MS acc; acc.m = -INFINITY; acc.s = 0.f;
    for (int c = threadIdx.x; c < W; c += 256) ms_add(acc, scale * r[c]);
    const MS tot = block_ms(acc, sh);
    const float l = tot.m + __logf(tot.s), dii = r[gi];
    float ns = 0.f, nc = 0.f;
    for (int c = threadIdx.x; c < W; c += 256) {
        if (c == gi) continue;
        const float x = r[c];
        const float mm = fmaxf(margin + x - dii, 0.f) + fmaxf(margin + x - diag[c], 0.f);
        if (mm > 0.f) {
            const float p = fminf(fmaxf(__expf(scale * x - l), 1e-6f), 1.0f - 1e-6f);
            ns += -__logf(1.0f - p); nc += 1.0f;
        }
    }
    ns = block_sum(ns, shf);
    nc = block_sum(nc, shf);
    if (threadIdx.x == 0) {
        const float p = fminf(fmaxf(__expf(scale * dii - l), 1e-6f), 1.0f - 1e-6f);
        pos_rows[i] = -__logf(p); neg_sum[i] = ns; neg_cnt[i] = nc; lse[i] = l;
    }
}
// dS for loss = coef[0] * sum_i pos_i + coef[1] * sum_(i,j) neg_ij  (coef on the device: c_pos / B and c_neg / count, times upstream)
template <typename TO>
__global__ __launch_bounds__(256) void negnce_bwd_kernel(const float* __restrict__ S, const float* __restrict__ diag, const float* __restrict__ lse,
                                                         const float* __restrict__ coef, int W, int row_offset, float scale, float margin,
                                                         TO* __restrict__ dS) {
    __shared__ float shf[4];
    const int i = blockIdx.x, gi = row_offset + i;
    const float* r = S + (long)i * W;
    const float l = lse[i], dii = r[gi], kp = coef[0], kn = coef[1];
    auto grad_p = [&](int c, float x, float p) -> float {  // d loss / d p_c  (zero where the clamp is active)
        if (!(p > 1e-6f && p < 1.0f - 1e-6f)) return 0.f;
        if (c == gi) return -kp / p;
        const float mm = fmaxf(margin + x - dii, 0.f) + fmaxf(margin + x - diag[c], 0.f);
        return mm > 0.f ? kn / (1.0f - p) : 0.f;
    };
    float gp = 0.f;
    for (int c = threadIdx.x; c < W; c += 256) {
        const float x = r[c], p = __expf(scale * x - l);
        gp += grad_p(c, x, p) * p;
    }
    gp = block_sum(gp, shf);
    for (int c = threadIdx.x; c < W; c += 256) {
        const float x = r[c], p = __expf(scale * x - l);
        st1<TO>(dS + (long)i * W + c, scale * p * (grad_p(c, x, p) - gp));
    }
}

extern "C" int antmmf_negnce_fwd(const float* S, const float* diag, int B, int W, int row_offset, float scale, float margin, float* pos_rows,
                                 float* neg_sum, float* neg_cnt, float* lse, hipStream_t s) {
    if (!S || !diag || !pos_rows || !neg_sum || !neg_cnt || !lse || B < 0 || W <= 0 || row_offset < 0 || row_offset + B > W) return ANTMMF_EINVAL;
    if (!B) return ANTMMF_OK;
    hipLaunchKernelGGL(negnce_fwd_kernel, dim3(B), dim3(256), 0, s, S, diag, W, row_offset, scale, margin, pos_rows, neg_sum, neg_cnt, lse);
    return antmmf_check_launch();
}
extern "C" int antmmf_negnce_bwd(const float* S, const float* diag, const float* lse, const float* coef, int B, int W, int row_offset, float scale,
                                 float margin, void* dS, int out_dtype, hipStream_t s) {
    if (!S || !diag || !lse || !coef || !dS || B < 0 || W <= 0 || row_offset < 0 || row_offset + B > W) return ANTMMF_EINVAL;
    if (!B) return ANTMMF_OK;
    if (out_dtype == ANTMMF_BF16) hipLaunchKernelGGL(negnce_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, s, S, diag, lse, coef, W, row_offset, scale, margin, (bf16_t*)dS);
    else if (out_dtype == ANTMMF_F32) hipLaunchKernelGGL(negnce_bwd_kernel<float>, dim3(B), dim3(256), 0, s, S, diag, lse, coef, W, row_offset, scale, margin, (float*)dS);
    else return ANTMMF_EINVAL;
    return antmmf_check_launch();
}

// ---- DMAE weighted token-wise interaction, reduction part (reference: DmaeUtils._get_wti_similarity,
// prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:85-131).  The reference materialises M[a,b,t,v] = <text_a,t, video_b,v> * masks
// as a 4-D tensor and reduces it with max / einsum / advanced indexing; here S = text x video^T comes out of the MFMA GEMM as a
// [A*T, B*V] slab and ONE pass per (a, b) pair produces
//     t2v[a,b,t] = max_v M  (+ 0.5 * f2f[b,z1] * M[a,b,t,z2_of[b,z1]],  z1 = argmax_v:  the "second best frame" term)
//     v2t[a,b,v] = max_t M
// plus the arg-max indices the backward pass routes gradients through.  HBM-bound: S is read once (4 B per (t, v) pair).
#define WTI_MAXV 32
__global__ __launch_bounds__(256) void wti_reduce_fwd_kernel(const float* __restrict__ S, int T, int B, int V, const float* __restrict__ tmask,
                                                             const float* __restrict__ vmask, const float* __restrict__ f2f, const int* __restrict__ z2_of,
                                                             float* __restrict__ t2v, float* __restrict__ v2t, int* __restrict__ z1_out, int* __restrict__ tmax_out) {
    const int a = blockIdx.x, b = blockIdx.y * 256 + threadIdx.x;
    if (b >= B) return;
    float vm[WTI_MAXV], best_t[WTI_MAXV];
    int arg_t[WTI_MAXV];
#pragma unroll 4
    for (int v = 0; v < V; ++v) { vm[v] = vmask[(long)b * V + v]; best_t[v] = -INFINITY; arg_t[v] = 0; }
    const long ab = (long)a * B + b;
    for (int t = 0; t < T; ++t) {
        const float tm = tmask[(long)a * T + t];
        const float* row = S + ((long)a * T + t) * ((long)B * V) + (long)b * V;
        float best = -INFINITY; int arg = 0;
        for (int v = 0; v < V; ++v) {
            const float m = row[v] * tm * vm[v];
            if (m > best) { best = m; arg = v; }           // first maximum, like torch.max
            if (m > best_t[v]) { best_t[v] = m; arg_t[v] = t; }
        }
        float out = best;
        if (f2f) {
            const int z2 = z2_of[(long)b * V + arg];
            out += 0.5f * f2f[(long)b * V + arg] * (row[z2] * tm * vm[z2]);
        }
        t2v[ab * T + t] = out;
        z1_out[ab * T + t] = arg;
    }
    for (int v = 0; v < V; ++v) { v2t[ab * V + v] = best_t[v]; tmax_out[ab * V + v] = arg_t[v]; }
}
// dS[a*T+t][b*V+v] = tm vm ( [v == z1] dt2v + [v == z2] 0.5 f2f[b,z1] dt2v + [t == tmax[v]] dv2t[v] );  df2f[b,z1] += 0.5 M[t][z2] dt2v
template <typename TO>
__global__ __launch_bounds__(256) void wti_reduce_bwd_kernel(const float* __restrict__ S, int T, int B, int V, const float* __restrict__ tmask,
                                                             const float* __restrict__ vmask, const float* __restrict__ f2f, const int* __restrict__ z2_of,
                                                             const int* __restrict__ z1_in, const int* __restrict__ tmax_in, const float* __restrict__ dt2v,
                                                             const float* __restrict__ dv2t, TO* __restrict__ dS, float* __restrict__ df2f) {
    const int a = blockIdx.x, b = blockIdx.y * 256 + threadIdx.x;
    if (b >= B) return;
    const long ab = (long)a * B + b;
    for (int t = 0; t < T; ++t) {
        const float tm = tmask[(long)a * T + t];
        const long ro = ((long)a * T + t) * ((long)B * V) + (long)b * V;
        const int z1 = z1_in[ab * T + t];
        const float g = dt2v[ab * T + t];
        int z2 = -1; float w2 = 0.f;
        if (f2f) {
            z2 = z2_of[(long)b * V + z1];
            w2 = 0.5f * f2f[(long)b * V + z1];
            if (df2f) atomicAdd(&df2f[(long)b * V + z1], 0.5f * g * (S[ro + z2] * tm * vmask[(long)b * V + z2]));
        }
        for (int v = 0; v < V; ++v) {
            float val = 0.f;
            if (v == z1) val += g;
            if (v == z2) val += g * w2;
            if (tmax_in[ab * V + v] == t) val += dv2t[ab * V + v];
            st1<TO>(dS + ro + v, val * tm * vmask[(long)b * V + v]);
        }
    }
}

extern "C" int antmmf_wti_reduce_fwd(const float* S, int A, int T, int B, int V, const float* tmask, const float* vmask, const float* f2f,
                                     const int* z2_of, float* t2v, float* v2t, int* z1, int* tmax, hipStream_t s) {
    if (!S || !tmask || !vmask || !t2v || !v2t || !z1 || !tmax || A < 0 || B < 0 || T <= 0 || V <= 0 || V > WTI_MAXV || (!f2f) != (!z2_of)) return ANTMMF_EINVAL;
    if (!A || !B) return ANTMMF_OK;
    hipLaunchKernelGGL(wti_reduce_fwd_kernel, dim3(A, (B + 255) / 256), dim3(256), 0, s, S, T, B, V, tmask, vmask, f2f, z2_of, t2v, v2t, z1, tmax);
    return antmmf_check_launch();
}
extern "C" int antmmf_wti_reduce_bwd(const float* S, int A, int T, int B, int V, const float* tmask, const float* vmask, const float* f2f,
                                     const int* z2_of, const int* z1, const int* tmax, const float* dt2v, const float* dv2t, void* dS, float* df2f,
                                     int out_dtype, hipStream_t s) {
    if (!S || !tmask || !vmask || !z1 || !tmax || !dt2v || !dv2t || !dS || A < 0 || B < 0 || T <= 0 || V <= 0 || V > WTI_MAXV || (!f2f) != (!z2_of)) return ANTMMF_EINVAL;
    if (!A || !B) return ANTMMF_OK;
    const dim3 grid(A, (B + 255) / 256);
    if (out_dtype == ANTMMF_BF16) hipLaunchKernelGGL(wti_reduce_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, S, T, B, V, tmask, vmask, f2f, z2_of, z1, tmax, dt2v, dv2t, (bf16_t*)dS, df2f);
    else if (out_dtype == ANTMMF_F32) hipLaunchKernelGGL(wti_reduce_bwd_kernel<float>, grid, dim3(256), 0, s, S, T, B, V, tmask, vmask, f2f, z2_of, z1, tmax, dt2v, dv2t, (float*)dS, df2f);
    else return ANTMMF_EINVAL;
    return antmmf_check_launch();
}

// ---- retrieval evaluation: rank of the ground truth in every row of a similarity matrix (reference: np.argsort(-sim) + a Python
// loop per row, antmmf/modules/metrics/global_retrieval_recall.py:13-89).  rank[i] = min over the row's ground-truth columns g of
// #{ j : S[i][j] > S[i][g] }  (0 = retrieved first).  HBM-bound: one pass over the row per ground-truth id.
__global__ __launch_bounds__(256) void rank_rows_kernel(const float* __restrict__ S, long ld, int cols, const int* __restrict__ gt_off,
                                                        const int* __restrict__ gt_idx, int* __restrict__ rank) {
    __shared__ float shf[4];
    const int i = blockIdx.x;
    const float* r = S + (long)i * ld;
    int best = cols;
    for (int g = gt_off[i]; g < gt_off[i + 1]; ++g) {
        // position of the ground truth in a STABLE descending sort (the reference argsorts -S): strictly larger scores, plus equal scores at a
        // lower column index -- a collapsed model (all scores equal) gets index-order ranks, not rank 0; NaN scores of other columns sort last
        // (never counted), a NaN ground-truth score is the worst rank
        const int gj = gt_idx[g];
        const float thr = r[gj];
        float c = 0.f;
        for (int j = threadIdx.x; j < cols; j += 256) { const float v = r[j]; c += (v > thr || (v == thr && j < gj)) ? 1.f : 0.f; }
        int cnt = (int)block_sum(c, shf);
        if (!(thr == thr)) cnt = cols - 1;
        best = cnt < best ? cnt : best;
    }
    if (threadIdx.x == 0) rank[i] = best;
}
extern "C" int antmmf_rank_rows(const float* S, long ld, int rows, int cols, const int* gt_off, const int* gt_idx, int* rank, hipStream_t s) {
    if (!S || !gt_off || !gt_idx || !rank || rows < 0 || cols <= 0 || ld < cols) return ANTMMF_EINVAL;
    if (!rows) return ANTMMF_OK;
    hipLaunchKernelGGL(rank_rows_kernel, dim3(rows), dim3(256), 0, s, S, ld, cols, gt_off, gt_idx, rank);
    return antmmf_check_launch();
}

extern "C" int antmmf_milnce_fwd(const float* Rm, const float* Cm, int B, int Wr, int Wc, int n_pair, int row_offset,
                                 float* loss_rows, float* denom, hipStream_t s) {
    if (!Rm || !Cm || !loss_rows || !denom || B < 0 || Wr <= 0 || Wc <= 0 || n_pair < 1 || row_offset < 0 || row_offset + B > Wc) return ANTMMF_EINVAL;
    if (!B) return ANTMMF_OK;
    hipLaunchKernelGGL(milnce_fwd_kernel, dim3(B), dim3(256), 0, s, Rm, Cm, Wr, Wc, n_pair, row_offset, loss_rows, denom);
    return antmmf_check_launch();
}
extern "C" int antmmf_milnce_bwd(const float* Rm, const float* Cm, const float* denom, const float* coef, int B, int Wr, int Wc, int n_pair,
                                 int row_offset, void* dRm, void* dCm, int out_dtype, hipStream_t s) {
    if (!Rm || !Cm || !denom || !coef || !dRm || !dCm || B < 0 || Wr <= 0 || Wc <= 0 || n_pair < 1) return ANTMMF_EINVAL;
    if (!B) return ANTMMF_OK;
    if (out_dtype == ANTMMF_BF16) hipLaunchKernelGGL(milnce_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, s, Rm, Cm, denom, coef, Wr, Wc, n_pair, row_offset, (bf16_t*)dRm, (bf16_t*)dCm);
    else if (out_dtype == ANTMMF_F32) hipLaunchKernelGGL(milnce_bwd_kernel<float>, dim3(B), dim3(256), 0, s, Rm, Cm, denom, coef, Wr, Wc, n_pair, row_offset, (float*)dRm, (float*)dCm);
    else return ANTMMF_EINVAL;
    return antmmf_check_launch();
}
extern "C" int antmmf_softmax_ce_fwd(const float* x, int B, int W, int row_offset, const float* log_scale, float scale_mul,
                                     float* loss_rows, float* lse, hipStream_t s) {
    if (!x || !loss_rows || !lse || B < 0 || W <= 0 || row_offset < 0 || row_offset + B > W) return ANTMMF_EINVAL;
    if (!B) return ANTMMF_OK;
    hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3(B), dim3(256), 0, s, x, W, row_offset, log_scale, scale_mul, loss_rows, lse);
    return antmmf_check_launch();
}
extern "C" int antmmf_softmax_ce_bwd(const float* x, const float* lse, const float* coef, int B, int W, int row_offset, const float* log_scale,
                                     float scale_mul, void* dx, float* dscale, int out_dtype, hipStream_t s) {
    if (!x || !lse || !coef || !dx || B < 0 || W <= 0) return ANTMMF_EINVAL;
    if (!B) return ANTMMF_OK;
    if (out_dtype == ANTMMF_BF16) hipLaunchKernelGGL(softmax_ce_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, s, x, lse, coef, W, row_offset, log_scale, scale_mul, (bf16_t*)dx, dscale);
    else if (out_dtype == ANTMMF_F32) hipLaunchKernelGGL(softmax_ce_bwd_kernel<float>, dim3(B), dim3(256), 0, s, x, lse, coef, W, row_offset, log_scale, scale_mul, (float*)dx, dscale);
    else return ANTMMF_EINVAL;
    return antmmf_check_launch();
}
