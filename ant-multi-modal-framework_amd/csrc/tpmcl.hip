// tpmcl.hip -- small fp32 row kernels of the DMAE stage-3 head: token-importance weights, per-pair token
// dot products / weighted token sums, and the token-importance selection mask.  Reference arithmetic:
//   prj/dmae_vtp/roi_univl/univl/model/dmae_utils.py:147-165 (text_weight_fc / video_weight_fc + masked softmax over the tokens),
//   :425-470 (wti_interaction_row: einsum 'ctd,cvd->ctv' on ALIGNED pairs), :411-418 (einsum 'abd,ab->ad': the predicted global feature),
//   prj/dmae_vtp/roi_univl/univl/model/tpmcl_utils.py:101-121 (TokenImportanceSelector: sort / cumsum / scatter).
// All of them are HBM-bound passes over [rows, tokens, D] fp32 tensors (D = 768, tokens <= 64): one 256-thread workgroup per row,
// coalesced 16-B loads along D, wave sums by DPP (common.h) -- the reference runs them as bmm / einsum / softmax / sort launches.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum4(float v, float* sh) {   // 256 threads = 4 waves; same value in every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// dot of two D-vectors by one wave (lanes stride the vector in 16-B pieces when D % 4 == 0)
__device__ __forceinline__ float wave_dot(const float* __restrict__ a, const float* __restrict__ b, int D, int lane) {
    float acc = 0.f;
    if ((D & 3) == 0) {
        for (int c = lane * 4; c < D; c += 256) {
            const float4 x = *reinterpret_cast<const float4*>(a + c), y = *reinterpret_cast<const float4*>(b + c);
            acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
    } else {
        for (int c = lane; c < D; c += 64) acc += a[c] * b[c];
    }
    return wave_sum(acc);
}

#define TW_MAXT 128   // tokens per item handled by the token-weight kernels (30 words / 13 frames on this path)

// ---- token weights: out[n, t] = softmax_t( feat[n, t, :] . w + b   masked to -inf where mask[n, t] < 0.5 )
__global__ __launch_bounds__(256) void token_weight_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ bias,
                                                               const float* __restrict__ mask, float* __restrict__ out, int N, int T, int D) {
    __shared__ float z[TW_MAXT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float b = bias ? bias[0] : 0.f;
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        for (int t = wave; t < T; t += 4) {
            const float d = wave_dot(feat + ((long)n * T + t) * D, w, D, lane);
            if (lane == 0) z[t] = (mask && mask[(long)n * T + t] < 0.5f) ? -INFINITY : d + b;
        }
        __syncthreads();
        if (wave == 0) {
            float m = -INFINITY;
            for (int t = lane; t < T; t += 64) m = fmaxf(m, z[t]);
            m = wave_max(m);
            float s = 0.f;
            for (int t = lane; t < T; t += 64) s += __expf(z[t] - m);   // (every token masked: exp(-inf - -inf) = NaN, as torch.softmax gives)
            s = wave_sum(s);
            for (int t = lane; t < T; t += 64) out[(long)n * T + t] = __expf(z[t] - m) / s;
        }
        __syncthreads();
    }
}
// dz[t] = p[t] (dout[t] - sum_s dout[s] p[s]);  dfeat[n, t, :] = dz[t] w;  partial[block, 0:D] += sum_t dz[t] feat[n, t, :],  partial[block, D] += sum_t dz[t].
// Persistent over items (grid = G blocks): deterministic partial sums instead of atomics; the caller adds the G partial rows.
__global__ __launch_bounds__(256) void token_weight_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ p,
                                                               const float* __restrict__ dout, float* __restrict__ dfeat, float* __restrict__ partial,
                                                               int N, int T, int D) {
    __shared__ float dz[TW_MAXT];
    __shared__ float sh[4];
    float dw_acc[4] = {0.f, 0.f, 0.f, 0.f};    // columns threadIdx.x + 256 k (D <= 1024)
    float db_acc = 0.f;
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        float s = 0.f;
        for (int t = threadIdx.x; t < T; t += 256) s += dout[(long)n * T + t] * p[(long)n * T + t];
        s = block_sum4(s, sh);
        for (int t = threadIdx.x; t < T; t += 256) {
            const float pv = p[(long)n * T + t];
            dz[t] = pv > 0.f ? pv * (dout[(long)n * T + t] - s) : 0.f;   // masked tokens: weight 0, no gradient
        }
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const float g = dz[t];
            const float* f = feat + ((long)n * T + t) * D;
            float* df = dfeat ? dfeat + ((long)n * T + t) * D : nullptr;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = threadIdx.x + 256 * k;
                if (c < D) {
                    dw_acc[k] += g * f[c];
                    if (df) df[c] = g * w[c];
                }
            }
            if (threadIdx.x == 0) db_acc += g;
        }
        __syncthreads();
    }
    float* row = partial + (long)blockIdx.x * (D + 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < D) row[c] = dw_acc[k];
    }
    if (threadIdx.x == 0) row[D] = db_acc;
}
// out[c] += sum_g partial[g * ld + c],  c < ncols
__global__ __launch_bounds__(256) void partial_rows_sum_kernel(const float* __restrict__ partial, int G, int ld, int ncols, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols) return;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += partial[(long)g * ld + c];
    out[c] += s;
}

// ---- aligned pairs: out[c, v] = x[c, :] . y[c, v, :]
__global__ __launch_bounds__(256) void pair_dots_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int C, int V, int D) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = blockIdx.x; c < C; c += gridDim.x)
        for (int v = wave; v < V; v += 4) {
            const float d = wave_dot(x + (long)c * D, y + ((long)c * V + v) * D, D, lane);
            if (lane == 0) out[(long)c * V + v] = d;
        }
}
// out[c, :] = sum_v w[c, v] y[c, v, :]
__global__ __launch_bounds__(256) void pair_wsum_kernel(const float* __restrict__ w, const float* __restrict__ y, float* __restrict__ out, int C, int V, int D) {
    __shared__ float ws[TW_MAXT];
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        for (int v = threadIdx.x; v < V; v += 256) ws[v] = w[(long)c * V + v];
        __syncthreads();
        for (int col = threadIdx.x; col < D; col += 256) {
            float acc = 0.f;
            for (int v = 0; v < V; ++v) acc += ws[v] * y[((long)c * V + v) * D + col];
            out[(long)c * D + col] = acc;
        }
        __syncthreads();
    }
}
// out[c, v, :] = w[c, v] x[c, :]
__global__ __launch_bounds__(256) void pair_outer_kernel(const float* __restrict__ w, const float* __restrict__ x, float* __restrict__ out, int C, int V, int D) {
    __shared__ float ws[TW_MAXT];
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        for (int v = threadIdx.x; v < V; v += 256) ws[v] = w[(long)c * V + v];
        __syncthreads();
        for (int col = threadIdx.x; col < D; col += 256) {
            const float xv = x[(long)c * D + col];
            for (int v = 0; v < V; ++v) out[((long)c * V + v) * D + col] = ws[v] * xv;
        }
        __syncthreads();
    }
}

// ---- token-importance selection: keep[r, t] = 0 for the tokens whose cumulative weight in DESCENDING order (the token itself included)
// is still below thresh, 1 otherwise.  One wave per row, one lane per token (T <= 64); the inclusive descending prefix of token t is
// sum_s w[s] [w[s] > w[t] or (w[s] == w[t] and s <= t)]  -- ties in index order, like a stable sort.
__global__ __launch_bounds__(256) void tis_keep_kernel(const float* __restrict__ w, float thresh, float* __restrict__ keep, int R, int T) {
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += gridDim.x * 4) {
        const float mine = lane < T ? w[(long)r * T + lane] : -INFINITY;
        float cum = 0.f;
        for (int s = 0; s < T; ++s) {
            const float other = __shfl(mine, s, 64);
            if (other > mine || (other == mine && s <= lane)) cum += other;
        }
        if (lane < T) keep[(long)r * T + lane] = cum < thresh ? 0.f : 1.f;
    }
}

// rows are walked with a grid stride: the grid is capped (the CPU lane emulator pays ~15 ms per emulated workgroup)
#ifdef ANTMMF_EMULATE
static inline int row_grid(int rows) { return rows < 6 ? rows : 6; }
#else
static inline int row_grid(int rows) { return rows < 16384 ? rows : 16384; }
#endif

}  // namespace

extern "C" int antmmf_token_weight_fwd(const float* feat, const float* w, const float* bias, const float* mask, float* out, int N, int T, int D,
                                       hipStream_t s) {
    if (!feat || !w || !out || N < 0 || T <= 0 || T > TW_MAXT || D <= 0) return ANTMMF_EINVAL;
    if (!N) return ANTMMF_OK;
    hipLaunchKernelGGL(token_weight_fwd_kernel, dim3(row_grid(N)), dim3(256), 0, s, feat, w, bias, mask, out, N, T, D);
    return antmmf_check_launch();
}
extern "C" int antmmf_token_weight_bwd(const float* feat, const float* w, const float* p, const float* dout, float* dfeat, float* dw, float* dbias,
                                       float* scratch, long scratch_floats, int N, int T, int D, hipStream_t s) {
    if (!feat || !w || !p || !dout || !dw || !scratch || N < 0 || T <= 0 || T > TW_MAXT || D <= 0 || D > 1024) return ANTMMF_EINVAL;
    if (!N) return ANTMMF_OK;
    int G = N < 512 ? N : 512;
    if (G > row_grid(N)) G = row_grid(N);
    if ((long)G * (D + 1) > scratch_floats) G = (int)(scratch_floats / (D + 1));
    if (G < 1) return ANTMMF_EINVAL;
    hipLaunchKernelGGL(token_weight_bwd_kernel, dim3(G), dim3(256), 0, s, feat, w, p, dout, dfeat, scratch, N, T, D);
    // dw [D] and dbias [1] are accumulated (+=): dw and dbias contiguous or not, two launches keep the ABI simple
    hipLaunchKernelGGL(partial_rows_sum_kernel, dim3(ceil_div(D, 256)), dim3(256), 0, s, scratch, G, D + 1, D, dw);
    if (dbias) hipLaunchKernelGGL(partial_rows_sum_kernel, dim3(1), dim3(256), 0, s, scratch + D, G, D + 1, 1, dbias);
    return antmmf_check_launch();
}
extern "C" int antmmf_pair_dots(const float* x, const float* y, float* out, int C, int V, int D, hipStream_t s) {
    if (!x || !y || !out || C < 0 || V <= 0 || D <= 0) return ANTMMF_EINVAL;
    if (!C) return ANTMMF_OK;
    hipLaunchKernelGGL(pair_dots_kernel, dim3(row_grid(C)), dim3(256), 0, s, x, y, out, C, V, D);
    return antmmf_check_launch();
}
extern "C" int antmmf_pair_wsum(const float* w, const float* y, float* out, int C, int V, int D, hipStream_t s) {
    if (!w || !y || !out || C < 0 || V <= 0 || V > TW_MAXT || D <= 0) return ANTMMF_EINVAL;
    if (!C) return ANTMMF_OK;
    hipLaunchKernelGGL(pair_wsum_kernel, dim3(row_grid(C)), dim3(256), 0, s, w, y, out, C, V, D);
    return antmmf_check_launch();
}
extern "C" int antmmf_pair_outer(const float* w, const float* x, float* out, int C, int V, int D, hipStream_t s) {
    if (!w || !x || !out || C < 0 || V <= 0 || V > TW_MAXT || D <= 0) return ANTMMF_EINVAL;
    if (!C) return ANTMMF_OK;
    hipLaunchKernelGGL(pair_outer_kernel, dim3(row_grid(C)), dim3(256), 0, s, w, x, out, C, V, D);
    return antmmf_check_launch();
}
extern "C" int antmmf_tis_keep(const float* w, float thresh, float* keep, int R, int T, hipStream_t s) {
    if (!w || !keep || R < 0 || T <= 0 || T > 64) return ANTMMF_EINVAL;
    if (!R) return ANTMMF_OK;
    hipLaunchKernelGGL(tis_keep_kernel, dim3(row_grid(ceil_div(R, 4))), dim3(256), 0, s, w, thresh, keep, R, T);
    return antmmf_check_launch();
}
