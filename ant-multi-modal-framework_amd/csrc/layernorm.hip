// layernorm.hip -- row LayerNorm forward / backward for gfx950.
//
// Replaces (reference, all stock torch ops there):
//   CLIP  LayerNorm (fp32 upcast)      antmmf/modules/vision/backbone/clip/model.py:213-219
//   BERT  BertLayerNorm eps 1e-12      antmmf/modules/vision/backbone/clip/modeling_bert.py:63
//   M2    LayerNorm eps 1e-5 (x4/layer, one of them over 4d)  prj/M2_Encoder/vlmo/torchscale/architecture/encoder.py:34,77
//
// HBM-bound.  One wave (64 lanes) owns a row; the row lives in registers (8-element vectors, 16 B per
// lane per load for bf16), statistics are a two-pass mean / centred variance in fp32, reductions are
// wave butterflies (no LDS).  Algorithmic bytes / row: fwd 2*cols*sizeof(T) (+8 B stats);
// bwd 3*cols*sizeof(T) (+ optional residual-gradient read).
#include "common.h"

template <typename T, int VPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long rows, int cols, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = cols >> 3;
    const float inv_n = 1.0f / (float)cols;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const T* xr = x + row * cols;
        float v[VPL][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
                ld8<T>(xr + vi * 8, v[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[i][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
            }
        }
        const float mean = wave_sum(s) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (lane + 64 * i < nvec) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
        T* yr = y + row * cols;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
                float g[8], b[8], o[8];
                ld8<float>(gamma + vi * 8, g);
                ld8<float>(beta + vi * 8, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
                st8<T>(yr + vi * 8, o);
            }
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) [+ dres];  dgamma += sum_rows dy*xhat;  dbeta += sum_rows dy
template <typename T, int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, long rows, int cols) {
    ANTMMF_DYN_LDS(float, red);  // [2][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = cols >> 3;
    const float inv_n = 1.0f / (float)cols;
    for (int i = threadIdx.x; i < 2 * cols; i += 256) red[i] = 0.f;
    float ag[VPL][8], ab[VPL][8], gm[VPL][8];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; gm[i][e] = 0.f; }
        if (vi < nvec) ld8<float>(gamma + vi * 8, gm[i]);
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        float xh[VPL][8], g[VPL][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
                float xv[8], dv[8];
                ld8<T>(x + row * cols + vi * 8, xv);
                ld8<T>(dy + row * cols + vi * 8, dv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[i][e] = (xv[e] - mean) * rstd;
                    g[i][e] = dv[e] * gm[i][e];
                    s1 += g[i][e];
                    s2 += g[i][e] * xh[i][e];
                    ag[i][e] += dv[e] * xh[i][e];
                    ab[i][e] += dv[e];
                }
            }
        }
        const float c1 = wave_sum(s1) * inv_n, c2 = wave_sum(s2) * inv_n;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (g[i][e] - c1 - xh[i][e] * c2);
                if (dres) {
                    float r[8];
                    ld8<T>(dres + row * cols + vi * 8, r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += r[e];
                }
                st8<T>(dx + row * cols + vi * 8, o);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                atomicAdd(&red[vi * 8 + e], ag[i][e]);
                atomicAdd(&red[cols + vi * 8 + e], ab[i][e]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cols; i += 256) {
        if (dgamma) atomicAdd(&dgamma[i], red[i]);
        if (dbeta) atomicAdd(&dbeta[i], red[cols + i]);
    }
}

template <typename T>
static int ln_fwd_launch(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, long rows,
                         int cols, float eps, hipStream_t s) {
    const int nvec = cols / 8;
    const int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
#define LN_FWD(V) hipLaunchKernelGGL((ln_fwd_kernel<T, V>), dim3(grid), dim3(256), 0, s, (const T*)x, g, b, (T*)y, mean, rstd, rows, cols, eps)
    if (nvec <= 64) LN_FWD(1);
    else if (nvec <= 128) LN_FWD(2);
    else if (nvec <= 256) LN_FWD(4);
    else if (nvec <= 512) LN_FWD(8);
    else return ANTMMF_EINVAL;
#undef LN_FWD
    return antmmf_check_launch();
}

template <typename T>
static int ln_bwd_launch(const void* dy, const void* x, const float* mean, const float* rstd, const float* g,
                         const void* dres, void* dx, float* dgamma, float* dbeta, long rows, int cols, hipStream_t s) {
    const int nvec = cols / 8;
    long want = (rows + 3) / 4;
    const int grid = (int)(want < 512 ? want : 512);
    const size_t lds = (size_t)2 * cols * sizeof(float);
#define LN_BWD(V) hipLaunchKernelGGL((ln_bwd_kernel<T, V>), dim3(grid), dim3(256), lds, s, (const T*)dy, (const T*)x, mean, rstd, g, (const T*)dres, (T*)dx, dgamma, dbeta, rows, cols)
    if (nvec <= 64) LN_BWD(1);
    else if (nvec <= 128) LN_BWD(2);
    else if (nvec <= 256) LN_BWD(4);
    else if (nvec <= 512) LN_BWD(8);
    else return ANTMMF_EINVAL;
#undef LN_BWD
    return antmmf_check_launch();
}

extern "C" int antmmf_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                    float* rstd, long rows, int cols, float eps, int dtype, hipStream_t stream) {
    if (!x || !gamma || !beta || !y || rows < 0 || cols <= 0 || (cols & 7) || cols > 4096) return ANTMMF_EINVAL;
    if (rows == 0) return ANTMMF_OK;
    return dtype == ANTMMF_BF16 ? ln_fwd_launch<bf16_t>(x, gamma, beta, y, mean, rstd, rows, cols, eps, stream)
         : dtype == ANTMMF_F32  ? ln_fwd_launch<float>(x, gamma, beta, y, mean, rstd, rows, cols, eps, stream)
                                : ANTMMF_EINVAL;
}

extern "C" int antmmf_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                                    const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                                    long rows, int cols, int dtype, hipStream_t stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || rows < 0 || cols <= 0 || (cols & 7) || cols > 4096) return ANTMMF_EINVAL;
    if (rows == 0) return ANTMMF_OK;
    return dtype == ANTMMF_BF16 ? ln_bwd_launch<bf16_t>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, rows, cols, stream)
         : dtype == ANTMMF_F32  ? ln_bwd_launch<float>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, rows, cols, stream)
                                : ANTMMF_EINVAL;
}
