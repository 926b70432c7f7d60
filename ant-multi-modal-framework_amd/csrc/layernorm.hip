// layernorm.hip -- row LayerNorm forward / backward for gfx950, optionally fused with a preceding activation.
//
//     y = LN(act(x)) = (act(x) - mean) * rstd * gamma + beta            act = none | gelu(erf) | quick_gelu | relu
//
// Replaces (reference, all stock torch ops there):
//   CLIP  LayerNorm (fp32 upcast)      antmmf/modules/vision/backbone/clip/model.py:213-219
//   BERT  BertLayerNorm eps 1e-12      antmmf/modules/vision/backbone/clip/modeling_bert.py:63
//   M2    LayerNorm eps 1e-5 (x4/layer) prj/M2_Encoder/vlmo/torchscale/architecture/encoder.py:34,77 and the sub-LN
//         over 4d behind the GELU: fc1 -> F.gelu -> ffn_layernorm -> fc2 (feedforward_network.py:117-128) -- the
//         activation is fused here, so gelu(u) never makes an HBM round trip (forward: 1 read + 1 write of the 4d-wide
//         tensor instead of 2 + 2; backward: 2 reads + 1 write instead of 4 + 2).
//
// HBM-bound.  Rows up to 1024 columns: one wave (64 lanes) owns a row, the row lives in registers (16-B loads), two-pass
// mean / centred variance in fp32, wave butterflies only.  Wider rows (the 4d-wide sub-LN): one 256-thread workgroup
// owns a row (<= 16 elements per thread keeps the register count low and the occupancy up), cross-wave sums through LDS.
// Algorithmic bytes / row: fwd 2*cols*sizeof(T) (+8 B stats); bwd 3*cols*sizeof(T) (+ optional residual-gradient read).
#include "common.h"

// ------------------------------------------------------------------ reductions over a "row group" (a wave or a 4-wave workgroup)
template <bool BLOCK>
__device__ __forceinline__ float row_sum(float v, float* sh) {
    v = wave_sum(v);
    if (!BLOCK) return v;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// BLOCK = false: wave per row, lane handles vectors lane + 64 i.  BLOCK = true: workgroup per row, thread handles vectors tid + 256 i.
template <typename T, int VPL, bool BLOCK, int ACT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long rows, int cols, float eps, int act) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = cols >> 3;
    const int v0 = BLOCK ? threadIdx.x : lane, vstep = BLOCK ? 256 : 64;
    const float inv_n = 1.0f / (float)cols;
    const long rstep = BLOCK ? (long)gridDim.x : (long)gridDim.x * 4;
    for (long row = BLOCK ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave; row < rows; row += rstep) {
        const T* xr = x + row * cols;
        float v[VPL][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) {
                ld8<T>(xr + vi * 8, v[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) { float dz_unused; act_fwd_grad<ACT>(v[i][e], act, v[i][e], dz_unused); s += v[i][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
            }
        }
        const float mean = row_sum<BLOCK>(s, sh) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (v0 + vstep * i < nvec) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(row_sum<BLOCK>(q, sh) * inv_n + eps);
        T* yr = y + row * cols;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) {
                float g[8], b[8], o[8];
                ld8<float>(gamma + vi * 8, g);
                ld8<float>(beta + vi * 8, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
                st8<T>(yr + vi * 8, o);
            }
        }
        if ((BLOCK ? threadIdx.x : lane) == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    }
}

// z = act(x);  dz = rstd * (g*dy - mean(g*dy) - zhat * mean(g*dy*zhat));  dx = dz * act'(x) [+ dres];
// dgamma += sum_rows dy*zhat;  dbeta += sum_rows dy
template <typename T, int VPL, bool BLOCK, int ACT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, long rows, int cols, int act, float* __restrict__ partials) {
    ANTMMF_DYN_LDS(float, red);  // wave-per-row: [2][cols] cross-wave column sums; workgroup-per-row: unused
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = cols >> 3;
    const int v0 = BLOCK ? threadIdx.x : lane, vstep = BLOCK ? 256 : 64;
    const float inv_n = 1.0f / (float)cols;
    if (!BLOCK) {
        for (int i = threadIdx.x; i < 2 * cols; i += 256) red[i] = 0.f;
    }
    float ag[VPL][8], ab[VPL][8], gm[VPL][8];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = v0 + vstep * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; gm[i][e] = 0.f; }
        if (vi < nvec) ld8<float>(gamma + vi * 8, gm[i]);
    }
    const long rstep = BLOCK ? (long)gridDim.x : (long)gridDim.x * 4;
    for (long row = BLOCK ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave; row < rows; row += rstep) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        float zh[VPL][8], g[VPL][8], da[VPL][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) {
                float xv[8], dv[8];
                ld8<T>(x + row * cols + vi * 8, xv);
                ld8<T>(dy + row * cols + vi * 8, dv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float z;
                    act_fwd_grad<ACT>(xv[e], act, z, da[i][e]);
                    zh[i][e] = (z - mean) * rstd;
                    g[i][e] = dv[e] * gm[i][e];
                    s1 += g[i][e];
                    s2 += g[i][e] * zh[i][e];
                    ag[i][e] += dv[e] * zh[i][e];
                    ab[i][e] += dv[e];
                }
            }
        }
        const float c1 = row_sum<BLOCK>(s1, sh) * inv_n, c2 = row_sum<BLOCK>(s2, sh) * inv_n;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (g[i][e] - c1 - zh[i][e] * c2) * da[i][e];
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
    if (BLOCK && partials) {  // per-workgroup partial column sums [grid][2][cols], reduced by ln_partials_reduce_kernel
        float* pg = partials + (long)blockIdx.x * 2 * cols;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) { st8<float>(pg + vi * 8, ag[i]); st8<float>(pg + cols + vi * 8, ab[i]); }
        }
        return;
    }
    if (BLOCK) {  // a thread owns its columns within the workgroup: straight to global
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (dgamma) atomicAdd(&dgamma[vi * 8 + e], ag[i][e]);
                    if (dbeta) atomicAdd(&dbeta[vi * 8 + e], ab[i][e]);
                }
            }
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = v0 + vstep * i;
        if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                atomicAdd(&red[vi * 8 + e], ag[i][e]);
                atomicAdd(&red[cols + vi * 8 + e], ab[i][e]);
            }
        }
    }
    __syncthreads();
    if (partials) {  // [grid][2][cols], reduced by ln_partials_reduce_kernel (no global atomics from 1000+ workgroups)
        float* pg = partials + (long)blockIdx.x * 2 * cols;
        for (int i = threadIdx.x; i < 2 * cols; i += 256) pg[i] = red[i];
        return;
    }
    for (int i = threadIdx.x; i < cols; i += 256) {
        if (dgamma) atomicAdd(&dgamma[i], red[i]);
        if (dbeta) atomicAdd(&dbeta[i], red[cols + i]);
    }
}

// dgamma[c] += sum_b partials[b][0][c];  dbeta[c] += sum_b partials[b][1][c].   grid (2*cols/256, 8): a workgroup sums 1/8 of the
// partial rows for 256 columns (4 row-slices x 64 float4 lanes), then one atomic per column.
__global__ __launch_bounds__(256) void ln_partials_reduce_kernel(const float* __restrict__ partials, int nblocks, int cols, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float4 red[4][64];
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lx * 4;            // column in the concatenated [2*cols] row
    const int per = (nblocks + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < 2 * cols) {
        for (int b = b0 + ly; b < b1; b += 4) {
            const float4 v = *reinterpret_cast<const float4*>(partials + (long)b * 2 * cols + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[ly][lx] = s;
    __syncthreads();
    if (ly == 0 && c < 2 * cols) {
        const float4 a = red[0][lx], b = red[1][lx], d = red[2][lx], e = red[3][lx];
        const float v[4] = {a.x + b.x + d.x + e.x, a.y + b.y + d.y + e.y, a.z + b.z + d.z + e.z, a.w + b.w + d.w + e.w};
        float* dst = c < cols ? dgamma : dbeta;
        const int cc = c < cols ? c : c - cols;
        if (dst) {
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(dst + cc + k, v[k]);
        }
    }
}

template <typename T>
static int ln_fwd_launch(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, long rows,
                         int cols, float eps, int act, hipStream_t s) {
    const int nvec = cols / 8;
#define LN_FWD_A(V, BLK, GRID, A) hipLaunchKernelGGL((ln_fwd_kernel<T, V, BLK, A>), dim3(GRID), dim3(256), 0, s, (const T*)x, g, b, (T*)y, mean, rstd, rows, cols, eps, act)
#define LN_FWD(V, BLK, GRID) do { if (act == ANTMMF_ACT_NONE) LN_FWD_A(V, BLK, GRID, ANTMMF_ACT_NONE); else if (act == ANTMMF_ACT_GELU_ERF) LN_FWD_A(V, BLK, GRID, ANTMMF_ACT_GELU_ERF); else LN_FWD_A(V, BLK, GRID, -1); } while (0)
    const int gw = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096), gb = (int)(rows < 4096 ? rows : 4096);
    if (nvec <= 64) LN_FWD(1, false, gw);
    else if (nvec <= 128) LN_FWD(2, false, gw);
    else if (nvec <= 256) LN_FWD(1, true, gb);
    else if (nvec <= 512) LN_FWD(2, true, gb);
    else return ANTMMF_EINVAL;
#undef LN_FWD
#undef LN_FWD_A
    return antmmf_check_launch();
}

template <typename T>
static int ln_bwd_launch(const void* dy, const void* x, const float* mean, const float* rstd, const float* g,
                         const void* dres, void* dx, float* dgamma, float* dbeta, long rows, int cols, int act, float* partials,
                         long partial_elems, hipStream_t s) {
    const int nvec = cols / 8;
    const long want = (rows + 3) / 4;
    const bool wide = nvec > 128;
    // wave-per-row kernels: 2048 workgroups (8 waves / SIMD in flight) when the column sums can go through the partials
    // scratch, otherwise 512 so that the closing global atomics (2 * cols per workgroup) stay cheap
    int gw = (int)(want < 512 ? want : 512);
    int gb = (int)(rows < 1024 ? rows : 1024);
    if (wide) {
        if (!(partials && partial_elems >= (long)gb * 2 * cols)) partials = nullptr;
    } else {
        const int gw_big = (int)(want < 2048 ? want : 2048);
        if (partials && gw_big > 512 && partial_elems >= (long)gw_big * 2 * cols) gw = gw_big; else partials = nullptr;
    }
    const int gp = wide ? gb : gw;
    const size_t lds = (size_t)2 * cols * sizeof(float);
#define LN_BWD_A(V, BLK, GRID, LDS, A) hipLaunchKernelGGL((ln_bwd_kernel<T, V, BLK, A>), dim3(GRID), dim3(256), LDS, s, (const T*)dy, (const T*)x, mean, rstd, g, (const T*)dres, (T*)dx, dgamma, dbeta, rows, cols, act, partials)
#define LN_BWD(V, BLK, GRID, LDS) do { if (act == ANTMMF_ACT_NONE) LN_BWD_A(V, BLK, GRID, LDS, ANTMMF_ACT_NONE); else if (act == ANTMMF_ACT_GELU_ERF) LN_BWD_A(V, BLK, GRID, LDS, ANTMMF_ACT_GELU_ERF); else LN_BWD_A(V, BLK, GRID, LDS, -1); } while (0)
    if (nvec <= 64) LN_BWD(1, false, gw, lds);
    else if (nvec <= 128) LN_BWD(2, false, gw, lds);
    else if (nvec <= 256) LN_BWD(1, true, gb, 16);
    else if (nvec <= 512) LN_BWD(2, true, gb, 16);
    else return ANTMMF_EINVAL;
#undef LN_BWD
#undef LN_BWD_A
    if (partials && (dgamma || dbeta))
        hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3((2 * cols + 255) / 256, 8), dim3(256), 0, s, partials, gp, cols, dgamma, dbeta);
    return antmmf_check_launch();
}

static int ln_args_ok(long rows, int cols) { return rows >= 0 && cols > 0 && !(cols & 7) && cols <= 4096; }

extern "C" int antmmf_act_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                        long rows, int cols, float eps, int act, int dtype, hipStream_t stream) {
    if (!x || !gamma || !beta || !y || !ln_args_ok(rows, cols)) return ANTMMF_EINVAL;
    if (rows == 0) return ANTMMF_OK;
    return dtype == ANTMMF_BF16 ? ln_fwd_launch<bf16_t>(x, gamma, beta, y, mean, rstd, rows, cols, eps, act, stream)
         : dtype == ANTMMF_F32  ? ln_fwd_launch<float>(x, gamma, beta, y, mean, rstd, rows, cols, eps, act, stream)
                                : ANTMMF_EINVAL;
}

extern "C" int antmmf_act_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                        const void* dres, void* dx, float* dgamma, float* dbeta, long rows, int cols, int act,
                                        int dtype, float* partials, long partial_elems, hipStream_t stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !ln_args_ok(rows, cols)) return ANTMMF_EINVAL;
    if (rows == 0) return ANTMMF_OK;
    return dtype == ANTMMF_BF16 ? ln_bwd_launch<bf16_t>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, rows, cols, act, partials, partial_elems, stream)
         : dtype == ANTMMF_F32  ? ln_bwd_launch<float>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, rows, cols, act, partials, partial_elems, stream)
                                : ANTMMF_EINVAL;
}

extern "C" int antmmf_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                    float* rstd, long rows, int cols, float eps, int dtype, hipStream_t stream) {
    return antmmf_act_layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, cols, eps, ANTMMF_ACT_NONE, dtype, stream);
}

extern "C" int antmmf_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                                    const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                                    long rows, int cols, int dtype, hipStream_t stream) {
    return antmmf_act_layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, rows, cols, ANTMMF_ACT_NONE, dtype, nullptr, 0, stream);
}
