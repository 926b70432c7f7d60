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

// raw (still packed) 8-element vector of a row: lets the NEXT row's loads be issued before the current row is touched
template <typename T> struct raw8;
template <> struct raw8<bf16_t> {
    uint4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void unpack(float (&f)[8]) const {
        f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
        f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
    }
};
template <> struct raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    __device__ __forceinline__ void unpack(float (&f)[8]) const {
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
};

// BLOCK = false: wave per row, lane handles vectors lane + 64 i.  BLOCK = true: workgroup per row, thread handles vectors tid + 256 i.
// Both walk rows with a grid stride and keep the next row's loads in flight while the current one is reduced.
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
    long row = BLOCK ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave;
    raw8<T> nx[VPL];
    auto fetch = [&](long r) {
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            if (v0 + vstep * i < nvec) nx[i].load(x + r * cols + (v0 + vstep * i) * 8);
    };
    if (row < rows) fetch(row);
    for (; row < rows; row += rstep) {
        float v[VPL][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (v0 + vstep * i < nvec) {
                nx[i].unpack(v[i]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
            }
        }
        if (row + rstep < rows) fetch(row + rstep);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (v0 + vstep * i < nvec) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { float dz_unused; act_fwd_grad<ACT>(v[i][e], act, v[i][e], dz_unused); s += v[i][e]; }
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
// dgamma += sum_rows dy*zhat;  dbeta += sum_rows dy;  DXSUM: dxsum += sum_rows dx -- the bias gradient of the Linear whose
// output this LayerNorm reads (fc1 for the fused gelu + ffn_layernorm; the attention out-projection for ln2 with the
// residual gradient added), which would otherwise be a separate full read of dx.
// Column sums leave through `partials` ([grid][NS][cols], reduced by ln_partials_reduce_kernel) or, without it, atomics.
template <typename T, int VPL, bool BLOCK, int ACT, bool DXSUM>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dxsum, long rows, int cols, int act,
                                                     float* __restrict__ partials) {
    constexpr int NS = DXSUM ? 3 : 2;
    ANTMMF_DYN_LDS(float, red);  // wave-per-row: [NS][cols] cross-wave column sums; workgroup-per-row: unused
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = cols >> 3;
    const int v0 = BLOCK ? threadIdx.x : lane, vstep = BLOCK ? 256 : 64;
    const float inv_n = 1.0f / (float)cols;
    if (!BLOCK) {
        for (int i = threadIdx.x; i < NS * cols; i += 256) red[i] = 0.f;
    }
    // gamma: registers for the wave-per-row variant; the workgroup-per-row variant (wide rows, three accumulator sets live) keeps it
    // in LDS instead -- 16 VGPRs less, which is the difference between 2 and 3 resident waves per SIMD for the fused-GELU backward
    float ag[VPL][8], ab[VPL][8], ad[DXSUM ? VPL : 1][8], gm[BLOCK ? 1 : VPL][8];
    if (BLOCK) {
        for (int i = threadIdx.x; i < cols; i += 256) red[i] = gamma[i];
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = v0 + vstep * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; if (!BLOCK) gm[i][e] = 0.f; if (DXSUM) ad[i][e] = 0.f; }
        if (!BLOCK && vi < nvec) ld8<float>(gamma + vi * 8, gm[i]);
    }
    const long rstep = BLOCK ? (long)gridDim.x : (long)gridDim.x * 4;
    long row = BLOCK ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave;
    raw8<T> nx[VPL], nd[VPL], nr[VPL];
    float nmean = 0.f, nrstd = 0.f;
    auto fetch = [&](long r) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) {
                nx[i].load(x + r * cols + vi * 8);
                nd[i].load(dy + r * cols + vi * 8);
                if (dres) nr[i].load(dres + r * cols + vi * 8);
            }
        }
        nmean = mean_in[r]; nrstd = rstd_in[r];
    };
    if (row < rows) fetch(row);
    for (; row < rows; row += rstep) {
        const float mean = nmean, rstd = nrstd;
        float zh[VPL][8], g[VPL][8], da[VPL][8], rs[VPL][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (v0 + vstep * i < nvec) {
                nx[i].unpack(zh[i]);   // x for now
                nd[i].unpack(g[i]);    // dy for now
                if (dres) nr[i].unpack(rs[i]);
            }
        }
        if (row + rstep < rows) fetch(row + rstep);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (v0 + vstep * i < nvec) {
                float gmv[8];
                if (BLOCK) ld8<float>(red + (v0 + vstep * i) * 8, gmv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float z;
                    const float dv = g[i][e];
                    act_fwd_grad<ACT>(zh[i][e], act, z, da[i][e]);
                    zh[i][e] = (z - mean) * rstd;
                    g[i][e] = dv * (BLOCK ? gmv[e] : gm[i][e]);
                    s1 += g[i][e];
                    s2 += g[i][e] * zh[i][e];
                    ag[i][e] += dv * zh[i][e];
                    ab[i][e] += dv;
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
                for (int e = 0; e < 8; ++e) {
                    o[e] = rstd * (g[i][e] - c1 - zh[i][e] * c2);
                    if (ACT != ANTMMF_ACT_NONE) o[e] *= da[i][e];
                    if (dres) o[e] += rs[i][e];
                    if (DXSUM) ad[i][e] += o[e];
                }
                st8<T>(dx + row * cols + vi * 8, o);
            }
        }
    }
    if (BLOCK && partials) {  // per-workgroup partial column sums [grid][NS][cols]
        float* pg = partials + (long)blockIdx.x * NS * cols;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) {
                st8<float>(pg + vi * 8, ag[i]); st8<float>(pg + cols + vi * 8, ab[i]);
                if (DXSUM) st8<float>(pg + 2 * cols + vi * 8, ad[i]);
            }
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
                    if (DXSUM) atomicAdd(&dxsum[vi * 8 + e], ad[i][e]);
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
                if (DXSUM) atomicAdd(&red[2 * cols + vi * 8 + e], ad[i][e]);
            }
        }
    }
    __syncthreads();
    if (partials) {
        float* pg = partials + (long)blockIdx.x * NS * cols;
        for (int i = threadIdx.x; i < NS * cols; i += 256) pg[i] = red[i];
        return;
    }
    for (int i = threadIdx.x; i < cols; i += 256) {
        if (dgamma) atomicAdd(&dgamma[i], red[i]);
        if (dbeta) atomicAdd(&dbeta[i], red[cols + i]);
        if (DXSUM) atomicAdd(&dxsum[i], red[2 * cols + i]);
    }
}

// out_k[c] += sum_b partials[b][k][c]  (k = 0 dgamma, 1 dbeta, 2 dxsum).  grid (ns*cols/256, 8): a workgroup sums 1/8 of the
// partial rows for 256 columns (4 row-slices x 64 float4 lanes), then one atomic per column.
__global__ __launch_bounds__(256) void ln_partials_reduce_kernel(const float* __restrict__ partials, int nblocks, int cols, int ns,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dxsum) {
    __shared__ float4 red[4][64];
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lx * 4;            // column in the concatenated [ns*cols] row
    const int per = (nblocks + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < ns * cols) {
        for (int b = b0 + ly; b < b1; b += 4) {
            const float4 v = *reinterpret_cast<const float4*>(partials + (long)b * ns * cols + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[ly][lx] = s;
    __syncthreads();
    if (ly == 0 && c < ns * cols) {
        const float4 a = red[0][lx], b = red[1][lx], d = red[2][lx], e = red[3][lx];
        const float v[4] = {a.x + b.x + d.x + e.x, a.y + b.y + d.y + e.y, a.z + b.z + d.z + e.z, a.w + b.w + d.w + e.w};
        const int k = c / cols;                          // cols % 8 == 0: the 4 columns stay inside one sum
        float* dst = k == 0 ? dgamma : k == 1 ? dbeta : dxsum;
        const int cc = c - k * cols;
        if (dst) {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(dst + cc + j, v[j]);
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
                         const void* dres, void* dx, float* dgamma, float* dbeta, float* dxsum, long rows, int cols, int act,
                         float* partials, long partial_elems, hipStream_t s) {
    const int nvec = cols / 8;
    const long want = (rows + 3) / 4;
    const bool wide = nvec > 128;
    const int ns = dxsum ? 3 : 2;
    // wave-per-row: 512 workgroups, closing LDS reduce + 2-3 * cols global atomics each (a bigger grid through the partials
    // scratch was measured SLOWER: the per-workgroup closing phase dominates).  workgroup-per-row: 1024 workgroups, partials.
    const int gw = (int)(want < 512 ? want : 512);
    const int gb = (int)(rows < 1024 ? rows : 1024);
    if (!(wide && partials && partial_elems >= (long)gb * ns * cols)) partials = nullptr;
    const size_t lds = (size_t)ns * cols * sizeof(float);
#define LN_BWD_D(V, BLK, GRID, LDS, A, D) hipLaunchKernelGGL((ln_bwd_kernel<T, V, BLK, A, D>), dim3(GRID), dim3(256), LDS, s, (const T*)dy, (const T*)x, mean, rstd, g, (const T*)dres, (T*)dx, dgamma, dbeta, dxsum, rows, cols, act, partials)
#define LN_BWD_A(V, BLK, GRID, LDS, A) do { if (dxsum) LN_BWD_D(V, BLK, GRID, LDS, A, true); else LN_BWD_D(V, BLK, GRID, LDS, A, false); } while (0)
#define LN_BWD(V, BLK, GRID, LDS) do { if (act == ANTMMF_ACT_NONE) LN_BWD_A(V, BLK, GRID, LDS, ANTMMF_ACT_NONE); else if (act == ANTMMF_ACT_GELU_ERF) LN_BWD_A(V, BLK, GRID, LDS, ANTMMF_ACT_GELU_ERF); else LN_BWD_A(V, BLK, GRID, LDS, -1); } while (0)
    if (nvec <= 64) LN_BWD(1, false, gw, lds);
    else if (nvec <= 128) LN_BWD(2, false, gw, lds);
    else if (nvec <= 256) LN_BWD(1, true, gb, (size_t)cols * sizeof(float));
    else if (nvec <= 512) LN_BWD(2, true, gb, (size_t)cols * sizeof(float));
    else return ANTMMF_EINVAL;
#undef LN_BWD
#undef LN_BWD_A
#undef LN_BWD_D
    if (partials && (dgamma || dbeta || dxsum))
        hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3((ns * cols + 255) / 256, 8), dim3(256), 0, s, partials, gb, cols, ns, dgamma, dbeta, dxsum);
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
                                        const void* dres, void* dx, float* dgamma, float* dbeta, float* dxsum, long rows, int cols, int act,
                                        int dtype, float* partials, long partial_elems, hipStream_t stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !ln_args_ok(rows, cols)) return ANTMMF_EINVAL;
    if (rows == 0) return ANTMMF_OK;
    return dtype == ANTMMF_BF16 ? ln_bwd_launch<bf16_t>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, dxsum, rows, cols, act, partials, partial_elems, stream)
         : dtype == ANTMMF_F32  ? ln_bwd_launch<float>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, dxsum, rows, cols, act, partials, partial_elems, stream)
                                : ANTMMF_EINVAL;
}

extern "C" int antmmf_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                    float* rstd, long rows, int cols, float eps, int dtype, hipStream_t stream) {
    return antmmf_act_layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, cols, eps, ANTMMF_ACT_NONE, dtype, stream);
}

extern "C" int antmmf_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                                    const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                                    long rows, int cols, int dtype, hipStream_t stream) {
    return antmmf_act_layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, nullptr, rows, cols, ANTMMF_ACT_NONE, dtype, nullptr, 0, stream);
}
