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
#include <cstdlib>

// ------------------------------------------------------------------ reductions over a "row group" (a wave or a 4-wave workgroup)
// Workgroup sums go through one of two LDS slots, alternated by the caller (`ph`): ONE barrier per reduction -- a slot is rewritten two
// reductions later, i.e. behind a barrier every wave has reached after its reads of that slot.
template <bool BLOCK>
__device__ __forceinline__ float row_sum(float v, float (*sh)[8], int& ph) {
    v = wave_sum(v);
    if (!BLOCK) return v;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s = sh[ph];
    ph ^= 1;
    if (lane == 0) s[wave] = v;
    __syncthreads();
    return (s[0] + s[1]) + (s[2] + s[3]);
}
template <bool BLOCK>
__device__ __forceinline__ void row_sum2(float& a, float& b, float (*sh)[8], int& ph) {
    a = wave_sum(a); b = wave_sum(b);
    if (!BLOCK) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s = sh[ph];
    ph ^= 1;
    if (lane == 0) { s[wave] = a; s[4 + wave] = b; }
    __syncthreads();
    a = (s[0] + s[1]) + (s[2] + s[3]);
    b = (s[4] + s[5]) + (s[6] + s[7]);
}

// raw (still packed) 8-element vector of a row: lets the NEXT row's loads be issued before the current row is touched
template <typename T> struct raw8;
template <> struct raw8<bf16_t> {
    uint4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = row_ld16(p); }
    __device__ __forceinline__ void unpack(f2_t (&f)[4]) const { f[0] = f2_bf(v.x); f[1] = f2_bf(v.y); f[2] = f2_bf(v.z); f[3] = f2_bf(v.w); }
};
template <> struct raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    __device__ __forceinline__ void unpack(f2_t (&f)[4]) const {
        f[0] = (f2_t){a.x, a.y}; f[1] = (f2_t){a.z, a.w}; f[2] = (f2_t){b.x, b.y}; f[3] = (f2_t){b.z, b.w};
    }
};

// BLOCK = false: wave per row, lane handles vectors lane + 64 i.  BLOCK = true: workgroup per row, thread handles vectors tid + 256 i.
// Both walk rows with a grid stride and keep the next row's loads in flight while the current one is reduced.  All per-element
// arithmetic is on f2_t (two elements per lane per VALU slot).
// ADJ (wave-per-row form, LAB library only): a wave owns two ADJACENT rows (4 KB of contiguous input for 1024 bf16 columns), requests both up front and never loops.  On the
// bare probe kernel of tools/stream_probe.hip that shape runs at 5.94 TB/s against 5.58 for one row per wave; in THIS kernel (statistics written, gamma / beta read per row)
// it measured 4 - 9 % SLOWER than one row per wave (profiles/r5_ln_fwd_adjacent_rows_ab.jsonl: 0.200 vs 0.192 ms on 263168 rows, 0.0686 vs 0.0624 on 78848), like two
// far-apart rows through the grid-stride prefetch (r5_ln_fwd_rows_per_wave_ab.jsonl): the product launches one row per wave
template <typename T, int VPL, bool BLOCK, int ACT, bool ADJ = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long rows, int cols, float eps, int act) {
    __shared__ float sh[2][8];
    int ph = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = cols >> 3;
    const int v0 = BLOCK ? threadIdx.x : lane, vstep = BLOCK ? 256 : 64;
    const float inv_n = 1.0f / (float)cols;
    const long rstep = ADJ ? 1L : (BLOCK ? (long)gridDim.x : (long)gridDim.x * 4);
    long row = ADJ ? ((long)blockIdx.x * 4 + wave) * 2 : (BLOCK ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave);
    // two rows in flight ahead of the one being reduced (BLOCK: 8 VGPRs per row and thread): nxa / nxb alternate
    raw8<T> nxa[VPL], nxb[VPL];
    auto fetch = [&](raw8<T> (&nx)[VPL], long r) {
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            if (v0 + vstep * i < nvec) nx[i].load(x + r * cols + (v0 + vstep * i) * 8);
    };
    auto body = [&](raw8<T> (&nx)[VPL], long row) {
        f2_t v[VPL][4];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (v0 + vstep * i < nvec) {
                nx[i].unpack(v[i]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = f2_splat(0.f);
            }
        }
        if (!ADJ && row + 2 * rstep < rows) fetch(nx, row + 2 * rstep);
        f2_t s2 = f2_splat(0.f);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (v0 + vstep * i < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { f2_t dz_unused; act_fwd_grad2<ACT>(v[i][e], act, v[i][e], dz_unused); s2 += v[i][e]; }
            }
        }
        const float mean = row_sum<BLOCK>(s2.x + s2.y, sh, ph) * inv_n;
        f2_t q2 = f2_splat(0.f);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (v0 + vstep * i < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[i][e] = v[i][e] - mean; q2 += v[i][e] * v[i][e]; }   // v is centred from here on
            }
        }
        const float rstd = rsqrtf(row_sum<BLOCK>(q2.x + q2.y, sh, ph) * inv_n + eps);
        T* yr = y + row * cols;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) {
                f2_t g[4], b[4], o[4];
                ld8_f2<float>(gamma + vi * 8, g);
                ld8_f2<float>(beta + vi * 8, b);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = v[i][e] * rstd * g[e] + b[e];
                st8_f2<T>(yr + vi * 8, o);
            }
        }
        if ((BLOCK ? threadIdx.x : lane) == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    };
    if (row < rows) fetch(nxa, row);
    if (row + rstep < rows) fetch(nxb, row + rstep);
    if (ADJ) {   // exactly this pair
        if (row < rows) body(nxa, row);
        if (row + 1 < rows) body(nxb, row + 1);
        return;
    }
    for (; row < rows; row += 2 * rstep) {
        body(nxa, row);
        if (row + rstep < rows) body(nxb, row + rstep);
    }
}

// z = act(x);  dz = rstd * (g*dy - mean(g*dy) - zhat * mean(g*dy*zhat));  dx = dz * act'(x) [+ dres];
// dgamma += sum_rows dy*zhat;  dbeta += sum_rows dy;  DXSUM: dxsum += sum_rows dx -- the bias gradient of the Linear whose
// output this LayerNorm reads (fc1 for the fused gelu + ffn_layernorm; the attention out-projection for ln2 with the
// residual gradient added), which would otherwise be a separate full read of dx.
// Column sums leave through `partials` ([grid][NS][cols], reduced by ln_partials_reduce_kernel) or, without it, atomics.
// YOUT (plain LayerNorm only): the forward output y = zhat * gamma + beta is written as well -- the backward pass of a transformer layer
// needs LN(x) again as the wgrad operand of the Linear behind it, and this kernel already holds zhat: one extra write instead of a separate
// recompute pass (read + write).
template <typename T, int VPL, bool BLOCK, int ACT, bool DXSUM, bool YOUT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dxsum, long rows, int cols, int act,
                                                     float* __restrict__ partials, const float* __restrict__ beta, T* __restrict__ yout) {
    constexpr int NS = DXSUM ? 3 : 2;
    ANTMMF_DYN_LDS(float, red);  // wave-per-row: [NS][cols] cross-wave column sums (+ [cols] beta with YOUT); workgroup-per-row: gamma (+ beta)
    __shared__ float sh[2][8];
    int ph = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = cols >> 3;
    const int v0 = BLOCK ? threadIdx.x : lane, vstep = BLOCK ? 256 : 64;
    const float inv_n = 1.0f / (float)cols;
    if (!BLOCK) {
        for (int i = threadIdx.x; i < NS * cols; i += 256) red[i] = 0.f;
    }
    // gamma: registers for the wave-per-row variant; the workgroup-per-row variant (wide rows, three accumulator sets live) keeps it
    // in LDS instead -- 16 VGPRs less, which is the difference between 2 and 3 resident waves per SIMD for the fused-GELU backward
    f2_t ag[VPL][4], ab[VPL][4], ad[DXSUM ? VPL : 1][4], gm[BLOCK ? 1 : VPL][4];
    float* const beta_s = red + (BLOCK ? 1 : NS) * cols;   // YOUT only
    if (BLOCK) {
        for (int i = threadIdx.x; i < cols; i += 256) red[i] = gamma[i];
    }
    if (YOUT) {
        for (int i = threadIdx.x; i < cols; i += 256) beta_s[i] = beta[i];
    }
    if (BLOCK || YOUT) __syncthreads();
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = v0 + vstep * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ag[i][e] = f2_splat(0.f); ab[i][e] = f2_splat(0.f);
            if (!BLOCK) gm[i][e] = f2_splat(0.f);
            if (DXSUM) ad[i][e] = f2_splat(0.f);
        }
        if (!BLOCK && vi < nvec) ld8_f2<float>(gamma + vi * 8, gm[i]);
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
        constexpr bool REG = BLOCK && ACT != ANTMMF_ACT_NONE;   // register diet of the wide fused-GELU variant: g = dy * gamma is rebuilt from the raw dy
        f2_t zh[VPL][4], g[REG ? 1 : VPL][4], da[VPL][4], rs[VPL][4];
        raw8<T> cd[REG ? VPL : 1];
        f2_t s1v = f2_splat(0.f), s2v = f2_splat(0.f);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (v0 + vstep * i < nvec) {
                nx[i].unpack(zh[i]);   // x for now
                if (REG) cd[i] = nd[i]; else nd[i].unpack(g[i]);    // dy for now
                if (dres) nr[i].unpack(rs[i]);
            }
        }
        if (row + rstep < rows) fetch(row + rstep);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (v0 + vstep * i < nvec) {
                f2_t gmv[4], yv[4];
                if (BLOCK) ld8_f2<float>(red + (v0 + vstep * i) * 8, gmv);
                if (YOUT) ld8_f2<float>(beta_s + (v0 + vstep * i) * 8, yv);
                f2_t dvv[4];
                if (REG) cd[i].unpack(dvv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f2_t z;
                    const f2_t dv = REG ? dvv[e] : g[i][e];
                    act_fwd_grad2<ACT>(zh[i][e], act, z, da[i][e]);
                    zh[i][e] = (z - mean) * rstd;
                    if (YOUT) yv[e] = zh[i][e] * (BLOCK ? gmv[e] : gm[i][e]) + yv[e];
                    const f2_t ge = dv * (BLOCK ? gmv[e] : gm[i][e]);
                    if (!REG) g[i][e] = ge;
                    s1v += ge;
                    s2v += ge * zh[i][e];
                    ag[i][e] += dv * zh[i][e];
                    ab[i][e] += dv;
                }
                if (YOUT) st8_f2<T>(yout + row * cols + (v0 + vstep * i) * 8, yv);
            }
        }
        float c1 = s1v.x + s1v.y, c2 = s2v.x + s2v.y;
        row_sum2<BLOCK>(c1, c2, sh, ph);
        c1 *= inv_n; c2 *= inv_n;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) {
                f2_t o[4], gg[4], gmv2[4];
                if (REG) { cd[i].unpack(gg); ld8_f2<float>(red + vi * 8, gmv2); }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f2_t ge = REG ? gg[e] * gmv2[e] : g[i][e];
                    o[e] = rstd * (ge - c1 - zh[i][e] * c2);
                    if (ACT != ANTMMF_ACT_NONE) o[e] *= da[i][e];
                    if (dres) o[e] += rs[i][e];
                    if (DXSUM) ad[i][e] += o[e];
                }
                st8_f2<T>(dx + row * cols + vi * 8, o);
            }
        }
    }
    if (BLOCK && partials) {  // per-workgroup partial column sums [grid][NS][cols]
        float* pg = partials + (long)blockIdx.x * NS * cols;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int vi = v0 + vstep * i;
            if (vi < nvec) {
                st8_f2<float>(pg + vi * 8, ag[i]); st8_f2<float>(pg + cols + vi * 8, ab[i]);
                if (DXSUM) st8_f2<float>(pg + 2 * cols + vi * 8, ad[i]);
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
                    if (dgamma) atomicAdd(&dgamma[vi * 8 + e], ag[i][e >> 1][e & 1]);
                    if (dbeta) atomicAdd(&dbeta[vi * 8 + e], ab[i][e >> 1][e & 1]);
                    if (DXSUM) atomicAdd(&dxsum[vi * 8 + e], ad[i][e >> 1][e & 1]);
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
                atomicAdd(&red[vi * 8 + e], ag[i][e >> 1][e & 1]);
                atomicAdd(&red[cols + vi * 8 + e], ab[i][e >> 1][e & 1]);
                if (DXSUM) atomicAdd(&red[2 * cols + vi * 8 + e], ad[i][e >> 1][e & 1]);
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

#define LN_FWD_ROWS_PER_WAVE 1   /* product default of the wave-per-row forward forms (see ln_fwd_launch) */
template <typename T>
static int ln_fwd_launch(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, long rows,
                         int cols, float eps, int act, hipStream_t s) {
    const int nvec = cols / 8;
#define LN_FWD_A(V, BLK, GRID, A) hipLaunchKernelGGL((ln_fwd_kernel<T, V, BLK, A>), dim3(GRID), dim3(256), 0, s, (const T*)x, g, b, (T*)y, mean, rstd, rows, cols, eps, act)
#define LN_FWD(V, BLK, GRID) do { if (act == ANTMMF_ACT_NONE) LN_FWD_A(V, BLK, GRID, ANTMMF_ACT_NONE); else if (act == ANTMMF_ACT_GELU_ERF) LN_FWD_A(V, BLK, GRID, ANTMMF_ACT_GELU_ERF); else LN_FWD_A(V, BLK, GRID, -1); } while (0)
    // one row per wave / workgroup, no grid-stride loop in practice (the loop only runs beyond 2^20 workgroups): measured against 4096
    // persistent workgroups walking rows with a two-row prefetch, the hardware's own workgroup turnover is 13-15 % faster on the plain
    // kernels (263168 x 1024: 0.220 -> 0.191 ms = 0.95x a torch copy of the same bytes; x 4096: 1.06 -> 0.91 ms) and 5 % on the GELU one
    const long grid_cap = 1L << 20;
    // rows per wave of the wave-per-row forms (the kernel keeps two rows' loads in flight: with half the grid every wave requests BOTH its rows up front and never loops).
    // LAB: ANTMMF_LN_FWD_RPW selects 1 / 2 / 4 for the A/B (tools/ln_bench.py, profiles/r5_ln_fwd_rows_per_wave_ab.jsonl)
    static const char* rpw_env = ANTMMF_LAB_ENV("ANTMMF_LN_FWD_RPW");
    const int rpw = rpw_env ? (atoi(rpw_env) > 0 ? atoi(rpw_env) : 1) : LN_FWD_ROWS_PER_WAVE;
    const long want_w = (rows + 4L * rpw - 1) / (4L * rpw);
    const int gw = (int)(want_w < grid_cap ? want_w : grid_cap), gb = (int)(rows < grid_cap ? rows : grid_cap);
    if (nvec <= 64) LN_FWD(1, false, gw);
#ifdef ANTMMF_LAB
    else if (nvec <= 128 && ANTMMF_LAB_ENV("ANTMMF_LN_FWD_ADJ") && ANTMMF_LAB_ENV("ANTMMF_LN_FWD_ADJ")[0] == '1' && act == ANTMMF_ACT_NONE && rows >= 256 && (rows + 7) / 8 <= grid_cap)
        hipLaunchKernelGGL((ln_fwd_kernel<T, 2, false, ANTMMF_ACT_NONE, true>), dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, s, (const T*)x, g, b, (T*)y, mean, rstd, rows, cols, eps, act);   // A/B only
#endif
    else if (nvec <= 128) LN_FWD(2, false, gw);
    else if (nvec <= 256) LN_FWD(1, true, gb);
    else if (nvec <= 512) {
        // 4096-wide bf16 rows (the M2 feed-forward's gelu -> ffn_layernorm): a WAVE per row with the whole row in its registers (8 vectors per lane; 204 VGPRs, two waves per
        // SIMD) beats the workgroup-per-row form with its two block reductions and barriers: 1.02 -> 0.97 ms on 263168 rows, 0.312 -> 0.283 ms on 78848 (round 3, same box,
        // insensitive to an occupancy limiter); fp32 rows would need > 256 VGPRs and keep the workgroup form
        if (sizeof(T) == 2 && nvec > 256) LN_FWD(8, false, gw); else LN_FWD(2, true, gb);
    }
    else return ANTMMF_EINVAL;
#undef LN_FWD
#undef LN_FWD_A
    return antmmf_check_launch();
}

// A persistent grid should be exactly what is RESIDENT: the workgroup-per-row backward launched 1024 workgroups, but at 168 VGPRs (the fused-GELU variant) only three
// 256-thread workgroups fit a CU -- 768 resident, the last 256 ran as a second, one-third-full round (round 3: 1.446 -> 1.339 ms on 263168 x 4096, 0.468 -> 0.439 ms on
// 78848 rows with a 768-workgroup grid; 512 and 1536 are both slower).  Asked from the runtime per kernel variant (they differ in registers), capped by `cap`.
template <typename K>
static int ln_resident_grid(K kern, size_t lds, int cap) {
#ifdef ANTMMF_EMULATE
    (void)kern; (void)lds;
    return cap;
#else
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess || per_cu <= 0) return cap;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return cap;
    const long g = (long)per_cu * cus;
    return (int)(g < cap ? g : cap);
#endif
}

template <typename T>
static int ln_bwd_launch(const void* dy, const void* x, const float* mean, const float* rstd, const float* g,
                         const void* dres, void* dx, float* dgamma, float* dbeta, float* dxsum, long rows, int cols, int act,
                         float* partials, long partial_elems, const float* beta, void* yout, hipStream_t s) {
    const int nvec = cols / 8;
    const long want = (rows + 3) / 4;
    const bool wide = nvec > 128;
    const int ns = dxsum ? 3 : 2;
    // wave-per-row: 512 workgroups, closing LDS reduce + 2-3 * cols global atomics each (a bigger grid through the partials
    // scratch was measured SLOWER: the per-workgroup closing phase dominates).  workgroup-per-row: 1024 workgroups, partials.
    // (re-measured in round 2 with 2x ... 16x the workgroups: 0.31 -> 0.35 / 0.41 / 0.52 / 0.76 ms for 263168 x 1024, 1.58 -> 1.58 / 1.61 / 1.69 / 1.86 ms
    // for the 4096-wide GELU backward -- the column-sum closing phase is per workgroup)
    // LAB: ANTMMF_ROW_CUS = n sizes the persistent backward grids for n CUs instead of the whole chip (tools/overlap_probe.py: the kernel on a CU-masked stream beside a GEMM)
    static const char* cus_env = ANTMMF_LAB_ENV("ANTMMF_ROW_CUS");
    const int row_cus = cus_env ? atoi(cus_env) : 0;
    const int gw_cap = row_cus > 0 ? 2 * row_cus : 512;
    const int gw = (int)(want < gw_cap ? want : gw_cap);
    const int gb = (int)(rows < 1024 ? rows : 1024);
    int gb_used = gb;   // the grid the workgroup-per-row kernel was launched with (<= gb: what is resident), = the number of partial rows it wrote
    if (!(wide && partials && partial_elems >= (long)gb * ns * cols)) partials = nullptr;
    const size_t lds = (size_t)(ns + (yout ? 1 : 0)) * cols * sizeof(float);
    const size_t ldsb = (size_t)(1 + (yout ? 1 : 0)) * cols * sizeof(float);
#define LN_BWD_Y(V, BLK, GRID, LDS, A, D, Y) \
    do { \
        int grid_ = GRID; \
        if (BLK) { static int res_ = 0; if (!res_) res_ = ln_resident_grid(ln_bwd_kernel<T, V, BLK, A, D, Y>, LDS, 1024); grid_ = grid_ < res_ ? grid_ : res_; if (row_cus > 0 && grid_ > res_ / 256 * row_cus) grid_ = res_ / 256 * row_cus; gb_used = grid_; } \
        LN_BWD_Z(V, BLK, grid_, LDS, A, D, Y); \
    } while (0)
#define LN_BWD_Z(V, BLK, GRID, LDS, A, D, Y) hipLaunchKernelGGL((ln_bwd_kernel<T, V, BLK, A, D, Y>), dim3(GRID), dim3(256), LDS, s, (const T*)dy, (const T*)x, mean, rstd, g, (const T*)dres, (T*)dx, dgamma, dbeta, dxsum, rows, cols, act, partials, beta, (T*)yout)
#define LN_BWD_D(V, BLK, GRID, LDS, A, D) do { if (yout && A == ANTMMF_ACT_NONE) LN_BWD_Y(V, BLK, GRID, LDS, ANTMMF_ACT_NONE, D, true); else LN_BWD_Y(V, BLK, GRID, LDS, A, D, false); } while (0)
#define LN_BWD_A(V, BLK, GRID, LDS, A) do { if (dxsum) LN_BWD_D(V, BLK, GRID, LDS, A, true); else LN_BWD_D(V, BLK, GRID, LDS, A, false); } while (0)
#define LN_BWD(V, BLK, GRID, LDS) do { if (act == ANTMMF_ACT_NONE) LN_BWD_A(V, BLK, GRID, LDS, ANTMMF_ACT_NONE); else if (act == ANTMMF_ACT_GELU_ERF) LN_BWD_A(V, BLK, GRID, LDS, ANTMMF_ACT_GELU_ERF); else LN_BWD_A(V, BLK, GRID, LDS, -1); } while (0)
    if (nvec <= 64) LN_BWD(1, false, gw, lds);
    else if (nvec <= 128) LN_BWD(2, false, gw, lds);
    else if (nvec <= 256) LN_BWD(1, true, gb, ldsb);
    else if (nvec <= 512) LN_BWD(2, true, gb, ldsb);
    else return ANTMMF_EINVAL;
#undef LN_BWD
#undef LN_BWD_A
#undef LN_BWD_D
#undef LN_BWD_Y
#undef LN_BWD_Z
    if (partials && (dgamma || dbeta || dxsum))
        hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3((ns * cols + 255) / 256, 8), dim3(256), 0, s, partials, gb_used, cols, ns, dgamma, dbeta, dxsum);
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
    return dtype == ANTMMF_BF16 ? ln_bwd_launch<bf16_t>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, dxsum, rows, cols, act, partials, partial_elems, nullptr, nullptr, stream)
         : dtype == ANTMMF_F32  ? ln_bwd_launch<float>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, dxsum, rows, cols, act, partials, partial_elems, nullptr, nullptr, stream)
                                : ANTMMF_EINVAL;
}

// plain LayerNorm backward that ALSO re-emits the forward output y = LN(x) (the layer backward needs it as a wgrad operand)
extern "C" int antmmf_layernorm_bwd_renorm(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                           const float* beta, const void* dres, void* dx, void* y, float* dgamma, float* dbeta, float* dxsum,
                                           long rows, int cols, int dtype, hipStream_t stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !beta || !dx || !y || !ln_args_ok(rows, cols)) return ANTMMF_EINVAL;
    if (rows == 0) return ANTMMF_OK;
    return dtype == ANTMMF_BF16 ? ln_bwd_launch<bf16_t>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, dxsum, rows, cols, ANTMMF_ACT_NONE, nullptr, 0, beta, y, stream)
         : dtype == ANTMMF_F32  ? ln_bwd_launch<float>(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, dxsum, rows, cols, ANTMMF_ACT_NONE, nullptr, 0, beta, y, stream)
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
