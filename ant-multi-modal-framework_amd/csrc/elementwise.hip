// elementwise.hip -- HBM-bound helpers of the contrastive step (gfx950): activations, L2 normalise,
// bias-gradient column sums, weight transposes, patch extraction, token assembly, embedding
// gather / scatter, fused AdamW over a flat parameter arena.
//
// Everything here is byte-moving work: 16-B vector accesses per lane, grid-stride loops capped at
// 2048 workgroups (256 CUs x 8), no LDS unless a transpose needs it.
#include "common.h"

static inline int ew_grid(long nvec) { long g = (nvec + 255) / 256; return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

// ------------------------------------------------------------------ activation forward / backward
// reference: QuickGELU clip/model.py:222-224; gelu(erf) modeling_bert.py:31-37; F.gelu feedforward_network.py:120
template <typename T>
__global__ __launch_bounds__(256) void act_fwd_kernel(const T* __restrict__ u, T* __restrict__ g, long n, int act) {
    const long nv = n >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        float v[8];
        ld8<T>(u + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act_fwd(v[e], act);
        st8<T>(g + i * 8, v);
    }
}
// du = dg * act'(u)
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dg, const T* __restrict__ u, T* __restrict__ du, long n, int act) {
    const long nv = n >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        float a[8], b[8];
        ld8<T>(dg + i * 8, a);
        ld8<T>(u + i * 8, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= act_grad(b[e], act);
        st8<T>(du + i * 8, a);
    }
}
extern "C" int antmmf_act_fwd(const void* u, void* g, long n, int act, int dtype, hipStream_t s) {
    if (!u || !g || n < 0 || (n & 7)) return ANTMMF_EINVAL;
    if (!n) return ANTMMF_OK;
    if (dtype == ANTMMF_BF16) hipLaunchKernelGGL(act_fwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (const bf16_t*)u, (bf16_t*)g, n, act);
    else if (dtype == ANTMMF_F32) hipLaunchKernelGGL(act_fwd_kernel<float>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (const float*)u, (float*)g, n, act);
    else return ANTMMF_EINVAL;
    return antmmf_check_launch();
}
extern "C" int antmmf_act_bwd(const void* dg, const void* u, void* du, long n, int act, int dtype, hipStream_t s) {
    if (!dg || !u || !du || n < 0 || (n & 7)) return ANTMMF_EINVAL;
    if (!n) return ANTMMF_OK;
    if (dtype == ANTMMF_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (const bf16_t*)dg, (const bf16_t*)u, (bf16_t*)du, n, act);
    else if (dtype == ANTMMF_F32) hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (const float*)dg, (const float*)u, (float*)du, n, act);
    else return ANTMMF_EINVAL;
    return antmmf_check_launch();
}

// ------------------------------------------------------------------ L2 normalise rows (one wave per row)
// reference: F.normalize(x, p=2, dim=-1) univl_video_base.py:114,158 (eps 1e-12); x / x.norm() vlmo_module.py:346-349 (eps 0)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const TI* __restrict__ x, TO* __restrict__ y, float* __restrict__ inv_norm,
                                                         long rows, int cols, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) { const float v = ld1<TI>(x + row * cols + c); s += v * v; }
        const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), eps);
        for (int c = lane; c < cols; c += 64) st1<TO>(y + row * cols + c, ld1<TI>(x + row * cols + c) * inv);
        if (lane == 0 && inv_norm) inv_norm[row] = inv;
    }
}
// y = x * inv ; dx = inv * (dy - y * <dy, y>)      (exact where the eps clamp is inactive)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const TO* __restrict__ dy, const TO* __restrict__ y, const float* __restrict__ inv_norm,
                                                         TI* __restrict__ dx, long rows, int cols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += ld1<TO>(dy + row * cols + c) * ld1<TO>(y + row * cols + c);
        s = wave_sum(s);
        const float inv = inv_norm[row];
        for (int c = lane; c < cols; c += 64)
            st1<TI>(dx + row * cols + c, inv * (ld1<TO>(dy + row * cols + c) - ld1<TO>(y + row * cols + c) * s));
    }
}
// in_dtype: dtype of x/dx; out_dtype: dtype of y/dy
extern "C" int antmmf_l2norm_fwd(const void* x, void* y, float* inv_norm, long rows, int cols, float eps, int in_dtype, int out_dtype, hipStream_t s) {
    if (!x || !y || rows < 0 || cols <= 0) return ANTMMF_EINVAL;
    if (!rows) return ANTMMF_OK;
    const int grid = (int)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
#define L2F(TI, TO) hipLaunchKernelGGL((l2norm_fwd_kernel<TI, TO>), dim3(grid), dim3(256), 0, s, (const TI*)x, (TO*)y, inv_norm, rows, cols, eps)
    if (in_dtype == ANTMMF_BF16 && out_dtype == ANTMMF_BF16) L2F(bf16_t, bf16_t);
    else if (in_dtype == ANTMMF_BF16 && out_dtype == ANTMMF_F32) L2F(bf16_t, float);
    else if (in_dtype == ANTMMF_F32 && out_dtype == ANTMMF_F32) L2F(float, float);
    else if (in_dtype == ANTMMF_F32 && out_dtype == ANTMMF_BF16) L2F(float, bf16_t);
    else return ANTMMF_EINVAL;
#undef L2F
    return antmmf_check_launch();
}
extern "C" int antmmf_l2norm_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, long rows, int cols, int in_dtype, int out_dtype, hipStream_t s) {
    if (!dy || !y || !inv_norm || !dx || rows < 0 || cols <= 0) return ANTMMF_EINVAL;
    if (!rows) return ANTMMF_OK;
    const int grid = (int)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
#define L2B(TI, TO) hipLaunchKernelGGL((l2norm_bwd_kernel<TI, TO>), dim3(grid), dim3(256), 0, s, (const TO*)dy, (const TO*)y, inv_norm, (TI*)dx, rows, cols)
    if (in_dtype == ANTMMF_BF16 && out_dtype == ANTMMF_BF16) L2B(bf16_t, bf16_t);
    else if (in_dtype == ANTMMF_BF16 && out_dtype == ANTMMF_F32) L2B(bf16_t, float);
    else if (in_dtype == ANTMMF_F32 && out_dtype == ANTMMF_F32) L2B(float, float);
    else if (in_dtype == ANTMMF_F32 && out_dtype == ANTMMF_BF16) L2B(float, bf16_t);
    else return ANTMMF_EINVAL;
#undef L2B
    return antmmf_check_launch();
}

// ------------------------------------------------------------------ column sum: out[c] (+)= sum_r x[r][c]   (bias / pos-emb gradients)
// Each workgroup takes a slab of rows and 512 columns (a lane owns 8 columns via one 16-B load), sums in fp32
// registers, then one atomicAdd per column per workgroup.  `out` must be zero-initialised or hold a running sum.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long rows, int cols, long ld, long rows_per_block) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 512 + lane * 8;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c0 < cols) {
        // four rows per trip, all four loads requested before the first add (the one-load-per-trip loop was load -> s_waitcnt vmcnt(0) -> add)
        long r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) ld8<T>(x + (r + 4 * u) * ld + c0, v[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (v[0][e] + v[1][e]) + (v[2][e] + v[3][e]);
        }
        for (; r < r1; r += 4) {
            float v[8];
            ld8<T>(x + r * ld + c0, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = acc[e];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int c = blockIdx.x * 512 + i;
        if (c < cols) atomicAdd(&out[c], red[0][i] + red[1][i] + red[2][i] + red[3][i]);
    }
}
extern "C" int antmmf_colsum(const void* x, float* out, long rows, int cols, long ld, int dtype, hipStream_t s) {
    if (!x || !out || rows < 0 || cols <= 0 || (cols & 7) || (ld & 7)) return ANTMMF_EINVAL;
    if (!rows) return ANTMMF_OK;
    const int gx = (cols + 511) / 512;
    int gy = 2048 / gx; if (gy < 1) gy = 1;
    long rpb = (rows + gy - 1) / gy; if (rpb < 16) rpb = 16;
    gy = (int)((rows + rpb - 1) / rpb);
    if (dtype == ANTMMF_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(gx, gy), dim3(256), 0, s, (const bf16_t*)x, out, rows, cols, ld, rpb);
    else if (dtype == ANTMMF_F32) hipLaunchKernelGGL(colsum_kernel<float>, dim3(gx, gy), dim3(256), 0, s, (const float*)x, out, rows, cols, ld, rpb);
    else return ANTMMF_EINVAL;
    return antmmf_check_launch();
}

// ------------------------------------------------------------------ 2-D transpose of a bf16 matrix  out[c][r] = in[r][c]   (weights, for dX = dY W)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows, int cols) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? in[(long)(r0 + r) * cols + c0 + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r0 + r < rows && c0 + c < cols) out[(long)(c0 + c) * rows + r0 + r] = tile[r][c];
    }
}
extern "C" int antmmf_transpose_bf16(const void* in, void* out, int rows, int cols, hipStream_t s) {
    if (!in || !out || rows <= 0 || cols <= 0) return ANTMMF_EINVAL;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, rows, cols);
    return antmmf_check_launch();
}

// Every transposed weight copy of a step in ONE launch: `table` (device, int64 x 5 per matrix: element offsets into in_base / out_base, rows,
// cols, index of the matrix's first 64 x 64 tile in the launch) -- the per-weight launches were 274 x 10 us per step for 0.9 GB of data.
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const bf16_t* __restrict__ in_base, bf16_t* __restrict__ out_base,
                                                                     const long* __restrict__ table, int n_mats) {
    __shared__ bf16_t tile[64][66];
    int lo = 0, hi = n_mats - 1;                 // last matrix whose first tile is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 5 + 4] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long* d = table + lo * 5;
    const bf16_t* in = in_base + d[0];
    bf16_t* out = out_base + d[1];
    const int rows = (int)d[2], cols = (int)d[3], t = (int)((long)blockIdx.x - d[4]);
    const int tx = (cols + 63) >> 6;
    const int r0 = (t / tx) * 64, c0 = (t % tx) * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? in[(long)(r0 + r) * cols + c0 + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (r0 + r < rows && c0 + c < cols) out[(long)(c0 + c) * rows + r0 + r] = tile[r][c];
    }
}
extern "C" int antmmf_transpose_bf16_batched(const void* in_base, void* out_base, const long* table, int n_mats, long total_tiles, hipStream_t s) {
    if (!in_base || !out_base || !table || n_mats <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffL) return ANTMMF_EINVAL;
    hipLaunchKernelGGL(transpose_bf16_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, (const bf16_t*)in_base, (bf16_t*)out_base, table, n_mats);
    return antmmf_check_launch();
}

// ------------------------------------------------------------------ cast fp32 -> bf16 (flat)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long)gridDim.x * 256 * 8) {
        if (i + 8 <= n) {
            float v[8];
            ld8<float>(in + i, v);
            st8<bf16_t>(out + i, v);
        } else {
            for (long j = i; j < n; ++j) out[j] = f2bf(in[j]);
        }
    }
}
extern "C" int antmmf_cast_f32_bf16(const float* in, void* out, long n, hipStream_t s) {
    if (!in || !out || n < 0) return ANTMMF_EINVAL;
    if (!n) return ANTMMF_OK;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid((n + 7) / 8)), dim3(256), 0, s, in, (bf16_t*)out, n);
    return antmmf_check_launch();
}

// ------------------------------------------------------------------ hi / lo split of an fp32 operand (the fp32-accurate similarity GEMMs: x = hi + lo + O(2^-16 |x|))
// x [rows, cols] fp32 (row stride ldx) -> hi = bf16(x), lo = bf16(x - float(hi)), both [rows_pad, cols_pad] dense, zero outside the operand.  One pass instead of the
// cast / cast back / subtract / cast (+ pad) launches torch makes of it; same round-to-nearest-even, bit-identical results.  cols_pad is a multiple of 8: a thread
// writes one 16-byte vector of each output.  Replaces (reference, stock torch ops): the fp32 `torch.matmul` of L2-normalised embeddings, univl_video_ret.py:357-387.
__global__ __launch_bounds__(256) void split_hi_lo_kernel(const float* __restrict__ x, long ldx, int rows, int cols, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                                          int rows_pad, int cols_pad) {
    const int vpr = cols_pad >> 3;
    const long total = (long)rows_pad * vpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / vpr), c0 = (int)(i % vpr) << 3;
        float v[8], h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (r < rows && c0 + e < cols) ? x[(long)r * ldx + c0 + e] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { h[e] = bf2f(f2bf(v[e])); l[e] = v[e] - h[e]; }
        st8<bf16_t>(hi + (long)r * cols_pad + c0, h);
        st8<bf16_t>(lo + (long)r * cols_pad + c0, l);
    }
}
extern "C" int antmmf_split_hi_lo_bf16(const float* x, long ldx, int rows, int cols, void* hi, void* lo, int rows_pad, int cols_pad, hipStream_t s) {
    if (!x || !hi || !lo || rows < 0 || cols < 0 || rows_pad < rows || cols_pad < cols || (cols_pad & 7) || ldx < cols) return ANTMMF_EINVAL;
    if (!rows_pad || !cols_pad) return ANTMMF_OK;
    hipLaunchKernelGGL(split_hi_lo_kernel, dim3(ew_grid((long)rows_pad * (cols_pad >> 3))), dim3(256), 0, s, x, ldx, rows, cols, (bf16_t*)hi, (bf16_t*)lo, rows_pad, cols_pad);
    return antmmf_check_launch();
}

// ------------------------------------------------------------------ patch extraction (im2col for a stride == kernel conv)
// image [B, C, H, W] (fp32 or bf16, optional affine (x - shift) * scale: M2's inception normalise) ->
// patches [B * Gh * Gw, Kpad] bf16, inner order (c, py, px) == Conv2d weight.flatten(1); columns >= C*P*P are zero.
// reference: nn.Conv2d(3, W, k=P, s=P) clip/model.py:289-295,310-312; VisionEmbedding.proj embedding.py:49,69; img_norm transforms/utils.py:48
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ img, bf16_t* __restrict__ out, int B, int C, int H, int W, int P,
                                                       int kpad, float shift, float scale) {
    const int gw = W / P, gh = H / P;
    const long total = (long)B * gh * gw * kpad;
    const int kk = C * P * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int col = (int)(i % kpad);
        const long prow = i / kpad;
        float v = 0.f;
        if (col < kk) {
            const int px = col % P, py = (col / P) % P, c = col / (P * P);
            const int gx = (int)(prow % gw), gy = (int)((prow / gw) % gh);
            const long b = prow / ((long)gw * gh);
            v = (ld1<T>(img + ((b * C + c) * H + gy * P + py) * W + gx * P + px) - shift) * scale;
        }
        out[i] = f2bf(v);
    }
}
extern "C" int antmmf_patchify(const void* img, void* out, int B, int C, int H, int W, int P, int kpad, float shift, float scale, int dtype, hipStream_t s) {
    if (!img || !out || B <= 0 || P <= 0 || H % P || W % P || kpad < C * P * P || (kpad & 7)) return ANTMMF_EINVAL;
    const long total = (long)B * (H / P) * (W / P) * kpad;
    if (dtype == ANTMMF_BF16) hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, (const bf16_t*)img, (bf16_t*)out, B, C, H, W, P, kpad, shift, scale);
    else if (dtype == ANTMMF_F32) hipLaunchKernelGGL(patchify_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, s, (const float*)img, (bf16_t*)out, B, C, H, W, P, kpad, shift, scale);
    else return ANTMMF_EINVAL;
    return antmmf_check_launch();
}

// ------------------------------------------------------------------ token assembly: x[b, 0] = cls + pos[0]; x[b, 1+p] = patch[b, p] (+ bias) + pos[1+p]
// reference: cat([class_embedding, x]) + positional_embedding clip/model.py:313-323; VisionEmbedding cls cat embedding.py:80-83 + PositionalEmbedding :92-110
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const bf16_t* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ out, long B, int G, int d) {
    const int dv = d >> 3;
    const long total = B * (G + 1) * dv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int v = (int)(i % dv);
        const long tok = i / dv;
        const int t = (int)(tok % (G + 1));
        const long b = tok / (G + 1);
        float o[8], p[8];
        if (t == 0) ld8<float>(cls + v * 8, o);
        else {
            ld8<bf16_t>(patch + ((b * G + t - 1) * d) + v * 8, o);
            if (bias) { float bb[8]; ld8<float>(bias + v * 8, bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += bb[e]; }
        }
        if (pos) { ld8<float>(pos + (long)t * d + v * 8, p);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += p[e]; }
        st8<bf16_t>(out + tok * d + v * 8, o);
    }
}
extern "C" int antmmf_assemble_tokens(const void* patch, const float* cls, const float* pos, const float* bias, void* out, long B, int G, int d, hipStream_t s) {
    if (!patch || !cls || !out || B <= 0 || G <= 0 || d <= 0 || (d & 7)) return ANTMMF_EINVAL;
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3(ew_grid(B * (G + 1) * (d / 8))), dim3(256), 0, s, (const bf16_t*)patch, cls, pos, bias, (bf16_t*)out, B, G, d);
    return antmmf_check_launch();
}
// backward of the assembly: dpatch[b, p] = dx[b, 1+p] (contiguous copy for the patch GEMM's dW);  the cls / pos / bias
// gradients are column sums of dx taken with antmmf_colsum on strided views by the host.
__global__ __launch_bounds__(256) void split_tokens_kernel(const bf16_t* __restrict__ dx, bf16_t* __restrict__ dpatch, long B, int G, int d) {
    const int dv = d >> 3;
    const long total = B * G * dv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int v = (int)(i % dv);
        const long tok = i / dv;
        const int p = (int)(tok % G);
        const long b = tok / G;
        *reinterpret_cast<uint4*>(dpatch + tok * d + v * 8) = *reinterpret_cast<const uint4*>(dx + ((b * (G + 1) + 1 + p) * d) + v * 8);
    }
}
extern "C" int antmmf_split_tokens(const void* dx, void* dpatch, long B, int G, int d, hipStream_t s) {
    if (!dx || !dpatch || B <= 0 || G <= 0 || d <= 0 || (d & 7)) return ANTMMF_EINVAL;
    hipLaunchKernelGGL(split_tokens_kernel, dim3(ew_grid(B * G * (d / 8))), dim3(256), 0, s, (const bf16_t*)dx, (bf16_t*)dpatch, B, G, d);
    return antmmf_check_launch();
}

// ------------------------------------------------------------------ embedding gather (sum of up to three tables) and scatter-add of its gradient
// out[r] = word[ids[r]] (+ pos[pos_ids ? pos_ids[r] : pos_offset + r % seq]) (+ type[type_ids ? type_ids[r] : 0]); padded rows can be zeroed (M2: encoder.py:440)
// reference: BertEmbeddings clip_text_encoder.py:36-60; TextEmbedding / PositionalEmbedding embedding.py:86-110
__global__ __launch_bounds__(256) void embed_gather_kernel(const long* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
                                                           const float* __restrict__ type, const long* __restrict__ type_ids,
                                                           const unsigned char* __restrict__ zero_rows, bf16_t* __restrict__ out,
                                                           long rows, int seq, int d, int pos_offset) {
    const int dv = d >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * dv; i += (long)gridDim.x * 256) {
        const int v = (int)(i % dv);
        const long r = i / dv;
        float o[8], t[8];
        ld8<float>(word + ids[r] * d + v * 8, o);
        if (pos) { ld8<float>(pos + (long)(pos_offset + (int)(r % seq)) * d + v * 8, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += t[e]; }
        if (type) { ld8<float>(type + (type_ids ? type_ids[r] : 0) * d + v * 8, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += t[e]; }
        if (zero_rows && zero_rows[r]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f; }
        st8<bf16_t>(out + r * d + v * 8, o);
    }
}
extern "C" int antmmf_embed_gather(const long* ids, const float* word, const float* pos, const float* type, const long* type_ids,
                                   const unsigned char* zero_rows, void* out, long rows, int seq, int d, int pos_offset, hipStream_t s) {
    if (!ids || !word || !out || rows < 0 || seq <= 0 || d <= 0 || (d & 7)) return ANTMMF_EINVAL;
    if (!rows) return ANTMMF_OK;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(ew_grid(rows * (d / 8))), dim3(256), 0, s, ids, word, pos, type, type_ids, zero_rows, (bf16_t*)out, rows, seq, d, pos_offset);
    return antmmf_check_launch();
}
// dtable[idx[r]] += dx[r]  (fp32 atomics; idx == NULL means idx[r] = offset + r % seq, i.e. the position table)
__global__ __launch_bounds__(256) void embed_scatter_kernel(const bf16_t* __restrict__ dx, const long* __restrict__ idx, const unsigned char* __restrict__ skip_rows,
                                                            float* __restrict__ dtable, long rows, int seq, int d, int offset) {
    const int dv = d >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * dv; i += (long)gridDim.x * 256) {
        const int v = (int)(i % dv);
        const long r = i / dv;
        if (skip_rows && skip_rows[r]) continue;
        const long t = idx ? idx[r] : (long)(offset + (int)(r % seq));
        float g[8];
        ld8<bf16_t>(dx + r * d + v * 8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(dtable + t * d + v * 8 + e, g[e]);
    }
}
// The word-table case without atomics: the caller sorts the token ids (sorted_idx ascending, src_row = the permutation; skipped rows carry an id >= n_table and sort to
// the end).  A wave that finds the HEAD of a run of equal ids sums that run's rows in registers (16-B row reads, 512 columns per trip) and adds the total into the table
// row -- exactly one writer per table row, so no atomics (the element-wise atomic form above: 80 M fp32 atomic adds = 1.19 ms at the bench size; this: ~0.1 ms incl. the sort).
__global__ __launch_bounds__(256) void embed_scatter_sorted_kernel(const bf16_t* __restrict__ dx, const long* __restrict__ sorted_idx, const long* __restrict__ src_row,
                                                                   float* __restrict__ dtable, long rows, long n_table, int d) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    for (long j = wid; j < rows; j += nw) {
        const long id = sorted_idx[j];
        if (id < 0 || id >= n_table) continue;
        if (j > 0 && sorted_idx[j - 1] == id) continue;          // not a run head (wave-uniform)
        long e = j + 1;
        while (e < rows && sorted_idx[e] == id) ++e;
        for (int c0 = lane * 8; c0 < d; c0 += 512) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (long k = j; k < e; ++k) {
                float g[8];
                ld8<bf16_t>(dx + src_row[k] * d + c0, g);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += g[q];
            }
            float* t = dtable + id * d + c0;
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] += acc[q];
        }
    }
}
extern "C" int antmmf_embed_scatter_add_sorted(const void* dx, const long* sorted_idx, const long* src_row, float* dtable, long rows, long n_table, int d, hipStream_t s) {
    if (!dx || !sorted_idx || !src_row || !dtable || rows < 0 || n_table <= 0 || d <= 0 || (d & 7)) return ANTMMF_EINVAL;
    if (!rows) return ANTMMF_OK;
    long blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(embed_scatter_sorted_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)dx, sorted_idx, src_row, dtable, rows, n_table, d);
    return antmmf_check_launch();
}
// The position-table case (idx == NULL: row r goes to table row offset + r % seq): every table row receives rows / seq contributions, so the element-wise
// atomics above were 1024 fp32 atomic adds per output element at the bench size (1.19 ms per call).  Here a thread owns one (position, 8-column vector) and
// walks a slice of the batch in registers; one atomic per slice and element closes it (gridDim.y slices: 64 x fewer atomics, coalesced 16-B row reads).
__global__ __launch_bounds__(256) void embed_scatter_pos_kernel(const bf16_t* __restrict__ dx, const unsigned char* __restrict__ skip_rows, float* __restrict__ dtable,
                                                                long nseq, int seq, int d, int offset) {
    const int dv = d >> 3;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= seq * dv) return;
    const int pos = i / dv, v = i - pos * dv;
    const long per = (nseq + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * per, b1 = b0 + per < nseq ? b0 + per : nseq;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long b = b0; b < b1; ++b) {
        const long r = b * seq + pos;
        if (skip_rows && skip_rows[r]) continue;
        float g[8];
        ld8<bf16_t>(dx + r * d + v * 8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += g[e];
    }
    if (b1 > b0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(dtable + (long)(offset + pos) * d + v * 8 + e, acc[e]);
    }
}
extern "C" int antmmf_embed_scatter_add(const void* dx, const long* idx, const unsigned char* skip_rows, float* dtable, long rows, int seq, int d, int offset, hipStream_t s) {
    if (!dx || !dtable || rows < 0 || seq <= 0 || d <= 0 || (d & 7)) return ANTMMF_EINVAL;
    if (!rows) return ANTMMF_OK;
    if (!idx && rows % seq == 0 && rows / seq >= 64) {
        const long nseq = rows / seq;
        const int slices = (int)(nseq / 16 < 64 ? nseq / 16 : 64);
        hipLaunchKernelGGL(embed_scatter_pos_kernel, dim3((unsigned)((seq * (d >> 3) + 255) / 256), (unsigned)slices), dim3(256), 0, s, (const bf16_t*)dx, skip_rows, dtable,
                           nseq, seq, d, offset);
        return antmmf_check_launch();
    }
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(ew_grid(rows * (d / 8))), dim3(256), 0, s, (const bf16_t*)dx, idx, skip_rows, dtable, rows, seq, d, offset);
    return antmmf_check_launch();
}

// ------------------------------------------------------------------ fused AdamW over a flat arena segment (decoupled weight decay, torch.optim.AdamW semantics)
// p, m, v fp32 master state; g fp32 gradient (scaled by grad_scale first, e.g. 1/world or a clip coefficient); writes the bf16 compute shadow.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ shadow, long n, float lr, float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2, float grad_scale, const float* __restrict__ dev_scale) {
    if (dev_scale) grad_scale *= *dev_scale;  // e.g. the gradient-clipping coefficient computed on the device (no host sync in the step)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * grad_scale;
        float pi = p[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        pi *= (1.f - lr * wd);
        pi -= (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (shadow) shadow[i] = f2bf(pi);
    }
}
// The same update on four elements per thread (16-B accesses; one workgroup per 1024 elements, no grid-stride loop): the 0.43 G-parameter arena
// moves 30 B per element, and the scalar kernel above reached 2.9 TB/s.  Same formula per element (the scalar kernel keeps the unaligned tail).
__global__ __launch_bounds__(256) void adamw_vec4_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                         bf16_t* __restrict__ shadow, long n4, float lr, float beta1, float beta2, float eps, float wd,
                                                         float bc1, float bc2, float grad_scale, const float* __restrict__ dev_scale) {
    if (dev_scale) grad_scale *= *dev_scale;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 g4 = reinterpret_cast<const float4*>(g)[i];
    float4 p4 = reinterpret_cast<float4*>(p)[i], m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
    float gi[4] = {g4.x, g4.y, g4.z, g4.w}, pi[4] = {p4.x, p4.y, p4.z, p4.w}, mi[4] = {m4.x, m4.y, m4.z, m4.w}, vi[4] = {v4.x, v4.y, v4.z, v4.w};
    // (v_sqrt_f32 / v_rcp_f32 -- 1 ulp each -- instead of the IEEE square root and division sequences, the hardware bf16 pack instead of the software
    // rounding: with the library forms this 30-B-per-element stream was VALU-bound at 3.3 TB/s)
    const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2), decay = 1.f - lr * wd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ge = gi[e] * grad_scale;
        mi[e] = beta1 * mi[e] + (1.f - beta1) * ge;
        vi[e] = beta2 * vi[e] + (1.f - beta2) * ge * ge;
#ifdef ANTMMF_EMULATE
        pi[e] = pi[e] * decay - step_size * mi[e] / (sqrtf(vi[e]) * inv_sqrt_bc2 + eps);
#else
        pi[e] = pi[e] * decay - step_size * mi[e] * fast_rcp(__builtin_amdgcn_sqrtf(vi[e]) * inv_sqrt_bc2 + eps);
#endif
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pi[0], pi[1], pi[2], pi[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(mi[0], mi[1], mi[2], mi[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vi[0], vi[1], vi[2], vi[3]);
    if (shadow) reinterpret_cast<uint2*>(shadow)[i] = make_uint2(pack_bf2(pi[0], pi[1]), pack_bf2(pi[2], pi[3]));
}
static int adamw_launch(float* p, const float* g, float* m, float* v, void* shadow, long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int step, float grad_scale, const float* dev_scale, hipStream_t s) {
    if (!p || !g || !m || !v || n < 0 || step < 1) return ANTMMF_EINVAL;
    if (!n) return ANTMMF_OK;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    const bool aligned = !(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) && !((uintptr_t)shadow & 7);
    const long n4 = aligned ? n / 4 : 0;
    if (n4 > 0 && (n4 + 255) / 256 < 0x7fffffffL)
        hipLaunchKernelGGL(adamw_vec4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p, g, m, v, (bf16_t*)shadow, n4, lr, beta1, beta2, eps, weight_decay,
                           bc1, bc2, grad_scale, dev_scale);
    const long done = (n4 > 0 && (n4 + 255) / 256 < 0x7fffffffL) ? n4 * 4 : 0;
    if (done < n)
        hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n - done)), dim3(256), 0, s, p + done, g + done, m + done, v + done,
                           shadow ? (bf16_t*)shadow + done : (bf16_t*)nullptr, n - done, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, dev_scale);
    return antmmf_check_launch();
}
extern "C" int antmmf_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, long n, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int step, float grad_scale, hipStream_t s) {
    return adamw_launch(p, g, m, v, shadow, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, s);
}
// the same step with an additional gradient scale read from device memory (one fp32; nullable)
extern "C" int antmmf_adamw_step_scaled(float* p, const float* g, float* m, float* v, void* shadow, long n, float lr, float beta1, float beta2, float eps,
                                        float weight_decay, int step, float grad_scale, const float* dev_scale, hipStream_t s) {
    return adamw_launch(p, g, m, v, shadow, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, dev_scale, s);
}

// sum of squares of a flat fp32 buffer into *out (fp32 atomics; out must be zeroed) -- gradient-norm clipping (antmmf/utils/general.py:47-56)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, long n) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float v = x[i]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
extern "C" int antmmf_sumsq(const float* x, float* out, long n, hipStream_t s) {
    if (!x || !out || n < 0) return ANTMMF_EINVAL;
    if (!n) return ANTMMF_OK;
    hipLaunchKernelGGL(sumsq_kernel, dim3(ew_grid(n)), dim3(256), 0, s, x, out, n);
    return antmmf_check_launch();
}

// ---- momentum (EMA) update of a key encoder: k = m k + (1 - m) q over a flat fp32 range, refreshing the bf16 compute shadow in
// the same pass.  Replaces the per-parameter Python loop of MocoUtils.momentum_update_key_encoder (moco_utils.py:55-69): with the
// key tower laid out at the query tower's arena offsets it is ONE launch over ~190 M parameters.  HBM-bound, 14 B / parameter.
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ k, const float* __restrict__ q, bf16_t* __restrict__ shadow, long n, float m) {
    const long nvec = n >> 2;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(k)[v], b = reinterpret_cast<const float4*>(q)[v];
        float4 r;
        r.x = a.x * m + b.x * (1.0f - m); r.y = a.y * m + b.y * (1.0f - m); r.z = a.z * m + b.z * (1.0f - m); r.w = a.w * m + b.w * (1.0f - m);
        reinterpret_cast<float4*>(k)[v] = r;
        if (shadow) reinterpret_cast<uint2*>(shadow)[v] = make_uint2(pack_bf2(r.x, r.y), pack_bf2(r.z, r.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (nvec << 2) + threadIdx.x;
        const float r = k[i] * m + q[i] * (1.0f - m);
        k[i] = r;
        if (shadow) shadow[i] = f2bf(r);
    }
}
extern "C" int antmmf_ema_update(float* k, const float* q, void* k_shadow_bf16, long n, float m, hipStream_t s) {
    if (!k || !q || n < 0 || ((uintptr_t)k & 15) || ((uintptr_t)q & 15) || (k_shadow_bf16 && ((uintptr_t)k_shadow_bf16 & 7))) return ANTMMF_EINVAL;
    if (!n) return ANTMMF_OK;
    hipLaunchKernelGGL(ema_kernel, dim3(ew_grid((n >> 2) + 1)), dim3(256), 0, s, k, q, (bf16_t*)k_shadow_bf16, n, m);
    return antmmf_check_launch();
}


// ---- hidden-state dropout of the BERT blocks (modeling_bert.py:175-186,227-238: LayerNorm(dropout(dense(x)) + residual)):
// y = x * keep / (1 - p) (+ residual);  backward dx = dy * keep / (1 - p) with the same counter-based mask.  Only launched when
// p > 0 in training (the flagship M2 encoder has p = 0 everywhere).  HBM-bound elementwise pass.
template <typename T>
__global__ __launch_bounds__(256) void dropout_add_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, long n, float scale,
                                                          uint32_t thr, uint64_t seed) {
    const long nvec = n >> 3;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        float a[8], r[8];
        ld8<T>(x + v * 8, a);
        if (res) ld8<T>(res + v * 8, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (DROPOUT_KEEP(v * 8 + e, seed, thr) ? a[e] * scale : 0.f) + (res ? r[e] : 0.f);
        st8<T>(y + v * 8, a);
    }
}
extern "C" int antmmf_dropout_add(const void* x, const void* residual, void* y, long n, float p, uint64_t seed, int dtype, hipStream_t s) {
    if (!x || !y || n < 0 || (n & 7) || !(p >= 0.f && p < 1.f)) return ANTMMF_EINVAL;
    if (!n) return ANTMMF_OK;
    const float scale = 1.0f / (1.0f - p);
    const uint32_t thr = dropout_threshold(p);
    if (dtype == ANTMMF_BF16) hipLaunchKernelGGL(dropout_add_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, n, scale, thr, seed);
    else if (dtype == ANTMMF_F32) hipLaunchKernelGGL(dropout_add_kernel<float>, dim3(ew_grid(n / 8)), dim3(256), 0, s, (const float*)x, (const float*)residual, (float*)y, n, scale, thr, seed);
    else return ANTMMF_EINVAL;
    return antmmf_check_launch();
}
