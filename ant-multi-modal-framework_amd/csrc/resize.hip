// resize.hip -- Pillow-exact antialiased bicubic resize of 8-bit interleaved images on gfx950 (the input-pipeline
// step in front of the M2 image tower).
//
// Replaces (reference): `square_transform(size)` = torchvision Resize((size, size), BICUBIC) + ToTensor
// (prj/M2_Encoder/vlmo/transforms/square_transform.py:8-14; called per image on the CPU by prj/M2_Encoder/m2_encoder.py:61-68), whose
// arithmetic is Pillow's src/libImaging/Resample.c (ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc).  Byte-exact:
// integer accumulation of 22-bit fixed-point coefficients with a rounding half, arithmetic shift, clamp; the horizontal pass is
// rounded to uint8 before the vertical one, exactly as Pillow does.  The coefficient tables (double-precision filter weights,
// normalised per output pixel) are a few KB per image shape and are computed by the HOST (antmmf/hip/image.py::bicubic_coeffs, the same
// statement order as Pillow's precompute_coeffs); the device does the byte work.
//
// Data layout: a batch of RAGGED images packed back to back in one uint8 buffer [h_i, w_i, C]; per image a row of 10 int64 in `desc`:
//   0 src byte offset   1 h   2 w   3 tmp byte offset (intermediate [h, out_w, C])
//   4 horizontal coeff offset (int32 index into `coeffs`, TAP-MAJOR table [kx rounded up to 4, out_w], zero-padded: lane xx reads
//     consecutive words)   5 kx (taps per output; 0 = pass skipped, w == out_w)
//   6 horizontal bounds offset (int32 index into `bounds`, table [out_w, 2] = first tap, tap count)
//   7 vertical coeff offset (output-major table [out_h, ky]: a wave shares one row of it)   8 ky (0 = skipped, h == out_h)
//   9 vertical bounds offset
// Both kernels are HBM-bound byte movers: pass 1 reads every input byte once (8 rows staged in LDS with aligned 4-B loads, then one
// thread per output column walks its taps over all staged rows), pass 2 reads the [h, out_w, C] intermediate once with fully coalesced rows.
// Algorithmic bytes per image: h*w*C + 2*h*out_w*C + out_h*out_w*C*(1 or 4).
#include "common.h"

#define RESIZE_PRECISION_BITS 22
#define RESIZE_DESC 10

// bytes [shift, shift + 4) of the 8-byte little-endian pair (lo, hi): v_alignbyte_b32
__device__ __forceinline__ uint32_t lds_alignbyte(uint32_t hi, uint32_t lo, uint32_t shift) {
#ifdef ANTMMF_EMULATE
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * shift));
#else
    return __builtin_amdgcn_alignbyte(hi, lo, shift);
#endif
}
// byte x 22-bit fixed-point coefficient (|k| < 2^23, checked by the host): v_mul_i32_i24 / v_mad_i32_i24 run at full rate, a 32-bit
// v_mul_lo_u32 at a quarter of it -- and these multiplies are most of the kernels' instructions
__device__ __forceinline__ int mul24(int a, int b) {
#ifdef ANTMMF_EMULATE
    return a * b;
#else
    return __mul24(a, b);
#endif
}
__device__ __forceinline__ int clip8_shift(int acc) {
    const int v = acc >> RESIZE_PRECISION_BITS;  // arithmetic shift (Pillow indexes a clamp table with it)
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One workgroup per (block of up to RESIZE_ROWS input rows, image): the rows are staged in LDS (every input byte is read from HBM
// once), then each lane produces one output column of all staged rows from its tap windows in LDS.
#define RESIZE_ROWS 8
template <int C>
__global__ __launch_bounds__(256) void resize_h_u8_kernel(const uint8_t* __restrict__ src, long src_bytes, const long* __restrict__ desc, int out_w,
                                                          const int* __restrict__ coeffs, const int* __restrict__ bounds, uint8_t* __restrict__ tmp,
                                                          int rows_per_wg) {
    ANTMMF_DYN_LDS(uint32_t, rows32);
    const long* d = desc + (long)blockIdx.y * RESIZE_DESC;
    const int h = (int)d[1], w = (int)d[2];
    const int y0 = blockIdx.x * rows_per_wg;
    if (y0 >= h) return;
    const int nrows = h - y0 < rows_per_wg ? h - y0 : rows_per_wg;
    const int kx = (int)d[5];
    if (kx == 0) {  // w == out_w: Pillow skips the horizontal pass
        const long n = (long)nrows * out_w * C, from = d[0] + (long)y0 * w * C, to = d[3] + (long)y0 * out_w * C;
        for (long i = threadIdx.x; i < n; i += 256) tmp[to + i] = src[from + i];
        return;
    }
    // The block's rows are adjacent in memory: ONE contiguous byte range, staged with aligned 16-B loads that are all independent
    // (a per-row loop exposed one global round trip per row).  LDS byte offset of row r = lead0 + r * w * C.
    const long block_start = d[0] + (long)y0 * w * C, base = block_start & ~15L;
    const int lead0 = (int)(block_start - base), n_qw = (int)((lead0 + (long)nrows * w * C + 15) >> 4);
    uint4* rows128 = reinterpret_cast<uint4*>(rows32);
#pragma unroll 4
    for (int i = threadIdx.x; i < n_qw; i += 256) {
        const long a = base + 16L * i;
        uint4 v;
        if (a + 16 <= src_bytes) v = *reinterpret_cast<const uint4*>(src + a);
        else {  // last 16 bytes of the buffer: never read past its end
            uint32_t t[4] = {0, 0, 0, 0};
            for (int b = 0; b < 16; ++b)
                if (a + b < src_bytes) t[b >> 2] |= (uint32_t)src[a + b] << (8 * (b & 3));
            v = make_uint4(t[0], t[1], t[2], t[3]);
        }
        rows128[i] = v;
    }
    __syncthreads();
    const int* kt = coeffs + d[4];
    const int* bt = bounds + d[6];
    // Thread xx produces output column xx of EVERY staged row: a coefficient is fetched once per tap and used for rows x channels
    // multiply-adds, and the next four taps' coefficients are requested before the current four are consumed (the loop is otherwise
    // a chain of dependent L1 round trips: 81 % of the wave cycles were waits, profiles/r1_resize_pmc_*).
    for (int xx = threadIdx.x; xx < out_w; xx += 256) {
        const int xmin = bt[2 * xx], n = bt[2 * xx + 1];
        const int* k = kt + xx;  // tap x of output xx at k[x * out_w]
        int boff[RESIZE_ROWS];   // byte offset of the window's first tap in the LDS array (index arithmetic keeps the ds_read path)
        int acc[RESIZE_ROWS][C];
#pragma unroll
        for (int r = 0; r < RESIZE_ROWS; ++r) {
            const int rr = r < nrows ? r : 0;  // rows past the block's end recompute row 0 and are not stored
            boff[r] = lead0 + rr * w * C + xmin * C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[r][c] = 1 << (RESIZE_PRECISION_BITS - 1);
        }
        if (C == 3) {
            // RGB: four taps = 12 bytes = three dwords per row.  The window is read as ALIGNED dwords (a quarter of the LDS
            // instructions of byte reads) and realigned with v_alignbyte; taps past the window's end multiply zero coefficients
            // (the host pads the tap-major table to a multiple of four taps with zeros).
            uint32_t carry[RESIZE_ROWS];
#pragma unroll
            for (int r = 0; r < RESIZE_ROWS; ++r) carry[r] = rows32[boff[r] >> 2];
            int k0 = k[0], k1 = k[out_w], k2 = k[2 * (long)out_w], k3 = k[3 * (long)out_w];
            for (int x = 0; x < n; x += 4) {
                const int c0 = k0, c1 = k1, c2 = k2, c3 = k3;
                if (x + 4 < n) { k0 = k[(long)(x + 4) * out_w]; k1 = k[(long)(x + 5) * out_w]; k2 = k[(long)(x + 6) * out_w]; k3 = k[(long)(x + 7) * out_w]; }
                const int base = (x >> 2) * 3;
#pragma unroll
                for (int r = 0; r < RESIZE_ROWS; ++r) {
                    const uint32_t* q = rows32 + (boff[r] >> 2) + base;
                    const uint32_t sh = (uint32_t)(boff[r] & 3);
                    const uint32_t w1 = q[1], w2 = q[2], w3 = q[3];
                    const uint32_t d0 = lds_alignbyte(w1, carry[r], sh), d1 = lds_alignbyte(w2, w1, sh), d2 = lds_alignbyte(w3, w2, sh);
                    carry[r] = w3;
                    acc[r][0] += mul24((int)(d0 & 255), c0) + mul24((int)(d0 >> 24), c1) + mul24((int)((d1 >> 16) & 255), c2) + mul24((int)((d2 >> 8) & 255), c3);
                    acc[r][1 % C] += mul24((int)((d0 >> 8) & 255), c0) + mul24((int)(d1 & 255), c1) + mul24((int)(d1 >> 24), c2) + mul24((int)((d2 >> 16) & 255), c3);
                    acc[r][2 % C] += mul24((int)((d0 >> 16) & 255), c0) + mul24((int)((d1 >> 8) & 255), c1) + mul24((int)(d2 & 255), c2) + mul24((int)(d2 >> 24), c3);
                }
            }
        } else {
            const uint8_t* bytes = reinterpret_cast<const uint8_t*>(rows32);
            for (int x = 0; x < n; ++x) {
                const int kv = k[(long)x * out_w];
#pragma unroll
                for (int r = 0; r < RESIZE_ROWS; ++r)
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[r][c] += mul24(bytes[boff[r] + x * C + c], kv);
            }
        }
#pragma unroll
        for (int r = 0; r < RESIZE_ROWS; ++r)
            if (r < nrows) {
                uint8_t* trow = tmp + d[3] + (long)(y0 + r) * out_w * C + xx * C;
#pragma unroll
                for (int c = 0; c < C; ++c) trow[c] = (uint8_t)clip8_shift(acc[r][c]);
            }
    }
}

// Vertical pass: one thread per 4 consecutive bytes of an output row (image, yy, 4 x (xx, c)) -- VEC = 4, one aligned dword per
// tap, when out_w * C is a multiple of 4 -- or per byte (VEC = 1).  Neighbouring threads walk neighbouring words of the same rows.
// OUT_F32: ToTensor layout and scaling, float32 [n, C, out_h, out_w] = u8 / 255 (IEEE division, equal to torch's .div(255)).
template <bool OUT_F32, int VEC>
__global__ __launch_bounds__(256) void resize_v_u8_kernel(const uint8_t* __restrict__ tmp, const long* __restrict__ desc, int n_images, int C, int out_h, int out_w,
                                                          const int* __restrict__ coeffs, const int* __restrict__ bounds, void* __restrict__ out) {
    const int rowb = out_w * C, roww = rowb / VEC;
    const long per_image = (long)out_h * roww, total = per_image * n_images;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int img = (int)(i / per_image);
        const int rem = (int)(i - (long)img * per_image);
        const int yy = rem / roww, e = (rem - yy * roww) * VEC;
        const long* d = desc + (long)img * RESIZE_DESC;
        const uint8_t* t = tmp + d[3];
        const int ky = (int)d[8];
        int v[VEC];
        if (ky == 0) {  // h == out_h: vertical pass skipped
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = t[(long)yy * rowb + e + j];
        } else {
            const int* bt = bounds + d[9];
            const int ymin = bt[2 * yy], n = bt[2 * yy + 1];
            const int* k = coeffs + d[7] + (long)yy * ky;
            int a[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[j] = 1 << (RESIZE_PRECISION_BITS - 1);
            const uint8_t* p = t + (long)ymin * rowb + e;
            for (int y = 0; y < n; ++y) {
                const int kv = k[y];
                if (VEC == 4) {
                    const uint32_t w = *reinterpret_cast<const uint32_t*>(p + (long)y * rowb);
                    a[0] += mul24((int)(w & 255), kv); a[1 % VEC] += mul24((int)((w >> 8) & 255), kv); a[2 % VEC] += mul24((int)((w >> 16) & 255), kv); a[3 % VEC] += mul24((int)(w >> 24), kv);
                } else a[0] += mul24(p[(long)y * rowb], kv);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = clip8_shift(a[j]);
        }
        if (OUT_F32) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int xx = (e + j) / C, c = (e + j) - xx * C;
                reinterpret_cast<float*>(out)[(((long)img * C + c) * out_h + yy) * out_w + xx] = (float)v[j] / 255.0f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) reinterpret_cast<uint8_t*>(out)[((long)img * out_h + yy) * rowb + e + j] = (uint8_t)v[j];
        }
    }
}

extern "C" int antmmf_resize_bicubic_u8(const void* src, int64_t src_bytes, const int64_t* desc, int n_images, int max_h, int max_w, int channels,
                                        int out_h, int out_w, const int32_t* coeffs, const int32_t* bounds, void* tmp, void* out, int out_f32,
                                        hipStream_t s) {
    if (!src || !desc || !coeffs || !bounds || !tmp || !out || n_images <= 0 || max_h <= 0 || max_w <= 0 || channels <= 0 || channels > 4 ||
        out_h <= 0 || out_w <= 0 || src_bytes <= 0)
        return ANTMMF_EINVAL;
    if (channels == 2) return ANTMMF_EINVAL;  // L, RGB, RGBA
    if (((uintptr_t)src & 15) != 0) return ANTMMF_EINVAL;
    const size_t row_bytes = (size_t)max_w * channels, slack = 16 + 16 + 16;  // alignment lead, tail of the last 16-B load, 4-tap look-ahead
    if (row_bytes + slack > 160 * 1024) return ANTMMF_EINVAL;
    int rows_per_wg = (int)((65536 - slack) / row_bytes);  // keep >= 2 workgroups per CU resident
    rows_per_wg = rows_per_wg < 1 ? 1 : (rows_per_wg > RESIZE_ROWS ? RESIZE_ROWS : rows_per_wg);
    const size_t lds = (row_bytes * rows_per_wg + slack + 15) / 16 * 16;
    static_assert(sizeof(long) == sizeof(int64_t), "descriptor rows are int64");
    const dim3 grid((unsigned)((max_h + rows_per_wg - 1) / rows_per_wg), (unsigned)n_images);
#define RESIZE_H(CH)                                                                                                                           \
    do {                                                                                                                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resize_h_u8_kernel<CH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(resize_h_u8_kernel<CH>, grid, dim3(256), lds, s, (const uint8_t*)src, (long)src_bytes, (const long*)desc, out_w,   \
                           (const int*)coeffs, (const int*)bounds, (uint8_t*)tmp, rows_per_wg);                                 \
    } while (0)
    if (channels == 3) RESIZE_H(3);
    else if (channels == 1) RESIZE_H(1);
    else RESIZE_H(4);
#undef RESIZE_H
    // tmp offsets are sums of h * out_w * C: the intermediate rows are dword-aligned whenever a row is a whole number of dwords
    const int vec = (out_w * channels) % 4 == 0 ? 4 : 1;
    const long total = (long)n_images * out_h * (out_w * channels / vec);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
#define RESIZE_V(F32, VEC)                                                                                                                  \
    hipLaunchKernelGGL((resize_v_u8_kernel<F32, VEC>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)tmp, (const long*)desc, n_images, \
                       channels, out_h, out_w, (const int*)coeffs, (const int*)bounds, out)
    if (out_f32) { if (vec == 4) RESIZE_V(true, 4); else RESIZE_V(true, 1); }
    else { if (vec == 4) RESIZE_V(false, 4); else RESIZE_V(false, 1); }
#undef RESIZE_V
    return antmmf_check_launch();
}
