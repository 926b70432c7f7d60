// resize.hip -- Pillow-exact antialiased bicubic resize of 8-bit interleaved images on gfx950 (the input-pipeline
// step in front of the M2 image tower).
//
// Replaces (reference): `square_transform(size)` = torchvision Resize((size, size), BICUBIC) + ToTensor
// (prj/M2_Encoder/vlmo/transforms/square_transform.py:8-14; called per image on the CPU by prj/M2_Encoder/m2_encoder.py:61-68), whose
// arithmetic is Pillow's src/libImaging/Resample.c (ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc).  Byte-exact:
// integer accumulation of 22-bit fixed-point coefficients with a rounding half, arithmetic shift, clamp; the horizontal pass is
// rounded to uint8 before the vertical one, exactly as Pillow does.  The coefficient tables (double-precision filter weights,
// normalised per output pixel) are a few KB per image shape and are computed by the HOST (antmmf/hip/ops.py::bicubic_coeffs, the same
// statement order as Pillow's precompute_coeffs); the device does the byte work.
//
// Data layout: a batch of RAGGED images packed back to back in one uint8 buffer [h_i, w_i, C]; per image a row of 10 int64 in `desc`:
//   0 src byte offset   1 h   2 w   3 tmp byte offset (intermediate [h, out_w, C])
//   4 horizontal coeff offset (int32 index into `coeffs`, table [out_w, kx])   5 kx (taps per output; 0 = pass skipped, w == out_w)
//   6 horizontal bounds offset (int32 index into `bounds`, table [out_w, 2] = first tap, tap count)
//   7 vertical coeff offset   8 ky (0 = skipped, h == out_h)   9 vertical bounds offset
// Both kernels are HBM-bound byte movers: pass 1 reads every input byte once (row staged in LDS with aligned 4-B loads, then one
// thread per output pixel walks its taps in LDS), pass 2 reads the [h, out_w, C] intermediate once with fully coalesced rows.
// Algorithmic bytes per image: h*w*C + 2*h*out_w*C + out_h*out_w*C*(1 or 4).
#include "common.h"

#define RESIZE_PRECISION_BITS 22
#define RESIZE_DESC 10

__device__ __forceinline__ int clip8_shift(int acc) {
    const int v = acc >> RESIZE_PRECISION_BITS;  // arithmetic shift (Pillow indexes a clamp table with it)
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One workgroup per (input row, image): stage the row in LDS, then thread xx produces output pixel xx of that row.
__global__ __launch_bounds__(256) void resize_h_u8_kernel(const uint8_t* __restrict__ src, long src_bytes, const long* __restrict__ desc, int C, int out_w,
                                                          const int* __restrict__ coeffs, const int* __restrict__ bounds, uint8_t* __restrict__ tmp) {
    ANTMMF_DYN_LDS(uint32_t, row32);
    const long* d = desc + (long)blockIdx.y * RESIZE_DESC;
    const int h = (int)d[1], w = (int)d[2];
    const int y = blockIdx.x;
    if (y >= h) return;
    uint8_t* trow = tmp + d[3] + (long)y * out_w * C;
    const long row_start = d[0] + (long)y * w * C;
    const int kx = (int)d[5];
    if (kx == 0) {  // w == out_w: Pillow skips the horizontal pass
        for (int i = threadIdx.x; i < out_w * C; i += 256) trow[i] = src[row_start + i];
        return;
    }
    const long base = row_start & ~3L;
    const int lead = (int)(row_start - base), n_dw = (lead + w * C + 3) >> 2;
    for (int i = threadIdx.x; i < n_dw; i += 256) {
        const long a = base + 4L * i;
        uint32_t v;
        if (a + 4 <= src_bytes) v = *reinterpret_cast<const uint32_t*>(src + a);
        else {  // last dword of the buffer: never read past its end
            v = 0;
            for (int b = 0; b < 4; ++b)
                if (a + b < src_bytes) v |= (uint32_t)src[a + b] << (8 * b);
        }
        row32[i] = v;
    }
    __syncthreads();
    const uint8_t* row = reinterpret_cast<const uint8_t*>(row32) + lead;
    const int* kt = coeffs + d[4];
    const int* bt = bounds + d[6];
    for (int xx = threadIdx.x; xx < out_w; xx += 256) {
        const int xmin = bt[2 * xx], n = bt[2 * xx + 1];
        const int* k = kt + (long)xx * kx;
        const uint8_t* p = row + xmin * C;
        if (C == 3) {
            int a0 = 1 << (RESIZE_PRECISION_BITS - 1), a1 = a0, a2 = a0;
            for (int x = 0; x < n; ++x) {
                const int kv = k[x];
                a0 += p[3 * x] * kv; a1 += p[3 * x + 1] * kv; a2 += p[3 * x + 2] * kv;
            }
            trow[3 * xx] = (uint8_t)clip8_shift(a0); trow[3 * xx + 1] = (uint8_t)clip8_shift(a1); trow[3 * xx + 2] = (uint8_t)clip8_shift(a2);
        } else {
            for (int c = 0; c < C; ++c) {
                int a = 1 << (RESIZE_PRECISION_BITS - 1);
                for (int x = 0; x < n; ++x) a += p[x * C + c] * k[x];
                trow[xx * C + c] = (uint8_t)clip8_shift(a);
            }
        }
    }
}

// One thread per output element (image, yy, xx, c); neighbouring threads walk neighbouring bytes of the same intermediate rows.
// OUT_F32: ToTensor layout and scaling, float32 [n, C, out_h, out_w] = u8 / 255 (IEEE division, equal to torch's .div(255)).
template <bool OUT_F32>
__global__ __launch_bounds__(256) void resize_v_u8_kernel(const uint8_t* __restrict__ tmp, const long* __restrict__ desc, int n_images, int C, int out_h, int out_w,
                                                          const int* __restrict__ coeffs, const int* __restrict__ bounds, void* __restrict__ out) {
    const int rowb = out_w * C;
    const long per_image = (long)out_h * rowb, total = per_image * n_images;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int img = (int)(i / per_image);
        const int r = (int)(i - (long)img * per_image);
        const int yy = r / rowb, e = r - yy * rowb;
        const long* d = desc + (long)img * RESIZE_DESC;
        const uint8_t* t = tmp + d[3];
        const int ky = (int)d[8];
        int v;
        if (ky == 0) v = t[(long)yy * rowb + e];  // h == out_h: vertical pass skipped
        else {
            const int* bt = bounds + d[9];
            const int ymin = bt[2 * yy], n = bt[2 * yy + 1];
            const int* k = coeffs + d[7] + (long)yy * ky;
            int a = 1 << (RESIZE_PRECISION_BITS - 1);
            const uint8_t* p = t + (long)ymin * rowb + e;
            for (int y = 0; y < n; ++y) a += p[(long)y * rowb] * k[y];
            v = clip8_shift(a);
        }
        if (OUT_F32) {
            const int xx = e / C, c = e - xx * C;
            reinterpret_cast<float*>(out)[(((long)img * C + c) * out_h + yy) * out_w + xx] = (float)v / 255.0f;
        } else reinterpret_cast<uint8_t*>(out)[i] = (uint8_t)v;
    }
}

extern "C" int antmmf_resize_bicubic_u8(const void* src, int64_t src_bytes, const int64_t* desc, int n_images, int max_h, int max_w, int channels,
                                        int out_h, int out_w, const int32_t* coeffs, const int32_t* bounds, void* tmp, void* out, int out_f32,
                                        hipStream_t s) {
    if (!src || !desc || !coeffs || !bounds || !tmp || !out || n_images <= 0 || max_h <= 0 || max_w <= 0 || channels <= 0 || channels > 4 ||
        out_h <= 0 || out_w <= 0 || src_bytes <= 0)
        return ANTMMF_EINVAL;
    const size_t lds = ((size_t)max_w * channels + 3 + 3) / 4 * 4 + 4;  // one input row + alignment lead
    if (lds > 160 * 1024) return ANTMMF_EINVAL;
    static_assert(sizeof(long) == sizeof(int64_t), "descriptor rows are int64");
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resize_h_u8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(resize_h_u8_kernel, dim3((unsigned)max_h, (unsigned)n_images), dim3(256), lds, s, (const uint8_t*)src, (long)src_bytes, (const long*)desc,
                       channels, out_w, (const int*)coeffs, (const int*)bounds, (uint8_t*)tmp);
    const long total = (long)n_images * out_h * out_w * channels;
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (out_f32)
        hipLaunchKernelGGL(resize_v_u8_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)tmp, (const long*)desc, n_images, channels, out_h, out_w,
                           (const int*)coeffs, (const int*)bounds, out);
    else
        hipLaunchKernelGGL(resize_v_u8_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)tmp, (const long*)desc, n_images, channels, out_h, out_w,
                           (const int*)coeffs, (const int*)bounds, out);
    return antmmf_check_launch();
}
