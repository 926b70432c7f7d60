// frames.hip -- the video-frame transform in front of the visual tower on gfx950: decoded uint8 frames ->
// float32 bilinear resize -> GroupNormalize, written straight into the zero-padded batch canvas.
//
// Replaces (reference, per video, on the CPU dataloader workers):
//   CustomTransforms.__call__          antmmf/datasets/processors/image_processors.py:520-547   uint8 -> float32, then the list below
//   ImageLongsideScaleAndPad           antmmf/utils/image_ops.py:127-189,191-223               torchvision F.resize(tensor, (h', w'), BILINEAR)
//   GroupNormalize                     antmmf/utils/image_ops.py:72-108                        if max > 1: / 255;  - mean;  / std
//   NestedTensor.from_tensor_list      antmmf/structures/nested_tensor.py:51-63                zero-pad to the batch extent (collate)
// torchvision's tensor resize is torch.nn.functional.interpolate(x, size, mode="bilinear", align_corners=False) (no antialiasing for
// tensors in the torchvision generations the reference pins), i.e. per output pixel
//     src = max((dst + 0.5) * in / out - 0.5, 0),  i0 = floor(src), i1 = min(i0 + 1, in - 1),  l1 = src - i0, l0 = 1 - l1
//     v   = (l0w p[i0h][i0w] + l1w p[i0h][i1w]) l0h + (l0w p[i1h][i0w] + l1w p[i1h][i1w]) l1h          (float32)
// Arithmetic is float32 and follows that expression order without fused multiply-adds (ATen's CPU kernels may contract them: parity is
// stated to 1e-6 relative, not bitwise).
//
// HBM-bound on the float32 output (4 C out_h out_w bytes per frame against C h w input bytes that stay L2-resident across the rows
// that reuse them): one thread produces 4 consecutive output pixels of one channel row and stores them as one 16-B vector, a wave
// writes 1 KiB of one output row.  Algorithmic bytes per frame: C h w + 4 C out_h out_w.
// The "max > 1" test of GroupNormalize is a device-side reduction (pass 1, no stores) that pass 2 reads: no host round trip.
#include "common.h"

struct FrameArgs {
    const uint8_t* src; float* out; int* maxbits;
    const float* mean; const float* stdv;   // [C] (NULL: no normalisation, plain float resize)
    long sn, sc, sh, sw;                    // source strides in bytes (= elements) of [frame, channel, row, column]
    long on, oc, oh;                        // output strides in floats of [frame, channel, row]; columns are contiguous
    int n, C, h, w, out_h, out_w;
    float scale_h, scale_w;
    int force_div255;                       // -1: decide from the device-side maximum (reference behaviour); 0 / 1: fixed
};

__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float s = __fsub_rn(__fmul_rn(scale, (float)dst + 0.5f), 0.5f);   // area_pixel_compute_source_index, align_corners = False
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 > in_size - 1 ? in_size - 1 : i0;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = __fsub_rn(s, (float)i0);
    l0 = __fsub_rn(1.f, l1);
}

__device__ __forceinline__ float bilerp(const uint8_t* __restrict__ r0, const uint8_t* __restrict__ r1, long o0, long o1, float l0w, float l1w, float l0h, float l1h) {
    const float t0 = __fadd_rn(__fmul_rn(l0w, (float)r0[o0]), __fmul_rn(l1w, (float)r0[o1]));
    const float t1 = __fadd_rn(__fmul_rn(l0w, (float)r1[o0]), __fmul_rn(l1w, (float)r1[o1]));
    return __fadd_rn(__fmul_rn(t0, l0h), __fmul_rn(t1, l1h));
}

// PASS: 0 = maximum of the resized values only (no stores), 1 = write
template <int PASS>
__global__ __launch_bounds__(256) void frames_bilinear_kernel(const FrameArgs a) {
    const int xq = (a.out_w + 3) >> 2;
    const long per_frame = (long)a.C * a.out_h * xq, total = per_frame * a.n;
    float vmax = 0.f;
    bool div255 = a.force_div255 == 1;
    if (PASS == 1 && a.force_div255 < 0) div255 = __int_as_float(*a.maxbits) > 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int f = (int)(i / per_frame);
        int rem = (int)(i - (long)f * per_frame);
        const int c = rem / (a.out_h * xq); rem -= c * a.out_h * xq;
        const int y = rem / xq, x0 = (rem - y * xq) * 4;
        int y0, y1; float l0h, l1h;
        src_index(y, a.scale_h, a.h, y0, y1, l0h, l1h);
        const uint8_t* base = a.src + (long)f * a.sn + (long)c * a.sc;
        const uint8_t* r0 = base + (long)y0 * a.sh;
        const uint8_t* r1 = base + (long)y1 * a.sh;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j < a.out_w ? x0 + j : a.out_w - 1;
            int xa, xb; float l0w, l1w;
            src_index(x, a.scale_w, a.w, xa, xb, l0w, l1w);
            v[j] = bilerp(r0, r1, (long)xa * a.sw, (long)xb * a.sw, l0w, l1w, l0h, l1h);
        }
        if (PASS == 0) {
            vmax = fmaxf(fmaxf(vmax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
        } else {
            if (a.mean) {
                const float m = a.mean[c], s = a.stdv[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = v[j];
                    if (div255) t = t / 255.0f;              // IEEE division, like tensor.div_(255.0)
                    v[j] = __fsub_rn(t, m) / s;
                }
            }
            float* o = a.out + (long)f * a.on + (long)c * a.oc + (long)y * a.oh + x0;
            if (x0 + 4 <= a.out_w && (((uintptr_t)o) & 15) == 0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (x0 + j < a.out_w) o[j] = v[j];
            }
        }
    }
    if (PASS == 0) {
        // pixel values are >= 0: their float bit patterns order like ints
#ifdef ANTMMF_EMULATE
        vmax = emu_wave_max(vmax);
        if ((threadIdx.x & 63) == 0) atomicMax(a.maxbits, (int)__float_as_uint(vmax));
#else
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0) atomicMax(a.maxbits, __float_as_int(vmax));
#endif
    }
}

// ---- antialiased variant: torchvision >= 0.17 resizes TENSORS with antialias=True by default (the reference's requirements admit it: `torchvision>=0.8.1`), i.e.
// torch.nn.functional.interpolate(x, size, mode="bilinear", align_corners=False, antialias=True) -- ATen's separable triangle filter
// (aten/src/ATen/native/cpu/UpSampleKernel.cpp: HelperInterpLinear::aa_filter, _compute_indices_min_size_weights_aa, horizontal pass into a float
// temporary, then the vertical pass):
//     scale = in / out,  support = scale >= 1 ? scale : 1,  centre = scale (i + 0.5),  invscale = scale >= 1 ? 1 / scale : 1
//     xmin = max(int(centre - support + 0.5), 0),  xsize = min(int(centre + support + 0.5), in) - xmin
//     w_j = max(0, 1 - |(j + xmin - centre + 0.5) invscale|),  normalised by their sum;   out_i = sum_j w_j in[xmin + j]   (sequential float32 sum)
// Selected by `antialias: true` on ImageLongsideScaleAndPad (this build's key; default false = the torchvision generations contemporary with the reference).
struct AaTaps { int xmin, xsize; float centre, invscale, total; };
__device__ __forceinline__ float aa_weight(const AaTaps& t, int j) {
    const float x = fabsf((float)(((double)((float)(j + t.xmin) - t.centre) + 0.5) * (double)t.invscale));
    return x < 1.0f ? 1.0f - x : 0.0f;
}
__device__ __forceinline__ AaTaps aa_taps(int i, int in_size, float scale) {
    AaTaps t;
    const float support = scale >= 1.0f ? scale : 1.0f;
    t.centre = (float)((double)scale * ((double)i + 0.5));
    t.invscale = scale >= 1.0f ? (float)(1.0 / (double)scale) : 1.0f;
    long lo = (long)((double)(t.centre - support) + 0.5), hi = (long)((double)(t.centre + support) + 0.5);
    lo = lo < 0 ? 0 : lo;
    hi = hi > in_size ? in_size : hi;
    const int maxsz = (int)ceilf(support) * 2 + 1;
    t.xmin = (int)lo;
    t.xsize = (int)(hi - lo) < 0 ? 0 : ((int)(hi - lo) > maxsz ? maxsz : (int)(hi - lo));
    float total = 0.f;
    for (int j = 0; j < t.xsize; ++j) total += aa_weight(t, j);
    t.total = total;   // (the divisor itself: every weight is DIVIDED by it, as ATen does)
    return t;
}
// horizontal pass: uint8 [n, C, h, w] (strides) -> float temp [n, C, h, out_w] (dense)
__global__ __launch_bounds__(256) void frames_aa_h_kernel(const FrameArgs a, float* __restrict__ temp) {
    const long total = (long)a.n * a.C * a.h * a.out_w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % a.out_w);
        long r = i / a.out_w;
        const int y = (int)(r % a.h); r /= a.h;
        const int c = (int)(r % a.C), f = (int)(r / a.C);
        const AaTaps t = aa_taps(x, a.w, a.scale_w);
        const uint8_t* row = a.src + (long)f * a.sn + (long)c * a.sc + (long)y * a.sh;
        float acc = 0.f;
        for (int j = 0; j < t.xsize; ++j) {
            const float w = t.total != 0.f ? aa_weight(t, j) / t.total : aa_weight(t, j);
            const float v = __fmul_rn((float)row[(long)(t.xmin + j) * a.sw], w);
            acc = j == 0 ? v : __fadd_rn(acc, v);
        }
        temp[i] = acc;
    }
}
// vertical pass (+ GroupNormalize): temp [n, C, h, out_w] -> out; PASS as in frames_bilinear_kernel
template <int PASS>
__global__ __launch_bounds__(256) void frames_aa_v_kernel(const FrameArgs a, const float* __restrict__ temp) {
    const long total = (long)a.n * a.C * a.out_h * a.out_w;
    float vmax = 0.f;
    bool div255 = a.force_div255 == 1;
    if (PASS == 1 && a.force_div255 < 0) div255 = __int_as_float(*a.maxbits) > 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % a.out_w);
        long r = i / a.out_w;
        const int y = (int)(r % a.out_h); r /= a.out_h;
        const int c = (int)(r % a.C), f = (int)(r / a.C);
        const AaTaps t = aa_taps(y, a.h, a.scale_h);
        const float* col = temp + (((long)f * a.C + c) * a.h) * a.out_w + x;
        float acc = 0.f;
        for (int j = 0; j < t.xsize; ++j) {
            const float w = t.total != 0.f ? aa_weight(t, j) / t.total : aa_weight(t, j);
            const float v = __fmul_rn(col[(long)(t.xmin + j) * a.out_w], w);
            acc = j == 0 ? v : __fadd_rn(acc, v);
        }
        if (PASS == 0) vmax = fmaxf(vmax, acc);
        else {
            if (a.mean) {
                if (div255) acc = acc / 255.0f;
                acc = __fsub_rn(acc, a.mean[c]) / a.stdv[c];
            }
            a.out[(long)f * a.on + (long)c * a.oc + (long)y * a.oh + x] = acc;
        }
    }
    if (PASS == 0) {
#ifdef ANTMMF_EMULATE
        vmax = emu_wave_max(vmax);
        if ((threadIdx.x & 63) == 0) atomicMax(a.maxbits, (int)__float_as_uint(vmax < 0.f ? 0.f : vmax));
#else
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0) atomicMax(a.maxbits, __float_as_int(vmax < 0.f ? 0.f : vmax));
#endif
    }
}

// src: uint8 frames, element strides (sn, sc, sh, sw) -- [n, C, h, w] (what reaches the reference's transform) or [n, h, w, C] (decoder
// output) are both just strides.  out: float32, frame f / channel c / row y at out + f on + c oc + y oh, out_w contiguous columns (a view
// into the zero-initialised padded canvas of the batch, or a dense [n, C, out_h, out_w]).  mean / std: DEVICE float[C] or NULL (resize only).
// div255: 1 / 0 fixed, -1 = the reference's test "resized.max() > 1" evaluated on the device into max_scratch (DEVICE int, any value).
extern "C" int antmmf_frames_bilinear_norm(const void* src, int n, int channels, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                           float* out, int out_h, int out_w, int64_t on, int64_t oc, int64_t oh, const float* mean, const float* stdv,
                                           int div255, int* max_scratch, hipStream_t s) {
    if (!src || !out || n <= 0 || channels <= 0 || h <= 0 || w <= 0 || out_h <= 0 || out_w <= 0 || (mean == nullptr) != (stdv == nullptr)) return ANTMMF_EINVAL;
    if (div255 < -1 || div255 > 1 || (div255 < 0 && mean && !max_scratch)) return ANTMMF_EINVAL;
    FrameArgs a;
    a.src = (const uint8_t*)src; a.out = out; a.maxbits = max_scratch; a.mean = mean; a.stdv = stdv;
    a.sn = sn; a.sc = sc; a.sh = sh; a.sw = sw; a.on = on; a.oc = oc; a.oh = oh;
    a.n = n; a.C = channels; a.h = h; a.w = w; a.out_h = out_h; a.out_w = out_w;
    a.scale_h = (float)h / (float)out_h; a.scale_w = (float)w / (float)out_w;   // interpolate(size=...): scale = in / out
    a.force_div255 = mean ? div255 : 0;
    const long total = (long)n * channels * out_h * ((out_w + 3) / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (mean && div255 < 0) {
        if (hipMemsetAsync(max_scratch, 0, sizeof(int), s) != hipSuccess) return ANTMMF_ELAUNCH;
        hipLaunchKernelGGL(frames_bilinear_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(frames_bilinear_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return antmmf_check_launch();
}

// The antialiased form (see above).  temp: DEVICE float[n * channels * h * out_w] scratch of the caller (the horizontally filtered frames).
extern "C" int antmmf_frames_bilinear_aa_norm(const void* src, int n, int channels, int h, int w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* temp,
                                              float* out, int out_h, int out_w, int64_t on, int64_t oc, int64_t oh, const float* mean, const float* stdv,
                                              int div255, int* max_scratch, hipStream_t s) {
    if (!src || !out || !temp || n <= 0 || channels <= 0 || h <= 0 || w <= 0 || out_h <= 0 || out_w <= 0 || (mean == nullptr) != (stdv == nullptr)) return ANTMMF_EINVAL;
    if (div255 < -1 || div255 > 1 || (div255 < 0 && mean && !max_scratch)) return ANTMMF_EINVAL;
    FrameArgs a;
    a.src = (const uint8_t*)src; a.out = out; a.maxbits = max_scratch; a.mean = mean; a.stdv = stdv;
    a.sn = sn; a.sc = sc; a.sh = sh; a.sw = sw; a.on = on; a.oc = oc; a.oh = oh;
    a.n = n; a.C = channels; a.h = h; a.w = w; a.out_h = out_h; a.out_w = out_w;
    a.scale_h = (float)h / (float)out_h; a.scale_w = (float)w / (float)out_w;
    a.force_div255 = mean ? div255 : 0;
    auto grid_of = [](long total) { long b = (total + 255) / 256; return (unsigned)(b > 16384 ? 16384 : b); };
    hipLaunchKernelGGL(frames_aa_h_kernel, dim3(grid_of((long)n * channels * h * out_w)), dim3(256), 0, s, a, temp);
    const unsigned gv = grid_of((long)n * channels * out_h * out_w);
    if (mean && div255 < 0) {
        if (hipMemsetAsync(max_scratch, 0, sizeof(int), s) != hipSuccess) return ANTMMF_ELAUNCH;
        hipLaunchKernelGGL(frames_aa_v_kernel<0>, dim3(gv), dim3(256), 0, s, a, (const float*)temp);
    }
    hipLaunchKernelGGL(frames_aa_v_kernel<1>, dim3(gv), dim3(256), 0, s, a, (const float*)temp);
    return antmmf_check_launch();
}
