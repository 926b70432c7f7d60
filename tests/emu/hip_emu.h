// hip_emu.h -- TEST INFRASTRUCTURE: a minimal lane-level CPU emulation of the HIP constructs used by
// ant-multi-modal-framework_amd/csrc/*.hip, so the kernels' index arithmetic (LDS swizzles, MFMA
// fragment bookkeeping, guards, reductions, epilogues) can be exercised in the GPU-less build
// container.  Every HIP thread is an OS thread; a wave's cross-lane operations (shuffles, MFMA)
// rendezvous on a per-wave barrier; __syncthreads() is a per-workgroup barrier; workgroups run one
// after another.  The MFMA model implements the documented gfx950 16x16x32 bf16 fragment layout
// (A: lane l holds A[l&15][8*(l>>4)+e]; B: lane l holds B[8*(l>>4)+e][l&15]; D: lane l holds
// D[4*(l>>4)+r][l&15]).  It is never linked into the product library and never runs on the GPU box's
// product path; the real-hardware parity tests (-m gpu) remain the authority.
#pragma once
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
static inline float2 make_float2(float a, float b) { return {a, b}; }
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return {a, b}; }
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return {a, b, c, d}; }
static inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
#define hipFuncAttributeMaxDynamicSharedMemorySize 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }

namespace emu {
struct Wave {
    std::barrier<> bar{64};
    float fbuf[64];
    uint16_t a[64][8], b[64][8];
};
struct Block {
    int nthreads;
    std::unique_ptr<std::barrier<>> bar;
    std::vector<std::unique_ptr<Wave>> waves;
    std::vector<char> dyn;
};
struct TLS { dim3 tid, bid, gdim, bdim; Block* blk; Wave* wave; int lane; };
inline thread_local TLS tls;

// One set of OS threads per LAUNCH (not per workgroup: spawning 256 - 512 threads costs ~10 ms, an elementwise launch of a few hundred workgroups was seconds):
// the threads walk the workgroups in order; between two workgroups they meet at a gate, thread 0 builds the next workgroup's state (fresh barriers --
// threads that return early DROP out of a workgroup's barriers -- and zeroed LDS), and they meet again.
template <typename F>
void launch(F&& body, dim3 grid, dim3 block, size_t lds) {
    const int nt = (int)(block.x * block.y * block.z);
    const long nblocks = (long)grid.x * grid.y * grid.z;
    if (nblocks <= 0 || nt <= 0) return;
    std::unique_ptr<Block> cur;
    auto make_block = [&] {
        cur.reset(new Block());
        cur->nthreads = nt;
        cur->bar = std::make_unique<std::barrier<>>(nt);
        for (int w = 0; w < (nt + 63) / 64; ++w) cur->waves.emplace_back(new Wave());
        cur->dyn.assign(lds + 64, 0);
    };
    make_block();
    std::barrier<> gate(nt);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            for (long bi = 0; bi < nblocks; ++bi) {
                if (bi > 0) {
                    gate.arrive_and_wait();          // every thread is out of the previous workgroup
                    if (t == 0) make_block();
                    gate.arrive_and_wait();
                }
                Block* blk = cur.get();
                tls.tid = dim3((unsigned)t, 0, 0);
                tls.bid = dim3((unsigned)(bi % grid.x), (unsigned)((bi / grid.x) % grid.y), (unsigned)(bi / ((long)grid.x * grid.y)));
                tls.gdim = grid;
                tls.bdim = block;
                tls.blk = blk;
                tls.wave = blk->waves[t / 64].get();
                tls.lane = t % 64;
                body();
                // a thread that returns early must keep the barriers balanced
                tls.wave->bar.arrive_and_drop();
                blk->bar->arrive_and_drop();
            }
        });
    for (auto& x : th) x.join();
}
inline char* dyn_smem() {
    char* p = tls.blk->dyn.data();
    return p + ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15);
}
}  // namespace emu

#define threadIdx (emu::tls.tid)
#define blockIdx (emu::tls.bid)
#define gridDim (emu::tls.gdim)
#define blockDim (emu::tls.bdim)
namespace emu {
// ANTMMF_EMU_TRACE=1: host seconds, launches and workgroups per kernel, printed at exit (where an emulated test spends its time)
struct Trace {
    std::map<std::string, std::array<double, 3>> acc;
    bool on = std::getenv("ANTMMF_EMU_TRACE") != nullptr;
    ~Trace() { if (on) for (auto& kv : acc) std::fprintf(stderr, "[emu] %9.3f s %6.0f launches %9.0f workgroups  %s\n", kv.second[0], kv.second[1], kv.second[2], kv.first.c_str()); }
};
inline Trace& trace() { static Trace t; return t; }
template <typename F>
void traced_launch(const char* name, F&& body, dim3 grid, dim3 block, size_t lds) {
    Trace& t = trace();
    if (!t.on) { launch(body, grid, block, lds); return; }
    const auto t0 = std::chrono::steady_clock::now();
    launch(body, grid, block, lds);
    auto& a = t.acc[name];
    a[0] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); a[1] += 1; a[2] += (double)grid.x * grid.y * grid.z;
}
}  // namespace emu
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) emu::traced_launch(#kernel, [&] { kernel(__VA_ARGS__); }, grid, block, lds)

static inline void __syncthreads() { emu::tls.blk->bar->arrive_and_wait(); }
static inline void emu_wave_barrier() { emu::tls.wave->bar.arrive_and_wait(); }
static inline float __shfl_xor(float v, int mask, int = 64) {
    auto* w = emu::tls.wave;
    w->fbuf[emu::tls.lane] = v;
    w->bar.arrive_and_wait();
    const float r = w->fbuf[emu::tls.lane ^ mask];
    w->bar.arrive_and_wait();
    return r;
}
static inline float __shfl(float v, int src, int = 64) {
    auto* w = emu::tls.wave;
    w->fbuf[emu::tls.lane] = v;
    w->bar.arrive_and_wait();
    const float r = w->fbuf[src & 63];
    w->bar.arrive_and_wait();
    return r;
}
// wave-wide reductions in TWO barriers instead of a six-step shuffle butterfly (12 barriers of 64 OS threads each): every lane adds the 64 values
// in butterfly order (pairs, then pairs of pairs, ...), so the result is the butterfly's bit for bit, the same in every lane
static inline float emu_wave_sum(float v) {
    auto* w = emu::tls.wave;
    w->fbuf[emu::tls.lane] = v;
    w->bar.arrive_and_wait();
    float t[64];
    for (int i = 0; i < 64; ++i) t[i] = w->fbuf[i];
    w->bar.arrive_and_wait();
    for (int n = 64; n > 1; n >>= 1)
        for (int i = 0; i < n / 2; ++i) t[i] = t[i] + t[i + n / 2];
    return t[0];
}
static inline float emu_wave_max(float v) {
    auto* w = emu::tls.wave;
    w->fbuf[emu::tls.lane] = v;
    w->bar.arrive_and_wait();
    float m = w->fbuf[0];
    for (int i = 1; i < 64; ++i) m = m > w->fbuf[i] ? m : (w->fbuf[i] > m ? w->fbuf[i] : m);
    w->bar.arrive_and_wait();
    return m;
}
static inline float atomicAdd(float* p, float v) {
    auto* a = reinterpret_cast<std::atomic<float>*>(p);
    float old = a->load();
    while (!a->compare_exchange_weak(old, old + v)) {}
    return old;
}
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return 0; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline int atomicMax(int* p, int v) {
    auto* a = reinterpret_cast<std::atomic<int>*>(p);
    int old = a->load();
    while (old < v && !a->compare_exchange_weak(old, v)) {}
    return old;
}
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float emu_expf(float x) { return std::exp(x); }
static inline float emu_logf(float x) { return std::log(x); }
#define __expf emu_expf
#define __logf emu_logf
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }

typedef __attribute__((ext_vector_type(8))) short emu_bf16x8;
typedef __attribute__((ext_vector_type(4))) float emu_f32x4;
static inline emu_f32x4 emu_mfma_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c) {
    auto* w = emu::tls.wave;
    const int l = emu::tls.lane;
    for (int e = 0; e < 8; ++e) { w->a[l][e] = (uint16_t)a[e]; w->b[l][e] = (uint16_t)b[e]; }
    w->bar.arrive_and_wait();
    emu_f32x4 d = c;
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float s = c[r];
        for (int k = 0; k < 32; ++k) {
            const float av = __uint_as_float((uint32_t)w->a[row + 16 * (k >> 3)][k & 7] << 16);
            const float bv = __uint_as_float((uint32_t)w->b[col + 16 * (k >> 3)][k & 7] << 16);
            s = std::fma(av, bv, s);
        }
        d[r] = s;
    }
    w->bar.arrive_and_wait();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma_16x16x32_bf16(a, b, c)

// wave vote: true in every lane if the predicate holds in any lane of the wave
static inline bool emu_wave_any(bool pred) {
    auto* w = emu::tls.wave;
    const int l = emu::tls.lane;
    w->fbuf[l] = pred ? 1.0f : 0.0f;
    w->bar.arrive_and_wait();
    bool any = false;
    for (int i = 0; i < 64; ++i) any = any || w->fbuf[i] != 0.0f;
    w->bar.arrive_and_wait();
    return any;
}
#define __any(p) emu_wave_any(p)

// v_mfma_f32_32x32x16_bf16: lane l holds A[l & 31][8 (l >> 5) + e], B[8 (l >> 5) + e][l & 31], D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31], r < 16
typedef __attribute__((ext_vector_type(16))) float emu_f32x16;
static inline emu_f32x16 emu_mfma_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c) {
    auto* w = emu::tls.wave;
    const int l = emu::tls.lane;
    for (int e = 0; e < 8; ++e) { w->a[l][e] = (uint16_t)a[e]; w->b[l][e] = (uint16_t)b[e]; }
    w->bar.arrive_and_wait();
    emu_f32x16 d = c;
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float s = c[r];
        for (int k = 0; k < 16; ++k) {
            const float av = __uint_as_float((uint32_t)w->a[row + 32 * (k >> 3)][k & 7] << 16);
            const float bv = __uint_as_float((uint32_t)w->b[col + 32 * (k >> 3)][k & 7] << 16);
            s = std::fma(av, bv, s);
        }
        d[r] = s;
    }
    w->bar.arrive_and_wait();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_32x32x16_bf16(a, b, c)

// v_permlane32_swap_b32 (gfx950): lanes 0-31 keep `a` and receive the upper half's `a` in `b`; lanes 32-63 receive the lower half's
// `b` in `a` and keep `b`  (vdst[32..63] <-> vsrc[0..31]).
typedef __attribute__((ext_vector_type(2))) unsigned emu_u2;
static inline emu_u2 emu_permlane32_swap(unsigned a, unsigned b) {
    auto* w = emu::tls.wave;
    const int l = emu::tls.lane;
    float fa, fb; std::memcpy(&fa, &a, 4); std::memcpy(&fb, &b, 4);
    w->fbuf[l] = l < 32 ? fb : fa;   // what this lane gives away
    w->bar.arrive_and_wait();
    float got = w->fbuf[l ^ 32];
    w->bar.arrive_and_wait();
    unsigned ug; std::memcpy(&ug, &got, 4);
    emu_u2 r;
    if (l < 32) { r[0] = a; r[1] = ug; } else { r[0] = ug; r[1] = b; }
    return r;
}
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) emu_permlane32_swap(a, b)

// ds_read_b64_tr_b16 as measured on gfx950 (profiles/r1_probe_gfx950_tr16_glds.txt): within each 16-lane group, lane j
// supplies the address of 4 contiguous 16-bit elements; result element r of lane i is element (i & 3) of the chunk
// supplied by lane 4 r + (i >> 2)  (a 4 x 16 block, row = supplying lane >> 2, read out by columns).
typedef __attribute__((ext_vector_type(4))) short emu_s4;
static inline emu_s4 emu_ds_read_tr16_b64(const void* p) {
    auto* w = emu::tls.wave;
    const int l = emu::tls.lane;
    std::memcpy(w->a[l], p, 8);
    w->bar.arrive_and_wait();
    emu_s4 out;
    const int base = l & ~15, i = l & 15;
    for (int r = 0; r < 4; ++r) out[r] = (short)w->a[base + 4 * r + (i >> 2)][i & 3];
    w->bar.arrive_and_wait();
    return out;
}
