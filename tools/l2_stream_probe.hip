// l2_stream_probe.hip -- how fast can one CU pull L2-resident operand panels?  256 workgroups x 8 waves stream a per-XCD-shared
// region (so every line is an L2 hit after first touch) as 1-KiB wave pieces, (a) with LDS-DMA (global_load_lds_dwordx4) or
// (b) with plain global_load_dwordx4 into registers; pieces are either 8 rows x 128 B of a row-major matrix (row stride `ld`
// bytes: the GEMM operand pattern) or 1 KiB contiguous.  Prints B/clk/CU-equivalent GB/s per CU and the aggregate.
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2_stream_probe.hip -o tools/l2_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u4;
// MODE bit 0: 1 = plain loads to registers, 0 = LDS-DMA; bit 1: 1 = contiguous 1-KiB pieces, 0 = 8 rows x 128 B
template <int MODE, int WAVES, int ROWS = 8, int DEPTH = 8>
__global__ __launch_bounds__(64 * WAVES) void k(const char* __restrict__ base, long region_bytes, long ld, int pieces_per_wave, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7;
    const char* reg = base + (long)xcd * region_bytes;  // all workgroups of an XCD share one region
    const long rows = region_bytes / ld;                // rows of `ld` bytes
    u4 accv = {0, 0, 0, 0};
    // piece p of this wave: (row block, 128-B column) walking the region; different waves / workgroups start at different places
    // all sizes are powers of two: index arithmetic is shifts / masks only (32-bit)
    const unsigned rmask = (unsigned)(region_bytes - 1);
    const unsigned ldb = (unsigned)ld;
    const unsigned nrb_mask = (unsigned)(region_bytes / ld / ROWS) - 1;   // row blocks - 1 (non-pow2 ld: rounded by the host to fit)
    constexpr unsigned SEG = 1024 / ROWS, LPR = SEG / 16;
    const unsigned ncol = ldb / SEG;                                       // segments per row (may be non-pow2: wrap by compare)
    unsigned pos = ((blockIdx.x >> 3) * WAVES + wave) * 977u;
    unsigned rb = pos & nrb_mask, col = (pos >> 3) % ncol;
    const unsigned lane_off = (lane / LPR) * ldb + (lane % LPR) * 16;
    for (int p = 0; p < pieces_per_wave; ++p) {
        const char* src;
        if (MODE & 2) { src = reg + ((pos * 1024u) & rmask) + lane * 16; pos += 61; }
        else {
            src = reg + (rb * ROWS) * ldb + col * SEG + lane_off;
            rb = (rb + 5) & nrb_mask;
            col = col + 1 == ncol ? 0 : col + 1;
        }
        if (MODE & 1) {
            u4 v = *reinterpret_cast<const u4*>(src);
            accv ^= v;  // the compiler waits per use; keep several in flight by unrolling below
        } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + wave * 16384 + (p & 15) * 1024), 16, 0, 0);
            if ((p % DEPTH) == DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (accv[0] == 0x12345678u && accv[1] == 42u) sink[threadIdx.x] = accv[2];
}
template <int MODE, int WAVES, int ROWS = 8, int DEPTH = 8>
static void run(const char* name, const char* buf, long region, long ld, int pieces, unsigned* sink) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, WAVES, ROWS, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<MODE, WAVES, ROWS, DEPTH>), dim3(256), dim3(64 * WAVES), 131072, 0, buf, region, ld, pieces, sink);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = 256.0 * WAVES * pieces * 1024.0;
        if (rep == 2) printf("{\"probe\": \"%s\", \"waves\": %d, \"rows\": %d, \"depth\": %d, \"region_MB\": %.1f, \"ld\": %ld, \"ms\": %.3f, \"TBps\": %.2f, \"GBps_per_CU\": %.1f}\n", name, WAVES, ROWS, DEPTH, region / 1048576.0, ld, ms,
                             bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
    }
}
int main() {
    const long region_max = 64L << 20;
    char* buf; unsigned* sink;
    CK(hipMalloc(&buf, 8 * region_max)); CK(hipMemset(buf, 1, 8 * region_max)); CK(hipMalloc(&sink, 4096));
    const int pieces = 4096;
    const long region = 2L << 20;
    for (long ld : {8192L, 2048L}) {
        run<0, 8, 8, 8>("ldsdma", buf, region, ld, pieces, sink);
        run<0, 8, 8, 16>("ldsdma", buf, region, ld, pieces, sink);
        run<0, 8, 8, 28>("ldsdma", buf, region, ld, pieces, sink);
    }
    run<0, 8, 4, 8>("ldsdma", buf, region, 8192, pieces, sink);
    run<0, 8, 2, 8>("ldsdma", buf, region, 8192, pieces, sink);
    run<0, 8, 1, 8>("ldsdma", buf, region, 8192, pieces, sink);
    run<0, 8, 4, 16>("ldsdma", buf, region, 8192, pieces, sink);
    run<0, 8, 2, 16>("ldsdma", buf, region, 8192, pieces, sink);
    run<0, 8, 1, 16>("ldsdma", buf, region, 8192, pieces, sink);
    run<2, 8, 8, 16>("ldsdma.contig", buf, region, 8192, pieces, sink);
    run<2, 8, 8, 28>("ldsdma.contig", buf, region, 8192, pieces, sink);
    return 0;
}
