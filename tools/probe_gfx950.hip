// probe_gfx950.hip -- prints the lane mappings of two gfx950 instructions the GEMM v2 design relies on:
//   ds_read_b64_tr_b16 (LDS transpose read) and global_load_lds_dwordx4 (async global -> LDS copy).
// Build: hipcc --offload-arch=gfx950 -O3 probe_gfx950.hip -o probe_gfx950 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4 lds_s4;

__global__ void k_tr(const unsigned short* in, unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x;
    int off;
    if (mode == 0) off = lane * 4;                                                  // lane-linear, 8 B per lane
    else if (mode == 1) off = (lane >> 4) * 1024 + ((lane & 15) >> 2) * 64 + (lane & 3) * 4;   // rows of 64 elems: row = (l&15)>>2, chunk = l&3
    else if (mode == 2) off = (lane >> 4) * 1024 + (lane & 3) * 64 + ((lane & 15) >> 2) * 4;   // row = l&3, chunk = (l&15)>>2
    else off = (lane >> 4) * 1024 + (lane & 15) * 64;                               // 16 distinct rows, same column chunk
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + off));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}

__global__ void k_glds(const unsigned int* in, unsigned int* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const int lane = threadIdx.x;
    // each lane points at 16 B of global memory: lane l -> dwords [4*perm(l) .. +3]
    const int src = mode == 0 ? lane : (lane ^ 5);
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(in + 4 * src),
                                     (void __attribute__((address_space(3)))*)(lds + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}

int main() {
    std::vector<unsigned short> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (unsigned short)i;
    unsigned short *din, *dout;
    hipMalloc(&din, 8192 * 2); hipMalloc(&dout, 256 * 2);
    hipMemcpy(din, h.data(), 8192 * 2, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, din, dout, mode);
        std::vector<unsigned short> o(256);
        hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("tr16_b64 mode %d (lds element index per lane, e0..e3):\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  l%02d: %5d %5d %5d %5d", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]); if (l % 4 == 3) printf("\n"); }
    }
    std::vector<unsigned int> g(4096);
    for (int i = 0; i < 4096; ++i) g[i] = i;
    unsigned int *gin, *gout;
    hipMalloc(&gin, 4096 * 4); hipMalloc(&gout, 2048 * 4);
    hipMemcpy(gin, g.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k_glds, dim3(1), dim3(64), 0, 0, gin, gout, mode);
        std::vector<unsigned int> o(2048);
        hipMemcpy(o.data(), gout, 2048 * 4, hipMemcpyDeviceToHost);
        printf("global_load_lds x4 mode %d: lds dword index -> value (non-poison only, first dword of each 16 B):\n", mode);
        int cnt = 0;
        for (int i = 0; i < 2048; ++i) if (o[i] != 0xdeadbeefu && (i % 4 == 0)) { printf("  [%4d]=%4u", i, o[i]); if (++cnt % 8 == 0) printf("\n"); }
        printf("\n");
    }
    return 0;
}
