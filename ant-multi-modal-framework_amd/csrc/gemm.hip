// gemm.hip -- bf16 MFMA GEMM with fused epilogues for gfx950 (CDNA4, MI355X).
//
//     out[i][j] = epilogue( alpha * sum_r P[i][r] * Q[j][r] )          i < I, j < J, r < R
//
// P and Q are bf16 matrices, each either "r-contiguous" (stored [I][R] / [J][R]) or "r-major"
// (stored [R][I] / [R][J]); the three layouts a training step needs are
//     forward   Y = X W^T      P = X  [M][K]  r-contig,  Q = W  [N][K]  r-contig          (pt=0, qt=0)
//     dgrad     dX = dY W      P = dY [M][N]  r-contig,  Q = W  [N][K]  r-major (R = N)    (pt=0, qt=1)
//     wgrad     dW = dY^T X    P = dY [M][N]  r-major,   Q = X  [M][K]  r-major (R = M)    (pt=1, qt=1)
// so no operand is ever transposed in HBM.  Replaces the cuBLAS calls behind (reference)
//   nn.MultiheadAttention in/out proj   antmmf/modules/vision/backbone/clip/model.py:231,251
//   BERT query/key/value/dense          antmmf/modules/vision/backbone/clip/modeling_bert.py:120-122,182-186,221-238
//   M2 q/k/v/out_proj, fc1/fc2          prj/M2_Encoder/vlmo/torchscale/component/multihead_attention.py:46-50, feedforward_network.py:117-128
//   patch-embed conv as GEMM            clip/model.py:289-295 ; embedding.py:49,69
//   similarity matrix                   prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:208-213 ; clip/model.py:442-444
//
// Kernel shape: 256 threads = 4 waves (2 x 2), workgroup tile 128 (i) x 128 (j) x 64 (r), each wave
// 64 x 64 as 4 x 4 v_mfma_f32_16x16x32_bf16 tiles (64 fp32 accumulators / lane).  Both operand tiles
// sit in LDS as [128 rows][64 r] bf16 (128-B rows) with the 16-B slot index XOR-swizzled by
// f(row) = ((row>>1) ^ (row>>3)) & 7, which makes the ds_read_b128 fragment reads (16 consecutive
// rows x one slot per 16-lane group) bank-conflict-free AND keeps the ds_write_b32 pattern of the
// transposing stager at <= 2-way.  r-major operands are transposed on the way into LDS: a thread
// loads two consecutive r-rows x 8 columns (2 x 16 B, coalesced along the row), interleaves them
// into 8 packed (r, r+1) words and writes one word per column.  Global loads for tile t+1 are
// issued before the MFMAs of tile t and written to the other LDS buffer after them (one barrier per
// r-step).  The MFMA is issued with Q rows as the A operand and P rows as the B operand so that a
// lane ends up holding 4 CONSECUTIVE j of one output row: bias / residual / store are 8-B (bf16) or
// 16-B (fp32) vector accesses.  Workgroup ids are remapped so that each XCD (private 4 MiB L2)
// walks consecutive j-tiles of the same i-panel.
//
// Roofline: MFMA-bound (2*I*J*R flop).  Algorithmic HBM bytes: 2*(I*R + J*R) + out bytes.
#include "common.h"
#include <cstdlib>

#define ANTMMF_GEMM_EINVAL ANTMMF_EINVAL

struct GemmArgs {
    const bf16_t* P; const bf16_t* Q; void* C;
    const float* bias;        // [J] added before the activation (nullable)
    const bf16_t* residual;   // [I][J] (ldr) added after the activation (nullable)
    bf16_t* aux;              // [I][J] (ldaux) receives the pre-activation value (nullable)
    const bf16_t* gate;       // [I][J] (ldgate): out *= act'(gate)  (dgrad through an activation; nullable)
    long ldp, ldq, ldc, ldr, ldaux, ldgate;
    int I, J, R;
    int act, c_dtype, accumulate, ksteps_per_split;
    float alpha;
    float* ws;   // split-K workspace [splits][I][J] fp32 (wgrad ring; NULL -> atomics)
    int raster;  // 0: dispatch order; 1: XCD-contiguous chunks + 4x8 patches (experiment knob)
    int aux_grad, gate_grad;  // ANTMMF_ACT_AUX_GRAD: aux receives act'(pre-activation) instead of the pre-activation; ANTMMF_ACT_GATE_GRAD: gate holds act' already
    int debug_nostore;  // LAB build only (antmmf_debug_set_gemm_variant bit 11 / 12): the staged epilogue skips its global stores / stores without the nt hint; the product kernels do not read it
    // ---- sub-LN fold (antmmf_ffn_* at the end of this file): the M2 feed-forward's gelu -> LayerNorm(4d) pair lives in the epilogues of its GEMMs
    //   1  fc1:   z = act(acc + bias) -> C,  act'(acc + bias) -> aux,  per-row (sum z, sum z^2) of the ROUNDED z -> part   (k64p; other kernels: a row pass follows)
    //   2  fc2:   C = rstd_i acc - rstd_i mu_i colv_j + bias_j + residual_ij                 rowv = [I][2] (mu, rstd);  Q = W2 gamma,  colv_j = sum_k Q[j][k]
    //   3  dgrad: C = gate_ij (A_i acc + B_i - G_i residual_ij),  A = rstd, B = -rstd m1 + mu rstd^2 m2, G = rstd^2 m2     rowv = [I][4] (mu, rstd, m1, m2),
    //             residual = z (fc1's output), gate = act'; per-column sums of C over the tile's rows -> part (k64p)
    int ffn_mode;
    const float* rowv;
    const float* colv;
    float* part;   // mode 1: [J / 64][I] float2 (column block major);  mode 3: [I / 128][J]
    // ---- tail round of the persistent NT kernel (gemm_nt_k64r_kernel): tail_cells != 0 -> the leftover tiles of an XCD chunk's walk are computed as 16 cells each
    // (one phase x one Q fragment per wave), spread over all the chunk's workgroups, inside the same launch
    int tail_cells;
};


// r-contiguous operand: 128 rows x 64 r = 1024 16-B vectors, 4 per thread
__device__ __forceinline__ void load_rc(const bf16_t* __restrict__ base, long ld, int row0, int nrows, int r0, int R, uint4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int id = threadIdx.x + 256 * q, row = id >> 3, slot = id & 7;
        const int gr = row0 + row, gc = r0 + slot * 8;
        v[q] = (gr < nrows && gc < R) ? *reinterpret_cast<const uint4*>(base + (long)gr * ld + gc) : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void store_rc(char* lds, const uint4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int id = threadIdx.x + 256 * q, row = id >> 3, slot = id & 7;
        *reinterpret_cast<uint4*>(lds + row * 128 + ((slot ^ lds_swz(row)) << 4)) = v[q];
    }
}
// r-major operand (stored [R][ncols]): 32 r-pairs x 16 column chunks = 512 units, 2 per thread
__device__ __forceinline__ void load_rm(const bf16_t* __restrict__ base, long ld, int col0, int ncols, int r0, int R, uint4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = threadIdx.x + 256 * q, c = u & 15, rp = u >> 4;
        const int gc = col0 + c * 8, gr = r0 + 2 * rp;
        const bool okc = gc < ncols;
        v[2 * q] = (okc && gr < R) ? *reinterpret_cast<const uint4*>(base + (long)gr * ld + gc) : make_uint4(0, 0, 0, 0);
        v[2 * q + 1] = (okc && gr + 1 < R) ? *reinterpret_cast<const uint4*>(base + (long)(gr + 1) * ld + gc) : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void store_rm(char* lds, const uint4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = threadIdx.x + 256 * q, c = u & 15, rp = u >> 4;
        const uint32_t a[4] = {v[2 * q].x, v[2 * q].y, v[2 * q].z, v[2 * q].w};
        const uint32_t b[4] = {v[2 * q + 1].x, v[2 * q + 1].y, v[2 * q + 1].z, v[2 * q + 1].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t lo = (e & 1) ? (a[e >> 1] >> 16) : (a[e >> 1] & 0xffffu);
            const uint32_t hi = (e & 1) ? (b[e >> 1] & 0xffff0000u) : (b[e >> 1] << 16);
            const int row = c * 8 + e;
            *reinterpret_cast<uint32_t*>(lds + row * 128 + ((((rp >> 2) ^ lds_swz(row))) << 4) + (rp & 3) * 4) = lo | hi;
        }
    }
}

// FFN: the sub-LN fold's element-wise forms are compiled in (only the generic NT kernel asks for them: inside the wgrad kernel, which shares this function, the extra
// branch cost its register allocation -- 238 VGPRs / no scratch became 234 / 528 B of spills and the kernel ran 17 % slower)
template <int TI, int TJ, bool FFN = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4_t (&acc)[TI][TJ], int i0, int j0, int wi, int wj, int l15, int grp, bool splitk) {
    // epilogue: lane holds out[i = .. + l15][j = .. + 4*grp + 0..3] for each (it, jt); wave tile = (16 TI) x (16 TJ)
#pragma unroll
    for (int it = 0; it < TI; ++it) {
        const int i = i0 + wi * (16 * TI) + it * 16 + l15;
        if (i >= g.I) continue;
#pragma unroll
        for (int jt = 0; jt < TJ; ++jt) {
            const int j = j0 + wj * (16 * TJ) + jt * 16 + grp * 4;
            if (j >= g.J) continue;  // J % 4 == 0 is required, so a 4-group is all-in or all-out
            float v[4] = {acc[it][jt][0] * g.alpha, acc[it][jt][1] * g.alpha, acc[it][jt][2] * g.alpha, acc[it][jt][3] * g.alpha};
            if (FFN && g.ffn_mode) {   // sub-LN fold, element-wise form (the statistics / column sums of these shapes come from their own small passes)
                bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (long)i * g.ldc + j;
                if (g.ffn_mode == 1) {
                    const float4 b = *reinterpret_cast<const float4*>(g.bias + j);
                    const float u[4] = {v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w};
                    float z[4], d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) act_fwd_grad<-1>(u[e], g.act, z[e], d[e]);
                    *reinterpret_cast<uint2*>(cp) = make_uint2(pack_bf2(z[0], z[1]), pack_bf2(z[2], z[3]));
                    *reinterpret_cast<uint2*>(g.aux + (long)i * g.ldaux + j) = make_uint2(pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3]));
                } else if (g.ffn_mode == 2) {
                    const float2 st = *reinterpret_cast<const float2*>(g.rowv + 2 * (long)i);
                    const float4 c = *reinterpret_cast<const float4*>(g.colv + j), b = *reinterpret_cast<const float4*>(g.bias + j);
                    const uint2 r = *reinterpret_cast<const uint2*>(g.residual + (long)i * g.ldr + j);
                    const float rs = st.y, rm = -st.y * st.x;
                    const float o0 = (rs * v[0] + rm * c.x + b.x) + bf_lo(r.x), o1 = (rs * v[1] + rm * c.y + b.y) + bf_hi(r.x);
                    const float o2 = (rs * v[2] + rm * c.z + b.z) + bf_lo(r.y), o3 = (rs * v[3] + rm * c.w + b.w) + bf_hi(r.y);
                    *reinterpret_cast<uint2*>(cp) = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
                } else {
                    const float4 st = *reinterpret_cast<const float4*>(g.rowv + 4 * (long)i);
                    const float A = st.y, G = st.y * st.y * st.w, Bc = -st.y * st.z + st.x * G;
                    const uint2 z = *reinterpret_cast<const uint2*>(g.residual + (long)i * g.ldr + j);
                    const uint2 d = *reinterpret_cast<const uint2*>(g.gate + (long)i * g.ldgate + j);
                    const float o0 = bf_lo(d.x) * ((A * v[0] + Bc) - G * bf_lo(z.x)), o1 = bf_hi(d.x) * ((A * v[1] + Bc) - G * bf_hi(z.x));
                    const float o2 = bf_lo(d.y) * ((A * v[2] + Bc) - G * bf_lo(z.y)), o3 = bf_hi(d.y) * ((A * v[3] + Bc) - G * bf_hi(z.y));
                    *reinterpret_cast<uint2*>(cp) = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
                }
                continue;
            }
            if (g.bias) {
                const float4 b = *reinterpret_cast<const float4*>(g.bias + j);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (g.aux) {
                if (g.aux_grad) *reinterpret_cast<uint2*>(g.aux + (long)i * g.ldaux + j) = make_uint2(pack_bf2(act_grad(v[0], g.act), act_grad(v[1], g.act)), pack_bf2(act_grad(v[2], g.act), act_grad(v[3], g.act)));
                else *reinterpret_cast<uint2*>(g.aux + (long)i * g.ldaux + j) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
            if (g.gate) {
                const uint2 u = *reinterpret_cast<const uint2*>(g.gate + (long)i * g.ldgate + j);
                if (g.gate_grad) { v[0] *= bf_lo(u.x); v[1] *= bf_hi(u.x); v[2] *= bf_lo(u.y); v[3] *= bf_hi(u.y); }
                else {
                    v[0] *= act_grad(bf_lo(u.x), g.act); v[1] *= act_grad(bf_hi(u.x), g.act);
                    v[2] *= act_grad(bf_lo(u.y), g.act); v[3] *= act_grad(bf_hi(u.y), g.act);
                }
            } else if (g.act != ANTMMF_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], g.act);
            }
            if (g.residual) {
                const uint2 u = *reinterpret_cast<const uint2*>(g.residual + (long)i * g.ldr + j);
                v[0] += bf_lo(u.x); v[1] += bf_hi(u.x); v[2] += bf_lo(u.y); v[3] += bf_hi(u.y);
            }
            if (g.c_dtype == ANTMMF_BF16) {
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + (long)i * g.ldc + j) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            } else {
                float* cp = reinterpret_cast<float*>(g.C) + (long)i * g.ldc + j;
                if (splitk) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(cp + e, v[e]);
                } else if (g.accumulate) {
                    float4 o = *reinterpret_cast<float4*>(cp);
                    o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                    *reinterpret_cast<float4*>(cp) = o;
                } else {
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

// bf16 outputs: the accumulator fragment layout (a lane owns 4 columns of 16 different rows) would store 16 x 32-B
// row segments per instruction -- measured on MI355X this scattered store tail costs as much as the whole K loop for the
// wide outputs (fc1 / qkv: 0.44 ms of 0.94 ms; tools/gemm_ablate.hip).  Instead each wave stages its (16 TI) x (16 TJ)
// sub-tile through its own LDS region (slot-swizzled, <= 2-way write conflicts) and writes whole 16-B-per-lane,
// 128-B-per-row segments.  bias / activation / gate / residual are applied on the fp32 accumulators before rounding.
// EPI: compile-time epilogue shape -- bit 0 bias, bit 1 residual (no alpha / activation / gate / aux); 4 = everything, decided at
// run time.  (With run-time checks inside the 32-tile unrolled loop hipcc emits ~600 basic blocks and the store tail of a
// 256 x 256 tile took 3x longer: measured 0.31 ms vs 0.11 ms of a 0.86 ms fc1 GEMM.)
// PASSES: the wave's (16 TI) x (16 TJ) tile goes through its LDS staging area in PASSES row blocks (staging bytes per wave =
// TI * 16 * TJ * 32 / PASSES)
template <int TI, int TJ, int EPI, int PASSES = 1>
__device__ __forceinline__ void gemm_epilogue_bf16_staged(const GemmArgs& g, f32x4_t (&acc)[TI][TJ], int i0, int j0, int wi, int wj,
                                                          int lane, char* wave_lds) {
    constexpr int ROWB = TJ * 32, SLOTS = ROWB / 16, TIP = TI / PASSES;
    constexpr bool GENERIC = EPI == 4, BIAS = GENERIC || (EPI & 1), RES = GENERIC || (EPI & 2);
    const int l15 = lane & 15, grp = lane >> 4;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    if (ps) WAVE_LDS_ORDER();  // the previous block's staged rows have been read back
#pragma unroll
    for (int itp = 0; itp < TIP; ++itp) {
        const int it = ps * TIP + itp;
        const int row = itp * 16 + l15;
        int i = i0 + wi * (16 * TI) + it * 16 + l15;
        i = i < g.I ? i : g.I - 1;  // clamped rows are computed but never stored
#pragma unroll
        for (int jt = 0; jt < TJ; ++jt) {
            int j = j0 + wj * (16 * TJ) + jt * 16 + grp * 4;
            j = j < g.J ? j : g.J - 4;
            float v[4] = {acc[it][jt][0], acc[it][jt][1], acc[it][jt][2], acc[it][jt][3]};
            if (GENERIC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= g.alpha;
            }
            if (BIAS && (!GENERIC || g.bias)) {
                const float4 b = *reinterpret_cast<const float4*>(g.bias + j);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (GENERIC) {
                if (g.aux) {
                    if (g.aux_grad) *reinterpret_cast<uint2*>(g.aux + (long)i * g.ldaux + j) = make_uint2(pack_bf2(act_grad(v[0], g.act), act_grad(v[1], g.act)), pack_bf2(act_grad(v[2], g.act), act_grad(v[3], g.act)));
                    else *reinterpret_cast<uint2*>(g.aux + (long)i * g.ldaux + j) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                }
                if (g.gate) {
                    const uint2 u = *reinterpret_cast<const uint2*>(g.gate + (long)i * g.ldgate + j);
                    if (g.gate_grad) { v[0] *= bf_lo(u.x); v[1] *= bf_hi(u.x); v[2] *= bf_lo(u.y); v[3] *= bf_hi(u.y); }
                    else {
                        v[0] *= act_grad(bf_lo(u.x), g.act); v[1] *= act_grad(bf_hi(u.x), g.act);
                        v[2] *= act_grad(bf_lo(u.y), g.act); v[3] *= act_grad(bf_hi(u.y), g.act);
                    }
                } else if (g.act != ANTMMF_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], g.act);
                }
            }
            if (RES && (!GENERIC || g.residual)) {
                const uint2 u = *reinterpret_cast<const uint2*>(g.residual + (long)i * g.ldr + j);
                v[0] += bf_lo(u.x); v[1] += bf_hi(u.x); v[2] += bf_lo(u.y); v[3] += bf_hi(u.y);
            }
            const int slot = jt * 2 + (grp >> 1);
            *reinterpret_cast<uint2*>(wave_lds + row * ROWB + ((slot ^ (row & (SLOTS - 1))) << 4) + (grp & 1) * 8) =
                make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        }
    }
    WAVE_LDS_ORDER();  // same-wave LDS write -> read
    constexpr int ROWS_PER_PASS = 64 / SLOTS;
    bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
    for (int pass = 0; pass < (TIP * 16) / ROWS_PER_PASS; ++pass) {
        const int row = pass * ROWS_PER_PASS + lane / SLOTS, ls = lane % SLOTS;
        const uint4 val = *reinterpret_cast<const uint4*>(wave_lds + row * ROWB + ((ls ^ (row & (SLOTS - 1))) << 4));
        const int gi = i0 + wi * (16 * TI) + ps * TIP * 16 + row, gj = j0 + wj * (16 * TJ) + ls * 8;
        if (gi < g.I) {
            bf16_t* dst = C + (long)gi * g.ldc + gj;
            if (gj + 8 <= g.J) *reinterpret_cast<uint4*>(dst) = val;
            else if (gj + 4 <= g.J) *reinterpret_cast<uint2*>(dst) = make_uint2(val.x, val.y);
        }
    }
  }
}

template <bool PT, bool QT>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
    ANTMMF_DYN_LDS(char, smem);  // [2 buffers][P tile 16 KiB | Q tile 16 KiB]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;

    // XCD-aware, bijective workgroup remap (8 XCDs): XCD x owns a contiguous range of tile ids
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tiles_j = (g.J + 127) >> 7;
    const int i0 = (wgid / tiles_j) << 7, j0 = (wgid % tiles_j) << 7;

    const int nk_total = (g.R + 63) >> 6;
    const int kbeg = blockIdx.z * g.ksteps_per_split;
    int kend = kbeg + g.ksteps_per_split;
    if (kend > nk_total) kend = nk_total;
    if (kbeg >= kend) return;

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 pv[4], qv[4];
    auto gload = [&](int kt) {
        if (PT) load_rm(g.P, g.ldp, i0, g.I, kt << 6, g.R, pv); else load_rc(g.P, g.ldp, i0, g.I, kt << 6, g.R, pv);
        if (QT) load_rm(g.Q, g.ldq, j0, g.J, kt << 6, g.R, qv); else load_rc(g.Q, g.ldq, j0, g.J, kt << 6, g.R, qv);
    };
    auto lstore = [&](int buf) {
        char* ps = smem + buf * 32768;
        if (PT) store_rm(ps, pv); else store_rc(ps, pv);
        if (QT) store_rm(ps + 16384, qv); else store_rc(ps + 16384, qv);
    };

    gload(kbeg);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kbeg; kt < kend; ++kt) {
        const bool more = kt + 1 < kend;
        if (more) gload(kt + 1);
        const char* ps = smem + cur * 32768;
        const char* qs = ps + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t qa[4], pb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int qrow = wj * 64 + t * 16 + l15;
                qa[t] = *reinterpret_cast<const bf16x8_t*>(qs + qrow * 128 + (((kk * 4 + grp) ^ lds_swz(qrow)) << 4));
                const int prow = wi * 64 + t * 16 + l15;
                pb[t] = *reinterpret_cast<const bf16x8_t*>(ps + prow * 128 + (((kk * 4 + grp) ^ lds_swz(prow)) << 4));
            }
            SCHED_FENCE();
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[jt], pb[it], acc[it][jt], 0, 0, 0);
            SCHED_FENCE();
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    gemm_epilogue<4, 4, !PT && !QT>(g, acc, i0, j0, wi, wj, l15, grp, gridDim.z > 1);
}

// ---- LDS-DMA variant for the all-r-contiguous layout (forward; dgrad against a pre-transposed weight) ----
// Same tile / fragment / epilogue code; operand tiles reach LDS with global_load_lds_dwordx4 instead of through
// VGPRs + ds_write_b128 (the VGPR->LDS store path, ~79 B/clk/CU, is what bounds the register-staged kernel: 32 KiB
// per K-step vs 512 MFMA cycles).  The DMA writes lane-linearly, so the slot swizzle is applied on the SOURCE
// address: LDS (row, physical slot p) is filled from global (row, p ^ lds_swz(row)) -- same 128-B line, coalescing
// untouched.  Rows past the matrix edge are clamped (their outputs are never stored); requires R % 64 == 0.
// Tile shapes: <NWI, NWJ, TI, TJ> = waves along i / j, 16x16 MFMA tiles per wave along i / j.
//   <2,2,4,4>: 128 x 128, 256 threads, 64 KiB LDS (2 workgroups / CU)      -- small problems
//   <2,4,8,4>: 256 x 256, 512 threads, 128 KiB LDS (1 workgroup / CU, 2 waves / SIMD): half the LDS fill bytes and
//              2/3 of the fragment reads per flop of the 128^2 tile                           -- the large GEMMs
template <int ROWS, int NW>
__device__ __forceinline__ void dma_tile(const bf16_t* __restrict__ base, long ld, int row0, int nrows, int r0, char* lds, int wave, int lane) {
    constexpr int PER_WAVE = ROWS / 8 / NW;  // 1-KiB chunks (8 rows x 128 B) per wave
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
        const int chunk = wave * PER_WAVE + q;
        const int row = chunk * 8 + (lane >> 3), pslot = lane & 7;
        int gr = row0 + row;
        gr = gr < nrows ? gr : nrows - 1;
        glds16(base + (long)gr * ld + r0 + ((pslot ^ lds_swz(row)) << 3), lds + chunk * 1024);
    }
}

template <int NWI, int NWJ, int TI, int TJ, int EPI>
__global__ __launch_bounds__(64 * NWI * NWJ) void gemm_nt_dma_kernel(const GemmArgs g) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int BM = NWI * TI * 16, BN = NWJ * TJ * 16, NW = NWI * NWJ;
    constexpr int PBYTES = BM * 128, QBYTES = BN * 128, BUF = PBYTES + QBYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave / NWJ, wj = wave % NWJ;
    const int l15 = lane & 15, grp = lane >> 4;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tiles_j = (g.J + BN - 1) / BN;
    const int i0 = (wgid / tiles_j) * BM, j0 = (wgid % tiles_j) * BN;
    const int nk = g.R >> 6;

    f32x4_t acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    dma_tile<BM, NW>(g.P, g.ldp, i0, g.I, 0, smem, wave, lane);
    dma_tile<BN, NW>(g.Q, g.ldq, j0, g.J, 0, smem + PBYTES, wave, lane);
    glds_wait_all();
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            char* nb = smem + (cur ^ 1) * BUF;
            dma_tile<BM, NW>(g.P, g.ldp, i0, g.I, (kt + 1) << 6, nb, wave, lane);
            dma_tile<BN, NW>(g.Q, g.ldq, j0, g.J, (kt + 1) << 6, nb + PBYTES, wave, lane);
        }
        const char* ps = smem + cur * BUF;
        const char* qs = ps + PBYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t qa[TJ], pb[TI];
#pragma unroll
            for (int t = 0; t < TJ; ++t) {
                const int qrow = wj * (16 * TJ) + t * 16 + l15;
                qa[t] = *reinterpret_cast<const bf16x8_t*>(qs + qrow * 128 + (((kk * 4 + grp) ^ lds_swz(qrow)) << 4));
            }
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const int prow = wi * (16 * TI) + t * 16 + l15;
                pb[t] = *reinterpret_cast<const bf16x8_t*>(ps + prow * 128 + (((kk * 4 + grp) ^ lds_swz(prow)) << 4));
            }
            SCHED_FENCE();
#pragma unroll
            for (int it = 0; it < TI; ++it)
#pragma unroll
                for (int jt = 0; jt < TJ; ++jt)
                    acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[jt], pb[it], acc[it][jt], 0, 0, 0);
            SCHED_FENCE();
        }
        glds_wait_all();
        __syncthreads();
        cur ^= 1;
    }
    if (g.c_dtype == ANTMMF_BF16 && !(g.ldc & 7)) gemm_epilogue_bf16_staged<TI, TJ, EPI>(g, acc, i0, j0, wi, wj, lane, smem + wave * (TI * 16 * TJ * 32));
    else gemm_epilogue<TI, TJ>(g, acc, i0, j0, wi, wj, l15, grp, false);
}

// ---- 4-stage LDS-DMA ring for the large all-r-contiguous GEMMs -------------------------------------------------------
// Measured on MI355X (profiles/r1_pmc_gemm_*.txt): the 2-buffer kernels above park their waves half of the time
// (SQ_WAIT_ANY 50 %, MFMA busy 29 %, LDS bank conflicts 0): every K-step issues its whole 64 KiB as one burst and
// then drains to vmcnt(0) before the barrier, so each step pays the loaded memory latency (~3 us with 27 % L2 misses).
// Here the tile is 256 x 256 x 32 with FOUR 32-KiB stages: three tiles are always in flight, a wave only waits until
// its pieces of the OLDEST tile have landed (counted vmcnt, never 0 in steady state) and the barrier is a raw s_barrier
// that does not drain the DMA queue.  LDS rows are 64 B (32 r); 16-B slot swizzle f(row) = T[(row>>2)&3], T = {0,2,3,1}
// makes the ds_read_b128 fragment reads conflict-free (each 16-lane group covers 16 rows x 1 slot = all 64 banks once).
__device__ __forceinline__ int swz32(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }  // {0,2,3,1}

template <int STAGES, int EPI>
__global__ __launch_bounds__(512) void gemm_nt_ring_kernel(const GemmArgs g) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int BM = 256, BN = 256, TI = 8, TJ = 4, NWJ = 4, G = 4;  // G: DMA pieces per wave per tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave / NWJ, wj = wave % NWJ;
    const int l15 = lane & 15, grp = lane >> 4;
    // XCD-aware remap; inside an XCD's contiguous range tiles are walked in 4 (i) x 8 (j) groups so that the ~32
    // workgroups resident on one XCD share 4 P panels and 8 Q panels through its L2
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int tiles_j = (g.J + BN - 1) / BN, tiles_i = (g.I + BM - 1) / BM;
    int ti, tj;
    if ((g.raster & 7) == 0) {          // dispatch order: workgroup b runs on XCD b % 8, so each XCD sees every 8th j-panel
        ti = bid / tiles_j; tj = bid % tiles_j;
    } else {
        const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
        const int wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
        if ((g.raster & 7) == 2) { ti = wgid / tiles_j; tj = wgid % tiles_j; }
        else {
            const int band = wgid / (4 * tiles_j), inb = wgid - band * 4 * tiles_j;
            const int rows_here = (tiles_i - band * 4) < 4 ? (tiles_i - band * 4) : 4;
            ti = band * 4 + inb % rows_here; tj = inb / rows_here;
        }
    }
    const int i0 = ti * BM, j0 = tj * BN;
    const int nk = g.R >> 5;

    f32x4_t acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // this wave's DMA source rows (clamped at the matrix edge) and slots are loop-invariant
    const int prow[2] = {wave * 32 + (lane >> 2), wave * 32 + 16 + (lane >> 2)};
    const bf16_t* psrc[2];
    const bf16_t* qsrc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = prow[q], sl = ((lane & 3) ^ swz32(row)) << 3;
        int gi = i0 + row; gi = gi < g.I ? gi : g.I - 1;
        int gj2 = j0 + row; gj2 = gj2 < g.J ? gj2 : g.J - 1;
        psrc[q] = g.P + (long)gi * g.ldp + sl;
        qsrc[q] = g.Q + (long)gj2 * g.ldq + sl;
    }
    auto issue = [&](int kt) {
        char* buf = smem + (kt % STAGES) * 32768;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            glds16(psrc[q] + (kt << 5), buf + (wave * 2 + q) * 1024);
            glds16(qsrc[q] + (kt << 5), buf + 16384 + (wave * 2 + q) * 1024);
        }
    };
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nk) issue(t);
    int issued = (STAGES - 1 < nk ? STAGES - 1 : nk) - 1;  // last tile this wave has issued
    // wait until this wave's pieces of tile kt have landed (later tiles may stay in flight: counted vmcnt)
    auto wait_tile = [&](int kt) {
        const int ahead = issued - kt;
        if (ahead >= 2) glds_wait_le<2 * G>();
        else if (ahead == 1) glds_wait_le<G>();
        else glds_wait_le<0>();
    };
    // Two-group staggered schedule (measured +21 % on the bare loop, tools/gemm_ablate.hip): waves 0-3 (group A) and waves
    // 4-7 (group B) sit pairwise on the same SIMDs; B runs half a K-step behind A, so while one group issues its DMA pieces
    // and reads its fragments the other group's 32 MFMAs keep the matrix pipe busy.  Phases are separated by workgroup
    // barriers (two per K-step); A: [wait kt | bar | issue, read kt | bar | mfma kt], B: the same one phase later, with its
    // wait for tile kt+1 pulled in front of the barrier that lets A start reading tile kt+1.
    const bool late = wave >= 4;
    if (late) { wait_tile(0); wg_barrier_lds_only(); }
    for (int kt = 0; kt < nk; ++kt) {
        if (!late) wait_tile(kt);
        wg_barrier_lds_only();
        if (kt + STAGES - 1 < nk) { issue(kt + STAGES - 1); issued = kt + STAGES - 1; }
        const char* ps = smem + (kt % STAGES) * 32768;
        const char* qs = ps + 16384;
        bf16x8_t qa[TJ], pb[TI];
#pragma unroll
        for (int t = 0; t < TJ; ++t) {
            const int row = wj * 64 + t * 16 + l15;
            qa[t] = *reinterpret_cast<const bf16x8_t*>(qs + row * 64 + ((grp ^ swz32(row)) << 4));
        }
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            const int row = wi * 128 + t * 16 + l15;
            pb[t] = *reinterpret_cast<const bf16x8_t*>(ps + row * 64 + ((grp ^ swz32(row)) << 4));
        }
        if (late && kt + 1 < nk) wait_tile(kt + 1);
        wg_barrier_lds_only();
        SCHED_FENCE();
#pragma unroll
        for (int it = 0; it < TI; ++it)
#pragma unroll
            for (int jt = 0; jt < TJ; ++jt)
                acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[jt], pb[it], acc[it][jt], 0, 0, 0);
        SCHED_FENCE();
    }
    if (!late) wg_barrier_lds_only();  // balance group B's extra leading barrier
#ifdef ANTMMF_LAB
    if (g.raster & 8) {  // lab experiment: no store tail
        float sacc = 0.f;
#pragma unroll
        for (int a2 = 0; a2 < TI; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < TJ; ++b2) sacc += acc[a2][b2][0] + acc[a2][b2][1] + acc[a2][b2][2] + acc[a2][b2][3];
        if (sacc == 123.456f) reinterpret_cast<float*>(g.C)[threadIdx.x] = sacc;
        return;
    }
#endif
    if (g.c_dtype == ANTMMF_BF16 && !(g.ldc & 7)) {
        wg_barrier_lds_only();  // every wave is done with the stage buffers (all DMA pieces were waited for above)
        gemm_epilogue_bf16_staged<TI, TJ, EPI>(g, acc, i0, j0, wi, wj, lane, smem + wave * (TI * 16 * TJ * 32));
    } else gemm_epilogue<TI, TJ>(g, acc, i0, j0, wi, wj, l15, grp, false);
}

// ---- LDS-DMA + transpose-read variant for the all-r-major layout (wgrad dW = dY^T X, reduction over tokens) ----
// Both operand tiles are DMA'd in their NATURAL layout [64 r][COLS] (rows of 2*COLS bytes) -- no register transpose, no
// ds_write -- and the MFMA fragments ("16 columns x 32 r, 8 consecutive r per lane") are produced by the hardware
// transpose read ds_read_b64_tr_b16, two per fragment.  Bank conflicts: a 32-lane half of a transpose read touches
// 8 rows (r, r+1, r+2, r+3, r+8, ..r+11) x 32 B at the same column; rows are a multiple of 256 B apart, so the 32-B
// chunk index is XOR-swizzled with ftr(r) = (r & 3) | ((r >> 3) & 1) << 2 (on the DMA source address and on the read).
// Requires R % 64 == 0, I % BM == 0, J % BN == 0 (otherwise the register-staged kernel runs).
__device__ __forceinline__ int ftr(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

template <int COLS, int NW>
__device__ __forceinline__ void dma_tile_rm(const bf16_t* __restrict__ base, long ld, int col0, int r0, char* lds, int wave, int lane) {
    constexpr int ROWB = COLS * 2, SLOTS = ROWB / 16, PER_WAVE = (64 * ROWB / 1024) / NW;
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
        const int chunk = wave * PER_WAVE + q;
        const int lin = chunk * 64 + lane;           // 16-B slot index inside the tile
        const int row = lin / SLOTS, s = lin % SLOTS;
        const int lc = (s >> 1) ^ ftr(row);          // logical 32-B chunk stored at this physical chunk
        glds16(base + (long)(r0 + row) * ld + col0 + lc * 16 + (s & 1) * 8, lds + chunk * 1024);
    }
}
// fragment: columns c16*16 .. +15 (lane l15), r = rbase .. rbase+7 with rbase = 32 kk + 8 grp
template <int ROWB>
__device__ __forceinline__ bf16x8_t frag_tr_raw(const char* tile, int rbase, int c16, int l15) {
    const int r0 = rbase + (l15 >> 2);
    const int x = ftr(r0);
    const char* p = tile + r0 * ROWB + ((c16 ^ x) << 5) + (l15 & 3) * 8;
    const bf16x4_t lo = lds_read_tr16_raw(p), hi = lds_read_tr16_raw(p + 4 * ROWB);
    return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// s_waitcnt lgkmcnt(0) that the compiler cannot move the consumers of q[0..3], p[0..7] across
__device__ __forceinline__ void lds_tr_fence(bf16x8_t (&q)[4], bf16x8_t (&p)[8]) {
#ifndef ANTMMF_EMULATE
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]),
                   "+v"(p[6]), "+v"(p[7]));
#endif
}
template <int ROWB>
__device__ __forceinline__ bf16x8_t frag_tr(const char* tile, int rbase, int c16, int l15) {
    const int r0 = rbase + (l15 >> 2);
    const int x = ftr(r0);  // ftr(r0 + 4) == ftr(r0)
    const char* p = tile + r0 * ROWB + ((c16 ^ x) << 5) + (l15 & 3) * 8;
    const bf16x4_t lo = lds_read_tr16(p), hi = lds_read_tr16(p + 4 * ROWB);
    return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int NWI, int NWJ, int TI, int TJ>
__global__ __launch_bounds__(64 * NWI * NWJ) void gemm_tn_dma_kernel(const GemmArgs g) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int BM = NWI * TI * 16, BN = NWJ * TJ * 16, NW = NWI * NWJ;
    constexpr int PBYTES = BM * 128, QBYTES = BN * 128, BUF = PBYTES + QBYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave / NWJ, wj = wave % NWJ;
    const int l15 = lane & 15, grp = lane >> 4;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tiles_j = g.J / BN;
    const int i0 = (wgid / tiles_j) * BM, j0 = (wgid % tiles_j) * BN;
    const int nk_total = g.R >> 6;
    const int kbeg = blockIdx.z * g.ksteps_per_split;
    int kend = kbeg + g.ksteps_per_split;
    if (kend > nk_total) kend = nk_total;
    if (kbeg >= kend) return;

    f32x4_t acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    dma_tile_rm<BM, NW>(g.P, g.ldp, i0, kbeg << 6, smem, wave, lane);
    dma_tile_rm<BN, NW>(g.Q, g.ldq, j0, kbeg << 6, smem + PBYTES, wave, lane);
    glds_wait_all();
    __syncthreads();
    int cur = 0;
    for (int kt = kbeg; kt < kend; ++kt) {
        if (kt + 1 < kend) {
            char* nb = smem + (cur ^ 1) * BUF;
            dma_tile_rm<BM, NW>(g.P, g.ldp, i0, (kt + 1) << 6, nb, wave, lane);
            dma_tile_rm<BN, NW>(g.Q, g.ldq, j0, (kt + 1) << 6, nb + PBYTES, wave, lane);
        }
        const char* ps = smem + cur * BUF;
        const char* qs = ps + PBYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t qa[TJ], pb[TI];
#pragma unroll
            for (int t = 0; t < TJ; ++t) qa[t] = frag_tr<BN * 2>(qs, 32 * kk + 8 * grp, wj * TJ + t, l15);
#pragma unroll
            for (int t = 0; t < TI; ++t) pb[t] = frag_tr<BM * 2>(ps, 32 * kk + 8 * grp, wi * TI + t, l15);
            SCHED_FENCE();
#pragma unroll
            for (int it = 0; it < TI; ++it)
#pragma unroll
                for (int jt = 0; jt < TJ; ++jt)
                    acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[jt], pb[it], acc[it][jt], 0, 0, 0);
            SCHED_FENCE();
        }
        glds_wait_all();
        __syncthreads();
        cur ^= 1;
    }
    gemm_epilogue<TI, TJ>(g, acc, i0, j0, wi, wj, l15, grp, gridDim.z > 1);
}

// acc += bias (+ residual) in place, in the fragment layout -- done BEFORE the next tile's DMA prologue is issued: loads return
// in order, so an epilogue that fetched its operands after the prologue would wait for the whole prologue to land first.
template <int TI, int TJ, int EPI>
__device__ __forceinline__ void epilogue_apply_operands(const GemmArgs& g, f32x4_t (&acc)[TI][TJ], int i0, int j0, int wi, int wj, int lane) {
    const int l15 = lane & 15, grp = lane >> 4;
    int jc[TJ];
#pragma unroll
    for (int jt = 0; jt < TJ; ++jt) {
        const int j = j0 + wj * (16 * TJ) + jt * 16 + grp * 4;
        jc[jt] = j < g.J ? j : g.J - 4;
    }
    if (EPI & 1) {
#pragma unroll
        for (int jt = 0; jt < TJ; ++jt) {
            const float4 b = *reinterpret_cast<const float4*>(g.bias + jc[jt]);
#pragma unroll
            for (int it = 0; it < TI; ++it) { acc[it][jt][0] += b.x; acc[it][jt][1] += b.y; acc[it][jt][2] += b.z; acc[it][jt][3] += b.w; }
        }
    }
    if (EPI & 2) {
        // residual in the fragment layout: lane (l15, grp) reads 8 B of row it * 16 + l15 per column tile, so the TJ loads of one `it`
        // cover whole 128-B lines of 16 rows.  `it` is the OUTER loop so that those loads are adjacent in time (L1 hits); with the
        // column tile outside, every line came back from L2 once per column tile (the wave's footprint exceeds L1).  Four row tiles
        // (16 loads) in flight at a time: all 32 would spill.
#pragma unroll
        for (int it = 0; it < TI; ++it) {
#ifndef ANTMMF_EMULATE
            if (!(it & 3)) asm volatile("" ::: "memory");
#endif
            int i = i0 + wi * (16 * TI) + it * 16 + l15;
            i = i < g.I ? i : g.I - 1;
            const bf16_t* rrow = g.residual + (long)i * g.ldr;
#pragma unroll
            for (int jt = 0; jt < TJ; ++jt) {
                const uint2 u = *reinterpret_cast<const uint2*>(rrow + jc[jt]);
                acc[it][jt][0] += bf_lo(u.x); acc[it][jt][1] += bf_hi(u.x); acc[it][jt][2] += bf_lo(u.y); acc[it][jt][3] += bf_hi(u.y);
            }
        }
    }
}

// bf16 tile store staged through LDS with the LDS traffic as inline asm: with LDS-DMA pieces of the next tile in flight hipcc
// guards every compiler-visible LDS access with `s_waitcnt vmcnt(0)` (see lds_read_tr16_raw), which would drain the prologue
// before the first staged row.  A wave's LDS instructions execute in order, so write -> read-back needs no wait; the read-back
// data is fenced with lgkmcnt(0) tied to the registers.  Operands (bias, residual) are already in acc.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
template <int TI, int TJ, int PASSES>
__device__ __forceinline__ void epilogue_store_bf16_staged_raw(const GemmArgs& g, f32x4_t (&acc)[TI][TJ], int i0, int j0, int wi, int wj,
                                                               int lane, char* wave_lds) {
    constexpr int ROWB = TJ * 32, SLOTS = ROWB / 16, TIP = TI / PASSES, ROWS_PER_PASS = 64 / SLOTS, NRD = (TIP * 16) / ROWS_PER_PASS;
    const int l15 = lane & 15, grp = lane >> 4;
    bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
        for (int itp = 0; itp < TIP; ++itp) {
            const int it = ps * TIP + itp, row = itp * 16 + l15;
#pragma unroll
            for (int jt = 0; jt < TJ; ++jt) {
                const int slot = jt * 2 + (grp >> 1);
                char* dst = wave_lds + row * ROWB + ((slot ^ (row & (SLOTS - 1))) << 4) + (grp & 1) * 8;
                const u32x2_t v = {pack_bf2(acc[it][jt][0], acc[it][jt][1]), pack_bf2(acc[it][jt][2], acc[it][jt][3])};
#ifdef ANTMMF_EMULATE
                *reinterpret_cast<u32x2_t*>(dst) = v;
#else
                asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dst), "v"(v) : "memory");
#endif
            }
        }
        WAVE_LDS_ORDER();
        u32x4_t val[NRD];
#pragma unroll
        for (int pass = 0; pass < NRD; ++pass) {
            const int row = pass * ROWS_PER_PASS + lane / SLOTS, ls = lane % SLOTS;
            const char* src = wave_lds + row * ROWB + ((ls ^ (row & (SLOTS - 1))) << 4);
#ifdef ANTMMF_EMULATE
            val[pass] = *reinterpret_cast<const u32x4_t*>(src);
#else
            asm volatile("ds_read_b128 %0, %1" : "=v"(val[pass]) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)src) : "memory");
#endif
        }
#ifndef ANTMMF_EMULATE
        static_assert(NRD == 4, "fence below lists four registers");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3])::"memory");
#endif
        WAVE_LDS_ORDER();
#pragma unroll
        for (int pass = 0; pass < NRD; ++pass) {
            const int row = pass * ROWS_PER_PASS + lane / SLOTS, ls = lane % SLOTS;
            const int gi = i0 + wi * (16 * TI) + ps * TIP * 16 + row, gj = j0 + wj * (16 * TJ) + ls * 8;
#ifdef ANTMMF_LAB
            if (gi < g.I && g.debug_nostore != 1) {
#else
            if (gi < g.I) {
#endif
                bf16_t* dst = C + (long)gi * g.ldc + gj;
                // non-temporal: the 0.5 GB output streams past an L2 that should keep the operand panels (same-box A/B of the step:
                // 1203.5 / 1203.6 vs 1193.8 / 1195.4 pairs/s; -1..2 % cycles per tile)
#ifndef ANTMMF_EMULATE
#ifdef ANTMMF_LAB
                if (gj + 8 <= g.J) { if (g.debug_nostore == 2) *reinterpret_cast<u32x4_t*>(dst) = val[pass]; else __builtin_nontemporal_store(val[pass], reinterpret_cast<u32x4_t*>(dst)); }
#else
                if (gj + 8 <= g.J) __builtin_nontemporal_store(val[pass], reinterpret_cast<u32x4_t*>(dst));
#endif
#else
                if (gj + 8 <= g.J) *reinterpret_cast<u32x4_t*>(dst) = val[pass];
#endif
                else if (gj + 4 <= g.J) *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){val[pass][0], val[pass][1]};
            }
        }
    }
}

// Persistent variant of gemm_nt_ring_kernel: one workgroup per CU walks its XCD's tile range; at a tile boundary the first three
// K-stages of the NEXT tile are issued before the epilogue of the current one (which stages through the one ring slot the
// prologue does not touch, 4 KiB per wave, four row blocks), so the DMA fill and most of the store drain overlap instead of
// leaving the CU idle between workgroups (one 128-KiB workgroup per CU: nothing else can hide them).  Ring slots follow a
// step counter that runs across tiles.  Same tile order as the non-persistent kernel (4 x 8 patches per XCD).
#ifdef ANTMMF_GEMM_PROF
// tools/gemm_prof.py: cycles wave 0 of every persistent workgroup spends in {K loops, operand fetch, next-tile prologue, store, tiles}
__device__ unsigned long long g_gemm_prof[256 * 8];
extern "C" int antmmf_debug_gemm_prof(unsigned long long* host) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_prof), sizeof(g_gemm_prof)) != hipSuccess) return -1;
    return 0;
}
#define PROF_DECL unsigned long long pt_[5] = {0, 0, 0, 0, 0}, pc_ = 0, pn_
#define PROF_START() (pc_ = __builtin_readcyclecounter())
#define PROF_MARK(i) (pn_ = __builtin_readcyclecounter(), pt_[i] += pn_ - pc_, pc_ = pn_)
#define PROF_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x < 256) { for (int i_ = 0; i_ < 5; ++i_) g_gemm_prof[blockIdx.x * 8 + i_] = pt_[i_]; } } while (0)
#else
#define PROF_DECL
#define PROF_START() ((void)0)
#define PROF_MARK(i) ((void)0)
#define PROF_FLUSH() ((void)0)
#endif
// CONT (default): the DMA ring runs CONTINUOUSLY across tiles -- the first three K-stages of the next tile are issued during the
// last three K-steps of the current one (the slots they free), not as a burst in front of the epilogue.  Measured reasons
// (tools/gemm_prof.py): the burst's 96 KB queued in the vector-memory path ahead of the epilogue's stores (store phase 5.7 k cycles
// per tile against a 2 k floor), and the next tile's first counted vmcnt wait had to drain those just-issued stores (loads and
// stores share the counter).  Now stages 0 and 1 of the next tile are waited for BEFORE the stores are issued, the first two steps
// of the next tile wait for nothing, and the first wait that covers the stores comes two K-steps later.
template <int EPI, bool CONT>
__global__ __launch_bounds__(512) void gemm_nt_pring_kernel(const GemmArgs g, int ntiles) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int STAGES = 4, BM = 256, BN = 256, TI = 8, TJ = 4, NWJ = 4, G = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave / NWJ, wj = wave % NWJ;
    const int l15 = lane & 15, grp = lane >> 4;
    const int tiles_j = (g.J + BN - 1) / BN, tiles_i = (g.I + BM - 1) / BM;
    const int nk = g.R >> 5;
    // XCD x owns the contiguous tile-id range [xbase, xbase + xcount); this workgroup takes ids lx, lx + per_xcd, ...
    const int xcd = blockIdx.x & 7, lx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int qd = ntiles >> 3, rm = ntiles & 7;
    const int xbase = xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd, xcount = qd + (xcd < rm ? 1 : 0);
    auto tile_origin = [&](int local, int& i0, int& j0) {
        const int wgid = xbase + local;
        const int band = wgid / (4 * tiles_j), inb = wgid - band * 4 * tiles_j;
        const int rows_here = (tiles_i - band * 4) < 4 ? (tiles_i - band * 4) : 4;
        i0 = (band * 4 + inb % rows_here) * BM; j0 = (inb / rows_here) * BN;
    };
    const int prow[2] = {wave * 32 + (lane >> 2), wave * 32 + 16 + (lane >> 2)};
    const bf16_t* psrc[2];
    const bf16_t* qsrc[2];
    auto set_sources = [&](int i0, int j0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = prow[q], sl = ((lane & 3) ^ swz32(row)) << 3;
            int gi = i0 + row; gi = gi < g.I ? gi : g.I - 1;
            int gj2 = j0 + row; gj2 = gj2 < g.J ? gj2 : g.J - 1;
            psrc[q] = g.P + (long)gi * g.ldp + sl;
            qsrc[q] = g.Q + (long)gj2 * g.ldq + sl;
        }
    };
    int gs = 0;         // ring step counter at the start of the current tile
    int issued_g = -1;  // ring step of the youngest stage this wave has issued
    // K-slice `k` of the tile psrc / qsrc point at -> ring slot of (global) step `step`
    auto issue = [&](int k, int step) {
        char* buf = smem + (step & (STAGES - 1)) * 32768;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            glds16(psrc[q] + (k << 5), buf + (wave * 2 + q) * 1024);
            glds16(qsrc[q] + (k << 5), buf + 16384 + (wave * 2 + q) * 1024);
        }
        issued_g = step;
    };
    int local = lx;
    if (local >= xcount) return;
    int i0, j0;
    tile_origin(local, i0, j0);
    set_sources(i0, j0);
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nk) issue(t, t);
    const bool late = wave >= 4;
    int prelanded = 0;  // CONT: leading stages of this tile already known to have landed (waited for before the previous tile's stores)
    PROF_DECL;
    for (;;) {
        f32x4_t acc[TI][TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < TJ; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        PROF_START();
        const bool more = local + per_xcd < xcount;
        int ni0 = 0, nj0 = 0;
        // counted vmcnt: the pieces of stage kt have landed once at most the pieces of the younger stages are outstanding.  Conservative
        // (never wrong) when stores of the previous epilogue are still in flight: loads complete in order among themselves.
        auto wait_tile = [&](int kt) {
            if (CONT && kt < prelanded) return;
            const int ahead = issued_g - (gs + kt);
            if (ahead >= 2) glds_wait_le<2 * G>();
            else if (ahead == 1) glds_wait_le<G>();
            else glds_wait_le<0>();
        };
        if (late) { wait_tile(0); wg_barrier_lds_only(); }
        for (int kt = 0; kt < nk; ++kt) {
            if (!late) wait_tile(kt);
            wg_barrier_lds_only();
            if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, gs + kt + STAGES - 1);
            else if (CONT && more) {  // the slot freed by step kt - 1 takes K-stage kt + 3 - nk of the NEXT tile
                if (kt + STAGES - 1 == nk) { tile_origin(local + per_xcd, ni0, nj0); set_sources(ni0, nj0); }
                issue(kt + STAGES - 1 - nk, gs + kt + STAGES - 1);
            }
            const char* ps = smem + ((gs + kt) & (STAGES - 1)) * 32768;
            const char* qs = ps + 16384;
            bf16x8_t qa[TJ], pb[TI];
#pragma unroll
            for (int t = 0; t < TJ; ++t) {
                const int row = wj * 64 + t * 16 + l15;
                qa[t] = *reinterpret_cast<const bf16x8_t*>(qs + row * 64 + ((grp ^ swz32(row)) << 4));
            }
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const int row = wi * 128 + t * 16 + l15;
                pb[t] = *reinterpret_cast<const bf16x8_t*>(ps + row * 64 + ((grp ^ swz32(row)) << 4));
            }
            if (late && kt + 1 < nk) wait_tile(kt + 1);
            wg_barrier_lds_only();
            SCHED_FENCE();
#pragma unroll
            for (int it = 0; it < TI; ++it)
#pragma unroll
                for (int jt = 0; jt < TJ; ++jt)
                    acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[jt], pb[it], acc[it][jt], 0, 0, 0);
            SCHED_FENCE();
        }
        if (!late) wg_barrier_lds_only();  // every wave has read its last fragments: the slot of the last stage is free
        PROF_MARK(0);
        const int ci0 = i0, cj0 = j0;
        if (EPI > 0) epilogue_apply_operands<TI, TJ, EPI>(g, acc, ci0, cj0, wi, wj, lane);
        // !CONT: the operand loads must be CONSUMED before any DMA piece is issued (else their wait would cover the pieces): an empty
        // asm that "modifies" the accumulators pins the adds here (LLVM otherwise sinks them below the prologue).  CONT: same pin, so
        // that the adds sit in front of the stage wait below and not inside the store passes.
#ifndef ANTMMF_EMULATE
        if (EPI > 0) {
#define ACC4(i) "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3])
            asm volatile("" : ACC4(0), ACC4(1), ACC4(2), ACC4(3));
            asm volatile("" : ACC4(4), ACC4(5), ACC4(6), ACC4(7));
#undef ACC4
        }
#endif
        SCHED_FENCE();
        PROF_MARK(1);
        gs += nk;
        local += per_xcd;
        if (more) {
            if (CONT) {
                i0 = ni0; j0 = nj0;
                glds_wait_le<G>();  // stages 0 and 1 of the next tile (issued two and one K-steps ago) have landed; stage 2 may fly
                prelanded = 2;
            } else {
                tile_origin(local, i0, j0);
                set_sources(i0, j0);
#pragma unroll
                for (int t = 0; t < STAGES - 1; ++t)
                    if (t < nk) issue(t, gs + t);  // slots gs .. gs+2; the epilogue below stages through slot gs+3
            }
        }
        PROF_MARK(2);
        char* stage = smem + ((gs + STAGES - 1) & (STAGES - 1)) * 32768 + wave * 4096;
#ifdef ANTMMF_GEMM_PROF
        if (g.raster & 8) {  // experiment: no store tail (does the store traffic slow the next tile's K loop?)
            float sacc = 0.f;
            for (int a = 0; a < TI; ++a) for (int b = 0; b < TJ; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
            if (sacc == 123.456f) reinterpret_cast<float*>(g.C)[threadIdx.x] = sacc;
        } else
#endif
        epilogue_store_bf16_staged_raw<TI, TJ, 4>(g, acc, ci0, cj0, wi, wj, lane, stage);
        PROF_MARK(3);
#ifdef ANTMMF_GEMM_PROF
        pt_[4] += 1;
#endif
        if (!more) { PROF_FLUSH(); return; }
    }
}

// ---- helpers of gemm_nt_k64p_kernel (asm LDS reads: invisible to hipcc's waitcnt pass, which would otherwise drain the DMA ring) ----
#ifdef ANTMMF_EMULATE
#define K64_READ(dst, addr, OFF) dst = *reinterpret_cast<const bf16x8_t*>(smem + (addr) + (OFF))
#define K64_SETPRIO(n) do {} while (0)
#define K64_FENCE8(a) do {} while (0)
#define K64_FENCE4(a) do {} while (0)
#define K64_TIE2(a) do {} while (0)
#else
#define K64_READ(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds0 + (addr)), "n"(OFF))
#define K64_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#define K64_FENCE4(a) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
#define K64_TIE2(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]))   /* orders the uses of two more asm-loaded registers behind the preceding fence (volatile asms keep their order) */
#define K64_FENCE8(a) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
#endif

#ifdef ANTMMF_EMULATE
#define K64_BARRIER() __syncthreads()
#else
#define K64_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif

// ---- gemm_nt_k64p_kernel: the large all-r-contiguous GEMMs (forward, pre-transposed dgrad) ----------------------------------------
// 256 x 256 output tile per workgroup (8 waves, 128 x 64 each), K-tiles of 64: LDS rows are whole 128-B cache lines, one LDS-DMA
// piece = 8 rows x 128 B (the BK = 32 ring moved 16 half-lines per piece), TWO 64-KiB K-tile stages.  A K-tile is consumed in FOUR
// phases of 16 MFMAs per wave: phase ph multiplies the wave's P rows [32 ph, +32) (2 fragments x 2 k-halves, read in that phase) with
// all four Q fragments (read in phase 0, kept for the K-tile): 12 / 4 / 4 / 4 fragment reads.
//   * Ping-pong with ONE barrier per phase: every wave runs the same stream L(0) M(0) L(1) M(1) ... (L = fragment reads + refills,
//     M = 16 MFMAs); waves 0-3 (group A) take their barrier behind every L, waves 4-7 (group B, same SIMDs pairwise) behind every M,
//     so inside a period A runs [M(g-1) L(g)] and B [L(g) M(g)]: the two waves of a SIMD start on opposite sides (matrix / memory)
//     and cross over, the matrix pipe is shared.  (Two barriers per phase measured 1-5 % slower; kept as variant bit 16.)
//   * Persistent, ring continuous across output tiles: a workgroup walks its XCD's tile range (4 x 8 patches per L2) and its refills
//     simply move on to the next tile's K-tiles -- no drain / refill bubble at a tile boundary.
//   * Ring discipline: a quarter read in period g (B early, A late; A's lgkmcnt(0) sits behind the next barrier) is refilled in period
//     g + 2 or later: two pieces per phase per wave, ph0 P q0 q1 (K-tile t+1) | ph1 P q2 q3 (t+1) | ph2 Q q0 q1 (t+2) | ph3 Q q2 q3 (t+2).
//     A piece is needed >= 4 periods after its issue; each wave waits for ITS pieces of what phase g + 1 reads before the barrier
//     that closes its period g: constant counted waits vmcnt {8, 9, 10, 7} in steady state, fixed ladders in the last two K-tiles of
//     the last tile.  Stores of an epilogue only add to the counter (never unsafe, at worst an early drain).
//   * Epilogues: plain / bias stage bf16 rows through 4 KiB per wave above the ring (128-B row segments).  Residual / generic
//     epilogues run from registers: the Q fragment rows are read in the order {0-3, 8-11, 4-7, 12-15}, so after one v_permlane32_swap
//     per accumulator register a lane holds 8 CONSECUTIVE columns of its output row -- residual / gate are read and the result stored
//     as 16-B accesses of 64-B row segments (the fragment-layout residual fetch cost the out-projection 19 % of its time).
// Measured on MI355X (profiles/r2_gemm_bench_*.jsonl, same box, bit-identical results): +12 ... +18 % over the BK = 32 persistent ring on
// every forward / dgrad shape of the ViT-L/14 step.  What the ablations said: the L2 -> LDS stream alone sustains 104 GB/s per CU
// (tools/l2_stream_probe.hip) but a wave issues at most one piece per ~160 cycles, so DMA issue has to be spread over all 8 waves
// and every period; without DMA the same loop runs at 1.5-1.6 PF.
// Requires I % 256 == 0, J % 256 == 0, R % 64 == 0, R >= 128, bf16 output.
#define K64F_PRIO 1
#define K64F_NODMA 2
#define K64F_DIST11 4    /* the second piece of a phase is issued behind the first MFMA of the MFMA block instead of behind the fragment reads */
#define K64F_ONEBAR 32   /* ONE barrier per 16-MFMA phase: group A runs [MFMA(g-1), LOAD(g)], group B [LOAD(g), MFMA(g)] inside each period */
#define K64F_CLK 16      /* experiment: workgroup 0 records shader-clock and 100-MHz-clock ticks across its run (effective clock under load) */
#ifndef ANTMMF_EMULATE
__device__ unsigned long long g_k64_clk[2];
extern "C" int antmmf_debug_gemm_clock(unsigned long long* host2) {
    return hipMemcpyFromSymbol(host2, HIP_SYMBOL(g_k64_clk), sizeof(g_k64_clk)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef ANTMMF_EMULATE
#define K64_NT_STORE16(ptr, val) (*reinterpret_cast<u32x4_t*>(ptr) = (val))
#else
#define K64_NT_STORE16(ptr, val) __builtin_nontemporal_store((val), reinterpret_cast<u32x4_t*>(ptr))
#endif
typedef __attribute__((ext_vector_type(2))) unsigned int k64_u2_t;

template <int EPI, int FLAGS>
__global__ __launch_bounds__(512) void gemm_nt_k64p_kernel(const GemmArgs g, int ntiles) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int BM = 256, BN = 256, TI = 8, TJ = 4, NWJ = 4;
    constexpr int STAGE = 65536, QOFF = 32768;
    constexpr bool PRIO = FLAGS & K64F_PRIO, NODMA = FLAGS & K64F_NODMA, DIST11 = FLAGS & K64F_DIST11,
                   ONEBAR = FLAGS & K64F_ONEBAR;
    // residual / generic epilogues run from registers (permuted Q rows + lane swap, 16-B accesses of 64-B row segments: the residual is read
    // coalesced); plain / bias epilogues stage through the 32 KiB of LDS above the ring (128-B row segments)
    constexpr bool SWAPEPI = (EPI & 2) || EPI == 4 || EPI == 8 || EPI == 9 || EPI >= 16;   // EPI 8: out = acc * gate (the dgrad through an activation whose derivative was stored); 16 / 32 / 64: the sub-LN fold (GemmArgs::ffn_mode 1 / 2 / 3)
    const int lane = threadIdx.x & 63;
#ifdef ANTMMF_EMULATE
    const int wave = threadIdx.x >> 6;
    const uint32_t lds0 = 0;
#else
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#endif
    (void)lds0;
    const int wi = wave / NWJ, wj = wave % NWJ;
    const int l15 = lane & 15, grp = lane >> 4;
    const int tiles_j = g.J / BN, tiles_i = g.I / BM;
    const int nk = g.R >> 6;
    // XCD x owns the contiguous tile-id range [xbase, xbase + xcount); this workgroup takes ids lx, lx + per_xcd, ... (4 x 8 patches)
    const int xcd = blockIdx.x & 7, lx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int qd = ntiles >> 3, rm = ntiles & 7;
    const int xbase = xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd, xcount = qd + (xcd < rm ? 1 : 0);
    auto tile_origin = [&](int local, int& i0, int& j0) {
        const int wgid = xbase + local;
        const int band = wgid / (4 * tiles_j), inb = wgid - band * 4 * tiles_j;
        const int rows_here = (tiles_i - band * 4) < 4 ? (tiles_i - band * 4) : 4;
        i0 = (band * 4 + inb % rows_here) * BM; j0 = (inb / rows_here) * BN;
    };
    int local = lx;
    if (local >= xcount) return;
#ifndef ANTMMF_EMULATE
    unsigned long long clk0 = 0, rt0 = 0;
    if ((FLAGS & K64F_CLK) && blockIdx.x == 0) { clk0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#endif

    // fragment read bases (see gemm_nt_k64_kernel); SWAPEPI: the Q fragment row of lane l15 is l15 with bits 2 and 3 exchanged
    const int pl15 = SWAPEPI ? ((l15 & 3) | (((l15 >> 3) & 1) << 2) | (((l15 >> 2) & 1) << 3)) : l15;
    uint32_t pbase[4], qbase[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        pbase[x] = (uint32_t)(l15 * 128 + (((grp ^ lds_swz(l15)) ^ (2 * x)) << 4) + wi * 128 * 128);
        qbase[x] = (uint32_t)(pl15 * 128 + (((grp ^ lds_swz(pl15)) ^ (2 * x)) << 4) + QOFF + wj * 64 * 128);
    }
    // DMA pieces (8 rows x 128 B); P quarter pq = rows {wi' 128 + 32 pq + [0, 32)}, wave w: wi' = w >> 2, 8-row block w & 3;
    // Q quarter pq = rows {wj' 64 + 16 pq + [0, 16)}, wave w: wj' = w >> 1, 8-row block w & 1
    const int pl = lane >> 3, pslot = lane & 7;
    const int prow0 = (wave >> 2) * 128 + (wave & 3) * 8, qrow0 = (wave >> 1) * 64 + (wave & 1) * 8;
    const long ldpb = g.ldp * 2, ldqb = g.ldq * 2;
    uint32_t pv[2], qv[4];  // per-lane byte offsets: row pl of the piece + the source-side swizzle of the 16-B slot
    {
        const uint32_t ps0 = (uint32_t)((pslot ^ lds_swz(prow0 + pl)) << 4), qs0 = (uint32_t)((pslot ^ lds_swz(qrow0 + pl)) << 4);
#pragma unroll
        for (int b = 0; b < 2; ++b) pv[b] = (uint32_t)(pl * (int)ldpb) + (ps0 ^ (uint32_t)(b << 6));
#pragma unroll
        for (int b = 0; b < 4; ++b) qv[b] = (uint32_t)(pl * (int)ldqb) + (qs0 ^ (uint32_t)(b << 5));
    }
    const char* pgc; const char* qgc; const char* pgn = nullptr; const char* qgn = nullptr;  // wave-uniform bases: current / next tile
    auto tile_bases = [&](int i0, int j0, const char*& pgx, const char*& qgx) {
        pgx = reinterpret_cast<const char*>(g.P) + (long)(i0 + prow0) * ldpb;
        qgx = reinterpret_cast<const char*>(g.Q) + (long)(j0 + qrow0) * ldqb;
    };
    auto dma_p = [&](const char* base, int pq, int kk, uint32_t stage_off) {
        glds16(base + (long)pq * 32 * ldpb + (long)kk * 128 + pv[pq & 1], smem + stage_off + (prow0 + 32 * pq) * 128);
    };
    auto dma_q = [&](const char* base, int pq, int kk, uint32_t stage_off) {
        glds16(base + (long)pq * 16 * ldqb + (long)kk * 128 + qv[pq], smem + stage_off + QOFF + (qrow0 + 16 * pq) * 128);
    };
    int i0, j0;
    tile_origin(local, i0, j0);
    tile_bases(i0, j0, pgc, qgc);
    // prologue: K-tiles 0 and 1 of the first tile (16 pieces per wave), drained once
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) { dma_p(pgc, pq, tt, tt * STAGE); dma_q(qgc, pq, tt, tt * STAGE); }
    glds_wait_all();
    wg_barrier_lds_only();

    bf16x8_t qa[8], pb[4];  // qa[2 jt + ks], pb[2 f + ks]
    const bool late = wave >= 4;
    uint32_t so = 0;     // byte offset of the current K-tile's stage (toggles every K-tile, across output tiles)
    bool first = true;   // the kernel's first K-tile: K-tile 1 is already complete, its first refills are skipped
    // one DMA piece: SLOT = 2 PH (LOAD interval) or 2 PH + 1 (MFMA interval), WT as in K64P_WAIT
    // DIST11 schedule (slot -> piece, target K-tile):  L0 P1 (t+1) | M0 P2 (t+1) | L1 P3 (t+1) | M1 Q0 (t+2) | L2 Q1 (t+2) | M2 Q2 (t+2) | L3 Q3 (t+2) | M3 P0 (t+2)
    // every quarter is refilled at least one full barrier interval after the interval in which the LAST wave waited for its reads of it
#define K64P_DMA11(SLOT, WT)                                                                                                        \
    do {                                                                                                                            \
        if (!NODMA && (WT == 0 || (WT == 1 && SLOT <= 2))) {                                                                        \
            int kk = t + (SLOT <= 2 ? 1 : 2);                                                                                       \
            const bool nx = kk >= nk;                                                                                               \
            if (nx) kk -= nk;                                                                                                       \
            const char* pbs = nx ? pgn : pgc;                                                                                       \
            const char* qbs = nx ? qgn : qgc;                                                                                       \
            const uint32_t st = SLOT <= 2 ? (so ^ STAGE) : so;                                                                      \
            if (SLOT == 0) { if (!first) dma_p(pbs, 1, kk, st); }                                                                   \
            else if (SLOT == 1) { if (!first) dma_p(pbs, 2, kk, st); }                                                              \
            else if (SLOT == 2) { if (!first) dma_p(pbs, 3, kk, st); first = false; }                                               \
            else if (SLOT == 3) dma_q(qbs, 0, kk, st);                                                                              \
            else if (SLOT == 4) dma_q(qbs, 1, kk, st);                                                                              \
            else if (SLOT == 5) dma_q(qbs, 2, kk, st);                                                                              \
            else if (SLOT == 6) dma_q(qbs, 3, kk, st);                                                                              \
            else dma_p(pbs, 0, kk, st);                                                                                             \
        }                                                                                                                           \
    } while (0)
    // WT: wait ladder -- 0 steady, 1 = K-tile nk - 2 of the LAST output tile, 2 = K-tile nk - 1 of the last output tile
#define K64P_WAIT(PH, WT)                                                                                                            \
    do {                                                                                                                            \
        if (WT == 0) { if (PH == 0) glds_wait_le<8>(); else if (PH == 1) glds_wait_le<9>(); else if (PH == 2) glds_wait_le<10>(); else glds_wait_le<7>(); } \
        else if (WT == 1) { if (PH == 0) glds_wait_le<8>(); else if (PH == 1) glds_wait_le<9>(); else if (PH == 2) glds_wait_le<8>(); else glds_wait_le<3>(); } \
        else { if (PH == 0) glds_wait_le<2>(); else if (PH == 1) glds_wait_le<1>(); else glds_wait_le<0>(); }                       \
    } while (0)
#define K64P_PHASE(PH, WT)                                                                                                          \
    do {                                                                                                                            \
        /* LOAD interval: fragment reads, then the refills */                                                                       \
        if (PH == 0) {  /* qa[2 jt + ks]: ks = 0 -> x = jt, ks = 1 -> x = jt ^ 2 */                                                 \
            K64_READ(qa[0], so + qbase[0], 0);    K64_READ(qa[1], so + qbase[2], 0);                                                \
            K64_READ(qa[2], so + qbase[1], 2048); K64_READ(qa[3], so + qbase[3], 2048);                                             \
            K64_READ(qa[4], so + qbase[2], 4096); K64_READ(qa[5], so + qbase[0], 4096);                                             \
            K64_READ(qa[6], so + qbase[3], 6144); K64_READ(qa[7], so + qbase[1], 6144);                                             \
        }                                                                                                                           \
        K64_READ(pb[0], so + pbase[((PH & 1) * 2 + 0)], PH * 4096);                                                                 \
        K64_READ(pb[1], so + pbase[((PH & 1) * 2 + 0) ^ 2], PH * 4096);                                                             \
        K64_READ(pb[2], so + pbase[((PH & 1) * 2 + 1)], PH * 4096 + 2048);                                                          \
        K64_READ(pb[3], so + pbase[((PH & 1) * 2 + 1) ^ 2], PH * 4096 + 2048);                                                      \
        K64P_DMA11(2 * PH, WT);                                                                                                     \
        K64P_WAIT(PH, WT);                                                                                                          \
        K64_BARRIER();                                                                                                              \
        /* MFMA interval */                                                                                                         \
        if (PH == 0) K64_FENCE8(qa);                                                                                                \
        K64_FENCE4(pb);                                                                                                             \
        SCHED_FENCE();                                                                                                              \
        if (PRIO) K64_SETPRIO(1);                                                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                            \
            _Pragma("unroll") for (int f = 0; f < 2; ++f)                                                                           \
                _Pragma("unroll") for (int jt = 0; jt < 4; ++jt) {                                                                  \
                    acc[2 * PH + f][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[2 * jt + ks], pb[2 * f + ks], acc[2 * PH + f][jt], 0, 0, 0); \
                    if ((ks * 8 + f * 4 + jt) == 0) { SCHED_FENCE(); K64P_DMA11(2 * PH + 1, WT); SCHED_FENCE(); } \
                }                                                                                                                   \
        if (PRIO) K64_SETPRIO(0);                                                                                                   \
        SCHED_FENCE();                                                                                                              \
        K64_BARRIER();                                                                                                              \
    } while (0)
#define K64P_TILE(WT) do { K64P_PHASE(0, WT); K64P_PHASE(1, WT); K64P_PHASE(2, WT); K64P_PHASE(3, WT); so ^= STAGE; } while (0)

    for (;;) {
        const bool more = local + per_xcd < xcount;
        int ni0 = 0, nj0 = 0;
        if (more) { tile_origin(local + per_xcd, ni0, nj0); tile_bases(ni0, nj0, pgn, qgn); }
        f32x4_t acc[TI][TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < TJ; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int tend = more ? nk : nk - 2;
        int t = 0;
        if (ONEBAR) {
            // One barrier per phase.  Period g = 4 t + ph:  group A (waves 0-3): MFMA(g - 1) then LOAD(g);  group B: LOAD(g) then MFMA(g).  The two
            // waves of a SIMD start a period on opposite sides (matrix / memory) and cross over in the middle, the matrix pipe is shared.
            // Refills (two pieces per period per wave, A then B):  ph0 P q0 q1 (t+1) | ph1 P q2 q3 (t+1) | ph2 Q q0 q1 (t+2) | ph3 Q q2 q3 (t+2):
            // a quarter read in period g (group B early, group A late; group A's lgkmcnt(0) sits behind the NEXT barrier) is refilled in period
            // g + 2 or later.  Waits at the end of a period cover what phase g + 1 reads: the same ladders as the DIST11 schedule.
#define K64Q_PIECE(IDX, WT)                                                                                                         \
    do {                                                                                                                            \
        if (!NODMA && (WT == 0 || (WT == 1 && IDX < 4))) {                                                                          \
            int kk = t + (IDX < 4 ? 1 : 2);                                                                                         \
            const bool nx = kk >= nk;                                                                                               \
            if (nx) kk -= nk;                                                                                                       \
            const char* pbs = nx ? pgn : pgc;                                                                                       \
            const char* qbs = nx ? qgn : qgc;                                                                                       \
            const uint32_t st = IDX < 4 ? (so ^ STAGE) : so;                                                                        \
            if (IDX < 4) { if (!first) dma_p(pbs, IDX, kk, st); if (IDX == 3) first = false; }                                      \
            else dma_q(qbs, IDX - 4, kk, st);                                                                                       \
        }                                                                                                                           \
    } while (0)
#define K64Q_READS(PH)                                                                                                              \
    do {                                                                                                                            \
        if (PH == 0) {                                                                                                              \
            K64_READ(qa[0], so + qbase[0], 0);    K64_READ(qa[1], so + qbase[2], 0);                                                \
            K64_READ(qa[2], so + qbase[1], 2048); K64_READ(qa[3], so + qbase[3], 2048);                                             \
            K64_READ(qa[4], so + qbase[2], 4096); K64_READ(qa[5], so + qbase[0], 4096);                                             \
            K64_READ(qa[6], so + qbase[3], 6144); K64_READ(qa[7], so + qbase[1], 6144);                                             \
        }                                                                                                                           \
        K64_READ(pb[0], so + pbase[((PH & 1) * 2 + 0)], PH * 4096);                                                                 \
        K64_READ(pb[1], so + pbase[((PH & 1) * 2 + 0) ^ 2], PH * 4096);                                                             \
        K64_READ(pb[2], so + pbase[((PH & 1) * 2 + 1)], PH * 4096 + 2048);                                                          \
        K64_READ(pb[3], so + pbase[((PH & 1) * 2 + 1) ^ 2], PH * 4096 + 2048);                                                      \
    } while (0)
    /* MPH: the phase whose MFMAs run; PIDX >= 0: the DMA piece issued behind the first MFMA (DIST11) */
#define K64Q_MFMA(MPH, PIDX, WT)                                                                                                    \
    do {                                                                                                                            \
        if (MPH == 0) K64_FENCE8(qa);                                                                                               \
        K64_FENCE4(pb);                                                                                                             \
        SCHED_FENCE();                                                                                                              \
        if (PRIO) K64_SETPRIO(1);                                                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                            \
            _Pragma("unroll") for (int f = 0; f < 2; ++f)                                                                           \
                _Pragma("unroll") for (int jt = 0; jt < 4; ++jt) {                                                                  \
                    acc[2 * MPH + f][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[2 * jt + ks], pb[2 * f + ks], acc[2 * MPH + f][jt], 0, 0, 0); \
                    if (PIDX >= 0 && (ks * 8 + f * 4 + jt) == 0) { SCHED_FENCE(); K64Q_PIECE((PIDX < 0 ? 0 : PIDX), WT); SCHED_FENCE(); } \
                }                                                                                                                   \
        if (PRIO) K64_SETPRIO(0);                                                                                                   \
        SCHED_FENCE();                                                                                                              \
    } while (0)
    /* ONE instruction stream  L(0) M(0) L(1) M(1) ...  for both groups; group A's barrier sits behind every LOAD, group B's behind every MFMA block,
       so A's periods are [M(g-1) L(g)] and B's [L(g) M(g)].  A waits (vmcnt) before the piece issued inside M(g): its ladder is B's minus that piece. */
#define K64Q_WAIT_A(PH, WT)                                                                                                         \
    do {                                                                                                                            \
        if (!DIST11) K64P_WAIT(PH, WT);                                                                                             \
        else if (WT == 0) { if (PH == 0) glds_wait_le<7>(); else if (PH == 1) glds_wait_le<8>(); else if (PH == 2) glds_wait_le<9>(); else glds_wait_le<6>(); } \
        else if (WT == 1) { if (PH == 0) glds_wait_le<7>(); else if (PH == 1) glds_wait_le<8>(); else if (PH == 2) glds_wait_le<8>(); else glds_wait_le<3>(); } \
        else K64P_WAIT(PH, WT);                                                                                                     \
    } while (0)
#define K64Q_PHASE(PH, WT)                                                                                                          \
    do {                                                                                                                            \
        K64Q_READS(PH);                                                                                                             \
        K64Q_PIECE(2 * PH, WT);                                                                                                     \
        if (!DIST11) K64Q_PIECE(2 * PH + 1, WT);                                                                                    \
        if (!late) { K64Q_WAIT_A(PH, WT); K64_BARRIER(); }                                                                          \
        if (DIST11) K64Q_MFMA(PH, 2 * PH + 1, WT); else K64Q_MFMA(PH, -1, WT);                                                      \
        if (late) { K64P_WAIT(PH, WT); K64_BARRIER(); }                                                                             \
    } while (0)
#define K64Q_TILE(WT) do { K64Q_PHASE(0, WT); K64Q_PHASE(1, WT); K64Q_PHASE(2, WT); K64Q_PHASE(3, WT); so ^= STAGE; } while (0)
            for (; t < tend; ++t) K64Q_TILE(0);
            if (!more) {
                K64Q_TILE(1);
                ++t;
                K64Q_TILE(2);
            }
#undef K64Q_TILE
#undef K64Q_PHASE
#undef K64Q_WAIT_A
#undef K64Q_MFMA
#undef K64Q_READS
#undef K64Q_PIECE
        } else {
        if (late) K64_BARRIER();  // group B runs one barrier interval behind group A
        for (; t < tend; ++t) K64P_TILE(0);
        if (!more) {
            K64P_TILE(1);
            ++t;
            K64P_TILE(2);
        }
        if (!late) K64_BARRIER();  // both groups enter the epilogue together
        }

        if (!SWAPEPI) {
            // bias in the fragment layout, then bf16 rows staged through this wave's 4 KiB above the ring (asm LDS traffic: pieces of the
            // next tile are in flight and hipcc would drain them in front of any LDS access it can see)
            if (EPI & 1) epilogue_apply_operands<TI, TJ, EPI & 1>(g, acc, i0, j0, wi, wj, lane);
            // (Measured and not kept: vmcnt is ONE in-order counter for LDS-DMA pieces and stores, so behind the 16 stores below the next tile's first
            // waits sit on the stores -- an ablation without the stores gains 6 ... 12 % on the R = 1024 shapes.  Issuing the next tile's first two
            // pieces in front of the stores and letting its first five waits allow 16 more operations in flight (every piece they cover is older
            // than the stores) kept the results bit-identical and made every shape 1.5 ... 4 % SLOWER: the vector-memory path itself is in order,
            // the pieces queue behind the store burst whatever the counter allows.  profiles/r2_gemm_store_ablation.txt)
            epilogue_store_bf16_staged_raw<TI, TJ, 4>(g, acc, i0, j0, wi, wj, lane, smem + 2 * STAGE + wave * 4096);
        } else {
            // ---- epilogue straight from the accumulators.  lane (l15, grp) holds out[i = it 16 + l15][j = jt 16 + 4 PB(grp) + r],
            // PB = {0, 2, 1, 3}; after the swap it holds 8 consecutive columns of fragment 2 p + (lane >> 5) at (grp & 1) * 8.
            constexpr bool FFN1 = EPI == 16, FFN2 = EPI == 32, FFN3 = EPI == 64;
            // GATEACT (EPI 9, round 6): out = acc * act'(gate) -- the dgrad through an activation whose PRE-ACTIVATION was kept (the recompute policy of the CLIP / BERT
            // feed-forwards): the gate tile is requested up front like GATEMUL's, the derivative is evaluated in packed fp32 in front of the multiply
            constexpr bool GENERIC = EPI == 4, GATEACT = EPI == 9, GATEMUL = EPI == 8 || GATEACT, BIAS = ((EPI & 1) && EPI < 8) || FFN1, RES = ((EPI & 2) && EPI < 8) || FFN2;
            const int pbg = ((grp & 1) << 1) | (grp >> 1);
            const int jw = j0 + wj * 64, iw = i0 + wi * 128 + l15;
            if (FFN2) {
                // fc2 with the sub-LN folded in, on the fragment layout: acc <- rstd_i acc - rstd_i mu_i c_j + b_j; the residual epilogue below does the rest
                float rs[TI], rm[TI];
#pragma unroll
                for (int it = 0; it < TI; ++it) {
                    const float2 st = *reinterpret_cast<const float2*>(g.rowv + 2 * (long)(iw + it * 16));
                    rs[it] = st.y; rm[it] = -st.y * st.x;
                }
#pragma unroll
                for (int jt = 0; jt < TJ; ++jt) {
                    const float4 c = *reinterpret_cast<const float4*>(g.colv + jw + jt * 16 + pbg * 4);
                    const float4 b = *reinterpret_cast<const float4*>(g.bias + jw + jt * 16 + pbg * 4);
#pragma unroll
                    for (int it = 0; it < TI; ++it) {
                        acc[it][jt][0] = rs[it] * acc[it][jt][0] + (rm[it] * c.x + b.x); acc[it][jt][1] = rs[it] * acc[it][jt][1] + (rm[it] * c.y + b.y);
                        acc[it][jt][2] = rs[it] * acc[it][jt][2] + (rm[it] * c.z + b.z); acc[it][jt][3] = rs[it] * acc[it][jt][3] + (rm[it] * c.w + b.w);
                    }
                }
                SCHED_FENCE();   // the 16 residual vectors below must not be requested while the row / column vectors above are live
            }
            if (FFN3) {
                // acc <- A_i acc + B_i  (the part of the LayerNorm backward that needs no tile operand)
#pragma unroll
                for (int it = 0; it < TI; ++it) {
                    const float4 st = *reinterpret_cast<const float4*>(g.rowv + 4 * (long)(iw + it * 16));
                    const float A = st.y, Bc = -st.y * st.z + st.x * (st.y * st.y * st.w);
#pragma unroll
                    for (int jt = 0; jt < TJ; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[it][jt][r] = A * acc[it][jt][r] + Bc;
                }
            }
            if (BIAS && !GENERIC) {
#pragma unroll
                for (int jt = 0; jt < TJ; ++jt) {
                    const float4 b = *reinterpret_cast<const float4*>(g.bias + jw + jt * 16 + pbg * 4);
#pragma unroll
                    for (int it = 0; it < TI; ++it) { acc[it][jt][0] += b.x; acc[it][jt][1] += b.y; acc[it][jt][2] += b.z; acc[it][jt][3] += b.w; }
                }
            }
            const int cj = jw + (lane >> 5) * 16 + (grp & 1) * 8;  // + 32 p
            bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
            // all 16 residual vectors of the wave's tile are requested up front (64 VGPRs: the K loop's fragment registers are free here): ONE
            // exposed L2 / HBM latency per output tile instead of one per half -- at R = 1024 (16 K-tiles per tile) each cost ~10 % of the tile
            // (the generic epilogue keeps two halves of 8: with gate / aux operands live as well, 16 vectors spill)
            if (FFN1) {
                // fc1 of the folded feed-forward: z = act(u) and act'(u) stored, row sums of the rounded z kept per lane and closed over the four lanes of a row
                float s1[TI], s2[TI];
#pragma unroll
                for (int it = 0; it < TI; ++it) { s1[it] = 0.f; s2[it] = 0.f; }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int it = q >> 1, p2 = q & 1, a = 2 * p2, b = 2 * p2 + 1;
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const k64_u2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[it][a][r]), __float_as_uint(acc[it][b][r]), false, false);
                        v[r] = __uint_as_float(sw[0]); v[4 + r] = __uint_as_float(sw[1]);
                    }
                    f2_t zz[4], dd[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) act_fwd_grad2<-1>((f2_t){v[2 * e], v[2 * e + 1]}, g.act, zz[e], dd[e]);
                    const u32x4_t zv = {pack_bf2(zz[0].x, zz[0].y), pack_bf2(zz[1].x, zz[1].y), pack_bf2(zz[2].x, zz[2].y), pack_bf2(zz[3].x, zz[3].y)};
                    const u32x4_t dv = {pack_bf2(dd[0].x, dd[0].y), pack_bf2(dd[1].x, dd[1].y), pack_bf2(dd[2].x, dd[2].y), pack_bf2(dd[3].x, dd[3].y)};
                    f2_t t1 = f2_splat(0.f), t2 = f2_splat(0.f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const f2_t zr = f2_bf(zv[e]); t1 += zr; t2 += zr * zr; }
                    s1[it] += t1.x + t1.y; s2[it] += t2.x + t2.y;
                    const long row = iw + it * 16;
                    const int col = cj + 32 * p2;
                    K64_NT_STORE16(C + row * g.ldc + col, zv);
                    K64_NT_STORE16(g.aux + row * g.ldaux + col, dv);
                }
#pragma unroll
                for (int it = 0; it < TI; ++it) { s1[it] = rows4_sum(s1[it]); s2[it] = rows4_sum(s2[it]); }
                if (grp == 0) {
                    float* pp = g.part + 2 * ((long)(jw >> 6) * g.I + iw);
#pragma unroll
                    for (int it = 0; it < TI; ++it) *reinterpret_cast<float2*>(pp + 2 * (it * 16)) = make_float2(s1[it], s2[it]);
                }
            } else if (FFN3) {
                // dgrad through the sub-LN and the activation: du = act' (acc' - G_i z);  column sums of du (the fc1 bias gradient) per 128-row block
                float G[TI];
#pragma unroll
                for (int it = 0; it < TI; ++it) {
                    const float4 st = *reinterpret_cast<const float4*>(g.rowv + 4 * (long)(iw + it * 16));
                    G[it] = st.y * st.y * st.w;
                }
                f2_t cs[2][4];
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) cs[p2][e] = f2_splat(0.f);
                // (quarters of the wave's tile: 4 + 4 operand vectors in flight -- halves of 8 + 8 spill next to the 128 accumulators)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    u32x4_t zv[4], dv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int it = h * 2 + (q >> 1), p2 = q & 1;
                        zv[q] = *reinterpret_cast<const u32x4_t*>(g.residual + (long)(iw + it * 16) * g.ldr + cj + 32 * p2);
                        dv[q] = *reinterpret_cast<const u32x4_t*>(g.gate + (long)(iw + it * 16) * g.ldgate + cj + 32 * p2);
                    }
                    SCHED_FENCE();
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int it = h * 2 + (q >> 1), p2 = q & 1, a = 2 * p2, b = 2 * p2 + 1;
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const k64_u2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[it][a][r]), __float_as_uint(acc[it][b][r]), false, false);
                            v[r] = __uint_as_float(sw[0]); v[4 + r] = __uint_as_float(sw[1]);
                        }
                        u32x4_t ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f2_t du = f2_bf(dv[q][e]) * ((f2_t){v[2 * e], v[2 * e + 1]} - G[it] * f2_bf(zv[q][e]));
                            ov[e] = pack_bf2(du.x, du.y);
                            cs[p2][e] += f2_bf(ov[e]);   // the gradient the wgrad sees: the rounded value
                        }
                        K64_NT_STORE16(C + (long)(iw + it * 16) * g.ldc + cj + 32 * p2, ov);
                    }
                    SCHED_FENCE();
                }
                float* pp = g.part + (long)((i0 + wi * 128) >> 7) * g.J + cj;
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    float c8[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { c8[2 * e] = row16_sum(cs[p2][e].x); c8[2 * e + 1] = row16_sum(cs[p2][e].y); }
                    if (l15 == 0) {
                        *reinterpret_cast<float4*>(pp + 32 * p2) = make_float4(c8[0], c8[1], c8[2], c8[3]);
                        *reinterpret_cast<float4*>(pp + 32 * p2 + 4) = make_float4(c8[4], c8[5], c8[6], c8[7]);
                    }
                }
            } else {
            constexpr int NH = GENERIC ? 2 : 1, NV = 16 / NH;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                u32x4_t rv[NV];
                if (RES || GATEMUL || (GENERIC && g.residual)) {
                    const bf16_t* rbase = GATEMUL ? g.gate : g.residual;
                    const long rld = GATEMUL ? g.ldgate : g.ldr;
#pragma unroll
                    for (int q = 0; q < NV; ++q) {
                        const int it = h * (NV / 2) + (q >> 1), p2 = q & 1;
                        rv[q] = *reinterpret_cast<const u32x4_t*>(rbase + (long)(iw + it * 16) * rld + cj + 32 * p2);
                    }
                }
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const int it = h * (NV / 2) + (q >> 1), p2 = q & 1, a = 2 * p2, b = 2 * p2 + 1;
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const k64_u2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[it][a][r]), __float_as_uint(acc[it][b][r]), false, false);
                        v[r] = __uint_as_float(sw[0]); v[4 + r] = __uint_as_float(sw[1]);
                    }
                    const long row = iw + it * 16;
                    const int col = cj + 32 * p2;
                    if (GENERIC) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= g.alpha;
                        if (g.bias) {
                            const float4 b0 = *reinterpret_cast<const float4*>(g.bias + col), b1 = *reinterpret_cast<const float4*>(g.bias + col + 4);
                            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                        }
                        f2_t zz[4], dd[4];   // activation and derivative of the four value pairs (both come out of one evaluation)
                        const bool want_act = !g.gate && g.act != ANTMMF_ACT_NONE;
                        if (want_act) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) act_fwd_grad2<-1>((f2_t){v[2 * e], v[2 * e + 1]}, g.act, zz[e], dd[e]);
                        }
                        if (g.aux) {
                            u32x4_t av;
                            if (g.aux_grad && want_act) av = (u32x4_t){pack_bf2(dd[0].x, dd[0].y), pack_bf2(dd[1].x, dd[1].y), pack_bf2(dd[2].x, dd[2].y), pack_bf2(dd[3].x, dd[3].y)};
                            else av = (u32x4_t){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
                            *reinterpret_cast<u32x4_t*>(g.aux + row * g.ldaux + col) = av;
                        }
                        // activation / its derivative on two elements per VALU slot: with scalar fp32 code this epilogue cost more issue cycles
                        // than a 12-K-tile main loop (fc1 of the ViT-B/16 tower: 632 TFLOP/s against 1300 for the plain shapes)
                        if (g.gate) {
                            const u32x4_t gv = *reinterpret_cast<const u32x4_t*>(g.gate + row * g.ldgate + col);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                f2_t z_unused, dz = f2_bf(gv[e]);   // gate_grad: the forward pass stored act'(pre-activation) itself
                                if (!g.gate_grad) act_fwd_grad2<-1>(f2_bf(gv[e]), g.act, z_unused, dz);
                                v[2 * e] *= dz.x; v[2 * e + 1] *= dz.y;
                            }
                        } else if (want_act) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[2 * e] = zz[e].x; v[2 * e + 1] = zz[e].y; }
                        }
                    }
                    if (GATEMUL) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            f2_t dz = f2_bf(rv[q][e]);
                            if (GATEACT) { f2_t z_unused; act_fwd_grad2<-1>(f2_bf(rv[q][e]), g.act, z_unused, dz); }
                            v[2 * e] *= dz.x; v[2 * e + 1] *= dz.y;
                        }
                    } else if (RES || (GENERIC && g.residual)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[2 * e] += bf_lo(rv[q][e]); v[2 * e + 1] += bf_hi(rv[q][e]); }
                    }
                    const u32x4_t ov = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
#ifdef ANTMMF_LAB
                    if (g.debug_nostore == 2) *reinterpret_cast<u32x4_t*>(C + row * g.ldc + col) = ov;   // (store ablations of the lab build: variant bits 11 / 12)
                    else if (g.debug_nostore != 1) K64_NT_STORE16(C + row * g.ldc + col, ov);
#else
                    K64_NT_STORE16(C + row * g.ldc + col, ov);
#endif
                }
            }
            }
        }
        if (!more) {
#ifndef ANTMMF_EMULATE
            if ((FLAGS & K64F_CLK) && blockIdx.x == 0 && threadIdx.x == 0) {
                g_k64_clk[0] = __builtin_readcyclecounter() - clk0; g_k64_clk[1] = __builtin_amdgcn_s_memrealtime() - rt0;
            }
#endif
            return;
        }
        local += per_xcd;
        i0 = ni0; j0 = nj0; pgc = pgn; qgc = qgn;
    }
#undef K64P_TILE
#undef K64P_PHASE
#undef K64P_WAIT
#undef K64P_DMA11
}

// ---- gemm_nt_k64r_kernel: the K loop of gemm_nt_k64p_kernel (one barrier per 16-MFMA phase) with a ROLLING epilogue -------------------------
// What the persistent kernel above leaves on the table (profiles/r2_gemm_store_ablation.txt, r3_gemm_residual_epilogue_ablation.txt): all 256
// workgroups reach their epilogues together, none of a CU's eight waves issues MFMAs while its 128-KB tile is converted / stored (6-12 % of
// the R = 1024 shapes) or while the residual tile is fetched (out-projection: 19 %).  Here no epilogue phase exists:
//   * phase ph of a K-tile multiplies the wave's rows [32 ph, +32), so after phase q of the LAST K-tile of an output tile the accumulators of
//     row quarter q are final: quarter q is converted (lane swap -> 8 consecutive columns per lane, then two exchanges of the register-pair index with lane bits 4
//     and 0 so that a 16-lane pass of a store writes 32 contiguous bytes of 8 rows instead of 16 bytes of 16 rows: K64R_EPIQ) and stored inside the LOAD interval of
//     phase q + 1, next to the partner wave's MFMA block; quarter 3 goes out in phase 0 of the NEXT tile's first K-tile;
//   * bias and residual are what the accumulators START from instead of what is added to the result: as soon as quarter q has been stored its
//     32 accumulator registers are dead, and the residual vectors of the NEXT tile's quarter q (four 16-B loads per lane in the store layout, exchanged
//     back before use; inline asm, so hipcc neither waits for them nor drains the DMA ring) are requested into that hole.  Three phases (> 1 us) later,
//     in phase q of the next tile's first K-tile, they are unpacked to fp32, taken back to the fragment layout (the lane swap is its own
//     inverse), the bias fragment (DMAed into a 256-B LDS strip per wave during the previous tile) is added and the MFMAs accumulate on top:
//     out = (bias + residual) + sum_r P Q in fp32 -- the same real number as "product first", rounded to bf16 once.  Without a residual the
//     first MFMA of every accumulator takes the bias fragment (or an inline zero) as its C operand: no initialisation pass at all;
//   * vmcnt ladders: one in-order counter covers DMA pieces, residual loads and stores; every wait allows exactly the operations younger than
//     the one it needs (tools/k64r_ladder.py replays the issue order and prints the tables below; smaller is safe, larger is a race);
//   * the prologue loads K-tile 0 and the Q half of K-tile 1 (what the steady-state refill schedule does not bring itself), so the first
//     K-tile issues the same operations as every other; the last tile's look-ahead refills wrap to its own first K-tiles (harmless re-reads,
//     drained before the kernel ends) instead of the two special wait ladders of the kernel above.
//   * tail round (g.tail_cells): the xcount % per_xcd tiles that would form a last, nearly empty round of the XCD chunk's walk (image tower, J = 1024: 4112 tiles on 256
//     workgroups = 16 rounds + 2 tiles per chunk -- a 17th round on 6 % of the CUs, 5.5 % of the launch) are left out of the walk and computed behind it as CELLS: the
//     256 x 256 tile is 4 phases x 4 Q fragments per wave, a cell is ONE phase x ONE Q fragment (64 x 64 outputs over the 8 waves, the full K range), so a tile is 16
//     independent cells and two leftover tiles occupy all 32 workgroups of the chunk for ~ 1 / 10 of a tile time.  No K split: no partial sums, no workspace, no second
//     launch (round 4's K-split over a second launch recovered <= 2 of the 5.5 %), and the accumulators start from bias + residual and add the K-tiles in the walk's
//     order, so a cell's outputs are BIT-IDENTICAL to what the walk would have stored.  The cell loop keeps 6 K-tiles of its two 8-KB operand pieces in flight in an
//     8-slot ring laid over the quarter regions of the two stages (a slot is refilled two barriers after its last read).
// Requires what gemm_nt_k64p_kernel requires, and R >= 192 (three K-tile roles).  EPI: bit 0 bias, bit 1 residual; 5 = bias + activation with TWO outputs (the
// activation -> C, its derivative or the pre-activation -> aux: the forward of a feed-forward whose activation output is kept for backward; g.act at run time);
// 21 = the same plus per-row (sum z, sum z^2) of the rounded activation over the wave's 64 columns -> g.part (fc1 of the sub-LN fold, GemmArgs::ffn_mode 1);
// 8 = out = acc * gate (the dgrad through an activation whose derivative the forward stored, gate_grad; round 5): the gate vectors of row quarter q are requested ONE phase
// before that quarter is stored (quarter 0 in phase 0 of the last K-tile, quarter q + 1 right behind the store of quarter q), in the store layout like the residual vectors,
// taken back to the fragment layout and multiplied into the accumulators right in front of the conversion -- one quarter's vectors (16 VGPRs) are alive at a time (two phases
// of lead = two quarters in flight put the kernel at 256 VGPRs + 24 B of scratch).
#ifdef ANTMMF_EMULATE
#define K64R_GLOAD16(dst, voff, sbase, OFF) dst = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(sbase) + (voff) + (OFF))
#define K64R_GSTORE16(voff, val, sbase, OFF) *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(sbase) + (voff) + (OFF)) = (val)
#define K64R_GSTORE16_NT K64R_GSTORE16
#define K64R_GSTORE8(voff, val, sbase) *reinterpret_cast<k64_u2_t*>(reinterpret_cast<char*>(sbase) + (voff)) = (val)
#define K64R_VMFENCE4(N, a) do {} while (0)
#define K64R_VMWAIT(N) do {} while (0)
#define K64R_LREAD16(dst, addr, OFF) dst = *reinterpret_cast<const f32x4_t*>(smem + (addr) + (OFF))
#define K64R_OPAQUE(x) do {} while (0)
#define K64R_OPAQUE_V(x) do {} while (0)
#define K64R_KEEP(x) do {} while (0)
#else
#define K64R_KEEP(x) asm volatile("" :: "v"(x))
#define K64R_GLOAD16(dst, voff, sbase, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF))
// (plain stores: inside the K loop the acknowledgement latency of a store is on the critical path -- the DMA pieces issued behind it retire behind it -- and a
// write-back store is acknowledged by the L2; measured against nt / sc0 / sc0 nt / sc1 / sc1 nt / sc0 sc1 in profiles/r4_gemm_rolling_epilogue_ab.txt)
#define K64R_GSTORE16(voff, val, sbase, OFF) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" :: "v"(voff), "v"(val), "s"(sbase), "n"(OFF) : "memory")
#define K64R_GSTORE16_NT(voff, val, sbase, OFF) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" :: "v"(voff), "v"(val), "s"(sbase), "n"(OFF) : "memory")
#define K64R_GSTORE8(voff, val, sbase) asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(voff), "v"(val), "s"(sbase) : "memory")
#define K64R_VMFENCE4(N, a) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N) : "memory")
#define K64R_VMWAIT(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory")
#define K64R_LREAD16(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds0 + (addr)), "n"(OFF))
#define K64R_OPAQUE(x) asm volatile("" : "+s"(x))
#define K64R_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif
// tools/k64r_ladder.py: end-of-phase waits W[role][phase] (roles T0 = first K-tile of an output tile, TR = steady, TE = last), the waits INIT[q] in
// front of the accumulator initialisation of row quarter q, BIASW in front of the bias strip read of a bias-only kernel
template <int EPI> struct K64RWaits;
template <> struct K64RWaits<0> { static constexpr int W[3][4] = {{24, 25, 26, 7}, {8, 9, 10, 7}, {8, 13, 18, 19}}; static constexpr int INIT[4] = {0, 0, 0, 0}; static constexpr int BIASW = 0; };
template <> struct K64RWaits<1> { static constexpr int W[3][4] = {{25, 25, 26, 7}, {8, 9, 10, 7}, {9, 14, 19, 20}}; static constexpr int INIT[4] = {0, 0, 0, 0}; static constexpr int BIASW = 24; };
template <> struct K64RWaits<2> { static constexpr int W[3][4] = {{40, 41, 42, 11}, {8, 9, 10, 7}, {8, 17, 26, 31}}; static constexpr int INIT[4] = {30, 22, 14, 6}; static constexpr int BIASW = 0; };
template <> struct K64RWaits<21> { static constexpr int W[3][4] = {{49, 49, 50, 7}, {8, 9, 10, 7}, {9, 20, 31, 38}}; static constexpr int INIT[4] = {0, 0, 0, 0}; static constexpr int BIASW = 48; };   // 8 stores + 2 row-sum stores per quarter
template <> struct K64RWaits<37> { static constexpr int W[3][4] = {{41, 41, 42, 7}, {8, 9, 10, 7}, {9, 18, 27, 32}}; static constexpr int INIT[4] = {0, 0, 0, 0}; static constexpr int BIASW = 40; };   // = EPI 5
template <> struct K64RWaits<69> { static constexpr int W[3][4] = {{41, 41, 42, 7}, {8, 9, 10, 7}, {9, 18, 27, 32}}; static constexpr int INIT[4] = {0, 0, 0, 0}; static constexpr int BIASW = 40; };   // = EPI 5
template <> struct K64RWaits<5> { static constexpr int W[3][4] = {{41, 41, 42, 7}, {8, 9, 10, 7}, {9, 18, 27, 32}}; static constexpr int INIT[4] = {0, 0, 0, 0}; static constexpr int BIASW = 40; };
template <> struct K64RWaits<3> { static constexpr int W[3][4] = {{41, 41, 42, 11}, {8, 9, 10, 7}, {9, 18, 27, 32}}; static constexpr int INIT[4] = {30, 22, 14, 6}; static constexpr int BIASW = 40; };
// EPI 8 (out = acc * gate, round 5): INIT[q] = the wait in front of the multiplication of row quarter q by its gate vectors, requested one phase earlier (only the two DMA
// pieces of the phase in between are younger; tools/k64r_ladder.py ladder_gate)
template <> struct K64RWaits<10> { static constexpr int W[3][4] = {{40, 37, 38, 7}, {8, 9, 10, 7}, {12, 21, 30, 35}}; static constexpr int INIT[4] = {2, 2, 2, 2}; static constexpr int BIASW = 0; };   // = EPI 8 (the column sums add LDS operations only)
template <> struct K64RWaits<8> { static constexpr int W[3][4] = {{40, 37, 38, 7}, {8, 9, 10, 7}, {12, 21, 30, 35}}; static constexpr int INIT[4] = {2, 2, 2, 2}; static constexpr int BIASW = 0; };

// ABL 8 (timing only, EPI 0, R = 1024): every K-tile stores ONE 16-row x 32-column block of the tile (out of the running sums: wrong values, right bytes,
// addresses and instruction count) in phase 1, nothing at the tile boundary: what a kernel whose rows finish at staggered K positions would pay for its stores.
struct K64RSpreadWaits { static constexpr int W[4] = {9, 11, 12, 8}; };
// ABL 24 (= 8 + 16): the same bytes, but the DMA pieces of all eight waves are issued by waves 4 - 7 (each also its partner's, 128 rows up) and all the stores
// by waves 0 - 3 (two blocks per K-tile): no wave that waits on its vmcnt counter for DMA pieces has a store in it, and the storing waves never wait.
struct K64RSplitWaits { static constexpr int W[4] = {16, 18, 20, 14}; };
// the two halves of that split on their own: ABL 24 = DMA by waves 4 - 7 only, every wave still stores one block per K-tile (waves 0 - 3 never wait);
// ABL 40 = every wave issues its own DMA, waves 0 - 3 store two blocks per K-tile, waves 4 - 7 none
struct K64RProxyWaits { static constexpr int W[4] = {17, 20, 22, 15}; };
struct K64RTwoStoreWaits { static constexpr int W[4] = {10, 13, 14, 9}; };
// ABL (ablations, variant bits 15 / 16): 1 = the converted rows are not stored (timing only: the ladders count the stores), 4 = non-temporal stores; 8 = spread stores
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_nt_k64r_kernel(const GemmArgs g, int ntiles) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int BM = 256, BN = 256, TI = 8, TJ = 4, NWJ = 4;
    constexpr int STAGE = 65536, QOFF = 32768, BIASOFF = 2 * STAGE;
    // EPI 10 (round 6): EPI 8 plus the column sums of the stored tile rows -- the bias gradient of the Linear in FRONT of the activation (fc1 of the CLIP / BERT feed-forwards: db1 = column
    // sums of du = (dy W2) * act'), which was a column-sum pass over the 4d-wide du per layer.  Per row quarter: the two fragments of a column are added, closed over the 16 row lanes by
    // DPP, and added into a per-workgroup strip of J floats in LDS (return-less ds_add_f32; the 30 KB behind the bias strips); the strip leaves as row blockIdx.x of g.part at the end.
    constexpr bool CSUM = EPI == 10, GATE = EPI == 8 || CSUM, BIAS = !GATE && (EPI & 1), RES = !GATE && (EPI & 2) != 0, FFN1 = EPI == 21, ACT2 = EPI == 5 || EPI == 37 || EPI == 69 || FFN1;
    // EPI 37 / 69: EPI 5 with the activation (erf-GELU / QuickGELU) and the stored-derivative policy fixed at compile time (no wave-uniform branches inside the conversion)
    constexpr int ACTC = EPI == 37 ? ANTMMF_ACT_GELU_ERF : -1;
    constexpr int CSOFF = 2 * 65536 + 2048;   // float strip [J <= 4096]
    using WT = K64RWaits<EPI>;
    const bf16_t* const rsrc = GATE ? g.gate : g.residual;   // what the "residual" vector loads fetch: the residual tile, or the gate tile
    const long rld = GATE ? g.ldgate : g.ldr;
    const int lane = threadIdx.x & 63;
#ifdef ANTMMF_EMULATE
    const int wave = threadIdx.x >> 6;
    const uint32_t lds0 = 0;
#else
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#endif
    (void)lds0;
    const int wi = wave / NWJ, wj = wave % NWJ;
    const int l15 = lane & 15, grp = lane >> 4;
    const int tiles_j = g.J / BN, tiles_i = g.I / BM;
    const int nk = g.R >> 6;
    const int xcd = blockIdx.x & 7, lx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int qd = ntiles >> 3, rm = ntiles & 7;
    const int xbase = xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd, xcount = qd + (xcd < rm ? 1 : 0);
    // (band height = rows of the tile patch an XCD's 32 workgroups cover at a time: 4 x 8 tiles in the product.  LAB: ANTMMF_GEMM_RASTER bits 5 - 6 select 8 / 16 / 2
    // rows -- the A/B for "is the weight panel, re-read once per band out of the Infinity Cache, a limiter?", profiles/r5_gemm_patch_shape_ab.txt)
#ifdef ANTMMF_LAB
    const int PR = ((g.raster >> 5) & 3) == 1 ? 8 : ((g.raster >> 5) & 3) == 2 ? 16 : ((g.raster >> 5) & 3) == 3 ? 2 : 4;
#else
    constexpr int PR = 4;
#endif
    auto tile_origin = [&](int local, int& i0, int& j0) {
        const int wgid = xbase + local;
        const int band = wgid / (PR * tiles_j), inb = wgid - band * PR * tiles_j;
        const int rows_here = (tiles_i - band * PR) < PR ? (tiles_i - band * PR) : PR;
        i0 = (band * PR + inb % rows_here) * BM; j0 = (inb / rows_here) * BN;
    };
    // tail round as cells (g.tail_cells): the xcount % per_xcd tiles that would run as a last, nearly empty round are left out of the walk; behind it the chunk's
    // workgroups share their 16 cells each (see the header)
    const int tail_r = g.tail_cells ? xcount % per_xcd : 0;
    const int xcount_main = xcount - tail_r;
    int local = lx;
    if (local >= xcount_main) return;
#ifndef ANTMMF_EMULATE
    // workgroup 0 records shader-clock and 100-MHz-clock ticks across its run (antmmf_debug_gemm_clock: the effective clock of this launch; two scalar reads)
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // fragment read bases; the Q fragment row of lane l15 is l15 with bits 2 and 3 exchanged (lane-swap store layout, see gemm_nt_k64p_kernel)
    const int pl15 = (l15 & 3) | (((l15 >> 3) & 1) << 2) | (((l15 >> 2) & 1) << 3);
    uint32_t pbase[4], qbase[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        pbase[x] = (uint32_t)(l15 * 128 + (((grp ^ lds_swz(l15)) ^ (2 * x)) << 4) + wi * 128 * 128);
        qbase[x] = (uint32_t)(pl15 * 128 + (((grp ^ lds_swz(pl15)) ^ (2 * x)) << 4) + QOFF + wj * 64 * 128);
    }
    const int pl = lane >> 3, pslot = lane & 7;
    const int prow0 = (wave >> 2) * 128 + (wave & 3) * 8, qrow0 = (wave >> 1) * 64 + (wave & 1) * 8;
    const long ldpb = g.ldp * 2, ldqb = g.ldq * 2;
    uint32_t pv[2], qv[4];
    {
        const uint32_t ps0 = (uint32_t)((pslot ^ lds_swz(prow0 + pl)) << 4), qs0 = (uint32_t)((pslot ^ lds_swz(qrow0 + pl)) << 4);
#pragma unroll
        for (int b = 0; b < 2; ++b) pv[b] = (uint32_t)(pl * (int)ldpb) + (ps0 ^ (uint32_t)(b << 6));
#pragma unroll
        for (int b = 0; b < 4; ++b) qv[b] = (uint32_t)(pl * (int)ldqb) + (qs0 ^ (uint32_t)(b << 5));
    }
    const int up128 = ((ABL & 16) && wave >= 4) ? 128 : 0;
    const char* pgc; const char* qgc; const char* pgn = nullptr; const char* qgn = nullptr;
    auto tile_bases = [&](int i0, int j0, const char*& pgx, const char*& qgx) {
        pgx = reinterpret_cast<const char*>(g.P) + (long)(i0 + prow0 - up128) * ldpb;
        qgx = reinterpret_cast<const char*>(g.Q) + (long)(j0 + qrow0 - up128) * ldqb;
    };
    // (ABL 16: waves 4 - 7 also issue the pieces of the wave 128 rows up; their tile bases then point at THAT wave's rows and their own pieces add 128 rows
    // to the per-lane offset: no second set of scalar addresses)
    uint32_t pvo[2], qvo[4];
#pragma unroll
    for (int b = 0; b < 2; ++b) pvo[b] = pv[b] + (uint32_t)(up128 * (int)ldpb);
#pragma unroll
    for (int b = 0; b < 4; ++b) qvo[b] = qv[b] + (uint32_t)(up128 * (int)ldqb);
    auto dma_p = [&](const char* base, int pq, int kk, uint32_t stage_off, int up = 0) {   // up = 1: the piece of the wave 128 rows up (ABL 16)
        glds16(base + (long)pq * 32 * ldpb + (long)kk * 128 + (up ? pv[pq & 1] : pvo[pq & 1]), smem + stage_off + (prow0 - up * 128 + 32 * pq) * 128);
    };
    auto dma_q = [&](const char* base, int pq, int kk, uint32_t stage_off, int up = 0) {
        glds16(base + (long)pq * 16 * ldqb + (long)kk * 128 + (up ? qv[pq] : qvo[pq]), smem + stage_off + QOFF + (qrow0 - up * 128 + 16 * pq) * 128);
    };
    auto dma_bias = [&](int tj0) {   // this wave's 64 bias values -> its LDS strip (4 B per lane)
#ifdef ANTMMF_EMULATE
        reinterpret_cast<float*>(smem + BIASOFF + wave * 256)[lane] = g.bias[tj0 + wj * 64 + lane];
#else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.bias + tj0 + wj * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(smem + BIASOFF + wave * 256), 4, 0, 0);
#endif
    };
    // per-lane byte offsets of the row-layout accesses (row l15 of a 16-row fragment, 8 consecutive columns at (lane >> 5) * 16 + (grp & 1) * 8; + 64 B for the second column pair)
    const uint32_t cvoff = (uint32_t)((l15 * (int)g.ldc + (lane >> 5) * 16 + (grp & 1) * 8) * 2);
    const uint32_t cvoff128 = cvoff + (uint32_t)(128 * (int)g.ldc * 2);   // (ABL 16: the same lane position 128 rows down)
    (void)cvoff128;
    // the store layout behind the two lane-bit exchanges of K64R_EPIQ: fragment row (l15 & 14) (+ 1 for the second register), 16-byte column chunk 4 l4 + 2 l5 + l0
    const int trow = l15 & 14, tchunk = ((lane >> 4) & 1) * 4 + (lane >> 5) * 2 + (lane & 1);
    const uint32_t cvoff_t0 = (uint32_t)((trow * (int)g.ldc + tchunk * 8) * 2), cvoff_t1 = cvoff_t0 + (uint32_t)((int)g.ldc * 2);
    const uint32_t avoff_t0 = ACT2 ? (uint32_t)((trow * (int)g.ldaux + tchunk * 8) * 2) : 0u, avoff_t1 = avoff_t0 + (ACT2 ? (uint32_t)((int)g.ldaux * 2) : 0u);
    const uint32_t rvoff_t0 = (RES || GATE) ? (uint32_t)((trow * (int)rld + tchunk * 8) * 2) : 0u, rvoff_t1 = rvoff_t0 + ((RES || GATE) ? (uint32_t)((int)rld * 2) : 0u);
    (void)cvoff_t0; (void)cvoff_t1; (void)avoff_t0; (void)avoff_t1; (void)rvoff_t0; (void)rvoff_t1;
    // (ABL 64, timing only: every store instruction writes 8 rows x 128 B -- eight FULL cache lines -- instead of 16 rows x 64 B; same bytes per tile, data in the wrong places)
    const uint32_t cvoff_fl0 = (uint32_t)(((lane >> 3) * (int)g.ldc + (lane & 7) * 8) * 2), cvoff_fl1 = cvoff_fl0 + (uint32_t)(8 * (int)g.ldc * 2);
    (void)cvoff_fl0; (void)cvoff_fl1;
    // (ABL 256 / 512, timing only: 8 rows x 32 B resp. 4 rows x 64 B per 16-lane pass -- what ONE resp. TWO of the lane-bit exchanges would give)
    const uint32_t cvoff_p32 = (uint32_t)((((lane >> 1) & 7) * (int)g.ldc + (2 * (lane >> 4) + (lane & 1)) * 8) * 2), cvoff_p32b = cvoff_p32 + (uint32_t)(8 * (int)g.ldc * 2);
    const uint32_t cvoff_p64 = (uint32_t)(((lane >> 2) * (int)g.ldc + (lane & 3) * 8) * 2);
    (void)cvoff_p32; (void)cvoff_p32b; (void)cvoff_p64;
    const uint32_t rvoff = (RES || GATE) ? (uint32_t)((l15 * (int)rld + (lane >> 5) * 16 + (grp & 1) * 8) * 2) : 0u;
    const uint32_t avoff = ACT2 ? (uint32_t)((l15 * (int)g.ldaux + (lane >> 5) * 16 + (grp & 1) * 8) * 2) : 0u;
    const int pbg = ((grp & 1) << 1) | (grp >> 1);
    u32x4_t rv[4][4] = {};   // residual vectors of row quarter q, in flight between the store of the previous tile's quarter q and this tile's phase q
    f32x4_t bf[4] = {};      // bias fragment of the current tile (live across its first K-tile only)
    // request the residual of row quarter QQ of the tile at (ti0, tj0)
#define K64R_RESLOAD(QQ, ti0, tj0)                                                                                                  \
    do {                                                                                                                            \
        _Pragma("unroll") for (int ih = 0; ih < 2; ++ih) {                                                                          \
            const char* rb = reinterpret_cast<const char*>(rsrc) + ((long)((ti0) + wi * 128 + (2 * (QQ) + ih) * 16) * rld + (tj0) + wj * 64) * 2; \
            if (ABL & 128) { K64R_GLOAD16(rv[QQ][2 * ih], rvoff, rb, 0); K64R_GLOAD16(rv[QQ][2 * ih + 1], rvoff, rb, 64); }        \
            else { K64R_GLOAD16(rv[QQ][2 * ih], rvoff_t0, rb, 0); K64R_GLOAD16(rv[QQ][2 * ih + 1], rvoff_t1, rb, 0); }              \
        }                                                                                                                           \
    } while (0)

    int i0, j0;
    tile_origin(local, i0, j0);
    tile_bases(i0, j0, pgc, qgc);
    if (CSUM) for (int c = threadIdx.x; c < g.J; c += 512) *reinterpret_cast<float*>(smem + CSOFF + c * 4) = 0.f;
    // prologue: K-tile 0 and the Q half of K-tile 1 (12 pieces per wave), the first tile's bias strip and residual quarters 0 - 2; drained once
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) { dma_p(pgc, pq, 0, 0); dma_q(qgc, pq, 0, 0); dma_q(qgc, pq, 1, STAGE); }
    if (BIAS) dma_bias(j0);
    if (RES) { K64R_RESLOAD(0, i0, j0); K64R_RESLOAD(1, i0, j0); K64R_RESLOAD(2, i0, j0); }
    if (RES) { K64R_VMFENCE4(0, rv[0]); K64R_VMFENCE4(0, rv[1]); K64R_VMFENCE4(0, rv[2]); }
    glds_wait_all();
    wg_barrier_lds_only();

    bf16x8_t qa[8], pb[4];
    f32x4_t acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool late = wave >= 4;
    uint32_t so = 0;
    bool pending = false;    // quarter 3 of the previous tile still sits in the accumulators
    int ei0 = 0, ej0 = 0;    // that tile's origin
    int ni0 = 0, nj0 = 0;    // the next tile's
    int t = 0;

    // convert + store row quarter QQ of the tile at (ti0, tj0): rows it 16 + l15, it in {2 QQ, 2 QQ + 1}
#define K64R_EPIQ(QQ, ti0, tj0)                                                                                                     \
    do {                                                                                                                            \
        _Pragma("unroll") for (int ih = 0; ih < 2; ++ih) {                                                                          \
            const int it = 2 * (QQ) + ih;                                                                                           \
            char* cb = reinterpret_cast<char*>(g.C) + ((long)((ti0) + wi * 128 + it * 16) * g.ldc + (tj0) + wj * 64) * 2;           \
            f2_t t1 = f2_splat(0.f), t2 = f2_splat(0.f);   /* FFN1: row sums of the ROUNDED activation (what fc2 multiplies) */      \
            u32x4_t ov2[2], av2[2];                                                                                                 \
            _Pragma("unroll") for (int p2 = 0; p2 < 2; ++p2) {                                                                      \
                u32x4_t ov, av;                                                                                                     \
                _Pragma("unroll") for (int rr = 0; rr < 4; rr += 2) {                                                               \
                    const k64_u2_t s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[it][2 * p2][rr]), __float_as_uint(acc[it][2 * p2 + 1][rr]), false, false); \
                    const k64_u2_t s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[it][2 * p2][rr + 1]), __float_as_uint(acc[it][2 * p2 + 1][rr + 1]), false, false); \
                    if (ACT2) {   /* two column pairs: activation and derivative out of one evaluation each (packed fp32 math) */           \
                        const f2_t u0 = (f2_t){__uint_as_float(s0[0]), __uint_as_float(s1[0])}, u1 = (f2_t){__uint_as_float(s0[1]), __uint_as_float(s1[1])}; \
                        f2_t z0, d0, z1, d1;                                                                                        \
                        act_fwd_grad2<ACTC>(u0, EPI == 69 ? ANTMMF_ACT_QUICK_GELU : g.act, z0, d0);                                 \
                        act_fwd_grad2<ACTC>(u1, EPI == 69 ? ANTMMF_ACT_QUICK_GELU : g.act, z1, d1);                                 \
                        ov[rr >> 1] = pack_bf2(z0.x, z0.y); ov[2 + (rr >> 1)] = pack_bf2(z1.x, z1.y);                               \
                        av[rr >> 1] = (EPI == 37 || EPI == 69 || g.aux_grad) ? pack_bf2(d0.x, d0.y) : pack_bf2(u0.x, u0.y);         \
                        av[2 + (rr >> 1)] = (EPI == 37 || EPI == 69 || g.aux_grad) ? pack_bf2(d1.x, d1.y) : pack_bf2(u1.x, u1.y);   \
                    } else {                                                                                                        \
                        ov[rr >> 1] = pack_bf2(__uint_as_float(s0[0]), __uint_as_float(s1[0]));                                     \
                        ov[2 + (rr >> 1)] = pack_bf2(__uint_as_float(s0[1]), __uint_as_float(s1[1]));                               \
                    }                                                                                                               \
                }                                                                                                                   \
                if (FFN1) {                                                                                                         \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) { const f2_t zr = f2_bf(ov[e]); t1 += zr; t2 += zr * zr; }        \
                }                                                                                                                   \
                ov2[p2] = ov; if (ACT2) av2[p2] = av;                                                                               \
                SCHED_FENCE();                                                                                                      \
            }                                                                                                                       \
            /* here a lane holds row l15 of the fragment, columns [8 grp, +8) in ov2[0] and [32 + 8 grp, +8) in ov2[1]: stored like that, each 16-lane pass of a store    \
               instruction touches 16 cache lines with 16 bytes each.  Two exchanges of the pair index with a lane bit (4: v_permlane16_swap, then 0: one DPP move + three   \
               selects) turn it into: lane holds fragment row (l15 & 14) + p, 16-byte column chunk 4 l4 + 2 l5 + l0 -- a pass writes 8 lines with 32 contiguous bytes each.   \
               Measured with the data misplaced, i.e. without any exchange code (profiles/r4_gemm_store_shapes.txt): 16 x 16 B 1238 - 1244 TF, 8 x 32 B 1276 - 1292,           \
               4 x 64 B 1286 - 1296, 2 whole lines 1292 - 1303; and with the exchanges that each layout needs: all five 1230 - 1238 (their ~ 60 VALU operations per 16 rows are  \
               not covered by the partner wave's MFMA block), these two: see the same file */                                                                              \
            if (!(ABL & (128 | 64 | 256 | 512 | 1))) {                                                                              \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                     \
                    uint32_t xa = ov2[0][e], xb = ov2[1][e];                                                                        \
                    lane_bit_exchange<4>(xa, xb, lane); lane_bit_exchange<0>(xa, xb, lane);                                         \
                    ov2[0][e] = xa; ov2[1][e] = xb;                                                                                 \
                    if (ACT2) {                                                                                                     \
                        uint32_t ya = av2[0][e], yb = av2[1][e];                                                                    \
                        lane_bit_exchange<4>(ya, yb, lane); lane_bit_exchange<0>(ya, yb, lane);                                     \
                        av2[0][e] = ya; av2[1][e] = yb;                                                                             \
                    }                                                                                                               \
                }                                                                                                                   \
            }                                                                                                                       \
            if (ACT2) {                                                                                                             \
                char* ab = reinterpret_cast<char*>(g.aux) + ((long)((ti0) + wi * 128 + it * 16) * g.ldaux + (tj0) + wj * 64) * 2;   \
                if (ABL & 2048) { K64R_KEEP(av2[0]); K64R_KEEP(av2[1]); }   /* timing only: the second output is not stored at all */ \
                else if (ABL & 1024) {   /* timing only: the second output as ONE byte per element (dense [rows][ldaux] bytes in the first half of the buffer): what an 8-bit gate would store */ \
                    char* ab1 = reinterpret_cast<char*>(g.aux) + ((long)((ti0) + wi * 128 + it * 16) * g.ldaux + (tj0) + wj * 64);  \
                    const k64_u2_t h0 = {av2[0][0], av2[0][1]}, h1 = {av2[1][0], av2[1][1]};                                        \
                    K64R_GSTORE8(avoff_t0 >> 1, h0, ab1); K64R_GSTORE8(avoff_t1 >> 1, h1, ab1);                                     \
                }                                                                                                                   \
                else if (ABL & 128) { K64R_GSTORE16(avoff, av2[0], ab, 0); K64R_GSTORE16(avoff, av2[1], ab, 64); }                  \
                else { K64R_GSTORE16(avoff_t0, av2[0], ab, 0); K64R_GSTORE16(avoff_t1, av2[1], ab, 0); }                            \
            }                                                                                                                       \
            if (ABL & 1) { K64R_KEEP(ov2[0]); K64R_KEEP(ov2[1]); }                                                                  \
            else if (ABL & 64) { K64R_GSTORE16(cvoff_fl0, ov2[0], cb, 0); K64R_GSTORE16(cvoff_fl1, ov2[1], cb, 0); }                \
            else if (ABL & 256) { K64R_GSTORE16(cvoff_p32, ov2[0], cb, 0); K64R_GSTORE16(cvoff_p32b, ov2[1], cb, 0); }              \
            else if (ABL & 512) { K64R_GSTORE16(cvoff_p64, ov2[0], cb, 0); K64R_GSTORE16(cvoff_p64, ov2[1], cb, 64); }              \
            else if (ABL & 4) { K64R_GSTORE16_NT(cvoff_t0, ov2[0], cb, 0); K64R_GSTORE16_NT(cvoff_t1, ov2[1], cb, 0); }             \
            else if (ABL & 128) { K64R_GSTORE16(cvoff, ov2[0], cb, 0); K64R_GSTORE16(cvoff, ov2[1], cb, 64); }                      \
            else { K64R_GSTORE16(cvoff_t0, ov2[0], cb, 0); K64R_GSTORE16(cvoff_t1, ov2[1], cb, 0); }                                \
            SCHED_FENCE();                                                                                                          \
            if (FFN1) {   /* close the row over its four lanes; all four write the same 8 bytes (no exec mask inside the asm store) */ \
                const float r1 = rows4_sum(t1.x + t1.y), r2 = rows4_sum(t2.x + t2.y);                                               \
                char* pb2 = reinterpret_cast<char*>(g.part) + ((long)(((tj0) + wj * 64) >> 6) * g.I + (ti0) + wi * 128 + it * 16) * 8; \
                const k64_u2_t pv2 = {__float_as_uint(r1), __float_as_uint(r2)};                                                    \
                K64R_GSTORE8((uint32_t)(l15 * 8), pv2, pb2);                                                                        \
                SCHED_FENCE();                                                                                                      \
            }                                                                                                                       \
        }                                                                                                                           \
    } while (0)
    // accumulators of row quarter QQ <- residual (row layout -> fragment layout: the lane swap is its own inverse) + bias fragment
#define K64R_INIT(QQ)                                                                                                               \
    do {                                                                                                                            \
        K64R_VMFENCE4(WT::INIT[QQ], rv[QQ]);                                                                                        \
        _Pragma("unroll") for (int ih = 0; ih < 2; ++ih) {                                                                          \
            if (!(ABL & 128)) {   /* the vectors were requested in the store layout (8 lines x 32 B per pass): back to "row l15, chunk grp / 4 + grp" */ \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                     \
                    uint32_t xa = rv[QQ][2 * ih][e], xb = rv[QQ][2 * ih + 1][e];                                                    \
                    lane_bit_exchange<0>(xa, xb, lane); lane_bit_exchange<4>(xa, xb, lane);                                         \
                    rv[QQ][2 * ih][e] = xa; rv[QQ][2 * ih + 1][e] = xb;                                                             \
                }                                                                                                                   \
            }                                                                                                                       \
            _Pragma("unroll") for (int p2 = 0; p2 < 2; ++p2) {                                                                      \
                const u32x4_t w = rv[QQ][2 * ih + p2];                                                                              \
                _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                                  \
                    const uint32_t lo = w[rr >> 1], hi = w[2 + (rr >> 1)];                                                          \
                    const float flo = (rr & 1) ? bf_hi(lo) : bf_lo(lo), fhi = (rr & 1) ? bf_hi(hi) : bf_lo(hi);                     \
                    const k64_u2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(flo), __float_as_uint(fhi), false, false); \
                    acc[2 * (QQ) + ih][2 * p2][rr] = __uint_as_float(sw[0]) + (BIAS ? bf[2 * p2][rr] : 0.f);                        \
                    acc[2 * (QQ) + ih][2 * p2 + 1][rr] = __uint_as_float(sw[1]) + (BIAS ? bf[2 * p2 + 1][rr] : 0.f);                \
                }                                                                                                                   \
                SCHED_FENCE();                                                                                                      \
            }                                                                                                                       \
        }                                                                                                                           \
    } while (0)
    // accumulators of row quarter QQ *= gate (vectors in the store layout -> row layout -> fragment layout, exactly as K64R_INIT takes the residual back)
#define K64R_GATEMUL(QQ)                                                                                                            \
    do {                                                                                                                            \
        K64R_VMFENCE4(WT::INIT[QQ], rv[QQ]);                                                                                        \
        _Pragma("unroll") for (int ih = 0; ih < 2; ++ih) {                                                                          \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                         \
                uint32_t xa = rv[QQ][2 * ih][e], xb = rv[QQ][2 * ih + 1][e];                                                        \
                lane_bit_exchange<0>(xa, xb, lane); lane_bit_exchange<4>(xa, xb, lane);                                             \
                rv[QQ][2 * ih][e] = xa; rv[QQ][2 * ih + 1][e] = xb;                                                                 \
            }                                                                                                                       \
            _Pragma("unroll") for (int p2 = 0; p2 < 2; ++p2) {                                                                      \
                const u32x4_t w = rv[QQ][2 * ih + p2];                                                                              \
                _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                                  \
                    const uint32_t lo = w[rr >> 1], hi = w[2 + (rr >> 1)];                                                          \
                    const float flo = (rr & 1) ? bf_hi(lo) : bf_lo(lo), fhi = (rr & 1) ? bf_hi(hi) : bf_lo(hi);                     \
                    const k64_u2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(flo), __float_as_uint(fhi), false, false); \
                    acc[2 * (QQ) + ih][2 * p2][rr] *= __uint_as_float(sw[0]);                                                       \
                    acc[2 * (QQ) + ih][2 * p2 + 1][rr] *= __uint_as_float(sw[1]);                                                   \
                }                                                                                                                   \
                SCHED_FENCE();                                                                                                      \
            }                                                                                                                       \
        }                                                                                                                           \
    } while (0)
#ifdef ANTMMF_EMULATE
#define K64R_LDS_ADD(OFF, V) (*reinterpret_cast<float*>(smem + (OFF)) += (V))
#else
#define K64R_LDS_ADD(OFF, V) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(smem + (OFF)), (V), 0, 0, false)
#endif
    // column sums of row quarter QQ (after the gate): lane (l15, grp) holds row l15 of fragment it, columns jt 16 + 4 pbg + rr
#define K64R_CSUM(QQ, tj0)                                                                                                          \
    do {                                                                                                                            \
        int ln_ = lane;                                                                                                             \
        K64R_OPAQUE_V(ln_);   /* (the strip offset is rebuilt here, not kept alive across the K loop) */                              \
        const int g_ = ln_ >> 4;                                                                                                    \
        const int cb_ = CSOFF + ((tj0) + wj * 64 + 4 * (((g_ & 1) << 1) | (g_ >> 1))) * 4;                                          \
        _Pragma("unroll") for (int jt = 0; jt < 4; ++jt) {                                                                          \
            _Pragma("unroll") for (int rr = 0; rr < 4; rr += 2) {   /* (two columns at a time: a handful of live temporaries) */     \
                const float c0 = row16_sum(acc[2 * (QQ)][jt][rr] + acc[2 * (QQ) + 1][jt][rr]);                                      \
                const float c1 = row16_sum(acc[2 * (QQ)][jt][rr + 1] + acc[2 * (QQ) + 1][jt][rr + 1]);                              \
                if ((ln_ & 15) == 0) { K64R_LDS_ADD(cb_ + jt * 64 + rr * 4, c0); K64R_LDS_ADD(cb_ + jt * 64 + rr * 4 + 4, c1); }    \
                SCHED_FENCE();                                                                                                      \
            }                                                                                                                       \
        }                                                                                                                           \
    } while (0)
#define K64R_PIECE(IDX)                                                                                                             \
    do {                                                                                                                            \
        int kk = t + (IDX < 4 ? 1 : 2);                                                                                             \
        const bool nx = kk >= nk;                                                                                                   \
        if (nx) kk -= nk;                                                                                                           \
        const char* pbs = nx ? pgn : pgc;                                                                                           \
        const char* qbs = nx ? qgn : qgc;                                                                                           \
        const uint32_t st = IDX < 4 ? (so ^ STAGE) : so;                                                                            \
        if (ABL & 16) {                                                                                                             \
            if (late) { if (IDX < 4) { dma_p(pbs, IDX, kk, st); dma_p(pbs, IDX, kk, st, 1); } else { dma_q(qbs, IDX - 4, kk, st); dma_q(qbs, IDX - 4, kk, st, 1); } } \
        } else if (IDX < 4) dma_p(pbs, IDX, kk, st); else dma_q(qbs, IDX - 4, kk, st);                                              \
    } while (0)
#define K64R_READS(PH)                                                                                                              \
    do {                                                                                                                            \
        if (PH == 0) {                                                                                                              \
            K64_READ(qa[0], so + qbase[0], 0);    K64_READ(qa[1], so + qbase[2], 0);                                                \
            K64_READ(qa[2], so + qbase[1], 2048); K64_READ(qa[3], so + qbase[3], 2048);                                             \
            K64_READ(qa[4], so + qbase[2], 4096); K64_READ(qa[5], so + qbase[0], 4096);                                             \
            K64_READ(qa[6], so + qbase[3], 6144); K64_READ(qa[7], so + qbase[1], 6144);                                             \
        }                                                                                                                           \
        K64_READ(pb[0], so + pbase[((PH & 1) * 2 + 0)], PH * 4096);                                                                 \
        K64_READ(pb[1], so + pbase[((PH & 1) * 2 + 0) ^ 2], PH * 4096);                                                             \
        K64_READ(pb[2], so + pbase[((PH & 1) * 2 + 1)], PH * 4096 + 2048);                                                          \
        K64_READ(pb[3], so + pbase[((PH & 1) * 2 + 1) ^ 2], PH * 4096 + 2048);                                                      \
    } while (0)
    // CMODE: what the first MFMA of an accumulator starts from: 0 = the accumulator (steady state, or initialised by K64R_INIT), 1 = the bias fragment, 2 = zero
#define K64R_MFMA(MPH, CMODE)                                                                                                       \
    do {                                                                                                                            \
        if (MPH == 0) K64_FENCE8(qa);                                                                                               \
        K64_FENCE4(pb);                                                                                                             \
        SCHED_FENCE();                                                                                                              \
        K64_SETPRIO(1);                                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                            \
            _Pragma("unroll") for (int f = 0; f < 2; ++f)                                                                           \
                _Pragma("unroll") for (int jt = 0; jt < 4; ++jt)                                                                    \
                    acc[2 * MPH + f][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[2 * jt + ks], pb[2 * f + ks],                 \
                        ((CMODE) == 0 || ks == 1) ? acc[2 * MPH + f][jt] : ((CMODE) == 1 ? bf[jt] : (f32x4_t){0.f, 0.f, 0.f, 0.f}), 0, 0, 0); \
        K64_SETPRIO(0);                                                                                                             \
        SCHED_FENCE();                                                                                                              \
    } while (0)
#define K64R_SPREAD_BLOCK(IT, P2) do { if (!(ABL & 32)) K64R_SPREAD_BLOCK1(IT, P2, 0); else if (!late) { K64R_SPREAD_BLOCK1(IT, P2, 0); K64R_SPREAD_BLOCK1(((IT + 4) & 7), P2, 128); } } while (0)
#define K64R_SPREAD_BLOCK1(IT, P2, ROWOFF)                                                                                          \
    do {                                                                                                                            \
        u32x4_t ov;                                                                                                                 \
        _Pragma("unroll") for (int rr = 0; rr < 4; rr += 2) {                                                                       \
            const k64_u2_t s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[IT][2 * P2][rr]), __float_as_uint(acc[IT][2 * P2 + 1][rr]), false, false); \
            const k64_u2_t s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[IT][2 * P2][rr + 1]), __float_as_uint(acc[IT][2 * P2 + 1][rr + 1]), false, false); \
            ov[rr >> 1] = pack_bf2(__uint_as_float(s0[0]), __uint_as_float(s1[0]));                                                 \
            ov[2 + (rr >> 1)] = pack_bf2(__uint_as_float(s0[1]), __uint_as_float(s1[1]));                                           \
        }                                                                                                                           \
        char* cb = reinterpret_cast<char*>(g.C) + ((long)(i0 + wi * 128 + ((IT - (ROWOFF ? 4 : 0)) & 7) * 16) * g.ldc + j0 + wj * 64 + P2 * 32) * 2; \
        K64R_GSTORE16((ROWOFF ? cvoff128 : cvoff), ov, cb, 0);                                                                                            \
        SCHED_FENCE();                                                                                                              \
    } while (0)
#define K64R_SPREAD_STORE(ROLE)                                                                                                     \
    do {   /* (every accumulator is read by some K-tile's store: none of the MFMAs is dead code) */                                  \
        if (ROLE == 0) K64R_SPREAD_BLOCK(7, 1);                                                                                     \
        else if (t == 1) K64R_SPREAD_BLOCK(0, 0); else if (t == 2) K64R_SPREAD_BLOCK(0, 1); else if (t == 3) K64R_SPREAD_BLOCK(1, 0);   \
        else if (t == 4) K64R_SPREAD_BLOCK(1, 1); else if (t == 5) K64R_SPREAD_BLOCK(2, 0); else if (t == 6) K64R_SPREAD_BLOCK(2, 1);   \
        else if (t == 7) K64R_SPREAD_BLOCK(3, 0); else if (t == 8) K64R_SPREAD_BLOCK(3, 1); else if (t == 9) K64R_SPREAD_BLOCK(4, 0);   \
        else if (t == 10) K64R_SPREAD_BLOCK(4, 1); else if (t == 11) K64R_SPREAD_BLOCK(5, 0); else if (t == 12) K64R_SPREAD_BLOCK(5, 1); \
        else if (t == 13) K64R_SPREAD_BLOCK(6, 0); else if (t == 14) K64R_SPREAD_BLOCK(6, 1); else K64R_SPREAD_BLOCK(7, 0);           \
    } while (0)
    // ROLE: 0 = first K-tile of the output tile, 1 = steady, 2 = last K-tile
#define K64R_HOOK(ROLE, PH)                                                                                                         \
    do {                                                                                                                            \
        if (ROLE == 0 && PH == 0) {                                                                                                 \
            if (pending) { if (GATE) K64R_GATEMUL(3); if (CSUM) K64R_CSUM(3, ej0); K64R_EPIQ(3, ei0, ej0); }                        \
            if (RES) K64R_RESLOAD(3, i0, j0);                                                                                       \
            if (BIAS) {                                                                                                             \
                if (!RES) K64R_VMWAIT(WT::BIASW);   /* with a residual the wait in front of K64R_INIT(0) is the stronger one */      \
                else K64R_VMFENCE4(WT::INIT[0], rv[0]);                                                                             \
                K64R_LREAD16(bf[0], BIASOFF + wave * 256 + pbg * 16, 0);   K64R_LREAD16(bf[1], BIASOFF + wave * 256 + pbg * 16, 64);  \
                K64R_LREAD16(bf[2], BIASOFF + wave * 256 + pbg * 16, 128); K64R_LREAD16(bf[3], BIASOFF + wave * 256 + pbg * 16, 192); \
                K64_FENCE4(bf);                                                                                                     \
            }                                                                                                                       \
        }                                                                                                                           \
        if (ROLE == 0 && RES) K64R_INIT(PH);                                                                                        \
        if (ROLE == 2 && PH == 0 && BIAS) dma_bias(nj0);                                                                            \
        if (GATE && ROLE == 2 && PH == 0) K64R_RESLOAD(0, i0, j0);                                                                  \
        if (ROLE == 2 && PH >= 1) {                                                                                                 \
            if (GATE) K64R_GATEMUL(PH - 1);                                                                                         \
            if (CSUM) K64R_CSUM(PH - 1, j0);                                                                                        \
            K64R_EPIQ(PH - 1, i0, j0);                                                                                              \
            if (RES) K64R_RESLOAD(PH - 1, ni0, nj0);                                                                                \
            if (GATE) K64R_RESLOAD(PH, i0, j0);                                                                                     \
        }                                                                                                                           \
        if ((ABL & 8) && PH == 1) K64R_SPREAD_STORE(ROLE);                                                                              \
    } while (0)
#define K64R_PHASE(ROLE, PH)                                                                                                        \
    do {                                                                                                                            \
        K64R_READS(PH);                                                                                                             \
        K64R_PIECE(2 * PH);                                                                                                         \
        K64R_PIECE(2 * PH + 1);                                                                                                     \
        SCHED_FENCE();                                                                                                              \
        K64R_HOOK(ROLE, PH);                                                                                                        \
        SCHED_FENCE();                                                                                                              \
        if (!late) { glds_wait_le<((ABL & 16) ? 63 : (ABL & 32) ? K64RTwoStoreWaits::W[PH] : (ABL & 8) ? K64RSpreadWaits::W[PH] : WT::W[ROLE][PH])>(); K64_BARRIER(); }   \
        K64R_MFMA(PH, ((ROLE != 0 || RES) ? 0 : (BIAS ? 1 : 2)));                                                                  \
        if (late) { glds_wait_le<((ABL & 48) == 48 ? K64RSplitWaits::W[PH] : (ABL & 16) ? K64RProxyWaits::W[PH] : (ABL & 32) ? WT::W[1][PH] : (ABL & 8) ? K64RSpreadWaits::W[PH] : WT::W[ROLE][PH])>(); K64_BARRIER(); }    \
    } while (0)
#define K64R_TILE(ROLE) do { K64R_PHASE(ROLE, 0); K64R_PHASE(ROLE, 1); K64R_PHASE(ROLE, 2); K64R_PHASE(ROLE, 3); so ^= STAGE; } while (0)

    for (;;) {
        const bool more = local + per_xcd < xcount_main;
        if (more) { tile_origin(local + per_xcd, ni0, nj0); tile_bases(ni0, nj0, pgn, qgn); }
        else { ni0 = i0; nj0 = j0; pgn = pgc; qgn = qgc; }   // the look-ahead of the last tile re-reads the tile's own operands (never consumed)
        // (t stays a run-time value: with literal K-tile numbers hipcc precomputes a 64-bit DMA address vector per (role, piece) and spills them)
        t = 0; K64R_OPAQUE(t); K64R_TILE(0);
        if (ABL & 8) {
            for (++t; t < nk; ++t) K64R_TILE(1);
        } else {
        for (++t; t < nk - 1; ++t) K64R_TILE(1);
        K64R_OPAQUE(t);
        K64R_TILE(2);
        }
        pending = !(ABL & 8); ei0 = i0; ej0 = j0;
        if (!more) break;
        local += per_xcd;
        i0 = ni0; j0 = nj0; pgc = pgn; qgc = qgn;
    }
    // the wrapped look-ahead of the last tile must land before its targets are reused: the residual vectors' registers (the fences keep them allocated
    // up to here) and, for the DMA pieces, this LDS
    if (RES) { K64R_VMFENCE4(0, rv[0]); K64R_VMFENCE4(0, rv[1]); K64R_VMFENCE4(0, rv[2]); }
    glds_wait_all();
    SCHED_FENCE();
    if (GATE) K64R_GATEMUL(3);   // (its wait allows 2 younger operations: none are in flight any more)
    if (CSUM) K64R_CSUM(3, ej0);
    if (!(ABL & 8)) K64R_EPIQ(3, ei0, ej0);
    if (CSUM) {   // this workgroup's column sums -> row blockIdx.x of g.part ([gridDim.x][J] fp32; workgroups without a tile returned above: the host zero-fills)
        wg_barrier_lds_only();
        for (int c = threadIdx.x; c < g.J; c += 512) g.part[(long)blockIdx.x * g.J + c] = *reinterpret_cast<const float*>(smem + CSOFF + c * 4);
    }
    if (tail_r > 0 && lx < 16 * tail_r) {
        // ---- leftover tiles as cells.  Cell c of the chunk = tile c >> 4, phase (c >> 2) & 3, Q fragment c & 3: per wave 2 P fragments (rows wi 128 + 32 ph + [0, 32)) x 1 Q
        // fragment (columns wj 64 + 16 jt + [0, 16)), i.e. 4 MFMAs per K-tile; operands per K-tile: the P quarter `ph` and the Q quarter `jt` of the tile's stage image
        // (one 8-row piece each per wave), parked in ring slot t & 7 = (stage (t >> 2) & 1, quarter region t & 3) -- the fragment read offsets of the walk apply as they are.
        constexpr int DEPTH = 6;   // K-tiles in flight; slot (t + DEPTH) & 7 was last read in iteration t - 2: every wave is past barrier(t - 1), i.e. done with it
        wg_barrier_lds_only();     // every wave is out of the walk's last stage reads (its DMA pieces were drained above)
        for (int c = lx; c < 16 * tail_r; c += per_xcd) {
            const int u = c >> 4, cph = (c >> 2) & 3, cjt = c & 3;
            tile_origin(xcount_main + u, i0, j0);
            tile_bases(i0, j0, pgc, qgc);
            auto cell_issue = [&](int kt) {
                const int sq = kt & 3;
                const uint32_t st = (uint32_t)((kt >> 2) & 1) * STAGE;
                glds16(pgc + (long)cph * 32 * ldpb + (long)kt * 128 + pv[sq & 1], smem + st + (prow0 + 32 * sq) * 128);
                glds16(qgc + (long)cjt * 16 * ldqb + (long)kt * 128 + qv[sq], smem + st + QOFF + (qrow0 + 16 * sq) * 128);
            };
            // (an L2 warm-up of the cell's operand lines in front of this loop -- every line requested once, 4 bytes each, all in flight -- was built and measured: no gain,
            // profiles/r5_gemm_tail_cells_l2_warmup_ab.jsonl.  The cells are not latency-bound; what the round count promised was never there: the lone 17th round of the
            // walk runs on an otherwise idle chip and costs far less than a full round)
            // accumulators start from bias + residual, as in the walk (same fp32 sum, same order -> same bits)
            f32x4_t cacc[2];
            const int ccol = j0 + wj * 64 + cjt * 16 + 4 * pbg;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const long crow = i0 + wi * 128 + (2 * cph + f) * 16 + l15;
                f32x4_t v = {0.f, 0.f, 0.f, 0.f};
                if (RES) {
                    const k64_u2_t rr = *reinterpret_cast<const k64_u2_t*>(g.residual + crow * g.ldr + ccol);
                    v = (f32x4_t){bf_lo(rr[0]), bf_hi(rr[0]), bf_lo(rr[1]), bf_hi(rr[1])};
                }
                if (BIAS) { const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(g.bias + ccol); v = RES ? v + b4 : b4; }
                cacc[f] = v;
            }
#pragma unroll
            for (int kt = 0; kt < DEPTH; ++kt) if (kt < nk) cell_issue(kt);
            bf16x8_t cp[4], cq[2];
#define K64R_CELL_ITER(SLOT)                                                                                                        \
            if (t0 + (SLOT) < nk) {                                                                                                 \
                const int kt = t0 + (SLOT);                                                                                         \
                if (kt + DEPTH < nk) { cell_issue(kt + DEPTH); glds_wait_le<2 * DEPTH>(); } else glds_wait_all();                   \
                K64_BARRIER();                                                                                                      \
                constexpr uint32_t cst = (uint32_t)(((SLOT) >> 2) & 1) * STAGE;                                                     \
                constexpr int csq = (SLOT) & 3, k0 = 2 * csq, k1 = 2 * csq + 1;                                                     \
                K64_READ(cq[0], cst + qbase[csq], csq * 2048);           K64_READ(cq[1], cst + qbase[csq ^ 2], csq * 2048);         \
                K64_READ(cp[0], cst + pbase[k0 & 3], k0 * 2048);         K64_READ(cp[1], cst + pbase[(k0 & 3) ^ 2], k0 * 2048);     \
                K64_READ(cp[2], cst + pbase[k1 & 3], k1 * 2048);         K64_READ(cp[3], cst + pbase[(k1 & 3) ^ 2], k1 * 2048);     \
                K64_FENCE4(cp);                                                                                                     \
                K64_TIE2(cq);                                                                                                       \
                SCHED_FENCE();                                                                                                      \
                cacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cq[0], cp[0], cacc[0], 0, 0, 0);                                  \
                cacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cq[0], cp[2], cacc[1], 0, 0, 0);                                  \
                cacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cq[1], cp[1], cacc[0], 0, 0, 0);                                  \
                cacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cq[1], cp[3], cacc[1], 0, 0, 0);                                  \
                SCHED_FENCE();                                                                                                      \
            }
            for (int t0 = 0; t0 < nk; t0 += 8) {
                K64R_CELL_ITER(0) K64R_CELL_ITER(1) K64R_CELL_ITER(2) K64R_CELL_ITER(3) K64R_CELL_ITER(4) K64R_CELL_ITER(5) K64R_CELL_ITER(6) K64R_CELL_ITER(7)
            }
#undef K64R_CELL_ITER
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const long crow = i0 + wi * 128 + (2 * cph + f) * 16 + l15;
                const k64_u2_t ov = {pack_bf2(cacc[f][0], cacc[f][1]), pack_bf2(cacc[f][2], cacc[f][3])};
                *reinterpret_cast<k64_u2_t*>(reinterpret_cast<bf16_t*>(g.C) + crow * g.ldc + ccol) = ov;
            }
            K64_BARRIER();   // the next cell's prologue refills slots this cell's last K-tiles were read from
        }
    }
#ifndef ANTMMF_EMULATE
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_k64_clk[0] = __builtin_readcyclecounter() - clk0; g_k64_clk[1] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
#undef K64R_TILE
#undef K64R_PHASE
#undef K64R_HOOK
#undef K64R_SPREAD_STORE
#undef K64R_SPREAD_BLOCK
#undef K64R_SPREAD_BLOCK1
#undef K64R_MFMA
#undef K64R_READS
#undef K64R_PIECE
#undef K64R_INIT
#undef K64R_GATEMUL
#undef K64R_EPIQ
#undef K64R_RESLOAD
}

// fp32 partial tile of a token split -> workspace, staged through the wave's LDS region in two halves so that every store
// instruction writes whole 256-B row segments (fp32 atomics straight from the fragment layout measured ~90 G adds / s:
// 0.37 ms for the 33 M adds of one fc1 wgrad, as long as its whole K loop).
template <int TI, int TJ>
__device__ __forceinline__ void store_partial_f32_staged(float* __restrict__ dst, long ld, f32x4_t (&acc)[TI][TJ], int wi, int wj, int lane, char* wave_lds) {
    const int l15 = lane & 15, grp = lane >> 4;
    constexpr int ROWB = TJ * 64, SLOTS = ROWB / 16, HALF = TI / 2;  // 256-B rows, 16 slots
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        WAVE_LDS_ORDER();
#pragma unroll
        for (int it = 0; it < HALF; ++it) {
            const int row = it * 16 + l15;
#pragma unroll
            for (int jt = 0; jt < TJ; ++jt) {
                const int slot = jt * 4 + grp;
                const f32x4_t v = acc[h * HALF + it][jt];
                *reinterpret_cast<float4*>(wave_lds + row * ROWB + ((slot ^ (row & (SLOTS - 1))) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        WAVE_LDS_ORDER();
#pragma unroll
        for (int pass = 0; pass < (HALF * 16 * SLOTS) / 64; ++pass) {
            const int row = pass * (64 / SLOTS) + lane / SLOTS, ls = lane % SLOTS;
            const float4 val = *reinterpret_cast<const float4*>(wave_lds + row * ROWB + ((ls ^ (row & (SLOTS - 1))) << 4));
            *reinterpret_cast<float4*>(dst + (long)(wi * (16 * TI) + h * HALF * 16 + row) * ld + wj * (16 * TJ) + ls * 4) = val;
        }
    }
}

// out[i][j] (+)= sum_z ws[z][i][j]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int splits, int I, int J, long ldc, int accumulate) {
    const long nvec = (long)I * J / 4, plane = (long)I * J;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        const long e = v * 4;
        const int i = (int)(e / J), j = (int)(e % J);
        float4 s = *reinterpret_cast<const float4*>(ws + e);
        for (int z = 1; z < splits; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(ws + z * plane + e);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float4* o = reinterpret_cast<float4*>(out + (long)i * ldc + j);
        if (accumulate) { const float4 c = *o; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
        *o = s;
    }
}

// The same reduction with the output rows cut into up to four equal SEGMENTS that live at unrelated addresses (antmmf_gemm_wgrad_bf16_seg: the q / k / v projection
// weights of a layer are separate parameters of the gradient arena, their wgrad is ONE token-major GEMM over the packed dQKV)
struct SegDst { float* p[4]; int rows; };
static thread_local const SegDst* tl_seg = nullptr;   // set by the _seg entry around its gemm_impl call: host-side dispatch state of that call only
#define ANTMMF_ESEG (-77)                              /* private: "this shape does not take the workspace path": the _seg entry then runs one call per segment */
__global__ __launch_bounds__(256) void splitk_reduce_seg_kernel(const float* __restrict__ ws, const SegDst d, int splits, int I, int J, long ldc) {
    const long nvec = (long)I * J / 4, plane = (long)I * J;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        const long e = v * 4;
        const int i = (int)(e / J), j = (int)(e % J);
        float4 s = *reinterpret_cast<const float4*>(ws + e);
        for (int z = 1; z < splits; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(ws + z * plane + e);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const int sg = i / d.rows;
        float4* o = reinterpret_cast<float4*>(d.p[sg] + (long)(i - sg * d.rows) * ldc + j);
        const float4 c = *o;
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        *o = s;
    }
}

// ---- 4-stage LDS-DMA ring for wgrad (all-r-major): 256 x 256 output tile, 32 tokens per stage, natural [r][cols] LDS
// image (512-B rows), fragments by ds_read_b64_tr_b16, split over the token range by gridDim.z (fp32 atomics when split).
// STAGGER = true (product): the two-group schedule of gemm_nt_ring_kernel; false: one barrier per K-step (kept for A/B runs,
// ANTMMF_GEMM_RASTER bit 4).  Measured on the ViT-L/14 wgrad shapes: 898 / 1014 / 921 vs 878 / 978 / 902 TFLOP/s.
template <bool STAGGER>
__global__ __launch_bounds__(512) void gemm_tn_ring_kernel(const GemmArgs g) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int STAGES = 4, BM = 256, BN = 256, TI = 8, TJ = 4, NWJ = 4, G = 4, ROWB = 512;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave / NWJ, wj = wave % NWJ;
    const int l15 = lane & 15, grp = lane >> 4;
    // grid.x = tiles x token-splits, linearised; XCD x (workgroups b % 8 == x) gets a CONTIGUOUS range of (split, tile) ids, so
    // the ~32 workgroups resident on one XCD share one token range and an 8 x 4 patch of output tiles through its L2
    // (measured: L2 hit rate 36 % with the dispatch-order mapping, FETCH_SIZE 3x the algorithmic bytes)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tiles_j = g.J / BN, tiles = (g.I / BM) * tiles_j;
    const int split = wgid / tiles, tile = wgid - split * tiles;
    const int i0 = (tile / tiles_j) * BM, j0 = (tile % tiles_j) * BN;
    const int nk_total = g.R >> 5;
    const int kbeg = split * g.ksteps_per_split;
    int kend = kbeg + g.ksteps_per_split;
    if (kend > nk_total) kend = nk_total;
    if (kbeg >= kend) return;
    const int nk = kend - kbeg;

    f32x4_t acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // DMA: a 1-KiB piece = 2 rows x 512 B; wave w moves rows 4w .. 4w+3 of each operand's 32-row stage
    const bf16_t* psrc[2];
    const bf16_t* qsrc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int lin = (wave * 2 + q) * 64 + lane;  // 16-B slot index in the [32][512 B] tile
        const int row = lin >> 5, sidx = lin & 31;
        const int col = (((sidx >> 1) ^ ftr(row)) << 4) + ((sidx & 1) << 3);
        psrc[q] = g.P + (long)row * g.ldp + i0 + col;
        qsrc[q] = g.Q + (long)row * g.ldq + j0 + col;
    }
    auto issue = [&](int t) {  // t: step index relative to kbeg
        char* buf = smem + (t % STAGES) * 32768;
        const long roff = (long)(kbeg + t) << 5;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            glds16(psrc[q] + roff * g.ldp, buf + (wave * 2 + q) * 1024);
            glds16(qsrc[q] + roff * g.ldq, buf + 16384 + (wave * 2 + q) * 1024);
        }
    };
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nk) issue(t);

    if (STAGGER) {  // two-group schedule (see gemm_nt_ring_kernel)
        int issued = (STAGES - 1 < nk ? STAGES - 1 : nk) - 1;
        auto wait_tile = [&](int kt) {
            const int ahead = issued - kt;
            if (ahead >= 2) glds_wait_le<2 * G>();
            else if (ahead == 1) glds_wait_le<G>();
            else glds_wait_le<0>();
        };
        const bool late = wave >= 4;
        if (late) { wait_tile(0); wg_barrier_lds_only(); }
        for (int kt = 0; kt < nk; ++kt) {
            if (!late) wait_tile(kt);
            wg_barrier_lds_only();
            if (kt + STAGES - 1 < nk) { issue(kt + STAGES - 1); issued = kt + STAGES - 1; }
            const char* ps = smem + (kt % STAGES) * 32768;
            const char* qs = ps + 16384;
            bf16x8_t qa[TJ], pb[TI];
#pragma unroll
            for (int t = 0; t < TJ; ++t) qa[t] = frag_tr_raw<ROWB>(qs, 8 * grp, wj * TJ + t, l15);
#pragma unroll
            for (int t = 0; t < TI; ++t) pb[t] = frag_tr_raw<ROWB>(ps, 8 * grp, wi * TI + t, l15);
            if (late && kt + 1 < nk) wait_tile(kt + 1);
            lds_tr_fence(qa, pb);
            wg_barrier_lds_only();
            SCHED_FENCE();
#pragma unroll
            for (int it = 0; it < TI; ++it)
#pragma unroll
                for (int jt = 0; jt < TJ; ++jt)
                    acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[jt], pb[it], acc[it][jt], 0, 0, 0);
            SCHED_FENCE();
        }
        if (!late) wg_barrier_lds_only();
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt;
        if (ahead >= STAGES - 2) glds_wait_le<(STAGES - 2) * G>();
        else if (ahead == 1) glds_wait_le<G>();
        else glds_wait_le<0>();
        wg_barrier_lds_only();
        if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1);
        const char* ps = smem + (kt % STAGES) * 32768;
        const char* qs = ps + 16384;
        bf16x8_t qa[TJ], pb[TI];
#pragma unroll
        for (int t = 0; t < TJ; ++t)
            qa[t] = frag_tr_raw<ROWB>(qs, 8 * grp, wj * TJ + t, l15);
#pragma unroll
        for (int t = 0; t < TI; ++t)
            pb[t] = frag_tr_raw<ROWB>(ps, 8 * grp, wi * TI + t, l15);
        lds_tr_fence(qa, pb);
        SCHED_FENCE();
#pragma unroll
        for (int it = 0; it < TI; ++it)
#pragma unroll
            for (int jt = 0; jt < TJ; ++jt)
                acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[jt], pb[it], acc[it][jt], 0, 0, 0);
        SCHED_FENCE();
    }
#ifdef ANTMMF_LAB
    if (g.raster & 8) {  // lab experiment: no store tail
        float sacc = 0.f;
#pragma unroll
        for (int a2 = 0; a2 < TI; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < TJ; ++b2) sacc += acc[a2][b2][0] + acc[a2][b2][1] + acc[a2][b2][2] + acc[a2][b2][3];
        if (sacc == 123.456f) reinterpret_cast<float*>(g.C)[threadIdx.x] = sacc;
        return;
    }
#endif
    if (g.ws) {
        wg_barrier_lds_only();  // stage buffers are free
        store_partial_f32_staged<TI, TJ>(g.ws + (long)split * g.I * g.J + (long)i0 * g.J + j0, g.J, acc, wi, wj, lane, smem + wave * 16384);
        return;
    }
    gemm_epilogue<TI, TJ>(g, acc, i0, j0, wi, wj, l15, grp, g.ksteps_per_split < nk_total);
}

// ---- gemm_tn_k64_kernel: wgrad dW += dY^T X with the schedule of gemm_nt_k64p_kernel ---------------------------------------------------
// K-tiles of 64 tokens, two 64-KiB stages ([64 tokens][256 cols] per operand, natural token-major image, 512-B rows, the 32-B chunk swizzle
// ftr of the ring kernel), fragments by ds_read_b64_tr_b16.  Four phases of 16 MFMAs per K-tile: phase (ks, ih) = 32-token half ks x
// output rows [64 ih, +64) of the wave (4 P fragments, read in that phase) x all four Q fragments of that half (read in (ks, 0)):
// 16 / 8 / 16 / 8 transposing reads.  One barrier per phase, same L M stream as the NT kernel (group A barrier behind L, group B behind M).
// Ring units are the 32-token halves (16 P + 16 Q pieces of 2 rows x 512 B, 4 per wave): half 0 of a stage is read in ph0, ph1 and
// refilled in ph3 (P pieces) and ph0' (Q pieces) with the data two K-tiles ahead; half 1 is read in ph2, ph3 and refilled in ph1', ph2'
// (>= 2 periods after the last read period, needed >= 4 periods later).  Waits: only before the phases that open a new half (ph1 -> ph2,
// ph3 -> ph0'): vmcnt(6) in steady state.  The token range is split over gridDim (fp32 partial tiles to the workspace, as before).
template <int FLAGS>
__global__ __launch_bounds__(512) void gemm_tn_k64_kernel(const GemmArgs g) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int BM = 256, BN = 256, TI = 8, TJ = 4, NWJ = 4, ROWB = 512, STAGE = 65536, QOFF = 32768;
    constexpr bool PRIO = FLAGS & K64F_PRIO;
    const int lane = threadIdx.x & 63;
#ifdef ANTMMF_EMULATE
    const int wave = threadIdx.x >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#endif
    const int wi = wave / NWJ, wj = wave % NWJ;
    const int l15 = lane & 15, grp = lane >> 4;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tiles_j = g.J / BN, tiles = (g.I / BM) * tiles_j;
    const int split = wgid / tiles, tile = wgid - split * tiles;
    const int i0 = (tile / tiles_j) * BM, j0 = (tile % tiles_j) * BN;
    const int nk_total = g.R >> 6;                  // 64-token K-tiles
    const int kbeg = split * g.ksteps_per_split;    // ksteps_per_split counts 64-token K-tiles here
    int kend = kbeg + g.ksteps_per_split;
    if (kend > nk_total) kend = nk_total;
    if (kbeg >= kend) return;
    const int nk = kend - kbeg;                     // >= 2 (host)

    f32x4_t acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // DMA: piece = 2 token rows x 512 B; half h of a stage = rows [32 h, +32) = pieces 16 h .. 16 h + 15 of each operand, wave w moves
    // pieces 16 h + 2 w, + 1.  Per-lane source offsets (bytes) inside a piece pair are loop-invariant.
    const char* pgb = reinterpret_cast<const char*>(g.P + i0) + (long)kbeg * 64 * g.ldp * 2;   // token row kbeg * 64 of this split
    const char* qgb = reinterpret_cast<const char*>(g.Q + j0) + (long)kbeg * 64 * g.ldq * 2;
    uint32_t pvo[2], qvo[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int lin = (wave * 2 + q) * 64 + lane;  // 16-B slot index within the half's [32][512 B] image
        const int row = lin >> 5, sidx = lin & 31;
        const int col = (((sidx >> 1) ^ ftr(row)) << 4) + ((sidx & 1) << 3);   // ftr(row + 32 h) == ftr(row): bits 0, 1, 3 only
        pvo[q] = (uint32_t)(row * (int)(g.ldp * 2) + col * 2);
        qvo[q] = (uint32_t)(row * (int)(g.ldq * 2) + col * 2);
    }
    auto dma_p = [&](int h, int kk, uint32_t st) {
        const char* b = pgb + ((long)kk * 64 + 32 * h) * g.ldp * 2;
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(b + pvo[q], smem + st + (16 * h + wave * 2 + q) * 1024);
    };
    auto dma_q = [&](int h, int kk, uint32_t st) {
        const char* b = qgb + ((long)kk * 64 + 32 * h) * g.ldq * 2;
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(b + qvo[q], smem + st + QOFF + (16 * h + wave * 2 + q) * 1024);
    };
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int h = 0; h < 2; ++h) { dma_p(h, tt, tt * STAGE); dma_q(h, tt, tt * STAGE); }
    glds_wait_all();
    wg_barrier_lds_only();

    bf16x8_t qa[4], pb[4];
    const bool late = wave >= 4;
    uint32_t so = 0;
    bool first = true;   // first K-tile: K-tile 1 is complete, the refills that target it are skipped
    // WT: 0 steady, 1 = K-tile nk - 2, 2 = K-tile nk - 1
#define TK_PIECES(PH, WT)                                                                                                           \
    do {                                                                                                                            \
        if (WT == 0 || (WT == 1 && PH < 3)) {                                                                                       \
            const int kk = t + (PH < 3 ? 1 : 2);                                                                                    \
            const uint32_t st = PH < 3 ? (so ^ STAGE) : so;                                                                         \
            if (PH == 0) { if (!first) dma_q(0, kk, st); }                                                                          \
            else if (PH == 1) { if (!first) dma_p(1, kk, st); }                                                                     \
            else if (PH == 2) { if (!first) dma_q(1, kk, st); first = false; }                                                      \
            else dma_p(0, kk, st);                                                                                                  \
        }                                                                                                                           \
    } while (0)
#define TK_WAIT(PH, WT)                                                                                                             \
    do {                                                                                                                            \
        if (PH == 1) { if (WT == 2) glds_wait_le<0>(); else glds_wait_le<6>(); }                                                    \
        else if (PH == 3) { if (WT == 0) glds_wait_le<6>(); else if (WT == 1) glds_wait_le<4>(); else glds_wait_le<0>(); }          \
    } while (0)
#define TK_PHASE(PH, WT)                                                                                                            \
    do {                                                                                                                            \
        const char* ps = smem + so;                                                                                                 \
        const char* qs = ps + QOFF;                                                                                                 \
        if ((PH & 1) == 0) {                                                                                                        \
            _Pragma("unroll") for (int jt = 0; jt < TJ; ++jt) qa[jt] = frag_tr_raw<ROWB>(qs, 32 * (PH >> 1) + 8 * grp, wj * TJ + jt, l15); \
        }                                                                                                                           \
        _Pragma("unroll") for (int f = 0; f < 4; ++f) pb[f] = frag_tr_raw<ROWB>(ps, 32 * (PH >> 1) + 8 * grp, wi * TI + 4 * (PH & 1) + f, l15); \
        TK_PIECES(PH, WT);                                                                                                          \
        if (!late) { TK_WAIT(PH, WT); K64_BARRIER(); }                                                                              \
        if ((PH & 1) == 0) K64_FENCE4(qa);                                                                                          \
        K64_FENCE4(pb);                                                                                                             \
        SCHED_FENCE();                                                                                                              \
        if (PRIO) K64_SETPRIO(1);                                                                                                   \
        _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                                               \
            _Pragma("unroll") for (int jt = 0; jt < TJ; ++jt)                                                                       \
                acc[4 * (PH & 1) + f][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[jt], pb[f], acc[4 * (PH & 1) + f][jt], 0, 0, 0); \
        if (PRIO) K64_SETPRIO(0);                                                                                                   \
        SCHED_FENCE();                                                                                                              \
        if (late) { TK_WAIT(PH, WT); K64_BARRIER(); }                                                                               \
    } while (0)
#define TK_TILE(WT) do { TK_PHASE(0, WT); TK_PHASE(1, WT); TK_PHASE(2, WT); TK_PHASE(3, WT); so ^= STAGE; } while (0)
    int t = 0;
    for (; t < nk - 2; ++t) TK_TILE(0);
    TK_TILE(1);
    ++t;
    TK_TILE(2);
#undef TK_TILE
#undef TK_PHASE
#undef TK_WAIT
#undef TK_PIECES
    wg_barrier_lds_only();  // every wave is past its last MFMA block and nothing is in flight: the stages are free
    if (g.ws) {
        store_partial_f32_staged<TI, TJ>(g.ws + (long)split * g.I * g.J + (long)i0 * g.J + j0, g.J, acc, wi, wj, lane, smem + wave * 16384);
        return;
    }
    gemm_epilogue<TI, TJ>(g, acc, i0, j0, wi, wj, l15, grp, nk < nk_total);
}

// LAB build only (`make lab`, tests/emu): the A/B knob of tools/gemm_bench and the kernel tests, initialised from ANTMMF_GEMM_VARIANT: 0 = the BK = 32 ring kernels of
// round 1; bit 2 (default) routes the large 256-aligned all-r-contiguous GEMMs to the BK = 64 persistent kernels; further bits select A/B forms (see the launch site).
// The PRODUCT library has neither the variable nor the setter: g_gemm_variant is the constant 4 and every branch on its other bits is compiled out.
#ifdef ANTMMF_LAB
static int g_gemm_variant = -1;
extern "C" int antmmf_debug_set_gemm_variant(int v) { g_gemm_variant = v; return ANTMMF_OK; }
#define GEMM_VARIANT_INIT() do { if (g_gemm_variant < 0) { const char* ve = getenv("ANTMMF_GEMM_VARIANT"); g_gemm_variant = ve ? atoi(ve) : 4; } } while (0)
#define LAB_ONLY(...) __VA_ARGS__
static long g_k64_cell_launches = 0;   // launches whose tail round ran as cells (tests: "the cells really ran")
extern "C" long antmmf_debug_gemm_cell_launches() { return g_k64_cell_launches; }
#else
static constexpr int g_gemm_variant = 4;
#define GEMM_VARIANT_INIT() do {} while (0)
#define LAB_ONLY(...)
#endif
static long g_k64_launches = 0;   // launch counter read by the tests (which kernel family served a call); not dispatch state
extern "C" long antmmf_debug_gemm_k64_launches() { return g_k64_launches; }

static thread_local float* tl_colsum_part = nullptr;   // set around gemm_impl by antmmf_gemm_bf16_gated_colsum only
// C ABI: see include/antmmf_hip.h for the contract.
static int gemm_impl(const void* P, const void* Q, void* C, int I, int J, int R, long ldp, long ldq, long ldc,
                     int p_rmajor, int q_rmajor, int c_dtype, float alpha, const float* bias, int act,
                     const void* residual, long ldr, void* aux, long ldaux, const void* gate, long ldgate,
                     int accumulate, int split_k, float* workspace, long workspace_bytes, hipStream_t stream) {
    if (!P || !Q || !C || I < 0 || J < 0 || R <= 0) return ANTMMF_EINVAL;
    if (I == 0 || J == 0) return ANTMMF_OK;
    if ((J & 3) || (ldc & 3)) return ANTMMF_EINVAL;
    if ((ldp & 7) || (ldq & 7)) return ANTMMF_EINVAL;
    if (!p_rmajor && (R & 7)) return ANTMMF_EINVAL;
    if (!q_rmajor && (R & 7)) return ANTMMF_EINVAL;
    if (p_rmajor && (I & 7)) return ANTMMF_EINVAL;
    if (q_rmajor && (J & 7)) return ANTMMF_EINVAL;
    if (p_rmajor && !q_rmajor) return ANTMMF_EINVAL;  // layout not needed by the step
    if (c_dtype != ANTMMF_BF16 && c_dtype != ANTMMF_F32) return ANTMMF_EINVAL;
    if ((residual && (ldr & 3)) || (aux && (ldaux & 3)) || (gate && (ldgate & 3))) return ANTMMF_EINVAL;
    if (accumulate && c_dtype != ANTMMF_F32) return ANTMMF_EINVAL;
    const int nk = (R + 63) / 64;
    if (split_k < 1) split_k = 1;
    if (split_k > nk) split_k = nk;
    if (split_k > 1 && (c_dtype != ANTMMF_F32 || !accumulate || bias || act != ANTMMF_ACT_NONE || residual || aux || gate)) return ANTMMF_EINVAL;
    // segmented destination (antmmf_gemm_wgrad_bf16_seg): only the BK = 64 wgrad path with its workspace + reduce launch knows it
    if (tl_seg && !(p_rmajor && q_rmajor && (R & 63) == 0 && (I & 255) == 0 && (J & 255) == 0 && c_dtype == ANTMMF_F32 && accumulate && R >= 4096 && workspace)) return ANTMMF_ESEG;
    GemmArgs g;
    g.P = (const bf16_t*)P; g.Q = (const bf16_t*)Q; g.C = C; g.bias = bias; g.residual = (const bf16_t*)residual;
    g.aux = (bf16_t*)aux; g.gate = (const bf16_t*)gate;
    g.ldp = ldp; g.ldq = ldq; g.ldc = ldc; g.ldr = ldr; g.ldaux = ldaux; g.ldgate = ldgate;
    g.aux_grad = (act & 0x100) ? 1 : 0; g.gate_grad = (act & 0x200) ? 1 : 0;
    g.ffn_mode = 0; g.rowv = nullptr; g.colv = nullptr; g.part = tl_colsum_part;   // (antmmf_gemm_bf16_gated_colsum: per-workgroup column sums of the gated dgrad)
    g.tail_cells = 0;
    act &= 0xff;
    // aux = act'(pre-activation) is defined for an activation epilogue without a gate only (the three epilogue forms would otherwise disagree
    // about what lands in aux); gate-holds-act' needs a gate
    if (g.aux_grad && (act == ANTMMF_ACT_NONE || gate || !aux)) return ANTMMF_EINVAL;
    if (g.gate_grad && !gate) return ANTMMF_EINVAL;
    g.I = I; g.J = J; g.R = R; g.act = act; g.c_dtype = c_dtype; g.accumulate = accumulate; g.alpha = alpha; g.ws = nullptr;
    g.ksteps_per_split = (nk + split_k - 1) / split_k;
    static const char* raster_env = ANTMMF_LAB_ENV("ANTMMF_GEMM_RASTER");
    g.raster = raster_env ? atoi(raster_env) : 1;
    g.debug_nostore = 0;
    LAB_ONLY(g.debug_nostore = g_gemm_variant > 0 ? ((g_gemm_variant & 2048) ? 1 : (g_gemm_variant & 4096) ? 2 : 0) : 0;)
    const int splits = (nk + g.ksteps_per_split - 1) / g.ksteps_per_split;
    const long tiles = (long)((I + 127) / 128) * ((J + 127) / 128);
    if (tiles > 0x7fffffffL) return ANTMMF_EINVAL;
    const dim3 grid((unsigned)tiles, 1, splits), block(256);
    const size_t lds = 65536;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_dma_kernel<2, 2, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    if (!p_rmajor && !q_rmajor && (R & 63) == 0 && splits == 1) {
        const long tiles256 = (long)((I + 255) / 256) * ((J + 255) / 256);
        static const char* force = ANTMMF_LAB_ENV("ANTMMF_GEMM_FORCE_TILE");  // lab / emulator tests only: "256" / "128" / "ring" / "k"
        const bool big = force ? (force[0] == '2' || force[0] == 'r' || force[0] == 'p') : tiles256 >= 512;
        static const char* persist_env = ANTMMF_LAB_ENV("ANTMMF_GEMM_PERSIST");  // A/B knob: "0" = one workgroup per tile
        const bool persist = force ? force[0] == 'p' : !(persist_env && persist_env[0] == '0');
        static const char* pwgs_env = ANTMMF_LAB_ENV("ANTMMF_GEMM_PERSIST_WGS");  // lab / emulator tests only: a small grid makes every workgroup walk several tiles
        const unsigned pwgs = pwgs_env ? (unsigned)atoi(pwgs_env) : 256u;
        static const char* cont_env = ANTMMF_LAB_ENV("ANTMMF_GEMM_CONT");  // A/B knob: "0" = next-tile prologue as a burst in front of the epilogue
        const bool cont = !(cont_env && cont_env[0] == '0');
        const int epi = (aux || gate || act != ANTMMF_ACT_NONE || alpha != 1.0f || c_dtype != ANTMMF_BF16) ? 4 : ((bias ? 1 : 0) | (residual ? 2 : 0));
        GEMM_VARIANT_INIT();
        const bool k64p = (g_gemm_variant & 4) && c_dtype == ANTMMF_BF16 && !(ldc & 7) && !(I & 255) && !(J & 255) && R >= 128 &&
                          (!residual || !(ldr & 7)) && (!aux || !(ldaux & 7)) && (!gate || !(ldgate & 7)) &&
                          (force ? force[0] == 'k' : tiles256 >= 512);
#define K64P_LAUNCH(E_, F_)                                                                                                       \
    do {                                                                                                                          \
        static bool oncep = false;                                                                                                \
        if (!oncep) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64p_kernel<E_, F_>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); oncep = true; } \
        hipLaunchKernelGGL((gemm_nt_k64p_kernel<E_, F_>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256);              \
    } while (0)
// Timing-only ablations of gemm_nt_k64r_kernel (they store wrong data by construction: profiles/r4_gemm_store_cost_ablations.txt, r4_gemm_store_shapes.txt) exist only in a
// library built with -DANTMMF_GEMM_ABLATIONS (`make ABLATIONS=1`); the product library cannot be talked into them through ANTMMF_GEMM_VARIANT.
#ifdef ANTMMF_GEMM_ABLATIONS
#define K64R_ABL_ATTR1(EE, K) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<EE, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840)
#define K64R_ABL_ATTRS(E) do { K64R_ABL_ATTR1((E < 4 ? E : 0), 1); K64R_ABL_ATTR1(0, 8); K64R_ABL_ATTR1(0, 24); K64R_ABL_ATTR1(0, 40); K64R_ABL_ATTR1(0, 56); K64R_ABL_ATTR1(0, 64); \
                               K64R_ABL_ATTR1(0, 256); K64R_ABL_ATTR1(0, 512); } while (0)
#define K64R_ABL_GO(EE, K) do { hipLaunchKernelGGL((gemm_nt_k64r_kernel<EE, K>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256); } while (0)
#define K64R_ABL_TRY(E, done)                                                                                                     \
    do {                                                                                                                          \
        done = true;                                                                                                              \
        if ((g_gemm_variant & 8388608) && E == 0) K64R_ABL_GO(0, 256);                                                            \
        else if ((g_gemm_variant & 16777216) && E == 0) K64R_ABL_GO(0, 512);                                                      \
        else if ((g_gemm_variant & 2097152) && E == 0) K64R_ABL_GO(0, 64);                                                        \
        else if ((g_gemm_variant & 1572864) == 1572864 && E == 0 && R == 1024) K64R_ABL_GO(0, 56);                                \
        else if ((g_gemm_variant & 1048576) && E == 0 && R == 1024) K64R_ABL_GO(0, 40);                                           \
        else if ((g_gemm_variant & 524288) && E == 0 && R == 1024) K64R_ABL_GO(0, 24);                                            \
        else if ((g_gemm_variant & 262144) && E == 0 && R == 1024) K64R_ABL_GO(0, 8);                                             \
        else if (g_gemm_variant & 32768) K64R_ABL_GO((E < 4 ? E : 0), 1);                                                         \
        else done = false;                                                                                                        \
    } while (0)
#else
#define K64R_ABL_ATTRS(E) do {} while (0)
#define K64R_ABL_TRY(E, done) do {} while (0)
#endif
// (k64p shapes the rolling kernel does not take -- the generic run-time epilogue, and R = 128 -- go to the BK = 32 ring kernels below in BOTH libraries (round 6; the
// product used to serve them with the burst kernel's run-time epilogue form gemm_nt_k64p_kernel<4, 33>, 256 VGPRs + 20 B of scratch, off every bench path).  The lab
// build keeps the burst-epilogue kernel's compile-time forms for its A/B runs: variant bit 14 (every shape) or bit 27 (only the shapes the rolling kernel does not take))
#define K64P_FALLBACK(E) K64P_LAUNCH(E, PROD | K64F_PRIO)
#define LAUNCH_NT(E)                                                                                                              \
    do {                                                                                                                          \
        static bool once = false;                                                                                                 \
        if (!once) {                                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_ring_kernel<4, E>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pring_kernel<(E < 4 ? E : 0), true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
            LAB_ONLY((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pring_kernel<(E < 4 ? E : 0), false>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);) \
            LAB_ONLY((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_dma_kernel<2, 4, 8, 4, E>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);) \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_dma_kernel<2, 2, 4, 4, E>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
            once = true;                                                                                                          \
        }                                                                                                                         \
        if (k64p && ((E < 4 && R >= 192) LAB_ONLY(|| (g_gemm_variant & (16384 | 16 | 128 | 256 | 134217728))))) {                 \
            ++g_k64_launches;                                                                                                     \
            const unsigned t8 = (unsigned)((tiles256 + 7) / 8 * 8);  /* the tile walk needs a multiple of 8 workgroups (XCD = id % 8) */ \
            const unsigned gridp = (g_gemm_variant & 64) ? t8 : (pwgs < t8 ? pwgs : t8);                                          \
            constexpr int PROD = K64F_ONEBAR | ((E & 2) && E != 4 ? K64F_DIST11 : 0);                                             \
            /* the rolling-epilogue kernel (bias / residual start the accumulators, stores inside the K loop; plain write-back stores).  Same-process A/B    \
               against the burst epilogue (profiles/r4_gemm_rolling_epilogue_ab.txt, r4_gemm_store_shapes.txt): + 1 ... + 5 % on every shape of the step   \
               since its stores and residual loads move 32 contiguous bytes per row and pass.  Variant bit 14 disables it, bit 22 selects the first layout  \
               (16 rows x 16 B per pass); bit 17 non-temporal stores; the timing-only ablations (bits 15, 18 - 24) need -DANTMMF_GEMM_ABLATIONS    */    \
            if (E < 4 && !(g_gemm_variant & 16384) && R >= 192) {                                                                 \
                static bool oncer = false;                                                                                        \
                if (!oncer) {                                                                                                     \
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<(E < 4 ? E : 0), 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); \
                    LAB_ONLY((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<(E < 4 ? E : 0), 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);) \
                    K64R_ABL_ATTRS(E);                                                                                            \
                    LAB_ONLY((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<(E < 4 ? E : 0), 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);) \
                    oncer = true;                                                                                                 \
                }                                                                                                                 \
                bool abl_done = false;                                                                                            \
                K64R_ABL_TRY(E, abl_done);                                                                                        \
                if (abl_done) {}                                                                                                  \
                LAB_ONLY(else if (g_gemm_variant & 4194304) hipLaunchKernelGGL((gemm_nt_k64r_kernel<(E < 4 ? E : 0), 128>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256);) \
                LAB_ONLY(else if (g_gemm_variant & 131072) hipLaunchKernelGGL((gemm_nt_k64r_kernel<(E < 4 ? E : 0), 4>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256);) \
                else {                                                                                                            \
                    /* tail round as cells (kernel header): when the leftover tiles of an XCD chunk are so few that their 16 cells each give every workgroup of the chunk at  \
                       most ONE cell (<= 2 tiles at 32 workgroups per chunk: the image tower's J = 1024 shapes, 16 rounds + 2) and every workgroup walks at least one tile.     \
                       Lab variant bit 26 lifts the one-cell limit (tests on small grids), bit 25 disables the cells (A/B: bit-identical output either way) */               \
                    g.tail_cells = 0;                                                                                             \
                    if (E < 4 && !(g_gemm_variant & 33554432) && !(gridp & 7)) {                                                  \
                        const int per_xcd = (int)(gridp >> 3);                                                                    \
                        const long qd = tiles256 >> 3, rm = tiles256 & 7;                                                         \
                        const int r0 = (int)(qd % per_xcd), r1 = rm ? (int)((qd + 1) % per_xcd) : r0, rmax = r0 > r1 ? r0 : r1;   \
                        const bool any_tail = (g_gemm_variant & 67108864) != 0;                                                   \
                        if (rmax > 0 && (any_tail || 16 * rmax <= per_xcd) && qd >= per_xcd) { g.tail_cells = 1; LAB_ONLY(++g_k64_cell_launches;) } \
                    }                                                                                                             \
                    hipLaunchKernelGGL((gemm_nt_k64r_kernel<(E < 4 ? E : 0), 0>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256); \
                }                                                                                                                 \
            }                                                                                                                     \
            /* variant bits: 16 = two barriers per phase (the earlier schedule, kept for A/B), 128 = no s_setprio, 256 = clock probe */ \
            LAB_ONLY(else if (g_gemm_variant & 16) K64P_LAUNCH(E, K64F_PRIO | K64F_DIST11);)                                      \
            LAB_ONLY(else if (g_gemm_variant & 128) K64P_LAUNCH(E, PROD);)                                                        \
            LAB_ONLY(else if (g_gemm_variant & 256) K64P_LAUNCH(E, PROD | K64F_PRIO | K64F_CLK);)                                 \
            LAB_ONLY(else K64P_FALLBACK(E);)                                                                                      \
        }                                                                                                                         \
        else if (big && persist && E < 4 && c_dtype == ANTMMF_BF16 && !(ldc & 7) && (R & 31) == 0 && R >= 128) {                  \
            if (cont) hipLaunchKernelGGL((gemm_nt_pring_kernel<(E < 4 ? E : 0), true>), dim3(pwgs), dim3(512), 131072, stream, g, (int)tiles256); \
            LAB_ONLY(else hipLaunchKernelGGL((gemm_nt_pring_kernel<(E < 4 ? E : 0), false>), dim3(pwgs), dim3(512), 131072, stream, g, (int)tiles256);) \
        }                                                                                                                         \
        else if (big && !(force && force[0] == '2')) hipLaunchKernelGGL((gemm_nt_ring_kernel<4, E>), dim3((unsigned)tiles256), dim3(512), 131072, stream, g); \
        LAB_ONLY(else if (big) hipLaunchKernelGGL((gemm_nt_dma_kernel<2, 4, 8, 4, E>), dim3((unsigned)tiles256), dim3(512), 131072, stream, g);) \
        else hipLaunchKernelGGL((gemm_nt_dma_kernel<2, 2, 4, 4, E>), grid, block, lds, stream, g);                                \
    } while (0)
        if (k64p && aux && bias && !gate && !residual && act != ANTMMF_ACT_NONE && alpha == 1.0f && R >= 192 && !(g_gemm_variant & 16384)) {
            // forward of a feed-forward whose activation output is kept (CLIP / BERT layers of the video workloads, keep-FFN policy): bias + activation with two outputs on
            // the rolling-epilogue kernel -- the generic burst epilogue ran this shape at 0.27 of peak (two output tensors per tile behind an idle matrix pipe)
            ++g_k64_launches;
            const unsigned t8 = (unsigned)((tiles256 + 7) / 8 * 8);
            const unsigned gridp = pwgs < t8 ? pwgs : t8;
            static bool once5 = false;
            if (!once5) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<5, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); once5 = true; }
            // LAB, timing only (variant bits 28 / 29): the second output as one byte per element / not stored at all -- what an 8-bit stored derivative could buy (profiles/r6b_gemm_two_output_store_ablation.txt)
            LAB_ONLY(if (g_gemm_variant & (1 << 28)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<5, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
                       hipLaunchKernelGGL((gemm_nt_k64r_kernel<5, 1024>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256); } else)
            LAB_ONLY(if (g_gemm_variant & (1 << 29)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<5, 2048>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
                       hipLaunchKernelGGL((gemm_nt_k64r_kernel<5, 2048>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256); } else)
            if (g.aux_grad && act == ANTMMF_ACT_GELU_ERF LAB_ONLY(&& !(g_gemm_variant & (1 << 30)))) {   // (lab variant bit 30: the run-time-activation form, the A/B)
                static bool once37 = false;
                if (!once37) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<37, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); once37 = true; }
                hipLaunchKernelGGL((gemm_nt_k64r_kernel<37, 0>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256);
            } else if (g.aux_grad && act == ANTMMF_ACT_QUICK_GELU LAB_ONLY(&& !(g_gemm_variant & (1 << 30)))) {
                static bool once69 = false;
                if (!once69) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<69, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); once69 = true; }
                hipLaunchKernelGGL((gemm_nt_k64r_kernel<69, 0>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256);
            } else
            hipLaunchKernelGGL((gemm_nt_k64r_kernel<5, 0>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256);
        } else
        if (k64p && gate && !g.gate_grad && act != ANTMMF_ACT_NONE && !aux && !bias && !residual && alpha == 1.0f && !(ldgate & 7)) {
            // out = acc * act'(gate) (the dgrad of the CLIP / BERT feed-forwards under the RECOMPUTE activation policy: the pre-activation was kept, not its derivative) on the
            // burst kernel's register-level epilogue with all 16 gate vectors requested up front -- round 6; the generic run-time epilogue served it at half the speed
            ++g_k64_launches;
            const unsigned t8 = (unsigned)((tiles256 + 7) / 8 * 8);
            const unsigned gridp = pwgs < t8 ? pwgs : t8;
            K64P_LAUNCH(9, K64F_ONEBAR | K64F_DIST11 | K64F_PRIO);
        } else
        if (k64p && gate && g.gate_grad && !aux && !bias && !residual && alpha == 1.0f && !(ldgate & 7)) {
            // out = acc * gate on the register-level epilogue of the residual kernels (16 gate vectors requested up front) instead of the generic one
            ++g_k64_launches;
            const unsigned t8 = (unsigned)((tiles256 + 7) / 8 * 8);
            const unsigned gridp = pwgs < t8 ? pwgs : t8;
            // round 5: on the rolling-epilogue kernel (gate vectors requested two phases ahead of each row quarter's store, multiplied into the accumulators in front of the
            // conversion); the burst form with all 16 gate vectors up front stays for R < 192 and, in the lab library, as the A/B (variant bit 14)
            if (R >= 192 && !(g_gemm_variant & 16384)) {
                static bool once8 = false;
                if (!once8) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<8, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); once8 = true; }
                g.tail_cells = 0;
                if (g.part && J <= 4096) {   // ... with the per-workgroup column sums (antmmf_gemm_bf16_gated_colsum)
                    static bool once10 = false;
                    if (!once10) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<10, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); once10 = true; }
                    hipLaunchKernelGGL((gemm_nt_k64r_kernel<10, 0>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256);
                } else
                hipLaunchKernelGGL((gemm_nt_k64r_kernel<8, 0>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256);
            } else
            K64P_LAUNCH(8, K64F_ONEBAR | K64F_DIST11 | K64F_PRIO);
        } else
        switch (epi) {
            case 0: LAUNCH_NT(0); break;
            case 1: LAUNCH_NT(1); break;
            case 2: LAUNCH_NT(2); break;
            case 3: LAUNCH_NT(3); break;
            default: LAUNCH_NT(4); break;
        }
#undef LAUNCH_NT
#undef K64P_FALLBACK
#undef K64P_LAUNCH
    }
    else if (!p_rmajor && !q_rmajor) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, lds, stream, g);
    else if (!p_rmajor && q_rmajor) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, lds, stream, g);
    else if (p_rmajor && q_rmajor && (R & 31) == 0 && (I & 255) == 0 && (J & 255) == 0 && c_dtype == ANTMMF_F32 && accumulate &&
             !bias && act == ANTMMF_ACT_NONE && !residual && !aux && !gate && R >= 4096) {
        // wgrad ring: the host-side split_k hint is replaced by "enough workgroups to fill 256 CUs twice"
        const int tiles = (I / 256) * (J / 256), nk32 = R / 32;
        static const char* wgs_env = ANTMMF_LAB_ENV("ANTMMF_WGRAD_WGS");  // experiments only
        const int want_wgs = wgs_env ? atoi(wgs_env) : 256;
        int sp = want_wgs / tiles;  // floor: one resident round of workgroups (36 tiles x 8 splits = 288 would need a second, 12 % full round)
        if (sp < 1) sp = 1;
        if (sp > 32) sp = 32;  // (d = 768 towers: 9 output tiles per 768 x 768 weight need 28 splits to fill the 256 CUs)
        if (sp > nk32 / 8) sp = nk32 / 8 > 0 ? nk32 / 8 : 1;
        g.ksteps_per_split = (nk32 + sp - 1) / sp;
        const int zs = (nk32 + g.ksteps_per_split - 1) / g.ksteps_per_split;
        const bool use_ws = zs > 1 && workspace && workspace_bytes >= (long)zs * I * J * 4;
        if (use_ws) g.ws = workspace;
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_ring_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            LAB_ONLY((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_ring_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);)
            once = true;
        }
        GEMM_VARIANT_INIT();
        // BK = 64 schedule (default; variant bit 10 = the BK = 32 ring for A/B): K-tiles of 64 tokens, every split needs >= 2 of them
        int k64_steps = 0, k64_zs = 0;
        if ((g_gemm_variant & 4) && !(g_gemm_variant & 1024) && (R & 63) == 0 && R >= 128) {
            const int nk64 = R / 64;
            int spl = sp < nk64 / 2 ? sp : nk64 / 2;
            if (spl < 1) spl = 1;
            k64_steps = (nk64 + spl - 1) / spl;
            while ((nk64 % k64_steps) == 1) ++k64_steps;   // a trailing split of a single K-tile would have no steady state
            k64_zs = (nk64 + k64_steps - 1) / k64_steps;
        }
        if (tl_seg && !(k64_steps >= 2 && k64_zs > 1 && workspace_bytes >= (long)k64_zs * I * J * 4)) return ANTMMF_ESEG;
        if (k64_steps >= 2) {
            g.ksteps_per_split = k64_steps;
            const bool ws64 = k64_zs > 1 && workspace && workspace_bytes >= (long)k64_zs * I * J * 4;
            g.ws = ws64 ? workspace : nullptr;
            static bool once64 = false;
            if (!once64) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_k64_kernel<K64F_PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072); once64 = true; }
            ++g_k64_launches;
            hipLaunchKernelGGL(gemm_tn_k64_kernel<K64F_PRIO>, dim3((unsigned)(tiles * k64_zs)), dim3(512), 131072, stream, g);
            if (ws64) {
                const long nvec = (long)I * J / 4;
                const int rg = (int)((nvec + 255) / 256 < 2048 ? (nvec + 255) / 256 : 2048);
                if (tl_seg) {
                    const SegDst sd = *tl_seg;   // (a copy made on THIS thread: the CPU lane emulator evaluates launch arguments on its worker threads, where the thread-local is null)
                    hipLaunchKernelGGL(splitk_reduce_seg_kernel, dim3(rg), dim3(256), 0, stream, workspace, sd, k64_zs, I, J, ldc);
                }
                else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, stream, workspace, reinterpret_cast<float*>(C), k64_zs, I, J, ldc, 1);
            }
            return antmmf_check_launch();
        }
        LAB_ONLY(if (g.raster & 16) hipLaunchKernelGGL(gemm_tn_ring_kernel<false>, dim3((unsigned)(tiles * zs)), dim3(512), 131072, stream, g); else)
        hipLaunchKernelGGL(gemm_tn_ring_kernel<true>, dim3((unsigned)(tiles * zs)), dim3(512), 131072, stream, g);
        if (use_ws) {
            const long nvec = (long)I * J / 4;
            const int rg = (int)((nvec + 255) / 256 < 2048 ? (nvec + 255) / 256 : 2048);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, stream, workspace, reinterpret_cast<float*>(C), zs, I, J, ldc, 1);
        }
    } else if (p_rmajor && q_rmajor && (R & 63) == 0 && (I & 127) == 0 && (J & 127) == 0) {
        hipLaunchKernelGGL((gemm_tn_dma_kernel<2, 2, 4, 4>), grid, block, lds, stream, g);
    } else hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, lds, stream, g);
    return antmmf_check_launch();
}

#ifdef ANTMMF_LAB   // the sub-LN fold is a measured-neutral option: its entry points (include/antmmf_hip_lab.h) exist in the lab library only
// =====================================================================================================================================
// Sub-LN fold: the M2 feed-forward  y = fc2(LayerNorm_4d(gelu(fc1(x)))) + residual  (reference prj/M2_Encoder/vlmo/torchscale/component/
// feedforward_network.py:117-128: fc1 -> activation_fn -> ffn_layernorm -> fc2) without the two 4d-wide LayerNorm passes.  With z = gelu(u),
// (mu_i, r_i) the row statistics of z, W2g = W2 diag(gamma), c_j = sum_k W2g[j][k], b2f = b2 + W2 beta:
//   forward   fc1's epilogue stores z and gelu'(u) and the row sums of z;  y_ij = r_i (z W2g^T)_ij - r_i mu_i c_j + b2f_j + res_ij  (fc2's epilogue)
//   backward  t = dy W2g (dgrad) ;  dz_ik = r_i (t_ik - m1_i - zhat_ik m2_i),  m1 = mean_k t = dy . c / F,  m2 = mean_k t zhat = dy . (y - b2f - res) / F
//             -- both row dot products over d-wide tensors the step keeps anyway (a row pass over 3 d-wide tensors instead of 3 4d-wide ones) --
//             du = gelu'(u) dz in the dgrad's epilogue, with the fc1 bias gradient (column sums of du) from the same tiles;
//             dW2_jk = gamma_k (Gm_jk - s_j) + beta_k cs_j,  Gm = (r dy)^T z,  s_j = sum_i r_i mu_i dy_ij,  cs = column sums of dy;
//             dgamma_k = sum_j W2_jk (Gm_jk - s_j),  dbeta_k = sum_j W2_jk cs_j.
// The 4d-wide tensors are touched by GEMM epilogues only.
static int gemm_ffn_launch(GemmArgs& g, hipStream_t stream) {
    const long tiles256 = (long)((g.I + 255) / 256) * ((g.J + 255) / 256);
    static const char* force = getenv("ANTMMF_GEMM_FORCE_TILE");
    const bool aligned = !(g.I & 255) && !(g.J & 255) && !(g.R & 63) && g.R >= 128 && !(g.ldc & 7) && !(g.ldp & 7) && !(g.ldq & 7) &&
                         (!g.residual || !(g.ldr & 7)) && (!g.aux || !(g.ldaux & 7)) && (!g.gate || !(g.ldgate & 7));
    const bool k64p = aligned && g.part && (force ? force[0] == 'k' : tiles256 >= 512);
    if (!k64p) {
        g.part = nullptr;
        const long tiles = (long)((g.I + 127) / 128) * ((g.J + 127) / 128);
        static bool once = false;
        if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); once = true; }
        hipLaunchKernelGGL((gemm_kernel<false, false>), dim3((unsigned)tiles, 1, 1), dim3(256), 65536, stream, g);
        return 0;
    }
    static const char* pwgs_env = getenv("ANTMMF_GEMM_PERSIST_WGS");
    const unsigned pwgs = pwgs_env ? (unsigned)atoi(pwgs_env) : 256u;
    const unsigned t8 = (unsigned)((tiles256 + 7) / 8 * 8);
    const unsigned gridp = pwgs < t8 ? pwgs : t8;
    ++g_k64_launches;
#define FFN_LAUNCH(E_, F_)                                                                                                        \
    do {                                                                                                                          \
        static bool oncep = false;                                                                                                \
        if (!oncep) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64p_kernel<E_, F_>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); oncep = true; } \
        hipLaunchKernelGGL((gemm_nt_k64p_kernel<E_, F_>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256);              \
    } while (0)
    if (g.ffn_mode == 1 && g.R >= 192 && !(g_gemm_variant > 0 && (g_gemm_variant & 16384))) {   // fc1 of the fold: the rolling two-output epilogue + row sums
        static bool once21 = false;
        if (!once21) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_k64r_kernel<21, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); once21 = true; }
        g.aux_grad = 1;
        hipLaunchKernelGGL((gemm_nt_k64r_kernel<21, 0>), dim3(gridp), dim3(512), 163840, stream, g, (int)tiles256);
    }
    else if (g.ffn_mode == 1) FFN_LAUNCH(16, K64F_ONEBAR | K64F_PRIO);
    else if (g.ffn_mode == 2) FFN_LAUNCH(32, K64F_ONEBAR | K64F_DIST11 | K64F_PRIO);
    else FFN_LAUNCH(64, K64F_ONEBAR | K64F_DIST11 | K64F_PRIO);
#undef FFN_LAUNCH
    return 1;
}

static void gemm_ffn_args(GemmArgs& g, const void* P, const void* Q, void* C, int I, int J, int R, long ldp, long ldq, long ldc) {
    g.P = (const bf16_t*)P; g.Q = (const bf16_t*)Q; g.C = C; g.bias = nullptr; g.residual = nullptr; g.aux = nullptr; g.gate = nullptr;
    g.ldp = ldp; g.ldq = ldq; g.ldc = ldc; g.ldr = 0; g.ldaux = 0; g.ldgate = 0;
    g.I = I; g.J = J; g.R = R; g.act = ANTMMF_ACT_NONE; g.c_dtype = ANTMMF_BF16; g.accumulate = 0; g.ksteps_per_split = (R + 63) / 64; g.alpha = 1.0f;
    g.ws = nullptr; g.raster = 1; g.aux_grad = 0; g.gate_grad = 0; g.debug_nostore = 0;
    g.ffn_mode = 0; g.rowv = nullptr; g.colv = nullptr; g.part = nullptr;
    g.tail_cells = 0;
}
static bool ffn_shape_ok(int I, int J, int R, long ldp, long ldq, long ldc) {
    return I > 0 && J > 0 && R > 0 && !(J & 7) && !(R & 7) && !(ldp & 7) && !(ldq & 7) && !(ldc & 7);
}

// (sum z, sum z^2) partials per 64-column block -> (mean, rstd) per row
__global__ __launch_bounds__(256) void ffn_stats_reduce_kernel(const float* __restrict__ part, int nblk, int I, int F, float eps, float* __restrict__ stats) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= I) return;
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < nblk; ++b) {
        const float2 p = *reinterpret_cast<const float2*>(part + 2 * ((long)b * I + i));
        s1 += p.x; s2 += p.y;
    }
    const float mu = s1 / (float)F;
    float var = s2 / (float)F - mu * mu;
    var = var > 0.f ? var : 0.f;
    *reinterpret_cast<float2*>(stats + 2 * (long)i) = make_float2(mu, rsqrtf(var + eps));
}
// the same statistics straight from a stored z (shapes that do not run on the persistent kernel): one wave per row, two-pass
__global__ __launch_bounds__(256) void ffn_row_stats_kernel(const bf16_t* __restrict__ Z, long ldz, int I, int F, float eps, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= I) return;
    const bf16_t* z = Z + i * ldz;
    float s1 = 0.f;
    for (int k = lane * 8; k < F; k += 512) { float v[8]; ld8<bf16_t>(z + k, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += v[e]; }
    const float mu = wave_sum(s1) / (float)F;
    float s2 = 0.f;
    for (int k = lane * 8; k < F; k += 512) { float v[8]; ld8<bf16_t>(z + k, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s2 += (v[e] - mu) * (v[e] - mu); }
    const float var = wave_sum(s2) / (float)F;
    if (lane == 0) *reinterpret_cast<float2*>(stats + 2 * i) = make_float2(mu, rsqrtf(var + eps));
}
// out[c] += sum over the partial rows (split over gridDim.y, closed with one atomic per column and slice)
__global__ __launch_bounds__(256) void ffn_colpart_reduce_kernel(const float* __restrict__ part, int nparts, int J, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= J) return;
    const int per = (nparts + gridDim.y - 1) / gridDim.y, p0 = blockIdx.y * per, p1 = p0 + per < nparts ? p0 + per : nparts;
    float s = 0.f;
    for (int p = p0; p < p1; ++p) s += part[(long)p * J + c];
    if (p1 > p0) atomicAdd(out + c, s);
}

extern "C" int antmmf_colsum(const void* x, float* out, long rows, int cols, long ld, int dtype, hipStream_t s);

extern "C" int antmmf_ffn_fc1_fwd(const void* X, const void* W1, const float* b1, void* Z, void* DACT, float* stats, int tokens, int n_ff, int n_in,
                                  long ldx, long ldw, long ldz, int act, float eps, float* workspace, long workspace_bytes, hipStream_t stream) {
    if (!X || !W1 || !b1 || !Z || !DACT || !stats || !ffn_shape_ok(tokens, n_ff, n_in, ldx, ldw, ldz) || act == ANTMMF_ACT_NONE) return ANTMMF_EINVAL;
    GemmArgs g;
    gemm_ffn_args(g, X, W1, Z, tokens, n_ff, n_in, ldx, ldw, ldz);
    g.bias = b1; g.aux = (bf16_t*)DACT; g.ldaux = ldz; g.act = act; g.ffn_mode = 1;
    const int nblk = n_ff / 64;
    if (workspace && !(n_ff & 63) && workspace_bytes >= (long)nblk * tokens * 8) g.part = workspace;
    if (gemm_ffn_launch(g, stream)) hipLaunchKernelGGL(ffn_stats_reduce_kernel, dim3((tokens + 255) / 256), dim3(256), 0, stream, workspace, nblk, tokens, n_ff, eps, stats);
    else hipLaunchKernelGGL(ffn_row_stats_kernel, dim3((tokens + 3) / 4), dim3(256), 0, stream, (const bf16_t*)Z, ldz, tokens, n_ff, eps, stats);
    return antmmf_check_launch();
}

extern "C" int antmmf_ffn_fc2_fwd(const void* Z, const void* W2g, const float* colsum_w2g, const float* b2f, const float* stats, const void* RES, void* Y,
                                  int tokens, int n_out, int n_ff, long ldz, long ldw, long ldres, long ldy, hipStream_t stream) {
    if (!Z || !W2g || !colsum_w2g || !b2f || !stats || !RES || !Y || !ffn_shape_ok(tokens, n_out, n_ff, ldz, ldw, ldy) || (ldres & 7)) return ANTMMF_EINVAL;
    GemmArgs g;
    gemm_ffn_args(g, Z, W2g, Y, tokens, n_out, n_ff, ldz, ldw, ldy);
    g.bias = b2f; g.residual = (const bf16_t*)RES; g.ldr = ldres; g.rowv = stats; g.colv = colsum_w2g; g.ffn_mode = 2;
    float dummy;   // the persistent kernel is chosen by g.part != NULL; mode 2 writes no partials
    g.part = &dummy;
    gemm_ffn_launch(g, stream);
    return antmmf_check_launch();
}

extern "C" int antmmf_ffn_fc2_dgrad(const void* dY, const void* W2gT, const void* Z, const void* DACT, const float* rowv4, void* dU, float* db1,
                                    int tokens, int n_ff, int n_out, long lddy, long ldw, long ldz, long lddu, float* workspace, long workspace_bytes,
                                    hipStream_t stream) {
    if (!dY || !W2gT || !Z || !DACT || !rowv4 || !dU || !ffn_shape_ok(tokens, n_ff, n_out, lddy, ldw, lddu) || (ldz & 7)) return ANTMMF_EINVAL;
    GemmArgs g;
    gemm_ffn_args(g, dY, W2gT, dU, tokens, n_ff, n_out, lddy, ldw, lddu);
    g.residual = (const bf16_t*)Z; g.ldr = ldz; g.gate = (const bf16_t*)DACT; g.ldgate = ldz; g.rowv = rowv4; g.ffn_mode = 3;
    const int nparts = tokens / 128;
    if (workspace && !(tokens & 127) && workspace_bytes >= (long)nparts * n_ff * 4) g.part = workspace;
    if (gemm_ffn_launch(g, stream)) {
        if (db1) hipLaunchKernelGGL(ffn_colpart_reduce_kernel, dim3((n_ff + 255) / 256, 16), dim3(256), 0, stream, workspace, nparts, n_ff, db1);
    } else if (db1) {
        const int rc = antmmf_colsum(dU, db1, tokens, n_ff, lddu, ANTMMF_BF16, stream);
        if (rc) return rc;
    }
    return antmmf_check_launch();
}

// Backward row pass over the d-wide tensors: rowv4[i] = (mu, rstd, m1, m2), dYs = bf16(rstd_i dy), s_col[j] += sum_i rstd_i mu_i dy_ij, cs_col[j] += sum_i dy_ij.
// One wave per row (n_out <= 2048: up to four 16-B chunks per lane), persistent over rows; the column sums are closed per workgroup (LDS) + one atomic per column.
template <int NCH>
__global__ __launch_bounds__(256) void ffn_bwd_rows_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ Y, const bf16_t* __restrict__ RES,
                                                           const float* __restrict__ b2f, const float* __restrict__ cw, const float* __restrict__ stats,
                                                           float* __restrict__ rowv4, bf16_t* __restrict__ dYs, float* __restrict__ s_col, float* __restrict__ cs_col,
                                                           int tokens, int n_out, float inv_f, long lddy, long ldy, long ldres, long lddys) {
    ANTMMF_DYN_LDS(float, red);   // [4 waves][2][n_out]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float cb[NCH][8], cc[NCH][8], as_[NCH][8], ac[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k = c * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { as_[c][e] = 0.f; ac[c][e] = 0.f; cb[c][e] = 0.f; cc[c][e] = 0.f; }
        if (k < n_out) { ld8<float>(b2f + k, cb[c]); ld8<float>(cw + k, cc[c]); }
    }
    for (long i = (long)blockIdx.x * 4 + wave; i < tokens; i += (long)gridDim.x * 4) {
        float dy[NCH][8];
        float d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = c * 512 + lane * 8;
            if (k < n_out) {
                float y[8], r[8];
                ld8<bf16_t>(dY + i * lddy + k, dy[c]); ld8<bf16_t>(Y + i * ldy + k, y); ld8<bf16_t>(RES + i * ldres + k, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) { d1 += dy[c][e] * cc[c][e]; d2 += dy[c][e] * ((y[e] - r[e]) - cb[c][e]); }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) dy[c][e] = 0.f;
            }
        }
        d1 = wave_sum(d1) * inv_f; d2 = wave_sum(d2) * inv_f;
        const float2 st = *reinterpret_cast<const float2*>(stats + 2 * i);
        if (lane == 0) *reinterpret_cast<float4*>(rowv4 + 4 * i) = make_float4(st.x, st.y, d1, d2);
        const float rm = st.y * st.x;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = c * 512 + lane * 8;
            if (k < n_out) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] = st.y * dy[c][e]; as_[c][e] += rm * dy[c][e]; ac[c][e] += dy[c][e]; }
                st8<bf16_t>(dYs + i * lddys + k, o);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k = c * 512 + lane * 8;
        if (k < n_out) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[(wave * 2 + 0) * n_out + k + e] = as_[c][e]; red[(wave * 2 + 1) * n_out + k + e] = ac[c][e]; }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_out; k += 256) {
        atomicAdd(s_col + k, (red[0 * n_out + k] + red[2 * n_out + k]) + (red[4 * n_out + k] + red[6 * n_out + k]));
        if (cs_col) atomicAdd(cs_col + k, (red[1 * n_out + k] + red[3 * n_out + k]) + (red[5 * n_out + k] + red[7 * n_out + k]));
    }
}

extern "C" int antmmf_ffn_bwd_rows(const void* dY, const void* Y, const void* RES, const float* b2f, const float* colsum_w2g, const float* stats,
                                   float* rowv4, void* dYs, float* s_col, float* cs_col, int tokens, int n_out, int n_ff,
                                   long lddy, long ldy, long ldres, long lddys, hipStream_t stream) {
    if (!dY || !Y || !RES || !b2f || !colsum_w2g || !stats || !rowv4 || !dYs || !s_col || tokens <= 0 || n_out <= 0 || (n_out & 7) || n_out > 2048 || n_ff <= 0 ||
        (lddy & 7) || (ldy & 7) || (ldres & 7) || (lddys & 7)) return ANTMMF_EINVAL;
    int wgs = (tokens + 3) / 4;
    if (wgs > 1024) wgs = 1024;
    const size_t lds = (size_t)8 * n_out * sizeof(float);
    const float inv_f = 1.0f / (float)n_ff;
#define ROWS_LAUNCH(N)                                                                                                                          \
    do {                                                                                                                                        \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_bwd_rows_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(ffn_bwd_rows_kernel<N>, dim3(wgs), dim3(256), lds, stream, (const bf16_t*)dY, (const bf16_t*)Y, (const bf16_t*)RES, b2f, colsum_w2g, \
                           stats, rowv4, (bf16_t*)dYs, s_col, cs_col, tokens, n_out, inv_f, lddy, ldy, ldres, lddys);                           \
    } while (0)
    const int nch = (n_out + 511) / 512;
    if (nch == 1) ROWS_LAUNCH(1); else if (nch == 2) ROWS_LAUNCH(2); else if (nch == 3) ROWS_LAUNCH(3); else ROWS_LAUNCH(4);
#undef ROWS_LAUNCH
    return antmmf_check_launch();
}

// dW2[j][k] += gamma_k (Gm[j][k] - s_j) + beta_k cs_j;  dgamma_k += sum_j W2[j][k] (Gm[j][k] - s_j);  dbeta_k += sum_j W2[j][k] cs_j.
// Workgroup = 16 rows j x 256 columns k (thread = one column): coalesced rows, the column sums of the slab go out as one atomic per column.
__global__ __launch_bounds__(256) void ffn_wgrad_post_kernel(const float* __restrict__ Gm, const float* __restrict__ W2, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ s, const float* __restrict__ cs,
                                                             float* __restrict__ dW2, float* __restrict__ dgamma, float* __restrict__ dbeta, int n_out, int n_ff) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_ff) return;
    const int j0 = blockIdx.y * 16, j1 = j0 + 16 < n_out ? j0 + 16 : n_out;
    const float ga = gamma[k], be = beta[k];
    float dg = 0.f, db = 0.f;
    for (int j = j0; j < j1; ++j) {
        const long o = (long)j * n_ff + k;
        const float h = Gm[o] - s[j], w = W2[o], c = cs[j];
        dW2[o] += ga * h + be * c;
        dg += w * h; db += w * c;
    }
    if (dgamma) atomicAdd(dgamma + k, dg);
    if (dbeta) atomicAdd(dbeta + k, db);
}
extern "C" int antmmf_ffn_wgrad_post(const float* Gm, const float* W2, const float* gamma, const float* beta, const float* s, const float* cs, float* dW2,
                                     float* dgamma, float* dbeta, int n_out, int n_ff, hipStream_t stream) {
    if (!Gm || !W2 || !gamma || !beta || !s || !cs || !dW2 || n_out <= 0 || n_ff <= 0) return ANTMMF_EINVAL;
    hipLaunchKernelGGL(ffn_wgrad_post_kernel, dim3((n_ff + 255) / 256, (n_out + 15) / 16), dim3(256), 0, stream, Gm, W2, gamma, beta, s, cs, dW2, dgamma, dbeta, n_out, n_ff);
    return antmmf_check_launch();
}

// Once per optimizer step and layer: W2g[j][k] = bf16(W2[j][k] gamma_k),  c_j = sum_k W2g[j][k] (of the ROUNDED operand the GEMM multiplies),
// b2f_j = b2_j + sum_k beta_k W2[j][k].  One workgroup per output row j.
__global__ __launch_bounds__(256) void ffn_prepare_w2_kernel(const float* __restrict__ W2, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ b2, bf16_t* __restrict__ W2g, float* __restrict__ c, float* __restrict__ b2f, int n_ff) {
    __shared__ float red[2][4];
    const int j = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float sc = 0.f, sb = 0.f;
    for (int k = threadIdx.x; k < n_ff; k += 256) {
        const float w = W2[(long)j * n_ff + k];
        const bf16_t q = f2bf(w * gamma[k]);
        W2g[(long)j * n_ff + k] = q;
        sc += bf2f(q); sb += beta[k] * w;
    }
    sc = wave_sum(sc); sb = wave_sum(sb);
    if (lane == 0) { red[0][wave] = sc; red[1][wave] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        c[j] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        b2f[j] = (b2 ? b2[j] : 0.f) + ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}
extern "C" int antmmf_ffn_prepare_w2(const float* W2, const float* gamma, const float* beta, const float* b2, void* W2g, float* c, float* b2f, int n_out, int n_ff,
                                     hipStream_t stream) {
    if (!W2 || !gamma || !beta || !W2g || !c || !b2f || n_out <= 0 || n_ff <= 0) return ANTMMF_EINVAL;
    hipLaunchKernelGGL(ffn_prepare_w2_kernel, dim3(n_out), dim3(256), 0, stream, W2, gamma, beta, b2, (bf16_t*)W2g, c, b2f, n_ff);
    return antmmf_check_launch();
}

#endif  // ANTMMF_LAB (sub-LN fold)

extern "C" int antmmf_gemm_bf16_ws(const void* P, const void* Q, void* C, int I, int J, int R, long ldp, long ldq, long ldc,
                                   int p_rmajor, int q_rmajor, int c_dtype, float alpha, const float* bias, int act,
                                   const void* residual, long ldr, void* aux, long ldaux, const void* gate, long ldgate,
                                   int accumulate, int split_k, float* workspace, long workspace_bytes, hipStream_t stream) {
    return gemm_impl(P, Q, C, I, J, R, ldp, ldq, ldc, p_rmajor, q_rmajor, c_dtype, alpha, bias, act, residual, ldr, aux, ldaux, gate, ldgate,
                     accumulate, split_k, workspace, workspace_bytes, stream);
}
extern "C" int antmmf_gemm_bf16(const void* P, const void* Q, void* C, int I, int J, int R, long ldp, long ldq, long ldc,
                                int p_rmajor, int q_rmajor, int c_dtype, float alpha, const float* bias, int act,
                                const void* residual, long ldr, void* aux, long ldaux, const void* gate, long ldgate,
                                int accumulate, int split_k, hipStream_t stream) {
    return gemm_impl(P, Q, C, I, J, R, ldp, ldq, ldc, p_rmajor, q_rmajor, c_dtype, alpha, bias, act, residual, ldr, aux, ldaux, gate, ldgate,
                     accumulate, split_k, nullptr, 0, stream);
}

// out = (P Q^T) * gate (gate = the activation derivative the forward stored) PLUS the column sums of `out`, as 256 partial rows: colsum_part [256][J] fp32, ZERO-FILLED by the
// caller (row w = what workgroup w's tiles contributed; workgroups without a tile leave theirs untouched) -- the bias gradient of the Linear in front of the activation is the column
// sum of those 256 rows instead of a pass over the [I][J] tensor.  Served by the rolling-epilogue kernel only: antmmf_gemm_bf16_gated_colsum_ok says whether a shape is.
extern "C" int antmmf_gemm_bf16_gated_colsum_ok(int I, int J, int R, long ldc, long ldgate) {
    const long tiles256 = (long)((I + 255) / 256) * ((J + 255) / 256);
    bool enough = tiles256 >= 512;
#ifdef ANTMMF_LAB
    static const char* force = ANTMMF_LAB_ENV("ANTMMF_GEMM_FORCE_TILE");   // (lab / emulator tests: "k" forces the BK = 64 kernels on small problems)
    if (force) enough = force[0] == 'k';
    GEMM_VARIANT_INIT();
    if (!(g_gemm_variant & 4) || (g_gemm_variant & 16384)) return 0;       // (lab A/B variants that send the gated dgrad to another kernel: that one has no column sums)
#endif
    return I > 0 && J > 0 && !(I & 255) && !(J & 255) && J <= 4096 && !(R & 63) && R >= 192 && enough && !(ldc & 7) && !(ldgate & 7) ? 1 : 0;
}
extern "C" int antmmf_gemm_bf16_gated_colsum(const void* P, const void* Q, void* C, int I, int J, int R, long ldp, long ldq, long ldc, const void* gate, long ldgate,
                                             float* colsum_part, hipStream_t stream) {
    if (!gate || !colsum_part || !antmmf_gemm_bf16_gated_colsum_ok(I, J, R, ldc, ldgate)) return ANTMMF_EINVAL;
    tl_colsum_part = colsum_part;
    const int rc = gemm_impl(P, Q, C, I, J, R, ldp, ldq, ldc, 0, 0, ANTMMF_BF16, 1.0f, nullptr, ANTMMF_ACT_GELU_ERF | 0x200, nullptr, 0, nullptr, 0, gate, ldgate, 0, 1, nullptr, 0, stream);
    tl_colsum_part = nullptr;
    return rc;
}

// dW[n_out][k_in] += dY[tokens][n_out]^T X[tokens][k_in]  (fp32 accumulate), with a caller-owned fp32 workspace for the token-split
// partial sums (the kernel picks the split; workspace_bytes >= 32 * n_out * k_in * 4 always suffices; NULL -> fp32 atomics).
extern "C" int antmmf_gemm_wgrad_bf16(const void* dY, const void* X, float* dW, long tokens, int n_out, int k_in, long ld_dy, long ld_x,
                                      long ld_dw, int split_k_hint, float* workspace, long workspace_bytes, hipStream_t stream) {
    if (tokens <= 0 || tokens > 0x7fffffffL) return ANTMMF_EINVAL;
    return gemm_impl(dY, X, dW, n_out, k_in, (int)tokens, ld_dy, ld_x, ld_dw, 1, 1, ANTMMF_F32, 1.0f, nullptr, ANTMMF_ACT_NONE, nullptr, 0, nullptr, 0,
                     nullptr, 0, 1, split_k_hint, workspace, workspace_bytes, stream);
}

// The same for n_seg (<= 4) weights of seg_rows x k_in each whose gradients live at unrelated addresses dW[0 .. n_seg) while their dY columns are adjacent
// (dY[tokens][n_seg * seg_rows]: the packed dQKV of a layer with separate q / k / v projections): ONE wgrad GEMM of n_seg * seg_rows output rows -- fewer, fuller token
// splits than n_seg launches of seg_rows rows (3072 x 1024: 48 tiles x 5 splits instead of 3 x (16 tiles x 16 splits); a third of the partial-sum traffic) -- whose
// reduce launch scatters the row segments.  Shapes that do not take the workspace path run as n_seg ordinary calls.
extern "C" int antmmf_gemm_wgrad_bf16_seg(const void* dY, const void* X, float* const* dW, int n_seg, int seg_rows, long tokens, int k_in, long ld_dy, long ld_x,
                                          long ld_dw, int split_k_hint, float* workspace, long workspace_bytes, hipStream_t stream) {
    if (!dW || n_seg < 1 || n_seg > 4 || seg_rows <= 0 || tokens <= 0 || tokens > 0x7fffffffL) return ANTMMF_EINVAL;
    for (int s = 0; s < n_seg; ++s) if (!dW[s]) return ANTMMF_EINVAL;
    if (n_seg > 1) {
        SegDst d; d.rows = seg_rows;
        for (int s = 0; s < 4; ++s) d.p[s] = dW[s < n_seg ? s : 0];
        tl_seg = &d;
        const int rc = gemm_impl(dY, X, dW[0], n_seg * seg_rows, k_in, (int)tokens, ld_dy, ld_x, ld_dw, 1, 1, ANTMMF_F32, 1.0f, nullptr, ANTMMF_ACT_NONE, nullptr, 0, nullptr, 0,
                                 nullptr, 0, 1, split_k_hint, workspace, workspace_bytes, stream);
        tl_seg = nullptr;
        if (rc != ANTMMF_ESEG) return rc;
    }
    for (int s = 0; s < n_seg; ++s) {
        const int rc = antmmf_gemm_wgrad_bf16((const bf16_t*)dY + (long)s * seg_rows, X, dW[s], tokens, seg_rows, k_in, ld_dy, ld_x, ld_dw, split_k_hint, workspace, workspace_bytes, stream);
        if (rc != ANTMMF_OK) return rc;
    }
    return ANTMMF_OK;
}
