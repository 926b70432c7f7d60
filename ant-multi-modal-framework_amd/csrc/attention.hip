// attention.hip -- fused multi-head attention (forward + backward) for short sequences on gfx950.
//
// The contrastive towers never see more than 257 image tokens / 77 text tokens and head_dim is 64
// everywhere (SURVEY.md 2.4 K3, section 5 "long-context: absent"), so this is a whole-row design, not
// a streaming flash kernel: one workgroup owns one (batch, head), parks that head's K and V (<= 288
// keys, 37 KiB each) in LDS once, and each wave walks 16-query tiles with the full score row held in
// registers.  No [B*h, N, N] tensor ever reaches HBM (the reference materialises it three times:
// modeling_bert.py:144-161, multihead_attention.py:120-145, nn.MultiheadAttention's math path).
//
// MFMA bookkeeping (v_mfma_f32_16x16x32_bf16; A: lane l holds A[l&15][8g+e], B: lane l holds B[8g+e][l&15],
// D: lane l holds D[4g+r][l&15], g = l>>4):
//   * scores are computed TRANSPOSED, S^T = K Q^T, so a lane ends up with one query (column l&15) and, per
//     16-key tile, keys 4g..4g+3: the row max / row sum need only two wave shuffles (xor 16, xor 32);
//   * the k-order inside an MFMA is free as long as A and B agree, so the 32 key slots of one P.V MFMA are
//     DEFINED as slot 8g+e -> key 4g+e of the even tile (e < 4) / key 4g+e-4 of the odd tile (e >= 4): the
//     probabilities a lane already holds ARE its B fragment -- no LDS round trip, no cross-lane traffic;
//   * the matching A fragment (V^T, K^T, Q^T, dO^T: "16 head-dim rows x those 32 token slots") comes straight out of
//     the ROW-MAJOR token tile with two ds_read_b64_tr_b16 (the gfx950 transposing LDS read: a 16-lane group reads a
//     4-token x 16-dh block and lane i receives column i) -- no transposed copies in LDS, which halves the LDS footprint
//     (2 workgroups per CU instead of 1 in the backward kernels) and removes the 8-ds_write_b32-per-row-pair staging;
//   * every output fragment is "4 consecutive head-dim values of one token": 8-B stores.
// LDS layout: token tiles [tokens][64] with 128-B rows, 16-B slot XOR-swizzled by attn_swz(row): conflict-free both for
// the ds_read_b128 fragment reads (16 consecutive rows x one slot) and for the transposing reads (8 consecutive rows x
// one 32-B slot pair per half-wave).
// Softmax runs in the exp2 domain (scale and bias pre-multiplied by log2 e): one v_exp_f32 per score, no extra multiply.
//
// Replaces (reference): CLIP nn.MultiheadAttention core clip/model.py:245-251; BertSelfAttention
// modeling_bert.py:144-161 (additive -10000 key mask); torchscale MultiheadAttention
// multihead_attention.py:120-145 (-inf key padding, fp32 softmax); ViLBERT co-attention vilbert.py:360-400
// (two calls with the streams swapped); DMAE TransformerClip dmae_utils.py:589-619.
//
// Roofline: MFMA-bound on paper (4*Nq*Nk*64 flop per (b, h) forward, 14*Nq*Nk*64 backward as written: scores and dP
// are recomputed in both backward kernels), in practice bounded by the softmax VALU work (~10 VALU slots per score) --
// and, at these lengths, nearer to an HBM floor than to either: a 257-token head moves 2*64*N bytes per tensor pass against
// 4*N*N*64 flop, i.e. N / 2 = 128 flop per byte where the part needs ~ 470.  Forward: 4 passes (0.41 ms of the 0.75 measured
// at 1024 x 16 x 257), two-kernel backward: 13 passes (1.32 of 2.07 ms), one-kernel backward below: 8 passes (0.81 of 1.55 ms).
// Algorithmic HBM bytes per (b, h): fwd 2*64*(2Nq + 2Nk) + 4Nq.
#include "common.h"
#include <cstdlib>
#include <type_traits>

// Compiler-level fence: keeps hipcc from hoisting every LDS fragment read of a fully unrolled tile loop to the
// top (which costs > 256 VGPRs and spills); a few tiles' worth of reads stay in flight between fences.
#ifdef ANTMMF_EMULATE
#define LDS_FENCE() do {} while (0)
#define EXP2F(x) exp2f(x)
#define LOG2F(x) log2f(x)
#else
#define LDS_FENCE() asm volatile("" ::: "memory")
#define EXP2F(x) __builtin_amdgcn_exp2f(x)
#define LOG2F(x) __builtin_amdgcn_logf(x)
#endif
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define ATTN_THREADS 512  // 8 waves: the 16 full query / key tiles of the 257-token image tower in two even rounds (the 17th tile holds the one remaining token and costs
                          // one wave a third round); two workgroups = 16 waves per CU, every kernel <= 128 VGPRs.  Round 3, same-box A/B against 6 waves (18 / 17 tiles
                          // split 3-3-3-3-3-3, 12 waves per CU): forward 0.80 -> 0.73 ms, backward 2.30 -> 2.16 ms, 77-token tower unchanged (profiles/r3_attn_waves_ab.txt)

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const float* key_bias;
    bf16_t* o; float* lse;
    const bf16_t* d_o; bf16_t* dq; bf16_t* dk; bf16_t* dv;
    long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, heads, Nq, Nk;
    float scale;
    // attention-probability dropout (BertSelfAttention, modeling_bert.py:157): p_drop = 0 -> off.  The mask is a pure function of
    // (seed, ((b * heads + h) * Nq + q) * Nk + k), regenerated identically by the forward and both backward kernels.
    float drop_scale;       // 1 / (1 - p)
    uint32_t drop_thr;      // 0 = no dropout
    uint64_t drop_seed;
    // one-kernel backward only (antmmf_attention_bwd_sums): per batch item, the sums over the tokens of dQ | dK | dV -- what the q / k / v bias gradients are made of --
    // sums[(b * 3 + {0: q, 1: k, 2: v}) * heads * 64 + h * 64 + e], fp32; NULL = not wanted
    float* sums;
    int fwd_stagger;   // (lab experiment, see attn_fwd_kernel; 0 in the product)
    int sums_v;   // 0: the dV sums are not wanted (their slots stay unwritten) -- torchscale's value bias comes out of the inner LayerNorm's backward
};

// Head size DH = 64 (every tower of the contrastive path) or 128 (ViLBERT co-attention, vilbert.py:326-416 with bi_hidden_size 1024 / 8 heads):
// token tiles [tokens][DH] with 2 DH-byte rows; the 16-B slot index of a row is XOR-swizzled so that BOTH access patterns sweep all 64 banks:
// 16 consecutive rows x one slot (ds_read_b128 fragments) and 8 consecutive rows x one 32-B slot pair (transposing reads).
template <int DH>
__device__ __forceinline__ int attn_swz(int row) {
    return DH == 64 ? ((((row >> 1) & 3) << 1) | ((row >> 3) & 1))    // 128-B rows: two rows per bank sweep, 4 slot pairs
                    : (((row & 7) << 1) | ((row >> 3) & 1));          // 256-B rows: one row per bank sweep, 8 slot pairs
}

// rows [0, n_pad) x DH bf16 -> swizzled token tile; rows >= n_valid are zero.  NPAD is a compile-time bound on n_pad: every thread requests ALL of
// its 16-B vectors first and writes them to LDS afterwards.  (As a run-time loop -- one load, s_waitcnt vmcnt(0), one ds_write per trip -- staging K and
// V cost twelve SERIAL HBM round trips per workgroup: most of a workgroup's lifetime, which is what the PMC pass saw as 63 % wait cycles.)
template <int DH, int NPAD>
__device__ __forceinline__ void stage_rows(char* dst, const bf16_t* __restrict__ src, long ld, int n_valid, int n_pad) {
    constexpr int NS = DH / 8, TRIPS = (NPAD * NS + ATTN_THREADS - 1) / ATTN_THREADS;
    uint4 v[TRIPS];
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        const int id = threadIdx.x + i * ATTN_THREADS, row = id / NS, slot = id % NS;
        v[i] = (id < n_pad * NS && row < n_valid) ? *reinterpret_cast<const uint4*>(src + (long)row * ld + slot * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        const int id = threadIdx.x + i * ATTN_THREADS, row = id / NS, slot = id % NS;
        if (id < n_pad * NS) *reinterpret_cast<uint4*>(dst + row * (2 * DH) + ((slot ^ attn_swz<DH>(row)) << 4)) = v[i];
    }
}
// two tiles at once (K and V, or Q and dO): all loads of both in flight together
template <int DH, int NPAD>
__device__ __forceinline__ void stage_rows2(char* dst0, const bf16_t* __restrict__ src0, long ld0, char* dst1, const bf16_t* __restrict__ src1, long ld1,
                                            int n_valid, int n_pad) {
    constexpr int NS = DH / 8, TRIPS = (NPAD * NS + ATTN_THREADS - 1) / ATTN_THREADS;
    uint4 v0[TRIPS], v1[TRIPS];
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        const int id = threadIdx.x + i * ATTN_THREADS, row = id / NS, slot = id % NS;
        const bool ok = id < n_pad * NS && row < n_valid;
        v0[i] = ok ? *reinterpret_cast<const uint4*>(src0 + (long)row * ld0 + slot * 8) : make_uint4(0, 0, 0, 0);
        v1[i] = ok ? *reinterpret_cast<const uint4*>(src1 + (long)row * ld1 + slot * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        const int id = threadIdx.x + i * ATTN_THREADS, row = id / NS, slot = id % NS;
        if (id < n_pad * NS) {
            const int off = row * (2 * DH) + ((slot ^ attn_swz<DH>(row)) << 4);
            *reinterpret_cast<uint4*>(dst0 + off) = v0[i];
            *reinterpret_cast<uint4*>(dst1 + off) = v1[i];
        }
    }
}
// fragment "token `row`, head-dim 8 slot .. 8 slot + 7" (A or B operand with the head dim as the contraction)
template <int DH>
__device__ __forceinline__ bf16x8_t frag_rows(const char* tile, int row, int slot) {
    return *reinterpret_cast<const bf16x8_t*>(tile + row * (2 * DH) + ((slot ^ attn_swz<DH>(row)) << 4));
}
// A fragment "head-dim row 16 dt + l15, token slots of 32-token chunk c" (tokens {32c+4g..+3, 32c+16+4g..+3}): the
// transposed view of the row-major tile, two transposing reads
template <int DH>
__device__ __forceinline__ bf16x8_t frag_tokens(const char* tile, int c, int dt, int grp, int l15) {
    const int row = 32 * c + 4 * grp + (l15 >> 2);           // row this lane supplies to the 4 x 16 block
    const int slot = 2 * dt + ((l15 >> 1) & 1), sub = (l15 & 1) << 3;
    const bf16x4_t lo = lds_read_tr16(tile + row * (2 * DH) + ((slot ^ attn_swz<DH>(row)) << 4) + sub);
    const bf16x4_t hi = lds_read_tr16(tile + (row + 16) * (2 * DH) + ((slot ^ attn_swz<DH>(row + 16)) << 4) + sub);
    return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8_t pack_frag(const float (&lo)[4], const float (&hi)[4]) {
    union { uint4 u; bf16x8_t f; } cv;
    cv.u = make_uint4(pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3]));
    return cv.f;
}
__device__ __forceinline__ bf16x8_t load_frag_global(const bf16_t* p) {
    union { uint4 u; bf16x8_t f; } cv;
    cv.u = *reinterpret_cast<const uint4*>(p);
    return cv.f;
}
__device__ __forceinline__ float grp_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float grp_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
// key bias in the exp2 domain; padding keys get -inf
__device__ __forceinline__ void stage_key_bias(float* kb, const AttnArgs& a, int b, int n_pad) {
    for (int i = threadIdx.x; i < n_pad; i += ATTN_THREADS)
        kb[i] = i < a.Nk ? (a.key_bias ? a.key_bias[(long)b * a.Nk + i] * LOG2E : 0.f) : -INFINITY;
}

// ---- forward ----------------------------------------------------------------------------------------
// DROP: attention-probability dropout compiled in (a separate instantiation: the mask arithmetic in the inner loops costs registers
// -- with it folded in at run time the no-dropout backward dropped from 3 to 2 waves per SIMD and ran 1.7x slower)
// Row stores of a transposed output tile (lane (l15, grp): row l15, columns 16 dt + 4 grp + [0, 4) per 16-column tile dt: 8 bytes per tile): the pieces of an even / odd pair of
// column tiles are exchanged between the lanes l and l ^ 16 (lane_bit_exchange<4> = v_permlane16_swap), after which a lane holds 8 consecutive columns of ONE tile of the pair --
// one 16-byte store per pair instead of two 8-byte stores (half the store instructions, twice the bytes per row and pass).  Every lane of the wave must call it (cross-lane).
template <int NDT>
__device__ __forceinline__ void store_rows_paired(bf16_t* rowp, uint2 (&w)[NDT], int lane, int grp, bool valid, bool al16) {
#pragma unroll
    for (int k = 0; k < NDT / 2; ++k) {
        lane_bit_exchange<4>(w[2 * k].x, w[2 * k + 1].x, lane);
        lane_bit_exchange<4>(w[2 * k].y, w[2 * k + 1].y, lane);
        if (valid) {
            bf16_t* p = rowp + 16 * (2 * k + (grp & 1)) + 4 * (grp & 2);
            if (al16) *reinterpret_cast<uint4*>(p) = make_uint4(w[2 * k].x, w[2 * k].y, w[2 * k + 1].x, w[2 * k + 1].y);
            else { *reinterpret_cast<uint2*>(p) = w[2 * k]; *reinterpret_cast<uint2*>(p + 4) = w[2 * k + 1]; }
        }
    }
}

// ABL (lab library only, TIMING-ONLY, wrong results -- what each part of the forward costs, same box, same process; profiles/r6b_attn_fwd_ablations.txt): 1 the exponential
// replaced by its argument, 2 no O / lse stores, 4 K / V not fetched (zeros staged), 8 no P.V contraction, 16 no Q.K contraction, 32 Q not fetched
template <int NCH, bool DROP, int DH, int ABL = 0>  // keys padded to 32 * NCH
__global__ __launch_bounds__(ATTN_THREADS, DH == 64 ? 2 : 1) void attn_fwd_kernel(const AttnArgs a) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int NKP = 32 * NCH, NT = 2 * NCH, RB = 2 * DH, KM = DH / 32, DT = DH / 16;
    char* Ks = smem;
    char* Vs = smem + NKP * RB;
    float* kb = reinterpret_cast<float*>(Vs + NKP * RB);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, grp = lane >> 4;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
#if defined(ANTMMF_LAB) && !defined(ANTMMF_EMULATE)
    // LAB experiment (round 6, measured: no effect -- profiles/r6b_attn_fwd_stagger.txt).  Two workgroups share a CU, and each is "fetch K / V (nothing to compute), then
    // compute (nothing to fetch)"; the timing-only ablations (profiles/r6b_attn_fwd_ablations.txt: 0.41 ms with no memory traffic, 0.50 without the K / V fetch, 0.63 without
    // the stores, 0.72 - 0.73 with everything) read like "no overlap", and one explanation would be that the launch's first 2 x 256 workgroups start together, every item takes the
    // same time, and the two on a CU stay in phase.  So: the workgroup in the UPPER half of its CU's LDS (HW_REG_LDS_ALLOC: a non-zero base) starts ANTMMF_ATTN_FWD_STAGGER x 4 us
    // late, once.  0.73 - 0.75 ms for 0 ... 24 us: the pairs are not in lock step; what is exposed is that ONE workgroup computing alone (2 waves per SIMD of dependent
    // LDS -> MFMA -> VALU chains) does not run the CU at the rate two do, while its partner waits ~ 3.4 us per item for K / V.
    if (a.fwd_stagger > 0 && blockIdx.x < 512u) {
        const unsigned lds_alloc = __builtin_amdgcn_s_getreg((31 << 11) | 6);   // LDS_BASE in the low bits
        if (lds_alloc & 0xfffu) for (int i = 0; i < a.fwd_stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    stage_rows2<DH, NKP>(Ks, a.k + (long)b * a.Nk * a.ldk + h * DH, a.ldk, Vs, a.v + (long)b * a.Nk * a.ldv + h * DH, a.ldv, (ABL & 4) ? 0 : a.Nk, NKP);
    stage_key_bias(kb, a, b, NKP);
    __syncthreads();

    const float scale2 = a.scale * LOG2E;
    const int nqt = (a.Nq + 15) >> 4;
    const bool o_al16 = !(a.ldo & 7) && !((uintptr_t)a.o & 15);
    // the query fragments of a tile come straight from global memory: the NEXT tile's are requested before the current tile is
    // processed (PMC: 68 % of the wave cycles were spent parked on s_waitcnt with the loads issued at the point of use)
    auto q_ptr = [&](int qt) {
        const int qi = qt * 16 + l15;
        return a.q + ((long)b * a.Nq + (qi < a.Nq ? qi : a.Nq - 1)) * a.ldq + h * DH + grp * 8;
    };
    bf16x8_t nq[KM] = {};
    if (wave < nqt && !(ABL & 32)) {
        const bf16_t* p0 = q_ptr(wave);
#pragma unroll
        for (int j = 0; j < KM; ++j) nq[j] = load_frag_global(p0 + 32 * j);
    }
    for (int qt = wave; qt < nqt; qt += ATTN_THREADS / 64) {
        const int qi = qt * 16 + l15;
        const int qrow = qi < a.Nq ? qi : a.Nq - 1;
        bf16x8_t qf[KM];
#pragma unroll
        for (int j = 0; j < KM; ++j) qf[j] = nq[j];
        if (qt + ATTN_THREADS / 64 < nqt && !(ABL & 32)) {
            const bf16_t* p1 = q_ptr(qt + ATTN_THREADS / 64);
#pragma unroll
            for (int j = 0; j < KM; ++j) nq[j] = load_frag_global(p1 + 32 * j);
        }
        float s[NT][4];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t % 3 == 0) LDS_FENCE();
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < KM; ++j)
                if constexpr (!(ABL & 16)) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Ks, 16 * t + l15, 4 * j + grp), qf[j], acc, 0, 0, 0);
            const float4 bias = *reinterpret_cast<const float4*>(kb + 16 * t + 4 * grp);
            s[t][0] = acc[0] * scale2 + bias.x; s[t][1] = acc[1] * scale2 + bias.y;
            s[t][2] = acc[2] * scale2 + bias.z; s[t][3] = acc[3] * scale2 + bias.w;
            m = fmaxf(m, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        }
        m = grp_max(m);
        if (m == -INFINITY) m = 0.f;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[t][r] = (ABL & 1) ? (s[t][r] - m) : EXP2F(s[t][r] - m); sum += s[t][r]; }
        sum = grp_sum(sum);
        float inv = sum > 0.f ? fast_rcp(sum) : 0.f;  // applied to the 16 output values, not the Nk probabilities
        if (DROP) {  // drop probabilities (the softmax denominator keeps every key, as in the reference)
            const uint32_t base = (uint32_t)((((long)b * a.heads + h) * a.Nq + qrow) * a.Nk);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (!DROPOUT_KEEP(base + 16 * t + 4 * grp + r, a.drop_seed, a.drop_thr)) s[t][r] = 0.f;
            inv *= a.drop_scale;
        }
        bf16x8_t pf[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) pf[c] = pack_frag(s[2 * c], s[2 * c + 1]);
        uint2 ow[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            LDS_FENCE();
            f32x4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if constexpr (!(ABL & 8)) o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tokens<DH>(Vs, c, dt, grp, l15), pf[c], o, 0, 0, 0);
                else if (c == 0) { union { bf16x8_t f; uint4 u; } cv_; cv_.f = pf[dt]; o[0] = __uint_as_float(cv_.u.x); o[1] = __uint_as_float(cv_.u.y); o[2] = __uint_as_float(cv_.u.z); o[3] = __uint_as_float(cv_.u.w); }   // (keeps the probabilities alive)
            ow[dt] = make_uint2(pack_bf2(o[0] * inv, o[1] * inv), pack_bf2(o[2] * inv, o[3] * inv));
        }
        if constexpr (ABL & 2) {   // (one dword per wave and tile keeps every value alive)
            float keep_ = sum + m;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) keep_ += __uint_as_float(ow[dt].x) + __uint_as_float(ow[dt].y);
            if (keep_ == 1.2345f) a.lse[0] = keep_;
        } else {
        store_rows_paired<DT>(a.o + ((long)b * a.Nq + qrow) * a.ldo + h * DH, ow, lane, grp, qi < a.Nq, o_al16);
        if (grp == 0 && qi < a.Nq) a.lse[((long)b * a.heads + h) * a.Nq + qi] = sum > 0.f ? (m + LOG2F(sum)) * LN2 : -INFINITY;
        }
    }
}

// ---- forward on 32 x 32 x 16 MFMA tiles (head size 64, no dropout: the M2 / CLIP image towers and every tower in evaluation) --------------------------
// Same residency as the kernel above (one workgroup per (batch, head), K and V token tiles in LDS), different unit of work: a wave owns 32 queries and walks
// the keys in blocks of 64 with an online softmax.
//   * v_mfma_f32_32x32x16_bf16 (A: lane l holds A[l & 31][8 (l >> 5) + e], B: B[8 (l >> 5) + e][l & 31], D: D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31], r < 16).
//     Scores are computed transposed, S^T = K Q^T: a lane holds one query (column l & 31) and 16 keys of a 32-key tile, so the row max / sum are in-lane plus ONE
//     exchange with lane l ^ 32 (v_permlane32_swap: no LDS crossbar);
//   * a K fragment (one ds_read_b128) now feeds 32 x 32 x 16 MACs instead of 16 x 16 x 32: half the LDS bytes, half the LDS and MFMA instructions per flop,
//     half the shuffles per score of the 16 x 16 kernel -- its waves spent 48 % of their cycles parked on s_waitcnt and another 24 % on issue stalls
//     (profiles/r3_pmc_attn_summary.txt), i.e. on the NUMBER of dependent LDS -> MFMA round trips, not on any pipe's throughput;
//   * the probabilities a lane holds ARE its P.V B fragment: the 8 k-slots of lane group g = l >> 5 in the MFMA of key half h of a tile are DEFINED as keys
//     {16 h + 4 g + 0..3, 16 h + 8 + 4 g + 0..3} (score registers 8 h .. 8 h + 7), and the matching A fragment (V^T: 32 head-dim rows x those 16 keys) is two
//     transposing reads of the row-major V tile;
//   * online softmax over 64-key blocks (32 score registers, 32 output accumulators, 16 packed probabilities: ~125 VGPRs, 4 waves per SIMD as before); the
//     running maximum only ever rescales when some lane of the wave saw a larger score (wave-uniform branch).  exp2 domain, scale folded into one FMA.
// Tile bookkeeping: 257 queries = 8 full tiles + one tile with a single row; the extra tile goes to a different wave (hence SIMD) in every workgroup.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__device__ __forceinline__ bf16_t f2bf_hw(float v) { return (bf16_t)(pack_bf2(v, 0.f) & 0xffffu); }
#ifdef ANTMMF_EMULATE
__device__ __forceinline__ float lane32_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float lane32_sum(float v) { return v + __shfl_xor(v, 32, 64); }
#else
typedef __attribute__((ext_vector_type(2))) unsigned int attn_u2_t;
__device__ __forceinline__ float lane32_max(float v) {
    const attn_u2_t s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
}
__device__ __forceinline__ float lane32_sum(float v) {
    const attn_u2_t s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
#endif
// A fragment "head-dim rows 32 d2 + (l & 31), key slots of key half h of 32-key tile t" (keys {32 t + 16 h + 4 g + 0..3, + 8}): two transposing reads
__device__ __forceinline__ bf16x8_t frag_tokens32(const char* tile, int t, int h, int d2, int lane) {
    const int g = lane >> 5, sub16 = (lane >> 4) & 1, j = lane & 15;
    const int row = 32 * t + 16 * h + 4 * g + (j >> 2);
    const int slot = 4 * d2 + 2 * sub16 + ((j >> 1) & 1), sub = (j & 1) << 3;
    const bf16x4_t lo = lds_read_tr16(tile + row * 128 + ((slot ^ attn_swz<64>(row)) << 4) + sub);
    const bf16x4_t hi = lds_read_tr16(tile + (row + 8) * 128 + ((slot ^ attn_swz<64>(row + 8)) << 4) + sub);
    return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
template <int NT32, bool HASBIAS, int KB>  // keys padded to 32 * NT32; HASBIAS: an additive key bias exists (BERT / torchscale padding masks); KB: 32-key tiles per softmax block
__global__ __launch_bounds__(ATTN_THREADS, 4) void attn_fwd32_kernel(const AttnArgs a) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int NKP = 32 * NT32;
    char* Ks = smem;
    char* Vs = smem + NKP * 128;
    float* kb = reinterpret_cast<float*>(Vs + NKP * 128);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, g = lane >> 5;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    // K keeps the 16 x 16 kernels' swizzle (conflict-free for the 32-row ds_read_b128 fragments as well); V is only ever read by transposing reads of 4 rows x 64 B
    // per half-wave here, so its 16-B slot index is XORed with ((row >> 1) & 1) << 2: rows r and r + 2 of such a read land in different 64-B halves of the bank row
    {
        constexpr int TRIPS = (NKP * 8 + ATTN_THREADS - 1) / ATTN_THREADS;
        const bf16_t* ksrc = a.k + (long)b * a.Nk * a.ldk + h * 64;
        const bf16_t* vsrc = a.v + (long)b * a.Nk * a.ldv + h * 64;
        uint4 kv[TRIPS], vv[TRIPS];
#pragma unroll
        for (int i = 0; i < TRIPS; ++i) {
            const int id = threadIdx.x + i * ATTN_THREADS, row = id >> 3, slot = id & 7;
            const bool ok = id < NKP * 8 && row < a.Nk;
            kv[i] = ok ? *reinterpret_cast<const uint4*>(ksrc + (long)row * a.ldk + slot * 8) : make_uint4(0, 0, 0, 0);
            vv[i] = ok ? *reinterpret_cast<const uint4*>(vsrc + (long)row * a.ldv + slot * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TRIPS; ++i) {
            const int id = threadIdx.x + i * ATTN_THREADS, row = id >> 3, slot = id & 7;
            if (id < NKP * 8) {
                *reinterpret_cast<uint4*>(Ks + row * 128 + ((slot ^ attn_swz<64>(row)) << 4)) = kv[i];
                *reinterpret_cast<uint4*>(Vs + row * 128 + ((slot ^ (((row >> 1) & 1) << 2)) << 4)) = vv[i];
            }
        }
    }
    if (HASBIAS) stage_key_bias(kb, a, b, NKP);
    __syncthreads();

    const float scale2 = a.scale * LOG2E;
    // A last tile with one or two valid queries (257 tokens = 8 x 32 + CLS) is not given to ONE wave -- the workgroup would live two tile times for 9 / 8
    // of a tile's work per wave -- but split over the KEYS: every wave takes the key tiles wave, wave + 8, .. of it, the partial (max, sum, output) triples
    // meet in LDS and one wave merges them (the flash-decoding combine).
    const int nfull = a.Nq >> 5, nrem = a.Nq & 31;
    const bool split_last = nrem > 0 && nrem <= 2;
    const int nqt = split_last ? nfull : ((a.Nq + 31) >> 5);
    const int first = (wave + blockIdx.x) & 7;   // an extra whole tile (nqt = 8 k + 1, not split) lands on a different wave -- hence SIMD -- in every workgroup
    float* comb = kb + NKP;                      // [8 waves][2 columns][68]: 64 outputs, max, the two lane groups' sums
    // per-lane LDS byte offsets of the fragment reads inside tile 0; tile t adds 4096 t, key half h adds 2048 h (the swizzle reads row bits 1 - 3 only)
    int kofs[4], vofs_lo[2], vofs_hi[2];
    {
        const int sub16 = (lane >> 4) & 1, j = lane & 15, vrow = 4 * g + (j >> 2);
#pragma unroll
        for (int c = 0; c < 4; ++c) kofs[c] = l31 * 128 + (((2 * c + g) ^ attn_swz<64>(l31)) << 4);
#pragma unroll
        for (int d2 = 0; d2 < 2; ++d2) {
            const int slot = 4 * d2 + 2 * sub16 + ((j >> 1) & 1), sub = (j & 1) << 3;
            vofs_lo[d2] = vrow * 128 + ((slot ^ (((vrow >> 1) & 1) << 2)) << 4) + sub;
            vofs_hi[d2] = (vrow + 8) * 128 + ((slot ^ (((vrow >> 1) & 1) << 2)) << 4) + sub;
        }
    }
    auto vfrag = [&](int t, int hh, int d2) {
        const char* p = Vs + t * 4096 + hh * 2048;
        const bf16x4_t lo = lds_read_tr16(p + vofs_lo[d2]), hi = lds_read_tr16(p + vofs_hi[d2]);
        return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    // one 32-query tile against the key tiles [t_begin, t_end) step t_step: -> running (m, lsum, o0, o1) of this lane
    auto run_tile = [&](int qt, int t_begin, int t_end, int t_step, float& m, float& lsum, f32x16_t& o0, f32x16_t& o1) {
        const int qi = qt * 32 + l31;
        const int qrow = qi < a.Nq ? qi : a.Nq - 1;
        const bf16_t* qp = a.q + ((long)b * a.Nq + qrow) * a.ldq + h * 64 + g * 8;
        bf16x8_t qf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) qf[c] = load_frag_global(qp + 16 * c);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        m = -INFINITY; lsum = 0.f;   // running maximum (exp2 domain, of the scaled + biased scores) and this lane's share of the running sum
        // one block of NTB 32-key tiles starting at tile t0; LASTB: the block that holds the padding keys.  (The full blocks run as a rolled loop: unrolled,
        // hipcc keeps an address register per tile and the 9-tile kernel spills.)
        auto block = [&](int t0, auto ntb_c, auto last_c) {
            constexpr int NTB = decltype(ntb_c)::value;
            constexpr bool LASTB = decltype(last_c)::value;
            f32x16_t s[NTB];
#pragma unroll
            for (int tt = 0; tt < NTB; ++tt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[tt][r] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    s[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(Ks + (t0 + tt) * 4096 + kofs[c]), qf[c], s[tt], 0, 0, 0);
            }
            LDS_FENCE();
            // scaled (+ biased) scores; padding keys (>= Nk) exist only in the last tile of the last block
            float mb = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < NTB; ++tt) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int key0 = 32 * (t0 + tt) + 8 * r4 + 4 * g;
                    if (HASBIAS) {
                        const float4 bias = *reinterpret_cast<const float4*>(kb + key0);
                        s[tt][4 * r4 + 0] = s[tt][4 * r4 + 0] * scale2 + bias.x; s[tt][4 * r4 + 1] = s[tt][4 * r4 + 1] * scale2 + bias.y;
                        s[tt][4 * r4 + 2] = s[tt][4 * r4 + 2] * scale2 + bias.z; s[tt][4 * r4 + 3] = s[tt][4 * r4 + 3] * scale2 + bias.w;
                    } else if (LASTB && tt == NTB - 1) {   // (without a bias the scores stay unscaled: the scale is positive, so the maximum can be taken first and
#pragma unroll                                  //  scaled once, and exp2(s scale2 - m) is ONE fused multiply-add in front of the exponential)
                        for (int e = 0; e < 4; ++e) s[tt][4 * r4 + e] = key0 + e < a.Nk ? s[tt][4 * r4 + e] : -INFINITY;
                    }
                    mb = fmaxf(mb, fmaxf(fmaxf(s[tt][4 * r4], s[tt][4 * r4 + 1]), fmaxf(s[tt][4 * r4 + 2], s[tt][4 * r4 + 3])));
                }
            }
            if (!HASBIAS) mb *= scale2;
            mb = lane32_max(mb);
            if (__any(mb > m)) {   // some query of this wave saw a larger score: rescale (wave-uniform branch; always taken in the first block)
                const float mn = fmaxf(m, mb);
                const float alpha = mn == -INFINITY ? 1.f : EXP2F(m - mn);   // m = -inf (nothing seen yet): exp2(-inf) = 0 on zeros
                lsum *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
                m = mn;
            }
            const float msub = m == -INFINITY ? 0.f : m;   // fully masked so far: every score is -inf, exp2(-inf - 0) = 0
            bf16x8_t pf[NTB][2];
#pragma unroll
            for (int tt = 0; tt < NTB; ++tt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[tt][r] = EXP2F(HASBIAS ? s[tt][r] - msub : __builtin_fmaf(s[tt][r], scale2, -msub)); lsum += s[tt][r]; }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    union { uint4 u; bf16x8_t f; } cv;
                    cv.u = make_uint4(pack_bf2(s[tt][8 * hh], s[tt][8 * hh + 1]), pack_bf2(s[tt][8 * hh + 2], s[tt][8 * hh + 3]),
                                      pack_bf2(s[tt][8 * hh + 4], s[tt][8 * hh + 5]), pack_bf2(s[tt][8 * hh + 6], s[tt][8 * hh + 7]));
                    pf[tt][hh] = cv.f;
                }
            }
#pragma unroll
            for (int tt = 0; tt < NTB; ++tt)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(t0 + tt, hh, 0), pf[tt][hh], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(t0 + tt, hh, 1), pf[tt][hh], o1, 0, 0, 0);
                }
            LDS_FENCE();
        };
        if (t_step == 1) {   // all keys: full blocks, then the block with the padding keys
            constexpr int NFULL = (NT32 - 1) / KB;
#pragma unroll 1
            for (int kbi = 0; kbi < NFULL; ++kbi) block(KB * kbi, std::integral_constant<int, KB>{}, std::false_type{});
            block(KB * NFULL, std::integral_constant<int, NT32 - KB * NFULL>{}, std::true_type{});
        } else {             // a share of the key tiles (split last query tile)
#pragma unroll 1
            for (int t = t_begin; t < t_end; t += t_step) {
                if (t == NT32 - 1) block(t, std::integral_constant<int, 1>{}, std::true_type{});
                else block(t, std::integral_constant<int, 1>{}, std::false_type{});
            }
        }
    };

    for (int qt = first; qt < nqt; qt += ATTN_THREADS / 64) {
        float m, lsum;
        f32x16_t o0, o1;
        run_tile(qt, 0, NT32, 1, m, lsum, o0, o1);
        const int qi = qt * 32 + l31;
        const float sum = lane32_sum(lsum);
        const float inv = sum > 0.f ? fast_rcp(sum) : 0.f;
        if (qi < a.Nq) {
            bf16_t* op = a.o + ((long)b * a.Nq + qi) * a.ldo + h * 64 + 4 * g;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                *reinterpret_cast<uint2*>(op + 8 * r4) = make_uint2(pack_bf2(o0[4 * r4] * inv, o0[4 * r4 + 1] * inv), pack_bf2(o0[4 * r4 + 2] * inv, o0[4 * r4 + 3] * inv));
                *reinterpret_cast<uint2*>(op + 32 + 8 * r4) = make_uint2(pack_bf2(o1[4 * r4] * inv, o1[4 * r4 + 1] * inv), pack_bf2(o1[4 * r4 + 2] * inv, o1[4 * r4 + 3] * inv));
            }
            if (g == 0) a.lse[((long)b * a.heads + h) * a.Nq + qi] = sum > 0.f ? (m + LOG2F(sum)) * LN2 : -INFINITY;
        }
    }
    if (split_last) {
        float m, lsum;
        f32x16_t o0, o1;
        run_tile(nfull, wave, NT32, ATTN_THREADS / 64, m, lsum, o0, o1);
        if (l31 < nrem) {
            float* cw = comb + (wave * 2 + l31) * 68;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dh = (r & 3) + 8 * (r >> 2) + 4 * g;
                cw[dh] = o0[r]; cw[32 + dh] = o1[r];
            }
            if (g == 0) cw[64] = m;
            cw[65 + g] = lsum;
        }
        __syncthreads();
        if (wave < nrem) {   // wave c merges column c: lane = head-dim index
            const int c = wave;
            float M = -INFINITY;
#pragma unroll
            for (int w = 0; w < ATTN_THREADS / 64; ++w) M = fmaxf(M, comb[(w * 2 + c) * 68 + 64]);
            float L = 0.f, acc = 0.f;
#pragma unroll
            for (int w = 0; w < ATTN_THREADS / 64; ++w) {
                const float* cw = comb + (w * 2 + c) * 68;
                const float f = cw[64] == -INFINITY ? 0.f : EXP2F(cw[64] - M);
                L += (cw[65] + cw[66]) * f;
                acc += cw[lane] * f;
            }
            const int qi = nfull * 32 + c;
            const float inv = L > 0.f ? fast_rcp(L) : 0.f;
            a.o[((long)b * a.Nq + qi) * a.ldo + h * 64 + lane] = f2bf_hw(acc * inv);
            if (lane == 0) a.lse[((long)b * a.heads + h) * a.Nq + qi] = L > 0.f ? (M + LOG2F(L)) * LN2 : -INFINITY;
        }
    }
}

// ---- backward 1: dQ (one workgroup per (b, h); K and V token tiles in LDS; waves walk query tiles) ----
// Head size 64, written out: the same statements as the DH-templated kernel below, but hipcc's schedule of THIS form is the tuned one (the
// templated form compiles the 96-key variant to 108 VGPRs instead of 60 -- 4 instead of 8 waves per SIMD -- and the text tower's backward
// ran 10 % slower with it).
template <int NCH, bool DROP>
__global__ __launch_bounds__(ATTN_THREADS, 2) void attn_bwd_dq64_kernel(const AttnArgs a) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int NKP = 32 * NCH;
    char* Ks = smem;
    char* Vs = smem + NKP * 128;
    float* kb = reinterpret_cast<float*>(Vs + NKP * 128);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, grp = lane >> 4;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    stage_rows2<64, NKP>(Ks, a.k + (long)b * a.Nk * a.ldk + h * 64, a.ldk, Vs, a.v + (long)b * a.Nk * a.ldv + h * 64, a.ldv, a.Nk, NKP);
    stage_key_bias(kb, a, b, NKP);
    __syncthreads();

    const float scale2 = a.scale * LOG2E, inv_scale2 = 1.0f / scale2;
    const int nqt = (a.Nq + 15) >> 4;
    const bool dq_al16 = !(a.lddq & 7) && !((uintptr_t)a.dq & 15);
    // (unlike the forward / dK-dV kernels this one loads its tile operands at the point of use: with a one-tile-ahead prefetch hipcc
    // hoists the unrolled LDS fragment reads as well and lands at 256 VGPRs + scratch, i.e. 2 waves per SIMD -- measured slower)
    bf16x8_t nq0 = {}, nq1 = {}, nd0 = {}, nd1 = {};
    float nlse = 0.f;
#define DQ_FETCH(QT)                                                                                      \
    do {                                                                                                  \
        const int qi_ = (QT) * 16 + l15;                                                                  \
        const long tok_ = (long)b * a.Nq + (qi_ < a.Nq ? qi_ : a.Nq - 1);                                 \
        const bf16_t* qp_ = a.q + tok_ * a.ldq + h * 64 + grp * 8;                                        \
        const bf16_t* dop_ = a.d_o + tok_ * a.lddo + h * 64 + grp * 8;                                    \
        nq0 = load_frag_global(qp_); nq1 = load_frag_global(qp_ + 32);                                    \
        nd0 = load_frag_global(dop_); nd1 = load_frag_global(dop_ + 32);                                  \
        nlse = a.lse[((long)b * a.heads + h) * a.Nq + (qi_ < a.Nq ? qi_ : a.Nq - 1)];                     \
    } while (0)
    for (int qt = wave; qt < nqt; qt += ATTN_THREADS / 64) {
        const int qi = qt * 16 + l15;
        const int qrow = qi < a.Nq ? qi : a.Nq - 1;
        DQ_FETCH(qt);
        const bf16x8_t qf0 = nq0, qf1 = nq1, df0 = nd0, df1 = nd1;
        const float lse = nlse;
        // D_q = <dO[q], O[q]> : this lane covers dh 8g..8g+7 and 32+8g..32+8g+7.  (O is fetched here, not a tile ahead: prefetching it
        // as well pushes the kernel over the 3-waves-per-SIMD register budget; its latency hides behind the first score MFMAs.)
        const bf16_t* op = a.o + ((long)b * a.Nq + qrow) * a.ldo + h * 64 + grp * 8;
        float dsum = 0.f;
        {
            union { bf16x8_t f; uint4 u; } d0, d1;
            d0.f = df0; d1.f = df1;
            const uint4 o0 = *reinterpret_cast<const uint4*>(op), o1 = *reinterpret_cast<const uint4*>(op + 32);
            const uint32_t dw[8] = {d0.u.x, d0.u.y, d0.u.z, d0.u.w, d1.u.x, d1.u.y, d1.u.z, d1.u.w};
            const uint32_t ow[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += bf_lo(dw[e]) * bf_lo(ow[e]) + bf_hi(dw[e]) * bf_hi(ow[e]);
        }
        dsum = grp_sum(dsum);
        // fully masked query (lse = -inf): subtracting +inf makes every z = -inf and exp2(z) = 0 without per-element selects
        const float lse2 = lse == -INFINITY ? INFINITY : lse * LOG2E;
        // -lse and -D_q are what the score / dP accumulators START from (z = s scale2 + bias - lse2 becomes ONE fused multiply-add behind the MFMAs, dS = p dP'
        // one multiply): 6.5 instead of 8.5 VALU slots per score.  A fully masked query starts at -inf and stays there: p = 0.
        const float s_init = -lse2 * inv_scale2, d_init = DROP ? 0.f : -dsum;
        const uint32_t dbase = (uint32_t)((((long)b * a.heads + h) * a.Nq + qrow) * a.Nk);
        bf16x8_t dsf[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            LDS_FENCE();
            float ds[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int t = 2 * c + hh;
                if (16 * t >= a.Nk) {   // a key tile of pure padding (257 keys: the 18th): dS = 0, no score work
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[hh][r] = 0.f;
                    continue;
                }
                f32x4_t sa = {s_init, s_init, s_init, s_init}, da = {d_init, d_init, d_init, d_init};
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<64>(Ks, 16 * t + l15, grp), qf0, sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<64>(Ks, 16 * t + l15, 4 + grp), qf1, sa, 0, 0, 0);
                da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<64>(Vs, 16 * t + l15, grp), df0, da, 0, 0, 0);
                da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<64>(Vs, 16 * t + l15, 4 + grp), df1, da, 0, 0, 0);
                const float4 bias = *reinterpret_cast<const float4*>(kb + 16 * t + 4 * grp);
                const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = EXP2F(__builtin_fmaf(sa[r], scale2, bb[r]));  // masked key: bias = -inf -> p = 0
                    float dp = da[r];
                    if (DROP) dp = (DROPOUT_KEEP(dbase + 16 * t + 4 * grp + r, a.drop_seed, a.drop_thr) ? dp * a.drop_scale : 0.f) - dsum;
                    ds[hh][r] = p * dp;
                }
            }
            dsf[c] = pack_frag(ds[0], ds[1]);
        }
        // request the next tile's operands now: the register-hungry score phase is over, the dQ phase (36 MFMAs) covers the latency
        uint2 gw[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            LDS_FENCE();
            f32x4_t g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                g = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tokens<64>(Ks, c, dt, grp, l15), dsf[c], g, 0, 0, 0);
            gw[dt] = make_uint2(pack_bf2(g[0] * a.scale, g[1] * a.scale), pack_bf2(g[2] * a.scale, g[3] * a.scale));
        }
        store_rows_paired<4>(a.dq + ((long)b * a.Nq + qrow) * a.lddq + h * 64, gw, lane, grp, qi < a.Nq, dq_al16);
    }
}
#undef DQ_FETCH

// ---- backward 1, any head size (used for 128)
template <int NCH, bool DROP, int DH>
__global__ __launch_bounds__(ATTN_THREADS, DH == 64 ? 2 : 1) void attn_bwd_dq_kernel(const AttnArgs a) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int NKP = 32 * NCH, RB = 2 * DH, KM = DH / 32, DT = DH / 16;
    char* Ks = smem;
    char* Vs = smem + NKP * RB;
    float* kb = reinterpret_cast<float*>(Vs + NKP * RB);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, grp = lane >> 4;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    stage_rows2<DH, NKP>(Ks, a.k + (long)b * a.Nk * a.ldk + h * DH, a.ldk, Vs, a.v + (long)b * a.Nk * a.ldv + h * DH, a.ldv, a.Nk, NKP);
    stage_key_bias(kb, a, b, NKP);
    __syncthreads();

    const float scale2 = a.scale * LOG2E, inv_scale2 = 1.0f / scale2;
    const int nqt = (a.Nq + 15) >> 4;
    const bool dq_al16 = !(a.lddq & 7) && !((uintptr_t)a.dq & 15);
    // (unlike the forward / dK-dV kernels this one loads its tile operands at the point of use: with a one-tile-ahead prefetch hipcc
    // hoists the unrolled LDS fragment reads as well and lands at 256 VGPRs + scratch, i.e. 2 waves per SIMD -- measured slower)
    bf16x8_t nq[KM] = {}, nd[KM] = {};
    float nlse = 0.f;
#define DQ_FETCH(QT)                                                                                      \
    do {                                                                                                  \
        const int qi_ = (QT) * 16 + l15;                                                                  \
        const long tok_ = (long)b * a.Nq + (qi_ < a.Nq ? qi_ : a.Nq - 1);                                 \
        const bf16_t* qp_ = a.q + tok_ * a.ldq + h * DH + grp * 8;                                        \
        const bf16_t* dop_ = a.d_o + tok_ * a.lddo + h * DH + grp * 8;                                    \
        _Pragma("unroll") for (int j_ = 0; j_ < KM; ++j_) nq[j_] = load_frag_global(qp_ + 32 * j_);       \
        _Pragma("unroll") for (int j_ = 0; j_ < KM; ++j_) nd[j_] = load_frag_global(dop_ + 32 * j_);      \
        nlse = a.lse[((long)b * a.heads + h) * a.Nq + (qi_ < a.Nq ? qi_ : a.Nq - 1)];                     \
    } while (0)
    for (int qt = wave; qt < nqt; qt += ATTN_THREADS / 64) {
        const int qi = qt * 16 + l15;
        const int qrow = qi < a.Nq ? qi : a.Nq - 1;
        DQ_FETCH(qt);
        bf16x8_t qf[KM], df[KM];
#pragma unroll
        for (int j = 0; j < KM; ++j) { qf[j] = nq[j]; df[j] = nd[j]; }
        const float lse = nlse;
        // D_q = <dO[q], O[q]> : this lane covers dh 8g..8g+7 of every 32-wide slice.  (O is fetched here, not a tile ahead: prefetching it
        // as well pushes the kernel over the 3-waves-per-SIMD register budget; its latency hides behind the first score MFMAs.)
        const bf16_t* op = a.o + ((long)b * a.Nq + qrow) * a.ldo + h * DH + grp * 8;
        float dsum = 0.f;
        {
            uint32_t dw[4 * KM], ow[4 * KM];
#pragma unroll
            for (int j = 0; j < KM; ++j) {
                union { bf16x8_t f; uint4 u; } d0;
                d0.f = df[j];
                const uint4 o0 = *reinterpret_cast<const uint4*>(op + 32 * j);
                dw[4 * j] = d0.u.x; dw[4 * j + 1] = d0.u.y; dw[4 * j + 2] = d0.u.z; dw[4 * j + 3] = d0.u.w;
                ow[4 * j] = o0.x; ow[4 * j + 1] = o0.y; ow[4 * j + 2] = o0.z; ow[4 * j + 3] = o0.w;
            }
#pragma unroll
            for (int e = 0; e < 4 * KM; ++e) dsum += bf_lo(dw[e]) * bf_lo(ow[e]) + bf_hi(dw[e]) * bf_hi(ow[e]);
        }
        dsum = grp_sum(dsum);
        // fully masked query (lse = -inf): subtracting +inf makes every z = -inf and exp2(z) = 0 without per-element selects
        const float lse2 = lse == -INFINITY ? INFINITY : lse * LOG2E;
        const float s_init = -lse2 * inv_scale2, d_init = DROP ? 0.f : -dsum;   // (see attn_bwd_dq64_kernel)
        const uint32_t dbase = (uint32_t)((((long)b * a.heads + h) * a.Nq + qrow) * a.Nk);
        bf16x8_t dsf[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            LDS_FENCE();
            float ds[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int t = 2 * c + hh;
                if (16 * t >= a.Nk) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[hh][r] = 0.f;
                    continue;
                }
                f32x4_t sa = {s_init, s_init, s_init, s_init}, da = {d_init, d_init, d_init, d_init};
#pragma unroll
                for (int j = 0; j < KM; ++j) sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Ks, 16 * t + l15, 4 * j + grp), qf[j], sa, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < KM; ++j) da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Vs, 16 * t + l15, 4 * j + grp), df[j], da, 0, 0, 0);
                const float4 bias = *reinterpret_cast<const float4*>(kb + 16 * t + 4 * grp);
                const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = EXP2F(__builtin_fmaf(sa[r], scale2, bb[r]));  // masked key: bias = -inf -> p = 0
                    float dp = da[r];
                    if (DROP) dp = (DROPOUT_KEEP(dbase + 16 * t + 4 * grp + r, a.drop_seed, a.drop_thr) ? dp * a.drop_scale : 0.f) - dsum;
                    ds[hh][r] = p * dp;
                }
            }
            dsf[c] = pack_frag(ds[0], ds[1]);
        }
        uint2 gw[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            LDS_FENCE();
            f32x4_t g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                g = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tokens<DH>(Ks, c, dt, grp, l15), dsf[c], g, 0, 0, 0);
            gw[dt] = make_uint2(pack_bf2(g[0] * a.scale, g[1] * a.scale), pack_bf2(g[2] * a.scale, g[3] * a.scale));
        }
        store_rows_paired<DT>(a.dq + ((long)b * a.Nq + qrow) * a.lddq + h * DH, gw, lane, grp, qi < a.Nq, dq_al16);
    }
}

#undef DQ_FETCH

// ---- backward 2: dK, dV (one workgroup per (b, h); Q and dO token tiles, lse, D in LDS; waves own key tiles) ----
template <bool DROP, int DH>
__global__ __launch_bounds__(ATTN_THREADS, DH == 64 ? 2 : 1) void attn_bwd_dkv_kernel(const AttnArgs a, int NQP) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int RB = 2 * DH, KM = DH / 32, DT = DH / 16, NS = DH / 8;
    char* Qs = smem;
    char* Ds = Qs + NQP * RB;
    float* lse_s = reinterpret_cast<float*>(Ds + NQP * RB);
    float* dsum_s = lse_s + NQP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, grp = lane >> 4;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const bf16_t* qbase = a.q + (long)b * a.Nq * a.ldq + h * DH;
    const bf16_t* dobase = a.d_o + (long)b * a.Nq * a.lddo + h * DH;
    const bf16_t* obase = a.o + (long)b * a.Nq * a.ldo + h * DH;
    stage_rows2<DH, 288>(Qs, qbase, a.ldq, Ds, dobase, a.lddo, a.Nq, NQP);
    for (int i = threadIdx.x; i < NQP; i += ATTN_THREADS) {
        float d = 0.f, l = INFINITY;  // padding queries: p = 0
        if (i < a.Nq) {
            l = a.lse[((long)b * a.heads + h) * a.Nq + i];
            l = l == -INFINITY ? INFINITY : l / a.scale;  // lse in units of the raw score (what the score accumulators start from, negated); fully masked query: -inf -> p = 0
#pragma unroll
            for (int v = 0; v < NS; ++v) {
                float x[8], y[8];
                ld8<bf16_t>(dobase + (long)i * a.lddo + v * 8, x);
                ld8<bf16_t>(obase + (long)i * a.ldo + v * 8, y);
#pragma unroll
                for (int e = 0; e < 8; ++e) d += x[e] * y[e];
            }
        }
        lse_s[i] = -l;
        dsum_s[i] = -d;   // (both negated: accumulator start values)
    }
    __syncthreads();

    const float scale2 = a.scale * LOG2E;
    const int nkt = (a.Nk + 15) >> 4, nqc = NQP >> 5;
    const bool dkv_al16 = !(a.lddk & 7) && !(a.lddv & 7) && !((uintptr_t)a.dk & 15) && !((uintptr_t)a.dv & 15);
    // (K / V fragments are loaded at the point of use: a one-tile-ahead prefetch costs 34 VGPRs here -- 128 -> 162, one resident wave
    // per SIMD less -- and measured 11 % slower)
    for (int kt = wave; kt < nkt; kt += ATTN_THREADS / 64) {
        const int ki = kt * 16 + l15;
        const int krow = ki < a.Nk ? ki : a.Nk - 1;
        const bf16_t* kp = a.k + ((long)b * a.Nk + krow) * a.ldk + h * DH + grp * 8;
        const bf16_t* vp = a.v + ((long)b * a.Nk + krow) * a.ldv + h * DH + grp * 8;
        bf16x8_t kf[KM], vf[KM];
#pragma unroll
        for (int j = 0; j < KM; ++j) { kf[j] = load_frag_global(kp + 32 * j); vf[j] = load_frag_global(vp + 32 * j); }
        const float kbias = ki < a.Nk ? (a.key_bias ? a.key_bias[(long)b * a.Nk + ki] * LOG2E : 0.f) : -INFINITY;
        f32x4_t dk[DT], dv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
        for (int c = 0; c < nqc; ++c) {
            float p[2][4], ds[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int q0 = 32 * c + 16 * hh;
                if (q0 >= a.Nq) {   // a 16-query tile of pure padding (257 queries: the last half chunk): p = dS = 0, no score work
#pragma unroll
                    for (int r = 0; r < 4; ++r) { p[hh][r] = 0.f; ds[hh][r] = 0.f; }
                    continue;
                }
                // lane holds (key = l15, q = q0 + 4g + r); the accumulators start from -lse (in raw-score units) and -D_q of their queries: the exponent is one
                // fused multiply-add behind the MFMAs, dS one multiply (see attn_bwd_dq64_kernel)
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0 + 4 * grp);
                const float4 d4 = *reinterpret_cast<const float4*>(dsum_s + q0 + 4 * grp);
                const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
                f32x4_t sa = {l4.x, l4.y, l4.z, l4.w}, da = {0.f, 0.f, 0.f, 0.f};
                if (!DROP) da = (f32x4_t){d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < KM; ++j) sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Qs, q0 + l15, 4 * j + grp), kf[j], sa, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < KM; ++j) da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Ds, q0 + l15, 4 * j + grp), vf[j], da, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = EXP2F(__builtin_fmaf(sa[r], scale2, kbias));  // masked / padding key or query: -inf -> 0
                    float pd = pr, dp = da[r];
                    if (DROP) {
                        const int qq = q0 + 4 * grp + r;
                        const bool keep = DROPOUT_KEEP((uint32_t)((((long)b * a.heads + h) * a.Nq + (qq < a.Nq ? qq : a.Nq - 1)) * a.Nk) + krow, a.drop_seed, a.drop_thr);
                        pd = keep ? pr * a.drop_scale : 0.f;
                        dp = (keep ? dp * a.drop_scale : 0.f) + dd[r];
                    }
                    p[hh][r] = pd;                     // dV uses the dropped probabilities
                    ds[hh][r] = pr * dp;               // dS uses the softmax probabilities
                }
            }
            const bf16x8_t pf = pack_frag(p[0], p[1]), dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tokens<DH>(Ds, c, dt, grp, l15), pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tokens<DH>(Qs, c, dt, grp, l15), dsf, dk[dt], 0, 0, 0);
            }
        }
        {
            uint2 vw[DT], kw[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                vw[dt] = make_uint2(pack_bf2(dv[dt][0], dv[dt][1]), pack_bf2(dv[dt][2], dv[dt][3]));
                kw[dt] = make_uint2(pack_bf2(dk[dt][0] * a.scale, dk[dt][1] * a.scale), pack_bf2(dk[dt][2] * a.scale, dk[dt][3] * a.scale));
            }
            store_rows_paired<DT>(a.dv + ((long)b * a.Nk + krow) * a.lddv + h * DH, vw, lane, grp, ki < a.Nk, dkv_al16);
            store_rows_paired<DT>(a.dk + ((long)b * a.Nk + krow) * a.lddk + h * DH, kw, lane, grp, ki < a.Nk, dkv_al16);
        }
    }
}

// ---- backward in ONE kernel (head size 64, 32 < Nk <= 272, no dropout: every tower of the contrastive step) --------------------------------------------------
// Why: at these lengths the backward is closer to its HBM floor than to the MFMA peak -- the two kernels above read Q, K, V, dO, O TWICE and form the
// scores, dP and the exponentials twice (13 tensor passes, 14 flop units, 2 exponentials per score); one kernel needs 8 passes, 10 units, 1 exponential.
// How: a workgroup owns one (b, h) with Q, dO AND K token tiles resident in LDS.  Wave w owns the key tiles w and w + 8 (dK / dV accumulators in registers
// for the whole kernel, as in attn_bwd_dkv_kernel) and walks the queries in 32-query chunks; per chunk it forms S, dP, P, dS for its 32 keys ONCE, feeds
// dV / dK, and leaves dS^T (bf16) in a double-buffered LDS tile [key][32 queries].  After ONE workgroup barrier per chunk the eight waves each
// contract one 16 x 16 tile of that chunk's dQ^T (head-dim tile w & 3, query tile w >> 2) over ALL keys straight out of that tile (transposing LDS reads)
// and store it: dQ is never accumulated across waves -- no atomics, no fp32 dQ buffer, a fixed summation order (bit-reproducible).
// The 17th key tile of the 257-token towers (ONE valid key) belongs to no wave: before the main loop its score blocks are spread over the waves by chunk
// (P and dS^T for that tile go to LDS), then waves 0 - 3 contract dV and waves 4 - 7 dK of that tile, one head-dim tile each, over all queries -- again
// no cross-wave sums.  Its dS^T stays in LDS as the ninth key slice of every chunk's dQ contraction.
// LDS: (2 NQP + 288) x 128 B token tiles + 2 x 16 KB dS^T + 1 KB per chunk for the extra tile + 8 NQP B row statistics = 151 KB at 257 x 257: one
// workgroup (8 waves, up to 256 VGPRs each) per CU.
// dS^T tile: key rows of 64 B (32 queries), 8-B slot s = 4 * (query tile) + (4-query group), XOR-swizzled so that both the 8-B row writes of a score block
// (16 consecutive keys x one slot) and the transposing reads (8 consecutive keys x 4 slots of one query tile) sweep all banks.
__device__ __forceinline__ int ds_swz(int key) { return (((key >> 2) & 1) << 2) | (((key >> 3) & 1) << 1) | ((key >> 1) & 1); }
__device__ __forceinline__ int ds_off(int key, int s) { return key * 64 + ((s ^ ds_swz(key)) << 3); }
// token sums of dQ | dK | dV (AttnArgs::sums): per-wave partials [dK: 8 x 64 | dV: 8 x 64 | dQ: 8 x 16 | the extra key tile: 8 x 16] floats behind the row statistics
#define ATTN_SUMS_FLOATS 1280
#ifndef ATTN_SUMS_MIN_KEYS
#define ATTN_SUMS_MIN_KEYS 32
#endif

static inline size_t attn_fused_lds_bytes(int Nq, int Nk, bool early = false, bool sums = false) {
    const int nqp = ((Nq + 31) / 32) * 32;
    return (size_t)nqp * 256 * (early ? 2 : 1) + 288 * 128 + 2 * 16384 + (Nk > 256 ? (size_t)(nqp / 32) * 1024 : 0) + (size_t)nqp * 8 + (sums ? ATTN_SUMS_FLOATS * 4 : 0);
}
static inline bool attn_fused_ok(const AttnArgs& a) {
    // (every shape of the one-kernel backward is served.  With the per-chunk dQ sums in LDS the sums cost ~ 1 us per (b, h) item -- 15 % of the 77 - 86-token forms, more than the
    // column-sum pass they save -- and the short towers were excluded; with the dQ tiles summed in registers it is 0.025 ms of 0.41 at 1024 x 16 x 77 against a pass of 0.055:
    // profiles/r6b_attn_bwd_token_sums_ab.txt)
    if (a.sums && a.Nk <= ATTN_SUMS_MIN_KEYS) return false;
    return !a.drop_thr && a.Nk > 32 && a.Nk <= 272 && attn_fused_lds_bytes(a.Nq, a.Nk, false, a.sums != nullptr) <= 160 * 1024;   // (one or two key tiles: the two-kernel form)
}

// one score block: 16 queries (tile at q0) x the 16 keys of (kf, vf): p = softmax probabilities, ds = p * (dP - D).  Lane: key l15, queries q0 + 4 grp + r.
#define FUSED_SCORE_BLOCK(KF, VF, KBIAS, P, DSV)                                                              \
    do {                                                                                                      \
        f32x4_t sa_ = {l4.x, l4.y, l4.z, l4.w}, da_ = {d4.x, d4.y, d4.z, d4.w};                               \
        sa_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, (KF)[0], sa_, 0, 0, 0);                            \
        sa_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, (KF)[1], sa_, 0, 0, 0);                            \
        da_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa0, (VF)[0], da_, 0, 0, 0);                            \
        da_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa1, (VF)[1], da_, 0, 0, 0);                            \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                                    \
            const float pr_ = FUSED_EXP(__builtin_fmaf(sa_[r_], scale2, (KBIAS)));                            \
            (P)[r_] = pr_;                                                                                    \
            (DSV)[r_] = pr_ * da_[r_];                                                                        \
        }                                                                                                     \
    } while (0)

// LDS-DMA of one 1-KB piece (8 token rows x 128 B) of a swizzled token tile: lane i lands at piece base + 16 i = row 8 piece + (i >> 3), physical slot i & 7, so it
// FETCHES the logical slot (i & 7) ^ swz(row) of that row.  Rows past the end re-read the last valid row (finite values: everything they feed is multiplied by a
// probability that is exactly 0).  The DMA is issued from inline asm on purpose: with the builtin hipcc puts `s_waitcnt vmcnt(0)` in front of every later ds_read
// while a piece is in flight (it cannot tell that the rows differ) -- the pieces requested here are read only after the next item's opening wait + barrier.
__device__ __forceinline__ void attn_dma_piece(char* tile, uint32_t tile_lds, const bf16_t* __restrict__ src, long ld, int n_valid, int piece, int lane) {
    const int row = 8 * piece + (lane >> 3), slot = (lane & 7) ^ attn_swz<64>(row);
    const bf16_t* g = src + (long)(row < n_valid ? row : n_valid - 1) * ld + slot * 8;
#ifdef ANTMMF_EMULATE
    (void)tile_lds;
    std::memcpy(tile + piece * 1024 + lane * 16, g, 16);
#else
    (void)tile;
    const uint32_t m = __builtin_amdgcn_readfirstlane(tile_lds + piece * 1024);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m), "v"(g) : "memory", "m0");   // (m0 is written here: the compiler must not keep a value of its own in it across the statement -- movrel, sendmsg, its own LDS-DMA builtins)
#endif
}

// one 32-query chunk C of attn_bwd_fused64_kernel.  NH = 2: both 16-query tiles hold queries; NH = 1: the second one is pure padding (no score work, zero dS^T rows).
// BEFORE / AFTER: statements run by every wave before / after the chunk's workgroup barrier (the persistent form's prefetch of the next item).
#define FUSED_CHUNK_SCORE(C, NH)                                                                                                           \
    do {                                                                                                                                    \
        char* sb_ = Sb + ((C) & 1) * 16384;                                                                                                 \
        if (tva) {   /* (a wave without a key tile -- fewer than eight tiles: the short towers -- only contracts its dQ tile) */               \
        float p_[2][2][4], ds_[2][2][4];   /* [tile][query tile][r] */                                                                      \
        _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                                                  \
            if (hh < (NH)) {                                                                                                                \
                const int q0 = 32 * (C) + 16 * hh;                                                                                          \
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0 + 4 * grp);                                                   \
                const float4 d4 = *reinterpret_cast<const float4*>(dsum_s + q0 + 4 * grp);                                                  \
                const bf16x8_t qa0 = frag_rows<64>(Qs, q0 + l15, grp), qa1 = frag_rows<64>(Qs, q0 + l15, 4 + grp);                          \
                const bf16x8_t oa0 = frag_rows<64>(Ds, q0 + l15, grp), oa1 = frag_rows<64>(Ds, q0 + l15, 4 + grp);                          \
                FUSED_SCORE_BLOCK(kf[0], vf[0], kbias[0], p_[0][hh], ds_[0][hh]);                                                           \
                if (tvb) FUSED_SCORE_BLOCK(kf[1], vf[1], kbias[1], p_[1][hh], ds_[1][hh]);                                                  \
                else { _Pragma("unroll") for (int r = 0; r < 4; ++r) { p_[1][hh][r] = 0.f; ds_[1][hh][r] = 0.f; } }                         \
            } else {                                                                                                                        \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                               \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) { p_[j][hh][r] = 0.f; ds_[j][hh][r] = 0.f; }                              \
            }                                                                                                                               \
        }                                                                                                                                   \
        bf16x8_t pf_[2], dsf_[2];                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                                     \
            pf_[j] = pack_frag(p_[j][0], p_[j][1]);                                                                                         \
            dsf_[j] = pack_frag(ds_[j][0], ds_[j][1]);                                                                                      \
            if (j == 0 || tvb) {                                                                                                            \
                union { bf16x8_t f; uint4 u; } dc_;                                                                                         \
                dc_.f = dsf_[j];                                                                                                            \
                const int key_ = kt[j] * 16 + l15;                                                                                          \
                *reinterpret_cast<uint2*>(sb_ + ds_off(key_, grp)) = make_uint2(dc_.u.x, dc_.u.y);                                          \
                *reinterpret_cast<uint2*>(sb_ + ds_off(key_, 4 + grp)) = make_uint2(dc_.u.z, dc_.u.w);                                      \
            }                                                                                                                               \
        }                                                                                                                                   \
        if constexpr (!(ABL & 8)) _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                        \
            const bf16x8_t fo_ = frag_tokens<64>(Ds, (C), dt, grp, l15), fq_ = frag_tokens<64>(Qs, (C), dt, grp, l15);                      \
            dv[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fo_, pf_[0], dv[0][dt], 0, 0, 0);                                           \
            dk[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq_, dsf_[0], dk[0][dt], 0, 0, 0);                                          \
            if (tvb) {                                                                                                                      \
                dv[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fo_, pf_[1], dv[1][dt], 0, 0, 0);                                       \
                dk[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq_, dsf_[1], dk[1][dt], 0, 0, 0);                                      \
            }                                                                                                                               \
        }                                                                                                                                   \
        }                                                                                                                                   \
    } while (0)
#define FUSED_CHUNK_DQ(C, NH)   /* this wave's 16 x 16 tile of chunk C's dQ^T, contracted over all keys out of the chunk's dS^T tile */              \
    do {                                                                                                                                    \
        char* sb_ = Sb + ((C) & 1) * 16384;                                                                                                 \
        if (!(ABL & 1) && qqt < (NH)) {                                                                                                     \
            f32x4_t g0_ = {0.f, 0.f, 0.f, 0.f}, g1_ = {0.f, 0.f, 0.f, 0.f};                                                                 \
            _Pragma("unroll") for (int kc = 0; kc < NKS; ++kc) {                                                                            \
                const bf16x4_t lo_ = lds_read_tr16(sb_ + ds_off(32 * kc + rr, s_q)), hi_ = lds_read_tr16(sb_ + ds_off(32 * kc + 16 + rr, s_q)); \
                const bf16x8_t bf_ = {lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]};                                      \
                if (kc & 1) g1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FUSED_KT(kc), bf_, g1_, 0, 0, 0);                                 \
                else g0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FUSED_KT(kc), bf_, g0_, 0, 0, 0);                                        \
            }                                                                                                                               \
            if constexpr (HASE) {                                                                                                           \
                const bf16x4_t lo_ = lds_read_tr16(Se + (C) * 1024 + ds_off(rr, s_q));                                                      \
                const bf16x8_t bf_ = {lo_[0], lo_[1], lo_[2], lo_[3], 0, 0, 0, 0};                                                          \
                g1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FUSED_KT(8), bf_, g1_, 0, 0, 0);                                              \
            }                                                                                                                               \
            const int qi_ = 32 * (C) + 16 * qqt + l15;                                                                                      \
            if (qi_ < a.Nq)                                                                                                                 \
                *reinterpret_cast<uint2*>(a.dq + ((long)b * a.Nq + qi_) * a.lddq + h * 64 + 16 * qdt + 4 * grp) =                           \
                    make_uint2(pack_bf2((g0_[0] + g1_[0]) * a.scale, (g0_[1] + g1_[1]) * a.scale),                                          \
                               pack_bf2((g0_[2] + g1_[2]) * a.scale, (g0_[3] + g1_[3]) * a.scale));                                         \
            if constexpr (SUMS && !(SUMS & 16)) FUSED_QSUM_ACC(g0_, g1_);   /* (padding queries: their dS^T rows are exactly 0) */                           \
        }                                                                                                                                   \
    } while (0)
#define FUSED_CHUNK(C, NH, BEFORE, AFTER)                                                                                                   \
    do {                                                                                                                                    \
        FUSED_CHUNK_SCORE(C, NH);                                                                                                           \
        BEFORE;                                                                                                                             \
        if constexpr (!(ABL & 2)) wg_barrier_lds_only();   /* every wave's dS^T rows of this chunk are in the tile; nobody reads this chunk's Q / dO rows any more */ \
        AFTER;                                                                                                                              \
        FUSED_CHUNK_DQ(C, NH);                                                                                                              \
    } while (0)

// NKS: 32-key slices of the resident key tiles (compile time: the dQ contraction is straight-line code); HASE: a 17th key tile exists; FULL: all sixteen resident
// tiles exist (no per-wave validity tests); NQC: number of query chunks when known at compile time (the extra tile's contraction), 0 = run-time loop.
// PERSIST: one workgroup per CU walks the items (b, h) blockIdx.x, blockIdx.x + gridDim.x, ... and hides the NEXT item's loads behind this item's arithmetic --
// with 151 KB of LDS there is no second workgroup on the CU to do that (timing-only ablations of the non-persistent form, profiles/r5_attn_bwd_one_kernel_v1_*:
// 1.87 ms of which ~ 1.3 ms remain with the dQ / dV / dK contractions, the exponentials and the barriers all removed).  Q and dO of the next item arrive by LDS-DMA
// chunk by chunk INTO THE ROWS THE CURRENT ITEM HAS JUST FINISHED WITH (chunk c's rows are dead after the chunk's barrier), its K tile as soon as every wave holds
// its K^T fragments in registers (KREG below), and its O rows, lse, V fragments and key biases go into registers behind this item's dK / dV stores; D = rowsum(dO o O)
// is formed at the item boundary from the landed dO tile and those O rows.
// ABL (lab library only, TIMING-ONLY, wrong results): 1 no dQ contraction, 2 no per-chunk barrier, 4 the exponential replaced by its argument, 8 no dV / dK contraction,
// 16 no O rows (D = rowsum(dO o O) comes out as 0), 32 no extra key tile -- what each part of the kernel costs, same box, same process; and two A/Bs of kept / rejected
// forms: 128 K^T fragments NOT in registers, 256 the dQ contraction one chunk late (see the main loop).
// EARLY (persistent form of the short towers, where LDS and registers allow it): Q / dO tiles double-buffered and EVERYTHING of the next item -- K, Q, dO, the register
// prefetch -- requested at the START of the current one, so the item boundary waits for nothing.  (The long towers have neither the 74 KB nor the 45 VGPRs.)
// SUMS (round 6): the token sums of dQ | dK | dV of every item -> a.sums (the q / k / v bias gradients' column-sum passes over the three tensors disappear): the dK / dV
// accumulators are closed over their 16 key lanes at the end of the item, the dQ^T tiles chunk by chunk, per-wave partials in LDS, three waves add them up and store.
template <int NKS, bool HASE, bool FULL, int NQC, bool PERSIST, int ABL = 0, bool EARLY = false, int SUMS = 0>   // SUMS: 0 none, 1 dQ | dK | dV, 2 dQ | dK
__global__ __launch_bounds__(ATTN_THREADS, 1) void attn_bwd_fused64_kernel(const AttnArgs a, int NQP, int n_items) {
    ANTMMF_DYN_LDS(char, smem);
    const int nkt = (a.Nk + 15) >> 4, nkr = nkt < 16 ? nkt : 16, nqc = NQC ? NQC : NQP >> 5;   // key tiles, resident key tiles, query chunks
    const int qd_bytes = NQP * 256;   // a Q tile + a dO tile
    char* Ks = smem + (EARLY ? 2 : 1) * qd_bytes;
    char* Sb = Ks + 288 * 128;
    char* Se = Sb + 2 * 16384;
    float* stats = reinterpret_cast<float*>(Se + (HASE ? nqc * 1024 : 0));   // [lse | dsum][NQP], both negated: accumulator start values
    float* sums = stats + 2 * NQP;                                           // SUMS: ATTN_SUMS_FLOATS per-wave partial token sums
    (void)sums;
    const int lane_w = threadIdx.x & 63;
#ifdef ANTMMF_EMULATE
    const int wave = threadIdx.x >> 6;
    const uint32_t lds0 = 0;
#else
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: everything decided per wave below is a uniform branch, not an exec mask
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#endif
    const float scale2 = a.scale * LOG2E;
    const bool dkv_al16 = !(a.lddk & 7) && !(a.lddv & 7) && !((uintptr_t)a.dk & 15) && !((uintptr_t)a.dv & 15);
#define FUSED_EXP(x) ((ABL & 4) ? (x) : EXP2F(x))
    // ---- persistent form: requests for an item's operands
    // K tile: 36 pieces over the 8 waves
#define FUSED_DMA_K(ITEM)                                                                                                                    \
    do {                                                                                                                                    \
        const int b_ = (ITEM) / a.heads, h_ = (ITEM) % a.heads;                                                                             \
        for (int pc_ = wave; pc_ < (HASE ? 36 : 4 * NKS); pc_ += ATTN_THREADS / 64)   /* (the rows the contractions read) */                \
            attn_dma_piece(Ks, lds0 + (uint32_t)(Ks - smem), a.k + (long)b_ * a.Nk * a.ldk + h_ * 64, a.ldk, a.Nk, pc_, lane);              \
    } while (0)
    // chunk C of Q (waves 0 - 3) and dO (waves 4 - 7): one piece per wave
#define FUSED_DMA_QD(ITEM, C, QB)   /* QB: byte offset of the destination Q tile (its dO tile follows it) */                                   \
    do {                                                                                                                                    \
        const int b_ = (ITEM) / a.heads, h_ = (ITEM) % a.heads;                                                                             \
        if (wave < 4) attn_dma_piece(smem + (QB), lds0 + (uint32_t)(QB), a.q + (long)b_ * a.Nq * a.ldq + h_ * 64, a.ldq, a.Nq, 4 * (C) + wave, lane); \
        else attn_dma_piece(smem + (QB) + NQP * 128, lds0 + (uint32_t)((QB) + NQP * 128), a.d_o + (long)b_ * a.Nq * a.lddo + h_ * 64, a.lddo, a.Nq, 4 * (C) + wave - 4, lane); \
    } while (0)
    // what an item needs in REGISTERS, requested while the previous item's dK / dV stores drain: its O rows in the staging pattern of the token tiles (thread -> row id >> 3,
    // 16-B slot id & 7, id = thread + 512 i) for D = rowsum(dO o O) -- dO comes out of the landed LDS tile --, its lse row, the V fragments and key biases of this wave's tiles
    uint4 po[5];
    float pl = -INFINITY;
    bf16x8_t vfn[2][2] = {}, vfEn[2] = {};
    float kbn[2] = {0.f, 0.f}, kbEn = 0.f;
#define FUSED_PREFETCH_REGS(ITEM)                                                                                                            \
    do {                                                                                                                                    \
        const int b_ = (ITEM) / a.heads, h_ = (ITEM) % a.heads;                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) {                                                                                  \
            const int id_ = wave * 64 + lane + i_ * ATTN_THREADS, row_ = id_ >> 3;                                                          \
            po[i_] = make_uint4(0, 0, 0, 0);                                                                                                \
            if (!(ABL & 16) && row_ < a.Nq) po[i_] = *reinterpret_cast<const uint4*>(a.o + ((long)b_ * a.Nq + row_) * a.ldo + h_ * 64 + (id_ & 7) * 8); \
        }                                                                                                                                   \
        pl = -INFINITY;                                                                                                                     \
        if (wave * 64 + lane < a.Nq) pl = a.lse[((long)b_ * a.heads + h_) * a.Nq + wave * 64 + lane];                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < (NKS <= 4 ? 1 : 2); ++j_) {                                                                 \
            const int ki_ = (wave + 8 * j_) * 16 + (lane & 15), krow_ = ki_ < a.Nk ? ki_ : a.Nk - 1;                                        \
            const bf16_t* vp_ = a.v + ((long)b_ * a.Nk + krow_) * a.ldv + h_ * 64 + (lane >> 4) * 8;                                        \
            vfn[j_][0] = load_frag_global(vp_); vfn[j_][1] = load_frag_global(vp_ + 32);                                                    \
            kbn[j_] = ki_ < a.Nk ? (a.key_bias ? a.key_bias[(long)b_ * a.Nk + ki_] : 0.f) : -INFINITY;   /* (x log2 e at the point of use: no wait here) */ \
        }                                                                                                                                   \
        if constexpr (HASE) {                                                                                                               \
            const int ki_ = 256 + (lane & 15), krow_ = ki_ < a.Nk ? ki_ : a.Nk - 1;                                                         \
            const bf16_t* vp_ = a.v + ((long)b_ * a.Nk + krow_) * a.ldv + h_ * 64 + (lane >> 4) * 8;                                        \
            vfEn[0] = load_frag_global(vp_); vfEn[1] = load_frag_global(vp_ + 32);                                                          \
            kbEn = ki_ < a.Nk ? (a.key_bias ? a.key_bias[(long)b_ * a.Nk + ki_] : 0.f) : -INFINITY;                                         \
        }                                                                                                                                   \
    } while (0)

    // SUMS: the per-wave partials of item ITEM (parked in LDS at the end of that item, a workgroup barrier ago) -> its 3 x 64 token sums.  Wave 0: dQ, 1: dK, 2: dV;
    // lane = head-dim element (= [head-dim tile][element of the tile]).  Runs behind the NEXT item's opening barrier (which exists anyway) -- as a block of its own at
    // the end of the item, with a barrier of its own, it cost the short towers 10 - 16 % (0.46 -> 0.51 ms at 77 tokens: ~ 1 us per item of 6 us)
#define FUSED_SUMS_FINALIZE(ITEM)                                                                                                            \
    do {                                                                                                                                    \
        if (wave < ((SUMS & 3) == 1 ? 3 : 2)) {                                                                                                   \
            const int b_ = (ITEM) / a.heads, h_ = (ITEM) % a.heads;                                                                         \
            FUSED_OPAQUE_LANE(le_);                                                                                                         \
            float s_;                                                                                                                       \
            if (wave == 0) s_ = (sums[1024 + le_] + sums[1024 + 64 + le_]) * a.scale;   /* the two query tiles of a chunk: waves qdt and 4 + qdt */ \
            else {                                                                                                                          \
                const float* w_ = sums + (wave == 1 ? 0 : 512) + le_;                                                                       \
                s_ = ((w_[0] + w_[64]) + (w_[128] + w_[192])) + ((w_[256] + w_[320]) + (w_[384] + w_[448]));                                \
                if constexpr (HASE && !(ABL & 32)) s_ += sums[1152 + (wave == 1 ? 64 : 0) + le_];                                           \
                if (wave == 1) s_ *= a.scale;                                                                                               \
            }                                                                                                                               \
            a.sums[((long)b_ * 3 + wave) * (a.heads * 64) + h_ * 64 + le_] = s_;                                                            \
        }                                                                                                                                   \
    } while (0)
#ifdef ANTMMF_EMULATE
#define FUSED_OPAQUE_LANE(x) int x = lane_w
#else
#define FUSED_OPAQUE_LANE(x) int x = lane_w; asm volatile("" : "+v"(x))   /* (offsets derived from it are recomputed at the point of use, not kept alive from the top of the item) */
#endif
    int item = blockIdx.x, cur = 0;
    int pitem = -1;   // SUMS: the item whose partial sums sit in LDS
    if constexpr (PERSIST) {
        const int lane = lane_w;
        if (item < n_items) {
            FUSED_DMA_K(item);
            for (int c = 0; c < nqc; ++c) FUSED_DMA_QD(item, c, 0);
            FUSED_PREFETCH_REGS(item);
        }
    }
    do {   // (the non-persistent form is launched with one workgroup per item: no loop in its code)
    // every per-lane LDS offset below derives from this copy of the lane number, opaque per item: hoisted out of the ITEM loop they would all stay live across the
    // whole kernel (measured: 256 VGPRs + 312 B of scratch); recomputed per item they cost a few VALU instructions in 40 k cycles
    int lane_o = lane_w;
#ifndef ANTMMF_EMULATE
    if constexpr (PERSIST) asm volatile("" : "+v"(lane_o));
#endif
    const int lane = lane_o, l15 = lane & 15, grp = lane >> 4;
    const int b = item / a.heads, h = item % a.heads;
    const int nxt = item + gridDim.x;
    const bool has_next = PERSIST && nxt < n_items;
    float* lse_s = stats;
    float* dsum_s = lse_s + NQP;
    char* Qs = smem + (EARLY ? cur * qd_bytes : 0);
    char* Ds = Qs + NQP * 128;
    if constexpr (PERSIST) {
        glds_wait_all();         // this wave's pieces of the item, its register prefetch (and everything older) have landed ...
        wg_barrier_lds_only();   // ... and so have the other waves' pieces
        if constexpr (SUMS && !(SUMS & 64)) { if (pitem >= 0) FUSED_SUMS_FINALIZE(pitem); }   // (the previous item's; this item's first write of that region is behind the next barrier)
#pragma unroll
        for (int i = 0; i < 5; ++i) {   // D = rowsum(dO o O): eight lanes per row, dO from the LDS tile
            const int id = wave * 64 + lane + i * ATTN_THREADS, row = id >> 3 < NQP ? id >> 3 : NQP - 1, slot = id & 7;
            const uint4 dw4 = *reinterpret_cast<const uint4*>(Ds + row * 128 + ((slot ^ attn_swz<64>(row)) << 4));
            const uint32_t ow[4] = {po[i].x, po[i].y, po[i].z, po[i].w}, dw[4] = {dw4.x, dw4.y, dw4.z, dw4.w};
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) d += bf_lo(dw[e]) * bf_lo(ow[e]) + bf_hi(dw[e]) * bf_hi(ow[e]);
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            if (slot == 0 && (id >> 3) < NQP) dsum_s[id >> 3] = -d;
        }
        if (wave * 64 + lane < NQP) lse_s[wave * 64 + lane] = pl == -INFINITY ? -INFINITY : -(pl / a.scale);   // raw-score units, negated; fully masked / padding query: -inf -> p = 0
        wg_barrier_lds_only();
    } else {
        const bf16_t* qbase = a.q + (long)b * a.Nq * a.ldq + h * 64;
        const bf16_t* dobase = a.d_o + (long)b * a.Nq * a.lddo + h * 64;
        const bf16_t* obase = a.o + (long)b * a.Nq * a.ldo + h * 64;
        stage_rows2<64, 288>(Qs, qbase, a.ldq, Ds, dobase, a.lddo, a.Nq, NQP);
        stage_rows<64, 288>(Ks, a.k + (long)b * a.Nk * a.ldk + h * 64, a.ldk, a.Nk, 288);
        for (int i = threadIdx.x; i < NQP; i += ATTN_THREADS) {   // (as attn_bwd_dkv_kernel; padding queries: p = 0)
            float d = 0.f, l = INFINITY;
            if (i < a.Nq) {
                l = a.lse[((long)b * a.heads + h) * a.Nq + i];
                l = l == -INFINITY ? INFINITY : l / a.scale;
                if constexpr (!(ABL & 16)) {
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        float x[8], y[8];
                        ld8<bf16_t>(dobase + (long)i * a.lddo + v * 8, x);
                        ld8<bf16_t>(obase + (long)i * a.ldo + v * 8, y);
#pragma unroll
                        for (int e = 0; e < 8; ++e) d += x[e] * y[e];
                    }
                }
            }
            lse_s[i] = -l;
            dsum_s[i] = -d;
        }
        __syncthreads();
    }

    // ---- the key tile beyond the sixteen resident ones
    if constexpr (HASE && !(ABL & 32)) {
        char* Pe = Sb;   // (the dS^T double buffer is free until the main loop)
        const int ki = 256 + l15, krow = ki < a.Nk ? ki : a.Nk - 1;
        const bf16_t* vp = a.v + ((long)b * a.Nk + krow) * a.ldv + h * 64 + grp * 8;
        const bf16x8_t kfE[2] = {frag_rows<64>(Ks, 256 + l15, grp), frag_rows<64>(Ks, 256 + l15, 4 + grp)};
        bf16x8_t vfE[2];
        float kbE;
        if constexpr (PERSIST) { vfE[0] = vfEn[0]; vfE[1] = vfEn[1]; kbE = kbEn * LOG2E; }
        else {
            vfE[0] = load_frag_global(vp); vfE[1] = load_frag_global(vp + 32);
            kbE = ki < a.Nk ? (a.key_bias ? a.key_bias[(long)b * a.Nk + ki] * LOG2E : 0.f) : -INFINITY;
        }
        for (int c = wave; c < nqc; c += ATTN_THREADS / 64) {   // (query tiles of pure padding: lse_s = -inf -> p = dS = 0 out of the same code)
            float p[2][4], ds[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int q0 = 32 * c + 16 * hh;
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0 + 4 * grp);
                const float4 d4 = *reinterpret_cast<const float4*>(dsum_s + q0 + 4 * grp);
                const bf16x8_t qa0 = frag_rows<64>(Qs, q0 + l15, grp), qa1 = frag_rows<64>(Qs, q0 + l15, 4 + grp);
                const bf16x8_t oa0 = frag_rows<64>(Ds, q0 + l15, grp), oa1 = frag_rows<64>(Ds, q0 + l15, 4 + grp);
                FUSED_SCORE_BLOCK(kfE, vfE, kbE, p[hh], ds[hh]);
            }
            union { bf16x8_t f; uint4 u; } pc, dc;
            pc.f = pack_frag(p[0], p[1]);
            dc.f = pack_frag(ds[0], ds[1]);
            *reinterpret_cast<uint2*>(Pe + c * 1024 + ds_off(l15, grp)) = make_uint2(pc.u.x, pc.u.y);
            *reinterpret_cast<uint2*>(Pe + c * 1024 + ds_off(l15, 4 + grp)) = make_uint2(pc.u.z, pc.u.w);
            *reinterpret_cast<uint2*>(Se + c * 1024 + ds_off(l15, grp)) = make_uint2(dc.u.x, dc.u.y);
            *reinterpret_cast<uint2*>(Se + c * 1024 + ds_off(l15, 4 + grp)) = make_uint2(dc.u.z, dc.u.w);
        }
        wg_barrier_lds_only();
        {   // waves 0 - 3: dV of that tile, head-dim tile `wave`; waves 4 - 7: dK, head-dim tile `wave - 4`; the contraction runs over all queries
            const int dt = wave & 3;
            const bool is_k = wave >= 4;
            const char* tok = is_k ? Qs : Ds;
            const char* src = is_k ? Se : Pe;
            f32x4_t g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
#define FUSED_E_STEP(C, ACC)                                                                                         \
            do {                                                                                                      \
                union { bf16x8_t f; uint4 u; } bc_;                                                                   \
                const uint2 lo_ = *reinterpret_cast<const uint2*>(src + (C) * 1024 + ds_off(l15, grp));               \
                const uint2 hi_ = *reinterpret_cast<const uint2*>(src + (C) * 1024 + ds_off(l15, 4 + grp));           \
                bc_.u = make_uint4(lo_.x, lo_.y, hi_.x, hi_.y);                                                       \
                ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tokens<64>(tok, (C), dt, grp, l15), bc_.f, ACC, 0, 0, 0); \
            } while (0)
            if constexpr (NQC > 0) {
#pragma unroll
                for (int c = 0; c < NQC; ++c) { if (c & 1) FUSED_E_STEP(c, g1); else FUSED_E_STEP(c, g0); }
            } else {
                for (int c = 0; c < nqc; ++c) FUSED_E_STEP(c, g0);
            }
#undef FUSED_E_STEP
            const float sc = is_k ? a.scale : 1.0f;
            if (ki < a.Nk) {
                bf16_t* dst = (is_k ? a.dk + ((long)b * a.Nk + ki) * a.lddk : a.dv + ((long)b * a.Nk + ki) * a.lddv) + h * 64 + 16 * dt + 4 * grp;
                *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf2((g0[0] + g1[0]) * sc, (g0[1] + g1[1]) * sc), pack_bf2((g0[2] + g1[2]) * sc, (g0[3] + g1[3]) * sc));
            }
            if ((SUMS & 3) == 1 || ((SUMS & 3) == 2 && is_k)) {   // this tile's share of the token sums (padding keys: exactly 0); waves 0 - 3: dV, 4 - 7: dK (unscaled), head-dim tile dt
                const float4 es = make_float4(row16_sum(g0[0] + g1[0]), row16_sum(g0[1] + g1[1]), row16_sum(g0[2] + g1[2]), row16_sum(g0[3] + g1[3]));
                if (l15 == 0) *reinterpret_cast<float4*>(sums + 1152 + wave * 16 + 4 * grp) = es;
            }
        }
        wg_barrier_lds_only();
    }

    // ---- resident key tiles of this wave: `wave` (exists whenever this kernel runs: more than 128 keys) and `wave + 8`
    const int kt[2] = {wave, wave + 8};
    constexpr bool SMALL = NKS <= 4;   // at most eight key tiles: no wave owns a second one, and waves >= the tile count own none
    const bool tva = !SMALL || kt[0] < nkr, tvb = !SMALL && (FULL || kt[1] < nkr);
    bf16x8_t kf[2][2], vf[2][2];
    float kbias[2];
    f32x4_t dk[2][4], dv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ki = kt[j] * 16 + l15, krow = ki < a.Nk ? ki : a.Nk - 1;
        const bf16_t* vp = a.v + ((long)b * a.Nk + krow) * a.ldv + h * 64 + grp * 8;
        kf[j][0] = frag_rows<64>(Ks, ki, grp);      // (rows >= Nk: zero, or a copy of the last row in the persistent form -- their probabilities are 0 either way)
        kf[j][1] = frag_rows<64>(Ks, ki, 4 + grp);
        if constexpr (PERSIST) { vf[j][0] = vfn[j][0]; vf[j][1] = vfn[j][1]; kbias[j] = kbn[j] * LOG2E; }
        else {
            vf[j][0] = load_frag_global(vp);
            vf[j][1] = load_frag_global(vp + 32);
            kbias[j] = ki < a.Nk ? (a.key_bias ? a.key_bias[(long)b * a.Nk + ki] * LOG2E : 0.f) : -INFINITY;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[j][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[j][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    }
    {   // the missing half of the last 32-key slice: its dS^T rows are read by the dQ contraction and never written
        const int tz = SMALL ? kt[0] : kt[1];
        if (!FULL && !(SMALL ? tva : tvb) && tz < 2 * NKS) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                *reinterpret_cast<uint2*>(Sb + u * 16384 + ds_off(tz * 16 + l15, grp)) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(Sb + u * 16384 + ds_off(tz * 16 + l15, 4 + grp)) = make_uint2(0u, 0u);
            }
        }
    }
    const int qdt = wave & 3, qqt = wave >> 2;   // this wave's dQ^T tile of every chunk
    // SUMS: ... summed over the chunks (unscaled) in this wave's own 16 floats of LDS (four accumulator registers across the main loop are four too many: 256 VGPRs + 156 B
    // of scratch in the 257-token form): closed over the 16 query lanes per chunk, read-modify-write by one lane per head-dim group -- only this wave touches the slot
    f32x4_t qsum = {0.f, 0.f, 0.f, 0.f};
    (void)qsum;
#ifdef ANTMMF_EMULATE
#define FUSED_LDS_ADD(P, V) (*(P) += (V))
#else
#define FUSED_LDS_ADD(P, V) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(P), (V), 0, 0, false)
#endif
#define FUSED_QSUM_ACC(G0, G1) do { qsum += (G0) + (G1); } while (0)
    const int s_q = 4 * qqt + (l15 & 3), rr = 4 * grp + (l15 >> 2);
    // KREG: the K^T fragments of this wave's head-dim tile -- the same for every chunk -- in registers (36 VGPRs): 18 transposing reads less per chunk, and the K tile is
    // free as soon as every wave holds its fragments, so the NEXT item's K is requested here, under the whole main loop, instead of at the item boundary where nothing hides
    // it (1.69 -> 1.57 ms, profiles/r5_attn_bwd_one_kernel_k_in_registers_ab.txt; ABL bit 7 switches it OFF: the A/B)
    constexpr bool KREG = PERSIST && !(ABL & 128);
    bf16x8_t ktf[KREG ? 9 : 1];
    if constexpr (KREG) {
#pragma unroll
        for (int kc = 0; kc < NKS + (HASE ? 1 : 0); ++kc) ktf[kc] = frag_tokens<64>(Ks, kc < NKS ? kc : 8, qdt, grp, l15);
        wg_barrier_lds_only();
        if (has_next) FUSED_DMA_K(nxt);
    }
    if constexpr (EARLY) {
        static_assert(!EARLY || (PERSIST && !(ABL & 128)), "EARLY needs the persistent form with K^T in registers");
        if (has_next) {
            for (int c_ = 0; c_ < nqc; ++c_) FUSED_DMA_QD(nxt, c_, (cur ^ 1) * qd_bytes);   // (the other Q / dO buffer: last read before this item's opening barrier)
            FUSED_PREFETCH_REGS(nxt);                                                        // (this item's copies were consumed above)
        }
    }
#define FUSED_KT(KC) (KREG ? ktf[KREG ? ((KC) < NKS ? (KC) : NKS) : 0] : frag_tokens<64>(Ks, (KC), qdt, grp, l15))

    const int nqc2 = (a.Nq + 15) >> 5;   // chunks whose second query tile holds queries
    int c = 0;
    if constexpr (PERSIST) {
        // (measured and not kept, profiles/r5_attn_bwd_one_kernel_l2_warmup_ab.txt: one dword per 64-B half row of the NEXT item's K, V and O requested during chunk 2, so
        // that the K tile's DMA and the register prefetch at the item boundary find their lines in L2: 1.69 -> 1.80 ms)
#define FUSED_POST(C) do { if (!EARLY && has_next) FUSED_DMA_QD(nxt, (C), 0); } while (0)
        if constexpr (ABL & 256) {
            // A/B (lab): the dQ contraction of chunk c - 1 in the SAME barrier interval as the score work of chunk c (independent instruction streams for the scheduler:
            // MFMA + LDS reads of the one next to the VALU-heavy softmax of the other), still one barrier per chunk and two dS^T buffers.  Measured 1.56 -> 1.60 ms
            // (profiles/r5_attn_bwd_one_kernel_delayed_dq_ab.txt): not the product's order
            // (assumes at least one full chunk: the instantiation this A/B exists for has eight and one padded chunk)
            FUSED_CHUNK_SCORE(0, 2);
            wg_barrier_lds_only();
            FUSED_POST(0);
            for (c = 1; c < nqc2; ++c) {
                FUSED_CHUNK_SCORE(c, 2);
                FUSED_CHUNK_DQ(c - 1, 2);
                wg_barrier_lds_only();
                FUSED_POST(c);
            }
            if (c < nqc) {   // the padded chunk (at most one)
                FUSED_CHUNK_SCORE(c, 1);
                FUSED_CHUNK_DQ(c - 1, 2);
                wg_barrier_lds_only();
                FUSED_POST(c);
                FUSED_CHUNK_DQ(c, 1);
                ++c;
            } else {
                FUSED_CHUNK_DQ(c - 1, 2);
            }
        } else {
        for (; c < nqc2; ++c) FUSED_CHUNK(c, 2, (void)0, FUSED_POST(c));
        for (; c < nqc; ++c) FUSED_CHUNK(c, 1, (void)0, FUSED_POST(c));
        }
#undef FUSED_POST
        wg_barrier_lds_only();   // every wave is through its last dQ contraction: K and the dS^T buffers are free
        if (!KREG && has_next) FUSED_DMA_K(nxt);
        cur ^= 1;
    } else {
        for (; c < nqc2; ++c) FUSED_CHUNK(c, 2, (void)0, (void)0);
        for (; c < nqc; ++c) FUSED_CHUNK(c, 1, (void)0, (void)0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j == 0 ? !tva : !tvb) continue;
        const int ki = kt[j] * 16 + l15, krow = ki < a.Nk ? ki : a.Nk - 1;
        uint2 vw[4], kw[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            vw[dt] = make_uint2(pack_bf2(dv[j][dt][0], dv[j][dt][1]), pack_bf2(dv[j][dt][2], dv[j][dt][3]));
            kw[dt] = make_uint2(pack_bf2(dk[j][dt][0] * a.scale, dk[j][dt][1] * a.scale), pack_bf2(dk[j][dt][2] * a.scale, dk[j][dt][3] * a.scale));
        }
        store_rows_paired<4>(a.dv + ((long)b * a.Nk + krow) * a.lddv + h * 64, vw, lane, grp, ki < a.Nk, dkv_al16);
        store_rows_paired<4>(a.dk + ((long)b * a.Nk + krow) * a.lddk + h * 64, kw, lane, grp, ki < a.Nk, dkv_al16);
    }
    if constexpr (SUMS && !(SUMS & 32)) {
        // token sums of this wave's dK / dV tiles (lane: key l15, head-dim 16 dt + 4 grp + r; tiles the wave does not own are zero): closed over the 16 lanes of a DPP
        // row, parked per wave in LDS
        FUSED_OPAQUE_LANE(ln);
        float* sw = sums + wave * 64 + 4 * (ln >> 4);
        {
            const float4 q4 = make_float4(row16_sum(qsum[0]), row16_sum(qsum[1]), row16_sum(qsum[2]), row16_sum(qsum[3]));
            if ((ln & 15) == 0) *reinterpret_cast<float4*>(sums + 1024 + wave * 16 + 4 * (ln >> 4)) = q4;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4_t ks = dk[0][dt] + dk[1][dt];
            const float4 k4 = make_float4(row16_sum(ks[0]), row16_sum(ks[1]), row16_sum(ks[2]), row16_sum(ks[3]));
            if ((ln & 15) == 0) *reinterpret_cast<float4*>(sw + 16 * dt) = k4;
            if constexpr ((SUMS & 3) == 1) {
                const f32x4_t vs = dv[0][dt] + dv[1][dt];
                const float4 v4 = make_float4(row16_sum(vs[0]), row16_sum(vs[1]), row16_sum(vs[2]), row16_sum(vs[3]));
                if ((ln & 15) == 0) *reinterpret_cast<float4*>(sw + 512 + 16 * dt) = v4;
            }
        }
    }
    if constexpr (PERSIST && !EARLY) { if (has_next) FUSED_PREFETCH_REGS(nxt); }   // (behind the stores: the accumulators' registers are free now; all of it lands under the K tile's latency)
    pitem = item;
    } while (PERSIST && (item += gridDim.x) < n_items);
    if constexpr (SUMS) {   // the last item's
        wg_barrier_lds_only();
        FUSED_SUMS_FINALIZE(pitem);
    }
#undef FUSED_SUMS_FINALIZE
#undef FUSED_KT
#undef FUSED_QSUM_ACC
#undef FUSED_LDS_ADD
#undef FUSED_OPAQUE_LANE
#undef FUSED_DMA_K
#undef FUSED_DMA_QD
#undef FUSED_PREFETCH_REGS
}
#undef FUSED_CHUNK
#undef FUSED_CHUNK_SCORE
#undef FUSED_CHUNK_DQ
#undef FUSED_SCORE_BLOCK
#undef FUSED_EXP

// ---- C ABI --------------------------------------------------------------------------------------------
static bool attn_args_ok(const AttnArgs& a, int dh) {
    return (dh == 64 || dh == 128) && a.B > 0 && a.heads > 0 && a.Nq > 0 && a.Nk > 0 && a.Nk <= 288 && a.Nq <= 288 && !(a.ldq & 7) && !(a.ldk & 7) &&
           !(a.ldv & 7) && !(a.ldo & 3);
}
template <typename K>
static void set_lds(K kern, size_t bytes) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); }

template <int DH>
static int attn_fwd_launch(const AttnArgs& a, hipStream_t stream) {
    const int nch = (a.Nk + 31) / 32;
    const dim3 grid((unsigned)(a.B * a.heads)), block(ATTN_THREADS);
#define FWD(N) do { const size_t lds = (size_t)(32 * N) * (4 * DH) + (32 * N) * 4; \
        if (a.drop_thr) { set_lds(attn_fwd_kernel<N, true, DH>, lds); hipLaunchKernelGGL((attn_fwd_kernel<N, true, DH>), grid, block, lds, stream, a); } \
        else { set_lds(attn_fwd_kernel<N, false, DH>, lds); hipLaunchKernelGGL((attn_fwd_kernel<N, false, DH>), grid, block, lds, stream, a); } } while (0)
    // attn_fwd32_kernel (32 x 32 x 16 tiles, online softmax) is an OPT-IN experiment: ANTMMF_ATTN_VARIANT bit 2 selects it for the long rows without dropout, bit 1 its
    // 64-key softmax blocks.  Measured same-box against the 16 x 16 whole-row kernel at 1024 x 16 heads x 257 tokens (profiles/r4_attn_fwd32_ab.txt): 0.78 - 0.81 ms vs
    // 0.72 - 0.73 ms -- half the LDS instructions and bytes, fewer parked cycles (34 % vs 48 %), but more issue stalls behind the 16-pass MFMAs (35 % vs 24 %) and the same
    // VALU work per score (one 16-cycle v_exp_f32 per score is as long as all the MFMAs of a tile): not faster, so not the default.
#ifdef ANTMMF_LAB
    static const char* av_env = ANTMMF_LAB_ENV("ANTMMF_ATTN_VARIANT");
    const int av = av_env ? atoi(av_env) : 0;
    if constexpr (DH == 64) {
        if (!a.drop_thr && (nch == 7 || nch == 9) && (av & 4)) {   // (its padding mask covers the LAST 32-key tile only: the key count must fill the instantiated tile count)
#define FWD32K(N, HB, KB) do { const size_t lds = (size_t)(32 * N) * 256 + (32 * N) * 4 + 8 * 2 * 68 * 4; set_lds(attn_fwd32_kernel<N, HB, KB>, lds); \
            hipLaunchKernelGGL((attn_fwd32_kernel<N, HB, KB>), grid, block, lds, stream, a); } while (0)
#define FWD32(N, HB) do { if (av & 2) FWD32K(N, HB, 2); else FWD32K(N, HB, 1); } while (0)   /* variant bit 1: 64-key softmax blocks */
            if (nch == 7) { if (a.key_bias) FWD32(7, true); else FWD32(7, false); }
            else { if (a.key_bias) FWD32(9, true); else FWD32(9, false); }
#undef FWD32
#undef FWD32K
            return antmmf_check_launch();
        }
    }
#endif
#ifdef ANTMMF_LAB
    if constexpr (DH == 64) {
        static const char* fa_env = ANTMMF_LAB_ENV("ANTMMF_ATTN_FWD_ABL");   // timing-only ablations of the 257-token forward (wrong results)
        const int fa = fa_env ? atoi(fa_env) : 0;
        if (fa && nch == 9 && !a.drop_thr) {
            const size_t lds = (size_t)(32 * 9) * (4 * DH) + (32 * 9) * 4;
#define FWDA(A) case A: set_lds(attn_fwd_kernel<9, false, 64, A>, lds); hipLaunchKernelGGL((attn_fwd_kernel<9, false, 64, A>), grid, block, lds, stream, a); return antmmf_check_launch();
            switch (fa) { FWDA(1) FWDA(2) FWDA(4) FWDA(8) FWDA(16) FWDA(32) FWDA(24) FWDA(25) FWDA(38) FWDA(63) FWDA(27) default: break; }
#undef FWDA
        }
    }
#endif
    if (nch <= 1) FWD(1); else if (nch <= 3) FWD(3); else if (nch <= 7) FWD(7); else FWD(9);
#undef FWD
    return antmmf_check_launch();
}

#ifdef ANTMMF_LAB
static long g_attn_fused_launches = 0;   // backward calls served by the one-kernel path (tests: "that kernel really ran")
extern "C" long antmmf_debug_attn_fused_launches() { return g_attn_fused_launches; }
#endif
template <int DH>
static int attn_bwd_launch(const AttnArgs& a, hipStream_t stream) {
    const int nch = (a.Nk + 31) / 32;
    const dim3 grid((unsigned)(a.B * a.heads)), block(ATTN_THREADS);
    if constexpr (DH == 64) {
        bool fused = attn_fused_ok(a);
#ifdef ANTMMF_LAB
        static const char* bv_env = ANTMMF_LAB_ENV("ANTMMF_ATTN_VARIANT");   // bit 3: the two-kernel backward everywhere (same-box A/B)
        if (bv_env && (atoi(bv_env) & 8) && !a.sums) fused = false;
#endif
        if (fused) {
#ifdef ANTMMF_LAB
            ++g_attn_fused_launches;
#endif
            const int nqp = ((a.Nq + 31) / 32) * 32, nkt = (a.Nk + 15) / 16, n_items = a.B * a.heads;
            // all sixteen resident key tiles present (the 197- / 257-token towers): the persistent form, one workgroup per CU walking the items
            const bool sm = a.sums != nullptr;
            const bool persist = (nkt <= 16 || nqp == 288) && attn_fused_lds_bytes(a.Nq, a.Nk, false, sm) <= 160 * 1024;   // (a 17th key tile with other query counts: run-time chunk loops, spills in the persistent form)
            const bool early = persist && nkt <= 8 && attn_fused_lds_bytes(a.Nq, a.Nk, true, sm) <= 160 * 1024;
            const size_t lds = attn_fused_lds_bytes(a.Nq, a.Nk, early, sm);
            unsigned pwgs = 256u;
#ifdef ANTMMF_LAB
            static const char* pw_env = ANTMMF_LAB_ENV("ANTMMF_ATTN_PERSIST_WGS");   // lab / emulator tests: a small grid makes every workgroup walk several items
            if (pw_env) pwgs = (unsigned)atoi(pw_env);
#endif
            const dim3 pgrid(persist ? ((unsigned)n_items < pwgs ? (unsigned)n_items : pwgs) : (unsigned)n_items);
#define BWDF_(NKS, HASE, FULL, NQC, PERS, ABL, EARLY, SUMS) do { set_lds(attn_bwd_fused64_kernel<NKS, HASE, FULL, NQC, PERS, ABL, EARLY, SUMS>, lds); \
            hipLaunchKernelGGL((attn_bwd_fused64_kernel<NKS, HASE, FULL, NQC, PERS, ABL, EARLY, SUMS>), pgrid, block, lds, stream, a, nqp, n_items); } while (0)
#define BWDF(NKS, HASE, FULL, NQC, PERS, ABL) do { if (sm && a.sums_v) BWDF_(NKS, HASE, FULL, NQC, PERS, 0, false, 1); else if (sm) BWDF_(NKS, HASE, FULL, NQC, PERS, 0, false, 2); \
                                                   else BWDF_(NKS, HASE, FULL, NQC, PERS, ABL, false, 0); } while (0)
#ifdef ANTMMF_LAB
            static const char* abl_env = ANTMMF_LAB_ENV("ANTMMF_ATTN_FUSED_ABL");   // timing-only ablations of the 257-token kernel (wrong results)
            const int abl = abl_env ? atoi(abl_env) : 0;
            if (abl && !sm && persist && nkt > 16 && nqp == 288) {
                switch (abl) {
                    case 1: BWDF(8, true, true, 9, true, 1); return antmmf_check_launch();
                    case 2: BWDF(8, true, true, 9, true, 2); return antmmf_check_launch();
                    case 4: BWDF(8, true, true, 9, true, 4); return antmmf_check_launch();
                    case 8: BWDF(8, true, true, 9, true, 8); return antmmf_check_launch();
                    case 16: BWDF(8, true, true, 9, true, 16); return antmmf_check_launch();
                    case 32: BWDF(8, true, true, 9, true, 32); return antmmf_check_launch();
                    case 15: BWDF(8, true, true, 9, true, 15); return antmmf_check_launch();
                    case 128: BWDF(8, true, true, 9, true, 128); return antmmf_check_launch();
                    case 256: BWDF(8, true, true, 9, true, 256); return antmmf_check_launch();
                    default: break;
                }
            }
#endif
#ifdef ANTMMF_LAB
            static const char* sa_env = ANTMMF_LAB_ENV("ANTMMF_ATTN_SUMS_ABL");   // timing-only (wrong sums): 16 no per-chunk dQ sums, 32 no end-of-item partials, 64 no finalisation
            const int sa = sa_env ? atoi(sa_env) : 0;
            if (sa && sm && persist && nkt > 16) {
                switch (sa) {
                    case 16: BWDF_(8, true, true, 9, true, 0, false, 2 | 16); return antmmf_check_launch();
                    case 32: BWDF_(8, true, true, 9, true, 0, false, 2 | 32); return antmmf_check_launch();
                    case 64: BWDF_(8, true, true, 9, true, 0, false, 2 | 64); return antmmf_check_launch();
                    case 48: BWDF_(8, true, true, 9, true, 0, false, 2 | 48); return antmmf_check_launch();
                    case 112: BWDF_(8, true, true, 9, true, 0, false, 2 | 112); return antmmf_check_launch();
                    default: break;
                }
            }
#endif
            if (persist) {
                if (nkt > 16) BWDF(8, true, true, 9, true, 0);
                else if (nkt == 16) BWDF(8, false, true, 0, true, 0);
                else if (nkt > 12) { if (nkt > 14) BWDF(8, false, false, 0, true, 0); else BWDF(7, false, false, 0, true, 0); }
                else if (nkt > 8) { if (nkt > 10) BWDF(6, false, false, 0, true, 0); else BWDF(5, false, false, 0, true, 0); }
                else if (early) {
#define BWDE(NKS) do { if (sm && a.sums_v) BWDF_(NKS, false, false, 0, true, 0, true, 1); else if (sm) BWDF_(NKS, false, false, 0, true, 0, true, 2); else BWDF_(NKS, false, false, 0, true, 0, true, 0); } while (0)
                    if (nkt > 6) BWDE(4); else if (nkt > 4) BWDE(3); else BWDE(2);
#undef BWDE
                }
                else if (nkt > 6) BWDF(4, false, false, 0, true, 0);
                else if (nkt > 4) BWDF(3, false, false, 0, true, 0);
                else BWDF(2, false, false, 0, true, 0);
            } else BWDF(8, true, true, 0, false, 0);
#undef BWDF
#undef BWDF_
            return antmmf_check_launch();
        }
    }
#define BWDQ_K(KERN, LDS) do { set_lds(KERN, LDS); hipLaunchKernelGGL(KERN, grid, block, LDS, stream, a); } while (0)
#define BWDQ(N) do { const size_t lds = (size_t)(32 * N) * (4 * DH) + (32 * N) * 4; \
        if constexpr (DH == 64) { if (a.drop_thr) BWDQ_K((attn_bwd_dq64_kernel<N, true>), lds); else BWDQ_K((attn_bwd_dq64_kernel<N, false>), lds); } \
        else { if (a.drop_thr) BWDQ_K((attn_bwd_dq_kernel<N, true, DH>), lds); else BWDQ_K((attn_bwd_dq_kernel<N, false, DH>), lds); } } while (0)
    if (nch <= 1) BWDQ(1); else if (nch <= 3) BWDQ(3); else if (nch <= 7) BWDQ(7); else BWDQ(9);
#undef BWDQ
#undef BWDQ_K
    int rc = antmmf_check_launch();
    if (rc) return rc;
    const int nqp = ((a.Nq + 31) / 32) * 32;
    const size_t lds2 = (size_t)nqp * (4 * DH) + (size_t)nqp * 8;
    if (a.drop_thr) { set_lds(attn_bwd_dkv_kernel<true, DH>, lds2); hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, DH>), grid, block, lds2, stream, a, nqp); }
    else { set_lds(attn_bwd_dkv_kernel<false, DH>, lds2); hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, DH>), grid, block, lds2, stream, a, nqp); }
    return antmmf_check_launch();
}

extern "C" int antmmf_attention_fwd_hd(const void* q, const void* k, const void* v, const float* key_bias, void* o, float* lse,
                                       int B, int heads, int head_dim, int Nq, int Nk, long ldq, long ldk, long ldv, long ldo, float scale,
                                       float dropout_p, uint64_t dropout_seed, hipStream_t stream) {
    AttnArgs a{};
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return ANTMMF_EINVAL;
    a.drop_thr = dropout_threshold(dropout_p); a.drop_scale = 1.0f / (1.0f - dropout_p); a.drop_seed = dropout_seed;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.key_bias = key_bias; a.o = (bf16_t*)o; a.lse = lse;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.B = B; a.heads = heads; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
    if (!q || !k || !v || !o || !lse || !attn_args_ok(a, head_dim)) return ANTMMF_EINVAL;
    a.fwd_stagger = 0;
#ifdef ANTMMF_LAB
    static const char* st_env = ANTMMF_LAB_ENV("ANTMMF_ATTN_FWD_STAGGER");
    if (st_env) a.fwd_stagger = atoi(st_env);
#endif
    return head_dim == 64 ? attn_fwd_launch<64>(a, stream) : attn_fwd_launch<128>(a, stream);
}

extern "C" int antmmf_attention_bwd_hd(const void* q, const void* k, const void* v, const float* key_bias, const void* o, const float* lse,
                                       const void* d_o, void* dq, void* dk, void* dv, int B, int heads, int head_dim, int Nq, int Nk, long ldq, long ldk,
                                       long ldv, long ldo, long lddo, long lddq, long lddk, long lddv, float scale, float dropout_p,
                                       uint64_t dropout_seed, hipStream_t stream) {
    AttnArgs a{};
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return ANTMMF_EINVAL;
    a.drop_thr = dropout_threshold(dropout_p); a.drop_scale = 1.0f / (1.0f - dropout_p); a.drop_seed = dropout_seed;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.key_bias = key_bias; a.o = (bf16_t*)const_cast<void*>(o);
    a.lse = const_cast<float*>(lse); a.d_o = (const bf16_t*)d_o; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.B = B; a.heads = heads; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
    if (!q || !k || !v || !o || !lse || !d_o || !dq || !dk || !dv || !attn_args_ok(a, head_dim) || (lddo & 7) || (lddq & 3) || (lddk & 3) || (lddv & 3) || (ldo & 7))
        return ANTMMF_EINVAL;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)d_o | (uintptr_t)o) & 15) || (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 7))
        return ANTMMF_EINVAL;   // 16-byte operand vectors (and LDS-DMA pieces), 8-byte gradient stores
    return head_dim == 64 ? attn_bwd_launch<64>(a, stream) : attn_bwd_launch<128>(a, stream);
}

// head size 64: the entry points every tower of the contrastive path uses
extern "C" int antmmf_attention_fwd(const void* q, const void* k, const void* v, const float* key_bias, void* o, float* lse,
                                    int B, int heads, int Nq, int Nk, long ldq, long ldk, long ldv, long ldo, float scale,
                                    float dropout_p, uint64_t dropout_seed, hipStream_t stream) {
    return antmmf_attention_fwd_hd(q, k, v, key_bias, o, lse, B, heads, 64, Nq, Nk, ldq, ldk, ldv, ldo, scale, dropout_p, dropout_seed, stream);
}

extern "C" int antmmf_attention_bwd(const void* q, const void* k, const void* v, const float* key_bias, const void* o, const float* lse,
                                    const void* d_o, void* dq, void* dk, void* dv, int B, int heads, int Nq, int Nk, long ldq, long ldk,
                                    long ldv, long ldo, long lddo, long lddq, long lddk, long lddv, float scale, float dropout_p,
                                    uint64_t dropout_seed, hipStream_t stream) {
    return antmmf_attention_bwd_hd(q, k, v, key_bias, o, lse, d_o, dq, dk, dv, B, heads, 64, Nq, Nk, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, scale,
                                   dropout_p, dropout_seed, stream);
}

// The backward plus the per-batch-item token sums of dQ | dK | dV (fp32 sums of the unrounded gradients): sums[(b * 3 + {0: q, 1: k, 2: v}) * heads * 64 + h * 64 + e] =
// sum_n dX[b, n, h, e] (the dV third only with want_dv != 0).  The q / k / v bias gradients of the layer are the column sums of this [B, 3 * heads * 64] matrix -- B rows instead of B * N: the column-sum passes
// over dQ | dK | dV (one tensor pass each per layer) disappear.  Served by the one-kernel backward only (head size 64, 33 ... 272 keys, no dropout):
// antmmf_attention_bwd_sums_ok says whether a shape is; anything else is ANTMMF_EINVAL (the caller then sums the columns of dq / dk / dv itself).
extern "C" int antmmf_attention_bwd_sums_ok(int head_dim, int Nq, int Nk, float dropout_p) {
    AttnArgs a{};
    float dummy = 0.f;
    a.Nq = Nq; a.Nk = Nk; a.drop_thr = (dropout_p > 0.f) ? 1u : 0u; a.sums = &dummy;
    return head_dim == 64 && Nq > 0 && Nq <= 288 && Nk > 0 && attn_fused_ok(a) ? 1 : 0;
}
extern "C" int antmmf_attention_bwd_sums(const void* q, const void* k, const void* v, const float* key_bias, const void* o, const float* lse,
                                         const void* d_o, void* dq, void* dk, void* dv, float* sums, int want_dv, int B, int heads, int Nq, int Nk, long ldq, long ldk,
                                         long ldv, long ldo, long lddo, long lddq, long lddk, long lddv, float scale, hipStream_t stream) {
    AttnArgs a{};
    a.drop_thr = 0; a.drop_scale = 1.0f; a.drop_seed = 0;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.key_bias = key_bias; a.o = (bf16_t*)const_cast<void*>(o);
    a.lse = const_cast<float*>(lse); a.d_o = (const bf16_t*)d_o; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.B = B; a.heads = heads; a.Nq = Nq; a.Nk = Nk; a.scale = scale; a.sums = sums; a.sums_v = want_dv ? 1 : 0;
    if (!q || !k || !v || !o || !lse || !d_o || !dq || !dk || !dv || !sums || !attn_args_ok(a, 64) || (lddo & 7) || (lddq & 3) || (lddk & 3) || (lddv & 3) || (ldo & 7))
        return ANTMMF_EINVAL;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)d_o | (uintptr_t)o) & 15) || (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 7))
        return ANTMMF_EINVAL;
    if (!attn_fused_ok(a)) return ANTMMF_EINVAL;
    return attn_bwd_launch<64>(a, stream);
}

// ---- key importance: column sums of the attention probabilities --------------------------------------------------------------------------------
// out[b][k] += weight * sum_h sum_q P_drop[b][h][q][k]  -- what the reference does with the attention MAPS it asks the text tower for in training
// (`output_attentions=True`, prj/base_vtp/roi_univl/univl/model/univl_video_base.py:131-143: cat over layers of the head mean, summed over layers and
// queries -> words_importance).  The maps never exist here; the probabilities are rebuilt from Q, K and the forward's log-sum-exp exactly as the backward
// kernels do (same dropout mask: HF's BertSelfAttention returns the probabilities AFTER dropout), summed over the queries in registers, over the 16 query
// lanes by DPP, over the waves in LDS, and leave as one atomic add per (b, h, key).  Head size 64.
template <int NCH, bool DROP>
__global__ __launch_bounds__(ATTN_THREADS, 2) void attn_key_importance_kernel(const AttnArgs a, float* __restrict__ out, float weight) {
    ANTMMF_DYN_LDS(char, smem);
    constexpr int NKP = 32 * NCH, NT = 2 * NCH;
    char* Ks = smem;
    float* kb = reinterpret_cast<float*>(Ks + NKP * 128);
    float* colsum = kb + NKP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, grp = lane >> 4;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    stage_rows<64, NKP>(Ks, a.k + (long)b * a.Nk * a.ldk + h * 64, a.ldk, a.Nk, NKP);
    stage_key_bias(kb, a, b, NKP);
    for (int i = threadIdx.x; i < NKP; i += ATTN_THREADS) colsum[i] = 0.f;
    __syncthreads();
    const float scale2 = a.scale * LOG2E;
    const int nqt = (a.Nq + 15) >> 4;
    float part[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[t][r] = 0.f;
    for (int qt = wave; qt < nqt; qt += ATTN_THREADS / 64) {
        const int qi = qt * 16 + l15;
        const int qrow = qi < a.Nq ? qi : a.Nq - 1;
        const bf16_t* qp = a.q + ((long)b * a.Nq + qrow) * a.ldq + h * 64 + grp * 8;
        const bf16x8_t qf0 = load_frag_global(qp), qf1 = load_frag_global(qp + 32);
        const float lse = a.lse[((long)b * a.heads + h) * a.Nq + qrow];
        // padding queries and fully masked ones (lse = -inf) contribute nothing: subtracting +inf makes every exponent -inf
        const float lse2 = (qi >= a.Nq || lse == -INFINITY) ? INFINITY : lse * LOG2E;
        const uint32_t dbase = (uint32_t)((((long)b * a.heads + h) * a.Nq + qrow) * a.Nk);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t % 3 == 0) LDS_FENCE();
            f32x4_t sa = {0.f, 0.f, 0.f, 0.f};
            sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<64>(Ks, 16 * t + l15, grp), qf0, sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<64>(Ks, 16 * t + l15, 4 + grp), qf1, sa, 0, 0, 0);
            const float4 bias = *reinterpret_cast<const float4*>(kb + 16 * t + 4 * grp);
            const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr = EXP2F(sa[r] * scale2 + bb[r] - lse2);
                if (DROP) pr = DROPOUT_KEEP(dbase + 16 * t + 4 * grp + r, a.drop_seed, a.drop_thr) ? pr * a.drop_scale : 0.f;
                part[t][r] += pr;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = row16_sum(part[t][r]);   // over the wave's 16 query lanes
            if (l15 == 0) atomicAdd(colsum + 16 * t + 4 * grp + r, v);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < a.Nk; i += ATTN_THREADS) atomicAdd(out + (long)b * a.Nk + i, colsum[i] * weight);
}

extern "C" int antmmf_attention_key_importance(const void* q, const void* k, const float* key_bias, const float* lse, float* out, int B, int heads, int Nq, int Nk,
                                               long ldq, long ldk, float scale, float dropout_p, uint64_t dropout_seed, float weight, hipStream_t stream) {
    AttnArgs a{};
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return ANTMMF_EINVAL;
    a.drop_thr = dropout_threshold(dropout_p); a.drop_scale = 1.0f / (1.0f - dropout_p); a.drop_seed = dropout_seed;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.key_bias = key_bias; a.lse = const_cast<float*>(lse);
    a.ldq = ldq; a.ldk = ldk; a.B = B; a.heads = heads; a.Nq = Nq; a.Nk = Nk; a.scale = scale;
    if (!q || !k || !lse || !out || B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0 || Nk > 288 || Nq > 288 || (ldq & 7) || (ldk & 7)) return ANTMMF_EINVAL;
    const int nch = (Nk + 31) / 32;
    const dim3 grid((unsigned)(B * heads)), block(ATTN_THREADS);
#define KIMP(N) do { const size_t lds = (size_t)(32 * N) * 128 + (32 * N) * 8; \
        if (a.drop_thr) { set_lds(attn_key_importance_kernel<N, true>, lds); hipLaunchKernelGGL((attn_key_importance_kernel<N, true>), grid, block, lds, stream, a, out, weight); } \
        else { set_lds(attn_key_importance_kernel<N, false>, lds); hipLaunchKernelGGL((attn_key_importance_kernel<N, false>), grid, block, lds, stream, a, out, weight); } } while (0)
    if (nch <= 1) KIMP(1); else if (nch <= 3) KIMP(3); else if (nch <= 7) KIMP(7); else KIMP(9);
#undef KIMP
    return antmmf_check_launch();
}

