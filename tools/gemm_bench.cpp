// gemm_bench.cpp -- within-process A/B of the NT GEMM kernel variants of libantmmf_hip.so on the ViT-L/14 step's shapes.
// Build: hipcc --offload-arch=gfx950 -O2 tools/gemm_bench.cpp -o tools/gemm_bench -ldl -lhipblaslt      Run (GPU box): tools/gemm_bench [pairs=1024] [rounds=3]
// GEMM_BENCH_HIPBLASLT=1 adds, per NT shape, the vendor library on the SAME buffers, interleaved in the same rounds ("the honest
// baseline to beat"): hipblasLtMatmul, bf16 in / bf16 out, fp32 compute, the same fused epilogue (bias vector; residual as beta = 1 on a
// separate C), the best of the first 16 heuristic algorithms (each timed once, the winner re-timed with the others).
// Variants are selected with antmmf_debug_set_gemm_variant (4 = the product default: BK = 64 persistent kernels; 0 = the round-1 BK = 32 ring).  Prints one JSON line per (shape, variant):
// median TFLOP/s over interleaved rounds, and the max |difference| of the variant's output against variant 0 (same K order: expected 0).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int (*gemm_fn)(const void*, const void*, void*, int, int, int, long, long, long, int, int, int, float, const float*, int, const void*, long,
                       void*, long, const void*, long, int, int, hipStream_t);
typedef int (*gemm_ws_fn)(const void*, const void*, void*, int, int, int, long, long, long, int, int, int, float, const float*, int, const void*, long,
                          void*, long, const void*, long, int, int, float*, long, hipStream_t);
typedef int (*setv_fn)(int);
typedef int (*wgrad_fn)(const void*, const void*, float*, long, int, int, long, long, long, int, float*, long, hipStream_t);
__global__ void maxdiff_f32(const float* a, const float* b, long n, unsigned* out, unsigned* outmag) {
    unsigned m = 0, mm = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = fabsf(a[i] - b[i]);
        const unsigned u = (d == d) ? __float_as_uint(d) : 0x7f800000u;
        m = u > m ? u : m;
        const unsigned v = __float_as_uint(fabsf(a[i]));
        mm = v > mm ? v : mm;
    }
    atomicMax(out, m); atomicMax(outmag, mm);
}
#define LT(x) do { hipblasStatus_t s_ = (x); if (s_ != HIPBLAS_STATUS_SUCCESS) { printf("hipBLASLt error %d at %d\n", (int)s_, __LINE__); exit(1); } } while (0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void fill_bf16(uint16_t* p, long n, uint32_t seed, float scale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 0x9E3779B1u + seed; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        const float f = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;  // uniform [-scale, scale)
        p[i] = (uint16_t)(__float_as_uint(f) >> 16);
    }
}
__global__ void fill_f32(float* p, long n, uint32_t seed) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 0x9E3779B1u + seed; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13;
        p[i] = (h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
}
__global__ void maxdiff(const uint16_t* a, const uint16_t* b, long n, unsigned* out) {
    unsigned m = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)a[i] << 16), y = __uint_as_float((uint32_t)b[i] << 16);
        const float d = fabsf(x - y);
        const unsigned u = (d == d) ? __float_as_uint(d) : 0x7f800000u;
        m = u > m ? u : m;
    }
    atomicMax(out, m);
}
int main(int argc, char** argv) {
    const int pairs = argc > 1 ? atoi(argv[1]) : 1024, rounds = argc > 2 ? atoi(argv[2]) : 3;
    const char* libpath = argc > 3 ? argv[3] : "ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so";   // the LAB library: the product library has no variant switch
    void* h = dlopen(libpath, RTLD_NOW);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    gemm_fn gemm = (gemm_fn)dlsym(h, "antmmf_gemm_bf16");
    gemm_ws_fn gemm_ws = (gemm_ws_fn)dlsym(h, "antmmf_gemm_bf16_ws");   // with a 64-MiB scratch: the tail round of the persistent kernel is split along K (variant bit 25 disables)
    float* tail_ws = nullptr;
    if (gemm_ws && !getenv("GEMM_BENCH_NO_WS")) { if (hipMalloc(&tail_ws, 64u << 20) != hipSuccess) tail_ws = nullptr; }
    setv_fn setv = (setv_fn)dlsym(h, "antmmf_debug_set_gemm_variant");
    if (!gemm || !setv) { printf("missing symbols\n"); return 1; }
    typedef int (*clk_fn)(unsigned long long*);
    clk_fn getclk = (clk_fn)dlsym(h, "antmmf_debug_gemm_clock");
    const long tokens = getenv("GEMM_BENCH_TOKENS") ? atol(getenv("GEMM_BENCH_TOKENS")) : 257L * pairs;   // (GEMM_BENCH_TOKENS=78848: the text tower's row count)
    const int pad = getenv("GEMM_BENCH_PAD") ? atoi(getenv("GEMM_BENCH_PAD")) : 0;  // extra elements in the operands' leading dimension
    struct Shape { const char* tag; int J, R; int bias, res; };
    const Shape shapes[] = {{"fc1", 4096, 1024, 1, 0}, {"fc2", 1024, 4096, 1, 1}, {"qkv", 3072, 1024, 1, 0}, {"out", 1024, 1024, 1, 1},
                            {"dgrad_fc1", 1024, 4096, 0, 0}, {"dgrad_fc2", 4096, 1024, 0, 0}, {"dgrad_qkv", 1024, 3072, 0, 0}, {"dgrad_out", 1024, 1024, 0, 0}};
    std::vector<int> variants = {0, 1, 4};
    if (getenv("GEMM_BENCH_VARIANTS")) { variants.clear(); const char* p = getenv("GEMM_BENCH_VARIANTS"); while (*p) { variants.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; } }
    uint16_t *A, *W, *C0, *C1, *Rz; float* bias; unsigned* dmax;
    CK(hipMalloc(&A, tokens * (4096 + pad) * 2)); CK(hipMalloc(&W, 4096L * (4096 + pad) * 2)); CK(hipMalloc(&C0, tokens * 4096 * 2)); CK(hipMalloc(&C1, tokens * 4096 * 2));
    CK(hipMalloc(&Rz, tokens * 4096 * 2)); CK(hipMalloc(&bias, 4096 * 4)); CK(hipMalloc(&dmax, 4));
    fill_bf16<<<4096, 256>>>(A, tokens * (4096 + pad), 1u, 1.0f);
    fill_bf16<<<4096, 256>>>(Rz, tokens * 4096, 3u, 1.0f);
    fill_f32<<<16, 256>>>(bias, 4096, 4u);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool with_lt = getenv("GEMM_BENCH_HIPBLASLT") != nullptr;
    hipblasLtHandle_t lt = nullptr;
    void* lt_ws = nullptr; const size_t lt_ws_bytes = 256u << 20;
    uint16_t* biasb = nullptr;   // the library's bias vector in the output type (bf16) -- ours reads fp32
    if (with_lt) { LT(hipblasLtCreate(&lt)); CK(hipMalloc(&lt_ws, lt_ws_bytes)); CK(hipMalloc(&biasb, 4096 * 2)); fill_bf16<<<16, 256>>>(biasb, 4096, 4u, 1.0f); }
    for (const Shape& s : shapes) {
        fill_bf16<<<4096, 256>>>(W, (long)s.J * (s.R + pad), 2u, 1.0f / sqrtf((float)s.R));
        CK(hipDeviceSynchronize());
        // GEMM_BENCH_RES_LD0=1: ablation -- every row of the residual tile reads the SAME 512 bytes (leading dimension 0: cache-resident): what is
        // left of the residual epilogue's cost is its instruction path, what disappears is the HBM burst / latency of the residual tile
        static const bool res_ld0 = getenv("GEMM_BENCH_RES_LD0") != nullptr;
        auto run = [&](int v, uint16_t* out) {
            setv(v);
            if (tail_ws) return gemm_ws(A, W, out, (int)tokens, s.J, s.R, s.R + pad, s.R + pad, s.J, 0, 0, 1, 1.0f, s.bias ? bias : nullptr, 0, s.res ? Rz : nullptr, res_ld0 ? 0 : s.J, nullptr, 0, nullptr, 0, 0, 1, tail_ws, 64L << 20, 0);
            return gemm(A, W, out, (int)tokens, s.J, s.R, s.R + pad, s.R + pad, s.J, 0, 0, 1, 1.0f, s.bias ? bias : nullptr, 0, s.res ? Rz : nullptr, res_ld0 ? 0 : s.J, nullptr, 0, nullptr, 0, 0, 1, 0);
        };
        std::vector<std::vector<double>> ms(variants.size());
        std::vector<float> diff(variants.size(), 0.f);
        run(variants[0], C0);
        for (size_t vi = 0; vi < variants.size(); ++vi) {
            CK(hipMemset(C1, 0xff, tokens * s.J * 2));
            const int rc = run(variants[vi], C1);
            if (rc) { printf("variant %d rc %d\n", variants[vi], rc); }
            CK(hipMemset(dmax, 0, 4));
            maxdiff<<<2048, 256>>>(C0, C1, tokens * s.J, dmax);
            unsigned u; CK(hipMemcpy(&u, dmax, 4, hipMemcpyDeviceToHost));
            diff[vi] = *reinterpret_cast<float*>(&u);
        }
        for (int r = 0; r < rounds; ++r)
            for (size_t vi = 0; vi < variants.size(); ++vi) {
                const int iters = 8;
                run(variants[vi], C1);
                CK(hipEventRecord(e0, 0));
                for (int it = 0; it < iters; ++it) run(variants[vi], C1);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                ms[vi].push_back(t / iters);
            }
        if (with_lt) {
            // row-major C[T, J] = A[T, R] W[J, R]^T  ==  column-major C^T (J x T, ld J) = op_T(W buffer: R x J, ld R) * (A buffer: R x T, ld R)
            hipblasLtMatmulDesc_t md; hipblasLtMatrixLayout_t la, lb, lc; hipblasLtMatmulPreference_t pref;
            LT(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
            hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
            LT(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)));
            LT(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)));
            if (s.bias) {
                hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS; hipDataType bt = HIP_R_16BF; const void* bp = biasb;
                LT(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
                LT(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
                LT(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)));
            }
            LT(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, s.R, s.J, s.R + pad));
            LT(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, s.R, tokens, s.R + pad));
            LT(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, s.J, tokens, s.J));
            LT(hipblasLtMatmulPreferenceCreate(&pref));
            uint64_t wsb = lt_ws_bytes;
            LT(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)));
            hipblasLtMatmulHeuristicResult_t heur[16]; int nh = 0;
            LT(hipblasLtMatmulAlgoGetHeuristic(lt, md, la, lb, lc, lc, pref, 16, heur, &nh));
            const float one = 1.0f, beta = s.res ? 1.0f : 0.0f;
            auto run_lt = [&](int a) { return hipblasLtMatmul(lt, md, &one, W, la, A, lb, &beta, s.res ? Rz : C1, lc, C1, lc, &heur[a].algo, lt_ws, lt_ws_bytes, 0); };
            const int own_v = 4;   // the product's kernels, whatever GEMM_BENCH_VARIANTS lists
            int best_a = -1; double best_ms = 1e30;
            for (int a = 0; a < nh; ++a) {
                if (run_lt(a) != HIPBLAS_STATUS_SUCCESS) continue;
                CK(hipEventRecord(e0, 0));
                for (int it = 0; it < 4; ++it) run_lt(a);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (t / 4 < best_ms) { best_ms = t / 4; best_a = a; }
            }
            if (best_a >= 0) {
                // interleaved with the product kernel (variant 0), same rounds, same iteration count
                std::vector<double> mlt, mown;
                for (int r = 0; r < rounds; ++r) {
                    const int iters = 8;
                    run(own_v, C1);
                    CK(hipEventRecord(e0, 0)); for (int it = 0; it < iters; ++it) run(own_v, C1); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float t; CK(hipEventElapsedTime(&t, e0, e1)); mown.push_back(t / iters);
                    run_lt(best_a);
                    CK(hipEventRecord(e0, 0)); for (int it = 0; it < iters; ++it) run_lt(best_a); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&t, e0, e1)); mlt.push_back(t / iters);
                }
                std::sort(mlt.begin(), mlt.end()); std::sort(mown.begin(), mown.end());
                // numerical agreement of the two libraries on the same operands (bf16 outputs; the bias vectors differ only by bf16 rounding)
                run(own_v, C0); run_lt(best_a);
                CK(hipMemset(dmax, 0, 4));
                maxdiff<<<2048, 256>>>(C0, C1, tokens * s.J, dmax);
                unsigned u; CK(hipMemcpy(&u, dmax, 4, hipMemcpyDeviceToHost));
                const double fl = 2.0 * tokens * s.J * s.R;
                printf("{\"shape\": \"%s\", \"I\": %ld, \"J\": %d, \"R\": %d, \"epi\": %d, \"hipblaslt_algos_tried\": %d, \"hipblaslt_ms_med\": %.4f, \"hipblaslt_tf_med\": %.1f, "
                       "\"own_ms_med\": %.4f, \"own_tf_med\": %.1f, \"own_over_hipblaslt\": %.3f, \"maxdiff_own_vs_hipblaslt\": %g}\n",
                       s.tag, tokens, s.J, s.R, s.bias | (s.res << 1), nh, mlt[mlt.size() / 2], fl / mlt[mlt.size() / 2] * 1e-9, mown[mown.size() / 2],
                       fl / mown[mown.size() / 2] * 1e-9, mlt[mlt.size() / 2] / mown[mown.size() / 2], *reinterpret_cast<float*>(&u));
                fflush(stdout);
            } else printf("{\"shape\": \"%s\", \"hipblaslt\": \"no algorithm\"}\n", s.tag);
            hipblasLtMatmulPreferenceDestroy(pref); hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(lb); hipblasLtMatrixLayoutDestroy(lc);
            hipblasLtMatmulDescDestroy(md);
        }
        for (size_t vi = 0; vi < variants.size(); ++vi) {
            std::sort(ms[vi].begin(), ms[vi].end());
            const double med = ms[vi][ms[vi].size() / 2], best = ms[vi][0];
            const double fl = 2.0 * tokens * s.J * s.R;
            double mhz = 0;
            if (getclk) { for (int it = 0; it < 8; ++it) run(variants[vi], C1); CK(hipDeviceSynchronize()); unsigned long long c2[2] = {0, 0}; getclk(c2); if (c2[1]) mhz = (double)c2[0] / (double)c2[1] * 100.0; }
            printf("{\"shape\": \"%s\", \"clock_mhz\": %.0f, \"I\": %ld, \"J\": %d, \"R\": %d, \"epi\": %d, \"variant\": %d, \"pad\": %d, \"ms_med\": %.4f, \"tf_med\": %.1f, \"tf_best\": %.1f, \"maxdiff_vs_v0\": %g}\n",
                   s.tag, mhz, tokens, s.J, s.R, s.bias | (s.res << 1), variants[vi], pad, med, fl / med * 1e-9, fl / best * 1e-9, diff[vi]);
            fflush(stdout);
        }
    }
    // ---- wgrad (TN): dW[n_out][k_in] += dY[tokens][n_out]^T X[tokens][k_in]
    wgrad_fn wgrad = (wgrad_fn)dlsym(h, "antmmf_gemm_wgrad_bf16");
    if (wgrad && !getenv("GEMM_BENCH_NO_TN")) {
        struct WS { const char* tag; int n_out, k_in; };
        const WS wss[] = {{"wgrad_fc1", 4096, 1024}, {"wgrad_fc2", 1024, 4096}, {"wgrad_out", 1024, 1024}, {"wgrad_qkv", 3072, 1024}};
        float *dW0, *dW1, *wsb; unsigned* dm2;
        const long wsbytes = 32L * 4096 * 1024 * 4;
        CK(hipMalloc(&dW0, 4096L * 4096 * 4)); CK(hipMalloc(&dW1, 4096L * 4096 * 4)); CK(hipMalloc(&wsb, wsbytes)); CK(hipMalloc(&dm2, 8));
        std::vector<int> tv = {1028, 4};
        for (const WS& w : wss) {
            auto runw = [&](int v, float* out) { setv(v); return wgrad(Rz, A, out, tokens, w.n_out, w.k_in, w.n_out, w.k_in, w.k_in, 1, wsb, wsbytes, 0); };
            CK(hipMemset(dW0, 0, (long)w.n_out * w.k_in * 4)); CK(hipMemset(dW1, 0, (long)w.n_out * w.k_in * 4));
            runw(tv[0], dW0); runw(tv[1], dW1);
            CK(hipMemset(dm2, 0, 8));
            maxdiff_f32<<<1024, 256>>>(dW0, dW1, (long)w.n_out * w.k_in, dm2, dm2 + 1);
            unsigned u2[2]; CK(hipMemcpy(u2, dm2, 8, hipMemcpyDeviceToHost));
            std::vector<std::vector<double>> ms(tv.size());
            for (int r = 0; r < rounds; ++r)
                for (size_t vi = 0; vi < tv.size(); ++vi) {
                    const int iters = 8;
                    runw(tv[vi], dW1);
                    CK(hipEventRecord(e0, 0));
                    for (int it = 0; it < iters; ++it) runw(tv[vi], dW1);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float t; CK(hipEventElapsedTime(&t, e0, e1));
                    ms[vi].push_back(t / iters);
                }
            if (with_lt) {
                // dW (row-major [n_out, k_in] fp32, accumulated) == column-major (k_in x n_out) = Xc (k_in x T) * Yc^T (Yc: n_out x T)
                hipblasLtMatmulDesc_t md; hipblasLtMatrixLayout_t la, lb, lc; hipblasLtMatmulPreference_t pref;
                LT(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
                hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
                LT(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &opN, sizeof(opN)));
                LT(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &opT, sizeof(opT)));
                LT(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, w.k_in, tokens, w.k_in));
                LT(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, w.n_out, tokens, w.n_out));
                LT(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, w.k_in, w.n_out, w.k_in));
                LT(hipblasLtMatmulPreferenceCreate(&pref));
                uint64_t wsb2 = lt_ws_bytes;
                LT(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb2, sizeof(wsb2)));
                hipblasLtMatmulHeuristicResult_t heur[16]; int nh = 0;
                hipblasStatus_t hs = hipblasLtMatmulAlgoGetHeuristic(lt, md, la, lb, lc, lc, pref, 16, heur, &nh);
                const float one = 1.0f;
                auto run_lt = [&](int a) { return hipblasLtMatmul(lt, md, &one, A, la, Rz, lb, &one, dW1, lc, dW1, lc, &heur[a].algo, lt_ws, lt_ws_bytes, 0); };
                int best_a = -1; double best_ms = 1e30;
                for (int a = 0; hs == HIPBLAS_STATUS_SUCCESS && a < nh; ++a) {
                    if (run_lt(a) != HIPBLAS_STATUS_SUCCESS) continue;
                    CK(hipEventRecord(e0, 0)); for (int it = 0; it < 4; ++it) run_lt(a); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float t; CK(hipEventElapsedTime(&t, e0, e1));
                    if (t / 4 < best_ms) { best_ms = t / 4; best_a = a; }
                }
                if (best_a >= 0) {
                    std::vector<double> mlt, mown;
                    for (int r = 0; r < rounds; ++r) {
                        const int iters = 8; float t;
                        runw(tv.back(), dW1);
                        CK(hipEventRecord(e0, 0)); for (int it = 0; it < iters; ++it) runw(tv.back(), dW1); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&t, e0, e1)); mown.push_back(t / iters);
                        run_lt(best_a);
                        CK(hipEventRecord(e0, 0)); for (int it = 0; it < iters; ++it) run_lt(best_a); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&t, e0, e1)); mlt.push_back(t / iters);
                    }
                    std::sort(mlt.begin(), mlt.end()); std::sort(mown.begin(), mown.end());
                    const double fl = 2.0 * tokens * w.n_out * w.k_in;
                    printf("{\"shape\": \"%s\", \"tokens\": %ld, \"n_out\": %d, \"k_in\": %d, \"hipblaslt_algos_tried\": %d, \"hipblaslt_ms_med\": %.4f, \"hipblaslt_tf_med\": %.1f, \"own_variant\": %d, "
                           "\"own_ms_med\": %.4f, \"own_tf_med\": %.1f, \"own_over_hipblaslt\": %.3f}\n", w.tag, tokens, w.n_out, w.k_in, nh, mlt[mlt.size() / 2],
                           fl / mlt[mlt.size() / 2] * 1e-9, tv.back(), mown[mown.size() / 2], fl / mown[mown.size() / 2] * 1e-9, mlt[mlt.size() / 2] / mown[mown.size() / 2]);
                    fflush(stdout);
                } else printf("{\"shape\": \"%s\", \"hipblaslt\": \"no algorithm (status %d, %d candidates)\"}\n", w.tag, (int)hs, nh);
                hipblasLtMatmulPreferenceDestroy(pref); hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(lb); hipblasLtMatrixLayoutDestroy(lc);
                hipblasLtMatmulDescDestroy(md);
            }
            for (size_t vi = 0; vi < tv.size(); ++vi) {
                std::sort(ms[vi].begin(), ms[vi].end());
                const double med = ms[vi][ms[vi].size() / 2], fl = 2.0 * tokens * w.n_out * w.k_in;
                printf("{\"shape\": \"%s\", \"tokens\": %ld, \"n_out\": %d, \"k_in\": %d, \"variant\": %d, \"ms_med\": %.4f, \"tf_med\": %.1f, \"maxdiff_new_vs_old\": %g, \"max_abs\": %g}\n",
                       w.tag, tokens, w.n_out, w.k_in, tv[vi], med, fl / med * 1e-9, *reinterpret_cast<float*>(&u2[0]), *reinterpret_cast<float*>(&u2[1]));
                fflush(stdout);
            }
        }
    }
    return 0;
}
