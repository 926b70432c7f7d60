// gemm_bench.cpp -- within-process A/B of the NT GEMM kernel variants of libantmmf_hip.so on the ViT-L/14 step's shapes.
// Build: hipcc -O2 tools/gemm_bench.cpp -o tools/gemm_bench -ldl      Run (GPU box): tools/gemm_bench [pairs=1024] [rounds=3]
// Variants are selected with antmmf_debug_set_gemm_variant (0 = product default).  Prints one JSON line per (shape, variant):
// median TFLOP/s over interleaved rounds, and the max |difference| of the variant's output against variant 0 (same K order: expected 0).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int (*gemm_fn)(const void*, const void*, void*, int, int, int, long, long, long, int, int, int, float, const float*, int, const void*, long,
                       void*, long, const void*, long, int, int, hipStream_t);
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
    const char* libpath = argc > 3 ? argv[3] : "ant-multi-modal-framework_amd/lib/libantmmf_hip.so";
    void* h = dlopen(libpath, RTLD_NOW);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    gemm_fn gemm = (gemm_fn)dlsym(h, "antmmf_gemm_bf16");
    setv_fn setv = (setv_fn)dlsym(h, "antmmf_debug_set_gemm_variant");
    if (!gemm || !setv) { printf("missing symbols\n"); return 1; }
    typedef int (*clk_fn)(unsigned long long*);
    clk_fn getclk = (clk_fn)dlsym(h, "antmmf_debug_gemm_clock");
    const long tokens = 257L * pairs;
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
    for (const Shape& s : shapes) {
        fill_bf16<<<4096, 256>>>(W, (long)s.J * (s.R + pad), 2u, 1.0f / sqrtf((float)s.R));
        CK(hipDeviceSynchronize());
        auto run = [&](int v, uint16_t* out) {
            setv(v);
            return gemm(A, W, out, (int)tokens, s.J, s.R, s.R + pad, s.R + pad, s.J, 0, 0, 1, 1.0f, s.bias ? bias : nullptr, 0, s.res ? Rz : nullptr, s.J, nullptr, 0, nullptr, 0, 0, 1, 0);
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
        for (size_t vi = 0; vi < variants.size(); ++vi) {
            std::sort(ms[vi].begin(), ms[vi].end());
            const double med = ms[vi][ms[vi].size() / 2], best = ms[vi][0];
            const double fl = 2.0 * tokens * s.J * s.R;
            double mhz = 0;
            if (getclk && (variants[vi] & 256)) { run(variants[vi], C1); CK(hipDeviceSynchronize()); unsigned long long c2[2] = {0, 0}; getclk(c2); if (c2[1]) mhz = (double)c2[0] / (double)c2[1] * 100.0; }
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
