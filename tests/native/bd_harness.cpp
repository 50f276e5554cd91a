// Native (no-Python) GPU harness: correctness spot checks + kernel timing for libbitdelta_hip.so and for
// experimental tile configurations of bd_gemm_mfma.h.  Development / measurement tool: pytest (-m gpu) holds the
// parity tests proper.  Build: see tests/native/Makefile.  Output: one JSON object per line on stdout.
//
//   bd_harness check            correctness over a shape grid, every kernel family (sampled exact reference)
//   bd_harness perf             timing of the shipped dispatcher on the headline shapes
//   bd_harness sweep            timing of experimental GemmCfg variants on 4096^3 (and 8192/16384 rows)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../bitdelta_amd/csrc/bd_gemm_mfma.h"
#include "ab/bd_gemm_pp.h"
#include "ab/bd_gemm_sp.h"
#include "../../bitdelta_amd/csrc/bd_gemm_pf.h"
#include "../../bitdelta_amd/csrc/bd_gemm_fx.h"
#include "../../include/bitdelta_hip.h"
#include "../../include/bitdelta_hip_test.h"

#define HIPCHECK(x)                                                                      \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)

// ---------------- host-side 16-bit float helpers ----------------
static inline float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t f_to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float f16_to_f(uint16_t h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }
static inline uint16_t f_to_f16(float f) { _Float16 x = (_Float16)f; uint16_t h; memcpy(&h, &x, 2); return h; }
static inline float h2f(uint16_t h, int dt) { return dt == BD_BF16 ? bf16_to_f(h) : f16_to_f(h); }
static inline uint16_t f2h(float f, int dt) { return dt == BD_BF16 ? f_to_bf16(f) : f_to_f16(f); }

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static inline float urand() { return (float)((rng() >> 40) * (1.0 / 16777216.0)); }
static inline float nrand() { float u1 = urand() + 1e-7f, u2 = urand(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }

struct Problem {
    int B, M, N, K, dt, out_dt, fused, tenants;   // tenants: 1 = broadcast mask (sPb = 0), else B masks
    std::vector<uint16_t> A, W;
    std::vector<uint32_t> P;
    std::vector<float> alpha;
    void *dA = nullptr, *dW = nullptr, *dP = nullptr, *dAl = nullptr, *dC = nullptr, *dWs = nullptr;
    int64_t ws_bytes = 0;
    size_t c_elems() const { return (size_t)B * M * N; }
};

static void make_problem(Problem& q) {
    const size_t na = (size_t)q.B * q.M * q.K, np = (size_t)(q.tenants == 1 ? 1 : q.B) * (q.K / 32) * q.N;
    q.A.resize(na);
    for (auto& v : q.A) v = f2h(nrand(), q.dt);
    q.P.resize(np);
    for (auto& v : q.P) v = (uint32_t)rng();
    q.alpha.resize(q.B);
    for (auto& v : q.alpha) v = 4e-4f * (0.75f + 0.5f * urand());
    if (q.fused) {
        q.W.resize((size_t)q.N * q.K);
        for (auto& v : q.W) v = f2h(0.02f * nrand(), q.dt);
    }
    HIPCHECK(hipMalloc(&q.dA, na * 2));
    HIPCHECK(hipMemcpy(q.dA, q.A.data(), na * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&q.dP, np * 4));
    HIPCHECK(hipMemcpy(q.dP, q.P.data(), np * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&q.dAl, q.B * 4));
    HIPCHECK(hipMemcpy(q.dAl, q.alpha.data(), q.B * 4, hipMemcpyHostToDevice));
    if (q.fused) {
        HIPCHECK(hipMalloc(&q.dW, q.W.size() * 2));
        HIPCHECK(hipMemcpy(q.dW, q.W.data(), q.W.size() * 2, hipMemcpyHostToDevice));
    }
    HIPCHECK(hipMalloc(&q.dC, q.c_elems() * 4));
    q.ws_bytes = bd_gemm_workspace_bytes(q.B, q.M, q.N, q.K);
    if (q.ws_bytes) { HIPCHECK(hipMalloc(&q.dWs, q.ws_bytes)); HIPCHECK(hipMemset(q.dWs, 0, q.ws_bytes)); }   // ticket-area contract
}
static void free_problem(Problem& q) {
    hipFree(q.dA); hipFree(q.dP); hipFree(q.dAl); hipFree(q.dC);
    if (q.dW) hipFree(q.dW);
    if (q.dWs) hipFree(q.dWs);
}

static double ref_value(const Problem& q, int b, int m, int n) {
    const uint16_t* a = &q.A[((size_t)b * q.M + m) * q.K];
    const uint32_t* p = &q.P[(size_t)(q.tenants == 1 ? 0 : b) * (q.K / 32) * q.N];
    double d = 0.0, base = 0.0;
    for (int k = 0; k < q.K; ++k) {
        const double x = h2f(a[k], q.dt);
        d += ((p[(size_t)(k >> 5) * q.N + n] >> (k & 31)) & 1u) ? x : -x;
        if (q.fused) base += x * (double)h2f(q.W[(size_t)n * q.K + k], q.dt);
    }
    return q.fused ? base + (double)q.alpha[b] * d : d;
}

// returns number of bad samples; fills max relative error (relative to the row-scale sqrt(K))
static int check_output(const Problem& q, const void* hC, int nsamples, double* max_err, double* max_ulp) {
    int bad = 0;
    *max_err = 0; *max_ulp = 0;
    const bool exhaustive = q.c_elems() <= (size_t)nsamples;
    const size_t total = exhaustive ? q.c_elems() : (size_t)nsamples;
    for (size_t s = 0; s < total; ++s) {
        size_t idx = exhaustive ? s : (size_t)(rng() % q.c_elems());
        if (!exhaustive && s < 64) {   // force the corners / edges into the sample
            const int bb = (s & 1) ? q.B - 1 : 0, mm = (s & 2) ? q.M - 1 : (int)(s % q.M), nn = (s & 4) ? q.N - 1 : (int)((s * 7919) % q.N);
            idx = ((size_t)bb * q.M + mm) * q.N + nn;
        }
        const int n = (int)(idx % q.N), m = (int)((idx / q.N) % q.M), b = (int)(idx / ((size_t)q.N * q.M));
        const double r = ref_value(q, b, m, n);
        double got, tol;
        if (q.out_dt == BD_F32) {
            got = ((const float*)hC)[idx];
            tol = 2e-6 * (fabs(r) + sqrt((double)q.K));
        } else {
            got = h2f(((const uint16_t*)hC)[idx], q.out_dt);
            int ex;
            frexp(r, &ex);                                   // |r| in [2^(ex-1), 2^ex)
            const double ulp = ldexp(1.0, ex - 1 - (q.out_dt == BD_BF16 ? 7 : 10));
            tol = 0.5 * ulp + 4e-6 * (fabs(r) + sqrt((double)q.K)) + (q.out_dt == BD_F16 ? 6e-8 : 0);
            if (ulp > 0) *max_ulp = fmax(*max_ulp, fabs(got - r) / ulp);
        }
        const double e = fabs(got - r);
        *max_err = fmax(*max_err, e / (fabs(r) + sqrt((double)q.K)));
        if (!(e <= tol)) {
            if (bad < 5) fprintf(stderr, "  MISMATCH b=%d m=%d n=%d got=%.8g ref=%.8g tol=%.3g\n", b, m, n, got, r, tol);
            ++bad;
        }
    }
    return bad;
}

static int call_api(const Problem& q, hipStream_t st) {
    const int64_t sPb = q.tenants == 1 ? 0 : (int64_t)(q.K / 32) * q.N;
    if (q.fused)
        return bd_binary_linear(q.dA, q.dW, (const int32_t*)q.dP, (const float*)q.dAl, q.dC, q.B, q.M, q.N, q.K,
                                (int64_t)q.M * q.K, q.K, q.K, sPb, 1, 1, (int64_t)q.M * q.N, q.N, q.dt, q.out_dt, q.dWs,
                                q.ws_bytes, st);
    return bd_delta_bmm(q.dA, (const int32_t*)q.dP, q.dC, q.B, q.M, q.N, q.K, (int64_t)q.M * q.K, q.K, sPb,
                        (int64_t)q.M * q.N, q.N, q.dt, q.out_dt, 0, nullptr, 0, 1, 0, q.dWs, q.ws_bytes, st);
}

template <class F>
static double time_ms(F&& f, int warm, int iters) {
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) f();
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) f();
    HIPCHECK(hipEventRecord(e1, 0));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / iters;
}

static const char* dtn(int dt) { return dt == BD_BF16 ? "bf16" : dt == BD_F16 ? "f16" : "f32"; }

static int run_case(const char* tag, int B, int M, int N, int K, int dt, int out_dt, int fused, int tenants, int variant,
                    int iters, int nsamples) {
    Problem q{B, M, N, K, dt, out_dt, fused, tenants};
    make_problem(q);
    HIPCHECK(hipMemset(q.dC, 0xFF, q.c_elems() * 4));
    bd_set_gemm_variant(variant);
    int rc = call_api(q, 0);
    hipError_t herr = hipDeviceSynchronize();
    const int used = bd_last_gemm_variant();
    int bad = -1;
    double max_err = 0, max_ulp = 0, ms = 0;
    if (rc == 0 && herr == hipSuccess) {
        std::vector<uint8_t> hC(q.c_elems() * (out_dt == BD_F32 ? 4 : 2));
        HIPCHECK(hipMemcpy(hC.data(), q.dC, hC.size(), hipMemcpyDeviceToHost));
        bad = nsamples > 0 ? check_output(q, hC.data(), nsamples, &max_err, &max_ulp) : 0;
        if (iters > 0) ms = time_ms([&] { call_api(q, 0); }, 3, iters);
    }
    const double flops = (fused ? 4.0 : 2.0) * B * M * (double)N * K;
    const double bytes = 2.0 * B * M * K + (double)(tenants == 1 ? 1 : B) * K * N / 8 + (out_dt == BD_F32 ? 4.0 : 2.0) * B * M * N +
                         (fused ? 2.0 * N * K : 0.0);
    printf("{\"tag\":\"%s\",\"B\":%d,\"M\":%d,\"N\":%d,\"K\":%d,\"dt\":\"%s\",\"out\":\"%s\",\"fused\":%d,\"tenants\":%d,"
           "\"variant\":%d,\"used\":%d,\"rc\":%d,\"hip\":\"%s\",\"bad\":%d,\"max_err\":%.3g,\"max_ulp\":%.3g,\"ms\":%.5f,"
           "\"tflops\":%.2f,\"gbps\":%.1f}\n",
           tag, B, M, N, K, dtn(dt), dtn(out_dt), fused, tenants, variant, used, rc, hipGetErrorString(herr), bad, max_err,
           max_ulp, ms, ms > 0 ? flops / ms * 1e-9 : 0.0, ms > 0 ? bytes / ms * 1e-6 : 0.0);
    fflush(stdout);
    bd_set_gemm_variant(-1);
    free_problem(q);
    if (herr != hipSuccess) exit(3);   // device is in an undefined state after a fault
    return (rc != 0 || bad != 0) ? 1 : 0;
}

// ---------------- experimental configs launched directly ----------------
template <class Cfg, int PP> struct KernSel { static auto get() { return bd::delta_gemm_kernel<Cfg>; } };
template <class Cfg> struct KernSel<Cfg, 1> { static auto get() { return bd::delta_gemm_pp_kernel<Cfg>; } };
template <class Cfg> struct KernSel<Cfg, 2> { static auto get() { return bd::delta_gemm_sp_kernel<Cfg>; } };
template <class Cfg> struct KernSel<Cfg, 3> { static auto get() { return bd::delta_gemm_pf_kernel<Cfg>; } };
template <class Cfg> struct KernSel<Cfg, 4> { static auto get() { return bd::delta_gemm_fx_kernel<Cfg>; } };

static int g_group_m = 1;
template <class Cfg, int PP = 0>
static void run_cfg(const char* name, int M, int N, int K, int iters, int nsamples) {
    auto kern = KernSel<Cfg, PP>::get();
    Problem q{1, M, N, K, Cfg::DT == bd::DT_BF16 ? BD_BF16 : BD_F16, Cfg::OUT_F32 ? BD_F32 : (Cfg::DT == bd::DT_BF16 ? BD_BF16 : BD_F16),
              Cfg::FUSED ? 1 : 0, 1};
    make_problem(q);
    bd::GemmParams p{};
    p.A = (const char*)q.dA; p.P = (const int32_t*)q.dP; p.C = (char*)q.dC; p.W = (const char*)q.dW; p.alpha = (const float*)q.dAl;
    p.M = M; p.N = N; p.K = K;
    p.tiles_m = (M + Cfg::BM - 1) / Cfg::BM; p.tiles_n = (N + Cfg::BN - 1) / Cfg::BN;
    p.sAb = (long long)M * K; p.sPb = 0; p.sCb = (long long)M * N; p.sAm = K; p.sCm = N; p.ldw = K; p.sAlb = 0; p.gsz = N;
    p.round_mode = 0; p.accumulate = 0; p.group_m = g_group_m < p.tiles_m ? g_group_m : p.tiles_m; p.ksplit = 1;
    HIPCHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    dim3 grid(p.tiles_m * p.tiles_n, 1);
    auto launch = [&] { hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, 0, p); };
    HIPCHECK(hipMemset(q.dC, 0xFF, q.c_elems() * 4));
    launch();
    hipError_t herr = hipDeviceSynchronize();
    int bad = -1; double max_err = 0, max_ulp = 0, ms = 0;
    if (herr == hipSuccess) {
        std::vector<uint8_t> hC(q.c_elems() * (Cfg::OUT_F32 ? 4 : 2));
        HIPCHECK(hipMemcpy(hC.data(), q.dC, hC.size(), hipMemcpyDeviceToHost));
        bad = nsamples > 0 ? check_output(q, hC.data(), nsamples, &max_err, &max_ulp) : 0;
        ms = time_ms(launch, 5, iters);
    }
    const double flops = (Cfg::FUSED ? 4.0 : 2.0) * M * (double)N * K;
    printf("{\"tag\":\"cfg\",\"name\":\"%s\",\"M\":%d,\"N\":%d,\"K\":%d,\"hip\":\"%s\",\"bad\":%d,\"max_err\":%.3g,\"max_ulp\":%.3g,"
           "\"ms\":%.5f,\"tflops\":%.2f,\"lds\":%d}\n",
           name, M, N, K, hipGetErrorString(herr), bad, max_err, max_ulp, ms, ms > 0 ? flops / ms * 1e-9 : 0.0, Cfg::LDS_BYTES);
    fflush(stdout);
    free_problem(q);
    if (herr != hipSuccess) exit(3);
}

using namespace bd;
#define CFG(name, ...) run_cfg<GemmCfg<__VA_ARGS__>>(name, M, N, K, iters, 2048)

static void sweep(int M, int N, int K, int iters) {
    //                      DT   BM   BN  WM WN NS fused f32 OPT
    CFG("256x256_2x4_ns4", DT_BF16, 256, 256, 2, 4, 4, false, false, 0);
    CFG("256x256_2x4_ns3", DT_BF16, 256, 256, 2, 4, 3, false, false, 0);
    CFG("256x256_2x4_ns4_sgb", DT_BF16, 256, 256, 2, 4, 4, false, false, 1);
    CFG("256x256_2x4_ns4_prio", DT_BF16, 256, 256, 2, 4, 4, false, false, 2);
    CFG("256x256_2x4_ns4_sgb_prio", DT_BF16, 256, 256, 2, 4, 4, false, false, 3);
    CFG("256x256_1x8_ns4", DT_BF16, 256, 256, 1, 8, 4, false, false, 0);
    CFG("256x256_1x8_ns4_prio", DT_BF16, 256, 256, 1, 8, 4, false, false, 2);
    CFG("256x256_4x2_ns4", DT_BF16, 256, 256, 4, 2, 4, false, false, 0);
    CFG("256x128_2x2_ns4", DT_BF16, 256, 128, 2, 2, 4, false, false, 0);
    CFG("128x256_1x4_ns4", DT_BF16, 128, 256, 1, 4, 4, false, false, 0);
    CFG("128x128_2x2_ns4", DT_BF16, 128, 128, 2, 2, 4, false, false, 0);
    CFG("256x256_2x4_ns4_f16", DT_F16, 256, 256, 2, 4, 4, false, false, 0);
    CFG("256x256_2x4_fused", DT_BF16, 256, 256, 2, 4, 4, true, false, 0);
    CFG("128x256_1x4_fused", DT_BF16, 128, 256, 1, 4, 4, true, false, 0);
}

#define CFGPP(name, ...) run_cfg<GemmCfg<__VA_ARGS__>, 1>(name, M, N, K, iters, 4096)
#define CFGSP(name, ...) run_cfg<GemmCfg<__VA_ARGS__>, 2>(name, M, N, K, iters, 4096)
#define CFGPF(name, ...) run_cfg<GemmCfg<__VA_ARGS__>, 3>(name, M, N, K, iters, 4096)
static void sweep_pf(int M, int N, int K, int iters) {
    for (int rep = 0; rep < 3; ++rep) {      // interleaved A/B rounds in one process
        CFGPP("pp_ns4_noprio", DT_BF16, 256, 256, 2, 4, 4, false, false, 2);
        CFGPF("pf_ns4", DT_BF16, 256, 256, 2, 4, 4, false, false, 0);
        CFGPF("pf_ns4_lut", DT_BF16, 256, 256, 2, 4, 4, false, false, 1);
        CFGPF("pf_256x128_lut", DT_BF16, 256, 128, 2, 4, 4, false, false, 1);
        CFGPF("pf_256x128", DT_BF16, 256, 128, 2, 4, 4, false, false, 0);
        CFGPP("pp_256x128", DT_BF16, 256, 128, 2, 4, 4, false, false, 2);
    }
}

#define CFGFX(name, ...) run_cfg<FxCfg<__VA_ARGS__>, 4>(name, M, N, K, iters, 4096)
// one-pass fused (two accumulator sets) against the two-loop fused kernels, interleaved in one process
static void sweep_fx(int M, int N, int K, int iters) {
    for (int rep = 0; rep < 3; ++rep) {
        CFGFX("fx_256x128_ns3_lut", DT_BF16, 256, 128, 3, false, 1);
        CFGFX("fx_256x128_ns3_valu", DT_BF16, 256, 128, 3, false, 0);
        CFGPF("pf_256x128_fused", DT_BF16, 256, 128, 2, 4, 4, true, false, 0);
        CFGPP("pp_256x256_fused", DT_BF16, 256, 256, 2, 4, 4, true, false, 2);
    }
}

// tile walk order for the one-pass fused kernel: group_m tile rows per group (1 = n fastest, 99 = m fastest)
static void sweep_order(int M, int N, int K, int iters) {
    for (int rep = 0; rep < 2; ++rep)
        for (int g : {1, 2, 4, 99}) {
            g_group_m = g;
            char nm[64]; snprintf(nm, sizeof nm, "fx_gm%d", g);
            CFGFX(nm, DT_BF16, 256, 128, 3, false, 1);
        }
    g_group_m = 1;
}

static void sweep_sp(int M, int N, int K, int iters) {
    for (int rep = 0; rep < 2; ++rep) {      // interleaved A/B: two rounds in one process
        CFGPP("pp_ns4_noprio", DT_BF16, 256, 256, 2, 4, 4, false, false, 2);
        CFGPP("pp_ns4_prio", DT_BF16, 256, 256, 2, 4, 4, false, false, 0);
        CFGPP("pp_ns4_xg2", DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + 64);
        CFGPP("pp_ns4_xg4", DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + 512);
        CFGPP("pp_ns4_asym", DT_BF16, 256, 256, 2, 4, 4, false, false, 6);
        CFGPP("pp_256x128_noprio", DT_BF16, 256, 128, 2, 4, 4, false, false, 2);
        CFGSP("sp_ns4", DT_BF16, 256, 256, 2, 4, 4, false, false, 0);
        CFGSP("sp_ns3", DT_BF16, 256, 256, 2, 4, 3, false, false, 0);
        CFGSP("sp_ns4_nopin", DT_BF16, 256, 256, 2, 4, 4, false, false, 1);
        CFGSP("sp_ns4_prio", DT_BF16, 256, 256, 2, 4, 4, false, false, 2);
        CFGSP("sp_1x8_ns4", DT_BF16, 256, 256, 1, 8, 4, false, false, 0);
        CFGSP("sp_256x128_ns4", DT_BF16, 256, 128, 2, 4, 4, false, false, 0);
        CFGSP("sp_128x256_1x4", DT_BF16, 128, 256, 1, 4, 4, false, false, 0);
    }
}

static void sweep_pp(int M, int N, int K, int iters) {
    if (K < 256) {   // fixed-cost probe: launch + prologue + epilogue with (almost) no main loop
        CFGPP("pp_256x256_ns4", DT_BF16, 256, 256, 2, 4, 4, false, false, 0);
        CFGPP("pp_256x256_ns4_f32out", DT_BF16, 256, 256, 2, 4, 4, false, true, 0);
        return;
    }
    CFG("v1_256x256_2x4_ns4", DT_BF16, 256, 256, 2, 4, 4, false, false, 0);
    CFGPP("pp_256x256_ns5", DT_BF16, 256, 256, 2, 4, 4, false, false, 0);
    CFGPP("pp_256x256_ns4", DT_BF16, 256, 256, 2, 4, 4, false, false, 0);
    CFGPP("pp_ns4_nopin", DT_BF16, 256, 256, 2, 4, 4, false, false, 1);
    CFGPP("pp_ns4_noprio", DT_BF16, 256, 256, 2, 4, 4, false, false, 2);
    CFGPP("pp_ns4_tail4", DT_BF16, 256, 256, 2, 4, 4, false, false, 4);
    CFGPP("pp_ns4_tail4_noprio", DT_BF16, 256, 256, 2, 4, 4, false, false, 6);
    CFGPP("pp_ns4_tail8", DT_BF16, 256, 256, 2, 4, 4, false, false, 8);
    CFGPP("pp_ns4_tail8_noprio", DT_BF16, 256, 256, 2, 4, 4, false, false, 10);
    CFGPP("pp_256x128_ns4", DT_BF16, 256, 128, 2, 4, 4, false, false, 2);
    CFGPP("pp_256x128_ns4_fused", DT_BF16, 256, 128, 2, 4, 4, true, false, 2);
    CFGPP("pp_256x256_ns4_f16", DT_F16, 256, 256, 2, 4, 4, false, false, 0);
    CFGPP("pp_256x256_ns4_f32out", DT_BF16, 256, 256, 2, 4, 4, false, true, 0);
    CFG("v1_256x256_fused", DT_BF16, 256, 256, 2, 4, 4, true, false, 0);
    CFGPP("pp_256x256_ns4_fused2", DT_BF16, 256, 256, 2, 4, 4, true, false, 2);
    CFGPP("pp_256x256_ns4_fused", DT_BF16, 256, 256, 2, 4, 4, true, false, 0);
}

static void ablate(int M, int N, int K, int iters) {
    // timing only: outputs are wrong by construction (nsamples = 0 skips the check)
#define ABL(name, opt) run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, opt>>(name, M, N, K, iters, 0)
    ABL("base", 0);
    ABL("no_dma", 4);
    ABL("no_expand", 8);
    ABL("no_dsread", 16);
    ABL("no_barrier_no_dma", 36);
    ABL("no_mfma", 64);
    ABL("no_dma_no_expand", 12);
    ABL("no_dma_no_dsread", 20);
    ABL("no_dma_no_expand_no_dsread", 28);
    ABL("mfma_only", 60);
    ABL("no_mfma_no_dma", 68);
#undef ABL
}

// HBM-bound bit-plane kernels: time + algorithmic bytes (spot-checked for self-consistency: unpack(pack(x)) == x, merge sign)
static void bits_bench() {
    const int64_t N = 11008, K = 4096;      // Llama-2-7B gate_proj sized
    std::vector<uint8_t> hb((size_t)K * N);
    for (auto& v : hb) v = (uint8_t)(rng() & 1);
    uint8_t *dbits, *dbits2; uint32_t* dwords; uint16_t *dW, *dF; float *dcoeff, *dws;
    HIPCHECK(hipMalloc(&dbits, (size_t)K * N)); HIPCHECK(hipMalloc(&dbits2, (size_t)K * N)); HIPCHECK(hipMalloc(&dwords, (size_t)K / 32 * N * 4));
    HIPCHECK(hipMalloc(&dW, (size_t)N * K * 2)); HIPCHECK(hipMalloc(&dF, (size_t)N * K * 2)); HIPCHECK(hipMalloc(&dcoeff, 4));
    const int64_t wsb = bd_binarize_workspace_bytes(N, K);
    HIPCHECK(hipMalloc(&dws, wsb));
    HIPCHECK(hipMemcpy(dbits, hb.data(), hb.size(), hipMemcpyHostToDevice));
    std::vector<uint16_t> hw((size_t)N * K), hf((size_t)N * K);
    for (size_t i = 0; i < hw.size(); ++i) { const float w = 0.02f * nrand(); hw[i] = f_to_bf16(w); hf[i] = f_to_bf16(bf16_to_f(hw[i]) + 5e-4f * nrand()); }
    HIPCHECK(hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(dF, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
    auto report = [&](const char* name, double ms, double bytes, int ok) {
        printf("{\"tag\":\"bits\",\"name\":\"%s\",\"N\":%lld,\"K\":%lld,\"ms\":%.5f,\"gbps\":%.1f,\"frac_of_8TBs\":%.3f,\"ok\":%d}\n", name, (long long)N, (long long)K, ms,
               bytes / ms * 1e-6, bytes / ms * 1e-6 / 8000.0, ok);
        fflush(stdout);
    };
    double ms = time_ms([&] { bd_pack(dbits, 1, K, N, K * N, N, 1, dwords, 32, 0); }, 3, 20);
    report("pack_rowmajor", ms, (double)K * N + (double)K * N / 8, 1);
    ms = time_ms([&] { bd_unpack(dwords, 1, K / 32, N, dbits2, 32, 0); }, 3, 20);
    std::vector<uint8_t> hb2((size_t)K * N);
    HIPCHECK(hipMemcpy(hb2.data(), dbits2, hb2.size(), hipMemcpyDeviceToHost));
    report("unpack", ms, (double)K * N + (double)K * N / 8, memcmp(hb.data(), hb2.data(), hb.size()) == 0);
    // k-major view (what BinaryDiff.__init__ packs: bits stored [N,K], logical [K,N] with s_k = 1, s_n = K)
    ms = time_ms([&] { bd_pack(dbits, 1, K, N, K * N, 1, K, dwords, 32, 0); }, 3, 20);
    report("pack_kmajor", ms, (double)K * N + (double)K * N / 8, 1);
    ms = time_ms([&] { bd_binarize(dW, dF, N, K, K, BD_BF16, (int32_t*)dwords, dcoeff, dws, wsb, 0); }, 3, 20);
    float hc = 0; HIPCHECK(hipMemcpy(&hc, dcoeff, 4, hipMemcpyDeviceToHost));
    report("binarize", ms, 4.0 * N * K + (double)K * N / 8, hc > 3.5e-4f && hc < 4.5e-4f);
    ms = time_ms([&] { bd_merge_delta(dW, K, (const int32_t*)dwords, dcoeff, N, K, BD_BF16, 0); }, 3, 20);
    report("merge_delta", ms, 4.0 * N * K + (double)K * N / 8, 1);
    hipFree(dbits); hipFree(dbits2); hipFree(dwords); hipFree(dW); hipFree(dF); hipFree(dcoeff); hipFree(dws);
}

// Decode timing with COLD weights: the same problem over `nset` device copies of W and P (nset * bytes > 2 x the 256 MB Infinity
// Cache), launched round-robin, so every launch streams its weights from HBM as a decode step of a real model does.  (run_case's
// loop re-reads one 46 MB problem from the Infinity Cache and flatters HBM-bound kernels.)  Prints warm and cold us per launch.
static int g_tiled = 0;      // decode calls go through bd_binary_linear_decode: 1 = tile-major sign words, 2 = packed (interleaved tenants)
static int tpad_of(int T) { return T <= 1 ? 1 : T <= 2 ? 2 : T <= 4 ? 4 : T <= 6 ? 6 : 8; }
static int call_decode(const Problem& q) {
    if (!g_tiled || !q.fused) return call_api(q, 0);
    const int T = q.tenants == 1 ? 1 : q.B;
    const int64_t sPb = q.tenants == 1 ? 0 : (int64_t)(q.K / 32) * ((q.N + 15) / 16 * 16);
    return bd_binary_linear_decode(q.dA, q.dW, (const int32_t*)q.dP, g_tiled, g_tiled == 2 ? tpad_of(T) : 0, (const float*)q.dAl, q.dC,
                                   q.B, q.M, q.N, q.K, (int64_t)q.M * q.K, q.K, q.K, sPb, 1, 1, (int64_t)q.M * q.N, q.N, q.dt,
                                   q.out_dt, 0, 0);
}
static int run_decode_cold(const char* tag, int B, int M, int N, int K, int dt, int fused, int variant, int iters) {
    Problem q{B, M, N, K, dt, dt, fused, B};
    make_problem(q);
    void* dP_ref = q.dP;
    size_t pbytes_dev = (size_t)(q.tenants == 1 ? 1 : B) * (K / 32) * N * 4;
    if (g_tiled && fused) {      // repack on the host (the host copy q.P keeps the reference order for the checker)
        const int KW = K / 32, NT = (N + 15) / 16, T = (q.tenants == 1 ? 1 : B);
        std::vector<uint32_t> tp;
        if (g_tiled == 1) {      // [T][K/32][N] -> [T][N/16][K/32][16]
            tp.assign((size_t)T * NT * KW * 16, 0u);
            for (int t = 0; t < T; ++t)
                for (int i = 0; i < KW; ++i)
                    for (int n = 0; n < N; ++n)
                        tp[(((size_t)t * NT + (n >> 4)) * KW + i) * 16 + (n & 15)] = q.P[((size_t)t * KW + i) * N + n];
        } else {                 // -> [N/16][K/128][4 g][16][t_pad], byte s of a dword = byte g of word row 4 it + s
            const int NIT = (K + 127) / 128, TP = tpad_of(T);
            tp.assign((size_t)NT * NIT * 4 * 16 * TP, 0u);
            for (int t = 0; t < T; ++t)
                for (int i = 0; i < KW; ++i)
                    for (int n = 0; n < N; ++n) {
                        const uint32_t w = q.P[((size_t)t * KW + i) * N + n];
                        const int it = i >> 2, sidx = i & 3;
                        for (int g = 0; g < 4; ++g)
                            tp[((((size_t)(n >> 4) * NIT + it) * 4 + g) * 16 + (n & 15)) * TP + t] |= ((w >> (8 * g)) & 0xffu) << (8 * sidx);
                    }
        }
        pbytes_dev = tp.size() * 4;
        HIPCHECK(hipMalloc(&q.dP, pbytes_dev));
        HIPCHECK(hipMemcpy(q.dP, tp.data(), pbytes_dev, hipMemcpyHostToDevice));
    }
    bd_set_gemm_variant(variant);
    int rc = call_decode(q);
    hipError_t herr = hipDeviceSynchronize();
    const int used = bd_last_gemm_variant();
    int bad = -1;
    double max_err = 0, max_ulp = 0;
    if (rc == 0 && herr == hipSuccess) {
        std::vector<uint8_t> hC(q.c_elems() * 2);
        HIPCHECK(hipMemcpy(hC.data(), q.dC, hC.size(), hipMemcpyDeviceToHost));
        bad = check_output(q, hC.data(), 2048, &max_err, &max_ulp);
    }
    const double wbytes = fused ? 2.0 * N * K : 0.0, pbytes = (double)pbytes_dev;
    const double bytes = 2.0 * B * M * K + (double)B * K * N / 8 + 2.0 * B * M * N + wbytes;
    int nset = (int)(600e6 / (wbytes + pbytes)) + 1;
    if (nset < 2) nset = 2;
    if (nset > 24) nset = 24;
    std::vector<void*> Ws(nset, nullptr), Ps(nset, nullptr);
    for (int i = 0; i < nset; ++i) {
        if (fused) { HIPCHECK(hipMalloc(&Ws[i], (size_t)wbytes)); HIPCHECK(hipMemcpy(Ws[i], q.dW, (size_t)wbytes, hipMemcpyDeviceToDevice)); }
        HIPCHECK(hipMalloc(&Ps[i], (size_t)pbytes)); HIPCHECK(hipMemcpy(Ps[i], q.dP, (size_t)pbytes, hipMemcpyDeviceToDevice));
    }
    double warm = 0, cold = 0;
    if (rc == 0 && herr == hipSuccess && bad == 0) {
        warm = time_ms([&] { call_decode(q); }, 5, iters) * 1e3;
        int idx = 0;
        Problem c = q;
        cold = time_ms([&] { c.dW = Ws[idx]; c.dP = Ps[idx]; idx = (idx + 1) % nset; call_decode(c); }, nset, iters) * 1e3;
    }
    printf("{\"tag\":\"%s\",\"B\":%d,\"M\":%d,\"N\":%d,\"K\":%d,\"dt\":\"%s\",\"fused\":%d,\"variant\":%d,\"used\":%d,\"rc\":%d,"
           "\"bad\":%d,\"max_ulp\":%.3g,\"MB\":%.1f,\"nset\":%d,\"warm_us\":%.2f,\"warm_gbps\":%.0f,\"cold_us\":%.2f,\"cold_gbps\":%.0f}\n",
           tag, B, M, N, K, dtn(dt), fused, variant, used, rc, bad, max_ulp, bytes * 1e-6, nset, warm, warm > 0 ? bytes / warm * 1e-3 : 0.0,
           cold, cold > 0 ? bytes / cold * 1e-3 : 0.0);
    fflush(stdout);
    bd_set_gemm_variant(-1);
    for (int i = 0; i < nset; ++i) { if (Ws[i]) hipFree(Ws[i]); hipFree(Ps[i]); }
    if (q.dP != dP_ref) { hipFree(q.dP); q.dP = dP_ref; }
    free_problem(q);
    if (herr != hipSuccess) exit(3);
    return (rc != 0 || bad != 0) ? 1 : 0;
}

// pure weight streaming: bd_tenant_linear (no sign operand) over `nset` cold copies of W; the ceiling of the decode kernel structure
static int run_tenant_cold(const char* tag, int T, int N, int K, int iters) {
    const size_t wbytes = (size_t)T * N * K * 2;
    int nset = (int)(600e6 / (double)wbytes) + 1;
    if (nset < 2) nset = 2;
    if (nset > 24) nset = 24;
    std::vector<uint16_t> hW((size_t)T * N * K), hX((size_t)T * K);
    for (auto& v : hW) v = f2h(0.02f * nrand(), BD_F16);
    for (auto& v : hX) v = f2h(nrand(), BD_F16);
    void *dX, *dY;
    std::vector<void*> Ws(nset);
    HIPCHECK(hipMalloc(&dX, hX.size() * 2)); HIPCHECK(hipMemcpy(dX, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&dY, (size_t)T * N * 2));
    for (int i = 0; i < nset; ++i) { HIPCHECK(hipMalloc(&Ws[i], wbytes)); HIPCHECK(hipMemcpy(Ws[i], hW.data(), wbytes, hipMemcpyHostToDevice)); }
    auto call = [&](int i) { return bd_tenant_linear(dX, Ws[i], dY, T, 1, N, K, K, K, (int64_t)N * K, K, N, N, BD_F16, BD_F16, 0); };
    int rc = call(0);
    hipError_t herr = hipDeviceSynchronize();
    int bad = 0;
    if (rc == 0 && herr == hipSuccess) {
        std::vector<uint16_t> hY((size_t)T * N);
        HIPCHECK(hipMemcpy(hY.data(), dY, hY.size() * 2, hipMemcpyDeviceToHost));
        for (int s = 0; s < 256; ++s) {
            const int t = (int)(rng() % T), n = (int)(rng() % N);
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)h2f(hX[(size_t)t * K + k], BD_F16) * (double)h2f(hW[((size_t)t * N + n) * K + k], BD_F16);
            const double got = h2f(hY[(size_t)t * N + n], BD_F16);
            if (fabs(got - ref) > 2e-3 * fabs(ref) + 2e-3) ++bad;
        }
    }
    double warm = 0, cold = 0;
    if (rc == 0 && herr == hipSuccess && bad == 0) {
        warm = time_ms([&] { call(0); }, 5, iters) * 1e3;
        int idx = 0;
        cold = time_ms([&] { call(idx); idx = (idx + 1) % nset; }, nset, iters) * 1e3;
    }
    printf("{\"tag\":\"%s\",\"T\":%d,\"N\":%d,\"K\":%d,\"rc\":%d,\"bad\":%d,\"MB\":%.1f,\"nset\":%d,\"warm_us\":%.2f,\"warm_gbps\":%.0f,"
           "\"cold_us\":%.2f,\"cold_gbps\":%.0f}\n", tag, T, N, K, rc, bad, wbytes * 1e-6, nset, warm, warm > 0 ? wbytes / warm * 1e-3 : 0.0,
           cold, cold > 0 ? wbytes / cold * 1e-3 : 0.0);
    fflush(stdout);
    hipFree(dX); hipFree(dY);
    for (auto w : Ws) hipFree(w);
    if (herr != hipSuccess) exit(3);
    return (rc != 0 || bad != 0) ? 1 : 0;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "check";
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"tag\":\"device\",\"name\":\"%s\",\"arch\":\"%s\",\"cus\":%d,\"clock_mhz\":%d,\"lds_per_block\":%zu}\n", prop.name,
           prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000, prop.sharedMemPerBlock);
    int fails = 0;
    if (mode == "check") {
        const int S = 4096;
        // every kernel family, both dtypes, delta-only and fused, broadcast and per-tenant masks
        for (int dt : {BD_BF16, BD_F16})
            for (int fused : {0, 1}) {
                for (int v : {0, 1, 2, 3, 4, 5, 6, 7, 100})
                    fails += run_case("tile", 2, 200, 520, 256, dt, BD_F32, fused, 2, v, 0, S);
                if (fused) {        // one-pass fused kernel: multi-tenant, k shorter than the ring (nk = 1, 2), odd N
                    fails += run_case("fx", 2, 200, 520, 256, dt, BD_F32, 1, 2, 8, 0, S);
                    fails += run_case("fx_nk1", 1, 257, 136, 64, dt, dt, 1, 1, 8, 0, S);
                    fails += run_case("fx_nk2", 3, 300, 264, 128, dt, dt, 1, 1, 8, 0, S);
                    fails += run_case("fx_big", 1, 1024, 1024, 2048, dt, dt, 1, 1, 8, 0, S);
                    fails += run_case("fx128", 2, 200, 520, 256, dt, BD_F32, 1, 2, 9, 0, S);
                    fails += run_case("fx128_nk1", 1, 257, 136, 64, dt, dt, 1, 1, 9, 0, S);
                    fails += run_case("fx128_nk2", 3, 300, 264, 128, dt, dt, 1, 1, 9, 0, S);
                    fails += run_case("fx128_big", 1, 640, 1024, 2048, dt, dt, 1, 1, 9, 0, S);
                    fails += run_case("splitk_forced", 2, 200, 520, 256, dt, BD_F32, 1, 2, 10, 0, S);      // KS = 2, 2 k-tiles per slice
                    fails += run_case("splitk_forced2", 1, 128, 384, 512, dt, dt, 1, 1, 10, 0, S);
                    fails += run_case("splitk_auto", 1, 40, 264, 1024, dt, dt, 1, 1, -1, 0, S);
                    fails += run_case("splitk_auto_q", 1, 96, 4096, 4096, dt, dt, 1, 1, -1, 0, S);         // KS = 8
                    fails += run_case("splitk_auto_t3", 3, 33, 1024, 2048, dt, dt, 1, 3, -1, 0, S);
                }
                fails += run_case("tile_bcast", 3, 130, 300, 128, dt, dt, fused, 1, 0, 0, S);
                fails += run_case("auto_big", 1, 512, 768, 1024, dt, dt, fused, 1, -1, 0, S);
                fails += run_case("generic_k96", 2, 17, 40, 96, dt, BD_F32, fused, 2, -1, 0, S);
                fails += run_case("gemv", 6, 1, 1000, 1024, dt, BD_F32, fused, 6, -1, 0, S);
                fails += run_case("gemv_m2", 3, 2, 512, 512, dt, dt, fused, 3, -1, 0, S);
                fails += run_case("gemv_b1", 1, 1, 4096, 4096, dt, dt, fused, 1, -1, 0, S);
                fails += run_case("gemv_bcast16", 16, 1, 256, 2048, dt, BD_F32, fused, 1, -1, 0, S);
                fails += run_case("gemv_m4_t4", 4, 4, 200, 160, dt, dt, fused, 4, -1, 0, S);          // M > 1, ragged N, K % 128 != 0
                fails += run_case("gemv_t16", 16, 1, 520, 1184, dt, BD_F32, fused, 16, -1, 0, S);    // 16 masks, partial last iteration
                fails += run_case("gemv_ks3", 5, 1, 300, 1536, dt, dt, fused, 5, 203, 0, S);
                fails += run_case("gemv_m16", 1, 16, 300, 512, dt, dt, fused, 1, -1, 0, S);           // 16 rows sharing one mask
                fails += run_case("gemv_chunks", 40, 1, 200, 256, dt, BD_F32, fused, 40, -1, 0, S);   // 40 tenants -> 3 chunks of <= 16 rows
                fails += run_case("gemv_chunks_m3", 7, 3, 136, 192, dt, dt, fused, 7, -1, 0, S);      // chunks of 5 batch entries
                for (int v : {300, 400, 500}) {      // every decode kernel family, forced
                    fails += run_case("gemv_forced", 6, 1, 1000, 1024, dt, BD_F32, fused, 6, v, 0, S);
                    fails += run_case("gemv_forced_m4", 4, 4, 200, 160, dt, dt, fused, 4, v, 0, S);
                    fails += run_case("gemv_forced_t16", 16, 1, 520, 1184, dt, BD_F32, fused, 16, v, 0, S);
                    fails += run_case("gemv_forced_ks3", 5, 1, 300, 1536, dt, dt, fused, 5, v + 3, 0, S);
                }
                fails += run_case("col16_m16", 1, 16, 300, 512, dt, dt, fused, 1, 500, 0, S);
                fails += run_case("col16_chunks", 40, 1, 200, 256, dt, BD_F32, fused, 40, 500, 0, S);
                fails += run_case("col16_k8192_t16", 16, 1, 72, 8192, dt, dt, fused, 16, 500, 0, S);      // R*K*2 > LDS -> 2 k slices + reduce
                fails += run_case("col16_q", 6, 1, 4096, 4096, dt, dt, fused, 6, 500, 0, S);
                bd_set_decode_small_lut(0);       // 16-copy sign LUT, forced for fused launches too
                fails += run_case("col16_lut64k", 6, 1, 1000, 1024, dt, BD_F32, fused, 6, 500, 0, S);
                fails += run_case("col16_lut64k_m2", 3, 2, 520, 544, dt, dt, fused, 3, 500, 0, S);
                bd_set_decode_small_lut(-1);
            }
        fails += run_case("edge_m1_tile", 1, 1, 256, 128, BD_BF16, BD_F32, 0, 1, 3, 0, S);
        fails += run_case("edge_n_odd", 1, 70, 77, 64, BD_BF16, BD_BF16, 0, 1, -1, 0, S);
        fails += run_case("edge_n_odd_t", 1, 70, 77, 64, BD_BF16, BD_BF16, 1, 1, 2, 0, S);
        printf("{\"tag\":\"check_done\",\"fails\":%d}\n", fails);
    } else if (mode == "perf") {
        const int it = 20;
        for (int M : {4096, 8192, 16384}) fails += run_case("delta_4096sq", 1, M, 4096, 4096, BD_BF16, BD_BF16, 0, 1, -1, it, 2048);
        for (int v : {-1, 0, 5}) {       // auto (one-pass fused) vs the two-loop kernels
        fails += run_case("fused_8192x4096sq", 1, 8192, 4096, 4096, BD_BF16, BD_BF16, 1, 1, v, it, 2048);
        fails += run_case("fused_4096sq", 1, 4096, 4096, 4096, BD_BF16, BD_BF16, 1, 1, v, it, 2048);
        fails += run_case("fused_2048x4096", 1, 2048, 4096, 4096, BD_BF16, BD_BF16, 1, 1, v, it, 2048);
        fails += run_case("fused_gate", 1, 2048, 11008, 4096, BD_BF16, BD_BF16, 1, 1, v, it, 2048);
        fails += run_case("fused_down", 1, 2048, 4096, 11008, BD_BF16, BD_BF16, 1, 1, v, it, 2048);
        fails += run_case("fused_512", 1, 512, 4096, 4096, BD_BF16, BD_BF16, 1, 1, v, it, 2048);
        fails += run_case("fused_256", 1, 256, 4096, 4096, BD_BF16, BD_BF16, 1, 1, v, it, 2048);
        }
        fails += run_case("fused_4096sq", 1, 4096, 4096, 4096, BD_BF16, BD_BF16, 1, 1, -1, it, 2048);
        fails += run_case("fused_2048x4096", 1, 2048, 4096, 4096, BD_BF16, BD_BF16, 1, 1, -1, it, 2048);
        fails += run_case("fused_gate", 1, 2048, 11008, 4096, BD_BF16, BD_BF16, 1, 1, -1, it, 2048);
        fails += run_case("fused_down", 1, 2048, 4096, 11008, BD_BF16, BD_BF16, 1, 1, -1, it, 2048);
        fails += run_case("c1_128", 1, 128, 4096, 4096, BD_BF16, BD_BF16, 1, 1, -1, it, 2048);
        for (int two : {1})
        for (int v : {-1, 300, 400, 500})
        for (int T : {1, 6, 16}) {
            bd_set_decode_two_launch(two);
            fails += run_case("decode_delta", T, 1, 4096, 4096, BD_BF16, BD_BF16, 0, T, v, 50, 2048);
            fails += run_case(two ? "decode_fused_2launch" : "decode_fused", T, 1, 4096, 4096, BD_BF16, BD_BF16, 1, T, v, 50, 2048);
        }
        bd_set_decode_two_launch(1);
        for (int v : {-1, 300, 400, 500}) {
            fails += run_case("decode_fused_gate", 6, 1, 14336, 4096, BD_F16, BD_F16, 1, 6, v, 50, 2048);
            fails += run_case("decode_fused_down", 6, 1, 4096, 14336, BD_F16, BD_F16, 1, 6, v, 50, 2048);
            fails += run_case("decode_fused_kv", 6, 1, 1024, 4096, BD_F16, BD_F16, 1, 6, v, 50, 2048);
        }
        fails += run_case("decode_fused_gate", 6, 1, 14336, 4096, BD_F16, BD_F16, 1, 6, -1, 50, 2048);
        fails += run_case("decode_fused_down", 6, 1, 4096, 14336, BD_F16, BD_F16, 1, 6, -1, 50, 2048);
        fails += run_case("decode_fused_kv", 6, 1, 1024, 4096, BD_F16, BD_F16, 1, 6, -1, 50, 2048);
        fails += run_case("prefill64_t6", 6, 64, 4096, 4096, BD_F16, BD_F16, 1, 6, -1, it, 2048);
    } else if (mode == "midm") {
        // fused Linear at 16 < M <= 512: split-k one-pass kernel (auto = variant 10 where the rule fires) vs unsplit tiles
        for (int rep = 0; rep < 2; ++rep)
            for (int M : {32, 64, 128, 256, 512}) {
                for (int v : {-1, 9, 2}) {
                    if (v == 2 && M > 64) continue;
                    fails += run_case("midm_q", 1, M, 4096, 4096, BD_BF16, BD_BF16, 1, 1, v, 20, 1024);
                    fails += run_case("midm_gate", 1, M, 11008, 4096, BD_BF16, BD_BF16, 1, 1, v, 20, 1024);
                    fails += run_case("midm_down", 1, M, 4096, 11008, BD_BF16, BD_BF16, 1, 1, v, 20, 1024);
                }
            }
        for (int v : {-1, 9, 2}) fails += run_case("prefill64_t6", 6, 64, 4096, 4096, BD_F16, BD_F16, 1, 6, v, 20, 1024);
    } else if (mode == "chunk") {
        // multi-round shapes: one launch vs single-round launch chunks (dispatcher hook), interleaved
        for (int rep = 0; rep < 2; ++rep)
            for (int on : {1, 0}) {
                bd_set_launch_chunking(on);
                const char* tg = on ? "chunked" : "one_launch";
                fails += run_case(tg, 1, 2048, 11008, 4096, BD_BF16, BD_BF16, 1, 1, -1, 20, 2048);
                fails += run_case(tg, 1, 4096, 4096, 4096, BD_BF16, BD_BF16, 1, 1, -1, 20, 2048);
                fails += run_case(tg, 1, 8192, 4096, 4096, BD_BF16, BD_BF16, 1, 1, -1, 20, 2048);
                fails += run_case(tg, 1, 4096, 11008, 4096, BD_BF16, BD_BF16, 1, 1, -1, 20, 2048);
                fails += run_case(tg, 1, 4096, 4096, 11008, BD_BF16, BD_BF16, 1, 1, -1, 20, 2048);
                fails += run_case(tg, 1, 8192, 4096, 4096, BD_BF16, BD_BF16, 0, 1, -1, 20, 2048);
                fails += run_case(tg, 1, 16384, 4096, 4096, BD_BF16, BD_BF16, 0, 1, -1, 20, 2048);
                fails += run_case(tg, 6, 1024, 4096, 4096, BD_F16, BD_F16, 1, 6, -1, 20, 2048);
                fails += run_case(tg, 1, 3000, 5000, 1024, BD_F16, BD_F32, 1, 1, -1, 20, 4096);      // ragged edges in both directions
            }
        bd_set_launch_chunking(0);
    } else if (mode == "smallm") {
        // fused Linear at 128 < M <= 1024: 256x128 one-pass tile (8) vs 128x128 one-pass tile (9) vs the two-loop 128x256 tile (1)
        for (int rep = 0; rep < 2; ++rep)
            for (int M : {192, 256, 384, 512, 768, 1024})
                for (int v : {8, 9, 1}) {
                    fails += run_case("smallm_q", 1, M, 4096, 4096, BD_BF16, BD_BF16, 1, 1, v, 20, 1024);
                    fails += run_case("smallm_gate", 1, M, 11008, 4096, BD_BF16, BD_BF16, 1, 1, v, 20, 1024);
                }
        fails += run_case("smallm_t6", 6, 200, 4096, 4096, BD_F16, BD_F16, 1, 6, 9, 20, 1024);
        fails += run_case("smallm_t6", 6, 200, 4096, 4096, BD_F16, BD_F16, 1, 6, 8, 20, 1024);
    } else if (mode == "dec_pmc2") {
        // one shape per kernel instantiation (NM = 1 / 6 / 16) so that counters can be keyed by kernel name: 4096^2 fused
        const bool big_lut = argc > 2 && atoi(argv[2]) == 64;
        bd_set_decode_small_lut(big_lut ? 0 : 1);
        for (int v : {500, 300})
            for (int T : {1, 6, 8, 16}) fails += run_case("q", T, 1, 4096, 4096, BD_BF16, BD_BF16, 1, T, v, 5, 0);
        bd_set_decode_small_lut(-1);
    } else if (mode == "wspec") {
        // VALU decode kernel: wave-specialised (4 base + 4 delta waves) vs single-role, fused launches, interleaved
        for (int rep = 0; rep < 2; ++rep)
            for (int on : {1, 0}) {
                bd_set_decode_wave_spec(on);
                const char* tg = on ? "spec" : "plain";
                for (int T : {1, 6, 16}) fails += run_case(tg, T, 1, 4096, 4096, BD_BF16, BD_BF16, 1, T, 300, 50, 2048);
                fails += run_case(tg, 6, 1, 14336, 4096, BD_F16, BD_F16, 1, 6, 300, 50, 2048);
                fails += run_case(tg, 6, 1, 4096, 14336, BD_F16, BD_F16, 1, 6, 300, 50, 2048);
                fails += run_case(tg, 6, 1, 1024, 4096, BD_F16, BD_F16, 1, 6, 300, 50, 2048);
                fails += run_case(tg, 3, 4, 1000, 1184, BD_BF16, BD_F32, 1, 3, 300, 20, 4096);
            }
        bd_set_decode_wave_spec(1);
    } else if (mode == "dec500") {
        // the no-split-k decode kernel with the 16-copy conflict-free sign LUT (default) vs the single 4-KiB table, interleaved
        for (int rep = 0; rep < 2; ++rep)
            for (int small : {0, 1, 2}) {
                bd_set_decode_small_lut(small == 2 ? 1 : small);
                bd_set_decode_generic_loop(small == 2 ? 1 : 0);
                const char* tg = small == 2 ? "lut4k_loop" : small ? "lut4k" : "lut64k";
                for (int T : {1, 6, 8, 16}) {
                    fails += run_case(tg, T, 1, 4096, 4096, BD_BF16, BD_BF16, 0, T, 500, 50, 2048);
                    fails += run_case(tg, T, 1, 4096, 4096, BD_BF16, BD_BF16, 1, T, 500, 50, 2048);
                }
                fails += run_case(tg, 6, 1, 14336, 4096, BD_F16, BD_F16, 1, 6, 500, 50, 2048);
                fails += run_case(tg, 6, 1, 1024, 4096, BD_F16, BD_F16, 1, 6, 500, 50, 2048);
                fails += run_case(tg, 4, 4, 4096, 4096, BD_BF16, BD_BF16, 1, 4, 500, 50, 2048);
            }
        bd_set_decode_small_lut(-1); bd_set_decode_generic_loop(0);
    } else if (mode == "dec600") {
        // streaming decode kernel (600, the automatic choice) vs the round-1 kernels (300 wave-specialised VALU, 500 one-launch col16),
        // warm (Infinity-Cache resident) and cold (HBM) weights.  argv[2] = "quick" -> headline shapes only
        const bool quick = argc > 2 && std::string(argv[2]) == "quick";
        g_tiled = argc > 3 ? (std::string(argv[3]) == "tile" ? 1 : std::string(argv[3]) == "pack" ? 2 : 0) : 0;   // sign-word layout of variant 600
        for (int v : {600, 300, 500}) {
            if (g_tiled && v != 600) continue;
            for (int T : {1, 4, 6, 8}) {
                if (quick && T != 6 && T != 1) continue;
                fails += run_decode_cold("dec_4096sq", T, 1, 4096, 4096, BD_F16, 1, v, 200);
                if (!quick) fails += run_decode_cold("dec_4096sq_delta", T, 1, 4096, 4096, BD_F16, 0, v, 200);
            }
            fails += run_decode_cold("dec_qkv6144", 6, 1, 6144, 4096, BD_F16, 1, v, 200);
            fails += run_decode_cold("dec_gateup28672", 6, 1, 28672, 4096, BD_F16, 1, v, 100);
            fails += run_decode_cold("dec_down14336", 6, 1, 4096, 14336, BD_F16, 1, v, 100);
            if (!quick) {
                fails += run_decode_cold("dec_gate14336", 6, 1, 14336, 4096, BD_F16, 1, v, 100);
                fails += run_decode_cold("dec_kv1024", 6, 1, 1024, 4096, BD_F16, 1, v, 200);
                fails += run_decode_cold("dec_gate11008", 1, 1, 11008, 4096, BD_BF16, 1, v, 100);
                fails += run_decode_cold("dec_m4_t4", 4, 4, 4096, 4096, BD_BF16, 1, v, 200);
                fails += run_decode_cold("dec_70b_gateup_shard", 1, 1, 7168, 8192, BD_BF16, 1, v, 100);
            }
        }
    } else if (mode == "stream_ab") {
        // A/B matrix of the streaming decode kernel (bd_set_stream_tuning: bit 0 natural-order W, bit 1 default cache policy, bit 2
        // 4-wave blocks, bit 3 deeper prefetch) x sign-word layout (reference [K/32, N] vs tile-major), cold weights.
        // argv[2] = list of tune codes (default: all 16)
        std::vector<int> tunes;
        if (argc > 2) { for (char* t = strtok(argv[2], ","); t; t = strtok(nullptr, ",")) tunes.push_back(atoi(t)); }
        else for (int t = 0; t < 16; ++t) tunes.push_back(t);
        for (int tune : tunes) {
            bd_set_stream_tuning(tune);
            char tg[64];
            snprintf(tg, sizeof tg, "ab%02d_stream_4096sq", tune);   fails += run_tenant_cold(tg, 1, 4096, 4096, 200);
            snprintf(tg, sizeof tg, "ab%02d_stream_28672", tune);    fails += run_tenant_cold(tg, 1, 28672, 4096, 60);
            for (int tiled : {0, 1, 2}) {
                g_tiled = tiled;
                const char* ly = tiled == 2 ? "pack" : tiled ? "tile" : "ref";
                snprintf(tg, sizeof tg, "ab%02d_%s_T1_4096sq", tune, ly);    fails += run_decode_cold(tg, 1, 1, 4096, 4096, BD_F16, 1, 600, 200);
                snprintf(tg, sizeof tg, "ab%02d_%s_T6_4096sq", tune, ly);    fails += run_decode_cold(tg, 6, 1, 4096, 4096, BD_F16, 1, 600, 200);
                snprintf(tg, sizeof tg, "ab%02d_%s_T6_qkv", tune, ly);       fails += run_decode_cold(tg, 6, 1, 6144, 4096, BD_F16, 1, 600, 200);
                snprintf(tg, sizeof tg, "ab%02d_%s_T6_gateup", tune, ly);    fails += run_decode_cold(tg, 6, 1, 28672, 4096, BD_F16, 1, 600, 60);
                snprintf(tg, sizeof tg, "ab%02d_%s_T6_down", tune, ly);      fails += run_decode_cold(tg, 6, 1, 4096, 14336, BD_F16, 1, 600, 60);
                snprintf(tg, sizeof tg, "ab%02d_%s_T4_4096sq", tune, ly);    fails += run_decode_cold(tg, 4, 1, 4096, 4096, BD_F16, 1, 600, 200);
            }
            g_tiled = 0;
        }
        bd_set_stream_tuning(0);
        fails += run_decode_cold("ref300_T6_4096sq", 6, 1, 4096, 4096, BD_F16, 1, 300, 200);
        fails += run_decode_cold("ref300_T6_gateup", 6, 1, 28672, 4096, BD_F16, 1, 300, 60);
    } else if (mode == "dec1") {
        // one decode case for rocprofv3 passes: dec1 <T> <N> <K> <variant> <tiled 0|1> [iters]
        if (argc < 7) { fprintf(stderr, "usage: dec1 T N K variant tiled [iters]\n"); return 2; }
        g_tiled = atoi(argv[6]);
        fails += run_decode_cold("dec1", atoi(argv[2]), 1, atoi(argv[3]), atoi(argv[4]), BD_F16, 1, atoi(argv[5]), argc > 7 ? atoi(argv[7]) : 20);
    } else if (mode == "dec600_pmc") {
        // few launches of the headline decode shapes for rocprofv3 --kernel-trace / --pmc passes (streaming kernel, tile-major signs;
        // then the round-1 kernels on the reference layout for comparison)
        g_tiled = 1;
        for (int T : {1, 6}) fails += run_decode_cold("pmc_4096sq", T, 1, 4096, 4096, BD_F16, 1, 600, 20);
        fails += run_decode_cold("pmc_qkv6144", 6, 1, 6144, 4096, BD_F16, 1, 600, 20);
        fails += run_decode_cold("pmc_gateup28672", 6, 1, 28672, 4096, BD_F16, 1, 600, 20);
        fails += run_decode_cold("pmc_down14336", 6, 1, 4096, 14336, BD_F16, 1, 600, 20);
        g_tiled = 0;
        for (int v : {500, 300}) fails += run_decode_cold("pmc_4096sq", 6, 1, 4096, 4096, BD_F16, 1, v, 20);
    } else if (mode == "notebook") {
        // the shapes the reference's notebook publishes (fp16, FLOP = 2*B*M*N*K, delta-only kernels)
        for (int NK : {4096, 8192}) {
            fails += run_case("nb_matmul_m1", 1, 1, NK, NK, BD_F16, BD_F16, 0, 1, -1, 100, 1024);
            fails += run_case("nb_matmul_m16", 1, 16, NK, NK, BD_F16, BD_F16, 0, 1, -1, 100, 1024);
            fails += run_case("nb_bmm_b16", 16, 1, NK, NK, BD_F16, BD_F16, 0, 16, -1, 100, 1024);
            fails += run_case("nb_bmm_b8", 8, 1, NK, NK, BD_F16, BD_F16, 0, 8, -1, 100, 1024);
            fails += run_case("nb_bmm_b1", 1, 1, NK, NK, BD_F16, BD_F16, 0, 1, -1, 100, 1024);
        }
        // the blog's "one Linear, decode" plots: B = 8 tenants at hidden ~10k; N = K = 8192 with 4..64 tenants (fused: base + deltas)
        fails += run_case("blog_b8_h10240", 8, 1, 10240, 10240, BD_F16, BD_F16, 1, 8, -1, 50, 1024);
        for (int T : {4, 16, 64}) fails += run_case("blog_8192_models", T, 1, 8192, 8192, BD_F16, BD_F16, 1, T, -1, 30, 1024);
    } else if (mode == "dec_pmc") {
        // few launches of each decode kernel family for rocprofv3 counter passes (kernel name = family, grid size = shape)
        for (int two : {1, 0})
        for (int v : {300, 400}) {
            bd_set_decode_two_launch(two);
            if (!two && v == 300) continue;
            fails += run_case("q_t1", 1, 1, 4096, 4096, BD_BF16, BD_BF16, 1, 1, v, 5, 0);
            fails += run_case("q_t6", 6, 1, 4096, 4096, BD_BF16, BD_BF16, 1, 6, v, 5, 0);
            fails += run_case("gate_t6", 6, 1, 14336, 4096, BD_BF16, BD_BF16, 1, 6, v, 5, 0);
            fails += run_case("q_t16_delta", 16, 1, 4096, 4096, BD_BF16, BD_BF16, 0, 16, v, 5, 0);
        }
    } else if (mode == "bits") {
        bits_bench();
    } else if (mode == "ks") {
        for (int ks : {1, 2, 4, 8}) {
            fails += run_case("gate_t6", 6, 1, 14336, 4096, BD_F16, BD_F16, 1, 6, 200 + ks, 50, 1024);
            fails += run_case("q_t6", 6, 1, 4096, 4096, BD_F16, BD_F16, 1, 6, 200 + ks, 50, 1024);
            fails += run_case("kv_t6", 6, 1, 1024, 4096, BD_F16, BD_F16, 1, 6, 200 + ks, 50, 1024);
            fails += run_case("down_t6", 6, 1, 4096, 14336, BD_F16, BD_F16, 1, 6, 200 + ks, 50, 1024);
        }
        fails += run_case("down_t6", 6, 1, 4096, 14336, BD_F16, BD_F16, 1, 6, 200 + 14, 50, 1024);
        fails += run_case("down_t6", 6, 1, 4096, 14336, BD_F16, BD_F16, 1, 6, 200 + 28, 50, 1024);
    } else if (mode == "pfab") {
        // shipped (full-tile ping-pong) vs half-tile ping-pong through the dispatcher, model shapes, 3 interleaved rounds
        for (int rep = 0; rep < 3; ++rep)
            for (int v : {5, 7}) {
                fails += run_case("fused_q", 1, 2048, 4096, 4096, BD_BF16, BD_BF16, 1, 1, v, 20, 1024);
                fails += run_case("fused_gate", 1, 2048, 11008, 4096, BD_BF16, BD_BF16, 1, 1, v, 20, 1024);
                fails += run_case("fused_down", 1, 2048, 4096, 11008, BD_BF16, BD_BF16, 1, 1, v, 20, 1024);
                fails += run_case("delta_q", 1, 2048, 4096, 4096, BD_BF16, BD_BF16, 0, 1, v, 20, 1024);
            }
        for (int rep = 0; rep < 3; ++rep)
            for (int v : {0, 6}) fails += run_case("delta_4096", 1, 4096, 4096, 4096, BD_BF16, BD_BF16, 0, 1, v, 20, 1024);
    } else if (mode == "pf_pmc") {
        const int M = 4096, N = 4096, K = 4096, iters = 10;
        CFGPP("pp_ns4_noprio", DT_BF16, 256, 256, 2, 4, 4, false, false, 2);
        CFGPF("pf_ns4", DT_BF16, 256, 256, 2, 4, 4, false, false, 0);
        CFGPF("pf_ns4_lut", DT_BF16, 256, 256, 2, 4, 4, false, false, 1);
        CFG("v1_256x256_2x4_ns4", DT_BF16, 256, 256, 2, 4, 4, false, false, 0);
    } else if (mode == "fx") {
        const int M = argc > 2 ? atoi(argv[2]) : 2048, N = argc > 3 ? atoi(argv[3]) : 4096, K = argc > 4 ? atoi(argv[4]) : 4096;
        sweep_fx(M, N, K, argc > 5 ? atoi(argv[5]) : 20);
    } else if (mode == "order") {
        const int M = argc > 2 ? atoi(argv[2]) : 2048, N = argc > 3 ? atoi(argv[3]) : 4096, K = argc > 4 ? atoi(argv[4]) : 4096;
        sweep_order(M, N, K, argc > 5 ? atoi(argv[5]) : 20);
    } else if (mode == "pf") {
        sweep_pf(argc > 2 ? atoi(argv[2]) : 4096, 4096, argc > 3 ? atoi(argv[3]) : 4096, 30);
    } else if (mode == "sp") {
        sweep_sp(argc > 2 ? atoi(argv[2]) : 4096, 4096, argc > 3 ? atoi(argv[3]) : 4096, 30);
    } else if (mode == "pp") {
        const int M = argc > 2 ? atoi(argv[2]) : 4096;
        sweep_pp(M, 4096, argc > 3 ? atoi(argv[3]) : 4096, 20);
    } else if (mode == "ablate") {
        ablate(4096, 4096, 4096, 20);
    } else if (mode == "ablate_pp") {
#define ABLP(name, opt) run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + (opt)>, 1>(name, 4096, 4096, 4096, 30, 0)
        ABLP("pp_base", 0);
        ABLP("pp_base_nostore", 128);
        ABLP("pp_base", 0);
#undef ABLP
    } else if (mode == "fixed") {
        // fixed-cost anatomy at K = 64 (one k-tile): full / no C stores / empty kernel, timed by rocprofv3 kernel-trace
        run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2>, 1>("full", 4096, 4096, 64, 10, 0);
        run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + 1024>, 1>("full_plain_store", 4096, 4096, 64, 10, 0);
        run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2>, 1>("k4096_nt", 4096, 4096, 4096, 10, 64);
        run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + 1024>, 1>("k4096_plain", 4096, 4096, 4096, 10, 64);
        run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + 128>, 1>("no_store", 4096, 4096, 64, 10, 0);
        run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + 256>, 1>("empty", 4096, 4096, 64, 10, 0);
        run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + 128>, 1>("no_store_k4096", 4096, 4096, 4096, 10, 0);
    } else if (mode == "one_pp") {
        const int M = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
        run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2>, 1>("pp_ns4_noprio", M, 4096, K, 10, 256);
    } else if (mode == "one") {
        // a single configuration, a few launches: for rocprofv3 counter passes
        const int M = argc > 2 ? atoi(argv[2]) : 4096;
        run_cfg<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 0>>("256x256_2x4_ns4", M, 4096, 4096, 10, 256);
    } else if (mode == "sweep") {
        const int M = argc > 2 ? atoi(argv[2]) : 4096;
        sweep(M, 4096, 4096, 20);
    }
    return fails ? 1 : 0;
}
