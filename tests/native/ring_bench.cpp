// Decode-kernel A/B harness against the SHIPPED library (links libbitdelta_hip.so; C ABI only): the streaming register-load kernel
// (variant 600) vs the LDS-DMA loader / consumer kernel (variant 700) on packed multi-tenant decode launches, cold weights.
// Development / measurement tool (pytest -m gpu holds the parity tests proper).  One JSON object per line.
//
//   ring_bench check [quick]                       correctness grid of variant 700 (and 600 as the control) vs an exact host reference
//   ring_bench one T M N K dt wtile variant tune [iters]     one configuration: check + warm / cold timing
//   ring_bench ab [iters]                          the Mistral-7B / Llama-2-7B decode shapes x {600, 700 x tunings}, cold
//
// "cold": every launch reads a different device copy of W / signs, the copies rotate through > 2x the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/bitdelta_hip.h"
#include "../../include/bitdelta_hip_test.h"

#define HIPCHECK(x)                                                                                  \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
            exit(2);                                                                                 \
        }                                                                                            \
    } while (0)

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
static int tpad_of(int T) { return T <= 1 ? 1 : T <= 2 ? 2 : T <= 4 ? 4 : T <= 6 ? 6 : 8; }

struct Case {
    int T, M, N, K, dt, wtile;
    std::vector<uint16_t> X, W;
    std::vector<uint32_t> P;          // reference layout [T][K/32][N]
    std::vector<float> alpha;
    void *dX = nullptr, *dW = nullptr, *dP = nullptr, *dAl = nullptr, *dY = nullptr;
    size_t wbytes = 0, pbytes = 0;
    int tp = 1;
};

static void make_case(Case& c) {
    const int KW = c.K / 32, NT = (c.N + 15) / 16, NIT = (c.K + 127) / 128;
    c.tp = tpad_of(c.T);
    c.X.resize((size_t)c.T * c.M * c.K);
    for (auto& v : c.X) v = f2h(nrand(), c.dt);
    c.W.resize((size_t)c.N * c.K);
    for (auto& v : c.W) v = f2h(0.02f * nrand(), c.dt);
    c.P.resize((size_t)c.T * KW * c.N);
    for (auto& v : c.P) v = (uint32_t)rng();
    c.alpha.resize(c.T);
    for (auto& v : c.alpha) v = 4e-4f * (0.75f + 0.5f * urand());
    // packed decode layout [N/16][K/128][4 g][16][tp]: byte s of the dword of (tile, it, g, col, t) = byte g of word row 4 it + s
    std::vector<uint32_t> pk((size_t)NT * NIT * 4 * 16 * c.tp, 0u);
    for (int t = 0; t < c.T; ++t)
        for (int i = 0; i < KW; ++i)
            for (int n = 0; n < c.N; ++n) {
                const uint32_t w = c.P[((size_t)t * KW + i) * c.N + n];
                const int it = i >> 2, sidx = i & 3;
                for (int g = 0; g < 4; ++g)
                    pk[((((size_t)(n >> 4) * NIT + it) * 4 + g) * 16 + (n & 15)) * c.tp + t] |= ((w >> (8 * g)) & 0xffu) << (8 * sidx);
            }
    c.pbytes = pk.size() * 4;
    HIPCHECK(hipMalloc(&c.dP, c.pbytes));
    HIPCHECK(hipMemcpy(c.dP, pk.data(), c.pbytes, hipMemcpyHostToDevice));
    if (c.wtile) {     // tile-major W'[n/16][k/128][s][n%16][g][8], k = 128 it + 32 s + 8 g + e
        std::vector<uint16_t> wt((size_t)NT * NIT * 2048, 0);
        for (int n = 0; n < c.N; ++n)
            for (int k = 0; k < c.K; ++k) {
                const int it = k >> 7, s = (k >> 5) & 3, g = (k >> 3) & 3, e = k & 7;
                wt[(((size_t)(n >> 4) * NIT + it) * 4 + s) * 512 + (n & 15) * 32 + g * 8 + e] = c.W[(size_t)n * c.K + k];
            }
        c.wbytes = wt.size() * 2;
        HIPCHECK(hipMalloc(&c.dW, c.wbytes));
        HIPCHECK(hipMemcpy(c.dW, wt.data(), c.wbytes, hipMemcpyHostToDevice));
    } else {
        c.wbytes = c.W.size() * 2;
        HIPCHECK(hipMalloc(&c.dW, c.wbytes));
        HIPCHECK(hipMemcpy(c.dW, c.W.data(), c.wbytes, hipMemcpyHostToDevice));
    }
    HIPCHECK(hipMalloc(&c.dX, c.X.size() * 2));
    HIPCHECK(hipMemcpy(c.dX, c.X.data(), c.X.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&c.dAl, c.T * 4));
    HIPCHECK(hipMemcpy(c.dAl, c.alpha.data(), c.T * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&c.dY, (size_t)c.T * c.M * c.N * 2));
}
static void free_case(Case& c) { hipFree(c.dX); hipFree(c.dW); hipFree(c.dP); hipFree(c.dAl); hipFree(c.dY); }

static int launch(const Case& c, const void* W, const void* P) {
    const int64_t sPb = c.T == 1 ? 0 : (int64_t)(c.K / 32) * ((c.N + 15) / 16 * 16);
    return bd_binary_linear_decode(c.dX, W, (const int32_t*)P, 2, c.tp, (const float*)c.dAl, c.dY, c.T, c.M, c.N, c.K, (int64_t)c.M * c.K, c.K,
                                   c.wtile ? 0 : c.K, sPb, 1, 1, (int64_t)c.M * c.N, c.N, c.dt, c.dt, 0, 0);
}

static double ref_value(const Case& c, int t, int m, int n) {
    const uint16_t* a = &c.X[((size_t)t * c.M + m) * c.K];
    const uint32_t* p = &c.P[(size_t)t * (c.K / 32) * c.N];
    double d = 0.0, base = 0.0;
    for (int k = 0; k < c.K; ++k) {
        const double x = h2f(a[k], c.dt);
        d += ((p[(size_t)(k >> 5) * c.N + n] >> (k & 31)) & 1u) ? x : -x;
        base += x * (double)h2f(c.W[(size_t)n * c.K + k], c.dt);
    }
    return base + (double)c.alpha[t] * d;
}
static int check(const Case& c, int nsamples, double* max_ulp) {
    std::vector<uint16_t> h((size_t)c.T * c.M * c.N);
    HIPCHECK(hipMemcpy(h.data(), c.dY, h.size() * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    *max_ulp = 0;
    const size_t tot = h.size();
    const bool all = tot <= (size_t)nsamples;
    for (size_t s = 0; s < (all ? tot : (size_t)nsamples); ++s) {
        size_t idx = all ? s : (size_t)(rng() % tot);
        if (!all && s < 64) {      // corners and edges
            const int tt = (s & 1) ? c.T - 1 : 0, mm = (s & 2) ? c.M - 1 : 0, nn = (s & 4) ? c.N - 1 - (int)(s >> 3) : (int)((s * 7919) % c.N);
            idx = ((size_t)tt * c.M + mm) * c.N + nn;
        }
        const int n = (int)(idx % c.N), m = (int)((idx / c.N) % c.M), t = (int)(idx / ((size_t)c.N * c.M));
        const double r = ref_value(c, t, m, n), got = h2f(h[idx], c.dt);
        int ex;
        frexp(r, &ex);
        const double ulp = ldexp(1.0, ex - 1 - (c.dt == BD_BF16 ? 7 : 10));
        const double tol = 0.5 * ulp + 4e-6 * (fabs(r) + sqrt((double)c.K)) + (c.dt == BD_F16 ? 6e-8 : 0);
        if (ulp > 0) *max_ulp = fmax(*max_ulp, fabs(got - r) / ulp);
        if (!(fabs(got - r) <= tol)) {
            if (bad < 4) fprintf(stderr, "  MISMATCH t=%d m=%d n=%d got=%.8g ref=%.8g tol=%.3g\n", t, m, n, got, r, tol);
            ++bad;
        }
    }
    return bad;
}

template <class F>
static double time_us(F&& f, int warm, int iters) {
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
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms * 1e3 / iters;
}

// one configuration on an existing case: returns bad count; prints a JSON line
// variant 600: `tune` = bd_set_stream_tuning flags (16 = nt on the tile-major weight loads, 32 = default policy); 700: bd_set_ring_tuning
static int run_cfg(const char* tag, Case& c, int variant, int tune, int iters, std::vector<void*>* Ws, std::vector<void*>* Ps) {
    bd_set_gemm_variant(variant);
    if (variant == 700) bd_set_ring_tuning(tune);
    else bd_set_stream_tuning(tune < 0 ? 0 : tune);
    HIPCHECK(hipMemset(c.dY, 0xff, (size_t)c.T * c.M * c.N * 2));      // NaN-poison: an unwritten output cannot pass
    const int rc = launch(c, c.dW, c.dP);
    const hipError_t herr = hipDeviceSynchronize();
    const int used = bd_last_gemm_variant();
    int bad = -1;
    double max_ulp = 0, warm = 0, cold = 0;
    if (rc == 0 && herr == hipSuccess) bad = check(c, 3072, &max_ulp);
    const double bytes = 2.0 * c.N * c.K + (double)c.T * c.K * c.N / 8 + 2.0 * c.T * c.M * (c.K + c.N);
    if (rc == 0 && herr == hipSuccess && bad == 0 && iters > 0) {
        warm = time_us([&] { launch(c, c.dW, c.dP); }, 5, iters);
        if (Ws && !Ws->empty()) {
            size_t idx = 0;
            const size_t ns = Ws->size();
            cold = time_us([&] { launch(c, (*Ws)[idx], (*Ps)[idx]); idx = (idx + 1) % ns; }, (int)ns, iters);
        }
    }
    printf("{\"tag\":\"%s\",\"T\":%d,\"M\":%d,\"N\":%d,\"K\":%d,\"dt\":\"%s\",\"wtile\":%d,\"variant\":%d,\"tune\":%d,\"used\":%d,\"rc\":%d,\"hip\":%d,"
           "\"bad\":%d,\"max_ulp\":%.3g,\"MB\":%.1f,\"warm_us\":%.2f,\"warm_gbps\":%.0f,\"cold_us\":%.2f,\"cold_gbps\":%.0f}\n",
           tag, c.T, c.M, c.N, c.K, c.dt == BD_BF16 ? "bf16" : "f16", c.wtile, variant, tune, used, rc, (int)herr, bad, max_ulp, bytes * 1e-6, warm,
           warm > 0 ? bytes / warm * 1e-3 : 0.0, cold, cold > 0 ? bytes / cold * 1e-3 : 0.0);
    fflush(stdout);
    bd_set_gemm_variant(-1);
    bd_set_ring_tuning(-1);
    bd_set_stream_tuning(0);
    if (herr != hipSuccess) exit(3);
    return (rc != 0 || bad != 0) ? 1 : 0;
}

static void make_cold_sets(const Case& c, std::vector<void*>& Ws, std::vector<void*>& Ps) {
    int nset = (int)(600e6 / (double)(c.wbytes + c.pbytes)) + 1;
    if (nset < 2) nset = 2;
    if (nset > 24) nset = 24;
    Ws.assign(nset, nullptr);
    Ps.assign(nset, nullptr);
    for (int i = 0; i < nset; ++i) {
        HIPCHECK(hipMalloc(&Ws[i], c.wbytes)); HIPCHECK(hipMemcpy(Ws[i], c.dW, c.wbytes, hipMemcpyDeviceToDevice));
        HIPCHECK(hipMalloc(&Ps[i], c.pbytes)); HIPCHECK(hipMemcpy(Ps[i], c.dP, c.pbytes, hipMemcpyDeviceToDevice));
    }
}
static void free_sets(std::vector<void*>& Ws, std::vector<void*>& Ps) {
    for (auto p : Ws) hipFree(p);
    for (auto p : Ps) hipFree(p);
    Ws.clear(); Ps.clear();
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "check";
    int fails = 0;
    if (mode == "check") {
        const bool quick = argc > 2 && !strcmp(argv[2], "quick");
        struct Sh { int T, M, N, K; };
        std::vector<Sh> shapes = {{1, 1, 512, 512},   {2, 1, 528, 640},   {3, 1, 1000, 1024}, {4, 1, 1024, 2048}, {5, 1, 2048, 1152},
                                  {6, 1, 4096, 4096}, {8, 1, 1536, 4096}, {6, 2, 1040, 1024}, {4, 4, 4096, 2048}, {1, 16, 768, 1024},
                                  {6, 1, 6144, 4096}, {2, 8, 520, 768},   {6, 1, 4096, 14336}};
        if (quick) shapes.resize(6);
        for (const Sh& sh : shapes)
            for (int dt : {BD_F16, BD_BF16})
                for (int wtile : {0, 1}) {
                    if (wtile && (sh.N % 16 || sh.M != 1)) continue;
                    if (quick && dt == BD_BF16 && wtile) continue;
                    Case c{sh.T, sh.M, sh.N, sh.K, dt, wtile};
                    make_case(c);
                    fails += run_cfg("check", c, 600, 32, 0, nullptr, nullptr);                      // control: the checker itself
                    if (wtile) fails += run_cfg("check", c, 600, 16, 0, nullptr, nullptr);           // nt weight loads
                    if (wtile) fails += run_cfg("check", c, 600, 16 | 64, 0, nullptr, nullptr);      // + resident activations (inside its envelope)
                    for (int tune : {1, 9, 5, 0, 3, 11, 7, 1 | (8 << 8), 11 | (8 << 8)}) fails += run_cfg("check", c, 700, tune, 0, nullptr, nullptr);
                    free_case(c);
                }
    } else if (mode == "one") {
        if (argc < 10) { fprintf(stderr, "usage: one T M N K dt(0 f16 | 1 bf16) wtile variant tune [iters]\n"); return 2; }
        Case c{atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]) ? BD_BF16 : BD_F16, atoi(argv[7])};
        make_case(c);
        std::vector<void*> Ws, Ps;
        make_cold_sets(c, Ws, Ps);
        fails += run_cfg("one", c, atoi(argv[8]), atoi(argv[9]), argc > 10 ? atoi(argv[10]) : 100, &Ws, &Ps);
        free_sets(Ws, Ps);
        free_case(c);
    } else if (mode == "ab") {
        const int iters = argc > 2 ? atoi(argv[2]) : 100;
        struct Sh { const char* name; int T, N, K; };
        const Sh shapes[] = {{"o_4096sq", 6, 4096, 4096},   {"qkv_6144", 6, 6144, 4096}, {"gateup_28672", 6, 28672, 4096}, {"down_14336", 6, 4096, 14336},
                             {"o_T1", 1, 4096, 4096},       {"o_T4", 4, 4096, 4096},     {"llama_qkv_12288_T1", 1, 12288, 4096},
                             {"llama_gateup_22016_T1", 1, 22016, 4096}, {"llama_down_11008_T1", 1, 4096, 11008}};
        for (const Sh& sh : shapes) {
            Case c{sh.T, 1, sh.N, sh.K, BD_F16, 1};
            make_case(c);
            std::vector<void*> Ws, Ps;
            make_cold_sets(c, Ws, Ps);
            for (int rep = 0; rep < 2; ++rep) {
                fails += run_cfg(sh.name, c, 600, 32 | 128, iters, &Ws, &Ps);       // default policy
                fails += run_cfg(sh.name, c, 600, 16 | 128, iters, &Ws, &Ps);       // nt weight loads
                fails += run_cfg(sh.name, c, 600, 16 | 64, iters, &Ws, &Ps);        // nt + resident activations, deeper prefetch
                fails += run_cfg(sh.name, c, 600, 32 | 64, iters, &Ws, &Ps);
                for (int tune : {9, 1}) fails += run_cfg(sh.name, c, 700, tune, iters, &Ws, &Ps);
            }
            free_sets(Ws, Ps);
            free_case(c);
        }
    } else {
        fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    }
    printf("{\"fails\":%d}\n", fails);
    return fails ? 1 : 0;
}
