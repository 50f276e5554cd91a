// Development harness for the 4-wave persistent kernels (bd_gemm_w4.h): bitwise check against the shipped 8-wave kernels
// (bd_gemm_pf.h / bd_gemm_fx.h -- same per-element MFMA accumulation sequence, so the outputs must be IDENTICAL) and an
// independent fp32 device reference, then event-timed launches.  Not part of the product or of pytest.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o w4_bench w4_bench.hip
//   ./w4_bench [reps] [M list comma separated]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../bitdelta_amd/csrc/bd_gemm_pf.h"
#include "../../bitdelta_amd/csrc/bd_gemm_fx.h"
#include "../../bitdelta_amd/csrc/bd_gemm_w4.h"

using namespace bd;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill_bf16(unsigned short* x, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        // two hashed uniforms -> Box-Muller normal
        uint32_t a = (uint32_t)i * 2654435761u + seed, c = ((uint32_t)(i >> 32) + 0x9e3779b9u) ^ (seed * 40503u);
        a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; a += c;
        uint32_t d = a * 0x9e3779b1u + 0x85ebca6bu; d ^= d >> 13; d *= 0xc2b2ae35u; d ^= d >> 16;
        const float u1 = ((a >> 8) + 1) * (1.0f / 16777217.0f), u2 = (d >> 8) * (1.0f / 16777216.0f);
        const float v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * scale;
        x[i] = (unsigned short)f32_to_bf16_bits(v);
    }
}
__global__ void fill_u32(uint32_t* x, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t a = (uint32_t)i * 2246822519u + seed; a ^= a >> 15; a *= 0x2c1b3c6du; a ^= a >> 12; a *= 0x297a2d39u; a ^= a >> 15;
        x[i] = a;
    }
}
__global__ void count_diff(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < n; i += stride) c += a[i] != b[i];
    if (c) atomicAdd(out, c);
}
// independent reference on sampled rows: fp32 sequential, natural k order
__global__ void ref_rows(const unsigned short* X, const uint32_t* P, const unsigned short* W, float alpha, const unsigned short* C, int M,
                         int N, int K, int nrows, const int* rows, float* maxerr) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, ri = blockIdx.y;
    if (n >= N || ri >= nrows) return;
    const int m = rows[ri];
    double accS = 0, accW = 0;
    for (int k = 0; k < K; ++k) {
        const float x = bf16_bits_to_f32(X[(size_t)m * K + k]);
        const int bit = (P[(size_t)(k >> 5) * N + n] >> (k & 31)) & 1;
        accS += bit ? x : -x;
        if (W) accW += (double)x * bf16_bits_to_f32(W[(size_t)n * K + k]);
    }
    const double ref = W ? accW + (double)alpha * accS : accS;
    const float got = bf16_bits_to_f32(C[(size_t)m * N + n]);
    const float err = fabsf(got - (float)ref) / (fabsf((float)ref) + 1e-2f * sqrtf((float)K));
    atomicMax((int*)maxerr, __float_as_int(err));
}

static int g_cus = 256;

template <class Cfg>
static void launch_w4(const GemmParams& p0, int B, hipStream_t st, int grid_cap = 0) {
    auto kern = delta_gemm_w4_kernel<Cfg>;
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES)); done = true; }
    GemmParams p = p0;
    p.tiles_m = (p.M + Cfg::BM - 1) / Cfg::BM; p.tiles_n = (p.N + Cfg::BN - 1) / Cfg::BN;
    const int nt = p.tiles_m * p.tiles_n;
    int G = std::min(nt, grid_cap > 0 ? grid_cap : g_cus);
    p.nbatch = B;
    G = std::min(nt * B, grid_cap > 0 ? grid_cap : g_cus);
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), Cfg::LDS_BYTES, st, p);
}
template <class Cfg, class K>
static void launch_old(K kern, const GemmParams& p0, int B, hipStream_t st) {
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES)); done = true; }
    GemmParams p = p0;
    p.tiles_m = (p.M + Cfg::BM - 1) / Cfg::BM; p.tiles_n = (p.N + Cfg::BN - 1) / Cfg::BN;
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, B), dim3(Cfg::NT), Cfg::LDS_BYTES, st, p);
}

template <class F>
static void time_it(const char* name, F&& f, int reps, double flops, double* out_us = nullptr) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    double avg = 0; for (float t : ts) avg += t; avg /= ts.size();
    const double med = ts[ts.size() / 2], mn = ts[0];
    printf("  %-34s med %8.2f us  min %8.2f  avg %8.2f   %7.1f TF (med)  %7.1f TF (avg)  frac %.3f\n", name, med * 1e3, mn * 1e3, avg * 1e3,
           flops / (med * 1e-3) / 1e12, flops / (avg * 1e-3) / 1e12, flops / (avg * 1e-3) / 2.5e15);
    if (out_us) *out_us = med * 1e3;
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

#ifdef BD_TRACE
template <class Cfg>
static void trace_one(const char* name, const GemmParams& p, double flops) {
    unsigned long long z[16][8]; memset(z, 0, sizeof(z));
    for (int i = 0; i < 40; ++i) launch_w4<Cfg>(p, 1, 0);            // steady-state clocks: the stamps below are from the LAST launch
    CK(hipDeviceSynchronize());
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bd_trace_w4), z, sizeof(z)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) launch_w4<Cfg>(p, 1, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(bd_trace_w4), sizeof(z)));
    printf("  trace %-20s launch %.2f us (%.1f TF); stamps of workgroup 0 / wave 0 in s_memtime ticks:\n", name, ms * 1e3, flops / (ms * 1e-3) / 1e12);
    const unsigned long long t00 = z[14][5];
    unsigned long long last = 0;
    for (int r = 0; r < 14 && z[r][0]; ++r) {
        printf("    tile %d: loop entry @%llu  k-tiles 8..39: %.1f ticks/k-tile  loop total %llu (%.1f/k-tile)  epilogue %llu  tile total %llu\n", r,
               z[r][0] - t00, (double)(z[r][2] - z[r][1]) / 32.0, z[r][3] - z[r][0], (double)(z[r][3] - z[r][0]) / (p.K / 64), z[r][4] - z[r][3],
               z[r][4] - z[r][0]);
        last = z[r][4];
    }
    printf("    kernel entry -> last epilogue done: %llu ticks; launch time / ticks = %.3f ns per tick\n", last - t00, ms * 1e6 / (double)(last - t00));
}
#endif

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    std::vector<int> Ms = {4096, 8192, 16384};
    if (argc > 2) { Ms.clear(); for (char* t = strtok(argv[2], ","); t; t = strtok(nullptr, ",")) Ms.push_back(atoi(t)); }
    const std::string what = argc > 3 ? argv[3] : "all";
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, g_cus);
    const int N = 4096, K = 4096, Mmax = *std::max_element(Ms.begin(), Ms.end());
    unsigned short *X, *W, *C0, *C1; uint32_t* P; float* alpha; unsigned long long* dcount; float* derr; int* drows;
    CK(hipMalloc(&X, (size_t)Mmax * K * 2)); CK(hipMalloc(&W, (size_t)12288 * K * 2)); CK(hipMalloc(&P, (size_t)(K / 32) * 12288 * 4));
    CK(hipMalloc(&C0, (size_t)Mmax * 12288 * 2)); CK(hipMalloc(&C1, (size_t)Mmax * 12288 * 2));
    CK(hipMalloc(&alpha, 4)); CK(hipMalloc(&dcount, 8)); CK(hipMalloc(&derr, 4)); CK(hipMalloc(&drows, 64 * 4));
    fill_bf16<<<2048, 256>>>(X, (size_t)Mmax * K, 1u, 1.0f);
    fill_bf16<<<2048, 256>>>(W, (size_t)12288 * K, 2u, 0.02f);
    fill_u32<<<2048, 256>>>(P, (size_t)(K / 32) * 12288, 3u);
    const float a_h = 4.2e-4f; CK(hipMemcpy(alpha, &a_h, 4, hipMemcpyHostToDevice));
    int rows_h[16];
    CK(hipDeviceSynchronize());

    auto params = [&](int M, int Nn, bool fused, unsigned short* C) {
        GemmParams p{};
        p.A = (const char*)X; p.P = (const int32_t*)P; p.C = (char*)C; p.W = fused ? (const char*)W : nullptr; p.alpha = alpha;
        p.M = M; p.N = Nn; p.K = K; p.tile_m0 = 0; p.tile_n0 = 0; p.ksplit = 1;
        p.sAb = 0; p.sPb = 0; p.sCb = 0; p.sAm = K; p.sCm = Nn; p.ldw = K; p.sAlb = 0; p.gsz = Nn; p.round_mode = 0; p.accumulate = 0;
        p.group_m = fused ? 4 : 1;
        return p;
    };
    auto check = [&](const char* name, int M, int Nn, bool fused, bool vs_old) {
        CK(hipDeviceSynchronize());
        if (vs_old) {
            unsigned long long z = 0; CK(hipMemcpy(dcount, &z, 8, hipMemcpyHostToDevice));
            count_diff<<<1024, 256>>>((const uint32_t*)C0, (const uint32_t*)C1, (size_t)M * Nn / 2, dcount);
            CK(hipMemcpy(&z, dcount, 8, hipMemcpyDeviceToHost));
            printf("  %-34s bitwise vs 8-wave kernel: %llu differing dwords of %zu %s\n", name, z, (size_t)M * Nn / 2, z ? "  <-- MISMATCH" : "(identical)");
        }
        for (int i = 0; i < 16; ++i) rows_h[i] = (int)(((long long)i * 2654435761ll + 12345) % M);
        rows_h[0] = 0; rows_h[1] = M - 1; rows_h[2] = 255 % M; rows_h[3] = 256 % M;
        CK(hipMemcpy(drows, rows_h, sizeof(rows_h), hipMemcpyHostToDevice));
        float z = 0; CK(hipMemcpy(derr, &z, 4, hipMemcpyHostToDevice));
        ref_rows<<<dim3((Nn + 255) / 256, 16), 256>>>(X, P, fused ? W : nullptr, a_h, C1, M, Nn, K, 16, drows, derr);
        CK(hipMemcpy(&z, derr, 4, hipMemcpyDeviceToHost));
        printf("  %-34s max scaled error vs fp64 reference on 16 rows: %.3e %s\n", name, z, z < 8e-3f ? "(ok)" : "  <-- BAD");
    };

    if (what.rfind("soak", 0) == 0) {      // soak<variant>: run one kernel back to back for `reps` x 0.1 s (power / clock sampling from outside)
        const int M = Ms[0];
        const int v = atoi(what.c_str() + (what.rfind("soakn", 0) == 0 ? 5 : 4));
        GemmParams p0 = params(M, N, (v >= 10 && v < 20) || v == 31 || v == 33 || v == 35, C0);   // (fused variants: 10, 11, 12, 13)
        using OldD = GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 1>;
        using OldF = FxCfg<DT_BF16, 256, 128, 3, false, 1>;
        auto one = [&] {
            if (v == 0) launch_old<OldD>(delta_gemm_pf_kernel<OldD>, p0, 1, 0);
            else if (v == 1) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 0>>(p0, 1, 0);
            else if (v == 2) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1>>(p0, 1, 0);
            else if (v == 5) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 2>>(p0, 1, 0);     // energy A/B: dependent MFMA pairs (results wrong)
            else if (v == 6) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 16>>(p0, 1, 0);    // energy A/B: dependent chains of 4
            else if (v == 7) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 8>>(p0, 1, 0);     // ablation: no LDS-DMA in the loop
            else if (v == 8) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 32>>(p0, 1, 0);    // ablation: no X-fragment reads in the loop
            else if (v == 9) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 8 | 32>>(p0, 1, 0);  // ablation: neither
            else if (v == 20) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 128>>(p0, 1, 0);   // operand-value A/B: sign LUT {0, 2.0}
            else if (v == 21) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 256>>(p0, 1, 0);   // operand-value A/B: sign LUT {0, 1.0}
            else if (v == 22) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 512>>(p0, 1, 0);   // operand-value A/B: sign LUT all zero
            else if (v == 13) launch_w4<W4Cfg<DT_BF16, 256, 128, true, false, 1 | 128>>(p0, 1, 0);    // fused, sign LUT {0, 2.0}
            else if (v == 23) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 1024>>(p0, 1, 0);  // MFMA-order A/B: X-stationary (results wrong)
            else if (v == 30) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 2048>>(p0, 1, 0);  // trickled epilogue, delta-only
            else if (v == 31) launch_w4<W4Cfg<DT_BF16, 256, 128, true, false, 1 | 2048>>(p0, 1, 0);   // trickled epilogue, fused
            else if (v == 32) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 8192>>(p0, 1, 0);  // k loop unrolled by four, no trickle
            else if (v == 33) launch_w4<W4Cfg<DT_BF16, 256, 128, true, false, 1 | 8192>>(p0, 1, 0);   // same, fused
            else if (v == 34) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 8192 | 16384>>(p0, 1, 0);  // + per-wave store stagger (round 6 A/B)
            else if (v == 35) launch_w4<W4Cfg<DT_BF16, 256, 128, true, false, 1 | 8192 | 16384>>(p0, 1, 0);   // same, fused
            else if (v == 3) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 64>>(p0, 1, 0);    // split-form DMA
            else if (v == 12) launch_w4<W4Cfg<DT_BF16, 256, 128, true, false, 0>>(p0, 1, 0);         // fused, VALU sign expansion (A/B)
            else if (v == 4) launch_w4<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 4>>(p0, 1, 0);     // energy A/B: sign fragment in the first MFMA slot (results wrong)
            else if (v == 10) launch_old<OldF>(delta_gemm_fx_kernel<OldF>, p0, 1, 0);
            else launch_w4<W4Cfg<DT_BF16, 256, 128, true, false, 1>>(p0, 1, 0);      // 11: the shipped fused configuration
        };
        if (what.rfind("soakn", 0) == 0) {        // soakn<variant>: exactly `reps` launches (for rocprofv3 passes)
            for (int i = 0; i < reps; ++i) one();
            CK(hipDeviceSynchronize());
            printf("ran %d launches of variant %d at M=%d\n", reps, v, M);
            return 0;
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        long long n = 0; float total = 0;
        CK(hipEventRecord(e0, 0));
        while (total < reps * 100.f) {
            for (int i = 0; i < 50; ++i) one();
            n += 50;
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&total, e0, e1));
        }
        const double fl = ((v >= 10 && v < 20) || v == 31 || v == 33 || v == 35 ? 4.0 : 2.0) * M * N * K;
        printf("soak variant %d M=%d: %lld launches in %.1f ms -> %.2f us each, %.1f TF\n", v, M, n, total, total * 1e3 / n, fl * n / (total * 1e-3) / 1e12);
        return 0;
    }
#ifdef BD_TRACE
    for (int M : Ms) {
        printf("== trace M=%d\n", M);
        trace_one<W4Cfg<DT_BF16, 256, 256, false, false, 1>>("w4 LUT", params(M, N, false, C1), 2.0 * M * N * K);
        trace_one<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 2048>>("w4 LUT trickle", params(M, N, false, C1), 2.0 * M * N * K);
        trace_one<W4Cfg<DT_BF16, 256, 128, true, false, 1 | 2048>>("w4 fused trickle", params(M, N, true, C1), 4.0 * M * N * K);
        trace_one<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 2048 | 4096>>("w4 LUT trickle, NO stores", params(M, N, false, C1), 2.0 * M * N * K);
        trace_one<W4Cfg<DT_BF16, 256, 256, false, false, 1 | 8192>>("w4 LUT, k loop x4", params(M, N, false, C1), 2.0 * M * N * K);
        trace_one<W4Cfg<DT_BF16, 256, 128, true, false, 1 | 8192>>("w4 fused, k loop x4", params(M, N, true, C1), 4.0 * M * N * K);
        trace_one<W4Cfg<DT_BF16, 256, 128, true, false, 1>>("w4 fused (LUT, shipped)", params(M, N, true, C1), 4.0 * M * N * K);
    }
    return 0;
#endif
    for (int M : Ms) {
        const double fl = 2.0 * M * N * K;
        printf("== delta-only  M=%d N=%d K=%d\n", M, N, K);
        using OldD = GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 1>;
        using W4V = W4Cfg<DT_BF16, 256, 256, false, false, 0>;
        using W4L = W4Cfg<DT_BF16, 256, 256, false, false, 1>;
        GemmParams p0 = params(M, N, false, C0), p1 = params(M, N, false, C1);
        launch_old<OldD>(delta_gemm_pf_kernel<OldD>, p0, 1, 0);
        CK(hipMemset(C1, 0xff, (size_t)M * N * 2));
        launch_w4<W4V>(p1, 1, 0);
        check("w4 VALU expansion", M, N, false, true);
        CK(hipMemset(C1, 0xff, (size_t)M * N * 2));
        launch_w4<W4L>(p1, 1, 0);
        check("w4 LUT expansion", M, N, false, true);
        using W4T = W4Cfg<DT_BF16, 256, 256, false, false, 1 | 2048>;
        CK(hipMemset(C1, 0xff, (size_t)M * N * 2));
        launch_w4<W4T>(p1, 1, 0);
        check("w4 LUT, trickled epilogue", M, N, false, true);
        CK(hipMemset(C1, 0xff, (size_t)M * N * 2));
        launch_w4<W4T>(p1, 1, 0, 100);          // 100 workgroups: 3 / 6 / 11 rounds with a ragged last one
        check("w4 LUT, trickled, grid 100", M, N, false, true);
        time_it("w4 LUT", [&] { launch_w4<W4L>(p1, 1, 0); }, reps, fl);
        time_it("w4 LUT trickle", [&] { launch_w4<W4T>(p1, 1, 0); }, reps, fl);
        time_it("w4 LUT", [&] { launch_w4<W4L>(p1, 1, 0); }, reps, fl);
        time_it("w4 LUT trickle", [&] { launch_w4<W4T>(p1, 1, 0); }, reps, fl);
    }
    if (what == "all" || what == "fused") {
        const int shapes[][2] = {{2048, 4096}, {2048, 11008}, {2048, 12288}, {8192, 4096}};
        for (auto& sh : shapes) {
            const int M = sh[0], Nn = sh[1];
            if (M > Mmax) continue;
            const double fl = 4.0 * M * Nn * K;
            printf("== fused  M=%d N=%d K=%d\n", M, Nn, K);
            using OldF = FxCfg<DT_BF16, 256, 128, 3, false, 1>;
            using W4F = W4Cfg<DT_BF16, 256, 128, true, false, 1>;
            GemmParams p0 = params(M, Nn, true, C0), p1 = params(M, Nn, true, C1);
            launch_old<OldF>(delta_gemm_fx_kernel<OldF>, p0, 1, 0);
            CK(hipMemset(C1, 0xff, (size_t)M * Nn * 2));
            launch_w4<W4F>(p1, 1, 0);
            check("w4 fused", M, Nn, true, true);
            using W4FT = W4Cfg<DT_BF16, 256, 128, true, false, 1 | 2048>;
            CK(hipMemset(C1, 0xff, (size_t)M * Nn * 2));
            launch_w4<W4FT>(p1, 1, 0);
            check("w4 fused, trickled epilogue", M, Nn, true, true);
            CK(hipMemset(C1, 0xff, (size_t)M * Nn * 2));
            launch_w4<W4FT>(p1, 1, 0, 100);
            check("w4 fused, trickled, grid 100", M, Nn, true, true);
            time_it("w4 fused", [&] { launch_w4<W4F>(p1, 1, 0); }, reps, fl);
            time_it("w4 fused trickle", [&] { launch_w4<W4FT>(p1, 1, 0); }, reps, fl);
            time_it("w4 fused", [&] { launch_w4<W4F>(p1, 1, 0); }, reps, fl);
            time_it("w4 fused trickle", [&] { launch_w4<W4FT>(p1, 1, 0); }, reps, fl);
        }
    }
    return 0;
}
