// Phase-timeline probe of the ping-pong kernel: s_memtime stamps (build with -DBD_TRACE).  Development tool.
#define BD_TRACE 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "ab/bd_gemm_pp.h"
#include "../../bitdelta_amd/csrc/bd_gemm_pf.h"
using namespace bd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
template <class Cfg> void run(const char* name) {
    const int M = 4096, N = 4096, K = 4096;
    void *dA, *dP, *dC;
    CK(hipMalloc(&dA, (size_t)M * K * 2)); CK(hipMalloc(&dP, (size_t)K / 32 * N * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    std::vector<unsigned short> a((size_t)M * K); std::vector<unsigned> pw((size_t)K / 32 * N);
    unsigned long long st = 88172645463325252ull;
    auto rng = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (auto& v : a) { float f = ((rng() >> 40) / 16777216.0f - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    for (auto& v : pw) v = (unsigned)rng();
    CK(hipMemcpy(dA, a.data(), a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dP, pw.data(), pw.size() * 4, hipMemcpyHostToDevice));
    GemmParams p{};
    p.A = (const char*)dA; p.P = (const int32_t*)dP; p.C = (char*)dC; p.M = M; p.N = N; p.K = K;
    p.tiles_m = M / Cfg::BM; p.tiles_n = N / Cfg::BN; p.sAb = (long long)M * K; p.sCb = (long long)M * N; p.sAm = K; p.sCm = N; p.gsz = N; p.group_m = 1; p.ksplit = 1;
    auto kern = delta_gemm_pp_kernel<Cfg>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, p);
    CK(hipDeviceSynchronize());
    unsigned long long t[2][8][8];
    CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(bd_trace_buf), sizeof(t)));
    printf("== %s   (ticks; L0 | sync(g1) | M0 | sync(g0) | L1 | sync(g1) | M1 | ->next L0 incl. sync(g0))\n", name);
    for (int g = 0; g < 2; ++g)
        for (int k = 2; k < 4; ++k) {
            auto* s = t[g][k];
            printf("grp%d kt%2d: L0 %5lld  b %5lld  M0 %5lld  b %5lld  L1 %5lld  b %5lld  M1 %5lld  b+ %5lld   tile %6lld\n", g, 16 + k,
                   (long long)(s[1] - s[0]), (long long)(s[2] - s[1]), (long long)(s[3] - s[2]), (long long)(s[4] - s[3]),
                   (long long)(s[5] - s[4]), (long long)(s[6] - s[5]), (long long)(s[7] - s[6]), (long long)(t[g][k + 1][0] - s[7]),
                   (long long)(t[g][k + 1][0] - s[0]));
        }
    hipFree(dA); hipFree(dP); hipFree(dC);
}
template <class Cfg> void run_pf(const char* name) {
    const int M = 4096, N = 4096, K = 4096;
    void *dA, *dP, *dC;
    CK(hipMalloc(&dA, (size_t)M * K * 2)); CK(hipMalloc(&dP, (size_t)K / 32 * N * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    std::vector<unsigned short> a((size_t)M * K); std::vector<unsigned> pw((size_t)K / 32 * N);
    unsigned long long st = 88172645463325252ull;
    auto rng = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (auto& v : a) { float f = ((rng() >> 40) / 16777216.0f - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    for (auto& v : pw) v = (unsigned)rng();
    CK(hipMemcpy(dA, a.data(), a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dP, pw.data(), pw.size() * 4, hipMemcpyHostToDevice));
    GemmParams p{};
    p.A = (const char*)dA; p.P = (const int32_t*)dP; p.C = (char*)dC; p.M = M; p.N = N; p.K = K;
    p.tiles_m = M / Cfg::BM; p.tiles_n = N / Cfg::BN; p.sAb = (long long)M * K; p.sCb = (long long)M * N; p.sAm = K; p.sCm = N; p.gsz = N; p.group_m = 1; p.ksplit = 1;
    auto kern = delta_gemm_pf_kernel<Cfg>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, p);
    CK(hipDeviceSynchronize());
    unsigned long long t[2][8][4];
    CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(bd_trace_pf), sizeof(t)));
    printf("== %s   (ticks; L = loads+expansion | wait+barrier | M = 32 MFMAs + DMA | barrier -> next L)\n", name);
    for (int g = 0; g < 2; ++g)
        for (int k = 2; k < 4; ++k) {
            auto* s = t[g][k];
            printf("grp%d kt%2d: L %5lld  b %5lld  M %5lld  b+ %5lld   tile %6lld\n", g, 16 + k, (long long)(s[1] - s[0]), (long long)(s[2] - s[1]),
                   (long long)(s[3] - s[2]), (long long)(t[g][k + 1][0] - s[3]), (long long)(t[g][k + 1][0] - s[0]));
        }
    hipFree(dA); hipFree(dP); hipFree(dC);
}
int main() {
    run_pf<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 0>>("full-tile ping-pong 256x256");
    run_pf<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 1>>("full-tile ping-pong 256x256, LUT expansion");
    return 0;
    run<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2>>("strict, noprio, 1 MFMA : 2 VALU");
    run<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + 64>>("strict, noprio, 2 MFMA : 4 VALU");
    run<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + 512>>("strict, noprio, 4 MFMA : 8 VALU");
    run<GemmCfg<DT_BF16, 256, 256, 2, 4, 4, false, false, 2 + 1>>("strict, noprio, compiler-scheduled");
    return 0;
}
