// Timeline harness of ONE streaming decode launch (bitdelta_amd/csrc/bd_gemv_stream.h) on cold weights: s_memtime stamps of every block's wave 0
// (BD_STREAM_TRACE) + event-timed launches over rotating weight sets.  Development tool; not part of the product or of pytest.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBD_STREAM_TRACE -I bitdelta_amd/csrc -o /tmp/stream_tl tests/native/stream_tl.hip
//   /tmp/stream_tl [N=4096] [K=4096] [T=6] [form: 0 = resident rows (o), 1 = fine grid hand-off consumer (q|k|v), 2 = plain per-stage loads (down)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "bd_gemv_stream.h"
using namespace bd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#ifndef TL_NS2
#define TL_NS2 4          // prefetch depth of the per-stage-load form (form 2: down)
#endif
#ifndef TL_NS
#define TL_NS 2
#endif
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 4096, T = argc > 3 ? atoi(argv[3]) : 6, form = argc > 4 ? atoi(argv[4]) : 0;
    const int SETS = 8, nit = (K + 127) / 128, tiles = (N + 15) / 16;
    const size_t wb = (size_t)tiles * nit * 4096, pb = (size_t)tiles * nit * 4 * 16 * 6 * 4;
    char *W, *P; unsigned short *X, *C, *XW, *NWT; float *alpha, *ssq;
    CK(hipMalloc(&W, wb * SETS)); CK(hipMalloc(&P, pb * SETS)); CK(hipMalloc(&X, (size_t)T * K * 2)); CK(hipMalloc(&C, (size_t)T * N * 2));
    CK(hipMalloc(&XW, (size_t)T * N * 2)); CK(hipMalloc(&NWT, (size_t)T * N * 2)); CK(hipMalloc(&alpha, T * 4)); CK(hipMalloc(&ssq, (size_t)(std::max(N, K) / 16) * 64));
    CK(hipMemset(W, 0x11, wb * SETS)); CK(hipMemset(P, 0x5a, pb * SETS)); CK(hipMemset(X, 0x3c, (size_t)T * K * 2)); CK(hipMemset(C, 0, (size_t)T * N * 2));
    CK(hipMemset(NWT, 0x3c, (size_t)T * N * 2)); CK(hipMemset(alpha, 0, T * 4)); CK(hipMemset(ssq, 0x40, (size_t)(std::max(N, K) / 16) * 64));
    int cus = 256;
    StreamParams sp{};
    GemvParams& g = sp.g;
    g.X = X; g.alpha = alpha; g.C = C; g.B = T; g.M = 1; g.N = N; g.K = K; g.R = T; g.sXb = K; g.sPb = 1; g.sCb = N; g.sXm = K; g.sCm = N; g.ldw = 0;
    g.sAlb = 1; g.gsz = N; g.KS = 1; g.kslice = K; g.round_mode = 0; g.out_f32 = 0;
    sp.tp = 6; sp.pts = 16; sp.prs = N; sp.x_bytes = (uint32_t)(((size_t)(T - 1) * K + K) * 2); sp.w_bytes = (uint32_t)wb; sp.p_bytes = (uint32_t)pb;
    sp.xrow = (uint32_t)K * 2 + 16; sp.jsh = 0; sp.eps = 1e-5f; sp.ssq_scale = 1.f;
    int cpb = ((N + cus - 1) / cus + 3) & ~3;
    unsigned grid;
    size_t lds;
    void (*kern)(const StreamParams);
    if (form == 0) {            // o projection of the step: resident rows, residual epilogue, hand-off producer
        kern = gemv_stream_kernel<DT_F16, 6, true, TL_NS, 4, 1, 2, 1, 2, 0, 1, 0>;
        g.accumulate = 1; sp.ssq_out = ssq; sp.xw_out = XW; sp.nw_next = NWT; sp.sNwNext = N; cpb = (cpb + 15) & ~15;
        sp.xs_off = STREAM_XS_OFF; lds = sp.xs_off + (size_t)T * sp.xrow; grid = (N + cpb - 1) / cpb;
    } else if (form == 1) {     // q|k|v: fine grid, hand-off consumer
        kern = gemv_stream_kernel<DT_F16, 6, true, TL_NS, 4, 1, 2, 1, 3, 0, 1, 1>;
        sp.ssq_in = ssq; cpb = 16; sp.xs_off = STREAM_FG_XS_OFF; lds = sp.xs_off + (size_t)T * sp.xrow; grid = N / 16;
    } else {                    // down: per-stage activation loads, residual epilogue, hand-off producer
        kern = gemv_stream_kernel<DT_F16, 6, true, TL_NS2, 4, 1, 2, 1, 0, 0, 1, 0>;
        g.accumulate = 1; sp.ssq_out = ssq; sp.xw_out = XW; sp.nw_next = NWT; sp.sNwNext = N; cpb = (cpb + 15) & ~15;
        lds = STREAM_LDS_BYTES; grid = (N + cpb - 1) / cpb;
    }
    sp.cpb = cpb;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto launch = [&](int set) { sp.g.W = (const unsigned short*)(W + wb * set); sp.g.P = (const uint32_t*)(P + pb * set); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, sp); };
    for (int i = 0; i < 16; ++i) launch(i % SETS);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch(i % SETS);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wb + (double)pb * T / 6.0;
    printf("form %d N=%d K=%d T=%d NS=%d grid=%u lds=%zu: %.2f us per launch back to back (%.2f TB/s algorithmic)\n", form, N, K, T, TL_NS, grid, lds, ms * 1e3 / reps,
           bytes / (ms * 1e-3 / reps) * 1e-12);
    {   // FNV-1a over the output rows: the build variants of one form must print the same word (same launches, same inputs)
        std::vector<unsigned short> hc((size_t)T * N);
        CK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
        unsigned long long h = 1469598103934665603ull;
        for (unsigned short v : hc) { h ^= v; h *= 1099511628211ull; }
        printf("  output checksum %016llx\n", h);
    }
#ifdef BD_STREAM_TRACE
    static unsigned long long tr[1024][8];
    CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_stream_trace), sizeof(tr)));
    // (the counters of different XCDs are not synchronised: every block is measured against ITS OWN entry stamp)
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx5 = 0;
    for (unsigned b = 0; b < grid && b < 1024; ++b) for (int i = 0; i < 8; ++i) s[i] += (double)(tr[b][i] - tr[b][0]);
    for (unsigned b = 0; b < grid && b < 1024; ++b) mx5 = std::max(mx5, (double)(tr[b][5] - tr[b][0]));
    const double nb = std::min(grid, 1024u);
    if (s[6] > 0) printf("  sub-stamps: [6] %.0f  [7] %.0f\n", s[6] / nb, s[7] / nb);
    printf("  shader cycles since the block's own entry, mean over %g blocks (last launch): prologue loads issued %.0f | first barrier %.0f | first stage "
           "consumed %.0f | loop done %.0f | exit %.0f (slowest block %.0f)\n", nb, s[1] / nb, s[2] / nb, s[3] / nb, s[4] / nb, s[5] / nb, mx5);
#endif
    return 0;
}
