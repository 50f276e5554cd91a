// s_memtime timeline of the loader / consumer decode kernel (variant 700), block 0: where a stage's time goes.
// Development probe: compiles bd_gemv_ring.h directly with -DBD_RING_TRACE (the shipped library carries no trace code).
//   ring_trace T N K [tune]         fp16, M = 1, tile-major W, random operands (timing only, no checker); tune = bd_set_ring_tuning flags
#define BD_RING_TRACE
#include "ab/bd_gemv_ring.h"

#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
using namespace bd;

template <int NM>
static void launch(const RingParams& rp, unsigned grid, unsigned lds, int nt) {
    if (nt) {
        auto k = gemv_ring_kernel<DT_F16, NM, 1>;
        HIPCHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_MAX));
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * (4 + rp.nl)), lds, 0, rp);
    } else {
        auto k = gemv_ring_kernel<DT_F16, NM, 0>;
        HIPCHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_MAX));
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * (4 + rp.nl)), lds, 0, rp);
    }
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: ring_trace T N K [tune]\n"); return 2; }
    const int T = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    const int tune = argc > 4 ? atoi(argv[4]) : 1;      // bd_set_ring_tuning flags: 1 nt, 2 x rides the ring, 4 one loader, 8 table signs, cap << 8
    const int tp = T <= 1 ? 1 : T <= 2 ? 2 : T <= 4 ? 4 : T <= 6 ? 6 : 8;
    const int nit = K / 128, NT16 = (N + 15) / 16;
    const size_t wbytes = (size_t)NT16 * nit * 4096, pbytes = (size_t)NT16 * nit * 64 * tp * 4;
    const int nset = std::max(2, (int)(600e6 / (double)(wbytes + pbytes)) + 1);
    std::vector<void*> Ws(nset), Ps(nset);
    std::vector<uint32_t> h(std::max(wbytes, pbytes) / 4);
    uint32_t s = 777u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s & 0x3fff3fffu) | 0x20002000u; }      // small finite fp16 values
    for (int i = 0; i < nset; ++i) {
        HIPCHECK(hipMalloc(&Ws[i], wbytes)); HIPCHECK(hipMemcpy(Ws[i], h.data(), wbytes, hipMemcpyHostToDevice));
        HIPCHECK(hipMalloc(&Ps[i], pbytes)); HIPCHECK(hipMemcpy(Ps[i], h.data(), pbytes, hipMemcpyHostToDevice));
    }
    void *dX, *dY, *dA;
    unsigned long long* dTr;
    HIPCHECK(hipMalloc(&dX, (size_t)T * K * 2)); HIPCHECK(hipMemcpy(dX, h.data(), (size_t)T * K * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&dY, (size_t)T * N * 2));
    std::vector<float> ha(T, 4e-4f);
    HIPCHECK(hipMalloc(&dA, T * 4)); HIPCHECK(hipMemcpy(dA, ha.data(), T * 4, hipMemcpyHostToDevice));
    const size_t trb = 6 * 1024 * 4 * 8;
    HIPCHECK(hipMalloc(&dTr, trb));

    RingParams rp{};
    GemvParams& gp = rp.g;
    gp.X = (const unsigned short*)dX; gp.alpha = (const float*)dA; gp.C = dY;
    gp.B = T; gp.M = 1; gp.N = N; gp.K = K; gp.R = T;
    gp.sXb = K; gp.sPb = T == 1 ? 0 : 1; gp.sCb = N; gp.sXm = K; gp.sCm = N; gp.ldw = K; gp.sAlb = 1; gp.gsz = N;
    gp.KS = 1; gp.kslice = K; gp.round_mode = 0; gp.accumulate = 0; gp.out_f32 = 0;
    int cus = 256;
    HIPCHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    int cpb = (N + cus - 1) / cus;
    cpb = std::max(4, (cpb + 3) & ~3);
    rp.cpb = cpb; rp.epi = 0;
    if (!ring_plan_geometry(rp, T, K, tp, true, K, false, tune)) { fprintf(stderr, "does not fit\n"); return 2; }
    const int nt = rp.nt, xmode = rp.xmode, nl = rp.nl, ns = (int)rp.nslot;
    const unsigned lds = ring_lds_bytes(rp), grid = (unsigned)((N + cpb - 1) / cpb);
    auto go = [&](int i, unsigned long long* tr) {
        rp.g.W = (const unsigned short*)Ws[i % nset]; rp.g.P = (const uint32_t*)Ps[i % nset]; rp.trace = tr;
        switch (tp) { case 1: launch<1>(rp, grid, lds, nt); break; case 2: launch<2>(rp, grid, lds, nt); break; case 4: launch<4>(rp, grid, lds, nt); break;
                      case 6: launch<6>(rp, grid, lds, nt); break; default: launch<8>(rp, grid, lds, nt); }
    };
    for (int i = 0; i < nset + 2; ++i) go(i, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipEventRecord(e0));
    const int iters = 50;
    for (int i = 0; i < iters; ++i) go(i, nullptr);
    HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
    float ms; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wbytes + pbytes;
    printf("# T=%d N=%d K=%d tune=%d nt=%d xmode=%d loaders=%d sx=%d slots=%d slot_bytes=%u lds=%u grid=%u cpb=%d: %.2f us per launch (cold), %.0f GB/s (W + packed signs)\n", T, N, K, tune, nt,
           xmode, nl, rp.sx, ns, rp.slot_bytes, lds, grid, cpb, ms * 1e3 / iters, bytes / (ms * 1e3 / iters) * 1e-3);
    HIPCHECK(hipMemset(dTr, 0, trb));
    go(1, dTr);
    HIPCHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> tr(trb / 8);
    HIPCHECK(hipMemcpy(tr.data(), dTr, trb, hipMemcpyDeviceToHost));
    auto at = [&](int w, int i, int k) { return tr[((size_t)w * 1024 + i) * 4 + k]; };
    const int ntile = (std::min(N, cpb) + 15) / 16, total = ntile * nit;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < 6; ++w) for (int i = 0; i < 1024; ++i) for (int k = 0; k < 4; ++k) { const auto v = at(w, i, k); if (v) { t0 = std::min(t0, v); t1 = std::max(t1, v); } }
    printf("# block 0: %d stages, first stamp -> last stamp %llu cycles\n", total, t1 - t0);
    printf("# loader: stage  free_seen  issued  published  (cycles from the first stamp; issue cost = issued - free_seen; landing = published - issued)\n");
    for (int lwv = 0; lwv < nl; ++lwv) {
        const int w = lwv ? 5 : 0, show = std::min((total - lwv + nl - 1) / nl, 1024);
        for (int j = 0; j < show; ++j)
            if (j < 24 || j >= show - 6)
                printf("L%d %4d  %8llu %8llu %8llu   issue %5lld  landing %6lld\n", lwv, j, at(w, j, 0) - t0, at(w, j, 1) - t0, at(w, j, 2) - t0,
                       (long long)(at(w, j, 1) - at(w, j, 0)), (long long)(at(w, j, 2) - at(w, j, 1)));
    }
    for (int q = 0; q < 4; ++q) {
        const int cq = q < nit ? (nit - q + 3) / 4 : 0, mine = std::min(cq * ntile, 1024);
        printf("# consumer %d: own stage  ready_seen  released  computed  tile_done   (wait = ready_seen - previous computed; comp = computed - released)\n", q);
        double wsum = 0, csum = 0, rsum = 0; int n = 0;
        for (int i = 0; i < mine; ++i) {
            const long long rs = at(1 + q, i, 0) ? (long long)(at(1 + q, i, 0) - t0) : -1, rl = (long long)(at(1 + q, i, 1) - t0), cp = (long long)(at(1 + q, i, 2) - t0);
            if (i < 12 || i >= mine - 3)
                printf("C%d %4d  %8lld %8lld %8lld %8lld\n", q, i, rs, rl, cp, at(1 + q, i, 3) ? (long long)(at(1 + q, i, 3) - t0) : -1);
            if (i > 0) { csum += (double)(cp - rl); rsum += (double)(cp - (long long)(at(1 + q, i - 1, 2) - t0)); ++n; }
        }
        if (n) printf("# consumer %d: avg (computed - released) %.0f cycles, avg period per own stage %.0f cycles over %d stages\n", q, csum / n, rsum / n, n);
    }
    return 0;
}
