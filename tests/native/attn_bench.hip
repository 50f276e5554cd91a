// Stand-alone harness of the prefill attention kernel (bitdelta_amd/csrc/bd_attn_prefill.h): sampled-row check against a double-precision
// CPU softmax, event-timed launches.
//   hipcc --offload-arch=gfx950 -O3 -I bitdelta_amd/csrc tests/native/attn_bench.hip -o tests/native/attn_bench
//   tests/native/attn_bench [S=2048] [H=32] [KVH=32] [B=1] [causal=1] [pad=0] [iters=50]
#ifdef KS2
#include "ab/bd_attn_prefill_ks2.h"      // the key-split A/B kernel (round 6: measured equal or slower)
#else
#include "bd_attn_prefill.h"
#endif
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace bd;
#ifdef KS2
#define KERN prefill_attn_ks2_kernel<DT_BF16>
#define NTHR 512
#else
#define KERN prefill_attn_kernel<DT_BF16>
#define NTHR 256
#endif
static float bf2f(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 2048, H = argc > 2 ? atoi(argv[2]) : 32, KVH = argc > 3 ? atoi(argv[3]) : 32;
    const int B = argc > 4 ? atoi(argv[4]) : 1, causal = argc > 5 ? atoi(argv[5]) : 1, pad = argc > 6 ? atoi(argv[6]) : 0;
    const int iters = argc > 7 ? atoi(argv[7]) : 50;
    const int mode = argc > 8 ? atoi(argv[8]) : 0;      // 1: V = 1 (O must be 1), 2: Q = 0 (O = mean of the valid V rows), 3: V[k][d] = k (O = softmax-weighted key index)
    const int HD = 128;
    // one fused [B, S, (H + 2 KVH) * 128] buffer like the q|k|v projection's output
    const long long row = (long long)(H + 2 * KVH) * HD;
    std::vector<unsigned short> qkv((size_t)B * S * row);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& x : qkv) x = f2bf(nd(rng));
    if (mode)
        for (int b = 0; b < B; ++b) for (int k = 0; k < S; ++k) {
            unsigned short* r = &qkv[((size_t)b * S + k) * row];
            if (mode == 1) for (int i = 0; i < KVH * HD; ++i) r[(H + KVH) * HD + i] = f2bf(1.f);
            if (mode == 2) for (int i = 0; i < H * HD; ++i) r[i] = 0;
            if (mode == 3) for (int i = 0; i < KVH * HD; ++i) r[(H + KVH) * HD + i] = f2bf((float)(k % 64));
            if (mode == 4) for (int i = 0; i < KVH * HD; ++i) r[(H + KVH) * HD + i] = f2bf((float)(i % HD));
        }
    std::vector<int> kvs(B);
    for (int b = 0; b < B; ++b) kvs[b] = pad ? (b * 37 + pad) % (S / 2) : 0;
    unsigned short *dq, *dout; int* dks;
    CK(hipMalloc(&dq, qkv.size() * 2)); CK(hipMalloc(&dout, (size_t)B * S * H * HD * 2)); CK(hipMalloc(&dks, B * 4));
    CK(hipMemcpy(dq, qkv.data(), qkv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dks, kvs.data(), B * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, (size_t)B * S * H * HD * 2));
    PrefillAttnParams p{};
    p.q = dq; p.k = dq + (long long)H * HD; p.v = dq + (long long)(H + KVH) * HD; p.o = dout;
    p.sqb = p.skb = p.svb = (long long)S * row; p.sqs = p.sks = p.svs = row;
    p.sob = (long long)S * H * HD; p.sos = (long long)H * HD;
    p.kv_start = pad ? dks : nullptr;
    p.B = B; p.S = S; p.H = H; p.KVH = KVH; p.nqb = (S + 127) / 128;
    const float scale = 1.f / sqrtf((float)HD);
    p.c = scale * 1.4426950408889634f; p.causal = causal;
    CK(hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, PREFILL_ATTN_LDS));
    const dim3 grid(p.nqb * H * B);
    KERN<<<grid, NTHR, PREFILL_ATTN_LDS>>>(p);
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> out((size_t)B * S * H * HD);
    CK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
    // sampled rows
    double worst = 0, worst_rel = 0; int bad = 0, nrows = 0;
    std::mt19937 pick(7);
    for (int it = 0; it < 400; ++it) {
        const int b = pick() % B, h = pick() % H;
        int q = it < 8 ? (it < 4 ? it : S - 1 - (it - 4)) : pick() % S;
        if (it >= 8 && it < 40) q = (q / 64) * 64 + ((it & 1) ? 63 : 0);        // tile edges
        const int kvh = h / (H / KVH);
        const unsigned short* qr = &qkv[((size_t)b * S + q) * row + (size_t)h * HD];
        const int k_lo = kvs[b], k_hi = causal ? q : S - 1;
        std::vector<double> sc(S, 0.0), o(HD, 0.0);
        double mx = -1e300;
        for (int k = k_lo; k <= k_hi; ++k) {
            const unsigned short* kr = &qkv[((size_t)b * S + k) * row + (size_t)(H + kvh) * HD];
            double s = 0;
            for (int d = 0; d < HD; ++d) s += (double)bf2f(qr[d]) * bf2f(kr[d]);
            sc[k] = s * scale; mx = std::max(mx, sc[k]);
        }
        double l = 0;
        for (int k = k_lo; k <= k_hi; ++k) { sc[k] = exp(sc[k] - mx); l += sc[k]; }
        for (int k = k_lo; k <= k_hi; ++k) {
            const unsigned short* vr = &qkv[((size_t)b * S + k) * row + (size_t)(H + KVH + kvh) * HD];
            for (int d = 0; d < HD; ++d) o[d] += sc[k] * bf2f(vr[d]);
        }
        double rn = 0, en = 0;
        for (int d = 0; d < HD; ++d) {
            const double ref = k_lo <= k_hi ? o[d] / l : 0.0, got = bf2f(out[((size_t)b * S + q) * H * HD + (size_t)h * HD + d]);
            const double e = fabs(ref - got);
            worst = std::max(worst, e); rn += ref * ref; en += e * e;
            if (!(e <= 2e-2 + 2e-2 * fabs(ref))) { if (bad < 5) printf("  mismatch b%d h%d q%d d%d: ref %g got %g\n", b, h, q, d, ref, got); ++bad; }
        }
        if (rn > 0) worst_rel = std::max(worst_rel, sqrt(en / rn));
        ++nrows;
    }
    printf("S=%d H=%d KVH=%d B=%d causal=%d pad=%d: %d sampled rows, max abs err %.3e, worst row rel-L2 %.3e, mismatches %d -> %s\n",
           S, H, KVH, B, causal, pad, nrows, worst, worst_rel, bad, bad ? "FAIL" : "ok");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) KERN<<<grid, NTHR, PREFILL_ATTN_LDS>>>(p);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) KERN<<<grid, NTHR, PREFILL_ATTN_LDS>>>(p);
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    const double flops = 4.0 * S * (double)S * HD * H * B * (causal ? 0.5 : 1.0);
    printf("  %.1f us per launch, %.0f TF (%s flops)\n", us, flops / us * 1e-6, causal ? "causal-half" : "full");
    return bad != 0;
}
