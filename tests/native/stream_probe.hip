// Pure HBM read-stream ceilings on MI355X (development probe; not part of the library).
//
//   reg  : every wave streams its own contiguous region with buffer_load_dwordx4 into registers, NS loads in flight
//          (the structure of gemv_stream_kernel, variant 600)
//   dma  : NL loader waves per CU stream 16-KiB fills into an LDS ring with global_load_lds_dwordx4 (optionally nt), hand-counted
//          vmcnt; optionally 4 consumer waves read every landed fill back from LDS behind the ready / free flag protocol of
//          gemv_ring_kernel (variant 700) and XOR-fold it, so the host can verify that no fill was read early, late or twice.
//
// Every launch reads `bytes` cold bytes: the source is a 1.5-GiB buffer and consecutive launches read consecutive windows of it
// (the 256-MiB Infinity Cache never holds the window a launch is about to read).
//
//   stream_probe [MB per launch, default 235 46]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ void dma16(uint32_t voff, const void* sbase, uint32_t lds_addr) {
    uint32_t keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- reg: 4 waves per CU, each its own contiguous region
template <int AUX, int NS>
__global__ void __launch_bounds__(256) reg_stream(const char* src, uint32_t per_wave, uint32_t* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* p = src + ((size_t)blockIdx.x * 4 + wave) * per_wave;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), (short)0, (int)per_wave, 0x00020000);
    const int n = per_wave / 1024;
    u32x4 buf[NS], acc = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < NS; ++u) buf[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (u * 64 + lane) * 16, 0, AUX));
    for (int i = 0; i < n; i += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            acc ^= buf[u];
            buf[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ((i + NS + u) * 64 + lane) * 16, 0, AUX));   // past the end: zeros
        }
    }
    sink[((size_t)blockIdx.x * 256 + threadIdx.x) * 4 + 0] = acc[0];
    sink[((size_t)blockIdx.x * 256 + threadIdx.x) * 4 + 1] = acc[1];
    sink[((size_t)blockIdx.x * 256 + threadIdx.x) * 4 + 2] = acc[2];
    sink[((size_t)blockIdx.x * 256 + threadIdx.x) * 4 + 3] = acc[3];
}

// ---- dma: NL loader waves + (CONS ? 4 : 0) consumer waves per CU; ring of NSLOT 16-KiB slots
constexpr int FILL = 16384;
constexpr int SPIN_LIMIT = 1 << 22;
template <int NT, int NL, int DEPTH, int CONS, int NSLOT>
__global__ void __launch_bounds__(64 * (NL + (CONS ? 4 : 0))) dma_stream(const char* src, uint32_t per_cu, uint32_t* sink, uint32_t* err) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    // explicit LDS address space: a volatile access through a generic pointer compiles to flat_load (vmcnt AND lgkmcnt: it would drain the DMA queue)
    typedef volatile __attribute__((address_space(3))) uint32_t* ldsflag_t;
    const ldsflag_t ready = (ldsflag_t)(__attribute__((address_space(3))) char*)(lds + NSLOT * FILL);     // [NSLOT]
    const ldsflag_t freef = ready + NSLOT;                                                                 // [NSLOT]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (size_t)blockIdx.x * per_cu;
    const int nfill = per_cu / FILL;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    if (threadIdx.x < 2 * NSLOT) ready[threadIdx.x] = 0u;
    __syncthreads();
    if (wave < NL) {
        // loader `wave`: fills f = wave, wave + NL, ...; slot of fill f = f % NSLOT (NSLOT % NL == 0)
        int issued = 0;
        for (int f = wave; f < nfill; f += NL, ++issued) {
            const int slot = f % NSLOT;
            if (CONS && f >= NSLOT) {
                const uint32_t want = (uint32_t)(f / NSLOT);                     // the slot's previous use has been consumed
                int spin = 0;
                while (freef[slot] != want) { __builtin_amdgcn_s_sleep(1); if (++spin > SPIN_LIMIT) { if (lane == 0) err[0] = 1; return; } }
            }
            const uint32_t voff = (uint32_t)f * FILL + lane * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) dma16<NT>(voff + i * 1024, base, lds0 + slot * FILL + i * 1024);
            if (issued >= DEPTH - 1) {
                wait_vmcnt<16 * (DEPTH - 1)>();
                const int fd = f - (DEPTH - 1) * NL;
                if (CONS) ready[fd % NSLOT] = (uint32_t)(fd / NSLOT + 1);
            }
        }
        wait_vmcnt<0>();
        if (CONS) {
            int lastf = wave + (issued - 1) * NL;
            for (int k = DEPTH - 2; k >= 0; --k) {
                const int fd = lastf - k * NL;
                if (fd >= 0 && fd >= wave) ready[fd % NSLOT] = (uint32_t)(fd / NSLOT + 1);
            }
        }
    } else if (CONS) {
        const int c = wave - NL;
        u32x4 acc = {0u, 0u, 0u, 0u};
        for (int f = c; f < nfill; f += 4) {
            const int slot = f % NSLOT;
            const uint32_t want = (uint32_t)(f / NSLOT + 1);
            int spin = 0;
            while (ready[slot] != want) { __builtin_amdgcn_s_sleep(1); if (++spin > SPIN_LIMIT) { if (lane == 0) err[0] = 2; return; } }
            asm volatile("" ::: "memory");
            const char* s = lds + slot * FILL + lane * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc ^= *(const u32x4*)(s + i * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            freef[slot] = want;
        }
        uint32_t* o = sink + ((size_t)blockIdx.x * 256 + c * 64 + lane) * 4;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
    }
}

int main(int argc, char** argv) {
    std::vector<size_t> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back((size_t)atoi(argv[i]) * 1000000);
    if (sizes.empty()) { sizes.push_back(235000000); sizes.push_back(46000000); }
    const size_t total = 1536ull << 20;
    char* src; uint32_t *sink, *err;
    HIPCHECK(hipMalloc(&src, total + (64 << 20)));
    HIPCHECK(hipMalloc(&sink, 256 * 256 * 16));
    HIPCHECK(hipMalloc(&err, 64));
    HIPCHECK(hipMemset(err, 0, 64));
    {   // pseudo-random fill (a constant buffer would let DVFS / compression flatter the numbers)
        std::vector<uint32_t> h((total + (64 << 20)) / 4);
        uint32_t s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
        HIPCHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    for (size_t want : sizes) {
        const uint32_t per_cu = (uint32_t)(want / 256 / (4 * FILL)) * (4 * FILL);      // whole fills, a multiple of 4 per CU
        const size_t bytes = (size_t)per_cu * 256;
        const int nwin = (int)(total / bytes);
        auto run = [&](const char* name, auto launch, bool check) {
            for (int i = 0; i < 3; ++i) launch(src + (size_t)(i % nwin) * bytes);
            HIPCHECK(hipDeviceSynchronize());
            const int iters = 30;
            HIPCHECK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) launch(src + (size_t)((i + 3) % nwin) * bytes);
            HIPCHECK(hipEventRecord(e1));
            HIPCHECK(hipEventSynchronize(e1));
            float ms = 0; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
            uint32_t herr = 0; HIPCHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            int bad = 0;
            if (check) {     // one more launch on window 0, XOR-fold on the host
                HIPCHECK(hipMemset(sink, 0, 256 * 256 * 16));
                launch(src);
                HIPCHECK(hipDeviceSynchronize());
                std::vector<uint32_t> hs(256 * 256 * 4), hsrc(bytes / 4);
                HIPCHECK(hipMemcpy(hs.data(), sink, hs.size() * 4, hipMemcpyDeviceToHost));
                HIPCHECK(hipMemcpy(hsrc.data(), src, bytes, hipMemcpyDeviceToHost));
                uint32_t g[4] = {0, 0, 0, 0}, w[4] = {0, 0, 0, 0};
                for (size_t i = 0; i < hs.size(); ++i) g[i & 3] ^= hs[i];
                for (size_t i = 0; i < hsrc.size(); ++i) w[i & 3] ^= hsrc[i];
                bad = memcmp(g, w, 16) != 0;
            }
            const double us = ms * 1e3 / iters;
            printf("{\"probe\":\"%s\",\"MB\":%.1f,\"us\":%.2f,\"GBps\":%.0f,\"err\":%u,\"checksum_bad\":%d}\n", name, bytes * 1e-6, us, bytes / us * 1e-3, herr, bad);
            fflush(stdout);
            if (herr) { HIPCHECK(hipMemset(err, 0, 64)); }
        };
#define REG(AUX, NS) run("reg aux" #AUX " ns" #NS, [&](const char* p) { hipLaunchKernelGGL((reg_stream<AUX, NS>), dim3(256), dim3(256), 0, 0, p, per_cu / 4, sink); }, true)
        REG(0, 4); REG(0, 8); REG(2, 4); REG(2, 8);
#define DMA(NT, NL, DEPTH, CONS, NSLOT)                                                                                              \
        do {                                                                                                                          \
            auto k = dma_stream<NT, NL, DEPTH, CONS, NSLOT>;                                                                           \
            HIPCHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * FILL + 256));            \
            run("dma nt" #NT " loaders" #NL " depth" #DEPTH " cons" #CONS " slots" #NSLOT, [&](const char* p) {                         \
                hipLaunchKernelGGL(k, dim3(256), dim3(64 * (NL + (CONS ? 4 : 0))), NSLOT * FILL + 256, 0, p, per_cu, sink, err); }, CONS); \
        } while (0)
        DMA(0, 1, 2, 0, 8); DMA(0, 1, 3, 0, 8); DMA(1, 1, 2, 0, 8); DMA(1, 1, 3, 0, 8);
        DMA(0, 2, 2, 0, 8); DMA(0, 2, 3, 0, 8); DMA(1, 2, 2, 0, 8); DMA(1, 2, 3, 0, 8);
        DMA(0, 4, 2, 0, 8); DMA(1, 4, 2, 0, 8);
        DMA(0, 1, 3, 1, 8); DMA(1, 1, 3, 1, 8); DMA(0, 2, 3, 1, 8); DMA(1, 2, 3, 1, 8);
        DMA(1, 1, 3, 1, 4); DMA(1, 2, 2, 1, 4);
    }
    return 0;
}
