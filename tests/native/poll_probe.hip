// Which loads can poll a word that another XCD's agent-scope atomic updates, and how fast?  (development probe for the decode-plan
// kernel's device-wide barrier, bd_plan.h)
//   block 1 (XCD 1) waits `delay` us, then atomically adds 1 to the word; block 0 (XCD 0) polls it with
//     mode 0: vector load, agent scope (global_load_dword sc1)          -- the form gemv_ticket_reduce / decode_attn already rely on
//     mode 1: scalar load (s_load_dword glc) after s_dcache_inv          -- would keep vmcnt free for the weight stream
//     mode 2: scalar load without glc after s_dcache_inv
//   on three allocations: hipMalloc, hipExtMallocWithFlags(uncached), hipExtMallocWithFlags(fine-grained).
// Prints: seen (1 = the update was observed within 20 ms), latency from the add to the observation in 100 MHz ticks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ unsigned sload(const unsigned* p, int glc) {
    unsigned v;
    if (glc) asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    else asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

__global__ void probe(unsigned* word, long long* out, int mode, int delay_ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 1) {
        while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(4);
        out[2] = (long long)__builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (blockIdx.x != 0) return;
    // touch the word first so that a stale copy can sit in this XCD's L2 / scalar cache
    unsigned first = mode == 0 ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : sload(word, mode == 1);
    int seen = 0;
    long long polls = 0;
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < 2000000) {        // 20 ms
        const unsigned v = mode == 0 ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : sload(word, mode == 1);
        ++polls;
        if (v != first) { seen = 1; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    out[0] = seen;
    out[1] = (long long)__builtin_amdgcn_s_memrealtime();
    out[3] = polls;
}

int main() {
    const char* an[3] = {"hipMalloc", "uncached", "finegrained"};
    for (int a = 0; a < 3; ++a) {
        unsigned* word = nullptr;
        hipError_t e = a == 0 ? hipMalloc(&word, 256) : hipExtMallocWithFlags((void**)&word, 256, a == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", an[a], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        long long* out;
        HIPCHECK(hipMalloc(&out, 64));
        for (int mode = 0; mode < 3; ++mode)
            for (int rep = 0; rep < 3; ++rep) {
                HIPCHECK(hipMemset(word, 0, 256));
                HIPCHECK(hipMemset(out, 0, 64));
                hipLaunchKernelGGL(probe, dim3(16), dim3(64), 0, 0, word, out, mode, 20000);      // add after 200 us
                HIPCHECK(hipDeviceSynchronize());
                long long h[4];
                HIPCHECK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
                printf("%-11s mode %d (%s): seen %lld, add -> observed %lld ticks (10 ns), polls %lld\n", an[a], mode,
                       mode == 0 ? "vector sc1" : mode == 1 ? "scalar glc" : "scalar", h[0], h[0] ? h[1] - h[2] : -1, h[3]);
            }
        HIPCHECK(hipFree(out));
        HIPCHECK(hipFree(word));
    }
    return 0;
}
