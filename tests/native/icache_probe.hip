// Is the start of a kernel instruction-FETCH bound?  A straight-line block of N VALU instructions (v_add_u32 chains on 4 registers: ~8 bytes each)
// is executed twice in a loop by one wave per SIMD; s_memtime around each pass.  Pass 1 runs with whatever the instruction cache holds at kernel
// start, pass 2 with the lines pass 1 just fetched.  Launched back to back with itself, and alternating with a different large kernel.
//   hipcc --offload-arch=gfx950 -O3 tests/native/icache_probe.hip -o /tmp/icache_probe && /tmp/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define A4 "v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0\n\t"
#define A16 A4 A4 A4 A4
#define A64 A16 A16 A16 A16
#define A256 A64 A64 A64 A64
#define A1024 A256 A256 A256 A256
template <int WHICH>
__global__ void __launch_bounds__(256) probe(unsigned long long* out, unsigned* sink, int passes) {
    unsigned a = threadIdx.x, b = blockIdx.x, c = 3, d = 5;
    unsigned long long t[5];
    t[0] = __builtin_readcyclecounter();
    for (int p = 0; p < passes; ++p) {
        asm volatile(A1024 A1024 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (WHICH) asm volatile(A256 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));     // (a different code object size for the "other" kernel)
        t[p + 1] = __builtin_readcyclecounter();
    }
    if (threadIdx.x == 0) for (int i = 0; i <= passes; ++i) out[blockIdx.x * 8 + i] = t[i];
    if (a + b + c + d == 0x12345) sink[0] = a;
}
int main() {
    unsigned long long* out; unsigned* sink;
    CK(hipMalloc(&out, 1024 * 8 * 8)); CK(hipMalloc(&sink, 4));
    std::vector<unsigned long long> h(1024 * 8);
    auto report = [&](const char* what) {
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double p1 = 0, p2 = 0, p3 = 0;
        for (int b = 0; b < 256; ++b) { p1 += h[b * 8 + 1] - h[b * 8]; p2 += h[b * 8 + 2] - h[b * 8 + 1]; p3 += h[b * 8 + 3] - h[b * 8 + 2]; }
        printf("%-52s pass 1 %6.0f cycles | pass 2 %6.0f | pass 3 %6.0f   (2048 VALU instructions per pass, mean over 256 blocks)\n", what, p1 / 256, p2 / 256, p3 / 256);
    };
    for (int i = 0; i < 3; ++i) probe<0><<<256, 256>>>(out, sink, 3);
    CK(hipDeviceSynchronize());
    report("same kernel, back to back (3rd launch):");
    probe<1><<<256, 256>>>(out + 4096, sink, 3);
    probe<0><<<256, 256>>>(out, sink, 3);
    CK(hipDeviceSynchronize());
    report("after a different kernel:");
    // after something that sweeps the L2 (256 MB memset) and a different kernel
    void* big; CK(hipMalloc(&big, 512u << 20));
    CK(hipMemsetAsync(big, 1, 512u << 20, 0));
    probe<1><<<256, 256>>>(out + 4096, sink, 3);
    probe<0><<<256, 256>>>(out, sink, 3);
    CK(hipDeviceSynchronize());
    report("after a 512-MB memset + a different kernel:");
    CK(hipMemsetAsync(big, 2, 512u << 20, 0));
    probe<0><<<256, 256>>>(out, sink, 3);
    CK(hipDeviceSynchronize());
    report("after a 512-MB memset, same kernel as before it:");
    return 0;
}
