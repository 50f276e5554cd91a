// Probe: lane / element mapping of ds_read_b64_tr_b16 on gfx950 (no ISA manual in this image).  LDS[e] = e (16-bit); lane l passes the
// byte address 8*l (elements 4l .. 4l+3); prints which elements every lane got.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(uint16_t* out, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + threadIdx.x * stride_bytes;
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
    out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int stride : {8, 32}) {
        probe<<<1, 64>>>(d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d bytes per lane: lane -> the 4 elements it received, as (source lane, element)\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) { const int e = h[l * 4 + j]; const int sl = (e * 2) / stride, se = (e * 2 - sl * stride) / 2; printf(" (%2d,%d)", sl, se); }
            printf("\n");
        }
    }
    return 0;
}
