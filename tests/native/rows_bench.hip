// Development harness for delta_rows_kernel (bd_gemv_rows.h): event-timed launches of the shipped instantiations and of its ablations
// (no LUT reads / no MFMAs / no re-issued loads) on the reference's published binary_bmm shapes, masks rotated through > 400 MB.
// Not part of the product or of pytest.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o rows_bench rows_bench.hip ; ./rows_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../bitdelta_amd/csrc/bd_gemv_rows.h"
using namespace bd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill_u32(uint32_t* x, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t a = (uint32_t)i * 2246822519u + seed; a ^= a >> 15; a *= 0x2c1b3c6du; a ^= a >> 12; a *= 0x297a2d39u; a ^= a >> 15;
        x[i] = a;
    }
}
__global__ void fill_f16(unsigned short* x, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t a = (uint32_t)i * 2654435761u + seed; a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15;
        x[i] = (unsigned short)(0x3800u | (a & 0x83ffu));          // +-[0.5, 1)
    }
}

template <int MC, int NS, int AUXP, int ABL>
float run(const char* tag, int B, int NK, int nset, unsigned short* X, uint32_t* P, unsigned short* C, int reps) {
    auto kern = delta_rows_kernel<DT_F16, MC, NS, AUXP, ABL, 4>;
    constexpr int lds = STREAM_LUT_BYTES + 4 * MC * 64 * 4;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const size_t pw = (size_t)B * (NK / 32) * NK;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    const int nl = getenv("ROWS_NL") ? atoi(getenv("ROWS_NL")) : 20;      // launches per timed repetition (the first repetition is the warm-up)
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < nl; ++i) {
            RowsParams rp{};
            rp.X = X; rp.P = P + (size_t)(i % nset) * pw; rp.C = C; rp.B = B; rp.M = 1; rp.rpm = 1; rp.N = NK; rp.K = NK;
            rp.sXb = NK; rp.sXm = NK; rp.sPb = (long long)(NK / 32) * NK; rp.sCb = NK; rp.sCm = NK;
            rp.x_bytes = (uint32_t)((size_t)B * NK * 2); rp.p_bytes = (uint32_t)(pw * 4);
            const unsigned grid = (unsigned)((NK / 64) * ((B + MC - 1) / MC));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, rp);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) ts.push_back(ms / nl * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    const float us = ts[ts.size() / 2];
    printf("  %-34s MC=%d NS=%d nt=%d abl=%d : %7.2f us  %5.2f TB/s\n", tag, MC, NS, AUXP ? 1 : 0, ABL, us, (double)B * NK * NK / 8 / us / 1e6);
    return us;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    const int only_cfg = getenv("ROWS_CFG") ? atoi(getenv("ROWS_CFG")) : -1;      // one shape (0..3) and the shipped form only: soak runs
    for (int cfg = 0; cfg < 4; ++cfg) {
        if (only_cfg >= 0 && cfg != only_cfg) continue;
        const int B = (cfg & 1) ? 16 : 8, NK = (cfg & 2) ? 8192 : 4096;
        const size_t pw = (size_t)B * (NK / 32) * NK;
        const int nset = (int)std::max<size_t>(2, 400000000 / (pw * 4) + 1);
        unsigned short *X, *C; uint32_t* P;
        CK(hipMalloc(&X, (size_t)B * NK * 2)); CK(hipMalloc(&C, (size_t)B * NK * 2)); CK(hipMalloc(&P, pw * 4 * nset));
        fill_f16<<<256, 256>>>(X, (size_t)B * NK, 7u); fill_u32<<<2048, 256>>>(P, pw * nset, 11u);
        CK(hipDeviceSynchronize());
        printf("B=%d N=K=%d (%d mask sets)\n", B, NK, nset);
        run<4, 4, 2, 0>("shipped form", B, NK, nset, X, P, C, reps);
        if (only_cfg >= 0) continue;
        run<2, 4, 2, 0>("2 masks per block", B, NK, nset, X, P, C, reps);
        run<4, 2, 2, 0>("2 stages", B, NK, nset, X, P, C, reps);
        run<4, 4, 2, 1>("no LUT reads", B, NK, nset, X, P, C, reps);
        run<4, 4, 2, 2>("no MFMAs", B, NK, nset, X, P, C, reps);
        run<4, 4, 2, 3>("no LUT reads, no MFMAs", B, NK, nset, X, P, C, reps);
        run<4, 4, 2, 4>("loads issued once", B, NK, nset, X, P, C, reps);
        run<4, 4, 2, 7>("loads once, no LUT, no MFMA", B, NK, nset, X, P, C, reps);
        CK(hipFree(X)); CK(hipFree(C)); CK(hipFree(P));
    }
    return 0;
}
