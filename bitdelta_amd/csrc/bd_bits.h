// Bit-plane kernels (all HBM-bound byte/integer work, bit-exact vs the reference):
//   pack_kernel      bitdelta/binary_gemm_kernel.py:6-32
//   unpack_kernel    bitdelta/binary_gemm_kernel.py:34-46
//   binarize_kernel  BinaryDiff.__init__, bitdelta/diff.py:9-31  (sign + mean|diff| + transposed pack, one pass)
//   merge_kernel     load_diff's dequant-merge line, bitdelta/diff.py:93-95
#pragma once
#include "bd_common.h"

namespace bd {

// ------------------------------------------------------------------------------------------------
// pack: word[b,i,n] = sum_j bit[b, nb*i + j, n] << j.  bits are torch.bool bytes with arbitrary element strides
// (diff.py:16 packs a transposed view).  One thread per output word, lanes along n (coalesced word stores; the
// bit loads are coalesced when s_n == 1).  WORD = uint8/uint16/uint32/uint64 for n_bits 8/16/32/64.
// The transposed-view case (s_k == 1) is served by pack_kmajor_kernel below.
template <typename WORD>
__global__ void __launch_bounds__(256) pack_kernel(const uint8_t* __restrict__ bits, WORD* __restrict__ out,
                                                   long long KW, long long N, long long s_b, long long s_k,
                                                   long long s_n) {
    constexpr int NB = sizeof(WORD) * 8;
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i = blockIdx.y, b = blockIdx.z;
    if (n >= N) return;
    const uint8_t* src = bits + b * s_b + (i * NB) * s_k + n * s_n;
    WORD w = 0;
#pragma unroll 8
    for (int j = 0; j < NB; ++j) w |= (WORD)(src[j * s_k] ? 1 : 0) << j;
    out[(b * KW + i) * N + n] = w;
}

// s_k == 1 (k contiguous, e.g. the .T view of an [N,K] bool matrix): a block transposes a [64 n] x [8 words] tile
// through LDS so both the 32-byte-per-word bit reads and the word stores are coalesced.  32-bit words only.
__global__ void __launch_bounds__(256) pack_kmajor_kernel(const uint8_t* __restrict__ bits, uint32_t* __restrict__ out,
                                                          long long KW, long long N, long long s_b, long long s_n) {
    __shared__ uint32_t tile[8][65];
    const long long n0 = (long long)blockIdx.x * 64, i0 = (long long)blockIdx.y * 8, b = blockIdx.z;
    // thread t: row n = t/4 (64 rows), handles words (t%4)*2, +1 of the 8 -> 64 contiguous bytes per thread
    const int rn = threadIdx.x >> 2, wq = (threadIdx.x & 3) * 2;
    const long long n = n0 + rn;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const long long i = i0 + wq + u;
        uint32_t w = 0;
        if (n < N && i < KW) {
            const uint8_t* src = bits + b * s_b + n * s_n + i * 32;
#pragma unroll
            for (int j = 0; j < 32; ++j) w |= (uint32_t)(src[j] ? 1u : 0u) << j;
        }
        tile[wq + u][rn] = w;
    }
    __syncthreads();
    const int wi = threadIdx.x >> 6, cn = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const long long i = i0 + wi * 2 + u;
        if (i < KW && n0 + cn < N) out[(b * KW + i) * N + n0 + cn] = tile[wi * 2 + u][cn];
    }
}

// ------------------------------------------------------------------------------------------------
// unpack: bit[b, nb*i + j, n] = (word[b,i,n] >> j) & 1  -> torch.bool bytes.  Thread = 4 consecutive n of one word
// row; 4-byte stores, lanes along n.
template <typename WORD>
__global__ void __launch_bounds__(256) unpack_kernel(const WORD* __restrict__ words, uint8_t* __restrict__ out,
                                                     long long KW, long long N) {
    constexpr int NB = sizeof(WORD) * 8;
    const long long n4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long long i = blockIdx.y, b = blockIdx.z;
    if (n4 >= N) return;
    const WORD* src = words + (b * KW + i) * N + n4;
    const int cnt = (int)min((long long)4, N - n4);
    unsigned long long w[4] = {0, 0, 0, 0};
    for (int c = 0; c < cnt; ++c) w[c] = (unsigned long long)src[c];
    uint8_t* dst = out + ((b * KW + i) * NB) * N + n4;
    const bool vec = (cnt == 4) && ((N & 3) == 0) && (((uintptr_t)out & 3) == 0);
#pragma unroll 8
    for (int j = 0; j < NB; ++j) {
        if (vec) {
            const uint32_t v = (uint32_t)((w[0] >> j) & 1) | ((uint32_t)((w[1] >> j) & 1) << 8) |
                               ((uint32_t)((w[2] >> j) & 1) << 16) | ((uint32_t)((w[3] >> j) & 1) << 24);
            *(uint32_t*)(dst + (long long)j * N) = v;
        } else {
            for (int c = 0; c < cnt; ++c) dst[(long long)j * N + c] = (uint8_t)((w[c] >> j) & 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// binarize: base, fine [N,K] row-major (ld elements) -> mask int32 [K/32, N] (bit k%32 of word [k/32, n] = (fine-base >= 0 at
// [n,k]), i.e. pack(bit.T), diff.py:14-16) and per-block partial sums of |diff| (diff rounded to the weights' dtype first,
// diff.py:11-12).  Block = 64 rows (n) x 256 columns (k): reads are 512-byte row runs, word stores are 256-byte runs
// after an LDS transpose.  partial[blockIdx.y * gridDim.x + blockIdx.x] = sum |diff| over the block (fp32; the final
// mean is taken in double by binarize_finish_kernel -> deterministic, no atomics).
template <int DT>
__global__ void __launch_bounds__(256) binarize_kernel(const unsigned short* __restrict__ base,
                                                       const unsigned short* __restrict__ fine, uint32_t* __restrict__ mask,
                                                       float* __restrict__ partial, int N, int K, long long ld) {
    __shared__ uint32_t tile[8][65];
    __shared__ float red[4];
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float asum = 0.f;
    // wave w owns rows 16w..16w+15; one pass = 2 rows x 256 k (lane: row = lane/32, 8 k at 8*(lane%32))
#pragma unroll 2
    for (int pss = 0; pss < 8; ++pss) {
        const int rn = wave * 16 + pss * 2 + (lane >> 5);
        const int n = n0 + rn, k = k0 + (lane & 31) * 8;
        uint32_t byte = 0;
        if (n < N && k < K) {     // K % 32 == 0 and k % 8 == 0 -> the 8 elements are all in range
            const unsigned short* pb = base + (long long)n * ld + k;
            const unsigned short* pf = fine + (long long)n * ld + k;
            unsigned short vb[8], vf[8];
            if ((((uintptr_t)pb | (uintptr_t)pf) & 15) == 0) {
                *(uint4*)vb = *(const uint4*)pb;
                *(uint4*)vf = *(const uint4*)pf;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { vb[e] = pb[e]; vf[e] = pf[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = half_bits_to_f32<DT>(f32_to_half_bits<DT>(half_bits_to_f32<DT>(vf[e]) - half_bits_to_f32<DT>(vb[e])));
                asum += fabsf(d);
                byte |= (d < 0.f ? 0u : 1u) << e;     // zero, -0.0 and NaN stay 1 (diff.py:14-15)
            }
        }
        // 4 adjacent lanes hold the 4 bytes of one word
        uint32_t w = byte << (8 * (lane & 3));
        w |= __shfl_xor(w, 1);
        w |= __shfl_xor(w, 2);
        if ((lane & 3) == 0) tile[(lane & 31) >> 2][rn] = w;
    }
    // block sum of |diff|
    for (int o = 32; o > 0; o >>= 1) asum += __shfl_xor(asum, o);
    if (lane == 0) red[wave] = asum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    // store 8 word rows x 64 n
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int wi = wave * 2 + u;
        const int i = (k0 >> 5) + wi;
        if (i < (K >> 5) && n0 + lane < N) mask[(long long)i * N + n0 + lane] = tile[wi][lane];
    }
}

__global__ void __launch_bounds__(256) binarize_finish_kernel(const float* __restrict__ partial, int nparts, double inv_count,
                                                              float* __restrict__ coeff) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *coeff = (float)(red[0] * inv_count);
}

// ------------------------------------------------------------------------------------------------
// merge: W[n,k] = round(W[n,k] + round(+-coeff)) with the sign from bit (k%32) of P[k/32, n]  (diff.py:93-95).
// Block = 64 n x 256 k; the 8 x 64 sign words are read as 256-byte runs into LDS, W is streamed in 512-byte row runs.
template <int DT>
__global__ void __launch_bounds__(256) merge_kernel(unsigned short* __restrict__ W, const uint32_t* __restrict__ P,
                                                    const float* __restrict__ coeff_ptr, int N, int K, long long ldw) {
    __shared__ uint32_t tile[8][65];
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int wi = wave * 2 + u, i = (k0 >> 5) + wi;
        tile[wi][lane] = (i < (K >> 5) && n0 + lane < N) ? P[(long long)i * N + n0 + lane] : 0u;
    }
    __syncthreads();
    const float c = *coeff_ptr;
    const float cp = half_bits_to_f32<DT>(f32_to_half_bits<DT>(c)), cn = half_bits_to_f32<DT>(f32_to_half_bits<DT>(-c));
#pragma unroll 2
    for (int pss = 0; pss < 8; ++pss) {
        const int rn = wave * 16 + pss * 2 + (lane >> 5);
        const int n = n0 + rn, k = k0 + (lane & 31) * 8;
        if (n >= N || k >= K) continue;
        const uint32_t byte = (tile[(lane & 31) >> 2][rn] >> (8 * (lane & 3))) & 0xffu;
        unsigned short* pw = W + (long long)n * ldw + k;
        unsigned short v[8];
        const bool al = (((uintptr_t)pw) & 15) == 0;
        if (al) *(uint4*)v = *(const uint4*)pw;
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = pw[e];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (unsigned short)f32_to_half_bits<DT>(half_bits_to_f32<DT>(v[e]) + (((byte >> e) & 1u) ? cp : cn));
        if (al) *(uint4*)pw = *(const uint4*)v;
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) pw[e] = v[e];
    }
}

}  // namespace bd
