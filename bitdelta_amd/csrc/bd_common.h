// Shared device/host declarations for the MI355X (gfx950) BitDelta hot path.
// gfx950 only: wave64, MFMA 32x32x16, LDS-DMA (global_load_lds), 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bd {

enum : int { DT_F16 = 0, DT_BF16 = 1, DT_F32 = 2 };

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// ---- 16-bit float helpers (raw bit patterns; round-to-nearest-even) ----
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;      // quiet NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t h) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)h);
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
    return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f);    // v_cvt_f16_f32: RNE, overflow -> inf
}
template <int DT> __device__ __forceinline__ float half_bits_to_f32(uint32_t h) {
    return DT == DT_BF16 ? bf16_bits_to_f32(h) : f16_bits_to_f32(h);
}
template <int DT> __device__ __forceinline__ uint32_t f32_to_half_bits(float f) {
    return DT == DT_BF16 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f);
}
// reference epilogue `accumulator.to(tl.float16)` (bitdelta/binary_gemm_kernel.py:143, :287)
__device__ __forceinline__ float round_through_f16(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }

// +1.0 in both halves of a dword, per dtype
template <int DT> struct One2 { static constexpr uint32_t v = (DT == DT_BF16) ? 0x3F803F80u : 0x3C003C00u; };

// D(32x32) += A(32x16) * B(16x32); lane l supplies A[i = l&31][k = 8*(l>>5)+0..7] and B[k = 8*(l>>5)+0..7][j = l&31];
// D: column j = l&31, row i = (reg&3) + 8*(reg>>2) + 4*(l>>5).
template <int DT>
__device__ __forceinline__ f32x16_t mfma32(u32x4_t a, u32x4_t b, f32x16_t c) {
    if constexpr (DT == DT_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// Expand 8 sign bits into 8 x (+-1.0) 16-bit floats (4 dwords).  `rep` holds a 16-bit chunk of the
// (already inverted) packed word replicated in both halves; pair q (bits 2q, 2q+1 of the chunk) is moved
// to the two sign positions by one packed 16-bit shift, then masked and OR-ed onto 1.0|1.0:
// 2 VALU per dword (v_pk_lshlrev_b16 + v_and_or_b32).  bit = 1 -> +1.0, bit = 0 -> -1.0
// (bitdelta/binary_gemm_kernel.py:128-129, :270-272), because the word was inverted beforehand.
__device__ __forceinline__ u32x4_t expand_signs8(uint32_t rep, int q0, uint32_t one2_vgpr) {
    u32x4_t r;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int q = q0 + d;
        u16x2_t v = __builtin_bit_cast(u16x2_t, rep);
        u16x2_t sh;
        sh.x = (unsigned short)(15 - 2 * q);
        sh.y = (unsigned short)(14 - 2 * q);
        v = v << sh;
        r[d] = (__builtin_bit_cast(uint32_t, v) & 0x80008000u) | one2_vgpr;
    }
    return r;
}

// ---- LDS-DMA issued from inline asm so hipcc's waitcnt pass neither sees it nor drains it with vmcnt(0)
//      before every ds_read (measured with the builtin: `s_waitcnt vmcnt(0)` ahead of the first ds_read_b128
//      of each k-tile).  Completion is counted by hand: s_waitcnt vmcnt(N) + s_barrier before the first read.
//      LDS destination = lds_addr (wave-uniform, goes through M0) + lane*size.
__device__ __forceinline__ void dma16(uint32_t voff, const void* sbase, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_addr)
        : "memory");
}
__device__ __forceinline__ void dma4(uint32_t voff, const void* sbase, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_addr)
        : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// bijective XCD-aware remap: hardware places block b on XCD b % 8; give each XCD a contiguous run of tiles.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

}  // namespace bd
