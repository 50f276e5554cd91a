// Decode-step glue of the multi-tenant serving loop (demo/demo_backend.py:190-258 is the loop, the HF
// Llama / Mistral decoder layer is what it calls between two of the reference's Linears).  These three kernels are NOT the hot
// path -- they move a few hundred KB per layer -- but at decode every torch op is a launch, and a layer written with stock ops is
// ~30 launches around 4 weight-streaming Linears.  HBM/latency-bound, elementwise or tiny reductions; fp32 math, results rounded
// where the torch ops they replace round (noted per kernel), so they are compared against those ops in tests/test_gpu_serving.py.
//
//   rmsnorm_tenant_kernel   y[r] = w[tenant(r)] * round16(x[r] * rsqrt(mean(x[r]^2) + eps))        (HF RMSNorm, per-tenant weight)
//   swiglu_kernel           y = round16(silu(g)) * u     from the fused gate|up output               (HF MLP act_fn(gate) * up)
//   rope_kernel             in-place rotary embedding of a q / k projection output (prefill)
//   decode_attn_kernel      RoPE(q, k_new) -> KV-cache append -> softmax(q.K^T / sqrt(d)) . V        one new token per tenant, GQA
#pragma once
#include "bd_common.h"

namespace bd {

template <int DT> __device__ __forceinline__ float round16(float v) { return half_bits_to_f32<DT>(f32_to_half_bits<DT>(v)); }

// sum over the 16 lanes of a DPP row, result in every lane: quad xor 1, quad xor 2, half-row mirror, row mirror -- four VALU ops.
// (__shfl_xor lowers to ds_bpermute_b32: an LDS-crossbar round trip per step; four dependent ones per head per key row made the
// decode attention kernel latency-chain-bound at ~3000 cycles per 32 rows.)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}

// sum over the wave, result in every lane: DPP within the 16-lane rows, then the four row sums through SGPRs in a fixed order
// (six dependent ds_bpermute round trips of a __shfl_xor butterfly cost ~400 cycles; this is ~12 VALU ops)
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    const int i = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
    return (r0 + r1) + (r2 + r3);
}

// The three pieces of HF RMSNorm / SwiGLU arithmetic, shared with the fused prologue / epilogue of the streaming decode kernel
// (bd_gemv_stream.h, XL / EPI) so that the fused and the stand-alone forms are bit-identical: explicit fmaf, fixed order.
template <int DT> __device__ __forceinline__ float sumsq8(u32x4_t v, float ss) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float a = half_bits_to_f32<DT>(v[d] & 0xffffu), b = half_bits_to_f32<DT>(v[d] >> 16);
        ss = __builtin_fmaf(a, a, ss);
        ss = __builtin_fmaf(b, b, ss);
    }
    return ss;
}
// w * round16(x * rs), 8 elements
template <int DT> __device__ __forceinline__ u32x4_t norm8(u32x4_t v, u32x4_t g, float rs) {
    u32x4_t o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float a = round16<DT>(half_bits_to_f32<DT>(v[d] & 0xffffu) * rs), b = round16<DT>(half_bits_to_f32<DT>(v[d] >> 16) * rs);
        const uint32_t lo = f32_to_half_bits<DT>(a * half_bits_to_f32<DT>(g[d] & 0xffffu));
        const uint32_t hi = f32_to_half_bits<DT>(b * half_bits_to_f32<DT>(g[d] >> 16));
        o[d] = lo | (hi << 16);
    }
    return o;
}
// round16(silu(g)) * u  for 16-bit g, u (already rounded projection outputs); result as 16-bit pattern
template <int DT> __device__ __forceinline__ uint32_t swiglu1(float g, float u) {
    const float a = round16<DT>(g / (1.f + __expf(-g)));
    return f32_to_half_bits<DT>(a * u);
}
__device__ __forceinline__ float rms_scale(float p0, float p1, float p2, float p3, int H, float eps) {
    return rsqrtf((p0 + p1 + p2 + p3) / (float)H + eps);
}

// one block (256 threads) per row; H % 8 == 0
template <int DT>
__global__ void __launch_bounds__(256) rmsnorm_tenant_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                             unsigned short* __restrict__ y, int H, long long sx, long long sy,
                                                             long long sw, int rows_per_tenant, float eps) {
    __shared__ float part[4];
    const int r = blockIdx.x, t = r / rows_per_tenant;
    const unsigned short* xr = x + (long long)r * sx;
    const unsigned short* wr = w + (long long)t * sw;
    unsigned short* yr = y + (long long)r * sy;
    float ss = 0.f;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) ss = sumsq8<DT>(*(const u32x4_t*)(xr + c), ss);
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rs = rms_scale(part[0], part[1], part[2], part[3], H, eps);
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8)
        *(u32x4_t*)(yr + c) = norm8<DT>(*(const u32x4_t*)(xr + c), *(const u32x4_t*)(wr + c), rs);
}

// residual add + RMSNorm in ONE launch (round 6; the tensor-parallel decoder: `x = residual + all_reduce(partial).to(dtype); h = norm(x)` was a cast, an
// add and a norm launch -- three ~5-us launches around every row-parallel Linear of a per-rank step whose Linears are 10-25 us).  Same arithmetic and
// roundings as the three ops: v = round16(y32), x = round16(residual + v) (torch's 16-bit add), then rmsnorm_tenant_kernel's two passes on x, which stays
// in registers.  One 256-thread block per row, H % 8 == 0, H <= 8192.
template <int DT>
__global__ void __launch_bounds__(256) add_rmsnorm_kernel(const unsigned short* __restrict__ resid, const float* __restrict__ y32,
                                                          const unsigned short* __restrict__ w, unsigned short* __restrict__ x_out,
                                                          unsigned short* __restrict__ h_out, int H, long long s_r, long long s_y, long long s_x,
                                                          long long s_h, long long sw, int rows_per_tenant, float eps) {
    __shared__ float part[4];
    const int r = blockIdx.x, t = r / rows_per_tenant;
    const unsigned short* rr = resid + (long long)r * s_r;
    const float* yr = y32 + (long long)r * s_y;
    const unsigned short* wr = w + (long long)t * sw;
    u32x4_t xs[4], wv[4];
    float ss = 0.f;
    // the norm weights are fetched with the row, not after the barrier: a one-row launch (the decode step of tp.py) is a chain of memory round trips
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x * 8 + 2048 * i;
        wv[i] = c < H ? *(const u32x4_t*)(wr + c) : u32x4_t{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x * 8 + 2048 * i;
        xs[i] = u32x4_t{0u, 0u, 0u, 0u};
        if (c < H) {
            const u32x4_t rv = *(const u32x4_t*)(rr + c);
            const f32x4_t ya = *(const f32x4_t*)(yr + c), yb = *(const f32x4_t*)(yr + c + 4);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float y0 = d < 2 ? ya[2 * d] : yb[2 * d - 4], y1 = d < 2 ? ya[2 * d + 1] : yb[2 * d - 3];
                const float lo = half_bits_to_f32<DT>(rv[d] & 0xffffu) + round16<DT>(y0), hi = half_bits_to_f32<DT>(rv[d] >> 16) + round16<DT>(y1);
                xs[i][d] = f32_to_half_bits<DT>(lo) | (f32_to_half_bits<DT>(hi) << 16);
            }
            *(u32x4_t*)(x_out + (long long)r * s_x + c) = xs[i];
            ss = sumsq8<DT>(xs[i], ss);
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rs = rms_scale(part[0], part[1], part[2], part[3], H, eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x * 8 + 2048 * i;
        if (c < H) *(u32x4_t*)(h_out + (long long)r * s_h + c) = norm8<DT>(xs[i], wv[i], rs);
    }
}

// Split-k reduce + residual + the NEXT RMSNorm in ONE launch (round 6; the multi-tenant prefill of short prompts: the o / down pair tiles of a
// 6 x 64-row request are split over k, and the layer ran splitk_reduce_kernel (5.7 us) and then rmsnorm_rows_kernel (7.5 us) on the same 384 rows).
// Arithmetic of the two launches, in their order: s = ((0 + part[0]) + part[1]) + ... in slice order, x = round16(round16(s) + residual) (or
// round16(s) without a residual), then rmsnorm_tenant_kernel's two passes on x, which stays in registers.  ws [B][KS][M][N] fp32; C (residual in,
// x out) [B][M][N] with strides sCb / sCm; h [B][M][N] with strides sHb / sHm; w [B][N] (stride sw).  One 256-thread block per row; N % 8 == 0,
// N <= 8192.
template <int DT>
__global__ void __launch_bounds__(256) splitk_reduce_norm_kernel(const float* __restrict__ ws, unsigned short* __restrict__ C,
                                                                 const unsigned short* __restrict__ w, unsigned short* __restrict__ h_out, int KS,
                                                                 int M, int N, long long sCb, long long sCm, long long sHb, long long sHm,
                                                                 long long sw, int accumulate, float eps) {
    __shared__ float part[4];
    const int b = blockIdx.x / M, m = blockIdx.x - b * M;
    const float* slab = ws + ((long long)b * KS * M + m) * N;           // slice k of this row: slab + k * M * N
    unsigned short* cr = C + (long long)b * sCb + (long long)m * sCm;
    const unsigned short* wr = w + (long long)b * sw;
    u32x4_t xs[4], wv[4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                       // (the norm weights with the first loads, not after the barrier)
        const int c = threadIdx.x * 8 + 2048 * i;
        wv[i] = c < N ? *(const u32x4_t*)(wr + c) : u32x4_t{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x * 8 + 2048 * i;
        xs[i] = u32x4_t{0u, 0u, 0u, 0u};
        if (c < N) {
            f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < KS; k0 += 4) {                        // four slices' loads in flight, then their adds in slice order
                f32x4_t va[4], vb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* q = slab + (long long)min(k0 + j, KS - 1) * M * N + c;
                    va[j] = *(const f32x4_t*)q;
                    vb[j] = *(const f32x4_t*)(q + 4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + j < KS) { sa += va[j]; sb += vb[j]; }
            }
            u32x4_t rv = {0u, 0u, 0u, 0u};
            if (accumulate) rv = *(const u32x4_t*)(cr + c);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float y0 = d < 2 ? sa[2 * d] : sb[2 * d - 4], y1 = d < 2 ? sa[2 * d + 1] : sb[2 * d - 3];
                uint32_t lo = f32_to_half_bits<DT>(y0), hi = f32_to_half_bits<DT>(y1);
                if (accumulate) {
                    lo = f32_to_half_bits<DT>(half_bits_to_f32<DT>(lo) + half_bits_to_f32<DT>(rv[d] & 0xffffu));
                    hi = f32_to_half_bits<DT>(half_bits_to_f32<DT>(hi) + half_bits_to_f32<DT>(rv[d] >> 16));
                }
                xs[i][d] = lo | (hi << 16);
            }
            *(u32x4_t*)(cr + c) = xs[i];
            ss = sumsq8<DT>(xs[i], ss);
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rs = rms_scale(part[0], part[1], part[2], part[3], N, eps);
    unsigned short* hr = h_out + (long long)b * sHb + (long long)m * sHm;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x * 8 + 2048 * i;
        if (c < N) *(u32x4_t*)(hr + c) = norm8<DT>(xs[i], wv[i], rs);
    }
}

// The same norm for MANY rows (prefill: hundreds to thousands of rows): ONE WAVE per row, four rows per block, the row held in registers between
// the two passes -- no block barrier, no second read of x.  Bit-identical to rmsnorm_tenant_kernel: lane l plays that kernel's threads l, l + 64,
// l + 128, l + 192 (its four waves), so the four per-wave sums are formed by the same lanes in the same order and meet in rms_scale as before.
// H = 2048 * NCH, NCH = 1 .. 4 (host-checked).  One block-per-row launch of 2048 rows x 4096 measured 13.1 us against torch's 8.5
// (profiles/r03_prefill_glue.txt): a 256-thread block per 8-KB row is one 16-byte load per thread and a barrier.
template <int DT, int NCH>
__global__ void __launch_bounds__(256) rmsnorm_rows_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                           unsigned short* __restrict__ y, int rows, long long sx, long long sy,
                                                           long long sw, int rows_per_tenant, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;                                                // (wave-uniform; no barrier in this kernel)
    const int t = r / rows_per_tenant;
    const unsigned short* xr = x + (long long)r * sx;
    const unsigned short* wr = w + (long long)t * sw;
    unsigned short* yr = y + (long long)r * sy;
    u32x4_t xv[4][NCH], wv[4][NCH];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int i = 0; i < NCH; ++i) xv[v][i] = *(const u32x4_t*)(xr + 8 * (lane + 64 * v) + 2048 * i);
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int i = 0; i < NCH; ++i) wv[v][i] = *(const u32x4_t*)(wr + 8 * (lane + 64 * v) + 2048 * i);
    float part[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) ss = sumsq8<DT>(xv[v][i], ss);
        part[v] = wave_sum(ss);
    }
    const float rs = rms_scale(part[0], part[1], part[2], part[3], 2048 * NCH, eps);
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int i = 0; i < NCH; ++i) *(u32x4_t*)(yr + 8 * (lane + 64 * v) + 2048 * i) = norm8<DT>(xv[v][i], wv[v][i], rs);
}

// g, u [rows, I] (row strides sg, su; the fused gate|up output passes u = g + I) -> y [rows, I];  I % 8 == 0.
// il8 = 1: the projection output is interleaved in blocks of 8 ([g0..7 | u0..7 | g8..15 | u8..15 ...], the row order
// FusedDeltaLinear gives a gate|up pair so that the decode kernel can apply SwiGLU in its epilogue): g = gp + 2c, u = gp + 2c + 8.
template <int DT>
__global__ void __launch_bounds__(256) swiglu_kernel(const unsigned short* __restrict__ gp, const unsigned short* __restrict__ up,
                                                     unsigned short* __restrict__ y, int I, long long sg, long long su, long long sy,
                                                     int il8) {
    const int r = blockIdx.y;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (c >= I) return;
    const u32x4_t g = *(const u32x4_t*)(gp + (long long)r * sg + (il8 ? 2 * c : c));
    const u32x4_t u = il8 ? *(const u32x4_t*)(gp + (long long)r * sg + 2 * c + 8) : *(const u32x4_t*)(up + (long long)r * su + c);
    u32x4_t o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t lo = swiglu1<DT>(half_bits_to_f32<DT>(g[d] & 0xffffu), half_bits_to_f32<DT>(u[d] & 0xffffu));
        const uint32_t hi = swiglu1<DT>(half_bits_to_f32<DT>(g[d] >> 16), half_bits_to_f32<DT>(u[d] >> 16));
        o[d] = lo | (hi << 16);
    }
    *(u32x4_t*)(y + (long long)r * sy + c) = o;
}

// The two ends of a greedy decode step (round 6; demo/demo_backend.py:190-258 is the loop): what `valid.index_fill_(1, pos, True)` + the per-tenant
// embedding gather do before the layers, and what argmax + the token / output / stop-flag / position updates do after the lm_head -- 2 launches
// instead of ~13 stock ones (each ~5 us inside the replayed graph: ~60 us of a 4.5-ms step).  One block per tenant; integer work, exact.
__global__ void __launch_bounds__(256) step_begin_kernel(const unsigned short* __restrict__ embed, long long sEt, long long sEv,
                                                         const long long* __restrict__ tok, unsigned short* __restrict__ x, long long sx,
                                                         unsigned char* __restrict__ valid, int Lc, const long long* __restrict__ pos, int V,
                                                         int H) {
    const int t = blockIdx.x;
    long long id = tok[t];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);                           // (device data the host cannot validate: clamp into the table)
    const unsigned short* row = embed + (long long)t * sEt + id * sEv;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) *(u32x4_t*)(x + (long long)t * sx + c) = *(const u32x4_t*)(row + c);
    if (threadIdx.x == 0) {
        const long long p = *pos;
        if (p >= 0 && p < Lc) valid[(long long)t * Lc + p] = 1;
    }
}

// torch.argmax's order: a NaN is the maximum, ties go to the lower index
__device__ __forceinline__ bool argmax_better(float a, int ia, float b, int ib) {
    const bool an = a != a, bn = b != b;
    if (an || bn) return an && (!bn || ia < ib);
    return a > b || (a == b && ia < ib);
}
template <int DT>
__global__ void __launch_bounds__(256) step_end_kernel(const unsigned short* __restrict__ logits, long long sl, int V, long long* __restrict__ tok,
                                                       long long* __restrict__ out, long long s_out, int out_cap,
                                                       const long long* __restrict__ stop_ids, int ns, unsigned char* __restrict__ stopped,
                                                       long long* pos, long long* step, unsigned int* ticket, int T) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int t = blockIdx.x;
    const unsigned short* row = logits + (long long)t * sl;
    float best = -__builtin_inff();
    int bidx = 0x7fffffff;
    for (int c = threadIdx.x * 8; c < V; c += 256 * 8) {
        const u32x4_t v = *(const u32x4_t*)(row + c);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float lo = half_bits_to_f32<DT>(v[d] & 0xffffu), hi = half_bits_to_f32<DT>(v[d] >> 16);
            if (argmax_better(lo, c + 2 * d, best, bidx)) { best = lo; bidx = c + 2 * d; }
            if (argmax_better(hi, c + 2 * d + 1, best, bidx)) { best = hi; bidx = c + 2 * d + 1; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        if (argmax_better(ov, oi, best, bidx)) { best = ov; bidx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (argmax_better(bv[w], bi[w], best, bidx)) { best = bv[w]; bidx = bi[w]; }
        const long long nxt = bidx, s = *step;
        tok[t] = nxt;
        if (s >= 0 && s < out_cap) out[(long long)t * s_out + s] = nxt;
        unsigned char st = stopped[t];
        for (int j = 0; j < ns; ++j) st |= (stop_ids[(long long)t * ns + j] == nxt) ? 1 : 0;
        stopped[t] = st;
        // the last tenant's block advances the shared position / step counters: every block has read them before its ticket
        __threadfence();
        if (atomicAdd(ticket, 1u) == (unsigned)(T - 1)) {
            *ticket = 0u;
            *pos += 1;
            *step += 1;
        }
    }
}

// In-place rotary embedding of [rows, heads * 128] (a q or k projection output before the head transpose), HF rotate-half form with
// the sign folded into `sin`: out[d] = round16(round16(x[d] * cos[d]) + x[d +- 64] * sin[d])  -- the rounding points of the stock
// composition `torch.addcmul(x * cos, rot, sin)`, which costs five passes (cat, mul, addcmul + two temporaries) instead of one.
// Position of row r = pos0 + r % seq.  One block per row, one thread per (head, 8 dims of the lower half + their partners).
template <int DT>
__global__ void __launch_bounds__(256) rope_kernel(unsigned short* __restrict__ x, const unsigned short* __restrict__ cos_t,
                                                   const unsigned short* __restrict__ sin_t, int heads, long long sx, int seq, int pos0) {
    const int r = blockIdx.x;
    const long long pos = pos0 + r % seq;
    const unsigned short* cs = cos_t + pos * 128;
    const unsigned short* sn = sin_t + pos * 128;
    for (int i = threadIdx.x; i < heads * 8; i += 256) {
        const int h = i >> 3, d0 = (i & 7) * 8;
        unsigned short* px = x + (long long)r * sx + h * 128;
        const u32x4_t lo = *(const u32x4_t*)(px + d0), hi = *(const u32x4_t*)(px + 64 + d0);
        const u32x4_t cl = *(const u32x4_t*)(cs + d0), ch = *(const u32x4_t*)(cs + 64 + d0);
        const u32x4_t sl = *(const u32x4_t*)(sn + d0), sh = *(const u32x4_t*)(sn + 64 + d0);
        u32x4_t ol, oh;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t rl = 0, rh = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int shf = 16 * e;
                const float a = half_bits_to_f32<DT>((lo[d] >> shf) & 0xffffu), b = half_bits_to_f32<DT>((hi[d] >> shf) & 0xffffu);
                const float c0 = half_bits_to_f32<DT>((cl[d] >> shf) & 0xffffu), c1 = half_bits_to_f32<DT>((ch[d] >> shf) & 0xffffu);
                const float s0 = half_bits_to_f32<DT>((sl[d] >> shf) & 0xffffu), s1 = half_bits_to_f32<DT>((sh[d] >> shf) & 0xffffu);
                rl |= f32_to_half_bits<DT>(round16<DT>(a * c0) + b * s0) << shf;
                rh |= f32_to_half_bits<DT>(round16<DT>(b * c1) + a * s1) << shf;
            }
            ol[d] = rl; oh[d] = rh;
        }
        *(u32x4_t*)(px + d0) = ol;
        *(u32x4_t*)(px + 64 + d0) = oh;
    }
}

// Prefill RoPE + KV-cache append in ONE launch (round 6): rope_kernel on the q and k heads of the fused q|k|v projection output [T * S, (H + 2 KVH) *
// 128] (in place -- the prefill attention kernel reads q / k / v from that buffer), and the rotated k rows and the v rows of every token also go to
// the caches [T, KVH, Lc, 128] at positions pos0 .. pos0 + S - 1: the two strided torch copies `cache[:, :, :S] = k.transpose(1, 2)` (6.3 + 5.6 us
// per layer on a 6 x 64-token request) ride on a launch that already has the rows in registers.  One block per token row.
template <int DT>
__global__ void __launch_bounds__(256) rope_kv_append_kernel(unsigned short* __restrict__ x, const unsigned short* __restrict__ cos_t,
                                                             const unsigned short* __restrict__ sin_t, unsigned short* __restrict__ kc,
                                                             unsigned short* __restrict__ vc, int H, int KVH, long long sx, int seq, int pos0,
                                                             int Lc) {
    const int r = blockIdx.x, t = r / seq;
    const long long pos = pos0 + (r - t * seq);
    const unsigned short* cs = cos_t + pos * 128;
    const unsigned short* sn = sin_t + pos * 128;
    unsigned short* row = x + (long long)r * sx;
    const int nrope = (H + KVH) * 8, ntot = nrope + KVH * 16;
    for (int i = threadIdx.x; i < ntot; i += 256) {
        if (i >= nrope) {                                               // v head (i - nrope) / 16, 16-byte chunk (i - nrope) % 16: copy
            const int j = i - nrope, hv = j >> 4, d0 = (j & 15) * 8;
            *(u32x4_t*)(vc + (((long long)t * KVH + hv) * Lc + pos) * 128 + d0) = *(const u32x4_t*)(row + (long long)(H + KVH + hv) * 128 + d0);
            continue;
        }
        const int h = i >> 3, d0 = (i & 7) * 8;
        unsigned short* px = row + h * 128;
        const u32x4_t lo = *(const u32x4_t*)(px + d0), hi = *(const u32x4_t*)(px + 64 + d0);
        const u32x4_t cl = *(const u32x4_t*)(cs + d0), ch = *(const u32x4_t*)(cs + 64 + d0);
        const u32x4_t sl = *(const u32x4_t*)(sn + d0), sh = *(const u32x4_t*)(sn + 64 + d0);
        u32x4_t ol, oh;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t rl = 0, rh = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int shf = 16 * e;
                const float a = half_bits_to_f32<DT>((lo[d] >> shf) & 0xffffu), b = half_bits_to_f32<DT>((hi[d] >> shf) & 0xffffu);
                const float c0 = half_bits_to_f32<DT>((cl[d] >> shf) & 0xffffu), c1 = half_bits_to_f32<DT>((ch[d] >> shf) & 0xffffu);
                const float s0 = half_bits_to_f32<DT>((sl[d] >> shf) & 0xffffu), s1 = half_bits_to_f32<DT>((sh[d] >> shf) & 0xffffu);
                rl |= f32_to_half_bits<DT>(round16<DT>(a * c0) + b * s0) << shf;
                rh |= f32_to_half_bits<DT>(round16<DT>(b * c1) + a * s1) << shf;
            }
            ol[d] = rl; oh[d] = rh;
        }
        *(u32x4_t*)(px + d0) = ol;
        *(u32x4_t*)(px + 64 + d0) = oh;
        if (h >= H) {
            unsigned short* pk = kc + (((long long)t * KVH + (h - H)) * Lc + pos) * 128;
            *(u32x4_t*)(pk + d0) = ol;
            *(u32x4_t*)(pk + 64 + d0) = oh;
        }
    }
}

struct AttnParams {
    const unsigned short* qkv;     // [T, (H + 2*KVH) * 128]: q heads, then k heads, then v heads (the fused q+k+v Linear's output row)
    const unsigned short* cos;     // [Lmax, 128] RoPE tables in the activation dtype, rotate-half sign folded into `sin`
    const unsigned short* sin;
    unsigned short* kc;            // KV cache [T, KVH, Lc, 128]
    unsigned short* vc;
    unsigned char* valid;          // [T, Lc] key-validity bytes (left padding = 0); the new position is set here
    const long long* pos;          // device scalar: cache position of the new token
    unsigned short* out;           // [T, H * 128]
    int T, H, KVH, Lc;
    long long s_qkv, s_out;        // row strides (elements)
    float scale;                   // 1 / sqrt(head_dim)
    float* ws;                     // nsplit > 1: partial (acc[128], max, sum) per (tenant, kv head, split, query head): [.., G, 130] fp32
    unsigned* tickets;             // nsplit > 1: one arrival counter per (tenant, kv head), zero when the launch is enqueued
    int nsplit;                    // key range split over blockIdx.y (one CU streams only ~12-25 GB/s: 48 blocks cannot feed on HBM)
};

// One block of 8 waves per (tenant, kv head): its G = H / KVH query heads share the K / V stream.  Lane (kq = l >> 4, d8 = l & 15)
// reads 16 bytes (dims 8*d8 .. +7) of key / value row l0 + kq, so a wave instruction covers 4 whole 256-byte rows and the block 32
// rows per iteration; a DEPTH-deep register ring keeps the rows of the next DEPTH iterations in flight, so the loop does not pay one
// memory round trip per iteration (DEPTH is picked by the launch code: 2 for short caches, 4 for long ones -- deeper measured slower
// on the decode step at 512 keys, like every other prefetch depth of the step: profiles/r04_decode_step_ab.txt).  Online softmax per head in fp32; the 4 row slots of a wave are merged
// with lane shuffles, the 8 waves through LDS.  head_dim = 128, G in {1, 4, 8}.
// (History, profiles/r02_decode_step.txt: 4 waves, no prefetch -> 16 waves, one iteration ahead: 21 us per layer -> this form.)
__device__ __forceinline__ void attn_ws_store(float* dst, float v) { __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float attn_ws_load(const float* src) {
    return __hip_atomic_load(const_cast<float*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MAXS = key-range splits per (tenant, kv head) the in-launch merge is written for (host: nsplit <= MAXS): 4, or 16 for launches with few
// (tenant, kv head) pairs -- a single sequence on 8 kv heads is 32 blocks at 4 splits
template <int DT, int G, int DEPTH = 4, int MAXS = 4>
__global__ void __launch_bounds__(512) decode_attn_kernel(const AttnParams p) {
    constexpr int ATTN_MAX_SPLITS = MAXS;
    constexpr int HD = 128, NWV = 8, RPI = 4 * NWV;
    __shared__ float q_lds[G][HD];              // rotated, pre-scaled queries
    __shared__ float kn_lds[HD], vn_lds[HD];    // the new token's rotated key / value (also written to the cache)
    __shared__ float m_lds[NWV][G], s_lds[NWV][G];
    __shared__ float a_lds[NWV][G][HD];
    const int t = blockIdx.x / p.KVH, kvh = blockIdx.x % p.KVH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long pos = *p.pos;
    const unsigned short* row = p.qkv + (long long)t * p.s_qkv;
    const unsigned short* cs = p.cos + pos * HD;
    const unsigned short* sn = p.sin + pos * HD;
    unsigned short* kbase = p.kc + ((long long)t * p.KVH + kvh) * p.Lc * HD;
    unsigned short* vbase = p.vc + ((long long)t * p.KVH + kvh) * p.Lc * HD;
    const unsigned char* vld = p.valid + (long long)t * p.Lc;
    const int kq = lane >> 4, d8 = lane & 15;
    const int slot = 4 * wave + kq;                      // this lane group's row slot (0 .. RPI-1)
    // this block's key rows [l_lo, l_hi) of 0 .. pos (whole iterations of RPI rows per split)
    const long long nrows_all = pos + 1;
    const long long per_split = ((nrows_all + p.nsplit - 1) / p.nsplit + RPI - 1) / RPI * RPI;
    const long long l_lo = (long long)blockIdx.y * per_split, l_hi = l_lo + per_split < nrows_all ? l_lo + per_split : nrows_all;
    const bool owns_new = l_lo <= pos && pos < l_hi;      // the split that holds the new token appends it to the cache

    // (the validity byte is kept RAW and tested where the row is used: compared here -- `ok = in && vld[lc] != 0` -- the byte load sat behind a
    //  branch and its compare waited vmcnt(0) right after the row's K / V loads were issued, i.e. one full memory round trip per ring slot in the
    //  prologue and per refill in the loop; read in the ISA, round 5)
    struct Rows { u32x4_t kk, vv; uint32_t vb; bool in; };
    auto load_row = [&](long long l, Rows& r) {
        const bool in = l < pos && l < l_hi;              // row `pos` itself comes from LDS (it is being written by its block)
        const long long lc = in ? l : 0;
        r.kk = *(const u32x4_t*)(kbase + lc * HD + 8 * d8);
        r.vv = *(const u32x4_t*)(vbase + lc * HD + 8 * d8);
        r.vb = vld[lc];
        r.in = in;
    };
    // ---- phase 0 inputs FIRST (raw bits; L2 hits -- the row was just written by the q|k|v launch), the K / V ring behind them, the arithmetic
    //      after both: loads return in issue order, so RoPE inputs queued behind the ring would wait for the ring's HBM round trip (and the ring,
    //      issued behind a wait for the RoPE inputs, would start late).  Thread i owns element i of the G query heads; the last two waves also own
    //      the new key / value element d = i - 384 = i % 128 (the same cos / sin entry).  Unconditional loads from clamped indices: no branch, no wait.
    const int i0 = min((int)threadIdx.x, G * HD - 1), g0 = i0 / HD, d0 = threadIdx.x % HD, dp0 = d0 < HD / 2 ? d0 + HD / 2 : d0 - HD / 2;
    const unsigned short* qrow0 = row + (long long)(kvh * G + g0) * HD;
    const unsigned short* krow = row + (long long)(p.H + kvh) * HD;
    const unsigned short rq_x = qrow0[d0], rq_r = qrow0[dp0], r_cs = cs[d0], r_sn = sn[d0];
    const unsigned short rk_x = krow[d0], rk_r = krow[dp0], r_vn = row[(long long)(p.H + p.KVH + kvh) * HD + d0];
    asm volatile("" ::: "memory");                       // (hipcc sank two of the seven loads below the ring without this)
    __builtin_amdgcn_sched_barrier(0);
    // the first DEPTH iterations' rows go in flight next (they do not depend on the new token)
    Rows ring[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) load_row(l_lo + (long long)u * RPI + slot, ring[u]);

    // ---- phase 0: RoPE of the G query heads and of the new key (torch: round16(round16(x*cos) + rot*sin)), cache append
    auto rope_bits = [&](unsigned short xb, unsigned short xrb, unsigned short cb, unsigned short sb) {
        const float a = round16<DT>(half_bits_to_f32<DT>(xb) * half_bits_to_f32<DT>(cb));
        return round16<DT>(a + half_bits_to_f32<DT>(xrb) * half_bits_to_f32<DT>(sb));
    };
    auto rope = [&](const unsigned short* v, int d) { return rope_bits(v[d], v[d < HD / 2 ? d + HD / 2 : d - HD / 2], cs[d], sn[d]); };
    if ((int)threadIdx.x < G * HD) q_lds[g0][d0] = rope_bits(rq_x, rq_r, r_cs, r_sn) * p.scale;
    for (int i = threadIdx.x + 64 * NWV; i < G * HD; i += 64 * NWV) {          // (G = 8: the second half of the query heads)
        const int g = i / HD, d = i % HD;
        q_lds[g][d] = rope(row + (long long)(kvh * G + g) * HD, d) * p.scale;
    }
    if (owns_new && threadIdx.x >= 64 * NWV - HD) {      // the last two waves (the first ones may be busy with the q heads)
        const int d = d0;
        const float kr = rope_bits(rk_x, rk_r, r_cs, r_sn);
        const unsigned short vn = r_vn;
        kn_lds[d] = kr;
        vn_lds[d] = half_bits_to_f32<DT>(vn);
        kbase[pos * HD + d] = (unsigned short)f32_to_half_bits<DT>(kr);
        vbase[pos * HD + d] = vn;
        if (d == 0 && kvh == 0) p.valid[(long long)t * p.Lc + pos] = 1;
    }
    __syncthreads();
    float q[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) q[g][e] = q_lds[g][8 * d8 + e];
    float m[G], s[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -1e30f; s[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    }
    auto score_row = [&](const float (&kf)[8], const float (&vf)[8], bool valid_row) {
        float scv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float sc = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sc += q[g][e] * kf[e];
            scv[g] = row16_sum(sc);                                                  // over the 16 lanes of this row (DPP)
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float sc = scv[g];
            if (valid_row) {                                                         // uniform within the 16 lanes
                const float mn = fmaxf(m[g], sc), f = __expf(m[g] - mn), pw = __expf(sc - mn);
                s[g] = s[g] * f + pw;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] = acc[g][e] * f + pw * vf[e];
                m[g] = mn;
            }
        }
    };
    const long long niter = l_hi > l_lo ? (l_hi - l_lo + RPI - 1) / RPI : 0;   // uniform over the block
    for (long long it0 = 0; it0 < niter; it0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const long long l = l_lo + (it0 + u) * RPI + slot;
            float kf[8], vf[8];
            bool use = ring[u].in && ring[u].vb != 0;
            if (l == pos && owns_new) {
                use = true;
#pragma unroll
                for (int e = 0; e < 8; ++e) { kf[e] = kn_lds[8 * d8 + e]; vf[e] = vn_lds[8 * d8 + e]; }
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    kf[2 * d] = half_bits_to_f32<DT>(ring[u].kk[d] & 0xffffu); kf[2 * d + 1] = half_bits_to_f32<DT>(ring[u].kk[d] >> 16);
                    vf[2 * d] = half_bits_to_f32<DT>(ring[u].vv[d] & 0xffffu); vf[2 * d + 1] = half_bits_to_f32<DT>(ring[u].vv[d] >> 16);
                }
            }
            score_row(kf, vf, use);                                         // rows past `pos` carry ok = false
            load_row(l + (long long)DEPTH * RPI, ring[u]);                   // refill this slot DEPTH iterations ahead
        }
    }
    // merge the 4 row slots of the wave (lanes l, l^16, l^32 hold the same dims of different rows)
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float mo = __shfl_xor(m[g], o, 64), so = __shfl_xor(s[g], o, 64);
            const float mn = fmaxf(m[g], mo), f1 = __expf(m[g] - mn), f2 = __expf(mo - mn);
            s[g] = s[g] * f1 + so * f2;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[g][e] = acc[g][e] * f1 + __shfl_xor(acc[g][e], o, 64) * f2;
            m[g] = mn;
        }
    }
    if (kq == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (d8 == 0) { m_lds[wave][g] = m[g]; s_lds[wave][g] = s[g]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) a_lds[wave][g][8 * d8 + e] = acc[g][e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < G * HD; i += 64 * NWV) {
        const int g = i / HD, d = i % HD;
        float mm = -1e30f;
#pragma unroll
        for (int j = 0; j < NWV; ++j) mm = fmaxf(mm, m_lds[j][g]);
        float ssum = 0.f, a = 0.f;
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const float f = __expf(m_lds[j][g] - mm);
            ssum += s_lds[j][g] * f;
            a += a_lds[j][g][d] * f;
        }
        if (p.nsplit == 1) {
            p.out[(long long)t * p.s_out + (long long)(kvh * G + g) * HD + d] = (unsigned short)f32_to_half_bits<DT>(a / ssum);
        } else {                                         // partial of this split: un-normalised accumulator, running max, sum
            float* w = p.ws + (((long long)blockIdx.x * p.nsplit + blockIdx.y) * G + g) * (HD + 2);
            attn_ws_store(&w[d], a);
            if (d == 0) { attn_ws_store(&w[HD], mm); attn_ws_store(&w[HD + 1], ssum); }
        }
    }
    if (p.nsplit == 1) return;
    // In-launch merge of the splits (same protocol as gemv_ticket_reduce, bd_gemv.h): the partials and the ticket are only ever
    // touched by agent-scope relaxed atomics (they go to the coherence point; the XCDs' L2s are not coherent for plain stores), the
    // order is explicit -- every partial store of this block acknowledged (vmcnt(0), barrier) -> ticket increment -> the block that
    // arrives LAST reads all partials and merges them in split order (the result does not depend on which block that is) and puts
    // the ticket back to 0 for the next launch.  Contract: the ticket words are zero when the launch is enqueued.
    __shared__ int s_last;
    __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0)
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(&p.tickets[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(p.nsplit - 1);
    __syncthreads();
    if (!s_last) return;
    for (int i = threadIdx.x; i < G * HD; i += 64 * NWV) {
        const int g = i / HD, d = i % HD;
        const float* w = p.ws + ((long long)blockIdx.x * p.nsplit * G + g) * (HD + 2);
        // every partial of this element is fetched FIRST (3 loads per split, all in flight together), then merged in split order: written as
        // two loops over a run-time split count, each relaxed-atomic load was waited for where it was issued -- up to 8 dependent L2 round
        // trips on the critical path of the last block (the launch is latency, not bandwidth: profiles/r05_decode_step.txt)
        float mv[ATTN_MAX_SPLITS], sv[ATTN_MAX_SPLITS], av[ATTN_MAX_SPLITS];
#pragma unroll
        for (int c = 0; c < ATTN_MAX_SPLITS; ++c) {
            const float* wc = w + (long long)min(c, p.nsplit - 1) * G * (HD + 2);
            mv[c] = attn_ws_load(&wc[HD]);
            sv[c] = attn_ws_load(&wc[HD + 1]);
            av[c] = attn_ws_load(&wc[d]);
        }
        float mm = -1e30f;
#pragma unroll
        for (int c = 0; c < ATTN_MAX_SPLITS; ++c) if (c < p.nsplit) mm = fmaxf(mm, mv[c]);
        float ssum = 0.f, a = 0.f;
#pragma unroll
        for (int c = 0; c < ATTN_MAX_SPLITS; ++c) {
            if (c < p.nsplit) {
                const float f = __expf(mv[c] - mm);
                ssum += sv[c] * f;
                a += av[c] * f;
            }
        }
        p.out[(long long)t * p.s_out + (long long)(kvh * G + g) * HD + d] = (unsigned short)f32_to_half_bits<DT>(a / ssum);
    }
    if (threadIdx.x == 0) __hip_atomic_store(&p.tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// cache_warm_kernel (round 6): reads two byte ranges and throws the values away -- a weight-prefetch node for a hipGraph SIDE BRANCH.  The decode
// step's attention launch is a latency-bound chain that leaves HBM idle for ~10 us; the o projection that follows streams 46 MB it has never
// touched.  Forked before the attention launch and joined before the o projection, this kernel pulls those bytes through HBM into the 256-MB
// Infinity Cache (memory-side, shared by all XCDs), default cache policy, 8 x 16-byte loads in flight per thread.  No dependency to respect: the
// weights are static.  Changes no arithmetic anywhere.
__global__ void __launch_bounds__(256) cache_warm_kernel(const u32x4_t* __restrict__ p0, long long n0, const u32x4_t* __restrict__ p1, long long n1) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nth = (long long)gridDim.x * 256;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    for (int r = 0; r < 2; ++r) {
        const u32x4_t* p = r ? p1 : p0;
        const long long n = r ? n1 : n0;
        long long i = tid;
        for (; i + 7 * nth < n; i += 8 * nth) {
            u32x4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[i + u * nth];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
        for (; i < n; i += nth) acc ^= p[i];
    }
    asm volatile("" :: "v"(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]));      // keeps the loads alive; nothing is stored
}

}  // namespace bd
