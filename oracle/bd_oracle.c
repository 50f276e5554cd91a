/*
 * bd_oracle.c -- CPU restatement of the BitDelta 1-bit-delta Linear hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bitdelta_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * Parity pinning: this restatement is checked (tests/test_oracle_golden.py)
 * against golden vectors produced by importing the reference itself in the
 * authoring container (tests/golden/make_golden.py): pack/unpack outputs,
 * BinaryDiff.__init__ buffers, the Triton kernel body run under
 * TRITON_INTERPRET=1 (fp16), BinaryDiff.forward and the load_diff merge.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference checkout, e.g. bitdelta/binary_gemm_kernel.py:6-32).
 *
 * Conventions: all strides are in ELEMENTS.  dtype codes: 0 = fp16, 1 = bf16,
 * 2 = fp32.  16-bit floats are handled as raw uint16 bit patterns with
 * software round-to-nearest-even conversions, so results do not depend on the
 * host's fp16/bf16 hardware support.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>

#define BDO_F16 0
#define BDO_BF16 1
#define BDO_F32 2

/* ------------------------------------------------------------------ */
/* scalar conversions                                                  */
/* ------------------------------------------------------------------ */
static inline float u32_as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

float bdo_bf16_to_f32(uint16_t h) { return u32_as_f32((uint32_t)h << 16); }

/* round-to-nearest-even, NaN kept quiet (matches torch's float->bfloat16) */
uint16_t bdo_f32_to_bf16(float f) {
    uint32_t u = f32_as_u32(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (uint16_t)(u >> 16);
}

float bdo_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return u32_as_f32(sign);
        /* subnormal: man * 2^-24 */
        float v = (float)man * (1.0f / 16777216.0f);
        return (sign ? -v : v);
    }
    if (exp == 31) return u32_as_f32(sign | 0x7f800000u | (man << 13));
    return u32_as_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

/* IEEE binary16 round-to-nearest-even with overflow to inf and gradual underflow */
uint16_t bdo_f32_to_f16(float f) {
    uint32_t u = f32_as_u32(f);
    uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);          /* NaN */
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);         /* >= 65520 -> inf */
    if (a < 0x33000001u) return sign;                                 /* <= 2^-25 -> 0 (tie to even) */
    int32_t e = (int32_t)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t half;
    if (e < -14) {                    /* subnormal half */
        shift = 13 + (-14 - e);       /* 14..24 */
        uint32_t q = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) q++;
        return (uint16_t)(sign | q);  /* q may carry into the normal range: correct encoding */
    }
    shift = 13;
    uint32_t q = ((uint32_t)(e + 15) << 10) | ((m >> shift) & 0x3ffu);
    uint32_t rem = m & 0x1fffu;
    half = 0x1000u;
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | q);
}

static inline float load_f(const void *p, int64_t i, int dtype) {
    if (dtype == BDO_F16) return bdo_f16_to_f32(((const uint16_t *)p)[i]);
    if (dtype == BDO_BF16) return bdo_bf16_to_f32(((const uint16_t *)p)[i]);
    return ((const float *)p)[i];
}
static inline float round_f(float v, int dtype) {
    if (dtype == BDO_F16) return bdo_f16_to_f32(bdo_f32_to_f16(v));
    if (dtype == BDO_BF16) return bdo_bf16_to_f32(bdo_f32_to_bf16(v));
    return v;
}
static inline void store_f(void *p, int64_t i, float v, int dtype) {
    if (dtype == BDO_F16) ((uint16_t *)p)[i] = bdo_f32_to_f16(v);
    else if (dtype == BDO_BF16) ((uint16_t *)p)[i] = bdo_f32_to_bf16(v);
    else ((float *)p)[i] = v;
}

/* ------------------------------------------------------------------ */
/* pack / unpack      bitdelta/binary_gemm_kernel.py:6-32 / :34-46     */
/* ------------------------------------------------------------------ */
/*
 * pack: word[b,i,n] = sum_{j<n_bits} bit[b, n_bits*i + j, n] << j   (:16-19)
 * The reference sums int64 and casts to uint8/int16/int32/int64 (:23-32); the
 * cast wraps, so bit 31 of a 32-bit word makes it negative.  Storing the
 * two's-complement low n_bits is exactly that.  K % n_bits must be 0 (:13).
 * bits: uint8 0/1 (torch.bool storage), arbitrary element strides (diff.py:16
 * packs a transposed view).  out: contiguous [batch, K/n_bits, N].
 * Returns 0, or -1 for K % n_bits != 0, -2 for unsupported n_bits.
 */
int bdo_pack(const uint8_t *bits, int64_t batch, int64_t K, int64_t N,
             int64_t s_b, int64_t s_k, int64_t s_n, void *out, int n_bits) {
    if (n_bits != 8 && n_bits != 16 && n_bits != 32 && n_bits != 64) return -2;
    if (K % n_bits) return -1;
    int64_t KW = K / n_bits;
    for (int64_t b = 0; b < batch; ++b)
        for (int64_t i = 0; i < KW; ++i)
            for (int64_t n = 0; n < N; ++n) {
                uint64_t w = 0;
                for (int j = 0; j < n_bits; ++j)
                    w |= (uint64_t)(bits[b * s_b + (i * n_bits + j) * s_k + n * s_n] ? 1u : 0u) << j;
                int64_t o = (b * KW + i) * N + n;
                if (n_bits == 8) ((uint8_t *)out)[o] = (uint8_t)w;
                else if (n_bits == 16) ((uint16_t *)out)[o] = (uint16_t)w;
                else if (n_bits == 32) ((uint32_t *)out)[o] = (uint32_t)w;
                else ((uint64_t *)out)[o] = w;
            }
    return 0;
}

/*
 * unpack: bit[b, n_bits*i + j, n] = (word[b,i,n] >> j) & 1   (:41-45)
 * (arithmetic shift on signed words; the & 1 makes the sign fill irrelevant.)
 * words: contiguous [batch, KW, N]; out: contiguous uint8 [batch, KW*n_bits, N].
 */
int bdo_unpack(const void *words, int64_t batch, int64_t KW, int64_t N, int n_bits, uint8_t *out) {
    if (n_bits != 8 && n_bits != 16 && n_bits != 32 && n_bits != 64) return -2;
    for (int64_t b = 0; b < batch; ++b)
        for (int64_t i = 0; i < KW; ++i)
            for (int64_t n = 0; n < N; ++n) {
                int64_t o = (b * KW + i) * N + n;
                uint64_t w;
                if (n_bits == 8) w = ((const uint8_t *)words)[o];
                else if (n_bits == 16) w = ((const uint16_t *)words)[o];
                else if (n_bits == 32) w = ((const uint32_t *)words)[o];
                else w = ((const uint64_t *)words)[o];
                for (int j = 0; j < n_bits; ++j)
                    out[(b * KW * n_bits + i * n_bits + j) * N + n] = (uint8_t)((w >> j) & 1u);
            }
    return 0;
}

/* ------------------------------------------------------------------ */
/* binary GEMM        bitdelta/binary_gemm_kernel.py:48-184, :186-335  */
/* ------------------------------------------------------------------ */
/*
 * One activation row against every column of one mask:  acc[n] = sum_k (bit[k,n] ? x[k] : -x[k]),  k ASCENDING for every n.
 * S[k,n] = 2*bit - 1 with bit = (word[k/32,n] >> (k%32)) & 1   (:109-111, :128-129, :270-272).
 * acc_mode 0: fp32 accumulate (one legal order of tl.dot's fp32 accumulate, :118/:134); 1: double accumulate (exact-sum reference).
 * The loop nest is k outer / n inner so the packed words are read as contiguous rows; per column the additions happen in the
 * same k order as a column-at-a-time dot product, so the result is bit-identical to it.
 */
static void delta_row(const float *x, const int32_t *P, int64_t N, int64_t K, int acc_mode, double *acc_d, float *acc_f) {
    if (acc_mode == 0) {
        for (int64_t n = 0; n < N; ++n) acc_f[n] = 0.0f;
        for (int64_t k = 0; k < K; ++k) {
            const uint32_t *row = (const uint32_t *)P + (k >> 5) * N;
            const int sh = (int)(k & 31);
            const float a = x[k], na = -a;
            for (int64_t n = 0; n < N; ++n) acc_f[n] += ((row[n] >> sh) & 1u) ? a : na;
        }
        for (int64_t n = 0; n < N; ++n) acc_d[n] = (double)acc_f[n];
        return;
    }
    for (int64_t n = 0; n < N; ++n) acc_d[n] = 0.0;
    for (int64_t k = 0; k < K; ++k) {
        const uint32_t *row = (const uint32_t *)P + (k >> 5) * N;
        const int sh = (int)(k & 31);
        const double a = (double)x[k], na = -a;
        for (int64_t n = 0; n < N; ++n) acc_d[n] += ((row[n] >> sh) & 1u) ? a : na;
    }
}

/*
 * C[b] = A[b] . (2*unpack(P[b]) - 1)
 *   A [B,M,K] in dtype_in (k contiguous), P [B or 1, K/32, N] int32 (sPb = 0 broadcasts one mask,
 *   which is what diff.py:38's mask.repeat materialises), C [B,M,N] in dtype_out (n contiguous).
 * round_mode 0: C = round_out(acc_fp32)
 * round_mode 1: C = round_out(fp16(acc_fp32))  -- the reference epilogue `accumulator.to(tl.float16)`
 *               (:143, :287) followed by the store-cast into a tensor of a.dtype (:167, :314).
 * acc_mode 0: fp32 accumulate k-ascending; 1: double accumulate then one rounding to fp32.
 * Edge semantics: arbitrary M, N; K % 32 == 0 required (pack :13); `activation` is accepted and
 * ignored by the reference (:73, :141-142) so it has no parameter here.
 * Rows are independent: the (b, m) loop is an OpenMP parallel loop (no reduction crosses threads).
 */
int bdo_delta_bmm(const void *A, const int32_t *P, void *C, int64_t B, int64_t M, int64_t N, int64_t K,
                  int64_t sAb, int64_t sAm, int64_t sPb, int64_t sCb, int64_t sCm,
                  int dtype_in, int dtype_out, int round_mode, int acc_mode) {
    if (K % 32) return -1;
    int fail = 0;
#pragma omp parallel
    {
        float *x = (float *)malloc((size_t)(K > 0 ? K : 1) * sizeof(float));
        float *af = (float *)malloc((size_t)(N > 0 ? N : 1) * sizeof(float));
        double *ad = (double *)malloc((size_t)(N > 0 ? N : 1) * sizeof(double));
        if (!x || !af || !ad) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int64_t r = 0; r < B * M; ++r) {
                const int64_t b = r / M, m = r - b * M;
                for (int64_t k = 0; k < K; ++k) x[k] = load_f(A, b * sAb + m * sAm + k, dtype_in);
                delta_row(x, P + b * sPb, N, K, acc_mode, ad, af);
                for (int64_t n = 0; n < N; ++n) {
                    float acc = (float)ad[n];
                    if (round_mode == 1) acc = bdo_f16_to_f32(bdo_f32_to_f16(acc));
                    store_f(C, b * sCb + m * sCm + n, acc, dtype_out);
                }
            }
        }
        free(x); free(af); free(ad);
    }
    return fail ? -9 : 0;
}

/* ------------------------------------------------------------------ */
/* fused Linear       bitdelta/diff.py:33-39, demo/demo_backend.py:93-98 */
/* ------------------------------------------------------------------ */
/*
 * y[b,m,n] = sum_k x[b,m,k] W[n,k]  +  alpha[b*G + g(n)] * sum_k x[b,m,k] S_b[k,n]
 *   W [N,K] row-major with leading dim ldw (BinaryDiff.base is the .T view of this, diff.py:18),
 *   P [B or 1, K/32, N] (sPb = 0 -> single-tenant BinaryDiff; sPb = K/32*N -> DiffCompressModule,
 *   demo_backend.py:133-134), alpha fp32 [B*G] (sAlb = 0 broadcasts; G groups split N evenly,
 *   G = 1 is the reference's one-scalar-per-matrix case, diff.py:12/:20-30).
 * round_mode 0: one rounding of the exact value to dtype_out.
 * round_mode 1: the reference's chain for `x @ base + coeff * binary_bmm(x, mask)` (diff.py:39):
 *     t1 = round_in(x.W)                       (GEMM output in x.dtype)
 *     t2 = round_in(fp16(acc_delta))           (binary_bmm :287 + :314)
 *     t3 = round_in(round_in(alpha) * t2)      (0-dim fp32 coeff joins a 16-bit tensor: torch casts
 *                                               the scalar tensor to the common dtype first)
 *     y  = round_in(t1 + t3)
 *   demo_backend.py:95-98 is the same chain with per-tenant fp16 coeff.
 * Both contractions accumulate in double, k ascending.  W is converted to fp32 once (exact), rows run in parallel (OpenMP).
 */
int bdo_binary_linear(const void *X, const void *W, const int32_t *P, const float *alpha, void *Y,
                      int64_t B, int64_t M, int64_t N, int64_t K,
                      int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int G,
                      int64_t sYb, int64_t sYm, int dtype_in, int dtype_out, int round_mode) {
    if (K % 32) return -1;
    if (G < 1 || N % G) return -3;
    const int64_t gsz = N / G;
    float *Wf = (float *)malloc((size_t)(N * K > 0 ? N * K : 1) * sizeof(float));
    if (!Wf) return -9;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k) Wf[n * K + k] = load_f(W, n * ldw + k, dtype_in);
    int fail = 0;
    /* few rows (decode): parallel over columns inside a row; many rows: parallel over rows */
    const int by_rows = (B * M >= 8);
#pragma omp parallel if (by_rows)
    {
        float *x = (float *)malloc((size_t)(K > 0 ? K : 1) * sizeof(float));
        float *af = (float *)malloc((size_t)(N > 0 ? N : 1) * sizeof(float));
        double *ad = (double *)malloc((size_t)(N > 0 ? N : 1) * sizeof(double));
        double *bs = (double *)malloc((size_t)(N > 0 ? N : 1) * sizeof(double));
        if (!x || !af || !ad || !bs) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int64_t r = 0; r < B * M; ++r) {
                const int64_t b = r / M, m = r - b * M;
                for (int64_t k = 0; k < K; ++k) x[k] = load_f(X, b * sXb + m * sXm + k, dtype_in);
#pragma omp parallel for schedule(static) if (!by_rows)
                for (int64_t n = 0; n < N; ++n) {
                    double base = 0.0;
                    const float *wr = Wf + n * K;
                    for (int64_t k = 0; k < K; ++k) base += (double)x[k] * (double)wr[k];
                    bs[n] = base;
                }
                delta_row(x, P + b * sPb, N, K, 1, ad, af);
                for (int64_t n = 0; n < N; ++n) {
                    const double base = bs[n], dl = ad[n];
                    const float al = alpha[b * sAlb + n / gsz];
                    float y;
                    if (round_mode == 0) {
                        y = (float)(base + (double)al * dl);
                    } else {
                        float t1 = round_f((float)base, dtype_in);
                        float t2 = round_f(bdo_f16_to_f32(bdo_f32_to_f16((float)dl)), dtype_in);
                        float t3 = round_f(round_f(al, dtype_in) * t2, dtype_in);
                        y = round_f(t1 + t3, dtype_in);
                    }
                    store_f(Y, b * sYb + m * sYm + n, y, dtype_out);
                }
            }
        }
        free(x); free(af); free(ad); free(bs);
    }
    free(Wf);
    return fail ? -9 : 0;
}

/* ------------------------------------------------------------------ */
/* BinaryDiff.__init__          bitdelta/diff.py:9-31                  */
/* ------------------------------------------------------------------ */
/*
 * diff = finetune - base (in the weights' dtype, :11); coeff = mean(|diff|.float()) (:12);
 * bit = 0 where diff < 0 else 1 (:14-15: zero, -0.0 and NaN stay 1); mask = pack(bit.T) (:16):
 * mask[k/32, n] bit (k%32) = bit[n, k].   base/fine: [N,K] row-major, leading dim ld.
 * coeff is accumulated in double (torch's fp32 reduction order is unspecified).
 */
int bdo_binarize(const void *base, const void *fine, int64_t N, int64_t K, int64_t ld, int dtype,
                 int32_t *mask, float *coeff) {
    if (K % 32) return -1;
    double s = 0.0;
    memset(mask, 0, (size_t)(K / 32) * (size_t)N * 4);
    for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k) {
            float d = round_f(load_f(fine, n * ld + k, dtype) - load_f(base, n * ld + k, dtype), dtype);
            s += fabs((double)d);
            if (!(d < 0.0f)) ((uint32_t *)mask)[(k >> 5) * N + n] |= 1u << (k & 31);
        }
    *coeff = (float)(s / ((double)N * (double)K));
    return 0;
}

/* ------------------------------------------------------------------ */
/* load_diff merge               bitdelta/diff.py:93-95                */
/* ------------------------------------------------------------------ */
/*
 * weight = (unpack(mask)*2-1) * coeff            int64 * fp32 0-dim -> fp32, exactly +-coeff
 * W.add_(weight.T.to(W.dtype))                   W[n,k] = round(W[n,k] + round(+-coeff))
 */
int bdo_merge_delta(void *W, int64_t ldw, const int32_t *P, float coeff, int64_t N, int64_t K, int dtype) {
    if (K % 32) return -1;
    float cp = round_f(coeff, dtype), cn = round_f(-coeff, dtype);
    for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k) {
            uint32_t w = (uint32_t)P[(k >> 5) * N + n];
            float d = ((w >> (k & 31)) & 1u) ? cp : cn;
            store_f(W, n * ldw + k, load_f(W, n * ldw + k, dtype) + d, dtype);
        }
    return 0;
}
