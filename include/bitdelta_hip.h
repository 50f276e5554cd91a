/*
 * bitdelta_hip.h -- C ABI of libbitdelta_hip.so: the MI355X (gfx950) implementation of BitDelta's
 * 1-bit-delta Linear hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless
 * noted; `stream` is a hipStream_t passed as void* (NULL = default stream).  All entry points are
 * re-entrant (the reference's serving threads call without locks, demo/demo_backend.py:261): the only state is
 * (a) the bd_set_* test / tuning overrides, which are THREAD-LOCAL (a thread that forces a kernel family does not
 * change what other threads launch), and (b) per-DEVICE caches (CU count, max-dynamic-LDS attribute already
 * raised) keyed by the current HIP device, so one process may drive several GPUs.
 * Return value: 0 on success, a negative BD_E_* code otherwise (bd_error_string() names it); nothing throws.
 *
 * The reference has no FFI of its own: the interface this library replaces is the Python function surface
 * of /root/reference/bitdelta/binary_gemm_kernel.py and bitdelta/diff.py (cited per function below).
 * bitdelta_amd/_lib.py is the ctypes binding; INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * dtype codes: 0 = fp16, 1 = bf16, 2 = fp32 (outputs only).  All strides are in ELEMENTS.
 */
#ifndef BITDELTA_HIP_H
#define BITDELTA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BD_F16 0
#define BD_BF16 1
#define BD_F32 2

#define BD_OK 0
#define BD_E_K_NOT_MULTIPLE (-1)   /* K % n_bits != 0  (reference assert, binary_gemm_kernel.py:13) */
#define BD_E_BAD_NBITS (-2)        /* n_bits not in {8,16,32,64} (reference: UnboundLocalError, :23-30) */
#define BD_E_BAD_GROUPS (-3)       /* G < 1 or N % G != 0 */
#define BD_E_BAD_DTYPE (-4)
#define BD_E_BAD_SHAPE (-5)        /* negative / overflowing dimension */
#define BD_E_WORKSPACE (-6)        /* workspace missing or too small */
#define BD_WS_TICKET_BYTES 65536    /* leading bytes of a GEMM workspace that must be zero on entry (see bd_delta_bmm) */
#define BD_E_LAUNCH (-7)           /* hipLaunchKernel failed (hipGetLastError has the detail) */
#define BD_E_NULL (-8)

int bd_version(void);
const char* bd_error_string(int code);

/* pack: replaces pack(x, n_bits)  -- bitdelta/binary_gemm_kernel.py:6-32.
 * bits: torch.bool bytes, logical shape [batch, K, N] with element strides (s_b, s_k, s_n) -- the reference packs a
 * transposed view (bitdelta/diff.py:16).  out: contiguous [batch, K/n_bits, N] of uint8/int16/int32/int64. */
int bd_pack(const void* bits, int64_t batch, int64_t K, int64_t N, int64_t s_b, int64_t s_k, int64_t s_n,
            void* out, int n_bits, void* stream);

/* unpack: replaces unpack(x, n_bits) -- bitdelta/binary_gemm_kernel.py:34-46.
 * words: contiguous [batch, KW, N]; out_bits: contiguous torch.bool bytes [batch, KW*n_bits, N]. */
int bd_unpack(const void* words, int64_t batch, int64_t KW, int64_t N, void* out_bits, int n_bits, void* stream);

/* delta GEMM: replaces binary_matmul / binary_bmm -- bitdelta/binary_gemm_kernel.py:153-184, :297-335 (kernels :48-151,
 * :186-295).   C[b] = A[b] . (2*unpack(P[b]) - 1)
 *   A [B,M,K] dtype (k contiguous), P int32 [B or 1, K/32, N] contiguous (sPb = 0 broadcasts one mask: what
 *   bitdelta/diff.py:38 materialises with mask.repeat), C [B,M,N] out_dtype (n contiguous).
 *   round_mode 1 = the reference epilogue fp32 -> fp16 -> out (:143/:287 then :167/:314); 0 = one rounding fp32 -> out.
 *   alpha != NULL: C = alpha[b, g(n)] * acc (accumulate = 0) or C = C_in + alpha[b, g(n)] * acc (accumulate = 1), fp32
 *   math, one rounding -- folds `coeff *` and `+` of bitdelta/diff.py:39 / demo_backend.py:97-98 into the epilogue.
 *   alpha: fp32 [B or 1, G], sAlb = its batch stride (0 = broadcast); G scale groups split N evenly.
 *   ws / ws_bytes: scratch of at least bd_gemm_workspace_bytes(B, M, N, K) bytes (may be NULL when that is 0).
 *     The first BD_WS_TICKET_BYTES bytes are reserved for the arrival counters of the decode path's optional in-launch split-k
 *     reduction (bd_set_decode_two_launch(0)).  ONLY in that mode they must be ZERO when the call is enqueued (hipMemsetAsync the
 *     buffer once after allocating it); the library leaves them zero when the call completes, and a kernel that finds a counter out of
 *     range traps (hipErrorLaunchFailure at the next synchronisation) rather than return wrong sums.  In the default mode (a second,
 *     tiny reduce launch) the workspace needs no initialisation.  Do not share a workspace between concurrently running launches. */
int bd_delta_bmm(const void* A, const int32_t* P, void* C, int B, int M, int N, int K,
                 int64_t sAb, int64_t sAm, int64_t sPb, int64_t sCb, int64_t sCm,
                 int dtype, int out_dtype, int round_mode,
                 const float* alpha, int64_t sAlb, int G, int accumulate,
                 void* ws, int64_t ws_bytes, void* stream);

/* fused 16-bit-base + 1-bit-delta Linear: replaces BinaryDiff.forward (bitdelta/diff.py:33-39) and
 * DiffCompressModule.forward (demo/demo_backend.py:93-98) in ONE launch:
 *   Y[b] = X[b] . W^T + alpha[b, g(n)] * (X[b] . S[b]),  fp32 accumulate, one rounding to out_dtype.
 *   W [N,K] row-major, leading dimension ldw (BinaryDiff.base is its .T view, diff.py:18; nn.Linear.weight as is). */
int bd_binary_linear(const void* X, const void* W, const int32_t* P, const float* alpha, void* Y,
                     int B, int M, int N, int K,
                     int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int G,
                     int64_t sYb, int64_t sYm, int dtype, int out_dtype,
                     void* ws, int64_t ws_bytes, void* stream);

/* decode form of the fused Linear with the sign words repacked for the streaming decode kernel.  The serving side repacks a
 * tenant set's masks once when it registers them (diff.pt and the nn.Module buffers keep the reference layout):
 *   mask_layout 1, TILE-MAJOR:  P int32 [B or 1, ceil(N/16), K/32, 16]: word (i, n) of the reference layout at [n / 16][i][n % 16]
 *     (a 16-column MFMA tile reads its words as one contiguous run over k); sPb = tenant stride in words (0 broadcasts); t_pad unused.
 *   mask_layout 2, PACKED:  P int32 [ceil(N/16), ceil(K/128), 4, 16, t_pad]: element [tile][it][g][c][t] is tenant t's dword whose
 *     byte s holds the 8 signs of k = 128 it + 32 s + 8 g .. + 7 of column 16 tile + c (a 4 x 4 byte transpose of the iteration's 4
 *     word rows), tenants interleaved and zero-padded to t_pad in {1, 2, 4, 6, 8}; B <= t_pad tenants, all in one call (B*M <= 16).
 *     Natural k order for every operand: one activation fragment set, sector-contiguous weight loads, 1-2 wide sign loads per stage.
 *     With this layout the BASE WEIGHT may be the serving side's tile-major decode copy as well: pass ldw = 0 and
 *     W' [N/16][K/128][4 steps s][16 rows c][4 groups g][8] with W'[tile][it][s][c][g][e] = W[16 tile + c][128 it + 32 s + 8 g + e]
 *     (M == 1, N % 16 == 0, K % 128 == 0): one stage of the kernel is then ONE contiguous 4-KiB block (gate+up 28672x4096 for 6
 *     tenants: 68.8 vs 73.0 us).  Same values, same arithmetic: bit-identical to the row-major operand.
 * Columns past N / k past K are zero padding.  Streaming decode kernel only: M <= 16, N >= 512, <= 8 masks, else BD_E_BAD_SHAPE.
 * accumulate = 1 adds onto Y (residual epilogue).  Needs no workspace. */
int bd_binary_linear_decode(const void* X, const void* W, const int32_t* P, int mask_layout, int t_pad, const float* alpha, void* Y,
                            int B, int M, int N, int K,
                            int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int G,
                            int64_t sYb, int64_t sYm, int dtype, int out_dtype, int accumulate, void* stream);

/* PREFILL-size fused gate|up projection with the MLP's activation in its epilogue (M > 16; the decode-size form is
 * bd_binary_linear_decode_fused with epilogue = 1).  W [N, K] / P [B or 1, K/32, N] (reference sign layout) / alpha [B or 1, 2] describe
 * the gate_proj and up_proj BinaryDiff modules of one MLP (the two Linears bitdelta/diff.py:60-64 selects by name `mlp.*proj`) stored
 * as ONE projection whose output rows are interleaved in blocks of 8 ([g0..7 | u0..7 | g8..15 | ...]); Y [B, M, N/2] receives
 *     round(silu(round(gate))) * round(up),   gate / up = X . W^T + alpha * (X . S)  (fp32, one rounding each)
 * i.e. exactly bd_binary_linear followed by bd_srv_swiglu (bit-identical), without the [M, N] round trip through HBM and the second
 * launch.  N % 16 == 0, K % 64 == 0, 16-byte aligned rows; anything else returns BD_E_BAD_SHAPE and the caller runs the two launches. */
int bd_binary_linear_swiglu(const void* X, const void* W, const int32_t* P, const float* alpha, void* Y,
                            int B, int M, int N, int K,
                            int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb,
                            int64_t sYb, int64_t sYm, int dtype, void* stream);

/* packed-layout decode Linear with the neighbouring glue of a decoder layer FUSED into the launch (bit-identical to the separate
 * launches; what disappears is a ~4 us kernel + a launch gap per fused op, on a step of a few hundred 15-65 us Linears):
 *   norm_w != NULL (optional when epilogue = 1): X is the UN-NORMALISED residual stream; every block computes HF RMSNorm
 *     norm_w[b] * round(X[b,m] * rsqrt(mean(X[b,m]^2) + eps))  for the B*M rows itself while its first weight stages are in flight and
 *     reads its activations from LDS (bd_srv_rmsnorm's arithmetic, same order).  norm_w [B or 1, K], stride s_norm elements.
 *     Needs M == 1 (one new token per tenant), K a power of two >= 2048, B*K <= 32768 and B*(2K+16) bytes of LDS next to the
 *     kernel's 82 KB (8 tenants x 4096, 4 x 8192).
 *   epilogue = 1: SwiGLU.  W / P / alpha describe a fused gate|up projection whose OUTPUT ROWS are interleaved in blocks of 8
 *     ([g0..7 | u0..7 | g8..15 | u8..15 | ...]; G = 2 scale groups: alpha[b][0] gate, alpha[b][1] up); Y [B, M, N/2] receives
 *     round(silu(round(gate))) * round(up)  -- the HF MLP's act_fn(gate_proj(x)) * up_proj(x) (bd_srv_swiglu's arithmetic).
 *     N % 16 == 0, out_dtype == dtype, accumulate = 0.
 * Shapes outside the envelope return BD_E_BAD_SHAPE and the caller runs the separate launches. */
int bd_binary_linear_decode_fused(const void* X, const void* W, const int32_t* P, int t_pad, const float* alpha, void* Y,
                                  int B, int M, int N, int K,
                                  int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int G,
                                  int64_t sYb, int64_t sYm, int dtype, int out_dtype, int accumulate,
                                  const void* norm_w, int64_t s_norm, float eps, int epilogue, void* stream);

/* RMSNorm HAND-OFF between two decode launches of a decoder layer (packed layout, M == 1, B <= 8; new in round 5 -- the reference runs
 * the HF RMSNorm module between its Linears, demo/demo_backend.py:62-79 wraps it per tenant).  bd_binary_linear_decode_fused with three
 * more pointers:
 *   PRODUCER, ssq_out != NULL (the o_proj / down_proj launch: accumulate = 1 or 0, 16-bit output, N % 16 == 0, epilogue = 0): besides Y the
 *     launch writes ssq_out[n / 16][row] = sum over the 16 columns of tile n / 16 of Y[row][n]^2 (fp32 [N/16][16], fixed order).  With
 *     xw_out != NULL, norm_w [B or 1, N] (stride s_norm) is the weight of the RMSNorm that FOLLOWS and xw_out [B, M, N] (Y's strides)
 *     receives round(Y (.) norm_w): the pre-multiplied copy the consumer reads instead of Y.
 *   CONSUMER, ssq_in != NULL (the q|k|v / gate|up launch that follows; K % 16 == 0, K <= 8192, norm_w = NULL): X is the producer's xw_out
 *     -- the consumer reads exactly what the resident-row form reads, nothing to multiply -- and rsqrt(sum(ssq_in[:, row]) / K + eps) scales
 *     the row's accumulators in the epilogue:  Y = rs * (W + alpha S) . (norm_w (.) x)  -- HF RMSNorm followed by the Linear with the row
 *     scale moved across the contraction (one rounding of x * norm_w instead of two; not bit-identical to the separate bd_srv_rmsnorm launch,
 *     same accuracy).
 *   Needs the tile-major base weight (ldw = 0) on the consumer, K >= 1024, B * K <= 32768.  Both stand-alone rmsnorm launches of a layer
 *   disappear without any block re-reducing or re-normalising the rows.  BD_E_BAD_SHAPE outside the envelope (no fallback).  *   PRODUCER launches (ssq_out != NULL) do not normalise anything: there the `eps` argument is the FACTOR the written sums of squares are
 *   multiplied by (<= 0: 1).  Intended use (round 6): pass norm_w = nw / s with s a power of two >= max |nw| and eps = 1 / s^2, and give the consumer
 *   eps / s^2 -- xw_out = round16(x . nw / s) then never exceeds |x| (no fp16 overflow whatever the norm weight), and the consumer's row scalar
 *   becomes s . rsqrt(mean(x^2) + eps): the same product exactly (bitdelta_amd.serving_loop.handoff_norm). */
int bd_binary_linear_decode_handoff(const void* X, const void* W, const int32_t* P, int t_pad, const float* alpha, void* Y,
                                    int B, int M, int N, int K,
                                    int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int G,
                                    int64_t sYb, int64_t sYm, int dtype, int out_dtype, int accumulate,
                                    const void* norm_w, int64_t s_norm, float eps, int epilogue,
                                    const float* ssq_in, float* ssq_out, void* xw_out, void* stream);

/* the same Linear with the residual connection folded into its epilogue:  Y[b] = Y_in[b] + X[b] . W^T + alpha * (X[b] . S[b])
 * -- the `hidden = residual + o_proj(...)` / `+ down_proj(...)` of the decoder layers that call the reference's modules.
 * Decode shapes (M <= 16, B*M <= 64): fp32 sum, one rounding.  M > 16 on the fused GEMM's fast path (K % 64 == 0, 16-byte aligned
 * rows): the Linear's output is rounded to the output type and the sum is rounded again, i.e. exactly the two roundings of the
 * separate `y = proj(x); hidden = residual + y` (bit-identical to it), one pass over [M, N] less.  Anything else returns
 * BD_E_BAD_SHAPE and the caller adds the residual itself. */
int bd_binary_linear_residual(const void* X, const void* W, const int32_t* P, const float* alpha, void* Y,
                              int B, int M, int N, int K,
                              int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int G,
                              int64_t sYb, int64_t sYm, int dtype, int out_dtype,
                              void* ws, int64_t ws_bytes, void* stream);

/* bd_binary_linear_residual at prefill sizes (M > 16) TOGETHER WITH the RMSNorm that follows it in the decoder layer (round 6):
 *   Y[b] = Y_in[b] + Linear(X[b]);   H[b] = norm_w[b] * round(Y[b] * rsqrt(mean(Y[b]^2) + eps))     (bd_srv_rmsnorm's arithmetic, tenant b's weight)
 * -- `hidden = residual + o_proj(attn); h = post_attention_layernorm(hidden)` (and down_proj + the next layer's input_layernorm) of the HF decoder
 * layers whose Linears are the reference's BinaryDiff modules (bitdelta/diff.py:38-39) and whose norms are DataParallelModule-wrapped
 * (demo/demo_backend.py:62-79).  When the dispatcher splits the Linear over k (several tenants of <= 64 rows: the o / down projections of a
 * short-prompt request) the norm rides on the reduce launch; otherwise it is one norm launch behind the Linear.  Bit-identical to
 * bd_binary_linear_residual followed by bd_srv_rmsnorm either way.  Y, H [B, M, N] dense in M (sYb = M sYm, sHb = M sHm), N % 8 == 0,
 * N <= 8192, 16-byte aligned rows; norm_w [B, N] (stride s_nw; 0 = one weight for all).  Anything else: BD_E_BAD_SHAPE. */
int bd_binary_linear_residual_norm(const void* X, const void* W, const int32_t* P, const float* alpha, void* Y,
                                   int B, int M, int N, int K,
                                   int64_t sXb, int64_t sXm, int64_t ldw, int64_t sPb, int64_t sAlb, int G,
                                   int64_t sYb, int64_t sYm, int dtype,
                                   const void* norm_w, int64_t s_nw, float eps, void* H, int64_t sHb, int64_t sHm,
                                   void* ws, int64_t ws_bytes, void* stream);

/* per-tenant dense Linear at decode: replaces the weight-swapping loop of DataParallelModule.forward
 * (demo/demo_backend.py:62-79) for nn.Linear leaves (lm_head): row block t runs through tenant t's OWN weight matrix,
 *   Y[t] = X[t] . W[t]^T,   X [T, M, K] (strides sXt, sXm), W [T, N, K] (strides sWt, ldw), Y [T, M, N] (strides sYt, sYm),
 * one launch for all tenants (every weight byte streamed once, fp32 accumulate, one rounding).  M <= 16 only (decode);
 * larger M returns BD_E_BAD_SHAPE: that is an ordinary batched GEMM, not this library's business. */
int bd_tenant_linear(const void* X, const void* W, void* Y, int T, int M, int N, int K,
                     int64_t sXt, int64_t sXm, int64_t sWt, int64_t ldw, int64_t sYt, int64_t sYm,
                     int dtype, int out_dtype, void* stream);

/* ---- decode-step glue of the multi-tenant serving loop (callers of the path, demo/demo_backend.py:190-258 + the HF decoder layer
 * between two of the reference's Linears).  Not the hot path: they exist because at decode every stock op is a launch.
 * bd_srv_rmsnorm: Y[r] = Wt[r / rows_per_tenant] * round(X[r] * rsqrt(mean(X[r]^2) + eps))   (HF RMSNorm with per-tenant weights,
 *   the DataParallelModule-wrapped norms of demo_backend.py:62-79); X, Y [rows, H] (strides sx, sy), Wt [tenants, H] (stride sw).
 * bd_srv_swiglu:  Y = round(silu(G)) * U, G / U [rows, I] (row strides sg / su; the fused gate|up output passes U = G + I) -> Y [rows, I];
 *   interleaved8 = 1: G is a [rows, 2I] projection output interleaved in blocks of 8 (see bd_binary_linear_decode_fused), U ignored.
 * bd_srv_decode_attention: one new token per tenant: RoPE of q and the new k (tables cos/sin [Lmax, 128], rotate-half sign folded
 *   into sin), append k/v at *pos to the caches [T, KVH, Lc, 128], mark valid[t, *pos], then softmax(q.K^T/sqrt(128)).V over the
 *   valid keys 0..*pos (left padding = 0 in valid [T, Lc] bytes), grouped-query (H/KVH in {1, 4}); QKV [T, (H+2*KVH)*128] is the
 *   fused q+k+v Linear's output; out [T, H*128].  `pos` is a DEVICE scalar so the step replays inside a hipGraph. */
int bd_srv_rmsnorm(const void* X, const void* Wt, void* Y, int rows, int H, int64_t sx, int64_t sy, int64_t sw,
                   int rows_per_tenant, float eps, int dtype, void* stream);
/* bd_srv_add_rmsnorm (round 6): x_out = round16(resid + round16(y32)); h_out = HF RMSNorm(x_out) with tenant t's weight -- the cast, the residual add
 * and the norm that follow a row-parallel Linear's all-reduce (bitdelta_amd/tp.py) in ONE launch, same roundings as the three ops.  resid / x_out /
 * h_out [rows, H] 16-bit, y32 [rows, H] fp32, row strides in elements; H % 8 == 0, H <= 8192.  No reference counterpart (the reference has no TP). */
int bd_srv_add_rmsnorm(const void* resid, const float* y32, const void* Wt, void* x_out, void* h_out, int rows, int H, int64_t s_r, int64_t s_y,
                       int64_t s_x, int64_t s_h, int64_t sw, int rows_per_tenant, float eps, int dtype, void* stream);
int bd_srv_swiglu(const void* G, const void* U, void* Y, int rows, int I, int64_t sg, int64_t su, int64_t sy, int interleaved8,
                  int dtype, void* stream);
/* bd_srv_rope: in-place rotary embedding of X [rows, heads*128] (row stride sx), position of row r = pos0 + r % seq, tables as for
 * bd_srv_decode_attention; rounds where `torch.addcmul(x * cos, rotate_half(x), sin)` rounds. */
int bd_srv_rope(void* X, const void* cos_t, const void* sin_t, int rows, int heads, int head_dim, int64_t sx, int seq, int pos0,
                int dtype, void* stream);
/* bd_srv_rope_kv_append (round 6): the prefill of a request from position pos0 -- bd_srv_rope on the q and k heads of the fused q|k|v projection
 * output QKV [T * S, (H + 2 KVH) * 128] (row stride sx; in place: bd_srv_prefill_attention reads q / k / v from it), and in the same launch the
 * rotated k rows and the v rows are written to the caches [T, KVH, Lc, 128] at positions pos0 .. pos0 + S - 1 (the `past_key_values` update of
 * the HF attention module the reference's Linears sit in; demo/demo_backend.py:297-315 prefills through it).  pos0 + S <= Lc. */
int bd_srv_rope_kv_append(void* QKV, const void* cos_t, const void* sin_t, void* kcache, void* vcache, int T, int S, int H, int KVH,
                          int head_dim, int64_t sx, int Lc, int pos0, int dtype, void* stream);
/* bd_srv_step_begin / bd_srv_step_end (round 6): the two ends of ONE greedy decode step of the serving loop (demo/demo_backend.py:190-258: HF's
 * generate loop over the tenant batch -- embedding lookup of the last tokens, attention-mask extension, argmax of the last logits, stop-token check).
 * step_begin: X[t] = embed[t][tok[t]] (embed [T, V, H] 16-bit, tenant stride sEt -- 0 = one shared table -- row stride sEv; X [T, H], row stride sx)
 *   and valid[t, *pos] = 1 (valid [T, Lc] bytes: the key mask bd_srv_decode_attention reads).
 * step_end: nxt[t] = argmax(logits[t, :V]) with torch.argmax's order (a NaN is the maximum, ties go to the lower index); tok[t] = nxt[t];
 *   out[t, *step] = nxt[t] when *step < out_cap (out [T, >= out_cap] int64, row stride s_out); stopped[t] |= nxt[t] in stop_ids[t, :ns]; then
 *   *pos += 1 and *step += 1, once (the last block to finish; `ticket` = a zero-initialised 4-byte device word the call leaves at zero).
 * tok / pos / step / out / stop_ids are int64 (torch.long), valid / stopped bytes (torch.bool): the state the loop keeps on the device so that the
 * step replays inside a hipGraph.  V % 8 == 0, H % 8 == 0.  Integer work: exact. */
int bd_srv_step_begin(const void* embed, int64_t sEt, int64_t sEv, const int64_t* tok, void* X, int64_t sx, void* valid, int Lc,
                      const int64_t* pos, int T, int V, int H, void* stream);
int bd_srv_step_end(const void* logits, int64_t sl, int V, int64_t* tok, int64_t* out, int64_t s_out, int out_cap,
                    const int64_t* stop_ids, int ns, void* stopped, int64_t* pos, int64_t* step, void* ticket, int T, int dtype, void* stream);
/* bd_srv_cache_warm (round 6): reads [p0, p0 + bytes0) and [p1, p1 + bytes1) (16-byte aligned; whole 16-byte chunks) and discards the values:
 * a weight-prefetch launch for a hipGraph side branch (the serving loop forks it next to the decode attention launch so that the o projection's
 * weight and sign words sit in the Infinity Cache when it starts).  blocks = 0: one 256-thread block per CU.  No reference counterpart (the
 * reference's decode step is demo/demo_backend.py:93-98 under HF's eager loop); changes no arithmetic. */
int bd_srv_cache_warm(const void* p0, int64_t bytes0, const void* p1, int64_t bytes1, int blocks, void* stream);
int bd_srv_decode_attention(const void* QKV, const void* cos_t, const void* sin_t, void* kcache, void* vcache, void* valid,
                            const int64_t* pos, void* out, int T, int H, int KVH, int head_dim, int Lc,
                            int64_t s_qkv, int64_t s_out, int dtype, void* ws, int64_t ws_bytes, void* stream);
/* bd_srv_prefill_attention: attention of a whole prompt (the prefill call of the serving loop, demo/demo_backend.py:262-275 -> the HF
 *   decoder layer's attention with the left-padded attention_mask; what F.scaled_dot_product_attention computes there), flash-style:
 *   O[b, s, h] = softmax_k(Q[b, s, h] . K[b, k, h / (H/KVH)] * scale) . V[b, k, h / (H/KVH)] over the keys kv_start[b] <= k (<= s when
 *   causal), fp32 online softmax, 16-bit in / out.  Q / K / V / O are [B, S, heads, 128] through their batch and sequence strides in
 *   elements (head h at + 128 h): the three slices of a fused q|k|v projection output are passed as they lie.  kv_start [B] int32 on
 *   the device (left padding) or NULL; query rows with no valid key return 0.  head_dim == 128, S % 64 == 0, strides % 8 == 0,
 *   16-byte aligned pointers; anything else returns BD_E_BAD_SHAPE and the caller keeps its own attention. */
int bd_srv_prefill_attention(const void* Q, const void* K, const void* V, void* O, int B, int S, int H, int KVH, int head_dim,
                             int64_t sqb, int64_t sqs, int64_t skb, int64_t sks, int64_t svb, int64_t svs, int64_t sob, int64_t sos,
                             const int32_t* kv_start, float scale, int causal, int dtype, void* stream);
/* scratch for bd_srv_decode_attention's split of the key range over 4 blocks per (tenant, kv head) (up to 16 for one or two sequences of a
 * grouped-query model; the size returned covers the largest split count and depends on the geometry only), merged inside the launch by the
 * block that finishes last (0 = the cache is short enough to run unsplit; ws may then be NULL).  Without a workspace the kernel
 * runs unsplit.  CONTRACT: the first 16 KiB of ws (arrival counters) are ZERO when the launch is enqueued; the kernel puts them
 * back to zero, so a buffer that is zero-filled once and used by one stream at a time can be reused by every call. */
int64_t bd_srv_decode_attention_workspace_bytes(int T, int H, int KVH, int head_dim, int Lc);

/* bytes of scratch bd_delta_bmm / bd_binary_linear may need for this problem (split-k partials of the decode path) */
int64_t bd_gemm_workspace_bytes(int B, int M, int N, int K);

/* BinaryDiff.__init__ buffers in one pass -- bitdelta/diff.py:9-31: mask = pack((fine - base >= 0).T) int32 [K/32, N],
 * coeff = mean|fine - base| (fp32, device scalar).  base, fine: [N,K] row-major, leading dimension ld. */
int bd_binarize(const void* base, const void* fine, int64_t N, int64_t K, int64_t ld, int dtype,
                int32_t* mask, float* coeff, void* ws, int64_t ws_bytes, void* stream);
int64_t bd_binarize_workspace_bytes(int64_t N, int64_t K);

/* load_diff's merge line -- bitdelta/diff.py:93-95: W[n,k] = round(W[n,k] + round(+-coeff)); coeff: device fp32 scalar. */
int bd_merge_delta(void* W, int64_t ldw, const int32_t* P, const float* coeff, int64_t N, int64_t K, int dtype,
                   void* stream);

/* The tuning / A-B hooks (bd_set_*, bd_last_*) are declared in bitdelta_hip_test.h: thread-local, not part of the stable interface a
 * reference-side binding needs (INTEGRATION.md binds none of them), free to change between versions. */

#ifdef __cplusplus
}
#endif
#endif
