/* bitdelta_hip_test.h -- TUNING / TEST HOOKS of libbitdelta_hip.so (moved out of bitdelta_hip.h in round 6).
 *
 * Everything here is thread-local state of the calling thread, exists for same-process A/B measurements and for tests that pin a kernel
 * family, and is NOT part of the drop-in boundary: no reference interface corresponds to any of it (the reference's only tunable is
 * Triton's autotune cache, bitdelta/binary_gemm_kernel.py:48-60 / :186-198, replaced here by a static shape -> variant table).  Production
 * callers never include this header. */
#ifndef BITDELTA_HIP_TEST_H
#define BITDELTA_HIP_TEST_H
#include "bitdelta_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* tuning / test hook (thread-local): force a kernel family for bd_delta_bmm / bd_binary_linear.
 * -1 auto (default); 0..3 MFMA tile configs (256x256 ping-pong, 128x256, 64x256, 32x256); 4 = 256x256 single-barrier schedule;
 * 5 = 256x128 ping-pong (picked automatically when it fills the CUs better); 6 / 7 = the half-tile ping-pong schedule at
 * 256x256 / 256x128 (A/B reference for the shipped full-tile schedule); 8 = one-pass fused 256x128 kernel with two accumulator
 * sets (bd_binary_linear only; the automatic choice for M > 128; 0 / 5 remain as the two-loop A/B references); 9 = the same kernel
 * with a 128x128 tile (picked when 256x128 tiles cannot fill the CUs); 10 = the 128x128 kernel with split-k over blockIdx.y and a
 * reduce launch (automatic for 16 < M <= 512 when the tiles would leave more than half the CUs idle; needs the workspace);
 * 100 generic edge kernel; 200 decode path (200 + KS forces a k-split), which picks between 600 (+ columns-per-block / 4) = the
 * streaming kernel (one launch, one 8-wave block per CU, the automatic choice for N >= 512 and <= 8 masks per chunk),
 * 300 (+ KS) = the VALU sign-flip kernel, 400 (+ KS) = the MFMA + sign-LUT kernel and 500 (+ KS) = the no-split-k kernel
 * (16 columns x all of k per block).  Variants 4 / 6 / 7 and fused 0 / 5 (rejected schedules kept as A/B references) exist
 * only in builds with -DBD_AB_VARIANTS (tests/native/bd_harness); the shipped library answers BD_E_BAD_SHAPE for them.
 * 13 / 14 = the FOUR-WAVE PERSISTENT kernels (bd_gemm_w4.h: one wave per SIMD, 16 AGPR accumulators, grid = min(tiles, CUs)):
 * 13 delta-only on 256x256 tiles (automatic once those tiles fill >= 80 % of the CU-rounds), 14 fused on 256x128 tiles (automatic
 * wherever 8 was; with fp32 output only its general-form epilogue); 15 = 14 with the SwiGLU epilogue (bd_binary_linear_swiglu only).
 * 16 / 17 = 8-wave PAIR tiles (two batch entries of <= 64 rows per 128x128 tile; 17 + split-k); 18 / 19 = the same on the four-wave schedule
 * (automatic since round 5); 20 = four-wave fused 128x128 tile, one entry per tile (automatic wherever 9 was, 16-bit outputs); 21 = 18 with the
 * SwiGLU epilogue (bd_binary_linear_swiglu with several entries of <= 64 rows; round 6).
 * 800 = delta_rows_kernel (bd_gemv_rows.h): delta only, reference sign layout, M <= 16, no scale -- the reference's published binary_bmm /
 * binary_matmul decode shapes: 64- or 32-column super-tiles x (1 / 2 masks per block x M rows each, or one mask shared by all B * M <= 16 rows),
 * the whole batch in one launch (automatic when those blocks occupy at least half of the chip: per-entry masks from 4 rows on, a shared mask
 * always; 801 / 802 / 804 force the masks per block).  Needs N % 32 == 0, K % 128 == 0.
 * A forced variant whose preconditions fail returns BD_E_BAD_SHAPE instead of silently falling back.
 * Environment: BD_GEMM_VARIANT=<n> is every thread's initial forced variant (overridden by this call), BD_TAIL_SPLIT=0 disables the tail split.
 * Further A/B hooks read once per thread from the environment (none is needed in production): BD_ROWS_TUNE (delta_rows_kernel: bits 1-2 masks per
 * block, bit 4 never automatic, bit 5 / 6 force 64- / 32-column super-tiles), BD_ROWS_SHARED_MIN (rows from which a shared-mask launch takes it),
 * BD_ATTN_DEPTH (decode attention K / V ring: 2 / 4), BD_ATTN_SPLITS_MAX (key-range splits: 4 = fixed, 16 = by tenants x kv heads),
 * BD_NORM_ROWS_MIN (rows from which bd_srv_rmsnorm runs its wave-per-row kernel), BD_PAIR_SPLITK (k slices of the split pair tiles, variant 19:
 * 0 = the rule; read once per process).
 * The bd_set_* entry points below are TUNING / TEST HOOKS: thread-local, not part of the stable interface a reference-side binding
 * needs (INTEGRATION.md binds none of them), and free to change between versions. */
int bd_set_gemm_variant(int variant);
/* which family the LAST call on this thread dispatched to (same codes as above) */
int bd_last_gemm_variant(void);
/* tuning hook: tile walk order of the MFMA tile kernels -- groups of `group_m` tile rows, m fastest inside a group, then n
 * (1 = n fastest, >= tiles_m = m fastest, 0 = automatic).  Results do not depend on it. */
int bd_set_tile_group_m(int group_m);
/* A/B hook: 1 = problems of more than one tile per CU are issued as consecutive single-round launches; 0 (default) = one launch */
int bd_set_launch_chunking(int on);
/* A/B hook: 1 (default) = a fused launch whose last round of 256x128 tiles would be mostly empty hands the tile columns of that round to
 * the 8-wave kernel on 128x128 tiles (a second launch on the same stream; same results); 0 = always one launch */
int bd_set_tail_split(int on);
/* 1 (default) = the decode path sums its split-k partials with a second launch (gemv_reduce_kernel); 0 = in-launch ticket
 * reduction (single launch; measured equal within noise, and it needs the zeroed ticket area described at bd_delta_bmm) */
int bd_set_decode_two_launch(int on);
/* A/B hook: 1 (default) = fused launches of the VALU decode kernel run wave-specialised (4 weight-streaming + 4 sign waves per block) */
int bd_set_decode_wave_spec(int on);
/* A/B hook of the streaming decode kernel.  Bits 0-3 are effective only in -DBD_AB_VARIANTS builds (the shipped library ignores them):
 * bit 0 = natural-order base-weight loads, bit 1 = nt cache policy, bit 2 = 8-wave blocks, bit 3 = deeper prefetch.
 * Bits 4 / 5 work in the shipped library: 16 = non-temporal policy ON for the tile-major base-weight loads of the packed-layout kernels,
 * 32 = OFF (neither: the library default); 64 = activation rows resident in LDS + deeper weight prefetch (tile-major weight, M = 1,
 * K >= 1024, B * K <= 32768) ON wherever it applies, 128 = OFF (neither: the library's shape rule); 1024 = the residual of an
 * `accumulate` launch is read in the epilogue (as before round 4) instead of at kernel start. */
int bd_set_stream_tuning(int flags);
/* ... bits 8 / 9 (round 6): 256 = never run the FINE-GRID form of the resident-row decode launches (single-tile blocks, two per CU, nibble
 * sign table), 512 = run it on every eligible launch of at most two 16-column tiles per CU (default: only between one and two tiles per CU). */
/* ... bits 13 / 14 (round 6; -DBD_AB_VARIANTS harness builds only, the shipped library ignores them): 8192 = run the TWO-PASS resident-row form where
 * the rows do not fit LDS at once (the multi-tenant down projection; measured slower, profiles/r06_decode_step.txt), 16384 = with 2 stages of prefetch;
 * bits 11 / 12: bd_tenant_linear's weight loads in natural order with (2048) / without (4096) the non-temporal policy (A/B: no difference). */
/* which form the LAST streaming decode launch of this thread took: 0 = one block per CU, 1 = fine grid, 2 = two-pass resident rows (harness builds) */
int bd_last_decode_form(void);
/* A/B hook, sign LUT of the no-split-k decode kernel: -1 (default) automatic, 1 = single 4-KiB table, 0 = 16-copy conflict-free
 * 64-KiB table whenever it fits in LDS */
int bd_set_decode_small_lut(int mode);
/* A/B hook: 1 = the no-split-k decode kernel always runs its generic one-iteration-ahead loop (default 0: delta-only K = 4096
 * launches use the straight-line instantiation: iterations 0-1 in flight during activation staging, 2-3 issued after the barrier) */
int bd_set_decode_generic_loop(int on);
/* A/B hook, effective only in -DBD_AB_VARIANTS builds (the native harness; the shipped library ignores it and answers BD_E_BAD_SHAPE
 * to bd_set_gemm_variant(700)): which kernel serves the packed-layout decode launches (mask_layout 2) of the calling thread:
 * -1 / 0 = streaming register-load kernel (gemv_stream_kernel, variant 600), 1 = LDS-DMA loader / consumer kernel (gemv_ring_kernel,
 * variant 700, tests/native/ab/bd_gemv_ring.h -- measured 26-45 % slower in round 4, profiles/r04_decode_ring_ab.txt). */
int bd_set_decode_engine(int engine);
/* Knobs of variant 700 (harness builds only; -1 = defaults): bit 0 = non-temporal policy on the weight / sign streams, bit 1 = the activation rows ride the
 * ring even when a resident LDS copy would fit, bit 2 = one loader wave instead of two, bit 3 = 4-copy LDS sign table instead of VALU expansion, bit 4 = per-block
 * rotation of the k walk, bits 8..13 = cap on the number of ring slots
 * (0 = as many as fit). */
int bd_set_ring_tuning(int flags);

#ifdef __cplusplus
}
#endif
#endif
