"""Multi-tenant greedy decoding without HF: SURVEY.md section 8(f) row 3, the caller of the batched 1-bit-delta Linear.

What the reference's demo does around the hot path (demo/demo_backend.py), restated for one process and one GPU:
  * one 16-bit base model + T fine-tunes, each a set of 1-bit deltas for the `*proj*` Linears plus its OWN dense embedding, norms
    and lm_head (register_diff_compress, :107-153); batch row t is tenant t (:182, :192);
  * a request is T prompts, left-padded with token 0 to max(2^ceil(log2(len)), 64) and refused beyond 1024 (:297-302);
  * prefill `model(input_ids, attention_mask)`, then greedy `argmax(logits[:, -1])` fed back one token per tenant with the KV cache
    until every tenant has produced a stop token or max_new_tokens is reached (:190-258).  Positions are the raw indices of the
    padded sequence (transformers 4.31's LlamaModel numbers positions from the cache length when none are passed).

This module is plumbing in PyTorch around the HIP ops -- attention, RoPE and norms are stock torch -- with three differences from
running the reference's modules under HF, all of them launch-count / traffic only, none numerical beyond rounding:
  * q+k+v and gate+up of a layer are ONE fused Linear each (weights and masks concatenated along the output dimension, one scale
    group per projection so every tenant keeps its own coeff per projection): 4 Linear launches per layer instead of 7;
  * the residual adds after o_proj / down_proj ride in the decode kernel's epilogue;
  * per-tenant embedding / norm / lm_head are batched (serving.DataParallelModule's mechanisms), not a Python loop;
  * the decode step is shape-static (KV cache of fixed length, a key-validity mask, the position held in a device tensor), so it
    is captured once as a hipGraph and replayed: greedy feedback happens on the device, the host only polls the stop flags.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .binary_gemm_kernel import (binary_linear, binary_linear_residual_norm, binary_linear_swiglu, binary_linear_decode, decode_shape_ok, fused_norm_ok, handoff_ok,
                                 pack_decode_masks, tenant_linear, tile_weight)
from .diff import binarize
from . import serving_ops as ops

MODEL_CONFIGS = {
    # name: (hidden, intermediate, layers, heads, kv_heads, vocab)
    "llama-2-7b": (4096, 11008, 32, 32, 32, 32000),
    "mistral-7b": (4096, 14336, 32, 32, 8, 32000),
    "llama-2-70b": (8192, 28672, 80, 64, 8, 32000),
    "tiny": (256, 512, 2, 4, 2, 512),
    "tiny128": (512, 1024, 2, 4, 1, 512),        # head_dim 128, 4 query heads per kv head: exercises the decode glue kernels
    "tiny4096": (4096, 4096, 2, 32, 8, 512),     # two Llama-width layers (the fused-norm launches need K >= 2048)
    "tiny2048": (2048, 2048, 2, 16, 4, 512),     # hidden % 2048 == 0: exercises the fused RMSNorm / SwiGLU launches
    "mistral-1layer": (4096, 14336, 1, 32, 8, 512),   # ONE full-width Mistral-7B layer (small vocabulary): full-size loop parity tests
}

NORM_HANDOFF_DEFAULT = True     # (profiles/r05_decode_step.txt)
PREFETCH_O_DEFAULT = False      # weight prefetch of the o projection on a graph side branch next to the attention launch (profiles/r06_decode_step.txt)
MAX_PROMPT = 1024      # demo/demo_backend.py:300-302
MIN_PAD = 64           # demo/demo_backend.py:299


def handoff_norm(nw):
    """RMSNorm hand-off, overflow-safe form (round 6).  The producing launch stores round16(x_raw * nw) of the UN-normalised residual
    stream; in fp16 that product could overflow on a massive-activation row with a large norm weight, where HF's order (normalise first) cannot.
    Returns (nw / s, s) with s the power of two >= max |nw| (over every tenant): |nw / s| <= 1, so |x * nw / s| <= |x| never overflows, whatever
    x is.  The producer is then given ssq_scale = 1 / s^2 and the consumer eps / s^2: its row scalar becomes s * rsqrt(mean(x^2) + eps), i.e. the
    SAME product -- powers of two, so the same bits wherever the unscaled form did not overflow / underflow."""
    m = float(nw.detach().abs().max())
    s = 1.0 if not math.isfinite(m) or m <= 0.0 else 2.0 ** math.ceil(math.log2(m))
    return (nw.float() / s).to(nw.dtype), s


def padded_length(longest):
    """demo/demo_backend.py:297-299: next power of two, at least 64."""
    return max(1 << max(longest - 1, 0).bit_length(), MIN_PAD)


class FusedDeltaLinear(nn.Module):
    """Several DiffCompress Linears that read the same input, launched as ONE:  W = cat(W_i), masks = cat(masks_i) along N,
    alpha[t, g] = coeff of the projection that owns scale group g (group size = gcd of the output widths).

    interleave8=True (two projections of equal width, a gate|up pair): the output rows are stored interleaved in blocks of 8
    ([g0..7 | u0..7 | g8..15 | u8..15 | ...]) so that a 16-column tile of the decode kernel holds 8 gate columns and the 8 matching
    up columns and SwiGLU can run in its epilogue.  `split()` undoes the order for callers that want the separate outputs."""

    tile_decode_weight = True     # keep a tile-major decode copy of the base weight (class switch)

    def __init__(self, weights, masks, coeffs, interleave8=False, decode_copies=True):
        """decode_copies=False skips the decode-only copies (packed sign words, tile-major base weight): a prefill-only user"""
        super().__init__()
        widths = [w.shape[0] for w in weights]
        self.widths = widths
        self.interleave8 = bool(interleave8)
        weight, mask = torch.cat(weights, 0), torch.cat(masks, 2)                               # [N, K], [T, K/32, N]
        if self.interleave8:
            assert len(widths) == 2 and widths[0] == widths[1] and widths[0] % 8 == 0
            inter = widths[0]
            perm = torch.arange(2 * inter, device=weight.device).view(2, inter // 8, 8).transpose(0, 1).reshape(-1)
            weight, mask = weight[perm], mask[:, :, perm]
            gsz = 8
            alpha = torch.stack([c.float().reshape(-1) for c in coeffs], 1)                     # [T, 2] (gate, up)
            self.register_buffer("alpha_pair", alpha.contiguous())
            alpha = alpha.repeat(1, inter // 8)                                                  # [T, N/8]: g, u, g, u, ...
        else:
            gsz = 0
            for n in widths:
                gsz = math.gcd(gsz, n)
            alpha = torch.cat([c.float().reshape(-1, 1).expand(-1, n // gsz) for c, n in zip(coeffs, widths)], 1)
            self.alpha_pair = None
        self.register_buffer("weight", weight.contiguous())
        self.register_buffer("mask", mask.contiguous())
        self.register_buffer("alpha", alpha.contiguous())                                        # [T, G]
        self.groups = alpha.shape[1]
        # decode copy of the sign words in the streaming kernel's packed order (tenants interleaved, natural k order); prefill keeps
        # the reference layout
        self.register_buffer("mask_packed", pack_decode_masks(self.mask) if (decode_copies and self.mask.shape[0] <= 16) else None)
        # ... and of the base weight in the kernel's tile-major order (one contiguous 4-KiB block per stage; +2 bytes per weight of HBM)
        N, K = self.weight.shape
        tiled = self.mask_packed is not None and self.tile_decode_weight and N % 16 == 0 and K % 128 == 0
        self.register_buffer("weight_tiled", tile_weight(self.weight) if tiled else None)

    def _decode_ok(self, x):
        B, M, K = x.shape
        return self.mask_packed is not None and B == self.mask.shape[0] and B * M <= 16 and \
            decode_shape_ok(B, M, self.weight.shape[0], K, B) and x.data_ptr() % 16 == 0 and \
            x.stride(0) % 8 == 0 and x.stride(1) % 8 == 0 and x.stride(2) == 1

    def fusable(self, x, swiglu=False):
        """True when forward_fused can take this input: decode shape inside the fused-norm envelope (and an interleaved pair for SwiGLU)"""
        B, M, K = x.shape
        return self._decode_ok(x) and fused_norm_ok(B, M, K) and (not swiglu or self.interleave8)

    def _dec_weight(self, x):
        """(weight, weight_tiled flag) for a decode launch: the tile-major copy when it exists and the launch has one row per tenant"""
        if self.weight_tiled is not None and self.use_tiled and x.shape[1] == 1:
            return self.weight_tiled, True
        return self.weight, False

    use_tiled = True              # (A/B switch)

    def forward(self, x, residual=None, out_dtype=None, ssq_out=None, next_norm=None, xw_out=None, out=None, ssq_scale=1.0):
        """out_dtype=torch.float32: un-rounded partial sums (the row-parallel shards of tp.py reduce them across ranks).
        ssq_out (decode, with residual): RMSNorm hand-off, producer side -- the launch also leaves the per-row partial sums of squares of the
        updated residual stream for the next launch (`handoff_producer_ok`), and with next_norm [T, N] + xw_out [T, 1, N] the copy of it
        pre-multiplied by the NEXT norm's weight.  out: caller-provided destination (tp.py: the all-reduce's peer-mapped buffer)."""
        if self._decode_ok(x):
            w, wt = self._dec_weight(x)
            return binary_linear_decode(x, w, self.mask_packed, self.alpha, layout="packed", groups=self.groups,
                                        residual=residual, weight_tiled=wt, out_dtype=out_dtype, ssq_out=ssq_out,
                                        norm_weight=next_norm if ssq_out is not None else None, xw_out=xw_out, out=out, ssq_scale=ssq_scale)
        assert ssq_out is None and xw_out is None
        return binary_linear(x, self.weight, self.mask, self.alpha, groups=self.groups, residual=residual, out_dtype=out_dtype, out=out)

    def residual_norm_ok(self, x, residual):
        """forward_residual_norm can take this input: prefill rows on the fused GEMM's fast path, the residual stream contiguous, rows the norm
        kernels take whole (N % 8 == 0, N <= 8192)"""
        B, M, K = x.shape
        N = self.weight.shape[0]
        return (M > 16 and K % 64 == 0 and N % 8 == 0 and N <= 8192 and x.stride(2) == 1 and x.data_ptr() % 16 == 0 and x.stride(0) % 8 == 0 and
                x.stride(1) % 8 == 0 and residual.is_contiguous() and self.mask.shape[0] in (1, B))

    def forward_residual_norm(self, x, residual, norm_weight, eps):
        """(residual + Linear(x), rmsnorm_tenant(of that, norm_weight)): the residual Linear of a decoder layer and the norm in front of the next
        Linear, one launch fewer when the Linear is split over k (bd_binary_linear_residual_norm); bit-identical to the two calls"""
        return binary_linear_residual_norm(x, self.weight, self.mask, self.alpha, residual, norm_weight, eps, groups=self.groups)

    def handoff_producer_ok(self, x):
        """this (residual) Linear can leave the sums of squares of its output behind: decode shape, one row per tenant, tile-major weight"""
        return (self._decode_ok(x) and x.shape[1] == 1 and x.shape[0] <= 8 and self.weight.shape[0] % 16 == 0 and
                self.weight_tiled is not None and self.use_tiled)

    def handoff_consumer_ok(self, x, swiglu=False):
        """forward_fused(..., ssq_in=...) can take this input: the resident-row envelope, tile-major weight (an interleaved pair for SwiGLU)"""
        B, M, K = x.shape
        return (self._decode_ok(x) and handoff_ok(B, M, K) and self.weight_tiled is not None and self.use_tiled and
                (not swiglu or self.interleave8))

    def forward_fused(self, x, norm_weight, eps, swiglu=False, ssq_in=None):
        """[RMSNorm(x; norm_weight) ->] this Linear [-> SwiGLU] in ONE launch.  With norm_weight, x is the un-normalised residual stream
        (`fusable(x)`); norm_weight=None keeps only the SwiGLU epilogue (x already normalised; needs `_decode_ok(x)`).
        ssq_in: the partial sums of squares of the residual stream left by the launch that produced it -- the norm then costs no reduction at
        all (`handoff_consumer_ok`); with norm_weight=None, x is the producer's pre-multiplied copy."""
        if swiglu:
            w, wt = self._dec_weight(x)
            return binary_linear_decode(x, w, self.mask_packed, self.alpha_pair, layout="packed", groups=2,
                                        norm_weight=norm_weight, eps=eps, swiglu=True, weight_tiled=wt, ssq_in=ssq_in)
        w, wt = self._dec_weight(x)
        return binary_linear_decode(x, w, self.mask_packed, self.alpha, layout="packed", groups=self.groups,
                                    norm_weight=norm_weight, eps=eps, weight_tiled=wt, ssq_in=ssq_in)

    def swiglu_ok(self, x):
        """True when forward_swiglu can take this input: an interleaved gate|up pair at prefill size on the fused GEMM's fast path"""
        B, M, K = x.shape
        return self.interleave8 and M > 16 and K % 64 == 0 and x.data_ptr() % 16 == 0 and x.stride(2) == 1 and \
            x.stride(0) % 8 == 0 and x.stride(1) % 8 == 0 and (self.mask.shape[0] in (1, B))

    def forward_swiglu(self, x):
        """act_fn(gate_proj(x)) * up_proj(x) of the MLP in ONE launch at prefill size (bd_binary_linear_swiglu): [B, M, inter]"""
        return binary_linear_swiglu(x, self.weight, self.mask, self.alpha_pair)

    def split(self, y):
        """per-projection outputs of y = forward(x)  (undoes the interleaved row order)"""
        if self.interleave8:
            v = y.reshape(*y.shape[:-1], self.widths[0] // 8, 2, 8)
            return v[..., 0, :].reshape(*y.shape[:-1], -1), v[..., 1, :].reshape(*y.shape[:-1], -1)
        return y.split(self.widths, dim=-1)

    def column_alpha(self, t):
        """tenant t's scale of every stored output row, [N]"""
        return self.alpha[t].repeat_interleave(self.weight.shape[0] // self.groups)

    def linear_bytes(self):
        """algorithmic HBM bytes of one decode launch: base once + every tenant's signs (activations / outputs are noise)"""
        return self.weight.numel() * self.weight.element_size() + self.mask.numel() * 4


def _rope_tables(length, dim, device, dtype, base=10000.0):
    inv = 1.0 / (base ** (torch.arange(0, dim, 2, device=device, dtype=torch.float32) / dim))
    ang = torch.outer(torch.arange(length, device=device, dtype=torch.float32), inv)
    ang = torch.cat([ang, ang], dim=-1)
    d = dim // 2
    sin = ang.sin()
    sin = torch.cat([-sin[:, :d], sin[:, d:]], dim=-1)                  # rotate-half sign folded in
    return ang.cos().to(dtype), sin.to(dtype)


def _rope(x, cos, sin):
    # x [T, H, S, D]; cos / sin [S, D]
    d = x.shape[-1] // 2
    rot = torch.cat([x[..., d:], x[..., :d]], dim=-1)
    return torch.addcmul(x * cos, rot, sin)


class TenantDecoder(nn.Module):
    """Llama / Mistral decoder for T tenants over one base: every `*proj*` Linear is base + T 1-bit deltas (fused HIP launches),
    embedding / norms / lm_head are per tenant (stacked)."""

    MIN_STOP_WIDTH = 8          # stop ids per tenant the static stop table holds before it has to grow (to the next power of two)
    MAX_STATIC_SLOTS = 4        # captured decode-step graphs kept alive at once (LRU over stop-table width x glue switches)

    def __init__(self, cfg, tenants, device, dtype, max_len=MAX_PROMPT + 64, eps=1e-5):
        super().__init__()
        self.cfg, self.T, self.dtype, self.dev, self.eps = cfg, tenants, dtype, torch.device(device), eps
        hid, inter, nl, heads, kvh, vocab = cfg
        self.hd = hid // heads
        self.max_len = max_len
        self.layers = nn.ModuleList()
        self.embed = None           # [T, vocab, hid]
        self.final_norm = None      # [T, hid]
        self.lm_head = None         # [T, vocab, hid]
        cos, sin = _rope_tables(max_len, self.hd, self.dev, dtype)
        self.register_buffer("cos", cos, persistent=False)
        self.register_buffer("sin", sin, persistent=False)
        self._graph = None
        self.fast_glue = True       # decode steps use the HIP glue kernels (serving_ops); False = stock torch ops everywhere
        self.swiglu_epilogue = False  # prefill: SwiGLU inside the gate|up GEMM (256-row tiles from 256 rows per tenant, pair tiles for <= 64-row
                                      # prompts): measured slower than GEMM + one pass both times (6 x 64 rows: gate|up 162 -> 186 us against a 9-us pass)
        self.step_kernels = True      # decode step: mask extension + embedding gather, and argmax + token / output / stop / position updates, as
                                      # one HIP launch each instead of ~13 stock ops (round 6)
        self.short_prompt_fusions = True   # multi-tenant prefill of short prompts (round 6): RoPE + KV-cache append in one launch; <= 64 rows per
                                           # tenant: the norms ride on the split-k reduce launches of o / down
        self.hip_prefill_attention = True      # prefill: RoPE + flash-style attention kernels instead of torch SDPA over a [L, Lc] mask
        self.fuse_glue = True       # ... and fold RMSNorm / SwiGLU into the Linear launches where the shapes allow (bit-identical)
        # Which glue is folded where, by same-process A/B of the whole step (ms per step, Mistral-7B x 6, tile-major weights).  Round 2
        # (profiles/r02_decode_step.txt): separate launches 5.35 | SwiGLU in gate|up's epilogue 5.11 | + RMSNorm in gate|up's prologue
        # 5.17 | + RMSNorm in q|k|v's prologue 5.28.  Round 4, nt weight loads (profiles/r04_decode_step_ab.txt): 5.28 | 5.02 | 4.94 | 5.05.
        # The epilogue is free; a norm prologue makes 256 blocks each re-read and re-normalise all rows (~6 us in front of the launch)
        # to save a 4.4 us kernel: it pays on the long gate|up launch, not on the short q|k|v one.
        self._static = {}               # (stop-table width, glue switches) -> static request state + captured decode-step graph (LRU)
        self._kv_cache = None           # ONE KV cache per decoder, shared by every slot
        self._capture_stream = None
        self.fuse_qkv_norm = False      # RMSNorm folded into the q|k|v launch
        # RMSNorm folded into the gate|up launch (SwiGLU stays in its epilogue either way): every block re-normalises all rows (rows + norm
        # weights from L2) to save one ~4.5 us launch.  Same-process A/Bs at the end of round 4, with the resident-row form (which needs no
        # norm) at 2 stages on every eligible launch: 1 tenant 3.353 fused vs 3.386 ms separate, 2 tenants 3.790 vs 3.823, 4 tenants 4.244
        # vs 4.226 (and 4.218 vs 4.243 in a later session: a wash), 6 tenants 4.771 vs 4.700 -- the prologue's cost grows with the rows,
        # the saved launch does not (profiles/r04_decode_step_ab.txt).
        self.fuse_gateup_norm = tenants <= 4
        # RMSNorm by HAND-OFF (round 5): o / down leave the per-row partial sums of squares of the residual stream they write; the next
        # q|k|v / gate|up launch applies the norm weight on its resident rows and 1/rms in its epilogue -- neither a stand-alone norm launch
        # nor a per-block reduction.  Takes precedence over the two switches above where its envelope holds (tile-major weights, <= 8
        # tenants, hidden >= 2048); not bit-identical to the separate launches (one rounding moves), same accuracy.
        self.norm_handoff = NORM_HANDOFF_DEFAULT
        # Weight prefetch on a hipGraph SIDE BRANCH (round 6): forked before the decode attention launch, joined before the o projection, a
        # cache_warm launch reads the o projection's tile-major weight and packed sign words so that they sit in the Infinity Cache when the
        # projection starts (the attention launch is a latency-bound chain that leaves HBM idle).  No arithmetic changes.
        self.prefetch_o = PREFETCH_O_DEFAULT
        self._pf_stream = None
        self._ssq = None                # [hidden / 16, 16] fp32: partial sums of squares of the current residual stream
        self._xw = None                 # [T, 1, hidden]: the residual stream times the weight of the norm that reads it next
        # (round 2 also shipped a persistent per-layer chain launch, bd_decode_chain: bit-identical but 5.91 vs 5.33 ms per step in every
        # same-process A/B, so it was removed from the library in round 3 -- profiles/r02_decode_chain_*.txt keep the measurements)

    # ---------------------------------------------------------------- construction
    @classmethod
    def synthetic(cls, name, tenants, device, dtype=torch.float16, seed=0, layers=None, max_len=MAX_PROMPT + 64, shared_heads=False):
        """Random weights with the statistics of SURVEY.md section 8(d): W ~ N(0, 0.02^2), fine-tune = W + N(0, (5e-4)^2) per
        tenant (alpha = mean|delta| ~ 4e-4).  Embedding / norm / lm_head are per tenant as in the reference's diff.pt files
        (`shared_heads=True` stores them once and expands -- same arithmetic, used only to keep test models small)."""
        cfg = MODEL_CONFIGS[name] if isinstance(name, str) else tuple(name)
        hid, inter, nl, heads, kvh, vocab = cfg
        nl = layers or nl
        self = cls(cfg, tenants, device, dtype, max_len=max_len)
        gen = torch.Generator(device=device).manual_seed(seed)
        hd = hid // heads

        def delta_linear(n_out, n_in):
            w = (torch.randn(n_out, n_in, device=device, generator=gen) * 0.02).to(dtype)
            masks, coeffs = [], []
            for _ in range(tenants):
                fine = (w.float() + torch.randn(n_out, n_in, device=device, generator=gen) * 5e-4).to(dtype)
                m, c = binarize(w, fine)
                masks.append(m)
                coeffs.append(c)
            return w, torch.stack(masks, 0), torch.stack(coeffs, 0)

        def fused(*shapes, interleave8=False):
            parts = [delta_linear(o, i) for o, i in shapes]
            return FusedDeltaLinear([p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts], interleave8=interleave8)

        def per_tenant(*shape, scale=None):
            reps = 1 if shared_heads else tenants
            if scale is None:
                t = 1.0 + 0.1 * torch.randn(reps, *shape, device=device, generator=gen)
            else:
                t = torch.randn(reps, *shape, device=device, generator=gen) * scale
            t = t.to(dtype)
            return t.expand(tenants, *shape) if shared_heads else t

        for _ in range(nl):
            layer = nn.Module()
            layer.qkv = fused((heads * hd, hid), (kvh * hd, hid), (kvh * hd, hid))
            layer.o = fused((hid, heads * hd))
            layer.gate_up = fused((inter, hid), (inter, hid), interleave8=(inter % 8 == 0))
            layer.down = fused((hid, inter))
            layer.norm1 = per_tenant(hid)
            layer.norm2 = per_tenant(hid)
            self.layers.append(layer)
        self.embed = per_tenant(vocab, hid, scale=0.02)
        self.final_norm = per_tenant(hid)
        self.lm_head = per_tenant(vocab, hid, scale=0.02)
        return self

    # ---------------------------------------------------------------- accounting
    def linear_bytes_per_step(self):
        """algorithmic HBM bytes of the Linear launches of one decode step (the roofline numerator): delta Linears + lm_heads"""
        b = sum(l.qkv.linear_bytes() + l.o.linear_bytes() + l.gate_up.linear_bytes() + l.down.linear_bytes() for l in self.layers)
        return b, self.T * self.lm_head.shape[1] * self.lm_head.shape[2] * self.lm_head.element_size()

    def linear_param_count(self):
        return sum(l.qkv.weight.numel() + l.o.weight.numel() + l.gate_up.weight.numel() + l.down.weight.numel() for l in self.layers)

    # ---------------------------------------------------------------- forward
    def new_cache(self, length=None):
        length = length or self.max_len
        _, _, _, heads, kvh, _ = self.cfg
        mk = lambda: torch.zeros(self.T, kvh, length, self.hd, device=self.dev, dtype=self.dtype)
        return {"k": [mk() for _ in self.layers], "v": [mk() for _ in self.layers],
                "valid": torch.zeros(self.T, length, dtype=torch.bool, device=self.dev)}

    # prefill: the HIP per-tenant RMSNorm (ONE launch).  Its block-per-row form is used up to this many rows and stock torch beyond (F.rms_norm +
    # the weight multiply = three launches, but faster per row on long prompts: 2048-row prefill 8.5 vs 13.1 us, profiles/r03_prefill_glue.txt);
    # since late in round 5 hidden sizes that are multiples of 2048 take the wave-per-row form at every size (rmsnorm_rows_kernel, bit-identical):
    # 6 tenants x 256 rows 10.7 us against 21.2 for the torch composition, 2048 rows 12.0 against 22.5 (tools/bench_norm.py; torch's rms_norm with
    # a FUSED weight -- one weight for all rows, not this loop's per-tenant weights -- is 10.1: bench_model keeps it).  A 6-tenant request
    # padded to 64 tokens (384 rows) spends 15.5 us per norm in the three torch kernels against ~5 us here (profiles/r05_mt_prefill_tiles.txt).
    HIP_NORM_MAX_ROWS = 1024

    def _norm(self, x, w):
        H = x.shape[-1]
        # (hidden sizes of 2048 / 4096 / 6144 / 8192: the wave-per-row kernel, any number of rows -- bd_srv_rmsnorm picks it from 64 rows on)
        if self.fast_glue and H % 8 == 0 and (x.shape[1] <= 16 or x.shape[0] * x.shape[1] <= self.HIP_NORM_MAX_ROWS or (H % 2048 == 0 and H <= 8192)):
            return ops.rmsnorm_tenant(x if x.is_contiguous() else x.contiguous(), w, self.eps)
        return F.rms_norm(x, (x.shape[-1],), None, self.eps) * w[:, None, :]

    def _hn(self, layer, which):
        """(nw / s, s) of a layer's norm for the hand-off (handoff_norm), computed once"""
        key = "_hn_" + which
        v = getattr(layer, key, None)
        if v is None:
            v = handoff_norm(getattr(layer, which))
            setattr(layer, key, v)
        return v

    def _layer(self, layer, x, cos, sin, cache, li, pos_idx, attn_mask, ssq_valid=False, next_layer=None, h_in=None):
        """one decoder layer; returns (x, ssq_valid): whether self._ssq / self._xw hold the partial sums of squares of the returned x and its
        copy pre-multiplied by the NEXT layer's input norm weight (next_layer; None after the last layer) -- both in handoff_norm's scaled form.
        h_in: norm1 of x when the previous layer's down projection already produced it (prefill, forward_residual_norm); the third return value is
        that tensor for the next layer (or None)."""
        T, S, hid = x.shape
        _, inter, _, heads, kvh, _ = self.cfg
        hd = self.hd
        fuse = S == 1 and self.fast_glue and self.fuse_glue and x.is_contiguous()
        handoff = fuse and self.norm_handoff and hid % 16 == 0
        if handoff and self._ssq is None:
            self._ssq = torch.empty(hid // 16, 16, dtype=torch.float32, device=x.device)
            self._xw = torch.empty(T, 1, hid, dtype=x.dtype, device=x.device)
        if handoff and ssq_valid:
            s1 = self._hn(layer, "norm1")[1]
            qkv = layer.qkv.forward_fused(self._xw, None, self.eps / (s1 * s1), ssq_in=self._ssq)    # rows already carry norm1's weight / s; s/rms in the epilogue
        elif fuse and self.fuse_qkv_norm and layer.qkv.fusable(x):
            qkv = layer.qkv.forward_fused(x, layer.norm1, self.eps)              # RMSNorm in the Linear's prologue: one launch
        else:
            qkv = layer.qkv(h_in if h_in is not None else self._norm(x, layer.norm1))
        ck, cv = cache["k"][li], cache["v"][li]
        short = S > 1 and self.fast_glue and self.short_prompt_fusions
        pairs = short and S <= 64 and T >= 2                                 # the request runs on pair tiles (two tenants per 128-row tile)
        pf = None
        if S == 1 and self.fast_glue and self.prefetch_o and layer.o.mask_packed is not None:
            # fork: the prefetch depends on nothing the layer computes; it is ordered behind the q|k|v launch only so that it runs next to attention
            if self._pf_stream is None:
                self._pf_stream = torch.cuda.Stream(device=x.device)
            pf, cur = self._pf_stream, torch.cuda.current_stream(x.device)
            pf.wait_stream(cur)
            with torch.cuda.stream(pf):
                w_o = layer.o.weight_tiled if (layer.o.weight_tiled is not None and layer.o.use_tiled) else layer.o.weight
                ops.cache_warm(w_o, layer.o.mask_packed)
        if S == 1 and self.fast_glue and ops.decode_attention_supported(heads, kvh, hd):
            # decode: RoPE + cache append + attention over the valid keys in ONE launch (pos_idx is a one-element device tensor)
            a = ops.decode_attention(qkv, self.cos, self.sin, ck, cv, cache["valid"], pos_idx, heads, kvh)
        elif (S > 1 and self.fast_glue and cache.get("kv_start") is not None and hd == 128 and S % 64 == 0 and qkv.is_contiguous()
              and not layer.qkv.interleave8):
            # prefill from position 0: in-place RoPE on the q and k slices of the fused projection output, flash-style attention over the
            # left-padded prompts (keys kv_start[t] .. query position), then the K / V rows go into the cache
            nq, nk = heads * hd, kvh * hd
            qf, kf, vf = qkv[..., :nq], qkv[..., nq:nq + nk], qkv[..., nq + nk:]
            k4, v4 = kf.view(T, S, kvh, hd), vf.view(T, S, kvh, hd)
            if short and ck.is_contiguous() and cv.is_contiguous():
                ops.rope_kv_append_(qkv, self.cos, self.sin, ck, cv, heads, kvh, 0)       # RoPE of q and k + both cache writes: one launch
                a = ops.prefill_attention(qf.view(T, S, heads, hd), k4, v4, kv_start=cache["kv_start"], causal=True)
            else:
                ops.rope_(qkv[..., :nq + nk], self.cos, self.sin, heads + kvh, S, 0)      # q and k heads: one launch
                a = ops.prefill_attention(qf.view(T, S, heads, hd), k4, v4, kv_start=cache["kv_start"], causal=True)
                ck[:, :, :S] = k4.transpose(1, 2)
                cv[:, :, :S] = v4.transpose(1, 2)
        else:
            q, k, v = layer.qkv.split(qkv)
            if cos is None:                                                     # (forward() skips the gather when it expects the HIP attention)
                cos, sin = self.cos[pos_idx], self.sin[pos_idx]
            q = _rope(q.view(T, S, heads, hd).transpose(1, 2), cos, sin)
            k = _rope(k.view(T, S, kvh, hd).transpose(1, 2), cos, sin)
            v = v.view(T, S, kvh, hd).transpose(1, 2)
            ck.index_copy_(2, pos_idx, k)
            cv.index_copy_(2, pos_idx, v)
            a = F.scaled_dot_product_attention(q, ck, cv, attn_mask=attn_mask, enable_gqa=(kvh != heads))
            a = a.transpose(1, 2).reshape(T, S, heads * hd)
        if pf is not None:
            torch.cuda.current_stream(x.device).wait_stream(pf)           # join before the o projection
        o_hand = handoff and layer.o.handoff_producer_ok(a) and layer.gate_up.handoff_consumer_ok(x, swiglu=True)
        n2h, s2 = self._hn(layer, "norm2") if o_hand else (None, 1.0)
        h2 = None
        if pairs and layer.o.residual_norm_ok(a, x):
            x, h2 = layer.o.forward_residual_norm(a, x, layer.norm2, self.eps)    # o + residual, norm2 on its split-k reduce launch
        else:
            x = layer.o(a, residual=x, ssq_out=self._ssq if o_hand else None, next_norm=n2h, xw_out=self._xw if o_hand else None,
                        ssq_scale=1.0 / (s2 * s2))
        if h2 is not None:
            if self.swiglu_epilogue and layer.gate_up.swiglu_ok(h2):
                act = layer.gate_up.forward_swiglu(h2)                            # gate|up -> SwiGLU in the pair tile's epilogue (A/B: slower)
            else:
                act = ops.swiglu_interleaved8(layer.gate_up(h2)) if layer.gate_up.interleave8 else ops.swiglu(layer.gate_up(h2), inter)
        elif o_hand:
            act = layer.gate_up.forward_fused(self._xw, None, self.eps / (s2 * s2), swiglu=True, ssq_in=self._ssq)   # (RMSNorm by hand-off) gate|up -> SwiGLU
        elif fuse and self.fuse_gateup_norm and layer.gate_up.fusable(x, swiglu=True):
            act = layer.gate_up.forward_fused(x, layer.norm2, self.eps, swiglu=True)   # RMSNorm -> gate|up -> SwiGLU: one launch
        elif fuse and layer.gate_up.interleave8 and layer.gate_up._decode_ok(x):
            act = layer.gate_up.forward_fused(self._norm(x, layer.norm2), None, self.eps, swiglu=True)   # gate|up -> SwiGLU: one launch
        elif self.fast_glue and self.swiglu_epilogue and S >= 256 and layer.gate_up.swiglu_ok(x):
            act = layer.gate_up.forward_swiglu(self._norm(x, layer.norm2))       # prefill: gate|up -> SwiGLU in the GEMM's epilogue (A/B)
        else:
            gu = layer.gate_up(self._norm(x, layer.norm2))
            if self.fast_glue and inter % 8 == 0 and (S <= 16 or layer.gate_up.interleave8):
                # one elementwise pass on the projection output (any prompt length when the rows are interleaved in blocks of 8)
                act = ops.swiglu_interleaved8(gu) if layer.gate_up.interleave8 else ops.swiglu(gu, inter)
            else:
                g, u = layer.gate_up.split(gu)
                act = F.silu(g) * u
        d_hand = handoff and next_layer is not None and layer.down.handoff_producer_ok(act) and layer.qkv.handoff_consumer_ok(x)
        n1h, s1n = self._hn(next_layer, "norm1") if d_hand else (None, 1.0)
        if h2 is not None and next_layer is not None and layer.down.residual_norm_ok(act, x):
            x, h_next = layer.down.forward_residual_norm(act, x, next_layer.norm1, self.eps)   # down + residual, the NEXT layer's norm1 on its reduce
            return x, False, h_next
        x = layer.down(act, residual=x, ssq_out=self._ssq if d_hand else None, next_norm=n1h, xw_out=self._xw if d_hand else None,
                       ssq_scale=1.0 / (s1n * s1n))
        return x, d_hand, None

    @torch.no_grad()
    def forward(self, ids, pos_idx, cache, attn_mask, x=None):
        """ids [T, S]; pos_idx [S] (device, positions of these tokens in the cache); attn_mask [T, 1, S, L] bool.
        Returns the logits of the LAST position, [T, vocab].  x: the embedded tokens when the caller already has them (step_begin)."""
        T, S = ids.shape
        _, _, _, heads, kvh, _ = self.cfg
        # (the rows of the rotary tables are gathered only for the stock-op attention paths: the HIP kernels index the tables themselves)
        hip_attn = self.fast_glue and ((S == 1 and ops.decode_attention_supported(heads, kvh, self.hd)) or
                                       (S > 1 and cache.get("kv_start") is not None and self.hd == 128 and S % 64 == 0))
        cos, sin = (None, None) if hip_attn else (self.cos[pos_idx], self.sin[pos_idx])
        if x is None:
            t_idx = torch.arange(T, device=ids.device).view(T, 1)
            x = self.embed[t_idx, ids]                                        # per-tenant embedding: one gather
        ssq_valid = False                                                     # (the embedding rows have no producer launch: layer 0 norms itself)
        h_in = None
        for li, layer in enumerate(self.layers):
            nxt = self.layers[li + 1] if li + 1 < len(self.layers) else None
            x, ssq_valid, h_in = self._layer(layer, x, cos, sin, cache, li, pos_idx, attn_mask, ssq_valid, nxt, h_in)
        last = self._norm(x[:, -1:, :], self.final_norm)
        return tenant_linear(last, self.lm_head)[:, 0, :]                     # per-tenant lm_head: one launch

    # ---------------------------------------------------------------- request handling (demo_backend.py:261-315 + :190-258)
    def prepare(self, prompts):
        """Left-pad T prompts with token 0 to a power of two >= 64; returns (ids [T, L], attention mask [T, L]) on the device."""
        assert len(prompts) == self.T, "one prompt per tenant (batch row t is tenant t)"
        longest = max(len(p) for p in prompts)
        L = padded_length(longest)
        if L > MAX_PROMPT:
            raise ValueError("max_len too large, please reduce the input length")      # the reference's refusal, as an exception
        ids = torch.zeros(self.T, L, dtype=torch.long)
        am = torch.zeros(self.T, L, dtype=torch.bool)
        for t, p in enumerate(prompts):
            ids[t, L - len(p):] = torch.tensor(p, dtype=torch.long)
            am[t, L - len(p):] = True
        return ids.to(self.dev), am.to(self.dev)

    @torch.no_grad()
    def prefill(self, ids, attention_mask, cache):
        T, L = ids.shape
        pos_idx = torch.arange(L, device=self.dev)
        cache["valid"].zero_()
        cache["valid"][:, :L] = attention_mask
        Lc = cache["k"][0].shape[2]
        causal = torch.ones(L, Lc, dtype=torch.bool, device=self.dev).tril()                     # query i sees keys <= i
        mask = causal[None, None] & cache["valid"][:, None, None, :]
        # fully masked query rows (left pads) would be NaN in softmax: let a pad see itself; its output is never used
        mask = mask | torch.eye(L, Lc, dtype=torch.bool, device=self.dev)[None, None]
        # left-padded prompts (prepare() builds them so): the HIP prefill attention takes the first valid key of each tenant instead of
        # the [T, 1, L, Lc] mask; any other mask shape keeps the torch attention
        am = attention_mask.bool()
        left_padded = bool((am[:, 1:] | ~am[:, :-1]).all()) if L > 1 else True
        cache["kv_start"] = (L - am.sum(dim=1)).to(torch.int32) if (left_padded and self.hip_prefill_attention) else None
        try:
            return self.forward(ids, pos_idx, cache, mask)
        finally:
            cache["kv_start"] = None

    def _decode_step(self, st):
        """one greedy step on static buffers: st['tok'] [T,1] -> logits -> argmax -> st['tok']; position / masks advance on device"""
        cache = st["cache"]
        if self.fast_glue and self.step_kernels and self.embed.shape[-1] % 8 == 0 and self.lm_head.shape[-2] % 8 == 0:
            # the two ends of the step as ONE launch each (serving_ops.step_begin / step_end): mask extension + embedding, then argmax + the
            # token / output / stop-flag / position updates
            if "ticket" not in st:
                st["ticket"] = torch.zeros(1, dtype=torch.int32, device=self.dev)      # (first call = outside any graph capture)
            x = ops.step_begin(self.embed, st["tok"], cache["valid"], st["pos"])
            logits = self.forward(st["tok"], st["pos"], cache, cache["valid"][:, None, None, :], x=x)
            if logits.stride(1) == 1 and logits.stride(0) % 8 == 0 and logits.data_ptr() % 16 == 0:
                ops.step_end(logits, st["tok"], st["out"], st["step"], st["pos"], st["stop_ids"], st["stopped"], st["ticket"])
                return
        else:
            cache["valid"].index_fill_(1, st["pos"], True)
            mask = cache["valid"][:, None, None, :]
            logits = self.forward(st["tok"], st["pos"], cache, mask)
        nxt = torch.argmax(logits, dim=-1)
        st["tok"].copy_(nxt[:, None])
        st["out"].index_copy_(1, st["step"], nxt[:, None])
        st["stopped"] |= (nxt[:, None] == st["stop_ids"]).any(dim=1)
        st["pos"] += 1
        st["step"] += 1

    @torch.no_grad()
    def generate(self, prompts, max_new_tokens=16, stop_token_ids=None, use_graph=True, check_every=1):
        """Greedy decoding of T prompts (one per tenant).  Returns (new_tokens [T, n] on the CPU, n_steps).
        stop_token_ids: per-tenant lists of token ids; generation ends when every tenant has produced one (or at max_new_tokens),
        exactly like the reference's loop; tokens after a tenant's stop token are still generated (the reference does the same and
        hides them in its response formatting)."""
        ids, am = self.prepare(prompts)
        T, L = ids.shape
        assert L + max_new_tokens <= self.max_len
        nstop = max((len(s) for s in stop_token_ids), default=0) if stop_token_ids else 0
        # Static request state, reused by every generate() call: ONE KV cache per decoder (it does not depend on the request), feedback
        # buffers sized to fixed maxima (`out` holds max_len tokens, the stop table MIN_STOP_WIDTH ids or the next power of two), so
        # the captured hipGraph of the decode step is keyed by the stop-table width and the glue switches only -- not by caller-chosen
        # max_new_tokens (every new key used to allocate a full KV cache + a graph and nothing was ever evicted).  The
        # few slots that can exist are kept in a small LRU; an evicted slot's graph and buffers are released.
        # Stale keys of an earlier request are masked by cache["valid"].
        width = max(self.MIN_STOP_WIDTH, 1 << max(nstop - 1, 0).bit_length())
        key = (width, self.fast_glue, self.fuse_glue, self.fuse_qkv_norm, self.fuse_gateup_norm, self.norm_handoff, FusedDeltaLinear.use_tiled,
               self.prefetch_o, self.step_kernels)
        if self._kv_cache is None:
            self._kv_cache = self.new_cache()
        slot = self._static.pop(key, None)
        if slot is None:
            slot = {"st": {
                "cache": self._kv_cache, "tok": torch.zeros(T, 1, dtype=torch.long, device=self.dev),
                "pos": torch.zeros(1, dtype=torch.long, device=self.dev), "step": torch.zeros(1, dtype=torch.long, device=self.dev),
                "stop_ids": torch.full((T, width), -1, dtype=torch.long, device=self.dev),
                "out": torch.zeros(T, self.max_len + 1, dtype=torch.long, device=self.dev),
                "stopped": torch.zeros(T, dtype=torch.bool, device=self.dev)}, "graph": None}
        self._static[key] = slot                              # (re-)inserted last: dict order is the LRU order
        while len(self._static) > self.MAX_STATIC_SLOTS:
            old = self._static.pop(next(iter(self._static)))
            old["graph"] = None                               # drops the captured graph (and with it the references to the buffers)
            old["st"].clear()
        st = slot["st"]
        cache = st["cache"]
        logits = self.prefill(ids, am, cache)
        st["stop_ids"].fill_(-1)
        if stop_token_ids:
            for t, s_ in enumerate(stop_token_ids):
                if len(s_):
                    st["stop_ids"][t, :len(s_)] = torch.tensor(sorted(s_), dtype=torch.long, device=self.dev)
        first = torch.argmax(logits, dim=-1)
        st["tok"].copy_(first[:, None])
        st["pos"].fill_(L)
        st["step"].fill_(1)
        st["out"].zero_()
        st["out"][:, 0] = first
        st["stopped"].copy_((first[:, None] == st["stop_ids"]).any(dim=1))
        n = 1
        if use_graph and max_new_tokens > 1:
            if slot["graph"] is None:
                slot["graph"] = self._graph_runner(st)
            runner = slot["graph"]
        else:
            runner = lambda: self._decode_step(st)
        while n < max_new_tokens:
            if (n - 1) % check_every == 0 and bool(st["stopped"].all()):
                break
            runner()
            n += 1
        return st["out"][:, :n].cpu(), n

    def _graph_runner(self, st):
        """Capture one decode step as a hipGraph on the request's static buffers and return a replay callable.  The launch-bound
        step (4 Linear launches + ~20 small torch ops per layer) replays without per-op host overhead."""
        # ONE capture stream per decoder: the library's scratch is keyed by (device, stream), so the warm-up step below allocates it on
        # the very stream the capture then runs on (nothing is allocated or memset inside the graph), and later captures reuse it
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream(device=self.dev)
        side = self._capture_stream
        snap = {k: v.clone() for k, v in st.items() if torch.is_tensor(v)}
        snap_valid = st["cache"]["valid"].clone()
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):                     # warm-up outside capture (allocations, workspace, lazy init)
            self._decode_step(st)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            self._decode_step(st)
        torch.cuda.synchronize(self.dev)
        for k, v in snap.items():                         # undo the two trial steps: replay starts from the request's real state
            st[k].copy_(v)
        st["cache"]["valid"].copy_(snap_valid)
        return g.replay
