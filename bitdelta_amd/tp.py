"""Tensor-parallel 1-bit-delta Linear for the one configuration that does not fit a GPU: Llama-2-70B base + chat delta, TP = 8
(BASELINE.json configs[3]; model pair of the reference's scripts/multigpu_train_example.bash:1-13, which itself only knows
accelerate's layer-wise device_map -- SURVEY.md section 2.4 / 8e row 3).  One process per GPU, torch.distributed (backend "nccl"
is RCCL over xGMI on ROCm).

Partitioning (Megatron style, the only exchange is one all-reduce per row-parallel Linear):
    q / k / v / gate / up   COLUMN parallel: rank r keeps output columns [r N/w, (r+1) N/w).  Base weight rows and packed-mask
                            columns are plain slices ([K/32, N] words: a column slice is a slice); no communication.
    o / down                ROW parallel: rank r keeps input features [r K/w, (r+1) K/w): base weight columns and mask WORD ROWS
                            [r K/32/w, ...) (K/w is a multiple of 32 for every Llama-2-70B shape: 8192/8 = 1024, 28672/8 = 3584).
                            Each rank computes  x_r . W_r^T + coeff * (x_r . S_r)  -- coeff is a per-matrix scalar, so it is applied
                            BEFORE the reduction inside the fused kernel epilogue -- and the partial sums are all-reduced.
    coeff                   replicated (it is the mean |delta| of the WHOLE matrix, bitdelta/diff.py:12).
    attention               head parallel (64 heads / 8 GQA groups: one kv head + 8 query heads per rank); embedding, norms, lm_head
                            replicated.
Reduction precision: partial sums leave the kernel in fp32 and are all-reduced in `reduce_dtype`: fp32 for decode-sized messages
(16-64 KB: latency-bound, precision is free), the activation dtype for prefill-sized ones (2048 x 8192: 32 MB instead of 64 MB per
all-reduce; one extra rounding of each of the 8 partials, <= 2^-9 relative each).  Messages per step: 2 per layer x 80 layers.
"""
import os
import time

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .binary_gemm_kernel import binary_linear
from .diff import binarize
from .dist import shard_mask_columns, shard_mask_rows
from . import serving_ops as ops


def _world():
    return dist.get_world_size() if dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_initialized() else 0


class ColumnParallelBinaryDiff(nn.Module):
    """Rank-local slice [N/w, K] of a BinaryDiff Linear, split along the OUTPUT dimension.  forward: x [.., K] -> y [.., N/w]."""

    def __init__(self, weight_shard, mask_shard, coeff):
        super().__init__()
        self.register_buffer("weight", weight_shard.contiguous())           # [N/w, K]
        self.register_buffer("mask", mask_shard.contiguous())               # [K/32, N/w]
        self.register_buffer("coeff", coeff.detach().float().reshape(1, 1))

    @classmethod
    def from_full(cls, weight, mask, coeff, rank=None, world=None):
        rank, world = _rank() if rank is None else rank, _world() if world is None else world
        n = weight.shape[0] // world
        assert weight.shape[0] % world == 0
        return cls(weight[rank * n:(rank + 1) * n], shard_mask_columns(mask, rank, world), coeff)

    def forward(self, x):
        shape = x.shape
        y = binary_linear(x.reshape(1, -1, shape[-1]), self.weight, self.mask.unsqueeze(0), self.coeff)
        return y.reshape(*shape[:-1], y.shape[-1])


class RowParallelBinaryDiff(nn.Module):
    """Rank-local slice [N, K/w] of a BinaryDiff Linear, split along the INPUT dimension.  forward: x_r [.., K/w] -> y [.., N]
    (partial sums all-reduced over the process group)."""

    def __init__(self, weight_shard, mask_shard, coeff, group=None):
        super().__init__()
        self.register_buffer("weight", weight_shard.contiguous())           # [N, K/w]  (made contiguous: rows are k-contiguous)
        self.register_buffer("mask", mask_shard.contiguous())               # [K/32/w, N]
        self.register_buffer("coeff", coeff.detach().float().reshape(1, 1))
        self.group = group
        self._reduce = None

    @classmethod
    def from_full(cls, weight, mask, coeff, rank=None, world=None, group=None):
        rank, world = _rank() if rank is None else rank, _world() if world is None else world
        k = weight.shape[1] // world
        assert weight.shape[1] % world == 0 and k % 32 == 0, "K / world must be a multiple of 32"
        return cls(weight[:, rank * k:(rank + 1) * k], shard_mask_rows(mask, rank, world), coeff, group)

    def partial(self, x, out_dtype=torch.float32, out=None):
        shape = x.shape
        if out is not None:
            out = out.view(1, -1, out.shape[-1])
        y = binary_linear(x.reshape(1, -1, shape[-1]), self.weight, self.mask.unsqueeze(0), self.coeff, out_dtype=out_dtype, out=out)
        return y.reshape(*shape[:-1], y.shape[-1])

    def forward(self, x, reduce_dtype=None):
        rows = x.numel() // x.shape[-1]
        if reduce_dtype is None:
            reduce_dtype = torch.float32 if rows <= 64 else x.dtype
        if self._reduce is None:
            self._reduce = PartialSumReducer(self.group)       # one-shot exchange for decode-sized messages, ring otherwise
        if reduce_dtype == torch.float32:
            # decode-sized messages: the Linear writes its fp32 partial sums STRAIGHT into the reducer's peer-mapped buffer (no staging
            # copy, one launch less per all-reduce); None when this message shape rides the ring
            buf = self._reduce.buffer((*x.shape[:-1], self.weight.shape[0]), torch.float32, x.device)
            if buf is not None:
                self.partial(x, out_dtype=torch.float32, out=buf)
                return self._reduce.reduce_buffer(buf).to(x.dtype)
        y = self.partial(x, out_dtype=torch.float32)
        if reduce_dtype != torch.float32:
            y = y.to(reduce_dtype)
        return self._reduce(y).to(x.dtype)


# ------------------------------------------------------------------------------------------------ small-message all-reduce
class _ShapeLike:
    """(shape, dtype, device) of a message, for allocating its symmetric buffer before any tensor of that shape exists"""

    def __init__(self, shape, dtype, device):
        self.shape, self.dtype, self.device = tuple(shape), dtype, device


class PartialSumReducer:
    """All-reduce of the row-parallel Linears' fp32 partial sums.

    Decode-sized messages ([rows <= 64, hidden] fp32: 32 KB at Llama-2-70B) are latency-bound: a ring all-reduce over 8 ranks is 14
    dependent hops of point-to-point xGMI traffic (10-20 us each call, 160 calls per step -- as long as the step's compute).  They go
    through a ONE-SHOT exchange instead: every rank keeps its partial in a symmetric (peer-mapped) buffer, signals, and sums the 8
    buffers itself in rank order -- one xGMI read per peer, no hop chain, deterministic.  The peer mapping, the signal pads and the
    kernel are torch's (torch.distributed._symmetric_memory / symm_mem.one_shot_all_reduce: device memory + collectives are plumbing
    here, the product is the Linear kernel that fills the buffer).  Everything else -- prefill-sized messages, process groups without
    peer access, the gloo groups of the CPU/one-GPU tests -- uses dist.all_reduce (RCCL ring over xGMI on a node).
    UNMEASURED on hardware: no multi-GPU box was available to this repo's builder; the single-GPU tests drive both branches' control
    flow (tests/test_dist_gloo.py)."""

    ONE_SHOT_MAX_BYTES = 256 * 1024

    def __init__(self, group=None, world=None):
        self.group = group
        self.world = world           # TP degree of the owner (None: the group's size); 1 = nothing to reduce
        self._bufs = {}              # (shape, dtype) -> symmetric buffer, or None = this message shape stays on the ring
        self._symm = None            # None = not probed yet, False = unavailable (on ANY rank: the decision is collective)
        self.calls = {"one_shot": 0, "ring": 0}      # which transport each call took (bench lines report it)
        self.why_ring = None         # first reason the one-shot path was ruled out, for the bench line

    # Every decision below is made by ALL ranks together: a rank whose local setup failed and went to the ring while its peers sat in
    # the symmetric-memory rendezvous would hang the job.  A local attempt is followed by an all-reduce(MIN) of an "ok" flag over the
    # ring, and the one-shot path is used only if every rank said yes.
    def _all_agree(self, ok, device):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item())

    def _local_enable(self):
        """this rank's attempt to switch symmetric memory on for the group; raises when the build / topology cannot"""
        import torch.distributed._symmetric_memory as sm
        g = self.group or dist.group.WORLD
        sm.enable_symm_mem_for_group(g.group_name)
        self._sm, self._gname = sm, g.group_name

    def _local_alloc(self, y):
        return self._sm.empty(tuple(y.shape), dtype=y.dtype, device=y.device)

    def _rendezvous(self, buf):
        self._sm.rendezvous(buf, self._gname)

    def _one_shot(self, buf):
        return torch.ops.symm_mem.one_shot_all_reduce(buf, "sum", self._gname)

    def _capable(self, device):
        """job-wide facts (same answer on every rank): RCCL process group and device memory"""
        return dist.get_backend(self.group) == "nccl" and device.type == "cuda"

    def _symm_ok(self, device):
        if self._symm is None:
            if not self._capable(device):
                self._symm = False                   # same answer on every rank (backend and device type are job-wide): no vote needed
                self.why_ring = f"backend {dist.get_backend(self.group)} / device {device.type}: no peer-mapped memory"
                return False
            ok, err = True, None
            try:
                self._local_enable()
            except Exception as e:                   # no peer access / unsupported build
                ok, err = False, f"{type(e).__name__}: {e}"
            self._symm = self._all_agree(ok, device)
            if not self._symm:
                self.why_ring = err or "symmetric-memory setup failed on another rank"
        return self._symm

    def _buffer_for(self, shape, dtype, device):
        key = (tuple(shape), dtype)
        if key in self._bufs:
            return self._bufs[key]
        buf, ok, err = None, True, None
        try:
            buf = self._local_alloc(_ShapeLike(shape, dtype, device))     # local; the collective rendezvous comes only after everybody has one
        except Exception as e:
            ok, err = False, f"{type(e).__name__}: {e}"
        if self._all_agree(ok, device):
            try:
                self._rendezvous(buf)
            except Exception as e:
                ok, err = False, f"{type(e).__name__}: {e}"
            ok = self._all_agree(ok, device)
        else:
            ok = False
        if not ok:
            buf = None
            self.why_ring = self.why_ring or err or "symmetric buffer setup failed on another rank"
        self._bufs[key] = buf
        return buf

    def _active(self):
        return not (self.world == 1 or not dist.is_initialized() or dist.get_world_size(self.group) == 1)

    def buffer(self, shape, dtype, device):
        """The peer-mapped buffer a producer may write its partial sums INTO (then `reduce_buffer`), or None when a message of this shape
        goes over the ring / nothing is reduced.  The decision is collective and cached per (shape, dtype): every rank gets the same answer."""
        key = (tuple(shape), dtype)
        if key in self._bufs:                    # (hot path: 160 calls per decode step)
            return self._bufs[key]
        nbytes = dtype.itemsize
        for d in shape:
            nbytes *= int(d)
        if self._active() and nbytes <= self.ONE_SHOT_MAX_BYTES and self._symm_ok(device):
            return self._buffer_for(shape, dtype, device)
        return None

    def reduce_buffer(self, buf):
        """sum over ranks of a buffer handed out by `buffer` and filled by the producing kernel (stream order: same stream)"""
        self.calls["one_shot"] += 1
        return self._one_shot(buf)

    def __call__(self, y):
        """y: fp32 (or 16-bit) partial sums, identical shape on every rank; returns the sum (may alias y)"""
        if not self._active():
            return y
        buf = self.buffer(y.shape, y.dtype, y.device)
        if buf is not None:              # a producer that could not write into the buffer itself (16-bit partials, non-Linear callers)
            buf.copy_(y)
            return self.reduce_buffer(buf)
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        self.calls["ring"] += 1
        return y

    def report(self):
        """what the bench line says about the exchange: the transport of each message class and, if the ring carried the small messages,
        why"""
        return {"one_shot_calls": self.calls["one_shot"], "ring_calls": self.calls["ring"],
                "one_shot_available": bool(self._symm), "one_shot_max_bytes": self.ONE_SHOT_MAX_BYTES,
                "ring_reason": self.why_ring}


# ------------------------------------------------------------------------------------------------ Llama-2-70B shaped TP decoder (bench)
LLAMA_70B = (8192, 28672, 80, 64, 8, 32000)      # hidden, intermediate, layers, heads, kv heads, vocab


def synth_full(n_out, n_in, device, dtype, gen):
    """full synthetic (base weight, mask, coeff) of one projection (SURVEY.md section 8d statistics)"""
    w = (torch.randn(n_out, n_in, device=device, generator=gen) * 0.02).to(dtype)
    fine = (w.float() + torch.randn(n_out, n_in, device=device, generator=gen) * 5e-4).to(dtype)
    mask, coeff = binarize(w, fine)
    return w, mask, coeff


def _col_shard(full, rank, world):
    w, m, c = full
    n = w.shape[0] // world
    assert w.shape[0] % world == 0
    return w[rank * n:(rank + 1) * n].contiguous(), shard_mask_columns(m, rank, world), c


def _row_shard(full, rank, world):
    w, m, c = full
    k = w.shape[1] // world
    assert w.shape[1] % world == 0 and k % 32 == 0, "K / world must be a multiple of 32"
    return w[:, rank * k:(rank + 1) * k].contiguous(), shard_mask_rows(m, rank, world), c


class TPDecoderLayer(nn.Module):
    """One decoder layer's rank-local shards, at the serving loop's standard: q|k|v and gate|up are ONE FusedDeltaLinear each (the k / v
    shards of Llama-2-70B are 128 columns wide -- far below what a launch of their own can stream), the glue runs on the HIP kernels of
    serving_ops at decode (RMSNorm, RoPE + KV append + attention in one launch, SwiGLU in gate|up's epilogue), and the row-parallel
    o / down shards leave fp32 partials that PartialSumReducer exchanges."""

    def __init__(self, cfg, shards, norms, device, dtype, world, reducer, eps=1e-5):
        super().__init__()
        from .serving_loop import FusedDeltaLinear
        hid, inter, _, heads, kvh, _ = cfg
        self.hd = hid // heads
        self.h_loc, self.kv_loc = heads // world, kvh // world
        self.inter_loc = inter // world
        self.eps = eps
        one = lambda t: t.reshape(1)
        q, k, v, o, gate, up, down = shards
        self.qkv = FusedDeltaLinear([q[0], k[0], v[0]], [q[1][None], k[1][None], v[1][None]], [one(q[2]), one(k[2]), one(v[2])])
        self.gate_up = FusedDeltaLinear([gate[0], up[0]], [gate[1][None], up[1][None]], [one(gate[2]), one(up[2])],
                                        interleave8=(self.inter_loc % 8 == 0))
        self.o = FusedDeltaLinear([o[0]], [o[1][None]], [one(o[2])])
        self.down = FusedDeltaLinear([down[0]], [down[1][None]], [one(down[2])])
        self.register_buffer("n1", norms[0].reshape(1, hid))
        self.register_buffer("n2", norms[1].reshape(1, hid))
        self.reduce = reducer

    def _norm(self, x, w):
        if x.shape[1] <= 16 and x.shape[-1] % 8 == 0:
            return ops.rmsnorm_tenant(x if x.is_contiguous() else x.contiguous(), w, self.eps)
        return F.rms_norm(x, (x.shape[-1],), w[0], self.eps)

    def _row_parallel_sum(self, lin, x):
        """all-reduce(x_r . W_r^T + coeff * (x_r . S_r)) in fp32, WITHOUT the residual add (decode: the caller folds the add and the next RMSNorm into
        one launch, serving_ops.add_rmsnorm)"""
        buf = self.reduce.buffer((*x.shape[:-1], lin.weight.shape[0]), torch.float32, x.device)
        if buf is not None:
            lin(x, out_dtype=torch.float32, out=buf)
            return self.reduce.reduce_buffer(buf)
        return self.reduce(lin(x, out_dtype=torch.float32))

    def _row_parallel(self, lin, x, residual):
        """residual + all-reduce(x_r . W_r^T + coeff * (x_r . S_r)): fp32 partials for decode-sized messages, the activation dtype for
        prefill-sized ones (half the bytes on the wire; one extra rounding per partial)"""
        rows = x.numel() // x.shape[-1]
        if rows <= 64:
            # decode-sized message: the Linear writes its fp32 partial sums straight into the reducer's peer-mapped buffer (no staging copy:
            # one launch less per all-reduce, 160 per decode step); None when this message shape rides the ring
            buf = self.reduce.buffer((*x.shape[:-1], lin.weight.shape[0]), torch.float32, x.device)
            if buf is not None:
                lin(x, out_dtype=torch.float32, out=buf)
                return residual + self.reduce.reduce_buffer(buf).to(residual.dtype)
        y = lin(x, out_dtype=torch.float32)
        if rows > 64:
            y = y.to(x.dtype)
        y = self.reduce(y)
        return residual + y.to(residual.dtype)

    def prefill_kernel_ok(self, B, S, dtype, device):
        """True when `forward(..., from_zero=True)` will take the flash-style prefill attention kernel for a [B, S] prompt: the gate of
        that branch evaluated on an (uninitialised) tensor of the fused projection's output shape.  TPDecoder.forward asks ONCE per call,
        so the dense [S, Lc] mask is either built once or not at all."""
        hd = self.hd
        nq, nk = self.h_loc * hd, self.kv_loc * hd
        if not (S > 1 and hd == 128 and S % 64 == 0 and not self.qkv.interleave8):
            return False
        qkv = torch.empty(B, S, nq + 2 * nk, device=device, dtype=dtype)
        return bool(ops.prefill_attention_supported(qkv[..., :nq].view(B, S, self.h_loc, hd), qkv[..., nq:nq + nk].view(B, S, self.kv_loc, hd),
                                                    qkv[..., nq + nk:].view(B, S, self.kv_loc, hd)))

    def forward(self, x, cos, sin, cache, pos_idx, attn_mask, from_zero=False, pending=None):
        """x [1, S, hidden] (replicated); cache = (k [1, kv_loc, L, hd], v, valid [1, L]); pos_idx [S] device positions;
        from_zero: the caller knows pos_idx == arange(S) (a prompt prefilled from position 0 into an empty cache).
        Returns (x, pending): at decode `pending` = this layer's reduced down-projection sums (fp32), which the NEXT layer (or the final norm) adds to x
        together with its RMSNorm; otherwise None and x is complete."""
        B, S, hid = x.shape
        hd = self.hd
        # decode: the residual add after each all-reduce and the RMSNorm that follows it are ONE launch (round 6: cast + add + norm were three ~5-us
        # launches around Linears of 10-25 us); `pending` = the previous layer's reduced down-projection sums, not yet added to x
        fuse = S == 1 and B * S <= 16 and hid % 8 == 0 and hid <= 8192 and x.is_contiguous()
        if pending is not None:
            x, h1 = ops.add_rmsnorm(x, pending, self.n1, self.eps)
        else:
            h1 = self._norm(x, self.n1)
        qkv = self.qkv(h1)
        ck, cv, valid = cache
        nq, nk = self.h_loc * hd, self.kv_loc * hd
        if S == 1 and ops.decode_attention_supported(self.h_loc, self.kv_loc, hd):
            a = ops.decode_attention(qkv, cos, sin, ck, cv, valid, pos_idx, self.h_loc, self.kv_loc)       # RoPE + append + attention
        elif (from_zero and S > 1 and hd == 128 and S % 64 == 0 and qkv.is_contiguous() and not self.qkv.interleave8 and
              ops.prefill_attention_supported(qkv[..., :nq].view(B, S, self.h_loc, hd), qkv[..., nq:nq + nk].view(B, S, self.kv_loc, hd),
                                              qkv[..., nq + nk:].view(B, S, self.kv_loc, hd))):
            # whole-prompt causal attention straight on the three slices of the fused projection output (as serving_loop does): in-place
            # RoPE of the q and k heads in one launch, flash-style kernel over S keys -- no [S, Lc] mask, no SDPA over the cache length
            k4, v4 = qkv[..., nq:nq + nk].view(B, S, self.kv_loc, hd), qkv[..., nq + nk:].view(B, S, self.kv_loc, hd)
            if ck.is_contiguous() and cv.is_contiguous() and ck.shape[0] == B:
                ops.rope_kv_append_(qkv, cos, sin, ck, cv, self.h_loc, self.kv_loc, 0)      # RoPE + both cache writes: one launch (round 6)
                a = ops.prefill_attention(qkv[..., :nq].view(B, S, self.h_loc, hd), k4, v4, causal=True)
            else:
                ops.rope_(qkv[..., :nq + nk], cos, sin, self.h_loc + self.kv_loc, S, 0)
                a = ops.prefill_attention(qkv[..., :nq].view(B, S, self.h_loc, hd), k4, v4, causal=True)
                ck[:, :, :S] = k4.transpose(1, 2)
                cv[:, :, :S] = v4.transpose(1, 2)
        else:
            from .serving_loop import _rope
            q, k, v = self.qkv.split(qkv)
            c, s_ = cos[pos_idx], sin[pos_idx]
            q = _rope(q.view(B, S, self.h_loc, hd).transpose(1, 2), c, s_)
            k = _rope(k.view(B, S, self.kv_loc, hd).transpose(1, 2), c, s_)
            v = v.view(B, S, self.kv_loc, hd).transpose(1, 2)
            ck.index_copy_(2, pos_idx, k)
            cv.index_copy_(2, pos_idx, v)
            if attn_mask is None:                     # (the caller expected the prefill-attention kernel to take this chunk)
                keypos = torch.arange(ck.shape[2], device=x.device)
                attn_mask = (keypos[None, :] <= pos_idx[:, None])[None, None] & valid[:, None, None, :]
            a = F.scaled_dot_product_attention(q, ck, cv, attn_mask=attn_mask, enable_gqa=(self.kv_loc != self.h_loc))
            a = a.transpose(1, 2).reshape(B, S, self.h_loc * hd)
        if fuse:
            x, h = ops.add_rmsnorm(x, self._row_parallel_sum(self.o, a), self.n2, self.eps)
        else:
            x = self._row_parallel(self.o, a, x)
            h = self._norm(x, self.n2)
        if S == 1 and self.gate_up.interleave8 and self.gate_up._decode_ok(h):
            act = self.gate_up.forward_fused(h, None, self.eps, swiglu=True)                                # gate|up -> SwiGLU: one launch
        else:
            gu = self.gate_up(h)
            if self.gate_up.interleave8 and gu.shape[-1] % 16 == 0:
                act = ops.swiglu_interleaved8(gu)
            else:
                g, u = self.gate_up.split(gu)
                act = F.silu(g) * u
        if fuse:
            return x, self._row_parallel_sum(self.down, act)
        return self._row_parallel(self.down, act, x), None


class TPDecoder(nn.Module):
    """Tensor-parallel decoder over `world` ranks.  `full=None`: every rank synthesises ITS OWN shards (no rank ever holds a full 70B
    matrix).  `full=[...]` (tests): per-layer lists of the 7 full (weight, mask, coeff) triples, sharded here -- identical on every
    rank, so that the TP result can be compared with the single-rank one."""

    def __init__(self, cfg, device, dtype, rank, world, layers=None, seed=0, full=None, group=None, max_len=4096):
        super().__init__()
        from .serving_loop import _rope_tables
        hid, inter, nl, heads, kvh, vocab = cfg
        assert heads % world == 0 and kvh % world == 0, "query and kv heads must both divide by the TP degree (no kv-head replication)"
        gen = torch.Generator(device=device).manual_seed(seed + 1000 * rank)       # shards differ per rank
        rep = torch.Generator(device=device).manual_seed(seed)                      # replicated tensors agree on every rank
        self.cfg, self.dtype, self.dev, self.rank, self.world = cfg, dtype, device, rank, world
        self.reducer = PartialSumReducer(group, world)
        hd = hid // heads
        self.hd = hd
        shapes = [("col", heads * hd, hid), ("col", kvh * hd, hid), ("col", kvh * hd, hid), ("row", hid, heads * hd),
                  ("col", inter, hid), ("col", inter, hid), ("row", hid, inter)]
        ls = []
        for li in range(layers or nl):
            shards = []
            for pi, (split, n_out, n_in) in enumerate(shapes):
                if full is not None:
                    shards.append(_col_shard(full[li][pi], rank, world) if split == "col" else _row_shard(full[li][pi], rank, world))
                else:
                    o_, i_ = (n_out // world, n_in) if split == "col" else (n_out, n_in // world)
                    shards.append(synth_full(o_, i_, device, dtype, gen))
            norms = (torch.ones(hid, device=device, dtype=dtype), torch.ones(hid, device=device, dtype=dtype))
            ls.append(TPDecoderLayer(cfg, shards, norms, device, dtype, world, self.reducer))
        self.layers = nn.ModuleList(ls)
        self.embed = (torch.randn(vocab, hid, device=device, generator=rep) * 0.02).to(dtype)
        self.lm_head = (torch.randn(vocab, hid, device=device, generator=rep) * 0.02).to(dtype)
        self.norm = torch.ones(1, hid, device=device, dtype=dtype)
        self.max_len = max_len
        self.cos, self.sin = _rope_tables(max_len, hd, device, dtype)
        self._capture_stream = None

    def new_cache(self, length):
        l0 = self.layers[0]
        mk = lambda: torch.zeros(1, l0.kv_loc, length, l0.hd, device=self.dev, dtype=self.dtype)
        return {"k": [mk() for _ in self.layers], "v": [mk() for _ in self.layers],
                "valid": torch.zeros(1, length, dtype=torch.bool, device=self.dev)}

    @torch.no_grad()
    def forward(self, ids, pos_idx, cache, from_zero=False):
        """ids [1, S]; pos_idx [S] int64 device tensor (cache positions of these tokens; S > 1: a prefill chunk starting anywhere).
        from_zero=True: the caller asserts pos_idx == arange(S) on an empty cache (a whole prompt) -- the layers then run RoPE + the
        flash-style prefill attention kernel instead of SDPA over a dense [S, cache length] mask.
        Returns the logits of the last position [1, 1, vocab] (replicated on every rank)."""
        B, S = ids.shape
        Lc = cache["valid"].shape[1]
        if from_zero and os.environ.get("BD_DEBUG_CHECKS"):
            # from_zero is a caller assertion; with it wrong, attention would silently see only keys [:S].  Debug builds check it (a sync).
            assert int(pos_idx[0]) == 0 and int(pos_idx[-1]) == S - 1 and not bool(cache["valid"].any()), \
                "from_zero=True needs pos_idx == arange(S) on an EMPTY cache"
        cache["valid"].index_fill_(1, pos_idx, True)
        mask = None
        # the attention path is decided ONCE here (layer 0's gate: every layer has the same shapes), not re-derived per layer
        from_zero = bool(from_zero and self.layers[0].prefill_kernel_ok(B, S, self.dtype, self.dev))
        if not from_zero:
            # query i (at position pos_idx[i]) sees the valid keys at positions <= pos_idx[i]: correct for a chunk that starts at pos > 0 too
            keypos = torch.arange(Lc, device=self.dev)
            mask = (keypos[None, :] <= pos_idx[:, None])[None, None] & cache["valid"][:, None, None, :]
        x = self.embed[ids]
        pending = None
        for li, layer in enumerate(self.layers):
            x, pending = layer(x, self.cos, self.sin, (cache["k"][li], cache["v"][li], cache["valid"]), pos_idx, mask, from_zero=from_zero,
                               pending=pending)
        if pending is not None:                      # decode: the last layer's down-projection sums + the final norm in one launch
            _, last = ops.add_rmsnorm(x, pending, self.norm, 1e-5)
            return last @ self.lm_head.T
        last = x[:, -1:, :]
        last = ops.rmsnorm_tenant(last.contiguous(), self.norm, 1e-5) if last.shape[-1] % 8 == 0 else F.rms_norm(last, (last.shape[-1],), self.norm[0], 1e-5)
        return last @ self.lm_head.T

    def decode_runner(self, tok, pos, cache, use_graph=True):
        """callable running ONE decode step on static buffers (tok [1,1], pos [1] device tensors, advanced by the step; logits are
        written to the returned buffer).  With use_graph the per-rank step -- kernels AND collectives -- is captured once as a
        hipGraph and replayed (RCCL and symmetric-memory collectives are capturable; gloo is not: those groups run eagerly)."""
        out = torch.empty(1, 1, self.cfg[5], device=self.dev, dtype=self.dtype)

        def step():
            out.copy_(self.forward(tok, pos, cache))
            pos.add_(1)
        capturable = use_graph and (self.world == 1 or not dist.is_initialized() or dist.get_backend() == "nccl")
        if not capturable:
            return step, out
        try:
            if self._capture_stream is None:
                self._capture_stream = torch.cuda.Stream(device=self.dev)
            side = self._capture_stream
            snap = (pos.clone(), cache["valid"].clone())
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                step()                                   # warm-up on the capture stream: scratch, symmetric buffers, rendezvous
            torch.cuda.current_stream(self.dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                step()
            torch.cuda.synchronize(self.dev)
            pos.copy_(snap[0]); cache["valid"].copy_(snap[1])
            return g.replay, out
        except Exception:                                # capture refused (driver / collective): the eager step is always valid
            torch.cuda.synchronize(self.dev)
            return step, out

    def linear_bytes_per_rank(self):
        from .serving_loop import FusedDeltaLinear
        return sum(m.linear_bytes() for m in self.modules() if isinstance(m, FusedDeltaLinear))


def bench_tp70b(args, dev, rank, world, timer):
    """bench.py --workload tp70b: Llama-2-70B shapes over `world` ranks (8 on a full node), prefill of one 2048-token sequence and
    decode steps; value = prefill tokens/s of the WHOLE job (strong scaling: the ranks share one sequence)."""
    from .dist import timed_region, runtime_info
    dec = TPDecoder(LLAMA_70B, dev, torch.bfloat16, rank, world, layers=args.layers, seed=77, max_len=args.seq + args.kv_len + 64)
    ids = torch.randint(0, 32000, (1, args.seq), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    cache = dec.new_cache(max(args.seq, args.kv_len) + 64)
    pos_prefill = torch.arange(args.seq, device=dev)

    def step():
        cache["valid"].zero_()
        return dec(ids, pos_prefill, cache, from_zero=True)
    for _ in range(args.warmup):
        step()
    timer.reset()
    timer.enabled = False
    dt = timed_region(step, args.steps, device_sync=torch.cuda.synchronize)          # `value`: no per-launch events inside the region
    timer.enabled = True
    dt_ev = timed_region(step, args.steps, device_sync=torch.cuda.synchronize)       # roofline pass: one HIP event pair per fused launch
    timer.enabled = False
    n_launch, k_ms, k_flops, _ = timer.summary()
    # decode: one token per step on a KV cache of args.kv_len, the per-rank step (kernels + collectives) replayed as a hipGraph
    cache["valid"].zero_()
    dec(ids[:, :args.kv_len], torch.arange(args.kv_len, device=dev), cache)
    tok = ids[:, :1].clone()
    pos = torch.tensor([args.kv_len], device=dev)
    run, _ = dec.decode_runner(tok, pos, cache, use_graph=True)

    def dstep():
        pos.fill_(args.kv_len)
        run()
    for _ in range(3):
        dstep()
    ddt = timed_region(dstep, 20, device_sync=torch.cuda.synchronize)
    nl = len(dec.layers)
    msg_prefill = args.seq * 8192 * 2
    return {
        "metric": "Llama-2-70B base + chat 1-bit delta, tensor parallel, prefill tokens/s (BASELINE.json configs[3])",
        "value": args.seq * args.steps / dt, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": f"llama-2-70b shapes, TP={world}, prefill seq {args.seq}; {nl} layers x (fused q|k|v and gate|up column-parallel "
                               "shards + o / down row-parallel shards), all-reduce after o_proj and down_proj: RCCL ring at prefill, "
                               "one-shot symmetric-memory exchange for the 32 KB decode messages",
                   "seq_len": args.seq, "parallelism": f"tp{world}", "valid": args.layers is None and world == 8,
                   "all_reduce_messages_per_step": 2 * nl, "all_reduce_bytes_each_prefill": msg_prefill,
                   "all_reduce_bytes_each_decode": 8192 * 4,
                   "measured_on_hardware": "never on > 1 GPU by the builder (no multi-GPU box); world = 1 / 2-rank gloo runs only"},
        "roofline": {"bound": "mfma", "achieved": k_flops / k_ms * 1e-9 if k_ms > 0 else None, "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": (k_flops / k_ms * 1e-9 / 2500.0) if k_ms > 0 else None, "traffic": None, "launches": n_launch,
                     "share_of_step_time": (k_ms / 1e3) / dt_ev if dt_ev > 0 else None},
        "decode": {"ms_per_step": ddt / 20 * 1e3, "kv_len": args.kv_len, "linear_bytes_per_rank": dec.linear_bytes_per_rank(),
                   "linear_gbs_per_rank_if_only_linears": dec.linear_bytes_per_rank() / (ddt / 20) * 1e-9},
        # which transport the partial-sum exchange actually took on this run (prefill messages: ring; decode messages: one-shot when
        # every rank could set symmetric memory up, else ring + the first reason it was ruled out)
        "all_reduce": dec.reducer.report() if hasattr(dec, "reducer") else None,
        **runtime_info(),
    }


def bench_tp70b_shard(dev, timer, layers=None, seq=2048, kv_len=512, steps=3, world=8):
    """BASELINE.json configs[3]'s per-GPU unit on ONE GPU (bench.py's default run; no 8-GPU node has ever been available to this repo): rank 0's
    Llama-2-70B TP = `world` shards, every layer, with the partial-sum exchange after o / down STUBBED OUT -- no process group exists, so
    PartialSumReducer is inactive and each row-parallel Linear's partial sums are used as if they were the reduced result.  What is measured is
    therefore the per-rank compute of a prefill-2048 step and of a decode step (hipGraph replay); what is NOT is the 2 x 80 all-reduces
    (DESIGN.md section 6 prices them as projections).  Weights are this rank's own synthetic shards (no rank ever holds a full matrix)."""
    from .dist import timed_region
    assert not dist.is_initialized() or dist.get_world_size() == 1, "tp70b_shard is the single-process, exchange-stubbed leg"
    dec = TPDecoder(LLAMA_70B, dev, torch.bfloat16, 0, world, layers=layers, seed=77, max_len=seq + kv_len + 64)
    assert not dec.reducer._active()
    nl = len(dec.layers)
    ids = torch.randint(0, 32000, (1, seq), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    cache = dec.new_cache(max(seq, kv_len) + 64)
    pos_prefill = torch.arange(seq, device=dev)

    def step():
        cache["valid"].zero_()
        return dec(ids, pos_prefill, cache, from_zero=True)
    step()
    timer.reset()
    timer.enabled = False
    dt = timed_region(step, steps, device_sync=torch.cuda.synchronize)
    timer.enabled = True
    timed_region(step, steps, device_sync=torch.cuda.synchronize)           # roofline pass: one HIP event pair per fused launch
    timer.enabled = False
    n_launch, k_ms, k_flops, _ = timer.summary()
    cache["valid"].zero_()
    dec(ids[:, :kv_len], torch.arange(kv_len, device=dev), cache)
    tok = ids[:, :1].clone()
    pos = torch.tensor([kv_len], device=dev)
    run, _ = dec.decode_runner(tok, pos, cache, use_graph=True)

    def dstep():
        pos.fill_(kv_len)
        run()
    for _ in range(5):
        dstep()
    ddt = timed_region(dstep, 20, device_sync=torch.cuda.synchronize) / 20
    lin_bytes = dec.linear_bytes_per_rank()
    head_bytes = dec.lm_head.numel() * dec.lm_head.element_size()
    out = {"what": f"configs[3] per-GPU unit: rank 0's Llama-2-70B TP={world} shards, {nl} layers, on one GPU",
           "exchange": "STUBBED OUT (single process: the 2 all-reduces per layer are not executed; compute only)",
           "layers": nl, "valid": layers is None, "world_modelled": world,
           "prefill_ms": dt / steps * 1e3, "prefill_tokens_per_s_compute_only": seq / (dt / steps),
           "prefill_fused_frac_of_mfma_peak": (k_flops / k_ms * 1e-9 / 2500.0) if k_ms > 0 else None,
           "prefill_fused_launches": n_launch, "prefill_fused_ms_total": k_ms,
           "decode_ms": ddt * 1e3, "decode_kv_len": kv_len, "linear_bytes_per_rank": lin_bytes, "lm_head_bytes": head_bytes,
           "decode_frac_of_hbm_peak": (lin_bytes + head_bytes) / ddt * 1e-9 / 8000.0,
           "all_reduce_messages_per_step_not_executed": 2 * nl}
    del dec, cache
    torch.cuda.empty_cache()
    return out
