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
import time

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .binary_gemm_kernel import binary_linear
from .diff import binarize
from .dist import shard_mask_columns, shard_mask_rows


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

    @classmethod
    def from_full(cls, weight, mask, coeff, rank=None, world=None, group=None):
        rank, world = _rank() if rank is None else rank, _world() if world is None else world
        k = weight.shape[1] // world
        assert weight.shape[1] % world == 0 and k % 32 == 0, "K / world must be a multiple of 32"
        return cls(weight[:, rank * k:(rank + 1) * k], shard_mask_rows(mask, rank, world), coeff, group)

    def partial(self, x, out_dtype=torch.float32):
        shape = x.shape
        y = binary_linear(x.reshape(1, -1, shape[-1]), self.weight, self.mask.unsqueeze(0), self.coeff, out_dtype=out_dtype)
        return y.reshape(*shape[:-1], y.shape[-1])

    def forward(self, x, reduce_dtype=None):
        rows = x.numel() // x.shape[-1]
        if reduce_dtype is None:
            reduce_dtype = torch.float32 if rows <= 64 else x.dtype
        y = self.partial(x, out_dtype=torch.float32)
        if reduce_dtype != torch.float32:
            y = y.to(reduce_dtype)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        return y.to(x.dtype)


# ------------------------------------------------------------------------------------------------ Llama-2-70B shaped TP decoder (bench)
LLAMA_70B = (8192, 28672, 80, 64, 8, 32000)      # hidden, intermediate, layers, heads, kv heads, vocab


def _synth_shard(n_out, n_in, device, dtype, gen, split, rank, world):
    """Rank-local shard of a synthetic (base, fine-tune) pair WITHOUT materialising the full matrix on every rank: the shard itself is
    generated (statistics of SURVEY.md section 8d); coeff is taken from the shard (same distribution as the whole matrix)."""
    if split == "col":
        n_out //= world
    else:
        n_in //= world
    w = (torch.randn(n_out, n_in, device=device, generator=gen) * 0.02).to(dtype)
    fine = (w.float() + torch.randn(n_out, n_in, device=device, generator=gen) * 5e-4).to(dtype)
    mask, coeff = binarize(w, fine)
    return w, mask, coeff


class TPDecoderLayer(nn.Module):
    def __init__(self, cfg, device, dtype, gen, rank, world):
        super().__init__()
        hid, inter, _, heads, kvh, _ = cfg
        self.hd = hid // heads
        self.h_loc, self.kv_loc = heads // world, max(kvh // world, 1)
        hd = self.hd
        mk_c = lambda o, i: ColumnParallelBinaryDiff(*_synth_shard(o, i, device, dtype, gen, "col", rank, world))
        mk_r = lambda o, i: RowParallelBinaryDiff(*_synth_shard(o, i, device, dtype, gen, "row", rank, world))
        self.q_proj, self.k_proj, self.v_proj = mk_c(heads * hd, hid), mk_c(kvh * hd, hid), mk_c(kvh * hd, hid)
        self.o_proj = mk_r(hid, heads * hd)
        self.gate_proj, self.up_proj = mk_c(inter, hid), mk_c(inter, hid)
        self.down_proj = mk_r(hid, inter)
        self.n1 = torch.ones(hid, device=device, dtype=dtype)
        self.n2 = torch.ones(hid, device=device, dtype=dtype)

    def forward(self, x, cos, sin, kv):
        from .serving_loop import _rope
        B, S, hid = x.shape
        h = F.rms_norm(x, (hid,), self.n1, 1e-5)
        q = _rope(self.q_proj(h).view(B, S, self.h_loc, self.hd).transpose(1, 2), cos, sin)
        k = _rope(self.k_proj(h).view(B, S, self.kv_loc, self.hd).transpose(1, 2), cos, sin)
        v = self.v_proj(h).view(B, S, self.kv_loc, self.hd).transpose(1, 2)
        if kv is not None:
            pos = kv[2]
            kv[0][:, :, pos:pos + S] = k
            kv[1][:, :, pos:pos + S] = v
            kv[2] = pos + S
            k, v = kv[0][:, :, :pos + S], kv[1][:, :, :pos + S]
        a = F.scaled_dot_product_attention(q, k, v, is_causal=(S > 1 and k.shape[2] == S), enable_gqa=(self.kv_loc != self.h_loc))
        x = x + self.o_proj(a.transpose(1, 2).reshape(B, S, self.h_loc * self.hd))
        h = F.rms_norm(x, (hid,), self.n2, 1e-5)
        return x + self.down_proj(F.silu(self.gate_proj(h)) * self.up_proj(h))


class TPDecoder(nn.Module):
    def __init__(self, cfg, device, dtype, rank, world, layers=None, seed=0):
        super().__init__()
        from .serving_loop import _rope_tables
        hid, inter, nl, heads, kvh, vocab = cfg
        assert heads % world == 0 and (kvh % world == 0 or world % kvh == 0)
        gen = torch.Generator(device=device).manual_seed(seed + 1000 * rank)       # shards differ per rank
        rep = torch.Generator(device=device).manual_seed(seed)                      # replicated tensors agree on every rank
        self.cfg, self.dtype, self.dev = cfg, dtype, device
        self.layers = nn.ModuleList([TPDecoderLayer(cfg, device, dtype, gen, rank, world) for _ in range(layers or nl)])
        self.embed = (torch.randn(vocab, hid, device=device, generator=rep) * 0.02).to(dtype)
        self.lm_head = (torch.randn(vocab, hid, device=device, generator=rep) * 0.02).to(dtype)
        self.norm = torch.ones(hid, device=device, dtype=dtype)
        self.cos, self.sin = _rope_tables(4096, hid // heads, device, dtype)

    def new_cache(self, batch, length):
        l0 = self.layers[0]
        mk = lambda: torch.zeros(batch, l0.kv_loc, length, l0.hd, device=self.dev, dtype=self.dtype)
        return [[mk(), mk(), 0] for _ in self.layers]

    @torch.no_grad()
    def forward(self, ids, pos0=0, cache=None):
        B, S = ids.shape
        cos, sin = self.cos[pos0:pos0 + S], self.sin[pos0:pos0 + S]
        x = self.embed[ids]
        for i, layer in enumerate(self.layers):
            x = layer(x, cos, sin, None if cache is None else cache[i])
        return F.rms_norm(x[:, -1:], (x.shape[-1],), self.norm, 1e-5) @ self.lm_head.T

    def linear_bytes_per_rank(self):
        return sum((m.weight.numel() * 2 + m.mask.numel() * 4) for m in self.modules()
                   if isinstance(m, (ColumnParallelBinaryDiff, RowParallelBinaryDiff)))


def bench_tp70b(args, dev, rank, world, timer):
    """bench.py --workload tp70b: Llama-2-70B shapes over `world` ranks (8 on a full node), prefill of one 2048-token sequence and
    decode steps; value = prefill tokens/s of the WHOLE job (strong scaling: the ranks share one sequence)."""
    from .dist import timed_region
    dec = TPDecoder(LLAMA_70B, dev, torch.bfloat16, rank, world, layers=args.layers, seed=77)
    ids = torch.randint(0, 32000, (1, args.seq), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    step = lambda: dec(ids)
    for _ in range(args.warmup):
        step()
    timer.reset()
    timer.enabled = True
    dt = timed_region(step, args.steps, device_sync=torch.cuda.synchronize)
    timer.enabled = False
    n_launch, k_ms, k_flops, _ = timer.summary()
    # decode: one token with a KV cache of args.kv_len
    cache = dec.new_cache(1, args.kv_len + 64)
    dec(ids[:, :args.kv_len], 0, cache)
    tok = ids[:, :1]
    pos = [args.kv_len]

    def dstep():
        for c in cache:
            c[2] = args.kv_len
        dec(tok, args.kv_len, cache)
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
        "config": {"workload": f"llama-2-70b shapes, TP={world}, prefill seq {args.seq}; {nl} layers x (5 column-parallel + 2 row-parallel "
                               "BinaryDiff Linears), RCCL all-reduce after o_proj and down_proj",
                   "seq_len": args.seq, "parallelism": f"tp{world}", "valid": args.layers is None and world == 8,
                   "all_reduce_messages_per_step": 2 * nl, "all_reduce_bytes_each_prefill": msg_prefill,
                   "all_reduce_bytes_each_decode": 8192 * 4},
        "roofline": {"bound": "mfma", "achieved": k_flops / k_ms * 1e-9 if k_ms > 0 else None, "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": (k_flops / k_ms * 1e-9 / 2500.0) if k_ms > 0 else None, "traffic": None, "launches": n_launch,
                     "share_of_step_time": (k_ms / 1e3) / dt if dt > 0 else None},
        "decode": {"ms_per_step": ddt / 20 * 1e3, "kv_len": args.kv_len, "linear_bytes_per_rank": dec.linear_bytes_per_rank(),
                   "linear_gbs_per_rank_if_only_linears": dec.linear_bytes_per_rank() / (ddt / 20) * 1e-9},
    }
