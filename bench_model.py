"""Synthetic-weight Llama/Mistral decoder used by bench.py (plain PyTorch plumbing around the HIP op).

The hot path (the 7 projections of every layer) runs through bitdelta_amd.BinaryDiff / DiffCompressModule -> the
fused HIP kernels.  Everything else (RMSNorm, RoPE, SDPA attention, SiLU, embedding, lm_head) is stock torch: it is
the caller of the path, not the product.  Weights are random with the statistics SURVEY.md section 8(d) prescribes
(W ~ N(0, 0.02^2) bf16, fine-tune = W + N(0, (5e-4)^2), alpha = mean|delta| ~ 4e-4); there is no network for real
checkpoints.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from bitdelta_amd.diff import BinaryDiff
from bitdelta_amd.serving import DiffCompressModule
from bitdelta_amd.diff import binarize
from bitdelta_amd import serving_ops as ops

CONFIGS = {
    # name: (hidden, intermediate, layers, heads, kv_heads, vocab)
    "llama-2-7b": (4096, 11008, 32, 32, 32, 32000),
    "mistral-7b": (4096, 14336, 32, 32, 8, 32000),
    "llama-2-70b": (8192, 28672, 80, 64, 8, 32000),
    "tiny": (256, 512, 2, 4, 4, 512),
}


def synth_pair(n_out, n_in, device, dtype, gen):
    w = (torch.randn(n_out, n_in, device=device, generator=gen) * 0.02).to(dtype)
    fine = (w.float() + torch.randn(n_out, n_in, device=device, generator=gen) * 5e-4).to(dtype)
    return w, fine


class RMSNorm(nn.Module):
    def __init__(self, dim, dtype, device, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, dtype=dtype, device=device), requires_grad=False)
        self.eps = eps

    def forward(self, x):
        if hasattr(F, "rms_norm"):                       # fused kernel (torch >= 2.4); same fp32-internal math
            return F.rms_norm(x, (x.shape[-1],), self.weight, self.eps)
        v = x.float()
        v = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + self.eps)
        return (v.to(x.dtype)) * self.weight


def rope_tables(seq, dim, device, base=10000.0):
    inv = 1.0 / (base ** (torch.arange(0, dim, 2, device=device, dtype=torch.float32) / dim))
    t = torch.arange(seq, device=device, dtype=torch.float32)
    f = torch.outer(t, inv)
    emb = torch.cat([f, f], dim=-1)
    return emb.cos(), emb.sin()


def apply_rope(x, cos, sin):
    # x [B, H, S, D]; cos/sin [S, D] already in x.dtype with the rotate-half sign folded into `sin`
    d = x.shape[-1] // 2
    rot = torch.cat([x[..., d:], x[..., :d]], dim=-1)
    return torch.addcmul(x * cos, rot, sin)


class SingleTenantLinear(nn.Module):
    """One BinaryDiff (base + 1 delta): training/eval form, reference bitdelta/diff.py:8-39."""

    def __init__(self, n_out, n_in, device, dtype, gen):
        super().__init__()
        w, fine = synth_pair(n_out, n_in, device, dtype, gen)
        self.lin = BinaryDiff(w, fine)
        self.lin.coeff.requires_grad_(False)
        self.flops_per_row = 4 * n_out * n_in
        self.residual_epilogue = True

    def forward(self, x, residual=None):
        return self.lin(x, residual=residual)


class FusedSingleTenantLinear(nn.Module):
    """Several single-tenant BinaryDiff projections that read the same input, stored and launched as ONE (q|k|v; gate|up with the
    rows interleaved in blocks of 8 so that SwiGLU runs in the GEMM's epilogue): serving_loop.FusedDeltaLinear with one tenant."""

    def __init__(self, shapes, device, dtype, gen, interleave8=False):
        super().__init__()
        from bitdelta_amd.serving_loop import FusedDeltaLinear
        ws, ms, cs = [], [], []
        for n_out, n_in in shapes:
            w, fine = synth_pair(n_out, n_in, device, dtype, gen)
            m, c = binarize(w, fine)
            ws.append(w); ms.append(m[None]); cs.append(c.reshape(1))
        self.lin = FusedDeltaLinear(ws, ms, cs, interleave8=interleave8, decode_copies=False)
        self.flops_per_row = sum(4 * o * i for o, i in shapes)

    def forward(self, x):
        return self.lin(x)


class MultiTenantLinear(nn.Module):
    """One base nn.Linear + T deltas, row i -> tenant i: serving form, reference demo/demo_backend.py:82-98."""

    def __init__(self, n_out, n_in, device, dtype, gen, tenants):
        super().__init__()
        base = nn.Linear(n_in, n_out, bias=False, device=device, dtype=dtype)
        masks, coeffs = [], []
        with torch.no_grad():
            base.weight.copy_((torch.randn(n_out, n_in, device=device, generator=gen) * 0.02).to(dtype))
            for _ in range(tenants):
                fine = (base.weight.float() + torch.randn(n_out, n_in, device=device, generator=gen) * 5e-4).to(dtype)
                m, c = binarize(base.weight.data, fine)
                masks.append(m)
                coeffs.append(c)
        base.weight.requires_grad_(False)
        self.lin = DiffCompressModule(base, torch.stack(masks, 0).contiguous(), torch.stack(coeffs, 0).to(dtype))
        self.flops_per_row = 4 * n_out * n_in
        self.residual_epilogue = False

    def forward(self, x, residual=None):
        y = self.lin(x)
        return y if residual is None else residual.add_(y)


class DecoderLayer(nn.Module):
    def __init__(self, cfg, device, dtype, gen, tenants=0, fuse=True):
        super().__init__()
        hid, inter, _, heads, kvh, _ = cfg
        self.heads, self.kvh, self.hd = heads, kvh, hid // heads
        mk = (lambda o, i: MultiTenantLinear(o, i, device, dtype, gen, tenants)) if tenants else \
             (lambda o, i: SingleTenantLinear(o, i, device, dtype, gen))
        # single tenant: q|k|v and gate|up as ONE launch each (4 GEMM launches per layer: q|k|v, o + residual, gate|up -> SwiGLU,
        # down + residual); `fuse=False` keeps the reference's seven separate BinaryDiff modules
        self.fused = (not tenants) and fuse and inter % 8 == 0
        self.swiglu_epilogue = False
        self.hip_attention = True      # prefill attention through bd_srv_prefill_attention (False: torch SDPA, for A/B)
        if self.fused:
            self.qkv_proj = FusedSingleTenantLinear([(hid, hid), (kvh * self.hd, hid), (kvh * self.hd, hid)], device, dtype, gen)
            self.gate_up_proj = FusedSingleTenantLinear([(inter, hid), (inter, hid)], device, dtype, gen, interleave8=True)
        else:
            self.q_proj = mk(hid, hid)
            self.k_proj = mk(kvh * self.hd, hid)
            self.v_proj = mk(kvh * self.hd, hid)
            self.gate_proj = mk(inter, hid)
            self.up_proj = mk(inter, hid)
        self.o_proj = mk(hid, hid)
        self.down_proj = mk(hid, inter)
        self.input_layernorm = RMSNorm(hid, dtype, device)
        self.post_attention_layernorm = RMSNorm(hid, dtype, device)

    def forward(self, x, cos, sin, kv=None, rope=None):
        B, S, _ = x.shape
        h = self.input_layernorm(x)
        if self.fused:
            qkv = self.qkv_proj(h)                                               # [B, S, (heads + 2 kvh) * hd]: one launch
            nq, nk = self.heads * self.hd, self.kvh * self.hd
            qf, kf, vf = qkv[..., :nq], qkv[..., nq:nq + nk], qkv[..., nq + nk:]
            if self.hd == 128 and rope is not None:
                ops.rope_(qkv[..., :nq + nk], rope[0], rope[1], self.heads + self.kvh, S, rope[2])      # q and k heads: one launch
                q4, k4, v4 = qf.view(B, S, self.heads, self.hd), kf.view(B, S, self.kvh, self.hd), vf.view(B, S, self.kvh, self.hd)
                if kv is None and S > 1 and self.hip_attention and ops.prefill_attention_supported(q4, k4, v4):
                    # whole-prompt causal attention straight on the three slices of the fused projection output (no head transposes,
                    # no repeat_interleave for grouped queries); returns [B, S, heads * hd], what o_proj consumes
                    return self._after_attention(x, ops.prefill_attention(q4, k4, v4, causal=True))
                q = qf.view(B, S, self.heads, self.hd).transpose(1, 2)
                k = kf.view(B, S, self.kvh, self.hd).transpose(1, 2)
            else:
                q = apply_rope(qf.reshape(B, S, self.heads, self.hd).transpose(1, 2), cos, sin)
                k = apply_rope(kf.reshape(B, S, self.kvh, self.hd).transpose(1, 2), cos, sin)
            v = vf.view(B, S, self.kvh, self.hd).transpose(1, 2)
        elif self.hd == 128 and rope is not None:
            # fused in-place RoPE on the projection outputs (one pass instead of cat + mul + addcmul and their temporaries)
            q = ops.rope_(self.q_proj(h), rope[0], rope[1], self.heads, S, rope[2]).view(B, S, self.heads, self.hd).transpose(1, 2)
            k = ops.rope_(self.k_proj(h), rope[0], rope[1], self.kvh, S, rope[2]).view(B, S, self.kvh, self.hd).transpose(1, 2)
            v = self.v_proj(h).view(B, S, self.kvh, self.hd).transpose(1, 2)
        else:
            q = self.q_proj(h).view(B, S, self.heads, self.hd).transpose(1, 2)
            k = self.k_proj(h).view(B, S, self.kvh, self.hd).transpose(1, 2)
            v = self.v_proj(h).view(B, S, self.kvh, self.hd).transpose(1, 2)
            q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        if kv is not None:                      # decode: write into the preallocated cache [B, kvh, Lmax, hd]
            pos = kv[2]
            kv[0][:, :, pos:pos + S] = k
            kv[1][:, :, pos:pos + S] = v
            kv[2] = pos + S
            k, v = kv[0][:, :, :pos + S], kv[1][:, :, :pos + S]
        if self.kvh != self.heads:
            rep = self.heads // self.kvh
            k = k.repeat_interleave(rep, dim=1)
            v = v.repeat_interleave(rep, dim=1)
        a = F.scaled_dot_product_attention(q, k, v, is_causal=(S > 1 and k.shape[2] == S))
        return self._after_attention(x, a.transpose(1, 2).reshape(B, S, self.heads * self.hd))

    def _after_attention(self, x, a):
        x = self.o_proj(a, residual=x) if self._res_epilogue(x, self.o_proj) else x + self.o_proj(a)
        h = self.post_attention_layernorm(x)
        if self.fused and self.swiglu_epilogue and self.gate_up_proj.lin.swiglu_ok(h):
            act = self.gate_up_proj.lin.forward_swiglu(h)                        # gate|up -> SwiGLU in the GEMM's epilogue (A/B: slower)
        elif self.fused:
            act = ops.swiglu_interleaved8(self.gate_up_proj(h))                  # one GEMM launch + one elementwise pass
        else:
            g, u = self.gate_proj(h), self.up_proj(h)
            act = ops.swiglu2(g, u) if g.shape[-1] % 8 == 0 else F.silu(g) * u   # one pass: round(silu(g)) * u
        x = self.down_proj(act, residual=x) if self._res_epilogue(x, self.down_proj) else x + self.down_proj(act)
        return x

    @staticmethod
    def _res_epilogue(x, proj):
        # the residual stream is updated in place by the Linear's epilogue (inference, contiguous [B, S, hidden])
        return proj.residual_epilogue and (not torch.is_grad_enabled()) and x.is_contiguous()


class Decoder(nn.Module):
    def __init__(self, name, device, dtype=torch.bfloat16, tenants=0, layers=None, seed=0, fuse=True):
        super().__init__()
        cfg = CONFIGS[name]
        hid, inter, nl, heads, kvh, vocab = cfg
        nl = layers or nl
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.cfg, self.dtype, self.device_, self.tenants = cfg, dtype, device, tenants
        self.embed = nn.Embedding(vocab, hid, device=device, dtype=dtype)
        self.layers = nn.ModuleList([DecoderLayer(cfg, device, dtype, gen, tenants, fuse) for _ in range(nl)])
        self.norm = RMSNorm(hid, dtype, device)
        self.lm_head = nn.Linear(hid, vocab, bias=False, device=device, dtype=dtype)
        for p in self.parameters():
            p.requires_grad_(False)
        self.hd = hid // heads

    def new_cache(self, batch, max_len):
        _, _, _, heads, kvh, _ = self.cfg
        mk = lambda: torch.zeros(batch, kvh, max_len, self.hd, device=self.device_, dtype=self.dtype)
        return [[mk(), mk(), 0] for _ in self.layers]

    def linear_flops_per_token(self):
        return sum(m.flops_per_row for m in self.modules() if hasattr(m, "flops_per_row"))

    def linear_param_count(self):
        return self.linear_flops_per_token() // 4

    @torch.no_grad()
    def forward(self, ids, pos0=0, cache=None):
        B, S = ids.shape
        key = (pos0 + S)
        if getattr(self, "_rope_key", None) != key:      # tables are position-only: build once per length
            cos, sin = rope_tables(pos0 + S, self.hd, ids.device)
            d = self.hd // 2
            sin = torch.cat([-sin[:, :d], sin[:, d:]], dim=-1)          # rotate-half sign folded in
            self._rope = (cos.to(self.dtype), sin.to(self.dtype))
            self._rope_key = key
        cos, sin = self._rope[0][pos0:], self._rope[1][pos0:]
        rope = (self._rope[0], self._rope[1], pos0)          # whole tables + first position, for the fused in-place kernel
        x = self.embed(ids)
        for i, layer in enumerate(self.layers):
            x = layer(x, cos, sin, None if cache is None else cache[i], rope)
        return self.lm_head(self.norm(x[:, -1:, :]))
