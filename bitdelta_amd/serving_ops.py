"""ctypes wrappers of the decode-step glue kernels (csrc/bd_serving.h): per-tenant RMSNorm, SwiGLU on the fused gate|up output,
and single-token attention with RoPE + KV-cache append.  Callers of the hot path, used by serving_loop.TenantDecoder at decode;
each has a stock-torch equivalent in that module (used at prefill and as the test reference)."""
import torch

from ._lib import DTYPE_CODE, check, lib, ptr, require_gpu, stream_ptr, workspace


def rmsnorm_tenant(x, w, eps):
    """x [T, M, H], w [T, H] -> w[t] * round(x * rsqrt(mean(x^2) + eps))   (HF RMSNorm, tenant t's weight for row block t)"""
    require_gpu(x, w)
    T, M, H = x.shape
    assert w.shape == (T, H) and w.dtype == x.dtype and x.stride(2) == 1 and w.stride(1) == 1
    assert T == 1 or x.stride(0) == M * x.stride(1), "rows must be evenly strided"
    y = torch.empty((T, M, H), device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        check(lib().bd_srv_rmsnorm(ptr(x), ptr(w), ptr(y), T * M, H, x.stride(1), H, w.stride(0), M, float(eps),
                                   DTYPE_CODE[x.dtype], stream_ptr()), "srv_rmsnorm")
    return y


def add_rmsnorm(resid, y32, w, eps):
    """(x, h) with x = resid + y32.to(resid.dtype) and h = rmsnorm_tenant(x, w, eps), in ONE launch and with exactly those ops' roundings.
    resid [T, M, H] 16-bit, y32 [T, M, H] fp32 (a row-parallel Linear's reduced partial sums), w [T, H]."""
    require_gpu(resid, y32, w)
    T, M, H = resid.shape
    assert y32.shape == resid.shape and y32.dtype == torch.float32 and w.shape == (T, H) and w.dtype == resid.dtype
    assert resid.stride(2) == 1 and y32.stride(2) == 1 and w.stride(1) == 1 and H % 8 == 0 and H <= 8192
    assert T == 1 or (resid.stride(0) == M * resid.stride(1) and y32.stride(0) == M * y32.stride(1)), "rows must be evenly strided"
    x = torch.empty((T, M, H), device=resid.device, dtype=resid.dtype)
    h = torch.empty((T, M, H), device=resid.device, dtype=resid.dtype)
    with torch.cuda.device(resid.device):
        check(lib().bd_srv_add_rmsnorm(ptr(resid), ptr(y32), ptr(w), ptr(x), ptr(h), T * M, H, resid.stride(1), y32.stride(1), H, H, w.stride(0), M,
                                       float(eps), DTYPE_CODE[resid.dtype], stream_ptr()), "srv_add_rmsnorm")
    return x, h


def swiglu(gu, inter):
    """gu [T, M, 2*inter] (gate columns, then up columns) -> round(silu(gate)) * up, [T, M, inter]"""
    assert gu.shape[2] == 2 * inter
    return swiglu2(gu[..., :inter], gu[..., inter:])


def swiglu2(g, u):
    """g, u [B, S, I] (views allowed: last dim contiguous, rows evenly strided) -> round(silu(g)) * u as a new [B, S, I] tensor"""
    require_gpu(g, u)
    B, S, I = g.shape
    assert u.shape == g.shape and u.dtype == g.dtype and g.stride(2) == 1 and u.stride(2) == 1
    assert B == 1 or (g.stride(0) == S * g.stride(1) and u.stride(0) == S * u.stride(1))
    y = torch.empty((B, S, I), device=g.device, dtype=g.dtype)
    with torch.cuda.device(g.device):
        check(lib().bd_srv_swiglu(ptr(g), ptr(u), ptr(y), B * S, I, g.stride(1), u.stride(1), I, 0, DTYPE_CODE[g.dtype], stream_ptr()),
              "srv_swiglu")
    return y


def swiglu_interleaved8(gu):
    """gu [B, S, 2*I]: the output of a gate|up projection whose rows are interleaved in blocks of 8 ([g0..7 | u0..7 | g8..15 | ...],
    FusedDeltaLinear(interleave8=True)) -> round(silu(gate)) * up, [B, S, I]"""
    require_gpu(gu)
    B, S, I2 = gu.shape
    I = I2 // 2
    assert I2 % 16 == 0 and gu.stride(2) == 1 and (B == 1 or gu.stride(0) == S * gu.stride(1))
    y = torch.empty((B, S, I), device=gu.device, dtype=gu.dtype)
    with torch.cuda.device(gu.device):
        check(lib().bd_srv_swiglu(ptr(gu), ptr(gu), ptr(y), B * S, I, gu.stride(1), gu.stride(1), I, 1, DTYPE_CODE[gu.dtype],
                                  stream_ptr()), "srv_swiglu")
    return y


def decode_attention(qkv, cos, sin, kcache, vcache, valid, pos, heads, kv_heads):
    """One new token per tenant.  qkv [T, 1, (heads + 2 kv_heads) * 128]; cos / sin [Lmax, 128]; caches [T, kv_heads, Lc, 128];
    valid [T, Lc] bool; pos: int64 device tensor with one element.  Returns [T, 1, heads * 128]; the caches and valid[:, pos] are
    updated in place."""
    require_gpu(qkv, cos, sin, kcache, vcache, valid, pos)
    T = qkv.shape[0]
    hd = kcache.shape[3]
    assert qkv.shape[1] == 1 and qkv.shape[2] == (heads + 2 * kv_heads) * hd and qkv.stride(2) == 1
    assert kcache.is_contiguous() and vcache.is_contiguous() and valid.is_contiguous() and valid.dtype == torch.bool
    assert cos.is_contiguous() and sin.is_contiguous() and cos.dtype == qkv.dtype and pos.dtype == torch.int64
    out = torch.empty((T, 1, heads * hd), device=qkv.device, dtype=qkv.dtype)
    L = lib()
    need = L.bd_srv_decode_attention_workspace_bytes(T, heads, kv_heads, hd, kcache.shape[2])
    # persistent per-stream scratch, zero-filled once: the kernel's arrival counters must be zero at launch and it restores them
    ws, need = workspace(need, qkv.device, zeroed=True) if need > 0 else (None, 0)
    with torch.cuda.device(qkv.device):
        check(L.bd_srv_decode_attention(ptr(qkv), ptr(cos), ptr(sin), ptr(kcache), ptr(vcache), ptr(valid), ptr(pos), ptr(out),
                                        T, heads, kv_heads, hd, kcache.shape[2], qkv.stride(0), out.stride(0),
                                        DTYPE_CODE[qkv.dtype], ptr(ws), need, stream_ptr()), "srv_decode_attention")
    return out


def prefill_attention_supported(q, k, v):
    """the shapes bd_srv_prefill_attention takes: [B, S, heads, 128] views (any batch / sequence strides that are multiples of 8 elements,
    heads contiguous), S a multiple of 64"""
    if q.dim() != 4 or q.shape[3] != 128 or q.dtype not in DTYPE_CODE or q.shape[1] % 64 or q.shape[2] % k.shape[2]:
        return False
    for t in (q, k, v):
        if t.stride(3) != 1 or t.stride(2) != 128 or t.stride(1) % 8 or t.stride(0) % 8 or t.data_ptr() % 16:
            return False
    return k.shape == v.shape and k.shape[:2] == q.shape[:2] and k.dtype == q.dtype == v.dtype


def prefill_attention(q, k, v, kv_start=None, causal=True, scale=None):
    """Flash-style attention of a whole prompt.  q [B, S, heads, 128], k / v [B, S, kv_heads, 128] -- views into a fused q|k|v projection
    output are fine (after RoPE); kv_start: optional int32 [B] device tensor, first valid key of each sequence (left padding).
    Returns [B, S, heads * 128] (what the o projection consumes)."""
    require_gpu(q, k, v)
    assert prefill_attention_supported(q, k, v), "prefill_attention: unsupported geometry"
    B, S, H, hd = q.shape
    out = torch.empty((B, S, H * hd), device=q.device, dtype=q.dtype)
    if kv_start is not None:
        assert kv_start.dtype == torch.int32 and kv_start.is_cuda and kv_start.numel() == B and kv_start.is_contiguous()
    with torch.cuda.device(q.device):
        check(lib().bd_srv_prefill_attention(ptr(q), ptr(k), ptr(v), ptr(out), B, S, H, k.shape[2], hd, q.stride(0), q.stride(1),
                                             k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1),
                                             ptr(kv_start) if kv_start is not None else None,
                                             float(scale if scale is not None else hd ** -0.5), int(bool(causal)),
                                             DTYPE_CODE[q.dtype], stream_ptr()), "srv_prefill_attention")
    return out


def cache_warm(a, b=None, blocks=0):
    """Read tensor a (and b) and discard the values on the CURRENT stream: the weight-prefetch node of a hipGraph side branch
    (serving_loop.TenantDecoder.prefetch_o).  Contiguous device tensors; whole 16-byte chunks are touched."""
    require_gpu(a)
    assert a.is_contiguous() and (b is None or (b.is_cuda and b.is_contiguous()))
    with torch.cuda.device(a.device):
        check(lib().bd_srv_cache_warm(ptr(a), a.numel() * a.element_size(), ptr(b) if b is not None else None,
                                      b.numel() * b.element_size() if b is not None else 0, int(blocks), stream_ptr()), "srv_cache_warm")


def decode_attention_supported(heads, kv_heads, head_dim):
    return head_dim == 128 and heads % kv_heads == 0 and heads // kv_heads in (1, 4, 8)


def rope_(x, cos, sin, heads, seq, pos0=0):
    """In-place rotary embedding of x [B, S, heads * 128] (a q / k projection output, before the head transpose); row r sits at position
    pos0 + r % seq.  cos / sin [Lmax, 128] in x.dtype with the rotate-half sign folded into sin.  Returns x."""
    require_gpu(x, cos, sin)
    assert x.dim() == 3 and x.shape[2] == heads * 128 and x.stride(2) == 1
    assert x.shape[0] == 1 or x.stride(0) == x.shape[1] * x.stride(1)
    assert cos.is_contiguous() and sin.is_contiguous() and cos.dtype == x.dtype and cos.shape[0] >= pos0 + seq
    with torch.cuda.device(x.device):
        check(lib().bd_srv_rope(ptr(x), ptr(cos), ptr(sin), x.shape[0] * x.shape[1], heads, 128, x.stride(1), seq, pos0,
                                DTYPE_CODE[x.dtype], stream_ptr()), "srv_rope")
    return x


def step_begin(embed, tok, valid, pos):
    """First launch of a greedy decode step: x[t] = embed[t, tok[t]] -> [T, 1, H], and valid[t, pos] = True (what `valid.index_fill_(1, pos, True)`
    and the per-tenant embedding gather do).  embed [T, V, H] (or [V, H]: one shared table), tok [T, 1] long, valid [T, Lc] bool, pos [1] long."""
    require_gpu(embed, tok, valid, pos)
    shared = embed.dim() == 2
    V, H = embed.shape[-2], embed.shape[-1]
    T = tok.shape[0]
    assert tok.dtype == torch.long and tok.is_contiguous() and tok.numel() == T and pos.dtype == torch.long and pos.numel() == 1
    assert valid.dtype == torch.bool and valid.is_contiguous() and valid.shape[0] == T and (shared or embed.shape[0] == T)
    assert embed.stride(-1) == 1 and H % 8 == 0
    x = torch.empty((T, 1, H), device=embed.device, dtype=embed.dtype)
    with torch.cuda.device(embed.device):
        check(lib().bd_srv_step_begin(ptr(embed), 0 if shared else embed.stride(0), embed.stride(-2), ptr(tok), ptr(x), H, ptr(valid), valid.shape[1],
                                      ptr(pos), T, V, H, stream_ptr()), "srv_step_begin")
    return x


def step_end(logits, tok, out, step, pos, stop_ids, stopped, ticket):
    """Last launch of a greedy decode step: nxt = argmax(logits, -1) (torch's order); tok[:, 0] = nxt; out[:, step] = nxt; stopped |= (nxt[:, None] ==
    stop_ids).any(1); pos += 1; step += 1 -- the five stock ops of the loop in one launch.  logits [T, V] 16-bit, ticket: a zeroed int32 word."""
    require_gpu(logits, tok, out, step, pos, stop_ids, stopped, ticket)
    T, V = logits.shape
    assert logits.stride(1) == 1 and V % 8 == 0 and tok.dtype == torch.long and tok.is_contiguous() and tok.numel() == T
    assert out.dtype == torch.long and out.shape[0] == T and out.stride(1) == 1 and stop_ids.dtype == torch.long and stop_ids.is_contiguous()
    assert stop_ids.shape[0] == T and stopped.dtype == torch.bool and stopped.is_contiguous() and stopped.numel() == T
    assert step.dtype == torch.long and pos.dtype == torch.long and step.numel() == 1 and pos.numel() == 1
    assert ticket.dtype == torch.int32 and ticket.numel() == 1
    with torch.cuda.device(logits.device):
        check(lib().bd_srv_step_end(ptr(logits), logits.stride(0), V, ptr(tok), ptr(out), out.stride(0), out.shape[1], ptr(stop_ids), stop_ids.shape[1],
                                    ptr(stopped), ptr(pos), ptr(step), ptr(ticket), T, DTYPE_CODE[logits.dtype], stream_ptr()), "srv_step_end")


def rope_kv_append_(qkv, cos, sin, kcache, vcache, heads, kvh, pos0=0):
    """Prefill from position pos0: rope_ on the q and k heads of the fused projection output qkv [T, S, (heads + 2 kvh) * 128] (in place) and, in the
    same launch, the rotated k rows and the v rows into the caches [T, kvh, Lc, 128] at positions pos0 .. pos0 + S - 1 (what
    `kcache[:, :, pos0:pos0 + S] = k.transpose(1, 2)` and its v twin do after rope_).  Returns qkv."""
    require_gpu(qkv, cos, sin, kcache, vcache)
    T, S, W = qkv.shape
    assert W == (heads + 2 * kvh) * 128 and qkv.stride(2) == 1 and (T == 1 or qkv.stride(0) == S * qkv.stride(1))
    assert cos.is_contiguous() and sin.is_contiguous() and cos.dtype == qkv.dtype and cos.shape[0] >= pos0 + S
    Lc = kcache.shape[2]
    assert kcache.shape == (T, kvh, Lc, 128) and vcache.shape == kcache.shape and kcache.is_contiguous() and vcache.is_contiguous()
    assert kcache.dtype == qkv.dtype and vcache.dtype == qkv.dtype and pos0 + S <= Lc
    with torch.cuda.device(qkv.device):
        check(lib().bd_srv_rope_kv_append(ptr(qkv), ptr(cos), ptr(sin), ptr(kcache), ptr(vcache), T, S, heads, kvh, 128, qkv.stride(1), Lc, pos0,
                                          DTYPE_CODE[qkv.dtype], stream_ptr()), "srv_rope_kv_append")
    return qkv
