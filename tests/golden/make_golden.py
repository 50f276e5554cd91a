"""Generate golden vectors by running the REFERENCE implementation in the authoring container.

Run once (needs /root/reference; never runs on the GPU box):
    TRITON_INTERPRET=1 python tests/golden/make_golden.py

Writes only DATA (inputs + the reference's outputs) next to this script:
    golden.pt            G1 pack/unpack, G2 BinaryDiff.__init__, G3 Triton kernel body (interpreter,
                         fp16), G5 BinaryDiff.forward, G6 DiffCompressModule-expression, G7 merge line
    tiny_llama_diff.pt   G4: a diff.pt written by the reference's save_diff for a tiny random Llama
    tiny_llama_merged.pt G4: base weights (fp16) before / after the reference's load_diff

SURVEY.md section 8(c) lists these vectors.  No reference source text is stored, only tensors.
"""
import os
import sys

os.environ.setdefault("TRITON_INTERPRET", "1")
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import bitdelta.binary_gemm_kernel as ref_k  # noqa: E402
import bitdelta.diff as ref_d  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
G = {}


def interp_bmm(a, b, BM=16, BN=32, BK=32):
    """Run the reference's binary_bmm_kernel BODY (bitdelta/binary_gemm_kernel.py:195-295) under the
    Triton interpreter on CPU tensors, bypassing the autotuner (which needs a GPU driver)."""
    import triton
    B, M, K = a.shape
    N = b.shape[-1]
    c = torch.empty((B, M, N), dtype=a.dtype)
    grid = (triton.cdiv(M, BM) * triton.cdiv(N, BN), B)
    ref_k.binary_bmm_kernel.fn[grid](
        a, b, c, M, N, K, 32,
        a.stride(1), a.stride(2), b.stride(1), b.stride(2), c.stride(1), c.stride(2),
        a.stride(0), b.stride(0), c.stride(0),
        BLOCK_SIZE_M=BM, BLOCK_SIZE_N=BN, BLOCK_SIZE_K=BK, GROUP_SIZE_M=8, ACTIVATION="")
    return c


def interp_mm(a, b, BM=16, BN=32, BK=32):
    import triton
    M, K = a.shape
    N = b.shape[-1]
    c = torch.empty((M, N), dtype=a.dtype)
    grid = (triton.cdiv(M, BM) * triton.cdiv(N, BN),)
    ref_k.binary_matmul_kernel.fn[grid](
        a, b, c, M, N, K, 32,
        a.stride(0), a.stride(1), b.stride(0), b.stride(1), c.stride(0), c.stride(1),
        BLOCK_SIZE_M=BM, BLOCK_SIZE_N=BN, BLOCK_SIZE_K=BK, GROUP_SIZE_M=8, ACTIVATION="")
    return c


# ---------------- G1: pack / unpack ----------------
g1 = []
for seed, shape in enumerate([(32, 1), (64, 48), (2, 3, 64, 5), (4096, 64)]):
    torch.manual_seed(seed)
    x = torch.rand(shape) > 0.5
    p = ref_k.pack(x)
    u = ref_k.unpack(p)
    assert torch.equal(u, x)
    g1.append({"bits": x, "packed": p})
ones = torch.ones(32, 3, dtype=torch.bool)
g1.append({"bits": ones, "packed": ref_k.pack(ones)})               # all-ones word == -1
b31 = torch.zeros(32, 2, dtype=torch.bool); b31[31, 0] = True; b31[0, 1] = True
g1.append({"bits": b31, "packed": ref_k.pack(b31)})                 # INT32_MIN / 1
G["g1_pack32"] = g1
g1n = {}
for nb in (8, 16, 64):
    torch.manual_seed(10 + nb)
    x = torch.rand(2, 128, 7) > 0.5
    p = ref_k.pack(x, n_bits=nb)
    assert torch.equal(ref_k.unpack(p, n_bits=nb), x)
    g1n[nb] = {"bits": x, "packed": p}
G["g1_pack_nbits"] = g1n
# transposed-view input, as BinaryDiff.__init__ passes (diff.py:16)
torch.manual_seed(5)
xt = (torch.rand(48, 64) > 0.5)
G["g1_pack_transposed"] = {"bits_NK": xt, "packed": ref_k.pack(xt.T)}

# ---------------- G2: BinaryDiff.__init__ ----------------
torch.manual_seed(20)
base = (torch.randn(96, 64) * 0.02).bfloat16()
fine = (base.float() + torch.randn(96, 64) * 5e-4).bfloat16()
fine[3, 7] = base[3, 7]            # exact-zero diff -> bit 1
fine[5, 33] = base[5, 33]
m = ref_d.BinaryDiff(base.clone(), fine.clone())
G["g2_binarydiff"] = {
    "base": base, "fine": fine, "mask": m.mask.clone(), "coeff": m.coeff.detach().clone(),
    "state_keys": list(m.state_dict().keys()),
    "base_buf_shape": tuple(m.base.shape), "base_buf_stride": tuple(m.base.stride()),
    "coeff_is_parameter": isinstance(m.coeff, nn.Parameter), "coeff_requires_grad": m.coeff.requires_grad,
}
torch.manual_seed(21)
base16 = (torch.randn(64, 96) * 0.02).half()
fine16 = (base16.float() + torch.randn(64, 96) * 5e-4).half()
m16 = ref_d.BinaryDiff(base16.clone(), fine16.clone())
G["g2_binarydiff_fp16"] = {"base": base16, "fine": fine16, "mask": m16.mask.clone(),
                           "coeff": m16.coeff.detach().clone()}

# ---------------- G3: kernel semantics (Triton interpreter, fp16) ----------------
g3 = []
for i, (B, M, K, N) in enumerate([(1, 1, 64, 32), (2, 16, 64, 32), (3, 17, 96, 40), (2, 33, 512, 72), (1, 128, 512, 64)]):
    torch.manual_seed(30 + i)
    a = torch.randn(B, M, K).half()
    bits = torch.randn(B, K, N) > 0.5
    p = ref_k.pack(bits)
    c = interp_bmm(a, p)
    g3.append({"a": a, "packed": p, "c": c})
G["g3_bmm_fp16"] = g3
torch.manual_seed(40)
a = torch.randn(48, 128).half()
p = ref_k.pack(torch.randn(128, 96) > 0.5)
G["g3_mm_fp16"] = {"a": a, "packed": p, "c": interp_mm(a, p)}
# larger magnitude inputs so the fp16 epilogue rounding (and overflow to inf) is exercised
torch.manual_seed(41)
a = (torch.randn(1, 8, 2048) * 40).half()
p = ref_k.pack(torch.ones(1, 2048, 32, dtype=torch.bool))
G["g3_bmm_fp16_big"] = {"a": a, "packed": p, "c": interp_bmm(a, p)}

# ---------------- G5: BinaryDiff.forward (bf16 and fp16) with binary_bmm = kernel semantics ----------------
# binary_bmm itself needs a GPU (torch.cuda.device + autotuner); the kernel's result for bf16 inputs is
# fp32 accumulate -> fp16 -> bf16 store-cast (binary_gemm_kernel.py:287, :314).  The interpreter's bf16
# path is broken (returns inf), so the stand-in computes that formula from the reference's own unpack.


def kernel_semantics_bmm(a, b, n_bits=32, activation=""):
    s = (ref_k.unpack(b).float() * 2 - 1)
    acc = torch.bmm(a.float(), s)
    return acc.half().to(a.dtype)


ref_d.binary_bmm = kernel_semantics_bmm
for tag, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
    torch.manual_seed(50)
    base = (torch.randn(40, 64) * 0.02).to(dt)
    fine = (base.float() + torch.randn(40, 64) * 5e-4).to(dt)
    mod = ref_d.BinaryDiff(base.clone(), fine.clone())
    x = torch.randn(2, 3, 64).to(dt)
    with torch.no_grad():
        y = mod(x)
    G[f"g5_forward_{tag}"] = {"base": base, "fine": fine, "x": x, "y": y, "mask": mod.mask.clone(),
                              "coeff": mod.coeff.detach().clone()}
# cross-check the stand-in against the interpreter on fp16
chk = kernel_semantics_bmm(G["g3_bmm_fp16"][2]["a"], G["g3_bmm_fp16"][2]["packed"])
assert torch.equal(chk, G["g3_bmm_fp16"][2]["c"]), "stand-in != interpreter"

# ---------------- G6: DiffCompressModule.forward expression (demo/demo_backend.py:93-98), T=3 tenants ----------------
torch.manual_seed(60)
T, M, K, N = 3, 5, 64, 48
lin_w = (torch.randn(N, K) * 0.02).half()
masks = torch.stack([ref_k.pack(torch.randn(K, N) > 0.0) for _ in range(T)], 0)
coeffs = torch.tensor([3.1e-4, 4.7e-4, 5.3e-4]).half()
h = torch.randn(T, M, K).half()
out = torch.nn.functional.linear(h, lin_w)
diff = kernel_semantics_bmm(h, masks) * coeffs[:, None, None]
G["g6_multitenant_fp16"] = {"w": lin_w, "masks": masks, "coeffs": coeffs, "h": h, "y": out + diff}

# ---------------- G7: load_diff merge line (diff.py:93-95) ----------------
torch.manual_seed(70)
w = (torch.randn(48, 64) * 0.02).half()
mask = ref_k.pack(torch.randn(64, 48) > 0.0)
coeff = torch.tensor(4.2e-4, dtype=torch.float32)
w2 = w.clone()
w2.add_(((ref_k.unpack(mask) * 2 - 1) * coeff).T.to(w2.dtype))
G["g7_merge_fp16"] = {"w": w, "mask": mask, "coeff": coeff, "w_merged": w2}
wb = w.bfloat16()
wb2 = wb.clone()
wb2.add_(((ref_k.unpack(mask) * 2 - 1) * coeff).T.to(wb2.dtype))
G["g7_merge_bf16"] = {"w": wb, "mask": mask, "coeff": coeff, "w_merged": wb2}

torch.save(G, os.path.join(HERE, "golden.pt"))

# ---------------- G4: tiny Llama diff.pt written by the reference ----------------
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

cfg = LlamaConfig(vocab_size=128, hidden_size=64, intermediate_size=96, num_hidden_layers=2,
                  num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=64)
torch.manual_seed(80)
base_m = LlamaForCausalLM(cfg).bfloat16()
torch.manual_seed(81)
fine_m = LlamaForCausalLM(cfg).bfloat16()
with torch.no_grad():
    for (n, pb), (_, pf) in zip(base_m.named_parameters(), fine_m.named_parameters()):
        pf.copy_((pb.float() + torch.randn_like(pb.float()) * 5e-4).bfloat16())
import copy  # noqa: E402
comp = copy.deepcopy(fine_m)
ref_d.compress_diff(base_m, fine_m, comp)
path = os.path.join(HERE, "tiny_llama_diff.pt")
ref_d.save_diff(comp, path)
dd = torch.load(path, weights_only=False)
eval_m = copy.deepcopy(base_m).half()
before = {k: v.detach().clone() for k, v in eval_m.state_dict().items()}
_orig_load = torch.load
torch.load = lambda f, *a, **k: _orig_load(f, weights_only=False)
ref_d.load_diff(eval_m, path)
torch.load = _orig_load
after = {k: v.detach().clone() for k, v in eval_m.state_dict().items()}
torch.save({"config": cfg.to_dict(), "before": before, "after": after,
            "diff_keys": list(dd.keys()),
            "diff_types": {k: (type(v).__name__, str(v.dtype), tuple(v.shape)) for k, v in dd.items()},
            "base_state": {k: v.detach().clone() for k, v in base_m.state_dict().items()},
            "fine_state": {k: v.detach().clone() for k, v in fine_m.state_dict().items()}},
           os.path.join(HERE, "tiny_llama_merged.pt"))
print("golden written:", {k: type(v).__name__ for k, v in G.items()})
print("diff.pt keys:", len(dd), "size", os.path.getsize(path))
