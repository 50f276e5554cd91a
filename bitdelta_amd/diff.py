"""MI355X drop-in for the reference's ``bitdelta/diff.py``: same public names, same module surface, same ``diff.pt`` format.

    BinaryDiff(base, finetune)        reference :8-39    buffers ``mask`` int32 [in/32,out], ``base`` = W.T view,
                                                          parameter ``coeff`` fp32 0-dim; state_dict order coeff, mask, base
    compress_diff / save_diff / load_diff / save_full_model      reference :41-116

The behaviour of the model-surgery helpers (which modules are replaced, what ``diff.pt`` contains and in which order, how it is
merged back) follows the reference because the on-disk format and the module tree ARE the drop-in contract; the code here is
written against that contract, not transcribed.  Where the arithmetic runs is what changes: construction is one fused HIP pass
(sign + mean|diff| + transposed pack), the inference forward is one fused HIP launch (base GEMM + delta GEMM + scale + add), and
load_diff's merge is one HIP pass without the reference's int64 / fp32 [in,out] temporaries.

BitDelta is Apache-2.0 (FasterDecoding/BitDelta); the function names, signatures and the diff.pt layout are theirs.
"""
import gc

import torch
import torch.nn as nn

from ._lib import DTYPE_CODE, check, lib, ptr, require_gpu, stream_ptr, workspace
from .binary_gemm_kernel import binary_bmm, binary_linear, delta_bmm, pack, unpack  # noqa: F401  (re-exported, as reference diff.py:5 does)


# ------------------------------------------------------------------------------------------------ device primitives
def binarize(base, finetune):
    """(mask, coeff) of BinaryDiff.__init__ (reference diff.py:11-16) in one pass over the two [out, in] weights."""
    require_gpu(base, finetune)
    assert base.shape == finetune.shape and base.dim() == 2 and base.dtype == finetune.dtype
    assert base.dtype in (torch.float16, torch.bfloat16), "weights must be fp16 or bf16"
    if base.stride(1) != 1 or finetune.stride(1) != 1 or finetune.stride(0) != base.stride(0):
        base, finetune = base.contiguous(), finetune.contiguous()
    N, K = base.shape
    assert K % 32 == 0, "K must be divisible by n_bits"
    mask = torch.empty((K // 32, N), dtype=torch.int32, device=base.device)
    coeff = torch.empty((), dtype=torch.float32, device=base.device)
    L = lib()
    ws, ws_bytes = workspace(L.bd_binarize_workspace_bytes(N, K), base.device)
    with torch.cuda.device(base.device):
        check(L.bd_binarize(ptr(base), ptr(finetune), N, K, base.stride(0), DTYPE_CODE[base.dtype], ptr(mask),
                            ptr(coeff), ptr(ws), ws_bytes, stream_ptr()), "binarize")
    return mask, coeff


def merge_delta_(weight, mask, coeff):
    """In place ``weight += ((unpack(mask)*2-1) * coeff).T.to(weight.dtype)`` (reference diff.py:93-95)."""
    require_gpu(weight, mask, coeff)
    N, K = weight.shape
    assert mask.shape == (K // 32, N) and mask.dtype == torch.int32 and mask.is_contiguous()
    assert weight.stride(1) == 1 and weight.dtype in (torch.float16, torch.bfloat16)
    c = coeff.detach().reshape(()).to(device=weight.device, dtype=torch.float32)
    with torch.cuda.device(weight.device):
        check(lib().bd_merge_delta(ptr(weight), weight.stride(0), ptr(mask), ptr(c), N, K, DTYPE_CODE[weight.dtype],
                                   stream_ptr()), "merge_delta")
    return weight


def transpose_mask(mask):
    """Packed signs of S^T: [K/32, N] words over k  ->  [N/32, K] words over n  (bit j of word [i, k] <-> n = 32 i + j).
    Needed by the backward of the delta term, d/dx (x.S) = g . S^T, which is the same W1A16 GEMM on the transposed pack."""
    KW, N = mask.shape
    assert N % 32 == 0, "out_features must be divisible by 32 to pack the transposed signs"
    return pack(unpack(mask).T)


class _DeltaLinearFn(torch.autograd.Function):
    """y = x . W^T + coeff * (x . S) with the FULL gradient (opt-in, BinaryDiff.delta_input_grad = True).

    forward : one fused HIP launch (bd_binary_linear), fp32 accumulate, one rounding.
    backward: dx = g . W + coeff * (g . S^T)   -- ONE launch of the same fused kernel on (W^T stored [in, out], S^T packed over n):
                                                  the backward of a fused Linear is a fused Linear with the roles of in / out swapped
              dcoeff = sum(g * (x . S))         -- the delta GEMM (fp32 output), reduced in fp32
    The reference has no backward for its kernel: see BinaryDiff.forward for the default, reference-identical behaviour."""

    @staticmethod
    def forward(ctx, x3, weight_nk, mask, mask_t, weight_kn, coeff):
        ctx.save_for_backward(x3, mask, mask_t, weight_kn, coeff)
        return binary_linear(x3, weight_nk, mask.unsqueeze(0), coeff.detach().reshape(1, 1))

    @staticmethod
    def backward(ctx, g):
        x3, mask, mask_t, weight_kn, coeff = ctx.saved_tensors
        g = g.contiguous()
        gx = gc_ = None
        if ctx.needs_input_grad[0]:
            gx = binary_linear(g, weight_kn, mask_t.unsqueeze(0), coeff.detach().reshape(1, 1))      # [1, M, N] -> [1, M, K]
        if ctx.needs_input_grad[5]:
            c = delta_bmm(x3, mask.unsqueeze(0), out_dtype=torch.float32, round_mode=0)
            gc_ = (g.float() * c).sum().reshape(coeff.shape)
        return gx, None, None, None, None, gc_


class _RefLinearFn(torch.autograd.Function):
    """The reference-faithful training forward (BinaryDiff.delta_input_grad = False, the default) as ONE launch.

    The reference computes `x @ base + coeff * binary_bmm(x, mask)` (bitdelta/diff.py:33-39): a BLAS GEMM, a Triton launch with no
    autograd.Function, and two elementwise passes.  Its gradients are therefore d/dx = g . W only (the delta path contributes nothing:
    the kernel output carries no grad_fn) and d/dcoeff = sum(g * c) with c the kernel's 16-bit output.  This function keeps exactly
    those gradients -- the quirk that train.py's scale distillation (bitdelta/train.py:60-88) was run with -- and replaces the four
    forward launches by the fused HIP kernel (fp32 accumulate, fp32 alpha, one rounding).
    backward: dx = g @ W            the op autograd runs for `x @ base`
              dcoeff = sum(g * c)   c = the delta GEMM with the reference's fp32 -> fp16 -> dtype epilogue, product and sum in the
                                    activation dtype like autograd's mul / sum_to_size, then cast to coeff's fp32"""

    @staticmethod
    def forward(ctx, x3, weight_nk, mask, coeff):
        ctx.save_for_backward(x3, weight_nk, mask, coeff)
        return binary_linear(x3, weight_nk, mask.unsqueeze(0), coeff.detach().reshape(1, 1))

    @staticmethod
    def backward(ctx, g):
        x3, weight_nk, mask, coeff = ctx.saved_tensors
        gx = gc_ = None
        if ctx.needs_input_grad[0]:
            gx = g @ weight_nk
        if ctx.needs_input_grad[3]:
            c = delta_bmm(x3, mask.unsqueeze(0), round_mode=1)
            gc_ = (g * c).sum().to(coeff.dtype).reshape(coeff.shape)
        return gx, None, None, gc_


# ------------------------------------------------------------------------------------------------ the module
class BinaryDiff(nn.Module):
    """16-bit base weight + 1-bit delta Linear (reference bitdelta/diff.py:8-39); same buffers, parameter and state_dict order."""

    # Opt-in: True makes the delta term differentiable w.r.t. the input as well (row f4 of SURVEY.md section 8).  The default
    # (False) reproduces the reference exactly, including its quirk that d/dx of the delta path is dropped (its Triton launch has
    # no autograd.Function, bitdelta/diff.py:39), which is what train.py's scale distillation was run with.
    delta_input_grad = False
    # True: run the training forward as the reference's four separate launches (BLAS GEMM + delta GEMM + two elementwise passes)
    # instead of the one fused launch with the same gradients (_RefLinearFn).  A/B and parity hook.
    reference_composition = False

    def __init__(self, base, finetune):
        super().__init__()
        mask, mean_abs = binarize(base, finetune)
        self.register_buffer("mask", mask)
        self.register_buffer("base", base.T)
        self.register_parameter("coeff", nn.Parameter(mean_abs.detach().clone().to(torch.float32).requires_grad_(True)))
        self._mask_t = None          # lazily packed S^T for the opt-in backward (not a buffer: the state_dict stays the reference's)
        self._weight_kn = None       # ... and the base weight stored [in, out] (the backward launch's "W")
        self._mask_t_key = self._weight_kn_key = None

    def _weight_nk(self):
        w = self.base.T                      # [out, in]; contiguous while `base` is still the .T view it was built as
        return w if w.stride(1) == 1 else w.contiguous()      # e.g. after a state_dict round trip densified the buffer

    def _transposed_weight(self):
        """base weight as a contiguous [in, out] matrix: what the fused kernel needs as its row-major `W` when it computes
        dx = g . W + coeff * (g . S^T).  `self.base` IS that matrix logically ([in, out] view of the [out, in] storage); this is its
        contiguous copy, made once (training only, opt-in: +2 bytes per parameter)."""
        key = self._src_key(self.base)
        if self._weight_kn is None or self._weight_kn_key != key:
            self._weight_kn, self._weight_kn_key = self.base.contiguous(), key
        return self._weight_kn

    @staticmethod
    def _src_key(t):
        """identity AND content version of a buffer: a load_state_dict, an in-place update, a dtype cast or a device move all change it,
        so the derived copies below are rebuilt instead of silently feeding the backward a stale W^T / S^T"""
        return (t.data_ptr(), t._version, t.dtype, t.device, tuple(t.shape), tuple(t.stride()))

    def _transposed_mask(self):
        key = self._src_key(self.mask)
        if self._mask_t is None or self._mask_t_key != key:
            self._mask_t, self._mask_t_key = transpose_mask(self.mask), key
        return self._mask_t

    def forward(self, x, *, residual=None):
        # [B, seq, in] @ [in, out] + coeff * ([B, seq, in] @ S),  S = +-1 from the packed mask (broadcast, never repeated).
        # `residual` (keyword-only extension of the reference signature; inference only): the decoder layer's
        # `hidden = residual + proj(x)` in the kernel epilogue; `residual` is updated in place and returned.
        shape = x.shape
        x3 = x.reshape(1, -1, shape[-1])     # one [B*seq, in] problem: the mask is shared by every row
        if x3.stride(-1) != 1:
            x3 = x3.contiguous()
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or self.coeff.requires_grad)
        if residual is not None:
            assert not needs_grad and residual.is_contiguous() and residual.shape[:-1] == shape[:-1]
            r3 = residual.view(1, -1, residual.shape[-1])
            binary_linear(x3, self._weight_nk(), self.mask.unsqueeze(0), self.coeff.reshape(1, 1), residual=r3)
            return residual
        if not needs_grad:
            y = binary_linear(x3, self._weight_nk(), self.mask.unsqueeze(0), self.coeff.reshape(1, 1))
        elif self.delta_input_grad:
            y = _DeltaLinearFn.apply(x3, self._weight_nk(), self.mask, self._transposed_mask(), self._transposed_weight(), self.coeff)
        elif self.reference_composition:
            # the reference's own four launches (diff.py:39), kept as the A/B / parity reference of the fused training forward:
            # d/dcoeff flows through `coeff * c`, d/dx only through `x @ base` -- the kernel output carries no grad_fn
            c = delta_bmm(x3.detach(), self.mask.unsqueeze(0), round_mode=1)
            y = x3 @ self.base + self.coeff * c
        else:
            # training form, reference gradients (d/dx through the base GEMM only, d/dcoeff = sum(g * c)), ONE forward launch
            y = _RefLinearFn.apply(x3, self._weight_nk(), self.mask, self.coeff)
        return y.reshape(*shape[:-1], y.shape[-1])


BinaryLinear = BinaryDiff   # name used by BASELINE.json's north star; the reference class is BinaryDiff


# ------------------------------------------------------------------------------------------------ model surgery / diff.pt
def _is_delta_target(module_name, child_name):
    """Selection rule of the reference (diff.py:60-64): `*proj*` children of modules whose name contains mlp / self_attn."""
    return ("mlp" in module_name or "self_attn" in module_name) and "proj" in child_name


def compress_diff(base_model, finetuned_model, finetuned_compressed_model):
    """Replace every targeted Linear of `finetuned_compressed_model` by BinaryDiff(base weight, fine-tuned weight), in place,
    one at a time so that at most one extra weight pair is alive (the replaced Linear is dropped before the next one is built)."""
    targets = [(mod_name, mod, child_name)
               for mod_name, mod in finetuned_compressed_model.named_modules()
               for child_name, _ in mod.named_children()
               if _is_delta_target(mod_name, child_name)]
    for mod_name, mod, child_name in targets:
        path = f"{mod_name}.{child_name}"
        device = getattr(mod, child_name).weight.device
        w_base = base_model.get_submodule(path).weight.detach().to(device)
        w_fine = finetuned_model.get_submodule(path).weight.detach().to(device)
        replacement = BinaryDiff(base=w_base, finetune=w_fine).to(device)
        setattr(mod, child_name, None)           # release the dense Linear before installing the compressed one
        del w_base, w_fine
        gc.collect()
        torch.cuda.empty_cache()
        setattr(mod, child_name, replacement)


def save_diff(finetuned_compressed_model, save_dir):
    """diff.pt = {<module>.mask, <module>.coeff for every BinaryDiff, in module order} followed by every trainable parameter
    under its `named_parameters()` name (embeddings, norms, lm_head and the coeffs again) -- reference diff.py:66-79."""
    entries = {}
    for name, module in finetuned_compressed_model.named_modules():
        if isinstance(module, BinaryDiff):
            entries[f"{name}.mask"] = module.mask.cpu()
            entries[f"{name}.coeff"] = module.coeff.cpu()
    for name, param in finetuned_compressed_model.named_parameters():
        if param.requires_grad:
            entries[name] = param.cpu()
    torch.save(entries, save_dir)


@torch.no_grad()
def load_diff(model, diff_dir):
    """Merge a diff.pt into a dense model (reference diff.py:81-106): 1-bit deltas are added onto the matching weights on the GPU,
    stored dense tensors replace theirs, optional low-rank `.A` / `.B` pairs are added as (A @ B).T.
    The file holds tensors / Parameters only, so it is read with weights_only=True: a diff.pt is exactly the artefact users
    download from third parties in the multi-tenant flow, and the default unpickler would execute whatever it contains."""
    device = model.device
    entries = torch.load(diff_dir, map_location="cpu", weights_only=True)
    for name, module in model.named_modules():
        if f"{name}.mask" in entries:
            merge_delta_(module.weight.data, entries[f"{name}.mask"].to(device).contiguous(), entries[f"{name}.coeff"].to(device))
        elif f"{name}.weight" in entries:
            module.weight = nn.Parameter(entries[f"{name}.weight"].to(device).to(module.weight.dtype))
        elif f"{name}.A" in entries:
            low_rank = entries[f"{name}.A"].to(device) @ entries[f"{name}.B"].to(device)
            module.weight.add_(low_rank.T.to(module.weight.dtype))
    model.config.vocab_size = model.lm_head.weight.size(0)


def save_full_model(base_model_name, finetuned_model_name, diff_dir, save_dir, device):
    """Materialise base + delta as a plain HF checkpoint (reference diff.py:108-116)."""
    from .utils import get_model, get_tokenizer
    model = get_model(base_model_name, device)
    load_diff(model, diff_dir)
    model.save_pretrained(save_dir)
    get_tokenizer(finetuned_model_name).save_pretrained(save_dir)
    del model
