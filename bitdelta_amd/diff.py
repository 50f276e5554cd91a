"""MI355X drop-in for the reference's ``bitdelta/diff.py``: same classes/functions, same ``diff.pt`` format.

    BinaryDiff(base, finetune)        reference :8-39    buffers ``mask`` int32 [in/32,out], ``base`` = W.T view,
                                                          parameter ``coeff`` fp32 0-dim; state_dict order coeff, mask, base
    compress_diff / save_diff / load_diff / save_full_model      reference :41-116

What changes is only WHERE the arithmetic runs: construction is one fused HIP pass (sign + mean|diff| + transposed
pack), inference forward is one fused HIP launch (base GEMM + delta GEMM + scale + add), load_diff's merge is one HIP
pass without the reference's int64/fp32 [in,out] temporaries.
"""
import gc

import torch
import torch.nn as nn

from . import _lib
from ._lib import DTYPE_CODE, check, lib, ptr, require_gpu, stream_ptr, workspace
from .binary_gemm_kernel import binary_bmm, binary_linear, delta_bmm, pack, unpack  # noqa: F401  (re-exported like the reference, diff.py:5)


def binarize(base, finetune):
    """mask, coeff of BinaryDiff.__init__ (reference diff.py:11-16) in one pass over the two [out,in] weights."""
    require_gpu(base, finetune)
    assert base.shape == finetune.shape and base.dim() == 2 and base.dtype == finetune.dtype
    assert base.dtype in (torch.float16, torch.bfloat16), "weights must be fp16 or bf16"
    if base.stride(1) != 1:
        base = base.contiguous()
    if finetune.stride(1) != 1 or finetune.stride(0) != base.stride(0):
        finetune = finetune.contiguous()
        base = base.contiguous()
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


class BinaryDiff(nn.Module):
    def __init__(self, base, finetune):
        super().__init__()
        mask, quantile = binarize(base, finetune)

        self.register_buffer("mask", mask)
        self.register_buffer("base", base.T)
        self.register_parameter(
            "coeff",
            nn.Parameter(quantile.detach().clone().to(torch.float32).requires_grad_(True)),
        )
        del base, finetune

    def _weight_nk(self):
        w = self.base.T                      # [out, in]; contiguous when `base` is still the .T view it was built as
        if w.stride(1) != 1:
            w = w.contiguous()               # e.g. after a state_dict round trip that densified the buffer
        return w

    def forward(self, x):
        # [B, seq, in] @ [in, out] + coeff * ([B, seq, in] @ S),  S = +-1 from the packed mask (broadcast, never repeated)
        shape = x.shape
        x3 = x.reshape(1, -1, shape[-1])     # one [B*seq, in] problem: the mask is shared by every row
        if x3.stride(-1) != 1:
            x3 = x3.contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or self.coeff.requires_grad):
            # training form: identical composition to the reference (diff.py:39) so autograd behaves the same --
            # d/dcoeff flows through `coeff * c`, d/dx only through `x @ base` (the kernel output carries no grad_fn).
            c = delta_bmm(x3.detach(), self.mask.unsqueeze(0), round_mode=1)
            y = x3 @ self.base + self.coeff * c
        else:
            y = binary_linear(x3, self._weight_nk(), self.mask.unsqueeze(0), self.coeff.reshape(1, 1))
        return y.reshape(*shape[:-1], y.shape[-1])


BinaryLinear = BinaryDiff   # name used by BASELINE.json's north star; the reference class is BinaryDiff


def compress_diff(base_model, finetuned_model, finetuned_compressed_model):
    def compress_submodule(name, subname, module, submodule):
        target_device = submodule.weight.device

        base_weight = base_model.get_submodule(f"{name}.{subname}").weight.detach().to(target_device)
        finetuned_weight = finetuned_model.get_submodule(f"{name}.{subname}").weight.detach().to(target_device)

        compressed = BinaryDiff(
            base=base_weight,
            finetune=finetuned_weight,
        ).to(target_device)

        del submodule, base_weight
        setattr(module, subname, None)
        gc.collect()
        torch.cuda.empty_cache()
        setattr(module, subname, compressed)

    # same selection rule as the reference (diff.py:60-64)
    for name, module in finetuned_compressed_model.named_modules():
        if "mlp" in name or "self_attn" in name:
            for subname, submodule in module.named_children():
                if "proj" in subname:
                    compress_submodule(name, subname, module, submodule)


def save_diff(finetuned_compressed_model, save_dir):
    diff_dict = {}

    for name, module in finetuned_compressed_model.named_modules():
        if isinstance(module, BinaryDiff):
            diff_dict[name + ".mask"] = module.mask.cpu()
            diff_dict[name + ".coeff"] = module.coeff.cpu()

    for name, param in finetuned_compressed_model.named_parameters():
        if param.requires_grad:
            diff_dict[name] = param.cpu()

    torch.save(diff_dict, save_dir)


@torch.no_grad()
def load_diff(model, diff_dir):
    device = model.device
    diff_dict = torch.load(diff_dir, weights_only=False)

    for name, module in model.named_modules():
        if name + ".mask" in diff_dict:
            coeff = diff_dict[name + ".coeff"].to(device)
            mask = diff_dict[name + ".mask"].to(device)
            merge_delta_(module.weight.data, mask.contiguous(), coeff)
        elif name + ".weight" in diff_dict:
            module.weight = nn.Parameter(diff_dict[name + ".weight"].to(device).to(module.weight.dtype))
        elif name + '.A' in diff_dict:
            A = diff_dict[name + '.A'].to(device)
            B = diff_dict[name + '.B'].to(device)

            mask = (A @ B).T
            module.weight.add_(mask.to(module.weight.dtype))

    model.config.vocab_size = model.lm_head.weight.size(0)


def save_full_model(base_model_name, finetuned_model_name, diff_dir, save_dir, device):
    from .utils import get_model, get_tokenizer
    base_model = get_model(base_model_name, device)
    tokenizer = get_tokenizer(finetuned_model_name)
    load_diff(base_model, diff_dir)

    base_model.save_pretrained(save_dir)
    tokenizer.save_pretrained(save_dir)

    del base_model
