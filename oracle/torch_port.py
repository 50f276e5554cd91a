"""CPU port of the reference's hot path written with the same torch ops the reference uses (multi-threaded).

TEST / BASELINE INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests): the reference's Python cannot travel to
the GPU box, so this file restates -- it does not copy -- the three lines that make up the path:
    unpack            bitdelta/binary_gemm_kernel.py:34-46   (shift / and / bool)
    BinaryDiff.forward  bitdelta/diff.py:38-39              x @ base + coeff * binary_bmm(x, mask)
with binary_bmm replaced by its CPU-executable meaning (a.dtype matmul with the unpacked +-1 matrix, fp32 -> fp16 ->
a.dtype epilogue, binary_gemm_kernel.py:270-272, :287, :314), because the Triton kernel itself needs a GPU.
It is checked against the C oracle in tests/test_oracle_golden.py.
"""
import torch


def unpack32(words):
    shift = torch.arange(32, device=words.device)
    x = words.reshape(-1, words.shape[-2], 1, words.shape[-1])
    x = (x >> shift[None, None, :, None]) & 0x1
    return x.reshape(*words.shape[:-2], -1, words.shape[-1]).bool()


def forward_unpack_in_loop(x, w_nk, mask, coeff):
    """Variant 1 of BASELINE.md section 3: unpack inside the timed region, as a CPU run of the reference would do."""
    s = (unpack32(mask) * 2 - 1).to(x.dtype)
    delta = (x.float() @ s.float()).half().to(x.dtype) if x.dtype == torch.bfloat16 else (x @ s)
    return x @ w_nk.T + coeff.to(x.dtype) * delta


def forward_preunpacked(x, w_nk, s_kn, coeff):
    """Variant 2: the pure torch.matmul baseline (signs already expanded to x.dtype)."""
    return x @ w_nk.T + coeff.to(x.dtype) * (x @ s_kn)
