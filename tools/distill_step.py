#!/usr/bin/env python3
"""Scale-distillation step on the HIP op: what the reference's train.py:60-88 does after
compress_diff -- AdamW over the compressed model's parameters (every BinaryDiff.coeff plus the non-`proj` parameters), MSE between
the fine-tuned teacher's logits and the compressed student's, batch 4 x length 128 -- with the student's forward AND backward running
through bitdelta_amd (forward: the fused base + delta kernel; backward: the same kernel on (W^T, S^T) for dx, the delta GEMM for
dcoeff).  There is no network here, so the model pair is synthetic: a random-init HF Llama as the base and base + N(0, sigma^2) as
the fine-tune (SURVEY.md 8d's statistics).

    python tools/distill_step.py [--hidden 1024 --inter 2752 --layers 4 --steps 10 --full-grad] [--check]

--check also runs the SAME steps on a dense fp32 student (every BinaryDiff replaced by y = x @ (W^T + coeff * S) under stock
autograd, same initial parameters, same data, same optimizer) and prints both loss traces: the test suite
(tests/test_gpu_serving.py::test_scale_distillation_step_matches_dense_fp32) asserts they agree.
"""
import argparse
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.nn as nn
import torch.nn.functional as F


class DenseDelta(nn.Module):
    """fp32 reference of one BinaryDiff under stock autograd: y = x @ (W^T + coeff * S)."""

    def __init__(self, bdiff):
        super().__init__()
        from bitdelta_amd import unpack
        self.register_buffer("wt", bdiff.base.detach().float().clone())                 # [in, out]
        self.register_buffer("s", unpack(bdiff.mask).float() * 2 - 1)                   # [in, out]
        self.coeff = nn.Parameter(bdiff.coeff.detach().clone().float())

    def forward(self, x):
        return (x.float() @ (self.wt + self.coeff * self.s)).to(x.dtype)


def build(hidden, inter, layers, heads, vocab, device, dtype, seed, sigma=5e-4):
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                      num_key_value_heads=heads, vocab_size=vocab, max_position_embeddings=512)
    base = LlamaForCausalLM(cfg).to(device=device, dtype=dtype).eval()
    fine = copy.deepcopy(base)
    with torch.no_grad():
        for p in fine.parameters():
            p.add_((torch.randn_like(p, dtype=torch.float32) * sigma).to(p.dtype))
    student = copy.deepcopy(fine)
    from bitdelta_amd.diff import compress_diff
    compress_diff(base, fine, student)
    for p in fine.parameters():
        p.requires_grad_(False)
    return base, fine, student


def dense_twin(student):
    """fp32 copy of the student with every BinaryDiff replaced by its dense autograd equivalent (same parameter values)"""
    from bitdelta_amd.diff import BinaryDiff
    twin = copy.deepcopy(student)
    for name, mod in list(twin.named_modules()):
        for cname, child in list(mod.named_children()):
            if isinstance(child, BinaryDiff):
                setattr(mod, cname, DenseDelta(child))
    return twin.float()


def run(student, teacher, batches, lr, steps):
    opt = torch.optim.AdamW(student.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, steps)
    losses, times = [], []
    for step in range(steps):
        ids = batches[step]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.inference_mode():
            t_logits = teacher(input_ids=ids).logits
        s_logits = student(input_ids=ids).logits
        loss = F.mse_loss(t_logits.clone().to(s_logits.dtype), s_logits)            # reference train.py:75-78
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        losses.append(loss.item())
    return losses, times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--inter", type=int, default=2752)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--vocab", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4)          # reference scripts: batch_size 4, max_length 128
    ap.add_argument("--length", type=int, default=128)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--full-grad", action="store_true", help="BinaryDiff.delta_input_grad = True: dx includes coeff * g.S^T (fused backward launch)")
    ap.add_argument("--check", action="store_true", help="also run the dense fp32 twin and print both traces")
    args = ap.parse_args()
    assert torch.cuda.is_available()
    dev = "cuda"
    from bitdelta_amd.diff import BinaryDiff
    BinaryDiff.delta_input_grad = bool(args.full_grad)
    base, fine, student = build(args.hidden, args.inter, args.layers, args.heads, args.vocab, dev, torch.bfloat16, seed=0)
    n_coeff = sum(1 for m in student.modules() if isinstance(m, BinaryDiff))
    g = torch.Generator(device=dev).manual_seed(1)
    batches = [torch.randint(0, args.vocab, (args.batch, args.length), device=dev, generator=g) for _ in range(args.steps)]
    twin = dense_twin(student) if args.check else None
    losses, times = run(student, fine, batches, args.lr, args.steps)
    ms = sorted(times[2:] or times)[len(times[2:] or times) // 2] * 1e3
    print(f"student: {n_coeff} BinaryDiff modules, hidden {args.hidden} inter {args.inter} layers {args.layers}; batch {args.batch} x {args.length}; "
          f"delta_input_grad={BinaryDiff.delta_input_grad}")
    print("HIP student loss trace :", " ".join(f"{v:.6e}" for v in losses))
    print(f"median step time (teacher forward + student forward/backward + AdamW): {ms:.2f} ms")
    if twin is not None:
        fine32 = copy.deepcopy(fine).float()
        l32, _ = run(twin, fine32, batches, args.lr, args.steps)
        print("dense fp32 twin trace  :", " ".join(f"{v:.6e}" for v in l32))
        rel = max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(losses, l32))
        print(f"max relative difference of the loss traces: {rel:.3e}")


if __name__ == "__main__":
    main()
