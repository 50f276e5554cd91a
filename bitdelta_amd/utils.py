"""Checkpoint loading used by save_full_model only (the reference's counterpart is bitdelta/utils.py:80-121; its argparse /
device-string / memory-map helpers belong to the drivers, which are out of scope -- SURVEY.md section 2).

Needs `transformers` and a local checkpoint directory (no network on the GPU boxes); nothing on the hot path imports this.
"""
import torch


def get_model(model_name, device, memory_map=None):
    """bf16 causal LM on `device`; `device="auto"` (or a list of devices) spreads the layers with accelerate's device map."""
    from transformers import AutoModelForCausalLM
    kwargs = dict(torch_dtype=torch.bfloat16, low_cpu_mem_usage=True)
    if device == "auto" or isinstance(device, (list, tuple)):
        return AutoModelForCausalLM.from_pretrained(model_name, device_map="auto", max_memory=memory_map, **kwargs)
    return AutoModelForCausalLM.from_pretrained(model_name, **kwargs).to(device)


def get_tokenizer(tokenizer_name):
    """Slow tokenizer with a pad token (falls back to EOS, then to id 0), as the reference's drivers expect."""
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(tokenizer_name, use_fast=False)
    if tok.pad_token_id is None:
        tok.pad_token_id = tok.eos_token_id if tok.eos_token_id is not None else 0
    return tok
