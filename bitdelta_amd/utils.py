"""Loader glue used by save_full_model (mirror of the reference's bitdelta/utils.py:80-121, minus argparse).
Out of the hot path; needs `transformers` and local checkpoints."""
import torch


def get_model(model_name, device, memory_map=None):
    import transformers
    if device == "auto" or isinstance(device, list):
        return transformers.AutoModelForCausalLM.from_pretrained(
            model_name, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True, device_map="auto",
            max_memory=memory_map)
    return transformers.AutoModelForCausalLM.from_pretrained(
        model_name, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True).to(device)


def get_tokenizer(tokenizer_name):
    import transformers
    tokenizer = transformers.AutoTokenizer.from_pretrained(tokenizer_name, use_fast=False)
    if tokenizer.pad_token_id is None:
        if tokenizer.eos_token_id is not None:
            tokenizer.pad_token_id = tokenizer.eos_token_id
        else:
            tokenizer.pad_token_id = 0
    return tokenizer
