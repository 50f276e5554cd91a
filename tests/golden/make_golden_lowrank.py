"""G8 (round 6): the reference's load_diff on a diff.pt that takes ALL THREE of its branches -- 1-bit `.mask` / `.coeff` entries, dense `.weight`
replacements and low-rank `.A` / `.B` pairs (bitdelta/diff.py:88-104).  Run once in the authoring container (needs /root/reference):

    python tests/golden/make_golden_lowrank.py

Writes DATA only:  tiny_llama_lowrank.pt = {"diff": the mixed diff dict, "after": the fp16 weights the reference's load_diff leaves behind for
every module the dict touches}.  The model is the tiny Llama of G4 (tests/golden/tiny_llama_merged.pt holds its config and base weights)."""
import copy
import os
import sys

sys.path.insert(0, "/root/reference")
import torch  # noqa: E402
import bitdelta.diff as ref_d  # noqa: E402
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
gm = torch.load(os.path.join(HERE, "tiny_llama_merged.pt"), weights_only=False)
g4 = torch.load(os.path.join(HERE, "tiny_llama_diff.pt"), weights_only=False)
cfg = LlamaConfig(**{k: v for k, v in gm["config"].items() if k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers",
                                                                    "num_attention_heads", "num_key_value_heads", "max_position_embeddings")})
base = LlamaForCausalLM(cfg).bfloat16()
base.load_state_dict(gm["base_state"])
torch.manual_seed(90)
diff = {}
# layer 0: the 1-bit entries the reference's save_diff wrote (G4); layer 1: low-rank pairs for two projections, (A @ B).T is added to W [N, K]
for k, v in g4.items():
    if k.startswith("model.layers.0.") and (k.endswith(".mask") or k.endswith(".coeff")):
        diff[k] = v.detach().clone()
for name, (n_out, n_in, r) in {"model.layers.1.self_attn.q_proj": (64, 64, 4), "model.layers.1.mlp.down_proj": (64, 96, 3)}.items():
    diff[name + ".A"] = (torch.randn(n_in, r) * 0.05).half()
    diff[name + ".B"] = (torch.randn(r, n_out) * 0.05).half()
# dense replacements: a norm and the lm_head (the `.weight` branch)
diff["model.norm.weight"] = (1 + 0.1 * torch.randn(64)).bfloat16()
diff["lm_head.weight"] = (torch.randn(128, 64) * 0.02).bfloat16()
path = os.path.join(HERE, "_tmp_lowrank_diff.pt")
torch.save(diff, path)
eval_m = copy.deepcopy(base).half()
_orig_load = torch.load
torch.load = lambda f, *a, **k: _orig_load(f, weights_only=False)
ref_d.load_diff(eval_m, path)
torch.load = _orig_load
os.remove(path)
touched = sorted({k.rsplit(".", 1)[0] for k in diff})
after = {m + ".weight": eval_m.get_submodule(m).weight.detach().clone() for m in touched}
torch.save({"diff": diff, "after": after}, os.path.join(HERE, "tiny_llama_lowrank.pt"))
print("touched modules:", len(touched), "file bytes:", os.path.getsize(os.path.join(HERE, "tiny_llama_lowrank.pt")))
