"""Golden vectors for the CLIP text encoder (reference call site dataloader.py:633-646).

The reference calls `transformers.CLIPTextModel(input_ids)[0]`.  transformers is importable in the build container
(not vendored, pinned by the reference at ==4.19.2 in INSTALL.md; the CLIP text tower's arithmetic is unchanged since),
so this script instantiates transformers' own CLIPTextModel with a small config, loads the seeded synthetic state dict
of distdiff_amd.weights.synthetic_text_encoder into it and records input ids -> last_hidden_state.  The committed
fixture pins oracle/sd_oracle.py::clip_text_encode (tests/test_oracle.py) and, through it, the HIP text encoder.

    python tests/golden/make_clip_fixture.py        # writes tests/golden/clip_fixture.pt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from distdiff_amd.config import tiny_config  # noqa: E402
from distdiff_amd.weights import synthetic_text_encoder  # noqa: E402


def main():
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel
    out = {"transformers_version": transformers.__version__, "cases": []}
    for act in ("quick_gelu", "gelu"):
        cfg = tiny_config()
        cfg.text.hidden_act = act
        t = cfg.text
        sd = synthetic_text_encoder(cfg, seed=0)
        hf = CLIPTextModel(CLIPTextConfig(vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                                          num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                                          max_position_embeddings=t.max_position_embeddings, hidden_act=act,
                                          layer_norm_eps=t.layer_norm_eps, projection_dim=t.hidden_size,
                                          # the tokenizer's real ids (bos 49406 / eos 49407) do not exist in the tiny vocab
                                          bos_token_id=0, eos_token_id=2, pad_token_id=1)).eval()
        # SD-1.x checkpoints (and transformers 4.x) prefix every key with `text_model.`; transformers 5.x dropped the wrapper
        own = set(hf.state_dict().keys())
        hsd = sd if any(k.startswith("text_model.") for k in own) else {k[len("text_model."):]: v for k, v in sd.items()}
        missing, unexpected = hf.load_state_dict(hsd, strict=False)
        assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
        g = torch.Generator().manual_seed(11)
        ids = torch.randint(3, t.vocab_size, (4, cfg.text_len), generator=g)
        ids[:, 0] = 0
        ids[1, 5:] = 2     # a short prompt: eos then padding with the same id, as CLIPTokenizer pads SD prompts
        with torch.no_grad():
            ref = hf(ids, attention_mask=None, return_dict=False)[0]
        out["cases"].append({"hidden_act": act, "input_ids": ids.int(), "last_hidden_state": ref.float()})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip_fixture.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes; transformers", transformers.__version__)


if __name__ == "__main__":
    main()
