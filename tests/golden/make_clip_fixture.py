"""Golden vectors for the CLIP text encoder (reference call site dataloader.py:633-646).

The reference calls `transformers.CLIPTextModel(input_ids)[0]`.  transformers is importable in the build container
(not vendored, pinned by the reference at ==4.19.2 in INSTALL.md; the CLIP text tower's arithmetic is unchanged since),
so this script instantiates transformers' own CLIPTextModel with a small config, loads the seeded synthetic state dict
of distdiff_amd.weights.synthetic_text_encoder into it and records input ids -> last_hidden_state.  The committed
fixture pins oracle/sd_oracle.py::clip_text_encode (tests/test_oracle.py) and, through it, the HIP text encoder.

SDXL's text side (StableDiffusionXLPipeline.encode_prompt; SURVEY.md 8 f-4, beyond the reference): the second half records, for the two
towers of distdiff_amd.config.tiny_sdxl_config, transformers' CLIPTextModel / CLIPTextModelWithProjection run with
output_hidden_states=True -> hidden_states[-2] of both and text_embeds of the second (clip_fixture_sdxl.pt).

    python tests/golden/make_clip_fixture.py        # writes tests/golden/clip_fixture.pt and clip_fixture_sdxl.pt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from distdiff_amd.config import tiny_config, tiny_sdxl_config  # noqa: E402
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


def main_sdxl():
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    cfg = tiny_sdxl_config()
    out = {"transformers_version": transformers.__version__, "towers": []}
    g = torch.Generator().manual_seed(12)
    for which, cls in ((0, CLIPTextModel), (1, CLIPTextModelWithProjection)):
        t = cfg.text2 if which else cfg.text
        sd = synthetic_text_encoder(cfg, seed=0, which=which)
        hf = cls(CLIPTextConfig(vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                                num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                                max_position_embeddings=t.max_position_embeddings, hidden_act=t.hidden_act,
                                layer_norm_eps=t.layer_norm_eps, projection_dim=t.projection_dim or t.hidden_size,
                                # eos_token_id = 2 is the legacy setting of the published SDXL text_encoder_2 config: pooled = argmax(ids)
                                bos_token_id=0, eos_token_id=2, pad_token_id=1)).eval()
        own = set(hf.state_dict().keys())
        hsd = sd if any(k.startswith("text_model.") for k in own) else {(k[len("text_model."):] if k.startswith("text_model.") else k): v
                                                                          for k, v in sd.items()}
        missing, unexpected = hf.load_state_dict(hsd, strict=False)
        assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
        ids = torch.randint(3, t.vocab_size - 1, (4, cfg.text_len), generator=g)
        ids[:, 0] = 0
        # the eos token is the LARGEST id of the vocabulary (as in the real CLIP tokenizers); tower 1 pads with eos, tower 2 with id 0 ("!")
        for r, n in enumerate((5, 9, cfg.text_len - 1, 3)):
            ids[r, n] = t.vocab_size - 1
            ids[r, n + 1:] = 0 if which else t.vocab_size - 1
        with torch.no_grad():
            o = hf(ids, attention_mask=None, output_hidden_states=True, return_dict=True)
        rec = {"which": which, "input_ids": ids.int(), "hidden_m2": o.hidden_states[-2].float(), "n_hidden_states": len(o.hidden_states)}
        if which:
            rec["text_embeds"] = o.text_embeds.float()
        out["towers"].append(rec)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip_fixture_sdxl.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes; transformers", transformers.__version__)


if __name__ == "__main__":
    main()
    main_sdxl()
