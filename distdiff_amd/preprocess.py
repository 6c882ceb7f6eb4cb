"""The stage before the loop: image latents and class prompt embeddings of `SDDataset`
(dataloader.py:750-811), computed by the HIP VAE encoder / CLIP text encoder of the engine.

Host side only: PIL image decoding + the reference's transform (Resize(size, BILINEAR) -> Center/RandomCrop(size) ->
ToTensor -> Normalize([0.5],[0.5]), dataloader.py:758-765), the Hugging Face CLIP tokenizer call
(dataloader.py:35-49) and the cache file format (`save/vae_embedding/<dataset>/<model>/image_latents.pt` = list of
[1,4,L,L] tensors, dataloader.py:788-796).  No arithmetic of the encoders happens here.
"""
import os

import numpy as np
import torch

# dataloader.py:52-62
CUSTOM_TEMPLATES = {
    "dtd": "{} texture.",
    "stanford_cars": "a photo of a {}.",
    "cifar100_subset": "a photo of a {}.",
    "stl10": "a photo of a {}.",
    "imagenette2-320": "a photo of a {}.",
    "caltech-101": "a photo of a {}.",
    "pathmnist": "a colon pathological image of {}.",
    "breastmnist": "a photo of {} ultrasound image.",
    "bloodmnist": "a photo of {}, a type of cell.",
}


def resize_crop_size(w, h, size):
    """torchvision Resize(int): the smaller edge becomes `size`, the other keeps the aspect ratio (truncated)."""
    if w <= h:
        return size, int(size * h / w)
    return int(size * w / h), size


def load_image(path, size, center_crop=False, rng=None):
    """dataloader.py:803-807 + :758-765 -> float32 [3, size, size] in [-1, 1]."""
    from PIL import Image, ImageOps
    img = ImageOps.exif_transpose(Image.open(path))
    if img.mode != "RGB":
        img = img.convert("RGB")
    nw, nh = resize_crop_size(img.width, img.height, size)
    img = img.resize((nw, nh), Image.BILINEAR)
    if center_crop:
        left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    else:
        rng = rng or np.random
        top = int(rng.randint(0, nh - size + 1))
        left = int(rng.randint(0, nw - size + 1))
    img = img.crop((left, top, left + size, top + size))
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
    return (x - 0.5) / 0.5


def encode_image_latents(engine, image_paths, size, center_crop=False, seed=0, progress=None):
    """SDDataset.encode_image (dataloader.py:798-811): one `[1,4,L,L]` latent per image, sampled from the posterior
    and multiplied by scaling_factor.  Images are batched to the engine's static batch (the last batch is padded)."""
    B = engine.B
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    L, Cl = engine.cfg.latent_size, engine.cfg.vae.latent_channels
    out = []
    for i in range(0, len(image_paths), B):
        chunk = image_paths[i:i + B]
        x = torch.stack([load_image(p, size, center_crop, rng) for p in chunk])
        noise = torch.randn(len(chunk), Cl, L, L, generator=g)
        if len(chunk) < B:
            pad = B - len(chunk)
            x = torch.cat([x, x[-1:].expand(pad, -1, -1, -1)])
            noise = torch.cat([noise, torch.zeros(pad, Cl, L, L)])
        lat = engine.vae_encode(x, noise).cpu()
        out.extend(lat[j:j + 1].clone() for j in range(len(chunk)))
        if progress:
            progress(len(out), len(image_paths))
    return out


def load_or_encode_latents(engine, dataset, model_name, image_paths, size, center_crop=False, seed=0, root="save/vae_embedding"):
    """The cache logic of dataloader.py:788-796 (same path, same file format: a positional list of [1,4,L,L] tensors).  The cache
    pairs latents with images by POSITION only, so a listing that differs from the one that wrote it would silently pair latents
    with the wrong paths / labels / prompts: a cache written here stores its path list next to it (`image_latents.paths.json`) and
    is refused when the current listing differs; a cache written by the reference (no side file) must at least have its length."""
    import json
    embed_dir = os.path.join(root, dataset, model_name.replace("/", "--"))
    embed_path = os.path.join(embed_dir, "image_latents.pt")
    side = os.path.join(embed_dir, "image_latents.paths.json")
    rel = [os.path.join(os.path.basename(os.path.dirname(p)), os.path.basename(p)) for p in image_paths]
    if os.path.exists(embed_path):
        latents = torch.load(embed_path, map_location="cpu")
        if len(latents) != len(image_paths):
            raise SystemExit("latent cache %s holds %d latents but the %s train listing has %d images: it was written for a different "
                             "listing; delete it to re-encode" % (embed_path, len(latents), dataset, len(image_paths)))
        if os.path.exists(side) and json.load(open(side)) != rel:
            raise SystemExit("latent cache %s was written for a different image order than the current %s listing (see %s); delete "
                             "both files to re-encode" % (embed_path, dataset, side))
        return latents
    os.makedirs(embed_dir, exist_ok=True)
    latents = encode_image_latents(engine, image_paths, size, center_crop, seed)
    torch.save(latents, embed_path)
    json.dump(rel, open(side, "w"))
    return latents


def load_tokenizer(model_dir, revision=None, subfolder="tokenizer"):
    """AutoTokenizer.from_pretrained(path, subfolder='tokenizer') as generate_data.py:880-888 (local files only); SDXL layouts also
    carry `tokenizer_2` for the second tower."""
    from transformers import CLIPTokenizer
    return CLIPTokenizer.from_pretrained(model_dir, subfolder=subfolder, local_files_only=True)


def tokenize_prompt(tokenizer, prompt, tokenizer_max_length=None):
    """dataloader.py:35-49."""
    max_length = tokenizer_max_length if tokenizer_max_length is not None else tokenizer.model_max_length
    return tokenizer(prompt, truncation=True, padding="max_length", max_length=max_length, return_tensors="pt")


def compute_text_embeddings(engine, tokenizer, prompts, tokenizer_max_length=None):
    """compute_text_embeddings (dataloader.py:651-661) for a list of prompts -> fp32 [n, text_len, cross_dim] on the CPU,
    batched to the engine's text batch (2 * max_batch)."""
    ids = torch.cat([tokenize_prompt(tokenizer, p, tokenizer_max_length).input_ids for p in prompts]).int()
    return encode_token_ids(engine, ids)


def encode_token_ids(engine, ids):
    step = 2 * engine.B
    return torch.cat([engine.text_encode(ids[i:i + step]).cpu() for i in range(0, ids.shape[0], step)])


def encode_token_ids_sdxl(engine, ids1, ids2):
    """Both towers of an SDXL-style model (diffusers StableDiffusionXLPipeline.encode_prompt): -> (prompt embeddings
    [n, text_len, w1 + w2] = cat of the two towers' hidden_states[-2], pooled [n, projection_dim] = the second tower's text_embeds)."""
    step = 2 * engine.B
    emb, pooled = [], []
    for i in range(0, ids1.shape[0], step):
        h1 = engine.text_encode_tower(0, ids1[i:i + step])
        h2, p = engine.text_encode_tower(1, ids2[i:i + step], pooled=True)
        emb.append(torch.cat([h1, h2], dim=-1).cpu())
        pooled.append(p.cpu())
    return torch.cat(emb), torch.cat(pooled)


def compute_text_embeddings_sdxl(engine, tokenizers, prompts):
    ids = [torch.cat([tokenize_prompt(tk, p).input_ids for p in prompts]).int() for tk in tokenizers]
    return encode_token_ids_sdxl(engine, ids[0], ids[1])


def class_prompt_embeddings_sdxl(engine, tokenizers, dataset, class_names):
    """The same class prompts through both towers.  -> (class embeddings [C, T, D], uncond [1, T, D], class pooled [C, P], uncond
    pooled [1, P]); with `force_zeros_for_empty_prompt` (model_index.json of the SDXL base repo) the empty negative prompt is all
    zeros, as StableDiffusionXLPipeline.encode_prompt does when no negative prompt is given."""
    template = CUSTOM_TEMPLATES.get(dataset, "a photo of a {}.")
    emb, pooled = compute_text_embeddings_sdxl(engine, tokenizers, [template.format(x) for x in class_names] + [""])
    ue, up = emb[-1:], pooled[-1:]
    if engine.cfg.force_zeros_for_empty_prompt:
        ue, up = torch.zeros_like(ue), torch.zeros_like(up)
    return emb[:-1], ue, pooled[:-1], up


def class_prompt_embeddings(engine, tokenizer, dataset, class_names, language_enhance=False, data_root="data"):
    """classes_prompts / uncond_input of SDDataset.__init__ (dataloader.py:766-786).  Returns (per-class embeddings, uncond):
    a tensor [C, T, D], or with --language_enhance a list of [n_sentences_c, T, D] tensors from `data/<dataset>_le.pkl`
    (dict class name -> sentences, `_` -> ' ' in the keys, :769-778)."""
    if language_enhance:
        template = np.load(os.path.join(data_root, "%s_le.pkl" % dataset), allow_pickle=True)
        template = {k.replace("_", " "): v for k, v in template.items()}
        sentences = [list(template[x]) for x in class_names]
        flat = [p for ss in sentences for p in ss] + [""]
        emb = compute_text_embeddings(engine, tokenizer, flat)
        out, o = [], 0
        for ss in sentences:
            out.append(emb[o:o + len(ss)])
            o += len(ss)
        return out, emb[-1:]
    template = CUSTOM_TEMPLATES.get(dataset, "a photo of a {}.")
    emb = compute_text_embeddings(engine, tokenizer, [template.format(x) for x in class_names] + [""])
    return emb[:-1], emb[-1:]
