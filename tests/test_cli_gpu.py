"""GPU test of the drop-in CLI end to end on the tiny synthetic configuration: PNG tree, resume, shard disjointness."""
import os

import pytest

pytestmark = pytest.mark.gpu


def _run(tmp_path, split, total, extra=()):
    from distdiff_amd import generate_data as G
    out = str(tmp_path / "out")
    argv = ["--synthetic", "6", "--tiny", "--synthetic_classes", "2", "--output_dir", out, "--train_batch_size", "1", "--engine_batch", "4", "--steps", "10",
            "--total_split", str(total), "--split", str(split), "--num_images_per_prompt", "2", "--guidance_type", "transform_guidance",
            "--guidance_step", "4", "--guidance_period", "2", "--strength", "0.5", "--constraint_value", "0.2",
            "--optimize_targets", "global_prototype-local_prototype", "--K", "3"] + list(extra)
    assert G.main(argv) == 0
    return out


def test_cli_writes_png_tree_and_resumes(hip_lib, tmp_path, capsys):
    from PIL import Image
    out = _run(tmp_path, 0, 2)
    files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs)
    # shard 0 of 2 over 6 images = indices 0..2, two expansions each
    assert len(files) == 6 and all(f.endswith(".png") for f in files)
    assert sorted(os.path.basename(f) for f in files) == sorted("image_%04d_expand_%d.png" % (i, j) for i in range(3) for j in range(2))
    im = Image.open(files[0])
    assert im.size == (128, 128) and im.mode == "RGB"
    capsys.readouterr()
    _run(tmp_path, 0, 2)                       # second run: everything exists -> skipped (generate_data.py:1132-1143)
    assert capsys.readouterr().out.count("exists, so skipped") == 6
    out2 = _run(tmp_path, 1, 2)                # the other shard adds the remaining images, no overlap
    files2 = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(out2) for f in fs)
    assert len(files2) == 12


def test_cli_synthetic_encode_runs_both_encoders(hip_lib, tmp_path):
    """--synthetic_encode: latents from the HIP VAE encoder, prompt embeddings from the HIP text encoder, then the same loop."""
    out = _run(tmp_path, 0, 1, extra=["--synthetic_encode"])
    files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs)
    assert len(files) == 12


def test_cli_sdxl_structure_synthetic(hip_lib, tmp_path):
    """--synthetic_arch sdxl (BASELINE configs[4] structure at test size): the CLI drives the SDXL UNet with its text_time conditioning --
    pooled text embeddings + add_time_ids through dd_set_added_cond per engine batch (the engine refuses to step without them) -- first
    from seeded embeddings, then with --synthetic_encode from BOTH text towers (hidden_states[-2] of each, pooled text_embeds)."""
    out = _run(tmp_path, 0, 1, extra=["--synthetic_arch", "sdxl"])
    files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs)
    assert len(files) == 12
    out = _run(tmp_path / "enc", 0, 1, extra=["--synthetic_arch", "sdxl", "--synthetic_encode"])
    files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs)
    assert len(files) == 12


def test_latent_cache_format_and_reuse(hip_lib, tmp_path):
    """dataloader.py:788-811: images on disk -> list of [1,4,L,L] latents saved to image_latents.pt; a second call loads the file."""
    import numpy as np
    import torch
    from PIL import Image
    from distdiff_amd.config import tiny_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.preprocess import load_image, load_or_encode_latents
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    cfg = tiny_config(max_batch=2)
    w = synthetic_weights(cfg, seed=0, num_classes=2, encoders=True)
    rng = np.random.RandomState(0)
    paths = []
    for i, (wd, ht) in enumerate([(160, 200), (300, 140), (128, 128)]):     # portrait, landscape, exact
        p = str(tmp_path / ("img%d.png" % i))
        Image.fromarray(rng.randint(0, 255, (ht, wd, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    size = 8 * cfg.latent_size
    eng = Engine(cfg, w, enable_grad=False, max_guidance_period=1)
    try:
        root = str(tmp_path / "save")
        lat = load_or_encode_latents(eng, "toy", "org/model", paths, size, center_crop=True, seed=5, root=root)
        f = os.path.join(root, "toy", "org--model", "image_latents.pt")
        assert os.path.exists(f)
        assert len(lat) == 3 and all(tuple(x.shape) == (1, 4, cfg.latent_size, cfg.latent_size) for x in lat)
        again = load_or_encode_latents(eng, "toy", "org/model", paths, size, root=root)   # served from the cache
        assert all(torch.equal(a, b) for a, b in zip(lat, again))
        # the cache is positional: a listing of another length or another order is refused instead of silently mis-pairing
        for other in (["/nonexistent.png"], paths[::-1]):
            try:
                load_or_encode_latents(eng, "toy", "org/model", other, size, root=root)
                raise AssertionError("a cache written for a different listing was accepted")
            except SystemExit:
                pass
        # against the oracle on the same preprocessed pixels and the same seeded noise stream (3 images = batches of 2 + 1)
        g = torch.Generator().manual_seed(5)
        x = torch.stack([load_image(p, size, True) for p in paths])
        n = torch.cat([torch.randn(2, 4, cfg.latent_size, cfg.latent_size, generator=g), torch.randn(1, 4, cfg.latent_size, cfg.latent_size, generator=g)])
        with torch.no_grad():
            ref, _ = O.vae_encode(cfg, w["vae"], x, n)
        got = torch.cat(lat)
        assert ((got - ref).norm() / ref.norm()).item() < 0.03
    finally:
        eng.close()


def test_prototype_extraction_from_image_files(hip_lib, tmp_path):
    """8f-1 end to end (dataloader.py:664-747): image files -> HIP guide features -> class / group prototypes, against the oracle's
    guide on the same preprocessed pixels and the same sklearn clustering."""
    import types
    import numpy as np
    import torch
    from PIL import Image
    from distdiff_amd.config import tiny_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.prototypes import _load_image, extract_prototypes_with_encoder, prototypes_from_features
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    cfg = tiny_config(max_batch=4)
    w = synthetic_weights(cfg, seed=0, num_classes=2)
    rng = np.random.RandomState(4)
    paths, targets = [], []
    for i in range(10):                                    # 2 classes x 5 images of different sizes
        p = str(tmp_path / ("c%d_%d.png" % (i % 2, i)))
        Image.fromarray(rng.randint(0, 255, (40 + 7 * i, 64 + 3 * i, 3), dtype=np.uint8)).save(p)
        paths.append(p)
        targets.append(i % 2)
    class _DS:
        image_paths, class_names = paths, ["a", "b"]

        def __len__(self):
            return len(self.image_paths)

    ds = _DS()
    ds.targets = torch.tensor(targets)
    eng = Engine(cfg, w, enable_grad=False, max_guidance_period=1)
    try:
        Pc, Pg = extract_prototypes_with_encoder(types.SimpleNamespace(K=2), eng, ds)
    finally:
        eng.close()
    assert tuple(Pc.shape) == (2, cfg.guide.feature_dim) and tuple(Pg.shape) == (2, 2, cfg.guide.feature_dim)
    guide = O.GuideOracle(cfg, w["guide"])
    x = torch.stack([_load_image(p, cfg.guide.input_size) for p in paths])
    with torch.no_grad():
        f = guide.encode_image(x)          # the HIP guide is exact fp32 (guide_f32.hip)
    f = f / f.norm(dim=-1, keepdim=True)
    g_ref, l_ref = prototypes_from_features(f.numpy(), np.array(targets), 2, 2)
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a) - b) / np.linalg.norm(b))
    assert rel(Pc.numpy(), g_ref) < 1e-4
    # group prototypes are compared as sets per class (cluster labels are arbitrary)
    for c in range(2):
        d = np.linalg.norm(Pg[c].numpy()[:, None] - l_ref[c][None], axis=-1)
        assert min(d[0, 0] + d[1, 1], d[0, 1] + d[1, 0]) < 1e-3 * np.linalg.norm(l_ref[c])


def _write_tokenizer(d):
    """A byte-level CLIP tokenizer without merges: every character is a token (vocab.json + merges.txt as CLIPTokenizer expects)."""
    import json
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    chars = [chr(c) for c in cs]
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    os.makedirs(d)
    json.dump(vocab, open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n")
    json.dump({"model_max_length": 13, "bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>",
               "pad_token": "<|endoftext|>"}, open(os.path.join(d, "tokenizer_config.json"), "w"))
    return len(vocab)


def test_cli_on_a_model_directory_and_image_files(hip_lib, tmp_path, monkeypatch):
    """The NON-synthetic product path end to end: a local Hugging Face-layout model directory (config.json + safetensors with the real
    key names, legacy VAE attention names, tokenizer files), a reference-format guide checkpoint (`module.` prefix, real ResNet-50
    shapes) and a 2-class tree of JPEG files -> from_model_dir + load_safetensors_dir + create_model(weight_path) + tokenizer + HIP text
    encoder + HIP VAE encoder (latent cache written) + prototype extraction + transform-guided expansion -> PNG files; a second run
    resumes from the files and the latent cache."""
    import numpy as np
    import torch
    from PIL import Image
    from distdiff_amd import generate_data as G
    from distdiff_amd.config import TextConfig, guide_config, sd15_config, tiny_config
    from distdiff_amd.weights import synthetic_guide, synthetic_weights
    from test_checkpoints import write_model_dir
    monkeypatch.chdir(tmp_path)
    model = str(tmp_path / "sd-tiny")
    cfg = tiny_config(max_batch=4)
    V = _write_tokenizer(os.path.join(model, "tokenizer"))
    cfg.text = TextConfig(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                          max_position_embeddings=13)
    write_model_dir(model, cfg, synthetic_weights(cfg, seed=0, num_classes=2, encoders=True))
    gcfg = sd15_config()
    gcfg.guide = guide_config("resnet50")
    ck = str(tmp_path / "model_best.pth.tar")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in synthetic_guide(gcfg, seed=1, num_classes=2).items()}, "acc": 0,
                "best_acc": 0, "optimizer": {}}, ck)
    rng = np.random.RandomState(0)
    for c in ("cat_a", "dog_b"):
        d = tmp_path / "data" / "toy" / "train" / c
        d.mkdir(parents=True)
        for i in range(3):
            Image.fromarray(rng.randint(0, 255, (150 + 10 * i, 140 + 20 * i, 3), dtype=np.uint8)).save(str(d / ("im%d.jpg" % i)), quality=90)
    out = str(tmp_path / "out")
    argv = ["--pretrained_model_name_or_path", model, "-d", "toy", "-a", "resnet50", "--encoder_weight_path", ck, "--data_root",
            str(tmp_path / "data"), "--output_dir", out, "--resolution", "128", "--steps", "10", "--strength", "0.5", "--guidance_type",
            "transform_guidance", "--guidance_step", "4", "--guidance_period", "2", "--optimize_targets", "global_prototype-local_prototype",
            "--K", "2", "--train_batch_size", "1", "--engine_batch", "4", "--num_images_per_prompt", "2", "--constraint_value", "0.2",
            "--offset_noise", "--total_split", "1", "--split", "0"]
    assert G.main(argv) == 0
    files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs)
    assert len(files) == 12 and os.path.basename(os.path.dirname(files[0])) == "cat a"      # `_` -> ' ' in the class directory (:1232)
    im = np.asarray(Image.open(files[0]))
    assert im.shape == (128, 128, 3) and im.std() > 1.0
    cache = os.path.join("save", "vae_embedding", "toy", model.replace("/", "--"), "image_latents.pt")
    assert os.path.exists(cache) and len(torch.load(cache)) == 6
    mt = {f: os.path.getmtime(f) for f in files}
    os.remove(files[3])
    assert G.main(argv) == 0                                 # resume: only the missing (batch, expand index) group is regenerated
    assert os.path.exists(files[3]) and all(os.path.getmtime(f) == mt[f] for f in files if f != files[3])


def test_cli_on_an_sdxl_model_directory(hip_lib, tmp_path, monkeypatch):
    """The SDXL layout through the non-synthetic product path: unet/config.json with per-level heads, transformer depths and text_time
    conditioning, text_encoder/ + text_encoder_2/ (with text_projection), tokenizer/ + tokenizer_2/, model_index.json; a class tree whose
    directories are WordNet-style ids named by classnames.txt (ImageNet subsets, BASELINE configs[4]).  The class prompts go through both
    tokenizers and both HIP towers; pooled embeddings and time ids reach the UNet; PNGs come out under the mapped class names."""
    import numpy as np
    import torch
    from PIL import Image
    from distdiff_amd import generate_data as G
    from distdiff_amd.config import guide_config, sd15_config, tiny_sdxl_config
    from distdiff_amd.weights import synthetic_guide, synthetic_weights
    from test_checkpoints import write_model_dir
    monkeypatch.chdir(tmp_path)
    model = str(tmp_path / "sdxl-tiny")
    cfg = tiny_sdxl_config(max_batch=4)
    V = _write_tokenizer(os.path.join(model, "tokenizer"))
    V2 = _write_tokenizer(os.path.join(model, "tokenizer_2"))
    cfg.text.vocab_size, cfg.text2.vocab_size = V, V2
    write_model_dir(model, cfg, synthetic_weights(cfg, seed=0, num_classes=2, encoders=True))
    gcfg = sd15_config()
    gcfg.guide = guide_config("resnet50")
    ck = str(tmp_path / "model_best.pth.tar")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in synthetic_guide(gcfg, seed=1, num_classes=2).items()}, "acc": 0,
                "best_acc": 0, "optimizer": {}}, ck)
    rng = np.random.RandomState(0)
    for c in ("n01440764", "n01443537"):
        d = tmp_path / "data" / "imagenet100" / "train" / c
        d.mkdir(parents=True)
        for i in range(2):
            Image.fromarray(rng.randint(0, 255, (150 + 10 * i, 140 + 20 * i, 3), dtype=np.uint8)).save(str(d / ("im%d.jpg" % i)), quality=90)
    open(str(tmp_path / "data" / "imagenet100" / "classnames.txt"), "w").write("n01440764 tench, Tinca tinca\nn01443537 goldfish, Carassius auratus\n")
    out = str(tmp_path / "out")
    argv = ["--pretrained_model_name_or_path", model, "-d", "imagenet100", "-a", "resnet50", "--encoder_weight_path", ck, "--data_root",
            str(tmp_path / "data"), "--output_dir", out, "--resolution", "128", "--steps", "10", "--strength", "0.5", "--guidance_type",
            "transform_guidance", "--guidance_step", "4", "--guidance_period", "2", "--optimize_targets", "global_prototype-local_prototype",
            "--K", "2", "--train_batch_size", "1", "--engine_batch", "4", "--num_images_per_prompt", "2", "--constraint_value", "0.2",
            "--total_split", "1", "--split", "0"]
    assert G.main(argv) == 0
    files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs)
    assert len(files) == 8 and sorted({os.path.basename(os.path.dirname(f)) for f in files}) == ["goldfish", "tench"]
    im = np.asarray(Image.open(files[0]))
    assert im.shape == (128, 128, 3) and im.std() > 1.0
