"""CPU tests of the drop-in CLI host logic (SURVEY.md section 8c pin 5): flags/defaults, shard ranges, output paths,
skip-if-exists resume and the per-image call sequence, against a recording fake engine (no GPU)."""
import pytest
import os

import torch

from distdiff_amd import generate_data as G
from distdiff_amd.config import tiny_config
from distdiff_amd.scheduler import DDIMSchedule


class FakeEngine:
    device = torch.device("cpu")

    def __init__(self, B):
        self.B, self.calls = B, []

    def set_prompt(self, emb):
        self.calls.append(("set_prompt", tuple(emb.shape)))

    def expand(self, lat, noise, e, b, tg, si, gt, gfirst, gcount, want_image=True):
        self.calls.append(("expand", tuple(lat.shape), si, gt, gfirst, gcount, tg.tolist()))
        B = lat.shape[0]
        return lat.clone(), torch.rand(B, 3, 16, 16), torch.tensor([1.25])


def test_reference_defaults_and_script_of_record_flags():
    a = G.parse_args([])
    assert (a.strength, a.constraint_value, a.guidance_step, a.guidance_period, a.total_split, a.num_images_per_prompt) == (0.9, 0.8, 1, 1, 8, 4)
    assert (a.guidance_scale, a.rho, a.gs, a.ls, a.K, a.seed, a.train_batch_size, a.steps) == (7.5, 10.0, 1.0, 1.0, 3, 42, 2, 50)
    assert a.pretrained_model_name_or_path == "CompVis/stable-diffusion-v1-4" and a.dataset == "caltech-101"
    rec = ("--guidance_type=transform_guidance -a resnet50 -d caltech-101 --output_dir OUT --pretrained_model_name_or_path M "
           "--gradient_checkpointing --K 3 --train_batch_size 1 --optimize_targets global_prototype-local_prototype --strength 0.5 "
           "--num_images_per_prompt 5 --guidance_step 20 --guidance_period 2 --encoder_weight_path W --guidance_scale 7.5 "
           "--constraint_value 0.2 --rho 10.0 --total_split 4 --split 3").split()
    a = G.parse_args(rec)
    assert a.guidance_type == "transform_guidance" and a.arch == "resnet50" and a.split == 3 and a.total_split == 4
    assert a.do_classifier_free_guidance is True
    # DreamBooth leftovers are accepted and ignored
    G.parse_args(["--mixed_precision", "fp16", "--report_to", "tensorboard", "--enable_xformers_memory_efficient_attention"])


def test_output_path_layout():
    assert G.output_path("out", "sea horse", "/d/101_ObjectCategories/sea_horse/image_0007.jpg", 3) == "out/sea horse/image_0007_expand_3.png"


def test_loop_shards_paths_resume_and_call_sequence(tmp_path):
    cfg = tiny_config(max_batch=2)
    ds = G.ExpansionDataset.synthetic(cfg, n=7, n_classes=2, seed=0)
    sched = DDIMSchedule(cfg.scheduler)
    sched.set_timesteps(50)
    out = str(tmp_path / "out")
    args = G.parse_args(["--synthetic", "7", "--output_dir", out, "--train_batch_size", "2", "--total_split", "2", "--split", "1",
                         "--num_images_per_prompt", "2", "--guidance_type", "transform_guidance", "--guidance_step", "20",
                         "--guidance_period", "2", "--strength", "0.5"])
    eng = FakeEngine(2)
    written = []
    n = G.run_expansion(args, eng, sched, ds, writer=lambda img, p: (written.append(p), os.makedirs(os.path.dirname(p), exist_ok=True),
                                                                      open(p, "wb").close()))
    # shard 1 of 2 over 7 images: ceil(7/2)=4 -> indices 4..6 (generate_data.py:1003-1007) -> batches [4,5], [6]
    assert n == 3 * 2 and len(written) == 6
    exp = [c for c in eng.calls if c[0] == "expand"]
    # units in the reference order: ([4,5],0) ([4,5],1) ([6],0) ([6],1) -> six independent units packed into engine batches of 2
    assert len(exp) == 3
    assert [c[6] for c in exp] == [[1, 1], [1, 1], [1, 1]]
    assert all(c[2] == 25 and c[3] == "transform_guidance" and c[4] == 30 and c[5] == 2 for c in exp)
    assert exp[0][1] == (2, 4, cfg.latent_size, cfg.latent_size)
    assert sorted(os.path.basename(p) for p in written) == sorted(
        "image_%04d_expand_%d.png" % (i, j) for i in (4, 5, 6) for j in (0, 1))
    assert all(os.path.dirname(p).endswith("class 1") for p in written)      # class-sorted dataset: shard 1 is the second class
    # resume: everything exists -> no engine call at all
    eng2 = FakeEngine(2)
    assert G.run_expansion(args, eng2, sched, ds, writer=lambda img, p: None) == 0
    assert not eng2.calls
    # --first_image_index skips earlier expand indices
    args.first_image_index = 1
    for p in written:
        os.remove(p)
    eng3 = FakeEngine(2)
    G.run_expansion(args, eng3, sched, ds, writer=lambda img, p: None)
    assert len([c for c in eng3.calls if c[0] == "expand"]) == 2      # 3 units (indices 4,5,6 x expand index 1) -> 2 engine batches


def test_prototype_builder_known_answer():
    from distdiff_amd.prototypes import prototypes_from_features
    import numpy as np
    rng = np.random.RandomState(0)
    centers = np.array([[5, 0, 0], [0, 5, 0], [0, 0, 5]], dtype=np.float32)
    feats, tg = [], []
    for c in range(2):
        for k in range(3):
            feats.append(centers[k] * (c + 1) + 0.01 * rng.randn(4, 3).astype(np.float32))
            tg += [c] * 4
    feats = np.concatenate(feats)
    g, l = prototypes_from_features(feats, np.array(tg), 2, 3)
    assert g.shape == (2, 3) and l.shape == (2, 3, 3)
    for c in range(2):
        got = sorted(map(tuple, np.round(l[c] / (c + 1) / 5).astype(int).tolist()))
        assert got == [(0, 0, 1), (0, 1, 0), (1, 0, 0)]
        assert np.allclose(g[c], feats[np.array(tg) == c].mean(0))


def test_average_linkage_labels_match_sklearn():
    """The prototype builder's own clustering against the call the reference makes (dataloader.py:704-717: sklearn
    AgglomerativeClustering(n_clusters=K, linkage='average')): identical labels, including their numbering, on unit-norm features."""
    import numpy as np
    cluster = pytest.importorskip("sklearn.cluster")
    from distdiff_amd.prototypes import average_linkage_labels
    rng = np.random.default_rng(0)
    for trial in range(40):
        n, d = int(rng.integers(3, 100)), int(rng.integers(2, 48))
        K = int(rng.integers(1, min(n, 6) + 1))
        X = rng.standard_normal((n, d)).astype(np.float32)
        if trial % 3 == 0:
            X[n // 2:] += 0.5
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        ref = cluster.AgglomerativeClustering(n_clusters=K, linkage="average").fit(X).labels_
        assert np.array_equal(average_linkage_labels(X, K), ref), (n, d, K)
    assert average_linkage_labels(np.zeros((1, 4), np.float32), 1).tolist() == [0]
    with pytest.raises(ValueError):
        average_linkage_labels(np.zeros((2, 4), np.float32), 3)


def test_image_transform_matches_reference_pipeline(tmp_path):
    """Resize(size, BILINEAR) -> CenterCrop(size) -> ToTensor -> Normalize([0.5],[0.5]) (dataloader.py:758-765)."""
    import numpy as np
    import torch
    from PIL import Image
    from distdiff_amd.preprocess import CUSTOM_TEMPLATES, load_image, resize_crop_size
    assert resize_crop_size(300, 200, 64) == (96, 64) and resize_crop_size(200, 300, 64) == (64, 96) and resize_crop_size(50, 50, 64) == (64, 64)
    rng = np.random.RandomState(1)
    p = str(tmp_path / "a.png")
    Image.fromarray(rng.randint(0, 255, (90, 150, 3), dtype=np.uint8)).save(p)
    x = load_image(p, 64, center_crop=True)
    assert x.shape == (3, 64, 64) and x.dtype == torch.float32 and float(x.min()) >= -1.0 and float(x.max()) <= 1.0
    # restated by hand: the smaller edge (90) becomes 64, the width int(64 * 150 / 90) = 106, centre crop offset round((106-64)/2) = 21
    ref = Image.open(p).resize((106, 64), Image.BILINEAR).crop((21, 0, 85, 64))
    ref = torch.from_numpy(np.asarray(ref).copy()).permute(2, 0, 1).float() / 255.0 * 2 - 1
    assert torch.equal(x, ref)
    # grey-scale files are converted to RGB (dataloader.py:805-806); random crop stays inside the image and is seedable
    Image.fromarray(rng.randint(0, 255, (70, 70), dtype=np.uint8)).save(str(tmp_path / "g.png"))
    g1 = load_image(str(tmp_path / "g.png"), 64, rng=np.random.RandomState(3))
    g2 = load_image(str(tmp_path / "g.png"), 64, rng=np.random.RandomState(3))
    assert g1.shape == (3, 64, 64) and torch.equal(g1, g2) and torch.equal(g1[0], g1[1])
    assert CUSTOM_TEMPLATES["caltech-101"].format("sea horse") == "a photo of a sea horse."


def test_auto_engine_batch_without_a_gpu():
    """--engine_batch 0: 8 for --tiny, 16 when there is no device to size the batch for (the HBM-sized choice needs a GPU)."""
    from distdiff_amd.generate_data import auto_engine_batch, parse_args
    a = parse_args(["--synthetic", "4", "--tiny"])
    assert auto_engine_batch(a, "cpu") == 8
    a = parse_args(["--synthetic", "4"])
    assert auto_engine_batch(a, "cpu") == 16
