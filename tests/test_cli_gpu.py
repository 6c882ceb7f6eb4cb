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
