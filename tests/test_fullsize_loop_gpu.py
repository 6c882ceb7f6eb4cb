"""WHOLE-LOOP parity at the benchmarked shape and batch: `dd_expand` (add_noise -> every executed DDIM step incl. the guided ones ->
final decode, generate_data.py:1161-1234) at 512x512 with 32 images per engine batch -- exactly what bench.py times -- against the fp32
CPU oracle's whole loop on the same seeds (tests/golden/fullsize_loop_fixture.pt, made by make_fullsize_loop_fixture.py):

  configs[1]  the script of record (expand_diff.sh:3-15): strength 0.5 = 25 executed steps, transform guidance P = 2 at t = 381 + re-step,
              rho 10, constraint 0.2, C = 100, K = 3
  configs[3]  StanfordCars sizes (C = 196), direct guidance on each of the last 10 steps, shortened to 15 executed steps (strength 0.3)

Nothing is pinned to a common point here (the per-step tests evaluate the guide at the oracle's image): the engine draws the guide's
ReLU / max-pool masks at its OWN bf16 images, takes its own clamp decisions, and its error accumulates over all steps -- this is the
quantity a user of the PNGs sees.  Two independent rows ("a", "b") alternate over the 32 batch positions; every row is compared with
its own oracle run.  Stated per quantity (bounds = measured worst row on MI355X x ~1.5):

  final latents   relative L2 vs fp32
  final image     PSNR (peak 1.0), max abs difference, fraction of uint8 bytes that differ / differ by more than 2 levels
  guidance score  relative difference

Bounds are 1.25 x the worst row measured on MI355X with the round-5 binary (the measured values are in the comment of each bound).
tests/test_fullsize_loop_draws_gpu.py runs the same configs[1] loop on two MORE weight draws (fullsize_loop_w_fixture.pt): an
independent seed and a moderately non-flat one (attn1.to_q / to_k x 2 in every transformer block: scores x 4).
"""
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

B = 32
LOOSE = os.environ.get("DD_LOOP_MEASURE") == "1"      # print the measurements without asserting the bounds (to re-derive them)
# bounds = 1.25 x the worst row measured on MI355X with the round-5 binary (PSNR: 1.25 x the rms error = -1.94 dB); measured values:
#   c1  latents 0.0200, PSNR 43.29 dB, max abs 0.0656, u8 bytes differing 0.716 (by > 2 levels 0.1485), score 1.1e-4
#   c3  latents 0.0136, PSNR 46.28 dB, max abs 0.0312, u8 0.642 (0.0557), score < 1e-5
# (re-derived on the final round-5 binary -- the short-key cross-attention kernel rounds differently from the streaming one, equally
#  accurate per op (tools/xattn_acc.py), and these whole-loop measures move with every such change: c1 2.00 -> 1.77 %, s1 1.71 -> 1.51 %,
#  qk14 1.44 -> 1.76 %.  Measured: c1 0.0177 / 44.76 dB / 0.0380 / 0.683 (0.0982) / 6e-5; c3 0.0135 / 46.32 dB / 0.0296 / 0.643 (0.0540) / 0)
# Round 6: the polynomial GELU of the GEGLU epilogues (1e-5 accurate per op, tests/test_oracle.py::test_gelu_polynomial) is the third
# equally accurate rounding of the loop to be measured, and c1's worst row moved again: latents 0.0200 (round-4 binary) / 0.0177 (round 5)
# / 0.0222 (round 6), PSNR 43.29 / 44.76 / 42.53 dB, max abs 0.0656 / 0.0380 / 0.0831, u8 > 2 levels 0.1485 / 0.0982 / 0.1712 (row a of
# the same run: 0.0136, 45.06 dB).  A bound at 1.25 x ONE build's value is a bound on the lottery of the guide's masks, so c1's bounds
# are now 1.25 x the WORST of the three builds; c3 / s1 / qk14 / qk2 needed no change (round 6: c3 0.0133 / 46.41 dB / 0.0311 / 0.640
# (0.0522); s1 0.0151 / 45.32 / 0.0412; qk14 0.0192 / 43.94 / 0.0584; qk2 0.1734 / 25.63).
C1 = {"z_rel": 0.0278, "psnr": 40.6, "img_max": 0.104, "u8_diff": 0.80, "u8_gt2": 0.214, "score_rel": 1.5e-4}
C3 = {"z_rel": 0.0169, "psnr": 44.3, "img_max": 0.037, "u8_diff": 0.72, "u8_gt2": 0.0675, "score_rel": 5e-5}
# the other weight draws (tests/test_fullsize_loop_draws_gpu.py); measured:
#   s1    latents 0.0151, PSNR 45.25 dB, max abs 0.0361, u8 0.656 (0.0846), score 6e-5
#   qk14  (scores x 2 in every block, well conditioned: oracle eps moves 0.0023 under one bf16 rounding of its input)
#         latents 0.0176, PSNR 44.34 dB, max abs 0.0498, u8 0.683 (0.1120), score 1.3e-4; +0.001 per step like the flat draws
#   qk2   (scores x 4 in every block, ILL-conditioned: the oracle's own fp16 / bf16 executions of the tiny loop are 21 % / 27 % from its
#         fp32 run, tests/test_oracle.py) latents 0.178, PSNR 25.28 dB: error grows by a steady 0.8 % per step (1.1 % after the first step
#         against 0.73 % for ONE bf16 rounding of the oracle's input); bounded by that reduced-precision family, not a parity claim
CW = {"s1": {"z_rel": 0.0189, "psnr": 43.2, "img_max": 0.0451, "u8_diff": 0.74, "u8_gt2": 0.106, "score_rel": 1.5e-4},
      "qk14": {"z_rel": 0.022, "psnr": 42.3, "img_max": 0.0623, "u8_diff": 0.76, "u8_gt2": 0.14, "score_rel": 2.6e-4},
      "qk2": {"z_rel": 0.30, "psnr": 21.0, "img_max": 0.50, "u8_diff": 0.97, "u8_gt2": 0.92, "score_rel": 1.5e-3}}


# configs[3] IN FULL (tests/golden/fullsize_loop_c3full_fixture.pt: strength 1.0 = all 50 schedule steps executed, 40 plain + 10 direct-guided)
# measured (round 6): latents 0.0151, PSNR 45.80 dB, max abs 0.0366, u8 0.653 (0.0677), score 1e-5; the latent error saturates after 15 steps
# (0.0129 after step 5, 0.0149 after 15, 0.0151 from step 25 on): bounds = 1.25 x
C3FULL = {"z_rel": 0.0189, "psnr": 43.8, "img_max": 0.0458, "u8_diff": 0.75, "u8_gt2": 0.085, "score_rel": 5e-5}


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    assert torch.isfinite(a).all()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.fixture(scope="module")
def world(hip_lib):
    from make_fullsize_fixture import inputs
    from make_fullsize_loop_fixture import loop_inputs
    from distdiff_amd.config import sd15_config
    from distdiff_amd.engine import Engine
    from distdiff_amd.scheduler import DDIMSchedule
    from distdiff_amd.weights import synthetic_weights
    free, total = torch.cuda.mem_get_info()
    need = 5.8e9 * B + 8e9            # 186.7 GB of workspace at B = 32
    if free < need:
        # the benchmarked batch must fit an empty MI355X: a smaller device is a skip, a full-size device that cannot take it is a failure
        if total < 280e9:
            pytest.skip("engine batch %d needs ~%.0f GB of HBM, device has %.0f GB" % (B, need / 1e9, total / 1e9))
        pytest.fail("engine batch %d needs ~%.0f GB of HBM, only %.0f of %.0f GB free" % (B, need / 1e9, free / 1e9, total / 1e9))
    fx = torch.load(os.path.join(HERE, "golden", "fullsize_loop_fixture.pt"), weights_only=False)
    cfg = sd15_config(latent_size=64, max_batch=B)
    w = synthetic_weights(cfg, seed=0, num_classes=100)
    chk = float(sum(v.double().sum() for v in w["unet"].values()))
    assert abs(chk - fx["weights_checksum"]) <= 1e-6 * abs(fx["weights_checksum"]), "synthetic weights differ from the fixture's"
    eng = Engine(cfg, w, enable_grad=True, max_guidance_period=2)
    sched = DDIMSchedule(cfg.scheduler)
    ts = sched.set_timesteps(50)
    rows = ["a" if i % 2 == 0 else "b" for i in range(B)]
    d = {t: loop_inputs(cfg, t) for t in ("a", "b")}
    inp = {k: torch.cat([d[r][k] for r in rows]) for k in ("latents", "noise", "e", "b", "neg", "pos", "t100", "t196")}
    eng.set_prompt(torch.cat([inp["neg"], inp["pos"]]).cuda())
    eng.set_sample_weights([1.0] * B)          # train_batch_size = 1: every row is its own reference batch (generate_data.py:709)
    yield {"eng": eng, "cfg": cfg, "sched": sched, "ts": ts, "rows": rows, "inp": inp, "fx": fx, "proto": inputs(cfg)}
    eng.close()


def _compare(tag, w, z, img, scores, bounds):
    """Per-row comparison of the loop's outputs with the oracle's; returns the worst-row numbers and asserts the bounds."""
    eng, rows, fx = w["eng"], w["rows"], w["fx"]
    u8 = eng.image_to_u8(img).cpu()
    img, z = img.cpu(), z.cpu()
    worst = {"z_rel": 0.0, "psnr": 1e9, "img_max": 0.0, "u8_diff": 0.0, "u8_gt2": 0.0, "score_rel": 0.0}
    per_row = []
    for i, r in enumerate(rows):
        f = fx["%s_%s" % (tag, r)]
        ref_img = f["image16"].float()
        zr = rel(z[i:i + 1], f["z_final"])
        mse = float(((img[i:i + 1] - ref_img) ** 2).mean())
        psnr = 10.0 * math.log10(1.0 / max(mse, 1e-20))
        imax = float((img[i:i + 1] - ref_img).abs().max())
        du8 = (u8[i:i + 1].int() - f["image_u8"].int()).abs()
        udiff, ugt2 = float((du8 > 0).float().mean()), float((du8 > 2).float().mean())
        s_ref = float(f["scores"][-1])
        srel = abs(float(scores[i]) - s_ref) / abs(s_ref)
        per_row.append((r, zr, psnr, imax, udiff, ugt2, srel))
        worst = {"z_rel": max(worst["z_rel"], zr), "psnr": min(worst["psnr"], psnr), "img_max": max(worst["img_max"], imax),
                 "u8_diff": max(worst["u8_diff"], udiff), "u8_gt2": max(worst["u8_gt2"], ugt2), "score_rel": max(worst["score_rel"], srel)}
    print("%s whole loop, B=%d, worst row: final latents rel-L2 %.4f | image PSNR %.2f dB, max abs %.4f, u8 bytes differing %.3f (by > 2 "
          "levels: %.4f) | score rel %.5f" % (tag, len(rows), worst["z_rel"], worst["psnr"], worst["img_max"], worst["u8_diff"],
                                            worst["u8_gt2"], worst["score_rel"]))
    for r in sorted(set(rows)):
        sel = [x for x in per_row if x[0] == r]
        print("   row %s: latents %.4f-%.4f  PSNR %.2f-%.2f" % (r, min(x[1] for x in sel), max(x[1] for x in sel),
                                                               min(x[2] for x in sel), max(x[2] for x in sel)))
    if not LOOSE:
        assert worst["z_rel"] <= bounds["z_rel"] and worst["psnr"] >= bounds["psnr"] and worst["img_max"] <= bounds["img_max"], (worst, bounds)
        assert worst["u8_diff"] <= bounds["u8_diff"] and worst["u8_gt2"] <= bounds["u8_gt2"] and worst["score_rel"] <= bounds["score_rel"], (worst, bounds)
    return worst


def test_config1_whole_loop_script_of_record(world):
    """expand_diff.sh:3-15 through dd_expand at B = 32 vs the oracle's expand loop, every row."""
    from distdiff_amd.scheduler import guide_window, start_index
    w = world
    eng, sched, ts, inp, fx = w["eng"], w["sched"], w["ts"], w["inp"], w["fx"]
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_scale=7.5, gs=1.0, ls=1.0, rho=10.0,
                     constraint_value=0.2, guidance_period=2)
    eng.set_prototypes(w["proto"]["Pc100"], w["proto"]["Pg100"])
    si = start_index(0.5, 50)
    first, cnt = guide_window(50, 20, 2)
    assert si == fx["c1_a"]["start_index"] and [ts[first], ts[first + 1]] == fx["c1_a"]["guide_timesteps"]
    z, img, _ = eng.expand(inp["latents"], inp["noise"], inp["e"], inp["b"], inp["t100"], si, "transform_guidance", first, cnt)
    scores = eng.image_scores().cpu()
    _compare("c1", w, z, img, scores, C1)
    # where the difference comes from: the same loop step by step, each row's latents against the oracle's trajectory
    zc = eng.add_noise(inp["latents"], inp["noise"], si)
    errs = []
    for k, i in enumerate(range(si, 50)):
        if i == first:
            zc, _, _ = eng.transform_guidance(zc, inp["t100"], inp["e"], inp["b"], first, cnt)
            g = max(rel(zc[j:j + 1], fx["c1_" + r]["z_guided"]) for j, r in enumerate(w["rows"]))
            errs.append("guided %.4f" % g)
        zc, _ = eng.denoise_step(zc, i)
        errs.append("%.4f" % max(rel(zc[j:j + 1], fx["c1_" + r]["traj"][k + 1:k + 2]) for j, r in enumerate(w["rows"])))
    print("c1 latents rel-L2 vs the oracle's trajectory, worst row, step by step: " + " ".join(errs))
    assert torch.equal(zc, z), "dd_expand and the step-by-step ABI calls differ"


def test_config3_whole_loop_direct_guidance_last_10_steps(world):
    """generate_data.py:1210-1216 (direct guidance on each of the last 10 steps, StanfordCars sizes) through dd_expand at B = 32."""
    from distdiff_amd.scheduler import guide_window, start_index
    w = world
    eng, sched, ts, inp, fx = w["eng"], w["sched"], w["ts"], w["inp"], w["fx"]
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_scale=7.5, gs=1.0, ls=1.0, rho=10.0,
                     constraint_value=0.2, guidance_period=10)
    eng.set_prototypes(w["proto"]["Pc196"], w["proto"]["Pg196"])
    si = start_index(fx["c3_strength"], 50)
    first, cnt = guide_window(50, 10, 10)
    assert si == fx["c3_a"]["start_index"] and [ts[first + k] for k in range(cnt)] == fx["c3_a"]["guide_timesteps"]
    z, img, _ = eng.expand(inp["latents"], inp["noise"], None, None, inp["t196"], si, "direct_guidance", first, cnt)
    scores = eng.image_scores().cpu()
    _compare("c3", w, z, img, scores, C3)



def test_config3_whole_50_step_schedule(world):
    """BASELINE configs[3] as stated -- "50 DDIM steps ... guidance on the last 10 steps" -- with nothing shortened: strength 1.0, i.e. pure
    noise at t = 981, 40 plain steps, direct guidance (generate_data.py:735-767, :1210-1216) on each of the last 10, final decode, uint8,
    through dd_expand at B = 32 against the fp32 oracle's own 50-step loop (its guide masks at its own images).  The shortened form
    above (15 executed steps) shares the last 10 steps with this one; this is the test of the first 40."""
    from distdiff_amd.scheduler import guide_window, start_index
    path = os.path.join(HERE, "golden", "fullsize_loop_c3full_fixture.pt")
    if not os.path.exists(path):
        pytest.skip("tests/golden/fullsize_loop_c3full_fixture.pt not generated (make_fullsize_loop_c3full_fixture.py)")
    fx = torch.load(path, weights_only=False)
    w = dict(world)
    w["fx"] = fx
    eng, sched, ts, inp = w["eng"], w["sched"], w["ts"], w["inp"]
    eng.set_schedule(ts, sched.alphas_cumprod, sched.final_alpha_cumprod, guidance_scale=7.5, gs=1.0, ls=1.0, rho=10.0,
                     constraint_value=0.2, guidance_period=10)
    eng.set_prototypes(w["proto"]["Pc196"], w["proto"]["Pg196"])
    si = start_index(1.0, 50)
    first, cnt = guide_window(50, 10, 10)
    assert si == 0 == fx["c3full_a"]["start_index"] and [ts[first + k] for k in range(cnt)] == fx["c3full_a"]["guide_timesteps"]
    z, img, _ = eng.expand(inp["latents"], inp["noise"], None, None, inp["t196"], si, "direct_guidance", first, cnt)
    scores = eng.image_scores().cpu()
    global LOOSE
    loose, LOOSE = LOOSE, LOOSE or C3FULL is None
    try:
        worst = _compare("c3full", w, z, img, scores, C3FULL)
    finally:
        LOOSE = loose
    # growth along the schedule: every fifth state of the oracle's trajectory against the step-by-step ABI calls
    zc = eng.add_noise(inp["latents"], inp["noise"], si)
    errs = []
    for i in range(50):
        if first <= i < first + cnt:
            zc, _, _, _ = eng.direct_guidance(zc, inp["t196"], i)
        else:
            zc, _ = eng.denoise_step(zc, i)
        if (i + 1) % 5 == 0:
            errs.append("%d: %.4f" % (i + 1, max(rel(zc[j:j + 1], fx["c3full_" + r]["traj"][(i + 1) // 5:(i + 1) // 5 + 1]) for j, r in enumerate(w["rows"]))))
    print("c3full latents rel-L2 vs the oracle's trajectory, worst row, after step " + " ".join(errs))
    assert torch.equal(zc, z), "dd_expand and the step-by-step ABI calls differ"
    assert C3FULL is not None or LOOSE or worst["z_rel"] < 0.05, worst
