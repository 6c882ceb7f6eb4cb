"""Generates tests/golden/tiny_fixture.pt — run in the BUILD container only (needs /root/reference).

Pins the oracle against the reference's OWN hot-path code: the FunctionDef nodes of `denoise_one_step`,
`tensor_clamp`, `linfball_proj`, `transform_guidance` and `direct_guidance` are read from
/root/reference/generate_data.py with `ast` (the module itself cannot be imported: torchvision / diffusers / timm
are not installed) and executed here, unchanged, against the oracle's duck-typed model objects (UNet / VAE /
scheduler / image processor / guide restated from the published diffusers / timm definitions). Their inputs and
outputs are the fixture. Nothing from the reference is copied into the repo: only tensors are saved.

    python tests/golden/make_fixtures.py
"""
import ast
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
REF = "/root/reference/generate_data.py"
WANTED = ["denoise_one_step", "tensor_clamp", "linfball_proj", "transform_guidance", "direct_guidance"]


def load_reference_functions(ns):
    src = open(REF).read()
    tree = ast.parse(src)
    found = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in WANTED:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, REF, "exec"), ns)
            found[node.name] = (node.lineno, node.end_lineno)
    missing = [w for w in WANTED if w not in found]
    assert not missing, missing
    return found


def main():
    from distdiff_amd.config import tiny_config
    from distdiff_amd.weights import synthetic_weights
    from oracle import sd_oracle as O
    from torch.autograd import Variable

    B = 2
    cfg = tiny_config(max_batch=B)
    w = synthetic_weights(cfg, seed=0, num_classes=5)
    unet, vae, guide, sched = O.build_models(cfg, w)
    n_steps = 10
    ts = sched.set_timesteps(n_steps)
    args = O.SamplerArgs(guidance_type="transform_guidance", num_inference_steps=n_steps, guidance_step=4, guidance_period=2, strength=0.5)

    # the reference reads a module-global `args` and calls .cuda(); give it both (CPU build: .cuda() is the identity)
    torch.Tensor.cuda = lambda self, *a, **k: self
    ns = {"torch": torch, "Variable": Variable, "args": args}
    lines = load_reference_functions(ns)
    # the reference hard-codes 224x224 for the guide input (generate_data.py:704); the tiny guide uses 56x56,
    # same 16/7 ratio: redirect F.interpolate's size only
    real_interp = torch.nn.functional.interpolate

    def interp(x, size=None, **kw):
        if size == (224, 224):
            size = (cfg.guide.input_size, cfg.guide.input_size)
        return real_interp(x, size=size, **kw)
    torch.nn.functional.interpolate = interp

    g = torch.Generator().manual_seed(1)
    L, D = cfg.latent_size, cfg.guide.feature_dim
    lat = torch.randn(B, 4, L, L, generator=g) * 0.9
    noise = torch.randn(B, 4, L, L, generator=g)
    pe = torch.randn(B, cfg.text_len, cfg.unet.cross_attention_dim, generator=g)
    ne = torch.randn(1, cfg.text_len, cfg.unet.cross_attention_dim, generator=g).expand(B, -1, -1).contiguous()
    Pc = F.normalize(torch.randn(5, D, generator=g), dim=-1)
    Pg = F.normalize(torch.randn(5, 3, D, generator=g), dim=-1)
    targets = torch.tensor([1, 3])
    embeds = torch.cat([ne, pe])
    si = O.start_index(args.strength, n_steps)
    z = sched.add_noise(lat, noise, ts[si])
    gts = O.guide_timesteps(ts, args.guidance_step, args.guidance_period)
    batch = {"targets": targets}
    iproc = O.ImageProcessorOracle()

    fx = {"n_steps": n_steps, "timesteps": ts.clone(), "start_index": si, "guide_timesteps": gts, "reference_lines": lines,
          "lat": lat, "noise": noise, "prompt_embeds": pe, "negative_embeds": ne, "Pc": Pc, "Pg": Pg, "targets": targets, "z": z,
          "args": {k: v for k, v in args.__dict__.items()}}

    # --- reference denoise_one_step (generate_data.py:109-121)
    with torch.no_grad():
        zp, x0 = ns["denoise_one_step"](z, sched, int(gts[0]), unet, embeds, None)
    fx["ref_denoise_z_prev"], fx["ref_denoise_x0"] = zp, x0

    # --- reference transform_guidance (generate_data.py:687-732); e, b come from the CPU global RNG (:692-695)
    seed = 1234
    torch.manual_seed(seed)
    e = torch.rand([B, 4, 1, 1])
    b = torch.zeros([B, 4, 1, 1]).data.normal_(0, 1)
    torch.manual_seed(seed)
    znew, score = ns["transform_guidance"](z.clone(), batch, gts, sched, unet, embeds, None, vae, guide, iproc, torch.float32, None, Pc, Pg)
    fx["e"], fx["b"], fx["ref_transform_z"], fx["ref_transform_score"] = e, b, znew.detach(), score.detach()

    # --- reference direct_guidance (generate_data.py:735-767)
    zn, x0d, sc = ns["direct_guidance"](z.clone(), batch, int(gts[0]), sched, unet, embeds, None, vae, guide, iproc, torch.float32, None, Pc, Pg)
    fx["ref_direct_z_next"], fx["ref_direct_x0"], fx["ref_direct_score"] = zn.detach(), x0d.detach(), sc.detach()

    # --- reference linfball_proj / tensor_clamp (generate_data.py:124-137)
    c = torch.randn(3, 5, generator=g)
    t = c + torch.randn(3, 5, generator=g)
    fx["clamp_center"], fx["clamp_t"] = c, t.clone()
    fx["ref_clamp_out"] = ns["linfball_proj"](c, 0.2, t.clone(), in_place=True)

    torch.nn.functional.interpolate = real_interp
    # --- oracle end-to-end vectors (restated loop, generate_data.py:1161-1228) for the three guidance modes
    for gt in ("transform_guidance", "direct_guidance", None):
        a2 = O.SamplerArgs(**{**args.__dict__, "guidance_type": gt})
        trace = []
        zf, img, s = O.expand_one(a2, cfg, (unet, vae, guide, sched), lat, noise, e, b, pe, ne, targets, Pc, Pg, trace=trace)
        key = gt or "none"
        fx["expand_%s_z" % key] = zf
        fx["expand_%s_img_u8" % key] = (img * 255 + 0.5).clamp(0, 255).to(torch.uint8)
        fx["expand_%s_score" % key] = None if s is None else s.detach()
        fx["expand_%s_trace" % key] = trace
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_fixture.pt")
    torch.save(fx, out)
    print("wrote", out, os.path.getsize(out), "bytes; reference functions at lines", lines)


if __name__ == "__main__":
    main()
