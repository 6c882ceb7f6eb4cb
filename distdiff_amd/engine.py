"""Python host side of the expansion engine: a thin ctypes layer over include/distdiff_hip.h.

PyTorch supplies device memory and the HIP stream; all arithmetic happens in libdistdiff_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import EngineConfig

DD_MAX_LEVELS = 8
vp = C.c_void_p
_IA = C.c_int * DD_MAX_LEVELS


class DDConfig(C.Structure):
    _fields_ = [
        ("unet_in_channels", C.c_int), ("unet_out_channels", C.c_int), ("unet_levels", C.c_int),
        ("unet_block_out_channels", _IA), ("unet_layers_per_block", C.c_int),
        ("unet_down_attn", _IA), ("unet_up_attn", _IA),
        ("unet_num_heads", C.c_int), ("unet_cross_dim", C.c_int), ("unet_groups", C.c_int),
        ("unet_eps", C.c_float), ("unet_freq_shift", C.c_float), ("unet_flip_sin_to_cos", C.c_int),
        ("vae_latent_channels", C.c_int), ("vae_out_channels", C.c_int), ("vae_levels", C.c_int),
        ("vae_block_out_channels", _IA), ("vae_layers_per_block", C.c_int), ("vae_groups", C.c_int),
        ("vae_eps", C.c_float), ("vae_scaling_factor", C.c_float),
        ("guide_stem", C.c_int), ("guide_stages", C.c_int), ("guide_planes", _IA), ("guide_blocks", _IA),
        ("guide_expansion", C.c_int), ("guide_input_size", C.c_int), ("guide_bn_eps", C.c_float),
        ("latent_size", C.c_int), ("text_len", C.c_int), ("max_batch", C.c_int),
        ("enable_grad", C.c_int), ("max_guidance_period", C.c_int),
        ("text_heads", C.c_int), ("text_act", C.c_int), ("text_eps", C.c_float),
        ("guide_kind", C.c_int), ("guide_strides", _IA), ("guide_vit_heads", C.c_int), ("guide_vit_patch", C.c_int), ("guide_vit_act", C.c_int),
        ("guide_feature_dim", C.c_int),
        ("unet_transformer_depth", _IA), ("unet_level_heads", _IA), ("unet_add_time_dim", C.c_int), ("unet_add_text_dim", C.c_int),
        ("text_hidden_layer", C.c_int), ("text2_heads", C.c_int), ("text2_act", C.c_int), ("text2_eps", C.c_float),
        ("unet_attn_fp8", C.c_int), ("abi_version", C.c_int),
    ]


class DDSamplerParams(C.Structure):
    _fields_ = [("guidance_scale", C.c_float), ("gs", C.c_float), ("ls", C.c_float), ("rho", C.c_float),
                ("constraint_value", C.c_float), ("use_global", C.c_int), ("use_local", C.c_int),
                ("guidance_period", C.c_int)]


class DDExpandArgs(C.Structure):
    _fields_ = [("image_latents", vp), ("noise", vp), ("e", vp), ("b", vp), ("targets", vp), ("B", C.c_int),
                ("start_index", C.c_int), ("guidance_type", C.c_int), ("guide_first", C.c_int), ("guide_count", C.c_int),
                ("z_out", vp), ("image_out", vp), ("score_out", vp)]


def _declare(l):
    i, f = C.c_int, C.c_float
    l.dd_abi_version.argtypes = []
    l.dd_create.argtypes = [C.POINTER(DDConfig), C.POINTER(vp)]
    l.dd_destroy.argtypes = [vp]
    l.dd_destroy.restype = None
    l.dd_last_error.argtypes = [vp]
    l.dd_last_error.restype = C.c_char_p
    l.dd_load_tensor.argtypes = [vp, C.c_char_p, C.c_char_p, vp, i, C.POINTER(C.c_int64)]
    l.dd_finalize_weights.argtypes = [vp]
    l.dd_declare_tensor.argtypes = [vp, C.c_char_p, C.c_char_p, i, C.POINTER(C.c_int64)]
    l.dd_packed_bytes.argtypes = [vp]
    l.dd_packed_bytes.restype = C.c_size_t
    l.dd_export_packed.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp]
    l.dd_import_packed.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp]
    l.dd_set_schedule.argtypes = [vp, vp, i, vp, i, f, C.POINTER(DDSamplerParams)]
    l.dd_set_prototypes.argtypes = [vp, vp, vp, i, i, i]
    l.dd_set_prompt.argtypes = [vp, vp, i, vp]
    l.dd_add_noise.argtypes = [vp, vp, vp, vp, i, i, vp]
    l.dd_set_added_cond.argtypes = [vp, vp, vp, i, vp]
    l.dd_denoise_step.argtypes = [vp, vp, i, vp, vp, i, vp]
    l.dd_transform_guidance.argtypes = [vp, vp, vp, vp, vp, i, i, vp, vp, vp, i, vp]
    l.dd_direct_guidance.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, i, vp]
    l.dd_decode.argtypes = [vp, vp, vp, i, i, vp]
    l.dd_expand.argtypes = [vp, C.POINTER(DDExpandArgs), vp]
    l.dd_guide_encode.argtypes = [vp, vp, vp, i, vp]
    l.dd_guide_encode_pooled.argtypes = [vp, vp, vp, i, i, vp]
    l.dd_vae_encode.argtypes = [vp, vp, vp, vp, vp, i, vp]
    l.dd_text_encode.argtypes = [vp, vp, vp, i, vp]
    l.dd_text_encode_tower.argtypes = [vp, i, vp, vp, vp, i, vp]
    l.dd_image_to_u8.argtypes = [vp, vp, vp, i, vp]
    l.dd_set_sample_weights.argtypes = [vp, vp, i]
    l.dd_get_image_scores.argtypes = [vp, vp, i, vp]
    l.dd_unet_forward.argtypes = [vp, vp, i, vp, i, vp]
    l.dd_unet_vjp.argtypes = [vp, vp, i, vp, vp, i, vp]
    l.dd_decode_vjp.argtypes = [vp, vp, vp, vp, i, vp]
    l.dd_guide_vjp.argtypes = [vp, vp, vp, vp, i, vp]
    l.dd_profile_enable.argtypes = [vp, i]
    l.dd_profile_read.argtypes = [vp, vp]
    l.dd_debug_tensor.argtypes = [vp, i, i, i, vp, vp]
    l.dd_debug_num_tensors.argtypes = [vp, i]
    l.dd_debug_set_image.argtypes = [vp, vp]
    l.dd_debug_set_images.argtypes = [vp, vp, i]
    l.dd_workspace_bytes.argtypes = [vp]
    l.dd_workspace_bytes.restype = C.c_size_t
    l.dd_flops_last.argtypes = [vp]
    l.dd_flops_last.restype = C.c_double


DD_ABI_VERSION = _lib.ABI_VERSION      # include/distdiff_hip.h


def _to_c_config(cfg: EngineConfig, enable_grad, max_guidance_period, attn_fp8=False):
    c = DDConfig()
    c.abi_version = DD_ABI_VERSION
    c.unet_attn_fp8 = int(bool(attn_fp8))
    u, v, g = cfg.unet, cfg.vae, cfg.guide

    def arr(xs):
        a = _IA()
        for k, x in enumerate(xs):
            a[k] = int(x)
        return a

    c.unet_in_channels, c.unet_out_channels, c.unet_levels = u.in_channels, u.out_channels, len(u.block_out_channels)
    c.unet_block_out_channels = arr(u.block_out_channels)
    c.unet_layers_per_block = u.layers_per_block
    c.unet_down_attn, c.unet_up_attn = arr(u.down_attn), arr(u.up_attn)
    c.unet_num_heads, c.unet_cross_dim, c.unet_groups = u.num_heads, u.cross_attention_dim, u.norm_num_groups
    c.unet_eps, c.unet_freq_shift, c.unet_flip_sin_to_cos = u.norm_eps, u.freq_shift, int(u.flip_sin_to_cos)
    c.vae_latent_channels, c.vae_out_channels, c.vae_levels = v.latent_channels, v.out_channels, len(v.block_out_channels)
    c.vae_block_out_channels = arr(v.block_out_channels)
    c.vae_layers_per_block, c.vae_groups, c.vae_eps, c.vae_scaling_factor = v.layers_per_block, v.norm_num_groups, v.norm_eps, v.scaling_factor
    c.guide_stem, c.guide_stages = g.stem_channels, len(g.planes)
    c.guide_planes, c.guide_blocks = arr(g.planes), arr(g.blocks)
    c.guide_expansion, c.guide_input_size, c.guide_bn_eps = g.expansion, g.input_size, g.bn_eps
    c.latent_size, c.text_len, c.max_batch = cfg.latent_size, cfg.text_len, cfg.max_batch
    c.enable_grad, c.max_guidance_period = int(enable_grad), int(max_guidance_period)
    t = cfg.text
    c.text_heads, c.text_act, c.text_eps = t.num_attention_heads, {"quick_gelu": 0, "gelu": 1}[t.hidden_act], t.layer_norm_eps
    c.guide_kind = {"resnet": 0, "vit": 1, "mbv2": 2}[g.kind]
    if g.kind == "mbv2":
        c.guide_stem, c.guide_stages = g.mb_stem, len(g.mb_channels)
        c.guide_planes, c.guide_blocks, c.guide_strides = arr(g.mb_channels), arr(g.mb_repeats), arr(g.mb_strides)
        c.guide_expansion = g.mb_expand
    c.guide_vit_heads, c.guide_vit_patch, c.guide_vit_act = g.vit_heads, g.vit_patch, {"quick_gelu": 0, "gelu": 1}[g.vit_act]
    c.guide_feature_dim = g.feature_dim
    if u.transformer_depth:
        c.unet_transformer_depth = arr(u.transformer_depth)
    if u.level_heads:
        c.unet_level_heads = arr(u.level_heads)
    c.unet_add_time_dim, c.unet_add_text_dim = u.add_time_dim, u.add_text_dim
    c.text_hidden_layer = cfg.text_hidden_layer
    if cfg.text2 is not None:
        t2 = cfg.text2
        c.text2_heads, c.text2_act, c.text2_eps = t2.num_attention_heads, {"quick_gelu": 0, "gelu": 1}[t2.hidden_act], t2.layer_norm_eps
    return c


def _p(t):
    return vp(t.data_ptr()) if t is not None else vp(0)


def _stream():
    return vp(torch.cuda.current_stream().cuda_stream)


# HBM budget of the 512x512 SD-1.x workload per image of the static batch (measured: 186.7 GB of engine workspace at 32 images with
# two chained guided steps = 5.83 GB per image; 3.0 GB without guidance) + weights, guide / VAE scratch and allocator slack.  The one
# place the CLI's and bench.py's automatic batch come from.
HBM_BYTES_PER_IMAGE_GUIDED = 5.85e9
HBM_BYTES_PER_IMAGE_PLAIN = 3.0e9
HBM_BYTES_FIXED = 12e9


def batch_for_free_hbm(free_bytes, guided=True):
    """Largest static engine batch of 32 / 16 / 8 whose workspace fits `free_bytes` of HBM."""
    per = HBM_BYTES_PER_IMAGE_GUIDED if guided else HBM_BYTES_PER_IMAGE_PLAIN
    for B in (32, 16):
        if free_bytes >= B * per + HBM_BYTES_FIXED:
            return B
    return 8


class Engine:
    """One engine per device. Mirrors the objects the reference builds at generate_data.py:863-922, 1100-1125."""

    def __init__(self, cfg: EngineConfig, weights, enable_grad=True, max_guidance_period=2, device="cuda:0", layout=None, attn_fp8=False):
        """attn_fp8: BASELINE configs[4]'s fp8 MFMA attention (P.V of the UNet's d = 64 heads in e4m3; dd_config.unet_attn_fp8).
        weights: {"unet" | "vae" | "guide" | "text" | "text2": state dict} -- packed on this rank; or None with `layout` = the
        `weight_layout()` of the rank that has them: the engine is then built from the tensor shapes alone and its packed weight
        buffers are filled by `import_packed` (launcher.broadcast_packed_weights: one RCCL broadcast of the packed device buffers
        instead of every process loading its own copy, scripts/exps/expand_diff.sh:19-24)."""
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.L = _lib.lib()
        self._h = vp()
        cc = _to_c_config(cfg, enable_grad, max_guidance_period, attn_fp8)
        self._chk(self.L.dd_create(C.byref(cc), C.byref(self._h)), "dd_create")
        if weights is not None:
            self.layout = []
            for model in ("unet", "vae", "guide", "text", "text2"):
                for key, t in weights.get(model, {}).items():
                    if key.startswith(("fc.", "classifier.")) or key.endswith("num_batches_tracked"):
                        continue
                    a = t.detach().float().contiguous().cpu()
                    shape = (C.c_int64 * a.dim())(*a.shape)
                    self._chk(self.L.dd_load_tensor(self._h, model.encode(), key.encode(), vp(a.data_ptr()), a.dim(), shape),
                              "dd_load_tensor " + key)
                    self.layout.append((model, key, tuple(a.shape)))
        else:
            assert layout is not None, "Engine needs weights or the layout of the rank that holds them"
            self.layout = list(layout)
            for model, key, shp in self.layout:
                shape = (C.c_int64 * len(shp))(*shp)
                self._chk(self.L.dd_declare_tensor(self._h, model.encode(), key.encode(), len(shp), shape), "dd_declare_tensor " + key)
        self._chk(self.L.dd_finalize_weights(self._h), "dd_finalize_weights")
        self.n_steps = 0
        self.B = cfg.max_batch
        self._keep = []

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.L.dd_last_error(self._h).decode() if self._h else ""
            raise RuntimeError("%s failed (%d): %s" % (what, rc, msg))

    def close(self):
        if self._h:
            torch.cuda.synchronize()
            self.L.dd_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- packed weights (multi-GPU start-up) ---------------------------------------------------
    def weight_layout(self):
        return list(self.layout)

    def packed_bytes(self):
        return int(self.L.dd_packed_bytes(self._h))

    def export_packed(self, buf, offset):
        """Copies bytes [offset, offset + buf.numel()) of the packed weights into the device uint8 tensor `buf`."""
        self._chk(self.L.dd_export_packed(self._h, _p(buf), offset, buf.numel(), _stream()), "dd_export_packed")

    def import_packed(self, buf, offset):
        self._chk(self.L.dd_import_packed(self._h, _p(buf), offset, buf.numel(), _stream()), "dd_import_packed")

    # ---- setup -------------------------------------------------------------------------------
    def set_schedule(self, timesteps, alphas_cumprod, final_alpha_cumprod, guidance_scale=7.5, gs=1.0, ls=1.0, rho=10.0,
                     constraint_value=0.2, use_global=True, use_local=True, guidance_period=2):
        ts = np.ascontiguousarray(np.asarray(timesteps, dtype=np.int32))
        ac = np.ascontiguousarray(np.asarray(alphas_cumprod, dtype=np.float32))
        sp = DDSamplerParams(guidance_scale, gs, ls, rho, constraint_value, int(use_global), int(use_local), int(guidance_period))
        self._chk(self.L.dd_set_schedule(self._h, ts.ctypes.data_as(vp), len(ts), ac.ctypes.data_as(vp), len(ac),
                                         float(final_alpha_cumprod), C.byref(sp)), "dd_set_schedule")
        self.n_steps = len(ts)
        self.timesteps = [int(t) for t in ts]

    def set_prototypes(self, Pc, Pg):
        pc = Pc.detach().float().contiguous().cpu() if Pc is not None else None
        pg = Pg.detach().float().contiguous().cpu() if Pg is not None else None
        ref = pc if pc is not None else pg
        Cn, D = ref.shape[0], ref.shape[-1]
        K = pg.shape[1] if pg is not None else 1
        self._chk(self.L.dd_set_prototypes(self._h, _p(pc), _p(pg), Cn, K, D), "dd_set_prototypes")

    def set_prompt(self, embeds_2b):
        """embeds_2b: device fp32 [2B, text_len, cross_dim], negative (uncond) half first (generate_data.py:1184)."""
        e = embeds_2b.to(self.device, torch.float32).contiguous()
        assert e.shape[0] == 2 * self.B
        self._chk(self.L.dd_set_prompt(self._h, _p(e), self.B, _stream()), "dd_set_prompt")
        self._keep = [e]

    def set_sample_weights(self, w):
        """w[i] = 1 / |reference batch of image i| (the `.mean()` of generate_data.py:709-719, 750-760 runs over train_batch_size
        images), 0 for padding rows; None restores the default 1/B."""
        if w is None:
            self._chk(self.L.dd_set_sample_weights(self._h, vp(0), self.B), "dd_set_sample_weights")
            self._sample_w = None
            return
        a = torch.as_tensor(w, dtype=torch.float32).contiguous().cpu()
        assert a.numel() == self.B
        # the call synchronises the device (the weights of a guidance call in flight must not change under it): the CLI's full batches
        # all carry the same weights, so an unchanged vector is not sent again and the host stays ahead of the GPU
        if getattr(self, "_sample_w", None) is not None and torch.equal(self._sample_w, a):
            return
        self._sample_w = a.clone()
        self._chk(self.L.dd_set_sample_weights(self._h, vp(a.data_ptr()), self.B), "dd_set_sample_weights")

    def image_scores(self):
        """Per-image energies of the most recent guidance call (device [B])."""
        out = torch.empty(self.B, device=self.device, dtype=torch.float32)
        self._chk(self.L.dd_get_image_scores(self._h, _p(out), self.B, _stream()), "dd_get_image_scores")
        return out

    def set_added_cond(self, text_embeds, time_ids):
        """SDXL `added_cond_kwargs`: text_embeds [2B, add_text_dim], time_ids [2B, 6] (negative half first); after set_schedule."""
        te, ti = self._f(text_embeds), self._f(time_ids)
        assert te.shape[0] == 2 * self.B and tuple(ti.shape) == (2 * self.B, 6)
        self._chk(self.L.dd_set_added_cond(self._h, _p(te), _p(ti), self.B, _stream()), "dd_set_added_cond")

    # ---- stage before the loop (dataloader.py:633-661, 750-811) ------------------
    def vae_encode(self, images, noise=None, return_moments=False):
        """`vae.encode(x).latent_dist.sample() * vae.config.scaling_factor` (dataloader.py:808-809).
        images [B,3,8L,8L] in [-1,1]; noise [B,4,L,L] ~ N(0,1), or None for the distribution's mode."""
        x = self._f(images)
        B, L = x.shape[0], self.cfg.latent_size
        n = self._f(noise) if noise is not None else None
        lat = torch.empty((B, self.cfg.vae.latent_channels, L, L), device=self.device, dtype=torch.float32)
        mom = torch.empty((B, 2 * self.cfg.vae.latent_channels, L, L), device=self.device, dtype=torch.float32) if return_moments else None
        self._chk(self.L.dd_vae_encode(self._h, _p(x), _p(n), _p(lat), _p(mom), B, _stream()), "dd_vae_encode")
        return (lat, mom) if return_moments else lat

    def text_encode(self, input_ids):
        """`text_encoder(input_ids)[0]` (dataloader.py:633-646): ids [n, text_len] -> [n, text_len, cross_dim] fp32, n <= 2B."""
        ids = input_ids.to(self.device, torch.int32).contiguous()
        n = ids.shape[0]
        out = torch.empty((n, self.cfg.text_len, self.cfg.unet.cross_attention_dim), device=self.device, dtype=torch.float32)
        self._chk(self.L.dd_text_encode(self._h, _p(ids), _p(out), n, _stream()), "dd_text_encode")
        return out

    def text_encode_tower(self, which, input_ids, pooled=False):
        """One tower of a two-tower (SDXL) model: ids [n, text_len] of THAT tower's tokenizer -> hidden states [n, text_len, width] at
        cfg.text_hidden_layer and, for the second tower with pooled=True, text_embeds [n, projection_dim]
        (CLIPTextModelWithProjection; diffusers StableDiffusionXLPipeline.encode_prompt)."""
        t = self.cfg.text2 if which else self.cfg.text
        ids = input_ids.to(self.device, torch.int32).contiguous()
        n = ids.shape[0]
        hid = torch.empty((n, self.cfg.text_len, t.hidden_size), device=self.device, dtype=torch.float32)
        pool = torch.empty((n, t.projection_dim), device=self.device, dtype=torch.float32) if pooled else None
        self._chk(self.L.dd_text_encode_tower(self._h, int(which), _p(ids), _p(hid), _p(pool), n, _stream()), "dd_text_encode_tower")
        return (hid, pool) if pooled else hid

    # ---- hot path ------------------------------------------------------------------------------
    def _f(self, t):
        t = t.to(self.device, torch.float32).contiguous()
        return t

    def add_noise(self, x, noise, step_index):
        x, noise = self._f(x), self._f(noise)
        out = torch.empty_like(x)
        self._chk(self.L.dd_add_noise(self._h, _p(x), _p(noise), _p(out), x.shape[0], step_index, _stream()), "dd_add_noise")
        return out

    def unet_forward(self, z, step_index):
        z = self._f(z)
        out = torch.empty((2 * z.shape[0],) + tuple(z.shape[1:]), device=self.device, dtype=torch.float32)
        self._chk(self.L.dd_unet_forward(self._h, _p(z), step_index, _p(out), z.shape[0], _stream()), "dd_unet_forward")
        return out

    def denoise_step(self, z, step_index):
        z = self._f(z)
        zp, x0 = torch.empty_like(z), torch.empty_like(z)
        self._chk(self.L.dd_denoise_step(self._h, _p(z), step_index, _p(zp), _p(x0), z.shape[0], _stream()), "dd_denoise_step")
        return zp, x0

    def transform_guidance(self, z, targets, e, b, first_step_index, P):
        z, e, b = self._f(z), self._f(e).reshape(-1), self._f(b).reshape(-1)
        tg = targets.to(self.device, torch.int32).contiguous()
        out = torch.empty_like(z)
        score = torch.zeros(1, device=self.device)
        gz0 = torch.empty_like(z)
        self._chk(self.L.dd_transform_guidance(self._h, _p(z), _p(tg), _p(e), _p(b), first_step_index, P, _p(out), _p(score),
                                               _p(gz0), z.shape[0], _stream()), "dd_transform_guidance")
        return out, score, gz0

    def direct_guidance(self, z, targets, step_index):
        z = self._f(z)
        tg = targets.to(self.device, torch.int32).contiguous()
        zn, x0, gz = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        score = torch.zeros(1, device=self.device)
        self._chk(self.L.dd_direct_guidance(self._h, _p(z), _p(tg), step_index, _p(zn), _p(x0), _p(score), _p(gz), z.shape[0],
                                            _stream()), "dd_direct_guidance")
        return zn, x0, score, gz

    def decode(self, z, denormalize=True):
        z = self._f(z)
        L8 = self.cfg.latent_size * 8
        img = torch.empty((z.shape[0], self.cfg.vae.out_channels, L8, L8), device=self.device, dtype=torch.float32)
        self._chk(self.L.dd_decode(self._h, _p(z), _p(img), int(denormalize), z.shape[0], _stream()), "dd_decode")
        return img

    def image_to_u8(self, img):
        """[B,3,H,W] fp32 in [0,1] (device) -> uint8 [B,H,W,3] (device), save_image quantisation."""
        img = self._f(img)
        out = torch.empty((img.shape[0], img.shape[2], img.shape[3], 3), device=self.device, dtype=torch.uint8)
        self._chk(self.L.dd_image_to_u8(self._h, _p(img), _p(out), img.shape[0], _stream()), "dd_image_to_u8")
        return out

    def guide_encode(self, images, pooling="avg"):
        """image_encoder.encode_image(x, pooling) (model_utils.py:29-41)."""
        x = self._f(images)
        f = torch.empty((x.shape[0], self.cfg.guide.feature_dim), device=self.device, dtype=torch.float32)
        self._chk(self.L.dd_guide_encode_pooled(self._h, _p(x), _p(f), x.shape[0], int(pooling == "max"), _stream()), "dd_guide_encode")
        return f

    def expand(self, image_latents, noise, e, b, targets, start_index, guidance_type, guide_first, guide_count, want_image=True):
        """guidance_type: None | 'transform_guidance' | 'direct_guidance' (generate_data.py:1203-1218)."""
        lat, nz = self._f(image_latents), self._f(noise)
        B = lat.shape[0]
        e = self._f(e).reshape(-1) if e is not None else torch.zeros(B * 4, device=self.device)
        b = self._f(b).reshape(-1) if b is not None else torch.zeros(B * 4, device=self.device)
        tg = targets.to(self.device, torch.int32).contiguous()
        a = DDExpandArgs()
        z_out = torch.empty_like(lat)
        L8 = self.cfg.latent_size * 8
        img = torch.empty((B, 3, L8, L8), device=self.device) if want_image else None
        score = torch.zeros(1, device=self.device)
        a.image_latents, a.noise, a.e, a.b, a.targets, a.B = _p(lat), _p(nz), _p(e), _p(b), _p(tg), B
        a.start_index = start_index
        a.guidance_type = {None: 0, "": 0, "transform_guidance": 1, "direct_guidance": 2}[guidance_type]
        a.guide_first, a.guide_count = guide_first, guide_count
        a.z_out, a.image_out, a.score_out = _p(z_out), _p(img), _p(score)
        self._chk(self.L.dd_expand(self._h, C.byref(a), _stream()), "dd_expand")
        return z_out, img, score

    # ---- per-module VJP diagnostics -------------------------------------------------------------
    def unet_vjp(self, z, step_index, g_eps2):
        z, g = self._f(z), self._f(g_eps2)
        out = torch.empty_like(z)
        self._chk(self.L.dd_unet_vjp(self._h, _p(z), step_index, _p(g), _p(out), z.shape[0], _stream()), "dd_unet_vjp")
        return out

    def decode_vjp(self, z, g_image):
        z, g = self._f(z), self._f(g_image)
        out = torch.empty_like(z)
        self._chk(self.L.dd_decode_vjp(self._h, _p(z), _p(g), _p(out), z.shape[0], _stream()), "dd_decode_vjp")
        return out

    def guide_vjp(self, images, g_feats):
        x, g = self._f(images), self._f(g_feats)
        out = torch.empty_like(x)
        self._chk(self.L.dd_guide_vjp(self._h, _p(x), _p(g), _p(out), x.shape[0], _stream()), "dd_guide_vjp")
        return out

    def profile_enable(self, on=True):
        self._chk(self.L.dd_profile_enable(self._h, int(on)), "dd_profile_enable")

    def profile_read(self):
        out = (C.c_double * 12)()
        self._chk(self.L.dd_profile_read(self._h, out), "dd_profile_read")
        names = ["conv_gemm", "attention", "norm", "other"]
        return {n: {"ms": out[3 * k], "flops": out[3 * k + 1], "ops": int(out[3 * k + 2])} for k, n in enumerate(names)}

    def debug_tensor(self, prog, idx, grad=False, instance=0):
        prog = prog | (instance << 4)
        info = (C.c_int * 4)()
        self._chk(self.L.dd_debug_tensor(self._h, prog, idx, int(grad), None, info), "dd_debug_tensor")
        rows, Cc, ld, f32 = list(info)
        out = torch.empty(rows * ld, dtype=torch.float32)
        self._chk(self.L.dd_debug_tensor(self._h, prog, idx, int(grad), vp(out.data_ptr()), info), "dd_debug_tensor")
        return out.reshape(rows, ld)[:, :Cc]

    def guided_image(self, instance=0):
        """The fp32 image [B,3,8L,8L] (not denormalised) the decoder produced in chained guided step `instance` of the most recent
        guidance call: the exact input of the bicubic resize + guide network (used by the parity tests)."""
        L8 = 8 * self.cfg.latent_size
        return self.debug_tensor(1, -1, instance=instance).reshape(self.B, L8, L8, 3).permute(0, 3, 1, 2).contiguous()

    def set_guide_image(self, image):
        """Parity-test hook (dd_debug_set_images): evaluate the guide of later guided forwards at `image` -- [B,3,8L,8L] for every
        chained step, or [P,B,3,8L,8L] with one image set per chained guided step; None = off."""
        self._override = self._f(image) if image is not None else None
        count = 0 if image is None else (self._override.shape[0] if self._override.dim() == 5 else 1)
        self._chk(self.L.dd_debug_set_images(self._h, _p(self._override), count), "dd_debug_set_images")

    def debug_num_tensors(self, prog):
        return int(self.L.dd_debug_num_tensors(self._h, prog))

    def workspace_bytes(self):
        return int(self.L.dd_workspace_bytes(self._h))

    def flops_last(self):
        return float(self.L.dd_flops_last(self._h))
