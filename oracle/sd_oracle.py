"""CPU ORACLE — test infrastructure only (never imported by the product path).

A plain torch fp32 CPU restatement of the hot path of haoweiz23/DistDiff `generate_data.py`:
the DDIM denoising loop with classifier-free guidance and hierarchical (class + group prototype)
energy guidance. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY STATUS: **parity unpinned for the third-party arithmetic.** The reference ships no golden vectors
or known-answer tests and its arithmetic lives in diffusers / timm / torchvision, none of
which is installed here nor vendored under /root/reference, so the UNet / VAE / ResNet-50 / DDIM modules
below restate the *published* diffusers-0.28 / timm / SD-v1 config definitions.
What IS pinned: the reference's own hot-path functions (`denoise_one_step`, `transform_guidance`,
`direct_guidance`, `tensor_clamp`, `linfball_proj`) are executed from /root/reference by
tests/golden/make_fixtures.py against this oracle's model objects (duck-typed like the diffusers objects)
and the committed fixtures tests/golden/*.pt hold their outputs; tests/test_oracle.py checks the restated
sampler below against those fixtures, and every primitive against torch.nn.functional.

Reference lines restated (file:line in /root/reference):
  denoise_one_step        generate_data.py:109-121
  tensor_clamp / linfball generate_data.py:124-137
  transform_guidance      generate_data.py:687-732
  direct_guidance         generate_data.py:735-767
  main denoise loop       generate_data.py:1161-1228
  encode_image            model_utils.py:29-41   (forward_features -> AdaptiveAvgPool2d(1) -> flatten)
  shard ranges            generate_data.py:1003-1007
  (8f-2) vae.encode / text_encoder(input_ids)[0]   dataloader.py:808-809, 633-646 -> vae_encode, clip_text_encode; the
          latter IS pinned: transformers (importable here) runs its own CLIPTextModel in tests/golden/make_clip_fixture.py.
"""
import math

import torch
import torch.nn.functional as F

from distdiff_amd.config import EngineConfig


# ------------------------------------------------------------------------------------------------
# DDIMScheduler (diffusers, SD-v1 scheduler_config.json) — SURVEY.md row A3
# ------------------------------------------------------------------------------------------------
class DDIMSchedulerOracle:
    def __init__(self, cfg):
        s = cfg.scheduler if isinstance(cfg, EngineConfig) else cfg
        self.cfg = s
        T = s.num_train_timesteps
        if s.beta_schedule == "scaled_linear":
            betas = torch.linspace(s.beta_start ** 0.5, s.beta_end ** 0.5, T, dtype=torch.float32) ** 2
        elif s.beta_schedule == "linear":
            betas = torch.linspace(s.beta_start, s.beta_end, T, dtype=torch.float32)
        else:
            raise NotImplementedError(s.beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if s.set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n):
        s = self.cfg
        self.num_inference_steps = n
        assert s.timestep_spacing == "leading"
        ratio = s.num_train_timesteps // n
        ts = (torch.arange(0, n) * ratio).round().flip(0).to(torch.int64) + s.steps_offset
        self.timesteps = ts
        return ts

    def scale_model_input(self, sample, t=None):
        return sample

    def coefficients(self, t):
        t = int(t)
        prev = t - self.cfg.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a_t, a_p

    def step(self, model_output, timestep, sample, return_dict=True, **kw):
        a_t, a_p = self.coefficients(timestep)
        beta_t = 1 - a_t
        x0 = (sample - beta_t ** 0.5 * model_output) / a_t ** 0.5
        direction = (1 - a_p) ** 0.5 * model_output          # eta = 0 -> std_dev_t = 0
        prev = a_p ** 0.5 * x0 + direction
        return {"prev_sample": prev, "pred_original_sample": x0}

    def add_noise(self, original, noise, timesteps):
        a = self.alphas_cumprod[int(timesteps)]
        return a ** 0.5 * original + (1 - a) ** 0.5 * noise


# ------------------------------------------------------------------------------------------------
# UNet2DConditionModel (SD-1.x topology) — SURVEY.md row A2
# ------------------------------------------------------------------------------------------------
def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def _resnet(sd, p, x, temb, groups, eps):
    h = F.silu(F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    if temb is not None:
        t = F.linear(F.silu(temb), sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"])
        h = h + t[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def _attention(sd, p, x, ctx, heads):
    B, N, Cc = x.shape
    q = F.linear(x, sd[p + ".to_q.weight"], sd.get(p + ".to_q.bias"))
    k = F.linear(ctx, sd[p + ".to_k.weight"], sd.get(p + ".to_k.bias"))
    v = F.linear(ctx, sd[p + ".to_v.weight"], sd.get(p + ".to_v.bias"))
    d = Cc // heads
    q, k, v = (t.reshape(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, N, Cc)
    return F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def _transformer(sd, p, x, ctx, heads, groups, depth=1):
    """diffusers Transformer2DModel: `depth` BasicTransformerBlocks; proj_in / proj_out are 1x1 convolutions (SD-1.x) or, with
    use_linear_projection (SDXL), nn.Linear applied after / before the NCHW <-> tokens permutation: the same arithmetic."""
    B, Cc, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    wi, wo = sd[p + ".proj_in.weight"], sd[p + ".proj_out.weight"]
    if wi.dim() == 4:
        h = F.conv2d(h, wi, sd[p + ".proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, Cc)
    else:
        h = F.linear(h.permute(0, 2, 3, 1).reshape(B, H * W, Cc), wi, sd[p + ".proj_in.bias"])
    for d in range(depth):
        t = p + ".transformer_blocks.%d" % d
        n = F.layer_norm(h, (Cc,), sd[t + ".norm1.weight"], sd[t + ".norm1.bias"], 1e-5)
        h = h + _attention(sd, t + ".attn1", n, n, heads)
        n = F.layer_norm(h, (Cc,), sd[t + ".norm2.weight"], sd[t + ".norm2.bias"], 1e-5)
        h = h + _attention(sd, t + ".attn2", n, ctx, heads)
        n = F.layer_norm(h, (Cc,), sd[t + ".norm3.weight"], sd[t + ".norm3.bias"], 1e-5)
        proj = F.linear(n, sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"])
        hid, gate = proj.chunk(2, dim=-1)
        h = h + F.linear(hid * F.gelu(gate), sd[t + ".ff.net.2.weight"], sd[t + ".ff.net.2.bias"])
    if wo.dim() == 4:
        h = F.conv2d(h.reshape(B, H, W, Cc).permute(0, 3, 1, 2), wo, sd[p + ".proj_out.bias"])
    else:
        h = F.linear(h, wo, sd[p + ".proj_out.bias"]).reshape(B, H, W, Cc).permute(0, 3, 1, 2)
    return h + res


class UNetOracle:
    """Callable like diffusers' UNet2DConditionModel: unet(x, t, encoder_hidden_states, return_dict=False)[0]."""

    def __init__(self, cfg: EngineConfig, sd):
        self.cfg, self.sd = cfg, sd
        self.added_cond = None       # SDXL: {"text_embeds": [B, add_text_dim], "time_ids": [B, 6]} (diffusers' added_cond_kwargs)

    def __call__(self, sample, timestep, encoder_hidden_states, class_labels=None, return_dict=False, added_cond_kwargs=None, **kw):
        u, sd = self.cfg.unet, self.sd
        g, eps = u.norm_num_groups, u.norm_eps
        B = sample.shape[0]
        t = torch.as_tensor(timestep).reshape(-1).expand(B)
        temb = timestep_embedding(t, u.block_out_channels[0], u.flip_sin_to_cos, u.freq_shift).to(sample.dtype)   # diffusers: t_emb.to(dtype=sample.dtype)
        temb = F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
        temb = F.linear(F.silu(temb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
        if u.add_time_dim:
            # addition_embed_type "text_time" (SDXL): emb += add_embedding(cat[text_embeds, add_time_proj(time_ids)])
            ac = added_cond_kwargs or self.added_cond
            tid = timestep_embedding(ac["time_ids"].reshape(-1), u.add_time_dim, True, 0.0).reshape(B, -1)
            a = torch.cat([ac["text_embeds"], tid], dim=-1)
            a = F.linear(a, sd["add_embedding.linear_1.weight"], sd["add_embedding.linear_1.bias"])
            temb = temb + F.linear(F.silu(a), sd["add_embedding.linear_2.weight"], sd["add_embedding.linear_2.bias"])
        h = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
        skips = [h]
        nlev = len(u.block_out_channels)
        for i in range(nlev):
            for j in range(u.layers_per_block):
                h = _resnet(sd, "down_blocks.%d.resnets.%d" % (i, j), h, temb, g, eps)
                if u.down_attn[i]:
                    h = _transformer(sd, "down_blocks.%d.attentions.%d" % (i, j), h, encoder_hidden_states, u.heads(i), g, u.depth(i))
                skips.append(h)
            if i < nlev - 1:
                p = "down_blocks.%d.downsamplers.0.conv" % i
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=1)
                skips.append(h)
        h = _resnet(sd, "mid_block.resnets.0", h, temb, g, eps)
        h = _transformer(sd, "mid_block.attentions.0", h, encoder_hidden_states, u.heads(nlev - 1), g, u.depth(nlev - 1))
        h = _resnet(sd, "mid_block.resnets.1", h, temb, g, eps)
        for i in range(nlev):
            for j in range(u.layers_per_block + 1):
                h = torch.cat([h, skips.pop()], dim=1)
                h = _resnet(sd, "up_blocks.%d.resnets.%d" % (i, j), h, temb, g, eps)
                if u.up_attn[i]:
                    h = _transformer(sd, "up_blocks.%d.attentions.%d" % (i, j), h, encoder_hidden_states, u.heads(nlev - 1 - i), g,
                                     u.depth(nlev - 1 - i))
            if i < nlev - 1:
                p = "up_blocks.%d.upsamplers.0.conv" % i
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        h = F.silu(F.group_norm(h, g, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps))
        h = F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
        return (h,)


# ------------------------------------------------------------------------------------------------
# AutoencoderKL.decode — SURVEY.md row A4
# ------------------------------------------------------------------------------------------------
class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class VAEOracle:
    def __init__(self, cfg: EngineConfig, sd):
        self.cfg, self.sd = cfg, sd
        self.config = _Cfg(scaling_factor=cfg.vae.scaling_factor)

    def decode(self, z, return_dict=False, generator=None):
        v, sd = self.cfg.vae, self.sd
        g, eps = v.norm_num_groups, v.norm_eps
        h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
        h = F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
        h = _resnet(sd, "decoder.mid_block.resnets.0", h, None, g, eps)
        a = "decoder.mid_block.attentions.0"
        B, Cc, H, W = h.shape
        n = F.group_norm(h, g, sd[a + ".group_norm.weight"], sd[a + ".group_norm.bias"], eps)
        n = n.reshape(B, Cc, H * W).transpose(1, 2)
        o = _attention(sd, a, n, n, 1)
        h = h + o.transpose(1, 2).reshape(B, Cc, H, W)
        h = _resnet(sd, "decoder.mid_block.resnets.1", h, None, g, eps)
        nlev = len(v.block_out_channels)
        for i in range(nlev):
            for j in range(v.layers_per_block + 1):
                h = _resnet(sd, "decoder.up_blocks.%d.resnets.%d" % (i, j), h, None, g, eps)
            if i < nlev - 1:
                p = "decoder.up_blocks.%d.upsamplers.0.conv" % i
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        h = F.silu(F.group_norm(h, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps))
        h = F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
        return (h,)


def vae_encode(cfg, sd, images, noise=None):
    """AutoencoderKL.encode(x).latent_dist.sample() * scaling_factor (dataloader.py:808-809), restated from the
    published diffusers-0.28 Encoder: conv_in; DownEncoderBlock2D x levels (resnets, then F.pad(0,1,0,1) + stride-2 3x3 conv);
    mid Res-Attn-Res; GN+SiLU+conv_out; quant_conv; DiagonalGaussianDistribution (logvar clamped to [-30, 20]).
    Returns (latents, moments[mean | clamped logvar]); noise=None gives the mode."""
    v = cfg.vae
    g, eps = v.norm_num_groups, v.norm_eps
    h = F.conv2d(images, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    nlev = len(v.block_out_channels)
    for i in range(nlev):
        for j in range(v.layers_per_block):
            h = _resnet(sd, "encoder.down_blocks.%d.resnets.%d" % (i, j), h, None, g, eps)
        if i < nlev - 1:
            p = "encoder.down_blocks.%d.downsamplers.0.conv" % i
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[p + ".weight"], sd[p + ".bias"], stride=2)
    h = _resnet(sd, "encoder.mid_block.resnets.0", h, None, g, eps)
    a = "encoder.mid_block.attentions.0"
    B, Cc, H, W = h.shape
    n = F.group_norm(h, g, sd[a + ".group_norm.weight"], sd[a + ".group_norm.bias"], eps)
    n = n.reshape(B, Cc, H * W).transpose(1, 2)
    h = h + _attention(sd, a, n, n, 1).transpose(1, 2).reshape(B, Cc, H, W)
    h = _resnet(sd, "encoder.mid_block.resnets.1", h, None, g, eps)
    h = F.silu(F.group_norm(h, g, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], eps))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    m = F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    mean, logvar = m.chunk(2, dim=1)
    logvar = logvar.clamp(-30.0, 20.0)
    z = mean if noise is None else mean + torch.exp(0.5 * logvar) * noise
    return z * v.scaling_factor, torch.cat([mean, logvar], 1)


def clip_text_encode(cfg, sd, input_ids, which=0, hidden_layer=0, pooled=False):
    """transformers CLIPTextModel(input_ids)[0] (dataloader.py:633-646): token + position embeddings, pre-LN layers with
    causal self-attention and quick_gelu / gelu MLP, final LayerNorm.  Pinned against transformers' own CLIPTextModel by
    tests/golden/make_clip_fixture.py (transformers is importable in the build container).
    SDXL's text side (diffusers StableDiffusionXLPipeline.encode_prompt; beyond the reference): which = 1 takes
    cfg.text2 (CLIPTextModelWithProjection); hidden_layer = -2 returns `hidden_states[-2]` (the input of the last layer, no final
    LayerNorm); pooled=True also returns `text_embeds` = text_projection(final_layer_norm(last layer)[input_ids.argmax(-1)]).
    Pinned by the second half of the same fixture."""
    t = cfg.text2 if which else cfg.text
    tm = "text_model."
    ids = input_ids.long()
    B, T = ids.shape
    x = sd[tm + "embeddings.token_embedding.weight"][ids] + sd[tm + "embeddings.position_embedding.weight"][:T][None]
    heads = t.num_attention_heads
    d = t.hidden_size // heads
    mask = torch.full((T, T), float("-inf")).triu(1)
    hidden = [x]
    for l in range(t.num_hidden_layers):
        p = tm + "encoder.layers.%d" % l
        h = F.layer_norm(x, (t.hidden_size,), sd[p + ".layer_norm1.weight"], sd[p + ".layer_norm1.bias"], t.layer_norm_eps)
        q, k, v = (F.linear(h, sd[p + ".self_attn.%s.weight" % n], sd[p + ".self_attn.%s.bias" % n]).reshape(B, T, heads, d).transpose(1, 2)
                   for n in ("q_proj", "k_proj", "v_proj"))
        w = (q @ k.transpose(-1, -2)) * d ** -0.5 + mask
        o = (w.softmax(-1) @ v).transpose(1, 2).reshape(B, T, t.hidden_size)
        x = x + F.linear(o, sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"])
        h = F.layer_norm(x, (t.hidden_size,), sd[p + ".layer_norm2.weight"], sd[p + ".layer_norm2.bias"], t.layer_norm_eps)
        h = F.linear(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])
        h = h * torch.sigmoid(1.702 * h) if t.hidden_act == "quick_gelu" else F.gelu(h)
        x = x + F.linear(h, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
        hidden.append(x)
    last = F.layer_norm(x, (t.hidden_size,), sd[tm + "final_layer_norm.weight"], sd[tm + "final_layer_norm.bias"], t.layer_norm_eps)
    out = last if hidden_layer == 0 else hidden[hidden_layer]
    if not pooled:
        return out
    eos = ids.argmax(-1)                          # the eos token is the largest id of the CLIP vocabulary (first occurrence)
    return out, F.linear(last[torch.arange(B), eos], sd["text_projection.weight"])


def sdxl_encode_prompt(cfg, sd1, sd2, ids1, ids2):
    """diffusers StableDiffusionXLPipeline.encode_prompt for one list of prompts: cat[tower 1, tower 2] of hidden_states[-2] along the
    width, pooled = tower 2's text_embeds."""
    h1 = clip_text_encode(cfg, sd1, ids1, which=0, hidden_layer=-2)
    h2, pooled = clip_text_encode(cfg, sd2, ids2, which=1, hidden_layer=-2, pooled=True)
    return torch.cat([h1, h2], dim=-1), pooled


class ImageProcessorOracle:
    """VaeImageProcessor.postprocess(output_type='pt'): denormalize = (x/2+0.5).clamp(0,1) — row A5."""

    def postprocess(self, image, output_type="pt", do_denormalize=None):
        if do_denormalize is None:
            do_denormalize = [True] * image.shape[0]
        return torch.stack([(image[i] / 2 + 0.5).clamp(0, 1) if do_denormalize[i] else image[i] for i in range(image.shape[0])])


# ------------------------------------------------------------------------------------------------
# timm resnet50 + model_utils.add_encoder_image_method (eval-mode BN) — SURVEY.md row A7
# ------------------------------------------------------------------------------------------------
class GuideOracle:
    def __init__(self, cfg: EngineConfig, sd):
        self.cfg, self.sd = cfg, sd

    def _bn(self, x, p):
        sd = self.sd
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0,
                            self.cfg.guide.bn_eps)

    def forward_features(self, x):
        g, sd = self.cfg.guide, self.sd
        x = F.relu(self._bn(F.conv2d(x, sd["conv1.weight"], None, stride=2, padding=3), "bn1"))
        x = F.max_pool2d(x, 3, 2, 1)
        for li, nb in enumerate(g.blocks):
            for bi in range(nb):
                p = "layer%d.%d" % (li + 1, bi)
                stride = 2 if (bi == 0 and li > 0) else 1
                sc = x
                o = F.relu(self._bn(F.conv2d(x, sd[p + ".conv1.weight"]), p + ".bn1"))
                w2 = sd[p + ".conv2.weight"]         # timm Bottleneck: groups = cardinality (resnext50_32x4d), 1 otherwise
                o = F.relu(self._bn(F.conv2d(o, w2, stride=stride, padding=1, groups=o.shape[1] // w2.shape[1]), p + ".bn2"))
                o = self._bn(F.conv2d(o, sd[p + ".conv3.weight"]), p + ".bn3")
                if (p + ".downsample.0.weight") in sd:
                    sc = self._bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), p + ".downsample.1")
                x = F.relu(o + sc)
        return x

    def encode_image(self, x, pooling="avg"):
        f = self.forward_features(x)
        f = F.adaptive_avg_pool2d(f, (1, 1)) if pooling == "avg" else F.adaptive_max_pool2d(f, (1, 1))
        return torch.flatten(f, 1)


class GuideOracleMBV2(GuideOracle):
    """timm mobilenetv2_100 forward_features (+ the pooling of model_utils.py:29-41), restated from the published timm definition."""

    def forward_features(self, x):
        g, sd = self.cfg.guide, self.sd
        r6 = lambda t: F.relu6(t)
        x = r6(self._bn(F.conv2d(x, sd["conv_stem.weight"], None, stride=2, padding=1), "bn1"))
        for s, rep in enumerate(g.mb_repeats):
            for bi in range(rep):
                p = "blocks.%d.%d" % (s, bi)
                stride = g.mb_strides[s] if bi == 0 else 1
                inp = x
                if (p + ".conv_pwl.weight") in sd:
                    o = r6(self._bn(F.conv2d(x, sd[p + ".conv_pw.weight"]), p + ".bn1"))
                    o = r6(self._bn(F.conv2d(o, sd[p + ".conv_dw.weight"], stride=stride, padding=1, groups=o.shape[1]), p + ".bn2"))
                    o = self._bn(F.conv2d(o, sd[p + ".conv_pwl.weight"]), p + ".bn3")
                else:
                    o = r6(self._bn(F.conv2d(x, sd[p + ".conv_dw.weight"], stride=stride, padding=1, groups=x.shape[1]), p + ".bn1"))
                    o = self._bn(F.conv2d(o, sd[p + ".conv_pw.weight"]), p + ".bn2")
                x = o + inp if (stride == 1 and o.shape[1] == inp.shape[1]) else o
        return r6(self._bn(F.conv2d(x, sd["conv_head.weight"]), "bn2"))


class GuideOracleViT:
    """open_clip VisionTransformer.forward (image tower of 'ViT-B-32'; `model.encode_image` of the CLIP guide, model_utils.py:80-87),
    restated from the published open_clip definition (open_clip is not installed: unpinned like the other third-party modules):
    conv1 -> [class_embedding; patches] + positional_embedding -> ln_pre -> residual attention blocks (nn.MultiheadAttention with the
    fused in_proj, MLP c_fc -> GELU -> c_proj) -> ln_post -> class token -> @ proj."""

    def __init__(self, cfg: EngineConfig, sd):
        self.cfg, self.sd = cfg, sd

    def encode_image(self, x, pooling="avg"):
        g, sd, v = self.cfg.guide, self.sd, "visual."
        W, H = g.vit_width, g.vit_heads
        x = F.conv2d(x, sd[v + "conv1.weight"], None, stride=g.vit_patch)
        B = x.shape[0]
        x = x.reshape(B, W, -1).permute(0, 2, 1)
        x = torch.cat([sd[v + "class_embedding"].to(x.dtype).expand(B, 1, -1), x], dim=1) + sd[v + "positional_embedding"]
        x = F.layer_norm(x, (W,), sd[v + "ln_pre.weight"], sd[v + "ln_pre.bias"], 1e-5)
        act = F.gelu if g.vit_act == "gelu" else (lambda t: t * torch.sigmoid(1.702 * t))
        for l in range(g.vit_layers):
            r = v + "transformer.resblocks.%d" % l
            n = F.layer_norm(x, (W,), sd[r + ".ln_1.weight"], sd[r + ".ln_1.bias"], 1e-5)
            qkv = F.linear(n, sd[r + ".attn.in_proj_weight"], sd[r + ".attn.in_proj_bias"])
            q, k, vv = (t.reshape(B, -1, H, W // H).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
            a = F.scaled_dot_product_attention(q, k, vv).transpose(1, 2).reshape(B, -1, W)
            x = x + F.linear(a, sd[r + ".attn.out_proj.weight"], sd[r + ".attn.out_proj.bias"])
            n = F.layer_norm(x, (W,), sd[r + ".ln_2.weight"], sd[r + ".ln_2.bias"], 1e-5)
            x = x + F.linear(act(F.linear(n, sd[r + ".mlp.c_fc.weight"], sd[r + ".mlp.c_fc.bias"])), sd[r + ".mlp.c_proj.weight"],
                             sd[r + ".mlp.c_proj.bias"])
        pooled = F.layer_norm(x[:, 0], (W,), sd[v + "ln_post.weight"], sd[v + "ln_post.bias"], 1e-5)
        return pooled @ sd[v + "proj"]


# ------------------------------------------------------------------------------------------------
# Sampler: restatement of generate_data.py:109-137, 687-767, 1161-1228
# ------------------------------------------------------------------------------------------------
class SamplerArgs:
    """The subset of the reference's module-global `args` the hot path reads."""

    def __init__(self, **kw):
        self.do_classifier_free_guidance = True
        self.guidance_scale = 7.5
        self.gs = 1.0
        self.ls = 1.0
        self.rho = 10.0
        self.guidance_period = 2
        self.guidance_step = 20
        self.constraint_value = 0.2
        self.strength = 0.5
        self.guidance_type = "transform_guidance"
        self.num_inference_steps = 50
        self.__dict__.update(kw)


def denoise_one_step(args, latents, scheduler, t, unet, prompt_embeds):
    """generate_data.py:109-121."""
    x = torch.cat([latents] * 2) if args.do_classifier_free_guidance else latents
    x = scheduler.scale_model_input(x, t)
    noise_pred = unet(x, t, prompt_embeds, class_labels=None, return_dict=False)[0]
    if args.do_classifier_free_guidance:
        u, c = noise_pred.chunk(2)
        noise_pred = u + args.guidance_scale * (c - u)
    out = scheduler.step(noise_pred, t, latents, return_dict=True)
    return out["prev_sample"], out["pred_original_sample"]


def energy(args, feats, targets, global_proto, local_proto):
    """generate_data.py:707-717 / 749-759."""
    score = 0.0
    if global_proto is not None:
        gp = global_proto[targets]
        score = score + torch.norm(feats - gp, dim=1, p=2).mean() * args.gs
    if local_proto is not None:
        lp = local_proto[targets]
        idx = torch.argmax(torch.bmm(feats.unsqueeze(1), lp.permute(0, 2, 1)), -1)
        lp = lp[torch.arange(lp.size(0)), idx.squeeze(-1)]
        score = score + torch.norm(feats - lp, dim=1, p=2).mean() * args.ls
    return score


def _guide_features(vae, guide, x0, size, image_at=None):
    img = vae.decode(x0 / vae.config.scaling_factor)[0]          # postprocess(do_denormalize=False) is the identity
    if image_at is not None:
        # test hook (not in the reference): evaluate the guide -- i.e. its ReLU / max-pool masks -- AT a given image while the
        # gradient still flows through this decoder (straight-through).  The guide's input-gradient is piecewise constant in the
        # image, so a comparison of two implementations of the energy gradient is only meaningful at the same image.
        img = img + (image_at - img).detach()
    img = F.interpolate(img, size=(size, size), mode="bicubic")
    return guide.encode_image(img).float()


def transform_guidance(args, latents, targets, sub_timesteps, scheduler, unet, prompt_embeds, vae, guide, e, b,
                       global_proto, local_proto, guide_size=224, images_at=None):
    """generate_data.py:687-732 with the random draws (e ~ U[0,1), b ~ N(0,1), :692-695) passed in explicitly."""
    e = e.clone().requires_grad_(True)
    b = b.clone().requires_grad_(True)
    x = latents * (1 + e) + b
    score = 0.0
    for k, t in enumerate(sub_timesteps):
        x, x0 = denoise_one_step(args, x, scheduler, t, unet, prompt_embeds)
        feats = _guide_features(vae, guide, x0, guide_size, images_at[k] if images_at is not None else None)
        score = score + energy(args, feats, targets, global_proto, local_proto)
    score = score / args.guidance_period
    ge, gb = torch.autograd.grad(score, [e, b])
    e2 = e.detach() - args.rho * ge
    b2 = b.detach() - args.rho * gb
    new = latents * (1 + e2) + b2
    lo, hi = latents - args.constraint_value, latents + args.constraint_value
    new = torch.where(new < lo, lo, new)        # tensor_clamp: lower bound first, then upper (:129-132)
    new = torch.where(new > hi, hi, new)
    return new.detach(), score.detach(), (ge, gb)


def direct_guidance(args, latents, targets, t, scheduler, unet, prompt_embeds, vae, guide, global_proto, local_proto,
                    guide_size=224, image_at=None):
    """generate_data.py:735-767."""
    z = latents.clone().requires_grad_(True)
    z_next, x0 = denoise_one_step(args, z, scheduler, t, unet, prompt_embeds)
    feats = _guide_features(vae, guide, x0, guide_size, image_at)
    feats = feats / feats.norm(dim=-1, keepdim=True)
    score = energy(args, feats, targets, global_proto, local_proto)
    (g,) = torch.autograd.grad(score, z)
    z_next = z_next - args.rho * g
    return z_next.detach(), x0.detach(), score.detach(), g


def start_index(strength, n):
    return int((1 - strength) * n)          # generate_data.py:1174


def guide_timesteps(timesteps, guidance_step, guidance_period):
    n = len(timesteps)
    return [int(t) for t in timesteps[n - guidance_step: n - guidance_step + guidance_period]]   # :1178


def shard_range(total, total_split, split):
    """generate_data.py:1003-1007."""
    per = math.ceil(total / total_split)
    if split == total_split - 1 and total < per * (split + 1):
        return list(range(per * split, total))
    return list(range(per * split, per * (split + 1)))


def expand_one(args, cfg, models, image_latents, noise, e, b, prompt_embeds, neg_embeds, targets, global_proto, local_proto,
               trace=None):
    """One (batch, expand-index) of the main loop, generate_data.py:1161-1228. Returns (final latents, image in [0,1], score)."""
    unet, vae, guide, sched = models
    timesteps = sched.set_timesteps(args.num_inference_steps)
    si = start_index(args.strength, len(timesteps))
    z = sched.add_noise(image_latents, noise, timesteps[si])
    gts = guide_timesteps(timesteps, args.guidance_step, args.guidance_period) if args.guidance_type else []
    embeds = torch.cat([neg_embeds, prompt_embeds]) if args.do_classifier_free_guidance else prompt_embeds
    score = None
    gsz = cfg.guide.input_size
    for t in timesteps[si:]:
        t = int(t)
        if gts and t == gts[0] and args.guidance_type == "transform_guidance":
            if trace is not None:
                trace.append(("transform_guidance", t))
            z, score, _ = transform_guidance(args, z, targets, gts, sched, unet, embeds, vae, guide, e, b, global_proto,
                                             local_proto, gsz)
            with torch.no_grad():
                z, _ = denoise_one_step(args, z, sched, t, unet, embeds)
            if trace is not None:
                trace.append(("denoise", t))
        elif gts and t in gts and args.guidance_type == "direct_guidance":
            if trace is not None:
                trace.append(("direct_guidance", t))
            z, _, score, _ = direct_guidance(args, z, targets, t, sched, unet, embeds, vae, guide, global_proto, local_proto, gsz)
        else:
            if trace is not None:
                trace.append(("denoise", t))
            with torch.no_grad():
                z, _ = denoise_one_step(args, z, sched, t, unet, embeds)
    with torch.no_grad():
        img = vae.decode(z / vae.config.scaling_factor)[0]
        img = (img / 2 + 0.5).clamp(0, 1)
    return z, img, score


def transform_guidance_3stage(args, cfg, models, z, targets, gts, emb, e0, b0, Pc, Pg):
    """transform_guidance (generate_data.py:687-732) for guidance_period = 2 with the chain rule applied per step, so that only ONE
    step's autograd graph is alive at a time (two chained graphs of the SD-1.5 oracle do not fit 64 GB): the same quantity
    torch.autograd.grad returns for the reference's loop, checked against the one-graph gradient at the tiny config
    (tests/golden/make_fullsize_p2_fixture.py).  Returns (new latents, score, (ge, gb))."""
    import gc
    unet, vae, guide, sched = models
    gsz = cfg.guide.input_size
    t0, t1 = gts
    P = args.guidance_period
    assert P == 2 and len(gts) == 2

    def E_of(x0):
        img = vae.decode(x0 / cfg.vae.scaling_factor)[0]                  # :701
        gi = F.interpolate(img, size=(gsz, gsz), mode="bicubic")          # :704
        return energy(args, guide.encode_image(gi).float(), targets, Pc, Pg)

    with torch.no_grad():                                                 # stage 1: first chained step, no graph
        z0 = z * (1 + e0) + b0                                            # :696
        z1, _ = denoise_one_step(args, z0, sched, t0, unet, emb)
    z1r = z1.detach().clone().requires_grad_(True)                        # stage 2: dE2/dz1
    _, x0_2 = denoise_one_step(args, z1r, sched, t1, unet, emb)
    E2 = E_of(x0_2)
    (g_z1,) = torch.autograd.grad(E2, z1r)
    E2 = E2.detach()
    del x0_2, z1r
    gc.collect()
    e = e0.clone().requires_grad_(True)                                   # stage 3: first step with its graph
    b = b0.clone().requires_grad_(True)
    z0 = z * (1 + e) + b
    z1g, x0_1 = denoise_one_step(args, z0, sched, t0, unet, emb)
    E1 = E_of(x0_1)
    total = (E1 + (g_z1 * z1g).sum()) / P                                 # d(E1 + E2)/P through z1
    ge, gb = torch.autograd.grad(total, [e, b])
    score = ((E1.detach() + E2) / P)                                      # :719
    del z1g, x0_1, total
    gc.collect()
    e2, b2 = e0 - args.rho * ge, b0 - args.rho * gb                       # :723-724
    new = z * (1 + e2) + b2
    lo, hi = z - args.constraint_value, z + args.constraint_value
    new = torch.where(new < lo, lo, new)                                  # tensor_clamp: lower bound first (:129-132)
    new = torch.where(new > hi, hi, new)
    return new.detach(), score, (ge, gb)



def build_models(cfg, weights):
    guide = {"vit": GuideOracleViT, "mbv2": GuideOracleMBV2}.get(cfg.guide.kind, GuideOracle)(cfg, weights["guide"])
    return (UNetOracle(cfg, weights["unet"]), VAEOracle(cfg, weights["vae"]), guide, DDIMSchedulerOracle(cfg))
