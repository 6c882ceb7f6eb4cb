"""Weight sources for the engine: seeded synthetic weights with the exact Hugging Face / timm key
names and shapes (there are no checkpoints offline), or real `safetensors` / torch checkpoints from a
local model directory (same key map, so real SD-1.x / guide weights drop in where they exist).

Key names follow the state dicts the reference loads: diffusers UNet2DConditionModel / AutoencoderKL
(generate_data.py:912-922) and the timm resnet50 guide checkpoint `{'state_dict': ...}` with an
optional `module.` prefix (model_utils.py:89-101, train.py:201-207).
"""
import hashlib
import math
import os

import torch

from .config import EngineConfig


def _seed(key, base):
    return (int(hashlib.md5(key.encode()).hexdigest()[:8], 16) + base) % (2 ** 31)


def _randn(key, shape, scale, base_seed):
    g = torch.Generator().manual_seed(_seed(key, base_seed))
    return torch.randn(shape, generator=g) * scale


class _Builder:
    def __init__(self, prefix, seed):
        self.sd = {}
        self.prefix = prefix
        self.seed = seed

    def conv(self, name, cout, cin, k, bias=True):
        self.sd[name + ".weight"] = _randn(self.prefix + name + ".w", (cout, cin, k, k), 1.0 / math.sqrt(cin * k * k), self.seed)
        if bias:
            self.sd[name + ".bias"] = _randn(self.prefix + name + ".b", (cout,), 0.02, self.seed)

    def linear(self, name, cout, cin, bias=True):
        self.sd[name + ".weight"] = _randn(self.prefix + name + ".w", (cout, cin), 1.0 / math.sqrt(cin), self.seed)
        if bias:
            self.sd[name + ".bias"] = _randn(self.prefix + name + ".b", (cout,), 0.02, self.seed)

    def norm(self, name, c):
        self.sd[name + ".weight"] = 1.0 + _randn(self.prefix + name + ".g", (c,), 0.1, self.seed)
        self.sd[name + ".bias"] = _randn(self.prefix + name + ".b", (c,), 0.1, self.seed)

    def bn(self, name, c):
        self.norm(name, c)
        self.sd[name + ".running_mean"] = _randn(self.prefix + name + ".m", (c,), 0.1, self.seed)
        self.sd[name + ".running_var"] = 1.0 + 0.2 * _randn(self.prefix + name + ".v", (c,), 1.0, self.seed).abs()


def unet_resnet_specs(cfg):
    """Yields (key_prefix, cin, cout) of every UNet ResnetBlock2D in execution order plus structure info."""
    u = cfg.unet
    ch = u.block_out_channels
    specs = {"down": [], "mid": [], "up": []}
    out = ch[0]
    for i, c in enumerate(ch):
        inp, out = out, c
        specs["down"].append([(inp if j == 0 else out, out) for j in range(u.layers_per_block)])
    rev = list(reversed(ch))
    out = rev[0]
    for i, c in enumerate(rev):
        prev, out = out, c
        inp = rev[min(i + 1, len(ch) - 1)]
        blk = []
        for j in range(u.layers_per_block + 1):
            skip = inp if j == u.layers_per_block else out
            rin = prev if j == 0 else out
            blk.append((rin + skip, out))
        specs["up"].append(blk)
    return specs


def synthetic_unet(cfg: EngineConfig, seed=0):
    u = cfg.unet
    b = _Builder("unet.", seed)
    ch = u.block_out_channels
    temb = u.time_embed_dim
    b.linear("time_embedding.linear_1", temb, ch[0])
    b.linear("time_embedding.linear_2", temb, temb)
    b.conv("conv_in", ch[0], u.in_channels, 3)
    if u.add_time_dim:                       # SDXL text_time conditioning: add_embedding = TimestepEmbedding(text + 6 time ids -> temb)
        b.linear("add_embedding.linear_1", temb, u.add_text_dim + 6 * u.add_time_dim)
        b.linear("add_embedding.linear_2", temb, temb)

    def resnet(p, cin, cout):
        b.norm(p + ".norm1", cin)
        b.conv(p + ".conv1", cout, cin, 3)
        b.linear(p + ".time_emb_proj", cout, temb)
        b.norm(p + ".norm2", cout)
        b.conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            b.conv(p + ".conv_shortcut", cout, cin, 1)

    def transformer(p, c, depth=1):
        b.norm(p + ".norm", c)
        if u.transformer_depth:              # SDXL: use_linear_projection -> nn.Linear weights [C, C]
            b.linear(p + ".proj_in", c, c)
        else:
            b.conv(p + ".proj_in", c, c, 1)
        for d in range(depth):
            t = p + ".transformer_blocks.%d" % d
            for n in ("norm1", "norm2", "norm3"):
                b.norm(t + "." + n, c)
            for a, kv in (("attn1", c), ("attn2", u.cross_attention_dim)):
                b.linear(t + "." + a + ".to_q", c, c, bias=False)
                b.linear(t + "." + a + ".to_k", c, kv, bias=False)
                b.linear(t + "." + a + ".to_v", c, kv, bias=False)
                b.linear(t + "." + a + ".to_out.0", c, c)
            b.linear(t + ".ff.net.0.proj", 8 * c, c)
            b.linear(t + ".ff.net.2", c, 4 * c)
        if u.transformer_depth:
            b.linear(p + ".proj_out", c, c)
        else:
            b.conv(p + ".proj_out", c, c, 1)

    specs = unet_resnet_specs(cfg)
    for i, blk in enumerate(specs["down"]):
        for j, (cin, cout) in enumerate(blk):
            resnet("down_blocks.%d.resnets.%d" % (i, j), cin, cout)
            if u.down_attn[i]:
                transformer("down_blocks.%d.attentions.%d" % (i, j), cout, u.depth(i))
        if i < len(ch) - 1:
            b.conv("down_blocks.%d.downsamplers.0.conv" % i, ch[i], ch[i], 3)
    resnet("mid_block.resnets.0", ch[-1], ch[-1])
    transformer("mid_block.attentions.0", ch[-1], u.depth(len(ch) - 1))
    resnet("mid_block.resnets.1", ch[-1], ch[-1])
    for i, blk in enumerate(specs["up"]):
        for j, (cin, cout) in enumerate(blk):
            resnet("up_blocks.%d.resnets.%d" % (i, j), cin, cout)
            if u.up_attn[i]:
                transformer("up_blocks.%d.attentions.%d" % (i, j), cout, u.depth(len(ch) - 1 - i))
        if i < len(ch) - 1:
            b.conv("up_blocks.%d.upsamplers.0.conv" % i, blk[-1][1], blk[-1][1], 3)
    b.norm("conv_norm_out", ch[0])
    b.conv("conv_out", u.out_channels, ch[0], 3)
    return b.sd


def synthetic_vae_decoder(cfg: EngineConfig, seed=0):
    v = cfg.vae
    b = _Builder("vae.", seed)
    ch = v.block_out_channels
    top = ch[-1]
    b.conv("post_quant_conv", v.latent_channels, v.latent_channels, 1)
    b.conv("decoder.conv_in", top, v.latent_channels, 3)

    def resnet(p, cin, cout):
        b.norm(p + ".norm1", cin)
        b.conv(p + ".conv1", cout, cin, 3)
        b.norm(p + ".norm2", cout)
        b.conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            b.conv(p + ".conv_shortcut", cout, cin, 1)

    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    b.norm(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        b.linear(a + "." + n, top, top)
    resnet("decoder.mid_block.resnets.1", top, top)
    rev = list(reversed(ch))
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(v.layers_per_block + 1):
            resnet("decoder.up_blocks.%d.resnets.%d" % (i, j), prev if j == 0 else c, c)
        prev = c
        if i < len(ch) - 1:
            b.conv("decoder.up_blocks.%d.upsamplers.0.conv" % i, c, c, 3)
    b.norm("decoder.conv_norm_out", ch[0])
    b.conv("decoder.conv_out", v.out_channels, ch[0], 3)
    return b.sd


def synthetic_vae_encoder(cfg: EngineConfig, seed=0):
    """AutoencoderKL encoder + quant_conv keys (the half of vae/ used by dataloader.py:808)."""
    v = cfg.vae
    b = _Builder("vae.", seed)
    ch = v.block_out_channels
    b.conv("encoder.conv_in", ch[0], v.out_channels, 3)

    def resnet(p, cin, cout):
        b.norm(p + ".norm1", cin)
        b.conv(p + ".conv1", cout, cin, 3)
        b.norm(p + ".norm2", cout)
        b.conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            b.conv(p + ".conv_shortcut", cout, cin, 1)

    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(v.layers_per_block):
            resnet("encoder.down_blocks.%d.resnets.%d" % (i, j), prev if j == 0 else c, c)
        prev = c
        if i < len(ch) - 1:
            b.conv("encoder.down_blocks.%d.downsamplers.0.conv" % i, c, c, 3)
    top = ch[-1]
    resnet("encoder.mid_block.resnets.0", top, top)
    a = "encoder.mid_block.attentions.0"
    b.norm(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        b.linear(a + "." + n, top, top)
    resnet("encoder.mid_block.resnets.1", top, top)
    b.norm("encoder.conv_norm_out", top)
    b.conv("encoder.conv_out", 2 * v.latent_channels, top, 3)
    b.conv("quant_conv", 2 * v.latent_channels, 2 * v.latent_channels, 1)
    return b.sd


def synthetic_text_encoder(cfg: EngineConfig, seed=0, which=0):
    """transformers CLIPTextModel state-dict keys (text_encoder/ of the SD-1.x repo); which = 1: the second tower of an SDXL-style
    model (text_encoder_2/, CLIPTextModelWithProjection: the same keys plus `text_projection.weight` [projection_dim, hidden])."""
    t = cfg.text2 if which else cfg.text
    pre = "text2." if which else "text."
    b = _Builder(pre, seed)
    tm = "text_model."
    b.sd[tm + "embeddings.token_embedding.weight"] = _randn(pre + "tok", (t.vocab_size, t.hidden_size), 0.5, seed)
    b.sd[tm + "embeddings.position_embedding.weight"] = _randn(pre + "pos", (t.max_position_embeddings, t.hidden_size), 0.5, seed)
    for l in range(t.num_hidden_layers):
        p = tm + "encoder.layers.%d" % l
        b.norm(p + ".layer_norm1", t.hidden_size)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            b.linear(p + ".self_attn." + n, t.hidden_size, t.hidden_size)
        b.norm(p + ".layer_norm2", t.hidden_size)
        b.linear(p + ".mlp.fc1", t.intermediate_size, t.hidden_size)
        b.linear(p + ".mlp.fc2", t.hidden_size, t.intermediate_size)
    b.norm(tm + "final_layer_norm", t.hidden_size)
    if which:
        b.sd["text_projection.weight"] = _randn(pre + "proj", (t.projection_dim, t.hidden_size), t.hidden_size ** -0.5, seed)
    return b.sd


def synthetic_guide_vit(cfg: EngineConfig, seed=0, num_classes=100):
    """open_clip VisionTransformer state dict (`visual.*` keys of the CLIP model the reference builds at model_utils.py:80-87)."""
    g = cfg.guide
    b = _Builder("guide.", seed)
    W, v = g.vit_width, "visual."
    b.conv(v + "conv1", W, 3, g.vit_patch, bias=False)
    n_tok = (g.input_size // g.vit_patch) ** 2 + 1
    sc = W ** -0.5
    b.sd[v + "class_embedding"] = _randn("guide.cls", (W,), sc, seed)
    b.sd[v + "positional_embedding"] = _randn("guide.pos", (n_tok, W), sc, seed)
    b.norm(v + "ln_pre", W)
    for l in range(g.vit_layers):
        r = v + "transformer.resblocks.%d" % l
        b.norm(r + ".ln_1", W)
        b.sd[r + ".attn.in_proj_weight"] = _randn("guide." + r + ".qkv.w", (3 * W, W), 1.0 / math.sqrt(W), seed)
        b.sd[r + ".attn.in_proj_bias"] = _randn("guide." + r + ".qkv.b", (3 * W,), 0.02, seed)
        b.linear(r + ".attn.out_proj", W, W)
        b.norm(r + ".ln_2", W)
        b.linear(r + ".mlp.c_fc", g.vit_mlp, W)
        b.linear(r + ".mlp.c_proj", W, g.vit_mlp)
    b.norm(v + "ln_post", W)
    b.sd[v + "proj"] = _randn("guide.proj", (W, g.vit_out), sc, seed)
    b.linear("fc", num_classes, g.vit_out)      # wrap_clip_forward's text-feature classifier head (model_utils.py:14-26): not on the hot path
    return b.sd


def synthetic_guide_mbv2(cfg: EngineConfig, seed=0, num_classes=100):
    """timm mobilenetv2_100 state dict: conv_stem / bn1, blocks.<stage>.<block>.{conv_dw, bn1, conv_pw, bn2} (stage 0, depthwise-separable)
    or {conv_pw, bn1, conv_dw, bn2, conv_pwl, bn3} (inverted residual), conv_head / bn2, classifier."""
    g = cfg.guide
    b = _Builder("guide.", seed)

    def dw(name, c):
        b.sd[name + ".weight"] = _randn("guide." + name + ".w", (c, 1, 3, 3), 1.0 / 3.0, seed)

    b.conv("conv_stem", g.mb_stem, 3, 3, bias=False)
    b.bn("bn1", g.mb_stem)
    inp = g.mb_stem
    for s, (out, rep) in enumerate(zip(g.mb_channels, g.mb_repeats)):
        for bi in range(rep):
            p = "blocks.%d.%d" % (s, bi)
            if s == 0:
                dw(p + ".conv_dw", inp)
                b.bn(p + ".bn1", inp)
                b.conv(p + ".conv_pw", out, inp, 1, bias=False)
                b.bn(p + ".bn2", out)
            else:
                mid = inp * g.mb_expand
                b.conv(p + ".conv_pw", mid, inp, 1, bias=False)
                b.bn(p + ".bn1", mid)
                dw(p + ".conv_dw", mid)
                b.bn(p + ".bn2", mid)
                b.conv(p + ".conv_pwl", out, mid, 1, bias=False)
                b.bn(p + ".bn3", out)
            inp = out
    b.conv("conv_head", g.mb_head, inp, 1, bias=False)
    b.bn("bn2", g.mb_head)
    b.linear("classifier", num_classes, g.mb_head)
    return b.sd


def synthetic_guide(cfg: EngineConfig, seed=0, num_classes=100):
    g = cfg.guide
    if g.kind == "vit":
        return synthetic_guide_vit(cfg, seed, num_classes)
    if g.kind == "mbv2":
        return synthetic_guide_mbv2(cfg, seed, num_classes)
    b = _Builder("guide.", seed)
    b.conv("conv1", g.stem_channels, 3, 7, bias=False)
    b.bn("bn1", g.stem_channels)
    inp = g.stem_channels
    for li, (planes, nb) in enumerate(zip(g.planes, g.blocks)):
        for bi in range(nb):
            p = "layer%d.%d" % (li + 1, bi)
            out = planes * g.expansion
            width = g.width(planes)              # timm Bottleneck: resnext50_32x4d / wide_resnet50_2 widen (and group) conv2
            b.conv(p + ".conv1", width, inp, 1, bias=False)
            b.bn(p + ".bn1", width)
            b.conv(p + ".conv2", width, width // g.cardinality, 3, bias=False)
            b.bn(p + ".bn2", width)
            b.conv(p + ".conv3", out, width, 1, bias=False)
            b.bn(p + ".bn3", out)
            if bi == 0 and (inp != out or li > 0):
                b.conv(p + ".downsample.0", out, inp, 1, bias=False)
                b.bn(p + ".downsample.1", out)
            inp = out
    b.linear("fc", num_classes, inp)
    return b.sd


def synthetic_weights(cfg: EngineConfig, seed=0, num_classes=100, encoders=False):
    """encoders=True adds the stage before the loop (VAE encoder + CLIP text encoder)."""
    w = {"unet": synthetic_unet(cfg, seed), "vae": synthetic_vae_decoder(cfg, seed),
         "guide": synthetic_guide(cfg, seed, num_classes)}
    if encoders:
        w["vae"].update(synthetic_vae_encoder(cfg, seed))
        w["text"] = synthetic_text_encoder(cfg, seed)
        if cfg.text2 is not None:
            w["text2"] = synthetic_text_encoder(cfg, seed, which=1)
    return w


_VAE_LEGACY = (("query", "to_q"), ("key", "to_k"), ("value", "to_v"), ("proj_attn", "to_out.0"))


def normalize_vae_keys(sd):
    """The published SD-1.x AutoencoderKL checkpoints (CompVis v1-4, runwayml v1-5, sd-vae-ft-mse) name the mid-block attention
    projections `query / key / value / proj_attn`; diffusers renames them to `to_q / to_k / to_v / to_out.0` when it loads them
    (generate_data.py:912-916 goes through that loader).  Same rename here, plus 1x1-conv-shaped projection weights squeezed to 2-D."""
    out = {}
    for k, v in sd.items():
        if ".attentions." in k:
            for old, new in _VAE_LEGACY:
                k = k.replace(".%s." % old, ".%s." % new)
            if k.endswith(".weight") and v.dim() == 4 and v.shape[2] == v.shape[3] == 1 and ".group_norm." not in k:
                v = v[:, :, 0, 0]
        out[k] = v
    return out


def load_safetensors_dir(model_dir, sub, names=("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors",
                                                 "model.safetensors", "diffusion_pytorch_model.bin", "pytorch_model.bin")):
    """State dict of one sub-model of a local Hugging Face directory (`unet/`, `vae/`, `text_encoder/`): safetensors first, the
    torch `.bin` pickles older repos ship otherwise.  Keys are the diffusers / transformers names the engine consumes."""
    for n in names:
        p = os.path.join(model_dir, sub, n)
        if not os.path.exists(p):
            continue
        if n.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(p)
        else:
            sd = torch.load(p, map_location="cpu", weights_only=True)
        sd = {k: v.float() for k, v in sd.items()}
        return normalize_vae_keys(sd) if sub == "vae" else sd
    raise FileNotFoundError("no weights under %s/%s (looked for %s)" % (model_dir, sub, ", ".join(names)))


def load_guide_checkpoint(path):
    """Reference checkpoint format: torch.save({'epoch','state_dict','acc','best_acc','optimizer'}) with an
    optional DataParallel `module.` prefix (model_utils.py:89-101)."""
    sd = torch.load(path, map_location="cpu")["state_dict"]
    return {(k[len("module."):] if k.startswith("module.") else k): v.float() for k, v in sd.items()}
