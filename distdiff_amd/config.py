"""Architecture / sampler configuration of the expansion engine.

Every constant that the reference obtains from the Hugging Face model directory
(`unet/config.json`, `vae/config.json`, `scheduler/scheduler_config.json`; reference loads them at
generate_data.py:863-922) lives here with the Stable-Diffusion-v1.x values as defaults, and is
overridden from those JSON files when a local model directory is given (SURVEY.md section 8a note).
"""
import json
import os
from dataclasses import asdict, dataclass, field
from typing import Optional, Tuple


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # SD-1.x: CrossAttnDownBlock2D x3 + DownBlock2D ; UpBlock2D + CrossAttnUpBlock2D x3
    down_attn: Tuple[bool, ...] = (True, True, True, False)
    up_attn: Tuple[bool, ...] = (False, True, True, True)
    num_heads: int = 8                 # diffusers' `attention_head_dim` = 8 is the head COUNT for SD-1.x
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    freq_shift: float = 0.0
    flip_sin_to_cos: bool = True
    # SDXL-style UNets (BASELINE.json configs[4]; unet/config.json `transformer_layers_per_block`, per-level `attention_head_dim`,
    # `addition_embed_type: text_time`): empty / 0 = the SD-1.x structure (one transformer block per attention, num_heads everywhere,
    # no additional conditioning)
    transformer_depth: Tuple[int, ...] = ()
    level_heads: Tuple[int, ...] = ()
    add_time_dim: int = 0              # addition_time_embed_dim (256): sinusoid width of each of the 6 time ids
    add_text_dim: int = 0              # pooled text embedding width (1280); add_embedding input = add_text_dim + 6 * add_time_dim

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4

    def depth(self, level):
        return self.transformer_depth[level] if self.transformer_depth else 1

    def heads(self, level):
        return self.level_heads[level] if self.level_heads else self.num_heads


@dataclass
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215


@dataclass
class TextConfig:
    """CLIPTextModel of the SD-1.x repo (text_encoder/config.json: CLIP ViT-L/14 text tower)."""
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    projection_dim: int = 0            # second tower only (CLIPTextModelWithProjection): width of text_projection's output


@dataclass
class GuideConfig:
    arch: str = "resnet50"
    stem_channels: int = 64
    planes: Tuple[int, ...] = (64, 128, 256, 512)
    blocks: Tuple[int, ...] = (3, 4, 6, 3)
    expansion: int = 4
    cardinality: int = 1               # timm Bottleneck: width = int(planes * base_width / 64) * cardinality; conv2 has `cardinality` groups
    base_width: int = 64               # resnet50 (1, 64), resnext50_32x4d (32, 4), wide_resnet50_2 (1, 128)  (model_utils.py:47-79)
    bn_eps: float = 1e-5
    input_size: int = 224              # F.interpolate(..., (224,224), 'bicubic') generate_data.py:704
    # kind "vit": the image tower of an open_clip ViT (open_clip_vit_b32 = 'ViT-B-32': width 768, 12 layers, 12 heads, patch 32, MLP
    # 4x, nn.GELU, projection to 512); encode_image = CLIP.encode_image = visual(x) (model_utils.py:80-87)
    kind: str = "resnet"
    vit_width: int = 768
    vit_layers: int = 12
    vit_heads: int = 12
    vit_patch: int = 32
    vit_mlp: int = 3072
    vit_out: int = 512
    vit_act: str = "gelu"
    # kind "mbv2": timm mobilenetv2_100 (model_utils.py:64-71): stem 32, stages (out channels, repeats, first stride), expansion 6
    # from stage 1 on (stage 0 is a depthwise-separable block), conv_head 1280
    mb_stem: int = 32
    mb_channels: Tuple[int, ...] = (16, 24, 32, 64, 96, 160, 320)
    mb_repeats: Tuple[int, ...] = (1, 2, 3, 4, 3, 3, 1)
    mb_strides: Tuple[int, ...] = (1, 2, 2, 2, 1, 2, 1)
    mb_expand: int = 6
    mb_head: int = 1280

    @property
    def feature_dim(self):
        if self.kind == "vit":
            return self.vit_out
        if self.kind == "mbv2":
            return self.mb_head
        return self.planes[-1] * self.expansion

    def width(self, planes):
        return int(planes * self.base_width / 64) * self.cardinality


# the reference's five guide architectures (model_utils.py:47-87): the three timm Bottleneck ResNets and mobilenetv2_100 (exact fp32
# programs: ReLU / ReLU6 masks) and the open_clip ViT-B/32 image tower (bf16 program)
GUIDE_ARCHS = {
    "resnet50": dict(cardinality=1, base_width=64),
    "resnext50": dict(cardinality=32, base_width=4),          # timm resnext50_32x4d
    "wideresnet50": dict(cardinality=1, base_width=128),      # timm wide_resnet50_2
    "open_clip_vit_b32": dict(kind="vit"),                    # open_clip 'ViT-B-32' image tower (the reference's default --arch)
    "mobilenetv2": dict(kind="mbv2"),                         # timm mobilenetv2_100
}


def guide_config(arch="resnet50", **kw):
    if arch not in GUIDE_ARCHS:
        raise NotImplementedError("guide arch %r is not built (built: %s)" % (arch, ", ".join(sorted(GUIDE_ARCHS))))
    return GuideConfig(arch=arch, **GUIDE_ARCHS[arch], **kw)


@dataclass
class SchedulerConfig:
    """DDIMScheduler as configured by the SD-1.x repo scheduler_config.json."""
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    beta_schedule: str = "scaled_linear"
    steps_offset: int = 1
    set_alpha_to_one: bool = False
    clip_sample: bool = False
    prediction_type: str = "epsilon"
    timestep_spacing: str = "leading"


@dataclass
class EngineConfig:
    unet: UNetConfig = field(default_factory=UNetConfig)
    vae: VAEConfig = field(default_factory=VAEConfig)
    guide: GuideConfig = field(default_factory=GuideConfig)
    text: TextConfig = field(default_factory=TextConfig)
    # SDXL's text side (diffusers StableDiffusionXLPipeline.encode_prompt): a second tower (text_encoder_2/, CLIPTextModelWithProjection),
    # both towers read at hidden_states[-2], prompt embedding = cat[text, text2] along the width, pooled embedding = text2's text_embeds
    text2: Optional[TextConfig] = None
    text_hidden_layer: int = 0         # 0: last_hidden_state (SD-1.x, dataloader.py:633-646); -2: hidden_states[-2]
    force_zeros_for_empty_prompt: bool = False    # model_index.json of the SDXL base repo: the empty negative prompt embeds as zeros
    scheduler: SchedulerConfig = field(default_factory=SchedulerConfig)
    latent_size: int = 64              # 512 / 8
    text_len: int = 77
    max_batch: int = 1                 # train_batch_size (B); the CFG batch is 2B

    def to_dict(self):
        return asdict(self)


def sd15_config(latent_size=64, max_batch=1):
    return EngineConfig(latent_size=latent_size, max_batch=max_batch)


def sdxl_config(latent_size=128, max_batch=1):
    """Stable-Diffusion-XL base UNet (BASELINE.json configs[4]: 1024x1024 -> 128x128 latents) with the SD-family AutoencoderKL
    (scaling_factor 0.13025).  The reference cannot run this model (single text encoder, no added_cond_kwargs: SURVEY.md 8d C5)."""
    cfg = EngineConfig(latent_size=latent_size, max_batch=max_batch)
    cfg.unet = UNetConfig(block_out_channels=(320, 640, 1280), layers_per_block=2, down_attn=(False, True, True), up_attn=(True, True, False),
                          num_heads=10, cross_attention_dim=2048, transformer_depth=(1, 2, 10), level_heads=(5, 10, 20),
                          add_time_dim=256, add_text_dim=1280)
    cfg.vae.scaling_factor = 0.13025
    # text_encoder/ = CLIP ViT-L/14 text tower (the SD-1.x one), text_encoder_2/ = OpenCLIP ViT-bigG/14 text tower
    cfg.text2 = TextConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20, hidden_act="gelu",
                           projection_dim=1280)
    cfg.text_hidden_layer = -2
    cfg.force_zeros_for_empty_prompt = True
    return cfg


def tiny_sdxl_config(latent_size=16, max_batch=2):
    """The SDXL structure at test size: 3 levels, attention on the last two, transformer depths (1, 2, 3), per-level head counts,
    text_time additional conditioning."""
    cfg = tiny_config(latent_size, max_batch)
    cfg.unet = UNetConfig(block_out_channels=(64, 128, 128), layers_per_block=1, down_attn=(False, True, True), up_attn=(True, True, False),
                          num_heads=2, cross_attention_dim=96, norm_num_groups=8, transformer_depth=(1, 2, 3), level_heads=(2, 2, 4),
                          add_time_dim=8, add_text_dim=24)
    # two text towers whose widths add up to the cross-attention width (32 + 64, head dim 32 like the tiny SD-1.x tower); the second
    # one projects its pooled token to add_text_dim
    cfg.text = TextConfig(vocab_size=97, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=1,
                          max_position_embeddings=13)
    cfg.text2 = TextConfig(vocab_size=97, hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                           max_position_embeddings=13, hidden_act="gelu", projection_dim=24)
    cfg.text_hidden_layer = -2
    return cfg


def tiny_config(latent_size=16, max_batch=2):
    """Small architecture with the same topology (all block types, up/down sampling, cross attention,
    GEGLU, VAE attention, bottleneck guide) used by the parity tests, golden vectors and smoke()."""
    return EngineConfig(
        unet=UNetConfig(block_out_channels=(64, 128, 128, 128), layers_per_block=1, num_heads=2,
                        cross_attention_dim=64, norm_num_groups=8),
        vae=VAEConfig(block_out_channels=(32, 64, 64, 64), layers_per_block=1, norm_num_groups=8),
        guide=GuideConfig(stem_channels=16, planes=(16, 32, 32, 64), blocks=(1, 2, 1, 1), input_size=56),
        text=TextConfig(vocab_size=97, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                        max_position_embeddings=13),
        latent_size=latent_size, text_len=13, max_batch=max_batch)


def from_model_dir(path, latent_size=64, max_batch=1):
    """Populates the config from a local HF Stable-Diffusion directory when its JSON files exist."""
    cfg = sd15_config(latent_size, max_batch)
    cfg.text_len = None

    def _load(sub):
        p = os.path.join(path, sub)
        return json.load(open(p)) if os.path.exists(p) else None

    u = _load("unet/config.json")
    if u:
        cfg.unet.block_out_channels = tuple(u.get("block_out_channels", cfg.unet.block_out_channels))
        cfg.unet.layers_per_block = u.get("layers_per_block", cfg.unet.layers_per_block)
        cfg.unet.cross_attention_dim = u.get("cross_attention_dim", cfg.unet.cross_attention_dim)
        cfg.unet.num_heads = u.get("attention_head_dim", cfg.unet.num_heads)
        cfg.unet.norm_num_groups = u.get("norm_num_groups", cfg.unet.norm_num_groups)
        cfg.unet.norm_eps = u.get("norm_eps", cfg.unet.norm_eps)
        cfg.unet.freq_shift = u.get("freq_shift", cfg.unet.freq_shift)
        cfg.unet.flip_sin_to_cos = u.get("flip_sin_to_cos", cfg.unet.flip_sin_to_cos)
        if "down_block_types" in u:
            cfg.unet.down_attn = tuple("CrossAttn" in t for t in u["down_block_types"])
            cfg.unet.up_attn = tuple("CrossAttn" in t for t in u["up_block_types"])
        nl = len(cfg.unet.block_out_channels)
        if isinstance(u.get("attention_head_dim"), (list, tuple)):          # SDXL: per-level head COUNTS
            cfg.unet.level_heads = tuple(u["attention_head_dim"])
            cfg.unet.num_heads = cfg.unet.level_heads[-1]
        tl = u.get("transformer_layers_per_block", 1)
        cfg.unet.transformer_depth = tuple(tl) if isinstance(tl, (list, tuple)) else ((tl,) * nl if tl != 1 else ())
        if u.get("addition_embed_type") == "text_time":
            cfg.unet.add_time_dim = u.get("addition_time_embed_dim", 256)
            cfg.unet.add_text_dim = u.get("projection_class_embeddings_input_dim", 2816) - 6 * cfg.unet.add_time_dim
        elif u.get("addition_embed_type"):
            raise NotImplementedError("unet addition_embed_type=%r is not built" % u["addition_embed_type"])
    v = _load("vae/config.json")
    if v:
        cfg.vae.block_out_channels = tuple(v.get("block_out_channels", cfg.vae.block_out_channels))
        cfg.vae.layers_per_block = v.get("layers_per_block", cfg.vae.layers_per_block)
        cfg.vae.scaling_factor = v.get("scaling_factor", cfg.vae.scaling_factor)
        cfg.vae.norm_num_groups = v.get("norm_num_groups", cfg.vae.norm_num_groups)
    t = _load("text_encoder/config.json")
    if t:
        for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                  "max_position_embeddings", "hidden_act", "layer_norm_eps"):
            if k in t:
                setattr(cfg.text, k, t[k])
    t2 = _load("text_encoder_2/config.json")
    if t2:                                                                  # SDXL layout: two towers, both read at hidden_states[-2]
        cfg.text2 = TextConfig()
        for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                  "max_position_embeddings", "hidden_act", "layer_norm_eps", "projection_dim"):
            if k in t2:
                setattr(cfg.text2, k, t2[k])
        cfg.text_hidden_layer = -2
        mi = _load("model_index.json") or {}
        cfg.force_zeros_for_empty_prompt = bool(mi.get("force_zeros_for_empty_prompt", True))
    cfg.text_len = cfg.text.max_position_embeddings
    if v and "latent_channels" in v:
        cfg.vae.latent_channels = v["latent_channels"]
    if u:
        cfg.unet.in_channels = u.get("in_channels", cfg.unet.in_channels)
        cfg.unet.out_channels = u.get("out_channels", cfg.unet.out_channels)
    s = _load("scheduler/scheduler_config.json")
    if s:
        for k in ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "steps_offset", "set_alpha_to_one",
                  "clip_sample", "prediction_type", "timestep_spacing"):
            if k in s:
                setattr(cfg.scheduler, k, s[k])
    # the engine implements the sampler the SD-1.x repos configure; anything else would run with silently wrong
    # results, so it is refused
    sc = cfg.scheduler
    if sc.prediction_type != "epsilon":
        raise NotImplementedError("scheduler prediction_type=%r: only epsilon-prediction models are built" % sc.prediction_type)
    if sc.clip_sample:
        raise NotImplementedError("scheduler clip_sample=true is not built (SD-1.x uses clip_sample=false)")
    if sc.timestep_spacing != "leading":
        raise NotImplementedError("scheduler timestep_spacing=%r: only 'leading' is built" % sc.timestep_spacing)
    if sc.beta_schedule != "scaled_linear":
        raise NotImplementedError("scheduler beta_schedule=%r: only 'scaled_linear' is built" % sc.beta_schedule)
    return cfg
