"""Pins of the oracle's restated guide networks to independent third-party implementations (timm and open_clip
are not installed, so `oracle/sd_oracle.py` restates their forward passes from the published definitions).  `transformers` IS
importable and carries its own ResNet (v1.5 bottleneck, stride in the 3x3) and CLIP vision tower: the same weights, renamed to the timm
/ open_clip key names the reference's checkpoints use (model_utils.py:47-55, :80-87), must give the same features.  CPU only.
"""
import pytest
import torch

from distdiff_amd.config import EngineConfig, GuideConfig, guide_config
from oracle import sd_oracle as O

transformers = pytest.importorskip("transformers")


def _randomise_bn(model, g):
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def _hf_resnet_to_timm(sd, depths):
    """transformers ResNetModel state dict -> timm `resnet50` key names."""
    out = {}

    def norm(src, dst):
        for k in ("weight", "bias", "running_mean", "running_var"):
            out[dst + "." + k] = sd[src + ".normalization." + k]

    out["conv1.weight"] = sd["embedder.embedder.convolution.weight"]
    norm("embedder.embedder", "bn1")
    for s, nb in enumerate(depths):
        for i in range(nb):
            src, dst = "encoder.stages.%d.layers.%d" % (s, i), "layer%d.%d" % (s + 1, i)
            for j in range(3):
                out["%s.conv%d.weight" % (dst, j + 1)] = sd["%s.layer.%d.convolution.weight" % (src, j)]
                norm("%s.layer.%d" % (src, j), "%s.bn%d" % (dst, j + 1))
            if (src + ".shortcut.convolution.weight") in sd:
                out[dst + ".downsample.0.weight"] = sd[src + ".shortcut.convolution.weight"]
                norm(src + ".shortcut", dst + ".downsample.1")
    return out


@pytest.mark.parametrize("pooling", ["avg"])
def test_resnet50_restatement_matches_transformers_resnet(pooling):
    from transformers import ResNetConfig, ResNetModel
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    depths = [3, 4, 6, 3]
    hf = ResNetModel(ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=depths,
                                  layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False)).eval()
    _randomise_bn(hf, g)
    x = torch.randn(2, 3, 96, 96, generator=g)
    with torch.no_grad():
        out = hf(x)
        ref_map, ref_feat = out.last_hidden_state, out.pooler_output.flatten(1)
    cfg = EngineConfig(guide=guide_config("resnet50"))
    guide = O.GuideOracle(cfg, _hf_resnet_to_timm(hf.state_dict(), depths))
    with torch.no_grad():
        fmap = guide.forward_features(x)
        feat = guide.encode_image(x, pooling)
    assert fmap.shape == ref_map.shape == (2, 2048, 3, 3)
    assert torch.allclose(fmap, ref_map, rtol=1e-4, atol=1e-5)
    assert torch.allclose(feat, ref_feat, rtol=1e-4, atol=1e-5)


def _hf_clip_vision_to_open_clip(sd, layers):
    """transformers CLIPVisionModelWithProjection state dict -> open_clip `visual.*` key names (the published conversion: fused
    in_proj = [q; k; v], proj = visual_projection^T)."""
    v, out = "vision_model.", {}
    out["visual.conv1.weight"] = sd[v + "embeddings.patch_embedding.weight"]
    out["visual.class_embedding"] = sd[v + "embeddings.class_embedding"]
    out["visual.positional_embedding"] = sd[v + "embeddings.position_embedding.weight"]
    for a, b in (("pre_layrnorm", "ln_pre"), ("post_layernorm", "ln_post")):
        for k in ("weight", "bias"):
            out["visual.%s.%s" % (b, k)] = sd["%s%s.%s" % (v, a, k)]
    for l in range(layers):
        s, d = "%sencoder.layers.%d." % (v, l), "visual.transformer.resblocks.%d." % l
        for k in ("weight", "bias"):
            out[d + "attn.in_proj_" + k] = torch.cat([sd["%sself_attn.%s_proj.%s" % (s, n, k)] for n in "qkv"])
            out[d + "attn.out_proj." + k] = sd[s + "self_attn.out_proj." + k]
            out[d + "ln_1." + k] = sd[s + "layer_norm1." + k]
            out[d + "ln_2." + k] = sd[s + "layer_norm2." + k]
            out[d + "mlp.c_fc." + k] = sd[s + "mlp.fc1." + k]
            out[d + "mlp.c_proj." + k] = sd[s + "mlp.fc2." + k]
    out["visual.proj"] = sd["visual_projection.weight"].t().contiguous()
    return out


@pytest.mark.parametrize("act,hf_act", [("quick_gelu", "quick_gelu"), ("gelu", "gelu")])
def test_vit_restatement_matches_transformers_clip_vision(act, hf_act):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(0)
    W, layers, heads, patch, size, mlp, proj = 64, 3, 4, 8, 32, 256, 48
    hf = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=W, intermediate_size=mlp, num_hidden_layers=layers, num_attention_heads=heads,
                                                        image_size=size, patch_size=patch, projection_dim=proj, hidden_act=hf_act,
                                                        layer_norm_eps=1e-5, attn_implementation="eager")).eval()
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for p in hf.parameters():          # default init leaves the biases at zero and the norms at identity
            p.add_(torch.randn(p.shape, generator=g) * 0.02)
    x = torch.randn(2, 3, size, size, generator=g)
    with torch.no_grad():
        ref = hf(pixel_values=x).image_embeds
    cfg = EngineConfig(guide=GuideConfig(arch="clip_vit", kind="vit", input_size=size, vit_width=W, vit_layers=layers, vit_heads=heads,
                                         vit_patch=patch, vit_mlp=mlp, vit_out=proj, vit_act=act))
    guide = O.GuideOracleViT(cfg, _hf_clip_vision_to_open_clip(hf.state_dict(), layers))
    with torch.no_grad():
        feat = guide.encode_image(x)
    assert feat.shape == ref.shape == (2, proj)
    assert torch.allclose(feat, ref, rtol=1e-4, atol=1e-5)


def _hf_mobilenetv2_to_timm(sd, repeats):
    """transformers MobileNetV2Model state dict -> timm `mobilenetv2_100` key names (the stem of the former holds the first
    depthwise-separable block of the latter)."""
    out = {}

    def cb(src, conv, bn):
        out[conv + ".weight"] = sd[src + ".convolution.weight"]
        for k in ("weight", "bias", "running_mean", "running_var"):
            out[bn + "." + k] = sd[src + ".normalization." + k]

    cb("conv_stem.first_conv", "conv_stem", "bn1")
    cb("conv_stem.conv_3x3", "blocks.0.0.conv_dw", "blocks.0.0.bn1")
    cb("conv_stem.reduce_1x1", "blocks.0.0.conv_pw", "blocks.0.0.bn2")
    i = 0
    for s in range(1, len(repeats)):
        for b in range(repeats[s]):
            p = "blocks.%d.%d" % (s, b)
            cb("layer.%d.expand_1x1" % i, p + ".conv_pw", p + ".bn1")
            cb("layer.%d.conv_3x3" % i, p + ".conv_dw", p + ".bn2")
            cb("layer.%d.reduce_1x1" % i, p + ".conv_pwl", p + ".bn3")
            i += 1
    cb("conv_1x1", "conv_head", "bn2")
    return out


def test_mobilenetv2_restatement_matches_transformers_mobilenetv2():
    from transformers import MobileNetV2Config, MobileNetV2Model
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(3)
    hf = MobileNetV2Model(MobileNetV2Config(tf_padding=False, layer_norm_eps=1e-5, hidden_act="relu6")).eval()
    _randomise_bn(hf, g)
    x = torch.randn(2, 3, 96, 96, generator=g)
    with torch.no_grad():
        out = hf(x)
        ref_map, ref_feat = out.last_hidden_state, out.pooler_output.flatten(1)
    gc = guide_config("mobilenetv2")
    guide = O.GuideOracleMBV2(EngineConfig(guide=gc), _hf_mobilenetv2_to_timm(hf.state_dict(), gc.mb_repeats))
    with torch.no_grad():
        fmap = guide.forward_features(x)
        feat = guide.encode_image(x)
    assert fmap.shape == ref_map.shape == (2, 1280, 3, 3)
    assert torch.allclose(fmap, ref_map, rtol=1e-4, atol=1e-5)
    assert torch.allclose(feat, ref_feat, rtol=1e-4, atol=1e-5)
