"""Guide-model plugin surface of the reference (model_utils.py:43-104): `create_model(...)` returns an object with
`encode_image(x[B,3,S,S]) -> [B,D]` (= forward_features -> global average pool -> flatten, model_utils.py:29-41),
`forward(x) -> logits`, `.eval() / .float() / .to() / .cuda()`, and loads `{'state_dict': ...}` checkpoints with an optional
`module.` prefix (model_utils.py:89-101). The arithmetic runs in the engine's guide program (BN folded, exact fp32 on v_mfma_f32_32x32x2_f32, guide_f32.hip).
Built: the three timm Bottleneck networks of the reference -- resnet50 (model_utils.py:47-55), resnext50 = resnext50_32x4d (:56-63,
grouped 3x3 convolutions) and wideresnet50 = wide_resnet50_2 (:72-79); the engine reads widths and groups from the weight shapes --
open_clip_vit_b32 (:80-87, the reference's default --arch): encode_image = the open_clip image tower (bf16 program, `visual.*` keys);
and mobilenetv2 = timm mobilenetv2_100 (:64-71; depthwise convolutions as grouped fp32 convolutions, ReLU6)."""
import torch

from .weights import load_guide_checkpoint, synthetic_guide

from .config import GUIDE_ARCHS

SUPPORTED = tuple(GUIDE_ARCHS)


class GuideModel:
    def __init__(self, name, state_dict, num_classes):
        self.name, self._sd, self.num_classes = name, state_dict, num_classes
        self._engine = None

    # torch.nn.Module-like surface the reference touches
    def state_dict(self):
        return self._sd

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._sd if k not in sd and not k.endswith("num_batches_tracked")]
        if strict and missing:
            raise RuntimeError("missing keys: %s" % missing[:5])
        self._sd = {k: v.float() for k, v in sd.items()}

    def eval(self):
        return self          # BatchNorm always uses running statistics

    def float(self):
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def bind(self, engine):
        """Attach to an Engine whose guide program was built from this state dict."""
        self._engine = engine
        return self

    def encode_image(self, x, pooling="avg"):
        if pooling not in ("avg", "max"):
            raise ValueError("Unsupported pooling type. Please use 'avg' or 'max'.")      # model_utils.py:36-37
        if self._engine is None:
            raise RuntimeError("GuideModel is not bound to an Engine (there is no CPU fallback): call .bind(engine)")
        B = self._engine.B
        outs = []
        for i in range(0, x.shape[0], B):
            xb = x[i:i + B]
            n = xb.shape[0]
            if n < B:
                xb = torch.cat([xb, xb[-1:].expand(B - n, -1, -1, -1)])
            outs.append(self._engine.guide_encode(xb, pooling)[:n].clone())
        return torch.cat(outs)

    def forward(self, x):
        f = self.encode_image(x)                  # CLIP: wrap_clip_forward's `fc(encode_image(x))` (model_utils.py:14-26)
        head = "classifier" if "classifier.weight" in self._sd else "fc"          # timm mobilenetv2: model.classifier (model_utils.py:69)
        w, b = self._sd[head + ".weight"].to(f.device), self._sd[head + ".bias"].to(f.device)
        return torch.nn.functional.linear(f, w, b)   # classifier head: not on the expansion hot path

    __call__ = forward


def create_model(model_name, num_classes=1000, pretrained=False, class_names=None, cache_dir=None, dataset_name=None,
                 weight_path=None, cfg=None):
    print("=> creating model '{}'".format(model_name))
    if model_name not in SUPPORTED:
        raise NotImplementedError("guide arch %r is not built (built: %s; SURVEY.md section 8f-4)" % (model_name, ", ".join(SUPPORTED)))
    from .config import guide_config, sd15_config
    if cfg is None:
        cfg = sd15_config()
        cfg.guide = guide_config(model_name)
    sd = synthetic_guide(cfg, seed=0, num_classes=num_classes)   # shapes of the timm network + fc(num_classes)
    model = GuideModel(model_name, sd, num_classes)
    if pretrained:
        model.load_state_dict(torch.load("save/%s_imagenet1k.pth" % model_name, map_location="cpu"), strict=False)
    if weight_path is not None and weight_path != "None":
        model.load_state_dict(load_guide_checkpoint(weight_path), strict=True)
        print("Load pretrained weights from : %s" % weight_path)
    return model
