"""Generate the golden fixtures under tests/golden/ by running the REAL reference code where it is importable in the build
container: src/models/emasc.py EMASC, src/utils/data_utils.py mask_features, and the installed
transformers CLIPEncoderLayer (the class src/models/inversion_adapter.py:2,9 wraps; v5 call convention `layer(x, None)`).

Run here (needs /root/reference):  python -m oracle.make_golden
The GPU box has no /root/reference: tests only read the committed .safetensors files.
"""
import os
import sys

import torch
from safetensors.torch import save_file

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    sys.path.insert(0, REF)
    from src.models.emasc import EMASC  # noqa: E402  (reference code, imported not copied)
    from src.utils.data_utils import mask_features  # noqa: E402
    from ladi_vton_amd import configs as C

    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)

    # ---- EMASC + mask_features (tiny channel config, 64x48 base resolution)
    ecfg = C.EMASC_TINY
    sd = C.synth_state_dict(C.emasc_shapes(ecfg), "emasc.")
    m = EMASC(list(ecfg["in_channels"]), list(ecfg["out_channels"]), kernel_size=3, padding=1, stride=1, type="nonlinear").eval()
    m.load_state_dict(sd, strict=True)
    B, H, W = 1, 32, 24
    g = torch.Generator().manual_seed(11)
    sizes = [(H, W), (H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
    feats = [torch.randn((B, c, h, w), generator=g).half().float() for c, (h, w) in zip(ecfg["in_channels"], sizes)]
    mask = (torch.rand((B, 1, H, W), generator=g) > 0.6).float()
    with torch.no_grad():
        outs = m([f.clone() for f in feats])
        masked = mask_features([o.clone() for o in outs], mask.clone())
    blob = {}
    for i in range(5):
        blob["feat%d" % i] = feats[i].contiguous()
        blob["emasc%d" % i] = outs[i].contiguous()
        blob["masked%d" % i] = masked[i].contiguous()
    blob["mask"] = mask
    save_file(blob, os.path.join(OUT, "emasc_tiny.safetensors"))

    # ---- CLIPEncoderLayer (adapter's encoder layer), tiny vision config
    from transformers import CLIPVisionConfig
    from transformers.models.clip.modeling_clip import CLIPEncoderLayer
    acfg = C.ADAPTER_TINY
    vc = CLIPVisionConfig(hidden_size=acfg["hidden"], intermediate_size=acfg["mlp_dim"], num_attention_heads=acfg["heads"],
                          layer_norm_eps=acfg["layer_norm_eps"], hidden_act="gelu", attention_dropout=0.0)
    try:
        vc._attn_implementation = "eager"
    except Exception:
        pass
    layer = CLIPEncoderLayer(vc).eval()
    asd = C.synth_state_dict(C.adapter_shapes(acfg), "adapter.")
    lsd = {k[len("encoder_layers.0."):]: v for k, v in asd.items() if k.startswith("encoder_layers.0.")}
    layer.load_state_dict(lsd, strict=True)
    x = torch.randn((2, 17, acfg["hidden"]), generator=g).half().float()
    with torch.no_grad():
        y = layer(x, None)
        if isinstance(y, (tuple, list)):
            y = y[0]
    save_file({"x": x, "y": y.contiguous()}, os.path.join(OUT, "clip_encoder_layer_tiny.safetensors"))
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
