"""Config keys the AFI path adds to detectron2's defaults (afigan/config/defaults.py:5-22), so the reference yamls load unchanged.

``add_afigan_config(cfg)`` declares them on a yacs / detectron2 CfgNode; without either package ``get_cfg()`` returns a plain attribute
tree with the same keys and defaults (enough for the builders and the guide network of this package)."""
from types import SimpleNamespace

# key -> default, exactly as declared by the reference (defaults.py:5-22)
AFIGAN_MODEL_KEYS = {
    "GUIDE_ARCHITECTURE": "",
    "GUIDE_WEIGHTS": "",
    "AFI_GEN_WEIGHTS": "",
    "AFI_DIS_WEIGHTS": "",
    "AF_EXTRACTOR_WEIGHTS": "",
    "AFI_FREEZE": False,
}
AFIGAN_GUIDE_BACKBONE_KEYS = {
    "NAME": "build_resnet_fpn_backbone",
    "FREEZE_AT": 2,
}


def add_afigan_config(cfg):
    """Declare MODEL.{GUIDE_ARCHITECTURE, GUIDE_WEIGHTS, AFI_GEN_WEIGHTS, AFI_DIS_WEIGHTS, AF_EXTRACTOR_WEIGHTS, AFI_FREEZE} and
    MODEL.GUIDE_BACKBONE.{NAME, FREEZE_AT} on `cfg` (a yacs CfgNode, detectron2's included, or the stand-in below); returns cfg."""
    model = cfg.MODEL
    for k, v in AFIGAN_MODEL_KEYS.items():
        if not hasattr(model, k):
            setattr(model, k, v)
    if not hasattr(model, "GUIDE_BACKBONE"):
        try:
            node = type(cfg)()                      # a CfgNode of the same flavour
        except Exception:
            node = SimpleNamespace()
        setattr(model, "GUIDE_BACKBONE", node)
    for k, v in AFIGAN_GUIDE_BACKBONE_KEYS.items():
        if not hasattr(model.GUIDE_BACKBONE, k):
            setattr(model.GUIDE_BACKBONE, k, v)
    return cfg


def get_cfg():
    """detectron2's defaults + the AFI keys when detectron2 is importable (config/config.py:3), else a minimal attribute tree."""
    try:
        from detectron2.config import get_cfg as d2_get_cfg
        return add_afigan_config(d2_get_cfg())
    except Exception:
        cfg = SimpleNamespace(MODEL=SimpleNamespace(DEVICE="cuda", PIXEL_MEAN=[103.530, 116.280, 123.675], PIXEL_STD=[1.0, 1.0, 1.0],
                                                    FPN=SimpleNamespace(IN_FEATURES=["res2", "res3", "res4", "res5"], OUT_CHANNELS=256, NORM="", FUSE_TYPE="sum"),
                                                    BACKBONE=SimpleNamespace(NAME="build_resnet_fpn_sr_backbone")),
                              INPUT=SimpleNamespace(FORMAT="BGR"))
        return add_afigan_config(cfg)
