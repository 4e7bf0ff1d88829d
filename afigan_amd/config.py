"""Config keys the AFI path adds to detectron2's defaults (afigan/config/defaults.py:5-94), so the reference yamls load unchanged.

``add_afigan_config(cfg)`` declares them on a yacs / detectron2 CfgNode; without either package ``get_cfg()`` returns a small attribute
tree (``Node``) with the same keys and defaults plus the handful of detectron2 defaults this package's builders read, and
``Node.merge_from_file`` / ``merge_from_dict`` follow yacs' rule that a yaml may only SET keys that are declared.

One key of the reference's own yamls is declared nowhere in its defaults.py: ``MODEL.SRF_FREEZE``
(configs/inference/AFI-GAN_cascade_rcnn_swint_BiFPN_ST.yaml -- the older spelling of ``MODEL.AFI_FREEZE``; with stock yacs that file
cannot merge).  It is declared here as well, so the file loads, and the builders read ``AFI_FREEZE or SRF_FREEZE``."""
import copy
import os

# section -> key -> default, exactly as declared by the reference; nested dicts are CfgNodes
AFIGAN_KEYS = {
    "MODEL": {
        "GUIDE_ARCHITECTURE": "",                          # defaults.py:5
        "GUIDE_WEIGHTS": "",                               # :7
        "AFI_GEN_WEIGHTS": "",                             # :8
        "AFI_DIS_WEIGHTS": "",                             # :9
        "AF_EXTRACTOR_WEIGHTS": "",                        # :10
        "AFI_FREEZE": False,                               # :11
        "SRF_FREEZE": False,                               # (undeclared in the reference, set by its BiFPN yaml: see the module docstring)
        "GUIDE_BACKBONE": {"NAME": "build_resnet_fpn_backbone", "FREEZE_AT": 2},                    # :16-22
        "RESNETS": {"RADIX": 1, "BOTTLENECK_WIDTH": 64, "DEEP_STEM": False, "AVD": False, "AVG_DOWN": False},   # :32-41 (ResNeSt)
        "BIFPN": {"IN_FEATURES": [], "OUT_CHANNELS": 256, "FPN_REPEAT": 3, "NORM": "SyncBN", "FUSE_TYPE": "sum"},   # :47-59
        "SWINT": {"EMBED_DIM": 96, "OUT_FEATURES": ["stage2", "stage3", "stage4", "stage5"], "DEPTHS": [2, 2, 6, 2],
                  "NUM_HEADS": [3, 6, 12, 24], "WINDOW_SIZE": 7, "MLP_RATIO": 4, "DROP_PATH_RATE": 0.2, "APE": False},   # :65-73
    },
    "SOLVER": {
        "OPTIMIZER": "SGD",                                # :81
        "AMP": {"ENABLED": False},                         # :82 (never read by the reference's trainers: the path is fp32-only)
        "CLIP_GRADIENTS": {"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0},       # :84-94
    },
}
# kept for callers of the round-1/2 names
AFIGAN_MODEL_KEYS = {k: v for k, v in AFIGAN_KEYS["MODEL"].items() if not isinstance(v, dict)}
AFIGAN_GUIDE_BACKBONE_KEYS = dict(AFIGAN_KEYS["MODEL"]["GUIDE_BACKBONE"])


class Node:
    """Attribute tree standing in for a yacs CfgNode when neither yacs nor detectron2 is installed."""

    def __init__(self, d=None):
        for k, v in (d or {}).items():
            setattr(self, k, Node(v) if isinstance(v, dict) else copy.deepcopy(v))

    def __contains__(self, k):
        return hasattr(self, k)

    def keys(self):
        return [k for k in vars(self)]

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Node) else v) for k, v in vars(self).items()}

    def merge_from_dict(self, d, _path=""):
        """yacs semantics: every key must already be declared (KeyError otherwise); tuples in the yaml meet list defaults and vice versa."""
        for k, v in d.items():
            if k == "_BASE_":
                continue
            if not hasattr(self, k):
                raise KeyError(f"Non-existent config key: {_path}{k}")
            cur = getattr(self, k)
            if isinstance(cur, Node):
                if not isinstance(v, dict):
                    raise ValueError(f"{_path}{k} is a section, got {type(v).__name__}")
                cur.merge_from_dict(v, _path + k + ".")
            else:
                if isinstance(cur, (list, tuple)) and isinstance(v, (list, tuple)):
                    v = type(cur)(v)
                setattr(self, k, v)
        return self

    def merge_from_file(self, path):
        """yacs semantics: ``yaml.safe_load`` (a config file cannot run code), then the literal parse yacs applies to string values --
        the reference yamls write tuples as "(a, b)" strings -- and ``!!python/tuple`` nodes accepted as plain tuples."""
        import ast
        import yaml

        class _Loader(yaml.SafeLoader):
            pass
        _Loader.add_constructor("tag:yaml.org,2002:python/tuple", lambda ld, node: tuple(ld.construct_sequence(node, deep=True)))

        def literal(v):
            if isinstance(v, dict):
                return {k: literal(x) for k, x in v.items()}
            if isinstance(v, list):
                return [literal(x) for x in v]
            if isinstance(v, str):
                try:                                       # yacs _decode_cfg_value: "(800, 1333)" -> (800, 1333), "1e-3" -> 0.001; other strings stay
                    return ast.literal_eval(v)
                except (ValueError, SyntaxError):
                    return v
            return v
        with open(path) as f:
            d = yaml.load(f, Loader=_Loader) or {}
        base = d.pop("_BASE_", None) if isinstance(d, dict) else None
        d = literal(d)
        if base is not None:
            d["_BASE_"] = base
        base = d.get("_BASE_")
        if base:
            self.merge_from_file(os.path.join(os.path.dirname(path), base))
        return self.merge_from_dict(d)


def _declare(node, decl, make):
    for k, v in decl.items():
        if isinstance(v, dict):
            if not hasattr(node, k):
                setattr(node, k, make())
            _declare(getattr(node, k), v, make)
        elif not hasattr(node, k):
            setattr(node, k, copy.deepcopy(v))


def add_afigan_config(cfg):
    """Declare every key of afigan/config/defaults.py on `cfg` (a yacs CfgNode, detectron2's included, or ``Node``); existing keys keep
    their values.  Returns cfg."""
    def make():
        try:
            return type(cfg)()                             # a CfgNode of the same flavour
        except Exception:
            return Node()
    for section, decl in AFIGAN_KEYS.items():
        if not hasattr(cfg, section):
            setattr(cfg, section, make())
        _declare(getattr(cfg, section), decl, make)
    return cfg


def afi_freeze(cfg) -> bool:
    """MODEL.AFI_FREEZE (fpn_sr.py:67-69), or its older yaml spelling MODEL.SRF_FREEZE."""
    m = getattr(cfg, "MODEL", None)
    return bool(getattr(m, "AFI_FREEZE", False) or getattr(m, "SRF_FREEZE", False))


def get_cfg():
    """detectron2's defaults + the AFI keys when detectron2 is importable (config/config.py:3), else a minimal attribute tree: the
    detectron2 defaults this package's builders and guide network read, plus everything above."""
    try:
        from detectron2.config import get_cfg as d2_get_cfg
        return add_afigan_config(d2_get_cfg())
    except Exception:
        cfg = Node({"MODEL": {"DEVICE": "cuda", "PIXEL_MEAN": [103.530, 116.280, 123.675], "PIXEL_STD": [1.0, 1.0, 1.0],
                              "FPN": {"IN_FEATURES": ["res2", "res3", "res4", "res5"], "OUT_CHANNELS": 256, "NORM": "", "FUSE_TYPE": "sum"},
                              "BACKBONE": {"NAME": "build_resnet_fpn_sr_backbone", "FREEZE_AT": 2}},
                    "INPUT": {"FORMAT": "BGR"}, "SOLVER": {}})
        return add_afigan_config(cfg)
