"""Registry surface of the AFI path (SURVEY.md 8b "Registry surface to keep").

The reference registers five backbone builders with detectron2's ``BACKBONE_REGISTRY`` (fpn_sr.py:201,224; pafpn_sr.py:237,260;
bifpn_sr.py:791) and keeps a registry of its own for the frozen guide network (meta_arch/build.py:5, rcnn_only.py:17).  Here the same
names are registered with detectron2's registry when detectron2 is importable and with a local one of the same interface otherwise,
so ``BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)`` resolves either way.  The bottom-up networks (ResNet, ResNeSt,
Swin) are NOT part of this package: a builder looks its bottom-up builder up when it is CALLED -- detectron2's / the reference's if
importable, or one installed with ``set_bottom_up_builder`` -- and fails with a clear message when there is none."""
from typing import Callable, Dict

from . import _lib


class Registry:
    """The subset of fvcore.common.registry.Registry the reference uses: ``@REG.register()`` on functions / classes, ``REG.get(name)``."""

    def __init__(self, name: str):
        self._name = name
        self._obj_map: Dict[str, object] = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(fn_or_class):
                self._do_register(fn_or_class.__name__, fn_or_class)
                return fn_or_class
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map


def _detectron2_backbone_registry():
    try:
        from detectron2.modeling import BACKBONE_REGISTRY as reg
        return reg
    except Exception:
        return None


_D2 = _detectron2_backbone_registry()
BACKBONE_REGISTRY = _D2 if _D2 is not None else Registry("BACKBONE")
USING_DETECTRON2_REGISTRY = _D2 is not None
GUIDE_ARCH_REGISTRY = Registry("GUIDE_ARCH")          # meta_arch/build.py:5

_BOTTOM_UP: Dict[str, Callable] = {}


def set_bottom_up_builder(kind: str, fn: Callable) -> None:
    """Install ``fn(cfg, input_shape) -> bottom-up backbone`` for kind in {"resnet", "resnest", "swint"} (overrides the lookup below)."""
    _BOTTOM_UP[kind] = fn


def bottom_up_builder(kind: str) -> Callable:
    if kind in _BOTTOM_UP:
        return _BOTTOM_UP[kind]
    try:
        if kind == "resnet":
            from detectron2.modeling.backbone.resnet import build_resnet_backbone as fn      # fpn_sr.py:13
        elif kind == "resnest":
            from afigan.modeling.backbone.resnest import build_resnest_backbone as fn        # fpn_sr.py:14 (vendored detectron2-ResNeSt)
        elif kind == "swint":
            from afigan.modeling.backbone.swin_transformer import build_swint_backbone as fn  # bifpn_sr.py:17 (vendored SwinT_detectron2)
        else:
            raise KeyError(kind)
        return fn
    except Exception as e:
        raise _lib.AfiError(f"no '{kind}' bottom-up builder available ({type(e).__name__}: {e}); the bottom-up networks are outside this "
                            f"package -- install detectron2 / the reference's backbone, or call registry.set_bottom_up_builder('{kind}', fn)")


def build_guide_model(cfg):
    """meta_arch/build.py:15-21: the frozen guide network named by cfg.MODEL.GUIDE_ARCHITECTURE."""
    return GUIDE_ARCH_REGISTRY.get(cfg.MODEL.GUIDE_ARCHITECTURE)(cfg)
