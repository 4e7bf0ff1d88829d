"""Import-path shim of the reference's afigan/modeling/meta_arch: the guide-network registry and RCNN_FPN_only of afigan_amd."""
from .build import GUIDE_ARCH_REGISTRY, build_guide_model  # noqa: F401
from .rcnn_only import RCNN_FPN_only  # noqa: F401
from .rcnn_extractor import GeneralizedRCNN_AFExtractor  # noqa: F401
