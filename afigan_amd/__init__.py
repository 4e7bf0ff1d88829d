"""afigan_amd -- MI355X-native implementation of AFI-GAN's adversarial feature-interpolation hot path.

Drop-in surface (same names / signatures / state_dict keys as the reference modules):
  * ``Generator``      <- afigan/modeling/feat_interpol/generator_rdb.py:73
  * ``Discriminator``  <- afigan/modeling/feat_interpol/feature_patch_discriminator.py:16
  * ``Stage1Step``     <- the G+D iteration of afigan/engine/stage1_trainer.py:305-435
  * ``FPN_AFIGAN`` / ``PAFPN_AFIGAN`` <- afigan/modeling/backbone/fpn_sr.py:20, pafpn_sr.py:20 (the callers of G in stages 2/3)
All forward/backward math runs in hand-written HIP kernels for gfx950 behind the C-ABI of
``include/afigan_hip.h`` (libafigan_hip.so).  There is no CPU / eager fallback.
"""
from . import _lib
from ._lib import AfiError, compute_dtype

_lib.load()     # fail loudly at import time when the HIP library is missing

from . import ops  # noqa: E402
from .generator_rdb import Generator  # noqa: E402
from .feature_patch_discriminator import Discriminator  # noqa: E402
from .stage1 import GuidePrefetcher, Stage1Step, warmup_multistep_lr  # noqa: E402
from .fpn_sr import FPN_AFIGAN, LastLevelMaxPool  # noqa: E402
from .pafpn_sr import PAFPN_AFIGAN  # noqa: E402
from .bifpn_sr import BiFPN_AFIGAN, LastLevelP6P7  # noqa: E402
from .stage2 import Stage2Adversarial, Stage2Step, l1_loss_common, nearest_half  # noqa: E402
from .dual_scale import DualScaleMapper, preprocess_images  # noqa: E402
from . import config, registry  # noqa: E402
from .registry import BACKBONE_REGISTRY, GUIDE_ARCH_REGISTRY, build_guide_model  # noqa: E402
from .rcnn_only import RCNN_FPN_only  # noqa: E402
from .rcnn_extractor import GeneralizedRCNN_AFExtractor, META_ARCH_REGISTRY  # noqa: E402
from .config import add_afigan_config, get_cfg  # noqa: E402

__all__ = ["Generator", "Discriminator", "Stage1Step", "GuidePrefetcher", "warmup_multistep_lr", "FPN_AFIGAN", "PAFPN_AFIGAN", "BiFPN_AFIGAN", "LastLevelP6P7", "LastLevelMaxPool", "Stage2Adversarial", "l1_loss_common", "nearest_half", "DualScaleMapper", "preprocess_images", "ops", "AfiError", "compute_dtype", "BACKBONE_REGISTRY", "GUIDE_ARCH_REGISTRY", "build_guide_model", "RCNN_FPN_only", "GeneralizedRCNN_AFExtractor", "META_ARCH_REGISTRY", "Stage2Step", "add_afigan_config", "get_cfg"]
