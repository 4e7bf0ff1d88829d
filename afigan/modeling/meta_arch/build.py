"""Drop-in for the reference module path afigan/modeling/meta_arch/build.py (GUIDE_ARCH_REGISTRY, build_guide_model)."""
from afigan_amd.registry import GUIDE_ARCH_REGISTRY, build_guide_model  # noqa: F401
