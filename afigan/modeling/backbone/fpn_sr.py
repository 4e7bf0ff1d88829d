"""Drop-in for the reference module path afigan/modeling/backbone/fpn_sr.py: re-exports the HIP-backed FPN_AFIGAN."""
from afigan_amd.fpn_sr import FPN_AFIGAN, LastLevelMaxPool  # noqa: F401
