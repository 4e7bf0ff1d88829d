"""Drop-in for the reference module path afigan/modeling/backbone/pafpn_sr.py: re-exports the HIP-backed PAFPN_AFIGAN."""
from afigan_amd.pafpn_sr import PAFPN_AFIGAN, LastLevelMaxPool  # noqa: F401
