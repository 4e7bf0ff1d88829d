"""Drop-in for the reference module path afigan/modeling/backbone/bifpn_sr.py: re-exports the HIP-backed BiFPN_AFIGAN (inference)."""
from afigan_amd.bifpn_sr import BiFPN_AFIGAN, LastLevelP6P7  # noqa: F401
