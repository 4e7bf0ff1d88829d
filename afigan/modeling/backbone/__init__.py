from . import fpn_sr, pafpn_sr  # noqa: F401
