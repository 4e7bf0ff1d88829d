from . import fpn_sr, pafpn_sr, bifpn_sr  # noqa: F401
