"""Import-path shim of the reference's afigan/config (config.py:3 get_cfg, defaults.py:5-22 keys)."""
from afigan_amd.config import add_afigan_config, get_cfg  # noqa: F401
