"""Drop-in for the reference module path afigan/modeling/meta_arch/rcnn_only.py (the frozen guide network)."""
from afigan_amd.rcnn_only import RCNN_FPN_only  # noqa: F401
