"""Drop-in for the reference module path afigan/modeling/meta_arch/rcnn_extractor.py (the stage-2 detector that returns its FPN features)."""
from afigan_amd.rcnn_extractor import GeneralizedRCNN_AFExtractor  # noqa: F401
