"""Drop-in for the reference module path (stage1_trainer.py:32, stage2_trainer.py:33): re-exports the HIP-backed class."""
from afigan_amd.feature_patch_discriminator import Discriminator  # noqa: F401
