"""Drop-in for the reference module path (stage1_trainer.py:31, fpn_sr.py:16): re-exports the HIP-backed classes."""
from afigan_amd.generator_rdb import Generator, ResidualDenseBlock, ResidualInResidual  # noqa: F401
