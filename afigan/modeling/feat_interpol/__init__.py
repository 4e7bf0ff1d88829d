from . import feature_patch_discriminator, generator_rdb  # noqa: F401
