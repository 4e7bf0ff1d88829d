"""Import-path compatibility with the reference tree: ``afigan.modeling.feat_interpol.{generator_rdb,
feature_patch_discriminator}`` resolve to the MI355X-native modules of ``afigan_amd/``."""
