"""Weight hand-off between the three training stages (SURVEY.md sections 3.4 and 8(f) row 2).

Host-side restatement of the key remapping of ``AF_DetectionCheckpointer`` (afigan/engine/checkpoint.py:64-125 and the two
``align_and_update_state_dicts_*`` methods, :127-258) as plain functions over state dicts, so stage-1 interpolator weights
(keys ``Generators.0...``) load into an AFI backbone of this package (keys ``backbone.srf_module.Generators.0...``) and a
stage-2 AF-extractor checkpoint hands only its ``srf_module`` tensors to the stage-3 detector.  No GPU work here.
"""
import logging
from typing import Dict, Tuple

import torch

logger = logging.getLogger(__name__)


def convert_afi_names(weights: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Dict[str, str]]:
    """Stage 1 -> stage 2 (checkpoint.py:78-111): every ``Generators`` in a key becomes ``backbone.srf_module.Generators``.
    Returns (renamed weights, renamed key -> original key)."""
    new_weights, new_to_old = {}, {}
    for orig in sorted(weights):
        renamed = orig.replace("Generators", "backbone.srf_module.Generators")
        if renamed in new_weights:
            raise ValueError(f"renaming {orig!r} collides with another key ({renamed!r})")
        new_weights[renamed] = weights[orig]
        new_to_old[renamed] = orig
    return new_weights, new_to_old


def remain_only_afi_names(weights: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Dict[str, str]]:
    """Stage 2 -> stage 3 (checkpoint.py:113-125): keep only the tensors whose key mentions ``srf_module``."""
    kept = {k: weights[k] for k in sorted(weights) if "srf_module" in k}
    return kept, {k: k for k in kept}


def align_and_update(model_state_dict: Dict[str, torch.Tensor], ckpt_state_dict: Dict[str, torch.Tensor]) -> Dict[str, str]:
    """The suffix matching shared by both loaders (checkpoint.py:127-197 / :199-258): a model key takes the checkpoint key
    that equals it or is its longest complete ``.``-suffix; a shape mismatch skips the pair with a warning; one checkpoint key
    feeding two model keys is an error.  Updates ``model_state_dict`` in place (clones) and returns {ckpt key: model key}."""
    model_keys, ckpt_keys = sorted(model_state_dict), sorted(ckpt_state_dict)
    matched: Dict[str, str] = {}
    for mk in model_keys:
        best, best_len = None, 0
        for ck in ckpt_keys:
            if (mk == ck or mk.endswith("." + ck)) and len(ck) > best_len:      # first longest match, as torch.max returns
                best, best_len = ck, len(ck)
        if best is None:
            continue
        value = ckpt_state_dict[best]
        if tuple(model_state_dict[mk].shape) != tuple(value.shape):
            logger.warning("Shape of %s in checkpoint is %s, while shape of %s in model is %s; not loaded.", best, tuple(value.shape),
                           mk, tuple(model_state_dict[mk].shape))
            continue
        model_state_dict[mk] = value.clone()
        if best in matched:
            raise ValueError(f"Cannot match one checkpoint key to multiple keys in the model: {best} -> {mk} and {matched[best]}")
        matched[best] = mk
    return matched


def _model_dict(checkpoint):
    return checkpoint["model"] if isinstance(checkpoint, dict) and "model" in checkpoint else checkpoint


def load_af_extractor_weights(model: torch.nn.Module, checkpoint) -> Dict[str, str]:
    """``_load_AFExtractor_weights_file`` (checkpoint.py:64-69) on an already-read checkpoint (``{"model": ...}`` or a bare
    state dict): stage-1 interpolator weights into ``model``'s ``backbone.srf_module``."""
    sd = model.state_dict()
    renamed, _ = convert_afi_names(_model_dict(checkpoint))
    matched = align_and_update(sd, renamed)
    model.load_state_dict(sd, strict=True)
    return matched


def load_target_detector_weights(model: torch.nn.Module, checkpoint) -> Dict[str, str]:
    """``_load_TargetDetector_weights_file`` (checkpoint.py:71-76): only the ``srf_module`` tensors of a stage-2 checkpoint."""
    sd = model.state_dict()
    kept, _ = remain_only_afi_names(_model_dict(checkpoint))
    matched = align_and_update(sd, kept)
    model.load_state_dict(sd, strict=True)
    return matched
