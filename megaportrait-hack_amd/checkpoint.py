"""Checkpoint utilities for the hot path: on-disk compatibility with the reference.

The reference writes two layouts: a raw `Gbase.state_dict()` (`train.py:429`, read back by `inference.py:59-60` with
`strict=False`) and a wrapped training checkpoint `{epoch, model_G_state_dict, model_D_state_dict,
optimizer_G_state_dict, optimizer_D_state_dict}` (`train.py:348-355`, read by `train.py:372-385`).  Keys of the hot
slice inside a Gbase state-dict: `warp_generator_s2c.*`, `warp_generator_c2d.*`, `G3d.*`, and (row f1)
`appearanceEncoder.resblock3D_*`.  A reference model built on a GPU host does not register
`adaptive_matrix_gamma/beta` (`model.py:934-935`: `nn.Parameter(...).to(device)` returns a plain tensor), so its
checkpoints lack those four keys; they are reported, not fatal.

Nothing here touches the GPU kernels: it is plain state-dict plumbing (torch.save / torch.load, CPU tensors).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch

WRAPPED_KEYS = ("model_G_state_dict", "model_D_state_dict", "optimizer_G_state_dict", "optimizer_D_state_dict")
HOT_PREFIXES = ("warp_generator_s2c.", "warp_generator_c2d.", "G3d.")
EAPP_TAIL_PREFIX = "appearanceEncoder."


def generator_state_dict(obj) -> Dict[str, torch.Tensor]:
    """Raw Gbase state-dict from either layout the reference writes (or an already-extracted dict)."""
    if not isinstance(obj, dict):
        raise TypeError(f"expected a state-dict or a training checkpoint dict, got {type(obj).__name__}")
    if "model_G_state_dict" in obj:
        return obj["model_G_state_dict"]
    if obj and all(isinstance(v, torch.Tensor) for v in obj.values()):
        return obj
    raise ValueError("neither a raw state-dict nor a train.py checkpoint (no 'model_G_state_dict' key)")


def strip_prefix(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def load_hot_path(module: torch.nn.Module, checkpoint, prefix: str = "", strict: bool = True,
                  map_location="cpu") -> Tuple[list, list, list]:
    """Loads the hot-path parameters of `module` (GbaseHotSlice, Eapp3DTail, a G3d, a warp generator ...) from a
    reference checkpoint: a path or an already loaded object, raw or wrapped.

    prefix: where `module` sits inside Gbase ('' for GbaseHotSlice; 'G3d.' for a bare G3d; 'appearanceEncoder.' for
    Eapp3DTail).  Keys of other sub-networks (Eapp trunk, Emtn, G2d, ...) are ignored.
    Returns (missing, tolerated_missing, unexpected): `tolerated_missing` are the adaptive_matrix_* keys a GPU-built
    reference checkpoint does not contain (kept at their current values); with strict=True any other missing key
    or a shape mismatch raises."""
    obj = torch.load(checkpoint, map_location=map_location) if isinstance(checkpoint, (str, os.PathLike)) else checkpoint
    sd = generator_state_dict(obj)
    if prefix:
        sd = strip_prefix(sd, prefix)
    own = module.state_dict()
    usable = {k: v for k, v in sd.items() if k in own}
    bad_shape = [k for k, v in usable.items() if tuple(v.shape) != tuple(own[k].shape)]
    if bad_shape:
        k = bad_shape[0]
        raise ValueError(f"shape mismatch for {k}: checkpoint {tuple(usable[k].shape)} vs module {tuple(own[k].shape)}"
                         + (f" (+{len(bad_shape) - 1} more)" if len(bad_shape) > 1 else ""))
    missing = [k for k in own if k not in usable]
    tolerated = [k for k in missing if "adaptive_matrix_" in k]
    hard = [k for k in missing if k not in tolerated]
    if strict and hard:
        raise KeyError(f"checkpoint lacks {len(hard)} hot-path keys, e.g. {hard[:4]}")
    module.load_state_dict(usable, strict=False)
    unexpected = [k for k in sd if k not in own and (not prefix and k.startswith(HOT_PREFIXES))]
    return hard, tolerated, unexpected


def merge_into_generator_state_dict(gbase_sd: Dict[str, torch.Tensor], module: torch.nn.Module, prefix: str = "",
                                    keep_unregistered_adaptive: bool = False) -> Dict[str, torch.Tensor]:
    """Writes `module`'s parameters back into a full Gbase state-dict (a copy), e.g. after fine-tuning the hot slice
    with the HIP path while the 2D networks stay in PyTorch.  keep_unregistered_adaptive=False drops adaptive_matrix_*
    keys the target dict did not have (a GPU-built reference model would reject them with strict loading)."""
    out = dict(gbase_sd)
    for k, v in module.state_dict().items():
        full = prefix + k
        if "adaptive_matrix_" in k and full not in gbase_sd and not keep_unregistered_adaptive:
            continue
        out[full] = v.detach().cpu().clone()
    return out


def save_training_checkpoint(path, epoch: int, model_G: torch.nn.Module, model_D: Optional[torch.nn.Module] = None,
                             optimizer_G: Optional[torch.optim.Optimizer] = None,
                             optimizer_D: Optional[torch.optim.Optimizer] = None, rank: int = 0) -> bool:
    """The wrapped layout of train.py:348-355 (`checkpoint_epoch{N}.pth`).  Under data parallelism only rank 0 writes
    (the replicas are identical after the gradient all-reduce); written to a temp file and renamed so a crash never
    leaves a truncated checkpoint.  Returns True if this rank wrote the file."""
    if rank != 0:
        return False
    obj = {"epoch": int(epoch), "model_G_state_dict": model_G.state_dict()}
    if model_D is not None:
        obj["model_D_state_dict"] = model_D.state_dict()
    if optimizer_G is not None:
        obj["optimizer_G_state_dict"] = optimizer_G.state_dict()
    if optimizer_D is not None:
        obj["optimizer_D_state_dict"] = optimizer_D.state_dict()
    tmp = f"{path}.tmp.{os.getpid()}"
    torch.save(obj, tmp)
    os.replace(tmp, path)
    return True


def load_training_checkpoint(path, model_G: torch.nn.Module, model_D: Optional[torch.nn.Module] = None,
                             optimizer_G: Optional[torch.optim.Optimizer] = None,
                             optimizer_D: Optional[torch.optim.Optimizer] = None, prefix: str = "", map_location="cpu") -> int:
    """train.py:372-385: restores what the checkpoint has and returns the epoch to resume from (0 if the file does
    not exist, like the reference).  model_G's hot-path keys are loaded with `load_hot_path` semantics."""
    if not os.path.isfile(path):
        return 0
    ck = torch.load(path, map_location=map_location)
    load_hot_path(model_G, ck, prefix=prefix)
    if model_D is not None and "model_D_state_dict" in ck:
        model_D.load_state_dict(ck["model_D_state_dict"])
    if optimizer_G is not None and "optimizer_G_state_dict" in ck:
        optimizer_G.load_state_dict(ck["optimizer_G_state_dict"])
    if optimizer_D is not None and "optimizer_D_state_dict" in ck:
        optimizer_D.load_state_dict(ck["optimizer_D_state_dict"])
    return int(ck.get("epoch", -1)) + 1


def load_gbase(gbase: torch.nn.Module, checkpoint, strict: bool = False, map_location="cpu") -> Tuple[list, list]:
    """inference.py:59-60 for the whole generator: `Gbase.load_state_dict(torch.load(path), strict=False)`, accepting
    the raw layout (train.py:429 `Gbase.pth`) and the wrapped training checkpoint (train.py:348-355) alike.
    Returns (missing, unexpected).  With strict=True anything missing except the adaptive_matrix_* keys of a
    GPU-built reference checkpoint (model.py:934-935) raises; shape mismatches always raise."""
    obj = torch.load(checkpoint, map_location=map_location) if isinstance(checkpoint, (str, os.PathLike)) else checkpoint
    sd = generator_state_dict(obj)
    own = gbase.state_dict()
    usable = {k: v for k, v in sd.items() if k in own}
    bad = [k for k, v in usable.items() if tuple(v.shape) != tuple(own[k].shape)]
    if bad:
        raise ValueError(f"shape mismatch for {bad[0]}: checkpoint {tuple(usable[bad[0]].shape)} vs model {tuple(own[bad[0]].shape)}")
    missing = [k for k in own if k not in usable]
    unexpected = [k for k in sd if k not in own]
    hard = [k for k in missing if "adaptive_matrix_" not in k]
    if strict and hard:
        raise KeyError(f"checkpoint lacks {len(hard)} Gbase keys, e.g. {hard[:4]}")
    gbase.load_state_dict(usable, strict=False)
    return missing, unexpected
