"""Import shims that let the reference (/root/reference) run in THIS container only.

Used by make_golden.py (fixture generation) and by nothing that travels to the GPU box.
The shims follow SURVEY.md §8c: stub the absent optional modules, alias the removed
transformers helpers, and replace pytorch_lightning.LightningModule by nn.Module.
"""
import os
import sys
import types

REF = "/root/reference"


def setup():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present; golden vectors can only be regenerated "
                           "in the build container")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    ply = types.ModuleType("plyfile")
    ply.PlyData = None
    ply.PlyElement = None
    sys.modules.setdefault("plyfile", ply)
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))

    import torch.nn as nn
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    sys.modules.setdefault("pytorch_lightning", pl)

    import transformers.pytorch_utils as pu
    if not hasattr(pu, "find_pruneable_heads_and_indices"):
        pu.find_pruneable_heads_and_indices = None
    try:
        import transformers.utils.backbone_utils as bu
        if not hasattr(bu, "get_aligned_output_features_output_indices"):
            bu.get_aligned_output_features_output_indices = None
    except Exception:
        pass

    # silence tqdm bars inside the reference
    import tqdm

    def _quiet_trange(*a, **k):
        k.pop("desc", None)
        return range(*a)

    tqdm.trange = _quiet_trange


class Cfg(dict):
    """Tiny attribute-dict standing in for the hydra config object."""

    def __getattr__(self, k):
        v = self[k]
        return Cfg(v) if isinstance(v, dict) else v


def ehem_cfg():
    return Cfg(model=dict(class_name="EHEM", context_size=8192, token_num=255, level_k=4, max_level=19),
               data=dict(extra_pos=False), train=dict(type="kitti", dropout=0.0))


def octattn_cfg():
    return Cfg(model=dict(class_name="OctAttention", max_octree_level=12, context_size=1024, token_num=255,
                          layer_num=3, head_num=4, abs_pos_embed_dim=12, occ_embed_dim=128,
                          level_embed_dim=6, octant_embed_dim=4, hidden_dimension=300, pos_max_len=5000,
                          level_k=4, pos_embed=True),
               data=dict(extra_pos=False), train=dict(type="kitti", dropout=0.0))
