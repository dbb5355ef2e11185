"""Model configs used by the tests (values of configs/model/ehem.yaml and oct_attn.yaml of the reference)."""


class Cfg(dict):
    def __getattr__(self, k):
        v = self[k]
        return Cfg(v) if isinstance(v, dict) else v


def ehem_cfg():
    return Cfg(model=dict(class_name="EHEM", context_size=8192, token_num=255, level_k=4, max_level=19),
               data=dict(extra_pos=False), train=dict(type="kitti", dropout=0.0))


def octattn_cfg():
    return Cfg(model=dict(class_name="OctAttention", max_octree_level=12, context_size=1024, token_num=255,
                          layer_num=3, head_num=4, abs_pos_embed_dim=12, occ_embed_dim=128, level_embed_dim=6,
                          octant_embed_dim=4, hidden_dimension=300, pos_max_len=5000, level_k=4, pos_embed=True),
               data=dict(extra_pos=False), train=dict(type="kitti", dropout=0.0))
