"""EHEM context model on MI355X - same constructor / forward / state_dict as the reference, HIP kernels inside.

Drop-in for models/ehem.py (class EHEM), models/dgcnn.py (GeoFeatGenerator) and models/swin_transformer.py
(SwinEncoder with the EHEM-specific 1-D windows): `EHEM(cfg).forward(data, pos, enc=True)` takes the
reference's tensors (data int64 [B,c,4,3] = (level, octant, occ) x (ggp, gp, p, self); pos float32 [B,3,c])
and returns (logits_even [B,ceil(c/2),255], logits_odd [B,floor(c/2),255]).  Parameter names are identical
to the reference, so published checkpoints load with `load_state_dict`.

What runs where:
  * kNN (distance + top-20), edge-conv gather/max, window attention, embedding gather: hand-written HIP
    through the C ABI (scp_amd/native.py)
  * dense layers: fp32 GEMMs (scp_amd.ops.linear)
The module refuses to run on a CPU tensor: there is no fallback path.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import native
from ..ops import linear, layer_norm, gelu_linear, leaky_mlp3, derived

WINDOW = 512
SHIFT = 256
HEADS = 4
SELF_DEPTHS = (4, 4, 4, 4, 2)
CROSS_DEPTHS = (2, 2, 1, 1)


def _mlp(dims):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(nn.LeakyReLU())
    return nn.Sequential(*layers)


# ---------------------------------------------------------------------------------------- parameter containers
class GeoFeatGenerator(nn.Module):
    """Parameters of models/dgcnn.py:74-119 (keys conv{1,2,3}.{0,1}, occ_enc, level_enc, octant_enc, mlp2, ...)."""

    def __init__(self, k=20, max_level=17):
        super().__init__()
        self.k = k

        def conv(cin, cout):
            return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout),
                                 nn.LeakyReLU(negative_slope=0.2))

        self.conv1 = conv(6, 64)
        self.conv2 = conv((64 + 80) * 2, 128)
        self.conv3 = conv((128 + 64) * 2, 256)
        self.occ_enc = nn.Embedding(256, 16)
        self.level_enc = nn.Embedding(max_level, 4)
        self.octant_enc = nn.Embedding(9, 4)
        self.mlp2 = _mlp([80, 80, 64, 64])
        self.mlp3 = _mlp([64, 128, 128, 128])
        self.edge_mlp1 = _mlp([448, 256, 256, 256])
        self.edge_mlp2 = _mlp([512, 256, 256, 128])

    def embed_occ(self, occ):
        return self.occ_enc(occ)


class _SelfAttention(nn.Module):
    def __init__(self, dim, heads, window):
        super().__init__()
        self.relative_position_bias_table = nn.Parameter(torch.zeros(2 * window - 1, heads))
        ar = torch.arange(window)
        self.register_buffer("relative_position_index", ar[:, None] - ar[None, :] + window - 1)
        self.query = nn.Linear(dim, dim)
        self.key = nn.Linear(dim, dim)
        self.value = nn.Linear(dim, dim)


class _Dense(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.dense = nn.Linear(i, o)


class _Attention(nn.Module):
    def __init__(self, dim, heads, window):
        super().__init__()
        self.self = _SelfAttention(dim, heads, window)
        self.output = _Dense(dim, dim)


class SwinLayer(nn.Module):
    def __init__(self, dim=256, heads=HEADS, window=WINDOW):
        super().__init__()
        self.layernorm_before = nn.LayerNorm(dim, eps=1e-5)
        self.attention = _Attention(dim, heads, window)
        self.layernorm_after = nn.LayerNorm(dim, eps=1e-5)
        self.intermediate = _Dense(dim, 4 * dim)
        self.output = _Dense(4 * dim, dim)


class SwinPatchMerging(nn.Module):
    def __init__(self, dim=256):
        super().__init__()
        self.reduction = nn.Linear(2 * dim, dim, bias=False)
        self.norm = nn.LayerNorm(2 * dim)


class SwinStage(nn.Module):
    def __init__(self, depth, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([SwinLayer() for _ in range(depth)])
        if downsample:
            self.downsample = SwinPatchMerging()


class SwinEncoder(nn.Module):
    def __init__(self, depths):
        super().__init__()
        self.depths = tuple(depths)
        self.layers = nn.ModuleList([SwinStage(d, i < len(depths) - 1) for i, d in enumerate(depths)])


# ---------------------------------------------------------------------------------------- forward pieces
def _edge_fold(conv):
    """The 1x1 convolution on (neighbour - centre, centre) as two products on the points ([W1; W2 - W1]) and the eval-mode
    BatchNorm as one scale / shift pair; derived from the parameters, rebuilt when they change."""
    W, bn = conv[0].weight, conv[1]
    Cout, C2 = W.shape[0], W.shape[1]
    C = C2 // 2

    def build():
        W2d = W.detach().reshape(Cout, C2)
        Wuv = torch.cat((W2d[:, :C], W2d[:, C:] - W2d[:, :C]), 0).contiguous()          # [2C', C]
        scale = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).contiguous()
        shift = (bn.bias.detach() - bn.running_mean * scale).contiguous()
        return Wuv, scale, shift
    return derived(conv, "fold", (W, bn.weight, bn.bias, bn.running_mean, bn.running_var), build)


def qkv_fused(layer, cross):
    """(weight, bias) of the layer's q|k|v projections as one product (k|v for a cross layer, whose q comes from the query stream)."""
    att = layer.attention.self
    mods = (att.key, att.value) if cross else (att.query, att.key, att.value)

    def build():
        return (torch.cat([m.weight for m in mods], 0).detach().contiguous(), torch.cat([m.bias for m in mods], 0).detach().contiguous())
    return derived(layer, "kv" if cross else "qkv", [m.weight for m in mods] + [m.bias for m in mods], build)


def _edge_conv(conv, feat, k, feeds_knn=True):
    """feat [B,n,C] -> [B,n,C'] : kNN in feature space + (split 1x1 conv) + BN + LeakyReLU(0.2) + max over k."""
    Cout = conv[0].weight.shape[0]  # [C', 2C, 1, 1]
    Wuv, scale, shift = _edge_fold(conv)
    idx = native.knn_topk(feat, k)
    # [B,n,2C'] plain fp32 where the features feed the next kNN search; the last convolution's only feed dense layers: bf16x3 like those
    uv = linear(feat, Wuv, None, exact=feeds_knn)
    u = uv[..., :Cout].contiguous()
    v = uv[..., Cout:].contiguous()
    return native.edge_gather_max(u, v, idx, scale, shift)


def _edge_conv_packed(conv, feat, ktab, feeds_knn=True, out=None):
    """packed layout: feat [T,C] (all windows back to back, padded to x512 rows), ktab int32 [T/512,2] -> [T,C'] (into `out` when given: a column
    view of the next search's feature buffer)."""
    Cout = conv[0].weight.shape[0]
    Wuv, scale, shift = _edge_fold(conv)
    feat = feat.contiguous()
    idx = native.knn_topk_packed(feat, ktab)
    uv = linear(feat, Wuv, None, exact=feeds_knn)
    return native.edge_gather_max_rows(uv[:, :Cout], uv[:, Cout:], idx, scale, shift, out=out)


def geo_feat_forward(g, ctx, pos):
    """ctx int64 [B,c,12] (level, octant, occ) x 4 with the self occupancy still present; pos [B,c,3] f32."""
    B, c = ctx.shape[:2]
    occ = ctx[:, :, 2:11:3]
    level = ctx[:, :, 0::3]
    octant = ctx[:, :, 1::3]
    x = torch.cat((F.embedding(occ, g.occ_enc.weight).reshape(B, c, -1),
                   F.embedding(level, g.level_enc.weight).reshape(B, c, -1),
                   F.embedding(octant, g.octant_enc.weight).reshape(B, c, -1)), 2)
    k = min(g.k, c)
    pos1 = _edge_conv(g.conv1, pos.contiguous(), k)
    pos2 = _edge_conv(g.conv2, torch.cat((pos1, x), 2), k)
    x = leaky_mlp3(g.mlp2, x, exact=True)     # feeds the third kNN search: keep plain fp32
    pos3 = _edge_conv(g.conv3, torch.cat((pos2, x), 2), k, feeds_knn=False)
    x = leaky_mlp3(g.mlp3, x)
    ec = leaky_mlp3(g.edge_mlp1, torch.cat((pos1, pos2, pos3), 2))
    ec = leaky_mlp3(g.edge_mlp2, torch.cat((pos3, ec), 2))
    return torch.cat((x, ec), 2)


def _pad_tokens(h, L):
    pad = (WINDOW - L % WINDOW) % WINDOW
    return F.pad(h, (0, 0, 0, pad)) if pad else h


def swin_layer_forward(layer, x, L, shift, query=None):
    """swin_transformer.py:654-706.  x [B,L,256]; query (cross stream) [B,L,256] or None."""
    att = layer.attention.self
    Wqkv, bqkv = qkv_fused(layer, query is not None)
    ln = layer.layernorm_before
    h = _pad_tokens(layer_norm(x, ln), L)        # zero rows AFTER LayerNorm, like the reference
    if query is None:
        qkv = linear(h, Wqkv, bqkv)               # [B,Lp,768]
        q, k, v = qkv[..., :256], qkv[..., 256:512], qkv[..., 512:]
    else:
        hq = _pad_tokens(layer_norm(query, ln), L)
        q = linear(hq, att.query.weight, att.query.bias)
        kv = linear(h, Wqkv, bqkv)
        k, v = kv[..., :256], kv[..., 256:]
    o = native.swin_attention(q, k, v, att.relative_position_bias_table, shift)
    if o.shape[1] != L:
        o = o[:, :L].contiguous()
    x = linear(o, layer.attention.output.dense.weight, layer.attention.output.dense.bias, residual=x)
    y = gelu_linear(layer_norm(x, layer.layernorm_after), layer.intermediate.dense.weight, layer.intermediate.dense.bias)
    return linear(y, layer.output.dense.weight, layer.output.dense.bias, residual=x)


def patch_merge_forward(m, x, L):
    if L % 2:
        x = F.pad(x, (0, 0, 0, 1))
    B = x.shape[0]
    x = x.reshape(B, (L + 1) // 2, 512)    # cat(even, odd) of consecutive tokens == reshape
    return linear(layer_norm(x, m.norm), m.reduction.weight, None)


def swin_encoder_forward(enc, x, L, query=None):
    hs = [x]
    for s, stage in enumerate(enc.layers):
        for b, blk in enumerate(stage.blocks):
            x = swin_layer_forward(blk, x, L, SHIFT if b % 2 else 0, query)
        hs.append(x)
        if s < len(enc.layers) - 1:
            x = patch_merge_forward(stage.downsample, x, L)
            if query is not None:
                query = patch_merge_forward(stage.downsample, query, L)
            L = (L + 1) // 2
    return hs


def concat_states(hs):
    """ehem.py:75-86 == gather of stage s at token index t >> (s-1)."""
    n = hs[1].shape[1]
    t = torch.arange(n, device=hs[1].device)
    return torch.cat([hs[1]] + [hs[s][:, t >> (s - 1)] for s in range(2, len(hs))], 2)


class EHEM(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.geo_feat_generator = GeoFeatGenerator(max_level=cfg.model.max_level)
        self.swin_self_transformer = SwinEncoder(SELF_DEPTHS)
        self.swin_cross_transformer = SwinEncoder(CROSS_DEPTHS)
        self.ancient_mlp = _mlp([1280, 1024, 512, 256])
        self.prob_pred_mlp1 = _mlp([256, 256, 256, 255])
        self.pre_occ_mlp = _mlp([16, 16, 16, 16])
        self.pre_attn_mlp = _mlp([256, 256, 240, 240])
        self.prob_pred_mlp2 = _mlp([1280, 768, 512, 255])
        self.eval()

    # ---- reference-compatible entry points -------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, path, cfg=None, map_location="cpu"):
        m = cls(cfg)
        sd = torch.load(path, map_location=map_location)
        m.load_state_dict(sd["state_dict"] if "state_dict" in sd else sd, strict=True)
        return m

    @torch.no_grad()
    def forward(self, data, pos, enc=True):
        """data int [B,c,4,3]; pos float32 [B,3,c] (reference layout) -> (logits_even, logits_odd)."""
        if not data.is_cuda:
            raise native.ScpError("EHEM runs on the MI355X only (no CPU fallback); move the inputs to cuda")
        B, c = data.shape[:2]
        return self.forward_ctx(data.reshape(B, c, 12).long(), pos.transpose(1, 2).contiguous())

    @torch.no_grad()
    def forward_packed(self, ctx, pos, lengths, plan=None, table=None):
        """All windows in one pass: ctx [T,12], pos [T,3] (windows back to back), lengths = window sizes.
        Returns (even rows [sum ceil(c/2),255], odd rows [sum floor(c/2),255]); see models/packed.py.  With `table` (fp32 view whose
        row 0 is the chunk's first coded row, 16-byte aligned rows) the logits are written in coding order and (None, None) returns."""
        from .packed import PackedPlan, ehem_forward_packed
        if not ctx.is_cuda:
            raise native.ScpError("EHEM runs on the MI355X only (no CPU fallback)")
        if plan is None:
            plan = PackedPlan(lengths, device=ctx.device)
        return ehem_forward_packed(self, ctx, pos, plan, table=table)

    @torch.no_grad()
    def forward_ctx(self, ctx, pos):
        """ctx int64/uint8 [B,c,12]; pos float32 [B,c,3] (token-major layout used by the kernels)."""
        prob1, st = self._phase1(ctx, pos)
        prob2 = self._phase2(st, st["pre_occ"])
        return prob1, prob2

    def _phase1(self, ctx, pos):
        """Everything that does not depend on the window's own occupancies (ehem.py:92-115): -> even-node logits + state."""
        ctx = ctx.long()
        if ctx.shape[1] % 2 == 1:                       # ehem.py:92-99: pad row (0,0,255) x 4, pos 0
            pad = torch.zeros_like(ctx[:, :1])
            pad[:, :, 2::3] = 255
            ctx = torch.cat((ctx, pad), 1)
            pos = torch.cat((pos, torch.zeros_like(pos[:, :1])), 1)
            padded = True
        else:
            padded = False
        B, c = ctx.shape[:2]
        feat = geo_feat_forward(self.geo_feat_generator, ctx, pos)
        hs = swin_encoder_forward(self.swin_self_transformer, feat, c)
        feat_a = leaky_mlp3(self.ancient_mlp, concat_states(hs))
        a1, a2 = feat_a[:, ::2].contiguous(), feat_a[:, 1::2].contiguous()
        prob1 = leaky_mlp3(self.prob_pred_mlp1, a1)
        return prob1, dict(a1=a1, a2=a2, padded=padded, pre_occ=ctx[:, ::2, 11])

    def _phase2(self, st, pre_occ):
        """Odd-node logits given the occupancies of the even nodes (ehem.py:117-127)."""
        a1, a2 = st["a1"], st["a2"]
        occ_feat = leaky_mlp3(self.pre_occ_mlp, F.embedding(pre_occ.long(), self.geo_feat_generator.occ_enc.weight))
        pre = torch.cat((occ_feat, leaky_mlp3(self.pre_attn_mlp, a1)), 2)
        hc = swin_encoder_forward(self.swin_cross_transformer, pre, a2.shape[1], query=a2)
        prob2 = leaky_mlp3(self.prob_pred_mlp2, torch.cat((concat_states(hc), a2), 2))
        if st["padded"]:
            prob2 = prob2[:, :-1]
        return prob2

    @torch.no_grad()
    def decode(self, data, pos, pre_occ=None):
        """Reference decoder API (ehem.py:138-177): decode(data, pos) -> even logits (state is kept on the module);
        decode(data, pos, pre_occ) -> odd logits given the decoded even occupancies [B, ceil(c/2)]."""
        if not data.is_cuda:
            raise native.ScpError("EHEM runs on the MI355X only (no CPU fallback); move the inputs to cuda")
        if pre_occ is None:
            B, c = data.shape[:2]
            prob1, self._dec_state = self._phase1(data.reshape(B, c, 12), pos.transpose(1, 2).contiguous())
            return prob1
        return self._phase2(self._dec_state, pre_occ)
