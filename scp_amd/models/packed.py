"""Packed ("varlen") EHEM forward: ALL windows of a frame in one pass.

The reference calls the model once per window (encode.py:112-118): ~100 forwards per frame, most of them far too small to
fill an MI355X and each paying ~350 kernel launches.  Windows are independent, and every dense layer / LayerNorm / activation
is token-wise, so the windows are concatenated along the token axis; only four operators need to know where a window starts:
kNN (neighbours inside the window), window attention (each window is zero-padded to a multiple of 512 tokens AFTER LayerNorm,
swin_transformer.py:641), patch merging (pairs inside the window, :342-357) and the stage gathers of concat_states
(ehem.py:75-86).  Each window therefore owns a run of rows padded to a multiple of 512 at every Swin stage ("layout"); the
kernels get a small per-512-row table, and the stage transitions are index gathers whose maps are built once per frame on the
host from the list of window lengths.  Numerically this is the same computation as the per-window forward (tests compare them).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .. import native
from ..ops import linear, layer_norm, leaky_mlp3
from .ehem import SHIFT, WINDOW, _edge_conv_packed


def _ceil512(a):
    return (a + WINDOW - 1) // WINDOW * WINDOW


class StageLayout:
    """Rows of one Swin stage: window i owns rows [base[i], base[i] + Lp[i]), the first L[i] of them real."""

    def __init__(self, L):
        self.L = np.asarray(L, np.int64)
        self.Lp = _ceil512(self.L)
        self.base = np.concatenate(([0], np.cumsum(self.Lp)[:-1]))
        self.rows = int(self.Lp.sum())

    def table(self, real=False):
        """int32 [rows/512, 2]: (sequence base row, padded length | real length) for every 512-row chunk."""
        reps = self.Lp // WINDOW
        return np.stack([np.repeat(self.base, reps), np.repeat(self.L if real else self.Lp, reps)], 1).astype(np.int32)

    def valid(self):
        v = np.zeros(self.rows, np.float32)
        for b, l in zip(self.base, self.L):
            v[b:b + l] = 1.0
        return v

    def row_index(self, counts=None):
        """global rows of the first counts[i] (default L[i]) tokens of every window, concatenated."""
        counts = self.L if counts is None else counts
        return np.concatenate([b + np.arange(c) for b, c in zip(self.base, counts)]) if len(self.base) else np.zeros(0, np.int64)


def _merge_maps(cur, nxt):
    """patch merging cur -> nxt: rows (even, odd) of cur feeding every row of nxt; `cur.rows` is the index of an all-zero row."""
    even = np.full(nxt.rows, cur.rows, np.int64)
    odd = np.full(nxt.rows, cur.rows, np.int64)
    for i in range(len(cur.L)):
        t = np.arange(nxt.L[i])
        even[nxt.base[i] + t] = cur.base[i] + 2 * t
        o = 2 * t + 1
        odd[nxt.base[i] + t] = np.where(o < cur.L[i], cur.base[i] + o, cur.rows)
    return even, odd


def _concat_map(l0, ls, s):
    """rows of stage-s layout `ls` that stage-0 token t of each window gathers (t >> s)."""
    m = np.zeros(l0.rows, np.int64)
    for i in range(len(l0.L)):
        t = np.arange(l0.L[i])
        m[l0.base[i] + t] = ls.base[i] + (t >> s)
    return m


class PackedPlan:
    """All index maps for one list of window lengths (host side, numpy; uploaded once)."""

    def __init__(self, lengths, n_self=5, n_cross=4, device=None):
        c = np.asarray(lengths, np.int64)
        self.c = c
        e = c + (c & 1)                                   # ehem.py:92-99: odd windows get one pad token
        self.self_layouts = [StageLayout(e)]
        for _ in range(n_self - 1):
            self.self_layouts.append(StageLayout((self.self_layouts[-1].L + 1) // 2))
        self.cross_layouts = [StageLayout(e // 2)]
        for _ in range(n_cross - 1):
            self.cross_layouts.append(StageLayout((self.cross_layouts[-1].L + 1) // 2))
        P0, Q0 = self.self_layouts[0], self.cross_layouts[0]
        # input gather: token rows of the frame arrays (window i = rows [start_i, start_i + c_i)); sentinel = pad token
        self.n_tokens = int(c.sum())
        starts = np.concatenate(([0], np.cumsum(c)[:-1]))
        inmap = np.full(P0.rows, self.n_tokens, np.int64)
        for i in range(len(c)):
            inmap[P0.base[i]:P0.base[i] + c[i]] = starts[i] + np.arange(c[i])
        self.inmap = inmap
        self.self_merge = [_merge_maps(self.self_layouts[s], self.self_layouts[s + 1]) for s in range(n_self - 1)]
        self.cross_merge = [_merge_maps(self.cross_layouts[s], self.cross_layouts[s + 1]) for s in range(n_cross - 1)]
        self.self_concat = [_concat_map(P0, self.self_layouts[s], s) for s in range(1, n_self)]
        self.cross_concat = [_concat_map(Q0, self.cross_layouts[s], s) for s in range(1, n_cross)]
        a1 = np.zeros(Q0.rows, np.int64)
        a2 = np.zeros(Q0.rows, np.int64)
        for i in range(len(c)):
            t = np.arange(Q0.L[i])
            a1[Q0.base[i] + t] = P0.base[i] + 2 * t
            a2[Q0.base[i] + t] = P0.base[i] + 2 * t + 1
        self.a1map, self.a2map = a1, a2
        # outputs in coding order: window i -> [evens (ceil(c/2)) | odds (floor(c/2))] (encode.py:126-131)
        self.even_rows = Q0.row_index((c + 1) // 2)
        self.odd_rows = Q0.row_index(c // 2)
        ne = (c + 1) // 2
        coded = np.concatenate(([0], np.cumsum(c)[:-1]))
        self.even_dst = np.concatenate([coded[i] + np.arange(ne[i]) for i in range(len(c))])
        self.odd_dst = np.concatenate([coded[i] + ne[i] + np.arange(c[i] // 2) for i in range(len(c))]) if (c // 2).sum() else np.zeros(0, np.int64)
        if device is not None:
            self.to(device)

    def to(self, dev):
        def up(a, dt=torch.int64):
            return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
        self.d = dict(
            inmap=up(self.inmap), a1map=up(self.a1map), a2map=up(self.a2map),
            even_rows=up(self.even_rows), odd_rows=up(self.odd_rows), even_dst=up(self.even_dst), odd_dst=up(self.odd_dst),
            self_merge=[(up(a), up(b)) for a, b in self.self_merge], cross_merge=[(up(a), up(b)) for a, b in self.cross_merge],
            self_concat=[up(m) for m in self.self_concat], cross_concat=[up(m) for m in self.cross_concat],
            self_tab=[up(l.table(), torch.int32) for l in self.self_layouts], cross_tab=[up(l.table(), torch.int32) for l in self.cross_layouts],
            knn_tab=up(self.self_layouts[0].table(real=True), torch.int32),
            self_valid=[up(l.valid(), torch.float32)[:, None] for l in self.self_layouts],
            cross_valid=[up(l.valid(), torch.float32)[:, None] for l in self.cross_layouts])
        return self


def _swin_layer(layer, x, valid, wtab, shift, query=None):
    """swin_transformer.py:654-706 on a packed layout (rows beyond a window's length are don't-care, except that the
    LayerNorm output is zeroed there - the reference zero-pads AFTER LayerNorm)."""
    att = layer.attention.self
    packed = getattr(layer, "_scp_packed_v", None)
    cross = query is not None
    if packed is None or packed[0].device != x.device or packed[2] != cross:
        if not cross:
            W = torch.cat((att.query.weight, att.key.weight, att.value.weight), 0).detach().contiguous()
            b = torch.cat((att.query.bias, att.key.bias, att.value.bias), 0).detach().contiguous()
        else:
            W = torch.cat((att.key.weight, att.value.weight), 0).detach().contiguous()
            b = torch.cat((att.key.bias, att.value.bias), 0).detach().contiguous()
        packed = (W, b, cross)
        layer._scp_packed_v = packed
    W, b, _ = packed
    h = layer_norm(x, layer.layernorm_before) * valid
    if not cross:
        qkv = linear(h, W, b)
        q, k, v = qkv[:, :256], qkv[:, 256:512], qkv[:, 512:]
    else:
        hq = layer_norm(query, layer.layernorm_before) * valid
        q = linear(hq, att.query.weight, att.query.bias)
        kv = linear(h, W, b)
        k, v = kv[:, :256], kv[:, 256:]
    o = native.swin_attention_packed(q, k, v, att.relative_position_bias_table, wtab, shift)
    x = linear(o, layer.attention.output.dense.weight, layer.attention.output.dense.bias, residual=x)
    y = linear(layer_norm(x, layer.layernorm_after), layer.intermediate.dense.weight, layer.intermediate.dense.bias, act="gelu")
    return linear(y, layer.output.dense.weight, layer.output.dense.bias, residual=x)


def _merge(m, x, maps):
    ev, od = maps
    xe = torch.cat((x, torch.zeros((1, x.shape[1]), dtype=x.dtype, device=x.device)))      # zero row for odd-length pads
    y = torch.cat((xe[ev], xe[od]), 1)
    return linear(layer_norm(y, m.norm), m.reduction.weight, None)


def _encoder(enc, x, valids, tabs, merges, query=None):
    hs = [x]
    for s, stage in enumerate(enc.layers):
        for b, blk in enumerate(stage.blocks):
            x = _swin_layer(blk, x, valids[s], tabs[s], SHIFT if b % 2 else 0, query)
        hs.append(x)
        if s < len(enc.layers) - 1:
            x = _merge(stage.downsample, x, merges[s])
            if query is not None:
                query = _merge(stage.downsample, query, merges[s])
    return hs


def _concat(hs, cmaps):
    return torch.cat([hs[1]] + [hs[s + 1][cmaps[s - 1]] for s in range(1, len(hs) - 1)], 1)


@torch.no_grad()
def ehem_forward_packed(model, ctx, pos, plan):
    """ctx uint8/int64 [T,12], pos float32 [T,3]: the frame's tokens, windows back to back (lengths = plan.c).
    Returns (logits_even_rows [sum ceil(c/2), 255], logits_odd_rows [sum floor(c/2), 255]) in window order."""
    d = plan.d
    g = model.geo_feat_generator
    dev = ctx.device
    pad_ctx = torch.tensor([[0, 0, 255] * 4], dtype=ctx.dtype, device=dev)
    ctx0 = torch.cat((ctx, pad_ctx))[d["inmap"]].long()                      # [P0,12]
    pos0 = torch.cat((pos, torch.zeros((1, 3), dtype=pos.dtype, device=dev)))[d["inmap"]].contiguous()
    P0 = ctx0.shape[0]
    x = torch.cat((F.embedding(ctx0[:, 2:11:3], g.occ_enc.weight).reshape(P0, -1),
                   F.embedding(ctx0[:, 0::3], g.level_enc.weight).reshape(P0, -1),
                   F.embedding(ctx0[:, 1::3], g.octant_enc.weight).reshape(P0, -1)), 1)
    ktab = d["knn_tab"]
    pos1 = _edge_conv_packed(g.conv1, pos0, ktab)
    pos2 = _edge_conv_packed(g.conv2, torch.cat((pos1, x), 1), ktab)
    x = leaky_mlp3(g.mlp2, x, exact=True)
    pos3 = _edge_conv_packed(g.conv3, torch.cat((pos2, x), 1), ktab)
    x = leaky_mlp3(g.mlp3, x)
    ec = leaky_mlp3(g.edge_mlp1, torch.cat((pos1, pos2, pos3), 1))
    ec = leaky_mlp3(g.edge_mlp2, torch.cat((pos3, ec), 1))
    feat = torch.cat((x, ec), 1)
    hs = _encoder(model.swin_self_transformer, feat, d["self_valid"], d["self_tab"], d["self_merge"])
    feat_a = leaky_mlp3(model.ancient_mlp, _concat(hs, d["self_concat"]))
    a1, a2 = feat_a[d["a1map"]], feat_a[d["a2map"]]
    prob1 = leaky_mlp3(model.prob_pred_mlp1, a1)
    pre_occ = ctx0[d["a1map"], 11]
    occ_feat = leaky_mlp3(model.pre_occ_mlp, F.embedding(pre_occ, g.occ_enc.weight))
    pre = torch.cat((occ_feat, leaky_mlp3(model.pre_attn_mlp, a1)), 1)
    hc = _encoder(model.swin_cross_transformer, pre, d["cross_valid"], d["cross_tab"], d["cross_merge"], query=a2)
    prob2 = leaky_mlp3(model.prob_pred_mlp2, torch.cat((_concat(hc, d["cross_concat"]), a2), 1))
    return prob1[d["even_rows"]], prob2[d["odd_rows"]]
