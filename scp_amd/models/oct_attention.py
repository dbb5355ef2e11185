"""OctAttention context model on MI355X (drop-in for models/oct_attention.py + models/attention_model.py).

`OctAttention(cfg).forward(data, pos)` with data int64 [B,c,4,3] = (occ, level, octant) x (ggp, gp, p, self) and
pos float32 [B,c,4,3] returns logits [B,c,255].  state_dict keys match the reference.
The dual-stream causal attention of attention_model.py:58-95 runs in one HIP kernel (csrc/octattn.hip); unlike
the reference, forward() is a pure function (it does not edit `data` in place, Appendix B-13).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import native
from .. import ops as _ops
from ..ops import linear as _linear

# False (tests only): the dense layers read fp32 rows and convert them in every tile (the round-2 form: identical bits)
PLANES = True


def linear(x, w, b=None, act=None, residual=None, scales=None):
    """OctAttention scales its embeddings by sqrt(600): the bf16x3 split (16-bit operands) leaves 1.8e-3 on the logits here, above
    the 1e-3 tolerance, so its dense layers run on the f16x3 kernel (22-bit operands, row scaled: the accuracy of an fp32 chain);
    the K = 12 position layer stays on the exact fp32 kernel."""
    return _linear(x, w, b, act=act, residual=residual, precise=True, scales=scales)


_KV_OFF = 640            # column of the value block inside the stacked key | value projection (600 rounded up to a multiple of 128)


def _kv_cat(a, D):
    """([2 * _KV_OFF, D] weight, [2 * _KV_OFF] bias) of the stacked key | value projection: rows [0, D) = key, [_KV_OFF, _KV_OFF + D) = value."""
    w = torch.zeros((2 * _KV_OFF, D), dtype=torch.float32, device=a.mlp_key.weight.device)
    b = torch.zeros((2 * _KV_OFF,), dtype=torch.float32, device=w.device)
    w[:D] = a.mlp_key.weight.detach()
    w[_KV_OFF:_KV_OFF + D] = a.mlp_value.weight.detach()
    b[:D] = a.mlp_key.bias.detach()
    b[_KV_OFF:_KV_OFF + D] = a.mlp_value.bias.detach()
    return w, b


class _PosEnc(nn.Module):
    def __init__(self, d_model, max_len):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.mlp_key = nn.Linear(d, d)
        self.mlp_query = nn.Linear(d, d)
        self.mlp_value = nn.Linear(d, d)


class _Layer(nn.Module):
    def __init__(self, d, hid):
        super().__init__()
        self.attn = _Attn(d)
        self.linear1 = nn.Linear(d, hid)
        self.linear2 = nn.Linear(hid, d)
        self.norm1 = nn.LayerNorm(d, eps=1e-5)
        self.norm2 = nn.LayerNorm(d, eps=1e-5)


class _Transformer(nn.Module):
    def __init__(self, d, hid, n_layers, ctx):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(d, hid) for _ in range(n_layers)])
        self.position_enc = _PosEnc(d, ctx)


class OctAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        m = cfg.model
        self.heads = m.head_num
        self.embed_dimension = 4 * (m.occ_embed_dim + m.level_embed_dim + m.octant_embed_dim + m.abs_pos_embed_dim)
        self.transformer_encoder = _Transformer(self.embed_dimension, m.hidden_dimension, m.layer_num, m.context_size)
        self.occ_enc = nn.Embedding(m.token_num + 1, m.occ_embed_dim)
        self.level_enc = nn.Embedding(m.max_octree_level + 1, m.level_embed_dim)
        self.octant_enc = nn.Embedding(9, m.octant_embed_dim)
        self.abs_pos_embed_dim = m.abs_pos_embed_dim
        if self.abs_pos_embed_dim:
            self.abs_pos_enc = nn.Linear(3, self.abs_pos_embed_dim)
        self.decoder0 = nn.Linear(self.embed_dimension, self.embed_dimension)
        self.decoder1 = nn.Linear(self.embed_dimension, m.token_num)
        mask = (torch.triu(torch.ones(m.context_size, m.context_size)) == 1).transpose(0, 1)
        self.register_buffer("mask", mask.float().masked_fill(mask == 0, float("-inf")).masked_fill(mask == 1, 0.0))
        self.eval()

    @classmethod
    def load_from_checkpoint(cls, path, cfg=None, map_location="cpu"):
        m = cls(cfg)
        sd = torch.load(path, map_location=map_location)
        m.load_state_dict(sd["state_dict"] if "state_dict" in sd else sd, strict=True)
        return m

    def _embed_torch(self, data, pos, cap):
        """The input stage as a sequence of torch index operations - the executable specification of csrc/octattn_embed.hip (tests
        compare the two bit for bit) and the path of the rows form (PLANES = False)."""
        B, c = data.shape[:2]
        data = data.long()
        occ, level, octant = data[..., 0], data[..., 1], data[..., 2]
        level = level - torch.clip(level[:, :, -1:] - cap, 0, None)           # oct_attention.py:57-61, out of place
        level = torch.clip(level, 0, self.cfg.model.max_octree_level)
        oe = F.embedding(occ, self.occ_enc.weight)
        ue = oe.clone()
        ue[:, :, -1] = self.occ_enc.weight[255]
        le = F.embedding(level, self.level_enc.weight)
        te = F.embedding(octant, self.octant_enc.weight)
        parts, parts_u = [oe, le, te], [ue, le, te]
        if self.abs_pos_embed_dim:
            pe = linear(pos, self.abs_pos_enc.weight, self.abs_pos_enc.bias)
            parts.append(pe)
            parts_u.append(pe)
        D = self.embed_dimension
        # both streams travel as one [2, B, c, D] tensor (0: known, 1: unknown): every layer they share runs as one launch
        E = torch.stack((torch.cat(parts, 3).reshape(B, c, D), torch.cat(parts_u, 3).reshape(B, c, D))) * math.sqrt(D)
        return E + self.transformer_encoder.position_enc.pe[:c]

    @torch.no_grad()
    def forward(self, data, pos=None):
        if not data.is_cuda:
            raise native.ScpError("OctAttention runs on the MI355X only (no CPU fallback)")
        B, c = data.shape[:2]
        cap = 10 if self.cfg.train.type == "obj" else 12
        D = self.embed_dimension
        planes = PLANES
        n = B * c
        pa = None                                      # planes of E, when the kernel that produced E wrote them
        pe_tab = self.transformer_encoder.position_enc.pe
        if planes and D <= 768 and D % 4 == 0 and pos is not None and c <= pe_tab.shape[0]:
            # the whole input stage in ONE kernel (csrc/octattn_embed.hip): three embedding lookups x four ancestors, the position
            # Linear, concatenation, sqrt(D) scale, position table - both streams - and the f16x3 operand of the first dense layers
            ctx8 = data.reshape(n, 12)
            ctx8 = ctx8 if ctx8.dtype == torch.uint8 else ctx8.to(torch.uint8)
            ap = self.abs_pos_enc if self.abs_pos_embed_dim else None
            E, pa = native.octattn_embed(ctx8.contiguous(), pos.reshape(n, 4, 3).float().contiguous(), c, self.occ_enc.weight, self.level_enc.weight,
                                         self.octant_enc.weight, None if ap is None else ap.weight, None if ap is None else ap.bias, pe_tab, cap,
                                         self.cfg.model.max_octree_level)
            E = E.reshape(2, B, c, D)
        else:
            E = self._embed_torch(data, pos, cap)

        def lin(a, x, w, b, act=None, residual=None, rows=None, scales=None, row_max=None):
            if planes:
                return native.linear_split_f16(a if rows is None else a.rows(*rows), _ops._split16(w), b, _ops._ACT[act], residual, row_max=row_max)
            return linear(x, w, b, act=act, residual=residual, scales=scales)

        for lyr in self.transformer_encoder.layers:
            a = lyr.attn
            E2 = E.reshape(-1, D)
            rs = rsq = None
            if planes:
                if pa is None:                         # the embedding stage's output: one standalone pass
                    pa = native.SplitActF16(E2 if E2.is_contiguous() else E2.contiguous())
            elif E.is_contiguous():
                # the three projections read the same rows: one pass for their power-of-two row scales (the query takes the unknown stream's half)
                rs = native.RowScales(E2)
                rsq = rs.rows(n, 2 * n)
            if planes and D <= _KV_OFF:
                # key | value as ONE product (round 4): the two [600, 600] weights stacked with 40 zero rows behind each (N = 1280 = five
                # 256-wide tiles, 256 x 256 tile configuration - half the LDS fill per flop of the 128 x 128 one): 1 339 against 2 x 887 us
                # at M = 262 144 (tools/mb_oa_cfg.py); same products in the same k order: identical bits.  The attention kernels take the
                # two column slices with their row stride.
                wkv, bkv = _ops.derived(a, "kv_cat", [a.mlp_key.weight, a.mlp_key.bias, a.mlp_value.weight, a.mlp_value.bias], lambda: _kv_cat(a, D))
                # round 5: the maxima the next kernels scale by come out of the producing GEMM's epilogue (one zeroed buffer per layer): max |v| of the
                # known stream for the attention kernel's V planes, the row maxima of linear1's output for linear2 - no pass over either tensor
                mxw = torch.zeros(2 * n + 1, dtype=torch.int32, device=E.device)
                vmax, h1max = mxw[2 * n:], mxw[:2 * n]
                kv = native.linear_split_f16(pa, _ops._split16(wkv), bkv, cfg=1, col_max=(vmax, _KV_OFF, _KV_OFF + D, n)).reshape(2, B, c, wkv.shape[0])
                key, val = kv[..., :D], kv[..., _KV_OFF:_KV_OFF + D]
            else:
                vmax = h1max = None
                key = lin(pa, E, a.mlp_key.weight, a.mlp_key.bias, scales=rs).reshape(2, B, c, D)
                val = lin(pa, E, a.mlp_value.weight, a.mlp_value.bias, scales=rs).reshape(2, B, c, D)
            q_u = lin(pa, E[1], a.mlp_query.weight, a.mlp_query.bias, rows=(n, 2 * n), scales=rsq).reshape(B, c, D)
            att = torch.empty_like(E)
            native.octattn_attention(q_u, key[0], key[1], val[0], val[1], self.heads, out=att[0], out_u=att[1], vmax=vmax)
            # norm(x + residual) in one pass; with `planes` the same pass writes the f16x3 operand of the layer that reads the result
            E, p1 = native.layernorm_add(att, E, lyr.norm1.weight, lyr.norm1.bias, 1e-5, planes=True) if planes else \
                (native.layernorm_add(att, E, lyr.norm1.weight, lyr.norm1.bias, 1e-5), None)
            h1 = lin(p1, E, lyr.linear1.weight, lyr.linear1.bias, act="relu", row_max=h1max)
            y2 = linear(h1, lyr.linear2.weight, lyr.linear2.bias, residual=E.reshape(h1.shape[:-1] + (D,)),
                        scales=None if h1max is None else native.RowScales.from_max(h1max)).reshape(E.shape)
            E, pa = native.layernorm_add(y2, None, lyr.norm2.weight, lyr.norm2.bias, 1e-5, planes=True) if planes else \
                (native.layernorm_add(y2, None, lyr.norm2.weight, lyr.norm2.bias, 1e-5), None)
        emu = E[1]
        if planes:
            d0 = native.linear_split_f16(pa.rows(n, 2 * n), _ops._split16(self.decoder0.weight), self.decoder0.bias, native.ACT_RELU)
            return linear(d0, self.decoder1.weight, self.decoder1.bias).reshape(B, c, -1)
        return linear(linear(emu, self.decoder0.weight, self.decoder0.bias, act="relu"), self.decoder1.weight, self.decoder1.bias)
