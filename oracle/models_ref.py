"""Plain-PyTorch fp32 restatement of the two context models (floating-point oracle).

TEST INFRASTRUCTURE ONLY (see oracle/scp_oracle.c header).  Functional style: every function takes
a `state_dict`-like mapping `sd` with the reference's key names and runs the
same op sequence as the reference on whatever device the tensors live on (CPU in tests).

Parity status: PINNED by tests/test_oracle_models.py against logits produced by the reference
modules themselves (tests/golden/logits_*.npz, swin_*.npz), tolerance 2e-5 abs on CPU.

Restated from: models/ehem.py:72-136, models/dgcnn.py:10-71,121-154,
models/swin_transformer.py:350-367,443-501,603-706,737-871, models/oct_attention.py:48-98,
models/attention_model.py:6-155.
"""
import math

import torch
import torch.nn.functional as F

WINDOW = 512
SHIFT = 256
HEADS = 4


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _mlp3(sd, p, x, slope=0.01):
    """nn.Sequential(Linear, LeakyReLU, Linear, LeakyReLU, Linear) with keys p.0/p.2/p.4."""
    x = F.leaky_relu(_lin(sd, p + ".0", x), slope)
    x = F.leaky_relu(_lin(sd, p + ".2", x), slope)
    return _lin(sd, p + ".4", x)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


# ----------------------------------------------------------------------------- DGCNN (dgcnn.py)
KNN_OVERRIDE = None   # tests may inject neighbour indices chosen elsewhere (callable(x [B,C,n], k) -> idx [B,n,k])


def knn(x, k):
    """dgcnn.py:10-45: x [B,C,n] -> idx [B,n,k] of the k largest  -|xi-xj|^2."""
    if KNN_OVERRIDE is not None:
        return KNN_OVERRIDE(x, k)
    return knn_default(x, k)


def knn_default(x, k):
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pd = -inner - xx - xx.transpose(2, 1)
    return torch.cat([pd[i:i + 4].topk(k=k, dim=-1)[1] for i in range(0, pd.shape[0], 4)], 0)


def edge_conv(sd, p, x, k):
    """get_graph_feature + conv/BN/LeakyReLU(0.2) + max over k (dgcnn.py:48-71,132-134)."""
    B, Cc, n = x.shape
    idx = knn(x, k)
    xt = x.transpose(2, 1).contiguous()
    nb = xt.reshape(B * n, Cc)[(idx + torch.arange(B, device=x.device).view(-1, 1, 1) * n).view(-1)].view(B, n, k, Cc)
    ctr = xt.view(B, n, 1, Cc).expand(B, n, k, Cc)
    feat = torch.cat((nb - ctr, ctr), 3).permute(0, 3, 1, 2).contiguous()
    y = F.conv2d(feat, sd[p + ".0.weight"])
    y = F.batch_norm(y, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"],
                     sd[p + ".1.bias"], False, 0.0, 1e-5)
    return F.leaky_relu(y, 0.2).max(dim=-1)[0]


def geo_feat(sd, data11, pos):
    """GeoFeatGenerator.forward, dgcnn.py:121-151.  data11 [B,c,11] int64, pos [B,3,c] f32 -> [B,c,256]."""
    g = "geo_feat_generator."
    B, c = data11.shape[:2]
    occ, level, octant = data11[:, :, 2::3], data11[:, :, ::3], data11[:, :, 1::3]
    x = torch.cat((F.embedding(occ, sd[g + "occ_enc.weight"]).reshape(B, c, -1),
                   F.embedding(level, sd[g + "level_enc.weight"]).reshape(B, c, -1),
                   F.embedding(octant, sd[g + "octant_enc.weight"]).reshape(B, c, -1)), 2)
    k = min(20, c)
    pos1 = edge_conv(sd, g + "conv1", pos, k)
    pos2 = edge_conv(sd, g + "conv2", torch.cat((pos1, x.transpose(1, 2)), 1), k)
    x = _mlp3(sd, g + "mlp2", x)
    pos3 = edge_conv(sd, g + "conv3", torch.cat((pos2, x.transpose(1, 2)), 1), k)
    x = _mlp3(sd, g + "mlp3", x)
    ec = _mlp3(sd, g + "edge_mlp1", torch.cat((pos1, pos2, pos3), 1).transpose(1, 2))
    ec = _mlp3(sd, g + "edge_mlp2", torch.cat((pos3.transpose(1, 2), ec), 2))
    return torch.cat((x, ec), 2)


# ----------------------------------------------------------------------------- 1-D Swin (swin_transformer.py)
def _windows(sd, p, x, L, shift):
    """proc_hidden_stats :631-652: LN -> zero-pad to x512 -> roll(-shift) -> [B*nW, 512, C]."""
    B, _, Cc = x.shape
    h = _ln(sd, p + ".layernorm_before", x)
    pad = (WINDOW - L % WINDOW) % WINDOW
    h = F.pad(h, (0, 0, 0, pad))
    if shift:
        h = torch.roll(h, shifts=(-shift,), dims=(1,))
    return h.reshape(-1, WINDOW, Cc), L + pad


def swin_layer(sd, p, x, L, shift, query=None):
    """SwinLayer.forward :654-706 with Attention.forward :443-501.  x [B,L,256]."""
    B, _, Cc = x.shape
    kv, Lp = _windows(sd, p, x, L, shift)
    qsrc = _windows(sd, p, query, L, shift)[0] if query is not None else kv
    a = p + ".attention.self."
    hd = Cc // HEADS

    def heads(t):
        return t.view(t.shape[0], WINDOW, HEADS, hd).permute(0, 2, 1, 3)

    q = heads(_lin(sd, a + "query", qsrc))
    kk = heads(_lin(sd, a + "key", kv))
    v = heads(_lin(sd, a + "value", kv))
    s = torch.matmul(q, kk.transpose(-1, -2)) / math.sqrt(hd)
    ar = torch.arange(WINDOW, device=x.device)
    rel = sd[a + "relative_position_bias_table"][(ar[:, None] - ar[None, :] + WINDOW - 1).view(-1)]
    s = s + rel.view(WINDOW, WINDOW, HEADS).permute(2, 0, 1).unsqueeze(0)
    if shift:
        # get_attn_mask :603-623: regions [0,Lp-512) / [Lp-512, Lp-256) / [Lp-256, Lp) in shifted coordinates
        reg = torch.zeros(Lp, device=x.device)
        reg[Lp - WINDOW:Lp - shift] = 1
        reg[Lp - shift:] = 2
        rw = reg.view(-1, WINDOW)
        m = rw.unsqueeze(1) - rw.unsqueeze(2)
        m = torch.where(m != 0, torch.full_like(m, -100.0), torch.zeros_like(m))
        nW = m.shape[0]
        s = (s.view(B, nW, HEADS, WINDOW, WINDOW) + m.unsqueeze(1).unsqueeze(0)).view(-1, HEADS, WINDOW, WINDOW)
    o = torch.matmul(torch.softmax(s, dim=-1), v).permute(0, 2, 1, 3).reshape(-1, WINDOW, Cc)
    o = _lin(sd, p + ".attention.output.dense", o).view(B, Lp, Cc)
    if shift:
        o = torch.roll(o, shifts=(shift,), dims=(1,))
    h = x + o[:, :L]
    y = _lin(sd, p + ".intermediate.dense", _ln(sd, p + ".layernorm_after", h))
    y = F.gelu(y)
    return h + _lin(sd, p + ".output.dense", y)


def patch_merge(sd, p, x, L):
    """SwinPatchMerging.forward :350-367."""
    if L % 2:
        x = F.pad(x, (0, 0, 0, 1))
    x = torch.cat([x[:, 0::2], x[:, 1::2]], -1)
    return F.linear(_ln(sd, p + ".norm", x), sd[p + ".reduction.weight"])


def swin_encoder(sd, p, x, L, depths, query=None):
    """SwinEncoder.forward :793-871 with output_hidden_states_before_downsampling -> list of hidden states."""
    hs = [x]
    for s, depth in enumerate(depths):
        for b in range(depth):
            x = swin_layer(sd, f"{p}.layers.{s}.blocks.{b}", x, L, SHIFT if b % 2 else 0, query)
        hs.append(x)
        if s < len(depths) - 1:
            x = patch_merge(sd, f"{p}.layers.{s}.downsample", x, L)
            if query is not None:
                query = patch_merge(sd, f"{p}.layers.{s}.downsample", query, L)
            L = (L + 1) // 2
    return hs


def concat_states(hs):
    """ehem.py:75-86: nearest-neighbour x2 upsampling of the coarser stages, truncated, channel-concat."""
    n = hs[1].shape[1]
    out = [hs[1]]
    for s in range(2, len(hs)):
        idx = torch.arange(n, device=hs[1].device) >> (s - 1)
        out.append(hs[s][:, idx])
    return torch.cat(out, 2)


# ----------------------------------------------------------------------------- EHEM (ehem.py:88-136)
SELF_DEPTHS = (4, 4, 4, 4, 2)
CROSS_DEPTHS = (2, 2, 1, 1)


def ehem_forward(sd, data, pos):
    """data [B,c,4,3] int64 = (level, octant, occ) x (ggp, gp, p, self); pos [B,3,c] f32.
    Returns (logits_even [B,ceil(c/2),255], logits_odd [B,floor(c/2),255])."""
    padded = data.shape[1] % 2 == 1
    if padded:
        pad = torch.zeros_like(data[:, :1])
        pad[:, :, :, 2] = 255
        data = torch.cat((data, pad), 1)
        pos = torch.cat((pos, torch.zeros_like(pos[:, :, :1])), 2)
    B, c = data.shape[:2]
    pre_occ = data[:, ::2, -1, -1]
    d11 = data.reshape(B, c, -1)[:, :, :-1]
    feat = geo_feat(sd, d11, pos)
    hs = swin_encoder(sd, "swin_self_transformer", feat, c, SELF_DEPTHS)
    feat_a = _mlp3(sd, "ancient_mlp", concat_states(hs))
    a1, a2 = feat_a[:, ::2], feat_a[:, 1::2]
    prob1 = _mlp3(sd, "prob_pred_mlp1", a1)
    occ_feat = _mlp3(sd, "pre_occ_mlp", F.embedding(pre_occ, sd["geo_feat_generator.occ_enc.weight"]))
    pre = torch.cat((occ_feat, _mlp3(sd, "pre_attn_mlp", a1)), 2)
    hc = swin_encoder(sd, "swin_cross_transformer", pre, a2.shape[1], CROSS_DEPTHS, query=a2)
    prob2 = _mlp3(sd, "prob_pred_mlp2", torch.cat((concat_states(hc), a2), 2))
    if padded:
        prob2 = prob2[:, :-1]
    return prob1, prob2


# ----------------------------------------------------------------------------- OctAttention
def octattn_forward(sd, data, pos, max_octree_level=12, train_type="kitti"):
    """oct_attention.py:48-83 + attention_model.py.  data [B,c,4,3] = (occ, level, octant); pos [B,c,4,3]."""
    B, c = data.shape[:2]
    occ, level, octant = data[..., 0], data[..., 1].clone(), data[..., 2]
    level = level - torch.clip(level[:, :, -1:] - (10 if train_type == "obj" else 12), 0, None)
    level = torch.clip(level, 0, max_octree_level)
    oe = F.embedding(occ, sd["occ_enc.weight"])
    ue = oe.clone()
    ue[:, :, -1] = sd["occ_enc.weight"][255]
    le = F.embedding(level, sd["level_enc.weight"])
    te = F.embedding(octant, sd["octant_enc.weight"])
    pe = _lin(sd, "abs_pos_enc", pos)
    E = oe.shape[-1] + le.shape[-1] + te.shape[-1] + pe.shape[-1]
    D = 4 * E
    emb = torch.cat((oe, le, te, pe), 3).reshape(B, c, D) * math.sqrt(D)
    emu = torch.cat((ue, le, te, pe), 3).reshape(B, c, D) * math.sqrt(D)
    pe_tab = sd["transformer_encoder.position_enc.pe"][:c]
    emb, emu = emb + pe_tab, emu + pe_tab
    mask = sd["mask"][:c, :c]
    nh = 4
    hd = D // nh
    eye = torch.eye(c, device=data.device)[None, None]
    n_layers = len({k.split(".")[2] for k in sd if k.startswith("transformer_encoder.layers.")})
    for l in range(n_layers):
        p = f"transformer_encoder.layers.{l}."

        def sl(t):
            return t.view(B, c, nh, hd).permute(0, 2, 1, 3)

        key, key_u = sl(_lin(sd, p + "attn.mlp_key", emb)), sl(_lin(sd, p + "attn.mlp_key", emu))
        q_u = sl(_lin(sd, p + "attn.mlp_query", emu))
        val, val_u = sl(_lin(sd, p + "attn.mlp_value", emb)), sl(_lin(sd, p + "attn.mlp_value", emu))
        score = torch.matmul(q_u, key.transpose(-1, -2)) / math.sqrt(hd)
        out = torch.matmul(torch.softmax(score + mask, -1), val)
        zero = torch.sum(q_u * key_u, dim=3) / math.sqrt(hd)
        su = (1 - eye) * score + torch.diag_embed(zero)
        au = torch.softmax(su + mask, -1)
        out_u = torch.matmul((1 - eye) * au, val)
        out_u = out_u + torch.einsum("ijk,ijkl->ijkl", torch.diagonal(au, dim1=2, dim2=3), val_u)

        def ctx(t):
            return t.permute(0, 2, 1, 3).reshape(B, c, D)

        emb = _ln(sd, p + "norm1", ctx(out) + emb)
        emu = _ln(sd, p + "norm1", ctx(out_u) + emu)
        emb = _ln(sd, p + "norm2", emb + _lin(sd, p + "linear2", torch.relu(_lin(sd, p + "linear1", emb))))
        emu = _ln(sd, p + "norm2", emu + _lin(sd, p + "linear2", torch.relu(_lin(sd, p + "linear1", emu))))
    return _lin(sd, "decoder1", torch.relu(_lin(sd, "decoder0", emu)))
