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
from ..ops import linear_s, leaky_mlp3_s, split_cat, linear, layer_norm, leaky_mlp3, _split, derived, frozen_epoch
from .ehem import SHIFT, WINDOW, _edge_conv_packed, qkv_fused


def _ceil512(a):
    return (a + WINDOW - 1) // WINDOW * WINDOW


def _tokens(L):
    """(window id, position in window) of every real token of a layout, windows concatenated (device tensors)."""
    wid = torch.repeat_interleave(torch.arange(L.numel(), device=L.device), L)
    first = torch.cumsum(L, 0) - L
    return wid, torch.arange(int(L.sum()), device=L.device) - first[wid]


class StageLayout:
    """Rows of one Swin stage: window i owns rows [base[i], base[i] + Lp[i]), the first L[i] of them real."""

    def __init__(self, L):
        self.L = L
        self.Lp = _ceil512(L)
        self.base = torch.cumsum(self.Lp, 0) - self.Lp
        self.rows = int(self.Lp.sum())
        self.wid, self.t = _tokens(L)
        self.real_rows = self.base[self.wid] + self.t          # global rows of the real tokens

    def table(self, real=False):
        """int32 [rows/512, 2]: (sequence base row, padded length | real length) for every 512-row chunk."""
        reps = self.Lp // WINDOW
        return torch.stack([torch.repeat_interleave(self.base, reps), torch.repeat_interleave(self.L if real else self.Lp, reps)],
                           1).to(torch.int32).contiguous()

    def valid(self):
        v = torch.zeros(self.rows, dtype=torch.float32, device=self.L.device)
        v[self.real_rows] = 1.0
        return v[:, None]

    def row_index(self, counts):
        """global rows of the first counts[i] tokens of every window, concatenated."""
        wid, t = _tokens(counts)
        return self.base[wid] + t


def _merge_maps(cur, nxt):
    """patch merging cur -> nxt: rows (even, odd) of cur feeding every row of nxt; `cur.rows` is the index of an all-zero row."""
    dev = cur.L.device
    even = torch.full((nxt.rows,), cur.rows, dtype=torch.int64, device=dev)
    odd = torch.full((nxt.rows,), cur.rows, dtype=torch.int64, device=dev)
    even[nxt.real_rows] = cur.base[nxt.wid] + 2 * nxt.t
    o = 2 * nxt.t + 1
    odd[nxt.real_rows] = torch.where(o < cur.L[nxt.wid], cur.base[nxt.wid] + o, torch.full_like(o, cur.rows))
    return even, odd


def _out_map(q0, counts, first):
    """coded position of token t of every cross-stage-0 row (first[w] + t for t < counts[w], else -1)."""
    m = torch.full((q0.rows,), -1, dtype=torch.int64, device=q0.L.device)
    wid, t = _tokens(counts)
    m[q0.base[wid] + t] = first[wid] + t
    return m


def _concat_map(l0, ls, s):
    """rows of stage-s layout `ls` that stage-0 token t of each window gathers (t >> s)."""
    m = torch.zeros(l0.rows, dtype=torch.int64, device=l0.L.device)
    m[l0.real_rows] = ls.base[l0.wid] + (l0.t >> s)
    return m


class PackedPlan:
    """All index maps for one list of window lengths, built ON THE DEVICE from the (tiny) list of lengths with a few dozen
    vectorised index ops - the host never touches a per-token array."""

    def __init__(self, lengths, n_self=5, n_cross=4, device=None, use_native=True):
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        dev = torch.device(dev)
        if use_native and dev.type == "cuda" and n_self == 5 and n_cross == 4:
            # one kernel launch for all 47 maps (csrc/plan.hip); the torch construction below is its executable specification
            # (tests/test_host_logic.py checks it on the CPU, tests/test_gpu_model.py checks the kernel against it)
            self.c = torch.as_tensor(np.asarray(lengths, np.int64))
            self.n_tokens = int(np.asarray(lengths, np.int64).sum())
            self.rows, self.d = native.packed_plan(lengths, dev)
            return
        c = torch.as_tensor(np.asarray(lengths, np.int64), device=dev)
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
        inmap = torch.full((P0.rows,), self.n_tokens, dtype=torch.int64, device=dev)
        inmap[P0.row_index(c)] = torch.arange(self.n_tokens, device=dev)
        a1 = torch.zeros(Q0.rows, dtype=torch.int64, device=dev)
        a2 = torch.zeros(Q0.rows, dtype=torch.int64, device=dev)
        a1[Q0.real_rows] = P0.base[Q0.wid] + 2 * Q0.t
        a2[Q0.real_rows] = P0.base[Q0.wid] + 2 * Q0.t + 1
        # outputs in coding order: window i -> [evens (ceil(c/2)) | odds (floor(c/2))] (encode.py:126-131)
        ne, no = (c + 1) // 2, c // 2
        coded = torch.cumsum(c, 0) - c
        we, te = _tokens(ne)
        wo, to = _tokens(no)
        self.d = dict(
            inmap=inmap, a1map=a1, a2map=a2, even_rows=Q0.row_index(ne), odd_rows=Q0.row_index(no),
            even_dst=coded[we] + te, odd_dst=coded[wo] + ne[wo] + to,
            self_merge=[_merge_maps(self.self_layouts[s], self.self_layouts[s + 1]) for s in range(n_self - 1)],
            cross_merge=[_merge_maps(self.cross_layouts[s], self.cross_layouts[s + 1]) for s in range(n_cross - 1)],
            # (stage-s row of stage-0 token t >> s: the direct form of concat_states; the forward now uses the parent maps below)
            self_concat=[_concat_map(P0, self.self_layouts[s], s) for s in range(1, n_self)],
            cross_concat=[_concat_map(Q0, self.cross_layouts[s], s) for s in range(1, n_cross)],
            self_tab=[l.table() for l in self.self_layouts], cross_tab=[l.table() for l in self.cross_layouts],
            knn_tab=P0.table(real=True),
            self_valid=[l.valid().contiguous() for l in self.self_layouts], cross_valid=[l.valid().contiguous() for l in self.cross_layouts],
            # parent rows: stage s token t -> stage s + 1 token t >> 1 (the hierarchical evaluation of a layer over concat_states)
            self_parent=[_concat_map(self.self_layouts[s], self.self_layouts[s + 1], 1) for s in range(n_self - 1)],
            cross_parent=[_concat_map(self.cross_layouts[s], self.cross_layouts[s + 1], 1) for s in range(n_cross - 1)],
            even_out=_out_map(Q0, ne, coded), odd_out=_out_map(Q0, no, coded + ne))


def _rowchain_weights(layer, cross):
    """Derived weights of a block for the row-chain kernels (rebuilt when a parameter changes): LayerNorm-folded q|k|v (or k|v and q for a
    cross layer) and the packed post-attention buffer."""
    att = layer.attention.self
    lnb, lna = layer.layernorm_before, layer.layernorm_after
    W, b = qkv_fused(layer, cross)
    fc1, fc2, proj = layer.intermediate.dense, layer.output.dense, layer.attention.output.dense
    srcs = [W, b, lnb.weight, lnb.bias, lna.weight, lna.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, proj.weight, proj.bias]
    if cross:
        srcs += [att.query.weight, att.query.bias]

    def build():
        out = dict(kv=native.LnFoldedWeight(W, lnb.weight, lnb.bias), b=b,
                   post=native.PostAttnWeights(proj.weight, proj.bias, lna.weight, lna.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias))
        if cross:
            out["q"] = native.LnFoldedWeight(att.query.weight, lnb.weight, lnb.bias)
        return out
    return derived(layer, "rowchain+" if cross else "rowchain", srcs, build)


def _block_consts(layer, cross):
    """(derived weights, bias table, query bias, eps before, eps after) of a block; inside ops.frozen_weights looked up once per frame."""
    ep = frozen_epoch()
    key = "_scp_fast_x" if cross else "_scp_fast"
    fast = layer.__dict__.get(key) if ep else None
    if fast is not None and fast[0] == ep:
        return fast[1:]
    att = layer.attention.self
    out = (_rowchain_weights(layer, cross), att.relative_position_bias_table, (att.query.bias if cross else None), layer.layernorm_before.eps, layer.layernorm_after.eps)
    if ep:
        layer.__dict__[key] = (ep,) + out
    return out


def _swin_layer_rowchain(layer, x, valid, wtab, shift, query=None, tiles=None, q_pre=None):
    """The same block on two row-chain launches around the attention kernel.  With `tiles` (the 128-row tiles that hold a real row) the
    block's second half runs IN PLACE on those tiles only and the attention skips the query tiles of pure window padding: their rows
    keep finite old values, which is all the next block needs (it multiplies their normalised rows by valid = 0)."""
    cross = query is not None
    # inside ops.frozen_weights (one frame) a block's derived weights and the attributes the launches need are looked up once per frame: the
    # decoder runs ~1 400 blocks per frame, and the nn.Module attribute walks + the validation of twelve source tensors were 45 us of each
    w, table, qbias, eps_b, eps_a = _block_consts(layer, cross)
    v1 = None if valid is None else valid.reshape(-1)
    if native.attention_bf16x3():
        # keys and values leave the projection as the bf16 planes the attention kernel stages by LDS-DMA (identical bits to the fp32 hand-over)
        if not cross:
            q, kvp = native.swin_ln_qkv(x, w["kv"], w["b"], eps_b, v1)
        else:
            # q_pre: the block's query projection, computed ahead by ehem_phase2_prepare (the decoder: it does not depend on the decoded symbols)
            q = q_pre if q_pre is not None else native.swin_ln_linear(query, w["q"], qbias, eps_b, v1)
            _, kvp = native.swin_ln_qkv(x, w["kv"], w["b"], eps_b, v1)
        o = native.swin_attention_packed_planes(q, kvp, table, wtab, shift, split=True, valid=v1 if tiles is not None else None)
        if tiles is not None and tiles.shape[0] > 0:
            return native.swin_post_attn(o, x, w["post"], eps_a, out=x, tiles=tiles)
        return native.swin_post_attn(o, x, w["post"], eps_a)
    att, lnb = layer.attention.self, layer.layernorm_before
    # numeric profile "attention = fp32 MFMA" (scp_ctx): the fp32-fed attention kernel takes q, k, v as rows
    if not cross:
        qkv = native.swin_ln_linear(x, w["kv"], w["b"], lnb.eps, v1)
        q, k, v = qkv[:, :256], qkv[:, 256:512], qkv[:, 512:]
    else:
        q = q_pre if q_pre is not None else native.swin_ln_linear(query, w["q"], att.query.bias, lnb.eps, v1)
        kv = native.swin_ln_linear(x, w["kv"], w["b"], lnb.eps, v1)
        k, v = kv[:, :256], kv[:, 256:]
    o = native.swin_attention_packed(q, k, v, att.relative_position_bias_table, wtab, shift, split=True)
    return native.swin_post_attn(o, x, w["post"], layer.layernorm_after.eps)


def _swin_layer(layer, x, valid, wtab, shift, query=None, tiles=None, q_pre=None):
    """swin_transformer.py:654-706 on a packed layout (rows beyond a window's length are don't-care, except that the
    LayerNorm output is zeroed there - the reference zero-pads AFTER LayerNorm): two row-chain launches around the attention kernel."""
    fc1, fc2 = layer.intermediate.dense, layer.output.dense
    if x.shape[1] != 256 or fc1.weight.shape != (1024, 256) or fc2.weight.shape != (256, 1024):
        raise native.ScpError("EHEM's Swin blocks are 256 wide with a 1024-wide MLP (configs/model/ehem.yaml); other widths are not built")
    return _swin_layer_rowchain(layer, x, valid, wtab, shift, query, tiles, q_pre)


def _merge(m, x, maps):
    """SwinPatchMerging: gather (even, odd) token of every pair + LayerNorm(512) + the 512 -> 256 reduction in one row-chain launch."""
    ev, od = maps          # index == x.shape[0] stands for the zero row an odd-length window is padded with
    if x.shape[1] != 256 or m.reduction.weight.shape != (256, 512) or m.reduction.bias is not None:
        raise native.ScpError("SwinPatchMerging: 2 x 256 -> 256 without bias expected")
    mw = derived(m, "rowchain_merge", [m.reduction.weight, m.norm.weight, m.norm.bias],
                 lambda: native.MergeWeights(m.reduction.weight, m.norm.weight, m.norm.bias))
    return native.swin_merge(x, ev, od, mw, m.norm.eps)


def _encoder(enc, x, valids, tabs, merges, query=None, tiles=None, q_pre=None):
    """q_pre[s][b]: the query projections of the cross blocks when they were computed ahead (then the query stream is not merged here)."""
    hs = [x]
    for s, stage in enumerate(enc.layers):
        for b, blk in enumerate(stage.blocks):
            x = _swin_layer(blk, x, valids[s], tabs[s], SHIFT if b % 2 else 0, query, None if tiles is None else tiles[s],
                            None if q_pre is None else q_pre[s][b])
        hs.append(x)
        if s < len(enc.layers) - 1:
            x = _merge(stage.downsample, x, merges[s])
            if query is not None and q_pre is None:
                query = _merge(stage.downsample, query, merges[s])
    return hs


def _concat(hs, cmaps, extra=None):
    """concat_states (ehem.py:75-86): stage s is gathered at token >> s straight into its 256-column slot of a split (bf16 hi/lo)
    buffer - the operand of the MLP that follows; `extra` = (src, map) appends one more gathered slot (the odd-token features of
    the cross branch, ehem.py:124)."""
    n = len(hs) - 1 + (1 if extra is not None else 0)
    out = native.SplitAct.empty(hs[1].shape[0], 256 * n, hs[1].device)
    native.split_rows(hs[1], out=out.cols(0, 256))
    for s in range(1, len(hs) - 1):
        native.split_rows(hs[s + 1], idx=cmaps[s - 1], out=out.cols(256 * s, 256 * (s + 1)))
    if extra is not None:
        k = len(hs) - 1
        native.split_rows(extra[0], idx=extra[1], out=out.cols(256 * k, 256 * (k + 1)))
    return out


# False (tests only): build the 1280-wide concatenation and run one product (the literal form of ehem.py:75-86); product: per stage
HIER = True


# False (tests, A/B; SCP_CONCAT_FUSE2=0): stages 0 and 1 of a layer over concat_states as two launches with an fp32 partial sum between them (rounds 1 - 5)
FUSE2 = __import__("os").environ.get("SCP_CONCAT_FUSE2", "1")[:1] != "0"


def _concat_layer(lin, hs, parents, extra=None, grand=None, a0_pre=None):
    """LeakyReLU(Linear(concat_states(hs))) (ehem.py:75-86 + the first layer of the MLP that consumes it) WITHOUT building the
    concatenation: the layer's weight is cut into one 256-column slab per Swin stage, stage s contributes h_s . W_s^T at its OWN
    resolution (rows / 2^s), and the partial sums flow from the coarsest stage down through the parent-row maps (token t of stage s
    adds row t >> 1 of stage s + 1) as a gathered residual of the next finer product.  1.94x instead of 5x the stage-0 row count
    in products and in split conversions; the result differs from the one-shot layer only in fp32 summation order.
    `extra`: one more 256-wide input at stage-0 resolution (the odd-token features of the cross branch, ehem.py:124)."""
    n = len(hs) - 1
    def build():
        W = lin.weight.detach()
        slabs = [W[:, 256 * s:256 * (s + 1)].contiguous() for s in range(n)]
        if extra is not None:
            slabs[0] = torch.cat((slabs[0], W[:, 256 * n:256 * (n + 1)]), 1).contiguous()
        return slabs
    cache = derived(lin, "slabs+" if extra is not None else "slabs", (lin.weight,), build)
    fused = FUSE2 and n >= 3 and grand is not None and native.WTILE and hs[1].shape[0] % 256 == 0
    z = None
    for s in range(n - 1, 1 if fused else 0, -1):
        z = linear_s(native.split_rows(hs[s + 1]), cache[s], None, residual=z, res_map=None if z is None else parents[s], res_first=z is not None)
    if a0_pre is not None:           # the operand's `extra` columns were written ahead (ehem_phase2_prepare): only stage 0's own columns are left
        native.split_rows(hs[1], out=a0_pre.cols(0, 256))
        a0 = a0_pre
    else:
        a0 = native.split_rows(hs[1]) if extra is None else split_cat((hs[1], extra))
    if fused:
        # round 6: stages 0 and 1 in ONE launch - the stage-1 product of a 256-token tile's own 128 parents stays in the accumulators
        # (csrc/gemm_split.hip: gemm_hier2_kernel); z = the partial sum of stages 2 .. n - 1 at stage-2 resolution, gathered through `grand`
        return native.linear_split_hier2(a0, _split(cache[0]), native.split_rows(hs[2]), _split(cache[1]), parents[0], lin.bias, native.ACT_LEAKY,
                                         residual=z, res_map=grand)
    return linear_s(a0, cache[0], lin.bias, act="leaky", residual=z, res_map=None if z is None else parents[0], res_first=z is not None,
                    want="split")


def _mlp_over_concat(seq, hs, parents, extra=None, grand=None, a0_pre=None):
    """leaky_mlp3 over concat_states: hierarchical first layer, then the two remaining layers on split activations.
    grand: stage-2 row of every stage-0 row (plan `self_concat[1]` / `cross_concat[1]`) - enables the fused two-stage launch."""
    a = _concat_layer(seq[0], hs, parents, extra, grand, a0_pre)
    a = linear_s(a, seq[2].weight, seq[2].bias, act="leaky", want="split")
    return linear_s(a, seq[4].weight, seq[4].bias)


# False (tests, A/B): the 256-wide heads as three split-GEMM launches each (rounds 1 - 5) instead of one row-chain launch
CHAIN_HEADS = __import__("os").environ.get("SCP_HEADS", "chain") != "split"      # (SCP_HEADS=split: A/B bracket)


def _mlp3_weights(owner, name, seq):
    """native.Mlp3Weights of a three-layer head, cached on the model and rebuilt when a parameter changes."""
    return derived(owner, "mlp3_" + name, [seq[i].weight for i in (0, 2, 4)] + [seq[i].bias for i in (0, 2, 4)], lambda: native.Mlp3Weights(seq))


def _head3(model, name, a, out=None, out_map=None, ncols=None):
    """A 256 -> .. -> .. three-layer head (ehem.py:113-121) on fp32 rows `a` [M, 256]: ONE row-chain launch (csrc/rowchain.hip: rc_mlp3_kernel; the
    hidden activations never leave the registers) - or, with CHAIN_HEADS off, three split-GEMM launches.  out: fp32 rows (may be a column-offset
    view); out_map: scatter rows (negative = dropped), then `out` is the table."""
    seq = getattr(model, name)
    N = seq[4].weight.shape[0]
    if CHAIN_HEADS:
        if out is None:
            out = torch.empty((a.shape[0], -(-N // 4) * 4), dtype=torch.float32, device=a.device)
        native.mlp3_rows(a, _mlp3_weights(model, name, seq), out, out_map=out_map, ncols=ncols)
        return out[:, :N] if out_map is None else None
    sa = native.split_rows(a)
    if out_map is not None:
        sa = linear_s(sa, seq[0].weight, seq[0].bias, act="leaky", want="split")
        sa = linear_s(sa, seq[2].weight, seq[2].bias, act="leaky", want="split")
        native.linear_split_scatter(sa, _split(seq[4].weight), seq[4].bias, out_map, out)
        return None
    return leaky_mlp3_s(seq, sa, out=out)


def _head_to_table(model, a, out_map, table):
    """prob_pred_mlp1 whose last layer writes its rows straight to their coding-order positions in `table` (rows out_map[m] >= 0)."""
    if CHAIN_HEADS and table.stride(0) < 256:
        raise native.ScpError("the coding-order table must have rows of at least 256 floats")
    _head3(model, "prob_pred_mlp1", a, out=table, out_map=out_map)


@torch.no_grad()
def ehem_phase1_packed(model, ctx, pos, plan, table=None):
    """Everything that does not depend on the windows' own occupancies (ehem.py:92-115).
    Returns (even-node logits in window order [sum ceil(c/2), 255], state for phase 2); with `table` (fp32 [>= n_tokens, ld] view
    starting at the chunk's first coded row) the logits go straight to their coding-order rows and None is returned for them."""
    d = plan.d
    g = model.geo_feat_generator
    dev = ctx.device
    fused_in = (ctx.dtype == torch.uint8 and g.occ_enc.weight.shape[1] == 16 and g.level_enc.weight.shape[1] == 4
                and g.octant_enc.weight.shape[1] == 4)
    if fused_in:   # gather into the packed layout + the three embedding lookups in one kernel
        x, pos0, occ_self = native.embed_gather(ctx, pos, d["inmap"], g.occ_enc.weight, g.level_enc.weight, g.octant_enc.weight)
        P0 = x.shape[0]
    else:
        pad_ctx = torch.tensor([[0, 0, 255] * 4], dtype=ctx.dtype, device=dev)
        ctx0 = torch.cat((ctx, pad_ctx))[d["inmap"]].long()                      # [P0,12]
        pos0 = torch.cat((pos, torch.zeros((1, 3), dtype=pos.dtype, device=dev)))[d["inmap"]].contiguous()
        P0 = ctx0.shape[0]
        x = torch.cat((F.embedding(ctx0[:, 2:11:3], g.occ_enc.weight).reshape(P0, -1),
                       F.embedding(ctx0[:, 0::3], g.level_enc.weight).reshape(P0, -1),
                       F.embedding(ctx0[:, 1::3], g.octant_enc.weight).reshape(P0, -1)), 1)
        occ_self = ctx0[:, 11]
    ktab = d["knn_tab"]
    # the feature rows of the second and third search, cat(pos1, x) and cat(pos2, mlp2(x)) (dgcnn.py:136,142): the edge convolutions write their
    # columns in place (round 6: two [P0, 144 | 192] concatenation copies per frame less); pos1 / pos2 stay views of these buffers
    c1, c2 = g.conv1[0].weight.shape[0], g.conv2[0].weight.shape[0]
    f2 = torch.empty((P0, c1 + x.shape[1]), dtype=torch.float32, device=dev)
    f2[:, c1:] = x
    pos1 = _edge_conv_packed(g.conv1, pos0, ktab, out=f2[:, :c1])
    x = leaky_mlp3(g.mlp2, x, exact=True)
    f3 = torch.empty((P0, c2 + x.shape[1]), dtype=torch.float32, device=dev)
    f3[:, c2:] = x
    pos2 = _edge_conv_packed(g.conv2, f2, ktab, out=f3[:, :c2])
    pos3 = _edge_conv_packed(g.conv3, f3, ktab, feeds_knn=False)
    # dense part on the split-operand GEMM: fp32 tensors are split once, the MLP chains stay in the split format, the two
    # halves of `feat` are written straight into their column slots
    nx = g.mlp3[4].weight.shape[0]
    feat = torch.empty((P0, nx + g.edge_mlp2[4].weight.shape[0]), dtype=torch.float32, device=dev)
    leaky_mlp3_s(g.mlp3, native.split_rows(x), out=feat[:, :nx])
    if (pos1.shape[1], pos2.shape[1], pos3.shape[1]) != (64, 128, 256) or feat.shape[1] - nx != 128:
        raise native.ScpError("GeoFeatGenerator: edge features of 64 / 128 / 256 channels expected (models/dgcnn.py:74-119)")
    # both edge MLPs in one row-chain launch: six layers chained through the accumulators, edge_mlp1's output never leaves the registers
    ew = derived(g, "rowchain_edge", [l.weight for l in (g.edge_mlp1[0], g.edge_mlp1[2], g.edge_mlp1[4], g.edge_mlp2[0], g.edge_mlp2[2], g.edge_mlp2[4])] +
                 [l.bias for l in (g.edge_mlp1[0], g.edge_mlp1[2], g.edge_mlp1[4], g.edge_mlp2[0], g.edge_mlp2[2], g.edge_mlp2[4])],
                 lambda: native.EdgeMlpWeights(g.edge_mlp1, g.edge_mlp2))
    native.geo_edge_mlps(pos1, pos2, pos3, ew, feat[:, nx:])
    hs = _encoder(model.swin_self_transformer, feat, d["self_valid"], d["self_tab"], d["self_merge"], tiles=d.get("self_tiles"))
    feat_a = _mlp_over_concat(model.ancient_mlp, hs, d["self_parent"], grand=d["self_concat"][1]) if HIER else leaky_mlp3_s(model.ancient_mlp, _concat(hs, d["self_concat"]))
    Q0 = d["a1map"].shape[0]
    # round 6: the even tokens' features travel as fp32 rows (like the odd tokens'): the one-launch heads read rows, not planes
    a1 = native.gather_rows(feat_a, d["a1map"], torch.empty((Q0, 256), dtype=torch.float32, device=dev))
    a2 = native.gather_rows(feat_a, d["a2map"], torch.empty((Q0, 256), dtype=torch.float32, device=dev))
    st = dict(a1=a1, a2=a2, pre_occ=occ_self[d["a1map"]])
    if table is not None:
        _head_to_table(model, a1, d["even_out"], table)
        return None, st
    prob1 = _head3(model, "prob_pred_mlp1", a1)
    return prob1[d["even_rows"]], st


@torch.no_grad()
def ehem_phase2_prepare(model, st, plan):
    """The part of phase 2 that does NOT depend on the even nodes' occupancies (ehem.py:117-127): pre_attn_mlp(a1) and the query stream of the
    cross transformer - LayerNorm + query projection of every block, patch merging of the queries between the stages.  The decoder runs it on
    a side stream while the host range-decodes the even symbols (the GPU is idle then); same kernels on the same rows as inside
    ehem_phase2_packed, so the bits are the same.  -> dict(pre = [Q0, 16 + 240] with the pre_attn columns filled, q[s][b] = query projections).
    A preparation (or a window's rows of it) is CONSUMED by the phase 2 that takes it: the cross blocks run in place on `pre`."""
    d = plan.d
    a1, a2 = st["a1"], st["a2"]
    no = model.pre_occ_mlp[4].weight.shape[0]
    pre = torch.empty((a2.shape[0], no + model.pre_attn_mlp[4].weight.shape[0]), dtype=torch.float32, device=a2.device)
    _head3(model, "pre_attn_mlp", a1, out=pre[:, no:])
    enc = model.swin_cross_transformer
    qs, query = [], a2
    for s, stage in enumerate(enc.layers):
        v1 = d["cross_valid"][s].reshape(-1)
        row = []
        for blk in stage.blocks:
            w, _, qbias, eps_b, _ = _block_consts(blk, True)
            row.append(native.swin_ln_linear(query, w["q"], qbias, eps_b, v1))
        qs.append(row)
        if s < len(enc.layers) - 1:
            query = _merge(stage.downsample, query, d["cross_merge"][s])
    # the odd tokens' half of prob_pred_mlp2's first operand (ehem.py:124: cat(concat_states, a2)) as planes, in its columns of the [Q0, 512] operand
    a0 = native.SplitAct.empty(a2.shape[0], 512, a2.device)
    native.split_rows(a2, out=a0.cols(256, 512))
    return dict(pre=pre, q=qs, a0=a0)


def phase2_prep_window(prep, bases, rows):
    """The rows of ONE window inside a level-wide ehem_phase2_prepare result: bases[s] / rows[s] = first row / padded row count of the window in
    cross stage s (a one-window plan has exactly these rows)."""
    return dict(pre=prep["pre"][bases[0]:bases[0] + rows[0]], q=[[q[bases[s]:bases[s] + rows[s]] for q in qs] for s, qs in enumerate(prep["q"])],
                a0=native.SplitAct(prep["a0"].t[:, bases[0]:bases[0] + rows[0]], prep["a0"].K))


@torch.no_grad()
def ehem_phase2_packed(model, st, plan, pre_occ=None, table=None, prep=None):
    """Odd-node logits given the occupancies of the even nodes (ehem.py:117-127).  pre_occ: int64 [Q0 rows] in the cross
    layout (None = the true occupancies taken from ctx, as the encoder does).  prep: the result of ehem_phase2_prepare for these rows."""
    d = plan.d
    g = model.geo_feat_generator
    a1, a2 = st["a1"], st["a2"]
    po = st["pre_occ"] if pre_occ is None else pre_occ
    # pre_occ_mlp(embed_occ(symbol)) (ehem.py:117-119) is a function of the symbol alone: a 256-row table, made once per weight by the same exact
    # fp32 kernels (their rows do not depend on the batch: same bits as evaluating the three layers on the gathered embeddings), then one gather per
    # call instead of an embedding lookup and three dense launches - the decoder calls this once per window
    lut = derived(model, "pre_occ_lut", [g.occ_enc.weight] + [model.pre_occ_mlp[i].weight for i in (0, 2, 4)] + [model.pre_occ_mlp[i].bias for i in (0, 2, 4)],
                  lambda: leaky_mlp3(model.pre_occ_mlp, g.occ_enc.weight.detach()).contiguous())
    no = lut.shape[1]
    if prep is not None:
        pre = prep["pre"]
    else:
        pre = torch.empty((a2.shape[0], no + model.pre_attn_mlp[4].weight.shape[0]), dtype=torch.float32, device=a2.device)
        _head3(model, "pre_attn_mlp", a1, out=pre[:, no:])
    native.gather_rows(lut, po, pre[:, :no])                # the table's rows straight into their columns of `pre`
    hc = _encoder(model.swin_cross_transformer, pre, d["cross_valid"], d["cross_tab"], d["cross_merge"], query=a2, tiles=d.get("cross_tiles"),
                  q_pre=None if prep is None else prep["q"])
    if table is not None:
        a = _concat_layer(model.prob_pred_mlp2[0], hc, d["cross_parent"], a2, d["cross_concat"][1], None if prep is None else prep.get("a0")) if HIER else linear_s(_concat(hc, d["cross_concat"], extra=(a2, None)), model.prob_pred_mlp2[0].weight, model.prob_pred_mlp2[0].bias, act="leaky", want="split")
        from ..ops import _split
        a = linear_s(a, model.prob_pred_mlp2[2].weight, model.prob_pred_mlp2[2].bias, act="leaky", want="split")
        native.linear_split_scatter(a, _split(model.prob_pred_mlp2[4].weight), model.prob_pred_mlp2[4].bias, d["odd_out"], table)
        return None
    prob2 = (_mlp_over_concat(model.prob_pred_mlp2, hc, d["cross_parent"], extra=a2, grand=d["cross_concat"][1], a0_pre=None if prep is None else prep.get("a0")) if HIER
             else leaky_mlp3_s(model.prob_pred_mlp2, _concat(hc, d["cross_concat"], extra=(a2, None))))
    return prob2[d["odd_rows"]]


@torch.no_grad()
def ehem_forward_packed(model, ctx, pos, plan, table=None):
    """ctx uint8/int64 [T,12], pos float32 [T,3]: the frame's tokens, windows back to back (lengths = plan.c).
    Returns (logits_even_rows [sum ceil(c/2), 255], logits_odd_rows [sum floor(c/2), 255]) in window order."""
    ev, st = ehem_phase1_packed(model, ctx, pos, plan, table=table)
    return ev, ehem_phase2_packed(model, st, plan, table=table)
