"""Stage M on the GPU: HIP kernels and the product modules against the reference's golden logits (tolerance 1e-3,
BASELINE.json north_star) and against the CPU oracle."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cfgs import ehem_cfg, octattn_cfg
from conftest import GOLDEN, golden, parity_record

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from scp_amd import native
    native.lib()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ehem(dev):
    from scp_amd.models import EHEM
    from scp_amd.weights import fill_weights
    m = EHEM(ehem_cfg())
    fill_weights(m, 0)
    return m.to(dev)


@pytest.fixture(scope="module")
def octattn(dev):
    from scp_amd.models import OctAttention
    from scp_amd.weights import fill_weights
    m = OctAttention(octattn_cfg())
    fill_weights(m, 0)
    return m.to(dev)


# ----------------------------------------------------------------------------------------------- kNN
def ref_knn_values(x, k):
    """dgcnn.py:18-20 distances in float64 (ground truth for 'is this a valid top-k set')."""
    x = x.double()
    inner = x @ x.transpose(1, 2)
    xx = (x ** 2).sum(2)
    return 2 * inner - xx[:, None, :] - xx[:, :, None]


@pytest.mark.parametrize("B,n,C", [(1, 1, 3), (1, 7, 3), (2, 300, 3), (1, 1000, 144), (1, 2500, 192), (1, 8192, 3)])
def test_knn_topk_is_a_valid_topk(dev, B, n, C):
    from scp_amd import native
    g = torch.Generator().manual_seed(n * 7 + C)
    if C == 3:   # lattice positions like the real input (many exact ties)
        x = torch.randint(0, 60, (B, n, C), generator=g).float() / 59.0
    else:
        x = torch.randn((B, n, C), generator=g)
    k = min(20, n)
    idx = native.knn_topk(x.to(dev), k).cpu().long()
    assert idx.shape == (B, n, k) and idx.min() >= 0 and idx.max() < n
    # no duplicates inside a row
    assert (torch.sort(idx, 2)[0].diff(dim=2) != 0).all() if k > 1 else True
    d = ref_knn_values(x, k)
    got = torch.gather(d, 2, idx)
    kth = torch.topk(d, k, dim=2)[0][..., -1:]
    # every returned neighbour is at least as close as the true k-th neighbour (up to float32 rounding of the distance)
    tol = 4e-6 * d.abs().max()
    assert (got >= kth - tol).all()
    # sorted by decreasing value
    assert (got.diff(dim=2) <= tol).all()


@pytest.mark.parametrize("n,C,span", [(20, 3, 2), (33, 3, 3), (1000, 3, 6), (8192, 3, 12), (2500, 144, 2), (4096, 192, 2), (300, 192, 1)])
def test_knn_exact_order_under_heavy_ties(dev, n, C, span):
    """Small-integer features: every product and sum is exact in fp32 (and in the f16x3 split), so the distances are exact and
    massively tied (duplicates included).  The result must be THE top-20 under (distance ascending, index ascending) - this
    pins the pruning bound (union of the two half-lists), the cut of pass 1 (ties must pass) and the final merge."""
    from scp_amd import native
    g = torch.Generator().manual_seed(n + C)
    x = torch.randint(0, span + 1, (1, n, C), generator=g).float()
    k = min(20, n)
    got = native.knn_topk(x.to(dev), k).cpu().long()[0]
    xi = x[0].long()
    d = ((xi[:, None, :] - xi[None, :, :]) ** 2).sum(-1) if n <= 2500 else None
    if d is None:   # blockwise exact integer distances
        d = torch.empty((n, n), dtype=torch.int64)
        for i0 in range(0, n, 512):
            d[i0:i0 + 512] = ((xi[i0:i0 + 512, None, :] - xi[None, :, :]) ** 2).sum(-1)
    key = d * n + torch.arange(n)[None, :]                            # distance major, index minor: unique keys
    want = torch.topk(key, k, dim=1, largest=False)[1]
    assert torch.equal(got, want), (got[:2], want[:2])


def test_knn_packed_workgroup_shapes_agree_on_ragged_windows(dev):
    """The packed f16x3 search in its workgroup shapes (256 queries on the XCD schedule with a barrier per group of 3 - 4 / 2 / 1 tiles,
    front-to-back or outward sweep; 128 queries in launch order) and the dense per-window entry point return the SAME lists: ragged windows (1 ... 8192 rows), heavily tied
    integer features, 144 and 192 features."""
    from scp_amd import native
    lengths = [1, 7, 20, 33, 300, 513, 2049, 8192, 600, 5000]
    g = torch.Generator().manual_seed(11)
    for C in (144, 192):
        rows = sum(-(-n // 512) * 512 for n in lengths)
        x = torch.zeros((rows, C))
        tab, base = [], 0
        for n in lengths:
            x[base:base + n] = torch.randint(0, 3, (n, C), generator=g).float()
            tab += [[base, n]] * (-(-n // 512))
            base += -(-n // 512) * 512
        xd, td = x.to(dev), torch.tensor(tab, dtype=torch.int32, device=dev)
        outs = {}
        try:
            for sh in (128, 258, 257, 256, 272):
                native.set_knn_workgroup(sh)
                outs[sh] = native.knn_topk_packed(xd, td).cpu()
        finally:
            native.set_knn_workgroup(256)
        base = 0
        for n in lengths:
            k = min(20, n)
            want = native.knn_topk(xd[base:base + n][None].contiguous(), k).cpu()[0].long() + base
            for sh, o in outs.items():
                got = o[base:base + n].long()
                assert torch.equal(got[:, :k], want), (C, n, sh)
                if k < 20:
                    assert (got[:, k:] == got[:, :1]).all()
            base += -(-n // 512) * 512


@pytest.mark.parametrize("lengths", [[8192], [2049], [300], [33], [8192, 4000], [700, 700, 700, 20], [5000] * 6])
def test_knn_short_launches_split_the_candidate_sweep_with_identical_lists(dev, lengths):
    """Round 5 (decoder latency): a packed search with too few 256-query blocks for the chip cuts every block's candidate sweep into 2 - 16 runs
    (one workgroup each) and merges the partial lists under the lists' own total order (value descending, index ascending) - the lists equal
    the single-sweep kernel's (workgroup shape 257 never splits) and the dense per-window entry point's, heavily tied features included."""
    from scp_amd import native
    g = torch.Generator().manual_seed(len(lengths) * 1000 + lengths[0])
    for C, ties in ((144, True), (192, False), (192, True)):
        rows = sum(-(-n // 512) * 512 for n in lengths)
        x = torch.zeros((rows, C))
        tab, base = [], 0
        for n in lengths:
            x[base:base + n] = torch.randint(0, 3, (n, C), generator=g).float() if ties else torch.randn((n, C), generator=g)
            tab += [[base, n]] * (-(-n // 512))
            base += -(-n // 512) * 512
        xd, td = x.to(dev), torch.tensor(tab, dtype=torch.int32, device=dev)
        try:
            native.set_knn_workgroup(256)
            split = native.knn_topk_packed(xd, td).cpu()
            native.set_knn_workgroup(257)
            single = native.knn_topk_packed(xd, td).cpu()
        finally:
            native.set_knn_workgroup(256)
        base = 0
        for n in lengths:
            assert torch.equal(split[base:base + n], single[base:base + n]), (C, ties, n)
            k = min(20, n)
            want = native.knn_topk(xd[base:base + n][None].contiguous(), k).cpu()[0].long() + base
            assert torch.equal(split[base:base + n, :k].long(), want), (C, ties, n)
            base += -(-n // 512) * 512


def test_knn_packed_many_short_windows(dev):
    """More sequences than the XCD schedule lists (2048): the launch falls back to the 128-query kernel; and a launch of 1500 one-chunk
    windows on the schedule (every run holds far more blocks than its even share of slots: the spill region takes them)."""
    from scp_amd import native
    g = torch.Generator().manual_seed(5)
    for nwin in (1500, 2100):
        C = 144
        lens = torch.randint(1, 60, (nwin,), generator=g)
        x = torch.zeros((nwin * 512, C))
        tab = []
        for w in range(nwin):
            n = int(lens[w])
            x[w * 512:w * 512 + n] = torch.randint(0, 3, (n, C), generator=g).float()
            tab.append([w * 512, n])
        xd, td = x.to(dev), torch.tensor(tab, dtype=torch.int32, device=dev)
        got = native.knn_topk_packed(xd, td).cpu().long()
        for w in (0, 1, nwin // 2, nwin - 1):
            n = int(lens[w]); k = min(20, n)
            want = native.knn_topk(xd[w * 512:w * 512 + n][None].contiguous(), k).cpu()[0].long() + w * 512
            assert torch.equal(got[w * 512:w * 512 + n, :k], want), (nwin, w)


def test_knn_scratch_survives_many_streams(dev):
    """The per-stream kNN scratch table has 8 slots: a ninth stream takes over the least recently used one (encoders come and go in a
    long-running process); results stay those of the default stream."""
    from scp_amd import native
    g = torch.Generator().manual_seed(2)
    x = torch.randn((1024, 144), generator=g).to(dev)
    tab = torch.tensor([[0, 600], [0, 600]], dtype=torch.int32, device=dev)
    want = native.knn_topk_packed(x, tab).cpu()
    streams = [torch.cuda.Stream(device=dev) for _ in range(12)]
    torch.cuda.synchronize()
    for rnd in range(2):
        for st in streams:
            with torch.cuda.stream(st):
                got = native.knn_topk_packed(x, tab)
            st.synchronize()
            assert torch.equal(got.cpu()[:600], want[:600])


def test_knn_matches_cpu_reference_topk_on_real_window(dev):
    from scp_amd import native
    from oracle import models_ref
    z = golden("logits_ehem_c1024")
    pos = torch.from_numpy(z["pos"])[None]          # [1,3,c]
    ref = models_ref.knn(pos, 20)                   # CPU torch, reference formula
    got = native.knn_topk(pos.transpose(1, 2).contiguous().to(dev), 20).cpu().long()
    same = (torch.sort(ref, 2)[0] == torch.sort(got, 2)[0]).all(2).float().mean().item()
    print(f"kNN neighbour sets identical to CPU torch.topk for {100 * same:.2f}% of the points")
    # lattice positions: every differing point must be a TIE at the 20th / 21st neighbour, i.e. a point where the reference's own
    # CPU and CUDA top-k disagree as well.  Squared lattice distances are integers (in lattice units) of a few tens at the 20th
    # neighbour: distinct ones differ by >= 2 %, tied ones by the float32 rounding of the normalised coordinates (<= 1e-4).
    x = pos[0].T.double()
    d = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    srt = torch.sort(d, 1)[0]
    tie = (srt[:, 19] - srt[:, 20]).abs() <= 1e-3 * srt[:, 20].abs().clamp_min(1e-30)
    differs = ~(torch.sort(ref, 2)[0] == torch.sort(got, 2)[0]).all(2)[0]
    parity_record("knn_pos_lattice_c1024", sets_identical=same, differing_points=int(differs.sum()), differing_not_tied=int((differs & ~tie).sum()))
    assert not (differs & ~tie).any()
    assert same > 0.95      # measured 0.969: 32 of 1024 points differ, all of them exact ties


# ----------------------------------------------------------------------------------------------- edge conv
@pytest.mark.parametrize("rows,C", [(1, 600), (7, 600), (4099, 600), (33, 256), (5, 1024), (9, 12)])
def test_layernorm_add_vs_torch(dev, rows, C):
    """norm(x + residual) of attention_model.py:117,123 in one pass, any row width."""
    from scp_amd import native
    g = torch.Generator().manual_seed(rows + C)
    a, b = torch.randn((rows, C), generator=g) * 30, torch.randn((rows, C), generator=g)
    gm, bt = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    want = torch.nn.functional.layer_norm((a + b).double(), (C,), gm.double(), bt.double(), 1e-5)
    got = native.layernorm_add(a.to(dev), b.to(dev), gm.to(dev), bt.to(dev)).cpu()
    assert (got.double() - want).abs().max() < 2e-5
    got1 = native.layernorm_add(a.to(dev), None, gm.to(dev), bt.to(dev)).cpu()
    want1 = torch.nn.functional.layer_norm(a.double(), (C,), gm.double(), bt.double(), 1e-5)
    assert (got1.double() - want1).abs().max() < 2e-5
    with pytest.raises(native.ScpError):
        native.layernorm_add(a.to(dev)[:, :C - 1], None, gm.to(dev), bt.to(dev))


def test_edge_gather_max_vs_torch(dev):
    from scp_amd import native
    g = torch.Generator().manual_seed(3)
    B, n, Co, k = 2, 777, 128, 20
    u = torch.randn((B, n, Co), generator=g)
    v = torch.randn((B, n, Co), generator=g)
    idx = torch.randint(0, n, (B, n, k), generator=g, dtype=torch.int32)
    scale = torch.randn(Co, generator=g)
    shift = torch.randn(Co, generator=g)
    got = native.edge_gather_max(u.to(dev), v.to(dev), idx.to(dev), scale.to(dev), shift.to(dev)).cpu()
    nb = torch.stack([u[b][idx[b].long()] for b in range(B)])           # [B,n,k,Co]
    y = torch.nn.functional.leaky_relu(scale * (nb + v[:, :, None]) + shift, 0.2).max(2)[0]
    assert torch.allclose(got, y, atol=1e-6, rtol=1e-6)


def gpu_knn_for_oracle(dev):
    """Neighbour indices chosen by the HIP kernel, handed to the CPU oracle (isolates everything but tie-breaking)."""
    from scp_amd import native

    def f(x, k):
        return native.knn_topk(x.transpose(1, 2).contiguous().to(dev), k).cpu().long()
    return f


class _KnnLists:
    """Records the neighbour lists of every kNN call of a GPU forward (both entry points) and replays them, in call order, to the CPU
    oracle as its KNN_OVERRIDE: the oracle then computes with EXACTLY the neighbour sets the product used.  (gpu_knn_for_oracle above
    runs the HIP search on the ORACLE's features instead - the same sets only as long as no two candidates lie within the float noise of
    the two feature computations; the full-size Ford-like window has such pairs.)"""

    def __init__(self, monkeypatch):
        from scp_amd import native
        self.lists = []
        k1, k2 = native.knn_topk, native.knn_topk_packed

        self.inputs = []          # what every search was asked: (kind, features on the host, k or window table)

        def topk(x, k):
            idx = k1(x, k)
            self.lists.append(("window", idx.cpu().long()))
            self.inputs.append(("window", x.detach().cpu(), k))
            return idx

        def topk_packed(x, ctab, thr0=None):
            idx = k2(x, ctab, thr0)
            self.lists.append(("packed", idx.cpu().long()))
            self.inputs.append(("packed", x.detach().cpu(), ctab.cpu()))
            return idx

        monkeypatch.setattr(native, "knn_topk", topk)
        monkeypatch.setattr(native, "knn_topk_packed", topk_packed)

    def check(self, max_windows=3):
        """The recorded lists are NEAREST-neighbour lists of the features each search was given - checked independently of the kernels, by a
        float64 brute force on the host (replaying the product's lists into the oracle cannot notice a wrong list; this can): for
        every query the farthest listed neighbour is no farther than the true k-th nearest (up to float32 rounding of the distance: exact ties
        and candidates within rounding may be exchanged), and a list holds no duplicates."""
        def one(feat, idx, k):
            f = feat.double()
            n = f.shape[0]
            kk = min(k, n)
            d = (f * f).sum(1)[:, None] + (f * f).sum(1)[None, :] - 2.0 * (f @ f.T)
            kth = torch.topk(d, kk, dim=1, largest=False)[0][:, -1]
            got = torch.gather(d, 1, idx[:, :kk])
            scale = (f * f).sum(1).max().clamp_min(1e-30)
            assert (got.max(1)[0] <= kth + 4e-6 * scale).all(), "a listed neighbour is farther than the k-th nearest"
            srt = torch.sort(idx[:, :kk], 1)[0]
            assert (srt[:, 1:] != srt[:, :-1]).all() or kk < 2, "duplicate neighbour in a list"
        for (kind, idx), (_, x, aux) in zip(self.lists, self.inputs):
            if kind == "window":
                for b in range(min(x.shape[0], max_windows)):
                    one(x[b], idx[b], aux)
            else:
                seqs = sorted({(int(b), int(n)) for b, n in aux.tolist() if n > 0})
                for base, n in seqs[:max_windows]:
                    one(x[base:base + n], idx[base:base + n] - base, idx.shape[1])

    def replay(self, B, c):
        """-> a KNN_OVERRIDE for models_ref.ehem_forward on B windows of c nodes (ce = c rounded up to even, ehem.py:92-99)."""
        ce = c + (c & 1)
        cp = -(-ce // 512) * 512
        calls = iter(self.lists)

        def f(x, k):
            kind, idx = next(calls)
            if kind == "window":
                return idx[:, :ce, :k]
            return torch.stack([idx[b * cp:b * cp + ce, :k] - b * cp for b in range(B)])      # (windows shorter than 20 nodes ask for k = n)
        return f


def test_edge_conv_matches_oracle_formulation(dev, ehem):
    """split-GEMM + gather/max == cat(f_j - f_i, f_i) conv + BN + LeakyReLU + max (dgcnn.py:48-71), same neighbours."""
    from oracle import models_ref
    from scp_amd.models.ehem import _edge_conv
    sd = {k: v.cpu() for k, v in ehem.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 500, 144), generator=g)
    models_ref.KNN_OVERRIDE = gpu_knn_for_oracle(dev)
    try:
        want = models_ref.edge_conv(sd, "geo_feat_generator.conv2", x.transpose(1, 2).contiguous(), 20).transpose(1, 2)
    finally:
        models_ref.KNN_OVERRIDE = None
    got = _edge_conv(ehem.geo_feat_generator.conv2, x.to(dev), 20).cpu()
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-5)


# ----------------------------------------------------------------------------------------------- dense layers
@pytest.mark.parametrize("M,N,K", [(1, 255, 256), (7, 16, 32), (300, 768, 256), (1000, 256, 1024), (4100, 255, 512), (513, 240, 80)])
@pytest.mark.parametrize("act", [None, "leaky", "gelu"])
def test_linear_bf16x3_vs_float64(dev, M, N, K, act):
    """bf16x3 split GEMM: fp32-class accuracy (the three-product split drops only the 2^-16 lo.lo term)."""
    from scp_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g)
    w = torch.randn((N, K), generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn((M, N), generator=g)
    y = ops.linear(x.to(dev), w.to(dev), b.to(dev), act=act, residual=r.to(dev)).cpu().double()
    ref = x.double() @ w.double().T + b.double()
    if act == "leaky":
        ref = torch.nn.functional.leaky_relu(ref, 0.01)
    elif act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    ref = ref + r.double()
    err = (y - ref).abs().max().item()
    f32 = (torch.nn.functional.linear(x, w, b).double() - (x.double() @ w.double().T + b.double())).abs().max().item()
    print(f"M={M} N={N} K={K} act={act}: bf16x3 err {err:.2e} (plain fp32 CPU matmul err {f32:.2e})")
    assert err < 3e-5, err


def _library_linear(x, w, b=None, act=None, residual=None, exact=False, precise=False, scales=None):
    """The test bracket that used to live in the product as SCP_GEMM=f32: the same layer on the fp32 library GEMM (rocBLAS / hipBLASLt)."""
    y = F.linear(x, w, b)
    if act == "leaky":
        y = F.leaky_relu(y, 0.01)
    elif act == "gelu":
        y = F.gelu(y)
    elif act == "relu":
        y = torch.relu(y)
    return y if residual is None else y + residual


def test_ehem_logits_bf16x3_vs_fp32_library_gemm(dev, ehem, monkeypatch):
    """Same window through the product's dense layers (bf16x3 split on bf16 MFMA) and through fp32 library GEMMs patched in by this test:
    the split changes the logits by far less than the 1e-3 tolerance."""
    from scp_amd import ops
    from scp_amd.models import ehem as ehem_mod
    z = golden("logits_ehem_c1024")
    data = torch.from_numpy(z["data"].astype(np.int64))[None].to(dev)
    pos = torch.from_numpy(z["pos"])[None].to(dev)
    b1, b2 = ehem(data, pos)
    with monkeypatch.context() as m:
        m.setattr(ops, "linear", _library_linear)
        m.setattr(ehem_mod, "linear", _library_linear)
        a1, a2 = ehem(data, pos)
    d = max((a1 - b1).abs().max().item(), (a2 - b2).abs().max().item())
    print(f"bf16x3 vs fp32 library GEMMs: max|dlogit| = {d:.3e}")
    assert 0 < d < 2e-4, d


def test_dense_layers_refuse_host_tensors():
    """One backend: a tensor that is not on the device raises (no F.linear fallback in the product)."""
    from scp_amd import native, ops
    with pytest.raises(native.ScpError):
        ops.linear(torch.zeros(4, 32), torch.zeros(8, 32))


# ----------------------------------------------------------------------------------------------- Swin
@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "swin_*_s*_L*.npz"))))
def test_swin_layer_vs_reference(dev, name):
    from scp_amd.models.ehem import SwinLayer, swin_layer_forward
    from scp_amd.weights import fill_weights
    z = golden(name)
    _, kind, s, L = name.split("_")
    shift, L = int(s[1:]), int(L[1:])
    layer = SwinLayer()
    fill_weights(layer, int(z["wseed"]))
    layer = layer.to(dev)
    rng = np.random.default_rng(int(z["x_seed"]))
    x = torch.from_numpy(rng.standard_normal((1, L, 256), dtype=np.float32)).to(dev)
    q = torch.from_numpy(rng.standard_normal((1, L, 256), dtype=np.float32)).to(dev)
    with torch.no_grad():
        y = swin_layer_forward(layer, x, L, shift, q if kind == "cross" else None)
    err = np.abs(y[0, ::int(z["stride"])].cpu().numpy() - z["y"]).max()
    assert err < 1e-4, err


def test_attention_bf16x3_vs_fp32_mfma(dev):
    """The two numerics of the window-attention kernel on the same inputs (dense and packed entry points)."""
    from scp_amd import native
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn((2, 1024, 768), generator=g).to(dev)
    tab = (torch.randn((1023, 4), generator=g) * 0.5).to(dev)
    q, k, v = qkv[..., :256], qkv[..., 256:512], qkv[..., 512:]
    try:
        for shift in (0, 256):
            native.set_attention_mode(False)
            a = native.swin_attention(q, k, v, tab, shift)
            native.set_attention_mode(True)
            b = native.swin_attention(q, k, v, tab, shift)
            err = (a - b).abs().max().item()
            print(f"shift {shift}: bf16x3 vs fp32 attention max diff {err:.2e} (|out| max {a.abs().max().item():.2f})")
            assert err < 5e-5
            # packed entry point == dense entry point
            wtab = torch.tensor([[0, 1024], [0, 1024], [1024, 1024], [1024, 1024]], dtype=torch.int32, device=dev)
            c = native.swin_attention_packed(q.reshape(2048, 256), k.reshape(2048, 256), v.reshape(2048, 256), tab, wtab, shift)
            assert torch.equal(c.reshape(2, 1024, 256), b)
    finally:
        native.set_attention_mode(True)


def test_swin_attention_argument_errors(dev):
    from scp_amd import native
    q = torch.zeros((1, 500, 256), device=dev)
    with pytest.raises(native.ScpError):
        native.swin_attention(q, q, q, torch.zeros((1023, 4), device=dev), 0)     # Lp not a multiple of 512


# ----------------------------------------------------------------------------------------------- EHEM
def _ehem_case(z):
    data = torch.from_numpy(z["data"].astype(np.int64))
    pos = torch.from_numpy(z["pos"])
    if data.dim() == 3:
        data, pos = data[None], pos[None]
    return data, pos


def _want_rows(z):
    """(stride, even rows, odd rows) of a logits fixture, batch axis first."""
    if "out1_sub" in z:
        return int(z["stride"]), z["out1_sub"][None], z["out2_sub"][None]
    st = int(z["stride"]) if "stride" in z else 1
    w1, w2 = z["out1"], z["out2"]
    if w1.ndim == 2:
        w1, w2 = w1[None], w2[None]
    return st, w1, w2


def _run_packed(ehem, data, pos, dev):
    """The product path (FrameEncoder / decoder): every batch entry is one window of ONE packed forward."""
    B, c = data.shape[:2]
    ctx = data.reshape(B * c, 12).to(torch.uint8).to(dev)
    p = pos.transpose(1, 2).reshape(B * c, 3).contiguous().to(dev)
    ev, od = ehem.forward_packed(ctx, p, [c] * B)
    return ev.reshape(B, (c + 1) // 2, 255).cpu().numpy(), od.reshape(B, c // 2, 255).cpu().numpy()


def _row_err(o1, o2, st, w1, w2):
    e1 = np.abs(o1[:, ::st] - w1).reshape(-1, 255)
    e2 = np.abs(o2[:, ::st] - w2).reshape(-1, 255)
    return np.concatenate([e1, e2]) if e2.size else e1


class _KnnSpy:
    """Records (features, neighbour lists) of every kNN call of a forward (both entry points)."""

    def __init__(self, monkeypatch):
        from scp_amd import native
        self.calls = []
        k1, k2 = native.knn_topk, native.knn_topk_packed

        def topk(x, k):
            idx = k1(x, k)
            self.calls.append((x[0].detach().cpu(), idx[0].cpu().long()))
            return idx

        def topk_packed(x, ctab, thr0=None):
            idx = k2(x, ctab, thr0)
            self.calls.append((x.detach().cpu(), idx.cpu().long()))
            return idx

        monkeypatch.setattr(native, "knn_topk", topk)
        monkeypatch.setattr(native, "knn_topk_packed", topk_packed)


def _knn_sets_vs_reference(feat, idx, want_sorted, c):
    """Fraction of points whose neighbour SET equals the reference's; every other point must be a near-tie: all exchanged
    candidates lie within fp32 rounding of the 20th-best distance (float64 distances of the features the kernel saw)."""
    want = want_sorted.astype(np.int64)
    c = want.shape[0]                 # odd windows: the reference searched the padded window (pad token at position 0, ehem.py:92-99)
    got = np.sort(idx[:c].numpy().astype(np.int64), axis=1)
    bad = np.where((got != want).any(1))[0]
    worst = 0.0
    x = feat[:c].double()
    sq = (x * x).sum(1)
    for i in bad:
        d = sq[i] + sq - 2 * (x @ x[i])                       # float64 squared distances to every candidate
        kth = torch.sort(d)[0][19]
        swapped = np.setxor1d(got[i], want[i])
        scale = float(sq[i] + sq.max())
        worst = max(worst, float((d[swapped] - kth).abs().max()) / scale)
    return 1.0 - len(bad) / c, worst


TIEFREE = ["tiefree_ehem_c600", "tiefree_ehem_c2049", "tiefree_ehem_c8192"]


@pytest.mark.parametrize("path", ["window", "packed"])
@pytest.mark.parametrize("name", TIEFREE)
def test_ehem_tiefree_every_row_vs_reference(dev, ehem, monkeypatch, name, path):
    """The strict pin of a11/a12 against the reference itself (models/ehem.py:88-136, dgcnn.py:10-45): random float positions,
    so no neighbour distance is exactly tied and EVERY logit row of the reference is a hard target (1e-3, BASELINE.json) on the
    default arithmetic (f16x3 feature kNN, bf16x3 dense layers and attention) - through the per-window forward and through the
    packed forward the encoder / decoder use.  The neighbour SETS of the three searches must be the reference's; a point may
    differ only by candidates within fp32 rounding of its 20th-best distance."""
    z = golden(name)
    data, pos = _ehem_case(z)
    c = data.shape[1]
    spy = _KnnSpy(monkeypatch)
    if path == "window":
        o1, o2 = ehem(data.to(dev), pos.to(dev), enc=True)
        o1, o2 = o1.cpu().numpy(), o2.cpu().numpy()
    else:
        o1, o2 = _run_packed(ehem, data, pos, dev)
    st, w1, w2 = _want_rows(z)
    e = _row_err(o1, o2, st, w1, w2)
    assert len(spy.calls) == 3
    rec = dict(max_dlogit=e.max(), rows=e.shape[0], rows_within_1e3=(e.max(1) <= LOGIT_TOL).mean())
    for i, (feat, idx) in enumerate(spy.calls):
        same, worst = _knn_sets_vs_reference(feat, idx, z[f"knn{i}"], c)
        rec[f"knn{i}_sets_identical"] = same
        rec[f"knn{i}_worst_exchange_rel"] = worst
        # search 0 is the exact fp32 chain on tie-free positions: the reference's sets, all of them
        assert same == 1.0 if i == 0 else same >= 0.995, (i, same)
        assert worst <= 4e-6, (i, worst)
    parity_record(f"{name}/{path}", **rec)
    print(name, path, rec)
    assert e.max() <= LOGIT_TOL, rec


def test_ehem_tiefree_exact_knn_mode(dev, ehem, monkeypatch):
    """The same pin under `scp_set_knn_mode(0)` (all three searches on the exact k-ordered fp32 chain): the reference's neighbour sets
    for every point of every search, every row within 1e-3."""
    from scp_amd import native
    z = golden("tiefree_ehem_c2049")
    data, pos = _ehem_case(z)
    spy = _KnnSpy(monkeypatch)
    native.set_knn_mode(False)
    try:
        o1, o2 = _run_packed(ehem, data, pos, dev)
    finally:
        native.set_knn_mode(True)
    st, w1, w2 = _want_rows(z)
    e = _row_err(o1, o2, st, w1, w2)
    for i, (feat, idx) in enumerate(spy.calls):
        same, worst = _knn_sets_vs_reference(feat, idx, z[f"knn{i}"], data.shape[1])
        assert same >= 0.999 and worst <= 4e-6, (i, same, worst)
    parity_record("tiefree_ehem_c2049/packed, exact fp32 kNN", max_dlogit=e.max(), rows_within_1e3=(e.max(1) <= LOGIT_TOL).mean())
    assert e.max() <= LOGIT_TOL


# fixtures on which no tie at rank 20 / 21 changes a row beyond the tolerance: there the bound against the reference's golden logits is ASSERTED
# (logits_ehem_c8192: 9.3e-4 on sub-sampled rows of a full window, measured in rounds 3 - 5); on the others it is recorded and every row beyond
# it must be explained by a tie (the two tests at the end of this file)
EVERY_ROW_VS_REFERENCE = {"logits_ehem_c1", "logits_ehem_c7", "logits_ehem_lvl1_c6", "logits_ehem_c8192"}


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "logits_ehem_*.npz"))))
def test_ehem_logits_vs_reference_packed_path(dev, ehem, monkeypatch, name):
    """The lattice-position goldens through forward_packed (the kernels the bench / CLI / decoder run: layernorm_rows,
    gemm_split, mlp_fused, hierarchical concat) with the same two assertions as the per-window test below."""
    z = golden(name)
    data, pos = _ehem_case(z)
    rec = _KnnLists(monkeypatch)
    o1, o2 = _run_packed(ehem, data, pos, dev)
    rec.check()
    st, w1, w2 = _want_rows(z)
    e = _row_err(o1, o2, st, w1, w2)
    rows_ok = (e.max(1) <= LOGIT_TOL).mean()
    from oracle import models_ref
    sd = {k: v.cpu() for k, v in ehem.state_dict().items()}
    models_ref.KNN_OVERRIDE = rec.replay(data.shape[0], data.shape[1])      # the oracle with the neighbour lists this forward used
    try:
        with torch.no_grad():
            r1, r2 = models_ref.ehem_forward(sd, data, pos)
    finally:
        models_ref.KNN_OVERRIDE = None
    d = max(np.abs(o1 - r1.numpy()).max(), np.abs(o2 - r2.numpy()).max() if o2.size else 0.0)
    parity_record(f"{name}/packed", max_dlogit_vs_reference=e.max(), rows_within_1e3_vs_reference=rows_ok, max_dlogit_vs_oracle_same_knn=d)
    if name in EVERY_ROW_VS_REFERENCE:
        assert e.max() <= LOGIT_TOL, e.max()
    # (the fraction is a recorded number, not a criterion: test_every_out_of_tolerance_lattice_row_is_explained_by_a_knn_tie and
    # test_given_the_reference_neighbour_choice_every_lattice_row_matches are the proof)
    assert d <= LOGIT_TOL, d


# measured fractions of rows within 1e-3 of the reference's golden logits (lattice positions: exact distance ties at the 20th /
# 21st neighbour are resolved by the reference's top-k implementation) minus a small margin; profiles/parity_r2.json has the values
ROWS_OK_MIN = {"logits_ehem_b2_c256": 0.985,   # (no longer asserted: kept as the record of what round 2 measured)
                "logits_ehem_c1024": 0.994, "logits_ehem_c600": 0.99, "logits_ehem_c1": 1.0, "logits_ehem_c7": 1.0,
               "logits_ehem_c8192": 1.0, "logits_ehem_lvl1_c6": 1.0}      # measured: 0.9902 / 0.9971 / 0.9933 / 1 / 1 / 1 / 1

@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "logits_ehem_*.npz"))))
def test_ehem_logits_vs_reference(dev, ehem, monkeypatch, name):
    z = golden(name)
    data = torch.from_numpy(z["data"].astype(np.int64))
    pos = torch.from_numpy(z["pos"])
    if data.dim() == 3:
        data, pos = data[None], pos[None]
    rec = _KnnLists(monkeypatch)
    o1, o2 = ehem(data.to(dev), pos.to(dev), enc=True)
    rec.check()
    o1, o2 = o1.cpu().numpy(), o2.cpu().numpy()
    if "out1_sub" in z:
        st = int(z["stride"])
        e1 = np.abs(o1[0, ::st] - z["out1_sub"])
        e2 = np.abs(o2[0, ::st] - z["out2_sub"])
    else:
        w1, w2 = z["out1"], z["out2"]
        if w1.ndim == 2:
            w1, w2 = w1[None], w2[None]
        assert o1.shape == w1.shape and o2.shape == w2.shape
        e1, e2 = np.abs(o1 - w1), np.abs(o2 - w2)
    e = np.concatenate([e1.reshape(-1, 255), e2.reshape(-1, 255)]) if e2.size else e1.reshape(-1, 255)
    rows_ok = (e.max(1) <= LOGIT_TOL).mean()
    print(f"{name}: vs reference (its own CPU top-k tie-breaking): max|dlogit| = {e.max():.3e}, "
          f"rows within 1e-3: {100 * rows_ok:.2f}%")
    parity_record(f"{name}/window", max_dlogit_vs_reference=e.max(), rows_within_1e3_vs_reference=rows_ok)
    if name in EVERY_ROW_VS_REFERENCE:        # asserted against the REFERENCE's own logits, on the product's own neighbour lists
        assert e.max() <= LOGIT_TOL, e.max()
    # The reference's neighbour choice among EXACTLY tied distances is an artefact of its top-k implementation
    # (std::partial_sort on CPU, radix-select on CUDA); rows whose 20th/21st neighbours tie can differ: the fraction of rows that
    # agree is recorded (profiles/parity_r3.json), the proof that every other row is a tie is in the two tests at the end of this file.
    # ALL rows agree to 1e-3 with the CPU oracle once it is given the same neighbour sets (the lists this very forward used):
    from oracle import models_ref
    sd = {k: v.cpu() for k, v in ehem.state_dict().items()}
    models_ref.KNN_OVERRIDE = rec.replay(data.shape[0], data.shape[1])
    try:
        with torch.no_grad():
            r1, r2 = models_ref.ehem_forward(sd, data, pos)
    finally:
        models_ref.KNN_OVERRIDE = None
    d = max(np.abs(o1 - r1.numpy()).max(), np.abs(o2 - r2.numpy()).max() if o2.size else 0.0)
    print(f"{name}: vs CPU oracle with identical neighbour sets: max|dlogit| = {d:.3e}")
    parity_record(f"{name}/window", max_dlogit_vs_oracle_same_knn=d)
    assert d <= LOGIT_TOL, d


def test_weights_replaced_after_a_forward_take_effect(dev):
    """Derived weights (fused q|k|v, conv / BN folds, concat slabs, bf16 planes) are caches keyed on their sources' versions:
    load_state_dict / in-place updates after a forward must give the logits of a freshly built model - on both paths."""
    from scp_amd.models import EHEM
    from scp_amd.weights import fill_weights
    z = golden("logits_ehem_c600")
    data, pos = _ehem_case(z)
    a = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
    a0 = _run_packed(a, data, pos, dev)
    w0 = [t.cpu().numpy() for t in a(data.to(dev), pos.to(dev))]
    b = fill_weights(EHEM(ehem_cfg()), 5).to(dev)
    want_p = _run_packed(b, data, pos, dev)
    want_w = [t.cpu().numpy() for t in b(data.to(dev), pos.to(dev))]
    assert np.abs(want_p[0] - a0[0]).max() > 1e-2                  # the two seeds really differ
    a.load_state_dict(b.state_dict())
    got_p = _run_packed(a, data, pos, dev)
    got_w = [t.cpu().numpy() for t in a(data.to(dev), pos.to(dev))]
    for g, w in zip(list(got_p) + got_w, list(want_p) + want_w):
        assert np.array_equal(g, w)
    fill_weights(a, 0)                                             # in-place copy_ of every tensor
    back = _run_packed(a, data, pos, dev)
    assert np.array_equal(back[0], a0[0]) and np.array_equal(back[1], a0[1])
    assert np.array_equal(a(data.to(dev), pos.to(dev))[0].cpu().numpy(), w0[0])


def test_ehem_forward_ctx_equals_reference_signature(dev, ehem):
    z = golden("logits_ehem_c600")
    data = torch.from_numpy(z["data"].astype(np.int64))[None].to(dev)
    pos = torch.from_numpy(z["pos"])[None].to(dev)
    a1, a2 = ehem(data, pos)
    b1, b2 = ehem.forward_ctx(data.reshape(1, -1, 12).to(torch.uint8), pos.transpose(1, 2).contiguous())
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    with pytest.raises(Exception):
        ehem(data.cpu(), pos.cpu())      # no CPU fallback


def test_packed_forward_equals_per_window_forward(dev, ehem):
    """models/packed.py: windows of assorted lengths in ONE pass == one forward per window (same kernels, same numerics)."""
    z = golden("logits_ehem_c1024")
    data = torch.from_numpy(z["data"].astype(np.int64)).to(dev)       # [1024,4,3]
    pos = torch.from_numpy(z["pos"]).to(dev)                            # [3,1024]
    lengths = [1, 7, 2, 300, 513, 1, 200]
    assert sum(lengths) == 1024
    ctx = data.reshape(1024, 12).to(torch.uint8)
    p = pos.T.contiguous()
    ev, od = ehem.forward_packed(ctx, p, lengths)
    a = 0
    e0 = o0 = 0
    worst = 0.0
    for c in lengths:
        r1, r2 = ehem.forward_ctx(ctx[a:a + c][None], p[a:a + c][None])
        ne, no = (c + 1) // 2, c // 2
        worst = max(worst, (ev[e0:e0 + ne] - r1[0]).abs().max().item())
        if no:
            worst = max(worst, (od[o0:o0 + no] - r2[0]).abs().max().item())
        a += c; e0 += ne; o0 += no
    assert e0 == ev.shape[0] and o0 == od.shape[0]
    print(f"packed vs per-window forward: max|dlogit| = {worst:.3e}")
    assert worst < 2e-4, worst


@pytest.mark.parametrize("M,N,K", [(1, 255, 600), (7, 16, 32), (300, 600, 300), (1000, 300, 600), (4100, 255, 512), (513, 240, 80)])
@pytest.mark.parametrize("act", [None, "relu"])
def test_linear_f16x3_vs_float64(dev, M, N, K, act):
    """f16x3 (22-bit operands, one power-of-two scale per row of A and of W): the error of an fp32 FMA chain, for rows whose
    magnitudes differ by many orders (each row has its own scale), zero rows included; one row alone gives the same bits as the
    row inside the batch."""
    from scp_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g) * 25.0
    rowmag = torch.pow(10.0, torch.randint(-6, 7, (M, 1), generator=g).float())
    x = x * rowmag
    if M > 4:
        x[3] = 0.0
    w = torch.randn((N, K), generator=g) / K ** 0.5
    w = w * torch.pow(10.0, torch.randint(-3, 4, (N, 1), generator=g).float())
    b = torch.randn(N, generator=g)
    r = torch.randn((M, N), generator=g)
    xd, wd, bd, rd = x.to(dev), w.to(dev), b.to(dev), r.to(dev)
    # the bare product, against the magnitude its rounding errors are relative to
    scale = (x.double().abs() @ w.double().abs().T) + 1e-300
    ref0 = x.double() @ w.double().T
    err = ((ops.linear(xd, wd, precise=True).cpu().double() - ref0).abs() / scale).max().item()
    err32 = ((ops.linear(xd, wd, exact=True).cpu().double() - ref0).abs() / scale).max().item()
    print(f"M={M} N={N} K={K}: f16x3 rel err {err:.2e} (exact fp32 MFMA chain {err32:.2e})")
    assert err < 1.5e-6, err
    # epilogue: bias, activation, residual
    y = ops.linear(xd, wd, bd, act=act, residual=rd, precise=True)
    ref = ref0 + b.double()
    if act == "relu":
        ref = torch.relu(ref)
    ref = ref + r.double()
    assert ((y.cpu().double() - ref).abs() / (scale + 1.0)).max().item() < 2e-6
    one = ops.linear(xd[M // 2:M // 2 + 1], wd, bd, act=act, residual=rd[M // 2:M // 2 + 1], precise=True)
    assert torch.equal(one[0], y[M // 2])


@pytest.mark.parametrize("M,N,K", [(1, 255, 600), (7, 16, 32), (300, 600, 300), (1000, 300, 600), (4100, 600, 600), (513, 240, 80), (70000, 600, 600)])
@pytest.mark.parametrize("act", [None, "relu"])
def test_linear_split_f16_is_bit_identical_to_the_fp32_row_kernel(dev, M, N, K, act):
    """OctAttention's dense layers on pre-split f16 planes (scp_split_rows_f16 + scp_linear_split_f16): the planes and row scales
    are what the fp32-row kernel stages (scp_row_scale_f16 + its in-tile conversion), products and epilogue in the same order, so
    the result has the SAME BITS - for rows of very different magnitude, zero rows, a residual, ragged M / N, and row ranges taken as
    views of one plane set (the query projection reads the second half of the stacked streams)."""
    from scp_amd import native, ops
    g = torch.Generator().manual_seed(3 * M + N + K)
    x = torch.randn((M, K), generator=g) * 25.0 * torch.pow(10.0, torch.randint(-6, 7, (M, 1), generator=g).float())
    if M > 4:
        x[3] = 0.0
    w = (torch.randn((N, K), generator=g) / K ** 0.5) * torch.pow(10.0, torch.randint(-3, 4, (N, 1), generator=g).float())
    b, r = torch.randn(N, generator=g), torch.randn((M, N), generator=g)
    xd, wd, bd, rd = x.to(dev), w.to(dev), b.to(dev), r.to(dev)
    sw = ops._split16(wd)
    pa = native.SplitActF16(xd)
    rs = native.RowScales(xd)
    assert torch.equal(pa.sc, rs.sc) and torch.equal(pa.isc, rs.isc)
    a = ops._ACT[act]
    want = native.linear_f16x3(xd, sw, bd, a, rd, scales=rs)
    got = native.linear_split_f16(pa, sw, bd, a, rd)
    assert torch.equal(got, want)
    assert torch.equal(native.linear_split_f16(pa, sw, None, a), native.linear_f16x3(xd, sw, None, a))
    if M >= 300:
        lo, hi = M // 3, M // 3 + 129
        assert torch.equal(native.linear_split_f16(pa.rows(lo, hi), sw, bd, a), native.linear_f16x3(xd[lo:hi], sw, bd, a))


@pytest.mark.parametrize("rows,C", [(1, 600), (37, 600), (5000, 600), (1000, 256), (129, 992), (64, 32)])
def test_layernorm_add_planes_equal_a_second_pass(dev, rows, C):
    """scp_layernorm_add_split_f16: the fp32 output is scp_layernorm_add's, and planes / scales are what scp_split_rows_f16 makes of that
    output (bit for bit, padding columns zero)."""
    from scp_amd import native
    g = torch.Generator().manual_seed(rows + C)
    a = (torch.randn((rows, C), generator=g) * torch.pow(10.0, torch.randint(-3, 4, (rows, 1), generator=g).float())).to(dev)
    b = torch.randn((rows, C), generator=g).to(dev)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.3 * torch.randn(C, generator=g)).to(dev)
    for bb in (b, None):
        want = native.layernorm_add(a, bb, gamma, beta, 1e-5)
        out, pl = native.layernorm_add(a, bb, gamma, beta, 1e-5, planes=True)
        ref = native.SplitActF16(want)
        assert torch.equal(out, want)
        assert torch.equal(pl.sc, ref.sc) and torch.equal(pl.isc, ref.isc)
        assert torch.equal(pl.hi.view(torch.int16), ref.hi.view(torch.int16)) and torch.equal(pl.lo.view(torch.int16), ref.lo.view(torch.int16))
        assert not pl.hi[:, C:].any() and not pl.lo[:, C:].any()


def test_octattn_forward_on_planes_equals_forward_on_rows(dev, octattn):
    """The model with its dense layers on pre-split planes (default) against SCP_OA_DENSE=rows: identical logits, bit for bit."""
    from scp_amd.models import oct_attention as oa
    z = golden(sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "logits_octattn_*.npz")))[-1])
    data = torch.from_numpy(z["data"].astype(np.int64))[None].to(dev).repeat(3, 1, 1, 1)
    pos = torch.from_numpy(z["pos"])[None].to(dev).repeat(3, 1, 1, 1)
    assert oa.PLANES
    a = octattn(data, pos)
    try:
        oa.PLANES = False
        b = octattn(data, pos)
    finally:
        oa.PLANES = True
    assert torch.equal(a, b)


# ----------------------------------------------------------------------------------------------- OctAttention
@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "logits_octattn_*.npz"))))
def test_octattn_logits_vs_reference(dev, octattn, name):
    z = golden(name)
    data = torch.from_numpy(z["data"].astype(np.int64))[None].to(dev)
    pos = torch.from_numpy(z["pos"])[None].to(dev)
    keep = data.clone()
    o = octattn(data, pos)[0].cpu().numpy()
    assert torch.equal(data, keep)            # forward is a pure function (reference edits `level` in place)
    e = np.abs(o - z["out"]).max()
    print(f"{name}: max|dlogit| = {e:.3e}")
    assert e <= LOGIT_TOL


@pytest.mark.parametrize("B,c", [(1, 1), (2, 37), (3, 300), (2, 1024), (1, 129)])
def test_octattn_attention_f16x3_vs_fp32_mfma(dev, B, c):
    """Dual-stream causal attention on f16 MFMA (22-bit operands, power-of-two scales) against the fp32 MFMA kernel and a float64
    evaluation of attention_model.py:58-95; rows of very different magnitude (each (token, head) has its own q / k scale)."""
    from scp_amd import native
    g = torch.Generator().manual_seed(B * 1000 + c)
    H, hd = 4, 150
    D = H * hd
    # per-token magnitudes 2^-6 .. 2 on both sides: scores up to ~N(0, 4^2), a softmax from flat to sharp
    mq = torch.pow(2.0, torch.randint(-6, 2, (B, c, 1), generator=g).float())
    mk = torch.pow(2.0, torch.randint(-6, 2, (B, c, 1), generator=g).float())
    q_u = (torch.randn((B, c, D), generator=g) * mq).to(dev)
    k = (torch.randn((B, c, D), generator=g) * mk).to(dev)
    k_u = (torch.randn((B, c, D), generator=g) * mk).to(dev)
    v = (torch.randn((B, c, D), generator=g) * 7.0).to(dev)
    v_u = (torch.randn((B, c, D), generator=g) * 7.0).to(dev)
    keep = native.OCTATTN_MODE
    try:
        native.OCTATTN_MODE = "f16x3"
        o16, ou16 = native.octattn_attention(q_u, k, k_u, v, v_u, H)
        native.OCTATTN_MODE = "f32"
        o32, ou32 = native.octattn_attention(q_u, k, k_u, v, v_u, H)
    finally:
        native.OCTATTN_MODE = keep

    def ref(qq, kk, kku, vv, vvu):
        qq, kk, kku, vv, vvu = [t.double().cpu().reshape(B, c, H, hd).transpose(1, 2) for t in (qq, kk, kku, vv, vvu)]
        s = qq @ kk.transpose(-1, -2) / hd ** 0.5
        du = (qq * kku).sum(-1) / hd ** 0.5
        eye = torch.eye(c, dtype=torch.bool)
        causal = torch.tril(torch.ones(c, c, dtype=torch.bool))
        s1 = s.masked_fill(~causal, -float("inf"))
        o = torch.softmax(s1, -1) @ vv
        s2 = torch.where(eye, du.unsqueeze(-1).expand_as(s), s).masked_fill(~causal, -float("inf"))
        p2 = torch.softmax(s2, -1)
        ou = (p2.masked_fill(eye, 0.0)) @ vv + torch.diagonal(p2, dim1=-2, dim2=-1).unsqueeze(-1) * vvu
        return [t.transpose(1, 2).reshape(B, c, D) for t in (o, ou)]

    r, ru = ref(q_u, k, k_u, v, v_u)
    e16 = max((o16.cpu().double() - r).abs().max().item(), (ou16.cpu().double() - ru).abs().max().item())
    e32 = max((o32.cpu().double() - r).abs().max().item(), (ou32.cpu().double() - ru).abs().max().item())
    print(f"B={B} c={c}: f16x3 max err {e16:.2e}, fp32 MFMA max err {e32:.2e} (|out| max {r.abs().max().item():.1f})")
    assert e16 < 1e-4 and e32 < 1e-4
    assert e16 < 4 * e32 + 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,act,res", [(1000, 256, 256, 0, True), (257, 300, 64, 3, False), (700, 255, 512, 1, False),
                                           (513, 240, 240, 0, False), (2049, 1024, 256, 2, False), (300, 128, 448, 1, True),
                                           (70001, 256, 256, 0, True), (140000, 512, 64, 2, False)])
def test_linear_split_matches_fp32_activation_kernel(M, N, K, act, res):
    """scp_linear_split (both operands pre-split, LDS-DMA staging) is bit-identical to scp_linear_bf16x3 (activation split while
    staging) in every tile configuration, for fp32 and for split outputs, and zero-fills the K padding of its split output.
    The two large cases give more tiles than CUs, i.e. they exercise the persistent loop (several tiles per workgroup, the next
    tile's first DMA in flight during the epilogue)."""
    from scp_amd import native
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g).to(dev)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn((M, N), generator=g).to(dev) if res else None
    sw = native.SplitWeight(w)
    ref = native.linear_bf16x3(a, sw, b, act, r)
    sa = native.split_rows(a)
    hi = a.bfloat16()
    assert torch.equal(sa.t[0, :, :K], hi) and torch.equal(sa.t[1, :, :K], (a - hi.float()).bfloat16())
    assert (sa.t[:, :, K:] == 0).all()
    ld = -(-N // 4) * 4
    for cfg in (0, 1, 2, 3):
        buf = torch.full((M, ld), float("nan"), device=dev)
        out = native.linear_split(sa, sw, b, act, r, out=buf, cfg=cfg)[:, :N]
        assert torch.equal(out, ref), cfg
        o = native.linear_split(sa, sw, b, act, r, want="split", cfg=cfg)
        whi = ref.bfloat16()
        wlo = (ref - whi.float()).bfloat16()
        assert torch.equal(o.t[0, :, :N], whi) and torch.equal(o.t[1, :, :N], wlo) and (o.t[:, :, N:] == 0).all(), cfg
    # gathered split with the zero-row sentinel
    idx = torch.tensor([0, M - 1, M, 5], device=dev)
    gs = native.split_rows(a, idx=idx)
    assert torch.equal(gs.float()[[0, 1, 3]], sa.float()[[0, M - 1, 5]]) and (gs.t[:, 2] == 0).all()


@pytest.mark.gpu
def test_split_producers_match_fp32_forms():
    """LayerNorm and window attention writing the split format = split_rows of their fp32 outputs, bit for bit."""
    from scp_amd import native
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    x = torch.randn((1024, 256), generator=g).to(dev)
    gam, bet = torch.randn(256, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
    valid = (torch.rand(1024, generator=g) > 0.1).float().to(dev)
    y = native.layernorm_rows(x, gam, bet, 1e-5, valid=valid)
    ys = native.layernorm_rows(x, gam, bet, 1e-5, valid=valid, split=True)
    assert torch.equal(ys.t, native.split_rows(y).t)
    ia = torch.arange(0, 1024, 2, device=dev)
    ib = ia + 1
    ib[-1] = 1024
    g2, b2 = torch.randn(512, generator=g).to(dev), torch.randn(512, generator=g).to(dev)
    y = native.layernorm_rows(x, g2, b2, 1e-5, ia=ia, ib=ib)
    ys = native.layernorm_rows(x, g2, b2, 1e-5, ia=ia, ib=ib, split=True)
    assert torch.equal(ys.t, native.split_rows(y).t)
    qkv = torch.randn((1536, 768), generator=g).to(dev)
    table = (0.5 * torch.randn((1023, 4), generator=g)).to(dev)
    wtab = torch.tensor([[0, 1024], [0, 1024], [1024, 512]], dtype=torch.int32, device=dev)
    for shift in (0, 256):
        o = native.swin_attention_packed(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], table, wtab, shift)
        os_ = native.swin_attention_packed(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], table, wtab, shift, split=True)
        assert torch.equal(os_.t, native.split_rows(o).t)


@pytest.mark.gpu
def test_tiled_weight_planes(dev):
    """scp_tile_weight_bf16 against the layout written out in Python (native._tile_planes), and the split-operand layers / the fused MLP on
    tiled planes (default) against a process with SCP_WTILE=0 (row-major planes): identical bits."""
    import subprocess, sys
    from scp_amd import native
    g = torch.Generator().manual_seed(5)
    w = torch.randn((300, 600), generator=g).to(dev)
    sw, rm = native.SplitWeight(w), native.SplitWeight(w, tiled=False)
    th, tl = sw.tiled()
    assert sw.tiled_layout and not rm.tiled_layout
    assert torch.equal(th.view(-1), native._tile_planes(rm.hi).view(-1)) and torch.equal(tl.view(-1), native._tile_planes(rm.lo).view(-1))
    code = ("import torch, hashlib, sys; sys.path.insert(0, %r); from scp_amd import native; dev = torch.device('cuda:0'); g = torch.Generator().manual_seed(9);"
            "x = torch.randn((1000, 256), generator=g).to(dev); w1 = (torch.randn((1024, 256), generator=g) / 16).to(dev); b1 = torch.randn(1024, generator=g).to(dev);"
            "w2 = (torch.randn((256, 1024), generator=g) / 32).to(dev); b2 = torch.randn(256, generator=g).to(dev); a = native.split_rows(x);"
            "s1, s2 = native.SplitWeight(w1), native.SplitWeight(w2);"
            "c = native.linear_split(native.linear_split(a, s1, b1, act=native.ACT_GELU, want='split'), s2, b2, residual=x); d = native.linear_split(a, s1, b1, act=native.ACT_LEAKY);"
            "print(hashlib.sha256(c.cpu().numpy().tobytes() + d.cpu().numpy().tobytes()).hexdigest())") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for wt in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SCP_WTILE=wt))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and len(outs[0]) == 64


@pytest.mark.gpu
def test_position_knn_tile_skipping_keeps_the_exact_neighbours(dev):
    """The position search skips 32-candidate tiles whose bounding box is farther from the box of a wavefront's 32 queries than every
    lane's 20th best.  Morton-sorted clustered points (where most tiles ARE skipped), dense and packed with ragged windows: every
    returned neighbour is at least as near as the float64 20th best (up to fp32 rounding), i.e. nothing nearer was dropped."""
    from scp_amd import native
    g = torch.Generator().manual_seed(3)

    def morton_sorted(n, scale):
        centres = torch.rand((16, 3), generator=g) * 100
        p = centres[torch.randint(0, 16, (n,), generator=g)] + torch.randn((n, 3), generator=g) * scale
        q = ((p - p.min(0)[0]) / (p.max(0)[0] - p.min(0)[0] + 1e-9) * 1023).long()
        key = torch.zeros(n, dtype=torch.long)
        for b in range(10):
            for c in range(3):
                key |= ((q[:, c] >> b) & 1) << (3 * b + c)
        return p[torch.argsort(key)].contiguous()

    def check(x, got):
        d = ref_knn_values(x[None], x.shape[0])[0]
        k = min(20, x.shape[0])
        kth = torch.topk(d, k, dim=1)[0][:, -1:]
        val = torch.gather(d, 1, got[:, :k])
        scale = (x.double() ** 2).sum(1).max()
        assert (val >= kth - 4e-6 * scale).all()
        assert (torch.sort(got[:, :k], 1)[0].diff(dim=1) != 0).all()          # k distinct neighbours

    for n in (8192, 5000):
        x = morton_sorted(n, 0.8)
        got = native.knn_topk(x[None].to(dev), 20).cpu().long()[0]
        check(x, got)
    lengths = [8192, 300, 4097, 8192, 31]
    rows = sum(-(-n // 512) * 512 for n in lengths)
    xp = torch.zeros((rows, 3))
    tab, base = [], 0
    for n in lengths:
        xp[base:base + n] = morton_sorted(n, 0.5)
        tab += [[base, n]] * (-(-n // 512))
        base += -(-n // 512) * 512
    out = native.knn_topk_packed(xp.to(dev), torch.tensor(tab, dtype=torch.int32, device=dev)).cpu().long()
    base = 0
    for n in lengths:
        check(xp[base:base + n], out[base:base + n] - base)
        base += -(-n // 512) * 512


@pytest.mark.gpu
@pytest.mark.parametrize("C", [144, 192])
def test_knn_f16x3_agrees_with_exact_fp32_chain(dev, C):
    """The default 144-/192-feature search (f16x3 on f16 MFMA) against the exact fp32 MFMA chain on the same input:
    identical neighbour SETS for almost every point; where they differ, only candidates whose distances are within fp32
    rounding of the 20th best are exchanged."""
    from scp_amd import native
    g = torch.Generator().manual_seed(C)
    n = 4096
    # smooth random walk + noise: neighbours are close in feature space, like octree siblings; wide dynamic range per column
    x = (torch.cumsum(torch.randn((1, n, C), generator=g) * 0.05, 1) + torch.randn((1, n, C), generator=g) * 0.3)
    x = (x * torch.logspace(-3, 1.5, C)[None, None, :]).contiguous()
    try:
        native.set_knn_mode(False)
        ref = native.knn_topk(x.to(dev), 20).cpu().long()
        native.set_knn_mode(True)
        got = native.knn_topk(x.to(dev), 20).cpu().long()
    finally:
        native.set_knn_mode(True)
    same = (torch.sort(ref, 2)[0] == torch.sort(got, 2)[0]).all(2)
    frac = same.float().mean().item()
    print(f"C={C}: neighbour sets identical for {100 * frac:.3f}% of the points")
    assert frac > 0.995
    d = ref_knn_values(x.double(), n)            # full distance matrix in float64
    kth = torch.topk(d, 20, dim=2)[0][..., -1:]
    val = torch.gather(d, 2, got)
    scale = (x.double() ** 2).sum(2).max()
    assert (val >= kth - 2e-6 * scale).all()


@pytest.mark.gpu
@pytest.mark.parametrize("C", [3, 144])
def test_knn_packed_with_a_priori_bound_is_unchanged(dev, C):
    """scp_knn_topk_packed_bounded: a valid pruning bound per row (here: slightly below the true 20th-best distance; -inf for some
    rows) only skips list insertions - the neighbour lists are identical."""
    from scp_amd import native
    g = torch.Generator().manual_seed(C + 1)
    lens = [1500, 8192, 30]
    rows = sum(-(-n // 512) * 512 for n in lens)
    x = torch.zeros((rows, C))
    ktab, r0 = [], 0
    for n in lens:
        x[r0:r0 + n] = torch.cumsum(torch.randn((n, C), generator=g) * 0.1, 0) + torch.randn((n, C), generator=g) * 0.2
        ktab += [[r0, n]] * (-(-n // 512))
        r0 += -(-n // 512) * 512
    x, ktab = x.to(dev), torch.tensor(ktab, dtype=torch.int32, device=dev)
    idx = native.knn_topk_packed(x, ktab)
    xx = (x * x).sum(1)
    d = 2 * (x[idx.long()] * x[:, None]).sum(2) - xx[idx.long()] - xx[:, None]
    thr = d.min(1)[0] - 1e-4 * xx.max()
    thr[::7] = float("-inf")
    assert torch.equal(native.knn_topk_packed(x, ktab, thr), idx)


@pytest.mark.gpu
@pytest.mark.parametrize("lengths", [[1], [1, 6, 20, 56, 208, 1372, 5605, 8192, 4938, 7, 2, 513, 1], [8192] * 5 + [3], [2, 1, 1, 1, 4096]])
def test_packed_plan_kernel_equals_torch_construction(dev, lengths):
    """csrc/plan.hip (all index maps of the packed forward in one launch) against the vectorised torch construction in
    models/packed.py, entry for entry."""
    from scp_amd.models.packed import PackedPlan
    a = PackedPlan(lengths, device=dev)                       # native kernel
    b = PackedPlan(lengths, device=dev, use_native=False)     # torch index ops
    assert a.n_tokens == b.n_tokens
    for k, vb in b.d.items():
        va = a.d[k]
        if isinstance(vb, (list, tuple)):
            assert len(va) == len(vb), k
            for xa, xb in zip(va, vb):
                if isinstance(xb, (list, tuple)):
                    for ya, yb in zip(xa, xb):
                        assert ya.dtype == yb.dtype and torch.equal(ya, yb), k
                else:
                    assert xa.dtype == xb.dtype and xa.shape == xb.shape and torch.equal(xa, xb), k
        else:
            assert va.dtype == vb.dtype and va.shape == vb.shape and torch.equal(va, vb), k


@pytest.mark.gpu
def test_hierarchical_concat_layers_equal_direct_form(dev, ehem):
    """The layers that consume concat_states (ehem.py:75-86) are evaluated per Swin stage with gathered partial sums
    (models/packed.py: _concat_layer); against the direct form (build the 1280-wide concatenation, one product) the logits differ
    only by fp32 summation order."""
    from scp_amd.models import packed
    z = golden("logits_ehem_c1024")
    ctx = torch.from_numpy(z["data"].astype(np.int64)).to(dev).reshape(1024, 12).to(torch.uint8)
    p = torch.from_numpy(z["pos"]).to(dev).T.contiguous()
    lengths = [1, 7, 2, 300, 513, 1, 200]
    try:
        packed.HIER = True
        a = ehem.forward_packed(ctx, p, lengths)
        packed.HIER = False
        b = ehem.forward_packed(ctx, p, lengths)
    finally:
        packed.HIER = True
    worst = max((a[0] - b[0]).abs().max().item(), (a[1] - b[1]).abs().max().item())
    print(f"hierarchical vs direct concat layers: max|dlogit| = {worst:.3e}")
    assert worst < 5e-5


@pytest.mark.gpu
def test_fused_two_stage_concat_layer_equals_the_two_launch_form(dev, ehem):
    """Round 6 (csrc/gemm_split.hip: gemm_hier2_kernel): stages 0 and 1 of a layer over concat_states in one launch - the stage-1 product of a
    256-token tile's own parents stays in the accumulators - against the two-launch form with an fp32 partial sum between them
    (packed.FUSE2 = False): the logits differ only by fp32 summation order; and the fused form is deterministic and batch-invariant (a window's
    rows do not depend on what else is in the launch)."""
    from scp_amd.models import packed
    z = golden("logits_ehem_c1024")
    ctx = torch.from_numpy(z["data"].astype(np.int64)).to(dev).reshape(1024, 12).to(torch.uint8)
    p = torch.from_numpy(z["pos"]).to(dev).T.contiguous()
    lengths = [1, 7, 2, 300, 513, 1, 200]
    try:
        packed.FUSE2 = True
        a = ehem.forward_packed(ctx, p, lengths)
        a2 = ehem.forward_packed(ctx, p, lengths)
        packed.FUSE2 = False
        b = ehem.forward_packed(ctx, p, lengths)
    finally:
        packed.FUSE2 = True
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1])
    worst = max((a[0] - b[0]).abs().max().item(), (a[1] - b[1]).abs().max().item())
    print(f"fused two-stage vs two-launch concat layers: max|dlogit| = {worst:.3e}")
    assert worst < 5e-5
    # batch invariance: the 513-token window alone gives the bits it has inside the packed launch
    s0 = sum(lengths[:4])
    c = ehem.forward_packed(ctx[s0:s0 + 513], p[s0:s0 + 513], [513])
    e0 = sum((l + 1) // 2 for l in lengths[:4])
    o0 = sum(l // 2 for l in lengths[:4])
    assert torch.equal(c[0], a[0][e0:e0 + 257]) and torch.equal(c[1], a[1][o0:o0 + 256])


@pytest.mark.gpu
@pytest.mark.parametrize("M,K0,N,with_res", [(256, 256, 1024, True), (512, 512, 768, True), (1024, 256, 256, False), (2560, 256, 1024, True)])
def test_linear_split_hier2_vs_float64(dev, M, K0, N, with_res):
    """scp_linear_split_hier2 against float64: out[m] = leaky(A0[m] . W0^T + A1[parent[m]] . W1^T + bias + res[res_map[m]]), parents of a 256-row
    tile consecutive (token t -> t >> 1), residual rows gathered; error of a bf16x3 product chain."""
    from scp_amd import native
    g = torch.Generator().manual_seed(M + K0 + N)
    a0 = torch.randn((M, K0), generator=g).to(dev)
    M1 = M // 2 + 300
    a1 = torch.randn((M1, 256), generator=g).to(dev)
    w0 = (torch.randn((N, K0), generator=g) / K0 ** 0.5).to(dev)
    w1 = (torch.randn((N, 256), generator=g) / 16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    # every 256-row tile is a run of tokens of one window: parents consecutive from an arbitrary base
    base = torch.randint(0, 300, (M // 256,), generator=g)
    parent = (base[:, None] + (torch.arange(256)[None, :] >> 1)).reshape(-1).to(dev)
    res = torch.randn((M // 4 + 7, N), generator=g).to(dev) if with_res else None
    rmap = torch.randint(0, M // 4 + 7, (M,), generator=g).to(dev) if with_res else None
    out = native.linear_split_hier2(native.split_rows(a0), native.SplitWeight(w0), native.split_rows(a1), native.SplitWeight(w1), parent, b,
                                    native.ACT_LEAKY, residual=res, res_map=rmap)
    got = out.t[0].float()[:, :N].double() + out.t[1].float()[:, :N].double()
    want = a0.double() @ w0.double().T + a1.double()[parent] @ w1.double().T + b.double()
    if with_res:
        want = want + res.double()[rmap]
    want = torch.where(want > 0, want, 0.01 * want)
    err = (got - want).abs().max().item()
    print(f"hier2 M={M} K0={K0} N={N}: max err {err:.3e}")
    assert err < 2e-4          # the split output itself carries 2^-17 relative; products 1e-5 relative of sum |a||w|


@pytest.mark.gpu
@pytest.mark.parametrize("M,dims,gather,scatter", [(1, (256, 256, 255), False, False), (300, (256, 240, 240), False, False), (1000, (256, 256, 255), True, True),
                                                   (33000, (256, 240, 240), True, False), (129, (256, 256, 128), False, True)])
def test_mlp3_rows_vs_float64(dev, M, dims, gather, scatter):
    """scp_mlp3_rows (csrc/rowchain.hip: rc_mlp3_kernel): Sequential(Linear, LeakyReLU, Linear, LeakyReLU, Linear) on 256-channel rows in one
    launch against float64 - ragged layer widths (240, 255), gathered input rows, scattered output rows with dropped rows, output as a
    column-offset view - and every row independent of what else is in the launch (bit-identical to a launch of a prefix)."""
    from scp_amd import native
    g = torch.Generator().manual_seed(M + sum(dims))
    seq = torch.nn.Sequential(torch.nn.Linear(256, dims[0]), torch.nn.LeakyReLU(), torch.nn.Linear(dims[0], dims[1]), torch.nn.LeakyReLU(),
                              torch.nn.Linear(dims[1], dims[2]))
    with torch.no_grad():
        for p_ in seq.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.08 if p_.dim() == 2 else 0.5))
    seq = seq.to(dev)
    n_src = M + 17
    x = (torch.randn((n_src, 256), generator=g) * 2.0).to(dev)
    in_map = torch.randint(0, n_src, (M,), generator=g).to(dev) if gather else None
    mw = native.Mlp3Weights(seq)
    N = dims[2]
    Np = -(-N // 4) * 4
    rows_out = M + 5
    wide = torch.full((rows_out, 16 + 256), 7.0, dtype=torch.float32, device=dev)
    out = wide[:, 16:]
    if scatter:
        perm = torch.randperm(rows_out, generator=g)[:M]
        out_map = perm.clone()
        out_map[::7] = -1                                   # dropped rows
        out_map = out_map.to(dev)
    else:
        out_map = None
    native.mlp3_rows(x if gather else x[:M], mw, out, in_map=in_map, out_map=out_map)
    xin = (x[in_map] if gather else x[:M]).double()
    h = xin
    for i in (0, 2, 4):
        h = h @ seq[i].weight.double().T + seq[i].bias.double()
        if i < 4:
            h = torch.where(h > 0, h, 0.01 * h)
    if scatter:
        keep = out_map >= 0
        got = out[out_map[keep]][:, :N].double()
        want = h[keep]
        untouched = torch.ones(rows_out, dtype=torch.bool, device=dev)
        untouched[out_map[keep]] = False
        assert (wide[untouched] == 7.0).all()               # dropped rows and rows nobody maps to are not written
    else:
        got, want = out[:M, :N].double(), h
        assert (wide[M:] == 7.0).all()
    err = (got - want).abs().max().item()
    print(f"mlp3 M={M} dims={dims}: max err {err:.3e}")
    assert err < 2e-4
    assert (wide[:, :16] == 7.0).all() and (wide[:, 16 + Np:] == 7.0).all()     # nothing outside the head's columns
    if Np > N:
        rows = out_map[out_map >= 0] if scatter else torch.arange(M, device=dev)
        assert (out[rows][:, N:Np] == 0).all()                                  # padding columns of a ragged head: zeros
    if not gather and not scatter and M > 40:
        out2 = torch.empty((40, 256), dtype=torch.float32, device=dev)
        native.mlp3_rows(x[:40], mw, out2)
        assert torch.equal(out2[:, :Np], out[:40, :Np])


@pytest.mark.gpu
def test_one_launch_heads_equal_three_launch_heads(dev, ehem):
    """Round 6: prob_pred_mlp1 / pre_attn_mlp as one row-chain launch each against the three split-GEMM launches per head of rounds 1 - 5
    (packed.CHAIN_HEADS = False): the logits differ only by fp32 summation order."""
    from scp_amd.models import packed
    z = golden("logits_ehem_c1024")
    ctx = torch.from_numpy(z["data"].astype(np.int64)).to(dev).reshape(1024, 12).to(torch.uint8)
    p = torch.from_numpy(z["pos"]).to(dev).T.contiguous()
    lengths = [1, 7, 2, 300, 513, 1, 200]
    try:
        packed.CHAIN_HEADS = True
        a = ehem.forward_packed(ctx, p, lengths)
        packed.CHAIN_HEADS = False
        b = ehem.forward_packed(ctx, p, lengths)
    finally:
        packed.CHAIN_HEADS = True
    worst = max((a[0] - b[0]).abs().max().item(), (a[1] - b[1]).abs().max().item())
    print(f"one-launch vs three-launch heads: max|dlogit| = {worst:.3e}")
    assert worst < 5e-5


# ---------------------------------------------------------------------------------------------------------------- row-chain kernels
@pytest.mark.gpu
@pytest.mark.parametrize("M,N", [(1, 256), (31, 768), (128, 768), (129, 512), (1000, 256), (70001, 768)])
def test_swin_ln_linear_vs_float64(dev, M, N):
    """scp_swin_ln_linear (csrc/rowchain.hip): LayerNorm + dense layer in one launch against float64 - LayerNorm affine folded into
    the weight, rows the window pads after LayerNorm (valid = 0) come out as the bias alone, any row count; and the rows of a launch do
    not depend on what else is in it (bit-identical to a launch of a prefix)."""
    from scp_amd import native
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn((M, 256), generator=g) * 1.5 + 0.3).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(256, generator=g)).to(dev), (0.1 * torch.randn(256, generator=g)).to(dev)
    W, b = (torch.randn((N, 256), generator=g) * 0.05).to(dev), (torch.randn(N, generator=g) * 0.1).to(dev)
    valid = (torch.rand(M, generator=g) > 0.2).float().to(dev)
    fw = native.LnFoldedWeight(W, gamma, beta)
    y = native.swin_ln_linear(x, fw, b, 1e-5, valid)
    ref = (F.layer_norm(x.double(), (256,), gamma.double(), beta.double(), 1e-5) * valid.double()[:, None]) @ W.double().T + b.double()
    err = (y.double() - ref).abs().max().item()
    assert err < 1e-4, err
    if (valid == 0).any():
        pad = (valid == 0).nonzero().flatten()[:5]
        assert torch.equal(y[pad], b[None].expand(len(pad), N))                  # exactly the bias
    y2 = native.swin_ln_linear(x, fw, None, 1e-5, None)
    assert (y2.double() - F.layer_norm(x.double(), (256,), gamma.double(), beta.double(), 1e-5) @ W.double().T).abs().max().item() < 1e-4
    k = max(1, M // 3)
    assert torch.equal(native.swin_ln_linear(x[:k].contiguous(), fw, b, 1e-5, valid[:k].contiguous()), y[:k])


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 129, 1000, 70001])
def test_geo_edge_mlps_vs_float64_and_the_launches_they_replace(dev, M):
    """scp_geo_edge_mlps: edge_mlp2(cat(pos3, edge_mlp1(cat(pos1, pos2, pos3)))) - six dense layers chained through the accumulators in one
    launch - against float64 and against the six split GEMMs; output into a column slice; a row's result does not depend on the launch."""
    import torch.nn as nn
    from scp_amd import native
    from scp_amd.ops import leaky_mlp3_s, split_cat
    torch.manual_seed(M)
    mk = lambda a, b, c, d: nn.Sequential(nn.Linear(a, b), nn.LeakyReLU(0.01), nn.Linear(b, c), nn.LeakyReLU(0.01), nn.Linear(c, d)).to(dev)
    m1, m2 = mk(448, 256, 256, 256), mk(512, 256, 256, 128)
    g = torch.Generator().manual_seed(M)
    p1, p2, p3 = (torch.randn((M, c), generator=g).to(dev) for c in (64, 128, 256))
    ew = native.EdgeMlpWeights(m1, m2)
    out = torch.full((M, 256), 7.0, device=dev)
    native.geo_edge_mlps(p1, p2, p3, ew, out[:, 128:])
    assert (out[:, :128] == 7.0).all()
    e_in = native.SplitAct.empty(M, 512, dev)
    native.split_rows(p3, out=e_in.cols(0, 256))
    leaky_mlp3_s(m1, split_cat((p1, p2, p3)), want="split", out_split=e_in.cols(256, 512))
    old = leaky_mlp3_s(m2, e_in)
    with torch.no_grad():
        ref = m2.double()(torch.cat((p3.double(), m1.double()(torch.cat((p1, p2, p3), 1).double())), 1))
    err, err0 = (out[:, 128:].double() - ref).abs().max().item(), (old.double() - ref).abs().max().item()
    print(f"M={M}: max err vs float64 {err:.2e} (six launches {err0:.2e})")
    assert err < 2e-5 and err < 3 * err0 + 2e-6
    if M > 300:
        lo, hi = M // 3, M // 3 + 131
        o2 = torch.empty((hi - lo, 128), device=dev)
        native.geo_edge_mlps(p1[lo:hi], p2[lo:hi], p3[lo:hi], ew, o2)
        assert torch.equal(o2, out[lo:hi, 128:])


@pytest.mark.gpu
@pytest.mark.parametrize("n_src", [1, 2, 255, 1000, 70001])
def test_swin_merge_vs_float64_and_the_launches_it_replaces(dev, n_src):
    """scp_swin_merge: gather of the (even, odd) tokens + LayerNorm(512) + 512 -> 256 reduction in one row-chain launch (the K = 512
    product as two K = 256 halves over the same accumulators), against float64 and against layernorm_rows(gather) + the split GEMM;
    zero rows (index == n_src) in either slot, ragged M, and a row's result does not depend on the launch it is in."""
    from scp_amd import native
    from scp_amd.ops import linear_s
    g = torch.Generator().manual_seed(n_src)
    x = (torch.randn((n_src, 256), generator=g) * 1.3 + 0.2).to(dev)
    M = (n_src + 1) // 2
    ev = torch.arange(0, 2 * M, 2)
    od = ev + 1
    od[od >= n_src] = n_src
    if M > 6:
        ev[5] = n_src
    ev, od = ev.to(dev), od.to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(512, generator=g)).to(dev), (0.1 * torch.randn(512, generator=g)).to(dev)
    W = (torch.randn((256, 512), generator=g) * 0.04).to(dev)
    mw = native.MergeWeights(W, gamma, beta)
    y = native.swin_merge(x, ev, od, mw)
    y0 = linear_s(native.layernorm_rows(x, gamma, beta, 1e-5, ia=ev, ib=od, split=True), W, None)
    xz = torch.cat((x, torch.zeros((1, 256), device=dev))).double()
    ref = F.layer_norm(torch.cat((xz[ev], xz[od]), 1), (512,), gamma.double(), beta.double(), 1e-5) @ W.double().T
    err, err0 = (y.double() - ref).abs().max().item(), (y0.double() - ref).abs().max().item()
    print(f"n_src={n_src}: max err vs float64 {err:.2e} (two launches {err0:.2e})")
    assert err < 1e-4 and err < 3 * err0 + 1e-5
    if M > 200:
        lo, hi = M // 3, M // 3 + 131
        assert torch.equal(native.swin_merge(x, ev[lo:hi].contiguous(), od[lo:hi].contiguous(), mw), y[lo:hi])


@pytest.mark.gpu
@pytest.mark.parametrize("M,N", [(512, 768), (512, 512), (1536, 768), (70144, 768), (70144, 512)])
def test_swin_ln_qkv_planes_are_the_split_of_the_fp32_projection(dev, M, N):
    """scp_swin_ln_qkv (keys / values leave the LayerNorm + projection kernel as the attention's bf16 planes, value heads computed with the
    MFMA operands swapped so that the accumulator IS a V^T tile): q and every plane bit-identical to scp_swin_ln_linear's fp32 output
    converted by scp_swin_kv_planes - with a valid mask, several tiles per CU, and the key | value form of the cross layers."""
    from scp_amd import native
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn((M, 256), generator=g) * 1.5 + 0.3).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(256, generator=g)).to(dev), (0.1 * torch.randn(256, generator=g)).to(dev)
    valid = (torch.rand(M, generator=g) > 0.1).float().to(dev)
    W, b = (torch.randn((N, 256), generator=g) * 0.05).to(dev), (torch.randn(N, generator=g) * 0.1).to(dev)
    fw = native.LnFoldedWeight(W, gamma, beta)
    for vm in (valid, None):
        ref = native.swin_ln_linear(x, fw, b, 1e-5, vm)
        nq = N - 512
        want = native.KvPlanes(ref[:, nq:nq + 256], ref[:, nq + 256:])
        q, kv = native.swin_ln_qkv(x, fw, b, 1e-5, vm)
        assert (q is None) == (N == 512)
        if q is not None:
            assert torch.equal(q, ref[:, :256].contiguous())
        assert torch.equal(kv.t.view(torch.int16), want.t.view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("N", [768, 512, 256])
def test_short_ln_linear_launches_split_a_tile_over_workgroups_with_identical_bits(dev, N):
    """Round 5 (decoder latency): a launch with fewer tiles than CUs gives every tile to several workgroups, each running LayerNorm and its own
    run of 64-channel steps (RcLnLinArgs.ngroups).  Every output channel is still one wave's accumulation chain: the rows of short launches
    (128 ... 8192 rows: 6, 3, 2 or 1 workgroups per tile) equal the same rows inside a long launch (548 tiles, one workgroup each) bit for bit -
    q, key planes, value planes, with and without a valid mask."""
    from scp_amd import native
    g = torch.Generator().manual_seed(N)
    Mbig = 70144
    x = (torch.randn((Mbig, 256), generator=g) * 1.5 + 0.3).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(256, generator=g)).to(dev), (0.1 * torch.randn(256, generator=g)).to(dev)
    valid = (torch.rand(Mbig, generator=g) > 0.1).float().to(dev)
    W, b = (torch.randn((N, 256), generator=g) * 0.05).to(dev), (torch.randn(N, generator=g) * 0.1).to(dev)
    fw = native.LnFoldedWeight(W, gamma, beta)
    for vm in (valid, None):
        big = native.swin_ln_linear(x, fw, b, 1e-5, vm)
        bigq = native.swin_ln_qkv(x, fw, b, 1e-5, vm) if N != 256 else None
        for M in (128, 512, 2048, 5504, 8192, 16384, 33024):
            xs, vs = x[:M].contiguous(), None if vm is None else vm[:M].contiguous()
            assert torch.equal(native.swin_ln_linear(xs, fw, b, 1e-5, vs), big[:M]), (N, M)
            if bigq is not None:
                q, kv = native.swin_ln_qkv(xs, fw, b, 1e-5, vs)
                if q is not None:
                    assert torch.equal(q, bigq[0][:M]), (N, M)
                assert torch.equal(kv.t.view(torch.int16), bigq[1].t[:, :M].view(torch.int16)), (N, M)


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [0, 256])
def test_plane_fed_attention_is_bit_identical_to_the_fp32_fed_kernel(dev, shift):
    """scp_swin_attention_packed_planes (K / V tiles staged by LDS-DMA from pre-split planes, 32-key tiles, one barrier per tile) against
    scp_swin_attention_packed: same products in the same order - identical output, fp32 and split form, sequences of 1 - 3 windows."""
    from scp_amd import native
    g = torch.Generator().manual_seed(shift + 3)
    lens = [1, 2, 3, 1, 2]
    T = sum(lens) * 512
    qkv = (torch.randn((T, 768), generator=g) * 2.0).to(dev)
    table = (torch.randn((1023, 4), generator=g) * 0.5).to(dev)
    rows, base = [], 0
    for n in lens:
        rows += [[base, n * 512]] * n
        base += n * 512
    wtab = torch.tensor(rows, dtype=torch.int32, device=dev)
    q, k, v = qkv[:, :256], qkv[:, 256:512], qkv[:, 512:]
    kv = native.KvPlanes(k, v)
    assert torch.equal(native.swin_attention_packed_planes(q, kv, table, wtab, shift), native.swin_attention_packed(q, k, v, table, wtab, shift))
    a, b = native.swin_attention_packed_planes(q, kv, table, wtab, shift, split=True), native.swin_attention_packed(q, k, v, table, wtab, shift, split=True)
    assert torch.equal(a.t.view(torch.int16), b.t.view(torch.int16))


def _attn_f64(q, k, v, table, nwin, shift):
    """models/swin_transformer.py:443-501, 603-652 for independent 512-token sequences of ONE window each, in float64."""
    T = q.shape[0]
    q, k, v = (t.double().reshape(nwin, 512, 4, 64).permute(0, 2, 1, 3) for t in (q, k, v))
    if shift:
        q, k, v = (torch.roll(t, -shift, 2) for t in (q, k, v))
    idx = torch.arange(512, device=q.device)
    bias = table.double()[(idx[:, None] - idx[None, :] + 511)].permute(2, 0, 1)            # [head][i][j]
    s = q @ k.transpose(-1, -2) / 8.0 + bias[None]
    if shift:
        reg = (idx >= 512 - shift).long()
        s = s + torch.where(reg[:, None] != reg[None, :], -100.0, 0.0)[None, None]
    o = torch.softmax(s, -1) @ v
    if shift:
        o = torch.roll(o, shift, 2)
    return o.permute(0, 2, 1, 3).reshape(T, 256)


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [0, 256])
def test_attention_fixed_reference_sweep_and_its_fallback(dev, shift):
    """Round 4: every workgroup of the bf16x3 attention first sweeps its keys with P = exp2(S) against the FIXED reference 0 (no running
    maximum, no rescale) and repeats the sweep in the standard online-softmax form only if a row sum left [2^-100, 2^100].
    (a) ordinary scores: the default is as close to float64 as the standard form; (b) scores beyond 2^+-100: the result IS the standard
    form's, bit for bit (the second sweep), in the plane-fed and in the rows-fed kernel; finite and right against float64."""
    from scp_amd import native
    L = native.lib()
    g = torch.Generator().manual_seed(31 + shift)
    nwin = 6
    T = nwin * 512
    table = (torch.randn((1023, 4), generator=g) * 0.5).to(dev)
    wtab = torch.tensor([[w * 512, 512] for w in range(nwin)], dtype=torch.int32, device=dev)
    try:
        for scale, fallback in ((1.0, False), (2.0, False), (1.0, True)):
            qkv = torch.randn((T, 768), generator=g).to(dev)
            qkv[:, :512] *= scale
            if fallback:
                qkv[512:1024, :512] *= 40.0                     # window 1: scores of +-2000 (log2 domain): exp2 overflows
                u = torch.full((1, 256), 3.0, device=dev)       # window 3: every score near -200: exp2 underflows to nothing
                qkv[3 * 512:4 * 512, :256] = u + 0.05 * qkv[3 * 512:4 * 512, :256]
                qkv[3 * 512:4 * 512, 256:512] = -2.0 * u + 0.05 * qkv[3 * 512:4 * 512, 256:512]
            q, k, v = qkv[:, :256], qkv[:, 256:512], qkv[:, 512:]
            kv = native.KvPlanes(k, v)
            want = _attn_f64(q, k, v, table, nwin, shift)
            out = {}
            for var in (0, 1):
                L.scp_set_attention_variant(var)
                out[var] = (native.swin_attention_packed_planes(q, kv, table, wtab, shift), native.swin_attention_packed(q, k, v, table, wtab, shift))
                assert torch.equal(out[var][0], out[var][1])                                    # plane-fed == rows-fed, either form
                assert bool(torch.isfinite(out[var][0]).all())
            plain = torch.ones(T, dtype=torch.bool, device=dev)
            if fallback:
                plain[512:1024] = False
                plain[3 * 512:4 * 512] = False
            e0, e1 = ((out[i][0].double() - want)[plain].abs().max().item() for i in (0, 1))
            print(f"shift {shift} scale {scale}: |standard - f64| {e0:.2e}, |default - f64| {e1:.2e}, |out| max {want.abs().max().item():.2f}")
            # bf16x3 carries 16 significant bits per operand and a score of magnitude m (log2 domain) is known to ulp(m): the error grows with the scale
            tol = 5e-5 * scale ** 2
            assert e0 < tol and e1 < tol and e1 < 1.5 * e0, (e0, e1)                          # and the default is as good as the standard form
            assert not torch.equal(out[1][0][:512], out[0][0][:512])                             # other last bits: the first sweep was kept
            if fallback:
                for wdw in (1, 3):      # these went through the second sweep: the standard form's bits
                    assert torch.equal(out[1][0][wdw * 512:(wdw + 1) * 512], out[0][0][wdw * 512:(wdw + 1) * 512])
                e3 = (out[1][0].double() - want)[3 * 512:4 * 512].abs().max().item()
                print(f"   window of scores near -200: |default - f64| {e3:.2e}")
                assert e3 < 1e-2, e3
    finally:
        L.scp_set_attention_variant(1)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 33, 128, 129, 1000, 70001])
def test_swin_post_attn_vs_float64_and_the_launches_it_replaces(dev, M):
    """scp_swin_post_attn: attention projection + residual + LayerNorm + fc1 + GELU + fc2 + residual in one launch, the intermediate
    activations chained through MFMA accumulators (never in memory), against float64 and against the three launches of rounds 1 - 2
    (scp_linear_split, scp_layernorm_rows_split, scp_linear_split twice); batch-invariant bit for bit; works in place."""
    from scp_amd import native
    from scp_amd.ops import linear_s, _split
    g = torch.Generator().manual_seed(M)
    rn = lambda *sh, s=1.0: (torch.randn(sh, generator=g) * s).to(dev)
    x, o = rn(M, 256), rn(M, 256)
    wp, bp = rn(256, 256, s=0.05), rn(256, s=0.1)
    gamma, beta = 1 + rn(256, s=0.1), rn(256, s=0.1)
    w1, b1, w2, b2 = rn(1024, 256, s=0.05), rn(1024, s=0.1), rn(256, 1024, s=0.03), rn(256, s=0.1)
    osp = native.split_rows(o)
    pw = native.PostAttnWeights(wp, bp, gamma, beta, w1, b1, w2, b2)
    y = native.swin_post_attn(osp, x, pw)
    x1 = x.double() + o.double() @ wp.double().T + bp.double()
    h = F.gelu(F.layer_norm(x1, (256,), gamma.double(), beta.double(), 1e-5) @ w1.double().T + b1.double())
    ref = x1 + h @ w2.double().T + b2.double()
    err = (y.double() - ref).abs().max().item()
    x1o = linear_s(osp, wp, bp, residual=x)
    hid = native.linear_split(native.layernorm_rows(x1o, gamma, beta, 1e-5, split=True), _split(w1), b1, act=native.ACT_GELU, want="split")
    old = native.linear_split(hid, _split(w2), b2, residual=x1o)
    err_old = (old.double() - ref).abs().max().item()
    print(f"M={M}: max err vs float64 {err:.2e} (the three launches: {err_old:.2e})")
    assert err < 1e-4 and err < 3 * err_old + 1e-5
    k = max(1, M // 2)
    assert torch.equal(native.swin_post_attn(native.split_rows(o[:k].contiguous()), x[:k].contiguous(), pw), y[:k])
    xc = x.clone()
    assert native.swin_post_attn(osp, xc, pw, out=xc) is xc and torch.equal(xc, y)      # in place: a tile reads its rows before it writes them


@pytest.mark.gpu
def test_wide_post_attn_kernel_has_the_chain_kernels_bits(dev):
    """Round 5 (decoder latency): short launches of scp_swin_post_attn run rc_post_attn_wide_kernel - a workgroup per 32 rows, its four waves
    splitting the output channels of every product instead of the rows (864 products per wave instead of 3 552: a quarter of the launch time
    when the launch has fewer tiles than CUs).  Every output element is one wave's accumulation chain with the chain kernel's operands in the
    chain kernel's order: the two kernels agree bit for bit - any M, ragged last tile, tile lists, in place - and so do the encoder (one packed
    launch through the chain kernel) and the decoder (short launches)."""
    from scp_amd import native
    L = native.lib()
    g = torch.Generator().manual_seed(5)
    rn = lambda *sh, s=1.0: (torch.randn(sh, generator=g) * s).to(dev)
    wp, bp = rn(256, 256, s=0.05), rn(256, s=0.1)
    gamma, beta = 1 + rn(256, s=0.1), rn(256, s=0.1)
    w1, b1, w2, b2 = rn(1024, 256, s=0.05), rn(1024, s=0.1), rn(256, 1024, s=0.03), rn(256, s=0.1)
    pw = native.PostAttnWeights(wp, bp, gamma, beta, w1, b1, w2, b2)
    try:
        for M in (1, 31, 32, 33, 128, 129, 1000, 4096, 8192, 20001):
            x, o = rn(M, 256, s=2.0), rn(M, 256)
            osp = native.split_rows(o)
            L.scp_rc_set_wide(0)
            want = native.swin_post_attn(osp, x, pw)
            L.scp_rc_set_wide(1)
            got = native.swin_post_attn(osp, x, pw)
            assert torch.equal(got, want), M
            xc = x.clone()
            assert native.swin_post_attn(osp, xc, pw, out=xc) is xc and torch.equal(xc, want), M           # in place
            if M >= 1000:                                                                                    # a tile list: the other tiles keep their rows
                nt = (M + 127) // 128
                tiles = torch.arange(nt, dtype=torch.int32, device=dev)[::3].contiguous()
                L.scp_rc_set_wide(0)
                a = native.swin_post_attn(osp, x.clone(), pw, out=None, tiles=None)
                xa, xb = x.clone(), x.clone()
                native.swin_post_attn(osp, xa, pw, out=xa, tiles=tiles)
                L.scp_rc_set_wide(1)
                native.swin_post_attn(osp, xb, pw, out=xb, tiles=tiles)
                assert torch.equal(xa, xb), M
                keep = torch.ones(nt, dtype=torch.bool)
                keep[::3] = False
                rows = torch.repeat_interleave(keep, 128)[:M].to(dev)
                assert torch.equal(xb[rows], x[rows]) and torch.equal(xb[~rows], a[~rows]), M
        # the automatic choice (by launch size) gives the same rows either way
        L.scp_rc_set_wide(-1)
        x, o = rn(70001, 256), rn(70001, 256)
        big = native.swin_post_attn(native.split_rows(o), x, pw)                                            # 548 tiles: the chain kernel
        for M in (512, 4096, 8192, 16384):
            assert torch.equal(native.swin_post_attn(native.split_rows(o[:M].contiguous()), x[:M].contiguous(), pw), big[:M]), M
    finally:
        L.scp_rc_set_wide(-1)


@pytest.mark.gpu
def test_rowchain_weights_follow_parameter_updates(dev, ehem):
    """The folded / permuted weights of the row-chain kernels are caches keyed on their sources: an in-place parameter update after a
    forward takes effect (the same contract as every other derived weight)."""
    from scp_amd.models.packed import _rowchain_weights
    layer = ehem.swin_self_transformer.layers[0].blocks[0]
    w0 = _rowchain_weights(layer, False)
    assert _rowchain_weights(layer, False) is w0
    with torch.no_grad():
        layer.layernorm_after.weight.mul_(1.5)
    try:
        w1 = _rowchain_weights(layer, False)
        assert w1 is not w0 and not torch.equal(w1["post"].packed, w0["post"].packed)
    finally:
        with torch.no_grad():
            layer.layernorm_after.weight.div_(1.5)


# ---------------------------------------------------------------------------------------------------------------- lattice rows: the proof
def _tied_points(feat, c, dev, rel=4e-6):
    """Points of a window whose 20th and 21st nearest candidates are tied (float64 distances of the features the kernel saw; `rel` x
    (|x_i|^2 + max |x|^2) = the rounding of an fp32 distance evaluation, the same yardstick as _knn_sets_vs_reference): the reference's
    choice among them is an artefact of its top-k (std::partial_sort on CPU, dgcnn.py:10-45), not of the model."""
    x = feat[:c].to(dev).double()
    sq = (x * x).sum(1)
    tied = torch.zeros(c, dtype=torch.bool, device=dev)
    if c <= 20:
        return tied.cpu().numpy()
    for a in range(0, c, 2048):
        d = sq[a:a + 2048, None] + sq[None, :] - 2.0 * (x[a:a + 2048] @ x.T)
        ds = torch.sort(d, 1)[0]
        scale = sq[a:a + 2048] + sq.max()
        tied[a:a + 2048] = (ds[:, 20] - ds[:, 19]) <= rel * scale
    return tied.cpu().numpy()


@pytest.mark.parametrize("name", ["logits_ehem_b2_c256", "logits_ehem_c600", "logits_ehem_c1024", "logits_ehem_c8192", "logits_ehem_c7", "logits_ehem_f17m_c8192"])
def test_every_out_of_tolerance_lattice_row_is_explained_by_a_knn_tie(dev, ehem, monkeypatch, name):
    """Octree positions are lattice points: exactly tied distances at the 20th / 21st neighbour are common, and which of the tied
    candidates the reference keeps is decided by its top-k implementation.  Every row of the lattice fixtures that misses the 1e-3
    tolerance against the reference must be EXPLAINED by such a tie: the row itself, or a row in one of its three neighbour lists, has
    its 20th and 21st candidates tied in one of the three searches (float64 distances of the features the kernel saw).  unexplained == 0
    is asserted; the fraction of rows inside the tolerance is only recorded.  A window WITHOUT any tied point has nothing to explain: there
    the bound against the reference itself is asserted - every row within 1e-3 (round 5; logits_ehem_c8192 is such a window)."""
    z = golden(name)
    data, pos = _ehem_case(z)
    B, c = data.shape[:2]
    spy = _KnnSpy(monkeypatch)
    o1, o2 = _run_packed(ehem, data, pos, dev)
    st, w1, w2 = _want_rows(z)
    assert len(spy.calls) == 3
    cp = -(-(c + (c & 1)) // 512) * 512                        # rows a window owns in the packed layout (even-padded, x512)
    total_bad = unexplained = tied_pts = 0
    for b in range(B):
        e1 = np.abs(o1[b, ::st] - w1[b]).max(1)
        e2 = np.abs(o2[b, ::st] - w2[b]).max(1) if o2.shape[1] else np.zeros(0)
        bad_tok = np.concatenate([2 * st * np.where(e1 > LOGIT_TOL)[0], 2 * st * np.where(e2 > LOGIT_TOL)[0] + 1]).astype(np.int64)
        total_bad += len(bad_tok)
        ce = c + (c & 1)                                       # the window as the model sees it (ehem.py:92-99 pads odd windows)
        tied = np.zeros(ce, bool)
        lists = []
        for feat, idx in spy.calls:
            f, ix = feat[b * cp:b * cp + ce], idx[b * cp:b * cp + ce].numpy().astype(np.int64) - b * cp
            tied |= _tied_points(f, ce, dev)
            lists.append(ix)
        tied_pts += int(tied.sum())
        for t in bad_tok:
            ok = tied[t] or any(tied[np.clip(ix[t], 0, ce - 1)].any() for ix in lists)
            unexplained += 0 if ok else 1
    rows = sum(len(np.abs(o1[b, ::st])) + len(np.abs(o2[b, ::st])) for b in range(B))
    worst = max(float(np.abs(o1[b, ::st] - w1[b]).max()) for b in range(B))
    if o2.shape[1]:
        worst = max(worst, max(float(np.abs(o2[b, ::st] - w2[b]).max()) for b in range(B)))
    parity_record(f"{name}/tie proof", rows=rows, rows_outside_1e3=total_bad, unexplained=unexplained, tied_points=tied_pts, max_dlogit_vs_reference=worst)
    print(f"{name}: {total_bad} of {rows} rows outside 1e-3, unexplained {unexplained} (points with a tie at rank 20 / 21: {tied_pts}), max|dlogit| vs reference {worst:.2e}")
    assert unexplained == 0
    if tied_pts == 0:                 # tie-free window: the product's own neighbour lists are the reference's, so is every logit row
        assert total_bad == 0 and worst <= LOGIT_TOL, (total_bad, worst)


@pytest.mark.parametrize("name", ["logits_ehem_b2_c256", "logits_ehem_c600", "logits_ehem_c1024", "logits_ehem_c8192", "logits_ehem_f17m_c8192"])
def test_given_the_reference_neighbour_choice_every_lattice_row_matches(dev, ehem, monkeypatch, name):
    """The constructive half of the proof: the CPU oracle (the reference's algorithm, torch.topk's own tie-breaking - its logits ARE the
    fixture's) records the neighbour lists of its three searches; the product forward (packed path, every kernel of the encoder) is
    then run with those lists in place of its own kNN results.  EVERY row of the fixture is then reproduced within 1e-3 - 100 %, no
    fraction: the neighbour choice among tied candidates is the only thing that separates the two implementations on lattice inputs."""
    from oracle import models_ref
    from scp_amd import native
    z = golden(name)
    data, pos = _ehem_case(z)
    B, c = data.shape[:2]
    ce = c + (c & 1)
    cp = -(-ce // 512) * 512
    sd = {k: v.cpu() for k, v in ehem.state_dict().items()}
    ref_lists = []

    def record(x, k):
        idx = models_ref.knn_default(x, k)
        ref_lists.append(idx.clone())
        return idx
    models_ref.KNN_OVERRIDE = record
    try:
        with torch.no_grad():
            r1, r2 = models_ref.ehem_forward(sd, data, pos)
    finally:
        models_ref.KNN_OVERRIDE = None
    assert len(ref_lists) == 3 and ref_lists[0].shape == (B, ce, 20)
    st, w1, w2 = _want_rows(z)
    assert _row_err(r1.numpy(), r2.numpy(), st, w1, w2).max() < 2e-4          # the oracle run IS the reference run (float noise of the CPU build)
    calls = []

    def forced(x, ktab, thr0=None):
        lists = ref_lists[len(calls)]
        calls.append(1)
        idx = torch.zeros((x.shape[0], 20), dtype=torch.int32)
        for b in range(B):
            idx[b * cp:b * cp + ce] = (lists[b] + b * cp).to(torch.int32)
        return idx.to(x.device)
    monkeypatch.setattr(native, "knn_topk_packed", forced)
    o1, o2 = _run_packed(ehem, data, pos, dev)
    assert len(calls) == 3
    e = _row_err(o1, o2, st, w1, w2)
    parity_record(f"{name}/reference neighbour lists", rows=e.shape[0], rows_within_1e3=(e.max(1) <= LOGIT_TOL).mean(), max_dlogit=e.max())
    print(f"{name}: with the reference's neighbour lists max|dlogit| = {e.max():.3e}, rows within 1e-3: {100 * (e.max(1) <= LOGIT_TOL).mean():.2f} %")
    assert e.max() <= LOGIT_TOL


@pytest.mark.gpu
def test_launch_brackets_record_inside_the_library(dev):
    """include/scp_debug.h scp_prof_*: while enabled, a bracketed entry point records a hipEvent pair around its launch - tag, algorithmic
    work and a positive duration per launch, in launch order; nothing is recorded outside the `with` block."""
    from scp_amd import native
    g = torch.Generator().manual_seed(3)
    x = torch.randn((4096, 256), generator=g).to(dev)
    w = torch.randn((512, 256), generator=g).to(dev)
    sw = native.SplitWeight(w)
    a = native.split_rows(x)
    torch.cuda.synchronize()
    with native.launch_profile() as p:
        native.linear_split(a, sw)
        native.split_rows(x)
        native.linear_split(a, sw)
    native.linear_split(a, sw)                       # outside: not recorded
    recs = p.records()
    assert [r[0] for r in recs] == ["gemm_split", "split_rows", "gemm_split"]
    assert all(0 < r[1] < 50 for r in recs)
    assert recs[0][2] == 2.0 * 4096 * 512 * 256 and recs[1][2] == 8.0 * 4096 * 256
    with native.launch_profile() as p:
        pass
    assert p.records() == []


@pytest.mark.gpu
@pytest.mark.parametrize("B,c,obj", [(1, 1, False), (2, 300, False), (3, 1024, False), (2, 257, True)])
def test_octattn_embed_kernel_equals_the_torch_input_stage(dev, B, c, obj):
    """scp_octattn_embed (embeddings x 4 ancestors, position Linear, concatenation, sqrt(D), position table, both streams, f16x3 planes -
    one launch) against the sequence of torch operations it replaces + the standalone split pass: identical bits, including levels
    above the cap (shifted and clipped as oct_attention.py:57-61 does)."""
    from scp_amd import native
    from scp_amd.models import OctAttention
    from scp_amd.weights import fill_weights
    cfg = octattn_cfg()
    if obj:
        cfg["train"]["type"] = "obj"
    m = fill_weights(OctAttention(cfg), 3).to(dev)
    g = torch.Generator().manual_seed(B * 1000 + c)
    data = torch.stack((torch.randint(0, 256, (B, c, 4), generator=g), torch.randint(0, 17, (B, c, 4), generator=g),
                        torch.randint(0, 9, (B, c, 4), generator=g)), 3).to(dev)
    pos = torch.rand((B, c, 4, 3), generator=g).to(dev)
    want = m._embed_torch(data, pos, 10 if obj else 12)
    ap = m.abs_pos_enc
    E, pa = native.octattn_embed(data.reshape(B * c, 12).to(torch.uint8), pos.reshape(B * c, 4, 3).contiguous(), c, m.occ_enc.weight, m.level_enc.weight,
                                 m.octant_enc.weight, ap.weight, ap.bias, m.transformer_encoder.position_enc.pe, 10 if obj else 12, cfg.model.max_octree_level)
    assert torch.equal(E.reshape(want.shape), want)
    ref = native.SplitActF16(want.reshape(-1, want.shape[-1]).contiguous())
    assert torch.equal(pa.hi, ref.hi) and torch.equal(pa.lo, ref.lo) and torch.equal(pa.sc, ref.sc) and torch.equal(pa.isc, ref.isc)


@pytest.mark.gpu
def test_pad_tiles_are_skipped_and_real_rows_keep_their_bits(dev):
    """Round 4: the block's second half (scp_swin_post_attn) walks a list of the 128-row tiles that hold a real row and the plane-fed
    attention leaves query tiles of pure window padding: the listed tiles get exactly the bits of a full run, the others are not touched."""
    from scp_amd import native
    g = torch.Generator().manual_seed(11)
    rn = lambda *sh, s=1.0: (torch.randn(sh, generator=g) * s).to(dev)
    lengths = [700, 3, 1030, 512, 129]
    st, _ = native.real_tiles(lengths, dev)
    tiles = st[0]
    Lp = [-(-(c + (c & 1)) // 512) * 512 for c in lengths]
    M = sum(Lp)
    valid = torch.zeros(M, device=dev)
    base = 0
    for c, lp in zip(lengths, Lp):
        valid[base:base + c + (c & 1)] = 1.0
        base += lp
    assert tiles.shape[0] < M // 128
    x, o = rn(M, 256), rn(M, 256)
    pw = native.PostAttnWeights(rn(256, 256, s=0.05), rn(256, s=0.1), 1 + rn(256, s=0.1), rn(256, s=0.1), rn(1024, 256, s=0.05), rn(1024, s=0.1),
                                rn(256, 1024, s=0.03), rn(256, s=0.1))
    osp = native.split_rows(o)
    full = native.swin_post_attn(osp, x, pw)
    xc = x.clone()
    got = native.swin_post_attn(osp, xc, pw, out=xc, tiles=tiles)
    listed = torch.zeros(M // 128, dtype=torch.bool, device=dev)
    listed[tiles.long()] = True
    rows = listed.repeat_interleave(128)
    assert torch.equal(got[rows], full[rows]) and torch.equal(got[~rows], x[~rows])
    assert bool((valid[~rows] == 0).all())                                    # nothing real was left out
    # attention: the same tiles, shifted and unshifted windows
    qkv = rn(M, 768, s=2.0)
    table = rn(1023, 4, s=0.5)
    wtab, b = [], 0
    for lp in Lp:
        wtab += [[b, lp]] * (lp // 512)
        b += lp
    wtab = torch.tensor(wtab, dtype=torch.int32, device=dev)
    kv = native.KvPlanes(qkv[:, 256:512], qkv[:, 512:])
    for shift in (0, 256):
        a = native.swin_attention_packed_planes(qkv[:, :256], kv, table, wtab, shift)
        bb = native.swin_attention_packed_planes(qkv[:, :256], kv, table, wtab, shift, valid=valid)
        assert torch.equal(a[rows], bb[rows])


@pytest.mark.gpu
def test_octattn_attention_takes_column_slices_of_one_projection(dev):
    """scp_octattn_attention_f16x3 with k / v as column slices of one stacked key | value projection output (row stride 1280) gives the
    bits of the call on dense copies (round 4: the projection is one N = 1280 product instead of two N = 600 ones)."""
    from scp_amd import native
    g = torch.Generator().manual_seed(5)
    B, c, D = 3, 700, 600
    kv = (torch.randn((2, B, c, 1280), generator=g) * 3).to(dev)
    q_u = torch.randn((B, c, D), generator=g).to(dev)
    key, val = kv[..., :D], kv[..., 640:640 + D]
    a, au = native.octattn_attention(q_u, key[0], key[1], val[0], val[1], 4)
    b, bu = native.octattn_attention(q_u, key[0].contiguous(), key[1].contiguous(), val[0].contiguous(), val[1].contiguous(), 4)
    assert torch.equal(a, b) and torch.equal(au, bu)


@pytest.mark.gpu
def test_phase2_prepared_ahead_has_the_bits_of_the_plain_phase2(dev, ehem):
    """Round 5 (decoder): pre_attn_mlp(a1) and the cross transformer's query stream do not depend on the decoded even symbols -
    ehem_phase2_prepare computes them for a whole level in one packed pass (on a side stream in the decoder) and phase 2 takes a window's rows
    of the result.  Same kernels on the same rows: the odd-node logits are bit-identical to the plain phase 2, level-wide and per window."""
    from scp_amd.models.packed import PackedPlan, ehem_phase1_packed, ehem_phase2_packed, ehem_phase2_prepare, phase2_prep_window
    from scp_amd import native
    g = torch.Generator().manual_seed(3)
    lengths = [2049, 700, 1, 1300]
    T = sum(lengths)
    ctx = torch.randint(0, 9, (T, 12), generator=g).to(torch.uint8)
    ctx[:, 2::3] = torch.randint(0, 255, (T, 4), generator=g).to(torch.uint8)
    ctx[:, 0::3] = 7
    ctx, pos = ctx.to(dev), torch.rand((T, 3), generator=g).to(dev)
    plan = PackedPlan(lengths, device=dev)
    _, st = ehem_phase1_packed(ehem, ctx, pos, plan)
    want = ehem_phase2_packed(ehem, st, plan)
    prep = ehem_phase2_prepare(ehem, st, plan)
    got = ehem_phase2_packed(ehem, st, plan, prep=prep)
    assert torch.equal(got, want)
    # one window at a time on its rows of the level-wide state and of the level-wide preparation (what FrameDecoder._decode_level does);
    # a preparation is consumed by the phase 2 that takes it (the blocks run in place on `pre`): a fresh one
    prep = ehem_phase2_prepare(ehem, st, plan)
    nst = len(ehem.swin_cross_transformer.layers)
    bases, q0, o0 = [0] * nst, 0, 0
    for c in lengths:
        rows, L = [], (c + (c & 1)) // 2
        for _ in range(nst):
            rows.append(-(-L // 512) * 512)
            L = (L + 1) // 2
        if c > 1:
            pw = PackedPlan([c], device=dev)
            stw = dict(a1=st["a1"][q0:q0 + rows[0]], a2=st["a2"][q0:q0 + rows[0]], pre_occ=st["pre_occ"][q0:q0 + rows[0]])
            one = ehem_phase2_packed(ehem, stw, pw, prep=phase2_prep_window(prep, bases, rows))
            assert torch.equal(one, want[o0:o0 + c // 2])
        q0 += rows[0]
        o0 += c // 2
        bases = [b + r for b, r in zip(bases, rows)]


@pytest.mark.gpu
@pytest.mark.parametrize("polar,last", [(True, False), (True, True), (False, False)])
def test_decode_expand_kernel_equals_the_torch_construction(dev, polar, last):
    """scp_decode_expand (the decoder's breadth-first regeneration, decode_ehem_mullevel.py:100-130, in one launch) against the torch index
    construction it replaced: children in (parent, digit) order, shifted ancestor windows, origins, and the next level's context rows and
    double-precision normalised positions - bit for bit."""
    from scp_amd import native
    g = torch.Generator().manual_seed(5)
    n, L, depth, lidar = 3000, 9, 13, 12
    sym = torch.randint(0, 255, (n,), generator=g)
    sym[-1] = -1                                                      # the dropped last node of a multi-level shell: no children
    pos = torch.randint(0, 2 ** 12, (n, 3), generator=g) << (depth - L + 1)
    anc = torch.randint(0, 255, (n, 9), generator=g)
    anc[:, 0::3] = torch.tensor([L - 3, L - 2, L - 1])
    octant = torch.randint(1, 9, (n,), generator=g)
    mn, mx = -3.0, 8191.5
    eps = 0.0 if last else 1e-9
    lvn = min(L + 1, lidar) if last else L + 1
    clamp = lidar if last else 255
    den = (mx - mn + eps) if polar else float(2 ** depth)
    # the torch construction (FrameDecoder._decode_tree until round 5)
    occ = sym + 1
    bits = ((occ[:, None] >> torch.arange(8)[None]) & 1).bool()
    par, dig = torch.nonzero(bits, as_tuple=True)
    sh = depth - L
    cpos = pos[par] + torch.stack((((dig >> 2) & 1) << sh, ((dig >> 1) & 1) << sh, (dig & 1) << sh), 1)
    canc = torch.cat((anc[par][:, 3:], torch.stack((torch.full_like(par, L), octant[par], sym[par]), 1)), 1)
    coct = dig + 1
    a = canc.clone()
    a[:, 0::3] = torch.clamp(a[:, 0::3], max=clamp)
    own = torch.stack((torch.full_like(par, lvn), coct, torch.full_like(par, 255)), 1)
    cctx = torch.cat((a, own), 1).to(torch.uint8)
    cposn = (((cpos.double() - mn) / den) if polar else (cpos.double() / den)).float()
    got = native.decode_expand(sym.to(dev), pos.to(torch.int32).to(dev), anc.to(torch.uint8).to(dev), octant.to(torch.uint8).to(dev), L, sh, lvn, clamp,
                               polar, mn if polar else 0.0, den)
    occ8, gpos, ganc, goct, gctx, gposn = [x.cpu() for x in got]
    assert torch.equal(occ8, occ.to(torch.uint8)) and torch.equal(gpos.long(), cpos) and torch.equal(ganc, canc.to(torch.uint8))
    assert torch.equal(goct, coct.to(torch.uint8)) and torch.equal(gctx, cctx)
    assert torch.equal(gposn.view(torch.int32), cposn.view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,cfg", [(1, 255, 600, 0), (300, 600, 300, 0), (4100, 2400, 600, 0), (4100, 1280, 600, 1), (513, 240, 80, 2), (70000, 600, 600, 3)])
@pytest.mark.parametrize("act", [None, "relu"])
def test_linear_split_f16_epilogue_maxima(dev, M, N, K, cfg, act):
    """Round 5: scp_linear_split_f16_max takes the maxima the NEXT f16x3 layer scales by in its epilogue - the output is unchanged bit for bit,
    row_max holds the bit patterns of max |out[m, :]| (so RowScales.from_max equals the scales of a pass over the output: scp_row_scale_f16),
    col_max the maximum over a row / column range (max |v| of OctAttention's stacked key | value projection)."""
    from scp_amd import native, ops
    g = torch.Generator().manual_seed(M + 7 * N + K)
    x = torch.randn((M, K), generator=g) * 25.0 * torch.pow(10.0, torch.randint(-4, 5, (M, 1), generator=g).float())
    w = torch.randn((N, K), generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    sw, pa, a = ops._split16(wd), native.SplitActF16(xd), ops._ACT[act]
    want = native.linear_split_f16(pa, sw, bd, a, cfg=cfg)
    rm = torch.zeros(M, dtype=torch.int32, device=dev)
    cm = torch.zeros(1, dtype=torch.int32, device=dev)
    lo, hi, rows = N // 3, N - 5 if N > 8 else N, max(1, M // 2)
    got = native.linear_split_f16(pa, sw, bd, a, cfg=cfg, row_max=rm, col_max=(cm, lo, hi, rows))
    assert torch.equal(got, want)
    assert torch.equal(rm.view(torch.float32), want.abs().amax(1))
    assert float(cm.view(torch.float32)) == float(want[:rows, lo:hi].abs().max())
    if N % 4 == 0:                                                     # (scp_row_scale_f16 reads rows as float4)
        rs, rs2 = native.RowScales(want), native.RowScales.from_max(rm)
        assert torch.equal(rs.sc, rs2.sc) and torch.equal(rs.isc, rs2.isc)
    only = torch.zeros(M, dtype=torch.int32, device=dev)
    assert torch.equal(native.linear_split_f16(pa, sw, bd, a, cfg=cfg, row_max=only), want) and torch.equal(only, rm)


@pytest.mark.gpu
def test_octattn_attention_with_the_value_maximum_given(dev):
    """scp_octattn_attention_f16x3_vmax (max |v| from the projection's epilogue) = scp_octattn_attention_f16x3 (its own pass over v), bit for bit."""
    from scp_amd import native
    g = torch.Generator().manual_seed(9)
    B, c, H, D = 3, 700, 4, 600
    q, k, ku, v, vu = [(torch.randn((B, c, D), generator=g) * s).to(dev) for s in (1.0, 1.0, 1.0, 3.0, 3.0)]
    a0, a1 = native.octattn_attention(q, k, ku, v, vu, H)
    vm = v.abs().max().reshape(1).view(torch.int32)
    b0, b1 = native.octattn_attention(q, k, ku, v, vu, H, vmax=vm)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
