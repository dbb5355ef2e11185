"""Host-side logic that needs no GPU: window plans, packed-layout index maps, CLI plumbing, file readers."""
import os

import numpy as np
import pytest
import torch


def test_encode_plan_matches_oracle_coding_order(orc):
    from scp_amd.encoder import EncodePlan, level_qs
    for sizes in ([1], [1, 6, 20, 8193, 1, 17000, 3], [2, 2, 2], [8192, 8192, 1]):
        plan = EncodePlan(sizes, 8192)
        _, want = orc.ehem_coding_plan(sizes, 8192, mullevel=True)
        assert np.array_equal(plan.coding_order(), want)
        assert np.array_equal(plan.coding_order_device(torch.device("cpu")).numpy(), want)
        assert sum(w[1] for w in plan.windows) == sum(sizes)
    assert level_qs("kitti", 16) == 400 / (2 ** 16 - 1) and level_qs("ford", 17) == 2.0
    groups = EncodePlan([8192] * 10 + [5], 8192).groups(8)
    assert [(c, len(ws)) for c, ws in groups] == [(8192, 8), (8192, 2), (5, 1)]


def _loop_maps(c):
    """Straightforward per-window construction of the packed layout (what models/packed.py vectorises)."""
    c = np.asarray(c)
    e = c + (c & 1)

    def layout(L):
        Lp = (L + 511) // 512 * 512
        return L, Lp, np.concatenate(([0], np.cumsum(Lp)[:-1])), int(Lp.sum())
    lay = [layout(e)]
    for _ in range(4):
        lay.append(layout((lay[-1][0] + 1) // 2))
    return lay


@pytest.mark.parametrize("lengths", [[1], [1, 6, 20, 56, 208, 1372, 5605, 8192, 4938, 7, 2, 513], [8192, 8192], [3, 3, 3]])
def test_packed_plan_maps(lengths):
    from scp_amd.models.packed import PackedPlan
    p = PackedPlan(lengths, device=torch.device("cpu"))   # pure index arithmetic: runs on any torch device
    c = np.asarray(lengths)
    lay = _loop_maps(c)
    for s, l in enumerate(p.self_layouts):
        L, Lp, base, rows = lay[s]
        assert np.array_equal(l.L.numpy(), L) and np.array_equal(l.base.numpy(), base) and l.rows == rows
        tab = p.d["self_tab"][s].numpy()
        assert tab.shape == (rows // 512, 2)
        for i in range(len(c)):
            ch = slice(base[i] // 512, (base[i] + Lp[i]) // 512)
            assert (tab[ch, 0] == base[i]).all() and (tab[ch, 1] == Lp[i]).all()
        v = p.d["self_valid"][s].numpy()[:, 0]
        want = np.zeros(rows)
        for i in range(len(c)):
            want[base[i]:base[i] + L[i]] = 1
        assert np.array_equal(v, want)
    # input map, merge maps, concat maps, outputs
    L0, Lp0, b0, rows0 = lay[0]
    starts = np.concatenate(([0], np.cumsum(c)[:-1]))
    inmap = np.full(rows0, c.sum())
    for i in range(len(c)):
        inmap[b0[i]:b0[i] + c[i]] = starts[i] + np.arange(c[i])
    assert np.array_equal(p.d["inmap"].numpy(), inmap)
    for s in range(4):
        (L, Lp, b, rows), (L2, Lp2, b2, rows2) = lay[s], lay[s + 1]
        ev, od = np.full(rows2, rows), np.full(rows2, rows)
        for i in range(len(c)):
            t = np.arange(L2[i])
            ev[b2[i] + t] = b[i] + 2 * t
            od[b2[i] + t] = np.where(2 * t + 1 < L[i], b[i] + 2 * t + 1, rows)
        assert np.array_equal(p.d["self_merge"][s][0].numpy(), ev) and np.array_equal(p.d["self_merge"][s][1].numpy(), od)
        m = np.zeros(rows0, np.int64)
        for i in range(len(c)):
            t = np.arange(L0[i])
            m[b0[i] + t] = b2[i] + (t >> (s + 1))
        assert np.array_equal(p.d["self_concat"][s].numpy(), m)
    ne, no = (c + 1) // 2, c // 2
    coded = np.concatenate(([0], np.cumsum(c)[:-1]))
    ed = np.concatenate([coded[i] + np.arange(ne[i]) for i in range(len(c))])
    od = np.concatenate([coded[i] + ne[i] + np.arange(no[i]) for i in range(len(c))]) if no.sum() else np.zeros(0)
    assert np.array_equal(p.d["even_dst"].numpy(), ed) and np.array_equal(p.d["odd_dst"].numpy(), od)
    assert sorted(np.concatenate([ed, od]).tolist()) == list(range(int(c.sum())))
    assert p.d["knn_tab"].numpy()[:, 1].max() == (c + (c & 1)).max()


def test_readers_and_cli_plumbing(tmp_path):
    from scp_amd.cli import expand_files, get_args, load_cfg
    from scp_amd.data_preproc import pt
    from scp_amd.synth import synth_frame, write_kitti_bin
    xyz = synth_frame(0)[:100]
    f = tmp_path / "a.bin"
    write_kitti_bin(str(f), xyz)
    assert np.array_equal(pt.ptread(str(f)), xyz)
    ply = tmp_path / "b.ply"
    pt.write_ply_data(str(ply), xyz)
    back = pt.ptread(str(ply))
    assert back.dtype == np.float32 and np.allclose(back, xyz)
    with pytest.raises(Exception):
        pt.ptread(str(tmp_path / "missing.bin"))
    assert expand_files([str(tmp_path)]) == sorted([str(f), str(ply)])
    a = get_args(["--type", "kitti", "--lidar_level", "16", "--spher", "--test_files", "x.bin", "--preproc_path", "pp/"], mullevel=True)
    assert a.type == "kitti" and a.lidar_level == 16 and a.spher and not a.cylin and a.preproc_path == "pp/" and not hasattr(a, "spher_circle")
    assert hasattr(get_args([], mullevel=False), "spher_circle")
    cfg = load_cfg("", "EHEM")
    assert cfg.model.context_size == 8192 and cfg.model.max_level == 19 and cfg.model.token_num == 255
    run = tmp_path / "run" / ".hydra"
    run.mkdir(parents=True)
    (run / "config.yaml").write_text("model:\n  class_name: OctAttention\n  context_size: 1024\n  token_num: 255\ntrain:\n  type: kitti\n")
    cfg = load_cfg(str(tmp_path / "run" / "ckpt" / "epoch=1.ckpt"))
    assert cfg.model.class_name == "OctAttention" and cfg.model.context_size == 1024 and cfg.data.extra_pos is False


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the product modules fail loudly on CPU inputs."""
    from cfgs import ehem_cfg
    from scp_amd import native
    from scp_amd.models import EHEM
    m = EHEM(ehem_cfg())
    with pytest.raises(native.ScpError):
        m(torch.zeros((1, 4, 4, 3), dtype=torch.int64), torch.zeros((1, 3, 4)))
    with pytest.raises(native.ScpError):
        native.knn_topk(torch.zeros((1, 4, 3)), 2)


def test_training_dataset_matches_reference(tmp_path):
    """scp_amd.dataloaders.EHEMDataset reproduces the reference class item for item (fixture: tests/golden/trainset.npz,
    produced by dataloaders/ehem_dataset.py with the same files, indices and torch seed)."""
    from types import SimpleNamespace
    from conftest import golden
    from scp_amd.dataloaders import EHEMDataset
    z = golden("trainset")
    np.save(tmp_path / "a_100.npy", z["rec_a"])
    np.save(tmp_path / "b_57.npy", z["rec_b"])
    ds = EHEMDataset(SimpleNamespace(root=str(tmp_path / "*.npy"), context_size=16, extra_pos=False))
    assert len(ds) == int(z["length"])
    torch.manual_seed(7)
    for k, i in enumerate((0, 1, 2, 3, 4, 5, 6, 7, 8)):
        d, p, lab = ds[i]
        assert np.array_equal(d, z["data"][k]) and np.array_equal(p, z["pos"][k]) and np.array_equal(lab, z["label"][k])
        assert d.dtype == np.int64 and p.dtype == np.float32 and p.shape == (3, 16)


def test_decoder_side_info_parsing(tmp_path):
    """decode_ehem.py:20-27 `extract_info` restated: coordinate system and (levels, bin_num, z_offset) from the file name, (min, max)
    pairs from the `.dat`; the `.scp.json` extension round-trips; per-shell quantisation steps follow the dataset type."""
    import torch
    from scp_amd import decoder as D
    f = str(tmp_path / "seq07000012_cylin_36_3279_-1.bin")
    open(f, "wb").write(b"\x00")
    torch.save(torch.Tensor(np.array([[0, 5], [1, 9]], np.float32)), f + ".dat")
    spher, cylin, pos_mm, n_levels, bin_num, z = D.extract_info(f)
    assert (spher, cylin, n_levels, bin_num, z) == (False, True, 36, 3279, -1) and pos_mm.tolist() == [[0, 5], [1, 9]]
    g = str(tmp_path / "000001_12_0_0.bin")          # Cartesian: no .dat is read
    open(g, "wb").write(b"\x00")
    e = D.extract_info(g)
    assert e[:2] == (False, False) and e[3:] == (12, 0, 0) and len(e[2]) == 0
    assert D.read_sidecar(f) is None

    class Enc:
        data_type = "kitti"; lidar_level = 14; mullevel = True; spher = False; cylin = True
    res = dict(n_points=120000, n_nodes=5, bin_num=3279.0, bin_nums=[3279.0, 6557.0, 13113.0], z_offset=-1.759)
    import scp_amd.native as native
    try:
        side = D.write_sidecar(f, Enc, res, "EHEM")
    except native.ScpError:
        pytest.skip("library not built")
    back = D.read_sidecar(f)
    assert back == side and back["bin_nums"] == [3279.0, 6557.0, 13113.0] and back["z_offset"] == -1.759 and back["profile"].startswith("ehem/")
    assert D.shell_qs("kitti", 16, True) == [400 / (2 ** 16 - 1), 400 / (2 ** 17 - 1), 400 / (2 ** 18 - 1)]
    assert D.shell_qs("ford", 17, True) == [2, 1, 0.5] and D.shell_qs("kitti", 12, False) == [400 / 4095]


def test_obj_quantisation_matches_proc_pc_defaults():
    """`--type obj`: proc_pc's defaults (data_preprocess.py:13-70): offset = per-axis minimum, qs = 1, numpy's round half to even on the
    float32 difference; MVUB names swap / negate axes first."""
    import torch
    from scp_amd.cli import obj_ints
    rng = np.random.default_rng(2)
    p = (rng.random((500, 3)) * 300 - 40).astype(np.float32)
    p[:7] = np.round(p[:7]) + 0.5                                       # exact halves: round half to even
    want = np.round(p - np.min(p, 0)).astype(np.int32)                  # the reference's arithmetic (float32 - float32, np.round)
    q, off = obj_ints(p, "thing_vox9.ply", torch.device("cpu"))
    assert np.array_equal(q.numpy(), want) and np.array_equal(np.float32(off), np.min(p, 0))
    r = p[:, [0, 2, 1]].copy(); r[:, 2] = -r[:, 2]
    want_r = np.round(r - np.min(r, 0)).astype(np.int32)
    assert np.array_equal(obj_ints(p, "data/mvub/phil9/frame0001.ply", torch.device("cpu"))[0].numpy(), want_r)


def test_host_quantize_equals_the_reference_integers_on_full_frames():
    """Strict-identity switch (--host_transform / SCP_XFORM=numpy): scp_amd.data_preproc.data_preprocess.host_quantize is the
    reference's float -> integer step (data_preprocess.py:40-70) in numpy.  On the 120 000-point frame its integers equal the ones the
    reference itself produced in the build container (tests/golden/frame_ints.npz) for every configuration: 0 differing points."""
    from scp_amd import native
    from scp_amd.data_preproc.data_preprocess import host_quantize
    from scp_amd.synth import synth_frame
    from conftest import GOLDEN
    ints = np.load(os.path.join(GOLDEN, "frame_ints.npz"))
    xyz = synth_frame(0)
    for key, mode, L, off in (("q_spher_L12", native.SPHER, 12, -200.0), ("q_spher_L16", native.SPHER, 16, 0.0), ("q_spher_L17", native.SPHER, 17, 0.0),
                              ("q_spher_L18", native.SPHER, 18, 0.0), ("q_cylin_L14", native.CYLIN, 14, -200.0), ("q_cart_L12", native.CART, 12, -200.0)):
        q, info = host_quantize(xyz, mode, 400 / (2 ** L - 1), off)
        assert q.dtype == np.int32 and int((q != ints[key]).any(1).sum()) == 0, key
    q, info = host_quantize(xyz, native.SPHER, 400 / (2 ** 12 - 1))
    assert info.bin_num == 820.0 and info.offset == [0.0, 0.0, 0.0] and abs(info.qs[1] - 2 * np.pi / 819) < 1e-9      # float32 step: numpy >= 2 keeps bin_num float32
    q, info = host_quantize(xyz, native.CYLIN, 400 / (2 ** 14 - 1))
    assert info.bin_num == 3279.0 and abs(info.offset[2] - (-1.75919378)) < 1e-6


def test_decode_cli_finds_exactly_the_stream_the_encoder_wrote(tmp_path):
    """decode_ehem*.py stream lookup: exact `<sequence><frame>` / `<stem>` names as the encoder writes them; substring matches of other
    sequences (`11000001` vs `000001`) and ambiguous directories are errors, not a silent first match."""
    from scp_amd import native
    from scp_amd.cli import find_stream
    out = tmp_path / "out"
    out.mkdir()
    for n in ("11000001_spher_10_820_0.bin", "12000001_spher_10_821_0.bin", "00000010_spher_10_800_0.bin", "frame7_cylin_12_3279_-1.bin",
              "cart3_12_0_-200.bin", "thing_vox6.bin", "11000001_spher_10_820_0.bin.dat"):
        (out / n).write_bytes(b"")
    root = str(out) + "/"
    # (stream name, the stem that matched): the stem names the decoder's .ply - no string surgery on the stream name
    assert find_stream(root, "/data/kitti/11/000001.bin") == ("11000001_spher_10_820_0.bin", "11000001")
    assert find_stream(root, "/data/kitti/12/000001.bin") == ("12000001_spher_10_821_0.bin", "12000001")
    assert find_stream(root, "/data/kitti/00/000010.bin") == ("00000010_spher_10_800_0.bin", "00000010")
    assert find_stream(root, "ford/frame7.ply") == ("frame7_cylin_12_3279_-1.bin", "frame7")
    assert find_stream(root, "x/cart3.bin") == ("cart3_12_0_-200.bin", "cart3") and find_stream(root, "thing_vox6.ply") == ("thing_vox6.bin", "thing_vox6")
    # an un-suffixed stream whose own stem holds three underscores keeps its whole stem (it used to be cut to `a`)
    (out / "a_b_c_d.bin").write_bytes(b"")
    assert find_stream(root, "objs/a_b_c_d.ply") == ("a_b_c_d.bin", "a_b_c_d")
    with pytest.raises(native.ScpError):
        find_stream(root, "/data/kitti/13/000001.bin")          # `000001` is a substring of three names, a match of none
    (out / "11000001_spher_10_999_0.bin").write_bytes(b"")
    with pytest.raises(native.ScpError):
        find_stream(root, "/data/kitti/11/000001.bin")


def test_packed_chunks_respect_the_padded_row_limit():
    """A packed forward addresses its K / V planes through one 32-bit buffer resource: chunks are cut on real tokens AND on rows of the
    padded layout (tail windows of a few nodes cost 512 rows each; a token bound alone overflowed the limit for batches of
    small frames and for max_tokens above 1M)."""
    from scp_amd.encoder import MAX_PACKED_ROWS, EncodePlan, chunk_windows
    sizes = [1, 6, 20, 56, 208, 1372, 5605, 21322, 34427, 52308] * 16            # 16 level-12-like frames back to back
    ws = EncodePlan(sizes, 8192).windows

    def rows(i, j):
        return sum(-(-(w[1] + (w[1] & 1)) // 512) * 512 for w in ws[i:j])
    for max_tokens in (1_000_000, 5_000_000, 100):
        ch = chunk_windows(ws, max_tokens)
        assert ch[0][0] == 0 and ch[-1][1] == len(ws) and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))      # a partition, in order
        for i, j in ch:
            assert j > i and rows(i, j) <= MAX_PACKED_ROWS
            assert sum(w[1] for w in ws[i:j]) <= max_tokens or j == i + 1
    assert len(chunk_windows(ws, 5_000_000)) >= 2 and 4 * MAX_PACKED_ROWS * 512 <= 0x7fffffff
    assert chunk_windows(ws[:5], 1_000_000) == [(0, 5)]


def test_frame_flop_formula_matches_the_survey():
    """bench.py prices a frame with SURVEY.md 8d's formulas: 310.2 GFLOP per full EHEM window (40.0 of them attention), 27.9 per
    OctAttention window (11.3 attention)."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("scp_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    assert abs(b.ehem_window_flops(8192) / 1e9 - 310.2) < 0.1 and abs(b.ehem_window_flops(8192, True) / 1e9 - 40.0) < 0.05
    assert abs(b.octattn_window_flops(1024) / 1e9 - 27.9) < 0.1 and abs(b.octattn_window_flops(1024, True) / 1e9 - 11.3) < 0.05
    assert b.ehem_window_flops(1) == b.ehem_window_flops(2) and b.ehem_window_flops(600) < b.ehem_window_flops(1024)


@pytest.mark.parametrize("lengths", [[1], [1, 6, 20, 56, 208, 1372, 5605, 8192, 4938, 7, 2, 513], [8192, 8192], [3, 3, 3], [129, 257, 1025]])
def test_real_tile_lists_match_the_valid_masks(lengths):
    """native.real_tiles (host arithmetic on the window lengths) lists exactly the 128-row tiles of every Swin stage whose first row is a
    real row of the packed layout - what the valid masks of the plan say; every other tile is pure window padding."""
    from scp_amd import native
    from scp_amd.models.packed import PackedPlan
    p = PackedPlan(lengths, device=torch.device("cpu"))
    st, ct = native.real_tiles(lengths, torch.device("cpu"))
    assert len(st) == 5 and len(ct) == 4
    for tiles, valid in list(zip(st, p.d["self_valid"])) + list(zip(ct, p.d["cross_valid"])):
        v = valid.reshape(-1)
        want = torch.nonzero(v[::128]).reshape(-1).to(torch.int32)
        assert torch.equal(tiles, want)
        # a tile is either listed or holds no real row at all
        allpad = v.reshape(-1, 128).sum(1) == 0
        assert torch.equal(torch.nonzero(~allpad).reshape(-1).to(torch.int32), tiles)
