"""The CPU oracle (oracle/) against the golden vectors produced by the reference itself."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


# ------------------------------------------------------------------ a2/a3 transform + quantiser
@pytest.mark.parametrize("seed", [0, 1])
def test_transform_and_quantiser(orc, seed):
    z = golden(f"xform_s{seed}")
    xyz = z["xyz"]
    for mode in ("spher", "cylin", "cart"):
        for L in (12, 14, 16, 18):
            tr, bin_num, qsv, off, pt = orc.quantise(xyz, 400 / (2 ** L - 1), mode)
            assert bin_num == float(z[f"{mode}_L{L}_bin"])
            if mode != "cart":
                # numpy's SIMD atan2/acos are not reproducible across CPUs (SURVEY.md §7 hard part 1):
                # the transform must agree to 2 ulp and the integers may differ for a handful of points
                ref = z[f"{mode}_tr"]
                ulp = np.abs(tr.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
                assert ulp.max() <= 2
            diff = (pt.astype(np.int64) != z[f"{mode}_L{L}_q"]).any(1).sum()
            assert diff <= 40, (mode, L, diff)


# ------------------------------------------------------------------ a4 octree + a6 records
@pytest.mark.parametrize("name", names("oct_"))
def test_octree_vs_reference_so(orc, name):
    z = golden(name)
    t = orc.octree_build(z["pts"])
    assert t.depth == int(z["depth"])
    assert np.array_equal(t.codes, z["codes"])
    assert np.array_equal(t.level, z["level"])
    assert np.array_equal(t.octant, z["octant"])
    assert np.array_equal(t.occ, z["occ"])
    assert np.array_equal(t.parent, z["parent"])
    assert np.array_equal(t.pos, z["pos"])
    assert np.array_equal(t.krecords(), z["krec"])
    # order invariance
    perm = np.random.default_rng(0).permutation(len(z["pts"]))
    t2 = orc.octree_build(z["pts"][perm])
    assert np.array_equal(t2.codes, z["codes"])


def test_full_size_frames_vs_reference_checksums(orc):
    """The oracle at BASELINE size: the reference quantiser's integers of the 120k-point frame (frame_ints.npz) -> the node counts
    and sha256 of occupancy stream / K-records the reference's own builders produced (frame_facts.json): same-level L12 / L16 /
    cylindrical L14 / Cartesian L12 through gen_octree's twin, the three L16-mullevel shells through mullevel_gen_octree's."""
    facts = json.load(open(os.path.join(GOLDEN, "frame_facts.json")))
    z = golden("frame_ints")
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    for key, name in (("L12-s", "q_spher_L12"), ("L16-s", "q_spher_L16"), ("C14", "q_cylin_L14"), ("L12-c", "q_cart_L12")):
        f = facts[key]
        t = orc.octree_build(np.unique(z[name], axis=0))
        assert t.n == f["N"] and t.depth == f["D"] and np.bincount(t.level)[1:].tolist() == f["per_level"]
        assert sha(t.occ) == f["codes_sha"]
        if "krec_sha_i32" in f:
            assert sha(t.krecords().astype(np.int32)) == f["krec_sha_i32"]
    for k, (L, path) in enumerate(((16, [0, 0]), (17, [0, 1]), (18, [1]))):
        f = facts["L16-m"][k]
        q = z[f"q_spher_L{L}"]
        _, idx = np.unique(q, axis=0, return_index=True)
        t = orc.octree_build(q[np.sort(idx)], path)
        assert t.n == f["records"] + 1 and np.bincount(t.level)[1:].tolist() == f["per_level"]
        assert sha(t.occ) == f["codes_sha"] and sha(t.krecords(True).astype(np.int32)) == f["krec_sha_i32"]


def test_ford_L17_mullevel_frame_vs_reference_checksums(orc):
    """BASELINE configs[3] at full size: the Ford-like frame (integer millimetres), level 17 --spher --mullevel (qs 2 / 1 / 0.5 mm).
    frame_facts.json["F17-m"] / frame_ints.npz q_spher_ford_L17..19 were written by running the reference's own `mul_proc_pc`
    (data_preprocess.py:95-167 with the arguments of encode_dataset_ehem_mullevel.py:154-186, data_type 'ford'; make_golden.py facts_ford).
    The oracle's quantiser must produce those integers from the floats (0 differing points: the Ford values are whole millimetres, the
    square root is exact up to float32 rounding and only arctan2 / arccos carry numpy's SIMD behaviour of the generating container - the
    count is asserted), and from the reference's integers the oracle's octree gives the reference's code streams and records."""
    from scp_amd.synth import ford_like, synth_frame
    facts = json.load(open(os.path.join(GOLDEN, "frame_facts.json")))["F17-m"]
    z = golden("frame_ints")
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    xyz = ford_like(synth_frame(0))
    for k, path in enumerate(([0, 0], [0, 1], [1])):
        f = facts[k]
        q = z[f"q_spher_ford_L{17 + k}"]
        assert f["qs"] == orc.ford_qs(17 + k)
        _, bin_num, _, _, pt = orc.quantise(xyz, orc.ford_qs(17 + k), "spher", cart_offset=0)
        assert bin_num == f["bin_num"]
        ndiff = int((pt.astype(np.int32) != q).any(1).sum())
        assert ndiff <= 8, (k, ndiff)                  # numpy float32 arctan2 / arccos are CPU dependent; 0 in the build container
        _, idx = np.unique(q, axis=0, return_index=True)
        t = orc.octree_build(q[np.sort(idx)], path)
        assert t.n == f["records"] + 1 and t.depth == f["D"] and np.bincount(t.level)[1:].tolist() == f["per_level"]
        assert sha(t.occ) == f["codes_sha"] and sha(t.krecords(True).astype(np.int32)) == f["krec_sha_i32"]


def test_octree_depth0_is_an_error(orc):
    with pytest.raises(ValueError):
        orc.octree_build(np.zeros((1, 3), np.int64))


@pytest.mark.parametrize("name", names("octmul_"))
def test_mullevel_octree(orc, name):
    z = golden(name)
    for tag, path in (("00", [0, 0]), ("01", [0, 1]), ("1", [1])):
        if f"p{tag}_empty" in z:
            with pytest.raises(ValueError):
                orc.octree_build(z["pts"], path)
            continue
        t = orc.octree_build(z["pts"], path)
        assert t.depth == int(z[f"p{tag}_depth"])
        assert np.array_equal(t.codes, z[f"p{tag}_codes"])
        assert np.array_equal(t.level, z[f"p{tag}_level"])
        assert np.array_equal(t.octant, z[f"p{tag}_octant"])
        assert np.array_equal(t.parent, z[f"p{tag}_parent"])
        assert np.array_equal(t.pos, z[f"p{tag}_pos"])
        assert np.array_equal(t.krecords(drop_last=True), z[f"p{tag}_krec"])
        assert np.array_equal(orc.deoctree(t.codes), z[f"p{tag}_deoct"])


# ------------------------------------------------------------------ a8/a9 context assembly
@pytest.mark.parametrize("name,mode,L", [("ctx_ehem_spher_L12", "spher", 12), ("ctx_ehem_cylin_L12", "cylin", 12),
                                         ("ctx_ehem_cart_L10", "cart", 10)])
def test_ehem_context_same_level(orc, name, mode, L):
    z = golden(name)
    r = orc.proc_pc(z["xyz"], 400 / (2 ** L - 1), mode)
    assert r["bin_num"] == float(z["bin_num"])
    ids, poss, pos_mm, data, oct_seq = orc.ehem_level_split(r["records"], L, polar=mode != "cart")
    assert len(data) == int(z["n_levels"])
    assert np.array_equal(oct_seq, z["oct_seq"])
    for l in range(len(data)):
        assert np.array_equal(data[l], z[f"data{l}"])
        assert np.array_equal(poss[l], z[f"pos{l}"]) and poss[l].dtype == np.float32
        assert np.array_equal(ids[l], z[f"ids{l}"])
    if mode != "cart":
        assert np.array_equal(np.array(pos_mm), z["pos_mm"])
    assert np.allclose(r["quant_pc"], z["quant_pc"], atol=1e-4)


def test_ehem_context_mullevel(orc):
    z = golden("ctx_ehem_mul_spher_L14")
    shells = orc.mullevel_shells(z["xyz"], 14, "spher")
    ids, poss, pos_mm, data, oct_seq = orc.ehem_mullevel_context([s["records"] for s in shells], 14)
    assert len(data) == int(z["n_levels"])
    assert np.array_equal(oct_seq, z["oct_seq"])
    for l in range(len(data)):
        assert np.array_equal(data[l], z[f"data{l}"])
        assert np.array_equal(poss[l], z[f"pos{l}"])
        assert np.array_equal(ids[l], z[f"ids{l}"])
    assert np.array_equal(np.array(pos_mm), z["pos_mm"])
    assert shells[0]["bin_num"] == float(z["bin_num"])
    q = np.vstack([s["quant_pc"] for s in shells])
    assert np.allclose(q, z["quant_pc"], atol=1e-4)


@pytest.mark.parametrize("lw", [False, True])
def test_octattn_mullevel_context_vs_reference_driver(orc, lw):
    """encode_dataset_mullevel.py:27-73 + encode_mullevel.py:23-86 restated: chunk sizes, coded symbols, file name and (first two
    chunks, through the oracle model) PMF rows of the reference driver's run (e2e_octattn_mul_*)."""
    import torch
    from cfgs import octattn_cfg
    from oracle import models_ref
    from scp_amd.models import OctAttention
    from scp_amd.weights import fill_weights
    z = golden("e2e_octattn_mul_lw_spher_L12" if lw else "e2e_octattn_mul_spher_L12")
    shells = orc.mullevel_shells(z["xyz"], 12, "spher")
    ids, pos, data, seq = orc.octattn_mullevel_context([s["records"] for s in shells], 1024, lw)
    assert [len(d) - 1023 for d in data] == z["chunk_sizes"].tolist()
    assert np.array_equal(seq[:, -1, 0].astype(np.int16), z["sym_coded"])
    assert orc.octattn_outfile_mullevel("f0", len(data), shells[0]["bin_num"], True, False) == str(z["fname"])
    sd = fill_weights(OctAttention(octattn_cfg()), 0).state_dict()
    rows, row0 = [], 0
    for d, p in list(zip(data, pos))[:2]:
        tab = np.zeros((len(d), 255), np.float32)
        with torch.no_grad():
            for i in range(0, len(d), 1024):
                o = models_ref.octattn_forward(sd, torch.from_numpy(d[i:i + 1024])[None], torch.from_numpy(p[i:i + 1024])[None])
                tab[i:i + 1024] = torch.softmax(o[0], 1).numpy()
        rows.append(tab[1023:])
    pmf = np.vstack(rows)
    st = int(z["pdf_stride"])
    k = len(pmf[::st])
    assert np.abs(pmf[::st] - z["pdf_sub"][:k]).max() < 1e-5


def test_octattn_context(orc):
    z = golden("ctx_octattn_spher_L12")
    r = orc.proc_pc(z["xyz"], 400 / (2 ** 12 - 1), "spher")
    ids, pos, data, oct_seq = orc.octattn_context(r["records"], 1024)
    assert np.array_equal(data, z["data"])
    assert np.array_equal(pos, z["pos"]) and pos.dtype == np.float32
    assert np.array_equal(ids, z["ids"])
    assert np.array_equal(oct_seq, z["oct_seq"])


# ------------------------------------------------------------------ a15 CDF, a16 range coder
def test_cdf_ints(orc):
    z = golden("cdf_mixed")
    assert np.array_equal(orc.pmf_to_cdf(z["pdf"]), z["cdf"])
    assert np.array_equal(orc.pmf_to_cdf_numpy(z["pdf"]), z["cdf"])


def _tile(base, n):
    return np.tile(base, (-(-n // len(base)), 1))[:n]


def test_range_coder_streams(orc):
    z = golden("ac_streams")
    for n in (1, 2, 1000):
        pdf = _tile(z[f"n{n}_pdfbase"], n)
        cdf = orc.pmf_to_cdf(pdf)
        assert np.array_equal(cdf[:300], z[f"n{n}_cdfbase"][:n])
        bs = orc.ac_encode(cdf, z[f"n{n}_sym"])
        assert bs == z[f"n{n}_bytes"].tobytes()
    n = 100000
    pdf = _tile(z[f"n{n}_pdfbase"], n)
    bs, bits = orc.encode_pmf(pdf, z[f"n{n}_sym"])
    assert len(bs) == int(z[f"n{n}_len"]) and bits == 8 * len(bs)
    assert hashlib.sha256(bs).hexdigest() == str(z[f"n{n}_sha"])
    assert bs[:64] == z[f"n{n}_head"].tobytes() and bs[-64:] == z[f"n{n}_tail"].tobytes()
    # decoder round trip (all but the very last symbol are recoverable, like the reference)
    dec = orc.AcDecoder(bs)
    cdf = orc.pmf_to_cdf(pdf[:5000])
    got = [dec.decode_cdf_row(r) for r in cdf]
    assert got == z[f"n{n}_sym"][:5000].tolist()
    # pending-bit stress
    pdf = np.tile(z["pend_pdf_row"], (len(z["pend_sym"]), 1))
    assert orc.encode_pmf(pdf, z["pend_sym"])[0] == z["pend_bytes"].tobytes()


def test_coding_plan(orc):
    w, order = orc.ehem_coding_plan([1, 5, 8193], 8192, mullevel=False)
    assert w == [(0, 0, 1), (1, 0, 5), (2, 0, 8192), (2, 8192, 8193)]
    assert order[:6].tolist() == [0, 1, 3, 5, 2, 4]
    assert order[6] == 6 and order[6 + 4096] == 7 and order[-1] == 6 + 8192
    assert sorted(order.tolist()) == list(range(1 + 5 + 8193))
    # mullevel: a later single-node level is offset by coded_cnt (encode_mullevel.py:120)
    _, om = orc.ehem_coding_plan([1, 4, 1, 3], 8192, mullevel=True)
    assert om.tolist() == [0, 1, 3, 2, 4, 5, 6, 8, 7]
    _, os_ = orc.ehem_coding_plan([1, 4, 1, 3], 8192, mullevel=False)
    assert os_.tolist() == [0, 1, 3, 2, 4, 0, 6, 8, 7]   # encode.py:122 quirk: no coded_cnt
    assert orc.ehem_outfile("a/b", 10, 820.0, 0, True, False) == "a/b_spher_10_820_0.bin"


def test_chamfer_psnr_matches_reference_tools(orc):
    """oracle.chamfer_psnr against the reference's distChamfer and the pc_error binary (tests/golden/metrics.json)."""
    import json
    from scp_amd.synth import synth_frame
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "metrics.json")))
    rng = np.random.default_rng(0)
    a = (rng.random((2000, 3)) * 50).astype(np.float32)
    b = np.round(a[::2].astype(np.float64) / 0.37) * 0.37
    ch, ps = orc.chamfer_psnr(a, b, 59.70)
    assert abs(ch - g["random_2000_vs_1000"]["chamfer"]) < 1e-12 and abs(ps - g["random_2000_vs_1000"]["psnr"]) < 2e-3
    for name in ("spher_L12_s0", "cart_L10_s1", "cylin_L12_s2"):
        e = g[name]
        xyz = synth_frame(e["seed"])[::24].copy() if e["sub"] else synth_frame(e["seed"])
        r = orc.proc_pc(xyz, 400 / (2 ** e["level"] - 1), e["mode"])
        assert r["quant_pc"].shape[0] == e["n_quant"]
        ch, ps = orc.chamfer_psnr(xyz, r["quant_pc"], e["peak"])
        # the de-quantised cloud goes through float32 sin/cos (numpy SIMD, ulp-level differences between builds) and pc_error
        # reads its inputs from a 6-digit ascii ply: both far below these tolerances
        assert abs(ch - e["chamfer"]) < 1e-6 * e["chamfer"] and abs(ps - e["psnr"]) < 2e-3, (name, ch, ps, e)
