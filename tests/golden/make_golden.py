#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden.py [group ...]      # groups: xform oct kseq ctx e2e logits logits_tiefree logits_full swin cdf ac facts facts_ford

Only data (inputs + expected outputs) is written; no reference source travels.  Every
fixture records which reference call produced it.  The script needs
/root/reference and is never run on the GPU box.
"""
import hashlib
import io
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_env  # noqa: E402

_ref_env.setup()

import torch  # noqa: E402

from scp_amd.synth import synth_frame, write_kitti_bin, ford_like  # noqa: E402
from scp_amd.weights import fill_weights  # noqa: E402

from data_preproc import Octree as RO  # noqa: E402
from data_preproc import data_preprocess as RDP  # noqa: E402
from data_preproc.OctreeCPP.Octreewarpper import gen_octree as so_gen_octree  # noqa: E402

torch.set_num_threads(8)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} kB")


def frame5k(seed):
    return synth_frame(seed)[::24].copy()


# ----------------------------------------------------------------------------- xform
def quantise_like_proc_pc(ref_pt, qs, mode):
    """Lines data_preprocess.py:40-68 executed verbatim through the reference functions."""
    import math
    points = ref_pt
    offset = 0
    bin_num = 0
    if mode == "cylin":
        points = RDP.cart2cylin(ref_pt)
        bin_num = np.round(points[:, 0].max() / qs) + 1
        qsv = np.array([qs, 2 * math.pi / (bin_num - 1), qs])[True]
        offset = np.array([0.0, 0.0, min(points[:, 2])])[True]
    elif mode == "spher":
        points = RDP.cart2spher(ref_pt)
        bin_num = np.round(points[:, 0].max() / qs) + 1
        qsv = np.array([qs, 2 * math.pi / (bin_num - 1), math.pi / (bin_num - 1)])[True]
        offset = 0
    else:
        qsv = qs
        offset = -200
    tr = points
    points = points - offset
    pt = np.round(points / qsv)
    return tr, float(bin_num), pt, offset


def gen_xform():
    print("[xform]")
    for seed in (0, 1):
        xyz = synth_frame(seed)[:4096].copy()
        out = {"xyz": xyz}
        for mode in ("spher", "cylin", "cart"):
            for L in (12, 14, 16, 18):
                qs = 400 / (2 ** L - 1)
                tr, bin_num, pt, offset = quantise_like_proc_pc(xyz, qs, mode)
                assert np.abs(pt).max() < 2 ** 31
                out[f"{mode}_L{L}_q"] = pt.astype(np.int32)
                out[f"{mode}_L{L}_bin"] = np.float64(bin_num)
                if mode == "cylin":
                    out[f"{mode}_L{L}_zoff"] = np.float64(offset[0, 2])
            if mode != "cart":
                out[f"{mode}_tr"] = tr.astype(np.float32)
                assert tr.dtype == np.float32
        save(f"xform_s{seed}", **out)


# ----------------------------------------------------------------------------- octree
def so_tree_tables(pts):
    """Run the reference's native builder and flatten it to arrays (BFS order)."""
    tree = so_gen_octree(pts)
    codes = np.array([tree.code[i] for i in range(len(tree.code))], np.int64)
    lv, octant, parent, occ, pos = [], [], [], [], []
    for L in range(len(tree)):
        level = tree[L]
        for i in range(len(level)):
            n = level[i]
            lv.append(L + 1)
            octant.append(n.octant)
            parent.append(n.parent)
            occ.append(n.oct)
            pos.append([n.pos[0], n.pos[1], n.pos[2]])
    return tree, codes, np.array(lv), np.array(octant), np.array(parent), np.array(occ), np.array(pos).reshape(-1, 3)


def py_tree_tables(octree):
    lv, octant, parent, occ, pos = [], [], [], [], []
    for L, level in enumerate(octree):
        for n in level.node:
            lv.append(L + 1)
            octant.append(n.octant)
            parent.append(n.parent)
            occ.append(n.oct)
            pos.append(np.asarray(n.pos).reshape(-1)[:3])
    return np.array(lv), np.array(octant), np.array(parent), np.array(occ), np.array(pos).reshape(-1, 3)


def rand_clouds():
    rng = np.random.default_rng(1234)
    c = {}
    c["cubic50"] = rng.integers(0, 16, (50, 3))
    c["cubic1000"] = rng.integers(0, 1024, (1000, 3))
    c["skew2000"] = np.stack([rng.integers(0, 820, 2000), rng.integers(0, 820, 2000), rng.integers(0, 523, 2000)], 1)
    c["deep300"] = np.stack([rng.integers(0, 13113, 300), rng.integers(0, 13106, 300), rng.integers(0, 8364, 300)], 1)
    c["pow2edge"] = np.array([[0, 0, 0], [15, 15, 15], [16, 0, 0], [0, 16, 0], [0, 0, 16], [31, 31, 31], [7, 8, 9]])
    c["pow2max15"] = np.array([[15, 0, 3], [1, 2, 3], [14, 15, 0]])
    c["single"] = np.array([[5, 9, 2]])
    c["single_one"] = np.array([[1, 0, 0]])  # [[0,0,0]] (depth 0) aborts inside the reference .so
    c["line_x"] = np.stack([np.arange(40), np.zeros(40, int), np.zeros(40, int)], 1)
    base = rng.integers(0, 200, (400, 3))
    c["dups_shuffled"] = rng.permutation(np.concatenate([base, base[:150], base[:10]]))
    c["frame5k_L12"] = None  # filled below from the synthetic frame
    return c


def gen_oct():
    print("[oct]")
    clouds = rand_clouds()
    xyz = frame5k(0)
    _, _, q, _ = quantise_like_proc_pc(xyz, 400 / (2 ** 12 - 1), "spher")
    clouds["frame5k_L12"] = np.unique(q, axis=0).astype(int)
    for name, pts in clouds.items():
        pts = np.asarray(pts).astype(np.int64)
        upts = np.unique(pts, axis=0)
        # the reference .so asserts on duplicate points (Octree.cpp:121); proc_pc always np.unique's first
        tree, codes, lv, octant, parent, occ, pos = so_tree_tables(upts if name.startswith("dups") else pts)
        pcodes, ptree, lmax = RO.GenOctree(upts)
        plv, poctant, pparent, pocc, ppos = py_tree_tables(ptree)
        assert np.array_equal(codes, np.array(pcodes)), name
        assert np.array_equal(lv, plv) and np.array_equal(octant, poctant) and np.array_equal(occ, pocc), name
        assert np.array_equal(pos, ppos), name
        assert np.array_equal(parent[1:], pparent[1:]), name
        rec_so = RO.gen_K_parent_seq(tree, 4)
        rec = np.concatenate((rec_so["Seq"][:, :, True], rec_so["Level"], rec_so["Pos"]), axis=2)
        rec_py = RO.gen_K_parent_seq(ptree, 4)
        recp = np.concatenate((rec_py["Seq"][:, :, True], rec_py["Level"], rec_py["Pos"]), axis=2)
        assert np.array_equal(rec, recp), name
        save(f"oct_{name}", pts=pts.astype(np.int32), codes=codes.astype(np.uint8), level=lv.astype(np.uint8),
             octant=octant.astype(np.uint8), parent=pparent.astype(np.int32), occ=occ.astype(np.uint8),
             pos=pos.astype(np.int32), depth=np.int32(lmax), krec=rec.astype(np.int32))

    # mullevel: shell filter on the rho axis (Octree.py:184-221) + records with the last node dropped
    print("[oct_mul]")
    rng = np.random.default_rng(99)
    mclouds = {
        "skew": np.stack([rng.integers(0, 1000, 1500), rng.integers(0, 900, 1500), rng.integers(0, 500, 1500)], 1),
        "near_only": np.stack([rng.integers(0, 200, 300), rng.integers(0, 1000, 300), rng.integers(0, 500, 300)], 1),
        "frame5k_L14": None,
    }
    _, _, q, _ = quantise_like_proc_pc(xyz, 400 / (2 ** 14 - 1), "spher")
    _, idx = np.unique(q, axis=0, return_index=True)
    mclouds["frame5k_L14"] = q[np.sort(idx)].astype(int)
    for name, pts in mclouds.items():
        pts = np.asarray(pts)
        _, idx = np.unique(pts, axis=0, return_index=True)
        pts = pts[np.sort(idx)]
        out = {"pts": pts.astype(np.int32)}
        for path in ([0, 0], [0, 1], [1]):
            tag = "".join(map(str, path))
            try:
                codes, tree, lmax, idxs = RO.mullevel_gen_octree(pts.astype(np.float64), morton_path=path)
            except Exception as e:  # empty shell: the reference itself fails
                out[f"p{tag}_empty"] = np.int32(1)
                print(f"    {name} path {path}: reference raises {type(e).__name__} (empty shell)")
                continue
            if len(codes) == 0:
                out[f"p{tag}_empty"] = np.int32(1)
                continue
            lv, octant, parent, occ, pos = py_tree_tables(tree)
            rec_d = RO.gen_K_parent_seq_mullevel(tree, 4)
            rec = np.concatenate((rec_d["Seq"][:, :, True], rec_d["Level"], rec_d["Pos"]), axis=2)
            dec = RO.DeOctree(np.array(codes))
            out[f"p{tag}_codes"] = np.array(codes, np.uint8)
            out[f"p{tag}_level"] = lv.astype(np.uint8)
            out[f"p{tag}_octant"] = octant.astype(np.uint8)
            out[f"p{tag}_parent"] = parent.astype(np.int32)
            out[f"p{tag}_occ"] = occ.astype(np.uint8)
            out[f"p{tag}_pos"] = pos.astype(np.int32)
            out[f"p{tag}_krec"] = rec.astype(np.int32)
            out[f"p{tag}_outer"] = rec_d["outer"].astype(np.int32)
            out[f"p{tag}_depth"] = np.int32(lmax)
            out[f"p{tag}_deoct"] = dec.astype(np.int32)
        save(f"octmul_{name}", **out)


# ----------------------------------------------------------------------------- ctx (dataset logic)
def _run_dataset_same(xyz, L, mode, tmp):
    """proc_pc + EncodeEHEMDataset.__getitem__ level split, through the reference classes."""
    from dataloaders.encode_dataset_ehem import EncodeEHEMDataset
    binf = os.path.join(tmp, "seq", "f0.bin")
    os.makedirs(os.path.dirname(binf), exist_ok=True)
    write_kitti_bin(binf, xyz)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        res = RDP.proc_pc(binf, os.path.join(tmp, "pp"), "f0", qs=400 / (2 ** L - 1), test=True,
                          spher=(mode == "spher"), cylin=(mode == "cylin"),
                          **({} if mode != "cart" else {"offset": -200}))
        ds = EncodeEHEMDataset([binf], 8192, "kitti", True, L, mode == "cylin", mode == "spher")
        if mode == "cylin":
            ds.preproc = lambda f: (res[0], res[2], 0.0, res[3], res[4][0, 2], 0.0)
        elif mode == "spher":
            ds.preproc = lambda f: (res[0], res[2], 0.0, res[3], 0.0)
        else:
            ds.preproc = lambda f: (res[0], res[2], 0.0, 0.0)
        item = ds[0]
    finally:
        os.chdir(cwd)
    return res, item


def gen_ctx():
    print("[ctx]")
    xyz = frame5k(0)
    with tempfile.TemporaryDirectory() as tmp:
        for mode, L in (("spher", 12), ("cylin", 12), ("cart", 10)):
            res, item = _run_dataset_same(xyz, L, mode, tmp)
            ids, poss, pos_mm, data, oct_seq = item[0], item[1], item[2], item[3], item[4]
            out = {"xyz": xyz, "n_levels": np.int32(len(data)), "oct_seq": oct_seq.astype(np.int32),
                   "bin_num": np.float64(item[7]), "z_offset": np.float64(item[8]),
                   "quant_pc": np.asarray(res[1], np.float32)}
            for l in range(len(data)):
                out[f"data{l}"] = data[l].astype(np.int16)
                out[f"pos{l}"] = poss[l]
                assert poss[l].dtype == np.float32
                out[f"ids{l}"] = ids[l].astype(np.int32)
            if pos_mm:
                out["pos_mm"] = np.array(pos_mm, np.int64)
            save(f"ctx_ehem_{mode}_L{L}", **out)

        # OctAttention dataset (a9): padded single sequence
        from dataloaders.encode_dataset import EncodeDataset
        binf = os.path.join(tmp, "seq", "f0.bin")
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            res = RDP.proc_pc(binf, os.path.join(tmp, "pp2"), "f0", qs=400 / (2 ** 12 - 1), test=True, spher=True)
            ds = EncodeDataset([binf], 1024, "kitti", False, 12, True)
            ds.preproc = lambda f: (res[0], res[2], 0.0, res[3], 0.0)
            ids, pos, data, oct_seq, npt, bin_num, _, _ = ds[0]
        finally:
            os.chdir(cwd)
        assert len(data) == 1
        save("ctx_octattn_spher_L12", xyz=xyz, data=data[0].astype(np.int16), pos=pos[0], ids=ids[0].astype(np.int32),
             oct_seq=oct_seq.astype(np.int32), bin_num=np.float64(bin_num))

        # mullevel EHEM dataset (three shells)
        from dataloaders.encode_dataset_ehem_mullevel import EncodeEHEMDataset as MulDS
        os.chdir(tmp)
        try:
            L = 14
            outs = []
            for qsl, path in ((L, [0, 0]), (L + 1, [0, 1]), (L + 2, [1])):
                outs.append(RDP.mul_proc_pc(binf, os.path.join(tmp, "ppm"), "f0", qs=400 / (2 ** qsl - 1), test=True,
                                            spher=True, morton_path=path))
            ds = MulDS([binf], 8192, "kitti", True, L, False, True)
            wq = np.vstack([o[1] for o in outs])
            ds.preproc = lambda f: ([o[0] for o in outs], outs[0][2], 0.0, outs[0][3], outs[0][4], 0.0)
            item = ds[0]
        finally:
            os.chdir(cwd)
        ids, poss, pos_mm, data, oct_seq = item[:5]
        out = {"xyz": xyz, "n_levels": np.int32(len(data)), "oct_seq": oct_seq.astype(np.int32),
               "bin_num": np.float64(item[7]), "z_offset": np.float64(item[8]), "quant_pc": wq.astype(np.float32),
               "pos_mm": np.array(pos_mm, np.int64)}
        for l in range(len(data)):
            out[f"data{l}"] = data[l].astype(np.int16)
            out[f"pos{l}"] = poss[l]
            out[f"ids{l}"] = ids[l].astype(np.int32)
        save(f"ctx_ehem_mul_spher_L{L}", **out)


# ----------------------------------------------------------------------------- models
def build_ref_ehem(seed=0):
    from models.ehem import EHEM
    m = EHEM(_ref_env.ehem_cfg()).eval()
    fill_weights(m, seed)
    return m


def build_ref_octattn(seed=0):
    from models.oct_attention import OctAttention
    m = OctAttention(_ref_env.octattn_cfg()).eval()
    fill_weights(m, seed)
    return m


def level_windows_from_ctx(name):
    z = np.load(os.path.join(HERE, name + ".npz"))
    n = int(z["n_levels"])
    return [(z[f"data{l}"].astype(np.int64), z[f"pos{l}"]) for l in range(n)]


def gen_logits():
    print("[logits]")
    m = build_ref_ehem(0)
    levels = level_windows_from_ctx("ctx_ehem_spher_L12")
    sizes = [len(d) for d, _ in levels]
    print("   level sizes", sizes)
    big = max(range(len(levels)), key=lambda i: sizes[i])
    cases = {}
    # c = 1 (root), a tiny odd level, a mid level, and slices of the biggest level
    cases["c1"] = levels[0]
    for want in (7, 600, 1024):
        d, p = levels[big]
        cases[f"c{want}"] = (d[:want], p[:, :want])
    lv_small = min((i for i in range(len(levels)) if 2 < sizes[i] < 200), key=lambda i: sizes[i])
    cases[f"lvl{lv_small}_c{sizes[lv_small]}"] = levels[lv_small]
    with torch.no_grad():
        for tag, (d, p) in cases.items():
            o1, o2 = m(torch.from_numpy(d)[None].clone(), torch.from_numpy(p)[None].clone(), enc=True)
            save(f"logits_ehem_{tag}", data=d.astype(np.int16), pos=p, out1=o1[0].numpy(), out2=o2[0].numpy(),
                 seed=np.int32(0))
        # batched call (B=2) pins the batch semantics of knn / windows
        d, p = levels[big]
        dd = np.stack([d[:256], d[256:512]])
        pp = np.stack([p[:, :256], p[:, 256:512]])
        o1, o2 = m(torch.from_numpy(dd).clone(), torch.from_numpy(pp).clone(), enc=True)
        save("logits_ehem_b2_c256", data=dd.astype(np.int16), pos=pp, out1=o1.numpy(), out2=o2.numpy(), seed=np.int32(0))

    # full 8192 window from the 120k frame (rows subsampled to keep the fixture small)
    xyz = synth_frame(0)
    _, _, q, _ = quantise_like_proc_pc(xyz, 400 / (2 ** 12 - 1), "spher")
    pt = np.unique(q, axis=0).astype(int)
    tree = so_gen_octree(pt)
    # fast K-record construction is NOT available from the reference; use its own (slow) routine once
    rec_d = RO.gen_K_parent_seq(tree, 4)
    rec = np.concatenate((rec_d["Seq"][:, :, True], rec_d["Level"], rec_d["Pos"]), axis=2)
    rec[:, :, 0] -= 1
    lv = rec[:, -1, 1]
    last = lv.max()
    sel = np.where(lv == last)[0]
    blk = rec[sel]
    dat = np.concatenate((blk[:, :, 1:3], blk[:, :, :1]), axis=2)
    cp = blk[:, -1, 3:6]
    pos = ((cp - cp.min()) / (cp.max() - cp.min() + 1e-9)).astype(np.float32).T
    d8, p8 = dat[:8192], np.ascontiguousarray(pos[:, :8192])
    with torch.no_grad():
        o1, o2 = m(torch.from_numpy(d8)[None].clone(), torch.from_numpy(p8)[None].clone(), enc=True)
    save("logits_ehem_c8192", data=d8.astype(np.int16), pos=p8, out1_sub=o1[0, ::16].numpy(), out2_sub=o2[0, ::16].numpy(),
         out1_sha=np.array(sha(o1.numpy())), seed=np.int32(0), stride=np.int32(16))

    mo = build_ref_octattn(0)
    z = np.load(os.path.join(HERE, "ctx_octattn_spher_L12.npz"))
    data, pos = z["data"].astype(np.int64), z["pos"]
    with torch.no_grad():
        for tag, sl in (("c1", slice(1023, 1024)), ("c300", slice(1023, 1323)), ("c1024", slice(1024, 2048)),
                        ("c1024pad", slice(0, 1024))):
            d, p = data[sl], pos[sl]
            o = mo(torch.from_numpy(d)[None].clone(), torch.from_numpy(p)[None].clone())
            save(f"logits_octattn_{tag}", data=d.astype(np.int16), pos=p, out=o[0].numpy(), seed=np.int32(0))


def gen_logits_tiefree():
    """EHEM.forward (models/ehem.py:88-136) on windows whose positions are random float32 triples instead of octree lattice points:
    no exactly tied neighbour distances (dgcnn.py:10-45 `topk` has no tie order to disagree about), so EVERY row of the
    reference's logits is a hard target.  The three neighbour lists the reference's `knn` produced are stored too (sorted per
    row: the edge convolution takes a max over the neighbours, the set is what matters)."""
    print("[logits_tiefree]")
    import models.dgcnn as RD
    m = build_ref_ehem(0)
    rec = []
    orig = RD.knn

    def spy(x, k):
        idx = orig(x, k)
        rec.append(np.sort(idx[0].numpy().astype(np.int16), axis=1))
        return idx

    RD.knn = spy
    try:
        z8 = np.load(os.path.join(HERE, "logits_ehem_c8192.npz"))
        for tag, c, stride in (("c600", 600, 1), ("c2049", 2049, 1), ("c8192", 8192, 8)):
            rng = np.random.default_rng(9000 + c)
            d = z8["data"][:c].astype(np.int64)
            p = rng.random((3, c), dtype=np.float32)
            del rec[:]
            with torch.no_grad():
                o1, o2 = m(torch.from_numpy(d)[None].clone(), torch.from_numpy(p)[None].clone(), enc=True)
            assert len(rec) == 3
            save(f"tiefree_ehem_{tag}", data=d.astype(np.int16), pos=p, out1=o1[0, ::stride].numpy(), out2=o2[0, ::stride].numpy(),
                 stride=np.int32(stride), knn0=rec[0], knn1=rec[1], knn2=rec[2], seed=np.int32(0))
    finally:
        RD.knn = orig


def gen_swin():
    print("[swin]")
    from models.swin_transformer import SwinLayer, SwinConfig, SwinPatchMerging
    cfg = SwinConfig(num_channels=256, embed_dim=256, depths=[2], num_heads=[4], window_size=512)
    for cross in (False, True):
        for shift in (0, 256):
            layer = SwinLayer(cfg, 256, 8192, 4, shift_size=shift, cross=cross).eval()
            fill_weights(layer, 7)
            for L in (2, 3, 511, 512, 513, 1024, 1500):
                rng = np.random.default_rng(1000 + L)
                x = rng.standard_normal((1, L, 256), dtype=np.float32)
                q = rng.standard_normal((1, L, 256), dtype=np.float32)
                with torch.no_grad():
                    y = layer(torch.from_numpy(x), L, query=torch.from_numpy(q) if cross else None)[0]
                stride = 1 if L <= 16 else 7
                save(f"swin_{'cross' if cross else 'self'}_s{shift}_L{L}", x_seed=np.int32(1000 + L),
                     x_head=x[0, :2], y=y[0, ::stride].numpy(), stride=np.int32(stride), wseed=np.int32(7))
    pm = SwinPatchMerging(8192, 256).eval()
    fill_weights(pm, 8)
    for L in (2, 3, 513):
        rng = np.random.default_rng(2000 + L)
        x = rng.standard_normal((1, L, 256), dtype=np.float32)
        with torch.no_grad():
            y = pm(torch.from_numpy(x), L)
        save(f"swin_merge_L{L}", x_seed=np.int32(2000 + L), y=y[0].numpy(), wseed=np.int32(8))


# ----------------------------------------------------------------------------- cdf / ac
def make_pmfs(rng, n):
    rows = []
    for i in range(n):
        kind = i % 5
        if kind == 0:
            lg = rng.standard_normal(255) * 4
        elif kind == 1:
            lg = np.zeros(255)
        elif kind == 2:
            lg = rng.standard_normal(255) * 0.1
            lg[rng.integers(0, 255)] += 30
        elif kind == 3:
            lg = rng.standard_normal(255) * 12
        else:
            lg = rng.standard_normal(255)
            lg[rng.integers(0, 255, 100)] = -200.0  # exact zeros after softmax
        p = torch.softmax(torch.from_numpy(lg.astype(np.float32)), 0).numpy()
        rows.append(p)
    return np.stack(rows).astype(np.float32)


def ref_cdf_int(pdf):
    import numpyAc.numpyAc as NA
    cdfF = NA.pdf_convert_to_cdf_and_normalize(pdf)
    return NA._convert_to_int_and_normalize(cdfF, True)


def gen_cdf():
    print("[cdf]")
    rng = np.random.default_rng(5)
    pdf = make_pmfs(rng, 64)
    cdf = ref_cdf_int(pdf)
    assert cdf.dtype == np.int16 and cdf.shape == (64, 256)
    save("cdf_mixed", pdf=pdf, cdf=cdf.view(np.uint16))


def gen_ac():
    print("[ac]")
    import numpyAc.numpyAc as NA
    rng = np.random.default_rng(6)
    out = {}
    for n in (1, 2, 1000, 100000):
        base = make_pmfs(rng, min(n, 300))
        reps = -(-n // len(base))
        pdf = np.tile(base, (reps, 1))[:n]   # consumers rebuild pdf by tiling `pdfbase` the same way
        # symbols: mostly sampled from the PMF, with forced 254s and a long run on a ~1/2-probability symbol
        cdfF = np.cumsum(pdf.astype(np.float64), 1)
        u = rng.random(n)
        sym = np.minimum((cdfF < u[:, None] * cdfF[:, -1:]).sum(1), 254).astype(np.int16)
        sym[:: max(1, n // 7)] = 254
        sym[0] = 254 if n > 1 else 17
        codec = NA.arithmeticCoding()
        bs, bits = codec.encode(pdf, sym, None)
        cdf = ref_cdf_int(pdf).view(np.uint16)
        dec = NA.arithmeticDeCoding(bs, n, 255, None)
        back = [dec.decode(pdf[i:i + 1]) for i in range(min(n, 3000))] if n > 1 else []
        assert back == sym[: len(back)].tolist()
        out[f"n{n}_pdfbase"] = base
        if n <= 1000:
            out[f"n{n}_cdfbase"] = cdf[:300]
            out[f"n{n}_bytes"] = np.frombuffer(bs, np.uint8)
        else:
            out[f"n{n}_sha"] = np.array(hashlib.sha256(bs).hexdigest())
            out[f"n{n}_head"] = np.frombuffer(bs[:64], np.uint8)
            out[f"n{n}_tail"] = np.frombuffer(bs[-64:], np.uint8)
            out[f"n{n}_len"] = np.int64(len(bs))
        out[f"n{n}_sym"] = sym
    # pending-bit stress: two symbols with probability ~1/2 each straddling the midpoint
    n = 4000
    pdf = np.full((n, 255), 1e-9, np.float32)
    pdf[:, 100] = 0.5
    pdf[:, 101] = 0.5
    sym = np.where(np.arange(n) % 2 == 0, 100, 101).astype(np.int16)
    bs, _ = NA.arithmeticCoding().encode(pdf, sym, None)
    out["pend_pdf_row"] = pdf[0]
    out["pend_sym"] = sym
    out["pend_bytes"] = np.frombuffer(bs, np.uint8)
    save("ac_streams", **out)


# ----------------------------------------------------------------------------- end-to-end (reference driver)
def _e2e_setup():
    """The reference's drivers importable and runnable on CPU: hydra stubbed, Tensor.cuda patched to identity, the range coder's
    `encode` wrapped so the PMF table and symbol vector it receives are captured."""
    import types
    hy = types.ModuleType("hydra")
    hy.initialize = lambda **k: None
    hy.compose = lambda **k: None
    sys.modules.setdefault("hydra", hy)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    import numpyAc.numpyAc as NA
    sys.modules["numpyAc"].arithmeticCoding = NA.arithmeticCoding
    sys.path.insert(0, _ref_env.REF)
    import importlib
    enc = importlib.import_module("encode")
    encm = importlib.import_module("encode_mullevel")
    captured = {}
    orig_encode = NA.arithmeticCoding.encode

    def spy(self, pdf, sym, binfile=None):
        captured["pdf"] = pdf.copy()
        captured["sym"] = sym.copy()
        return orig_encode(self, pdf, sym, binfile)

    NA.arithmeticCoding.encode = spy

    def restore():
        NA.arithmeticCoding.encode = orig_encode
    return enc, encm, captured, restore


def gen_e2e_octattn_mul():
    """encode_mullevel.py:23-86 `compress` (OctAttention over the three rho shells) fed by dataloaders/encode_dataset_mullevel.py
    `get_data` (:44-73), without and with --level_wise.  The dataset's own `preproc` cannot run (it unpacks the 3-value
    `_meta.npy` of the multi-level test-set generator into 2 names, :80), so the batch is put together from `get_data` the way
    `__getitem__` (:27-42) + the DataLoader's collate do."""
    print("[e2e_octattn_mul]")
    enc, encm, captured, restore = _e2e_setup()
    from dataloaders.encode_dataset_mullevel import EncodeDataset as MulDS
    xyz = frame5k(0)
    mo = build_ref_octattn(0)
    L = 12
    with tempfile.TemporaryDirectory() as tmp:
        binf = os.path.join(tmp, "seq", "f0.bin")
        os.makedirs(os.path.dirname(binf))
        write_kitti_bin(binf, xyz)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            outs3 = [RDP.mul_proc_pc(binf, os.path.join(tmp, "ppm"), "f0", qs=400 / (2 ** (L + k) - 1), test=True, spher=True, morton_path=path)
                     for k, path in enumerate(([0, 0], [0, 1], [1]))]
            bin_num = int(outs3[0][3])
            for lw in (False, True):
                ds = MulDS([binf], 1024, "kitti", lw, L, True, os.path.join(tmp, "ppm") + "/")
                ids, pos, data, oct_seq = ds.get_data(outs3[0][0])
                for o in outs3[1:]:
                    i2, p2, d2, s2 = ds.get_data(o[0])
                    ids += i2; pos += p2; data += d2
                    oct_seq = np.vstack((oct_seq, s2))
                batch = ([torch.from_numpy(a)[None] for a in ids], [torch.from_numpy(a)[None] for a in pos],
                         [torch.from_numpy(a)[None] for a in data], torch.from_numpy(oct_seq)[None], torch.tensor([len(xyz)]),
                         torch.tensor([bin_num]))

                class A:
                    spher = True
                    cylin = False
                    sequential = False

                tag = "lw_" if lw else ""
                odir = os.path.join(tmp, "out" + tag)
                bpp, _ = encm.compress(batch, os.path.join(odir, "f0"), mo, A)
                outs = [f for f in os.listdir(odir) if f.endswith(".bin")]
                assert len(outs) == 1
                bs = open(os.path.join(odir, outs[0]), "rb").read()
                pdf = captured["pdf"]
                save(f"e2e_octattn_mul_{tag}spher_L{L}", xyz=xyz, fname=np.array(outs[0]), bytes=np.frombuffer(bs, np.uint8),
                     sym_coded=captured["sym"], pdf_sub=pdf[::37], pdf_stride=np.int32(37), bpp=np.float64(bpp), n_nodes=np.int64(len(pdf)),
                     chunk_sizes=np.array([d.shape[0] - 1023 for d in data], np.int64), cdf_sha=np.array(sha(ref_cdf_int(pdf))), wseed=np.int32(0))
                print("   octattn mullevel", "level-wise" if lw else "per shell", "bpp", bpp, "bytes", len(bs), outs[0])
        finally:
            os.chdir(cwd)
            restore()


def gen_e2e():
    """Run the reference's own compress_ehem / compress drivers on CPU (Tensor.cuda patched to identity)."""
    print("[e2e]")
    enc, encm, captured, restore = _e2e_setup()
    from torch.utils.data import DataLoader

    class A:
        spher = True
        cylin = False
        sequential = False

    xyz = frame5k(0)
    model = build_ref_ehem(0)
    with tempfile.TemporaryDirectory() as tmp:
        binf = os.path.join(tmp, "seq", "f0.bin")
        os.makedirs(os.path.dirname(binf))
        write_kitti_bin(binf, xyz)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            # ---- same-level EHEM, L12 --spher
            from dataloaders.encode_dataset_ehem import EncodeEHEMDataset
            res = RDP.proc_pc(binf, os.path.join(tmp, "pp"), "f0", qs=400 / (2 ** 12 - 1), test=True, spher=True)
            ds = EncodeEHEMDataset([binf], 8192, "kitti", True, 12, False, True)
            ds.preproc = lambda f: (res[0], res[2], 0.25, res[3], 33.0)
            batch = next(iter(DataLoader(ds, batch_size=1, shuffle=False)))
            bpp, _ = enc.compress_ehem(batch[:-2], os.path.join(tmp, "out", "seqf0"), model, A)
            outs = [f for f in os.listdir(os.path.join(tmp, "out")) if f.endswith(".bin")]
            assert len(outs) == 1
            bs = open(os.path.join(tmp, "out", outs[0]), "rb").read()
            dat = torch.load(os.path.join(tmp, "out", outs[0] + ".dat")).numpy()
            pdf = captured["pdf"]
            save("e2e_ehem_spher_L12", xyz=xyz, fname=np.array(outs[0]), bytes=np.frombuffer(bs, np.uint8),
                 dat=dat, sym_coded=captured["sym"], pdf_sub=pdf[::37], pdf_stride=np.int32(37),
                 bpp=np.float64(bpp), n_nodes=np.int64(len(pdf)),
                 cdf_sha=np.array(sha(ref_cdf_int(pdf))), pdf_sha=np.array(sha(pdf)), wseed=np.int32(0))
            print("   same-level bpp", bpp, "bytes", len(bs), outs[0])

            # ---- mullevel EHEM, L14 --spher
            from dataloaders.encode_dataset_ehem_mullevel import EncodeEHEMDataset as MulDS
            L = 14
            outs3 = []
            for qsl, path in ((L, [0, 0]), (L + 1, [0, 1]), (L + 2, [1])):
                outs3.append(RDP.mul_proc_pc(binf, os.path.join(tmp, "ppm"), "f0", qs=400 / (2 ** qsl - 1), test=True,
                                             spher=True, morton_path=path))
            ds = MulDS([binf], 8192, "kitti", True, L, False, True)
            ds.preproc = lambda f: ([o[0] for o in outs3], outs3[0][2], 0.25, outs3[0][3], outs3[0][4], 33.0)
            batch = next(iter(DataLoader(ds, batch_size=1, shuffle=False)))
            bpp, _ = encm.compress_ehem(batch[:-2], os.path.join(tmp, "outm", "seqf0"), model, A)
            outs = [f for f in os.listdir(os.path.join(tmp, "outm")) if f.endswith(".bin")]
            bs = open(os.path.join(tmp, "outm", outs[0]), "rb").read()
            dat = torch.load(os.path.join(tmp, "outm", outs[0] + ".dat")).numpy()
            pdf = captured["pdf"]
            save("e2e_ehem_mul_spher_L14", xyz=xyz, fname=np.array(outs[0]), bytes=np.frombuffer(bs, np.uint8),
                 dat=dat, sym_coded=captured["sym"], pdf_sub=pdf[::37], pdf_stride=np.int32(37),
                 bpp=np.float64(bpp), n_nodes=np.int64(len(pdf)),
                 cdf_sha=np.array(sha(ref_cdf_int(pdf))), pdf_sha=np.array(sha(pdf)), wseed=np.int32(0))
            print("   mullevel bpp", bpp, "bytes", len(bs), outs[0])

            # ---- OctAttention, L12 --spher (reference `compress`)
            from dataloaders.encode_dataset import EncodeDataset
            mo = build_ref_octattn(0)
            res = RDP.proc_pc(binf, os.path.join(tmp, "pp3"), "f0", qs=400 / (2 ** 12 - 1), test=True, spher=True)
            ds = EncodeDataset([binf], 1024, "kitti", False, 12, True)
            ds.preproc = lambda f: (res[0], res[2], 0.25, res[3], 33.0)
            batch = next(iter(DataLoader(ds, batch_size=1, shuffle=False)))
            bpp, _ = enc.compress(batch[:-2], os.path.join(tmp, "outo", "f0"), mo, A)
            bs = open(os.path.join(tmp, "outo", "f0.bin"), "rb").read()
            pdf = captured["pdf"]
            save("e2e_octattn_spher_L12", xyz=xyz, bytes=np.frombuffer(bs, np.uint8), sym_coded=captured["sym"],
                 pdf_sub=pdf[::37], pdf_stride=np.int32(37), bpp=np.float64(bpp), n_nodes=np.int64(len(pdf)),
                 cdf_sha=np.array(sha(ref_cdf_int(pdf))), wseed=np.int32(0))
            print("   octattn bpp", bpp, "bytes", len(bs))
        finally:
            os.chdir(cwd)
            restore()


# ----------------------------------------------------------------------------- full-frame facts
def gen_facts():
    """Full 120k-point frame structure facts + checksums, from the reference."""
    print("[facts]")
    facts = {}
    xyz = synth_frame(0)

    def same(mode, L, key, cloud=xyz, qs=None):
        qs = 400 / (2 ** L - 1) if qs is None else qs
        _, bin_num, q, off = quantise_like_proc_pc(cloud, qs, mode)
        pt = np.unique(q, axis=0).astype(int)
        tree = so_gen_octree(pt)
        codes = np.array([tree.code[i] for i in range(len(tree.code))], np.uint8)
        per = [len(tree[i]) for i in range(len(tree))]
        facts[key] = dict(bin_num=bin_num, U=int(len(pt)), D=len(tree), N=int(len(codes)), per_level=per,
                          max_ints=[int(v) for v in pt.max(0)], codes_sha=sha(codes), pts_sha=sha(pt.astype(np.int32)))
        print("   ", key, facts[key]["N"], per)
        return tree

    tree = same("spher", 12, "L12-s")
    rec_d = RO.gen_K_parent_seq(tree, 4)
    rec = np.concatenate((rec_d["Seq"][:, :, True], rec_d["Level"], rec_d["Pos"]), axis=2)
    facts["L12-s"]["krec_sha_i32"] = sha(rec.astype(np.int32))
    same("spher", 16, "L16-s")
    same("cylin", 14, "C14")
    same("cart", 12, "L12-c")
    with open(os.path.join(HERE, "frame_facts.json"), "w") as f:
        json.dump(facts, f, indent=1)

    # mullevel: pure-Python reference octree is slow (~50 s) - run once
    for key, cloud, qss in (("L16-m", xyz, [400 / (2 ** l - 1) for l in (16, 17, 18)]),):
        shells = []
        for qs, path in zip(qss, ([0, 0], [0, 1], [1])):
            _, bin_num, q, off = quantise_like_proc_pc(cloud, qs, "spher")
            _, idx = np.unique(q, axis=0, return_index=True)
            pt = q[np.sort(idx)]
            codes, tree, lmax, idxs = RO.mullevel_gen_octree(pt, morton_path=path)
            rec_d = RO.gen_K_parent_seq_mullevel(tree, 4)
            rec = np.concatenate((rec_d["Seq"][:, :, True], rec_d["Level"], rec_d["Pos"]), axis=2)
            shells.append(dict(bin_num=float(bin_num), leaves=int(len(idxs)), D=int(lmax), records=int(len(rec)),
                               codes_sha=sha(np.array(codes, np.uint8)), krec_sha_i32=sha(rec.astype(np.int32)),
                               per_level=[len(l.node) for l in tree]))
            print("   ", key, path, shells[-1]["records"])
        facts[key] = shells
        with open(os.path.join(HERE, "frame_facts.json"), "w") as f:
            json.dump(facts, f, indent=1)

def gen_frame_ints():
    """The reference quantiser's integers (data_preprocess.py:40-68 through numpy's float32 SIMD arctan2 / arccos of THIS
    container) for the 120k-point frame at every full-size configuration of frame_facts.json, in point order.  numpy's float32
    trigonometry is CPU dependent, so the GPU box cannot re-derive them: with these as input the full-size stream / record
    checksums are asserted unconditionally."""
    print("[frame_ints]")
    xyz = synth_frame(0)
    facts = json.load(open(os.path.join(HERE, "frame_facts.json")))
    out = {}
    for mode, L, key in (("spher", 12, "L12-s"), ("spher", 16, "L16-s"), ("spher", 17, None), ("spher", 18, None), ("cylin", 14, "C14"),
                         ("cart", 12, "L12-c")):
        _, bin_num, q, off = quantise_like_proc_pc(xyz, 400 / (2 ** L - 1), mode)
        q = np.asarray(q)
        assert np.array_equal(q, q.astype(np.int32))
        if key is not None:
            assert sha(np.unique(q, axis=0).astype(np.int32)) == facts[key]["pts_sha"] and bin_num == facts[key]["bin_num"]
        out[f"q_{mode}_L{L}"] = np.ascontiguousarray(q.astype(np.int32))
    save("frame_ints", **out)


def gen_keys():
    """state_dict key/shape/dtype inventory of the reference modules (drop-in checkpoint contract, Appendix D)."""
    print("[keys]")
    out = {}
    for name, m in (("EHEM", build_ref_ehem(0)), ("OctAttention", build_ref_octattn(0))):
        out[name] = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
        print("   ", name, len(out[name]), "entries")
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump(out, f)


# ----------------------------------------------------------------------------- metrics
def gen_metrics():
    """Chamfer distance through the reference's `distChamfer` (data_preproc/pt.py:88-95) and D1 PSNR through the reference's
    `pcerror` + `get_psnr` (pt.py:13-85, utils/__init__.py:3-15), which shell out to the MPEG `pc_error` binary shipped under
    /root/reference/utils (no exec bit on this mount: a temporary copy is made executable).  Inputs are re-derivable from seeds
    (synthetic frames + the reference quantiser), so the fixture holds the de-quantised clouds' hashes and the numbers only."""
    import shutil
    import stat
    from data_preproc import pt as RPT
    from utils import get_psnr
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "pc_error")
        shutil.copy("/root/reference/utils/pc_error", exe)
        os.chmod(exe, os.stat(exe).st_mode | stat.S_IXUSR)
        os.makedirs(os.path.join(tmp, "temp", "data"), exist_ok=True)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            def measure(pc, q, peak):
                res = os.path.join(tmp, "temp", "r.txt")
                RPT.pcerror(pc.copy(), q.copy(), None, "-r " + peak, res, pcerror_path=exe)
                psnr = get_psnr(res)[0]
                return float(RPT.distChamfer(pc.copy(), q.copy())), float(psnr)
            rng = np.random.default_rng(0)
            a = (rng.random((2000, 3)) * 50).astype(np.float32)
            b = np.round(a[::2].astype(np.float64) / 0.37) * 0.37
            ch, ps = measure(a, b, "59.70")
            out["random_2000_vs_1000"] = dict(chamfer=ch, psnr=ps, peak=59.70)
            for name, seed, L, mode in (("spher_L12_s0", 0, 12, "spher"), ("cart_L10_s1", 1, 10, "cart"), ("cylin_L12_s2", 2, 12, "cylin")):
                xyz = frame5k(seed) if mode == "cart" else synth_frame(seed)
                binf = os.path.join(tmp, "seq", name + ".bin")
                os.makedirs(os.path.dirname(binf), exist_ok=True)
                write_kitti_bin(binf, xyz)
                res = RDP.proc_pc(binf, os.path.join(tmp, "pp"), name, qs=400 / (2 ** L - 1), test=True, spher=(mode == "spher"),
                                  cylin=(mode == "cylin"), **({} if mode != "cart" else {"offset": -200}))
                q, pc = res[1], res[2]
                ch, ps = measure(pc, q, "59.70")
                out[name] = dict(chamfer=ch, psnr=ps, peak=59.70, seed=seed, level=L, mode=mode, n_pc=int(pc.shape[0]), n_quant=int(q.shape[0]),
                                 sub=(mode == "cart"))
                print("  ", name, out[name])
        finally:
            os.chdir(cwd)
    with open(os.path.join(HERE, "metrics.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("  wrote metrics.json")


# ----------------------------------------------------------------------------- training dataset
def gen_trainset():
    """dataloaders/ehem_dataset.py EHEMDataset on two synthetic record files, torch seed 7: the first 9 items."""
    from dataloaders.ehem_dataset import EHEMDataset
    from types import SimpleNamespace
    rng = np.random.default_rng(3)
    with tempfile.TemporaryDirectory() as tmp:
        recs = {}
        for name, n in (("a", 100), ("b", 57)):
            r = rng.integers(1, 256, size=(n, 4, 6)).astype(np.int64)
            r[:, :, 1] = rng.integers(1, 13, size=(n, 4))
            r[:, :, 2] = rng.integers(1, 9, size=(n, 4))
            r[:, :, 3:] = rng.integers(0, 4096, size=(n, 4, 3))
            np.save(os.path.join(tmp, f"{name}_{n}.npy"), r)
            recs[name] = r
        ds = EHEMDataset(SimpleNamespace(root=os.path.join(tmp, "*.npy"), context_size=16, extra_pos=False))
        torch.manual_seed(7)
        items = [ds[i] for i in (0, 1, 2, 3, 4, 5, 6, 7, 8)]
        save("trainset", rec_a=recs["a"], rec_b=recs["b"], length=np.int64(len(ds)),
             data=np.stack([it[0] for it in items]), pos=np.stack([it[1] for it in items]), label=np.stack([it[2] for it in items]))


def gen_logits_full():
    """Value checks of the model output at the sizes of BASELINE.json configs[3] / configs[4]: one FULL 8192-node window of the
    Ford-like level-17 multi-level frame through the reference EHEM, and one full 1024-row window of the level-14 --cylin frame through the
    reference OctAttention - rows sub-sampled to keep the fixtures small.  The windows' inputs come from the repository's CPU oracle (its tables are
    pinned against the reference's datasets by the ctx_* fixtures); the expected outputs are the REFERENCE models' own."""
    print("[logits_full]")
    from oracle import scp_oracle as orc
    # --- EHEM, Ford-like L17 multi-level: the last full window of the deepest level of shell 0
    xyz = ford_like(synth_frame(0))
    shells = orc.mullevel_shells(xyz, 17, "spher", data_type="ford")
    ids, poss, pos_mm, data, _ = orc.ehem_mullevel_context([s["records"] for s in shells], 17)
    sizes = [len(d) for d in data]
    windows, _ = orc.ehem_coding_plan(sizes, 8192, True)
    full = [(l, i, j) for (l, i, j) in windows if j - i == 8192]
    l, i, j = full[len(full) // 2]
    d8, p8 = data[l][i:j].astype(np.int64), np.ascontiguousarray(poss[l][:, i:j])
    m = build_ref_ehem(0)
    with torch.no_grad():
        o1, o2 = m(torch.from_numpy(d8)[None].clone(), torch.from_numpy(p8)[None].clone(), enc=True)
    save("logits_ehem_f17m_c8192", data=d8.astype(np.int16), pos=p8, out1_sub=o1[0, ::16].numpy(), out2_sub=o2[0, ::16].numpy(),
         out1_sha=np.array(sha(o1.numpy())), seed=np.int32(0), stride=np.int32(16), level_index=np.int32(l), window_start=np.int32(i),
         level_sizes=np.asarray(sizes, np.int64))
    # --- OctAttention, L14 --cylin: a full window from the middle of the padded sequence
    rec = orc.proc_pc(synth_frame(0), orc.kitti_qs(14), "cylin")["records"]
    ids, pos, data, _ = orc.octattn_context(rec, 1024)
    w0 = (len(data) // 2048) * 1024
    d, p = data[w0:w0 + 1024].astype(np.int64), pos[w0:w0 + 1024]
    mo = build_ref_octattn(0)
    with torch.no_grad():
        o = mo(torch.from_numpy(d)[None].clone(), torch.from_numpy(p)[None].clone())
    save("logits_octattn_L14cylin_c1024", data=d.astype(np.int16), pos=p, out=o[0].numpy(), seed=np.int32(0), window_start=np.int32(w0),
         n_rows=np.int32(len(data)))


def gen_facts_ford():
    """BASELINE configs[3] (Ford-like frame = integer millimetres, --spher --mullevel, lidar_level 17): the reference's OWN `mul_proc_pc`
    (data_preprocess.py:95-167) run with the three (qs, morton_path) pairs of encode_dataset_ehem_mullevel.py:154-186 for data_type 'ford'
    (qs = 2^(18-L), 2^(17-L), 2^(16-L) = 2, 1, 0.5 mm).  Written: frame_facts.json["F17-m"] (per shell: bin_num, leaves, depth, records,
    per-level node counts, sha256 of the occupancy codes and of the [N,4,6] records the reference saved) and
    frame_ints.npz: q_spher_ford_L17/18/19 = the reference quantiser's integers in point order (whose octree is checked here to BE the
    one mul_proc_pc built: same code stream).  ~4 min (pure-Python octree of 760 k nodes)."""
    print("[facts_ford]")
    L = 17
    xyz = ford_like(synth_frame(0))
    path_json = os.path.join(HERE, "frame_facts.json")
    facts = json.load(open(path_json))
    ints = dict(np.load(os.path.join(HERE, "frame_ints.npz")))
    shells = []
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "ford0.bin")            # ptread -> loadbin: float32 [P,4] rows, the values are whole millimetres
        write_kitti_bin(src, xyz)
        for k, mpath in enumerate(([0, 0], [0, 1], [1])):
            qs = 2.0 ** (18 - L - k)
            out_file, quant_pc, ref_pt, bin_num, _ = RDP.mul_proc_pc(src, tmp, "ford0", normalize=False, qs=qs, test=True, spher=True,
                                                                     morton_path=mpath)
            assert np.array_equal(ref_pt, xyz)
            rec = np.load(out_file + ".npy")
            # the integers in point order, by the same lines (:107-137) through the reference's cart2spher
            _, bin2, q, _ = quantise_like_proc_pc(xyz, qs, "spher")
            q = np.asarray(q)
            assert bin2 == bin_num and np.array_equal(q, q.astype(np.int32))
            _, idx = np.unique(q, axis=0, return_index=True)
            codes, tree, lmax, idxs = RO.mullevel_gen_octree(q[np.sort(idx)], morton_path=mpath)
            rec_d = RO.gen_K_parent_seq_mullevel(tree, 4)
            rec2 = np.concatenate((rec_d["Seq"][:, :, True], rec_d["Level"], rec_d["Pos"]), axis=2)
            assert np.array_equal(rec, rec2), "quantise_like_proc_pc does not reproduce mul_proc_pc's records"
            shells.append(dict(qs=qs, bin_num=float(bin_num), leaves=int(len(idxs)), D=int(lmax), records=int(len(rec)),
                               codes_sha=sha(np.array(codes, np.uint8)), krec_sha_i32=sha(rec.astype(np.int32)),
                               per_level=[len(l.node) for l in tree], quant_pc_sha_f64=sha(np.asarray(quant_pc, np.float64))))
            ints[f"q_spher_ford_L{L + k}"] = np.ascontiguousarray(q.astype(np.int32))
            print("   F17-m", mpath, shells[-1]["records"], shells[-1]["per_level"])
    facts["F17-m"] = shells
    with open(path_json, "w") as f:
        json.dump(facts, f, indent=1)
    save("frame_ints", **ints)


GROUPS = {"trainset": gen_trainset, "metrics": gen_metrics, "keys": gen_keys, "xform": gen_xform, "oct": gen_oct, "ctx": gen_ctx, "logits": gen_logits, "logits_tiefree": gen_logits_tiefree, "swin": gen_swin,
          "cdf": gen_cdf, "ac": gen_ac, "e2e": gen_e2e, "e2e_octattn_mul": gen_e2e_octattn_mul, "facts": gen_facts, "frame_ints": gen_frame_ints,
          "logits_full": gen_logits_full, "facts_ford": gen_facts_ford}

if __name__ == "__main__":
    want = sys.argv[1:] or list(GROUPS)
    for g in want:
        GROUPS[g]()
