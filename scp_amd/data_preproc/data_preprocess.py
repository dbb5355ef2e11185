"""Offline preprocessing with the reference's on-disk contract, octree work on the MI355X.

Drop-in for data_preproc/data_preprocess.py:13-167 (`proc_pc`, `mul_proc_pc`) and the writers of data_preproc/test_gene.py:24-106:
  <out>/<name>.npy            int64 [N,4,6]  (occ 1..256, level, octant, x, y, z) x (great-grandparent, grandparent, parent, self)
  <out>/<name>_0_0|_0_1|_1.npy the three multi-level shells (rows 1:n - the last BFS node is dropped, Octree.py:259-262)
  <out>/<name>_loc.npy        the original float32 cloud
  <out>/<name>_quant.ply      the de-quantised cloud (ascii)
  <out>/<name>_meta.npy       [bin_num, chamfer(, z_offset)]
The integer path (quantiser, Morton sort, octree, K-records) runs through the C ABI; the de-quantised cloud and the chamfer distance
are offline quality tooling (numpy / scipy) and are not part of the encode hot path.
"""
import math
import os

import numpy as np
import torch

from .. import native
from . import pt as pointCloud


def cart2spher(points):
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    rho = np.sqrt(x ** 2 + y ** 2 + z ** 2)
    phi = np.arctan2(y, x + 1e-9)
    phi[np.where(phi < 0)[0]] += 2 * math.pi
    return np.vstack((rho, phi, np.arccos(z / rho))).transpose(1, 0)


def spher2cart(points):
    rho, phi, theta = points[:, 0], points[:, 1], points[:, 2]
    return np.vstack((rho * np.sin(theta) * np.cos(phi), rho * np.sin(theta) * np.sin(phi), rho * np.cos(theta))).transpose(1, 0)


def cylin2cart(points):
    rho, phi, z = points[:, 0], points[:, 1], points[:, 2]
    return np.vstack((rho * np.cos(phi), rho * np.sin(phi), z)).transpose(1, 0)


def cart2cylin(points):
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    rho = np.sqrt(x ** 2 + y ** 2)
    phi = np.arctan2(y, x + 1e-9)
    phi[np.where(phi < 0)[0]] += 2 * math.pi
    return np.vstack((rho, phi, z)).transpose(1, 0)


class HostQuantInfo:
    """What scp_quantize reports in its scp_quant_info, for integers produced on the host."""

    def __init__(self, bin_num, qs, offset):
        self.bin_num = float(bin_num)
        self.qs = [float(v) for v in np.broadcast_to(np.asarray(qs, np.float64).reshape(-1), (3,))]
        self.offset = [float(v) for v in np.broadcast_to(np.asarray(offset, np.float64).reshape(-1), (3,))]


def host_quantize(xyz, mode, qs, cart_offset=-200.0):
    """The reference's own float -> integer step, on the host, in numpy (data_preprocess.py:40-70 / :107-138; cart2spher :200-207,
    cart2cylin :171-177): float32 sqrt / arctan2 / arccos as numpy evaluates them, `bin_num = round(rho.max() / qs) + 1`, the float64
    divide `(points - offset) / qs_vec` and round-half-even.  This is the STRICT-IDENTITY path: the device transform
    (csrc/geom.hip: transform_kernel) evaluates atan2 / acos in float64 and rounds once, which is more accurate than numpy's float32
    routines and therefore not the same integers for a few points per frame.  Returns (int32 [P,3], HostQuantInfo)."""
    return host_quantize_shells(xyz, mode, [qs], cart_offset)[0]


def host_quantize_shells(xyz, mode, qs_list, cart_offset=-200.0):
    """host_quantize for every shell of a multi-level frame: the transform depends on the points only (the reference recomputes the same
    float32 values per shell, data_preprocess.py:107-138), so it runs ONCE and each quantisation step quantises its own copy.
    -> [(int32 [P,3], HostQuantInfo)] in the order of qs_list."""
    pts = np.ascontiguousarray(xyz[:, :3], np.float32)
    tr = cart2cylin(pts) if mode == native.CYLIN else (cart2spher(pts) if mode == native.SPHER else pts)
    rho_max = tr[:, 0].max() if mode != native.CART else None
    z_min = min(tr[:, 2]) if mode == native.CYLIN else None
    return [_quantize_transformed(tr, mode, qs, cart_offset, rho_max, z_min) for qs in qs_list]


def _quantize_transformed(tr, mode, qs, cart_offset, rho_max, z_min):
    if mode == native.CYLIN:
        bin_num = np.round(rho_max / qs) + 1
        qsv = np.array([qs, 2 * math.pi / (bin_num - 1), qs])[True]
        off = np.array([0.0, 0.0, z_min])[True]
    elif mode == native.SPHER:
        bin_num = np.round(rho_max / qs) + 1
        qsv = np.array([qs, 2 * math.pi / (bin_num - 1), math.pi / (bin_num - 1)])[True]
        off = 0
    else:
        bin_num, qsv, off = 0.0, qs, cart_offset
    q = np.round((tr - off) / qsv)
    return q.astype(np.int32), HostQuantInfo(bin_num, qsv, off)


def _mode(cylin, spher):
    return native.CYLIN if cylin else (native.SPHER if spher else native.CART)


def _dequant(leaves, info, cylin, spher):
    qs = np.array(list(info.qs))[None]
    off = np.array(list(info.offset))[None]
    out = leaves.astype(np.float64) * qs + off
    if cylin:
        return cylin2cart(out)
    if spher:
        return spher2cart(out)
    return out


def proc_pc(inp_path, out_dir, out_name, qs=1, offset="min", qlevel=None, rotation=False, normalize=False, test=False, cylin=False,
            spher=False, device=None):
    """data_preprocess.py:13-92.  Returns [out_file, out_points, ref_pt(, bin_num(, offset))] like the reference when test=True."""
    if rotation or normalize or qlevel is not None:
        raise NotImplementedError("only the LiDAR encode configurations are supported (no rotation / normalize / qlevel)")
    os.makedirs(out_dir, exist_ok=True)
    dev = device or torch.device("cuda", torch.cuda.current_device())
    p = pointCloud.ptread(inp_path)
    cart_offset = 0.0 if isinstance(offset, str) else float(offset)
    q, info, _ = native.quantize(torch.from_numpy(np.ascontiguousarray(p, np.float32)).to(dev), _mode(cylin, spher), qs, cart_offset)
    g = native.Geom()
    g.build(q, [(0, q.shape[0], None, False)])
    out_pc = g.krecords(0).cpu().numpy()
    out_file = os.path.join(out_dir, out_name) if test else os.path.join(out_dir, out_name + "_" + str(out_pc.shape[0]))
    if test:
        np.save(out_file + "_loc", p)
    np.save(out_file, out_pc)
    if not test:
        return
    out_points = _dequant(g.leaves(0).cpu().numpy(), info, cylin, spher).astype(np.float32)
    if cylin:
        return [out_file, out_points, p, info.bin_num, np.array(list(info.offset))[None]]
    if spher:
        return [out_file, out_points, p, info.bin_num]
    return [out_file, out_points, p]


def mul_proc_pc(inp_path, out_dir, out_name, qs=1, offset=0, qlevel=None, rotation=False, normalize=False, test=False, cylin=False,
                spher=False, morton_path=(0,), device=None):
    """data_preprocess.py:95-167: one rho shell; the record file drops the last BFS node, the de-quantised cloud keeps every leaf."""
    if rotation or normalize or qlevel is not None:
        raise NotImplementedError("only the LiDAR encode configurations are supported")
    os.makedirs(out_dir, exist_ok=True)
    dev = device or torch.device("cuda", torch.cuda.current_device())
    p = pointCloud.ptread(inp_path)
    q, info, _ = native.quantize(torch.from_numpy(np.ascontiguousarray(p, np.float32)).to(dev), _mode(cylin, spher), qs, float(offset))
    g = native.Geom()
    g.build(q, [(0, q.shape[0], list(morton_path), True)])
    out_pc = g.krecords(0).cpu().numpy()
    if test:
        for m in morton_path:
            out_name += "_" + str(m)
        out_file = os.path.join(out_dir, out_name)
        np.save(out_file + "_loc", p)
    else:
        out_file = os.path.join(out_dir, out_name + "_" + str(out_pc.shape[0]))
    np.save(out_file, out_pc)
    out_points = _dequant(g.leaves(0).cpu().numpy(), info, cylin, spher)
    if cylin:
        return [out_file, out_points, p, info.bin_num, info.offset[2]]
    return [out_file, out_points, p, info.bin_num, 0]


def dist_chamfer(a, b):
    """data_preproc/pt.py:88-95 (KD-tree chamfer) - offline metric."""
    from scipy.spatial import cKDTree
    da, _ = cKDTree(b).query(a)
    db, _ = cKDTree(a).query(b)
    return float(np.max([np.mean(da), np.mean(db)]))


def write_testset(ori_file, out_dir, data_type="kitti", lidar_level=16, spher=False, cylin=False, mullevel=False, chamfer=True):
    """data_preproc/test_gene.py:24-106: write the --preproc_path files for one frame; returns the base name."""
    from pathlib import Path
    ori = Path(ori_file)
    out_name = (str(ori.parent).split("/")[-1] + ori.stem) if data_type == "kitti" else ori.stem
    f = (lambda l: 400 / (2 ** l - 1)) if data_type == "kitti" else (lambda l: 2 ** (18 - l))
    if mullevel:
        parts = [mul_proc_pc(ori_file, out_dir, out_name, qs=f(lidar_level + k), test=True, spher=spher, cylin=cylin, morton_path=path)
                 for k, path in enumerate(([0, 0], [0, 1], [1]))]
        quant = np.vstack([r[1] for r in parts])
        pc, bin_num, z_off = parts[0][2], parts[0][3], parts[0][4]
        meta = [bin_num, dist_chamfer(pc, quant) if chamfer else 0.0, z_off]
    else:
        r = proc_pc(ori_file, out_dir, out_name, qs=f(lidar_level), test=True, spher=spher, cylin=cylin,
                    **({} if (spher or cylin) else {"offset": -200 if data_type == "kitti" else -2 ** 17}))
        quant, pc = r[1], r[2]
        meta = [r[3] if len(r) > 3 else 0, dist_chamfer(pc, quant) if chamfer else 0.0] + ([float(r[4][0, 2])] if cylin else [])
    pointCloud.write_ply_data(os.path.join(out_dir, out_name + "_quant.ply"), quant)
    np.save(os.path.join(out_dir, out_name + "_meta"), meta)
    return out_name
