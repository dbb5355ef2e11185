"""Distortion of the quantiser, on the device.

`chamfer_psnr(pc, quant, peak)` = what the reference obtains from `pointCloud.distChamfer(pc, quantized_pc)`
(data_preproc/pt.py:88-95: max of the two mean nearest-neighbour distances, KD-tree in float64) and from the MPEG `pc_error`
tool it shells out to with `-r peak` (pt.py:13-85; `get_psnr`, utils/__init__.py:3-15 reads "mseF,PSNR (p2point)"):
    mse_ab = mean_i min_j |a_i - b_j|^2,  mseF = max(mse_ab, mse_ba),  PSNR = 10 log10(3 peak^2 / mseF)
peak = 59.70 (KITTI) / 30000 (Ford) (encode_dataset_ehem.py:115-117).  pc_error merges exactly duplicated input points
before measuring (--dropdups=2, its default): `chamfer_psnr(..., dropdups=True)` does the same for the PSNR only.
"""
import math

import torch

from . import native

PEAK = {"kitti": 59.70, "ford": 30000.0, "obj": 1.0}


def _unique_rows(x):
    return torch.unique(x, dim=0)


def chamfer_psnr(pc, quant, peak, dropdups=True):
    """pc [P,3], quant [U,3] device tensors (any float dtype; compared in float64).  Returns dict(chamfer, psnr, mse_ab, mse_ba)."""
    a = pc.to(torch.float64).contiguous()
    b = quant.to(torch.float64).contiguous()
    dab = native.nn_sqdist(a, b)
    dba = native.nn_sqdist(b, a)
    chamfer = max(float(dab.sqrt().mean().item()), float(dba.sqrt().mean().item()))
    if dropdups and (a.shape[0] != _unique_rows(a).shape[0] or b.shape[0] != _unique_rows(b).shape[0]):
        ua, ub = _unique_rows(a), _unique_rows(b)
        m_ab = float(native.nn_sqdist(ua, ub).mean().item())
        m_ba = float(native.nn_sqdist(ub, ua).mean().item())
    else:
        m_ab, m_ba = float(dab.mean().item()), float(dba.mean().item())
    mse = max(m_ab, m_ba)
    psnr = 10.0 * math.log10(3.0 * peak * peak / mse) if mse > 0 else float("inf")
    return dict(chamfer=chamfer, psnr=psnr, mse_ab=m_ab, mse_ba=m_ba)


def dequantize(leaves, qs, offset, spher=False, cylin=False, f32=False):
    """leaves int [U,3] (device) -> de-quantised Cartesian points [U,3]: `pt * qs + offset` in float64, then spher2cart /
    cylin2cart (data_preprocess.py:186-229).  f32=True: the same-level path rounds to float32 before the inverse transform
    (data_preprocess.py:85-91), the multi-level path stays in float64 (:160-167)."""
    q = torch.as_tensor([float(x) for x in qs], dtype=torch.float64, device=leaves.device)
    o = torch.as_tensor([float(x) for x in offset], dtype=torch.float64, device=leaves.device)
    p = leaves.to(torch.float64) * q + o
    if f32:
        p = p.to(torch.float32)
    if cylin:
        rho, phi, z = p[:, 0], p[:, 1], p[:, 2]
        return torch.stack((rho * torch.cos(phi), rho * torch.sin(phi), z), 1)
    if spher:
        rho, phi, th = p[:, 0], p[:, 1], p[:, 2]
        return torch.stack((rho * torch.sin(th) * torch.cos(phi), rho * torch.sin(th) * torch.sin(phi), rho * torch.cos(th)), 1)
    return p
