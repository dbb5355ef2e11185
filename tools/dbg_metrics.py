import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cfgs import ehem_cfg
from scp_amd.encoder import FrameEncoder
from scp_amd.models import EHEM
from scp_amd.synth import synth_frame
from scp_amd import metrics
from oracle import scp_oracle as orc
dev = torch.device('cuda:0')
xyz = synth_frame(0)
enc = FrameEncoder(EHEM(ehem_cfg()).to(dev), "kitti", 12, spher=True, device=dev)
x = torch.from_numpy(xyz).to(dev)
enc.preprocess(x)
info = enc._infos[0]
print('qs', list(info.qs), 'off', list(info.offset), 'bin', info.bin_num)
lv = enc.geom.leaves(0).cpu().numpy()
r = orc.proc_pc(xyz, 400 / (2 ** 12 - 1), "spher")
print('oracle qsv', r['qsv'], 'offset', r['offset'], 'n', r['pts'].shape, lv.shape)
a = {tuple(t) for t in lv.tolist()}; b = {tuple(t) for t in r['pts'].tolist()}
print('leaf sets: common', len(a & b), 'only dev', len(a - b), 'only orc', len(b - a))
dq = metrics.dequantize(enc.geom.leaves(0), info.qs, info.offset, spher=True, f32=True).double().cpu().numpy()
# compare clouds by sorting on integer leaves
oi = np.lexsort(r['pts'].T[::-1]); di = np.lexsort(lv.T[::-1])
if lv.shape == r['pts'].shape:
    print('max |dq - oracle|', np.abs(dq[di] - r['quant_pc'][oi]).max())
ch, ps = orc.chamfer_psnr(xyz, r['quant_pc'], 59.7)
print('oracle', ch, ps, 'oracle on device cloud', orc.chamfer_psnr(xyz, dq, 59.7), 'device', enc.distortion(x))
