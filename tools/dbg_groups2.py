import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfgs import ehem_cfg
from scp_amd import native
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.decoder import FrameDecoder
from scp_amd.synth import synth_frame
dev = torch.device("cuda:0")
orig = native.swin_ln_qkv
def spy(x, fw, bias, eps=1e-5, valid=None):
    try:
        return orig(x, fw, bias, eps, valid)
    except Exception as e:
        print("FAILED", e, "x", tuple(x.shape), x.stride(), x.data_ptr() % 16, x.dtype, "N", fw.N, "valid", None if valid is None else (tuple(valid.shape), valid.stride(), valid.dtype, valid.data_ptr() % 16),
              "bias", None if bias is None else (tuple(bias.shape), bias.data_ptr() % 16), "planes", [t.data_ptr() % 16 for t in fw.planes], flush=True)
        raise
native.swin_ln_qkv = spy
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
enc = FrameEncoder(model, "kitti", 16, spher=True, mullevel=True, device=dev)
res = enc.encode(synth_frame(0))
print("encoded", res["n_nodes"], flush=True)
dec = FrameDecoder(model, 16, mullevel=True, polar=True, device=dev)
out = dec.decode(res["bytes"], res["n_levels"], res["pos_mm"])
print("decoded ok")
