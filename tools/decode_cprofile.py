"""Where the decoder's host thread spends a frame: cProfile of FrameDecoder.decode on the L16-m stream.  python tools/decode_cprofile.py"""
import os, sys, cProfile, pstats, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfgs import ehem_cfg
from scp_amd.models import EHEM
from scp_amd.weights import fill_weights
from scp_amd.encoder import FrameEncoder
from scp_amd.decoder import FrameDecoder
from scp_amd.synth import synth_frame
dev = torch.device("cuda:0")
model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
enc = FrameEncoder(model, "kitti", 16, spher=True, mullevel=True, device=dev)
res = enc.encode(synth_frame(0))
dec = FrameDecoder(model, 16, mullevel=True, polar=True, device=dev)
dec.decode(res["bytes"], res["n_levels"], res["pos_mm"])
pr = cProfile.Profile(); pr.enable()
dec.decode(res["bytes"], res["n_levels"], res["pos_mm"])
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
