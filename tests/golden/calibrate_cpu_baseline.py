#!/usr/bin/env python3
"""CPU-baseline calibration: time the oracle port (oracle/cpu_encode.py - what bench.py's
`cpu_baseline` runs on the GPU box, where /root/reference does not exist) against the TRUE reference, in this container, on
the same frame, same weights, same thread count.  The reference side is its own code end to end: `proc_pc` (numpy transform +
the shipped Octree .so + gen_K_parent_seq), `EncodeEHEMDataset.__getitem__`, `compress_ehem` (EHEM on PyTorch-CPU, per-window
softmax, numpyAc).  Writes profiles/cpu_calibration_r2.json; the ratio goes into BASELINE.md.

    python tests/golden/calibrate_cpu_baseline.py [level]        # default: level 12 --spher same-level, synth frame seed 0
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import make_golden as G  # noqa: E402  (sets up the reference import shims)
import torch  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    threads = os.cpu_count()
    torch.set_num_threads(threads)
    xyz = G.synth_frame(0)
    enc, encm, captured, restore = G._e2e_setup()
    from torch.utils.data import DataLoader
    from dataloaders.encode_dataset_ehem import EncodeEHEMDataset
    model = G.build_ref_ehem(0)
    out = {"level": L, "mode": "spher same-level", "threads": threads, "frame": "synth_frame(0), 120000 points"}

    class A:
        spher = True
        cylin = False
        sequential = False

    with tempfile.TemporaryDirectory() as tmp:
        binf = os.path.join(tmp, "seq", "f0.bin")
        os.makedirs(os.path.dirname(binf))
        G.write_kitti_bin(binf, xyz)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            t0 = time.perf_counter()
            res = G.RDP.proc_pc(binf, os.path.join(tmp, "pp"), "f0", qs=400 / (2 ** L - 1), test=True, spher=True)
            t_pre = time.perf_counter() - t0
            ds = EncodeEHEMDataset([binf], 8192, "kitti", True, L, False, True)
            ds.preproc = lambda f: (res[0], res[2], 0.25, res[3], 33.0)
            t0 = time.perf_counter()
            batch = next(iter(DataLoader(ds, batch_size=1, shuffle=False)))
            t_ctx = time.perf_counter() - t0
            t0 = time.perf_counter()
            bpp, t_model_ref = enc.compress_ehem(batch[:-2], os.path.join(tmp, "out", "seqf0"), model, A)
            t_comp = time.perf_counter() - t0
            ref_bits = 8 * os.path.getsize(os.path.join(tmp, "out", [f for f in os.listdir(os.path.join(tmp, "out")) if f.endswith(".bin")][0]))
        finally:
            os.chdir(cwd)
            restore()
    out["reference"] = dict(proc_pc_s=t_pre, dataset_s=t_ctx, compress_ehem_s=t_comp, model_s_as_printed=float(t_model_ref),
                            total_s=t_pre + t_ctx + t_comp, bits=ref_bits, n_nodes=int(len(captured["pdf"])))
    from cfgs import ehem_cfg
    from oracle import cpu_encode
    from scp_amd.models import EHEM
    from scp_amd.weights import fill_weights
    sd = fill_weights(EHEM(ehem_cfg()), 0).state_dict()
    torch.set_num_threads(threads)
    r = cpu_encode.encode_frame(xyz, sd, L, mullevel=False, mode="spher")
    out["oracle_port"] = dict(stage_s=r["stage_s"], total_s=r["total_s"], bits=r["bits"], n_nodes=r["n_nodes"])
    r2 = cpu_encode.encode_frame(xyz, sd, L, mullevel=False, mode="spher", full_window_runs=3)
    out["oracle_port_bounded_sample"] = dict(stage_s=r2["stage_s"], total_s=r2["total_s"], full_windows=r2["full_windows"],
                                             full_window_s=r2["full_window_s"])
    out["ratio_port_over_reference"] = r["total_s"] / out["reference"]["total_s"]
    out["ratio_bounded_sample_over_full_port"] = r2["total_s"] / r["total_s"]
    out["cpu"] = cpu_encode.cpu_model_name()
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "cpu_calibration_r2.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
