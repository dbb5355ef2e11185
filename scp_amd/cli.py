"""Drop-in command lines: `encode.py` (encode.py:315-339) and `encode_mullevel.py` (encode_mullevel.py:235-260).

Same flags, same output files (`<run>/test_output<ckpt-stem>/<frame>[_spher|_cylin]_<levels>_<bin_num>_<z_offset>.bin` + `.dat`),
same stdout block per frame, same summary file.  Differences, all additive:
  * `--gpus N` / torchrun: frames are sharded across ranks and the summary is all-reduced (scp_amd/distributed.py);
    the GPU is selected by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES like CUDA_VISIBLE_DEVICES in the reference;
  * the hydra run directory is read with a plain YAML loader; `--random_weights SEED` replaces the checkpoint for
    smoke runs (no trained checkpoint can be fetched offline);
  * `--test_files` accepts a glob, a directory or a list (the reference's `glob.glob(list)` cannot work, SURVEY B-3).
  * `--metrics`: chamfer distance and D1 PSNR of every frame, computed on the device (scp_amd/metrics.py; the reference
    shells out to pc_error and a CPU KD-tree for these) - they enter the per-frame block and the all-reduced summary.
"""
import argparse
import glob
import os
import time
from pathlib import Path

import numpy as np
import torch

from . import distributed as D
from . import native
from .data_preproc import pt as pointCloud
from .encoder import FrameEncoder, OctAttnFrameEncoder


class Cfg(dict):
    def __getattr__(self, k):
        v = self[k]
        return Cfg(v) if isinstance(v, dict) else v


EHEM_DEFAULT = dict(class_name="EHEM", context_size=8192, token_num=255, level_k=4, max_level=19)
OCTATTN_DEFAULT = dict(class_name="OctAttention", max_octree_level=12, context_size=1024, token_num=255, layer_num=3,
                       head_num=4, abs_pos_embed_dim=12, occ_embed_dim=128, level_embed_dim=6, octant_embed_dim=4,
                       hidden_dimension=300, pos_max_len=5000, level_k=4, pos_embed=True)


def load_cfg(ckpt_path, model_name=None):
    """encode.py:238-244: the hydra snapshot next to the checkpoint, else the reference's config defaults."""
    root = ckpt_path.split("ckpt")[0] if ckpt_path else ""
    y = Path(root, ".hydra", "config.yaml")
    if ckpt_path and y.exists():
        import yaml
        with open(y) as f:
            raw = yaml.safe_load(f)
        raw.setdefault("data", {}).setdefault("extra_pos", False)
        raw.setdefault("train", {}).setdefault("type", "kitti")
        return Cfg(raw)
    model = dict(EHEM_DEFAULT if (model_name or "EHEM") == "EHEM" else OCTATTN_DEFAULT)
    return Cfg(model=model, data=dict(extra_pos=False), train=dict(type="kitti", dropout=0.0))


def get_args(argv=None, mullevel=False):
    p = argparse.ArgumentParser()
    p.add_argument("--ckpt_path", type=str, default="", help="example: outputs/obj/2023-04-28/10-43-45/ckpt/epoch=7-step=64088.ckpt")
    p.add_argument("--test_files", nargs="*", default=["data/obj/mpeg/8iVLSF_910bit/boxer_viewdep_vox9.ply"])
    p.add_argument("--sequential", action="store_true")
    p.add_argument("--type", type=str, default="obj", choices=["obj", "kitti", "ford"])
    p.add_argument("--lidar_level", type=int, default=12)
    p.add_argument("--level_wise", action="store_true")
    p.add_argument("--cylin", action="store_true")
    p.add_argument("--spher", action="store_true")
    if not mullevel:
        p.add_argument("--spher_circle", action="store_true")
    p.add_argument("--preproc_path", type=str, default="")
    # additions
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--model", type=str, default=None, choices=[None, "EHEM", "OctAttention"])
    p.add_argument("--random_weights", type=int, default=None)
    p.add_argument("--out_dir", type=str, default=None)
    p.add_argument("--metrics", action="store_true", help="chamfer distance + D1 PSNR per frame (computed on the device)")
    return p.parse_args(argv)


def expand_files(specs):
    out = []
    for s in specs:
        if os.path.isdir(s):
            out += sorted(glob.glob(os.path.join(s, "**", "*.bin"), recursive=True) + glob.glob(os.path.join(s, "**", "*.ply"), recursive=True))
        elif "*" in s:
            out += sorted(glob.glob(s))
        else:
            out.append(s)
    return out


def main(argv=None, mullevel=False):
    args = get_args(argv, mullevel)
    rank, world, local = D.init()
    if not torch.cuda.is_available():
        raise native.ScpError("encode needs an MI355X: the SCP hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    native.lib()

    from .models import EHEM, OctAttention
    cfg = load_cfg(args.ckpt_path, args.model)
    name = cfg.model.class_name
    cls = OctAttention if name == "OctAttention" else EHEM
    if args.random_weights is not None or not args.ckpt_path:
        from .weights import fill_weights
        model = fill_weights(cls(cfg), args.random_weights or 0)
    else:
        model = cls.load_from_checkpoint(args.ckpt_path, cfg=cfg)
    model = model.to(dev).eval()

    if args.out_dir:
        out_root = args.out_dir.rstrip("/") + "/"
    else:
        root = args.ckpt_path.split("ckpt")[0] if args.ckpt_path else "./"
        out_root = root + "test_output" + (args.ckpt_path.split("ckpt")[1][:-1] if args.ckpt_path else "") + "/"
    os.makedirs(out_root, exist_ok=True)

    files = expand_files(args.test_files)
    combine = len(files) > 1
    spher = args.spher or getattr(args, "spher_circle", False)
    if name == "OctAttention":
        enc = OctAttnFrameEncoder(model, args.type, args.lidar_level, spher=spher, cylin=args.cylin, device=dev)
    else:
        enc = FrameEncoder(model, args.type, args.lidar_level, spher=spher, cylin=args.cylin, mullevel=mullevel, device=dev)

    sums = [0.0, 0.0, 0.0, 0.0, 0.0]
    if rank == 0:
        print("Encoding with", name)
    for i, cur in D.shard(files, rank, world):
        print("Encoding ", cur, i, "/", len(files))
        xyz = pointCloud.ptread(cur)
        stem = (cur.split("/")[-2] + Path(cur).stem) if (args.type == "kitti" and name != "OctAttention" and "/" in cur.rstrip("/")
                                                          and len(cur.split("/")) >= 2) else Path(cur).stem
        t0 = time.time()
        if args.preproc_path and name != "OctAttention":
            # encode_dataset_ehem.py:149-157 / ..._mullevel.py:147-155: records + meta written by the test-set generator
            pp = args.preproc_path + ((cur.split("/")[-2] + Path(cur).stem) if args.type == "kitti" else Path(cur).stem)
            meta = np.load(pp + "_meta.npy")
            recs = [np.load(pp + sfx + ".npy") for sfx in (("_0_0", "_0_1", "_1") if mullevel else ("",))]
            res = enc.encode_records(recs, float(meta[0]), float(meta[2]) if len(meta) > 2 else 0.0, len(xyz))
        else:
            res = enc.encode(xyz, sequential=True) if (args.sequential and name == "OctAttention") else enc.encode(xyz)
        elapsed = time.time() - t0
        outfile = enc.outfile(out_root + stem, res)
        with open(outfile, "wb") as f:
            f.write(res["bytes"])
        if name != "OctAttention":
            torch.save(torch.Tensor(res["pos_mm"].astype(np.float32)), outfile + ".dat")     # encode.py:150
        print("outputfile                  :", outfile)
        print("time(s)                     :", elapsed)
        print("pt num                      :", res["n_points"])
        print("oct num                     :", res["n_nodes"])
        print("total binsize               :", res["bits"])
        print("bit per oct                 :", res["bits"] / res["n_nodes"])
        print("bit per pixel               :", res["bpp"])
        dist = None
        if args.metrics and name != "OctAttention" and not args.preproc_path:
            dist = enc.distortion(torch.from_numpy(np.ascontiguousarray(xyz[:, :3], np.float32)).to(dev))
            print("chamfer distance            :", dist["chamfer"])
            print("PSNR (D1)                   :", dist["psnr"])
        sums = [sums[0] + res["bpp"], sums[1] + (dist["psnr"] if dist else 0.0), sums[2] + (dist["chamfer"] if dist else 0.0),
                sums[3] + elapsed, sums[4] + 1]
    total = D.reduce_summary(sums, dev)
    m = D.summary_means(total)
    if rank == 0:
        print("sample number:", m["count"])
        print("times:", m["time"])
        print("bpp:", m["bpp"])
        if combine and args.type in ("kitti", "ford"):
            tag = "mul" if mullevel else "same"
            out = (f"{tag} {args.lidar_level} {args.test_files} {args.ckpt_path}\nsample number: {m['count']}\ntimes: {m['time']}\n"
                   f"bpp: {m['bpp']}\nchamfer_dist: {m['chamfer']}\nPSNR: {m['psnr']}\n\n")
            with open(f"test_results_{tag}_{args.type}_{args.lidar_level}.txt", "a") as f:
                f.write(out)
    D.finalize()
    return m
