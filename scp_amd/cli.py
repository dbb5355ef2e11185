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
from .decoder import decode_file, write_sidecar
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
    p.add_argument("--host_transform", action="store_true",
                   help="strict identity with the reference from the frame on: the coordinate transform + quantiser run in numpy float32 on the "
                        "host exactly as data_preprocess.py:42-70 does (the device transform is more accurate, hence not bit-identical); "
                        "also SCP_XFORM=numpy")
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


MVUB_NAMES = ("andrew10", "david10", "phil10", "phil9", "ricardo10", "ricardo9", "sarah10")   # data_preprocess.py:242


def obj_ints(xyz, name, dev):
    """`--type obj` (encode_dataset.py:69-77 -> proc_pc defaults, data_preprocess.py:13-70): optional MVUB axis swap, offset = per-axis
    minimum, qs = 1, round half to even.  float32 subtraction then the float64 round of numpy, as index plumbing on the device."""
    p = torch.from_numpy(np.ascontiguousarray(xyz[:, :3], np.float32)).to(dev)
    if any(m in name for m in MVUB_NAMES):
        p = torch.stack((p[:, 0], p[:, 2], -p[:, 1]), 1)
    off = p.min(0)[0]
    return torch.round((p - off).double()).to(torch.int32).contiguous(), [float(v) for v in off.cpu()]


def refuse_unsupported(args, name, mullevel):
    """Flag combinations the reference accepts on its command line but cannot execute (or executes wrongly) are refused loudly
    instead of silently encoding something else."""
    if getattr(args, "spher_circle", False):
        raise native.ScpError("--spher_circle: the reference passes circle= to proc_pc, which has no such parameter "
                              "(encode_dataset_ehem.py:159-169 -> TypeError); there is no behaviour to reproduce")
    if name == "OctAttention" and args.level_wise and not mullevel:
        raise native.ScpError("--level_wise with OctAttention in encode.py stacks the padded per-level tables without removing the 1023 "
                              "pad rows (encode.py:59), so symbols are coded with the wrong rows; encode_mullevel.py has the fixed form "
                              "(encode_mullevel.py:60) - use it")
    if name == "OctAttention" and args.preproc_path:
        raise native.ScpError("--preproc_path with OctAttention is not supported (records are rebuilt on the device from the frame)")
    if name == "OctAttention" and args.metrics:
        raise native.ScpError("--metrics is available for the EHEM encoders only")
    if args.type == "obj" and (args.spher or args.cylin or mullevel and name != "OctAttention"):
        raise native.ScpError("--type obj is Cartesian and single-level in the reference (proc_pc defaults); drop --spher/--cylin/mullevel")
    if args.sequential and name != "OctAttention":
        raise native.ScpError("--sequential is an OctAttention mode (encode.py:38-41)")


def spawn_ranks(n, argv0, argv):
    """`--gpus N` outside torchrun: start N ranks (one per GPU) as a child launcher and hand back its exit code.  Runs before
    anything touches the GPU."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), argv0] + list(argv)
    return subprocess.call(cmd)


class Prefetch:
    """File reads (and the ascii PLY parse) one frame ahead on a reader thread, so the launch thread only enqueues GPU work.
    `post` (strict-identity mode: the host transform + quantiser) runs on the reader thread too; get() then returns (xyz, post(xyz))."""

    def __init__(self, files, depth=2, post=None):
        from concurrent.futures import ThreadPoolExecutor
        self.files = files
        self.pool = ThreadPoolExecutor(max_workers=1)
        self.depth = depth
        self.futs = {}
        self.next = 0
        self.post = post

    def _read(self, path):
        xyz = pointCloud.ptread(path)
        return xyz if self.post is None else (xyz, self.post(xyz))

    def get(self, k):
        while self.next < len(self.files) and self.next <= k + self.depth:
            self.futs[self.next] = self.pool.submit(self._read, self.files[self.next][1])
            self.next += 1
        return self.futs.pop(k).result()


def main(argv=None, mullevel=False):
    import sys
    args = get_args(argv, mullevel)
    rank, world, local = D.env_rank()
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        script = os.path.abspath(sys.argv[0])
        raise SystemExit(spawn_ranks(args.gpus, script, sys.argv[1:] if argv is None else argv))
    rank, world, local = D.init()
    if not torch.cuda.is_available():
        raise native.ScpError("encode needs an MI355X: the SCP hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    native.lib()
    D.pin_rank_threads(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))

    from .models import EHEM, OctAttention
    cfg = load_cfg(args.ckpt_path, args.model)
    name = cfg.model.class_name
    refuse_unsupported(args, name, mullevel)
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
    obj = args.type == "obj"
    if name == "OctAttention":
        mul = mullevel and args.spher and not obj       # encode_dataset_mullevel.py:76: the three-shell form exists for --spher
        enc = OctAttnFrameEncoder(model, args.type, args.lidar_level, spher=args.spher and not obj, cylin=args.cylin and not obj, device=dev,
                                  mullevel=mul, level_wise=args.level_wise and mullevel, named=mullevel,
                                  host_transform=True if args.host_transform else None)
    else:
        enc = FrameEncoder(model, args.type, args.lidar_level, spher=args.spher, cylin=args.cylin, mullevel=mullevel, device=dev,
                           host_transform=True if args.host_transform else None)

    mine = D.shard(files, rank, world)
    # fast path: frames are enqueued with encode_async (stage G on a side stream, two model lanes, range coder on a worker thread)
    # and finished three frames later - what bench.py measures.  Flows that need the frame's octree after the encode (--metrics),
    # come from record files, or run the one-window-per-node mode stay on the synchronous call.
    pipelined = not (args.preproc_path or args.metrics or args.sequential or obj)
    DEPTH = 3
    host_ints = enc.host_ints if (pipelined and name != "OctAttention" and enc.host_transform) else None
    reader = Prefetch(mine, post=host_ints)
    pending = []
    sums = [0.0, 0.0, 0.0, 0.0, 0.0]
    last_done = [time.time()]

    def stem_of(cur):
        return (cur.split("/")[-2] + Path(cur).stem) if (args.type == "kitti" and name != "OctAttention" and "/" in cur.rstrip("/")
                                                          and len(cur.split("/")) >= 2) else Path(cur).stem

    def report(cur, res, t_submit, dist=None):
        now = time.time()
        elapsed = now - max(last_done[0], t_submit)      # wall time this frame added to the run (frames overlap in the pipeline)
        last_done[0] = now
        outfile = enc.outfile(out_root + stem_of(cur), res)
        with open(outfile, "wb") as f:
            f.write(res["bytes"])
        if name != "OctAttention":
            torch.save(torch.Tensor(res["pos_mm"].astype(np.float32)), outfile + ".dat")     # encode.py:150
            write_sidecar(outfile, enc, res, name)
        print("outputfile                  :", outfile)
        print("time(s)                     :", elapsed)
        print("pt num                      :", res["n_points"])
        print("oct num                     :", res["n_nodes"])
        print("total binsize               :", res["bits"])
        print("bit per oct                 :", res["bits"] / res["n_nodes"])
        print("bit per pixel               :", res["bpp"])
        if dist:
            print("chamfer distance            :", dist["chamfer"])
            print("PSNR (D1)                   :", dist["psnr"])
        sums[0] += res["bpp"]; sums[1] += dist["psnr"] if dist else 0.0; sums[2] += dist["chamfer"] if dist else 0.0
        sums[3] += elapsed; sums[4] += 1

    if rank == 0:
        print("Encoding with", name)
    def fetch(k):
        d = reader.get(k)
        return d if host_ints is not None else (d, None)

    # the front part (stage G + window plans) of frame k + 1 runs on the encoder's front thread while this thread enqueues the model part of
    # frame k (FrameEncoder.front_async: the front part blocks its host thread for most of a frame time)
    front_ahead = pipelined and hasattr(enc, "front_async")
    nxt = None
    for k, (i, cur) in enumerate(mine):
        print("Encoding ", cur, i, "/", len(files))
        if front_ahead:
            xyz, ints, fr = nxt if nxt is not None else (*fetch(k), None)
            if fr is None:
                fr = enc.front_async(xyz, ints)
            nxt = None
            if k + 1 < len(mine):
                x2, i2 = fetch(k + 1)
                nxt = (x2, i2, enc.front_async(x2, i2))
        else:
            xyz, ints = fetch(k)
        t0 = time.time()
        if pipelined:
            pending.append((cur, enc.encode_async(xyz, ints, front=fr) if front_ahead else enc.encode_async(xyz), t0))
            if len(pending) > DEPTH:
                c0, h0, ts = pending.pop(0)
                report(c0, enc.finish(h0), ts)
            continue
        dist = None
        if args.preproc_path:
            # encode_dataset_ehem.py:149-157 / ..._mullevel.py:147-155: records + meta written by the test-set generator
            pp = args.preproc_path + ((cur.split("/")[-2] + Path(cur).stem) if args.type == "kitti" else Path(cur).stem)
            meta = np.load(pp + "_meta.npy")
            recs = [np.load(pp + sfx + ".npy") for sfx in (("_0_0", "_0_1", "_1") if mullevel else ("",))]
            res = enc.encode_records(recs, float(meta[0]), float(meta[2]) if len(meta) > 2 else 0.0, len(xyz))
        elif obj:
            q, off = obj_ints(xyz, cur, dev)
            res = enc.encode_ints(q, 0, len(xyz), sequential=args.sequential) if name == "OctAttention" else enc.encode_ints([q], 0, 0.0, len(xyz))
            res["quant"] = [dict(qs=[1.0, 1.0, 1.0], offset=off)]     # the sidecar is the only carrier of the per-axis minimum
        elif name == "OctAttention":
            res = enc.encode(xyz, sequential=args.sequential)
        else:
            res = enc.encode(xyz)
            if args.metrics:
                dist = enc.distortion(torch.from_numpy(np.ascontiguousarray(xyz[:, :3], np.float32)).to(dev))
        report(cur, res, t0, dist)
    for c0, h0, ts in pending:
        report(c0, enc.finish(h0), ts)
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


# ---------------------------------------------------------------------------------------------------------------- decode
def get_decode_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--ckpt_path", type=str, default="", help="example: outputs/obj/2023-04-28/10-43-45/ckpt/epoch=7-step=64088.ckpt")
    p.add_argument("--test_files", nargs="*", default=["data/obj/mpeg/8iVLSF_910bit/boxer_viewdep_vox9.ply"])
    p.add_argument("--preproc_path", type=str, default="")
    p.add_argument("--sequential_enc", action="store_true")
    p.add_argument("--level_wise", action="store_true")
    # additions
    p.add_argument("--random_weights", type=int, default=None)
    p.add_argument("--out_dir", type=str, default=None)
    p.add_argument("--lidar_level", type=int, default=None, help="overrides the side-info file / the reference's level-count rule")
    p.add_argument("--type", type=str, default=None, choices=[None, "obj", "kitti", "ford"])
    return p.parse_args(argv)


def find_stream(out_root, ori):
    """The stream the encode CLIs wrote for the original file `ori`: `<stem>[_spher|_cylin]_<levels>_<bin_num>_<z_offset>.bin` with
    stem = `<sequence dir><frame>` for KITTI EHEM runs (encode.py:140-144 via the dataset's file name) or the plain file stem.  The
    reference takes the first name that merely CONTAINS the frame number (decode_ehem.py:206-214), which picks the wrong sequence's
    stream as soon as two sequences hold the same frame number; here the name must match exactly, and 0 or several candidates are
    an error.  Returns (file name, the stem that matched) - the stem names the decoder's `.ply`."""
    import re
    p = Path(ori)
    names = [f for f in os.listdir(out_root) if f.endswith(".bin")]
    tried = []
    for stem in ([p.parent.name + p.stem] if p.parent.name else []) + [p.stem]:
        pat = re.compile("^" + re.escape(stem) + r"((_spher|_cylin)?_\d+_-?\d+_-?\d+)?\.bin$")
        c = sorted(f for f in names if pat.match(f))
        if len(c) == 1:
            return c[0], stem
        if len(c) > 1:
            raise native.ScpError(f"{len(c)} streams match {ori} in {out_root}: {c}")
        tried.append(stem)
    raise native.ScpError(f"no stream for {ori} in {out_root} (looked for {tried})")


def decode_main(argv=None, mullevel=False):
    """Drop-in for decode_ehem.py:191-255 / decode_ehem_mullevel.py:209-277: for every original file, find its stream in the
    test_output directory (`find_stream`: the exact name the encoder wrote, sequence included), decode it with the side info of `extract_info`,
    check the occupancy codes against the `--preproc_path` record files when they exist (the reference asserts this window by
    window, decode_ehem.py:184), rebuild the points (DeOctree -> de-quantise -> spher2cart / cylin2cart) and write
    `<test_output>/<stream stem>.ply` (KITTI: `<sequence><frame>.ply`, so two sequences never overwrite each other)."""
    args = get_decode_args(argv)
    if not torch.cuda.is_available():
        raise native.ScpError("decode needs an MI355X: the SCP hot path has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    native.lib()
    from .models import EHEM
    cfg = load_cfg(args.ckpt_path, "EHEM")
    if cfg.model.class_name != "EHEM":
        raise native.ScpError("decode_ehem*.py decode EHEM streams (the reference has no OctAttention decoder for these files)")
    if args.random_weights is not None or not args.ckpt_path:
        from .weights import fill_weights
        model = fill_weights(EHEM(cfg), args.random_weights or 0)
    else:
        model = EHEM.load_from_checkpoint(args.ckpt_path, cfg=cfg)
    model = model.to(dev).eval()
    if args.out_dir:
        out_root = args.out_dir.rstrip("/") + "/"
    else:
        root = args.ckpt_path.split("ckpt")[0] if args.ckpt_path else "./"
        out_root = root + "test_output" + (args.ckpt_path.split("ckpt")[1][:-1] if args.ckpt_path else "") + "/"
    files = args.test_files
    if files and os.path.isdir(files[0]):      # decode_ehem.py:202-203
        files = sorted(os.path.join(files[0], f) for f in os.listdir(files[0]) if f.endswith((".ply", ".bin")))
    else:
        files = expand_files(files)
    elapsed, results = 0.0, []
    for i, ori in enumerate(files):
        print(f"{i}/{len(files)}")
        name, stem = find_stream(out_root, ori)                # the stem that matched, not string surgery on the stream name
        binfile = out_root + name
        t0 = time.time()
        out = decode_file(binfile, model, args.lidar_level, args.type, mullevel, dev)
        torch.cuda.synchronize()
        t = time.time() - t0
        elapsed += t
        # the reference needs the record files (their length drives its loop and every window is asserted against them); here
        # they are an optional check
        npy = (args.preproc_path.rstrip("/") + "/" + Path(ori).stem) if args.preproc_path else str(ori).rsplit(".")[0]
        cands_npy = [npy + s + ".npy" for s in (("_0_0", "_0_1", "_1") if mullevel else ("",))]
        if not all(os.path.exists(c) for c in cands_npy) and args.preproc_path:
            alt = args.preproc_path.rstrip("/") + "/" + Path(ori).parent.name + Path(ori).stem       # kitti: <sequence><frame>
            cands_npy = [alt + s + ".npy" for s in (("_0_0", "_0_1", "_1") if mullevel else ("",))]
        if all(os.path.exists(c) for c in cands_npy):
            for codes, c in zip(out["codes"], cands_npy):
                want = np.load(c)[:, -1, 0]
                got = codes.cpu().numpy()[:len(want)]
                assert np.array_equal(got.astype(np.int64), want), f"decoded occupancy differs from {c}"
            print("checked against", ", ".join(cands_npy))
        print("decode succeeded, time:", t)
        print("oct len:", int(sum(len(c) for c in out["codes"])))
        print("avg dec time:", elapsed / (i + 1))
        ply = out_root + stem + ".ply"
        pointCloud.write_ply_data(ply, out["points"].cpu().numpy())
        print(ply)
        results.append((ply, out))
    print(elapsed / max(len(files), 1))
    return results
