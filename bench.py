#!/usr/bin/env python3
"""Headline benchmark: KITTI-like frames/s encode, SCP-EHEM, lidar_level 16 --spher --mullevel (BASELINE.json configs[2]).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...            (starts N ranks itself: a child `python -m torch.distributed.run`, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --all-configs            (the five BASELINE.json configurations, one JSON line each, into profiles/ with --out-dir)

One step = one synthetic 120 000-point frame through the whole hot path on one GPU: quantiser (3 shells) -> octree
serialisation -> context tables -> EHEM over every <= 8192-node window -> softmax/integer CDF -> range coder.  The frame
is resident in HBM before the timed region; the (c_low, c_high) pairs (4 B/node) go D2H and the serial range coder runs
on the host inside the timed region, as in a real encode.  Frames are independent: rank r encodes its own frames
(weak scaling), the only collective is the end-of-run all-reduce of the 5 summary scalars (RCCL).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="ehem-L16-m", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full", "none"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=4, help="frames in flight (encode_async handles not yet finished); 4 measured 1.2 % above 3, 5 no better")
    ap.add_argument("--batch", type=int, default=None, help="frames per stage-G / packed-forward / CDF launch (FrameEncoder.encode_batch_async); "
                    "default: 4 for the level-12 EHEM configuration (a 115 k-node frame is 22 windows), 1 elsewhere")
    ap.add_argument("--host-transform", action="store_true", help="strict-identity mode: numpy float32 transform + quantiser on the host (inside the timed region)")
    ap.add_argument("--all-configs", action="store_true", help="run every configuration of CONFIGS in turn (child processes), one JSON line each")
    ap.add_argument("--out-dir", default=None, help="with --all-configs: also write <out-dir>/<tag>_bench_<config>.json")
    ap.add_argument("--tag", default="r3", help="file-name prefix used with --out-dir")
    return ap.parse_args()


def spawn_ranks(n, argv):
    """`bench.py --gpus N` outside a torchrun environment: N ranks (one per GPU) as a CHILD `python -m torch.distributed.run`,
    started before this process has touched the GPU; its exit code is ours and rank 0's JSON line passes through on stdout."""
    from scp_amd.cli import spawn_ranks as _spawn
    return _spawn(n, os.path.abspath(__file__), argv)


def run_all_configs(args, argv):
    """One child process per configuration (a fresh process per run keeps lanes, caches and allocator state apart)."""
    import subprocess
    rest = [a for a in argv if a != "--all-configs"]
    for drop in ("--config", "--out-dir", "--tag"):
        while drop in rest:
            i = rest.index(drop)
            del rest[i:i + 2]
    rc = 0
    for name in ("ehem-L16-m", "ehem-L12-s", "ehem-F17-m", "octattn-L12-spher", "octattn-L14-cylin"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", name] + rest, stdout=subprocess.PIPE, text=True)
        rc = rc or r.returncode
        line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
        if line is None:
            print(json.dumps({"config": name, "error": f"exit code {r.returncode}"}), flush=True)
            continue
        print(line, flush=True)
        if args.out_dir:
            os.makedirs(args.out_dir, exist_ok=True)
            with open(os.path.join(args.out_dir, f"{args.tag}_bench_{name.replace('-', '_')}.json"), "w") as f:
                f.write(line + "\n")
    return rc


def measure_dominant_kernel(enc, xyz_dev):
    """Live HIP-event timing over one frame, on the stream the kernels are launched on (torch's current stream, which is where
    the C ABI launches them): the dominant kernel (gemm_split_kernel, all tile / epilogue variants) and, for the secondary
    roofline entries, the window attention and the feature-space kNN searches.  ALGORITHMIC flops only."""
    from scp_amd import native
    recs = {"gemm": [], "attn": [], "knn": [], "mlp": [], "post": [], "lnlin": []}

    def ev():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    o_lin, o_att, o_knn, o_mlp = native.linear_split, native.swin_attention_packed, native.knn_topk_packed, native.mlp_split_fused
    o_post, o_lnlin = native.swin_post_attn, native.swin_ln_linear
    o_attp, o_lnqkv = native.swin_attention_packed_planes, native.swin_ln_qkv

    def post(o, x, pw, *args, **kw):
        s, e = ev(); s.record(); y = o_post(o, x, pw, *args, **kw); e.record()
        recs["post"].append((s, e, 2.0 * x.shape[0] * (256 * 256 + 2 * 256 * 1024)))     # proj + fc1 + fc2
        return y

    def lnlin(x, fw, *args, **kw):
        s, e = ev(); s.record(); y = o_lnlin(x, fw, *args, **kw); e.record()
        recs["lnlin"].append((s, e, 2.0 * x.shape[0] * fw.N * 256))
        return y

    def lnqkv(x, fw, *args, **kw):                  # the same projection with the keys / values leaving as attention planes
        s, e = ev(); s.record(); y = o_lnqkv(x, fw, *args, **kw); e.record()
        recs["lnlin"].append((s, e, 2.0 * x.shape[0] * fw.N * 256))
        return y

    def attp(q, *args, **kw):                       # plane-fed form of the window attention: same flops
        s, e = ev(); s.record(); y = o_attp(q, *args, **kw); e.record()
        recs["attn"].append((s, e, q.shape[0] * 2.0 * 2.0 * 512 * 256))
        return y

    def lin(a, sw, *args, **kw):
        s, e = ev(); s.record(); y = o_lin(a, sw, *args, **kw); e.record()
        recs["gemm"].append((s, e, 2.0 * a.M * sw.N * sw.K))
        return y

    def att(q, *args, **kw):
        s, e = ev(); s.record(); y = o_att(q, *args, **kw); e.record()
        recs["attn"].append((s, e, q.shape[0] * 2.0 * 2.0 * 512 * 256))      # per row: QK^T and PV over 512 keys x 256 channels
        return y

    def knn(x, ktab):
        s, e = ev(); s.record(); y = o_knn(x, ktab); e.record()
        n = ktab[:, 1].double()
        recs["knn"].append((s, e, float((n * 512).sum().item()) * 2.0 * max(4, x.shape[1]), x.shape[1]))   # sum over 512-row chunks of n * 512 pairs
        return y

    def mlp(a, *args, **kw):
        s, e = ev(); s.record(); y = o_mlp(a, *args, **kw); e.record()
        recs["mlp"].append((s, e, 2.0 * a.M * 256 * 1024 * 2))              # fc1 + fc2
        return y

    native.linear_split, native.swin_attention_packed, native.knn_topk_packed, native.mlp_split_fused = lin, att, knn, mlp
    native.swin_post_attn, native.swin_ln_linear = post, lnlin
    native.swin_attention_packed_planes, native.swin_ln_qkv = attp, lnqkv
    try:
        enc.encode(xyz_dev)
        torch.cuda.synchronize()
    finally:
        native.linear_split, native.swin_attention_packed, native.knn_topk_packed, native.mlp_split_fused = o_lin, o_att, o_knn, o_mlp
        native.swin_post_attn, native.swin_ln_linear = o_post, o_lnlin
        native.swin_attention_packed_planes, native.swin_ln_qkv = o_attp, o_lnqkv

    def summ(rs):
        ms = sum(r[0].elapsed_time(r[1]) for r in rs)
        fl = sum(r[2] for r in rs)
        return dict(launches=len(rs), avg_launch_us=1e3 * ms / max(1, len(rs)), flops_per_launch=fl / max(1, len(rs)),
                    tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, total_ms=ms)
    # dominant kernel: the post-attention row-chain kernel when the Swin blocks run on it (default), else the split GEMM of rounds 1 - 2
    out = summ(recs["post"]) if recs["post"] else summ(recs["gemm"])
    out["dominant"] = "post" if recs["post"] else "gemm"
    out["gemm"] = summ(recs["gemm"])
    out["lnlin"] = summ(recs["lnlin"])
    out["attn"] = summ(recs["attn"])
    out["mlp"] = summ(recs["mlp"])
    out["knn_feat"] = summ([r for r in recs["knn"] if r[3] > 4])
    out["knn_pos"] = summ([r for r in recs["knn"] if r[3] <= 4])
    return out


def measure_dominant_kernel_octattn(enc, xyz_dev):
    """Same for the OctAttention path: its dense layers (f16x3 split GEMM) and the dual-stream causal attention."""
    from scp_amd import native
    recs = {"gemm": [], "attn": []}

    def ev():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    o_lin, o_att = native.linear_f16x3, native.octattn_attention

    def lin(x, sw, *args, **kw):
        s, e = ev(); s.record(); y = o_lin(x, sw, *args, **kw); e.record()
        recs["gemm"].append((s, e, 2.0 * (x.numel() // sw.K) * sw.N * sw.K))
        return y

    def att(q_u, *args, **kw):
        s, e = ev(); s.record(); y = o_att(q_u, *args, **kw); e.record()
        B, c, D = q_u.shape
        recs["attn"].append((s, e, B * 3.0 * 2.0 * c * c * D))       # heads x (2 c^2 150) x 3 score / AV products per layer
        return y

    native.linear_f16x3, native.octattn_attention = lin, att
    try:
        enc.encode(xyz_dev)
        torch.cuda.synchronize()
    finally:
        native.linear_f16x3, native.octattn_attention = o_lin, o_att

    def summ(rs):
        ms = sum(r[0].elapsed_time(r[1]) for r in rs)
        fl = sum(r[2] for r in rs)
        return dict(launches=len(rs), avg_launch_us=1e3 * ms / max(1, len(rs)), flops_per_launch=fl / max(1, len(rs)),
                    tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, total_ms=ms)
    out = summ(recs["gemm"])
    out["attn"] = summ(recs["attn"])
    return out


def cpu_baseline(cfg, xyz, full=False):
    """The CPU oracle (a port of the reference path: C octree / records / CDF / range coder + the functional PyTorch-CPU model,
    oracle/cpu_encode.py) on ONE frame of the bench's own workload, every stage timed.  Bounded sample: all windows shorter than
    the model's context are run; full windows (identical shapes, data-independent cost) are run three times - one warm-up, the
    the faster of the other two stands for the rest.  `--cpu-baseline full` runs every window (minutes per frame)."""
    from cfgs import ehem_cfg, octattn_cfg
    from oracle import cpu_encode
    from scp_amd.models import EHEM, OctAttention
    from scp_amd.weights import fill_weights
    threads = min(os.cpu_count() or 1, 32)   # PyTorch-CPU oversubscribes badly beyond this on 128-core hosts
    torch.set_num_threads(threads)
    runs = None if full else 3
    if cfg["model"] == "EHEM":
        sd = fill_weights(EHEM(ehem_cfg()), 0).state_dict()
        r = cpu_encode.encode_frame(xyz, sd, cfg["level"], mullevel=cfg["mullevel"], mode=cfg["mode"], full_window_runs=runs,
                                    data_type=cfg.get("type", "kitti"))
    else:
        sd = fill_weights(OctAttention(octattn_cfg()), 0).state_dict()
        r = cpu_encode.encode_frame_octattn(xyz, sd, cfg["level"], mode=cfg["mode"], full_window_runs=runs)
    what = (f"one whole frame, all {r['windows']} windows run" if full else
            f"one frame: quantiser/octree/records/context of the whole frame, all {r['partial_windows']} partial windows, "
            f"{r['full_windows_run']} of the {r['full_windows']} full windows (first = warm-up, the fastest of the rest x {r['full_windows']}), "
            f"CDF + range coder on the {r['rows_coded']} rows produced, scaled to {r['n_nodes']} nodes")
    return dict(value=1.0 / r["total_s"], unit="frames/s", cores=threads, kind="port", sample=what, seconds_per_frame=r["total_s"],
                stage_s=r["stage_s"], full_window_s=r["full_window_s"], host_cpu=cpu_encode.cpu_model_name(), host_cores=os.cpu_count())


CONFIGS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "ehem-L16-m": dict(model="EHEM", level=16, mullevel=True, mode="spher",
                       workload="SCP-EHEM KITTI-like synthetic 120k-pt frames, --spher --mullevel lidar_level=16 (BASELINE.json configs[2])"),
    # configs[1]: same-level level 12 (run it with --steps 16 for the batch of 16 frames)
    "ehem-L12-s": dict(model="EHEM", level=12, mullevel=False, mode="spher",
                       workload="SCP-EHEM KITTI-like synthetic 120k-pt frames, --spher lidar_level=12 (BASELINE.json configs[1])"),
    # configs[3]: Ford-like frames (the same clouds in integer millimetres), level 17 multi-level
    "ehem-F17-m": dict(model="EHEM", level=17, mullevel=True, mode="spher", type="ford",
                       workload="SCP-EHEM Ford-like synthetic 120k-pt frames (integer mm), --spher --mullevel lidar_level=17 (BASELINE.json configs[3])"),
    # configs[0]'s workload on the GPU / configs[4]
    "octattn-L12-spher": dict(model="OctAttention", level=12, mullevel=False, mode="spher",
                              workload="SCP-OctAttention KITTI-like synthetic 120k-pt frames, --spher lidar_level=12 (BASELINE.json configs[0] workload)"),
    "octattn-L14-cylin": dict(model="OctAttention", level=14, mullevel=False, mode="cylin",
                              workload="SCP-OctAttention KITTI-like synthetic 120k-pt frames, --cylin lidar_level=14 (BASELINE.json configs[4])"),
}


def main():
    args = parse()
    if args.all_configs:
        raise SystemExit(run_all_configs(args, sys.argv[1:]))
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))       # nothing has touched the GPU yet
    cfg = CONFIGS[args.config]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # test hooks (used by a 2-rank dry run on a 1-GPU box): SCP_FORCE_DEVICE pins every rank to one GPU, SCP_DIST_BACKEND=gloo
    local_dev = int(os.environ.get("SCP_FORCE_DEVICE", local))
    backend = os.environ.get("SCP_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    from cfgs import ehem_cfg, octattn_cfg
    from scp_amd import distributed as D
    from scp_amd import native
    from scp_amd.encoder import EncodePlan, FrameEncoder, OctAttnFrameEncoder
    from scp_amd.models import EHEM, OctAttention
    from scp_amd.synth import synth_frame
    from scp_amd.weights import fill_weights
    native.lib()
    # every rank keeps its launch thread, range-coder worker and reader on its own cores (scp_amd/distributed.py)
    pinned = D.pin_rank_threads(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    ehem = cfg["model"] == "EHEM"
    if ehem:
        model = fill_weights(EHEM(ehem_cfg()), 0).to(dev)
        enc = FrameEncoder(model, cfg.get("type", "kitti"), cfg["level"], spher=cfg["mode"] == "spher", cylin=cfg["mode"] == "cylin",
                           mullevel=cfg["mullevel"], device=dev, host_transform=True if args.host_transform else None)
    else:
        model = fill_weights(OctAttention(octattn_cfg()), 0).to(dev)
        enc = OctAttnFrameEncoder(model, "kitti", cfg["level"], spher=cfg["mode"] == "spher", cylin=cfg["mode"] == "cylin", device=dev,
                                  host_transform=True if args.host_transform else None)

    total = args.warmup + args.steps
    frames_host = [synth_frame(rank * 1000 + i) for i in range(total)]
    from scp_amd.synth import ford_like
    if cfg.get("type") == "ford":
        frames_host = [ford_like(f) for f in frames_host]
    frames = [torch.from_numpy(f).to(dev) for f in frames_host]      # resident in HBM before the timed region
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    batch = args.batch if args.batch is not None else (4 if args.config == "ehem-L12-s" else 1)
    if batch > 1 and not ehem:
        raise SystemExit("--batch is an EHEM option")
    for i in range(args.warmup):
        enc.finish(enc.encode_async(frames[i]))
    if batch > 1:
        enc.finish_batch(enc.encode_batch_async(frames[:batch]))
    barrier()
    cpu0 = time.process_time()                      # CPU seconds of this rank, all threads (launch thread, coder worker, reader)
    t0 = time.perf_counter()
    # frame i is range-coded on a worker thread while frames i+1 .. i+depth run on the GPU; a handle pins its ~590 MB logits table,
    # so at most `depth` frames are in flight (memory stays O(depth), not O(steps))
    pending, results = [], []
    if batch > 1:      # `batch` frames per launch sequence, two batches in flight (the range coder of one under the kernels of the next)
        for i in range(args.warmup, total, batch):
            pending.append(enc.encode_batch_async(frames[i:min(total, i + batch)]))
            if len(pending) > 1:
                results += enc.finish_batch(pending.pop(0))
        for h in pending:
            results += enc.finish_batch(h)
    else:
        for i in range(args.warmup, total):
            pending.append(enc.encode_async(frames[i]))
            if len(pending) > args.depth:
                results.append(enc.finish(pending.pop(0)))
        results += [enc.finish(h) for h in pending]
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0               # this rank's own time for its frames (before the barrier)
    cpu_ms = 1e3 * (time.process_time() - cpu0) / args.steps
    barrier()
    dt = time.perf_counter() - t0
    rank_stats = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # per-rank view for the scaling curve: is the limiter on the host side?  Every rank also encodes ONE shared frame (outside
        # the timed region): the streams must be byte-identical on every GPU.
        import hashlib
        shared = enc.finish(enc.encode_async(torch.from_numpy(synth_frame(0) if cfg.get("type") != "ford" else ford_like(synth_frame(0))).to(dev)))
        mine = dict(rank=rank, device=torch.cuda.current_device(), fps=args.steps / dt_own, host_cpu_ms_per_frame=cpu_ms,
                    shared_frame_sha256=hashlib.sha256(shared["bytes"]).hexdigest())
        rank_stats = [None] * world
        dist.all_gather_object(rank_stats, mine)

    # end-of-run summary reduction (encode.py:293-305): [sum bpp, sum psnr, sum chamfer, sum time, count] over all ranks
    summ = torch.tensor([sum(r["bpp"] for r in results), 0.0, 0.0, dt, len(results)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(summ, op=dist.ReduceOp.SUM)
    summ = summ.cpu().numpy()

    if rank == 0:
        n_nodes = results[-1]["n_nodes"]
        P = results[-1]["n_points"]
        peak3 = BF16_MFMA_PEAK_TFLOPS / 3.0
        bytes_G = 12 * P + 25 * n_nodes            # SURVEY.md §8d algorithmic bytes of stage G
        bytes_C = n_nodes * (255 * 4 + 4)
        traffic = traffic_src = None
        if ehem:
            dom = measure_dominant_kernel(enc, frames[-1])
            st = enc.encode(frames[-1], timing=True)["times"]     # per-stage wall times with a device sync after every stage
            key = "rc_post_attn_kernel" if dom["dominant"] == "post" else "gemm_split_all_variants"
            for name in ("r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1z_pmc_traffic.json"):   # HBM bytes per launch from the committed PMC passes
                try:
                    with open(os.path.join(ROOT, "profiles", name)) as f:
                        traffic = json.load(f)[key]["hbm_bytes_per_launch"]
                    traffic_src = "profiles/" + name + " (separate rocprofv3 --pmc passes over the same frame; not measured by this run)"
                    break
                except Exception:
                    pass
            metric = "KITTI frames/sec encode (SCP-EHEM, level 16) + bpp match vs ref"
            if args.config == "ehem-F17-m":
                metric = "Ford-like frames/sec encode (SCP-EHEM, level 17 multi-level) + bpp match vs ref"
            elif args.config != "ehem-L16-m":
                metric = f"KITTI frames/sec encode (SCP-EHEM, level {cfg['level']} same-level) + bpp match vs ref"
            dtype = ("f32 (dense layers and attention as bf16x3 split on bf16 MFMA, feature kNN as f16x3 split on f16 MFMA, fp32 accumulate; "
                     "position kNN / CDF in fp32)")
            kernel = ("rc_post_attn_kernel (attention projection + residual + LayerNorm + fc1 + GELU + fc2 + residual of a Swin block in one launch, "
                      "accumulators chained through registers: 3x v_mfma_f32_32x32x16_bf16 per fp32-class product)" if dom["dominant"] == "post" else
                      "gemm_split_kernel (dense layers, both operands pre-split: 3x v_mfma_f32_32x32x16_bf16 per fp32-class product)")
            windows = len(EncodePlan(results[-1]["level_sizes"], 8192).windows)
        else:
            dom = measure_dominant_kernel_octattn(enc, frames[-1])
            st = {"total": enc.encode(frames[-1])["times"]["total"]}
            metric = f"KITTI frames/sec encode (SCP-OctAttention, level {cfg['level']} --{cfg['mode']})"
            dtype = "f32 (dense layers and attention as f16x3 split on f16 MFMA with power-of-two row scales, fp32 accumulate; CDF in fp32)"
            kernel = "gemm_bf16x3_kernel<ACT, F16=true> (OctAttention dense layers: 3x v_mfma_f32_32x32x16_f16 per fp32-class product)"
            windows = -(-(n_nodes + 1023) // 1024)
        out = {
            "metric": metric,
            "value": world * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": cfg["workload"] + ", seeded random weights", "nodes_per_frame": int(n_nodes), "windows_per_frame": windows,
                       "frames_per_gpu": args.steps, "frames_in_flight": args.depth if batch == 1 else 2 * batch, "frames_per_launch_sequence": batch, "transform": "host-numpy (strict identity)" if enc.host_transform else "device", "parallelism": f"frame-sharded x{world}",
                       "rank_cores": len(pinned) if pinned else None},
            "rccl_world": dist.get_world_size() if world > 1 else 1, "dist_backend": backend if world > 1 else None,
            "host_cpu_ms_per_frame": cpu_ms,
            "bpp_mean": float(summ[0] / summ[4]),
            "stage_ms": {k: round(1e3 * v, 3) for k, v in st.items()},
            # dominant kernel: the x3-split dense layer.  `achieved` counts ALGORITHMIC flops (2*M*N*K of the fp32 product it
            # replaces); the kernel spends three 16-bit MFMAs per product, so its own ceiling is a third of the dense 16-bit peak.
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": dom["tflops"], "peak": peak3, "unit": "TFLOP/s", "frac": dom["tflops"] / peak3,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "peak_note": "2500 TFLOP/s dense 16-bit MFMA / 3 products; the fp32 MFMA peak this replaces is 157.3.  With all 256 CUs multiplying the clock settles at 1.85 GHz (tools/src/mb_power.cpp): 1.95 PFLOP/s sustained, 650 per fp32-class product",
                         "launches_per_frame": dom["launches"], "avg_launch_us": dom["avg_launch_us"], "flops_per_launch": dom["flops_per_launch"]},
        }

        def entry(d, **kw):
            return dict(bound="mfma", achieved=d["tflops"], peak=peak3, unit="TFLOP/s", frac=d["tflops"] / peak3, launches_per_frame=d["launches"],
                        avg_launch_us=d["avg_launch_us"], **kw)
        if ehem:
            # secondary kernels, same convention; the position search (3 features) is selection-bound, its MFMA share is negligible
            out["roofline_kernels"] = {
                "rc_ln_linear_kernel": entry(dom["lnlin"], note="LayerNorm + q|k|v projection in one launch, rows resident as MFMA B fragments; keys / values leave as the attention kernel's bf16 planes"),
                "gemm_split_kernel": entry(dom["gemm"], note="the remaining dense layers (geometry MLPs, patch merges, concat layers, probability heads)"),
                "mlp_fused_kernel": entry(dom["mlp"], note="fc1 + GELU + fc2 + residual of a Swin block in one launch, hidden activation in LDS (SCP_SWIN=split only)"),
                "swin_attn_planes_kernel": entry(dom["attn"], note="window attention, K / V tiles staged by LDS-DMA from pre-split planes (SCP_ATTN_KV=rows: swin_attn_bf16x3_kernel)"),
                "knn_f16x3_wg256_kernel": entry(dom["knn_feat"], note="fused distance + top-20 selection, 256-query workgroups on the XCD-affine schedule; every phase of a wave is latency-bound (DESIGN.md 4.5), L2-miss traffic 2.0 GB per launch"),
                "knn_mfma_kernel<2,16> (positions)": {"bound": "valu", "launches_per_frame": dom["knn_pos"]["launches"],
                                                      "avg_launch_us": dom["knn_pos"]["avg_launch_us"]}}
            out["roofline_kernels"] = {k: v for k, v in out["roofline_kernels"].items() if v.get("launches_per_frame", 1)}   # kernels this configuration never launched
            out["roofline_stages"] = {
                "G": {"bound": "hbm", "achieved": bytes_G / st["geom"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": bytes_G / st["geom"] / 1e9 / HBM_PEAK_GBS, "bytes": bytes_G, "note": "host wall time of the whole stage incl. its small D2H syncs"},
                "C": {"bound": "hbm", "achieved": bytes_C / st["cdf"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": bytes_C / st["cdf"] / 1e9 / HBM_PEAK_GBS, "bytes": bytes_C, "note": "includes the 4 B/node D2H copy"}}
        else:
            out["roofline_kernels"] = {"oa_attn_f16x3_kernel": entry(dom["attn"], note="dual-stream causal attention, non-causal flop count (SURVEY.md 8d)")}
        if rank_stats:
            fps = [r["fps"] for r in rank_stats]
            cpu = [r["host_cpu_ms_per_frame"] for r in rank_stats]
            out["ranks"] = {"fps_min": min(fps), "fps_max": max(fps), "host_cpu_ms_per_frame_min": min(cpu), "host_cpu_ms_per_frame_max": max(cpu),
                            "shared_frame_streams_identical": len({r["shared_frame_sha256"] for r in rank_stats}) == 1, "per_rank": rank_stats}
        mode = "none" if args.no_cpu_baseline else args.cpu_baseline
        if world == 1 and mode != "none":
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, frames_host[-1], full=mode == "full")
            except Exception as e:   # the baseline is a reported number, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
