#!/usr/bin/env python3
"""Headline benchmark: KITTI-like frames/s encode, SCP-EHEM, lidar_level 16 --spher --mullevel (BASELINE.json configs[2]).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...            (starts N ranks itself: a child `python -m torch.distributed.run`, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --all-configs            (the five BASELINE.json configurations, one JSON line each, into profiles/ with --out-dir)
    python bench.py --decode                 (the decoder on the same configuration: frames/s, stage split)
    python bench.py --no-legs                (the headline only: no side legs - see below)

The headline `value` is measured in STRICT-IDENTITY mode (round 5): the float -> integer step is the reference's own numpy float32 arithmetic
(data_preprocess.py:42-70,171-214) on a prefetch thread, inside the timed region, and the run asserts that 0 points of the seed-0 frame differ
from the reference's integers (`transform_parity`); the device-transform rate is the extra key `device_transform`.
After the headline (outside its timed region, N = 1 only) the default run adds short legs, each a child process running this file
(5 warm-up + 16 timed frames; 32 for the batched L12 workload): the other four BASELINE.json workloads and the decoder under `configs`, and a `cli` leg - `.bin` files on disk
through the drop-in `encode_mullevel.py` (file read, parse, host -> device copy and the written `.bin` / `.dat` inside) - with `cli_over_bench`;
`decode_2_procs` / `decode_4_procs`: independent decoder processes sharing the GPU (a decode is a serial chain of short launches: streams scale by process).

One step = one synthetic 120 000-point frame through the whole hot path on one GPU: quantiser (3 shells) -> octree
serialisation -> context tables -> EHEM over every <= 8192-node window -> softmax/integer CDF -> range coder.  The frame
is resident in HBM before the timed region; the (c_low, c_high) pairs (4 B/node) go D2H and the serial range coder runs
on the host inside the timed region, as in a real encode.  Frames are independent: rank r encodes its own frames
(weak scaling), the only collective is the end-of-run all-reduce of the 5 summary scalars (RCCL).
Prints ONE JSON line on rank 0.

How the roofline numbers of the line are measured (round 4): `native.launch_profile` switches on the launch brackets of the C ABI
(include/scp_debug.h: scp_prof_*): inside libscp_hip.so every hot entry point records a hipEvent immediately before and
immediately after its hipLaunchKernelGGL, on the stream the kernel is launched on - no Python, no allocation between the two
records.  One untimed frame warms the measuring stream's allocator pool, three frames are measured, every launch keeps its MINIMUM
over the three, and the line carries a self-check (`roofline.valid`): the event-timed kernels of a frame must sum to no more than the
wall time of the same frame's model stage.  `profiles/r4*_frame_kernel_stats.csv` (rocprofv3 --kernel-trace --stats of the same work)
must agree with `avg_launch_us` within a few percent.
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
L2_PEAK_GBS = 34500.0            # MI355X_MICROARCH.md: aggregate L2 bandwidth
PMC_PROFILES = ("r6_ehem_L16m_frame_pmc_traffic.json", "r6_octattn_L14_frame_pmc_traffic.json", "r5_ehem_L16m_frame_pmc_traffic.json", "r5_octattn_L14_frame_pmc_traffic.json",     # profiles/: HBM bytes per launch / per frame from separate rocprofv3 --pmc passes
                "r4_pmc_traffic.json", "r4_pmc_traffic_octattn_L14_cylin.json")                       # (tools/r5_profiles.sh), one file per configuration; the newest that exists is used


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames per rank (default 48; 3 with --decode)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="ehem-L16-m", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full", "none"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=4, help="frames in flight (encode_async handles not yet finished); 4 measured 1.2 % above 3, 5 no better")
    ap.add_argument("--batch", type=int, default=None, help="frames per stage-G / packed-forward / CDF launch (FrameEncoder.encode_batch_async); "
                    "default: 8 for the level-12 EHEM configuration (a 115 k-node frame is 22 windows), 1 elsewhere")
    ap.add_argument("--host-transform", action="store_true", help="(default for the EHEM configurations since round 5) strict-identity mode as the "
                    "HEADLINE: the reference's numpy float32 transform + quantiser on a prefetch thread one to two frames ahead")
    ap.add_argument("--device-transform", action="store_true", help="headline with the device transform (rounds 1 - 4); the strict leg becomes the extra key")
    ap.add_argument("--no-strict-leg", action="store_true", help="skip the second timed loop (the other transform mode)")
    ap.add_argument("--no-legs", action="store_true", help="skip the side legs (the other four workloads, the decoder, the CLI), which run as child processes "
                    "after the headline, outside its timed region")
    ap.add_argument("--leg-steps", type=int, default=16)
    ap.add_argument("--leg-warmup", type=int, default=5)
    ap.add_argument("--oa-batch", type=int, default=None, help="OctAttention: windows per forward (OctAttnFrameEncoder.max_batch)")
    ap.add_argument("--decode", action="store_true", help="time the decoder (FrameDecoder) on the configuration's frame instead of the encoder")
    ap.add_argument("--proc-barrier", default=None, metavar="DIR:N", help="with --decode: this process is one of N decoder processes sharing the GPU; after its warm-up it "
                    "drops a file into DIR and starts its timed loop when N files are there (the parent computes the aggregate rate from the t_begin / t_end each line carries)")
    ap.add_argument("--decode-streams", type=int, default=1, help="with --decode: frames decoded CONCURRENTLY, each by its own FrameDecoder on its own host thread and "
                    "HIP stream (a frame's decode is a chain of dependent launch sequences that uses a fraction of the GPU: independent frames overlap)")
    ap.add_argument("--all-configs", action="store_true", help="run every configuration of CONFIGS in turn (child processes), one JSON line each")
    ap.add_argument("--out-dir", default=None, help="with --all-configs: also write <out-dir>/<tag>_bench_<config>.json")
    ap.add_argument("--tag", default="r4", help="file-name prefix used with --out-dir")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 3 if a.decode else 48      # (four frames are in flight: 24 timed frames left the fill / drain of the pipeline at 2 - 4 % of the figure, 14.7 - 15.2 against 15.3 - 15.4)
    return a


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
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", name, "--no-legs"] + rest, stdout=subprocess.PIPE, text=True)
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


# ------------------------------------------------------------------------------------------------ algorithmic work
def _swin_layer_flops(L):
    Lp = -(-L // 512) * 512
    return Lp * 524288 + L * 1048576 + Lp * 524288          # q|k|v + proj, MLP, attention (QK^T + PV over 512-token windows)


def ehem_window_flops(c, attention_only=False):
    """Algorithmic flop of EHEM.forward on one window of c nodes (ehem.py:88-136): the formula of the survey, on the
    window's real length.  c = 8192 gives 310.2 GFLOP, of which 40.0 are the attention products."""
    c = c + (c & 1)
    att = tot = 0
    L = c
    for s, depth in enumerate((4, 4, 4, 4, 2)):             # self encoder
        Lp = -(-L // 512) * 512
        tot += depth * _swin_layer_flops(L)
        att += depth * Lp * 524288
        if s < 4:
            tot += 2 * (-(-L // 2)) * 512 * 256
            L = -(-L // 2)
    L = c // 2
    for s, depth in enumerate((2, 2, 1, 1)):                # cross encoder on the odd / even halves
        Lp = -(-L // 512) * 512
        tot += depth * _swin_layer_flops(L)
        att += depth * Lp * 524288
        if s < 3:
            tot += 2 * 2 * (-(-L // 2)) * 512 * 256          # both streams are merged
            L = -(-L // 2)
    tot += 2 * c * c * (3 + 144 + 192)                                                                  # three kNN searches
    tot += 2 * 20 * c * (6 * 64 + 288 * 128 + 384 * 256)                                                # edge convolutions
    tot += c * 2 * (80 * 80 + 80 * 64 + 64 * 64 + 64 * 128 + 2 * 128 * 128 + 448 * 256 + 2 * 256 * 256 + 512 * 256 + 256 * 256 + 256 * 128)
    tot += c * 2 * (1280 * 1024 + 1024 * 512 + 512 * 256)                                               # ancient_mlp
    tot += (c // 2) * 2 * (2 * 256 * 256 + 256 * 255 + 256 * 256 + 256 * 240 + 240 * 240 + 3 * 16 * 16 + 1280 * 768 + 768 * 512 + 512 * 255)
    return att if attention_only else tot


def octattn_window_flops(c, attention_only=False):
    """per layer 5 projections, 3 score / AV products per head (non-causal count), the FFN of both streams; 3 layers + head."""
    att = 3 * 4 * (2 * c * c * 150) * 3
    if attention_only:
        return att
    return 3 * (5 * 2 * c * 600 * 600 + 2 * 2 * (2 * c * 600 * 300)) + att + 2 * c * (600 * 600 + 600 * 255)


def frame_flops(cfg, level_sizes, n_nodes):
    if cfg["model"] == "EHEM":
        from scp_amd.encoder import EncodePlan
        ws = EncodePlan(level_sizes, 8192).windows
        return float(sum(ehem_window_flops(w[1]) for w in ws)), float(sum(ehem_window_flops(w[1], True) for w in ws))
    total = n_nodes + 1023
    full, tail = total // 1024, total % 1024
    return (float(full * octattn_window_flops(1024) + (octattn_window_flops(tail) if tail else 0)),
            float(full * octattn_window_flops(1024, True) + (octattn_window_flops(tail, True) if tail else 0)))


# ------------------------------------------------------------------------------------------------ live kernel measurement
KERNEL_OF = {   # launch-bracket tag -> (kernel name as rocprofv3 prints it, bound, note)
    "post_attn": ("rc_post_attn_kernel", "mfma", "attention projection + residual + LayerNorm + fc1 + GELU + fc2 + residual of a Swin block, one launch"),
    "ln_linear": ("rc_ln_linear_kernel", "mfma", "LayerNorm + q|k|v projection in one launch, rows resident as MFMA B fragments; keys / values leave as the attention kernel's bf16 planes"),
    "attention": ("swin_attn_planes_kernel", "mfma", "window attention, K / V tiles staged by LDS-DMA from pre-split planes"),
    "knn_feat": ("knn_f16x3_wg256_kernel", "mfma", "fused distance + top-20 selection on 144 / 192 features, 256-query workgroups on the XCD-affine schedule"),
    "knn_pos": ("knn_mfma_kernel<2,16> (positions)", "valu", "exact fp32 chain on 3 features with tile skipping: selection-bound"),
    "gemm_split": ("gemm_split_kernel", "mfma", "the remaining dense layers (geometry MLPs, concat layers incl. the fused two-stage gemm_hier2_kernel, the 512 / 768-wide head layers; OctAttention: every layer on planes)"),
    "edge_mlp": ("rc_edge_mlp_kernel", "mfma", "both edge MLPs of the geometry generator, six layers chained through the accumulators"),
    "mlp3": ("rc_mlp3_kernel", "mfma", "the 256-wide three-layer heads (prob_pred_mlp1, pre_attn_mlp), one row-chain launch each (round 6)"),
    "merge": ("rc_merge_kernel", "mfma", "patch merging: gather + LayerNorm(512) + reduction"),
    "gemm_f32": ("gemm_f32_kernel", "mfma_f32", "exact k-ordered fp32 layers feeding a kNN search"),
    "gemm_rows": ("gemm_bf16x3_kernel", "mfma", "dense layers reading fp32 rows (split in the tile)"),
    "oa_attention": ("oa_attn_f16x3_kernel", "mfma", "dual-stream causal attention, non-causal flop count (SURVEY.md 8d)"),
    "edge_gather": ("edge_gather_max_kernel", "hbm", "neighbour gather + max + BN + LeakyReLU; algorithmic bytes = every u / v / output row once + the index lists; the 20 gathered "
                    "rows per point are re-reads that fall out of L2: PMC 2.9 x the algorithmic bytes at 6.4 TB/s of HBM (profiles/r4_pmc_traffic.json) - the kernel runs at the achievable "
                    "HBM ceiling on wasted traffic"),
    "cdf": ("cdf_kernel", "hbm", "softmax + serial fp32 cumsum -> (c_low, c_high)"),
    "geom": ("stage_G_front_and_context_kernels", "hbm", "front_transform_kernel, front_key_kernel, ctx_ehem_all_kernel (the sort and tree kernels between them are not bracketed)"),
    "split_rows": ("split_rows_kernel", "hbm", "fp32 rows -> hi / lo planes"),
    "layernorm": ("layernorm_*_kernel", "hbm", "LayerNorm passes left outside the row-chain kernels"),
    "other": ("(operand preparation)", "hbm", "oa_prep / oa_absmax"),
}


def measure_kernels(run, frames=3):
    """run() encodes ONE frame synchronously on the current stream.  -> [(tag, ms, work)] in launch order, ms = the launch's minimum over
    `frames` measured runs (after one untimed run that warms this stream's allocator pool and every lazily built weight cache)."""
    from scp_amd import native
    run()
    torch.cuda.synchronize()
    recs = []
    for _ in range(frames):
        with native.launch_profile() as p:
            run()
            torch.cuda.synchronize()
        recs.append(p.records())
    tags = [r[0] for r in recs[0]]
    same = all([r[0] for r in rr] == tags for rr in recs)
    if not same or any(r[1] < 0 for rr in recs for r in rr):
        raise RuntimeError("launch brackets: the measured frames did not issue the same launches")
    ms = np.min(np.asarray([[r[1] for r in rr] for rr in recs], np.float64), 0)
    return [(t, float(m), r[2]) for t, m, r in zip(tags, ms, recs[0])]


def knn_pairs(enc, level_sizes):
    """Pair count of every packed kNN launch of a frame, in launch order (one per chunk of <= max_tokens tokens): the sum over 512-row
    chunks of n x 512 for the window they belong to (n = the window's even-padded length)."""
    from scp_amd.encoder import EncodePlan, chunk_windows
    ws = EncodePlan(level_sizes, enc.context_size).windows
    out = []
    for i, j in chunk_windows(ws, enc.max_tokens):
        pairs = 0
        for w in ws[i:j]:
            n = w[1] + (w[1] & 1)
            pairs += (-(-n // 512)) * 512 * n
        out.append(float(pairs))
    return out


def summarise(records, pairs=None):
    """per tag: launches, total ms, algorithmic work, average launch, achieved rate"""
    out = {}
    kn = {"knn_feat": 0, "knn_pos": 0}
    for tag, ms, work in records:
        if tag in kn and pairs is not None:
            per_chunk = 2 if tag == "knn_feat" else 1              # launches of this tag per chunk (144 + 192 features | positions)
            work = 2.0 * work * pairs[min(kn[tag] // per_chunk, len(pairs) - 1)]
            kn[tag] += 1
        d = out.setdefault(tag, dict(launches=0, total_ms=0.0, work=0.0))
        d["launches"] += 1
        d["total_ms"] += ms
        d["work"] += work
    for tag, d in out.items():
        d["avg_launch_us"] = 1e3 * d["total_ms"] / d["launches"]
        d["work_per_launch"] = d["work"] / d["launches"]
        d["rate"] = d["work"] / (d["total_ms"] * 1e-3) if d["total_ms"] > 0 else 0.0       # flop/s or B/s
    return out


def roofline_entry(tag, d):
    name, bound, note = KERNEL_OF.get(tag, (tag, "hbm", ""))
    peak3 = BF16_MFMA_PEAK_TFLOPS / 3.0
    e = dict(kernel=name, bound="mfma" if bound.startswith("mfma") else bound, launches_per_frame=d["launches"], avg_launch_us=d["avg_launch_us"],
             total_ms_per_frame=d["total_ms"], note=note)
    if bound == "mfma":
        e.update(achieved=d["rate"] / 1e12, peak=peak3, unit="TFLOP/s", frac=d["rate"] / 1e12 / peak3, flops_per_launch=d["work_per_launch"])
    elif bound == "mfma_f32":
        e.update(achieved=d["rate"] / 1e12, peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=d["rate"] / 1e12 / F32_MFMA_PEAK_TFLOPS,
                 flops_per_launch=d["work_per_launch"])
    elif bound == "hbm":
        e.update(achieved=d["rate"] / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=d["rate"] / 1e9 / HBM_PEAK_GBS, bytes_per_launch=d["work_per_launch"])
    elif bound == "l2":
        e.update(achieved=d["rate"] / 1e9, peak=L2_PEAK_GBS, unit="GB/s", frac=d["rate"] / 1e9 / L2_PEAK_GBS, bytes_per_launch=d["work_per_launch"])
    return e


def pmc_traffic(config):
    """(HBM bytes per launch of the dominant kernel, HBM bytes per frame, source) from the committed PMC passes - only for the
    configuration they were collected on."""
    for name in PMC_PROFILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                z = json.load(f)
        except Exception:
            continue
        if z.get("config") == config:
            return (z.get("dominant_hbm_bytes_per_launch"), z.get("frame_hbm_bytes"),
                    f"profiles/{name} (separate rocprofv3 --pmc passes over this configuration's frame, {z.get('dominant_kernel')}; not measured by this run)")
    return None, None, None


def last_full_frame_cpu(config):
    """The last recorded whole-frame run of the CPU port for this configuration (`--cpu-baseline full`, every window run), from profiles/."""
    for name in ("r6_bench_cpu_full.json", "r5_bench_cpu_full.json", "r3e_bench_cpu_full.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                z = json.loads(f.read().strip().splitlines()[-1])
        except Exception:
            continue
        cb = z.get("cpu_baseline") or {}
        if z.get("config", {}).get("workload", "").startswith(CONFIGS[config]["workload"]) and cb.get("seconds_per_frame") and "SAMPLED" not in cb.get("sample", ""):
            return dict(seconds_per_frame=cb["seconds_per_frame"], value=cb["value"], cores=cb.get("cores"), host_cpu=cb.get("host_cpu"), source="profiles/" + name)
    return None


def cpu_baseline(cfg, xyz, full=False, config=None):
    """The CPU oracle (a port of the reference path: C octree / records / CDF / range coder + the functional PyTorch-CPU model,
    oracle/cpu_encode.py) on ONE frame of the bench's own workload, every stage timed.  Bounded sample: all windows shorter than
    the model's context are run; full windows (identical shapes, data-independent cost) are run ten times - four warm-ups (the
    first windows of a run are 20 - 30 % slower than the settled ones and the times keep falling for a dozen windows: thread pool, allocator,
    caches, clocks), the FASTEST of the other six stands for the rest (the choice that favours the CPU; measured within 5 % of a run of every
    window on the same box).  `--cpu-baseline full` runs every window (minutes per frame); the last recorded such run is printed beside
    the sample (`full_frame_recorded`)."""
    from cfgs import ehem_cfg, octattn_cfg
    from oracle import cpu_encode
    from scp_amd.models import EHEM, OctAttention
    from scp_amd.weights import fill_weights
    threads = min(os.cpu_count() or 1, 32)   # PyTorch-CPU oversubscribes badly beyond this on 128-core hosts
    torch.set_num_threads(threads)
    runs, warm = (None, 0) if full else (10, 4)
    if cfg["model"] == "EHEM":
        sd = fill_weights(EHEM(ehem_cfg()), 0).state_dict()
        r = cpu_encode.encode_frame(xyz, sd, cfg["level"], mullevel=cfg["mullevel"], mode=cfg["mode"], full_window_runs=runs,
                                    data_type=cfg.get("type", "kitti"), full_window_warmup=warm)
    else:
        sd = fill_weights(OctAttention(octattn_cfg()), 0).state_dict()
        r = cpu_encode.encode_frame_octattn(xyz, sd, cfg["level"], mode=cfg["mode"], full_window_runs=runs, full_window_warmup=warm)
    what = (f"one whole frame, all {r['windows']} windows run" if full else
            f"SAMPLED: one frame: quantiser/octree/records/context of the whole frame, all {r['partial_windows']} partial windows, "
            f"{r['full_windows_run']} of the {r['full_windows']} full windows (first {warm} = warm-up, the fastest of the rest x {r['full_windows']}), "
            f"CDF + range coder on the {r['rows_coded']} rows produced, scaled to {r['n_nodes']} nodes")
    out = dict(value=1.0 / r["total_s"], unit="frames/s", cores=threads, kind="port", sample=what, seconds_per_frame=r["total_s"],
               stage_s=r["stage_s"], full_window_s=r["full_window_s"], full_window_cost_s=r.get("full_window_cost_s"),
               host_cpu=cpu_encode.cpu_model_name(), host_cores=os.cpu_count())
    rec = None if (full or config is None) else last_full_frame_cpu(config)
    if rec is not None:
        out["full_frame_recorded"] = rec
        out["sampled_over_full_frame_recorded"] = r["total_s"] / rec["seconds_per_frame"]
    return out


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


class IntsAhead:
    """Strict-identity front end of the bench loop: enc.host_ints (the reference's numpy float32 transform + quantiser,
    data_preprocess.py:42-70,171-214) of frame i + 1 ... i + ahead on ONE worker thread while frame i is enqueued - what cli.py does with
    `Prefetch(post=enc.host_ints)`.  All of it inside the timed region."""

    def __init__(self, enc, frames_host, lo, hi, ahead=2, workers=1):
        from concurrent.futures import ThreadPoolExecutor
        self.enc, self.frames, self.hi, self.ahead = enc, frames_host, hi, ahead
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.futs, self.next = {}, lo

    def get(self, i):
        while self.next < self.hi and self.next <= i + self.ahead:
            self.futs[self.next] = self.pool.submit(self.enc.host_ints, self.frames[self.next])
            self.next += 1
        return self.futs.pop(i).result()

    def close(self):
        self.pool.shutdown(wait=True)


def differing_points(enc, cfg, dev):
    """Seed-0 frame: points whose quantised integers differ from the REFERENCE's (tests/golden/frame_ints.npz, produced by running the
    reference's proc_pc here): host transform (must be 0) and device transform, per shell."""
    ford = cfg.get("type") == "ford"
    name = {"spher": "q_spher_L", "cylin": "q_cylin_L", "cart": "q_cart_L"}[cfg["mode"]]
    if ford:       # tests/golden/make_golden.py facts_ford: the reference's mul_proc_pc on ford_like(synth_frame(0))
        name = name[:-1] + "ford_L"
    path = os.path.join(ROOT, "tests", "golden", "frame_ints.npz")
    if not os.path.exists(path):
        return None
    z = np.load(path)
    from scp_amd.synth import ford_like, synth_frame
    xyz = ford_like(synth_frame(0)) if ford else synth_frame(0)
    keys = [f"{name}{lv}" for _, lv in enc.shells()]
    if any(k not in z.files for k in keys):
        return None
    host = None
    if hasattr(enc, "host_ints"):
        hq, _ = enc.host_ints(xyz)
        host = [int((np.asarray(q) != z[k]).any(1).sum()) for q, k in zip(hq, keys)]
    dq = enc.quantize(torch.from_numpy(xyz).to(dev))[0] if not enc.host_transform else None
    devc = None if dq is None else [int((q.cpu().numpy() != z[k]).any(1).sum()) for q, k in zip(dq, keys)]
    return dict(host_transform=host, device_transform=devc, reference="tests/golden/frame_ints.npz (the reference's proc_pc run on this frame)")


def run_decode(args, cfg, enc, model, dev, frame_host):
    """`--decode`: FrameDecoder on the stream of the configuration's frame (decode_ehem_mullevel.py:56-189)."""
    from scp_amd.decoder import FrameDecoder
    res = enc.encode(frame_host)
    nodes = enc.geom.nodes(("occ",))["occ"]
    want = [nodes[i.node_base:i.node_base + i.n_nodes].cpu().numpy() for i in enc.geom.info]
    dec = FrameDecoder(model, cfg["level"], mullevel=cfg["mullevel"], polar=cfg["mode"] != "cart", device=dev)
    ok = True
    for i in range(args.warmup):
        shells = dec.decode(res["bytes"], res["n_levels"], res["pos_mm"])
    # one more decode with a device synchronisation per stage: where the time goes (these stamps slow the decode down; the timed loop below
    # runs without them)
    dec.stats = {}
    dec.decode(res["bytes"], res["n_levels"], res["pos_mm"])
    stage = {k: round(1e3 * v, 2) for k, v in dec.stats.items()}
    dec.stats = None
    torch.cuda.synchronize()
    nstreams = min(8, max(1, args.decode_streams))           # (16 decoders did not fit next to each other: every decoder holds its own frame's tables)
    if nstreams > 1:
        # `nstreams` decoders, each on its own thread and stream, `steps` frames each (every weight cache is warm: decoder 0 has run)
        import threading
        decs = [dec] + [FrameDecoder(model, cfg["level"], mullevel=cfg["mullevel"], polar=cfg["mode"] != "cart", device=dev) for _ in range(nstreams - 1)]
        outs = [None] * nstreams

        def work(k, n):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                for _ in range(n):
                    outs[k] = decs[k].decode(res["bytes"], res["n_levels"], res["pos_mm"])
                torch.cuda.current_stream().synchronize()

        ths = [threading.Thread(target=work, args=(k, 1)) for k in range(nstreams)]     # warm-up: every decoder's plans
        [t.start() for t in ths]; [t.join() for t in ths]
        torch.cuda.synchronize()
        cpu0, t0 = time.process_time(), time.perf_counter()
        ths = [threading.Thread(target=work, args=(k, args.steps)) for k in range(nstreams)]
        [t.start() for t in ths]; [t.join() for t in ths]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nstreams          # per frame: nstreams x steps frames were decoded
        cpu_ms = 1e3 * (time.process_time() - cpu0) / (args.steps * nstreams)
        shells = outs[-1]
    else:
        if args.proc_barrier:
            bdir, nproc = args.proc_barrier.rsplit(":", 1)
            open(os.path.join(bdir, f"ready_{os.getpid()}"), "w").close()
            t_wait = time.perf_counter()
            while len([f for f in os.listdir(bdir) if f.startswith("ready_")]) < int(nproc):
                if time.perf_counter() - t_wait > 600:
                    raise SystemExit("--proc-barrier: the other decoder processes never arrived")
                time.sleep(0.002)
        t_begin = time.time()
        cpu0, t0 = time.process_time(), time.perf_counter()
        for i in range(args.steps):
            shells = dec.decode(res["bytes"], res["n_levels"], res["pos_mm"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t_end = time.time()
        cpu_ms = 1e3 * (time.process_time() - cpu0) / args.steps
    for (codes, _), w in zip(shells, want):
        got = torch.cat(codes).cpu().numpy()
        ok = ok and len(got) == len(w) and (np.array_equal(got[:-1], w[:-1]) if cfg["mullevel"] else np.array_equal(got, w))
    from scp_amd.encoder import EncodePlan
    ws = EncodePlan(res["level_sizes"], 8192).windows
    with_phase2 = sum(1 for w in ws if w[1] > 1)
    stats = None
    out = {"metric": f"KITTI frames/sec decode (SCP-EHEM, level {cfg['level']}{' multi-level' if cfg['mullevel'] else ''})", "value": args.steps / dt,
           "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32 (same kernels and numeric profile as the encoder)", "data": "synthetic",
           "config": {"workload": cfg["workload"] + ", seeded random weights; DECODER", "nodes_per_frame": int(res["n_nodes"]), "windows_per_frame": len(ws),
                      "levels": len(res["level_sizes"]), "phase1_launch_sequences_per_frame": len(res["level_sizes"]),
                      "phase2_launch_sequences_per_frame": with_phase2, "frames_decoded_concurrently": nstreams},
           "decoded_occupancy_equals_encoded": bool(ok), "host_cpu_ms_per_frame": cpu_ms, "stream_bytes": len(res["bytes"])}
    if nstreams == 1:
        out["t_begin"], out["t_end"] = t_begin, t_end
    out["stage_ms"] = stage
    out["stage_ms_note"] = "one extra decode of the same stream with a device synchronisation after every stage (slower than the timed decodes)"
    print(json.dumps(out), flush=True)


def _child_line(argv, timeout=900):
    """Run this file as a CHILD process (the parent keeps its GPU context; nothing is exec'ed over it) and return its last JSON line."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
    if line is None:
        return {"error": f"exit code {r.returncode}: {r.stderr[-300:]}"}
    return json.loads(line)


def decode_procs_leg(nproc, steps):
    """`nproc` decoder PROCESSES side by side on this GPU (a decoding server's shape: a decode is a serial chain of short launches that keeps one host
    thread busy and fills a fraction of the GPU, so independent streams are decoded by independent processes - threads of one process share the interpreter
    lock).  Every child is `bench.py --decode`; their timed loops start together (file barrier); aggregate = all frames / (last end - first begin)."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, os.path.abspath(__file__), "--decode", "--steps", str(steps), "--warmup", "1", "--proc-barrier", f"{tmp}:{nproc}"]
        ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(nproc)]
        zs = []
        for q in ps:
            try:
                so, se = q.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                q.kill()
                so, se = q.communicate()
            line = next((l for l in reversed(so.splitlines()) if l.startswith("{")), None)
            if line is None:
                return {"error": f"exit code {q.returncode}: {se[-300:]}"}
            zs.append(json.loads(line))
    span = max(z["t_end"] for z in zs) - min(z["t_begin"] for z in zs)
    return dict(fps=nproc * steps / span, processes=nproc, frames=nproc * steps, per_process_fps=[z["value"] for z in zs],
                start_skew_ms=1e3 * (max(z["t_begin"] for z in zs) - min(z["t_begin"] for z in zs)),
                decoded_occupancy_equals_encoded=all(z["decoded_occupancy_equals_encoded"] for z in zs),
                host_cpu_ms_per_frame=max(z["host_cpu_ms_per_frame"] for z in zs),
                note="independent decoder processes sharing ONE GPU (each: its own host thread, HIP queues, model replica), timed loops started together; "
                     "fps = all frames / (last end - first begin)")


def cli_leg(n_warm, n_timed):
    """`.bin` files on disk -> `encode_mullevel.py --spher --lidar_level 16 --host_transform` -> `.bin` / `.dat` / side-info files written: the
    reader (a1), the host -> device copy and the file writes are inside; frames/s from the per-frame `time(s)` lines (completion intervals of
    the pipelined path) after the first `n_warm` frames."""
    import subprocess
    import tempfile
    from scp_amd.synth import synth_frame, write_kitti_bin
    tail = 4                         # the last frames complete while the pipeline drains (nothing new competes for the GPU): not counted
    n = n_warm + n_timed + tail
    with tempfile.TemporaryDirectory() as tmp:
        seq = os.path.join(tmp, "seq00")
        os.makedirs(seq)
        for i in range(n):
            write_kitti_bin(os.path.join(seq, f"{i:06d}.bin"), synth_frame(i))
        cmd = [sys.executable, os.path.join(ROOT, "encode_mullevel.py"), "--test_files", os.path.join(seq, "*.bin"), "--type", "kitti", "--lidar_level", "16",
               "--spher", "--random_weights", "0", "--out_dir", os.path.join(tmp, "out"), "--host_transform"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, timeout=900)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": r.stderr[-300:]}
        times = [float(l.split(":")[1]) for l in r.stdout.splitlines() if l.startswith("time(s)")]
        written = len([f for f in os.listdir(os.path.join(tmp, "out")) if f.endswith(".bin")])
    steady = times[n_warm:n_warm + n_timed]
    fps = len(steady) / sum(steady)
    return dict(fps=fps, ms_per_step=1e3 / fps, files=n, files_timed=len(steady), streams_written=written, process_wall_s=wall,
                note="drop-in CLI (encode_mullevel.py -> scp_amd/cli.py), strict-identity transform on the reader thread, frames pipelined three deep; "
                     "file read + parse + H2D + .bin / .dat / .scp.json writes inside; process start, weight preparation, the first frames and the last four "
                     "(pipeline drain) excluded")


def side_legs(args, out):
    """The other four BASELINE.json workloads, the decoder and the CLI, after the headline and outside its timed region (N = 1 only)."""
    torch.cuda.empty_cache()
    common = ["--steps", str(args.leg_steps), "--warmup", str(args.leg_warmup), "--cpu-baseline", "none", "--no-legs", "--no-strict-leg"]

    def brief(z):
        if "error" in z:
            return z
        rf = z.get("roofline", {})
        return dict(fps=z["value"], ms_per_step=z["ms_per_step"], steps=z["steps"], warmup=z["warmup"], bpp_mean=z.get("bpp_mean"),
                    workload=z["config"]["workload"], transform=z["config"].get("transform"), frames_per_launch_sequence=z["config"].get("frames_per_launch_sequence"),
                    nodes_per_frame=z["config"].get("nodes_per_frame"), strict_identity_verified=z.get("strict_identity_verified"),
                    transform_parity=z.get("transform_parity"),
                    roofline=dict(kernel=rf.get("kernel", "").split(":")[0], frac=rf.get("frac"), valid=rf.get("valid"), avg_launch_us=rf.get("avg_launch_us")),
                    roofline_frame_frac_mfma=z.get("roofline_frame", {}).get("frac_mfma"))
    rf = out["roofline"]
    configs = {"ehem-L16-m": dict(fps=out["value"], ms_per_step=out["ms_per_step"], steps=out["steps"], warmup=out["warmup"], bpp_mean=out["bpp_mean"],
                                  workload=out["config"]["workload"], transform=out["config"]["transform"], strict_identity_verified=out.get("strict_identity_verified"),
                                  roofline=dict(kernel=rf["kernel"].split(":")[0], frac=rf["frac"], valid=rf["valid"], avg_launch_us=rf["avg_launch_us"]),
                                  roofline_frame_frac_mfma=out["roofline_frame"]["frac_mfma"], note="the headline of this line")}
    for name in ("ehem-L12-s", "ehem-F17-m", "octattn-L12-spher", "octattn-L14-cylin"):
        extra = ["--steps", "32"] if name == "ehem-L12-s" else []      # BASELINE.json configs[1] is a batch of 16 frames: two of them (four launch sequences of 8)
        configs[name] = brief(_child_line(["--config", name] + common + extra))
    out["configs"] = configs
    z = _child_line(["--decode", "--steps", "3", "--warmup", "1"])
    out["decode"] = z if "error" in z else dict(fps=z["value"], ms_per_step=z["ms_per_step"], steps=z["steps"], decoded_occupancy_equals_encoded=z["decoded_occupancy_equals_encoded"],
                                               stage_ms=z.get("stage_ms"), host_cpu_ms_per_frame=z.get("host_cpu_ms_per_frame"), workload=z["config"]["workload"])
    out["decode_2_procs"] = decode_procs_leg(2, 4)
    out["decode_4_procs"] = decode_procs_leg(4, 4)
    out["cli"] = cli_leg(args.leg_warmup, 2 * args.leg_steps)            # (files are cheap: a longer steady-state window than the bench legs')
    if "fps" in out["cli"]:
        out["cli_over_bench"] = out["cli"]["fps"] / out["value"]
    out["legs_note"] = ("`configs` / `decode` / `cli`: child processes of this run, started after the headline's timed region; each is this file's own "
                        "timed loop (barrier + synchronize on both sides) on that workload")


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
    # SCP_DIST_FORCE=1: run every distributed step of this file (process group, barriers, the all-reduces, the per-rank gather) in a world of
    # ONE too - what a one-GPU box can exercise of the RCCL path (tests/test_gpu_dist.py)
    dist_on = world > 1 or os.environ.get("SCP_DIST_FORCE", "0") == "1"
    if dist_on:
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
                           mullevel=cfg["mullevel"], device=dev, host_transform=False)
    else:
        model = fill_weights(OctAttention(octattn_cfg()), 0).to(dev)
        enc = OctAttnFrameEncoder(model, "kitti", cfg["level"], spher=cfg["mode"] == "spher", cylin=cfg["mode"] == "cylin", device=dev,
                                  host_transform=True if args.host_transform else None, **({"max_batch": args.oa_batch} if args.oa_batch else {}))

    total = args.warmup + args.steps
    frames_host = [synth_frame(rank * 1000 + i) for i in range(total)]
    from scp_amd.synth import ford_like
    if cfg.get("type") == "ford":
        frames_host = [ford_like(f) for f in frames_host]
    if args.decode:
        if not ehem or world > 1:
            raise SystemExit("--decode: the EHEM configurations, one GPU")
        return run_decode(args, cfg, enc, model, dev, frames_host[0])
    frames = [torch.from_numpy(f).to(dev) for f in frames_host]      # resident in HBM before the timed region
    torch.cuda.synchronize()

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    batch = args.batch if args.batch is not None else (8 if args.config == "ehem-L12-s" else 1)     # (4 / 8 / 16 frames per launch sequence: 69.8 / 71.3 / 68.8 frames/s strict, round 5)
    if batch > 1 and not ehem:
        raise SystemExit("--batch is an EHEM option")

    def timed_loop(strict):
        """`--steps` frames through the pipelined encoder; strict: the reference's host transform one frame ahead on a worker thread.
        -> (dt of this rank before the barrier, dt after it, host CPU ms per frame, results)"""
        ahead = IntsAhead(enc, frames_host, args.warmup, total, ahead=max(2, 2 * batch), workers=2 if batch > 1 else 1) if strict else None
        barrier()
        cpu0 = time.process_time()                      # CPU seconds of this rank, all threads (launch thread, coder worker, reader)
        t0 = time.perf_counter()
        # frame i is range-coded on a worker thread while frames i+1 .. i+depth run on the GPU; a handle pins its ~590 MB logits table,
        # so at most `depth` frames are in flight (memory stays O(depth), not O(steps))
        pending, results = [], []
        if batch > 1:      # `batch` frames per launch sequence, two batches in flight (the range coder of one under the kernels of the next)
            for i in range(args.warmup, total, batch):
                j = min(total, i + batch)
                pending.append(enc.encode_batch_async(frames[i:j], ints=[ahead.get(k) for k in range(i, j)] if strict else None))
                if len(pending) > 1:
                    results += enc.finish_batch(pending.pop(0))
            for h in pending:
                results += enc.finish_batch(h)
        else:
            # the front part (stage G + plans) of frame i + 1 runs on the encoder's front thread while this thread enqueues the model part of
            # frame i (FrameEncoder.front_async: the front part BLOCKS its host thread for 33 - 43 ms per frame - its small kernels queue
            # behind the whole-GPU model kernels -, which with the strict transform in the same thread made that leg launch-bound)
            ahead_front = hasattr(enc, "front_async")
            start = lambda j: enc.front_async(frames[j], ints=ahead.get(j) if strict else None)
            nxt = start(args.warmup) if ahead_front and args.warmup < total else None
            for i in range(args.warmup, total):
                if ahead_front:
                    cur, nxt = nxt, (start(i + 1) if i + 1 < total else None)
                    pending.append(enc.encode_async(frames[i], front=cur))
                else:
                    pending.append(enc.encode_async(frames[i]))
                if len(pending) > args.depth:
                    results.append(enc.finish(pending.pop(0)))
            results += [enc.finish(h) for h in pending]
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0               # this rank's own time for its frames (before the barrier)
        cpu_ms = 1e3 * (time.process_time() - cpu0) / args.steps
        barrier()
        dt = time.perf_counter() - t0
        if ahead:
            ahead.close()
        return dt_own, dt, cpu_ms, results

    for i in range(args.warmup):
        enc.finish(enc.encode_async(frames[i]))
    if batch > 1:
        enc.finish_batch(enc.encode_batch_async(frames[:batch]))
    # Headline mode (round 5): STRICT IDENTITY for the EHEM configurations - the float -> integer step is the reference's own numpy arithmetic
    # (enc.host_ints on a prefetch thread, inside the timed region), so the occupancy stream is the reference's bit for bit from the float
    # frame on; `--device-transform` gives rounds 1 - 4's headline.  The other mode runs as a second timed loop and is reported beside it.
    headline_strict = ehem and not args.device_transform
    dt_own, dt, cpu_ms, results = timed_loop(headline_strict)
    other = None
    if ehem and world == 1 and not args.no_strict_leg:
        s_own, s_dt, s_cpu, s_res = timed_loop(not headline_strict)
        other = dict(fps=args.steps / s_dt, ms_per_step=1e3 * s_dt / args.steps, host_cpu_ms_per_frame=s_cpu,
                     ratio_to_headline=(args.steps / s_dt) / (args.steps / dt), bpp_mean=float(np.mean([r["bpp"] for r in s_res])),
                     note=("the same frames with the transform + quantiser on the device (front_transform_kernel: float64 atan2 / acos rounded once - more accurate "
                           "than numpy's float32 routines, hence not always the reference's integers: see transform_parity)") if headline_strict else
                          ("enc.host_ints (numpy float32 transform + quantiser of data_preprocess.py:42-70,171-214) on one worker thread one to two "
                           "frames ahead, inside the timed region; everything after the integers on the device as in the headline"))
    rank_stats = None
    if dist_on:
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
    if dist_on:
        dist.all_reduce(summ, op=dist.ReduceOp.SUM)
    summ = summ.cpu().numpy()

    if rank == 0:
        n_nodes = results[-1]["n_nodes"]
        P = results[-1]["n_points"]
        peak3 = BF16_MFMA_PEAK_TFLOPS / 3.0
        bytes_G = 12 * P + 25 * n_nodes            # SURVEY.md §8d algorithmic bytes of stage G
        bytes_C = n_nodes * (255 * 4 + 4)
        level_sizes = results[-1]["level_sizes"]
        # ---- live kernel measurement on the caller's stream (single stream: a bracket never spans another lane's kernel)
        recs = measure_kernels(lambda: enc.encode(frames[-1]))
        per_tag = summarise(recs, knn_pairs(enc, level_sizes) if ehem else None)
        # per-stage wall times of the same frame with a device sync after every stage: the faster of two runs
        if ehem:
            st = min((enc.encode(frames[-1], timing=True)["times"] for _ in range(2)), key=lambda t: t["total"])
        else:
            torch.cuda.synchronize()
            t0 = time.perf_counter(); enc.encode(frames[-1]); torch.cuda.synchronize()
            st = {"total": time.perf_counter() - t0}
        model_tags = [t for t in per_tag if t not in ("cdf", "geom")]
        kernel_ms = sum(per_tag[t]["total_ms"] for t in model_tags)
        ref_ms = 1e3 * (st["model"] if ehem else st["total"])
        valid = kernel_ms <= 1.02 * ref_ms
        dom_tag = max((t for t in per_tag if KERNEL_OF.get(t, ("", ""))[1] == "mfma"), key=lambda t: per_tag[t]["total_ms"])
        dom = roofline_entry(dom_tag, per_tag[dom_tag])
        traffic, frame_bytes, traffic_src = pmc_traffic(args.config)
        fl_total, fl_att = frame_flops(cfg, level_sizes, n_nodes)
        ms_step = 1e3 * dt / args.steps
        if ehem:
            metric = "KITTI frames/sec encode (SCP-EHEM, level 16) + bpp match vs ref"
            if args.config == "ehem-F17-m":
                metric = "Ford-like frames/sec encode (SCP-EHEM, level 17 multi-level) + bpp match vs ref"
            elif args.config != "ehem-L16-m":
                metric = f"KITTI frames/sec encode (SCP-EHEM, level {cfg['level']} same-level) + bpp match vs ref"
            dtype = ("f32 (dense layers and attention as bf16x3 split on bf16 MFMA, feature kNN as f16x3 split on f16 MFMA, fp32 accumulate; "
                     "position kNN / CDF in fp32)")
            windows = len(EncodePlan(level_sizes, 8192).windows)
        else:
            metric = f"KITTI frames/sec encode (SCP-OctAttention, level {cfg['level']} --{cfg['mode']})"
            dtype = "f32 (dense layers and attention as f16x3 split on f16 MFMA with power-of-two row scales, fp32 accumulate; CDF in fp32)"
            windows = -(-(n_nodes + 1023) // 1024)
        out = {
            "metric": metric,
            "value": world * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": cfg["workload"] + ", seeded random weights", "nodes_per_frame": int(n_nodes), "windows_per_frame": windows,
                       "frames_per_gpu": args.steps, "frames_in_flight": args.depth if batch == 1 else 2 * batch, "frames_per_launch_sequence": batch,
                       "transform": "host-numpy on a prefetch thread (strict identity: the reference's integers)" if (headline_strict or enc.host_transform) else "device",
                       "parallelism": f"frame-sharded x{world}", "rank_cores": len(pinned) if pinned else len(os.sched_getaffinity(0)),
                       "rank_cores_pinned": bool(pinned)},
            "rccl_world": dist.get_world_size() if dist_on else 1, "dist_backend": backend if dist_on else None,
            "host_cpu_ms_per_frame": cpu_ms,
            "bpp_mean": float(summ[0] / summ[4]),
            "stage_ms": {k: round(1e3 * v, 3) for k, v in st.items()},
            # dominant kernel = the MFMA-bound kernel with the most time per frame.  `achieved` counts ALGORITHMIC flops (2*M*N*K of the fp32
            # product it replaces); a split kernel spends three 16-bit MFMAs per product, so its own ceiling is a third of the dense 16-bit peak.
            "roofline": {"bound": "mfma", "kernel": dom["kernel"] + ": " + dom["note"] + " (3x v_mfma_f32_32x32x16 per fp32-class product)",
                         "achieved": dom["achieved"], "peak": peak3, "unit": "TFLOP/s", "frac": dom["frac"],
                         "traffic": traffic, "traffic_source": traffic_src,
                         "valid": bool(valid), "kernel_ms_sum": kernel_ms, "kernel_ms_bound": ref_ms,
                         "validity_rule": "event-timed kernels of one frame (per-launch minimum over 3 frames) must sum to <= 1.02 x the synchronised wall "
                                          "time of the same frame's model stage; false = do not use these fractions",
                         "method": "hipEvents recorded inside libscp_hip.so around each launch (include/scp_debug.h), allocator-warm stream, per-launch minimum of 3 frames",
                         "peak_note": "2500 TFLOP/s dense 16-bit MFMA / 3 products; the fp32 MFMA peak this replaces is 157.3.  With all 256 CUs multiplying the clock settles at 1.85 GHz (tools/src/mb_power.cpp): 1.95 PFLOP/s sustained, 650 per fp32-class product; this kernel runs at 1.91 GHz inside the frame (GRBM_GUI_ACTIVE / duration, profiles/r4k_kernel_clocks.md)",
                         "launches_per_frame": dom["launches_per_frame"], "avg_launch_us": dom["avg_launch_us"], "flops_per_launch": dom["flops_per_launch"]},
            # the whole frame against both roofs: algorithmic flop of what the reference's window loop computes (encode_mullevel.py:106-133;
            # SURVEY.md 8d formulas on the frame's real window lengths) / the bench's own ms_per_step
            "roofline_frame": {"flops": fl_total, "attention_flops": fl_att, "achieved_tflops": fl_total / (ms_step * 1e-3) / 1e12 / world,
                               "peak_tflops": peak3, "frac_mfma": fl_total / (ms_step * 1e-3) / 1e12 / world / peak3,
                               "hbm_bytes": frame_bytes, "achieved_gbs": None if frame_bytes is None else frame_bytes / (ms_step * 1e-3) / 1e9 / world,
                               "frac_hbm": None if frame_bytes is None else frame_bytes / (ms_step * 1e-3) / 1e9 / world / HBM_PEAK_GBS,
                               "note": "per GPU; flops priced against 2500 / 3 TFLOP/s as if every product were a three-MFMA split; hbm_bytes from " + (traffic_src or "no PMC pass for this configuration")},
            "roofline_kernels": {KERNEL_OF.get(t, (t,))[0]: roofline_entry(t, per_tag[t]) for t in sorted(per_tag, key=lambda t: -per_tag[t]["total_ms"])},
            "launches_bracketed_per_frame": len(recs),
        }
        if ehem:
            out["roofline_stages"] = {
                "G": {"bound": "hbm", "achieved": bytes_G / st["geom"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": bytes_G / st["geom"] / 1e9 / HBM_PEAK_GBS, "bytes": bytes_G, "note": "host wall time of the whole stage incl. its small D2H syncs"},
                "C": {"bound": "hbm", "achieved": bytes_C / st["cdf"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": bytes_C / st["cdf"] / 1e9 / HBM_PEAK_GBS, "bytes": bytes_C, "note": "includes the 4 B/node D2H copy"}}
            if other is not None:
                out["device_transform" if headline_strict else "strict_identity"] = other
                out["strict_identity_fps"] = out["value"] if headline_strict else other["fps"]
                out["device_transform_fps"] = other["fps"] if headline_strict else out["value"]
            try:
                out["transform_parity"] = differing_points(enc, cfg, dev)
            except Exception as e:
                out["transform_parity"] = {"error": str(e)}
            tp = out["transform_parity"]
            if headline_strict and tp is not None and "host_transform" in tp:
                # asserted in-run: the headline's integers ARE the reference's (tests/golden/frame_ints.npz, written by running the reference's proc_pc)
                out["strict_identity_verified"] = all(v == 0 for v in tp["host_transform"])
                if not out["strict_identity_verified"]:
                    raise SystemExit(f"strict-identity headline, but the host transform's integers differ from the reference's: {tp}")
            elif headline_strict:
                out["strict_identity_verified"] = None      # no reference integers for this workload's frames in tests/golden/
        if not ehem:
            try:
                out["transform_parity"] = differing_points(enc, cfg, dev)     # OctAttention legs run the device transform: how many points it misses
            except Exception as e:
                out["transform_parity"] = {"error": str(e)}
        if rank_stats:
            fps = [r["fps"] for r in rank_stats]
            cpu = [r["host_cpu_ms_per_frame"] for r in rank_stats]
            out["ranks"] = {"fps_min": min(fps), "fps_max": max(fps), "host_cpu_ms_per_frame_min": min(cpu), "host_cpu_ms_per_frame_max": max(cpu),
                            "shared_frame_streams_identical": len({r["shared_frame_sha256"] for r in rank_stats}) == 1, "per_rank": rank_stats}
        mode = "none" if args.no_cpu_baseline else args.cpu_baseline
        if world == 1 and mode != "none":
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, frames_host[-1], full=mode == "full", config=args.config)
            except Exception as e:   # the baseline is a reported number, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        if world == 1 and not args.no_legs and args.config == "ehem-L16-m":
            try:
                side_legs(args, out)
            except Exception as e:     # a leg is extra information, never a reason to lose the headline
                out["legs_error"] = repr(e)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
