"""Whole-frame CPU encode through the oracle - TEST / BASELINE INFRASTRUCTURE, never the product path.

The reference's encode flow (encode.py:85-160 `compress_ehem`, encode_mullevel.py:88-154) restated on the oracle's pieces:
quantiser + octree + K records (scp_oracle.proc_pc / mul_proc_pc = data_preprocess.py:13-167), level split / context tensors
(ehem_level_split = encode_dataset_ehem*.py), one EHEM forward per window of <= 8192 nodes (models_ref.ehem_forward =
models/ehem.py:88-136 on PyTorch-CPU), softmax, PMF table in coding order, integer CDFs + range coder (numpyAc).  Every stage is
timed on its own.  Only bench.py's `cpu_baseline` leg, the tests and tests/golden/calibrate_cpu_baseline.py call this.
"""
import os
import time

import numpy as np
import torch

from . import models_ref
from . import scp_oracle as orc


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _full_window_cost(full_times, sampled, warmup):
    """The cost that stands for every full window of a SAMPLED run: the first `warmup` runs are dropped (thread pool, allocator and
    caches settle over the first windows: 2.1 - 2.2 s against 1.6 - 1.8 s later in a full-frame run, profiles/r5_bench_cpu_full.json);
    the FASTEST of the remaining runs is used (the choice that favours the CPU: window times keep falling over the first dozen windows of
    a run, so even the minimum of windows 5 - 10 is above the settled cost of a whole-frame run)."""
    use = full_times[warmup:] if sampled and len(full_times) > warmup else full_times
    return float(np.min(use)) if use else 0.0


def encode_frame(xyz, sd, level, mullevel=True, mode="spher", full_window_runs=None, context_size=8192, data_type="kitti", full_window_warmup=1):
    """One frame, every stage timed.  sd: EHEM state_dict (CPU tensors).

    full_window_runs=None: every window is run (a whole-frame measurement, bits are the real stream's).
    full_window_runs=k   : windows of exactly `context_size` nodes all cost the same (same shapes, data-independent work), so only
                           the first k of them are run (the first is a warm-up when k > 1) and the FASTEST of the rest stands
                           for the others (the choice that favours the CPU: measured against a run of every window it is
                           within a few percent, profiles/cpu_calibration_r2.json); every shorter window is run.  Coder time is measured on the rows that exist and scaled by the
                           node count.  The returned dict says what was measured and what was multiplied."""
    t = {}
    t0 = time.perf_counter()
    if mullevel:
        shells = orc.mullevel_shells(xyz, level, mode, data_type)
        recs = [s["records"] for s in shells]
    else:
        recs = [orc.proc_pc(xyz, orc.kitti_qs(level) if data_type == "kitti" else orc.ford_qs(level), mode)["records"]]
    t["quantise_octree_records"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if mullevel:
        ids, poss, pos_mm, data, oct_seq = orc.ehem_mullevel_context(recs, level)
    else:
        ids, poss, pos_mm, data, oct_seq = orc.ehem_level_split(recs[0], level, polar=mode != "cart")
    t["context"] = time.perf_counter() - t0
    sizes = [len(d) for d in data]
    windows, order = orc.ehem_coding_plan(sizes, context_size, mullevel)
    n_nodes = int(sum(sizes))
    sym_all = np.concatenate([d[:, -1, 2] for d in data]).astype(np.int16)
    level_off = np.concatenate(([0], np.cumsum(sizes)))
    pmf = np.zeros((n_nodes, 255), np.float32)
    have = np.zeros(n_nodes, bool)
    full_times, part_time, n_full, n_part, full_run = [], 0.0, 0, 0, 0
    with torch.no_grad():
        for (l, i, j) in windows:
            c = j - i
            is_full = c == context_size
            n_full += is_full
            n_part += not is_full
            if is_full and full_window_runs is not None and full_run >= full_window_runs:
                continue
            d = torch.from_numpy(data[l][i:j])[None]
            p = torch.from_numpy(np.ascontiguousarray(poss[l][:, i:j]))[None]
            t0 = time.perf_counter()
            o1, o2 = models_ref.ehem_forward(sd, d, p)
            p1 = torch.softmax(o1[0], 1).numpy()
            p2 = torch.softmax(o2[0], 1).numpy() if o2.shape[1] else np.zeros((0, 255), np.float32)
            dt = time.perf_counter() - t0
            r0 = level_off[l] + i
            ne = (c + 1) // 2
            pmf[r0:r0 + ne] = p1                     # coding order inside the window: evens, then odds
            pmf[r0 + ne:r0 + c] = p2
            have[r0:r0 + c] = True
            if is_full:
                full_times.append(dt)
                full_run += 1
            else:
                part_time += dt
    full_med = _full_window_cost(full_times, full_window_runs is not None, full_window_warmup)
    t["model"] = part_time + (full_med * n_full if full_window_runs is not None else float(sum(full_times)))
    sym_coded = sym_all[order]
    rows = np.where(have)[0]
    t0 = time.perf_counter()
    stream, _ = orc.encode_pmf(pmf[rows], sym_coded[rows])
    t_code = time.perf_counter() - t0
    t["cdf_rangecoder"] = t_code * (n_nodes / max(1, len(rows)))
    total = float(sum(t.values()))
    return dict(stage_s={k: round(v, 4) for k, v in t.items()}, total_s=total, n_nodes=n_nodes, windows=len(windows),
                full_windows=int(n_full), partial_windows=int(n_part), full_windows_run=int(full_run),
                full_window_s=[round(x, 3) for x in full_times], full_window_cost_s=round(full_med, 4), full_window_warmup=int(full_window_warmup),
                rows_coded=int(len(rows)),
                bits=8 * len(stream) if len(rows) == n_nodes else None)


def baseline(xyz, sd, level, mullevel, mode="spher", threads=None, full_window_runs=3):
    threads = threads or min(os.cpu_count() or 1, 32)     # PyTorch-CPU oversubscribes badly beyond this on 128-core hosts
    torch.set_num_threads(threads)
    r = encode_frame(xyz, sd, level, mullevel, mode, full_window_runs=full_window_runs)
    r.update(threads=threads, cpu=cpu_model_name(), host_cores=os.cpu_count())
    return r


def encode_frame_octattn(xyz, sd, level, mode="spher", full_window_runs=None, context_size=1024, full_window_warmup=1):
    """The OctAttention flow (encode.py:23-82 `compress` over dataloaders/encode_dataset.py:32-55), same conventions as
    `encode_frame`: every window of exactly `context_size` rows costs the same, the last (shorter) window is always run."""
    t = {}
    t0 = time.perf_counter()
    rec = orc.proc_pc(xyz, orc.kitti_qs(level), mode)["records"]
    t["quantise_octree_records"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ids, pos, data, oct_seq = orc.octattn_context(rec, context_size)
    t["context"] = time.perf_counter() - t0
    n_nodes = len(oct_seq)
    total = len(data)
    pmf = np.zeros((total, 255), np.float32)
    have = np.zeros(total, bool)
    full_times, part_time, n_full, full_run = [], 0.0, 0, 0
    with torch.no_grad():
        for i in range(0, total, context_size):
            c = min(context_size, total - i)
            is_full = c == context_size
            n_full += is_full
            if is_full and full_window_runs is not None and full_run >= full_window_runs:
                continue
            t0 = time.perf_counter()
            o = models_ref.octattn_forward(sd, torch.from_numpy(data[i:i + c])[None], torch.from_numpy(pos[i:i + c])[None])
            pmf[i:i + c] = torch.softmax(o[0], 1).numpy()
            dt = time.perf_counter() - t0
            have[i:i + c] = True
            if is_full:
                full_times.append(dt)
                full_run += 1
            else:
                part_time += dt
    full_med = _full_window_cost(full_times, full_window_runs is not None, full_window_warmup)
    t["model"] = part_time + (full_med * n_full if full_window_runs is not None else float(sum(full_times)))
    real = np.where(have[context_size - 1:])[0]
    sym = oct_seq[:, -1, 0].astype(np.int16)
    t0 = time.perf_counter()
    stream, _ = orc.encode_pmf(pmf[context_size - 1:][real], sym[real])
    t["cdf_rangecoder"] = (time.perf_counter() - t0) * (n_nodes / max(1, len(real)))
    return dict(stage_s={k: round(v, 4) for k, v in t.items()}, total_s=float(sum(t.values())), n_nodes=int(n_nodes),
                windows=-(-total // context_size), full_windows=int(n_full), partial_windows=int(total % context_size != 0),
                full_windows_run=int(full_run), full_window_s=[round(x, 3) for x in full_times], full_window_cost_s=round(full_med, 4),
                full_window_warmup=int(full_window_warmup), rows_coded=int(len(real)),
                bits=8 * len(stream) if len(real) == n_nodes else None)
