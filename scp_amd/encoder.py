"""Frame encoder: LiDAR frame -> range-coded occupancy bitstream, everything between on the MI355X.

This is the hot path of encode.py:85-160 (compress_ehem) / encode_mullevel.py:88-154 and of the datasets feeding it
(dataloaders/encode_dataset_ehem*.py), restructured for the device:

  xyz (H2D once) -> scp_quantize (1 or 3 shells) -> scp_geom_build (one launch sequence for all shells)
     -> scp_geom_context_ehem (ctx u8 [N,12], pos f32 [N,3], sym u8 [N])
     -> EHEM windows (<= 8192 nodes, equal-length windows batched) -> logits scattered straight into CODING ORDER
     -> scp_softmax_cdf over all N rows -> (c_low, c_high) pairs (4 B/node D2H) -> scp_ac_encode_lohi (host)

Per-window softmax, the [N,255] PMF table on the host, the int64 [N,4,6] records and the per-level Python loops of
the reference do not exist here.  The returned dict carries the same scalars the reference prints.
"""
import time
from concurrent.futures import ThreadPoolExecutor

import os

import numpy as np
import torch

from . import native

KITTI, FORD = "kitti", "ford"


def _wait_event(ev, poll_s=0.0005):
    """Wait for a CUDA / HIP event on a worker thread WITHOUT burning a core: hipEventSynchronize spins (measured on the GPU box with
    tools/host_cpu_threads.py: each of the two coder threads sat at 100 % of a core for the whole run - 130 of a rank's 200 CPU-ms per
    frame - blocking-sync flag or not), so the coder thread polls the event and sleeps in between.  The coder is off the critical path
    (frames in flight hide it); half a millisecond of extra latency per frame costs nothing."""
    while not ev.query():
        time.sleep(poll_s)


def level_qs(data_type, level):
    """encode_dataset_ehem.py:164 / encode_dataset_ehem_mullevel.py:162-185."""
    return 400 / (2 ** level - 1) if data_type == KITTI else 2 ** (18 - level)


class EncodePlan:
    """Window list + coding order for one frame (encode.py:109-136): pure host bookkeeping on level sizes."""

    def __init__(self, level_sizes, context_size):
        self.level_sizes = list(level_sizes)
        self.windows = []          # (row_start, length, coded_start)
        row = 0
        for n in self.level_sizes:
            for i in range(0, n, context_size):
                c = min(context_size, n - i)
                self.windows.append((row + i, c, row + i))   # inside a window: evens first, then odds
            row += n
        self.n_rows = row

    def groups(self, max_batch):
        """Windows of equal length, chunked to at most max_batch per model call."""
        by_len = {}
        for w in self.windows:
            by_len.setdefault(w[1], []).append(w)
        out = []
        for c in sorted(by_len, reverse=True):
            ws = by_len[c]
            for i in range(0, len(ws), max_batch):
                out.append((c, ws[i:i + max_batch]))
        return out

    def coding_order_device(self, dev):
        """coding_order() built with vectorised device ops (no per-window host loop over 577k rows)."""
        w = torch.as_tensor(np.asarray(self.windows, np.int64), device=dev)          # [W,3] (start, length, coded)
        start, c = w[:, 0], w[:, 1]
        ne = (c + 1) // 2
        wid = torch.repeat_interleave(torch.arange(len(self.windows), device=dev), c)
        pos = torch.arange(self.n_rows, device=dev) - start[wid]                        # coded position inside the window
        even = pos < ne[wid]
        return start[wid] + torch.where(even, 2 * pos, 2 * (pos - ne[wid]) + 1)

    def coding_order(self):
        """Row index (frame order) of every coded position - for tests and for the reference-compatible view."""
        order = np.empty(self.n_rows, np.int64)
        for start, c, coded in self.windows:
            ne = (c + 1) // 2
            order[coded:coded + ne] = start + np.arange(0, c, 2)
            order[coded + ne:coded + c] = start + np.arange(1, c, 2)
        return order


# scp_swin_ln_qkv addresses its K / V planes through one 32-bit buffer resource: 4 planes x Tp rows x 512 B must stay below 2^31, i.e. a
# packed forward holds at most 1 048 575 rows of the PADDED layout (every window owns a multiple of 512 rows)
MAX_PACKED_ROWS = 1_048_064


def chunk_windows(windows, max_tokens):
    """Cut a frame's window list into the chunks of one packed forward each: [(first window, one past the last)].  A chunk holds at most
    `max_tokens` real tokens AND at most MAX_PACKED_ROWS rows of the padded layout (a window of c nodes owns ceil((c + c % 2) / 512) * 512
    rows at stage 0: tail windows of a few nodes still cost 512 rows each, which a token bound alone does not see)."""
    out, i = [], 0
    while i < len(windows):
        j, tok, rows = i, 0, 0
        while j < len(windows):
            c = windows[j][1]
            r = -(-(c + (c & 1)) // 512) * 512
            if tok and (tok + c > max_tokens or rows + r > MAX_PACKED_ROWS):
                break
            tok += c
            rows += r
            j += 1
        out.append((i, j))
        i = j
    return out


class _PackedInts(list):
    """Per-shell integer clouds that are consecutive views of ONE buffer (`.packed`: the shells back to back)."""
    packed = None


def _quant_of(infos):
    """Per shell the quantiser's (qs[3], offset[3]) as plain floats: what de-quantisation needs and the reference's file name cannot
    carry exactly (decoder.write_sidecar)."""
    return [dict(qs=[float(v) for v in i.qs], offset=[float(v) for v in i.offset]) for i in infos]


class FrameEncoder:
    def __init__(self, model, data_type=KITTI, lidar_level=12, spher=True, cylin=False, mullevel=False, max_batch=8,
                 device=None, packed=True, max_tokens=1_000_000, host_transform=None, profile=None):
        self.model = model
        self.profile = profile          # native.NumericProfile of THIS encoder (None: the process default); current around every launch
        # strict-identity switch (CLI --host_transform, SCP_XFORM=numpy): the float -> integer step of the reference on the host
        # (numpy float32 arctan2 / arccos, data_preprocess.py:42-70) instead of the device transform, whose float64 atan2 / acos is
        # more accurate and therefore gives other integers for a few points per frame.  Everything after the
        # integers is bit-exact on the device either way.
        self.host_transform = (os.environ.get("SCP_XFORM", "") == "numpy") if host_transform is None else bool(host_transform)
        self.packed = packed            # one packed forward for all windows (default) vs one forward per group of equal windows
        self.max_tokens = max_tokens
        self.data_type = data_type
        self.lidar_level = lidar_level
        self.mode = native.CYLIN if cylin else (native.SPHER if spher else native.CART)
        self.spher, self.cylin = spher and not cylin, cylin
        self.mullevel = mullevel
        if mullevel and self.mode == native.CART:
            # encode_dataset_ehem_mullevel.py:97-186 has a cylindrical and a spherical branch only
            raise native.ScpError("--mullevel needs --spher or --cylin (the reference has no Cartesian multi-level path)")
        self.max_batch = max_batch
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.context_size = model.cfg.model.context_size
        self.geom = native.Geom()
        self.cart_offset = -200.0 if data_type == KITTI else -float(2 ** 17)

    # ------------------------------------------------------------------------------------------ stage G
    def shells(self):
        L = self.lidar_level
        return [([0, 0], L), ([0, 1], L + 1), ([1], L + 2)] if self.mullevel else [(None, L)]

    def host_ints(self, xyz):
        """The strict-identity front end: numpy [P,>=3] float32 -> (per-shell int32 arrays, infos) exactly as the reference computes
        them (scp_amd/data_preproc/data_preprocess.py: host_quantize).  Runs on the host; the CLI calls it on its reader thread."""
        from .data_preproc.data_preprocess import host_quantize_shells
        xyz = np.ascontiguousarray(xyz.cpu().numpy() if isinstance(xyz, torch.Tensor) else xyz, np.float32)
        out = host_quantize_shells(xyz, self.mode, [level_qs(self.data_type, lv) for _, lv in self.shells()], 0.0 if self.mullevel else self.cart_offset)
        # the shells' integers back to back in ONE pinned buffer (this runs on the reader / prefetch thread): the launch thread then issues a
        # single asynchronous host -> device copy and no concatenation kernel; the list holds numpy views of the buffer
        n = [q.shape[0] for q, _ in out]
        buf = torch.empty((sum(n), 3), dtype=torch.int32, pin_memory=torch.cuda.is_available())
        views, a = _PackedInts(), 0
        for (q, _), k in zip(out, n):
            v = buf[a:a + k].numpy()
            v[...] = q
            views.append(v)
            a += k
        views.packed = buf
        return views, [i for _, i in out]

    def quantize(self, xyz_dev, ints=None):
        """xyz -> list of per-shell integer clouds (device int32 [P,3]) + (bin_num, z_offset)."""
        if self.host_transform or ints is not None:
            hq, infos = ints if ints is not None else self.host_ints(xyz_dev)
            self._infos = infos
            packed = getattr(hq, "packed", None)
            if packed is not None:       # one pinned buffer -> one async copy; the per-shell clouds are views of it
                qcat = packed.to(self.device, non_blocking=True)
                qs, a = _PackedInts(), 0
                for q in hq:
                    qs.append(qcat[a:a + q.shape[0]])
                    a += q.shape[0]
                qs.packed = qcat
            else:
                qs = [torch.from_numpy(np.ascontiguousarray(q, np.int32)).to(self.device, non_blocking=True) for q in hq]
            return qs, infos[0].bin_num, (infos[0].offset[2] if self.cylin else 0.0)
        qs, infos = [], []
        for path, lv in self.shells():
            q, qi, _ = native.quantize(xyz_dev, self.mode, level_qs(self.data_type, lv),
                                       0.0 if self.mullevel else self.cart_offset)
            qs.append(q)
            infos.append(qi)
        self._infos = infos
        return qs, infos[0].bin_num, (infos[0].offset[2] if self.cylin else 0.0)

    def distortion(self, xyz_dev, infos=None):
        """Chamfer distance and D1 PSNR of the frame just encoded (the octree of the last `encode` / `preprocess` call) against
        its input cloud, on the device - what the reference gets from distChamfer + the pc_error tool
        (encode_dataset_ehem.py:170-171, encode_dataset_ehem_mullevel.py:141-144; scp_amd/metrics.py)."""
        from . import metrics
        pts = []
        for s, info in enumerate(infos or self._infos):   # infos: per shell, anything with .qs[3] and .offset[3] (tests: the oracle's)
            pts.append(metrics.dequantize(self.geom.leaves(s), info.qs, info.offset, spher=self.spher, cylin=self.cylin,
                                          f32=not self.mullevel).double())
        return metrics.chamfer_psnr(xyz_dev, torch.cat(pts), metrics.PEAK.get(self.data_type, 1.0))

    def preprocess(self, xyz_dev, ints=None):
        if ints is None and not self.host_transform:
            # device transform: stage G1 + G2 as ONE launch sequence (scp_geom_build_xyz: every point transformed once, keys for all
            # shells and the sort's first histogram out of one kernel, two host read-backs per frame)
            self._infos = self.geom.build_xyz([xyz_dev], self.mode, [level_qs(self.data_type, lv) for _, lv in self.shells()],
                                              0.0 if self.mullevel else self.cart_offset, [(path, self.mullevel) for path, _ in self.shells()])
            pre = self._tables(self._infos[0].bin_num, self._infos[0].offset[2] if self.cylin else 0.0, xyz_dev.shape[0])
        else:
            qs, bin_num, z_off = self.quantize(xyz_dev, ints)
            pre = self.preprocess_ints(qs, bin_num, z_off, xyz_dev.shape[0])
        pre["bin_nums"] = [float(i.bin_num) for i in self._infos]      # every shell's own (the file name carries the first)
        pre["quant"] = _quant_of(self._infos)                          # the steps and offsets the integers were made with (sidecar)
        return pre

    def _tables(self, bin_num, z_offset, n_points):
        """ctx / pos / coded symbols of every segment of the geometry just built, one launch (scp_geom_context_ehem_all)."""
        pos_mode = native.POS_MINMAX_MUL if self.mullevel else (native.POS_POW2 if self.mode == native.CART else native.POS_MINMAX)
        ctx, pos, sym_coded, mm = self.geom.context_ehem_all(pos_mode, self.lidar_level, self.context_size)
        sizes = []
        for s in range(len(self.geom.info)):
            counts = self.geom.level_counts(s)
            if self.mullevel:
                counts[-1] -= 1          # Octree.py:259-262: the records drop the last BFS node
            sizes += counts
        return dict(ctx=ctx, pos=pos, sym_coded=sym_coded, pos_mm=mm, level_sizes=sizes, bin_num=bin_num, z_offset=z_offset, n_points=n_points)

    def preprocess_ints(self, qs, bin_num, z_offset, n_points):
        """qs: per-shell quantised integer clouds (device int32 [P_s,3]) - the entry point for already-quantised input
        (the reference's --preproc_path flow).  -> ctx/pos/sym device tensors for all shells, level sizes, meta."""
        packed = getattr(qs, "packed", None)
        q = packed if packed is not None else (torch.cat(qs) if len(qs) > 1 else qs[0])
        segs, off = [], 0
        for (path, _), qq in zip(self.shells(), qs):
            segs.append((off, qq.shape[0], path, self.mullevel))
            off += qq.shape[0]
        self.geom.build(q.contiguous(), segs)
        return self._tables(bin_num, z_offset, n_points)

    def preprocess_records(self, records, bin_num, z_offset, n_points):
        """--preproc_path flow (encode_dataset_ehem.py:149-157 + :52-105): the reference's int64 [N,4,6] record files (one per
        shell) -> the same ctx/pos/sym tables the device octree path produces.  Pure index plumbing on the device."""
        L = self.lidar_level
        ctxs, poss, syms, mms, sizes = [], [], [], [], []
        for rec in records:
            r = (torch.from_numpy(np.ascontiguousarray(rec, np.int64)) if isinstance(rec, np.ndarray) else rec).to(self.device)
            lvl = r[:, 3, 1]
            depth = int(lvl.max())
            counts = torch.bincount(lvl, minlength=depth + 1)[1:]
            last = (lvl == depth)[:, None]
            levels = torch.where(last, torch.clamp(r[:, :, 1], max=L), r[:, :, 1])          # ehem:86 clips the last chunk
            ctx = torch.stack((levels, r[:, :, 2], r[:, :, 0] - 1), 2).reshape(-1, 12).to(torch.uint8)
            p = r[:, 3, 3:6]
            if self.mode == native.CART and not self.mullevel:
                pos = (p.double() / float(2 ** depth)).float()
                mm = torch.zeros((depth, 2), dtype=torch.int64, device=self.device)
            else:
                mn = torch.full((depth + 1,), 2 ** 62, dtype=torch.int64, device=self.device).scatter_reduce(0, lvl, p.min(1)[0], "amin")
                mx = torch.full((depth + 1,), -2 ** 62, dtype=torch.int64, device=self.device).scatter_reduce(0, lvl, p.max(1)[0], "amax")
                eps = torch.where((lvl == depth) & torch.tensor(self.mullevel, device=self.device), 0.0, 1e-9).double()
                pos = ((p - mn[lvl][:, None]).double() / ((mx - mn)[lvl].double() + eps)[:, None]).float()
                mm = torch.stack((mn[1:], mx[1:]), 1)
            ctxs.append(ctx); poss.append(pos.contiguous()); syms.append((r[:, 3, 0] - 1).to(torch.uint8)); mms.append(mm)
            sizes += counts.tolist()
        return dict(ctx=torch.cat(ctxs), pos=torch.cat(poss), sym=torch.cat(syms), pos_mm=torch.cat(mms), level_sizes=sizes,
                    bin_num=bin_num, z_offset=z_offset, n_points=n_points)

    def encode_records(self, records, bin_num, z_offset, n_points, timing=False):
        """Encode from the reference's preprocessed record files (list of int64 [N,4,6], one per shell)."""
        t0 = time.perf_counter()
        return self._encode_pre(self.preprocess_records(records, bin_num, z_offset, n_points), t0, timing)

    # ------------------------------------------------------------------------------------------ stage M + C
    def logits_in_coding_order(self, pre, plan):
        with native.use_profile(self.profile):
            return self._logits_in_coding_order(pre, plan)

    def profile_string(self):
        return native.numeric_profile("EHEM", self.profile)

    def _logits_in_coding_order(self, pre, plan):
        if self.packed:
            return self.logits_packed(pre, plan)
        N = plan.n_rows
        table = torch.empty((N, 255), dtype=torch.float32, device=self.device)
        ctx, pos = pre["ctx"], pre["pos"]
        for c, ws in plan.groups(self.max_batch):
            starts = [w[0] for w in ws]
            if len(ws) == 1:
                bctx, bpos = ctx[starts[0]:starts[0] + c][None], pos[starts[0]:starts[0] + c][None]
            else:
                bctx = torch.stack([ctx[s:s + c] for s in starts])
                bpos = torch.stack([pos[s:s + c] for s in starts])
            o1, o2 = self.model.forward_ctx(bctx, bpos)
            ne = (c + 1) // 2
            for b, (start, _, coded) in enumerate(ws):
                table[coded:coded + ne] = o1[b]
                if c > 1:
                    table[coded + ne:coded + c] = o2[b]
        return table

    def logits_packed(self, pre, plan):
        """All windows of the frame through ONE packed forward per chunk of <= max_tokens tokens (models/packed.py); the last layer
        of both probability heads writes its rows straight into the coding-order table (row stride 256 floats: 16-byte rows)."""
        full = torch.empty((plan.n_rows, 256), dtype=torch.float32, device=self.device)
        ctx, pos = pre["ctx"], pre["pos"]
        for r0, tok, pp in (pre.get("packed_plans") or self.packed_plans(plan)):
            self.model.forward_packed(ctx[r0:r0 + tok], pos[r0:r0 + tok], None, plan=pp, table=full[r0:])
        return full[:, :255]

    def packed_plans(self, plan):
        """The frame's windows cut into chunks of <= max_tokens tokens, each with its index maps: [(first row, tokens, PackedPlan)]."""
        from .models.packed import PackedPlan
        ws = plan.windows
        return [(ws[i][0], sum(w[1] for w in ws[i:j]), PackedPlan([w[1] for w in ws[i:j]], device=self.device))
                for i, j in chunk_windows(ws, self.max_tokens)]

    def encode(self, xyz, timing=False):
        """xyz: numpy / torch float32 [P,3].  Returns dict(bytes, bits, bpp, n_nodes, n_points, bin_num, z_offset,
        n_levels, pos_mm (numpy [n_levels,2]), times)."""
        t0 = time.perf_counter()
        if isinstance(xyz, np.ndarray):
            xyz = torch.from_numpy(np.ascontiguousarray(xyz, np.float32))
        ints = self.host_ints(xyz) if self.host_transform else None        # from the caller's copy (no D2H of a frame just uploaded)
        xyz_dev = xyz.to(self.device, non_blocking=True)
        return self._encode_pre(self.preprocess(xyz_dev, ints), t0, timing)

    def encode_ints(self, qs, bin_num, z_offset, n_points, timing=False):
        """Encode from already-quantised integer coordinates (list of per-shell int32 [P_s,3] arrays / tensors)."""
        t0 = time.perf_counter()
        dq = [(torch.from_numpy(np.ascontiguousarray(q, np.int32)) if isinstance(q, np.ndarray) else q).to(self.device)
              for q in qs]
        return self._encode_pre(self.preprocess_ints(dq, bin_num, z_offset, n_points), t0, timing)

    # ------------------------------------------------------------------------------------------ pipelined variant
    def _init_streams(self):
        if hasattr(self, "_pool"):
            return
        self._pool = ThreadPoolExecutor(max_workers=2)
        self._front_pool = ThreadPoolExecutor(max_workers=1)     # front_async: stage G + plans of the next frame (see there)
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._front_stream = torch.cuda.Stream(device=self.device, priority=-1)   # high priority: its tiny kernels slip in between the model's
        # Lanes: consecutive frames run their model part on alternating streams, so the launch gaps and tails of one frame's
        # kernels are filled by the other's (the kernels are whole-GPU persistent launches: they interleave rather than
        # co-run).  SCP_LANES=1 keeps every frame on the caller's stream.
        self._lanes = [torch.cuda.Stream(device=self.device) for _ in range(max(1, int(os.environ.get("SCP_LANES", "2"))))]
        self._lane_i = 0

    def _front(self, xyz, ints, caller):
        """Stage G and the window plans of one frame on the front stream (a side stream: launch-bound work with two small D2H syncs, which
        runs under the previous frames' model kernels).  -> everything the model part needs + the event it waits for."""
        torch.cuda.set_device(self.device)                     # (the current device is per host thread: this may be the front thread)
        if isinstance(xyz, np.ndarray):
            xyz = torch.from_numpy(np.ascontiguousarray(xyz, np.float32))
        if self.host_transform and ints is None:
            ints = self.host_ints(xyz)
        with torch.cuda.stream(self._front_stream):
            self._front_stream.wait_stream(caller)             # whatever produced `xyz` on the caller's stream
            pre = self.preprocess(xyz.to(self.device, non_blocking=True), ints)
            plan = EncodePlan(pre["level_sizes"], self.context_size)
            if self.packed:
                pre["packed_plans"] = self.packed_plans(plan)
            sym_coded = self._sym_coded(pre, plan)
            ready = torch.cuda.Event()
            ready.record()
        return dict(pre=pre, plan=plan, sym_coded=sym_coded, ready=ready)

    def front_async(self, xyz, ints=None):
        """Start a frame's front part (stage G, plans) on the encoder's FRONT THREAD and return a future for `encode_async(..., front=)`.
        The front part blocks its host thread for most of a frame time - its ~35 small dependent kernels each wait for a gap between the
        whole-GPU model kernels of the frames in flight (33 - 43 ms per frame in `bench.py`, against 0.4 ms alone on the GPU) - so a loop that
        calls this one frame ahead keeps the launch thread free for the model part (ctypes and torch release the GIL while they wait).
        Call order = frame order: the front thread takes the frames one at a time (one `scp_geom` handle)."""
        self._init_streams()
        caller = torch.cuda.current_stream(self.device)
        return self._front_pool.submit(self._front, xyz, ints, caller)

    def encode_async(self, xyz, ints=None, front=None):
        """Like encode(), but the D2H copy of the (c_low, c_high) pairs and the serial host range coder run on a worker
        thread (ctypes releases the GIL) behind an event on a side stream, so the caller can enqueue the next frame while this
        one is being coded.  Returns a handle; `finish(handle)` blocks and returns the usual result dict.
        ints: the result of `host_ints(xyz)` when the caller has already computed it (strict-identity mode, CLI reader thread).
        front: the future `front_async(xyz, ints)` returned for this frame (then xyz / ints are not looked at again)."""
        t0 = time.perf_counter()
        self._init_streams()
        # Front part on its own stream: stage G (with its small D2H syncs) and the window plans (one plan_kernel launch per chunk)
        # are launch-bound; on a side stream they run under the previous frame's model kernels instead of in front of this
        # frame's.  Everything allocated there stays referenced by the handle until finish(), i.e. past its last use on the
        # main stream, so the caching allocator cannot hand it out again early.
        caller = torch.cuda.current_stream(self.device)
        if len(self._lanes) > 1:
            main = self._lanes[self._lane_i % len(self._lanes)]
            self._lane_i += 1
            main.wait_stream(caller)
        else:
            main = caller
        fills0 = native.CACHE_FILLS
        f = front.result() if front is not None else self._front(xyz, ints, caller)
        pre, plan, sym_coded, ready = f["pre"], f["plan"], f["sym_coded"], f["ready"]
        main.wait_event(ready)
        with torch.cuda.stream(main):
            table = self.logits_in_coding_order(pre, plan)
            lohi = native.softmax_cdf(table, sym_coded)["lohi"]
            done = torch.cuda.Event()
            done.record()
            if native.CACHE_FILLS != fills0:               # device-side caches were built during this call (first frames only):
                main.synchronize()                         # finish them before anything is enqueued on another lane
        host = torch.empty(lohi.shape, dtype=lohi.dtype, pin_memory=True)
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(done)
            host.copy_(lohi, non_blocking=True)
            lohi.record_stream(self._copy_stream)
            copied = torch.cuda.Event(blocking=True)      # the coder thread SLEEPS until the pairs have landed (a default event spins a core)
            copied.record()

        def work():
            _wait_event(copied)
            return native.ac_encode_lohi(host.numpy())

        fut = self._pool.submit(work)
        return dict(future=fut, pre=pre, plan=plan, t0=t0, keep=(sym_coded, table, lohi))

    # ------------------------------------------------------------------------------------------ batches of small frames
    def preprocess_batch(self, frames, ints=None):
        """Stage G of k frames at once: one scp_geom_build with every (frame, shell) as a segment - one radix sort, one tree pass - and
        the frames' context tables back to back.  Frames share nothing (encode.py:274-291 rebuilds everything per frame): the batch is
        ONE sequence of levels - the frames' level lists one after the other - for the window plan, the packed forward and the CDF
        kernel, and is cut into frames again in front of the range coder.  Returns (combined `pre`, per-frame meta).
        ints: per frame the result of `host_ints(frame)` when the caller has computed it already (strict-identity mode, a prefetch thread)."""
        ns = len(self.shells())
        if len(frames) * ns > 62:
            raise native.ScpError("a batch holds at most 62 (frame, shell) trees (SCP_MAX_SEGMENTS)")
        if self.host_transform or ints is not None:
            qs_all, infos = [], []
            for f, xyz_dev in enumerate(frames):
                hq, inf = ints[f] if ints is not None else self.host_ints(xyz_dev)
                # copy from the PINNED TENSOR itself, as quantize() does: torch's host allocator records the stream use of a tensor it knows,
                # so the block is not handed to the next frame's host_ints while this asynchronous copy still reads it (a from_numpy view of
                # the same memory is invisible to the allocator: the next frame's integers could overwrite an earlier frame's in flight)
                packed = getattr(hq, "packed", None)
                if packed is not None:
                    qcat, a = packed.to(self.device, non_blocking=True), 0
                    for q in hq:
                        qs_all.append(qcat[a:a + q.shape[0]])
                        a += q.shape[0]
                else:      # plain per-shell arrays computed elsewhere (as quantize() accepts them): pageable memory, synchronous copies
                    qs_all += [torch.from_numpy(np.ascontiguousarray(q, np.int32)).to(self.device) for q in hq]
                infos += inf
            segs, off = [], 0
            for f in range(len(frames)):
                for (path, _), qq in zip(self.shells(), qs_all[f * ns:(f + 1) * ns]):
                    segs.append((off, qq.shape[0], path, self.mullevel))
                    off += qq.shape[0]
            self.geom.build(torch.cat(qs_all).contiguous(), segs)
        else:
            infos = self.geom.build_xyz(list(frames), self.mode, [level_qs(self.data_type, lv) for _, lv in self.shells()],
                                        0.0 if self.mullevel else self.cart_offset, [(path, self.mullevel) for path, _ in self.shells()])
        self._infos = infos[-ns:]
        pre = self._tables(infos[0].bin_num, 0.0, 0)
        metas, sizes, mm0 = [], pre["level_sizes"], 0
        depths = [i.depth for i in self.geom.info]
        for f, xyz_dev in enumerate(frames):
            fi = infos[f * ns:(f + 1) * ns]
            nl = sum(depths[f * ns:(f + 1) * ns])
            fsizes = sizes[mm0:mm0 + nl]
            metas.append(dict(bin_num=fi[0].bin_num, z_offset=(fi[0].offset[2] if self.cylin else 0.0), n_points=int(xyz_dev.shape[0]),
                              bin_nums=[float(i.bin_num) for i in fi], quant=_quant_of(fi), pos_mm=pre["pos_mm"][mm0:mm0 + nl],
                              level_sizes=fsizes, n_nodes=int(sum(fsizes))))
            mm0 += nl
        return dict(ctx=pre["ctx"], pos=pre["pos"], sym_coded=pre["sym_coded"], level_sizes=sizes), metas

    def encode_batch_async(self, frames, ints=None):
        """k frames through ONE stage G, ONE packed forward and ONE CDF launch (the small-frame configurations: a level-12 frame has
        22 windows, far too few for an MI355X - BASELINE.json configs[1] is a batch of 16 of them).  The streams are byte-identical
        to the per-frame path (every kernel is per window / per row).  Returns a handle for finish_batch().
        ints: [host_ints(frame) for every frame] when the caller has them already (strict-identity mode)."""
        t0 = time.perf_counter()
        self._init_streams()
        # the same stream structure as encode_async: stage G + plans on the high-priority front stream, the model part on the next lane
        caller = torch.cuda.current_stream(self.device)
        if len(self._lanes) > 1:
            main = self._lanes[self._lane_i % len(self._lanes)]
            self._lane_i += 1
            main.wait_stream(caller)
        else:
            main = caller
        fills0 = native.CACHE_FILLS
        with torch.cuda.stream(self._front_stream):
            self._front_stream.wait_stream(caller)
            dev_frames = [(torch.from_numpy(np.ascontiguousarray(x, np.float32)) if isinstance(x, np.ndarray) else x).to(self.device, non_blocking=True)
                          for x in frames]
            pre, metas = self.preprocess_batch(dev_frames, ints)
            plan = EncodePlan(pre["level_sizes"], self.context_size)
            if self.packed:
                pre["packed_plans"] = self.packed_plans(plan)
            sym_coded = self._sym_coded(pre, plan)
            ready = torch.cuda.Event()
            ready.record()
        main.wait_event(ready)
        with torch.cuda.stream(main):
            table = self.logits_in_coding_order(pre, plan)
            lohi = native.softmax_cdf(table, sym_coded)["lohi"]
            done = torch.cuda.Event()
            done.record()
            if native.CACHE_FILLS != fills0:
                main.synchronize()
        host = torch.empty(lohi.shape, dtype=lohi.dtype, pin_memory=True)
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(done)
            host.copy_(lohi, non_blocking=True)
            lohi.record_stream(self._copy_stream)
            copied = torch.cuda.Event(blocking=True)      # the coder thread SLEEPS until the pairs have landed (a default event spins a core)
            copied.record()
        cuts = np.concatenate(([0], np.cumsum([m["n_nodes"] for m in metas])))

        def work():
            _wait_event(copied)
            h = host.numpy()
            return [native.ac_encode_lohi(h[cuts[f]:cuts[f + 1]]) for f in range(len(metas))]
        return dict(future=self._pool.submit(work), metas=metas, t0=t0, keep=(pre, plan, sym_coded, table, lohi, dev_frames))

    def _sym_coded(self, pre, plan):
        """The coded symbols in coding order: written by the context kernel itself (scp_geom_context_ehem_all), or - for tables that
        did not come out of it (the --preproc_path record files) - gathered through the plan's coding-order index."""
        if pre.get("sym_coded") is not None:
            return pre["sym_coded"]
        return pre["sym"][plan.coding_order_device(self.device)].contiguous()

    def finish_batch(self, h):
        out = []
        for stream, m in zip(h["future"].result(), h["metas"]):
            bits = 8 * len(stream)
            out.append(dict(bytes=stream, bits=bits, bpp=bits / m["n_points"], n_nodes=m["n_nodes"], n_points=m["n_points"], bin_num=m["bin_num"],
                            z_offset=m["z_offset"], n_levels=len(m["level_sizes"]), pos_mm=m["pos_mm"].cpu().numpy(), level_sizes=m["level_sizes"],
                            bin_nums=m["bin_nums"], quant=m.get("quant"), times=dict(total=(time.perf_counter() - h["t0"]) / len(h["metas"]))))
        return out

    def finish(self, h):
        stream = h["future"].result()
        pre, plan = h["pre"], h["plan"]
        bits = 8 * len(stream)
        return dict(bytes=stream, bits=bits, bpp=bits / pre["n_points"], n_nodes=plan.n_rows, n_points=pre["n_points"],
                    bin_num=pre["bin_num"], z_offset=pre["z_offset"], n_levels=len(pre["level_sizes"]),
                    pos_mm=pre["pos_mm"].cpu().numpy(), level_sizes=pre["level_sizes"], bin_nums=pre.get("bin_nums", [float(pre["bin_num"])]),
                    quant=pre.get("quant"), times=dict(total=time.perf_counter() - h["t0"]))

    def _encode_pre(self, pre, t0, timing):
        if timing:
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        plan = EncodePlan(pre["level_sizes"], self.context_size)
        table = self.logits_in_coding_order(pre, plan)
        sym_coded = self._sym_coded(pre, plan)
        if timing:
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        lohi = native.softmax_cdf(table, sym_coded)["lohi"].cpu().numpy()
        t3 = time.perf_counter()
        stream = native.ac_encode_lohi(lohi)
        t4 = time.perf_counter()
        bits = 8 * len(stream)
        return dict(bytes=stream, bits=bits, bpp=bits / pre["n_points"], n_nodes=plan.n_rows, n_points=pre["n_points"],
                    bin_num=pre["bin_num"], z_offset=pre["z_offset"], n_levels=len(pre["level_sizes"]),
                    pos_mm=pre["pos_mm"].cpu().numpy(), level_sizes=pre["level_sizes"], bin_nums=pre.get("bin_nums", [float(pre["bin_num"])]),
                    quant=pre.get("quant"), times=dict(geom=t1 - t0, model=t2 - t1, cdf=t3 - t2, coder=t4 - t3, total=t4 - t0),
                    _debug=dict(table=table, sym_coded=sym_coded, pre=pre))

    def outfile(self, base, res):
        """encode.py:140-144 file name."""
        if self.spher:
            base += "_spher"
        elif self.cylin:
            base += "_cylin"
        return base + "_" + str(res["n_levels"]) + "_" + str(int(res["bin_num"])) + "_" + str(int(res["z_offset"])) + ".bin"


class OctAttnFrameEncoder:
    """OctAttention path.  Same-level (encode.py:23-82 `compress` + dataloaders/encode_dataset.py:32-55): one BFS sequence,
    front-padded with context_size-1 rows (occ 255), cut into consecutive 1024-windows; node r is predicted at position
    (r+1023) % 1024 of window (r+1023) // 1024.  Plain BFS coding order.  `--cylin` is wired here (the reference forgot to, SURVEY B-7).

    Multi-level / level-wise (encode_mullevel.py:23-86 `compress` + dataloaders/encode_dataset_mullevel.py:27-119): the frame is a
    LIST of sequences ("chunks") - one per rho shell ([0,0] at qs(L), [0,1] at qs(L+1), [1] at qs(L+2); records without the last
    BFS node, Octree.py:259-262), and with `level_wise` one per octree level of every shell - each front-padded and windowed on
    its own, positions divided by 2^(the shell's deepest level); the PMF rows of the chunks are stacked in order
    (`probabilities[:-1023]`) and coded as one stream.  File name `<base>[_spher]_<chunks>_<bin_num>_0.bin`."""

    def __init__(self, model, data_type=KITTI, lidar_level=12, spher=True, cylin=False, max_batch=128, device=None, mullevel=False,
                 level_wise=False, named=False, host_transform=None):
        self.model = model
        self.host_transform = (os.environ.get("SCP_XFORM", "") == "numpy") if host_transform is None else bool(host_transform)   # see FrameEncoder
        self.data_type = data_type
        self.lidar_level = lidar_level
        self.mode = native.CYLIN if cylin else (native.SPHER if spher else native.CART)
        self.spher, self.cylin = spher and not cylin, cylin
        self.mullevel = mullevel
        self.level_wise = level_wise
        self.named = named or mullevel        # encode_mullevel.py's file-name scheme (also for its single-shell Cartesian input)
        if mullevel and self.mode != native.SPHER:
            # encode_dataset_mullevel.py:76-86: the three-shell records exist for --spher only
            raise native.ScpError("OctAttention multi-level encoding needs --spher (encode_dataset_mullevel.py:76)")
        self.max_batch = max_batch
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.context_size = model.cfg.model.context_size
        self.geom = native.Geom()
        self.cart_offset = -200.0 if data_type == KITTI else -float(2 ** 17)

    def shells(self):
        L = self.lidar_level
        return [([0, 0], L), ([0, 1], L + 1), ([1], L + 2)] if self.mullevel else [(None, L)]

    def quantize(self, xyz_dev):
        if self.host_transform:
            from .data_preproc.data_preprocess import host_quantize_shells
            xyz = np.ascontiguousarray(xyz_dev.cpu().numpy(), np.float32)
            out = host_quantize_shells(xyz, self.mode, [level_qs(self.data_type, lv) for _, lv in self.shells()], 0.0 if self.mullevel else self.cart_offset)
            return [torch.from_numpy(q).to(self.device) for q, _ in out], out[0][1].bin_num
        qs, bin_num = [], None
        for path, lv in self.shells():
            q, qi, _ = native.quantize(xyz_dev, self.mode, level_qs(self.data_type, lv), 0.0 if self.mullevel else self.cart_offset)
            qs.append(q)
            bin_num = qi.bin_num if bin_num is None else bin_num
        return qs, bin_num

    def encode(self, xyz, timing=False, sequential=False):
        t0 = time.perf_counter()
        if isinstance(xyz, np.ndarray):
            xyz = torch.from_numpy(np.ascontiguousarray(xyz, np.float32))
        xyz_dev = xyz.to(self.device)
        if not self.host_transform:
            bin_num = self.build_from_xyz(xyz_dev)
            return self.encode_ints(None, bin_num, xyz_dev.shape[0], t0, sequential=sequential, front=self._front(None))
        qs, bin_num = self.quantize(xyz_dev)
        return self.encode_ints(qs, bin_num, xyz_dev.shape[0], t0, sequential=sequential)

    def _sequential_rows(self, seq_ctx, seq_pos, N, table):
        """`--sequential` (encode.py:38-41,55-56): a window starts at EVERY row of the padded sequence and only the prediction
        of its last position is kept, i.e. node r is predicted from its full 1023-node history (window r .. r + 1023).
        The reference's loop runs past the last full window with shrinking windows that all end at the last node and
        overwrite its row; the last one (that node alone) wins - reproduced.  ~context_size x the model calls of the default
        mode: batches of `max_batch` windows are gathered with a strided view, one forward each."""
        cs = self.context_size
        wc = seq_ctx.unfold(0, cs, 1).permute(0, 2, 1)               # [N, cs, 12] view: window i = rows i .. i + cs - 1
        wp = seq_pos.unfold(0, cs, 1).permute(0, 3, 1, 2)            # [N, cs, 4, 3]
        for b0 in range(0, N, self.max_batch):
            b1 = min(N, b0 + self.max_batch)
            out = self.model(wc[b0:b1].reshape(b1 - b0, cs, 4, 3), wp[b0:b1].contiguous())
            table[b0:b1] = out[:, -1]
        if cs > 1:
            table[N - 1] = self.model(seq_ctx[-1:].reshape(1, 1, 4, 3), seq_pos[-1:].reshape(1, 1, 4, 3))[0, -1]

    def build_from_xyz(self, xyz_dev):
        """Device transform: stage G1 + G2 in one launch sequence (scp_geom_build_xyz) -> bin_num of the first shell."""
        infos = self.geom.build_xyz([xyz_dev], self.mode, [level_qs(self.data_type, lv) for _, lv in self.shells()],
                                    0.0 if self.mullevel else self.cart_offset, [(path, self.mullevel) for path, _ in self.shells()])
        return infos[0].bin_num

    def _front(self, qs):
        """stage G + the front-padded context sequences (encode_dataset.py:32-55 / encode_dataset_mullevel.py:44-73) on the
        current stream.  qs: one integer cloud, or the list of per-shell clouds; None: the geometry has been built already
        (build_from_xyz).  -> (chunks [(seq_ctx, seq_pos, n)], sym, N)."""
        if qs is not None:
            if not isinstance(qs, (list, tuple)):
                qs = [qs]
            if len(qs) != len(self.shells()):
                raise native.ScpError(f"expected {len(self.shells())} integer cloud(s), got {len(qs)}")
            segs, off = [], 0
            for (path, _), qq in zip(self.shells(), qs):
                segs.append((off, qq.shape[0], path, self.mullevel))
                off += qq.shape[0]
            self.geom.build((torch.cat(qs) if len(qs) > 1 else qs[0]).contiguous(), segs)
        segs = self.geom.segments
        cs = self.context_size
        pad_ctx = torch.zeros((cs - 1, 12), dtype=torch.uint8, device=self.device)
        pad_ctx[:, 0::3] = 255
        pad_pos = torch.zeros((cs - 1, 4, 3), dtype=torch.float32, device=self.device)
        chunks, syms = [], []
        for s in range(len(segs)):
            ctx, pos, sym = self.geom.context_octattn(s)
            counts = self.geom.level_counts(s)
            if self.mullevel:
                counts[-1] -= 1                      # the records drop the last BFS node (Octree.py:259-262)
                if counts[-1] == 0:                  # ... and with it the deepest level: positions are over 2^(max level present)
                    counts.pop()
                    pos = pos * 2.0
            syms.append(sym)
            cuts = counts if self.level_wise else [ctx.shape[0]]
            a = 0
            for n in cuts:
                chunks.append((torch.cat((pad_ctx, ctx[a:a + n])), torch.cat((pad_pos, pos[a:a + n])), n))
                a += n
        sym = torch.cat(syms) if len(syms) > 1 else syms[0]
        return chunks, sym, int(sym.shape[0])

    def _chunk_rows(self, chunks, table):
        """Default mode: consecutive windows of context_size over every padded chunk; all windows of the frame (whatever chunk they belong to)
        share batched forwards.  Round 6: a chunk's shorter TAIL window rides in the same forwards, padded behind its last row to a full window
        - the model is causal (attention_model.py:58-95: row i attends to rows <= i; everything else is per row), so the rows in front of the
        padding come out as in a forward of the short window alone, and the frame saves that forward's ~45 launches on one window (1 ms of a
        15 ms L12 frame)."""
        cs = self.context_size
        full_c, full_p, dst = [], [], []       # windows + (table row of the window's first real node, leading pad rows, real rows in the window)
        row = 0
        for seq_ctx, seq_pos, n in chunks:
            total = n + cs - 1
            n_full = total // cs
            for w in range(n_full):
                full_c.append(seq_ctx[w * cs:(w + 1) * cs])
                full_p.append(seq_pos[w * cs:(w + 1) * cs])
                lo = max(w * cs - (cs - 1), 0)
                dst.append((row + lo, lo + (cs - 1) - w * cs, cs))
            k = total - n_full * cs
            if k:
                # (padding rows: the front padding's rows - valid embedding indices - and zero positions)
                full_c.append(torch.cat((seq_ctx[n_full * cs:], seq_ctx[:cs - k])))
                full_p.append(torch.cat((seq_pos[n_full * cs:], torch.zeros_like(seq_pos[:cs - k]))))
                lo = n_full * cs - (cs - 1)
                dst.append((row + max(lo, 0), max(-lo, 0), k))
            row += n
        # forwards of EQUAL size (round 6): 286 windows under max_batch = 128 ran as 128 + 128 + 30 - the short third forward fills the chip
        # badly (26.4 - 26.7 frames/s at L14 --cylin) - and run as 96 + 96 + 94 (27.2)
        nfw = -(-len(dst) // self.max_batch) if dst else 0
        step = -(-len(dst) // nfw) if nfw else 1
        for b0 in range(0, len(dst), step):
            b1 = min(len(dst), b0 + step)
            contiguous = len(chunks) == 1 and all(d[2] == cs for d in dst[b0:b1])
            if contiguous:                     # one sequence: its full windows are one contiguous block (no copy)
                d = chunks[0][0][b0 * cs:b1 * cs].reshape(b1 - b0, cs, 4, 3)
                p = chunks[0][1][b0 * cs:b1 * cs].reshape(b1 - b0, cs, 4, 3)
            else:
                d = torch.stack(full_c[b0:b1]).reshape(b1 - b0, cs, 4, 3)
                p = torch.stack(full_p[b0:b1])
            out = self.model(d, p).reshape(-1, 255)
            # consecutive full windows of one chunk cover consecutive table rows: one copy per run (the first window of a chunk
            # starts with its cs - 1 pad rows; a tail window ends its run with its real rows)
            i = 0
            while i < b1 - b0:
                j = i + 1
                while (j < b1 - b0 and dst[b0 + j - 1][2] == cs and dst[b0 + j][1] == 0 and
                       dst[b0 + j][0] == dst[b0 + j - 1][0] + cs - dst[b0 + j - 1][1]):
                    j += 1
                r0, skip, _ = dst[b0 + i]
                rows = (j - 1 - i) * cs + dst[b0 + j - 1][2] - skip
                table[r0:r0 + rows] = out[i * cs + skip:i * cs + skip + rows]
                i = j

    def encode_ints(self, q, bin_num, n_points, t0=None, sequential=False, defer=False, front=None):
        """q: integer cloud (numpy / tensor int32 [P,3]) or the list of per-shell clouds of the multi-level form."""
        t0 = t0 or time.perf_counter()
        if front is None:
            qs = q if isinstance(q, (list, tuple)) else [q]
            qs = [(torch.from_numpy(np.ascontiguousarray(x, np.int32)) if isinstance(x, np.ndarray) else x).to(self.device) for x in qs]
            front = self._front(qs)
        chunks, sym, N = front
        table = torch.empty((N, 255), dtype=torch.float32, device=self.device)
        if sequential:
            row = 0
            for seq_ctx, seq_pos, n in chunks:
                self._sequential_rows(seq_ctx, seq_pos, n, table[row:row + n])
                row += n
        else:
            self._chunk_rows(chunks, table)
        lohi = native.softmax_cdf(table, sym)["lohi"]
        meta = dict(n_nodes=N, n_points=n_points, bin_num=bin_num, z_offset=0.0, n_levels=len(chunks), pos_mm=np.zeros((0, 2)),
                    level_sizes=[c[2] for c in chunks])
        if defer:
            return lohi, meta, (table, sym, chunks)
        stream = native.ac_encode_lohi(lohi.cpu().numpy())
        bits = 8 * len(stream)
        return dict(bytes=stream, bits=bits, bpp=bits / n_points, times=dict(total=time.perf_counter() - t0),
                    _debug=dict(table=table, sym_coded=sym), **meta)

    def encode_async(self, xyz):
        """Like encode(), but the D2H copy of the (c_low, c_high) pairs and the serial host range coder run on a worker thread
        behind an event on a side stream (the same scheme as FrameEncoder.encode_async): the caller can enqueue the next frame
        while this one is being coded.  `finish(handle)` blocks and returns the usual result dict."""
        t0 = time.perf_counter()
        if isinstance(xyz, np.ndarray):
            xyz = torch.from_numpy(np.ascontiguousarray(xyz, np.float32))
        if not hasattr(self, "_pool"):
            self._pool = ThreadPoolExecutor(max_workers=2)
            self._copy_stream = torch.cuda.Stream(device=self.device)
            self._front_stream = torch.cuda.Stream(device=self.device, priority=-1)
            # two lanes (with four frames in flight: 56.3 against 47.8 frames/s at L12, 23.1 against 21.0 at L14; three lanes: no better)
            self._lanes = [torch.cuda.Stream(device=self.device) for _ in range(max(1, int(os.environ.get("SCP_LANES_OCTATTN", "2"))))]
            self._lane_i = 0
        # stage G has small D2H syncs: on a high-priority side stream they wait for stage G only, not for the previous frame's
        # model kernels still queued on the main stream; consecutive frames run their model part on alternating streams (lanes,
        # see FrameEncoder.encode_async)
        caller = torch.cuda.current_stream(self.device)
        if len(self._lanes) > 1:
            main = self._lanes[self._lane_i % len(self._lanes)]
            self._lane_i += 1
            main.wait_stream(caller)
        else:
            main = caller
        fills0 = native.CACHE_FILLS
        with torch.cuda.stream(self._front_stream):
            self._front_stream.wait_stream(caller)
            xyz_dev = xyz.to(self.device, non_blocking=True)
            if not self.host_transform:
                q, bin_num = None, self.build_from_xyz(xyz_dev)
            else:
                q, bin_num = self.quantize(xyz_dev)
            front = self._front(q)
            ready = torch.cuda.Event()
            ready.record()
        main.wait_event(ready)
        with torch.cuda.stream(main):
            lohi, meta, keep = self.encode_ints(q, bin_num, xyz_dev.shape[0], t0, defer=True, front=front)
            done = torch.cuda.Event()
            done.record()
            if native.CACHE_FILLS != fills0:
                main.synchronize()
        host = torch.empty(lohi.shape, dtype=lohi.dtype, pin_memory=True)
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(done)
            host.copy_(lohi, non_blocking=True)
            lohi.record_stream(self._copy_stream)
            copied = torch.cuda.Event(blocking=True)      # the coder thread SLEEPS until the pairs have landed (a default event spins a core)
            copied.record()

        def work():
            _wait_event(copied)
            return native.ac_encode_lohi(host.numpy())

        return dict(future=self._pool.submit(work), meta=meta, t0=t0, keep=(keep, lohi, q, xyz_dev))

    def finish(self, h):
        stream = h["future"].result()
        bits = 8 * len(stream)
        return dict(bytes=stream, bits=bits, bpp=bits / h["meta"]["n_points"], times=dict(total=time.perf_counter() - h["t0"]), **h["meta"])

    def outfile(self, base, res):
        """encode.py:24 (`<base>.bin`) / encode_mullevel.py:68-72 (`<base>[_spher|_cylin]_<chunks>_<bin_num>_0.bin`)."""
        if not self.named:
            return base + ".bin"
        if self.spher:
            base += "_spher"
        elif self.cylin:
            base += "_cylin"
        return base + "_" + str(res["n_levels"]) + "_" + str(int(res["bin_num"])) + "_0.bin"
