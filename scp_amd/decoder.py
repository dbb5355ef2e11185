"""Frame decoder: bitstream -> occupancy codes -> quantised points, the inverse of scp_amd/encoder.py.

Counterpart of decode_ehem.py:56-189 / decode_ehem_mullevel.py:56-189.  The octree is regenerated breadth first: the nodes of
level L+1 are the set children of level L in (parent order, child digit) = Morton order, so context rows and node origins are
known before their occupancy symbol is decoded.  Per window of <= 8192 nodes the model runs in two phases (EHEM.decode):
even-node logits -> decode the even symbols -> odd-node logits given them -> decode the odd symbols; the integer CDFs come
from the same device kernel the encoder used (csrc/cdf.hip) and the symbols from the host range decoder (csrc/rangecoder.cpp).
Side information = what the reference stores: the `.bin` file name (levels, bin_num, z_offset) and the `.dat` (min, max) pairs.
Like the reference decoder, the last BFS node of every multi-level shell is not coded (Octree.py:259-262) and stays unknown.
Phase 1 (ancestors only) runs once per level for all its windows; phase 2 is sequential per window (the bitstream interleaves
them).  Decoding is not on the metric path.
"""
import json
import math
import os

import numpy as np
import torch

from . import native

KITTI = "kitti"
SIDECAR = ".scp.json"


def extract_info(binfile):
    """decode_ehem.py:20-27 / decode_ehem_mullevel.py:20-27: everything the reference's decoders know about a stream - coordinate
    system from the file name, (levels, bin_num, z_offset) = its last three `_` fields (integers: the encoder wrote
    `int(z_offset)`), and the per-level (min, max) pairs from `<binfile>.dat` for the polar systems."""
    name = os.path.basename(binfile)
    spher, cylin = "spher" in name, "cylin" in name
    n_levels, bin_num, z_offset = (int(x) for x in name[:-len(".bin")].split("_")[-3:])
    pos_mm = torch.load(binfile + ".dat").numpy() if (spher or cylin) else np.zeros((0, 2), np.float32)
    return spher, cylin, pos_mm, n_levels, bin_num, z_offset


def write_sidecar(outfile, enc, res, model_name):
    """`<outfile>.scp.json` - what the reference's two side-info carriers cannot hold (an extension; the `.bin` / `.dat` pair stays
    exactly the reference's): lidar level and dataset type (the reference decoder takes the level count for the lidar level,
    decode_ehem.py:218), each shell's own bin_num (the file name has the first shell's only, but every shell de-quantises its
    angles with its own), the un-truncated z offset, `quant` = per shell the (qs[3], offset[3]) the integers were made with when the encoder quantised the
    frame itself (None on the --preproc_path / encode_ints paths: the dataset rules of `shell_qs` then apply, and `--type obj`, whose
    per-axis-minimum offset exists nowhere else, cannot be de-quantised), and the numeric profile of the kernels that produced the CDFs."""
    side = dict(model=model_name, type=enc.data_type, lidar_level=int(enc.lidar_level), mullevel=bool(enc.mullevel),
                spher=bool(enc.spher), cylin=bool(enc.cylin), n_points=int(res["n_points"]), n_nodes=int(res["n_nodes"]),
                bin_nums=[float(b) for b in res.get("bin_nums", [res["bin_num"]])],
                z_offset=float(res["z_offset"]), quant=res.get("quant"), profile=native.numeric_profile(model_name, getattr(enc, "profile", None)))
    with open(outfile + SIDECAR, "w") as f:
        json.dump(side, f)
    return side


def read_sidecar(binfile):
    p = binfile + SIDECAR
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return json.load(f)


def shell_qs(data_type, lidar_level, mullevel):
    f = (lambda l: 400 / (2 ** l - 1)) if data_type == KITTI else (lambda l: 2 ** (18 - l))
    return [f(lidar_level + k) for k in range(3)] if mullevel else [f(lidar_level)]


def dequantise_leaves(leaves, qs, bin_num, z_offset, spher, cylin, data_type=KITTI):
    """Leaf integers -> Cartesian points, decode_ehem.py:237-253 (`pt_rec * qs + offset`, then spher2cart / cylin2cart,
    data_preprocess.py:186-229), float64 on the device."""
    from . import metrics
    if spher:
        q, off = [qs, 2 * math.pi / (bin_num - 1), math.pi / (bin_num - 1)], [0.0, 0.0, 0.0]
    elif cylin:
        q, off = [qs, 2 * math.pi / (bin_num - 1), qs], [0.0, 0.0, float(z_offset)]
    else:
        o = -200.0 if data_type == KITTI else -float(2 ** 17)
        q, off = [qs, qs, qs], [o, o, o]
    return metrics.dequantize(leaves, q, off, spher=spher, cylin=cylin)


def decode_file(binfile, model, lidar_level=None, data_type=None, mullevel=False, device=None, profile=None):
    """A stream file written by the encode CLIs -> dict(codes per shell, leaves per shell, points [U,3] float64 Cartesian).
    Side information exactly as the reference's decoders take it (`extract_info`); the `.scp.json` written next to the stream
    supplies what that cannot carry.  Without it the reference's own rules apply: lidar level = the level count (decode_ehem.py:218),
    integer z offset, and - for the two outer shells - bin numbers extrapolated from the first shell's."""
    spher, cylin, pos_mm, n_levels, bin_num, z_offset = extract_info(binfile)
    side = read_sidecar(binfile)
    if side is not None:
        prof = native.numeric_profile(side.get("model", "EHEM"), profile)
        if side["profile"] != prof:
            raise native.ScpError(f"{binfile}: coded under numeric profile {side['profile']!r}, this process runs {prof!r}: the "
                                  "integer CDFs would differ and the range decoder would desynchronise")
        lidar_level = side["lidar_level"] if lidar_level is None else lidar_level
        data_type = side["type"] if data_type is None else data_type
        z_offset = side["z_offset"]
    data_type = data_type or ("ford" if "ford" in binfile else KITTI)                       # decode_ehem.py:241
    if lidar_level is None:
        lidar_level = n_levels // 3 - 1 if mullevel else n_levels                            # the reference's rule
    qs = shell_qs(data_type, lidar_level, mullevel)
    if side is not None and len(side["bin_nums"]) == len(qs):
        bins = side["bin_nums"]
    else:
        bins = [bin_num] + [round((bin_num - 1) * qs[0] / q) + 1 for q in qs[1:]]
    dec = FrameDecoder(model, lidar_level, mullevel=mullevel, polar=spher or cylin, device=device, profile=profile)
    with open(binfile, "rb") as f:
        stream = f.read()
    shells = dec.decode(stream, n_levels, pos_mm)
    # KITTI / Ford: the reference decoder's own rule (steps from the integer bin_num in float64, decode_ehem.py:237-249) - the cloud it
    # would write.  obj: there is no rule (the offset is the frame's per-axis minimum): the sidecar's `quant` entry, or nothing.
    quant = side.get("quant") if side is not None else None
    if data_type == "obj" and quant is not None and len(quant) == len(shells):
        from . import metrics
        pts = [metrics.dequantize(lv, qd["qs"], qd["offset"], spher=spher, cylin=cylin) for (_, lv), qd in zip(shells, quant)]
    elif data_type == "obj":
        raise native.ScpError(f"{binfile}: --type obj streams are quantised with the frame's per-axis minimum as offset "
                              "(data_preprocess.py:31-37), which only the sidecar's `quant` entry records - it is missing here")
    else:
        pts = [dequantise_leaves(lv, q, b, z_offset, spher, cylin, data_type) for (_, lv), q, b in zip(shells, qs, bins)]
    return dict(codes=[torch.cat(c) for c, _ in shells], leaves=[lv for _, lv in shells], points=torch.cat(pts),
                spher=spher, cylin=cylin, n_levels=n_levels, bin_num=bin_num, z_offset=z_offset, lidar_level=lidar_level)


class FrameDecoder:
    def __init__(self, model, lidar_level=12, mullevel=False, polar=True, device=None, profile=None):
        self.model = model
        self.profile = profile        # native.NumericProfile: must be the one the stream was coded under (None: process default)
        self.lidar_level = lidar_level
        self.mullevel = mullevel
        self.polar = polar            # spherical / cylindrical: positions normalised with the .dat (min, max) pairs
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.context_size = model.cfg.model.context_size
        self.stats = None             # set to {} to collect wall seconds per stage (adds a device synchronisation per stamp: bench.py --decode)
        self._plans = {}              # one-window plans by window length (index maps depend on the length only): built once, reused
        self._pin = [None, None]      # pinned staging of the CDF rows (grown on demand)
        self._pin_po = None           # pinned staging of a window's even symbols on their way up
        self._side = None             # side stream of ehem_phase2_prepare (created on first use, on the decoding thread's device)
        self.prepare_ahead = os.environ.get("SCP_DEC_PREP", "1") != "0"       # A/B switch: 0 = everything of phase 2 after the even symbols

    def _stamp(self, key, t0):
        if self.stats is None:
            return t0
        import time
        torch.cuda.synchronize()
        t = time.perf_counter()
        self.stats[key] = self.stats.get(key, 0.0) + (t - t0)
        return t

    def _t0(self):
        if self.stats is None:
            return 0.0
        import time
        torch.cuda.synchronize()
        return time.perf_counter()

    def _plan1(self, c):
        """The packed plan of ONE window of c nodes (cached: a frame has 55 windows of 8192 nodes and the plan is a function of c alone)."""
        from .models.packed import PackedPlan
        p = self._plans.get(c)
        if p is None:
            if len(self._plans) > 64:
                self._plans.clear()
            p = self._plans[c] = PackedPlan([c], device=self.device)
        return p

    def _cdf_to_host(self, cdf_dev, then=None, slot=0):
        """int16 CDF rows -> numpy, through a pinned buffer with an asynchronous copy: `then()` (launches for the side stream) runs on the host while
        the GPU finishes phase 1 and the copy; returns (rows, then's result).  The rows are a view of pinned buffer `slot` (0: a level's phase-1 rows,
        1: a window's phase-2 rows): consumed before the slot's next use."""
        n = cdf_dev.numel()
        if self._pin[slot] is None or self._pin[slot].numel() < n:
            self._pin[slot] = torch.empty(max(n, 1 << 20), dtype=torch.int16, pin_memory=True)
        host = self._pin[slot][:n].view(cdf_dev.shape)
        host.copy_(cdf_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        r = then() if then is not None else None
        ev.synchronize()
        return host.numpy(), r

    def _prepare(self, st, plan):
        """Start the symbol-independent part of phase 2 (models/packed.py: ehem_phase2_prepare) on the side stream, behind everything the current
        stream holds so far (phase 1); returns (prep, event).  The host decodes the even symbols meanwhile; phase 2 waits for the event."""
        from .models.packed import ehem_phase2_prepare
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            prep = ehem_phase2_prepare(self.model, st, plan)
            done = torch.cuda.Event()
            done.record(self._side)
        return prep, done

    def _even_to_device(self, even, qp):
        """The decoded even symbols of a window (int16 numpy [ne]) -> int64 device tensor [qp] in the cross layout (zeros behind the real rows):
        one asynchronous copy out of a pinned buffer.  (The buffer's previous contents were consumed: a window's phase-2 rows are read back,
        with a synchronisation, before the next window gets here.)"""
        if self._pin_po is None or self._pin_po.numel() < qp:
            self._pin_po = torch.empty(max(qp, 8192), dtype=torch.int64, pin_memory=True)
        h = self._pin_po[:qp].numpy()
        h[:even.shape[0]] = even
        h[even.shape[0]:] = 0
        po = torch.empty(qp, dtype=torch.int64, device=self.device)
        po.copy_(self._pin_po[:qp], non_blocking=True)
        return po

    def _decode_window(self, dec, ctx, pos, sym):
        """ctx uint8 [c,12] (own occupancy = 255 placeholder), pos f32 [c,3]; the symbols go to sym[:c] (host int64).
        Runs the SAME packed kernels as the encoder (a one-window plan): encoder and decoder must produce bit-identical
        integer CDFs, and every kernel on the path is deterministic per row / per window, independent of the batch."""
        from .models.packed import PackedPlan, ehem_phase1_packed, ehem_phase2_packed
        c = ctx.shape[0]
        t = self._t0()
        plan = self._plan1(c)
        prob1, st = ehem_phase1_packed(self.model, ctx, pos, plan)
        t = self._stamp("phase1_model", t)
        cdf_dev = native.softmax_cdf(prob1.contiguous(), want_lohi=False, want_cdf=True)["cdf"]
        cdf, prep = self._cdf_to_host(cdf_dev, (lambda: self._prepare(st, plan)) if (c > 1 and self.prepare_ahead) else None)
        t = self._stamp("cdf_d2h", t)
        even = dec.run(cdf)
        t = self._stamp("range_decoder", t)
        sym[0:c:2] = even
        if c > 1:
            po = self._even_to_device(even, plan.d["a1map"].shape[0])   # one window: its real rows are the first rows of the cross layout
            t = self._stamp("index_ops", t)
            if prep is not None:
                torch.cuda.current_stream(self.device).wait_event(prep[1])
            prob2 = ehem_phase2_packed(self.model, st, plan, po, prep=None if prep is None else prep[0])
            t = self._stamp("phase2_model", t)
            cdf, _ = self._cdf_to_host(native.softmax_cdf(prob2.contiguous(), want_lohi=False, want_cdf=True)["cdf"], slot=1)
            t = self._stamp("cdf_d2h", t)
            sym[1:c:2] = dec.run(cdf)
            t = self._stamp("range_decoder", t)

    def _decode_level(self, dec, ctx, pos, n_total):
        """All windows of one level -> int64 device tensor [n_total]: the symbols of the ctx.shape[0] coded nodes, -1 behind them (the dropped last
        node of a multi-level shell).  The even-node logits of a window depend on ancestors only (ehem.py:92-115), so phase 1 runs
        ONCE for the whole level as a packed forward - bit-identical to one-window launches (every kernel is per-row / per-window
        deterministic: tests/test_gpu_e2e.py::test_full_frame_packed_forward_is_batch_invariant) and 10 - 50 x better at filling the
        GPU; phase 2 needs the window's decoded even symbols and the bitstream interleaves the windows (evens, odds, evens, ...),
        so it runs window by window on that window's slice of the phase-1 state (a one-window plan has exactly that layout).
        The decoded symbols stay on the host until the level is complete (one copy); a window sends only its even symbols up."""
        from .models.packed import PackedPlan, ehem_phase1_packed, ehem_phase2_packed, phase2_prep_window
        cs = self.context_size
        n = ctx.shape[0]
        sym = np.full(n_total, -1, np.int64)
        lengths = [min(cs, n - i) for i in range(0, n, cs)]
        if len(lengths) == 1:
            self._decode_window(dec, ctx, pos, sym)
            return torch.from_numpy(sym).to(self.device)
        t = self._t0()
        plan = PackedPlan(lengths, device=self.device)
        prob1, st = ehem_phase1_packed(self.model, ctx, pos, plan)
        t = self._stamp("phase1_model", t)
        cdf1_dev = native.softmax_cdf(prob1.contiguous(), want_lohi=False, want_cdf=True)["cdf"]
        # the level's query stream + pre_attn_mlp: one packed pass over all windows, launched while the CDF rows travel
        cdf1, prep = self._cdf_to_host(cdf1_dev, (lambda: self._prepare(st, plan)) if self.prepare_ahead else None)
        t = self._stamp("cdf_d2h", t)
        row0 = e0 = q0 = 0
        nst = len(self.model.swin_cross_transformer.layers)
        bases = [0] * nst                                                       # first row of the current window in every cross stage
        for c in lengths:
            ne = (c + 1) // 2
            qp = -(-((c + (c & 1)) // 2) // 512) * 512                     # cross-layout rows of this window (padded to 512)
            rows, L = [], (c + (c & 1)) // 2                                    # (models/packed.py: StageLayout - every stage pads a window to x512 rows)
            for _ in range(nst):
                rows.append(-(-L // 512) * 512)
                L = (L + 1) // 2
            even = dec.run(cdf1[e0:e0 + ne])
            t = self._stamp("range_decoder", t)
            sym[row0:row0 + c:2] = even
            if c > 1:
                pw = self._plan1(c)
                stw = dict(a1=st["a1"][q0:q0 + qp], a2=st["a2"][q0:q0 + qp],
                           pre_occ=st["pre_occ"][q0:q0 + qp])
                po = self._even_to_device(even, qp)
                t = self._stamp("index_ops", t)
                pwin = None
                if prep is not None:
                    if prep[1] is not None:
                        torch.cuda.current_stream(self.device).wait_event(prep[1])
                        prep = (prep[0], None)
                    pwin = phase2_prep_window(prep[0], bases, rows)
                prob2 = ehem_phase2_packed(self.model, stw, pw, po, prep=pwin)
                t = self._stamp("phase2_model", t)
                cdf, _ = self._cdf_to_host(native.softmax_cdf(prob2.contiguous(), want_lohi=False, want_cdf=True)["cdf"], slot=1)
                t = self._stamp("cdf_d2h", t)
                sym[row0 + 1:row0 + c:2] = dec.run(cdf)
                t = self._stamp("range_decoder", t)
            row0 += c
            e0 += ne
            q0 += qp
            bases = [b + r for b, r in zip(bases, rows)]
        return torch.from_numpy(sym).to(self.device)

    def _decode_tree(self, dec, depth, pos_mm):
        """One octree: returns (codes per level as device uint8 tensors, leaf integer coordinates [U,3]).
        Level state: pos int32 [n,3] node origins, anc uint8 [n,9] = (level, octant, symbol) of (ggp, gp, p) with pad = (0, 0, 255), octant uint8 [n];
        the children of a decoded level and the model inputs of the level they form come from one launch (native.decode_expand)."""
        dev = self.device
        anc = torch.tensor([[0, 0, 255] * 3], dtype=torch.uint8, device=dev)
        octant = torch.ones(1, dtype=torch.uint8, device=dev)
        pos = torch.zeros((1, 3), dtype=torch.int32, device=dev)

        def level_params(L):
            """(lv, ancestor level clamp, mn, den) of level L's inputs."""
            last = L == depth
            lv = min(L, self.lidar_level) if last else L                      # encode_dataset_ehem.py:86 clips the last chunk
            if self.polar:
                mn, mx = float(pos_mm[L - 1][0]), float(pos_mm[L - 1][1])
                eps = 0.0 if (self.mullevel and last) else 1e-9
                return lv, (self.lidar_level if last else 255), mn, mx - mn + eps
            return lv, (self.lidar_level if last else 255), 0.0, float(2 ** depth)

        # level 1: the root
        lv, clamp, mn, den = level_params(1)
        ctx = torch.tensor([[0, 0, 255] * 3 + [lv, 1, 255]], dtype=torch.uint8, device=dev)
        posn = ((pos.double() - mn) / den).float() if self.polar else (pos.double() / den).float()
        codes = []
        t = self._t0()
        for L in range(1, depth + 1):
            n = pos.shape[0]
            last = L == depth
            rows = n - (1 if (self.mullevel and last) else 0)                  # the dropped last node is never coded
            t = self._stamp("tree_expansion", t)
            sym = self._decode_level(dec, ctx[:rows], posn[:rows], n) if rows > 0 else torch.full((n,), -1, dtype=torch.int64, device=dev)
            t = self._t0()
            # children in (parent, digit) order; occupancy 1..255, 0 = unknown (dropped node)
            if last:
                lvn, clamp, mn, den = 0, 255, 0.0, 1.0
            else:
                lvn, clamp, mn, den = level_params(L + 1)
            occ8, pos, anc, octant, ctx, posn = native.decode_expand(sym, pos, anc, octant, L, depth - L, lvn, clamp, self.polar and not last, mn, den)
            codes.append(occ8)
            if last:
                return codes, pos.long()

    def decode(self, stream, n_levels, pos_mm):
        """stream: bytes; n_levels: total level count from the file name; pos_mm: [n_levels,2] array from the .dat file.
        Returns list of (codes_per_level, leaf_points int64 [U,3]) - one entry per shell."""
        dec = native.AcDecoder(stream)
        if self.mullevel:
            m = n_levels // 3                                                   # decode_ehem_mullevel.py:199
            depths = [m - 1, m, m + 1]
        else:
            depths = [n_levels]
        out, off = [], 0
        from . import ops
        with native.use_profile(self.profile), ops.frozen_weights():            # (the weights do not change inside a frame: validated once per frame)
            for d in depths:
                out.append(self._decode_tree(dec, d, pos_mm[off:off + d] if self.polar else None))
                off += d
        return out
