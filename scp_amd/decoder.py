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
import numpy as np
import torch

from . import native

KITTI = "kitti"


class FrameDecoder:
    def __init__(self, model, lidar_level=12, mullevel=False, polar=True, device=None):
        self.model = model
        self.lidar_level = lidar_level
        self.mullevel = mullevel
        self.polar = polar            # spherical / cylindrical: positions normalised with the .dat (min, max) pairs
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.context_size = model.cfg.model.context_size

    def _decode_window(self, dec, ctx, pos):
        """ctx uint8 [c,12] (own occupancy = 255 placeholder), pos f32 [c,3] -> int64 symbols [c] (device).
        Runs the SAME packed kernels as the encoder (a one-window plan): encoder and decoder must produce bit-identical
        integer CDFs, and every kernel on the path is deterministic per row / per window, independent of the batch."""
        from .models.packed import PackedPlan, ehem_phase1_packed, ehem_phase2_packed
        c = ctx.shape[0]
        plan = PackedPlan([c], device=self.device)
        prob1, st = ehem_phase1_packed(self.model, ctx, pos, plan)
        cdf = native.softmax_cdf(prob1.contiguous(), want_lohi=False, want_cdf=True)["cdf"].cpu().numpy()
        even = torch.from_numpy(dec.run(cdf).astype(np.int64)).to(self.device)
        sym = torch.empty(c, dtype=torch.int64, device=self.device)
        sym[0::2] = even
        if c > 1:
            Q0 = plan.d["a1map"].shape[0]
            po = torch.zeros(Q0, dtype=torch.int64, device=self.device)
            po[:even.shape[0]] = even                       # one window: its real rows are the first rows of the cross layout
            prob2 = ehem_phase2_packed(self.model, st, plan, po)
            cdf = native.softmax_cdf(prob2.contiguous(), want_lohi=False, want_cdf=True)["cdf"].cpu().numpy()
            sym[1::2] = torch.from_numpy(dec.run(cdf).astype(np.int64)).to(self.device)
        return sym

    def _decode_level(self, dec, ctx, pos):
        """All windows of one level.  The even-node logits of a window depend on ancestors only (ehem.py:92-115), so phase 1 runs
        ONCE for the whole level as a packed forward - bit-identical to one-window launches (every kernel is per-row / per-window
        deterministic: tests/test_gpu_e2e.py::test_full_frame_packed_forward_is_batch_invariant) and 10 - 50 x better at filling the
        GPU; phase 2 needs the window's decoded even symbols and the bitstream interleaves the windows (evens, odds, evens, ...),
        so it runs window by window on that window's slice of the phase-1 state (a one-window plan has exactly that layout)."""
        from .models.packed import PackedPlan, ehem_phase1_packed, ehem_phase2_packed
        cs = self.context_size
        n = ctx.shape[0]
        lengths = [min(cs, n - i) for i in range(0, n, cs)]
        if len(lengths) == 1:
            return self._decode_window(dec, ctx, pos)
        plan = PackedPlan(lengths, device=self.device)
        prob1, st = ehem_phase1_packed(self.model, ctx, pos, plan)
        cdf1 = native.softmax_cdf(prob1.contiguous(), want_lohi=False, want_cdf=True)["cdf"].cpu().numpy()
        sym = torch.empty(n, dtype=torch.int64, device=self.device)
        row0 = e0 = q0 = 0
        for c in lengths:
            ne = (c + 1) // 2
            qp = -(-((c + (c & 1)) // 2) // 512) * 512                     # cross-layout rows of this window (padded to 512)
            even = torch.from_numpy(dec.run(cdf1[e0:e0 + ne]).astype(np.int64)).to(self.device)
            sym[row0:row0 + c:2] = even
            if c > 1:
                pw = PackedPlan([c], device=self.device)
                stw = dict(a1=native.SplitAct(st["a1"].t[:, q0:q0 + qp], st["a1"].K), a2=st["a2"][q0:q0 + qp],
                           pre_occ=st["pre_occ"][q0:q0 + qp])
                po = torch.zeros(qp, dtype=torch.int64, device=self.device)
                po[:ne] = even
                prob2 = ehem_phase2_packed(self.model, stw, pw, po)
                cdf = native.softmax_cdf(prob2.contiguous(), want_lohi=False, want_cdf=True)["cdf"].cpu().numpy()
                sym[row0 + 1:row0 + c:2] = torch.from_numpy(dec.run(cdf).astype(np.int64)).to(self.device)
            row0 += c
            e0 += ne
            q0 += qp
        return sym

    def _decode_tree(self, dec, depth, pos_mm):
        """One octree: returns (codes per level as device uint8 tensors, leaf integer coordinates [U,3])."""
        dev = self.device
        # level 1: the root. columns of `anc`: (level, octant, occ) of (ggp, gp, p); pad = (0, 0, 255)
        anc = torch.tensor([[0, 0, 255] * 3], dtype=torch.int64, device=dev)
        octant = torch.ones(1, dtype=torch.int64, device=dev)
        pos = torch.zeros((1, 3), dtype=torch.int64, device=dev)
        codes = []
        for L in range(1, depth + 1):
            n = pos.shape[0]
            last = L == depth
            lv = min(L, self.lidar_level) if last else L                      # encode_dataset_ehem.py:86 clips the last chunk
            a = anc.clone()
            if last:
                a[:, 0::3] = torch.clamp(a[:, 0::3], max=self.lidar_level)
            own = torch.stack((torch.full((n,), lv, dtype=torch.int64, device=dev), octant,
                               torch.full((n,), 255, dtype=torch.int64, device=dev)), 1)
            ctx = torch.cat((a, own), 1).to(torch.uint8)
            if self.polar:
                mn, mx = float(pos_mm[L - 1][0]), float(pos_mm[L - 1][1])
                eps = 0.0 if (self.mullevel and last) else 1e-9
                posn = ((pos.double() - mn) / (mx - mn + eps)).float()
            else:
                posn = (pos.double() / float(2 ** depth)).float()
            rows = n - (1 if (self.mullevel and last) else 0)                  # the dropped last node is never coded
            sym = torch.full((n,), -1, dtype=torch.int64, device=dev)
            if rows > 0:
                sym[:rows] = self._decode_level(dec, ctx[:rows].contiguous(), posn[:rows].contiguous())
            occ = sym + 1                                                       # 1..255; 0 = unknown (dropped node)
            codes.append(occ.to(torch.uint8))
            # children in (parent, digit) order
            bits = ((occ[:, None] >> torch.arange(8, device=dev)[None]) & 1).bool()
            par, dig = torch.nonzero(bits, as_tuple=True)
            sh = depth - L
            cpos = pos[par] + torch.stack((((dig >> 2) & 1) << sh, ((dig >> 1) & 1) << sh, (dig & 1) << sh), 1)
            if last:
                return codes, cpos
            anc = torch.cat((anc[par][:, 3:], torch.stack((torch.full_like(par, L), octant[par], sym[par]), 1)), 1)
            octant = dig + 1
            pos = cpos

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
        for d in depths:
            out.append(self._decode_tree(dec, d, pos_mm[off:off + d] if self.polar else None))
            off += d
        return out
