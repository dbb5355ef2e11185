#!/usr/bin/env python3
"""CU-partitioned lanes: does an HBM-bound launch sequence of one frame co-run with an MFMA-bound launch of
another when each gets its own CU set (hipExtStreamCreateWithCUMask), compared with the same work back to back on the whole chip and
with two unmasked streams (what FrameEncoder's lanes do today)?

HBM side: k x layernorm_rows(split) over 590 848 rows (1.2 GB moved per launch).  MFMA side: one scp_swin_post_attn over the same rows.
    python tools/mb_cumask.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scp_amd import native  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def main():
    M = 590848
    dev = torch.device("cuda:0")
    L = native.lib()
    L.scp_rc_set_grid.argtypes = [ctypes.c_int32]
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((M, 256), generator=g).to(dev)
    o = native.split_rows((torch.randn((M, 256), generator=g) * 0.5).to(dev))
    gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    mk = lambda *s: (torch.randn(s, generator=g) * 0.05).to(dev)
    pw = native.PostAttnWeights(mk(256, 256), mk(256), gamma, beta, mk(1024, 256), mk(1024), mk(256, 1024), mk(256))
    y = torch.empty_like(x)
    K_LN = 10

    def hbm_side():
        for _ in range(K_LN):
            native.layernorm_rows(x, gamma, beta, 1e-5, split=True)

    def mfma_side():
        native.swin_post_attn(o, x, pw, out=y)

    def run(sa, sb, grid, reps=6):
        """ms for both sides to finish (launched together), and each side's own elapsed time"""
        L.scp_rc_set_grid(grid)
        res = []
        for _ in range(reps):
            torch.cuda.synchronize()
            e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(torch.cuda.current_stream())
            sa.wait_event(e0); sb.wait_event(e0)
            with torch.cuda.stream(sb):
                mfma_side(); eb.record(sb)
            with torch.cuda.stream(sa):
                hbm_side(); ea.record(sa)
            torch.cuda.synchronize()
            res.append((max(e0.elapsed_time(ea), e0.elapsed_time(eb)), e0.elapsed_time(ea), e0.elapsed_time(eb)))
        L.scp_rc_set_grid(0)
        res.sort()
        return res[len(res) // 2]

    cur = torch.cuda.current_stream()
    # alone, whole chip
    def alone(f, n=6):
        f(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): f()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    ta, tb = alone(hbm_side), alone(mfma_side)
    print(f"alone on 256 CUs: {K_LN} x layernorm_rows(split) {ta:.3f} ms, swin_post_attn {tb:.3f} ms, back to back {ta + tb:.3f} ms", flush=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    t = run(s1, s2, 0)
    print(f"two unmasked streams:                      both done {t[0]:.3f} ms (HBM side {t[1]:.3f}, MFMA side {t[2]:.3f})", flush=True)
    layouts = {
        "128 / 128, bits 0-127 | 128-255": (set(range(128)), set(range(128, 256)), 128),
        "128 / 128, bit % 8 < 4 | >= 4 (XCD halves if bits interleave over XCDs)": ({b for b in range(256) if b % 8 < 4}, {b for b in range(256) if b % 8 >= 4}, 128),
        "64 / 192, bits 0-63 | 64-255": (set(range(64)), set(range(64, 256)), 192),
        "64 / 192, bit % 8 < 2 | >= 2": ({b for b in range(256) if b % 8 < 2}, {b for b in range(256) if b % 8 >= 2}, 192),
    }
    for name, (ma, mb, grid) in layouts.items():
        sa, sb = masked_stream(ma), masked_stream(mb)
        # each side alone on its CU set
        L.scp_rc_set_grid(grid)
        with torch.cuda.stream(sa):
            a1 = alone(hbm_side)
        with torch.cuda.stream(sb):
            b1 = alone(mfma_side)
        L.scp_rc_set_grid(0)
        t = run(sa, sb, grid)
        print(f"{name}:\n    alone on its set: HBM side {a1:.3f} ms, MFMA side {b1:.3f} ms;  together: both done {t[0]:.3f} ms (HBM side {t[1]:.3f}, MFMA side {t[2]:.3f})", flush=True)


if __name__ == "__main__":
    main()
