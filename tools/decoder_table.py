#!/usr/bin/env python3
"""Kernels of a decode by launch count and average launch time, before / after: two rocprofv3 --kernel-trace --stats summaries of
`bench.py --decode --steps 2 --warmup 1` (1 encode + 4 decodes each).   python tools/decoder_table.py before.csv after.csv > profiles/<tag>_decoder_kernels.md"""
import csv, sys
def load(f):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f))}
A, B = load(sys.argv[1]), load(sys.argv[2])
ND = 4.0
print("# Kernels of one decode of the L16-m frame (rocprofv3 --kernel-trace --stats of `bench.py --decode --steps 2 --warmup 1`: one encode + four decodes; per-decode figures = totals / 4,")
print("# the encode's launches - about 190 - included).  before = the round-4 kernels (with round 5's range decoder), after = short launches split along the output channels")
print("# (rc_post_attn_wide_kernel, step groups of rc_ln_linear_kernel, small GEMM tiles).\n")
print("| kernel | launches / decode | before: avg us | ms / decode | after: avg us | ms / decode |")
print("|---|---|---|---|---|---|")
names = sorted(set(A) | set(B), key=lambda k: -(A.get(k, (0, 0, 0))[1] + B.get(k, (0, 0, 0))[1]))
ta = tb = 0.0
for k in names[:26]:
    a, b = A.get(k, (0, 0.0, 0.0)), B.get(k, (0, 0.0, 0.0))
    print(f"| `{k[:72]}` | {max(a[0], b[0]) / ND:.0f} | {a[2]:.1f} | {a[1] / ND:.1f} | {b[2]:.1f} | {b[1] / ND:.1f} |")
for k in names:
    ta += A.get(k, (0, 0.0, 0.0))[1]; tb += B.get(k, (0, 0.0, 0.0))[1]
print(f"\nKernel time per decode (all kernels, encode share included): before {ta / ND:.0f} ms, after {tb / ND:.0f} ms; launches per decode {sum(v[0] for v in A.values()) / ND:.0f} / {sum(v[0] for v in B.values()) / ND:.0f}.")
