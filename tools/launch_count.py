#!/usr/bin/env python3
"""Steady-state launches per frame by kernel: the difference of two rocprofv3 --kernel-trace --stats runs of bench.py with different --steps
(everything one-time - weight preparation, warm-up, the roofline measurement frames - cancels).    python tools/launch_count.py A.csv B.csv stepsA stepsB"""
import csv, sys
a, b, na, nb = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
def load(f):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
A, B = load(a), load(b)
rows = []
for k in set(A) | set(B):
    ca, ta = A.get(k, (0, 0.0)); cb, tb = B.get(k, (0, 0.0))
    rows.append((k, (cb - ca) / (nb - na), (tb - ta) / (nb - na) / 1e6))
rows.sort(key=lambda r: -r[1])
own = lambda k: not (k.startswith("void at::native") or "rocclr" in k or k.startswith("void (anonymous namespace)::elementwise") or "rocblas" in k or "at::native" in k)
tot = sum(r[1] for r in rows); t_own = sum(r[1] for r in rows if own(r[0])); ms = sum(r[2] for r in rows)
print(f"# launches per frame, steady state (difference of a {na}-frame and a {nb}-frame bench run): {tot:.1f} in all, {t_own:.1f} of libscp_hip.so, "
      f"{tot - t_own:.1f} torch / runtime (at::native, rocclr copies and fills); kernel time {ms:.2f} ms per frame")
print("| kernel | launches / frame | ms / frame |\n|---|---|---|")
for k, c, t in rows:
    if abs(c) >= 0.05:
        print(f"| `{k[:110]}` | {c:.1f} | {t:.3f} |")
