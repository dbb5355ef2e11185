#!/usr/bin/env python3
"""Coefficients of the round-5 GELU (csrc/scp_internal.h: scp_gelu_scaled) and its error budget.

    GELU(y) = y Phi(y) = max(y, 0) - T(|y|),      T(a) = a Phi(-a) ~= a exp(-beta a^2) / P_n(a)

exp(-a^2 / 2) / Phi(-a) (the inverse Mills ratio's relative) is smooth and grows linearly, so a low-degree polynomial P_n follows it,
and letting the exponent's scale beta float buys a factor 2.4 of accuracy at n = 4.  The fit is minimax (Lawson re-weighting of a
Levenberg-Marquardt least-squares fit) on [0, 6.5]; P_4 has no root on the real axis.  The kernel works in y' = s y with
s = sqrt(beta log2 e), so exp(-beta a^2) is v_exp_f32(-y'^2).  Printed: the constants of scp_internal.h and the maximum absolute
error of a float32 evaluation (every operation rounded, FMA emulated in float64) against the float64 erf form on |y| <= 20 - next to
the degree-12 erf polynomial of rounds 1 - 4.            python tools/fit_gelu.py [n]"""
import sys
import numpy as np
from scipy import special, optimize

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
a = np.concatenate([np.linspace(0, 2, 8001), np.linspace(2, 6.5, 12001)[1:]])
phic = 0.5 * special.erfc(a / np.sqrt(2))
x = np.concatenate([np.polyfit(a, np.exp(-a * a / 2) / phic, n)[::-1], [0.5]])


def err(x):
    return a * (np.exp(-x[n + 1] * a * a) / np.polyval(x[:n + 1][::-1], a) - phic)


w, best = np.ones_like(a), None
for it in range(300):
    x = optimize.least_squares(lambda c: np.sqrt(w) * err(c), x, method="lm", xtol=1e-15, ftol=1e-15, max_nfev=4000).x
    e = np.abs(err(x))
    if best is None or e.max() < best[0]:
        best = (e.max(), x.copy())
    w = w * (e / e.max() + 1e-4)
    w /= w.mean()
m, x = best
c, beta = x[:n + 1], x[n + 1]
s = np.sqrt(beta * np.log2(np.e))
cs = (c / s ** np.arange(n + 1)).astype(np.float32)
aa = np.linspace(0, 1e3, 2000001)
print(f"n = {n}: minimax |error| of the activation {m:.3e}, beta = {beta:.7f}, min P on [0, 1000] = {np.polyval(c[::-1], aa).min():.4f}")
print(f"#define SCP_GELU_S {s:.9f}f\n#define SCP_GELU_INV_S {1 / s:.9f}f")
for i, v in enumerate(cs):
    print(f"#define SCP_GELU_C{i} {float(v)!r}f")

f32 = np.float32


def fma(p, q, r):
    return (p.astype(np.float64) * q.astype(np.float64) + r.astype(np.float64)).astype(f32)


def gelu_new(yp):
    ab = np.abs(yp)
    e = np.exp2((-(yp.astype(np.float64) * yp.astype(np.float64))).astype(f32).astype(np.float64)).astype(f32)
    p = fma(np.full_like(yp, cs[n]), ab, np.full_like(yp, cs[n - 1]))
    for i in range(n - 2, -1, -1):
        p = fma(p, ab, np.full_like(yp, cs[i]))
    r = (1.0 / p.astype(np.float64)).astype(f32)
    return fma(-(ab.astype(np.float64) * e).astype(f32), r, np.maximum(yp, f32(0)))


def gelu_r4(y):
    z = (y * f32(0.70710678118654752)).astype(f32)
    zc = np.minimum(np.abs(z), f32(3.5))
    u = fma((zc * zc).astype(f32), np.full_like(y, f32(2.0 / 12.25)), np.full_like(y, f32(-1)))
    cf = [1.480935152e-03, -3.987360327e-03, 4.474287011e-03, -7.227925849e-03, 1.704961757e-02, -3.003174999e-02, 4.501544287e-02,
          -6.477065166e-02, 8.840217622e-02, -1.146127499e-01, 1.467501802e-01, -2.007010379e-01, 4.038729840e-01]
    p = np.full_like(y, f32(cf[0]))
    for q in cf[1:]:
        p = fma(p, u, np.full_like(y, f32(q)))
    ee = np.copysign((p * zc).astype(f32), z)
    hy = (f32(0.5) * y).astype(f32)
    return fma(hy, ee, hy)


y = np.concatenate([np.linspace(-20, 20, 2000001), np.random.default_rng(0).normal(0, 1.5, 1000000)])
yp = (y * s).astype(f32)
yy = yp.astype(np.float64) / s
e_new = np.abs(gelu_new(yp).astype(np.float64) / s - 0.5 * yy * special.erfc(-yy / np.sqrt(2)))
yo = y.astype(f32)
e_old = np.abs(gelu_r4(yo).astype(np.float64) - 0.5 * yo.astype(np.float64) * special.erfc(-yo.astype(np.float64) / np.sqrt(2)))
print(f"float32 evaluation, |y| <= 20: this form {e_new.max():.3e} (at y = {yy[e_new.argmax()]:.2f}); degree-12 erf polynomial {e_old.max():.3e} "
      f"(at y = {yo[e_old.argmax()]:.2f}; {e_old[np.abs(yo) < 4].max():.3e} on |y| < 4)")
