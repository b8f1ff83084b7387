"""Derives the rational approximation used by the device kernel for the Ewald real-space
correction force

    g(w) = -( erf(z)/z - (2/sqrt(pi)) exp(-z^2) ) / z^2 ,   w = z^2 = (beta r)^2

so that F/r = qq * ( int_bit / r^3 + beta^3 g(beta^2 r^2) ), the analytical-Ewald form the
reference GPU kernels use (nbnxn_cuda_kernel.cuh:449-450).  The fit is our own: a
least-squares rational P6(w)/Q4(w) on [0, WMAX] refined by Sanathanan-Koerner iteration in
double precision (scipy.special.erf as the truth).  Prints the coefficients as a C header.
"""
import numpy as np
from scipy.special import erf

WMAX = 16.0   # (beta*rc)^2 = 9.76 at ewald_rtol 1e-5, 12.0 at 1e-6


def g(w):
    w = np.asarray(w, dtype=np.float64)
    z = np.sqrt(np.maximum(w, 1e-300))
    out = -(erf(z) / z - 2 / np.sqrt(np.pi) * np.exp(-w)) / np.maximum(w, 1e-300)
    # series near 0: g = -2/sqrt(pi) * sum_k (-1)^k w^k (1/k! - 1/((2k+1)k!)) / ... use direct series
    small = w < 1e-3
    ws = w[small]
    # erf(z)/z = 2/sqrt(pi) sum (-1)^k w^k /(k!(2k+1)); exp(-w) = sum (-1)^k w^k/k!
    # difference/w = 2/sqrt(pi) sum_{k>=1} (-1)^k w^(k-1) (1/(k!(2k+1)) - 1/k!)
    s = np.zeros_like(ws)
    from math import factorial
    for k in range(1, 8):
        s += (-1) ** k * ws ** (k - 1) * (1 / (factorial(k) * (2 * k + 1)) - 1 / factorial(k))
    out[small] = -2 / np.sqrt(np.pi) * s
    return out


def v(w):
    """erf(z)/z, w = z^2: the Ewald real-space potential correction (energy kernels)."""
    w = np.asarray(w, dtype=np.float64)
    z = np.sqrt(np.maximum(w, 1e-300))
    out = erf(z) / z
    out[w < 1e-12] = 2 / np.sqrt(np.pi)
    return out


def fit(np_, nq_, n=4000, func=None):
    func = func or g
    # Chebyshev-distributed sample points
    t = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    w = 0.5 * WMAX * (t + 1)
    y = func(w)
    u = w / WMAX   # scaled variable for conditioning
    wt = np.ones_like(w)
    for it in range(30):
        A = np.concatenate([np.stack([u ** k for k in range(np_ + 1)], 1),
                            -y[:, None] * np.stack([u ** k for k in range(1, nq_ + 1)], 1)], 1)
        sol, *_ = np.linalg.lstsq(A * wt[:, None], y * wt, rcond=None)
        p, q = sol[:np_ + 1], np.concatenate([[1.0], sol[np_ + 1:]])
        Q = sum(q[k] * u ** k for k in range(nq_ + 1))
        wt = 1 / np.abs(Q)
    p = np.array([p[k] / WMAX ** k for k in range(np_ + 1)])
    q = np.array([q[k] / WMAX ** k for k in range(nq_ + 1)])
    return p, q


def jointfit(npf, npv, nq, n=4000, iters=40):
    """Force and potential corrections with ONE denominator: Pf(w)/Q(w) ~ g(w), Pv(w)/Q(w) ~ v(w),
    so that the energy kernels need a single reciprocal per pair.  Same linearised least
    squares + Sanathanan-Koerner reweighting as fit(), both residuals stacked."""
    t = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    w = 0.5 * WMAX * (t + 1)
    u = w / WMAX
    yf, yv = g(w), v(w)
    wt = np.ones_like(w)
    for it in range(iters):
        Uf = np.stack([u ** k for k in range(npf + 1)], 1)
        Uv = np.stack([u ** k for k in range(npv + 1)], 1)
        Uq = np.stack([u ** k for k in range(1, nq + 1)], 1)
        A = np.concatenate([np.concatenate([Uf, np.zeros_like(Uv), -yf[:, None] * Uq], 1) * wt[:, None],
                            np.concatenate([np.zeros_like(Uf), Uv, -yv[:, None] * Uq], 1) * wt[:, None]], 0)
        sol, *_ = np.linalg.lstsq(A, np.concatenate([yf * wt, yv * wt]), rcond=None)
        pf, pv = sol[:npf + 1], sol[npf + 1:npf + npv + 2]
        q = np.concatenate([[1.0], sol[npf + npv + 2:]])
        wt = 1 / np.abs(sum(q[k] * u ** k for k in range(nq + 1)))
    sc = lambda c: np.array([c[k] / WMAX ** k for k in range(len(c))])
    return sc(pf), sc(pv), sc(q)


def evalf32(p, q, w):
    w = w.astype(np.float32)
    P = np.float32(p[-1]) * np.ones_like(w)
    for c in p[-2::-1]:
        P = P * w + np.float32(c)
    Q = np.float32(q[-1]) * np.ones_like(w)
    for c in q[-2::-1]:
        Q = Q * w + np.float32(c)
    return P / Q


if __name__ == "__main__":
    p, q = fit(6, 4)
    w = np.linspace(0, WMAX, 200001)
    err64 = np.abs(sum(p[k] * w ** k for k in range(7)) / sum(q[k] * w ** k for k in range(5)) - g(w))
    err32 = np.abs(evalf32(p, q, w).astype(np.float64) - g(w))
    print("// max abs error double %.3e, float32 evaluation %.3e (|g(0)| = 0.7523)" % (err64.max(), err32.max()))
    print("// min Q on range: %.3e" % min(sum(q[k] * w ** k for k in range(5))))
    for k, c in enumerate(p):
        print("#define NBX_PMEF_P%d %.9ef" % (k, c))
    for k, c in enumerate(q):
        print("#define NBX_PMEF_Q%d %.9ef" % (k, c))
    p, q = fit(5, 5, func=v)
    err64 = np.abs(sum(p[k] * w ** k for k in range(6)) / sum(q[k] * w ** k for k in range(6)) - v(w))
    err32 = np.abs(evalf32(p, q, w).astype(np.float64) - v(w))
    print("// erf(z)/z: max abs error double %.3e, float32 evaluation %.3e (v(0) = 1.1284)" % (err64.max(), err32.max()))
    print("// min Q on range: %.3e" % min(sum(q[k] * w ** k for k in range(6))))
    for k, c in enumerate(p):
        print("#define NBX_PMEV_P%d %.9ef" % (k, c))
    for k, c in enumerate(q):
        print("#define NBX_PMEV_Q%d %.9ef" % (k, c))
    pf, pv, q = jointfit(6, 5, 5)
    ef = np.abs(evalf32(pf, q, w).astype(np.float64) - g(w)).max()
    ev = np.abs(evalf32(pv, q, w).astype(np.float64) - v(w)).max()
    print("// joint fit, common denominator: float32 evaluation max abs error force %.3e potential %.3e" % (ef, ev))
    print("// min Q on range: %.3e" % min(sum(q[k] * w ** k for k in range(6))))
    for k, c in enumerate(pf):
        print("#define NBX_PMEJ_F%d %.9ef" % (k, c))
    for k, c in enumerate(pv):
        print("#define NBX_PMEJ_V%d %.9ef" % (k, c))
    for k, c in enumerate(q):
        print("#define NBX_PMEJ_Q%d %.9ef" % (k, c))
