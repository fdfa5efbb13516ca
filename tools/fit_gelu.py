#!/usr/bin/env python3
"""Fit of `gelu_sig5` (csrc/isp_common.h): GELU(x) = x * Phi(x) with Phi(x) ~ sigmoid(x * (c0 + c1 x^2 + c2 x^4)).

Solves for q(x) = logit(Phi(x)) / x (an even function, q(0) = sqrt(8/pi)) as a polynomial in x^2 by weighted least
squares on |x| <= 6 -- weight x * Phi'(x)-like, so that the error of x * sigmoid(x q(x)) itself is what is minimised --
followed by a few Gauss-Newton steps on the max-norm surrogate, and prints the coefficients and the maximum absolute
error of the resulting GELU over the real line (fp64).  CPU only; run once, the constants are pasted into the header."""
import numpy as np
from scipy.special import erf


def gelu(x):
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def model(c, x):
    x2 = np.minimum(x * x, 36.0)  # the kernel's clamp: the quartic turns negative at |x| = 11.1
    return x / (1.0 + np.exp(-x * (c[0] + x2 * (c[1] + x2 * c[2]))))


def main():
    x = np.linspace(-8.0, 8.0, 400001)
    x = x[np.abs(x) > 1e-6]
    phi = 0.5 * (1.0 + erf(x / np.sqrt(2.0)))
    inner = np.abs(x) < 5.0  # beyond that Phi saturates in fp64 and logit() is noise; the tails are checked below
    xi, pi = x[inner], np.clip(phi[inner], 1e-300, 1 - 1e-16)
    q = np.log(pi / (1.0 - pi)) / xi
    A = np.stack([np.ones_like(xi), xi ** 2, xi ** 4], 1)
    w = np.abs(xi) * pi * (1 - pi) * np.abs(xi)  # d(x sigmoid(x q))/dq = x^2 s (1 - s)
    c = np.linalg.lstsq(A * w[:, None], q * w, rcond=None)[0]
    for _ in range(50):  # iteratively re-weighted least squares towards the minimax fit
        r = model(c, x) - gelu(x)
        x2 = np.minimum(x * x, 36.0)
        s = 1.0 / (1.0 + np.exp(-x * (c[0] + x2 * (c[1] + x2 * c[2]))))
        J = (x * x * s * (1 - s))[:, None] * np.stack([np.ones_like(x), x2, x2 ** 2], 1)
        wt = (np.abs(r) / np.abs(r).max()) ** 2 + 1e-3
        dc = np.linalg.lstsq(J * wt[:, None], -r * wt, rcond=None)[0]
        c = c + 0.5 * dc
    err = np.abs(model(c, x) - gelu(x))
    print("c0, c1, c2 =", ", ".join(repr(float(v)) for v in c))
    print(f"max |gelu_sig5 - gelu| on [-8, 8]: {err.max():.3e} at x = {x[err.argmax()]:.3f}")
    shipped = np.array([1.595015725363722, 0.07401132856622687, -0.0007030391178699941])
    es = np.abs(model(shipped, x) - gelu(x))
    print(f"shipped constants: max error {es.max():.3e} at x = {x[es.argmax()]:.3f}")
    big = np.array([-1e4, -30.0, -12.0, 12.0, 30.0, 1e4])
    with np.errstate(over="ignore"):
        print("tails (must be 0):", model(shipped, big) - gelu(big))


if __name__ == "__main__":
    main()
