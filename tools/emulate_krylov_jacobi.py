"""CPU (numpy) emulations behind two design decisions of the block Krylov truncation.  Nothing here
runs in the product or in a test; the GPU behaviour it predicts was then measured (tools/probe_sweep_conv.py).

  basis  [spectrum.npy]   rows of Krylov basis needed for the leading k triplets vs the block size b of the block Golub-Kahan
                          recurrence with full re-orthogonalisation, on diag(spectrum) -- the recurrence only sees the spectrum.
                          spectrum.npy: singular values of one unit's M = R^T Rt (tools/dump_unit_spectrum.py); default: a synthetic
                          spectrum with a steep head and a slowly decaying tail.
  jacobi                  sweeps of the one-sided block Jacobi SVD (32-row panels -> here 16, same structure) of the Ritz matrix T
                          with full 2b x 2b pair passes, or with cross-pair passes in all rounds of a sweep but the first.
  prerot                  the same after T was rotated by the exact SVD of its leading blocks (would a pre-computed partial SVD,
                          overlapped with the last block steps, shorten the extraction?  no: 9, 9, 8 sweeps instead of 10)."""
import sys, time
import numpy as np
rng = np.random.default_rng(1)


def synthetic(n, k):
    i = np.arange(n); g = 0.08
    return np.exp(-(16.1 / ((k + 1) ** g - 1)) * ((i + 1) ** g - 1))


def block_gk(s, b, steps, k=None, tol=None):
    """block Golub-Kahan on diag(s), full re-orthogonalisation.  Returns (T, E) after `steps`, or -- with tol -- the first
    (steps, rows) at which the residual estimate max_i |x_i^T E| of the k leading Ritz triplets is <= tol."""
    n = len(s)
    Us, Vs = [], [np.linalg.qr(rng.standard_normal((n, b)))[0]]
    for j in range(steps):
        W = s[:, None] * Vs[-1]
        if Us:
            Ua = np.concatenate(Us, 1)
            for _ in range(2): W -= Ua @ (Ua.T @ W)
        Us.append(np.linalg.qr(W)[0])
        Z = s[:, None] * Us[-1]
        Va = np.concatenate(Vs, 1)
        for _ in range(2): Z -= Va @ (Va.T @ Z)
        Vs.append(np.linalg.qr(Z)[0])
        m = (j + 1) * b
        if tol is None or m < k: continue
        Ua, Va = np.concatenate(Us, 1), np.concatenate(Vs[:-1], 1)
        T, E = Ua.T @ (s[:, None] * Va), Ua.T @ (s[:, None] * Vs[-1])
        X = np.linalg.svd(T)[0]
        est = np.linalg.norm(X[:, :k].T @ E, axis=1).max()
        print(f"  b={b:3d} step {j + 1:2d} basis {m:4d} ({m / k:.2f} k)  residual estimate {est:.2e}", flush=True)
        if est <= tol: return j + 1, m
    if tol is not None: return None
    Ua, Va = np.concatenate(Us, 1), np.concatenate(Vs[:-1], 1)
    return Ua.T @ (s[:, None] * Va), Ua.T @ (s[:, None] * Vs[-1])


def rr_rounds(m):
    out = []
    for r in range(m - 1):
        ps = [(m - 1, r)] + [((r + k) % (m - 1), (r - k + m - 1) % (m - 1)) for k in range(1, m // 2)]
        out.append(np.array([(min(a, b), max(a, b)) for a, b in ps]))
    return out


def jac_pass(G, rounds, tol, tau2):
    """one pass of two-sided Jacobi rotations over `rounds` (disjoint pairs per round) on symmetric G: J, eigenvalues sorted descending"""
    G = G.copy(); J = np.eye(G.shape[0])
    for ps in rounds:
        p, q = ps[:, 0], ps[:, 1]
        a, d, g = G[p, p], G[q, q], G[p, q]
        rot = (g != 0) & (g * g > tol * tol * np.maximum(np.abs(a * d), tau2 * tau2))
        if not rot.any(): continue
        dd, g2 = d - a, 2 * g
        h = np.hypot(dd, g2); den = dd + np.where(dd >= 0, h, -h)
        t = np.where(rot, g2 / np.where(den == 0, 1, den), 0.0)
        c = 1 / np.sqrt(1 + t * t); s = t * c
        for M_, axis in ((G, 1), (G, 0), (J, 1)):
            if axis == 1:
                A, B = M_[:, p].copy(), M_[:, q].copy(); M_[:, p] = c * A - s * B; M_[:, q] = s * A + c * B
            else:
                A, B = M_[p, :].copy(), M_[q, :].copy(); M_[p, :] = c[:, None] * A - s[:, None] * B; M_[q, :] = s[:, None] * A + c[:, None] * B
    return J[:, np.argsort(-np.diag(G), kind="stable")]


def block_jacobi_rows(X, b, ktop, mode="full", tol=1e-14, quad_exit=1e-9, max_sweeps=40):
    X = X.copy(); nb = X.shape[0] // b
    full = rr_rounds(2 * b); cross = [np.array([(k, b + (k + r) % b) for k in range(b)]) for r in range(b)]
    hist = []
    for _ in range(max_sweeps):
        nr = np.sqrt((X * X).sum(1))
        tau2 = max((1e-14 * np.sqrt((nr * nr).sum())) ** 2, np.sort(nr)[::-1][ktop - 1] ** 2)
        worst = 0.0
        for r in range(nb - 1):
            for k in range(nb // 2):
                i, j = (nb - 1, r % (nb - 1)) if k == 0 else ((r + k) % (nb - 1), (r - k + nb - 1) % (nb - 1))
                i, j = min(i, j), max(i, j)
                idx = np.r_[i * b:(i + 1) * b, j * b:(j + 1) * b]
                P = X[idx]; G = P @ P.T; dg = np.diag(G)
                off = np.abs(G - np.diag(dg)) / np.maximum(np.maximum(np.sqrt(np.abs(np.outer(dg, dg))), tau2), 1e-300)
                worst = max(worst, off.max())
                if off.max() > 0.1 * tol:
                    X[idx] = jac_pass(G, cross if (mode == "cross" and r > 0) else full, 0.1 * tol, tau2).T @ P
        hist.append(worst)
        if worst <= max(tol, quad_exit): break
    return hist, X


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "basis"
    if what == "basis":
        s = np.load(sys.argv[2]) if len(sys.argv) > 2 else synthetic(4608, 129)
        s = s / s[0]; k = 129
        for b, mx in ((16, 40), (32, 24), (64, 12), (128, 7)):
            print(b, block_gk(s, b, mx, k, 4e-14))
    else:
        n, k, b, steps, pb = 4096, 129, 32, 14, 16
        s = synthetic(n, k); T, E = block_gk(s, b, steps)
        sv = np.linalg.svd(T, compute_uv=False)
        if what == "jacobi":
            for mode in ("full", "cross"):
                t0 = time.time(); h, Xf = block_jacobi_rows(T, pb, k, mode)
                err = np.abs(np.sort(np.sqrt((Xf * Xf).sum(1)))[::-1][:k] - sv[:k]).max()
                print(mode, "sweeps", len(h), " ".join(f"{x:.1e}" for x in h), "err", err, f"{time.time() - t0:.1f} s")
        else:
            for j1 in (8, 10, 12):
                m1 = j1 * b; U1, _, V1t = np.linalg.svd(T[:m1, :m1])
                L = np.eye(T.shape[0]); L[:m1, :m1] = U1.T; R = np.eye(T.shape[0]); R[:m1, :m1] = V1t.T
                h, _ = block_jacobi_rows(L @ T @ R, pb, k)
                print(f"leading {j1} of {steps} blocks pre-rotated: sweeps", len(h), " ".join(f"{x:.1e}" for x in h))
