"""CPU (numpy) emulation for the NEXT step of the C4v truncation while the environment moves:
how many applications of the enlarged corner A_t (n = chi D^2) does a solver need per sweep, started from the previous sweep's kept
eigenvectors, to bring the kk = chi + 8 leading |lambda| pairs to the product's acceptance threshold (residual <= 2e-14 |lambda_0|)?

Nothing here runs in the product or in a test.  The CTM sweep is restated in numpy for this tool only (one-site C4v move: enlarged
corner, eigendecomposition truncated by |lambda|, half-row tensor absorbed and symmetrised, inf-norm normalisation), on the signed
random state of the bench (bench.synth_sites("c4v", 4, signed=True), chi = 64).

  orth     what csrc/eigh.hip:eigh_orth_iter does: Q <- orth(Q A), p = 128 rows (kk previous vectors + pseudo-random guard rows),
           one Rayleigh-Ritz when the residual passes.  (The product measured 24 ... 4 applications over the 22 moving sweeps.)
  krylov   block Krylov from the same start: Rayleigh-Ritz on span[Q, Q A, ..., Q A^m], fully re-orthogonalised, block = the kk previous
           vectors only (no guard rows) or the 128 rows.
  restarted  block Krylov of m applications, Rayleigh-Ritz on (m + 1) kk rows, restarted from its kk Ritz vectors.
  lobpcg   locally optimal block iteration: Rayleigh-Ritz on [X, R, P] (current Ritz vectors, their residuals, previous directions),
           one application per iteration (A R; A X and A P follow by linearity), block = kk.

usage: emulate_c4v_moving.py [nsweeps=16]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "peps-torch_amd")); sys.path.insert(0, ROOT)
from bench import synth_sites

CHI, D = 64, 4
TOL = 2e-14
rng = np.random.default_rng(7)


def enlarged_corner(a, C, T):
    Tv = T.reshape(CHI, CHI, D, D)
    x = np.einsum('xy,cyuU->xcuU', C, Tv)
    x = np.einsum('xcuU,xelL->cuUelL', x, Tv)
    x = np.einsum('cuUelL,suldr->cUeLsdr', x, a)
    x = np.einsum('cUeLsdr,sULDR->edDcrR', x, a.conj())
    n = CHI * D * D
    A = x.reshape(n, n)
    return 0.5 * (A + A.T)


def top_pairs(A, k):
    w, v = np.linalg.eigh(A)
    o = np.argsort(-np.abs(w))[:k]
    return w[o], v[:, o]


def move(a, C, T, Pprev=None):
    A = enlarged_corner(a, C, T)
    w, P = top_pairs(A, CHI)
    if Pprev is not None:                      # the product returns its vectors in the gauge of the previous call (sign of <p_i, p_i_prev>)
        sg = np.sign(np.sum(P * Pprev, 0)); sg[sg == 0] = 1.0
        P = P * sg
    Pv = P.reshape(CHI, D, D, CHI)
    Tv = T.reshape(CHI, CHI, D, D)
    y = np.einsum('xuUi,xelL->uUielL', Pv, Tv)
    y = np.einsum('uUielL,suldr->UieLsdr', y, a)
    y = np.einsum('UieLsdr,sULDR->iedDrR', y, a.conj())
    nT = np.einsum('iedDrR,edDj->ijrR', y, Pv.conj()).reshape(CHI, CHI, D * D)
    nT = 0.5 * (nT + nT.transpose(1, 0, 2))
    nC = np.diag(w) / abs(w[0])
    return A, nC, nT / np.abs(nT).max(), P


def init_env(a):
    c = np.einsum('mijef,mijab->eafb', a, a).reshape(D * D, D * D); c = c / np.abs(c).max()
    w, U = top_pairs(0.5 * (c + c.T), D * D)
    C = np.zeros((CHI, CHI)); C[:D * D, :D * D] = np.diag(w)
    t = np.einsum('meifg,maibc->eafbgc', a, a).reshape(D * D, D * D, D * D); t = t / np.abs(t).max()
    t = np.einsum('ai,abs,bj->ijs', U, t, U)
    T = np.zeros((CHI, CHI, D * D)); T[:D * D, :D * D, :] = t
    return C, T


def worst_residual(A, X, kk):
    """Rayleigh-Ritz of A on the orthonormal columns X; max residual of the kk leading |theta| pairs, relative to |theta_0|."""
    H = X.T @ (A @ X)
    th, Z = np.linalg.eigh(0.5 * (H + H.T))
    o = np.argsort(-np.abs(th))[:kk]
    V = X @ Z[:, o]
    R = A @ V - V * th[o]
    return np.linalg.norm(R, axis=0).max() / abs(th[o][0]), V, th[o], R


def orth(X):
    return np.linalg.qr(X)[0]


def run_orth(A, V0, kk, p=128, max_it=40):
    n = A.shape[0]
    Q = orth(np.concatenate([V0, rng.standard_normal((n, p - kk))], 1))
    for it in range(1, max_it + 1):
        Q = orth(A @ Q)
        if it >= 3 and worst_residual(A, Q, kk)[0] <= TOL:
            return it + 1                      # (+1: the product Q A the Rayleigh-Ritz itself needs)
    return None


def run_krylov(A, V0, kk, p, max_it=20):
    n = A.shape[0]
    Q = V0 if p == kk else np.concatenate([V0, rng.standard_normal((n, p - kk))], 1)
    Q = orth(Q)
    K = Q
    for it in range(1, max_it + 1):
        W = A @ Q
        for _ in range(2): W -= K @ (K.T @ W)
        Q = orth(W)
        K = np.concatenate([K, Q], 1)
        if worst_residual(A, K, kk)[0] <= TOL:
            return it + 1, K.shape[1]           # applications (the last block needs its product for the Rayleigh-Ritz), rows of the RR
    return None, K.shape[1]


def run_lobpcg(A, V0, kk, max_it=40):
    X = orth(V0)
    P = None
    for it in range(max_it + 1):
        r, X, th, R = worst_residual(A, X, kk)          # (X has kk columns: this rotates it to its Ritz vectors)
        if r <= TOL:
            return it + 1                                # applications: A X at the start, then A W once per iteration
        W = R - X @ (X.T @ R)
        S = orth(np.concatenate([X, orth(W)] + ([P] if P is not None else []), 1))
        S = orth(S)
        Xn = worst_residual(A, S, kk)[1]
        Pn = Xn - X @ (X.T @ Xn)
        P = orth(Pn)
        X = Xn
    return None


def run_restarted(A, V0, kk, m, max_cycles=12):
    """Block Krylov of m applications per cycle from the kk current vectors, Rayleigh-Ritz on (m + 1) kk rows, restart from its kk Ritz vectors."""
    X = orth(V0)
    apps = 0
    for cyc in range(1, max_cycles + 1):
        K, Q = X, X
        for _ in range(m):
            W = A @ Q
            for _ in range(2): W -= K @ (K.T @ W)
            Q = orth(W); K = np.concatenate([K, Q], 1)
        apps += m + 1
        r, X, _, _ = worst_residual(A, K, kk)
        if r <= TOL:
            return apps, cyc
    return None, max_cycles


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    a = synth_sites("c4v", D, signed=True)[(0, 0)]
    C, T = init_env(a)
    kk = CHI + 8
    prev = None; Pp = None
    print("sweep  moved     |l_kk/l_0|  |l_129/l_kk|   orth(p=128)  krylov(b=kk) rows   krylov(b=128) rows   lobpcg(b=kk)   restarted m=2 (apps, RRs of 216)   m=3 (apps, RRs of 288)")
    for sw in range(1, ns + 1):
        A, C, T, Pp = move(a, C, T, Pp)
        if prev is not None:
            w_all = np.sort(np.abs(np.linalg.eigvalsh(A)))[::-1]
            moved = worst_residual(A, orth(prev), kk)[0]
            o = run_orth(A, prev, kk)
            k1, r1 = run_krylov(A, prev, kk, kk)
            k2, r2 = run_krylov(A, prev, kk, 128)
            l = run_lobpcg(A, prev, kk) if sw <= 4 else "-"
            r2a, r3a = run_restarted(A, prev, kk, 2), run_restarted(A, prev, kk, 3)
            print(f"{sw:4d}  {moved:9.2e}  {w_all[kk - 1] / w_all[0]:9.2e}  {w_all[128] / w_all[kk - 1]:9.3f}   {str(o):>8s}   {str(k1):>8s} {r1:6d}   {str(k2):>8s} {r2:6d}   {str(l):>8s}   {str(r2a):>14s}   {str(r3a):>14s}", flush=True)
        prev = top_pairs(A, kk)[1]


if __name__ == "__main__":
    main()
