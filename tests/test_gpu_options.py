"""GPU: the option surface of the engine (ctm_set_option, csrc/ctm_runtime.hip).  40 keys; each is named here with its default, is
accepted by the library, and -- where it selects a route or a kernel -- is exercised: the same small problems (a truncated SVD on a
slowly decaying spectrum through the block Krylov solver, a symmetric truncation with a warm workspace, a complex GEMM, a fused
projector unit) give the same answers with the option at its other value.  Keys the library no longer has (variants that lost their
measurement and were deleted in round 5) are refused."""
import numpy as np
import pytest
import torch
from helpers import dev

pytestmark = pytest.mark.gpu

# key: (default, another legal value or None when the key only tunes/instruments)
OPTIONS = {
    # accuracy / semantics
    "jacobi_tol": (1e-14, 2e-14), "jacobi_max_sweeps": (30, 40), "svd_null_tol": (1e-11, 1e-10), "rank_tol": (5e-13, 1e-12), "si_tol": (2e-14, 4e-14),
    "svd_abs_accuracy": (1, 0), "svd_polar": (1, 0),
    # routes
    "si_enable": (1, 0), "si_min_n": (256, 128), "si_max_iter": (40, 30), "lz_enable": (1, 0), "lz_min_k": (48, 64), "lz_block": (0, 64), "lz_block_c": (32, 64),
    "lz_async": (1, 0), "lz_local_project": (1, 0), "lz_verify_op": (0, 1), "jacobi_cross_only": (1, 0), "eigh_warm": (1, 0), "eigh_orth_iter": (1, 0),
    "eigh_orth_double": (2, 0), "proj_from_krylov": (1, 0), "use_layer2": (1, 0), "gemm_fast": (1, 0), "xgemm_stack_rows": (1, 0),
    # round 6: Ritz extraction warm started from the unit's previous one, orientation of returned vectors follows the previous decomposition,
    # two Cholesky-QR passes where the unit's history allows (tests/test_gpu_stationary.py / test_gpu_iterative.py drive them on sweeps)
    "ritz_warm": (1, 0), "sign_follow": (1, 0), "lz_two_pass": (1, 0),
    # stationary fast path (tests/test_gpu_stationary.py drives it)
    "warm_accept_tol": (0.0, None), "warm_try_factor": (1e-4, None), "warm_accept_max_run": (32, None),
    # row-block GEMM epilogues (tests/test_gpu_gemm_rows.py drives them)
    "rows_fused_reduce": (1, 0), "rows_kernel_min_m": (1, 1000), "rows_kernel_min_m_kc": (1, 1000), "rows_min_klen": (576, 256), "rows_target_wgs": (512, 256),
    # instrumentation
    "gemm_timing": (0, None), "timing_min_flops": (5e9, None), "profile": (0, None), "jacobi_verbose": (0, None),
}
REMOVED = ["jacobi_persist", "jacobi_rot_apply", "heavy_serial", "heavy_min_flops", "lz_jacobi_block", "eig64_bpt", "eig64_pingpong", "layer2_reg",
           "gemm_strip", "rows_quantise", "lz_first", "lz_stride", "splitk_target_wgs", "no_such_option"]


def _problems(eng):
    rng = np.random.default_rng(5)
    out = {}
    # (1) slowly decaying spectrum, n = 768, chi = 64: the leading-k iteration hands over to the block Krylov solver
    n, chi = 768, 64
    U, _ = np.linalg.qr(rng.standard_normal((n, n))); V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    s = np.concatenate([[1.0], 1e-3 * 0.992 ** np.arange(n - 1)])
    M = (U * s) @ V.T
    Ug, Sg, Vg = eng.truncated_svd(dev(M), chi, eng.cfg(keep_multiplets=False))
    out["svd_S"] = Sg.cpu().numpy() / s[0]
    out["svd_resid"] = np.array([float((dev(M) @ Vg - Ug * Sg).abs().max() / s[0])])
    # (2) symmetric truncation, twice with a warm workspace (second call: restart / orthogonal iteration routes)
    lam = 0.9 ** np.arange(n) * np.where(np.arange(n) % 3 == 1, -1.0, 1.0)
    A = (U * lam) @ U.T
    E = rng.standard_normal((n, n)); E = (E + E.T) / np.linalg.norm(E + E.T, 2)
    basis = eng.warm_basis_c4v(48, n)
    eng.truncated_eigh(dev(A), 48, basis=basis)
    D, W = eng.truncated_eigh(dev(A + 1e-5 * E), 48, basis=basis)
    out["eigh_D"] = D.cpu().numpy()
    # (3) complex GEMM with 48 rows against a long K (stacked-row route) and a plain aligned real GEMM
    X = rng.standard_normal((48, 2048)) + 1j * rng.standard_normal((48, 2048)); Y = rng.standard_normal((2048, 1024)) + 1j * rng.standard_normal((2048, 1024))
    out["cgemm"] = eng.gemm(dev(X), dev(Y)).cpu().numpy()
    P = rng.standard_normal((256, 512)); Q = rng.standard_normal((512, 384))
    out["gemm"] = eng.gemm(dev(P), dev(Q)).cpu().numpy()
    R32 = rng.standard_normal((32, 4096)); Bg = rng.standard_normal((4096, 1024))
    out["rows"] = eng.gemm(dev(R32), dev(Bg)).cpu().numpy()
    # (4) a full SVD with vectors (the differentiable route's decomposition)
    F = rng.standard_normal((160, 160))
    Uf, Sf, Vf = eng.truncated_svd(dev(F), 160, eng.cfg(keep_multiplets=False))
    out["full_S"] = Sf.cpu().numpy()
    out["full_recon"] = np.array([float(((Uf * Sf) @ Vf.T - dev(F)).abs().max())])
    return out


TOL = {"svd_S": 1e-12, "svd_resid": 1.0, "eigh_D": 1e-11, "cgemm": 1e-10, "gemm": 1e-10, "rows": 1e-10, "full_S": 1e-11, "full_recon": 1.0}


def test_every_option_is_accepted_and_removed_ones_are_refused(eng):
    import _native
    assert len(OPTIONS) == 40
    for k, (default, _) in OPTIONS.items():
        eng.set_option(k, default)
    for k in REMOVED:
        with pytest.raises(_native.NativeError):
            eng.set_option(k, 0)
        eng._options.pop(k, None)


@pytest.fixture(scope="module")
def reference(eng):
    for k, (default, _) in OPTIONS.items():
        eng.set_option(k, default)
    return _problems(eng)


@pytest.mark.parametrize("key", [k for k, (_, other) in OPTIONS.items() if other is not None])
def test_option_at_its_other_value_gives_the_same_answers(eng, reference, key):
    default, other = OPTIONS[key]
    try:
        eng.set_option(key, other)
        got = _problems(eng)
    finally:
        eng.set_option(key, default)
    for name, ref in reference.items():
        if TOL[name] >= 1.0:
            assert got[name][0] < 1e-10, (key, name, got[name])             # residual-type entries: small in every variant
        else:
            assert np.abs(got[name] - ref).max() <= TOL[name] * max(1.0, np.abs(ref).max()), (key, name)
