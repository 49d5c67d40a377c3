#!/usr/bin/env python
"""bench.py -- CTM sweeps/s of the MI355X-native CTMRG engine (driver contract, see task prompt).

    python bench.py --gpus N --steps K --warmup W [--config NAME]

A "step" is one CTM sweep (`_ctmrg_iter` of the reference, ctm/generic/ctmrg.py:63-69: 4 directions x
lX-or-lY repeats x all sites, conv_check excluded) on a synthetic random iPEPS built exactly like the
reference scripts do (A ~ U[0,1), shape (2,D,D,D,D), A /= max|A|; env from init_from_ipeps_pbc).
With N > 1 the per-site units of every directional move are sharded over the N ranks (one MI355X each,
RCCL all-gather of projectors and of new C/T tensors) -- total work is fixed, so scaling is "strong".
Rank 0 prints ONE JSON line with the metric, the roofline of the dominant kernel (FP64-MFMA GEMM, timed
live with HIP events on the engine's stream) and a CPU baseline (numpy oracle on the host cores,
bounded sample).
"""
import argparse, json, math, os, sys, time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "peps-torch_amd"))
sys.path.insert(0, REPO)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # one hardware queue per stream of the concurrent units (see config.py)
import numpy as np
import torch

CONFIGS = {
    # name: (kind, D, chi, dtype)          BASELINE.json configs[1..4]
    "c4v_D4_chi64": ("c4v", 4, 64, "f64"),
    "c4v_D4_chi64_c128": ("c4v", 4, 64, "c128"),               # the complex C4v ansatz A1 + i A2 (ipeps_c4v.py:60-66)
    "generic_D4_chi64": ("generic", 4, 64, "f64"),
    "generic_D6_chi128": ("generic", 6, 128, "f64"),
    "generic_D8_chi256": ("generic", 8, 256, "f64"),
    "generic_D3_chi36": ("generic", 3, 36, "f64"),
    "generic_D4_chi64_c128": ("generic", 4, 64, "c128"),
    "generic_D6_chi128_c128": ("generic", 6, 128, "c128"),
    "generic_D8_chi384_c128": ("generic", 8, 384, "c128"),     # BASELINE.json configs[4] (quoted there on 8 GPUs)
}
DEFAULT_CONFIG = "generic_D8_chi256"       # the configuration BASELINE.json's north_star quotes the metric on (fits one GPU)
FP64_MFMA_PEAK_TFLOPS = 78.6               # MI355X FP64 matrix peak (v_mfma_f64_16x16x4_f64, 32 flop/clk/SIMD)


def synth_sites(kind, D, seed=1, dtype="f64", signed=False):
    """Random site tensors as the reference scripts build them: entries U[0,1) (re and im parts for complex128), A /= max|A|.
    signed: entries U(-1,1) -- drawn from the same stream, re and im parts centred SEPARATELY before the normalisation (round 3 centred
    the normalised complex tensor with 2 A - (1 + i): |A| <= 0.71 after A /= max|A|, so the parts kept a mean of -0.3 and the
    "signed" complex128 state was as low-rank as the positive one: 13 corner values above 1e-8)."""
    rng = np.random.default_rng(seed)
    if signed:
        base = rng.random
        class _R:                                   # same stream, centred
            @staticmethod
            def random(shape): return 2.0 * base(shape) - 1.0
        rng = _R
    if kind == "c4v" and dtype == "c128":
        from groups.pg import make_c4v_symm
        A = make_c4v_symm(torch.from_numpy(rng.random((2, D, D, D, D)))) \
            + 1j * make_c4v_symm(torch.from_numpy(rng.random((2, D, D, D, D)) - 0.5), irreps=["A2"])
        A = A.numpy()
        return {(0, 0): A / np.abs(A).max()}
    if dtype == "c128":      # re and im parts each U[0,1)
        sites = {}
        for y in range(2):
            for x in range(2):
                A = rng.random((2, D, D, D, D)) + 1j * rng.random((2, D, D, D, D))
                sites[(x, y)] = A / np.abs(A).max()
        return sites
    if kind == "c4v":
        from groups.pg import make_c4v_symm                      # the host layer's own symmetriser (reference groups/pg.py:27-63)
        if signed:
            # the round-3 construction, kept for comparability: 2 * sym(U[0,1)) / max - 1 (mean ~ +0.25).  The zero-mean variant sym(U(-1,1)) is
            # not a convergent CTM problem at D = 4 chi = 64: the environment moves by O(1) in every sweep (|R|_F / |l0| = 0.4 ... 3 of the previous
            # subspace on the new corner), the spectrum behind the kept pairs is flat (0.54 per application) and every sweep takes the regular
            # block iteration: 73 ms (tools/probe_c4v_signed.py new)
            A = make_c4v_symm(torch.from_numpy(base((2, D, D, D, D)))).numpy()
            A = 2.0 * (A / np.abs(A).max()) - 1.0
            return {(0, 0): A / np.abs(A).max()}
        A = make_c4v_symm(torch.from_numpy(rng.random((2, D, D, D, D)))).numpy()
        return {(0, 0): A / np.abs(A).max()}
    sites = {}
    for y in range(2):
        for x in range(2):
            A = rng.random((2, D, D, D, D))
            sites[(x, y)] = A / np.abs(A).max()
    return sites


def gemm_macs_per_unit(D, chi, p=2):
    """real MACs of the GEMM-shaped work of one (site, direction) unit, 'sl' mode."""
    return 3 * chi ** 3 * D ** 6 + 8 * chi ** 3 * D ** 4 + 8 * chi ** 3 * D ** 2 + 10 * p * chi ** 2 * D ** 6


def _cpu_env(O, ost, chi, env_np, sites):
    """Environment for the CPU leg: the tensors the GPU run ended with (downloaded), i.e. the same environment the timed sweeps
    worked on -- dense LAPACK/BLAS does the same work on any environment, but the comparison should not rest on that."""
    env = O.init_env_ctmrg(ost, chi)
    if env_np is not None:
        C, T = env_np
        for k in env.C: env.C[k] = np.ascontiguousarray(C[k])
        for k in env.T: env.T[k] = np.ascontiguousarray(T[k])
        return env, "environment of the timed GPU run"
    rng = np.random.default_rng(7)
    cx = np.iscomplexobj(sites[(0, 0)])
    for k in env.C: env.C[k] = rng.random(env.C[k].shape) + (1j * rng.random(env.C[k].shape) if cx else 0.0)
    for k in env.T: env.T[k] = rng.random(env.T[k].shape) + (1j * rng.random(env.T[k].shape) if cx else 0.0)
    return env, "dense random environment"


def cpu_baseline(kind, D, chi, sites, env_np=None, budget_s=25.0, svd_n=None):
    """numpy oracle on the host cores: time ONE unit (site (0,0), UP move: halves -> projectors -> absorb) of
    the generic sweep -- or ONE C4v sweep -- and extrapolate to sweeps/s.  Bounded sample."""
    from oracle import ctm_oracle as O, c4v_oracle as O4
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    if kind == "c4v":
        A = sites[(0, 0)]
        C, T = O4.init_env_ctmrg(A, chi)
        for _ in range(2):
            C, T = O4.ctm_move_sl(A, C, T)
        n, t0 = 0, time.perf_counter()
        while True:
            C, T = O4.ctm_move_sl(A, C, T); n += 1
            if time.perf_counter() - t0 > min(budget_s, 10.0) or n >= 20:
                break
        dt = (time.perf_counter() - t0) / n
        return {"value": 1.0 / dt, "unit": "sweeps/s", "cores": threads, "kind": "port",
                "sample": f"{n} full C4v sweeps of the numpy oracle (LAPACK eigh), {os.cpu_count()} host cpus"}
    ost = O.State(sites)
    env, env_what = _cpu_env(O, ost, chi, env_np, sites)
    n = chi * D * D
    if not np.iscomplexobj(sites[(0, 0)]):
        # C++ restatement on threaded OpenBLAS/LAPACK (oracle/cpu_unit.cpp): ONE full unit -- four enlarged corners, R, Rt,
        # M = R^T Rt (three n^3 dgemm), full dgesdd, projectors, absorb -- measured; only at n > 6000 the dgesdd is measured on
        # the leading svd_n x svd_n block of M and scaled by (n / svd_n)^3 (a full n = 16384 dgesdd alone takes ~20 minutes)
        from oracle import cpu_unit
        nsub = 0 if n <= 6000 else min(n, svd_n or 4096)
        r = cpu_unit.run_unit(O.UP, (0, 0), ost, env, svd_nsub=nsub)
        t = dict(r["times"])
        t_svd = t["svd"] * ((n / nsub) ** 3 if nsub else 1.0)
        dt = t["corners"] + t["halves"] + t_svd + t["proj"] + t["absorb"]
        how = "measured" if not nsub else f"dgesdd measured at n_s={nsub} ({t['svd']:.1f} s) and scaled by (n/n_s)^3 = {t_svd:.0f} s, everything else measured at full size"
        # (the driver keeps the first 120 characters of the sample: what is extrapolated comes first)
        lead = (f"EXTRAPOLATED: dgesdd timed at n_s={nsub}, scaled x(n/n_s)^3 to n={n} = {100 * t_svd / dt:.0f}% of 1 unit; x32 units/sweep. " if nsub
                else "1 unit measured in full, x32 units/sweep (extrapolated). ")
        return {"value": 1.0 / (32 * dt), "unit": "sweeps/s", "cores": r["threads"], "kind": "port",
                "sample": lead + f"1 of the 32 (site,direction) units of one sweep, C++ restatement on OpenBLAS/LAPACK ({r['threads']} threads of "
                          f"{os.cpu_count()} host cpus) on the {env_what}: 4 corners {t['corners']:.1f} s + R, Rt, M (3 n^3 dgemm) {t['halves']:.1f} s + "
                          f"dgesdd {t_svd:.1f} s + projectors {t['proj']:.2f} s + absorb {t['absorb']:.1f} s = {dt:.1f} s/unit ({how}), x32 units/sweep"}
    if n > 6000:
        return cpu_baseline_large(O, ost, env, D, chi, threads, env_what)
    t0 = time.perf_counter()
    P, Pt = O.get_projectors_4x4(O.UP, (0, 0), ost, env)
    Pd = {c: P for c in ost.sites}; Ptd = {c: Pt for c in ost.sites}
    O.absorb_truncate(O.UP, (0, 0), ost, env, Pd, Ptd)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (32 * dt), "unit": "sweeps/s", "cores": threads, "kind": "port",
            "sample": f"1 of the 32 (site,direction) units of one sweep (numpy oracle: 4 corners, 2 halves, M, LAPACK gesdd, "
                      f"projectors, absorb) on the {env_what} = {dt:.2f} s, extrapolated x32; {os.cpu_count()} host cpus"}


def energy_parity(dev, dtype="f64", D=3, chi=36, nsweeps=3, signed=False):
    """rdm2x2 energy (J1-J2, j2 = 0.5) after `nsweeps` sweeps from the CTMRG init: native engine vs the numpy oracle on the SAME
    synthetic state, at a size the oracle finishes in seconds (the second half of BASELINE.json's metric)."""
    import config as cfg
    from oracle import ctm_oracle as O, j1j2_oracle as OJ
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from models import j1j2
    sites = synth_sites("generic", D, seed=3, dtype=dtype, signed=signed)
    st = IPEPS({k: torch.from_numpy(v).to(dev) for k, v in sites.items()})
    env = ENV(chi, st); init_env(st, env)
    import _native
    eng = _native.engine()
    l0 = eng.stat("lz_hits")
    for _ in range(nsweeps):
        for d in cfg.ctm_args.ctm_move_sequence:
            for _r in range(2):
                ctmrg.ctm_MOVE(d, st, env)
    e = float(j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env))
    spec = {k: v.cpu().numpy() for k, v in env.get_spectra().items()}
    nkry = int(eng.stat("lz_hits") - l0)
    t0 = time.perf_counter()
    ost = O.State(sites); oe = O.init_env_ctmrg(ost, chi)
    for _ in range(nsweeps):
        O.ctm_sweep(ost, oe)
    eo = OJ.energy_per_site([O.rdm2x2(c, ost, oe) for c in ost.sites], 1.0, 0.5)
    ospec = O.corner_spectra(oe)
    return {"workload": f"generic 2x2 D={D} chi={chi} {dtype}{' signed' if signed else ''}, {nsweeps} sweeps from the CTMRG init, J1-J2 j2=0.5", "energy_native": e,
            "energy_oracle": float(eo), "rel_err": abs(e - eo) / max(abs(eo), 1e-300), "abs_err": abs(e - eo),
            "max_abs_err_corner_spectra": float(max(np.abs(spec[k] - ospec[k]).max() for k in ospec)), "tolerance": 1e-10,
            # how many of the truncations of these sweeps were block Krylov solves (the full-rank route) rather than block power iterations
            "block_krylov_solves": nkry, "truncations": 32 * nsweeps, "oracle_seconds": round(time.perf_counter() - t0, 1)}


def cpu_baseline_large(O, ost, env, D, chi, threads, env_what="dense random environment"):
    """n = chi D^2 > 6000: one full unit on the CPU takes many minutes (LAPACK gesdd of an n x n matrix), so the unit is
    assembled from bounded pieces: the four enlarged corners and the absorb are timed at full size with the oracle, ONE
    n x n x n product is timed and counted three times (two halves + M = R^T Rt), and the SVD is timed at n_s = 4096 and
    scaled by (n / n_s)^3."""
    rng = np.random.default_rng(7)
    cx = np.iscomplexobj(next(iter(ost.sites.values())))
    n = chi * D * D
    t0 = time.perf_counter()
    cs = [O.c2x2(cid, (0, 0), ost, env) for cid in range(4)]
    t_corners = time.perf_counter() - t0
    t0 = time.perf_counter(); R = cs[0] @ cs[1]; t_gemm = time.perf_counter() - t0
    ns = 4096
    t0 = time.perf_counter(); np.linalg.svd(cs[2][:ns, :ns]); t_svd_s = time.perf_counter() - t0
    t_svd = t_svd_s * (n / ns) ** 3
    P = rng.random((n, chi)) + (1j * rng.random((n, chi)) if cx else 0.0)
    t0 = time.perf_counter(); _ = R @ P; _ = R @ P; t_proj = time.perf_counter() - t0
    Pd = {c: P for c in ost.sites}
    t0 = time.perf_counter(); O.absorb_truncate(O.UP, (0, 0), ost, env, Pd, Pd); t_abs = time.perf_counter() - t0
    dt = t_corners + 3 * t_gemm + t_svd + t_proj + t_abs
    return {"value": 1.0 / (32 * dt), "unit": "sweeps/s", "cores": threads, "kind": "port",
            # (the driver keeps the first 120 characters: what is extrapolated comes first)
            "sample": f"EXTRAPOLATED: dgesdd timed at n_s=4096, scaled x(n/n_s)^3 to n={n} = {100 * t_svd / dt:.0f}% of 1 unit; x32 units/sweep. "
                      f"One (site,direction) unit of the numpy oracle on the {env_what}, assembled from bounded pieces: 4 corners {t_corners:.1f} s + 3 x (n^3 GEMM "
                      f"{t_gemm:.1f} s) + gesdd at n_s=4096 {t_svd_s:.1f} s scaled = {t_svd:.0f} s + projector GEMMs {t_proj:.1f} s + "
                      f"absorb {t_abs:.1f} s = {dt:.0f} s/unit; {os.cpu_count()} host cpus"}


def _union_ms(ivals, kind):
    """Independent units run on concurrent streams, so launches overlap: the time the chip spends on a kernel class is the UNION
    of the [start, end] intervals of its launches (equal to the sum of durations when nothing overlaps)."""
    iv = sorted((a, b) for k_, a, b, _ in ivals if int(k_) == kind)
    tot, cur_a, cur_b = 0.0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None: tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    return tot + ((cur_b - cur_a) if cur_b is not None else 0.0)


HBM_PEAK_GBPS = 8000.0
# instrumented kernel classes (csrc: timing_end kinds).  Class 3 (streaming strip / row-block kernels) reports ALGORITHMIC BYTES
# (the big operand read once + the row block in and out) AND is priced against HBM; the others report flops.
CLS = {0: ("gemm_f64_fast_kernel / gemm_f64_kernel<4,4> (128x128 tile)", "mfma", ("gemm_f64_fast_kernel", "gemm_f64_kernel<4, 4>")),
       1: ("gemm_f64_kernel<2,2> (64x64 tile: small, segmented and batched products)", "mfma", ("gemm_f64_kernel<2, 2>",)),
       2: ("layer2_reg_kernel<KT> (float64) / layer2_c_kernel<KT> (complex128): fused two-layer enlarged-corner kernel", "mfma", ("layer2_",)),
       3: ("gemm_rows_kernel<TM<=2,BNF> (gemm_strip_kernel): <= 32-row block times an n x n corner, streamed once", "hbm",
           ("gemm_rows_kernel<1,", "gemm_rows_kernel<2,", "gemm_strip_kernel")),
       4: ("gemm_rows_kernel<TM>=3,BNF>: 33..64-row block times an n x n corner (16 flop per byte of the corner)", "mfma",
           ("gemm_rows_kernel<3,", "gemm_rows_kernel<4,"))}


def _rate(work, ms, bound):              # TFLOP/s or GB/s
    return work / max(ms * 1e-3, 1e-30) / (1e12 if bound == "mfma" else 1e9)


def _describe(i, kms, kfl, kn, ums=None):
    """frac = plain per-launch figure: total algorithmic work / SUM of the launch durations (what rocprofv3's per-kernel
    average confirms); frac_union = the same work / union of the launch intervals (what the chip delivered on that class while
    units share it)."""
    name, bound, _ = CLS[i]
    peak = FP64_MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBPS
    ach = _rate(kfl[i], kms[i], bound) if kn[i] else 0.0
    d = {"kernel": name, "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
         "frac": round(ach / peak, 4), "launches": int(kn[i]), "avg_launch_ms": round(kms[i] / max(kn[i], 1), 5),
         "algorithmic_per_launch": round(kfl[i] / max(kn[i], 1)),          # flop (mfma classes) or bytes (hbm class) per launch: compare `traffic` with it on the hbm class
         "sum_launch_ms": round(kms[i], 3)}
    if ums is not None and kn[i]:
        d["busy_ms_union"] = round(ums[i], 3)
        d["achieved_union"] = round(_rate(kfl[i], ums[i], bound), 3)
        d["frac_union"] = round(d["achieved_union"] / peak, 4)
    return d


def stationary_block(eng, step, env, warm_tol=1e-9, max_sweeps=14, timed=3):
    """The stationary-environment fast path (ctm_args.projector_warm_tol, csrc/svd_leading.hip: svd_stationary) on the environment the timed
    sweeps ended with: sweeps with the option on until one sweep's truncations were ALL accepted from the previous basis (or max_sweeps),
    then `timed` sweeps timed.  Reported next to the rate of the sweeps that solve every truncation from scratch; the option is off again
    on return (it is opt-in: a residual tolerance instead of the rounding-level threshold)."""
    import config as cfg
    old = cfg.ctm_args.projector_warm_tol
    cfg.ctm_args.projector_warm_tol = warm_tol
    out = {"projector_warm_tol": warm_tol}
    try:
        per, n_before = [], None
        spec0 = {k: (v / v[0]).cpu() for k, v in env.get_spectra().items()}
        for i in range(max_sweeps):
            a0, r0 = eng.stat("warm_accepts"), eng.stat("warm_rejects")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0, int(eng.stat("warm_accepts") - a0), int(eng.stat("warm_rejects") - r0)))
            if per[-1][1] >= 32:
                n_before = i
                break
        out["sweeps_until_every_truncation_is_accepted"] = n_before
        out["accepted_per_sweep_on_the_way"] = [p_[1] for p_ in per]
        if n_before is None:
            out["note"] = "the environment did not become stationary to the tolerance within the sweeps tried"
            return out
        a0, l0 = eng.stat("warm_accepts"), eng.stat("lz_hits")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(timed):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        spec1 = {k: (v / v[0]).cpu() for k, v in env.get_spectra().items()}
        out.update({"sweeps_per_sec": timed / dt, "ms_per_step": 1e3 * dt / timed, "steps": timed,
                    "accepted_truncations": int(eng.stat("warm_accepts") - a0), "full_solves": int(eng.stat("lz_hits") - l0),
                    "max_change_of_corner_spectra_since_the_timed_region": float(max(float((spec1[k] - spec0[k]).abs().max()) for k in spec0))})
    finally:
        cfg.ctm_args.projector_warm_tol = old
        for e in [eng] + list(getattr(eng, "workers", [])):
            e.set_option("warm_accept_tol", 0.0); e._warm_tol = 0.0
    return out


def run_workload(args, eng, dev, kind, D, chi, dtype, signed, steps, warmup, world, rank, dist):
    """Build the synthetic state, warm up, time `steps` sweeps; returns (dict for the JSON line, sites, state, env)."""
    import config as cfg
    from ipeps.ipeps import IPEPS
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env as init_env_c4v
    from ctm.one_site_c4v import ctmrg_c4v
    sites = synth_sites(kind, D, dtype=dtype, signed=signed)
    if kind == "c4v":
        state = IPEPS_C4V(torch.from_numpy(sites[(0, 0)]).to(dev))
        env = ENV_C4V(chi, state); init_env_c4v(state, env)
        a = state.site()
        def step():
            ctmrg_c4v.ctm_MOVE_sl(a, env)
    else:
        state = IPEPS({k: torch.from_numpy(v).to(dev) for k, v in sites.items()})
        env = ENV(chi, state); init_env(state, env)
        def step():
            for d in cfg.ctm_args.ctm_move_sequence:
                for _ in range(state.lX if d in [(-1, 0), (1, 0)] else state.lY):
                    ctmrg.ctm_MOVE(d, state, env)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    import parallel
    parallel.comm_time_s(reset=True); parallel.comm_timing = world > 1
    eng.set_option("gemm_timing", 1)
    eng.timers(reset=True)
    if args.profile:
        eng.set_option("profile", 1)
    step_ms, step_hits = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        if kind == "c4v":                 # ms-scale steps: per-step wall times (the sync costs ~10 us) for the steady-state figure
            torch.cuda.synchronize()
            step_ms.append(time.perf_counter())
            step_hits.append(eng.stat("eigh_warm_hits"))
    fence()
    dt = time.perf_counter() - t0
    if step_ms:
        step_ms = [1e3 * (b - a) for a, b in zip([t0] + step_ms[:-1], step_ms)]
    ivals = eng.gemm_intervals()
    KK = range(5)
    k_ms = [eng.stat(f"k_ms{i}") for i in KK]
    k_fl = [eng.stat(f"k_flops{i}") for i in KK]
    k_n = [eng.stat(f"k_calls{i}") for i in KK]
    phase = eng.timers()
    # time this rank spent in the two exchanges of every move (all-gather of P, Pt and of the new C, C, T): its own phase entry
    phase["comm"] = parallel.comm_time_s(reset=True); parallel.comm_timing = False
    absorb_bytes, absorb_calls = eng.stat("absorb_bytes"), eng.stat("absorb_calls")
    # flop the engine EXECUTED in the timed region: every GEMM launch (all kernels of csrc/gemm_f64.hip) + the fused two-layer kernel
    executed_flop = eng.stat("gemm_flops") + eng.stat("layer2_flops")
    svd = {"decompositions": int(eng.stat("jacobi_calls")),
           "avg_jacobi_sweeps": round(eng.stat("total_sweeps") / max(eng.stat("jacobi_calls"), 1), 2),
           "power_iter_hits": int(eng.stat("si_hits")), "power_iter_fallbacks_to_full": int(eng.stat("si_fallbacks")),
           "block_krylov_solves": int(eng.stat("lz_hits")),
           "avg_block_krylov_steps": round(eng.stat("lz_total_steps") / max(eng.stat("lz_hits"), 1), 2),
           "ritz_extractions": int(eng.stat("lz_extractions")),
           # ... of which started from the rotations of the unit's previous extraction, and the Jacobi sweeps an extraction took on average
           "ritz_warm_starts": int(eng.stat("ritz_warm_starts")),
           # corner passes this rank computed as one half of a rank pair that shares the unit (twice as many ranks as sites; 0 otherwise)
           "shared_corner_passes": int(eng.stat("comm_calls")),
           "avg_sweeps_per_ritz_extraction": round(eng.stat("ritz_sweeps") / max(eng.stat("lz_extractions"), 1), 2)}
    eng.set_option("gemm_timing", 0)
    if args.profile:
        eng.set_option("profile", 0)
    # reference pass for the kernel-quality figure: ONE more sweep with the units issued serially on one stream (nothing
    # co-scheduled), outside the timed region -- per-launch rates of the same kernels without sharing the chip
    serial = None
    if world == 1 and kind != "c4v" and not args.serial_units and not args.no_serial_pass:
        cfg.ctm_args.concurrent_units = False
        eng.set_option("gemm_timing", 1)
        step(); fence()
        serial = ([eng.stat(f"k_ms{i}") for i in KK], [eng.stat(f"k_flops{i}") for i in KK], [eng.stat(f"k_calls{i}") for i in KK])
        eng.set_option("gemm_timing", 0)
        cfg.ctm_args.concurrent_units = True
    stationary = None
    # both dtypes (svd_stationary / svd_stationary_c); bounded: up to 14 + 3 more sweeps -- skipped where that is more than ~2 minutes
    # (configs[4] at 18-35 s per sweep: run `bench.py --config generic_D8_chi384_c128 --signed --stationary-budget-s 900` for that block)
    if world == 1 and kind != "c4v" and signed and not args.no_stationary and not args.warm_tol and 17 * dt / steps > args.stationary_budget_s:
        stationary = {"skipped": f"17 sweeps of {dt / steps:.1f} s exceed --stationary-budget-s {args.stationary_budget_s:.0f}"}
    elif world == 1 and kind != "c4v" and signed and not args.no_stationary and not args.warm_tol:
        # the block advances the environment by up to 14 + 3 approximately truncated sweeps: the `state` block below and the energy
        # blocks of main() describe the environment the TIMED sweeps ended with (what `--gpus N` lines report too), so the block works
        # on the environment and the old tensors are put back afterwards (a move rebinds env.C / env.T entries, it writes into none)
        C_timed, T_timed = dict(env.C), dict(env.T)
        try:
            stationary = stationary_block(eng, step, env)
        except Exception as e:                         # reporting only
            stationary = {"error": repr(e)}
        env.C, env.T = C_timed, T_timed
    comm_ranks = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # time inside the exchanges per rank (min / max over ranks: an idle rank waits in the collective for the busiest one)
        c = torch.tensor([phase["comm"], -phase["comm"], executed_flop], dtype=torch.float64, device=dev)
        cmax = c.clone(); dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        csum = c.clone(); dist.all_reduce(csum)
        comm_ranks = {"max_s": float(cmax[0]), "min_s": float(-cmax[1]), "mean_s": float(csum[0]) / world}
        executed_flop = float(csum[2])
    u_ms = [_union_ms(ivals, i) for i in KK]
    # the dominant kernel = the class the chip spends most time on (sum of its launch durations)
    dom = max((0, 2, 3, 4), key=lambda i: k_ms[i])
    roof = _describe(dom, k_ms, k_fl, k_n, u_ms)
    roof = {"bound": roof.pop("bound"), **roof, "traffic": None,
            "concurrent_streams": max(1, len(getattr(eng, "workers", []))),
            "time_share_of_sweep": round(u_ms[dom] * 1e-3 / dt, 4),
            "algorithmic_work_per_launch": ("2*M*N*K flop of each product" if dom in (0, 1) else
                                            "2 * p * chi_x * chi_y * (D^2)^2 * 2 D^2 flop per launch (both layers)" if dom == 2 else
                                            "2*M*N*K flop: M <= 64 rows times the N x K corner" if dom == 4 else
                                            "8 * (K*N + M*K + M*N) bytes: the n x n corner read once, the <=64-row block in and out"),
            "other_kernels": {str(i): _describe(i, k_ms, k_fl, k_n, u_ms) for i in KK if i != dom and k_n[i]}}
    if serial is not None:
        s_ms, s_fl, s_n = serial
        roof["serial_pass"] = {"note": "one extra sweep after the timed region, units issued serially on one stream (no co-scheduling): kernel quality without sharing the chip",
                               **{("dominant" if i == dom else str(i)): _describe(i, s_ms, s_fl, s_n) for i in KK if s_n[i]}}
    # north_star: achieved HBM GB/s on the absorb step = algorithmic bytes of the absorb calls (operands read once, results written
    # once) / device time of the absorb phase (HIP events on the engines' streams, summed over streams)
    if absorb_calls and phase.get("absorb", 0.0) > 0:
        roof["absorb_step"] = {"bound": "hbm", "calls": int(absorb_calls), "algorithmic_bytes_per_call": round(absorb_bytes / absorb_calls),
                               "device_ms_per_call": round(1e3 * phase["absorb"] / absorb_calls, 4),
                               "achieved": round(absorb_bytes / phase["absorb"] / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(absorb_bytes / phase["absorb"] / 1e9 / HBM_PEAK_GBPS, 5),
                               "note": "the absorb is a chain of skinny GEMMs and the fused two-layer kernel over chi^2 D^4-sized intermediates: "
                                       "its time is set by those kernels (MFMA / latency), not by streaming its O(n chi) operands"}
        # ... so its MFMA fraction next to the GB/s figure: algorithmic flops of one absorb (4 chi^3 D^2 + 2 chi^3 D^4 +
        # 2 p chi^2 D^6 MACs, x4 for complex) / device time per call; only when the absorbs ran on all chi projector columns
        ncols = max((env.__dict__.get("_ncol") or {}).values(), default=chi) if kind != "c4v" else chi
        if 2 * ncols > chi:
            fl = 2.0 * (4.0 * chi ** 3 * D ** 2 + 2.0 * chi ** 3 * D ** 4 + 2.0 * 2 * chi ** 2 * D ** 6) * (4.0 if dtype == "c128" else 1.0)
            tf = fl * absorb_calls / phase["absorb"] / 1e12
            roof["absorb_step"]["mfma"] = {"flops_per_call": fl, "achieved": round(tf, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
                                           "note": "device time per call is measured on the unit's stream while the other units of the move share the chip"}
    # sweep-level figure north_star asks for ("fraction of the FP64 MFMA roofline"): executed flop / wall time / (peak x GPUs).  NOT
    # SURVEY 8d's dense flop count (3 n^3 GEMMs + a full SVD per unit): the engine never forms R, Rt, M
    roof["executed_flop_per_sweep"] = executed_flop / steps
    roof["sweep_mfma_frac"] = round(executed_flop / dt / (FP64_MFMA_PEAK_TFLOPS * 1e12 * world), 4)
    if comm_ranks is not None:
        roof["comm_per_rank"] = comm_ranks
    out = {"value": steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup, "roofline": roof, "svd": svd,
           "phase_s": {k: round(v, 4) for k, v in phase.items()},
           "phase_s_note": "device time per phase from HIP events on the engines' streams, summed over concurrent streams (not wall time)"}
    if stationary is not None:
        # the generic twin of the C4v block's moving / stationary split: `value` is the rate of sweeps that solve every truncation from
        # scratch (what every sweep of a default run pays, converged or not); with ctm_args.projector_warm_tol a converged run pays this
        stationary["solve_from_scratch_sweeps_per_sec"] = steps / dt
        if stationary.get("sweeps_per_sec"):
            stationary["speedup"] = round(stationary["sweeps_per_sec"] * dt / steps, 3)
        out["stationary_environment"] = stationary
    if step_ms:
        srt = sorted(step_ms)
        med = srt[len(srt) // 2]
        # sweeps in which the environment still moved (the warm restart of the eigensolver was NOT accepted: the regular subspace
        # iteration ran) against sweeps on a stationary corner (restart accepted)
        hit = [b > a for a, b in zip([0] + step_hits[:-1], step_hits)]          # (the counter was reset with the timers above)
        mov = [t for t, h in zip(step_ms, hit) if not h]
        sta = [t for t, h in zip(step_ms, hit) if h]
        out["moving_environment"] = {"sweeps": len(mov), "ms_per_step": round(sum(mov) / max(len(mov), 1), 4),
                                     "sweeps_per_sec": round(1e3 * len(mov) / sum(mov), 1) if mov else None}
        out["stationary_environment"] = {"sweeps": len(sta), "ms_per_step": round(sum(sta) / max(len(sta), 1), 4),
                                         "sweeps_per_sec": round(1e3 * len(sta) / sum(sta), 1) if sta else None}
        out["steady_state"] = {"ms_per_step_median": round(med, 4), "sweeps_per_sec_at_median": round(1e3 / med, 1),
                               "ms_per_step_first": round(step_ms[0], 3), "ms_per_step_max": round(srt[-1], 3),
                               "warm_restarts_accepted": int(eng.stat("eigh_warm_hits")), "warm_restarts_rejected_by_probe": int(eng.stat("eigh_warm_rejects")),
                               "note": "`value` is the mean over all timed sweeps, including the first ones after the warm-up in which the environment still moves "
                                       "and the eigensolver iterates from the previous subspace; once the enlarged corner is stationary to the residual "
                                       "threshold a sweep is one Rayleigh-Ritz in the previous subspace plus a deflated probe for missed directions"}
    if kind != "c4v":
        # what the engine could exploit on THIS state: numerical rank of the truncation and the number of projector columns above
        # projector_svd_reltol, out of chi
        nc = env.__dict__.get("_ncol") or {}
        spec = list(env.get_spectra().values())
        S = [float((s_ > 1e-8 * s_[0]).sum()) for s_ in spec]
        out["state"] = {"chi": chi, "signed": bool(signed),
                        # sum of the normalised corner spectra after the last timed sweep: the same number from 1, 2, 4 ranks (tests/test_gpu_bench_ranks.py)
                        "corner_spectra_checksum": float(sum(float((s_ / s_[0]).sum()) for s_ in spec)),
                        "corner_values_above_1e-8": int(min(S)) if S else None,
                        "nonzero_projector_columns": (max(nc.values()) if nc else chi),
                        "effective_rank_much_smaller_than_chi": bool(S and min(S) < 0.25 * chi),
                        "corner_cache_hits": int(eng.stat("corner_cache_hits")),
                        "note": ("signed random tensors A ~ U(-1,1): full-rank environment (all chi projector columns significant), "
                                 "block-Krylov truncation on every unit") if signed else
                                ("positive random tensors A ~ U[0,1) (the state SURVEY 8d prescribes): numerically low-rank environment, "
                                 "see the full_rank block for the same shape on a full-rank state")}
    return out, dom, sites, state, env


def energy_block(eng, state, env, world):
    """The energy half of the metric at the size of the timed run: seconds for ONE evaluation of E/site = mean over the 4 sites of
    tr(rho_2x2(coord) h_p) (reference models/j1j2.py:236-240, ctm/generic/rdm.py:1362-1592) on the environment the timed sweeps
    ended with, plus the invariants of the four plaquette RDMs (trace, Hermiticity, smallest eigenvalue)."""
    from models import j1j2
    from ctm.generic import rdm
    model = j1j2.J1J2(j1=1.0, j2=0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = float(model.energy_per_site(state, env))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out = {"seconds_per_energy_4_sites": round(dt, 3), "energy_per_site_j2_0.5": e, "n_gpus": world,
           "what": "rdm2x2 of the 4 sites (chunked open halves, split-K n^3 GEMMs) + tr(rho h_p), after the timed sweeps of this block"}
    if world == 1:
        r = rdm.rdm2x2((0, 0), state, env).reshape(16, 16).cpu()
        out["rdm2x2_invariants"] = {"trace_minus_1": float(abs(torch.trace(r) - 1.0)), "hermiticity": float((r - r.conj().T).abs().max()),
                                    "min_eigenvalue": float(torch.linalg.eigvalsh(0.5 * (r + r.conj().T)).min())}
    return out


def compact(res):
    """One BASELINE configuration as a short block of the JSON line."""
    roof = res["roofline"]
    out = {"value": res["value"], "unit": "sweeps/s", "ms_per_step": res["ms_per_step"], "steps": res["steps"], "warmup": res["warmup"],
           "dominant_kernel": {**{k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_union")},
                               "frac_alone": ((roof.get("serial_pass") or {}).get("dominant") or {}).get("frac")},
           "executed_flop_per_sweep": roof.get("executed_flop_per_sweep"), "sweep_mfma_frac": roof.get("sweep_mfma_frac"),
           "svd": res["svd"]}
    for k in ("state", "steady_state", "moving_environment", "stationary_environment"):
        if k in res:
            out[k] = res[k]
    if "state" in out:
        out["state"] = {k: v for k, v in out["state"].items() if k != "note"}
    return out


def other_configs(args, eng, dev, world, rank, dist):
    """Every other single-GPU BASELINE configuration, a few sweeps each (the default line used to carry configs[3] only):
    configs[1] C4v D4 chi64 (with the rate while the environment still moves), configs[2] generic D6 chi128 on both states,
    configs[4] generic D8 chi384 complex128 (1 warm-up + 1 timed sweep per state: 10-20 s per sweep on one GPU)."""
    import gc
    out = {}
    saved = (args.no_serial_pass,)
    args.no_serial_pass = True
    try:
        # warm-up = ceil(chi / D^2) sweeps: C4v 4, D6 chi128 4, D8 chi384 complex128 6 on the prescribed state.  The
        # signed complex128 block -- configs[4] where it is hard: every truncation a complex block Krylov solve at n = 24576, ~35 s per sweep
        # on one GPU -- warms up with 2 sweeps: its corners carry 192 / 225 of 384 values above 1e-8 after the first / second sweep
        # (gpurun probe r4c), i.e. the timed sweep IS the full-rank regime, and 6 warm-up sweeps would add 2.5 minutes to the default run
        for name, signed, steps, warmup in (("c4v_D4_chi64", False, 100, 4), ("c4v_D4_chi64", True, 100, 4), ("generic_D6_chi128", False, 3, 4), ("generic_D6_chi128", True, 3, 4),
                                            ("generic_D8_chi384_c128", False, 1, 3), ("generic_D8_chi384_c128", True, 1, 2)):
            kind, D, chi, dtype = CONFIGS[name]
            key = name + ("_signed" if signed else "")
            try:
                t0 = time.perf_counter()
                # the serially issued extra sweep (`frac_alone`) where a sweep is a fraction of a second (configs[2]); configs[4] at 10-35 s per
                # sweep stays without it (bench.py --config generic_D8_chi384_c128 [--signed] prints it)
                args.no_serial_pass = not (kind == "generic" and dtype == "f64")
                res, dom, sites, state, env = run_workload(args, eng, dev, kind, D, chi, dtype, signed, steps, warmup, world, rank, dist)
                out[key] = compact(res)
                out[key]["dtype"] = dtype
                out[key]["wall_s_incl_warmup"] = round(time.perf_counter() - t0, 1)
                out[key]["units_in_flight"] = max(1, len([w for w in getattr(eng, "workers", []) if w.own_stat("arena_total") > 0]))
                out[key]["hbm_free_GiB_after"] = round(torch.cuda.mem_get_info()[0] / 2 ** 30, 1)
                del state, env, sites
            except Exception as e:
                out[key] = {"error": repr(e)}
            gc.collect(); torch.cuda.empty_cache()
            if hasattr(eng, "trim"):
                eng.trim()
    finally:
        args.no_serial_pass, = saved
    return out


def traffic_from_profile(args, world, dom, signed):
    """HBM traffic of the kernel families from the committed PMC pass of this same command (bench.py cannot attach counters to
    itself); None when no profile of this workload is committed."""
    try:
        import csv, glob
        tag = "signed" if signed else "default"
        # the newest committed round: profiles/rNN_bench_<tag>_pmc_hbm_traffic.csv next to the line the profiled command printed
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_bench_{tag}_pmc_hbm_traffic.csv")))
        if not cands or world != 1:
            return None
        fcsv = cands[-1]
        rnd = os.path.basename(fcsv).split("_")[0]
        line = [f for f in (os.path.join(REPO, "profiles", f"{rnd}_bench_{tag}_under_rocprof.json"), os.path.join(REPO, "profiles", f"{rnd}_bench_{tag}.json"))
                if os.path.exists(f)]
        prof = json.loads(open(line[0]).read().strip().splitlines()[-1])
        if prof["config"]["workload"] != args.config:
            return None
        rows = list(csv.DictReader(open(fcsv)))

        def pmc(keys):
            tb = tn = 0.0
            for row in rows:
                if any(k_ in row["kernel"] for k_ in keys):
                    tb += float(row["hbm_bytes_per_launch(2x_fetch_corrected)"]) * float(row["launches"]); tn += float(row["launches"])
            return round(tb / tn) if tn else None
        cfile = os.path.join(REPO, "profiles", f"{rnd}_pmc_commit.txt")
        commit = open(cfile).read().strip() if os.path.exists(cfile) else "unrecorded"
        return {"dominant": pmc(CLS[dom][2]), "others": {str(i): pmc(CLS[i][2]) for i in CLS if i != dom},
                "source": f"profiles/{os.path.basename(fcsv)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on the build of commit {commit}, "
                          "average HBM bytes per launch; read from the committed file, not measured by this run)"}
    except Exception:
        return None


def _short_kernel(name):
    return name.split(" ")[0].split("(")[0] if name else None


def _compact_block(b):
    """A block of the detail (res of run_workload / compact()) as a handful of scalars."""
    if not isinstance(b, dict) or "error" in b:
        return b
    dk = b.get("dominant_kernel") or b.get("roofline") or {}
    o = {"value": round(b["value"], 5), "ms_per_step": round(b["ms_per_step"], 3), "steps": b["steps"], "warmup": b["warmup"],
         "kernel": _short_kernel(dk.get("kernel")), "bound": dk.get("bound"), "frac": dk.get("frac"), "frac_union": dk.get("frac_union"),
         "frac_alone": dk.get("frac_alone", ((dk.get("serial_pass") or {}).get("dominant") or {}).get("frac")),
         "sweep_mfma_frac": b.get("sweep_mfma_frac", dk.get("sweep_mfma_frac"))}
    sv = b.get("svd") or {}
    if sv.get("block_krylov_solves") is not None:
        o["block_krylov_solves"] = sv["block_krylov_solves"]; o["power_iter_solves"] = sv.get("power_iter_hits")
    stt = b.get("state") or {}
    if stt:
        o["corner_rank_1e-8"] = stt.get("corner_values_above_1e-8"); o["low_rank"] = stt.get("effective_rank_much_smaller_than_chi")
    for k_ in ("moving_environment", "stationary_environment"):
        if k_ in b:
            o[k_.split("_")[0] + "_sweeps_per_sec"] = b[k_].get("sweeps_per_sec"); o[k_.split("_")[0] + "_sweeps"] = b[k_].get("sweeps", b[k_].get("steps"))
            if "solve_from_scratch_sweeps_per_sec" in b[k_]:        # generic full-rank block: the timed sweeps solve every truncation from scratch
                o["moving_sweeps_per_sec"] = b[k_]["solve_from_scratch_sweeps_per_sec"]; o["stationary_warm_tol"] = b[k_].get("projector_warm_tol")
    for k_ in ("dtype", "units_in_flight", "wall_s_incl_warmup"):
        if k_ in b:
            o[k_] = b[k_]
    return o


def metric_line(d):
    """The ONE stdout line: the driver's contract keys, `roofline` with the scalars a reader plans with (both states of the default
    workload: per-launch / union / alone fractions of the dominant kernel, executed flop per sweep, sweep-level MFMA fraction), the CPU
    baseline, and every other configuration as a few scalars.  Everything else is in the detail (see main)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")
    line = {k: d[k] for k in keep if k in d}
    r = d["roofline"]
    if r.get("traffic_source"):
        r = dict(r, traffic_source=r["traffic_source"][:120])
    roof = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_launch_ms", "frac_union", "concurrent_streams", "algorithmic_per_launch",
                                  "time_share_of_sweep", "traffic_source", "executed_flop_per_sweep", "sweep_mfma_frac", "comm_per_rank")
            if r.get(k) is not None or k == "traffic"}
    sp = (r.get("serial_pass") or {}).get("dominant")
    if sp:
        roof["frac_alone"] = sp["frac"]; roof["avg_launch_ms_alone"] = sp["avg_launch_ms"]
    if "absorb_step" in r:
        a = r["absorb_step"]
        roof["absorb_step"] = {"achieved_GBps": a["achieved"], "frac_hbm": a["frac"], "device_ms_per_call": a["device_ms_per_call"],
                               "frac_mfma": (a.get("mfma") or {}).get("frac")}
    roof["note"] = ("frac: work / sum of launch durations in the timed region (HIP events, units share the chip); frac_union: / union of launch intervals; "
                    "frac_alone: serially issued sweep; sweep_mfma_frac: executed flop / wall / FP64 MFMA peak; definitions in DESIGN.md section 5")
    fr = d.get("full_rank")
    if fr:
        fro = fr["roofline"]
        roof.update({"full_rank_value": round(fr["value"], 5), "full_rank_ms_per_step": round(fr["ms_per_step"], 2), "full_rank_kernel": _short_kernel(fro.get("kernel")),
                     "full_rank_bound": fro.get("bound"), "full_rank_frac": fro.get("frac"), "full_rank_frac_union": fro.get("frac_union"),
                     "full_rank_frac_alone": ((fro.get("serial_pass") or {}).get("dominant") or {}).get("frac"),
                     "full_rank_executed_flop_per_sweep": fro.get("executed_flop_per_sweep"), "full_rank_sweep_mfma_frac": fro.get("sweep_mfma_frac"),
                     "full_rank_traffic": fro.get("traffic"), "full_rank_algorithmic_per_launch": fro.get("algorithmic_per_launch"),
                     "full_rank_traffic_source": (fro.get("traffic_source") or "")[:60]})
        blk = {"value": fr["value"], "ms_per_step": fr["ms_per_step"], "steps": fr["steps"], "warmup": fr["warmup"], "state": {k: v for k, v in fr["state"].items() if k != "note"},
               "svd": fr["svd"], "phase_s": fr["phase_s"]}
        if "stationary_environment" in fr:
            se = fr["stationary_environment"]
            blk["stationary_environment"] = {k: (float(f"{v:.5g}") if isinstance(v, float) else v) for k, v in se.items() if k != "note"}
            roof.update({"full_rank_stationary_value": se.get("sweeps_per_sec"), "full_rank_moving_value": se.get("solve_from_scratch_sweeps_per_sec")})
        if "energy" in fr:
            e = fr["energy"]
            blk["energy"] = e if "error" in e else {"seconds_per_energy_4_sites": e["seconds_per_energy_4_sites"], "energy_per_site_j2_0.5": e["energy_per_site_j2_0.5"],
                                                    "n_gpus": e.get("n_gpus"), "rdm2x2_invariants": e.get("rdm2x2_invariants")}
        if "energy_parity" in fr:
            ep_ = fr["energy_parity"]
            blk["energy_parity"] = ep_ if "error" in ep_ else {k: ep_.get(k) for k in ("workload", "rel_err", "abs_err", "max_abs_err_corner_spectra", "tolerance",
                                                                                       "block_krylov_solves", "oracle_seconds")}
        line["full_rank"] = blk
    line["roofline"] = roof
    if "cpu_baseline" in d:
        line["cpu_baseline"] = d["cpu_baseline"]
    for k in ("svd", "phase_s"):
        if k in d:
            line[k] = d[k]
    if "state" in d:
        line["state"] = {k: v for k, v in d["state"].items() if k != "note"}
    for k in ("steady_state", "moving_environment", "stationary_environment"):
        if k in d:
            line[k] = {kk: vv for kk, vv in d[k].items() if kk != "note"}
    if "signed_state" in d:
        line["signed_state"] = _compact_block(d["signed_state"])
    if "energy_parity" in d:
        ep = d["energy_parity"]
        line["energy_parity"] = ep if "error" in ep else {k: ep.get(k) for k in ("workload", "rel_err", "abs_err", "max_abs_err_corner_spectra", "tolerance",
                                                                                  "block_krylov_solves", "oracle_seconds")}
    if "other_configs" in d:
        line["other_configs"] = {k: _compact_block(v) for k, v in d["other_configs"].items()}
    return line


def traffic_live(args, dom, signed, warmup):
    """HBM bytes per launch of the kernel families, MEASURED by this run: two child runs of this same script (one timed sweep of the same
    workload and state) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters only, no tracing, as
    MI355X_MICROARCH.md prescribes -- aggregated as tools/pmc_summary.py does: bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024 (rocprofv3 reports
    KiB; FETCH_SIZE under-counts by 2x on gfx950).  None when rocprofv3 is not there or a pass fails (the committed pass is used then)."""
    import csv, glob, shutil, subprocess, tempfile
    from collections import defaultdict
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None or os.environ.get("CTM_BENCH_CHILD") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) \
            or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None               # (no profiler, or this process is itself a child / running under one)
    tot = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="ctm_pmc_", dir="/tmp")
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--config", args.config,
                   "--steps", "1", "--warmup", str(warmup), "--no-cpu-baseline", "--no-serial-pass", "--no-other-configs", "--no-energy", "--no-live-traffic", "--no-stationary"] + \
                  (["--signed"] if signed else ["--no-full-rank"]) + [x for kv in args.opt for x in ("--opt", kv)]
            env = dict(os.environ, CTM_BENCH_CHILD="1", TMPDIR="/tmp")
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=300)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGKILL)        # (the profiler wrapper and the child it started: this process group only)
                pr.wait()
                return None
            if rc != 0:
                return None
            t, c = defaultdict(float), defaultdict(int)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter:
                            t[row["Kernel_Name"]] += float(row["Counter_Value"]); c[row["Kernel_Name"]] += 1
            tot[counter] = (t, c)
            shutil.rmtree(d, ignore_errors=True)

        def per_launch(keys):
            b = n = 0.0
            for k, v in tot["FETCH_SIZE"][0].items():
                if any(k_ in k for k_ in keys):
                    b += 2.0 * v * 1024.0; n += tot["FETCH_SIZE"][1][k]
            for k, v in tot["WRITE_SIZE"][0].items():
                if any(k_ in k for k_ in keys):
                    b += v * 1024.0
            return round(b / n) if n else None
        return {"dominant": per_launch(CLS[dom][2]), "others": {str(i): per_launch(CLS[i][2]) for i in CLS if i != dom},
                "source": "measured by this run: child passes of this command under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one timed sweep each, "
                          "counters only), bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch of the kernel family"}
    except Exception:
        return None


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run with N ranks on this node (one per
    GPU, RCCL), which is exactly what the driver's own command line does."""
    import socket, subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    ngpu = torch.cuda.device_count()
    if ngpu < n and not os.environ.get("CTM_BENCH_ONE_DEVICE"):
        raise SystemExit(f"bench.py --gpus {n}: only {ngpu} GPU(s) visible on this node")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default=DEFAULT_CONFIG, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-svd-n", type=int, default=None, help="CPU baseline at n > 6000: size of the dgesdd that is measured (default 4096)")
    ap.add_argument("--profile", action="store_true", help="per-phase HOST timers (adds stream syncs; phase_s is event-based without it)")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (development)")
    ap.add_argument("--signed", action="store_true", help="time ONLY the full-rank state A ~ U(-1,1) (as `value`)")
    ap.add_argument("--no-full-rank", action="store_true", help="skip the second (full-rank) block of the default run")
    ap.add_argument("--no-serial-pass", action="store_true", help="skip the extra serially issued sweep behind roofline.serial_pass")
    ap.add_argument("--serial-units", action="store_true", help="do not overlap the independent site-units of a move on streams")
    ap.add_argument("--cold-start", action="store_true", help="no warm start of the leading-chi iteration")
    ap.add_argument("--warm-tol", type=float, default=0.0, help="ctm_args.projector_warm_tol for the WHOLE run (timed sweeps included): with enough warm-up sweeps the timed region is the stationary regime")
    ap.add_argument("--stationary-budget-s", type=float, default=120.0, help="skip the stationary-environment block where its 14 + 3 sweeps would take longer than this")
    ap.add_argument("--no-stationary", action="store_true", help="skip the stationary-environment block of the full-rank state (projector_warm_tol fast path)")
    ap.add_argument("--no-energy", action="store_true", help="skip the energy block (E/site from rdm2x2 at the size of the timed run)")
    ap.add_argument("--live-traffic-full-rank", action="store_true", help="measure the HBM traffic of the full-rank block by child runs under rocprofv3 too (default: the primary block only)")
    ap.add_argument("--energy", action="store_true", help="the energy block also for a --config other than the default one")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact blocks of the other single-GPU BASELINE configurations")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC pass instead of two rocprofv3 --pmc child runs of this command")
    args = ap.parse_args()
    kind, D, chi, dtype = CONFIGS[args.config]
    steps = args.steps if args.steps is not None else (100 if kind == "c4v" else 2)
    # untimed warm-up: ceil(chi / D^2) sweeps, the number after which the environment from the CTMRG init has filled its chi
    # (reference ctmrg.py:81)
    warmup = args.warmup if args.warmup is not None else (3 if kind == "c4v" else -(-chi // (D * D)))

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("CTM_BENCH_ONE_DEVICE"):      # development hook: several ranks on ONE GPU (gloo), to exercise the sharded path
        local = 0
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("CTM_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
        world, rank = dist.get_world_size(), dist.get_rank()        # n_gpus of the JSON line = size of the process group that ran

    import config as cfg
    cfg.global_args.device = f"cuda:{local}"
    cfg.ctm_args.concurrent_units = not args.serial_units
    cfg.ctm_args.projector_warm_start = not args.cold_start
    cfg.ctm_args.projector_warm_tol = float(args.warm_tol)
    import _native
    eng = _native.engine()
    for kv in args.opt:
        k_, v_ = kv.split("="); eng.set_option(k_, float(v_))
    dev = torch.device("cuda", local)

    primary_signed = bool(args.signed)
    res, dom, sites, state, env = run_workload(args, eng, dev, kind, D, chi, dtype, primary_signed, steps, warmup, world, rank, dist)
    # (the child needs the workload's memory a second time: single-GPU float64 workloads up to n = 16384)
    live = world == 1 and rank == 0 and not args.no_live_traffic and kind != "c4v" and dtype == "f64" and chi * D * D <= 16384
    tr = (traffic_live(args, dom, primary_signed, min(warmup, 2)) if live else None) or traffic_from_profile(args, world, dom, primary_signed)
    if tr:
        res["roofline"]["traffic"] = tr["dominant"]; res["roofline"]["traffic_source"] = tr["source"]
        for i, dsc in res["roofline"]["other_kernels"].items():
            dsc["traffic"] = tr["others"].get(i)
    full = None
    if kind != "c4v" and not primary_signed and not args.no_full_rank:
        # second required block: the SAME shape on a full-rank state (the workload real iPEPS produce lies between the two)
        env_np = None
        if rank == 0 and not args.no_cpu_baseline and world == 1:
            env_np = ({k: v.cpu().numpy() for k, v in env.C.items()}, {k: v.cpu().numpy() for k, v in env.T.items()})
        del state, env
        import gc; gc.collect(); torch.cuda.empty_cache()
        full, fdom, fsites, fstate, fenv = run_workload(args, eng, dev, kind, D, chi, dtype, True, steps, warmup, world, rank, dist)
        # (the full-rank block takes its traffic from the committed PMC pass unless asked: two more child runs of the 50 000-launch sweep
        # under counter collection were 2.5 of the 11 minutes of the default command)
        ftr = (traffic_live(args, fdom, True, min(warmup, 2)) if live and args.live_traffic_full_rank else None) or traffic_from_profile(args, world, fdom, True)
        if ftr:
            full["roofline"]["traffic"] = ftr["dominant"]; full["roofline"]["traffic_source"] = ftr["source"]
        if not args.no_energy and (args.config == DEFAULT_CONFIG or args.energy):
            try:
                full["energy"] = energy_block(eng, fstate, fenv, world)
            except Exception as e:                        # reporting only
                full["energy"] = {"error": repr(e)}
        del fstate, fenv
        gc.collect(); torch.cuda.empty_cache()
    else:
        env_np = None
        if kind != "c4v" and rank == 0 and not args.no_cpu_baseline and world == 1:
            env_np = ({k: v.cpu().numpy() for k, v in env.C.items()}, {k: v.cpu().numpy() for k, v in env.T.items()})
    signed_c4v = None
    if kind == "c4v" and not primary_signed and not args.no_full_rank and world == 1:
        # the same C4v shape on signed random tensors: the environment keeps moving for ~20 sweeps instead of ~7 and the corner's
        # spectrum decays more slowly (the truncation contracts by 0.3 per application instead of 0.01)
        try:
            signed_c4v = compact(run_workload(args, eng, dev, kind, D, chi, dtype, True, steps, warmup, world, rank, dist)[0])
        except Exception as e:                            # reporting only
            signed_c4v = {"error": repr(e)}

    others = None
    if args.config == DEFAULT_CONFIG and not primary_signed and not args.no_other_configs and world == 1 and not args.no_full_rank:
        del sites
        others = other_configs(args, eng, dev, world, rank, dist)
        sites = synth_sites(kind, D, dtype=dtype)                 # (the CPU baseline below takes the site tensors of the primary block)
    if rank == 0:
        out = {"metric": "ctm_sweeps_per_sec", "value": res["value"], "unit": "sweeps/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": {"workload": args.config, "variant": kind, "D": D, "chi": chi, "n": chi * D * D,
                          "state": "A ~ U(-1,1) (full rank)" if primary_signed else "A ~ U[0,1) (reference scripts, SURVEY 8d)",
                          "unit_cell": "1x1 C4v" if kind == "c4v" else "2x2 (4 sites, 32 units/sweep)",
                          "parallelism": f"site-sharded x{world}" if world > 1 else "single GPU"},
               "roofline": res["roofline"], "svd": res["svd"], "phase_s": res["phase_s"], "phase_s_note": res["phase_s_note"]}
        if "state" in res:
            out["state"] = res["state"]
        for k_ in ("steady_state", "moving_environment", "stationary_environment"):
            if k_ in res:
                out[k_] = res[k_]
        if full is not None:
            out["full_rank"] = {"metric": "ctm_sweeps_per_sec", "value": full["value"], "unit": "sweeps/s", "ms_per_step": full["ms_per_step"],
                                "steps": full["steps"], "warmup": full["warmup"],
                                "config": {"workload": args.config, "state": "A ~ U(-1,1), A /= max|A| (signed random tensors)", "n": chi * D * D},
                                "state": full["state"], "roofline": full["roofline"], "svd": full["svd"], "phase_s": full["phase_s"]}
            if "stationary_environment" in full:
                out["full_rank"]["stationary_environment"] = full["stationary_environment"]
        if full is not None and "energy" in full:
            out["full_rank"]["energy"] = full["energy"]
        if signed_c4v is not None:
            out["signed_state"] = signed_c4v
        if others is not None:
            out["other_configs"] = others
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(kind, D, chi, sites, env_np, svd_n=args.cpu_svd_n)
            except Exception as e:                      # the baseline is reporting only
                out["cpu_baseline"] = {"error": repr(e)}
            try:
                out["energy_parity"] = energy_parity(dev, dtype)
            except Exception as e:
                out["energy_parity"] = {"error": repr(e)}
            if full is not None:
                # the metric's second half where the block Krylov solver actually runs: the largest full-rank size the oracle finishes
                # in about a minute (D = 4 chi = 64 signed, n = 1024, 3 sweeps + four plaquette RDMs on the host)
                try:
                    out["full_rank"]["energy_parity"] = energy_parity(dev, dtype, D=4, chi=64, nsweeps=3, signed=True)
                except Exception as e:
                    out["full_rank"]["energy_parity"] = {"error": repr(e)}
        line = metric_line(out)
        # the verbose blocks (per-class rooflines, serial passes, every other configuration in full) go to stderr and to a file; stdout
        # carries ONE line that fits the driver's record
        try:
            os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
            with open(os.path.join(REPO, "gpurun_out", "bench_detail.json"), "w") as f:
                f.write(json.dumps(out) + "\n")
            line["detail"] = "gpurun_out/bench_detail.json (and stderr, prefix BENCH_DETAIL)"
        except OSError:
            line["detail"] = "stderr, prefix BENCH_DETAIL"
        print("BENCH_DETAIL " + json.dumps(out), file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the metric line is the LAST thing on stdout: RCCL announces itself through C stdio ("Librccl path : ..."), which sits in a
        # buffer until it is flushed -- at exit, i.e. after a line printed here -- unless it is flushed first
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
