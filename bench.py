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

import numpy as np
import torch

CONFIGS = {
    # name: (kind, D, chi, dtype)          BASELINE.json configs[1..4]
    "c4v_D4_chi64": ("c4v", 4, 64, "f64"),
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


def synth_sites(kind, D, seed=1, dtype="f64"):
    rng = np.random.default_rng(seed)
    if dtype == "c128":      # re and im parts each U[0,1)
        sites = {}
        for y in range(2):
            for x in range(2):
                A = rng.random((2, D, D, D, D)) + 1j * rng.random((2, D, D, D, D))
                sites[(x, y)] = A / np.abs(A).max()
        return sites
    if kind == "c4v":
        from groups.pg import make_c4v_symm                      # the host layer's own symmetriser (reference groups/pg.py:27-63)
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


def cpu_baseline(kind, D, chi, sites, budget_s=25.0):
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
    env = O.init_env_ctmrg(ost, chi)
    if chi * D * D > 6000:
        return cpu_baseline_large(O, ost, env, D, chi, threads)
    # make the environment dense with the cheapest possible warm-up: random dense env of the right shapes
    rng = np.random.default_rng(7)
    cx = np.iscomplexobj(sites[(0, 0)])
    for k in env.C: env.C[k] = rng.random(env.C[k].shape) + (1j * rng.random(env.C[k].shape) if cx else 0.0)
    for k in env.T: env.T[k] = rng.random(env.T[k].shape) + (1j * rng.random(env.T[k].shape) if cx else 0.0)
    t0 = time.perf_counter()
    P, Pt = O.get_projectors_4x4(O.UP, (0, 0), ost, env)
    Pd = {c: P for c in ost.sites}; Ptd = {c: Pt for c in ost.sites}
    O.absorb_truncate(O.UP, (0, 0), ost, env, Pd, Ptd)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (32 * dt), "unit": "sweeps/s", "cores": threads, "kind": "port",
            "sample": f"1 of the 32 (site,direction) units of one sweep (numpy oracle: 4 corners, 2 halves, M, LAPACK gesdd, "
                      f"projectors, absorb) = {dt:.2f} s, extrapolated x32; {os.cpu_count()} host cpus"}


def energy_parity(dev, dtype="f64", D=3, chi=36, nsweeps=3):
    """rdm2x2 energy (J1-J2, j2 = 0.5) after `nsweeps` sweeps from the CTMRG init: native engine vs the numpy oracle on the SAME
    synthetic state, at a size the oracle finishes in seconds (the second half of BASELINE.json's metric)."""
    import config as cfg
    from oracle import ctm_oracle as O, j1j2_oracle as OJ
    from ipeps.ipeps import IPEPS
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from models import j1j2
    sites = synth_sites("generic", D, seed=3, dtype=dtype)
    st = IPEPS({k: torch.from_numpy(v).to(dev) for k, v in sites.items()})
    env = ENV(chi, st); init_env(st, env)
    for _ in range(nsweeps):
        for d in cfg.ctm_args.ctm_move_sequence:
            for _r in range(2):
                ctmrg.ctm_MOVE(d, st, env)
    e = float(j1j2.J1J2(j1=1.0, j2=0.5).energy_per_site(st, env))
    spec = {k: v.cpu().numpy() for k, v in env.get_spectra().items()}
    ost = O.State(sites); oe = O.init_env_ctmrg(ost, chi)
    for _ in range(nsweeps):
        O.ctm_sweep(ost, oe)
    eo = OJ.energy_per_site([O.rdm2x2(c, ost, oe) for c in ost.sites], 1.0, 0.5)
    ospec = O.corner_spectra(oe)
    return {"workload": f"generic 2x2 D={D} chi={chi} {dtype}, {nsweeps} sweeps from the CTMRG init, J1-J2 j2=0.5", "energy_native": e,
            "energy_oracle": float(eo), "rel_err": abs(e - eo) / max(abs(eo), 1e-300),
            "max_abs_err_corner_spectra": float(max(np.abs(spec[k] - ospec[k]).max() for k in ospec)), "tolerance": 1e-10}


def cpu_baseline_large(O, ost, env, D, chi, threads):
    """n = chi D^2 > 6000: one full unit on the CPU takes many minutes (LAPACK gesdd of an n x n matrix), so the unit is
    assembled from bounded pieces: the four enlarged corners and the absorb are timed at full size with the oracle, ONE
    n x n x n product is timed and counted three times (two halves + M = R^T Rt), and the SVD is timed at n_s = 4096 and
    scaled by (n / n_s)^3."""
    rng = np.random.default_rng(7)
    cx = np.iscomplexobj(next(iter(ost.sites.values())))
    for k in env.C: env.C[k] = rng.random(env.C[k].shape) + (1j * rng.random(env.C[k].shape) if cx else 0.0)
    for k in env.T: env.T[k] = rng.random(env.T[k].shape) + (1j * rng.random(env.T[k].shape) if cx else 0.0)
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
            "sample": f"one (site,direction) unit of the numpy oracle assembled from bounded pieces: 4 corners {t_corners:.1f} s + 3 x (n^3 GEMM "
                      f"{t_gemm:.1f} s) + gesdd at n_s=4096 {t_svd_s:.1f} s scaled by (n/n_s)^3 = {t_svd:.0f} s + projector GEMMs {t_proj:.1f} s + "
                      f"absorb {t_abs:.1f} s = {dt:.0f} s/unit, x32 units/sweep (extrapolated); {os.cpu_count()} host cpus"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default=DEFAULT_CONFIG, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile", action="store_true", help="per-phase host timers (adds stream syncs)")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (development)")
    ap.add_argument("--signed", action="store_true", help="development: A ~ U(-1,1) instead of the reference's U[0,1) (flatter spectra)")
    ap.add_argument("--serial-units", action="store_true", help="do not overlap the independent site-units of a move on streams")
    ap.add_argument("--cold-start", action="store_true", help="no warm start of the leading-chi iteration")
    args = ap.parse_args()
    kind, D, chi, dtype = CONFIGS[args.config]
    steps = args.steps if args.steps is not None else (20 if kind == "c4v" else 2)
    # untimed warm-up: ceil(chi / D^2) sweeps, the number after which the environment from the CTMRG init has filled its chi
    # (reference ctmrg.py:81)
    warmup = args.warmup if args.warmup is not None else (3 if kind == "c4v" else -(-chi // (D * D)))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if os.environ.get("CTM_BENCH_ONE_DEVICE"):      # development hook: several ranks on ONE GPU (gloo), to exercise the sharded path
        local = 0
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("CTM_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    import config as cfg
    cfg.global_args.device = f"cuda:{local}"
    cfg.ctm_args.concurrent_units = not args.serial_units
    cfg.ctm_args.projector_warm_start = not args.cold_start
    import _native
    from ipeps.ipeps import IPEPS
    from ipeps.ipeps_c4v import IPEPS_C4V
    from ctm.generic.env import ENV, init_env
    from ctm.generic import ctmrg
    from ctm.one_site_c4v.env_c4v import ENV_C4V, init_env as init_env_c4v
    from ctm.one_site_c4v import ctmrg_c4v
    eng = _native.engine()
    for kv in args.opt:
        k_, v_ = kv.split("="); eng.set_option(k_, float(v_))
    dev = torch.device("cuda", local)
    sites = synth_sites(kind, D, dtype=dtype)
    if args.signed:
        sites = {k: (2.0 * v - (1.0 + 1.0j if np.iscomplexobj(v) else 1.0)) for k, v in sites.items()}
        sites = {k: v / np.abs(v).max() for k, v in sites.items()}
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
    eng.set_option("gemm_timing", 1)
    eng.timers(reset=True)
    if args.profile:
        eng.set_option("profile", 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ivals = eng.gemm_intervals()
    KK = range(4)
    k_ms = [eng.stat(f"k_ms{i}") for i in KK]
    k_fl = [eng.stat(f"k_flops{i}") for i in KK]
    k_n = [eng.stat(f"k_calls{i}") for i in KK]
    eng.set_option("gemm_timing", 0)
    # reference pass for the kernel-quality figure: ONE more sweep with the units issued serially on one stream (nothing
    # co-scheduled), outside the timed region -- per-launch rates of the same kernels without sharing the chip
    serial = None
    if world == 1 and kind != "c4v" and not args.serial_units:
        cfg.ctm_args.concurrent_units = False
        eng.set_option("gemm_timing", 1)
        step(); fence()
        serial = ([eng.stat(f"k_ms{i}") for i in KK], [eng.stat(f"k_flops{i}") for i in KK], [eng.stat(f"k_calls{i}") for i in KK])
        eng.set_option("gemm_timing", 0)
        cfg.ctm_args.concurrent_units = True
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        # instrumented kernel classes (csrc: timing_end kinds).  Class 3 (streaming strip kernel) is HBM-bound: the engine
        # reports its ALGORITHMIC BYTES (the big operand read once + the block in and out) where the others report flops.
        CLS = {0: ("gemm_f64_fast_kernel / gemm_f64_kernel<4,4> (128x128 tile)", "mfma", "gemm_f64_fast_kernel"),
               1: ("gemm_f64_kernel<2,2> (64x64 tile: small, segmented and batched products)", "mfma", "gemm_f64_kernel<2, 2>"),
               2: ("layer2_reg_kernel<KT> (float64) / layer2_c_kernel<KT> (complex128): fused two-layer enlarged-corner kernel", "mfma", "layer2_"),
               3: ("gemm_strip_kernel<TM,BNF>: <= 64-row block times an n x n corner, streamed once", "hbm", "gemm_strip_kernel")}
        # Independent units run on concurrent streams, so launches of the kernel overlap: the time the chip spends on
        # them is the UNION of their [start, end] intervals (equal to the sum of durations when nothing overlaps).
        def union_ms(kind):
            iv = sorted((a, b) for k_, a, b, _ in ivals if int(k_) == kind)
            tot, cur_a, cur_b = 0.0, None, None
            for a, b in iv:
                if cur_b is None or a > cur_b:
                    if cur_b is not None: tot += cur_b - cur_a
                    cur_a, cur_b = a, b
                else:
                    cur_b = max(cur_b, b)
            return tot + ((cur_b - cur_a) if cur_b is not None else 0.0)
        u_ms = [union_ms(i) for i in KK]
        HBM_PEAK_GBPS = 8000.0

        def rate(work, ms, bound):              # TFLOP/s or GB/s
            return work / max(ms * 1e-3, 1e-30) / (1e12 if bound == "mfma" else 1e9)

        def describe(i, kms, kfl, kn, ums=None):
            name, bound, _ = CLS[i]
            peak = FP64_MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBPS
            busy = ums[i] if ums is not None else kms[i]
            ach = rate(kfl[i], busy, bound) if kn[i] else 0.0
            d = {"kernel": name, "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                 "frac": round(ach / peak, 4), "launches": int(kn[i]), "avg_launch_ms": round(kms[i] / max(kn[i], 1), 5),
                 "sum_launch_ms": round(kms[i], 3)}
            if ums is not None:
                d["busy_ms_union"] = round(ums[i], 3)
            return d
        # the dominant kernel = the class the chip spends most time on
        dom = max((0, 2, 3), key=lambda i: u_ms[i])
        roof = describe(dom, k_ms, k_fl, k_n, u_ms)
        roof = {"bound": roof.pop("bound"), **roof, "traffic": None,
                "per_launch_rate_while_sharing_the_chip": round(rate(k_fl[dom], k_ms[dom], CLS[dom][1]), 3) if k_n[dom] else 0.0,
                "concurrent_streams": max(1, len(getattr(eng, "workers", []))),
                "time_share_of_sweep": round(u_ms[dom] * 1e-3 / dt, 4),
                "algorithmic_work_per_launch": ("2*M*N*K flop of each product" if dom in (0, 1) else
                                                "2 * p * chi_x * chi_y * (D^2)^2 * 2 D^2 flop per launch (both layers)" if dom == 2 else
                                                "8 * (K*N + M*K + M*N) bytes: the n x n corner read once, the <=64-row block in and out"),
                "other_kernels": {str(i): describe(i, k_ms, k_fl, k_n, u_ms) for i in KK if i != dom and k_n[i]}}
        if serial is not None:
            s_ms, s_fl, s_n = serial
            roof["serial_pass"] = {"note": "one extra sweep after the timed region, units issued serially on one stream (no co-scheduling): kernel quality without sharing the chip",
                                   **{("dominant" if i == dom else str(i)): describe(i, s_ms, s_fl, s_n) for i in KK if s_n[i]}}
        # HBM traffic of the dominant kernel family from the committed PMC pass of this same command (bench.py cannot attach
        # counters to itself); null when the profile is for another workload
        try:
            import csv
            prof = json.load(open(os.path.join(REPO, "profiles", "r01_bench_default.json")))
            if prof["config"]["workload"] == args.config and world == 1 and not args.signed:
                rows = list(csv.DictReader(open(os.path.join(REPO, "profiles", "r01_bench_default_pmc_hbm_traffic.csv"))))

                def pmc(key):
                    tb = tn = 0.0
                    for row in rows:
                        if key in row["kernel"]:
                            tb += float(row["hbm_bytes_per_launch(2x_fetch_corrected)"]) * float(row["launches"]); tn += float(row["launches"])
                    return round(tb / tn) if tn else None
                roof["traffic"] = pmc(CLS[dom][2])
                roof["traffic_source"] = "profiles/r01_bench_default_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, average HBM bytes per launch)"
                for i, dsc in roof["other_kernels"].items():
                    dsc["traffic"] = pmc(CLS[int(i)][2])
        except Exception:
            pass
        out = {"metric": "ctm_sweeps_per_sec", "value": steps / dt, "unit": "sweeps/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": {"workload": args.config, "variant": kind, "D": D, "chi": chi, "n": chi * D * D,
                          "unit_cell": "1x1 C4v" if kind == "c4v" else "2x2 (4 sites, 32 units/sweep)",
                          "parallelism": f"site-sharded x{world}" if world > 1 else "single GPU"},
               "roofline": roof,
               "svd": {"decompositions": int(eng.stat("jacobi_calls")),
                       "avg_jacobi_sweeps": round(eng.stat("total_sweeps") / max(eng.stat("jacobi_calls"), 1), 2),
                       "power_iter_hits": int(eng.stat("si_hits")), "power_iter_fallbacks_to_full": int(eng.stat("si_fallbacks")),
                       "avg_half_steps": round(eng.stat("si_total_iters") / max(eng.stat("si_hits") + eng.stat("si_fallbacks"), 1), 2)},
               "phase_s": {k: round(v, 4) for k, v in eng.timers().items()}}
        if kind != "c4v":
            # what the engine could exploit on THIS state (caveat): numerical rank of the truncation and
            # the number of projector columns above projector_svd_reltol, out of chi
            nc = env.__dict__.get("_ncol") or {}
            wr = [w.stat("si_last_rank") for w in getattr(eng, "workers", [])]
            out["state"] = {"chi": chi, "numerical_rank_of_truncated_operator": int(max(wr + [eng.stat("si_last_rank") - sum(wr)])),
                            "nonzero_projector_columns": (max(nc.values()) if nc else None),
                            "corner_cache_hits": int(eng.stat("corner_cache_hits")),
                            "note": "positive random tensors give a numerically low-rank environment; see --signed for the full-rank extreme"
                                    if not args.signed else "signed random tensors: full-rank spectrum"}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(kind, D, chi, sites)
            except Exception as e:                      # the baseline is reporting only
                out["cpu_baseline"] = {"error": repr(e)}
            try:
                out["energy_parity"] = energy_parity(dev, dtype)
            except Exception as e:
                out["energy_parity"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
