"""GPU: `bench.py --gpus N` end to end, as the driver launches it, before a multi-GPU node ever does: N ranks under torch.distributed.run
sharing the ONE GPU of the test box (development hook CTM_BENCH_ONE_DEVICE=1, gloo instead of RCCL: the collectives of parallel.py are
backend-agnostic; RCCL itself is executed by test_gpu_dist.py::test_rccl_backend_executes_every_collective_on_one_rank).  Asserted: ONE
parseable JSON line is the last thing on stdout, it carries n_gpus = N, the exchanges were timed (phase_s.comm > 0), and the environment
the sharded sweeps produce is the single-process one (checksum of the corner spectra)."""
import json, os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--config", "generic_D6_chi128", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-energy", "--no-live-traffic",
         "--no-serial-pass", "--no-stationary"]      # (both add sweeps to the single-process run only: the environments would differ by them)


def _bench(n, flags=None):
    env = dict(os.environ)
    env.update({"CTM_BENCH_ONE_DEVICE": "1", "CTM_BENCH_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n)] + (FLAGS if flags is None else flags), cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    line = json.loads(lines[-1])                       # the LAST stdout line is the metric line (the driver's parser takes that one)
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1, "more than one JSON line on stdout"
    return line


@pytest.fixture(scope="module")
def single():
    return _bench(1)


# (8 ranks on a 4-site cell: the pairs {i, i + 4} SHARE unit i -- every corner pass split in the pair and all-gathered, host-driven over gloo
#  here, i.e. a stream drain and a host round trip per pass: at configs[2] size that was 119 s of this file, so the 8-rank case runs the
#  n = 324 configuration (still the iterative route: every application of the operator is four shared passes); `--soak` adds the configs[2] one)
SMALL = ["--config", "generic_D3_chi36", "--signed"] + FLAGS[2:]          # (the signed state only: `value` is the full-rank state then)


@pytest.fixture(scope="module")
def single_small():
    return _bench(1, SMALL)


@pytest.mark.parametrize("n,small", [(2, False), (4, False), (8, True), pytest.param(8, False, marks=pytest.mark.soak)],
                         ids=["2", "4", "8-n324", "8"])
def test_bench_with_n_ranks_on_one_device(single, single_small, n, small):
    line = _bench(n, SMALL if small else None)
    if small:
        single = single_small
    # twice as many ranks as sites: the units were shared by rank pairs (rank 0's count of split corner passes)
    blk = line if small else line["full_rank"]
    assert (blk["svd"].get("shared_corner_passes", 0) > 0) == (n == 8), blk["svd"]
    assert line["n_gpus"] == n and single["n_gpus"] == 1
    assert line["metric"] == "ctm_sweeps_per_sec" and line["value"] > 0 and line["scaling"] == "strong"
    assert line["steps"] == 1 and line["warmup"] == 1
    assert line["phase_s"]["comm"] > 0, "the exchanges of the sharded moves were not timed"
    assert line["roofline"].get("comm_per_rank"), line["roofline"].keys()
    # same environment as the single-process run: both states of the workload (the 8-rank case times the signed state only)
    a, b = single["state"]["corner_spectra_checksum"], line["state"]["corner_spectra_checksum"]
    assert abs(a - b) <= 1e-9 * abs(a), (a, b)
    if not small:
        a, b = single["full_rank"]["state"]["corner_spectra_checksum"], line["full_rank"]["state"]["corner_spectra_checksum"]
        assert abs(a - b) <= 1e-9 * abs(a), (a, b)


EFLAGS = ["--config", "generic_D6_chi128", "--energy", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-live-traffic",
          "--no-serial-pass", "--no-stationary"]


@pytest.fixture(scope="module")
def single_with_energy():
    return _bench(1, EFLAGS)


@pytest.mark.parametrize("n", [2, pytest.param(8, marks=pytest.mark.soak)])
def test_energy_block_of_the_line_with_n_ranks(single_with_energy, n):
    """After the full-rank sweeps the driver's command evaluates the energy once: with more than one rank the plaquette RDMs of the four
    sites are shared out -- by site with 2 ranks, and with 8 ranks (more ranks than sites) the p^4 slices of a site's plaquette among the
    ranks of its group, one all-reduce inside the group (parallel.site_groups, rdm._rdm2x2_raw) -- and reduced.  N ranks on the one
    device against the single process (configs[2] shape, so that eight ranks fit one GPU): same environment, same energy."""
    one, many = single_with_energy, _bench(n, EFLAGS)
    assert many["n_gpus"] == n
    for blk in (one, many):
        assert "error" not in blk["full_rank"]["energy"], blk["full_rank"]["energy"]
    e1, e2 = one["full_rank"]["energy"], many["full_rank"]["energy"]
    assert e1["n_gpus"] == 1 and e2["n_gpus"] == n
    a, b = e1["energy_per_site_j2_0.5"], e2["energy_per_site_j2_0.5"]
    assert abs(a - b) <= 1e-10 * abs(a), (a, b)
    a, b = one["full_rank"]["state"]["corner_spectra_checksum"], many["full_rank"]["state"]["corner_spectra_checksum"]
    assert abs(a - b) <= 1e-9 * abs(a), (a, b)
