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


def _bench(n):
    env = dict(os.environ)
    env.update({"CTM_BENCH_ONE_DEVICE": "1", "CTM_BENCH_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n)] + FLAGS, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    line = json.loads(lines[-1])                       # the LAST stdout line is the metric line (the driver's parser takes that one)
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1, "more than one JSON line on stdout"
    return line


@pytest.fixture(scope="module")
def single():
    return _bench(1)


@pytest.mark.parametrize("n", [2, 4])
def test_bench_with_n_ranks_on_one_device(single, n):
    line = _bench(n)
    assert line["n_gpus"] == n and single["n_gpus"] == 1
    assert line["metric"] == "ctm_sweeps_per_sec" and line["value"] > 0 and line["scaling"] == "strong"
    assert line["steps"] == 1 and line["warmup"] == 1
    assert line["phase_s"]["comm"] > 0, "the exchanges of the sharded moves were not timed"
    assert line["roofline"].get("comm_per_rank"), line["roofline"].keys()
    # same environment as the single-process run: both states of the workload
    a, b = single["state"]["corner_spectra_checksum"], line["state"]["corner_spectra_checksum"]
    assert abs(a - b) <= 1e-9 * abs(a), (a, b)
    a, b = single["full_rank"]["state"]["corner_spectra_checksum"], line["full_rank"]["state"]["corner_spectra_checksum"]
    assert abs(a - b) <= 1e-9 * abs(a), (a, b)
