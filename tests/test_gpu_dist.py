"""GPU: the site-sharded move on two ranks (one process per rank, gloo, both ranks on this one GPU) against one process --
at n = 8192, where the corner cache, the cache-aware unit ownership and the masked-column absorb are all active.  The
environments must agree exactly (same kernels, same operands; sharding only decides who computes which unit)."""
import json, os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp, nranks, port):
    out = os.path.join(tmp, f"d{nranks}.json")
    env = dict(os.environ, CTM_BENCH_ONE_DEVICE="1", CTM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    tool = os.path.join(REPO, "tools", "check_dist_gpu.py")
    if nranks == 1:
        cmd = [sys.executable, tool, out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), tool, out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.load(open(out))


def test_two_ranks_on_one_gpu_equal_one_process(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    a = _run(str(tmp_path), 1, port)
    b = _run(str(tmp_path), 2, port)
    assert b["ncol"] and max(b["ncol"].values()) <= 64            # the masked-column absorb was active on the ranks
    for k in a:
        if k in ("checksum", "ncol"):
            continue
        assert a[k] == b[k], k
    assert a["checksum"] == b["checksum"]


@pytest.mark.parametrize("name", ["generic_ad_D2_chi8_f64", "generic_ad_D2_chi8_c128"])
def test_differentiable_moves_on_two_ranks_give_the_reference_gradient(tmp_path, name):
    """The differentiable route sharded over two ranks (gloo, both on this GPU): exchanges as autograd nodes, local gradients
    averaged; energy and gradient on both ranks against the reference's single-process autograd (tests/golden/generic_ad_*.npz)."""
    import socket
    import numpy as np
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = os.path.join(str(tmp_path), "ad")
    env = dict(os.environ, CTM_BENCH_ONE_DEVICE="1", CTM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tools", "check_dist_gpu_ad.py"), out, name]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    g = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
    for rank in range(2):
        o = np.load(out + f".rank{rank}.npz")
        assert abs(float(o["energy"]) - float(g["energy"])) < 1e-11
        for k in o.files:
            if k.startswith("grad_"):
                assert float(np.abs(o[k] - g[k]).max()) < 1e-9 * max(1.0, float(np.abs(g[k]).max())), (rank, k)


def test_rccl_backend_executes_every_collective_on_one_rank():
    """backend "nccl" (RCCL) with a one-rank process group on this GPU: tools/check_rccl_world1.py drives every collective of parallel.py
    on device buffers (both dtypes, both branches of exchange(), sub-groups, autograd exchange), then two sharded sweeps + energy
    bit-identical to the run without a process group, and the communicator entry of the C-ABI."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_rccl_world1.py"), str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and len(out["checks"]) == 2
    assert "needs a communicator" in out["set_comm_two_ranks"]
    assert out["rccl_shared_passes_one_rank"] > 0           # ncclAllGather on the library's stream inside the truncation of a unit, same bits
    assert out["comm_s_torch.float64"] > 0.0 and out["comm_s_torch.complex128"] > 0.0


def test_a_unit_shared_by_a_pair_of_ranks_equals_the_unsplit_solve(tmp_path):
    """Twice as many ranks as sites (8 GPUs on a 4-site cell; here 2 gloo ranks on this GPU, one-site cell): the pair shares the
    unit -- every corner pass of its truncation computes this rank's half of the output columns inside the native solver and the halves are
    all-gathered (ctm_set_comm_ops, host-driven; csrc/svd_leading.hip: rows_times_shared).  Three sweeps of a signed D = 4 chi = 48 state
    (n = 768: block Krylov solves): both ranks end with the SAME bits, equal to the one-process run to rounding (the half-width products
    slice K differently), and the passes really were shared."""
    import socket
    import numpy as np
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, CTM_BENCH_ONE_DEVICE="1", CTM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    tool = os.path.join(REPO, "tools", "check_pair_split.py")
    one = os.path.join(str(tmp_path), "one"); two = os.path.join(str(tmp_path), "two")
    r = subprocess.run([sys.executable, tool, one], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), tool, two], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a = json.load(open(one + ".rank0.json"))
    b0, b1 = json.load(open(two + ".rank0.json")), json.load(open(two + ".rank1.json"))
    assert a["shared_passes"] == 0 and b0["shared_passes"] > 100 and b0["shared_passes"] == b1["shared_passes"]
    assert b0["krylov_solves"] > 0
    for k in a:
        if k in ("shared_passes", "krylov_solves", "power_iteration_solves"):
            continue
        assert b0[k] == b1[k], k                                         # the pair: bit for bit
        if k == "checksum":
            assert abs(a[k] - b0[k]) < 1e-9 * abs(a[k])
        else:
            assert np.abs(np.array(a[k]) - np.array(b0[k])).max() < 1e-11, k
