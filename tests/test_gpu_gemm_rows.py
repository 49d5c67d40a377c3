"""Shape sweep of the two kernels behind `(<= 64 rows) x (big operand)` (csrc/gemm_f64.hip: gemm_rows_kernel, gemm_strip_kernel) through
the C-ABI entry `ctm_gemm`, each product against a host fp64 product.

The K-slice rule of the dispatcher (`rows_target_wgs`, `rows_min_klen`) is driven through every epilogue of the row-block kernel:
a single slice (direct write: alpha, beta, ldc > N), two slices, the default rule, odd slice lengths with a short last slice, the
in-launch combine and the separate reduce kernel -- on both layouts of the big operand.  (Round 3 ended one full test run with
`rows_min_klen=576` in a core dump; this is the sweep that run did not have.  It passes, and so does the whole suite under
AddressSanitizer with that value: the option is cleared and is the default since round 4.  The crash itself recurred once and
sits elsewhere -- inside ctm_svd_backward of a 25 x 25 complex gradcheck)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (N, K): N % 128 == 0, K % 16 == 0, K >= 1024, N * K >= 2^22 (what the dispatcher sends to the row-block kernel)
    (128, 32768), (256, 16384), (1152, 4608), (4608, 1024), (4608, 1040), (4608, 1152), (4608, 4608), (2048, 2064), (1024, 16384),
    (4224, 1008 + 16 * 9),
]
ROWS = [16, 17, 33, 48, 64]


def _host(a, b):
    return torch.from_numpy(a.cpu().numpy() @ b.cpu().numpy())


DEFAULTS = {"rows_target_wgs": 512, "rows_min_klen": 576, "rows_fused_reduce": 1, "rows_kernel_min_m": 1, "rows_kernel_min_m_kc": 1}   # csrc/ctm_common.h


@pytest.fixture()
def opts(eng):
    before = dict(eng._options)          # (a run may carry CTM_ENGINE_OPTS)
    yield eng
    for k, v in DEFAULTS.items():
        eng.set_option(k, before.get(k, v))
        if k not in before:
            eng._options.pop(k, None)


@pytest.mark.parametrize("N,K", SHAPES)
def test_row_block_kernel_shape_sweep(opts, N, K):
    eng = opts
    g = torch.Generator(device="cuda").manual_seed(N * 7 + K)
    B = torch.randn(K, N, dtype=torch.float64, device="cuda", generator=g)
    Bt = B.t().contiguous()
    A64 = torch.randn(64, K, dtype=torch.float64, device="cuda", generator=g)
    gx = N // 128
    for M in ROWS:
        A = A64[:M].contiguous()
        ref = _host(A, B)
        scale = float(ref.abs().max())
        C0 = torch.randn(M, N + 6, dtype=torch.float64, device="cuda", generator=g)      # ldc > N: the columns beyond N must survive
        for klen in (256, 576):
            eng.set_option("rows_min_klen", klen)
            # one slice (direct-write epilogue), two slices, three (odd lengths), the default, many
            for target in (1, 2 * gx, 3 * gx, 768, 64 * gx):
                eng.set_option("rows_target_wgs", target)
                for fused in (1, 0):
                    eng.set_option("rows_fused_reduce", fused)
                    for tB, Bop in ((False, B), (True, Bt)):
                        what = (M, N, K, klen, target, fused, tB)
                        out = eng.gemm(A, Bop, transB=tB)
                        assert float((out.cpu() - ref).abs().max()) <= 1e-12 * scale * np.sqrt(K), what
                        # alpha, beta and a padded output
                        C = C0.clone()
                        eng.gemm(A, Bop, transB=tB, alpha=-0.5, beta=2.0, out=C)
                        exp = 2.0 * C0[:, :N].cpu() - 0.5 * ref
                        assert float((C[:, :N].cpu() - exp).abs().max()) <= 1e-12 * (scale + 2.0 * 5.0) * np.sqrt(K), what
                        assert torch.equal(C[:, N:], C0[:, N:]), what


def test_row_block_kernel_slice_counts_are_bitwise_consistent_between_combine_and_reduce(opts):
    """Same partials, same summation order: the in-launch combine and the separate reduce kernel agree bit for bit for every slice
    count from 2 up (one slice included: both then write alpha * acc), repeatedly, on a mid-size operand (n = 4608: the D = 6 chi = 128 corner)."""
    eng = opts
    g = torch.Generator(device="cuda").manual_seed(11)
    n = 4608
    B = torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g)
    for M in (32, 64):
        A = torch.randn(M, n, dtype=torch.float64, device="cuda", generator=g)
        for klen in (256, 576, 1152, 4608):
            eng.set_option("rows_min_klen", klen)
            eng.set_option("rows_target_wgs", 768)
            eng.set_option("rows_fused_reduce", 0)
            r0, r1 = eng.gemm(A, B), eng.gemm(A, B, transB=True)
            eng.set_option("rows_fused_reduce", 1)
            for _ in range(4):
                assert torch.equal(eng.gemm(A, B), r0), (M, klen)
                assert torch.equal(eng.gemm(A, B, transB=True), r1), (M, klen)


def test_streaming_strip_kernel_shape_sweep(opts):
    """The LDS-free strip kernel (N a multiple of 32 but not of 128, or selected by option) over the same slice rules."""
    eng = opts
    eng.set_option("rows_kernel_min_m", 1000); eng.set_option("rows_kernel_min_m_kc", 1000)
    g = torch.Generator(device="cuda").manual_seed(5)
    for N, K in ((2080, 2048), (4640, 1040), (4608, 4608), (160, 32768)):
        B = torch.randn(K, N, dtype=torch.float64, device="cuda", generator=g)
        Bt = B.t().contiguous()
        for M in (1, 16, 31, 48, 64):
            A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
            ref = _host(A, B)
            for tB, Bop in ((False, B), (True, Bt)):
                out = eng.gemm(A, Bop, transB=tB)
                assert float((out.cpu() - ref).abs().max()) <= 1e-12 * float(ref.abs().max()) * np.sqrt(K), (M, N, K, tB)
