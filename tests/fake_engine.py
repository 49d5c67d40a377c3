"""Oracle-backed stand-in for the native engine (TEST INFRASTRUCTURE ONLY).

Lets the host logic (move orchestration, sharding over ranks, convergence checks) run on CPU torch
tensors without a GPU: every compute method defers to the numpy oracle.  Installed with
`backend.set_engine(FakeEngine())` by tests; the product path never sees it."""
import numpy as np
import torch
from oracle import ctm_oracle as O


def _n(t):
    return t.detach().cpu().numpy()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class _Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class FakeEngine:
    device = torch.device("cpu")

    def cfg(self, svd_reltol=1e-8, eps_multiplet=1e-8, multiplet_abstol=1e-14, keep_multiplets=True, fix_signs=True):
        return _Cfg(svd_reltol=svd_reltol, eps_multiplet=eps_multiplet, multiplet_abstol=multiplet_abstol,
                    keep_multiplets=keep_multiplets)

    def sync(self):
        pass

    def gemm(self, A, B, transA=False, transB=False, alpha=1.0):
        a, b = _n(A), _n(B)
        return _t(alpha * ((a.T if transA else a) @ (b.T if transB else b)))

    def einsum(self, expr, *tensors, conj=()):
        ops = [(_n(t).conj() if i in conj else _n(t)) for i, t in enumerate(tensors)]
        return _t(O.seq_einsum(expr, *ops))

    def truncated_eigh(self, A, chi, cfg=None, basis=None):
        km = True if cfg is None else cfg.keep_multiplets
        D, U = O.truncated_eig_sym(_n(A), chi, abs_tol=1e-14 if cfg is None else cfg.multiplet_abstol, keep_multiplets=km,
                                   eps_multiplet=1e-12 if cfg is None else cfg.eps_multiplet)
        return _t(D), _t(U)

    def truncated_svd(self, M, chi, cfg=None, basis=None):
        km = True if cfg is None else cfg.keep_multiplets
        U, S, V = O.truncated_svd_gesdd(_n(M), chi, abs_tol=1e-14 if cfg is None else cfg.multiplet_abstol, keep_multiplets=km,
                                        eps_multiplet=1e-8 if cfg is None else cfg.eps_multiplet)
        return _t(U), _t(S), _t(V)

    def svd_backward(self, U, S, V, gU=None, gS=None, gV=None, eps=1.0e-12):
        f = lambda x: None if x is None else _n(x)
        return _t(O.svd_backward(_n(U), _n(S), _n(V), f(gU), f(gS), f(gV), eps))

    def eigh_backward(self, D, U, gD=None, gU=None, reg=1.0e-12):
        return _t(O.eigh_backward(_n(D), _n(U), None if gD is None else _n(gD), None if gU is None else _n(gU), reg))

    def permute(self, x, perm):
        return x.permute(*perm).contiguous()

    def svdvals(self, M):
        return _t(np.linalg.svd(_n(M), compute_uv=False))

    def c2x2(self, corner, C, T1, T2, a, open_=False):
        return _t(O.c2x2_sl(corner, _n(C), _n(T1), _n(T2), _n(a), open_=open_))

    def halves(self, direction, tensors16):
        d = direction if isinstance(direction, tuple) else [(0, -1), (-1, 0), (0, 1), (1, 0)][direction]
        out = []
        for h, key in enumerate(('R', 'Rt')):
            cA, _sA, cB, _sB, oA, oB = O._HALVES[d][key]
            tA = [_n(t) for t in tensors16[8 * h:8 * h + 4]]
            tB = [_n(t) for t in tensors16[8 * h + 4:8 * h + 8]]
            A, B = O.c2x2_sl(cA, *tA), O.c2x2_sl(cB, *tB)
            out.append(_t(O._op(A, oA) @ O._op(B, oB)))
        return out[0], out[1]

    def projectors(self, R, Rt, chi, cfg=None, return_S=False):
        cfg = cfg or self.cfg()
        P, Pt, S = O.projectors_from_matrices(_n(R), _n(Rt), chi, svd_reltol=cfg.svd_reltol, eps_multiplet=cfg.eps_multiplet,
                                              multiplet_abstol=cfg.multiplet_abstol, return_S=True)
        return (_t(P), _t(Pt), _t(S)) if return_S else (_t(P), _t(Pt))

    def absorb(self, direction, tensors10, normalize=True):
        d = direction if isinstance(direction, tuple) else [(0, -1), (-1, 0), (0, 1), (1, 0)][direction]
        sp = O._ABSORB[d]
        C1, T1, T, T2, C2, A, P2, Pt2, P1, Pt1 = (_n(t) for t in tensors10)
        chi = C1.shape[0]
        as3 = lambda P: P.reshape(chi, P.shape[0] // chi, P.shape[1])
        nC1 = O.seq_einsum(sp['nC1'], as3(Pt1), C1, T1)
        nC2 = O.seq_einsum(sp['nC2'], C2, T2, as3(P2))
        Tv = O._split(T, sp['tsplit'][0], A.shape[sp['tsplit'][1]])
        nT = O.seq_einsum(sp['nT'], Tv, O._split(as3(Pt2), 1, A.shape[sp['pt2']]), A, A.conj(), O._split(as3(P1), 1, A.shape[sp['p1']]))
        f0, f1 = sp['fuse']
        sh = list(nT.shape)
        nT = nT.reshape(sh[:f0] + [sh[f0] * sh[f1]] + sh[f1 + 1:])
        if normalize == 2:
            nC1, nC2, nT = (x / np.linalg.norm(x.ravel()) for x in (nC1, nC2, nT))
        elif normalize:
            nC1, nC2, nT = O._nrm(nC1), O._nrm(nC2), O._nrm(nT)
        return _t(nC1), _t(nC2), _t(nT)

    def init_piece(self, kind, a):
        A = _n(a)
        expr = ['mijef,mijab->eafb', 'miefj,miabj->eafb', 'mefij,mabij->eafb', 'meijf,maijb->eafb',
                'miefg,miabc->eafbgc', 'meifg,maibc->eafbgc', 'mefig,mabic->eafbgc', 'mefgi,mabci->eafbgc'][kind]
        r = np.einsum(expr, A, A.conj())
        sh = r.shape
        r = r.reshape([sh[2 * i] * sh[2 * i + 1] for i in range(len(sh) // 2)])
        return _t(r / np.abs(r).max())

    def move_c4v(self, a, C, T, cfg=None, normalize=1, basis=None):
        from oracle import c4v_oracle as O4
        nC, nT, Dv, _P = O4.ctm_move_sl(_n(a), _n(C), _n(T), return_P=True, norm_type='inf' if normalize == 1 else '2')
        return _t(np.ascontiguousarray(nC)), _t(np.ascontiguousarray(nT)), _t(np.ascontiguousarray(Dv))

    def rdm_c4v(self, which, a, C, T):
        """C4v RDMs (native ctm_rdm_c4v returns them raw; the host's hermitisation / trace normalisation leaves the oracle's
        already normalised ones unchanged)."""
        from oracle import c4v_oracle as O4
        f = [O4.rdm2x1_sl, O4.rdm2x2_NN_lowmem_sl, O4.rdm2x2_NNN_lowmem_sl, O4.rdm2x2][which]
        return _t(np.ascontiguousarray(f(_n(a), _n(C), _n(T))))

    def rdm2x2(self, tensors16):
        cs = []
        for i, cid in enumerate((O.LU, O.RU, O.RD, O.LD)):
            cs.append(O.c2x2_sl(cid, *[_n(t) for t in tensors16[4 * i:4 * i + 4]], open_=True))
        up = np.einsum('akst,kbuv->abstuv', cs[0], cs[1], optimize=True)
        lo = np.einsum('akst,bkuv->abstuv', cs[3], cs[2], optimize=True)
        r = np.einsum('abstuv,abwxyz->stuvwxyz', up, lo, optimize=True)
        return _t(r.transpose(0, 2, 4, 6, 1, 3, 5, 7))

    def rdm2x2_part(self, tensors16, lo0, lo1):
        """R[(s0 t0 s1 t1), lo0:lo1] of the same contraction (the native ctm_rdm2x2_part)."""
        cs = []
        for i, cid in enumerate((O.LU, O.RU, O.RD, O.LD)):
            cs.append(O.c2x2_sl(cid, *[_n(t) for t in tensors16[4 * i:4 * i + 4]], open_=True))
        up = np.einsum('akst,kbuv->abstuv', cs[0], cs[1], optimize=True)
        lo = np.einsum('akst,bkuv->abstuv', cs[3], cs[2], optimize=True)
        p = up.shape[2]
        lo = lo.reshape(lo.shape[0], lo.shape[1], p ** 4)[:, :, lo0:lo1]
        return _t(np.einsum('abstuv,abl->stuvl', up, lo, optimize=True).reshape(p ** 4, lo1 - lo0))

    @staticmethod
    def rdm2x2_from_parts(R, p):
        return R.reshape([p] * 8).permute(0, 2, 4, 6, 1, 3, 5, 7).contiguous()

