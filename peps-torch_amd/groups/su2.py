"""SU(2) spin operators (reference groups/su2.py:5-175).  Basis ordered by descending S^z."""
import torch
from math import sqrt


def get_op(op, m, dtype=torch.float64, device='cpu', dbg=False):
    S = 0.5 * (m - 1)
    res = torch.zeros((m, m), dtype=dtype, device=device)
    if op == "I":
        return torch.eye(m, dtype=dtype, device=device)
    if op == "sz":
        for i in range(m):
            res[i, i] = S - i
        return res
    if op == "sp":          # S^+ |S,M> = sqrt(S(S+1) - M(M+1)) |S,M+1>
        for i in range(m - 1):
            M = -S + i
            res[i, i + 1] = sqrt(S * (S + 1) - M * (M + 1))
        return res
    if op == "sm":
        for i in range(1, m):
            M = -S + i
            res[i, i - 1] = sqrt(S * (S + 1) - M * (M - 1))
        return res
    raise Exception("Unsupported operator requested: " + op)


def get_rot_op(m, dtype=torch.float64, device='cpu'):
    res = torch.zeros((m, m), dtype=dtype, device=device)
    for i in range(m):
        res[i, m - 1 - i] = (-1) ** i
    return res


class SU2():
    def __init__(self, J, dtype=torch.float64, device='cpu'):
        self.J, self.dtype, self.device = J, dtype, device

    def _op(self, name):
        return get_op(name, self.J, dtype=self.dtype, device=self.device)

    def I(self): return self._op("I")
    def SZ(self): return self._op("sz")
    def SP(self): return self._op("sp")
    def SM(self): return self._op("sm")

    def I_N(self, N):
        return get_op("I", self.J ** N, dtype=self.dtype, device=self.device).view([self.J] * (2 * N))

    def SY(self):
        assert self.dtype in [torch.complex128], "SY requires complex dtype"
        return -0.5j * (self.SP() - self.SM())

    def BP_rot(self):
        return get_rot_op(self.J, dtype=self.dtype, device=self.device)

    def S(self):
        S = torch.zeros(3, self.J, self.J, dtype=self.dtype, device=self.device)
        S[0] = self.SZ()
        S[1] = 0.5 * (self.SP() + self.SM())
        if S.is_complex():
            S[2] = -0.5j * (self.SP() - self.SM())
        return S

    def SS(self, xyz=(1., 1., 1.)):
        k = 'ij,ab->iajb'
        return xyz[0] * torch.einsum(k, self.SZ(), self.SZ()) + 0.5 * xyz[1] * torch.einsum(k, self.SP(), self.SM()) \
            + 0.5 * xyz[2] * torch.einsum(k, self.SM(), self.SP())
