"""SVD of a symmetric matrix via its eigendecomposition (reference linalg/svd_symeig.py:12-34): forward pass on the native
symmetric eigensolver.  `SVDSYMEIG.apply(A)` returns the FULL decomposition U, S, V with S = |D| descending, V = U sign(D)."""
from backend import get_engine


class SVDSYMEIG:
    @staticmethod
    def apply(A):
        return get_engine().svd_symeig(A)

    forward = apply
