"""Driver of the C++ CPU restatement of one CTM unit (oracle/cpu_unit.cpp): builds the binary against the threaded OpenBLAS/LAPACK
that ships with scipy, generates the unit's contraction program from the oracle's own tables and runs it.

TEST / MEASUREMENT INFRASTRUCTURE (see oracle/__init__.py): used by bench.py's `cpu_baseline` leg and by tests/test_cpu_unit.py
(which pins the C++ restatement against the numpy oracle on a golden state).  float64 only (the BASELINE configurations that are
runnable on a CPU are real)."""
import glob, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "bin", "cpu_unit")


def _openblas():
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if not libs:
        raise RuntimeError("scipy's bundled OpenBLAS not found")
    return sorted(libs, key=len)[0]


def build(force=False):
    """g++ -O2 -fopenmp oracle/cpu_unit.cpp against scipy's libscipy_openblas (dgemm, dgesdd; threaded).  Output: oracle/bin/cpu_unit
    (git-ignored; travels to the GPU box with the snapshot like the HIP library)."""
    src = os.path.join(HERE, "cpu_unit.cpp")
    if not force and os.path.exists(BIN) and os.path.getmtime(BIN) >= os.path.getmtime(src):
        return BIN
    lib = _openblas()
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    cmd = ["g++", "-O2", "-fopenmp", "-std=c++17", src, "-o", BIN, "-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib),
           "-Wl,-rpath," + os.path.dirname(lib)]
    subprocess.check_call(cmd)
    return BIN


class _Prog:
    def __init__(self):
        self.lines, self.arrays, self.n = [], [], 0

    def load(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        self.arrays.append(a)
        self.lines.append(f"load {name} " + " ".join(str(d) for d in a.shape))
        return name

    def tmp(self):
        self.n += 1
        return f"t{self.n}"

    def seq_einsum(self, expr, names):
        """The pairwise left-to-right evaluation of oracle.ctm_oracle.seq_einsum, as `ein` lines."""
        lhs, out = expr.split('->')
        ins = lhs.split(',')
        cur, cidx = names[0], ins[0]
        for k in range(1, len(ins)):
            later = set(out).union(*[set(x) for x in ins[k + 1:]]) if k + 1 < len(ins) else set(out)
            nidx = ''.join(dict.fromkeys([c for c in cidx + ins[k] if c in later]))
            nm = self.tmp()
            self.lines.append(f"ein {nm} {cidx},{ins[k]}->{nidx} {cur} {names[k]}")
            if cur.startswith("t"):
                self.lines.append(f"free {cur}")
            cur, cidx = nm, nidx
        if cidx != out:
            nm = self.tmp()
            self.lines.append(f"perm {nm} {cidx}->{out} {cur}")
            self.lines.append(f"free {cur}")
            cur = nm
        return cur


def unit_program(direction, coord, ost, oenv, svd_nsub=0, reltol=1e-8):
    """Program + input arrays of ONE unit: 4 enlarged corners, R, Rt, M = R^T Rt, full dgesdd, P, Pt, absorb of `coord`."""
    from . import ctm_oracle as O
    chi = oenv.chi
    pr = _Prog()
    pr.lines.append("tic corners")
    halves = {}
    for key in ('R', 'Rt'):
        cA, sA, cB, sB, oA, oB = O._HALVES[direction][key]
        mats = []
        for cid, sh in ((cA, sA), (cB, sB)):
            c = (coord[0] + sh[0], coord[1] + sh[1])
            C, T1, T2, a = O.c2x2_tensors(cid, c, ost, oenv)
            sp = O._CORNER[cid]
            tag = f"{key}{len(mats)}"
            nC, nT1, nT2, nA = (pr.load(f"{tag}_{x}", v) for x, v in (("C", C), ("T1", T1), ("T2", T2), ("a", a)))
            T1v = O._split(T1, sp['s1'][0], a.shape[sp['s1'][1]]).shape
            T2v = O._split(T2, sp['s2'][0], a.shape[sp['s2'][1]]).shape
            pr.lines.append(f"view {nT1}v {nT1} " + " ".join(map(str, T1v)))
            pr.lines.append(f"view {nT2}v {nT2} " + " ".join(map(str, T2v)))
            r = pr.seq_einsum(sp['closed'], [nC, nT1 + "v", nT2 + "v", nA, nA])          # conj(a) = a (float64)
            n0 = chi * a.shape[_leg0(cid)] ** 2; n1 = chi * a.shape[_leg1(cid)] ** 2
            pr.lines.append(f"view {tag} {r} {n0} {n1}")
            mats.append(tag)
        halves[key] = (mats, oA, oB)
    pr.lines.append("toc corners")
    pr.lines.append("tic halves")
    for key in ('R', 'Rt'):
        (mA, mB), oA, oB = halves[key]
        pr.lines.append(f"mm {key} {mA} {oA} {mB} {oB}")
        pr.lines.append(f"free {mA} {mB}")
    pr.lines.append("mm M R T Rt N")
    pr.lines.append("toc halves")
    pr.lines.append("tic svd")
    pr.lines.append(f"svd U S V M {chi} {int(svd_nsub)}")
    pr.lines.append("toc svd")
    pr.lines.append("free M")
    pr.lines.append("tic proj")
    pr.lines.append(f"proj P R U S {reltol}")
    pr.lines.append(f"proj Pt Rt V S {reltol}")
    pr.lines.append("toc proj")
    pr.lines.append("free R Rt U V")
    # absorb of this site with the same projector pair on both sides (timing: the neighbour's projectors have the same shape)
    sp = O._ABSORB[direction]
    c = ost.vertexToSite(coord)
    C1, T1, T, T2, C2 = (oenv.C[(c, sp['C1'])], oenv.T[(c, sp['T1'])], oenv.T[(c, sp['T'])], oenv.T[(c, sp['T2'])], oenv.C[(c, sp['C2'])])
    A = ost.site(coord)
    n = chi * A.shape[1] ** 2
    names = {k: pr.load("ab_" + k, v) for k, v in (("C1", C1), ("T1", T1), ("T", T), ("T2", T2), ("C2", C2), ("A", A))}
    pr.lines.append("tic absorb")
    D2 = n // chi
    pr.lines.append(f"view P3 P {chi} {D2} {chi}")
    pr.lines.append(f"view Pt3 Pt {chi} {D2} {chi}")
    nC1 = pr.seq_einsum(sp['nC1'], ["Pt3", names["C1"], names["T1"]])
    nC2 = pr.seq_einsum(sp['nC2'], [names["C2"], names["T2"], "P3"])
    Tv = O._split(T, sp['tsplit'][0], A.shape[sp['tsplit'][1]]).shape
    Dp2, Dp1 = A.shape[sp['pt2']], A.shape[sp['p1']]
    pr.lines.append("view ab_Tv ab_T " + " ".join(map(str, Tv)))
    pr.lines.append(f"view Pt2v Pt {chi} {Dp2} {Dp2} {chi}")
    pr.lines.append(f"view P1v P {chi} {Dp1} {Dp1} {chi}")
    nT = pr.seq_einsum(sp['nT'], ["ab_Tv", "Pt2v", names["A"], names["A"], "P1v"])
    pr.lines.append("toc absorb")
    pr.finals = {"nC1": nC1, "nC2": nC2, "nT": nT}
    return pr


def _leg0(cid):
    return (3, 2, 1, 1)[cid]


def _leg1(cid):
    return (4, 3, 2, 4)[cid]


def run_unit(direction, coord, ost, oenv, svd_nsub=0, threads=None, dump=("S",), keep=False):
    """Execute the unit on the CPU; returns {"times": {phase: s}, "threads": n, "dumps": {name: (fro, max, first values)}}."""
    exe = build()
    pr = unit_program(direction, coord, ost, oenv, svd_nsub=svd_nsub)
    lines, finals = list(pr.lines), pr.finals
    for lab, nm in finals.items():
        if lab in dump:
            lines.append(f"dump {nm}")
    for nm in dump:
        if nm not in finals:
            lines.append(f"dump {nm}")
    d = tempfile.mkdtemp(prefix="cpu_unit_")
    fp, fb = os.path.join(d, "prog.txt"), os.path.join(d, "tensors.bin")
    with open(fp, "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(fb, "wb") as f:
        for a in pr.arrays:
            f.write(a.tobytes())
    cmd = [exe, fp, fb] + ([str(threads)] if threads else [])
    out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    if not keep:
        for x in (fp, fb): os.remove(x)
        os.rmdir(d)
    res = {"times": {}, "dumps": {}, "threads": None}
    rev = {v: k for k, v in finals.items()}
    for l in out.splitlines():
        w = l.split()
        if w[0] == "time": res["times"][w[1]] = float(w[2])
        elif w[0] == "threads": res["threads"] = int(w[1])
        elif w[0] == "dump":
            nm = rev.get(w[1], w[1])
            res["dumps"][nm] = (float(w[2].split("=")[1]), float(w[3].split("=")[1]), [float(x) for x in l.split("first=")[1].split()])
    return res
