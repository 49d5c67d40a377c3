"""JSON (de)serialisation of bare tensors in the reference's two formats (ipeps/tensor_io.py:30-130):
the "1D" format {dtype, dims, data[]} and the legacy sparse format whose "entries" are strings
"i0 i1 ... re [im]"."""
import numpy as np


def read_bare_json_tensor_np(json_obj):
    dtype_str = json_obj["dtype"].lower()
    assert dtype_str in ["float64", "complex128"], "Invalid dtype" + dtype_str
    raw = np.asarray(json_obj["data"], dtype=np.complex128 if "complex" in dtype_str else np.float64)
    return raw.reshape(json_obj["dims"])


def read_bare_json_tensor_np_legacy(json_obj):
    t = json_obj
    dtype_str = t["dtype"].lower() if "dtype" in t else "float64"
    assert dtype_str in ["float64", "complex128"], "Invalid dtype" + dtype_str
    dims = t["dims"] if "dims" in t else [t["physDim"]] + [t["auxDim"]] * 4
    X = np.zeros(dims, dtype=dtype_str)
    for entry in t["entries"]:
        tok = entry.split()
        if dtype_str == "complex128":
            X[tuple(int(i) for i in tok[:-2])] = float(tok[-2]) + 1.0j * float(tok[-1])
        else:
            k = 1 if len(tok) == len(dims) + 1 else 2        # real tensor stored with a zero imaginary column
            X[tuple(int(i) for i in tok[:-k])] += float(tok[-k])
    return X


def serialize_bare_tensor_np(t):
    a = t.detach().cpu().numpy()
    return {"format": "1D", "dtype": "complex128" if np.iscomplexobj(a) else "float64", "dims": list(a.shape),
            "data": [repr(x) if not np.iscomplexobj(a) else str(x) for x in a.reshape(-1).tolist()]}


def serialize_bare_tensor_legacy(t, tol=1.0e-14):
    a = t.detach().cpu().numpy()
    cplx = np.iscomplexobj(a)
    entries = []
    for idx in np.argwhere(np.abs(a) > tol):
        v = a[tuple(idx)]
        s = " ".join(str(int(i)) for i in idx)
        entries.append(f"{s} {float(v.real)!r} {float(v.imag)!r}" if cplx else f"{s} {float(v)!r}")
    return {"dtype": "complex128" if cplx else "float64", "dims": list(a.shape), "numEntries": len(entries), "entries": entries}
