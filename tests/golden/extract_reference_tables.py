#!/usr/bin/env python3
"""Reads the parameter tables out of the reference's shipped model files WITHOUT unpickling them.

The reference ships its pretrained vocabularies as joblib pickles of scikit-learn objects
(/root/reference/pyvisim/res/model_files/*.pkl; enum members at pyvisim/encoders/_base_encoder.py:117-155).  A pickle is a
program; this reader does not run it.  It walks the opcode stream with the decoding tables of `pickletools` and keeps an INERT
picture of what the stream describes:

  * GLOBAL / STACK_GLOBAL     -> a (module, name) record.  Nothing is imported, nothing is looked up.
  * NEWOBJ / REDUCE / BUILD   -> an `Obj(cls record, args, state)` record.  No constructor, no __reduce__, no __setstate__ runs.
  * containers, strings, numbers -> plain Python values.
  * an Obj whose class record is joblib.numpy_pickle.NumpyArrayWrapper is the header joblib writes in front of an array's raw
    bytes (shape, dtype string, memory order, alignment): the bytes that follow are copied with numpy.frombuffer -- by shape
    and dtype, plain numeric dtypes only -- exactly as joblib's own reader would place them, and the walk goes on behind them.

Any opcode outside that small set, an object dtype, or a class record other than the two scikit-learn estimators / numpy
helpers expected here stops the reader with an error.  Output: plain `.npz` files (arrays only) that
`pvsim.encoders.GMMWeights` / `_PCA` load with numpy.load(allow_pickle=False).

    python tests/golden/extract_reference_tables.py            # writes python-visual-similarity_amd/pvsim/res/model_files/*.npz
    python tests/golden/extract_reference_tables.py --check    # re-reads the .pkl files and compares with the committed .npz

Runs in the build container only (it reads /root/reference); the GPU box gets the .npz files.
"""
from __future__ import annotations

import io
import os
import pickletools
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = "/root/reference/pyvisim/res/model_files"
DST = os.path.join(REPO, "python-visual-similarity_amd", "pvsim", "res", "model_files")

ALLOWED_GLOBALS = {
    ("sklearn.mixture._gaussian_mixture", "GaussianMixture"),
    ("sklearn.decomposition._pca", "PCA"),
    ("joblib.numpy_pickle", "NumpyArrayWrapper"),
    ("numpy", "ndarray"),
    ("numpy", "dtype"),
    ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "scalar"),
}
PLAIN_DTYPES = {"f4", "f8", "i4", "i8", "u1", "b1"}


class Global:
    def __init__(self, module, name):
        self.module, self.name = module, name

    def key(self):
        return (self.module, self.name)

    def __repr__(self):
        return f"<global {self.module}.{self.name}>"


class Obj:
    """What NEWOBJ / REDUCE describe: class record + constructor arguments (+ the state BUILD attaches).  Never instantiated."""

    def __init__(self, cls, args):
        self.cls, self.args, self.state = cls, args, None

    def __repr__(self):
        return f"<obj of {self.cls!r}>"


class _Mark:
    pass


def _dtype_of(rec) -> np.dtype:
    """numpy.dtype record -> np.dtype, plain little-endian numeric types only."""
    if not (isinstance(rec, Obj) and isinstance(rec.cls, Global) and rec.cls.key() == ("numpy", "dtype")):
        raise ValueError(f"not a dtype record: {rec!r}")
    code = rec.args[0]
    if code not in PLAIN_DTYPES:
        raise ValueError(f"dtype {code!r} is not a plain numeric type")
    order = rec.state[1] if isinstance(rec.state, tuple) and len(rec.state) > 1 else "<"
    if order not in ("<", "|", "="):
        raise ValueError(f"byte order {order!r} not supported")
    return np.dtype(("<" if code[0] in "fiu" and code != "u1" else "|") + code)


def read_tables(path: str):
    """-> (class name, {attribute: value}) of the estimator the file describes; arrays as np.ndarray."""
    data = open(path, "rb").read()
    f = io.BytesIO(data)
    stack, memo = [], []
    while True:
        code = f.read(1)
        if not code:
            raise ValueError("stream ended without STOP")
        op = pickletools.code2op[code.decode("latin-1")]
        arg = op.arg.reader(f) if op.arg is not None else None
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        if n == "STOP":
            break
        if n in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8", "BININT", "BININT1", "BININT2", "BINFLOAT", "LONG1",
                 "SHORT_BINBYTES", "BINBYTES", "SHORT_BINSTRING", "BINSTRING"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "MEMOIZE":
            memo.append(stack[-1])
        elif n in ("BINGET", "LONG_BINGET"):
            stack.append(memo[arg])
        elif n == "STACK_GLOBAL":
            name, module = stack.pop(), stack.pop()
            g = Global(module, name)
            if g.key() not in ALLOWED_GLOBALS:
                raise ValueError(f"{path}: unexpected class record {g!r}")
            stack.append(g)
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "MARK":
            stack.append(_Mark)
        elif n in ("TUPLE1", "TUPLE2", "TUPLE3"):
            k = int(n[-1])
            t = tuple(stack[-k:])
            del stack[-k:]
            stack.append(t)
        elif n in ("TUPLE", "SETITEMS", "APPENDS"):
            i = len(stack) - 1
            while stack[i] is not _Mark:
                i -= 1
            items = stack[i + 1:]
            del stack[i:]
            if n == "TUPLE":
                stack.append(tuple(items))
            elif n == "SETITEMS":
                stack[-1].update(zip(items[0::2], items[1::2]))
            else:
                stack[-1].extend(items)
        elif n == "SETITEM":
            v, k = stack.pop(), stack.pop()
            stack[-1][k] = v
        elif n == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif n in ("NEWOBJ", "REDUCE"):
            args, cls = stack.pop(), stack.pop()
            if not isinstance(cls, Global):
                raise ValueError(f"{n} on {cls!r}")
            stack.append(Obj(cls, args))
        elif n == "BUILD":
            state = stack.pop()
            o = stack[-1]
            if not isinstance(o, Obj):
                raise ValueError(f"BUILD on {o!r}")
            o.state = state
            if o.cls.key() == ("joblib.numpy_pickle", "NumpyArrayWrapper"):
                # joblib/numpy_pickle.py NumpyArrayWrapper.read_array: [1 byte padding length + padding] raw C-order bytes
                st = state
                dt = _dtype_of(st["dtype"])
                shape = tuple(int(x) for x in st["shape"])
                if st.get("numpy_array_alignment_bytes") is not None:
                    pad = f.read(1)[0]
                    f.read(pad)
                count = int(np.prod(shape, dtype=np.int64))
                raw = f.read(count * dt.itemsize)
                if len(raw) != count * dt.itemsize:
                    raise ValueError("array bytes cut short")
                a = np.frombuffer(raw, dtype=dt).copy()
                if st["order"] == "F":
                    a = a.reshape(shape[::-1]).transpose()
                else:
                    a = a.reshape(shape)
                stack[-1] = a
                # the array replaces its header wherever the header was memoised
                for j, m in enumerate(memo):
                    if m is o:
                        memo[j] = a
        else:
            raise ValueError(f"{path}: opcode {n} is outside the subset this reader accepts")
    top = stack[-1]
    if not (isinstance(top, Obj) and isinstance(top.state, dict)):
        raise ValueError("top-level record is not an estimator with attributes")
    attrs = {}
    for k, v in top.state.items():
        if isinstance(v, Obj) and v.cls.name == "scalar":      # numpy scalar: (dtype record, raw bytes)
            v = np.frombuffer(v.args[1], dtype=_dtype_of(v.args[0]))[0]
        attrs[k] = v
    return top.cls.name, attrs


def tables_of(path: str) -> dict:
    """The arrays pvsim needs from one model file (names as scikit-learn stores them)."""
    cls, a = read_tables(path)
    # keys `kind`, `weights` / `means` / `covariances`, `components` / `mean`: the schema of pvsim.models.save_model / load_model
    if cls == "GaussianMixture":
        if a.get("covariance_type") != "diag":
            raise ValueError("only diagonal mixtures are expected here")
        out = {"kind": np.array("gmm"), "weights": a["weights_"], "means": a["means_"], "covariances": a["covariances_"],
               "precisions_cholesky": a["precisions_cholesky_"], "converged": np.array(bool(a.get("converged_", False))),
               "n_iter": np.array(int(a.get("n_iter_", 0))), "reg_covar": np.array(float(a.get("reg_covar", 1e-6)))}
    elif cls == "PCA":
        out = {"kind": np.array("pca"), "components": np.ascontiguousarray(a["components_"]), "mean": a["mean_"],
               "explained_variance": a["explained_variance_"], "n_samples": np.array(int(a.get("n_samples_", 0))),
               "whiten": np.array(bool(a.get("whiten", False)))}
    else:
        raise ValueError(cls)
    out["sklearn_version"] = np.array(str(a.get("_sklearn_version", "")))
    return out


def main():
    check = "--check" in sys.argv
    os.makedirs(DST, exist_ok=True)
    ok = True
    for name in sorted(os.listdir(SRC)):
        if not name.endswith(".pkl"):
            continue
        t = tables_of(os.path.join(SRC, name))
        dst = os.path.join(DST, name[:-4] + ".npz")
        desc = ", ".join(f"{k}{tuple(v.shape)}:{v.dtype}" for k, v in t.items() if v.ndim)
        if check:
            old = np.load(dst, allow_pickle=False)
            same = set(old.files) == set(t) and all(np.array_equal(old[k], t[k]) for k in t)
            ok &= same
            print(("same   " if same else "DIFFERS") + f" {name}: {desc}")
        else:
            np.savez_compressed(dst, **t)
            print(f"wrote {os.path.relpath(dst, REPO)}: {desc}")
            if t["kind"] == "gmm":
                print(f"        covariance entries at the reg_covar floor (<= 1.0000001e-6): {int((t['covariances'] <= 1.0000001e-6).sum())}")
    # The reference's own KMeans codebooks (k_means_k256_*.pkl) are absent from its checkout (.MISSING_LARGE_BLOBS:3-8).  Each
    # shipped mixture was k-means-initialised (init_params='kmeans'), so its float32 means_ are the stand-in codebook SURVEY.md
    # section 0.3 / 8c names; the file says what it is (`derived_from`), and KMeansWeights.load() warns about it.
    for name in sorted(os.listdir(SRC)):
        if not (name.startswith("gmm_") and name.endswith(".pkl")):
            continue
        t = tables_of(os.path.join(SRC, name))
        kname = "k_means_" + name[len("gmm_"):-4] + ".npz"
        dst = os.path.join(DST, kname)
        arrays = {"kind": np.array("kmeans"), "cluster_centers": np.ascontiguousarray(t["means"], dtype=np.float32),
                  "derived_from": np.array(name + " means_ (the reference's own " + kname[:-4] + ".pkl is absent from its checkout)")}
        if check:
            old = np.load(dst, allow_pickle=False)
            same = all(np.array_equal(old[k], arrays[k]) for k in arrays)
            ok &= same
            print(("same   " if same else "DIFFERS") + f" {kname} (derived)")
        else:
            np.savez_compressed(dst, **arrays)
            print(f"wrote {os.path.relpath(dst, REPO)}: cluster_centers{arrays['cluster_centers'].shape} float32, derived from {name}")
    if check and not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
