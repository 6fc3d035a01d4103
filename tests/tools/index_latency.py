"""One query image against a resident database: eval.retrieve_top_k_similar with the plain dict (the reference's calling convention:
the matrix is rebuilt and uploaded on every call) and with a pvsim.index.DeviceIndex (resident, normalised once).
    python tests/tools/index_latency.py [N] [L]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "python-visual-similarity_amd"))
import pvsim                                  # noqa: E402
from pvsim import eval as pe                  # noqa: E402
from pvsim.index import DeviceIndex           # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8189
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
ctx = pvsim.Context(0)
rng = np.random.default_rng(0)
db = rng.standard_normal((N, L), dtype=np.float32)
paths = [f"img_{i:05d}.jpg" for i in range(N)]
enc_map = dict(zip(paths, db))


class Enc:
    context = ctx

    def encode(self, image):
        return db[int(image[0, 0])].reshape(1, -1) + np.float32(0.01)


def timed(fn, reps):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return np.median(t) * 1e3, min(t) * 1e3


q = np.array([[123]], dtype=np.int64)
a = pe.retrieve_top_k_similar(q, enc_map, Enc(), k=5)
t0 = time.perf_counter()
index = DeviceIndex(enc_map, ctx)
t_build = (time.perf_counter() - t0) * 1e3
b = pe.retrieve_top_k_similar(q, index, Enc(), k=5)
assert a == b
d_med, d_min = timed(lambda: pe.retrieve_top_k_similar(q, enc_map, Enc(), k=5), 5)
i_med, i_min = timed(lambda: pe.retrieve_top_k_similar(q, index, Enc(), k=5), 50)
print(f"N = {N}, L = {L} float32 ({db.nbytes / 1e9:.2f} GB): retrieve_top_k_similar, one query, k = 5")
print(f"  plain dict    median {d_med:9.2f} ms  min {d_min:9.2f} ms   (matrix rebuilt from the dict + uploaded every call)")
print(f"  DeviceIndex   median {i_med:9.2f} ms  min {i_min:9.2f} ms   (built once in {t_build:.0f} ms); same list, same scores")
