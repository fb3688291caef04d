"""Phase timing of sim_part_kernel from in-kernel wall-clock stamps (EXPERIMENTS build only):
    python -m nafae_amd.build --experiments
    NAFAE_LIB=nafae_amd/csrc/libnafae_hip_exp.so python scripts/sim_stamps.py [c2|c5] [hist|dense]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nafae_amd import _lib, ops, synthetic as syn
W = {"c2": (8, 8, 128, 16), "c4": (8, 8, 256, 32), "c5": (8, 8, 300, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
kind = sys.argv[2] if len(sys.argv) > 2 else "hist"
Na, Ns, Nb, Ne = W[name]
lens = syn.entity_lengths(Na, Ne, seed=1234) if kind == "hist" else [Ne] * Na
V, Wt = syn.embeddings(Na * Ns * Nb, Na * Ne, 512, seed=1)
V, Wt = V.cuda(), Wt.cuda()
lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
for _ in range(5):
    ops.sim_max_fwd(V, Wt, lt, Na, Ns, Nb, Ne, lens=lens)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 4096))()
L = _lib.lib()
L.nafae_sim_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.nafae_sim_debug_stamps(buf, 8 * 4096) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
names = ["start", "prefix", "qmap", "W staged", "k-loop", "k-split sum", "epilogue"]
print(name, kind, "waves stamped:", len(st))
for k in range(7):
    col = st[:, k][st[:, k] > 0]
    if len(col):
        print("%-12s median %+7.2f us   min %+7.2f  max %+7.2f   (since first wave start)" % (names[k], np.median(col - t0) / 100.0, (col.min() - t0) / 100.0, (col.max() - t0) / 100.0))
d = np.diff(st[:, :7], axis=1) / 100.0
print("phase durations (median us):", dict(zip(names[1:], np.round(np.median(d, axis=0), 2))))
