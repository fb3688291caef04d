"""Phase timing of the third-generation similarity kernels from in-kernel wall-clock stamps (EXPERIMENTS build only):
    NAFAE_LIB=nafae_amd/csrc/libnafae_hip_exp.so python scripts/simfused_stamps.py [c2|c4|c5] [hist|dense]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nafae_amd import _lib, ops, synthetic as syn
W = {"c2": (8, 8, 128, 16), "c4": (8, 8, 256, 32), "c5": (8, 8, 300, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
kind = sys.argv[2] if len(sys.argv) > 2 else "hist"
Na, Ns, Nb, Ne = W[name]
lens = syn.entity_lengths(Na, Ne, seed=1234) if kind == "hist" else [Ne] * Na
V, Wt = syn.embeddings(Na * Ns * Nb, Na * Ne, 512, seed=1)
V, Wt = V.cuda(), Wt.cuda()
lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
for _ in range(5):
    ops.sim_max_fwd(V, Wt, lt, Na, Ns, Nb, Ne, lens=lens)
torch.cuda.synchronize()
N = 8 * 8192
buf = (ctypes.c_ulonglong * N)()
L = _lib.lib()
L.nafae_simfused_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.nafae_simfused_debug_stamps(buf, N) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
few = sum(lens) <= 32
names = (["start", "prefix+qmap", "MFMAs done", "K quarters in LDS", "end", "-", "-"] if few else
         ["start", "chunk 0 staged", "k-loop", "scan", "records", "phase A", "end (slow list)"])
print(name, kind, "waves stamped:", len(st), "(waves 0-3 of a frame-kernel workgroup issue MFMAs, 4-7 stage)")
for k in range(7):
    col = st[:, k][st[:, k] > 0]
    if len(col):
        print("%-16s median %+7.2f us   min %+7.2f  max %+7.2f   (since the first wave's start)"
              % (names[k], np.median(col - t0) / 100.0, (col.min() - t0) / 100.0, (col.max() - t0) / 100.0))
