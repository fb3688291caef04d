"""Where a trip of the frame kernel's k-loop goes (EXPERIMENTS build, NAFAE_SIM_DBG=256 [+ other bits]):
    NAFAE_LIB=nafae_amd/csrc/libnafae_hip_exp.so NAFAE_SIM_DBG=256 python scripts/simfused_trip.py [c5|c4] dense"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nafae_amd import _lib, ops, synthetic as syn
W = {"c2": (8, 8, 128, 16), "c4": (8, 8, 256, 32), "c5": (8, 8, 300, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
Na, Ns, Nb, Ne = W[name]
lens = [Ne] * Na
V, Wt = syn.embeddings(Na * Ns * Nb, Na * Ne, 512, seed=1)
V, Wt = V.cuda(), Wt.cuda()
lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
for _ in range(3):
    ops.sim_max_fwd(V, Wt, lt, Na, Ns, Nb, Ne, lens=lens)
torch.cuda.synchronize()
N = 8 * 8192
buf = (ctypes.c_ulonglong * N)()
L = _lib.lib()
L.nafae_simfused_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.nafae_simfused_debug_stamps(buf, N) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8, 8).astype(np.int64)     # [wg][wave][slot]
st = st[st[:, 0, 0] > 0]
kl = (st[:, :, 2] - st[:, :, 1]) / 100.0
print(name, "dense, dbg", os.environ.get("NAFAE_SIM_DBG"), "workgroups", len(st), "k-loop us (wall clock): MFMA waves %.2f staging waves %.2f" % (np.median(kl[:, :4]), np.median(kl[:, 4:])))
m, g = st[:, :4, :], st[:, 4:, :]
tot = np.median(m[:, :, 3] + m[:, :, 4])
print("MFMA waves   : issue reads+MFMAs %8.0f cycles, barrier wait %8.0f   (per 16 trips; sum %.0f)" % (np.median(m[:, :, 3]), np.median(m[:, :, 4]), tot))
tot = np.median(g[:, :, 3] + g[:, :, 4] + g[:, :, 5])
print("staging waves: wait for loads    %8.0f cycles, convert+write+issue %8.0f, barrier wait %8.0f   (sum %.0f)" % (np.median(g[:, :, 3]), np.median(g[:, :, 4]), np.median(g[:, :, 5]), tot))
