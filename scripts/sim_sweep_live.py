"""Similarity forward time against the number of LIVE query slots at a BASELINE shape (hipGraph of 20 calls, best of 3):
    python scripts/sim_sweep_live.py [c2|c4|c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nafae_amd import ops, synthetic as syn
W = {"c2": (8, 8, 128, 16), "c4": (8, 8, 256, 32), "c5": (8, 8, 300, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
Na, Ns, Nb, Ne = W[name]
V, Wt = syn.embeddings(Na * Ns * Nb, Na * Ne, 512, seed=1)
V, Wt = V.cuda(), Wt.cuda()
st = torch.cuda.Stream()
for L in (17, 32, 33, 48, 64, 96, 128, 192, 256, Na * Ne):
    if L > Na * Ne: continue
    lens = [L // Na + (1 if a < L % Na else 0) for a in range(Na)]
    if max(lens) > Ne: continue
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(st):
        for _ in range(3): ops.sim_max_fwd(V, Wt, lt, Na, Ns, Nb, Ne, lens=lens)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(20): ops.sim_max_fwd(V, Wt, lt, Na, Ns, Nb, Ne, lens=lens)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); g.replay(); e1.record(st); st.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
    print("%s live %4d  %7.2f us" % (name, L, best * 1000))
