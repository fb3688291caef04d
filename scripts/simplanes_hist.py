"""Few live columns (entity lengths from the data set's histogram) through either route, stand-alone hipGraph timing:
    NAFAE_LIB=...exp.so [NAFAE_SIM_LIVE_MAX=0] python scripts/simplanes_hist.py [c2|c4|c5] [none|bf16x3|f16 ...]
NAFAE_SIM_LIVE_MAX=0 (experiments build) sends every shape to the many-live-column kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nafae_amd import ops, synthetic as syn
W = {"c2": (8, 8, 128, 16), "c4": (8, 8, 256, 32), "c5": (8, 8, 300, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
kinds = sys.argv[2:] or ["none", "f16"]
Na, Ns, Nb, Ne = W[name]
lens = syn.entity_lengths(Na, Ne, seed=1234)
V, Wt = syn.embeddings(Na * Ns * Nb, Na * Ne, 512, seed=1)
V, Wt = V.cuda(), Wt.cuda()
lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
iters, ref = 20, None
for kind in kinds:
    pl = False if kind == "none" else (ops.sim_planes(V, kind), ops.sim_planes(Wt, kind))
    fn = lambda: ops.sim_max_fwd(V, Wt, lt, Na, Ns, Nb, Ne, lens=lens, planes=pl)
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    ref = ref or out
    same = bool(torch.equal(out[1], ref[1]) and torch.allclose(out[0], ref[0], rtol=0, atol=3e-6 * float(ref[0].abs().max())))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
        g.replay(); st.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); g.replay(); e1.record(st); st.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters)
    by = 4.0 * 512 * (V.shape[0] + Wt.shape[0]) + 12.0 * Na * Ns * Na * Ne
    print("%s hist (%d live) LIVE_MAX=%s planes=%-7s %.2f us per call (%.3f of 8 TB/s)  same result: %s"
          % (name, sum(lens), os.environ.get("NAFAE_SIM_LIVE_MAX", "default"), kind, best * 1e3, by / (best * 1e-3) / 8e12, same))
