"""Stand-alone timing (hipGraph of back-to-back calls) + in-kernel phase stamps of the many-live-column similarity kernels:
    [NAFAE_LIB=nafae_amd/csrc/libnafae_hip_exp.so] python scripts/simplanes_time.py [c2|c4|c5] [none|bf16x3|f16 ...]
planes 'none' = fp32 operands only: the entry point's own pre-pass + the planes kernel; otherwise the planes are produced OUTSIDE
the timed region (in a step they come out of the embedding epilogue)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nafae_amd import _lib, ops, synthetic as syn
W = {"c2": (8, 8, 128, 16), "c4": (8, 8, 256, 32), "c5": (8, 8, 300, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
kinds = sys.argv[2:] or ["none", "bf16x3", "f16"]
Na, Ns, Nb, Ne = W[name]
lens = syn.entity_lengths(Na, Ne, seed=1234) if os.environ.get("SIM_LENS") == "hist" else [Ne] * Na   # SIM_LENS=hist: few live columns
V, Wt = syn.embeddings(Na * Ns * Nb, Na * Ne, 512, seed=1)
V, Wt = V.cuda(), Wt.cuda()
lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
exp = "exp" in os.environ.get("NAFAE_LIB", "")
iters = 20
ref = None
for kind in kinds:
    pl = False if kind == "none" else (ops.sim_planes(V, kind), ops.sim_planes(Wt, kind))
    fn = lambda: ops.sim_max_fwd(V, Wt, lt, Na, Ns, Nb, Ne, lens=lens, planes=pl)
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    if ref is None:
        ref = out
    same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
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
    print("%s " % name + ("hist" if os.environ.get("SIM_LENS") == "hist" else "all-live") + " planes=%-7s %.2f us per call  (%.3f of 8 TB/s on the algorithmic bytes)  equal to first: %s"
          % (kind, best * 1e3, by / (best * 1e-3) / 8e12, same))
    if exp and kind != "none":
        fn(); torch.cuda.synchronize()
        N = 8 * 8192
        buf = (ctypes.c_ulonglong * N)()
        L = _lib.lib()
        L.nafae_simplanes_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.nafae_simplanes_debug_stamps.restype = ctypes.c_int
        assert L.nafae_simplanes_debug_stamps(buf, N) == 0
        s = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
        wave = np.arange(8192) % 8
        ok = s[:, 0] > 0
        t0 = s[ok, 0].min()
        names = ["start (prologue done)", "chunk 0 landed", "k-loop done", "scan + lists", "work list built", "exact done", "end"]
        for role, sel in (("MFMA waves", ok & (wave < 4)), ("staging waves", ok & (wave >= 4))):
            for k in range(7):
                col = s[sel, k]
                col = col[col > 0]
                if len(col):
                    print("   %-14s %-22s median %+7.2f us  min %+7.2f  max %+7.2f" % (role, names[k], np.median(col - t0) / 100.0,
                                                                                  (col.min() - t0) / 100.0, (col.max() - t0) / 100.0))
