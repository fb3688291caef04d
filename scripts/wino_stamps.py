"""Phase clocks of the Winograd conv kernel (experiments build): where a workgroup's cycles go, per VGG layer at C2.

    python -m nafae_amd.build --experiments
    NAFAE_LIB=nafae_amd/csrc/libnafae_hip_exp.so python scripts/wino_stamps.py [layer ...]

Per layer: tiles per workgroup and the median / max over (workgroup, wave) of the cycles in the prologue, the first chunk of a
tile, the next-tile setup, the remaining chunks and the epilogue, next to the ideal MFMA cycles (64 per MFMA)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops, _lib
L = _lib.lib()
L.nafae_wino_debug_stamps.restype = None
L.nafae_wino_debug_stamps.argtypes = [ctypes.c_void_p]
g = torch.Generator(device='cuda').manual_seed(0)
F = int(os.environ.get("F", "64"))
LAYERS = {"conv1_2": (224, 64, 64, True), "conv2_1": (112, 64, 128, False), "conv2_2": (112, 128, 128, True), "conv3_1": (56, 128, 256, False),
          "conv3_2": (56, 256, 256, False), "conv4_1": (28, 256, 512, False), "conv4_2": (28, 512, 512, False), "conv5_1": (14, 512, 512, False)}
names = sys.argv[1:] or list(LAYERS)
for name in names:
    H, Cin, Cout, pool = LAYERS[name]
    x = torch.relu(torch.randn(F, H, H, Cin, device='cuda', generator=g)); w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * 0.02
    b = torch.zeros(Cout, device='cuda')
    U = ops.conv3x3_wino_pack(w)
    for _ in range(3):
        ops.conv3x3_wino(x, U, b, Cout, pool=pool)
    st = torch.zeros(256 * 4 * 16, device='cuda', dtype=torch.int64)
    L.nafae_wino_debug_stamps(ctypes.c_void_p(st.data_ptr()))
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); ops.conv3x3_wino(x, U, b, Cout, pool=pool); e.record()
    torch.cuda.synchronize()
    L.nafae_wino_debug_stamps(None)
    ms = s.elapsed_time(e)
    t = st.view(256, 4, 16).cpu().double()
    t = t[t[:, 0, 0] > 0]
    tiles = t[:, :, 0]
    per = lambda i: t[:, :, i] / tiles
    ideal = (Cin // 8) * 64 * 64
    print("%s %d->%d @%d: %.3f ms with stamps | tiles/WG %d..%d | ideal MFMA cycles per tile %d (chunk %d)" %
          (name, Cin, Cout, H, ms, int(tiles.min()), int(tiles.max()), ideal, 4096))
    for i, nm in ((1, "prologue (once)"), (2, "chunk 0"), (3, "setup"), (4, "chunks 1.."), (5, "epilogue")):
        v = (t[:, :, i] if i == 1 else per(i)).flatten()
        print("   %-16s per tile: median %7.0f  max %7.0f cycles" % (nm, float(v.median()), float(v.max())))
    for i, nm in ((8, "even chunks: vmcnt wait"), (9, "even chunks: s_barrier"), (10, "odd chunks: vmcnt wait"), (11, "odd chunks: s_barrier")):
        v = per(i)
        print("   %-26s per tile: median %7.0f  max %7.0f | by wave (median) %s" % (nm, float(v.flatten().median()), float(v.max()),
              " ".join("%.0f" % float(v[:, w].median()) for w in range(4))))
    tot = (t[:, :, 6] / tiles).flatten()
    rt = t[:, :, 7]
    print("   total per tile median %.0f max %.0f | kernel span %.1f us (100 MHz clock) | MFMA share of the median tile %.3f" %
          (float(tot.median()), float(tot.max()), float(rt.max() - rt.min()) / 100.0, ideal / float(tot.median())))
    del x, w, U
