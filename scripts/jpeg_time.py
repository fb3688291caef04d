"""Decode time of one C2 batch of JPEG frames on the GPU (64 x 224 x 224, 4:2:0), host preparation and device part apart:
    python scripts/jpeg_time.py [restart_interval_in_MCUs]"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_jpeg_cpu import make_jpeg, pil_bgr
from nafae_amd import jpeg
rst = int(sys.argv[1]) if len(sys.argv) > 1 else 0
files = [make_jpeg(224, 224, 90, 2, restart=rst, seed=i) for i in range(64)]
print("64 files, %.1f KB each, restart interval %d" % (sum(len(f) for f in files) / 64 / 1024, rst))
t0 = time.perf_counter(); P = jpeg.prepare(files); t1 = time.perf_counter()
print("host prepare (header parse + restart scan + concat): %.2f ms, %d segments" % ((t1 - t0) * 1e3, P["seg"].shape[0]))
out = jpeg.decode_batch(files); torch.cuda.synchronize()
ref = np.stack([pil_bgr(f) for f in files])
print("bit-exact vs libjpeg:", bool(np.array_equal(out.cpu().numpy(), ref)))
for _ in range(3):
    t0 = time.perf_counter(); out = jpeg.decode_batch(files); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("decode_batch end to end (host prepare + H2D + 3 kernels): %.2f ms" % ((t1 - t0) * 1e3))
t0 = time.perf_counter(); [pil_bgr(f) for f in files]; t1 = time.perf_counter()
print("PIL / libjpeg-turbo on one host core, 64 frames: %.2f ms" % ((t1 - t0) * 1e3))
