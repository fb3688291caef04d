"""GPU JPEG decoding (csrc/jpeg.hip through nafae_amd/jpeg.py) against libjpeg itself (PIL: the library behind cv2.imread, which
the reference's loader calls per frame, lib/datasets/youcook2.py:212): bit-exact on every pixel, batches of frames, all supported
samplings, restart intervals, odd sizes; and the frames go straight into the detector's first layer."""
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
Image = pytest.importorskip("PIL.Image")
from test_jpeg_cpu import CASES, host_harness, make_jpeg, pil_bgr      # noqa: E402,F401  (same generated files as the CPU-side checks)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_q%d_s%d_r%d%s" % (c[0], c[1], c[2], c[3], c[4], "_grey" if c[5] else ""))
def test_gpu_decode_equals_libjpeg(case):
    from nafae_amd import jpeg
    files = [make_jpeg(*case, seed=s) for s in range(5)]
    got = jpeg.decode_batch(files).cpu().numpy()
    ref = np.stack([pil_bgr(f) for f in files])
    assert got.shape == ref.shape and got.dtype == np.uint8
    assert np.array_equal(got, ref), "max |diff| %d at %d pixels" % (np.abs(got.astype(int) - ref.astype(int)).max(), (got != ref).sum())


def test_gpu_decode_a_segment_batch_and_feed_the_detector(tmp_path):
    """64 frames 224 x 224 (one C2 batch), 4:2:0, written to disk like genframes.py's output, read back by loader.load_segment_gpu:
    equal to the host decode, and the first conv layer on the decoded bytes equals the fp32 path of the reference's loader
    (youcook2.py:212-214: imread, astype(float32), -= 127.5)."""
    from types import SimpleNamespace
    from nafae_amd import jpeg, loader, ops
    paths = []
    for i in range(64):
        p = tmp_path / ("%04d%06d.jpg" % (3, i))
        p.write_bytes(make_jpeg(224, 224, 90 - (i % 3) * 10, 2, restart=(7 if i % 5 == 0 else 0), seed=100 + i))
        paths.append(str(p))
    got = jpeg.decode_files(paths)
    ref = np.stack([pil_bgr(open(p, "rb").read()) for p in paths])
    assert np.array_equal(got.cpu().numpy(), ref)
    args = SimpleNamespace(img_h=224, img_w=224, fix_seg_len=True, fix_seg_len_val=True, sample_num=8, sample_num_val=16, sample_rate=4,
                           sample_rate_val=4)
    frames, used = loader.load_segment_gpu(paths, "val", args)
    assert frames.dtype == torch.uint8 and tuple(frames.shape)[1:] == (224, 224, 3)
    idx = [paths.index(u) for u in used]
    assert np.array_equal(frames.cpu().numpy(), ref[idx])
    g = torch.Generator(device="cuda").manual_seed(0)
    w27 = torch.randn(64, 27, device="cuda", generator=g) * 0.1
    b = torch.randn(64, device="cuda", generator=g) * 0.1
    a = ops.conv1_3x3_relu(frames, w27, b)                                                     # decoded bytes, -127.5 in-kernel
    x = torch.from_numpy(ref[idx].astype(np.float32) - 127.5).permute(0, 3, 1, 2).contiguous().cuda()      # the reference's blob
    assert torch.equal(a, ops.conv1_3x3_relu(x, w27, b))


def test_gpu_decode_rejects_what_it_cannot_decode():
    from nafae_amd import jpeg
    with pytest.raises(jpeg.JpegUnsupported):
        jpeg.decode_batch([make_jpeg(32, 32, 90, 2, progressive=True)])


def test_gpu_decode_large_files_take_the_one_lane_decoder():
    """A frame whose entropy-coded data exceeds what the many-lane decoder keeps in LDS (112 KB): the kernel flags the file before it writes
    anything and the one-lane decoder (reading through its LDS ring, refilled many times) takes it -- mixed in one batch with a file
    that does fit, same size and sampling."""
    from nafae_amd import jpeg
    big = make_jpeg(1024, 768, 100, 0, seed=1)                # 4:4:4 at quality 100: several hundred KB of scan data
    small = make_jpeg(1024, 768, 15, 0, seed=2)
    assert len(big) > 200 * 1024 and len(small) < 100 * 1024
    got = jpeg.decode_batch([big, small, big]).cpu().numpy()
    ref = np.stack([pil_bgr(f) for f in (big, small, big)])
    assert np.array_equal(got, ref)


def test_gpu_decode_truncated_file_equals_the_sequential_rules(host_harness):
    """Half of the scan data gone: libjpeg (and the one-lane decoder) go on with zero bits after the marker; the many-lane decoder counts
    fewer MCUs than the frame has and hands the file over.  Checked against the same functions run on the host."""
    from nafae_amd import jpeg
    from test_jpeg_cpu import host_decode
    f = make_jpeg(96, 64, 90, 2, seed=5)
    hdr = jpeg.parse_header(f)
    cut = f[:hdr["scan"] + (len(f) - hdr["scan"]) // 2] + b"\xff\xd9"
    whole = make_jpeg(96, 64, 90, 2, seed=6)
    got = jpeg.decode_batch([cut, whole]).cpu().numpy()
    want, _ = host_decode(host_harness, [cut, whole])
    assert np.array_equal(got, want)
    assert np.array_equal(got[1], pil_bgr(whole))
