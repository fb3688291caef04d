"""JPEG frame decoding, CPU side (no GPU): (1) the numpy restatement of libjpeg's default decompression (oracle/jpeg.py) and
(2) the DEVICE arithmetic of csrc/jpeg.hip -- jpeg_core.h, the very functions the kernels call per lane, compiled for the host by
g++ (tests/jpeg_host_harness.cpp) and fed by the product's host parser (nafae_amd/jpeg.py prepare) -- both against PIL, which links
the library cv2.imread uses (libjpeg-turbo) with the same defaults (ISLOW IDCT, fancy upsampling).  The reference call being
replaced: `cv2.imread(img_path)`, lib/datasets/youcook2.py:212.  Bit-exact on every pixel."""
import ctypes
import io
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Image = pytest.importorskip("PIL.Image")


def make_jpeg(h, w, quality, subsampling, restart=0, grey=False, smooth=True, seed=0, progressive=False):
    rs = np.random.RandomState(seed)
    if smooth:
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([128 + 100 * np.sin(xx / 9.0 + yy / 17.0 + seed), 128 + 90 * np.cos(xx / 5.0), 128 + 80 * np.sin(yy / 7.0)], -1)
        a = a + rs.randn(h, w, 3) * 12
    else:
        a = rs.rand(h, w, 3) * 255
    a = np.clip(a, 0, 255).astype(np.uint8)
    im = Image.fromarray(a[..., 0] if grey else a)
    kw = dict(quality=quality, progressive=progressive)
    if not grey:
        kw["subsampling"] = subsampling
    if restart:
        kw["restart_marker_blocks"] = restart
    b = io.BytesIO()
    im.save(b, "JPEG", **kw)
    return b.getvalue()


def pil_bgr(data):
    """what cv2.imread returns for the file (IMREAD_COLOR): libjpeg's RGB output in BGR order; grey files replicated"""
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., ::-1]


CASES = [  # h, w, quality, subsampling (0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0), restart interval in MCUs, grey, smooth content
    (16, 16, 90, 0, 0, False, True), (48, 64, 90, 2, 0, False, True), (33, 47, 75, 2, 0, False, True), (40, 56, 95, 1, 0, False, True),
    (31, 29, 50, 1, 0, False, False), (64, 64, 85, 2, 3, False, True), (17, 23, 90, 0, 0, True, True), (9, 5, 90, 2, 0, False, False),
    (8, 3, 90, 1, 0, False, False), (112, 96, 30, 2, 0, False, False), (64, 80, 100, 2, 5, False, False), (1, 1, 90, 2, 0, False, False),
    (224, 224, 92, 2, 0, False, True),       # the frame size of the reference's default configuration (--img_h / --img_w 224)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_q%d_s%d_r%d%s" % (c[0], c[1], c[2], c[3], c[4], "_grey" if c[5] else ""))
def test_oracle_jpeg_matches_libjpeg(case):
    from oracle import jpeg as OJ
    data = make_jpeg(*case)
    assert np.array_equal(OJ.decode(data), pil_bgr(data))


@pytest.fixture(scope="module")
def host_harness():
    out = os.path.join(ROOT, "oracle", "_build", "libjpeghost.so")
    src = os.path.join(ROOT, "tests", "jpeg_host_harness.cpp")
    dep = os.path.join(ROOT, "nafae_amd", "csrc", "jpeg_core.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(dep)):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", out, src])
    return ctypes.CDLL(out)


def host_decode(L, files, lanes=0):
    """lanes = 0: the one-lane entropy decoder; lanes > 0: the many-lane (self-synchronising) algorithm of jpeg_huffman_par_kernel,
    its lanes emulated one after the other.  Returns (frames, prepared tables[, most re-decoding rounds any interval needed])."""
    from nafae_amd import jpeg as NJ
    P = NJ.prepare(files)
    W, H, nc, h0, v0 = P["geom"]
    out = np.zeros((len(files), H, W, 3), np.uint8)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rounds = ctypes.c_int(0)
    rc = L.jpeg_host_decode_lanes(ptr(P["stream"]), ptr(P["desc"]), ptr(P["seg"]), ptr(P["qtabs"]), ptr(P["hufftabs"]), len(files),
                                  P["seg"].shape[0], W, H, nc, h0, v0, ptr(out), int(lanes), ctypes.byref(rounds))
    assert rc == 0
    return (out, P, rounds.value) if lanes else (out, P)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_q%d_s%d_r%d%s" % (c[0], c[1], c[2], c[3], c[4], "_grey" if c[5] else ""))
def test_device_arithmetic_on_host_matches_libjpeg(case, host_harness):
    """jpeg_core.h (what the kernels execute) + nafae_amd/jpeg.py's tables, a batch of three files with different content."""
    files = [make_jpeg(*case, seed=s) for s in range(3)]
    got, P = host_decode(host_harness, files)
    assert np.array_equal(got, np.stack([pil_bgr(f) for f in files]))
    if case[4]:
        assert P["seg"].shape[0] > 3             # restart intervals became independent work items
    assert len(P["hufftabs"]) <= 4 and len(P["qtabs"]) <= 2        # identical tables are shared across the batch


@pytest.mark.parametrize("lanes", [2, 7, 64, 256])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_q%d_s%d_r%d%s" % (c[0], c[1], c[2], c[3], c[4], "_grey" if c[5] else ""))
def test_many_lane_entropy_decoder_on_host_matches_libjpeg(case, lanes, host_harness):
    """The self-synchronising decoder (jpeg_core.h span_decode + the round structure of jpeg_huffman_par_kernel) with 2 ... 256 lanes
    per restart interval: wrong guesses at the lane starts must all fall into step -- bit-exact frames -- in a few rounds."""
    files = [make_jpeg(*case, seed=s) for s in range(2)]
    got, P, rounds = host_decode(host_harness, files, lanes=lanes)
    assert np.array_equal(got, np.stack([pil_bgr(f) for f in files]))
    assert 1 <= rounds <= lanes


def test_many_lane_decoder_hands_truncated_data_to_the_one_lane_decoder(host_harness):
    """Fewer MCUs in the data than the frame has (a truncated file): the many-lane pass notices and the one-lane decoder, which feeds
    zero bits as libjpeg does, takes the interval -- same frames as lanes = 0."""
    f = make_jpeg(64, 48, 90, 2, seed=5)
    from nafae_amd import jpeg as NJ
    hdr = NJ.parse_header(f)
    cut = f[:hdr["scan"] + (len(f) - hdr["scan"]) // 2] + b"\xff\xd9"
    a, _ = host_decode(host_harness, [cut])
    b, _, _ = host_decode(host_harness, [cut], lanes=16)
    assert np.array_equal(a, b)


def test_unsupported_files_are_refused_loudly():
    from nafae_amd import jpeg as NJ
    with pytest.raises(NJ.JpegUnsupported):
        NJ.parse_header(make_jpeg(32, 32, 90, 2, progressive=True))
    with pytest.raises(NJ.JpegUnsupported):
        NJ.parse_header(b"\\x89PNG\\r\\n\\x1a\\n" + b"\\0" * 32)
    with pytest.raises(Exception):
        NJ.prepare([make_jpeg(32, 32, 90, 2), make_jpeg(32, 48, 90, 2)])        # one call = one geometry


def test_malformed_files_raise_jpeg_unsupported():
    """the host parser's contract is `JpegUnsupported` for every file the GPU decoder does not take -- truncated files,
    empty segments, missing tables, zero sizes, a rotating EXIF orientation -- never IndexError / KeyError / ValueError (an untrusted
    data-set file must not crash load_segment_gpu)."""
    from nafae_amd import jpeg as J
    good = make_jpeg(32, 40, 90, 2)
    assert J.parse_header(good)["W"] == 40
    for cut in (1, 3, 5, 20, 60, 150, len(good) // 2):                    # truncated anywhere in the headers
        try:
            info = J.parse_header(good[:cut])
        except J.JpegUnsupported:
            continue
        assert info["scan"] <= cut                                         # (a cut behind SOS leaves a complete header)
    # a DQT segment removed: the component refers to a missing table
    i = good.index(b"\xff\xdb")
    L = (good[i + 2] << 8) | good[i + 3]
    with pytest.raises(J.JpegUnsupported):
        J.prepare([good[:i] + good[i + 2 + L:]])
    # every DHT segment removed
    b = good
    while b"\xff\xc4" in b[:b.index(b"\xff\xda")]:
        i = b.index(b"\xff\xc4")
        L = (b[i + 2] << 8) | b[i + 3]
        b = b[:i] + b[i + 2 + L:]
    with pytest.raises(J.JpegUnsupported):
        J.prepare([b])
    # zero height in SOF0
    i = good.index(b"\xff\xc0")
    with pytest.raises(J.JpegUnsupported):
        J.parse_header(good[:i + 5] + b"\x00\x00" + good[i + 7:])
    # an empty DRI segment (length 2)
    i = good.index(b"\xff\xda")
    with pytest.raises(J.JpegUnsupported):
        J.parse_header(good[:i] + b"\xff\xdd\x00\x02" + good[i:])
    # EXIF orientation 6 (cv2.imread would rotate): refused; orientation 1: accepted
    def exif(o):
        tiff = b"II*\x00\x08\x00\x00\x00" + b"\x01\x00" + b"\x12\x01\x03\x00\x01\x00\x00\x00" + bytes([o, 0, 0, 0]) + b"\x00\x00\x00\x00"
        seg = b"Exif\x00\x00" + tiff
        return good[:2] + b"\xff\xe1" + bytes([(len(seg) + 2) >> 8, (len(seg) + 2) & 255]) + seg + good[2:]
    assert J.parse_header(exif(1))["W"] == 40
    with pytest.raises(J.JpegUnsupported, match="orientation 6"):
        J.parse_header(exif(6))
    assert not isinstance(J.JpegUnsupported("x"), (IndexError, KeyError))
