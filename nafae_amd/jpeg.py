"""GPU decoding of baseline JPEG frames: the device-side replacement of `cv2.imread(img_path)` in the reference's loader
(lib/datasets/youcook2.py:212).  The host parses the few hundred header bytes of each file (quantisation
and Huffman tables, frame geometry, restart interval) and locates the restart intervals; entropy decoding, inverse DCT, chroma
upsampling and colour conversion run in libnafae_hip.so (csrc/jpeg.hip) and reproduce libjpeg's default decompression bit for
bit, so `decode_batch(files)` equals `[cv2.imread(f) for f in files]` stacked -- uint8 [n, H, W, 3], BGR -- and feeds the first
conv layer directly (ops.conv1_3x3_relu reads uint8 HWC frames).

Supported: baseline sequential DCT (SOF0), 8-bit, Huffman, one interleaved scan, 4:4:4 / 4:2:2 / 4:2:0 / grey, restart markers.
Anything else (progressive, arithmetic coding, CMYK, 12-bit, multi-scan, an EXIF orientation other than 1 -- which cv2.imread would
apply --, truncated or corrupt headers) raises JpegUnsupported: the caller then decodes that file on the host explicitly -- nothing
here falls back silently.  All files of one call must share size and sampling (the frames of
a video do; genframes.py writes them with one ffmpeg command)."""
import ctypes

import numpy as np
import torch

from . import _lib
from .ops import NafaeOpError, _p, _rc, _stream

DESC_INTS, HT_INTS, LOOK = 32, 384, 9
_ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                    47, 55, 62, 63])


class JpegUnsupported(ValueError):
    pass


def _exif_orientation(seg):
    """Orientation tag (0x0112) of an APP1 Exif segment, or 1.  cv2.imread applies it (IMREAD_COLOR honours EXIF), this decoder does
    not rotate: anything but 1 is refused so that the caller decodes that file on the host."""
    if len(seg) < 14 or seg[:6] != b"Exif\x00\x00":
        return 1
    t = seg[6:]
    if t[:2] == b"II":
        u16 = lambda o: t[o] | (t[o + 1] << 8)
        u32 = lambda o: t[o] | (t[o + 1] << 8) | (t[o + 2] << 16) | (t[o + 3] << 24)
    elif t[:2] == b"MM":
        u16 = lambda o: (t[o] << 8) | t[o + 1]
        u32 = lambda o: (t[o] << 24) | (t[o + 1] << 16) | (t[o + 2] << 8) | t[o + 3]
    else:
        return 1
    ifd = u32(4)
    n = u16(ifd)
    for k in range(n):
        e = ifd + 2 + 12 * k
        if u16(e) == 0x0112:
            return u16(e + 8)
    return 1


def parse_header(data):
    """Marker segments up to and including SOS -> dict(W, H, comps [(id, h, v, tq, td, ta)], q {slot: uint16[64] natural order},
    dc / ac {slot: (bits[16], vals)}, restart, scan = offset of the entropy-coded data).  EVERY malformed input -- truncated file,
    empty or short segment, missing table, zero size -- raises JpegUnsupported (the documented contract: the caller catches it and
    decodes that file on the host), never IndexError / KeyError / ValueError."""
    try:
        info = _parse_header(data)
    except JpegUnsupported:
        raise
    except (IndexError, KeyError, ValueError, TypeError) as e:
        raise JpegUnsupported("malformed JPEG header (%s: %s)" % (type(e).__name__, e))
    if info.get("W", 0) <= 0 or info.get("H", 0) <= 0:
        raise JpegUnsupported("frame size %sx%s" % (info.get("W"), info.get("H")))
    for c in info["comps"]:
        if c["tq"] not in info["q"] or c.get("td") not in info["dc"] or c.get("ta") not in info["ac"]:
            raise JpegUnsupported("component %d refers to a quantisation / Huffman table the file does not define" % c["id"])
    return info


def _parse_header(data):
    d = bytes(data)
    if len(d) < 4 or d[0] != 0xFF or d[1] != 0xD8:
        raise JpegUnsupported("not a JPEG file (no SOI)")
    i, q, dc, ac = 2, {}, {}, {}
    info = {"restart": 0, "comps": None}
    while i + 4 <= len(d):
        if d[i] != 0xFF:
            raise JpegUnsupported("marker expected at byte %d" % i)
        while d[i + 1] == 0xFF:
            i += 1
        m = d[i + 1]
        L = (d[i + 2] << 8) | d[i + 3]
        seg = d[i + 4:i + 2 + L]
        if m == 0xDB:
            p = 0
            while p < len(seg):
                pq, tq = seg[p] >> 4, seg[p] & 15
                n = 128 if pq else 64
                if p + 1 + n > len(seg):
                    raise JpegUnsupported("quantisation table segment truncated")
                vals = np.frombuffer(seg[p + 1:p + 1 + n], dtype=">u2" if pq else np.uint8).astype(np.uint16)
                t = np.zeros(64, dtype=np.uint16)
                t[_ZIGZAG] = vals
                q[tq] = t
                p += 1 + n
        elif m == 0xC4:
            p = 0
            while p < len(seg):
                tc, th = seg[p] >> 4, seg[p] & 15
                bits = list(seg[p + 1:p + 17])
                n = sum(bits)
                if len(bits) != 16 or n > 256 or p + 17 + n > len(seg):
                    raise JpegUnsupported("Huffman table segment truncated or with more than 256 codes")
                (ac if tc else dc)[th] = (bits, list(seg[p + 17:p + 17 + n]))
                p += 17 + n
        elif m == 0xC0:
            if seg[0] != 8:
                raise JpegUnsupported("%d-bit samples" % seg[0])
            info["H"], info["W"] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            info["comps"] = [dict(id=seg[6 + 3 * c], h=seg[7 + 3 * c] >> 4, v=seg[7 + 3 * c] & 15, tq=seg[8 + 3 * c])
                             for c in range(seg[5])]
        elif 0xC1 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise JpegUnsupported("only baseline sequential Huffman JPEG (SOF0) is decoded on the GPU, this file is SOF%d" % (m - 0xC0))
        elif m == 0xDD:
            info["restart"] = (seg[0] << 8) | seg[1]
        elif m == 0xE1:
            o = _exif_orientation(seg)
            if o != 1:
                raise JpegUnsupported("EXIF orientation %d: cv2.imread would rotate / mirror this frame, the GPU decoder does not" % o)
        elif m == 0xDA:
            if info["comps"] is None or seg[0] != len(info["comps"]):
                raise JpegUnsupported("a single interleaved scan over all components is expected")
            for k in range(seg[0]):
                cid, t = seg[1 + 2 * k], seg[2 + 2 * k]
                c = info["comps"][k]
                if c["id"] != cid:
                    raise JpegUnsupported("scan component order differs from the frame header")
                c["td"], c["ta"] = t >> 4, t & 15
            info.update(q=q, dc=dc, ac=ac, scan=i + 2 + L)
            return info
        i += 2 + L
    raise JpegUnsupported("no SOS marker")


def huff_table(bits, vals):
    """jdhuff.c jpeg_make_d_derived_tbl as the kernel's 384-int record: 512 x u16 look-ahead entries (len << 8 | symbol; 0 = a code
    longer than 9 bits), maxcode[1..16] at [256 + l] (-1 = no code of that length), valoffset at [274 + l], huffval bytes at [292]."""
    rec = np.zeros(HT_INTS, dtype=np.int32)
    look = np.zeros(1 << LOOK, dtype=np.uint16)
    maxcode = np.full(18, -1, dtype=np.int32)
    valoff = np.zeros(17, dtype=np.int32)
    code, k = 0, 0
    for l in range(1, 17):
        n = bits[l - 1]
        if n:
            valoff[l] = k - code
            for _ in range(n):
                if l <= LOOK:
                    lo = code << (LOOK - l)
                    look[lo:lo + (1 << (LOOK - l))] = (l << 8) | vals[k]
                code += 1
                k += 1
            maxcode[l] = code - 1
        code <<= 1
    rec[:256] = look.view(np.int32)
    rec[256:274] = maxcode
    rec[274:291] = valoff
    hv = np.zeros(256, dtype=np.uint8)
    hv[:len(vals)] = vals
    rec[292:356] = hv.view(np.int32)
    return rec


def prepare(files):
    """files: list of bytes objects (whole .jpg files) -> the arrays nafae_jpeg_decode_batch takes + the shared geometry."""
    infos = [parse_header(f) for f in files]
    g0 = None
    qt, qidx, ht, hidx = [], {}, [], {}
    desc = np.zeros((len(files), DESC_INTS), dtype=np.int32)
    segs, chunks, pos = [], [], 0
    for n, (f, info) in enumerate(zip(files, infos)):
        comps = info["comps"]
        ncomp = len(comps)
        if ncomp not in (1, 3) or any((c["h"], c["v"]) != (1, 1) for c in comps[1:]) or (comps[0]["h"], comps[0]["v"]) not in \
                ((1, 1), (2, 1), (2, 2)) or (ncomp == 1 and (comps[0]["h"], comps[0]["v"]) != (1, 1)):
            raise JpegUnsupported("sampling factors %s" % [(c["h"], c["v"]) for c in comps])
        g = (info["W"], info["H"], ncomp, comps[0]["h"], comps[0]["v"])
        if g0 is None:
            g0 = g
        elif g != g0:
            raise NafaeOpError("all files of one decode_batch call must share size and sampling: %s vs %s" % (g, g0))
        f = bytes(f)
        desc[n, 0], desc[n, 1], desc[n, 2] = pos + info["scan"], len(f) - info["scan"], info["restart"]
        for c, comp in enumerate(comps):
            tq = info["q"][comp["tq"]]
            key = tq.tobytes()
            if key not in qidx:
                qidx[key] = len(qt)
                qt.append(tq)
            desc[n, 3 + c] = qidx[key]
            slots = []
            for cls, tid in (("dc", comp["td"]), ("ac", comp["ta"])):
                bits, vals = info[cls][tid]
                key = (tuple(bits), tuple(vals))
                if key not in hidx:
                    hidx[key] = len(ht)
                    ht.append(huff_table(bits, vals))
                slots.append(hidx[key])
            desc[n, 6 + c] = (slots[0] << 16) | slots[1]
        # restart intervals: every FF Dn inside the entropy-coded data is a marker (data bytes FF are followed by 00)
        W, H, _, h0, v0 = g
        nmcu = -(-W // (8 * h0)) * -(-H // (8 * v0))
        a = np.frombuffer(f, dtype=np.uint8)[info["scan"]:]
        starts = [0]
        if info["restart"]:
            ff = np.flatnonzero((a[:-1] == 0xFF) & (a[1:] >= 0xD0) & (a[1:] <= 0xD7))
            starts += [int(x) + 2 for x in ff]
            need = -(-nmcu // info["restart"])
            if len(starts) < need:
                raise JpegUnsupported("restart markers missing: %d intervals, %d expected" % (len(starts), need))
            starts = starts[:need]
        ri = info["restart"] or nmcu
        for k, s in enumerate(starts):
            segs.append((n, pos + info["scan"] + s, k * ri, min(ri, nmcu - k * ri)))
        chunks.append(a if False else np.frombuffer(f, dtype=np.uint8))
        pos += (len(f) + 7) & ~7                                    # every file starts 8-byte aligned
    stream = np.zeros(pos + 64, dtype=np.uint8)                      # (+ padding: the kernel reads whole 8-byte words)
    p = 0
    for c in chunks:
        stream[p:p + len(c)] = c
        p += (len(c) + 7) & ~7
    return dict(stream=stream, desc=desc, seg=np.array(segs, dtype=np.int32).reshape(-1, 4), qtabs=np.stack(qt).astype(np.uint16),
                hufftabs=np.stack(ht).astype(np.int32), geom=g0)


_ws = {}


def decode_batch(files, device="cuda", out=None):
    """list of JPEG file contents (bytes) -> uint8 [n, H, W, 3] BGR on `device`, bit-identical to cv2.imread / libjpeg."""
    P = prepare(files)
    W, H, ncomp, h0, v0 = P["geom"]
    n = len(files)
    dev = torch.device(device)
    t = {k: torch.from_numpy(P[k]).to(dev, non_blocking=True) for k in ("stream", "desc", "seg", "qtabs", "hufftabs")}
    L = _lib.lib()
    nws = int(L.nafae_jpeg_workspace_bytes(n, W, H, ncomp, h0, v0))
    if nws < 0:
        raise NafaeOpError("nafae_jpeg_workspace_bytes failed (%d)" % nws)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _ws.get(key)
    if ws is None or ws.numel() < nws:
        ws = _ws[key] = torch.empty(nws, device=dev, dtype=torch.uint8)
    if out is None:
        out = torch.empty(n, H, W, 3, device=dev, dtype=torch.uint8)
    elif tuple(out.shape) != (n, H, W, 3) or out.dtype != torch.uint8 or not out.is_contiguous() or not out.is_cuda:
        raise NafaeOpError("out must be a contiguous uint8 CUDA tensor [%d, %d, %d, 3]" % (n, H, W))
    _rc(L.nafae_jpeg_decode_batch(_p(t["stream"]), t["stream"].numel(), _p(t["desc"]), _p(t["seg"]), _p(t["qtabs"]), _p(t["hufftabs"]),
                                  n, t["seg"].shape[0], W, H, ncomp, h0, v0, _p(ws), ws.numel(), _p(out), _stream()),
        "nafae_jpeg_decode_batch")
    out._jpeg_inputs = t            # keep the staged inputs alive until the (asynchronous) kernels have run
    return out


def decode_files(paths, device="cuda"):
    """`[cv2.imread(p) for p in paths]` on the GPU: reads the files, decodes them in one batch."""
    datas = []
    for p in paths:
        with open(p, "rb") as f:
            datas.append(f.read())
    return decode_batch(datas, device=device)
