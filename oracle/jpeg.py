"""CPU restatement (numpy / plain Python; TEST INFRASTRUCTURE, never imported by nafae_amd/) of what `cv2.imread(path)` does to a
baseline JPEG frame in the reference's loader (lib/datasets/youcook2.py:212): libjpeg's default decompression --

    jdhuff.c   decode_mcu          Huffman decoding of the interleaved scan (DC prediction, EOB / ZRL, restart intervals)
    jidctint.c jpeg_idct_islow     dequantisation + the 13-bit fixed-point LL&M inverse DCT (the default JDCT_ISLOW)
    jdsample.c h2v1 / h2v2_fancy_upsample   triangle-filter chroma upsampling (do_fancy_upsampling = TRUE, the default)
    jdcolor.c  ycc_rgb_convert     16-bit fixed-point YCbCr -> RGB, then cv2's BGR channel order

-- all integer arithmetic, so the result is defined bit for bit.  cv2 is not installed here; PIL 12.2 links the same library
(libjpeg-turbo, API 6.2) with the same defaults, and tests/test_oracle_jpeg.py pins this restatement against `PIL.Image.open`
on generated files (4:4:4 / 4:2:2 / 4:2:0 / grey, odd sizes, restart markers, several qualities).  The header parser that the
product uses (nafae_amd/jpeg.py) is exercised by the same tests; this module re-parses on its own.
"""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47,
                   55, 62, 63], dtype=np.int64)          # jutils.c jpeg_natural_order


def parse(data):
    """Markers up to SOS of a baseline (SOF0), Huffman-coded, single-scan JPEG -> dict."""
    d = memoryview(data)
    assert d[0] == 0xFF and d[1] == 0xD8, "not a JPEG"
    i, q, dc, ac, info = 2, {}, {}, {}, {"restart": 0}
    while True:
        assert d[i] == 0xFF, "marker expected"
        while d[i + 1] == 0xFF:
            i += 1
        m = d[i + 1]
        L = (d[i + 2] << 8) | d[i + 3]
        seg = bytes(d[i + 4:i + 2 + L])
        if m == 0xDB:
            p = 0
            while p < len(seg):
                pq, tq = seg[p] >> 4, seg[p] & 15
                n = 128 if pq else 64
                vals = np.frombuffer(seg[p + 1:p + 1 + n], dtype=">u2" if pq else np.uint8).astype(np.int64)
                t = np.zeros(64, dtype=np.int64)
                t[ZIGZAG] = vals
                q[tq] = t
                p += 1 + n
        elif m == 0xC4:
            p = 0
            while p < len(seg):
                tc, th = seg[p] >> 4, seg[p] & 15
                bits = list(seg[p + 1:p + 17])
                n = sum(bits)
                (ac if tc else dc)[th] = (bits, list(seg[p + 17:p + 17 + n]))
                p += 17 + n
        elif m == 0xC0:
            assert seg[0] == 8, "8-bit samples only"
            info["H"], info["W"] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            info["comps"] = [dict(id=seg[6 + 3 * c], h=seg[7 + 3 * c] >> 4, v=seg[7 + 3 * c] & 15, tq=seg[8 + 3 * c]) for c in range(seg[5])]
        elif m in (0xC1, 0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise ValueError("only baseline sequential Huffman JPEG (SOF0) is supported, found SOF%d" % (m - 0xC0))
        elif m == 0xDD:
            info["restart"] = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            ns = seg[0]
            assert ns == len(info["comps"]), "single interleaved scan expected"
            for k in range(ns):
                cid, t = seg[1 + 2 * k], seg[2 + 2 * k]
                c = [c for c in info["comps"] if c["id"] == cid][0]
                c["td"], c["ta"] = t >> 4, t & 15
            info.update(q=q, dc=dc, ac=ac, scan=i + 2 + L)
            return info
        i += 2 + L


def _huff_table(bits, vals):
    """jdhuff.c jpeg_make_d_derived_tbl: code -> symbol as a dict keyed by (length, code)."""
    codes, code, k = {}, 0, 0
    for l in range(1, 17):
        for _ in range(bits[l - 1]):
            codes[(l, code)] = vals[k]
            code += 1
            k += 1
        code <<= 1
    return codes


class _Bits:
    def __init__(self, data, pos):
        self.d, self.p, self.acc, self.n = data, pos, 0, 0

    def _fill(self):
        while self.n <= 24:
            b = 0
            if self.p < len(self.d):
                b = self.d[self.p]
                if b == 0xFF:
                    if self.p + 1 < len(self.d) and self.d[self.p + 1] == 0:
                        self.p += 2
                    else:
                        b = 0                              # a marker: feed zeros (jdhuff.c jpeg_fill_bit_buffer)
                else:
                    self.p += 1
            self.acc = ((self.acc << 8) | b) & 0xFFFFFFFFFFFF
            self.n += 8

    def get(self, k):
        if k == 0:
            return 0
        self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def decode(self, table):
        code = 0
        for l in range(1, 17):
            code = (code << 1) | self.get(1)
            s = table.get((l, code))
            if s is not None:
                return s
        raise ValueError("bad Huffman code")

    def restart(self):
        self.acc = self.n = 0                              # discard the partial byte, expect RSTn
        assert self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7, "RSTn expected"
        self.p += 2


def _extend(r, s):
    return r if r >= (1 << (s - 1)) else r - (1 << s) + 1     # HUFF_EXTEND


def decode_coefficients(data, info):
    """-> list per component of int array [blocks_y, blocks_x, 64] (natural order, quantised)."""
    comps = info["comps"]
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    mx, my = -(-info["W"] // (8 * hmax)), -(-info["H"] // (8 * vmax))
    out = [np.zeros((my * c["v"], mx * c["h"], 64), dtype=np.int64) for c in comps]
    dct = {k: _huff_table(*v) for k, v in info["dc"].items()}
    act = {k: _huff_table(*v) for k, v in info["ac"].items()}
    br = _Bits(data, info["scan"])
    pred = [0] * len(comps)
    for mcu in range(mx * my):
        if info["restart"] and mcu and mcu % info["restart"] == 0:
            br.restart()
            pred = [0] * len(comps)
        y0, x0 = divmod(mcu, mx)
        for ci, c in enumerate(comps):
            for by in range(c["v"]):
                for bx in range(c["h"]):
                    blk = out[ci][y0 * c["v"] + by, x0 * c["h"] + bx]
                    s = br.decode(dct[c["td"]])
                    if s:
                        pred[ci] += _extend(br.get(s), s)
                    blk[0] = pred[ci]
                    k = 1
                    while k < 64:
                        rs = br.decode(act[c["ta"]])
                        r, s = rs >> 4, rs & 15
                        if s:
                            k += r
                            blk[ZIGZAG[k]] = _extend(br.get(s), s)
                            k += 1
                        elif r == 15:
                            k += 16
                        else:
                            break
    return out


C = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069, f2053=16819,
         f2562=20995, f3072=25172)


def _idct_1d(x, shift):
    """jidctint.c, one pass over the last axis of x [..., 8] (int64), DESCALE by `shift`."""
    z2, z3 = x[..., 2], x[..., 6]
    z1 = (z2 + z3) * C["f0541"]
    tmp2 = z1 - z3 * C["f1847"]
    tmp3 = z1 + z2 * C["f0765"]
    tmp0 = (x[..., 0] + x[..., 4]) << 13
    tmp1 = (x[..., 0] - x[..., 4]) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = x[..., 7], x[..., 5], x[..., 3], x[..., 1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * C["f1175"]
    t0, t1, t2, t3 = t0 * C["f0298"], t1 * C["f2053"], t2 * C["f3072"], t3 * C["f1501"]
    z1, z2, z3, z4 = -z1 * C["f0899"], -z2 * C["f2562"], -z3 * C["f1961"] + z5, -z4 * C["f0390"] + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    r = 1 << (shift - 1)
    return np.stack([(tmp10 + t3 + r) >> shift, (tmp11 + t2 + r) >> shift, (tmp12 + t1 + r) >> shift, (tmp13 + t0 + r) >> shift,
                     (tmp13 - t0 + r) >> shift, (tmp12 - t1 + r) >> shift, (tmp11 - t2 + r) >> shift, (tmp10 - t3 + r) >> shift], -1)


def idct_islow(coef, quant):
    """coef [..., 64] quantised, natural order -> uint8 samples [..., 8, 8]."""
    x = (coef * quant).reshape(coef.shape[:-1] + (8, 8))
    ws = _idct_1d(np.swapaxes(x, -1, -2), 13 - 2)              # pass 1: columns -> ws[col][row]
    out = _idct_1d(np.swapaxes(ws, -1, -2), 13 + 2 + 3)        # pass 2: rows
    return np.clip(out + 128, 0, 255).astype(np.uint8)


def _plane(blocks):
    by, bx = blocks.shape[:2]
    return blocks.transpose(0, 2, 1, 3).reshape(by * 8, bx * 8)


def upsample_h2v1(p, wc):
    """jdsample.c h2v1_fancy_upsample (downsampled_width > 2), else box replication; p [rows, >= wc] -> [rows, 2 wc]."""
    p = p[:, :wc].astype(np.int64)
    if wc <= 2:
        return np.repeat(p, 2, 1)
    out = np.empty((p.shape[0], 2 * wc), dtype=np.int64)
    left = np.concatenate([p[:, :1], p[:, :-1]], 1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out


def upsample_h2v2(p, wc, hc):
    """jdsample.c h2v2_fancy_upsample with jdmainct.c's edge context rows; p [>= hc, >= wc] -> [2 hc, 2 wc]."""
    p = p[:hc, :wc].astype(np.int64)
    if wc <= 2:
        return np.repeat(np.repeat(p, 2, 0), 2, 1)
    above = np.concatenate([p[:1], p[:-1]], 0)
    below = np.concatenate([p[1:], p[-1:]], 0)
    out = np.empty((2 * hc, 2 * wc), dtype=np.int64)
    for v, far in ((0, above), (1, below)):
        cs = 3 * p + far                                        # thiscolsum of every column
        last = np.concatenate([cs[:, :1], cs[:, :-1]], 1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], 1)
        o = np.empty((hc, 2 * wc), dtype=np.int64)
        o[:, 0::2] = (3 * cs + last + 8) >> 4
        o[:, 1::2] = (3 * cs + nxt + 7) >> 4
        o[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        o[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2] = o
    return out


def ycc_to_bgr(y, cb, cr):
    """jdcolor.c ycc_rgb_convert (SCALEBITS 16) in cv2's channel order."""
    y, cb, cr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([b, g, r], -1), 0, 255).astype(np.uint8)


def decode(data):
    """bytes of a baseline JPEG -> uint8 [H, W, 3] BGR, what cv2.imread returns (IMREAD_COLOR)."""
    info = parse(data)
    H, W, comps = info["H"], info["W"], info["comps"]
    coefs = decode_coefficients(data, info)
    planes = [_plane(idct_islow(cf, info["q"][c["tq"]])) for cf, c in zip(coefs, comps)]
    if len(comps) == 1:
        yy = planes[0][:H, :W]
        return np.stack([yy, yy, yy], -1)
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    full = []
    for p, c in zip(planes, comps):
        wc, hc = -(-W * c["h"] // hmax), -(-H * c["v"] // vmax)
        if (c["h"], c["v"]) == (hmax, vmax):
            full.append(p[:H, :W])
        elif c["h"] * 2 == hmax and c["v"] == vmax:
            full.append(upsample_h2v1(p[:hc], wc)[:H, :W])
        elif c["h"] * 2 == hmax and c["v"] * 2 == vmax:
            full.append(upsample_h2v2(p, wc, hc)[:H, :W])
        else:
            raise ValueError("unsupported sampling factors %s" % [(c["h"], c["v"]) for c in comps])
    return ycc_to_bgr(*full)
