// jpeg_core.h -- the arithmetic of csrc/jpeg.hip as plain host / device functions: the kernels call them per lane, and
// tests/jpeg_host_harness.cpp compiles the very same functions with g++ to check them against libjpeg (PIL) without a GPU.
// No HIP built-ins in here.  What each piece restates: see jpeg.hip.
#pragma once
#include <stdint.h>

#ifndef NAFAE_HD
#define NAFAE_HD __host__ __device__ __forceinline__       // (the g++ harness defines it as `inline`)
#define NAFAE_DEVCONST __device__ __constant__
#endif

namespace nafae_jpeg {

constexpr int DESC_INTS = 32;      // per-image descriptor (nafae_amd/jpeg.py builds it)
// [0] byte offset of the scan's entropy data in `stream`  [1] bytes available from there (up to the end of the file)
// [2] restart interval in MCUs (0: none)                   [3 + c] quantisation table slot of component c
// [6 + c] (DC table slot << 16) | AC table slot of component c
constexpr int HT_INTS = 384;       // per Huffman table: 256 ints = 512 x u16 look-ahead (len << 8 | symbol; 0: longer than 9 bits),
                                   // 18 ints maxcode[1..16] at [256 + l] (-1: no code of that length), 17 ints valoffset at [274 + l],
                                   // 64 ints = huffval[256] bytes at [292]
constexpr int LOOK = 9;

NAFAE_DEVCONST unsigned char k_natural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Geom {
  int W, H, ncomp, h0, v0;           // luma sampling (chroma 1 x 1)
  int mx, my;                        // MCUs
  int bx[3], by[3], boff[3], nblk;   // blocks per component plane, block offset of the plane inside an image, blocks per image
  int pw[3], ph[3], poff[3], psize;  // sample planes (padded to whole blocks): width, height, byte offset inside an image, bytes per image
};
NAFAE_HD Geom make_geom(int W, int H, int ncomp, int h0, int v0) {
  Geom g;
  g.W = W; g.H = H; g.ncomp = ncomp; g.h0 = h0; g.v0 = v0;
  g.mx = (W + 8 * h0 - 1) / (8 * h0);
  g.my = (H + 8 * v0 - 1) / (8 * v0);
  int bo = 0, po = 0;
  for (int c = 0; c < 3; c++) {
    const int h = c == 0 ? h0 : 1, v = c == 0 ? v0 : 1;
    g.bx[c] = c < ncomp ? g.mx * h : 0;
    g.by[c] = c < ncomp ? g.my * v : 0;
    g.boff[c] = bo;
    bo += g.bx[c] * g.by[c];
    g.pw[c] = g.bx[c] * 8;
    g.ph[c] = g.by[c] * 8;
    g.poff[c] = po;
    po += g.pw[c] * g.ph[c];
  }
  g.nblk = bo;
  g.psize = po;
  return g;
}

// ---------------------------------------------------------------------------------------------------- entropy decoding
// The entropy-coded bytes are read through `buf[i & mask]`: on the GPU `buf` is a ring in LDS that the wave refills between MCUs
// (jpeg.hip), on the host it is the stream itself (mask = all ones).  `pos` / `lim`: next unread byte and the end of the data, as
// indices.  jdhuff.c jpeg_fill_bit_buffer: FF 00 is a stuffed FF, any other FF xx ends the interval (zero bits from there on).
struct BitReader {
  const unsigned char *buf;
  unsigned mask, pos, lim;
  unsigned long long acc;            // bit reservoir, newest bits at the bottom
  int nbits;
  bool marker;                       // a marker was reached: zeros from here on (jdhuff.c: "fill with zero bits")

  NAFAE_HD void init(const unsigned char *b, unsigned m, unsigned p0, unsigned l) {
    buf = b; mask = m; pos = p0; lim = l; acc = 0; nbits = 0; marker = false;
  }
  NAFAE_HD unsigned byte_at(unsigned i) const { return buf[i & mask]; }
  // at least 32 valid bits afterwards (a Huffman code is at most 16 bits, a value at most 16)
  NAFAE_HD void fill() {
    if (nbits >= 32) return;
    if (!marker && pos + 4 <= lim) {                       // four bytes at once unless one of them is FF
      const unsigned w = (byte_at(pos) << 24) | (byte_at(pos + 1) << 16) | (byte_at(pos + 2) << 8) | byte_at(pos + 3);
      const unsigned v = ~w;
      if (((v - 0x01010101u) & ~v & 0x80808080u) == 0) {   // no zero byte in ~w
        acc = (acc << 32) | w;
        nbits += 32;
        pos += 4;
        return;
      }
    }
    while (nbits < 32) {
      unsigned b = 0;
      if (!marker) {
        if (pos >= lim) {
          marker = true;
        } else {
          b = byte_at(pos++);
          if (b == 0xffu) {
            const unsigned b2 = pos < lim ? byte_at(pos) : 0xd9u;
            pos++;
            if (b2 != 0) {             // RSTn / EOI / anything else: the interval's data ends here
              marker = true;
              b = 0;
            }
          }
        }
      }
      acc = (acc << 8) | b;
      nbits += 8;
    }
  }
  NAFAE_HD int get(int s) {       // s >= 1 bits (after fill)
    nbits -= s;
    return (int)((acc >> nbits) & ((1ull << s) - 1));
  }
};

NAFAE_HD int huff_decode(BitReader &br, const int *tab) {
  br.fill();
  const unsigned look = (unsigned)(br.acc >> (br.nbits - LOOK)) & ((1u << LOOK) - 1);
  const unsigned e = reinterpret_cast<const unsigned short *>(tab)[look];
  if (e) {
    br.nbits -= (int)(e >> 8);
    return (int)(e & 0xffu);
  }
  const unsigned code16 = (unsigned)(br.acc >> (br.nbits - 16)) & 0xffffu;
  for (int l = LOOK + 1; l <= 16; l++) {
    const int c = (int)(code16 >> (16 - l));
    if (c <= tab[256 + l]) {
      br.nbits -= l;
      return (int)reinterpret_cast<const unsigned char *>(tab + 292)[(c + tab[274 + l]) & 255];
    }
  }
  br.nbits -= 16;                      // corrupt data: consume and carry on (libjpeg warns and returns 0)
  return 0;
}

NAFAE_HD int huff_extend(int r, int s) { return r < (1 << (s - 1)) ? r - (1 << s) + 1 : r; }

// One MCU (jdhuff.c decode_mcu) of image-local coefficient array `cimg` (int16 [g.nblk][64], zero-initialised).  tabs: the
// components' Huffman records, [2 c] = DC, [2 c + 1] = AC, HT_INTS ints each; nat: the zigzag -> natural map; pred: the DC predictions.
// An MCU consumes at most 6 blocks x 64 x 26 bits = 1 248 bytes, twice that with every byte stuffed.
NAFAE_HD void huffman_mcu(BitReader &br, const int *tabs, const unsigned char *nat, const Geom &g, int mcu, int (&pred)[3], short *cimg) {
  const int y0 = mcu / g.mx, x0 = mcu - y0 * g.mx;
  for (int c = 0; c < g.ncomp; c++) {
    const int h = c == 0 ? g.h0 : 1, v = c == 0 ? g.v0 : 1;
    const int *dct = tabs + (2 * c) * HT_INTS, *act = tabs + (2 * c + 1) * HT_INTS;
    for (int by = 0; by < v; by++)
      for (int bx = 0; bx < h; bx++) {
        short *blk = cimg + ((size_t)g.boff[c] + (size_t)(y0 * v + by) * g.bx[c] + (x0 * h + bx)) * 64;
        int sy = huff_decode(br, dct);
        if (sy) {
          br.fill();
          pred[c] += huff_extend(br.get(sy), sy);
        }
        blk[0] = (short)pred[c];
        for (int k = 1; k < 64;) {
          const int rs = huff_decode(br, act);
          const int r = rs >> 4;
          sy = rs & 15;
          if (sy) {
            k += r;
            br.fill();
            const int val = huff_extend(br.get(sy), sy);
            if (k < 64) blk[nat[k]] = (short)val;
            k++;
          } else if (r == 15) {
            k += 16;
          } else {
            break;
          }
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------- parallel entropy decoding
// The same decoding by MANY lanes per restart interval (self-synchronising Huffman decoding: a decoder started at a wrong place of a
// Huffman-coded stream falls into step with the right one after a few symbols).  The interval's bytes are first compacted -- stuffed
// zeros removed, cut at the first marker -- into big-endian 32-bit words `d32`, so that a position is a plain bit index and a reader
// carries no state.  Lane i owns the bits [i S, (i + 1) S): round 0 decodes them from a guessed state (block 0, DC next), every later
// round re-decodes them from the exit state of lane i - 1, until no exit state changes -- then every lane's entry state is the
// sequential decoder's (lane 0's always was).  A prefix sum of the MCUs each lane completed gives it its absolute position, a last
// pass writes the coefficients (DC as DIFFERENCES: the predictions are a prefix sum over the blocks afterwards).
struct SpanState {
  unsigned bp;                       // next unread bit
  int b, k;                          // block inside the MCU, next coefficient (0: the DC symbol comes next)
};
NAFAE_HD bool same_state(const SpanState &a, const SpanState &b) { return a.bp == b.bp && a.b == b.b && a.k == b.k; }

// 32 bits from bit bp on (MSB first); d32 has two zero words beyond the data
NAFAE_HD unsigned peek32(const uint32_t *d32, unsigned bp) {
  const unsigned j = bp >> 5, sh = bp & 31;
  const unsigned long long w = ((unsigned long long)d32[j] << 32) | d32[j + 1];
  return (unsigned)((w << sh) >> 32);
}
// the symbol whose code starts at the top of v; len: its code length
NAFAE_HD int huff_peek(unsigned v, const int *tab, int &len) {
  const unsigned e = reinterpret_cast<const unsigned short *>(tab)[v >> (32 - LOOK)];
  if (e) {
    len = (int)(e >> 8);
    return (int)(e & 0xffu);
  }
  const unsigned code16 = v >> 16;
  for (int l = LOOK + 1; l <= 16; l++) {
    const int c = (int)(code16 >> (16 - l));
    if (c <= tab[256 + l]) {
      len = l;
      return (int)reinterpret_cast<const unsigned char *>(tab + 292)[(c + tab[274 + l]) & 255];
    }
  }
  len = 16;                          // corrupt data: consume and carry on, as huff_decode does
  return 0;
}

// Decode symbols from state `s` while s.bp < end_bp (and < total_bits); returns the MCUs completed.  WRITE: store the coefficients of
// MCUs [.., mcu_end) -- `mcu` is the index (inside the image) of the MCU the state is in.
template <bool WRITE>
NAFAE_HD int span_decode(const uint32_t *d32, unsigned total_bits, unsigned end_bp, SpanState &s, const int *tabs, const unsigned char *nat,
                         const Geom &g, int mcu, int mcu_end, short *cimg) {
  const int ny = g.h0 * g.v0, nb = g.ncomp == 1 ? 1 : ny + 2;
  const unsigned stop = end_bp < total_bits ? end_bp : total_bits;
  int done = 0;
  unsigned bp = s.bp;
  int b = s.b, k = s.k;
  while (bp < stop) {
    const int c = b < ny ? 0 : b - ny + 1;
    const unsigned v = peek32(d32, bp);
    int len;
    if (k == 0) {
      const int sy = huff_peek(v, tabs + (2 * c) * HT_INTS, len) & 15;
      int diff = 0;
      if (sy) diff = huff_extend((int)((v << len) >> (32 - sy)), sy);
      bp += (unsigned)(len + sy);
      if (WRITE && mcu < mcu_end) {
        const int y0 = mcu / g.mx, x0 = mcu - y0 * g.mx;
        const int h = c == 0 ? g.h0 : 1, vv = c == 0 ? g.v0 : 1, by = c == 0 ? b / g.h0 : 0, bx = c == 0 ? b - by * g.h0 : 0;
        cimg[((size_t)g.boff[c] + (size_t)(y0 * vv + by) * g.bx[c] + (x0 * h + bx)) * 64] = (short)diff;
      }
      k = 1;
    } else {
      const int rs = huff_peek(v, tabs + (2 * c + 1) * HT_INTS, len);
      const int r = rs >> 4, sy = rs & 15;
      if (sy) {
        k += r;
        const int val = huff_extend((int)((v << len) >> (32 - sy)), sy);
        if (WRITE && k < 64 && mcu < mcu_end) {
          const int y0 = mcu / g.mx, x0 = mcu - y0 * g.mx;
          const int h = c == 0 ? g.h0 : 1, vv = c == 0 ? g.v0 : 1, by = c == 0 ? b / g.h0 : 0, bx = c == 0 ? b - by * g.h0 : 0;
          cimg[((size_t)g.boff[c] + (size_t)(y0 * vv + by) * g.bx[c] + (x0 * h + bx)) * 64 + nat[k]] = (short)val;
        }
        k++;
        bp += (unsigned)(len + sy);
      } else {
        bp += (unsigned)len;
        k = r == 15 ? k + 16 : 64;   // ZRL / end of block
      }
    }
    if (k >= 64) {
      k = 0;
      if (++b == nb) {
        b = 0;
        mcu++;
        done++;
      }
    }
  }
  s.bp = bp;
  s.b = b;
  s.k = k;
  return done;
}

// block n (decode order) of component c inside an interval that starts at MCU m0 -> its index in the image's coefficient array
NAFAE_HD size_t dc_block(const Geom &g, int c, int m0, int n) {
  const int h = c == 0 ? g.h0 : 1, v = c == 0 ? g.v0 : 1;
  const int per = h * v, mcu = m0 + n / per, sub = n - (n / per) * per;
  const int y0 = mcu / g.mx, x0 = mcu - y0 * g.mx, by = sub / h, bx = sub - by * h;
  return (size_t)g.boff[c] + (size_t)(y0 * v + by) * g.bx[c] + (x0 * h + bx);
}

// ---------------------------------------------------------------------------------------------------- inverse DCT
constexpr int FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270, FIX_0_899976223 = 7373,
              FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137, FIX_1_961570560 = 16069,
              FIX_2_053119869 = 16819, FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;

// one 8-point pass of jidctint.c: x[0..7] -> o[0..7], DESCALE by SHIFT
template <int SHIFT>
NAFAE_HD void idct8(const int (&x)[8], int (&o)[8]) {
  int z2 = x[2], z3 = x[6];
  int z1 = (z2 + z3) * FIX_0_541196100;
  const int tmp2 = z1 + z3 * (-FIX_1_847759065);
  const int tmp3 = z1 + z2 * FIX_0_765366865;
  const int tmp0 = (x[0] + x[4]) << 13, tmp1 = (x[0] - x[4]) << 13;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  int t0 = x[7], t1 = x[5], t2 = x[3], t3 = x[1];
  z1 = t0 + t3;
  z2 = t1 + t2;
  z3 = t0 + t2;
  int z4 = t1 + t3;
  const int z5 = (z3 + z4) * FIX_1_175875602;
  t0 *= FIX_0_298631336;
  t1 *= FIX_2_053119869;
  t2 *= FIX_3_072711026;
  t3 *= FIX_1_501321110;
  z1 *= -FIX_0_899976223;
  z2 *= -FIX_2_562915447;
  z3 = z3 * (-FIX_1_961570560) + z5;
  z4 = z4 * (-FIX_0_390180644) + z5;
  t0 += z1 + z3;
  t1 += z2 + z4;
  t2 += z2 + z3;
  t3 += z1 + z4;
  constexpr int R = 1 << (SHIFT - 1);
  o[0] = (tmp10 + t3 + R) >> SHIFT;
  o[7] = (tmp10 - t3 + R) >> SHIFT;
  o[1] = (tmp11 + t2 + R) >> SHIFT;
  o[6] = (tmp11 - t2 + R) >> SHIFT;
  o[2] = (tmp12 + t1 + R) >> SHIFT;
  o[5] = (tmp12 - t1 + R) >> SHIFT;
  o[3] = (tmp13 + t0 + R) >> SHIFT;
  o[4] = (tmp13 - t0 + R) >> SHIFT;
}


NAFAE_HD int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// ---------------------------------------------------------------------------------------------------- upsampling + colour
// chroma sample of component plane p at full-resolution position (X, Y)
NAFAE_HD int chroma_at(const unsigned char *p, int pw, int wc, int hc, int h0, int v0, int X, int Y) {
  if (h0 == 1) return p[(size_t)Y * pw + X];                                   // 4:4:4
  const int i = X >> 1;
  if (v0 == 1) {                                                                // 4:2:2: h2v1_fancy_upsample
    const unsigned char *row = p + (size_t)Y * pw;
    if (wc <= 2) return row[i];
    const int v = row[i];
    if (X & 1) return i == wc - 1 ? v : (3 * v + row[i + 1] + 2) >> 2;
    return i == 0 ? v : (3 * v + row[i - 1] + 1) >> 2;
  }
  const int jr = Y >> 1;                                                        // 4:2:0: h2v2_fancy_upsample
  if (wc <= 2) return p[(size_t)jr * pw + i];
  int jf = (Y & 1) ? jr + 1 : jr - 1;                                           // the further row, replicated at the edges
  jf = jf < 0 ? 0 : (jf > hc - 1 ? hc - 1 : jf);
  const unsigned char *rn = p + (size_t)jr * pw, *rf = p + (size_t)jf * pw;
  const int cs = 3 * rn[i] + rf[i];
  if (X & 1) {
    if (i == wc - 1) return (cs * 4 + 7) >> 4;
    return (cs * 3 + 3 * rn[i + 1] + rf[i + 1] + 7) >> 4;
  }
  if (i == 0) return (cs * 4 + 8) >> 4;
  return (cs * 3 + 3 * rn[i - 1] + rf[i - 1] + 8) >> 4;
}


// pixel (X, Y) of an image whose component planes start at pi -> B, G, R
NAFAE_HD void color_pixel(const unsigned char *pi, const Geom &g, int X, int Y, unsigned char *bgr) {
  const int y = pi[g.poff[0] + (size_t)Y * g.pw[0] + X];
  int R = y, G = y, B = y;
  if (g.ncomp == 3) {
    const int wc = (g.W + g.h0 - 1) / g.h0, hc = (g.H + g.v0 - 1) / g.v0;       // downsampled_width / _height
    const int cb = chroma_at(pi + g.poff[1], g.pw[1], wc, hc, g.h0, g.v0, X, Y) - 128;
    const int cr = chroma_at(pi + g.poff[2], g.pw[2], wc, hc, g.h0, g.v0, X, Y) - 128;
    R = clamp255(y + ((91881 * cr + 32768) >> 16));                              // jdcolor.c build_ycc_rgb_table, SCALEBITS 16
    G = clamp255(y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    B = clamp255(y + ((116130 * cb + 32768) >> 16));
  }
  bgr[0] = (unsigned char)B;
  bgr[1] = (unsigned char)G;
  bgr[2] = (unsigned char)R;
}

}  // namespace nafae_jpeg
