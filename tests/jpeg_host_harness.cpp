// Test infrastructure: the arithmetic of nafae_amd/csrc/jpeg.hip (jpeg_core.h -- the very functions the kernels call per lane)
// compiled for the HOST with g++ and driven sequentially, so that the bit reader, the Huffman records nafae_amd/jpeg.py builds, the
// ISLOW IDCT and the fancy upsampling can be checked against libjpeg (PIL) without a GPU (tests/test_jpeg_cpu.py).  Never part of
// libnafae_hip.so and never loaded by nafae_amd/.
#include <stdint.h>
#include <string.h>
#include <vector>
#define NAFAE_HD inline
#define NAFAE_DEVCONST static const
#include "../nafae_amd/csrc/jpeg_core.h"

using namespace nafae_jpeg;

// The many-lane entropy decoder of jpeg.hip (jpeg_huffman_par_kernel) with its lanes run one after the other: compaction, round 0 from
// guessed states, re-decoding rounds until no exit state changes, MCU prefix sum, writing pass, DC prefix sums.  Returns the number of
// rounds, or -1 where the kernel would hand the interval to the one-lane decoder (fewer MCUs in the data than the interval has).
static int par_interval(const uint8_t *raw, long avail, const int *tabs, const Geom &g, int m0, int nm, short *cimg, int lanes) {
  std::vector<uint32_t> d32((size_t)(avail > 0 ? avail : 0) / 4 + 4, 0);
  unsigned n = 0;
  for (long i = 0; i < avail; i++) {
    if (raw[i] == 0xff) {
      const unsigned nx = i + 1 < avail ? raw[i + 1] : 0xd9u;
      if (nx != 0) break;                                    // marker: the data ends
      d32[n >> 2] |= 0xffu << (8 * (3 - (n & 3)));
      n++;
      i++;                                                   // the stuffed zero
      continue;
    }
    d32[n >> 2] |= (uint32_t)raw[i] << (8 * (3 - (n & 3)));
    n++;
  }
  const unsigned total = n * 8;
  unsigned S = ((total + lanes - 1) / lanes + 31) & ~31u;
  if (S < 128) S = 128;
  std::vector<SpanState> in(lanes), ex(lanes), ex2(lanes);
  std::vector<int> cnt(lanes, 0);
  for (int i = 0; i < lanes; i++) {
    in[i] = SpanState{(unsigned)i * S, 0, 0};
    ex[i] = in[i];
    cnt[i] = span_decode<false>(d32.data(), total, (unsigned)(i + 1) * S, ex[i], tabs, k_natural, g, 0, 0, nullptr);
  }
  int rounds = 1;
  for (;; rounds++) {
    bool changed = false;
    ex2 = ex;
    for (int i = 1; i < lanes; i++)
      if (!same_state(ex[i - 1], in[i])) {
        in[i] = ex[i - 1];
        SpanState s = in[i];
        cnt[i] = span_decode<false>(d32.data(), total, (unsigned)(i + 1) * S, s, tabs, k_natural, g, 0, 0, nullptr);
        if (!same_state(s, ex[i])) changed = true;
        ex2[i] = s;
      }
    ex = ex2;
    if (!changed) break;
  }
  int tot = 0;
  std::vector<int> start(lanes);
  for (int i = 0; i < lanes; i++) {
    start[i] = tot;
    tot += cnt[i];
  }
  if (tot < nm) return -1;
  for (int i = 0; i < lanes; i++) {
    SpanState s = in[i];
    span_decode<true>(d32.data(), total, (unsigned)(i + 1) * S, s, tabs, k_natural, g, m0 + start[i], m0 + nm, cimg);
  }
  for (int c = 0; c < g.ncomp; c++) {
    const int per = c == 0 ? g.h0 * g.v0 : 1;
    int pred = 0;
    for (int b = 0; b < nm * per; b++) {
      short *blk = cimg + dc_block(g, c, m0, b) * 64;
      pred += blk[0];
      blk[0] = (short)pred;
    }
  }
  return rounds;
}

// lanes == 0: the one-lane decoder (huffman_mcu); lanes > 0: the many-lane algorithm emulated with that many lanes per interval
extern "C" int jpeg_host_decode_lanes(const uint8_t *stream, const int32_t *desc, const int32_t *seg, const uint16_t *qtabs,
                                      const int32_t *hufftabs, int n, int nseg, int W, int H, int ncomp, int h0, int v0, uint8_t *out,
                                      int lanes, int *max_rounds) {
  const Geom g = make_geom(W, H, ncomp, h0, v0);
  std::vector<short> coef((size_t)n * g.nblk * 64, 0);
  std::vector<unsigned char> planes((size_t)n * g.psize, 0);
  std::vector<int> tabs(6 * HT_INTS);
  for (int s = 0; s < nseg; s++) {
    const int32_t *sg = seg + 4 * s;
    const int img = sg[0];
    const int32_t *d = desc + (size_t)img * DESC_INTS;
    for (int c = 0; c < ncomp; c++) {
      const int t = d[6 + c];
      memcpy(&tabs[(2 * c) * HT_INTS], hufftabs + (size_t)(t >> 16) * HT_INTS, HT_INTS * 4);
      memcpy(&tabs[(2 * c + 1) * HT_INTS], hufftabs + (size_t)(t & 0xffff) * HT_INTS, HT_INTS * 4);
    }
    const long off = sg[1], avail = (long)d[0] + d[1] - off;
    short *cimg = coef.data() + (size_t)img * g.nblk * 64;
    int rounds = -1;
    if (lanes > 0) {
      rounds = par_interval(stream + off, avail, tabs.data(), g, sg[2], sg[3], cimg, lanes);
      if (max_rounds && rounds > *max_rounds) *max_rounds = rounds;
    }
    if (rounds < 0) {
      BitReader br;
      br.init(stream + off, 0xffffffffu, 0u, (unsigned)(avail > 0 ? avail : 0));
      int pred[3] = {0, 0, 0};
      for (int mcu = sg[2]; mcu < sg[2] + sg[3]; mcu++) huffman_mcu(br, tabs.data(), k_natural, g, mcu, pred, cimg);
    }
  }
  for (long b = 0; b < (long)n * g.nblk; b++) {
    const int img = (int)(b / g.nblk);
    int bi = (int)(b - (long)img * g.nblk);
    const int c = (ncomp > 2 && bi >= g.boff[2]) ? 2 : ((ncomp > 1 && bi >= g.boff[1]) ? 1 : 0);
    bi -= g.boff[c];
    const uint16_t *q = qtabs + (size_t)desc[(size_t)img * DESC_INTS + 3 + c] * 64;
    const short *cf = coef.data() + (size_t)b * 64;
    int ws[64];
    for (int j = 0; j < 8; j++) {
      int x[8], o[8];
      for (int r = 0; r < 8; r++) x[r] = (int)cf[r * 8 + j] * (int)q[r * 8 + j];
      idct8<13 - 2>(x, o);
      for (int r = 0; r < 8; r++) ws[r * 8 + j] = o[r];
    }
    const int byy = bi / g.bx[c], bxx = bi - byy * g.bx[c];
    for (int j = 0; j < 8; j++) {
      int x[8], o[8];
      for (int r = 0; r < 8; r++) x[r] = ws[j * 8 + r];
      idct8<13 + 2 + 3>(x, o);
      unsigned char *dst = planes.data() + (size_t)img * g.psize + g.poff[c] + (size_t)(byy * 8 + j) * g.pw[c] + bxx * 8;
      for (int r = 0; r < 8; r++) dst[r] = (unsigned char)clamp255(o[r] + 128);
    }
  }
  for (int img = 0; img < n; img++)
    for (int Y = 0; Y < H; Y++)
      for (int X = 0; X < W; X++)
        color_pixel(planes.data() + (size_t)img * g.psize, g, X, Y, out + (((size_t)img * H + Y) * W + X) * 3);
  return 0;
}

extern "C" int jpeg_host_decode(const uint8_t *stream, const int32_t *desc, const int32_t *seg, const uint16_t *qtabs,
                                const int32_t *hufftabs, int n, int nseg, int W, int H, int ncomp, int h0, int v0, uint8_t *out) {
  return jpeg_host_decode_lanes(stream, desc, seg, qtabs, hufftabs, n, nseg, W, H, ncomp, h0, v0, out, 0, nullptr);
}
