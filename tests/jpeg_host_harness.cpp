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

extern "C" int jpeg_host_decode(const uint8_t *stream, const int32_t *desc, const int32_t *seg, const uint16_t *qtabs,
                                const int32_t *hufftabs, int n, int nseg, int W, int H, int ncomp, int h0, int v0, uint8_t *out) {
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
    huffman_interval(stream + off, stream + off + (avail > 0 ? avail : 0), tabs.data(), k_natural, g, sg[2], sg[2] + sg[3],
                     coef.data() + (size_t)img * g.nblk * 64);
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
