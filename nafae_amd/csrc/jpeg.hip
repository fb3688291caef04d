// jpeg.hip -- baseline JPEG decoding on the GPU: what `cv2.imread(img_path)` does per frame in the reference's loader
// (lib/datasets/youcook2.py:212; SURVEY.md section 8(f)2, "replace cv2 JPEG decode with a GPU-side decode").  libjpeg's default
// decompression restated, integer arithmetic throughout, so the frames are bit-identical to the library's (tests: PIL = the same
// libjpeg-turbo, and the numpy restatement in oracle/jpeg.py):
//
//   jpeg_huffman_kernel   jdhuff.c decode_mcu.  Entropy decoding is sequential inside a restart interval, so ONE LANE decodes one
//                         interval (an image without restart markers is one interval); the batch's images / intervals run as
//                         independent waves (64 frames = 64 waves on 64 CUs).  The wave's other lanes copy the image's Huffman tables
//                         into LDS: a 9-bit look-ahead table resolves a code in one LDS read (jdhuff.c HUFF_LOOKAHEAD), longer codes
//                         walk maxcode[].  The byte stream is read 8 bytes at a time; FF 00 unstuffing and the stop at a marker follow
//                         jpeg_fill_bit_buffer.  Output: quantised coefficients, natural order, int16 [block][64].
//   jpeg_idct_kernel      jidctint.c jpeg_idct_islow (JDCT_ISLOW, the default): dequantisation + 13-bit fixed-point LL&M, eight
//                         threads per block (a column each, then a row each, through LDS).  Output: uint8 component planes.
//   jpeg_color_kernel     jdsample.c h2v1 / h2v2_fancy_upsample (do_fancy_upsampling, the default; edge rows replicated as
//                         jdmainct.c's context rows do) + jdcolor.c ycc_rgb_convert (16-bit fixed point), written as cv2's BGR HWC.
//
// Supported: SOF0, 8 bit, Huffman, one interleaved scan, 1 or 3 components with luma sampling 1x1 / 2x1 / 2x2 and chroma 1x1 (4:4:4,
// 4:2:2, 4:2:0, grey), restart intervals, any size.  All images of a call share size and sampling (frames of one video do).  The
// header segments (DQT / DHT / SOF0 / DRI / SOS) are parsed on the host (nafae_amd/jpeg.py): a few hundred bytes per file.
// Bandwidth note: the pixel path is trivial next to the detector (9.6 MB out per 64 frames); the Huffman stage is latency-bound
// (one dependent LDS look-up per symbol) -- it runs on a side stream under the detector of the previous batch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hip_util.h"
#include "jpeg_core.h"

using namespace nafae;
using namespace nafae_jpeg;

namespace {

// grid = segments (restart intervals); block = one wave.  seg: [image, byte offset of the interval in `stream`, first MCU, MCUs]
__global__ __launch_bounds__(64) void jpeg_huffman_kernel(const unsigned char *__restrict__ stream, const int *__restrict__ desc,
                                                          const int *__restrict__ seg, const int *__restrict__ hufftabs, Geom g,
                                                          short *__restrict__ coef) {
  __shared__ int tabs[6 * HT_INTS];    // [component][DC, AC]
  __shared__ unsigned char nat[64];
  const int lane = threadIdx.x;
  const int *sg = seg + (size_t)blockIdx.x * 4;
  const int img = sg[0];
  const int *d = desc + (size_t)img * DESC_INTS;
  for (int c = 0; c < g.ncomp; c++) {
    const int t = d[6 + c];
    const int *dc = hufftabs + (size_t)(t >> 16) * HT_INTS, *ac = hufftabs + (size_t)(t & 0xffff) * HT_INTS;
    for (int i = lane; i < HT_INTS; i += 64) {
      tabs[(2 * c) * HT_INTS + i] = dc[i];
      tabs[(2 * c + 1) * HT_INTS + i] = ac[i];
    }
  }
  nat[lane] = k_natural[lane];
  __syncthreads();
  if (lane != 0) return;
  const long off = sg[1];
  const long avail = (long)d[0] + d[1] - off;             // bytes from the interval's start to the end of the file
  huffman_interval(stream + off, stream + off + (avail > 0 ? avail : 0), tabs, nat, g, sg[2], sg[2] + sg[3],
                   coef + (size_t)img * g.nblk * 64);
}

// eight threads per 8x8 block; grid covers n_images * g.nblk blocks
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const short *__restrict__ coef, const int *__restrict__ desc,
                                                        const unsigned short *__restrict__ qtabs, Geom g, long nblocks,
                                                        unsigned char *__restrict__ planes) {
  __shared__ int ws[32 * 64];
  const int t = threadIdx.x, j = t & 7, lb = t >> 3;
  const long b = (long)blockIdx.x * 32 + lb;
  const bool ok = b < nblocks;
  int c = 0, bi = 0, img = 0;
  if (ok) {
    img = (int)(b / g.nblk);
    bi = (int)(b - (long)img * g.nblk);
    c = (g.ncomp > 2 && bi >= g.boff[2]) ? 2 : ((g.ncomp > 1 && bi >= g.boff[1]) ? 1 : 0);
    bi -= g.boff[c];
    const unsigned short *q = qtabs + (size_t)desc[(size_t)img * DESC_INTS + 3 + c] * 64;
    const short *cf = coef + (size_t)b * 64;
    int x[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = (int)cf[r * 8 + j] * (int)q[r * 8 + j];      // column j, dequantised
    idct8<13 - 2>(x, o);
#pragma unroll
    for (int r = 0; r < 8; r++) ws[lb * 64 + r * 8 + j] = o[r];
  }
  __syncthreads();
  if (!ok) return;
  int x[8], o[8];
#pragma unroll
  for (int r = 0; r < 8; r++) x[r] = ws[lb * 64 + j * 8 + r];                        // row j
  idct8<13 + 2 + 3>(x, o);
  const int byy = bi / g.bx[c], bxx = bi - byy * g.bx[c];
  unsigned char *dst = planes + (size_t)img * g.psize + g.poff[c] + (size_t)(byy * 8 + j) * g.pw[c] + bxx * 8;
  unsigned lo = 0, hi = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    int a = o[r] + 128, bq = o[r + 4] + 128;
    a = clamp255(a);                                      // range_limit
    bq = clamp255(bq);
    lo |= (unsigned)a << (8 * r);
    hi |= (unsigned)bq << (8 * r);
  }
  *reinterpret_cast<uint2 *>(dst) = make_uint2(lo, hi);
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const unsigned char *__restrict__ planes, Geom g, int n,
                                                         unsigned char *__restrict__ out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long per = (long)g.W * g.H;
  if (idx >= per * n) return;
  const int img = (int)(idx / per);
  const int r = (int)(idx - (long)img * per);
  const int Y = r / g.W, X = r - Y * g.W;
  color_pixel(planes + (size_t)img * g.psize, g, X, Y, out + (size_t)idx * 3);
}

inline bool sampling_ok(int ncomp, int h0, int v0) {
  if (ncomp == 1) return h0 == 1 && v0 == 1;
  return ncomp == 3 && ((h0 == 1 && v0 == 1) || (h0 == 2 && v0 == 1) || (h0 == 2 && v0 == 2));
}

}  // namespace

extern "C" {

int64_t nafae_jpeg_workspace_bytes(int n_images, int W, int H, int ncomp, int h0, int v0) {
  if (n_images <= 0 || W <= 0 || H <= 0 || W > 65535 || H > 65535 || !sampling_ok(ncomp, h0, v0)) return NAFAE_EINVAL;
  const Geom g = make_geom(W, H, ncomp, h0, v0);
  return (int64_t)n_images * g.nblk * 128 + (int64_t)n_images * g.psize + 256;
}

int nafae_jpeg_decode_batch(const uint8_t *stream, int64_t stream_bytes, const int32_t *img_desc, const int32_t *seg_desc,
                            const uint16_t *qtabs, const int32_t *hufftabs, int n_images, int n_segments, int W, int H, int ncomp,
                            int h0, int v0, void *workspace, int64_t workspace_bytes, uint8_t *out_bgr, void *strm) {
  if (!stream || !img_desc || !seg_desc || !qtabs || !hufftabs || !workspace || !out_bgr) return NAFAE_EINVAL;
  if (n_images <= 0 || n_segments < n_images || stream_bytes <= 0) return NAFAE_EINVAL;
  const int64_t need = nafae_jpeg_workspace_bytes(n_images, W, H, ncomp, h0, v0);
  if (need < 0) return (int)need;
  if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(stream) & 7)) return NAFAE_EINVAL;
  const Geom g = make_geom(W, H, ncomp, h0, v0);
  hipStream_t st = as_stream(strm);
  short *coef = reinterpret_cast<short *>(workspace);
  const size_t coef_bytes = (size_t)n_images * g.nblk * 128;
  unsigned char *planes = reinterpret_cast<unsigned char *>(workspace) + ((coef_bytes + 255) & ~(size_t)255);
  if (hipMemsetAsync(coef, 0, coef_bytes, st) != hipSuccess) return NAFAE_ELAUNCH;        // blocks end at EOB: the rest is zero
  hipLaunchKernelGGL(jpeg_huffman_kernel, dim3(n_segments), dim3(64), 0, st, stream, img_desc, seg_desc, hufftabs, g, coef);
  const long nblocks = (long)n_images * g.nblk;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((nblocks + 31) / 32)), dim3(256), 0, st, coef, img_desc, qtabs, g, nblocks, planes);
  const long npix = (long)n_images * W * H;
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, planes, g, n_images, out_bgr);
  return launch_status();
}

}  // extern "C"
