// jpeg.hip -- baseline JPEG decoding on the GPU: what `cv2.imread(img_path)` does per frame in the reference's loader
// (lib/datasets/youcook2.py:212; SURVEY.md section 8(f)2, "replace cv2 JPEG decode with a GPU-side decode").  libjpeg's default
// decompression restated, integer arithmetic throughout, so the frames are bit-identical to the library's (tests: PIL = the same
// libjpeg-turbo, and the numpy restatement in oracle/jpeg.py):
//
//   jpeg_huffman_par_kernel  jdhuff.c decode_mcu by 256 lanes per restart interval (an image without restart markers is one
//                         interval): the stream is compacted into LDS (stuffed zeros dropped, cut at the first marker), every lane
//                         decodes its own stretch of bits from a guessed state and re-decodes it from its left neighbour's exit state
//                         until no exit state changes (self-synchronisation), a prefix sum of the completed MCUs places the lanes, a
//                         last pass writes the coefficients and prefix sums turn the DC differences into values (jpeg_core.h).
//   jpeg_huffman_kernel   the same by ONE lane per interval, for what the kernel above hands over (an interval beyond its LDS budget,
//                         a truncated stream): the wave's other lanes keep a 32 KB ring of the interval's bytes in LDS; a 9-bit
//                         look-ahead table resolves a code in one LDS read (jdhuff.c HUFF_LOOKAHEAD), longer codes walk maxcode[];
//                         FF 00 unstuffing and the stop at a marker follow jpeg_fill_bit_buffer.
//                         Output of both: quantised coefficients, natural order, int16 [block][64].
//   jpeg_idct_kernel      jidctint.c jpeg_idct_islow (JDCT_ISLOW, the default): dequantisation + 13-bit fixed-point LL&M, eight
//                         threads per block (a column each, then a row each, through LDS).  Output: uint8 component planes.
//   jpeg_color_kernel     jdsample.c h2v1 / h2v2_fancy_upsample (do_fancy_upsampling, the default; edge rows replicated as
//                         jdmainct.c's context rows do) + jdcolor.c ycc_rgb_convert (16-bit fixed point), written as cv2's BGR HWC.
//
// Supported: SOF0, 8 bit, Huffman, one interleaved scan, 1 or 3 components with luma sampling 1x1 / 2x1 / 2x2 and chroma 1x1 (4:4:4,
// 4:2:2, 4:2:0, grey), restart intervals, any size.  All images of a call share size and sampling (frames of one video do).  The
// header segments (DQT / DHT / SOF0 / DRI / SOS) are parsed on the host (nafae_amd/jpeg.py): a few hundred bytes per file.
// Cost (64 frames of 224 x 224): entropy decoding 0.83 ms (one lane per file: 13.2 ms), IDCT 9 us, upsampling + colour 21 us; the
// pixel path is trivial next to the detector (9.6 MB out per 64 frames).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hip_util.h"
#include "jpeg_core.h"

using namespace nafae;
using namespace nafae_jpeg;

namespace {

// grid = segments (restart intervals); block = one wave.  seg: [image, byte offset of the interval in `stream`, first MCU, MCUs].
// Lane 0 decodes; ALL lanes keep a 32 KB ring in LDS filled with the interval's bytes (1 KB per refill step, 16 B per lane), so the
// decoder's byte reads cost an LDS access instead of a trip to memory every eight bytes (round 4, first form: 13.2 ms per 64 frames
// of 224 x 224 without restart markers, 2 000 cycles per symbol -- almost all of it the lone lane waiting for its next 8 bytes).
constexpr int JRING = 32768, JMARGIN = 4096;   // refill when fewer than JMARGIN bytes lie ahead of the reader (an MCU takes < 2.5 KB)
__global__ __launch_bounds__(64) void jpeg_huffman_kernel(const unsigned char *__restrict__ stream, long stream_bytes,
                                                          const int *__restrict__ desc, const int *__restrict__ seg,
                                                          const int *__restrict__ hufftabs, Geom g, short *__restrict__ coef,
                                                          const int *__restrict__ redo) {
  __shared__ int tabs[6 * HT_INTS];    // [component][DC, AC]
  __shared__ unsigned char nat[64];
  __shared__ __attribute__((aligned(16))) unsigned char ring[JRING];
  const int lane = threadIdx.x;
  if (redo && redo[blockIdx.x] == 0) return;              // the many-lane kernel has decoded this interval
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
  const long off = sg[1];
  const long avail = (long)d[0] + d[1] - off;             // bytes from the interval's start to the end of the file
  // ring index i <-> stream byte base + i, base = the interval's start rounded down to 16
  const long base = off & ~15L;
  const unsigned p0 = (unsigned)(off - base), lim = p0 + (unsigned)(avail > 0 ? avail : 0);
  unsigned loaded = 0;                                    // bytes in the ring so far (uniform, a multiple of 1024)
  auto refill = [&](unsigned reader) {                   // up to a full ring ahead of the reader's 16-byte line
    const unsigned stop = (reader & ~15u) + JRING;
    while (loaded < lim && loaded + 1024 <= stop) {
      const long a = base + loaded + lane * 16;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a + 16 <= stream_bytes) v = *reinterpret_cast<const uint4 *>(stream + a);
      *reinterpret_cast<uint4 *>(ring + ((loaded + lane * 16) & (JRING - 1))) = v;
      loaded += 1024;
    }
  };
  refill(p0);
  __syncthreads();
  BitReader br;
  br.init(ring, JRING - 1, p0, lim);
  int pred[3] = {0, 0, 0};
  short *cimg = coef + (size_t)img * g.nblk * 64;
  for (int mcu = sg[2]; mcu < sg[2] + sg[3]; mcu++) {
    const unsigned reader = (unsigned)__builtin_amdgcn_readfirstlane((int)br.pos);
    if (loaded < lim && loaded - reader < JMARGIN) refill(reader);
    if (lane == 0) huffman_mcu(br, tabs, nat, g, mcu, pred, cimg);
  }
}

// ---- the many-lane entropy decoder (jpeg_core.h, "parallel entropy decoding"): one workgroup of 256 lanes per restart interval.
//   1. compaction: the interval's bytes, 4 KB per step (16 per lane), lose their stuffed zeros and everything from the first marker
//      on, and land in LDS as big-endian words (a block-wide prefix sum of the bytes each lane keeps gives the positions);
//   2. round 0: lane i decodes the bits [i S, (i + 1) S) from a guessed state; rounds 1 ..: from the exit state of lane i - 1, until
//      no exit state changes (two to four rounds on photographs);
//   3. prefix sum of the MCUs each lane completed -> its absolute MCU; writing pass; DC predictions = prefix sums over the blocks.
// An interval that does not fit (JP_CAP bytes of LDS) or holds fewer MCUs than it should (truncated file: libjpeg feeds zero bits)
// is left to the one-lane kernel through `redo` -- nothing has been written for it at that point.
constexpr int JP_T = 256, JP_CAP = 112 * 1024;
struct ParLds {
  int d32, tabs, nat, scan, st, total;
};
__host__ __device__ inline ParLds par_lds() {
  ParLds o;
  int p = 0;
  o.d32 = p;  p += JP_CAP + 16;
  o.tabs = p; p += 6 * HT_INTS * 4;
  o.nat = p;  p += 64;
  o.scan = p; p += 16 * 4;
  o.st = p;   p += 7 * JP_T * 4;          // exit states (two buffers of bp / (b, k)), MCU counts, ...
  o.total = p;
  return o;
}

// inclusive prefix sum over the workgroup's 256 lanes; *total = the sum.  `tmp`: 8 ints of LDS.  Contains two barriers.
__device__ __forceinline__ int block_scan_incl(int v, int *tmp, int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  __syncthreads();                       // (tmp may still be read from the previous call)
  if (lane == 63) tmp[wave] = x;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < JP_T / 64; w++) {
    const int t = tmp[w];
    if (w < wave) off += t;
    tot += t;
  }
  *total = tot;
  return x + off;
}

__global__ __launch_bounds__(JP_T) void jpeg_huffman_par_kernel(const unsigned char *__restrict__ stream, long stream_bytes,
                                                                const int *__restrict__ desc, const int *__restrict__ seg,
                                                                const int *__restrict__ hufftabs, Geom g, short *__restrict__ coef,
                                                                int *__restrict__ redo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ParLds lo = par_lds();
  uint32_t *d32 = reinterpret_cast<uint32_t *>(smem + lo.d32);
  unsigned char *d8 = smem + lo.d32;
  int *tabs = reinterpret_cast<int *>(smem + lo.tabs);
  unsigned char *nat = smem + lo.nat;
  int *scan = reinterpret_cast<int *>(smem + lo.scan);        // [0..7] wave totals, [8] marker position, [9], [10] changed flags
  unsigned *ex_bp[2] = {reinterpret_cast<unsigned *>(smem + lo.st), reinterpret_cast<unsigned *>(smem + lo.st) + JP_T};
  int *ex_bk[2] = {reinterpret_cast<int *>(smem + lo.st) + 2 * JP_T, reinterpret_cast<int *>(smem + lo.st) + 3 * JP_T};
  const int tid = threadIdx.x;
  const int *sg = seg + (size_t)blockIdx.x * 4;
  const int img = sg[0], m0 = sg[2], nm = sg[3];
  const int *d = desc + (size_t)img * DESC_INTS;
  for (int c = 0; c < g.ncomp; c++) {
    const int t = d[6 + c];
    const int *dc = hufftabs + (size_t)(t >> 16) * HT_INTS, *ac = hufftabs + (size_t)(t & 0xffff) * HT_INTS;
    for (int i = tid; i < HT_INTS; i += JP_T) {
      tabs[(2 * c) * HT_INTS + i] = dc[i];
      tabs[(2 * c + 1) * HT_INTS + i] = ac[i];
    }
  }
  if (tid < 64) nat[tid] = k_natural[tid];
  for (int i = tid; i < (JP_CAP + 16) / 16; i += JP_T) reinterpret_cast<uint4 *>(smem + lo.d32)[i] = make_uint4(0, 0, 0, 0);
  const long off = sg[1];
  long avail = (long)d[0] + d[1] - off;                   // bytes from the interval's start to the end of the file
  avail = avail > 0 ? avail : 0;
  const long base = off & ~15L, end = off + avail;
  // ---- 1. compaction
  unsigned nkept = 0;                                     // compacted bytes so far (uniform)
  bool fits = true;
  for (long c0 = base; c0 < end; c0 += JP_T * 16) {
    if (tid == 0) scan[8] = 0x7fffffff;
    __syncthreads();
    const long a = c0 + tid * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (a + 16 <= stream_bytes) v = *reinterpret_cast<const uint4 *>(stream + a);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned prev = (a > off && a - 1 < stream_bytes) ? stream[a - 1] : 0u;
    const unsigned after = (a + 16 < end && a + 16 < stream_bytes) ? stream[a + 16] : 0xd9u;
    unsigned keep = 0;                                    // bit j: byte j of this lane is a data byte
    int mark = 0x7fffffff;                                // first marker (its FF) among this lane's bytes, as an index into the chunk
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const unsigned b = (w[j >> 2] >> (8 * (j & 3))) & 0xffu;
      const unsigned nx = j < 15 ? (w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xffu : after;
      const long i = a + j;
      if (i >= off && i < end) {
        const unsigned nxe = i + 1 < end ? nx : 0xd9u;
        if (b == 0xffu && nxe != 0) {
          if (mark == 0x7fffffff) mark = tid * 16 + j;
        }
        if (!(b == 0 && prev == 0xffu)) keep |= 1u << j;
      }
      prev = b;
    }
    if (mark != 0x7fffffff) atomicMin(&scan[8], mark);
    __syncthreads();
    const int mk = scan[8];                               // everything from this chunk index on is no data
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (tid * 16 + j >= mk) keep &= ~(1u << j);
    int tot;
    const int incl = block_scan_incl(__builtin_popcount(keep), scan, &tot);
    if (nkept + (unsigned)tot > (unsigned)JP_CAP) {
      fits = false;
      break;
    }
    unsigned o = nkept + (unsigned)(incl - __builtin_popcount(keep));
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (keep & (1u << j)) {
        d8[o ^ 3u] = (unsigned char)((w[j >> 2] >> (8 * (j & 3))) & 0xffu);
        o++;
      }
    nkept += (unsigned)tot;
    if (mk != 0x7fffffff) break;
  }
  __syncthreads();
  if (!fits) {
    if (tid == 0) redo[blockIdx.x] = 1;
    return;
  }
  // ---- 2. decode until the exit states stand still
  const unsigned total_bits = nkept * 8;
  unsigned S = ((total_bits + JP_T - 1) / JP_T + 31) & ~31u;
  S = S < 128 ? 128 : S;
  const unsigned my_end = (unsigned)(tid + 1) * S;
  SpanState in = {(unsigned)tid * S, 0, 0}, out = in;
  int cnt = span_decode<false>(d32, total_bits, my_end, out, tabs, nat, g, 0, 0, nullptr);
  int cur = 0;
  ex_bp[0][tid] = out.bp;
  ex_bk[0][tid] = (out.b << 8) | out.k;
  if (tid < 2) scan[9 + tid] = 0;
  __syncthreads();
  for (int round = 1; round <= JP_T; round++) {
    bool changed = false;
    if (tid > 0) {
      const SpanState p = {ex_bp[cur][tid - 1], ex_bk[cur][tid - 1] >> 8, ex_bk[cur][tid - 1] & 0xff};
      if (!same_state(p, in)) {
        in = p;
        SpanState s2 = p;
        cnt = span_decode<false>(d32, total_bits, my_end, s2, tabs, nat, g, 0, 0, nullptr);
        changed = !same_state(s2, out);
        out = s2;
      }
    }
    ex_bp[cur ^ 1][tid] = out.bp;
    ex_bk[cur ^ 1][tid] = (out.b << 8) | out.k;
    // ("some exit state changed in round r" lives in flag r & 1: the next round writes the other one, and the barrier of that round
    // lies between this round's readers and the writers of round r + 2)
    if (changed) scan[9 + (round & 1)] = round;
    __syncthreads();
    cur ^= 1;
    if (scan[9 + (round & 1)] != round) break;
  }
  // ---- 3. positions, coefficients, DC predictions
  int total_mcu;
  const int start = block_scan_incl(cnt, scan, &total_mcu) - cnt;
  if (total_mcu < nm) {
    if (tid == 0) redo[blockIdx.x] = 1;
    return;
  }
  short *cimg = coef + (size_t)img * g.nblk * 64;
  {
    SpanState s2 = in;
    span_decode<true>(d32, total_bits, my_end, s2, tabs, nat, g, m0 + start, m0 + nm, cimg);
  }
  __syncthreads();
  for (int c = 0; c < g.ncomp; c++) {
    const int per = c == 0 ? g.h0 * g.v0 : 1, nblocks = nm * per;
    int carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += JP_T) {
      const int n = b0 + tid;
      short *blk = n < nblocks ? cimg + dc_block(g, c, m0, n) * 64 : nullptr;
      int tot;
      const int x = block_scan_incl(blk ? (int)blk[0] : 0, scan, &tot);
      if (blk) blk[0] = (short)(carry + x);
      carry += tot;
    }
  }
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
  // coefficients | one flag per restart interval (at most one per MCU) | sample planes
  return (int64_t)n_images * g.nblk * 128 + (int64_t)n_images * g.mx * g.my * 4 + (int64_t)n_images * g.psize + 768;
}

int nafae_jpeg_decode_batch(const uint8_t *stream, int64_t stream_bytes, const int32_t *img_desc, const int32_t *seg_desc,
                            const uint16_t *qtabs, const int32_t *hufftabs, int n_images, int n_segments, int W, int H, int ncomp,
                            int h0, int v0, void *workspace, int64_t workspace_bytes, uint8_t *out_bgr, void *strm) {
  if (!stream || !img_desc || !seg_desc || !qtabs || !hufftabs || !workspace || !out_bgr) return NAFAE_EINVAL;
  if (n_images <= 0 || n_segments < n_images || stream_bytes <= 0) return NAFAE_EINVAL;
  const int64_t need = nafae_jpeg_workspace_bytes(n_images, W, H, ncomp, h0, v0);
  if (need < 0) return (int)need;
  if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(stream) & 15)) return NAFAE_EINVAL;
  const Geom g = make_geom(W, H, ncomp, h0, v0);
  hipStream_t st = as_stream(strm);
  if ((long)n_segments > (long)n_images * g.mx * g.my) return NAFAE_EINVAL;
  short *coef = reinterpret_cast<short *>(workspace);
  const size_t coef_bytes = ((size_t)n_images * g.nblk * 128 + 255) & ~(size_t)255;
  int *redo = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(workspace) + coef_bytes);
  const size_t redo_bytes = ((size_t)n_images * g.mx * g.my * 4 + 255) & ~(size_t)255;
  unsigned char *planes = reinterpret_cast<unsigned char *>(workspace) + coef_bytes + redo_bytes;
  // (blocks end at EOB: the rest is zero; flags: 0 = decoded by the many-lane kernel)
  if (hipMemsetAsync(coef, 0, coef_bytes + redo_bytes, st) != hipSuccess) return NAFAE_ELAUNCH;
  const char *pe = nafae::experiment_env("NAFAE_JPEG_PAR");       // =0 (experiments build): the one-lane decoder for everything
  const bool par = !(pe && pe[0] == '0');
  if (par) {
    const int lds = par_lds().total;
    if (nafae::allow_dynamic_lds(reinterpret_cast<const void *>(jpeg_huffman_par_kernel), lds) != NAFAE_OK) return NAFAE_ELAUNCH;
    hipLaunchKernelGGL(jpeg_huffman_par_kernel, dim3(n_segments), dim3(JP_T), lds, st, stream, (long)stream_bytes, img_desc, seg_desc,
                       hufftabs, g, coef, redo);
  }
  hipLaunchKernelGGL(jpeg_huffman_kernel, dim3(n_segments), dim3(64), 0, st, stream, (long)stream_bytes, img_desc, seg_desc, hufftabs, g, coef,
                     par ? redo : (const int *)nullptr);
  const long nblocks = (long)n_images * g.nblk;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((nblocks + 31) / 32)), dim3(256), 0, st, coef, img_desc, qtabs, g, nblocks, planes);
  const long npix = (long)n_images * W * H;
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, planes, g, n_images, out_bgr);
  return launch_status();
}

}  // extern "C"
