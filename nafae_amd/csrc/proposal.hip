// proposal.hip -- RPN proposal path and ROI-Align for gfx950 (device-resident, batched over frames).
//
// This translation unit is compiled with -ffp-contract=off: every +,-,*,/ below is a single IEEE-754
// operation in the order written, so that the integer results (sort order, NMS keep lists) are bit-identical
// to the CPU oracle (oracle/native.c, oracle/detector.py) on identical inputs.
//
//   rpn_decode_kernel   rpn/rpn.py:67-69 + proposal_layer.py:67-109 + bbox_transform.py:77-103,125-133
//   sort_kernel         proposal_layer.py:125   (one workgroup per frame, bitonic network in LDS)
//   nms_kernel          nms_cuda_kernel.cu:31-161 + proposal_layer.py:150-163 (one wavefront per frame)
//   roi_align kernels   roi_align_kernel.cu:15-70 (+ modules/roi_align.py:26-29 fused 2x2 mean)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nafae_hip.h"
#include "hip_util.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }
// launch failures (bad configuration, missing code object, wrong runtime) must be loud, never silent
inline int launched() { return hipGetLastError() == hipSuccess ? NAFAE_OK : NAFAE_ELAUNCH; }

// ------------------------------------------------------------------------------------------------ decode
__global__ __launch_bounds__(256) void rpn_decode_kernel(const float *__restrict__ head,
                                                         const float *__restrict__ anchors,
                                                         const float *__restrict__ im_info,
                                                         float *__restrict__ scores, float *__restrict__ boxes, int F,
                                                         int H, int W, int A, int feat_stride) {
  const long total = (long)F * H * W * A;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int a = i % A;
  const long pix = i / A;  // f*H*W + h*W + w
  const int w = pix % W;
  const int h = (pix / W) % H;
  const int f = pix / ((long)W * H);
  const float *hp = head + pix * (6L * A);
  // 2-way softmax over (bg = channel a, fg = channel A + a): rpn/rpn.py:67-69
  const float s0 = hp[a], s1 = hp[A + a];
  const float m = fmaxf(s0, s1);
  const float e0 = expf(s0 - m), e1 = expf(s1 - m);
  scores[i] = e1 / (e0 + e1);
  // anchor = base + shift (proposal_layer.py:79-93), fp32
  const float sx = (float)(w * feat_stride), sy = (float)(h * feat_stride);
  const float ax1 = anchors[a * 4 + 0] + sx, ay1 = anchors[a * 4 + 1] + sy;
  const float ax2 = anchors[a * 4 + 2] + sx, ay2 = anchors[a * 4 + 3] + sy;
  const f32x4 d = *reinterpret_cast<const f32x4 *>(hp + 2 * A + 4 * a);
  // bbox_transform_inv (bbox_transform.py:77-103)
  const float widths = ax2 - ax1 + 1.0f, heights = ay2 - ay1 + 1.0f;
  const float ctr_x = ax1 + 0.5f * widths, ctr_y = ay1 + 0.5f * heights;
  const float pcx = d[0] * widths + ctr_x, pcy = d[1] * heights + ctr_y;
  const float pw = expf(d[2]) * widths, ph = expf(d[3]) * heights;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
  // clip_boxes (bbox_transform.py:125-133)
  const float xmax = im_info[f * 3 + 1] - 1.0f, ymax = im_info[f * 3 + 0] - 1.0f;
  x1 = fminf(fmaxf(x1, 0.f), xmax);
  y1 = fminf(fmaxf(y1, 0.f), ymax);
  x2 = fminf(fmaxf(x2, 0.f), xmax);
  y2 = fminf(fmaxf(y2, 0.f), ymax);
  f32x4 o = {x1, y1, x2, y2};
  *reinterpret_cast<f32x4 *>(boxes + i * 4) = o;
}

// ------------------------------------------------------------------------------------------------ sort
// key = (descending-orderable score bits << 32) | index : ascending u64 order == descending score,
// ties by ascending index (torch's stable CPU sort, which the oracle pins).
__global__ __launch_bounds__(1024) void sort_kernel(const float *__restrict__ scores, int32_t *__restrict__ order,
                                                    int n, int P) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const int f = blockIdx.x;
  const float *s = scores + (long)f * n;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    unsigned long long k = ~0ull;
    if (i < n) {
      uint32_t b = __float_as_uint(s[i]);
      uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending-orderable
      k = ((unsigned long long)(~asc) << 32) | (uint32_t)i;
    }
    keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool up = (i & k) == 0;
          unsigned long long a = keys[i], b = keys[ixj];
          if ((a > b) == up) {
            keys[i] = b;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) order[(long)f * n + i] = (int32_t)(keys[i] & 0xffffffffu);
}

// ------------------------------------------------------------------------------------------------ NMS
// devIoU (nms_cuda_kernel.cu:31-39), same operation order; a = earlier (kept) box, b = candidate.
__device__ __forceinline__ float dev_iou(const f32x4 a, const f32x4 b) {
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
  const float interS = width * height;
  const float Sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
  const float Sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
  return interS / (Sa + Sb - interS);
}

// One wavefront per frame.  Candidates are taken 64 at a time in score order:
//   phase 1: every lane tests its candidate against all boxes kept so far (LDS broadcast reads);
//   phase 2: survivors of the block are resolved in order with ballots / shuffles.
// This visits exactly the pairs the greedy definition needs (kept x later) instead of the reference's full
// n x n bitmask (nms_cuda_kernel.cu:41-85) followed by a host sweep (:123-144), and stops as soon as
// `topN` boxes are kept (proposal_layer.py:153-154 only ever uses the first post_nms_topN).
__global__ __launch_bounds__(64) void nms_kernel(const float *__restrict__ boxes, int box_stride,
                                                 const float *__restrict__ scores, const int32_t *__restrict__ order,
                                                 int n, int n_sorted, float thresh, int topN,
                                                 int32_t *__restrict__ keep_out, int32_t *__restrict__ num_out,
                                                 float *__restrict__ rois, float *__restrict__ roi_scores) {
  extern __shared__ __attribute__((aligned(16))) f32x4 kept[];
  const int f = blockIdx.x;
  const int lane = threadIdx.x;
  const float *fb = boxes + (long)f * n * box_stride;
  int nk = 0;
  bool done = false;
  for (int base = 0; base < n_sorted && !done; base += 64) {
    const int c = base + lane;
    const bool valid = c < n_sorted;
    f32x4 box = {0.f, 0.f, 0.f, 0.f};
    float score = 0.f;
    if (valid) {
      const int src = order ? order[(long)f * n + c] : c;
      const float *p = fb + (long)src * box_stride;
      box[0] = p[0];
      box[1] = p[1];
      box[2] = p[2];
      box[3] = p[3];
      if (scores) score = scores[(long)f * n + src];
    }
    bool sup = !valid;
    for (int j = 0; j < nk; j++) {
      const f32x4 kb = kept[j];
      if (dev_iou(kb, box) > thresh) sup = true;
    }
    unsigned long long alive = __ballot(!sup);
    while (alive) {
      const int i = __ffsll((long long)alive) - 1;
      f32x4 bi;
      bi[0] = __shfl(box[0], i);
      bi[1] = __shfl(box[1], i);
      bi[2] = __shfl(box[2], i);
      bi[3] = __shfl(box[3], i);
      const float si = __shfl(score, i);
      if (lane == 0) {
        kept[nk] = bi;
        if (keep_out) keep_out[(long)f * n + nk] = base + i;
        if (rois) {
          float *r = rois + ((long)f * topN + nk) * 5;
          r[0] = (float)f;
          r[1] = bi[0];
          r[2] = bi[1];
          r[3] = bi[2];
          r[4] = bi[3];
          roi_scores[(long)f * topN + nk] = si;
        }
      }
      nk++;
      if (nk >= topN) {
        done = true;
        break;
      }
      const bool s = (lane > i) && !sup && (dev_iou(bi, box) > thresh);
      sup = sup || s;
      alive = __ballot(!sup) & ~((2ull << i) - 1ull);
    }
    __syncthreads();  // single-wave workgroup: orders the lane-0 LDS writes before the next block's reads
  }
  if (lane == 0) num_out[f] = nk;
  if (keep_out)
    for (int k = nk + lane; k < n; k += 64) keep_out[(long)f * n + k] = 0;
  if (rois)
    for (int k = nk + lane; k < topN; k += 64) {
      float *r = rois + ((long)f * topN + k) * 5;
      r[0] = (float)f;  // proposal_layer.py:160: column 0 is set for every row, padded ones included
      r[1] = r[2] = r[3] = r[4] = 0.f;
      roi_scores[(long)f * topN + k] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ ROI-Align
struct RoiGeom {
  float start_w, start_h, bin_w, bin_h;
  int img;
};

// roi_align_kernel.cu:33-43, with the reference's float/double mixing (literals `1.` are double there).
__device__ __forceinline__ RoiGeom roi_geom(const float *r, float scale, int AH, int AW) {
  RoiGeom g;
  g.img = (int)r[0];
  g.start_w = r[1] * scale;
  g.start_h = r[2] * scale;
  const float end_w = r[3] * scale, end_h = r[4] * scale;
  const float roi_w = fmaxf((float)((double)(end_w - g.start_w) + 1.), 0.f);
  const float roi_h = fmaxf((float)((double)(end_h - g.start_h) + 1.), 0.f);
  g.bin_h = (float)((double)roi_h / ((double)AH - 1.));
  g.bin_w = (float)((double)roi_w / ((double)AW - 1.));
  return g;
}

__device__ __forceinline__ float bilerp(float ul, float ur, float dl, float dr, float hr, float wr) {
  // roi_align_kernel.cu:64-67: evaluated in double, rounded once to float
  const double h1 = 1. - (double)hr, w1 = 1. - (double)wr;
  double v = (double)ul * h1 * w1 + (double)ur * h1 * (double)wr + (double)dl * (double)hr * w1 +
             (double)dr * (double)hr * (double)wr;
  return (float)v;
}

// Drop-in for ROIAlignForward: NCHW features, [N,C,AH,AW] output, one thread per output element.
__global__ __launch_bounds__(256) void roi_align_nchw_kernel(long nthreads, const float *__restrict__ bottom,
                                                             float scale, int H, int W, int C, int AH, int AW,
                                                             const float *__restrict__ rois, float *__restrict__ top) {
  const long index = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= nthreads) return;
  const int pw = index % AW;
  const int ph = (index / AW) % AH;
  const int c = (index / AW / AH) % C;
  const int n = index / AW / AH / C;
  const RoiGeom g = roi_geom(rois + (long)n * 5, scale, AH, AW);
  const float h = (float)ph * g.bin_h + g.start_h;
  const float w = (float)pw * g.bin_w + g.start_w;
  const int hs = (int)fminf(floorf(h), (float)(H - 2));
  const int ws = (int)fminf(floorf(w), (float)(W - 2));
  float out = 0.f;
  if (!(h < 0 || h >= H || w < 0 || w >= W)) {
    const float hr = h - (float)hs, wr = w - (float)ws;
    const float *p = bottom + (((long)g.img * C + c) * H + hs) * W + ws;
    out = bilerp(p[0], p[1], p[W], p[W + 1], hr, wr);
  }
  top[index] = out;
}

// Drop-in for ROIAlignBackward (roi_align_kernel.cu:93-141): one thread per top element scatters its gradient to the four
// taps with hardware fp32 atomic adds (the reference uses atomicAdd too, so the summation order is unspecified on both
// sides).  Float/double mixing as written there: upper taps in double rounded once, lower taps in float.
__global__ __launch_bounds__(256) void roi_align_bwd_nchw_kernel(long nthreads, const float *__restrict__ top_diff,
                                                                 float scale, int H, int W, int C, int AH, int AW,
                                                                 const float *__restrict__ rois, float *bottom_diff) {
  const long index = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= nthreads) return;
  const int pw = index % AW;
  const int ph = (index / AW) % AH;
  const int c = (index / AW / AH) % C;
  const int n = index / AW / AH / C;
  const RoiGeom g = roi_geom(rois + (long)n * 5, scale, AH, AW);
  const float h = (float)ph * g.bin_h + g.start_h;
  const float w = (float)pw * g.bin_w + g.start_w;
  if (h < 0 || h >= H || w < 0 || w >= W) return;
  const int hs = (int)fminf(floorf(h), (float)(H - 2));
  const int ws = (int)fminf(floorf(w), (float)(W - 2));
  const float hr = h - (float)hs, wr = w - (float)ws;
  const float td = top_diff[index];
  float *p = bottom_diff + (((long)g.img * C + c) * H + hs) * W + ws;
  unsafeAtomicAdd(p, (float)((double)td * (1. - (double)hr) * (double)(1.f - wr)));
  unsafeAtomicAdd(p + 1, (float)((double)td * (1. - (double)hr) * (double)wr));
  unsafeAtomicAdd(p + W, (td * hr) * (1.f - wr));
  unsafeAtomicAdd(p + W + 1, (td * hr) * wr);
}

// Fused RoIAlignAvg on NHWC: one workgroup per ROI, a thread owns 2 adjacent channels and walks the 8x8 sample
// grid row by row keeping the previous sample row in registers, so each 7x7 output bin is produced from
// registers and written once, channel-contiguous (coalesced 512 B per wave).
constexpr int PS = 7;      // pooled size
constexpr int AS = PS + 1;  // aligned (sample) grid

__global__ __launch_bounds__(256) void roi_align_avg_nhwc_kernel(const float *__restrict__ feat, int H, int W, int C,
                                                                 const float *__restrict__ rois, float scale,
                                                                 float *__restrict__ out) {
  const int n = blockIdx.x;
  const RoiGeom g = roi_geom(rois + (long)n * 5, scale, AS, AS);
  // per-ROI sample geometry, computed once by 8 lanes and broadcast through LDS (keeps it out of SGPRs)
  __shared__ int s_hs[AS], s_ws[AS], s_hv[AS], s_wv[AS];
  __shared__ float s_hr[AS], s_wr[AS];
  if (threadIdx.x < AS) {
    const int p = threadIdx.x;
    const float h = (float)p * g.bin_h + g.start_h;
    const float w = (float)p * g.bin_w + g.start_w;
    const int hsp = (int)fminf(floorf(h), (float)(H - 2));
    const int wsp = (int)fminf(floorf(w), (float)(W - 2));
    s_hs[p] = hsp;
    s_ws[p] = wsp;
    s_hv[p] = !(h < 0 || h >= H);
    s_wv[p] = !(w < 0 || w >= W);
    s_hr[p] = h - (float)hsp;
    s_wr[p] = w - (float)wsp;
  }
  __syncthreads();
  const float *fimg = feat + (long)g.img * H * W * C;
  float *o = out + (long)n * PS * PS * C;
  // four adjacent channels per thread where C allows (16-byte loads and stores: half the instructions of the 8-byte form through
  // the texture path, which is what bounds this kernel -- 256 tap loads per thread); the arithmetic per channel is unchanged
  if ((C & 3) == 0) {
    for (int c = threadIdx.x * 4; c < C; c += 4 * blockDim.x) {
      f32x4 prev[AS], cur[AS];
#pragma unroll
      for (int ph = 0; ph < AS; ph++) {
#pragma unroll
        for (int pw = 0; pw < AS; pw++) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (s_hv[ph] && s_wv[pw]) {
            const float *p = fimg + ((long)s_hs[ph] * W + s_ws[pw]) * C + c;
            const f32x4 ul = *reinterpret_cast<const f32x4 *>(p);
            const f32x4 ur = *reinterpret_cast<const f32x4 *>(p + C);
            const f32x4 dl = *reinterpret_cast<const f32x4 *>(p + (long)W * C);
            const f32x4 dr = *reinterpret_cast<const f32x4 *>(p + (long)W * C + C);
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = bilerp(ul[q], ur[q], dl[q], dr[q], s_hr[ph], s_wr[pw]);
          }
          cur[pw] = v;
        }
        if (ph > 0) {
#pragma unroll
          for (int pw = 0; pw < PS; pw++) {
            f32x4 s4;
#pragma unroll
            for (int q = 0; q < 4; q++) s4[q] = (((prev[pw][q] + prev[pw + 1][q]) + cur[pw][q]) + cur[pw + 1][q]) / 4.0f;
            *reinterpret_cast<f32x4 *>(o + ((ph - 1) * PS + pw) * C + c) = s4;
          }
        }
#pragma unroll
        for (int pw = 0; pw < AS; pw++) prev[pw] = cur[pw];
      }
    }
    return;
  }
  for (int c = threadIdx.x * 2; c < C; c += 2 * blockDim.x) {
    f32x2 prev[AS], cur[AS];
#pragma unroll
    for (int ph = 0; ph < AS; ph++) {
#pragma unroll
      for (int pw = 0; pw < AS; pw++) {
        f32x2 v = {0.f, 0.f};
        if (s_hv[ph] && s_wv[pw]) {
          const float *p = fimg + ((long)s_hs[ph] * W + s_ws[pw]) * C + c;
          const f32x2 ul = *reinterpret_cast<const f32x2 *>(p);
          const f32x2 ur = *reinterpret_cast<const f32x2 *>(p + C);
          const f32x2 dl = *reinterpret_cast<const f32x2 *>(p + (long)W * C);
          const f32x2 dr = *reinterpret_cast<const f32x2 *>(p + (long)W * C + C);
          v[0] = bilerp(ul[0], ur[0], dl[0], dr[0], s_hr[ph], s_wr[pw]);
          v[1] = bilerp(ul[1], ur[1], dl[1], dr[1], s_hr[ph], s_wr[pw]);
        }
        cur[pw] = v;
      }
      if (ph > 0) {
#pragma unroll
        for (int pw = 0; pw < PS; pw++) {
          // avg_pool2d(k=2, s=1): ((t[y][x] + t[y][x+1]) + t[y+1][x]) + t[y+1][x+1], then / 4
          f32x2 s;
          s[0] = (((prev[pw][0] + prev[pw + 1][0]) + cur[pw][0]) + cur[pw + 1][0]) / 4.0f;
          s[1] = (((prev[pw][1] + prev[pw + 1][1]) + cur[pw][1]) + cur[pw + 1][1]) / 4.0f;
          *reinterpret_cast<f32x2 *>(o + ((ph - 1) * PS + pw) * C + c) = s;
        }
      }
#pragma unroll
      for (int pw = 0; pw < AS; pw++) prev[pw] = cur[pw];
    }
  }
}

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

// Same fused RoIAlignAvg on bf16 planes (hi [, lo]): feature value = hi + lo (split-bf16), output re-split.
// 128 threads per ROI, 4 adjacent channels per thread (8-byte plane loads / stores).
// Planes may be separate tensors or interleaved per 32 channels (lo == hi + 32 elements: "I32", see bf16_tile.h).
__global__ __launch_bounds__(128) void roi_align_avg_nhwc_bf16_kernel(const __bf16 *fhi, const __bf16 *flo, int H, int W, int C,
                                                                      const float *__restrict__ rois, float scale, __bf16 *ohi,
                                                                      __bf16 *olo, float *__restrict__ of32) {
  const int n = blockIdx.x;
  const bool il = flo != nullptr && reinterpret_cast<const char *>(flo) == reinterpret_cast<const char *>(fhi) + 64;
  const int CS = il ? 2 * C : C;  // elements per pixel row behind one plane pointer
  const RoiGeom g = roi_geom(rois + (long)n * 5, scale, AS, AS);
  __shared__ int s_hs[AS], s_ws[AS], s_hv[AS], s_wv[AS];
  __shared__ float s_hr[AS], s_wr[AS];
  if (threadIdx.x < AS) {
    const int p = threadIdx.x;
    const float h = (float)p * g.bin_h + g.start_h;
    const float w = (float)p * g.bin_w + g.start_w;
    const int hsp = (int)fminf(floorf(h), (float)(H - 2));
    const int wsp = (int)fminf(floorf(w), (float)(W - 2));
    s_hs[p] = hsp;
    s_ws[p] = wsp;
    s_hv[p] = !(h < 0 || h >= H);
    s_wv[p] = !(w < 0 || w >= W);
    s_hr[p] = h - (float)hsp;
    s_wr[p] = w - (float)wsp;
  }
  __syncthreads();
  const long img = (long)g.img * H * W * CS;
  const long ob = (long)n * PS * PS * CS;
  auto ld4 = [&](long off) -> f32x4 {
    const bf16x4_t h = *reinterpret_cast<const bf16x4_t *>(fhi + off);
    f32x4 v = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    if (flo) {
      const bf16x4_t l = *reinterpret_cast<const bf16x4_t *>(flo + off);
      v[0] += (float)l[0];
      v[1] += (float)l[1];
      v[2] += (float)l[2];
      v[3] += (float)l[3];
    }
    return v;
  };
  for (int c = threadIdx.x * 4; c < C; c += 512) {
    f32x4 prev[AS], cur[AS];
#pragma unroll
    for (int ph = 0; ph < AS; ph++) {
#pragma unroll
      for (int pw = 0; pw < AS; pw++) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (s_hv[ph] && s_wv[pw]) {
          const int co = il ? ((c >> 5) << 6) + (c & 31) : c;   // 4 channels c..c+3 stay inside one 32-channel piece
          const long p = img + ((long)s_hs[ph] * W + s_ws[pw]) * CS + co;
          const f32x4 ul = ld4(p), ur = ld4(p + CS), dl = ld4(p + (long)W * CS), dr = ld4(p + (long)W * CS + CS);
#pragma unroll
          for (int q = 0; q < 4; q++) v[q] = bilerp(ul[q], ur[q], dl[q], dr[q], s_hr[ph], s_wr[pw]);
        }
        cur[pw] = v;
      }
      if (ph > 0) {
#pragma unroll
        for (int pw = 0; pw < PS; pw++) {
          bf16x4_t hv, lv;
          f32x4 fv;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float sv = (((prev[pw][q] + prev[pw + 1][q]) + cur[pw][q]) + cur[pw + 1][q]) / 4.0f;
            const __bf16 hq = (__bf16)sv;
            hv[q] = hq;
            lv[q] = (__bf16)(sv - (float)hq);
            fv[q] = olo ? (float)hq + (float)lv[q] : (float)hq;   // exactly what merging the planes would give
          }
          if (of32) *reinterpret_cast<f32x4 *>(of32 + (long)n * PS * PS * C + ((ph - 1) * PS + pw) * (long)C + c) = fv;
          const long o = ob + ((ph - 1) * PS + pw) * (long)CS + (il ? ((c >> 5) << 6) + (c & 31) : c);
          *reinterpret_cast<bf16x4_t *>(ohi + o) = hv;
          if (olo) *reinterpret_cast<bf16x4_t *>(olo + o) = lv;
        }
      }
#pragma unroll
      for (int pw = 0; pw < AS; pw++) prev[pw] = cur[pw];
    }
  }
}

// RoIAlignAvg from an fp32 NHWC feature map straight to the bf16 planes fc6 consumes (and, optionally, the fp32
// `pooled_feat`).  The detector merges the conv5_3 planes to fp32 once per step (25.7 MB at C2) instead of once per tap
// here (4.3 G tap reads), and the bilinear weights of a sample -- the reference's double-precision products
// (roi_align_kernel.cu:64-67), rounded once -- are shared by all channels, so a sample costs 4 fp32 FMAs per channel.
// The result differs from the exact-fp32 kernel above by the rounding of the weights (~1e-7), far inside the 2^-17 of
// the planes it is stored in; the exact kernel stays the one behind precision = 'f32'.
__global__ __launch_bounds__(128) void roi_align_avg_to_planes_kernel(const float *__restrict__ feat, int H, int W, int C,
                                                                      const float *__restrict__ rois, float scale, __bf16 *ohi,
                                                                      __bf16 *olo, float *__restrict__ of32) {
  const int n = blockIdx.x;
  const bool il = olo != nullptr && reinterpret_cast<const char *>(olo) == reinterpret_cast<const char *>(ohi) + 64;
  const int CS = il ? 2 * C : C;  // elements per output pixel row behind one plane pointer
  const RoiGeom g = roi_geom(rois + (long)n * 5, scale, AS, AS);
  __shared__ int s_off[AS * AS];           // element offset of the up-left tap, or -1: sample outside the map
  __shared__ f32x4 s_wt[AS * AS];          // (1-hr)(1-wr), (1-hr)wr, hr(1-wr), hr wr
  if (threadIdx.x < AS * AS) {
    const int ph = threadIdx.x / AS, pw = threadIdx.x - ph * AS;
    const float h = (float)ph * g.bin_h + g.start_h;
    const float w = (float)pw * g.bin_w + g.start_w;
    const int hs = (int)fminf(floorf(h), (float)(H - 2));
    const int ws = (int)fminf(floorf(w), (float)(W - 2));
    const bool ok = !(h < 0 || h >= H) && !(w < 0 || w >= W);
    const double hr = (double)(h - (float)hs), wr = (double)(w - (float)ws);
    s_off[threadIdx.x] = ok ? (hs * W + ws) * C : -1;
    s_wt[threadIdx.x] = f32x4{(float)((1. - hr) * (1. - wr)), (float)((1. - hr) * wr), (float)(hr * (1. - wr)), (float)(hr * wr)};
  }
  __syncthreads();
  const float *img = feat + (long)g.img * H * W * C;
  const long ob = (long)n * PS * PS * CS;
  const long rs = (long)W * C;
  for (int c = threadIdx.x * 4; c < C; c += 512) {
    f32x4 prev[AS], cur[AS];
#pragma unroll
    for (int ph = 0; ph < AS; ph++) {
#pragma unroll
      for (int pw = 0; pw < AS; pw++) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int off = s_off[ph * AS + pw];
        if (off >= 0) {
          const float *p = img + off + c;
          const f32x4 wt = s_wt[ph * AS + pw];
          const f32x4 ul = *reinterpret_cast<const f32x4 *>(p), ur = *reinterpret_cast<const f32x4 *>(p + C);
          const f32x4 dl = *reinterpret_cast<const f32x4 *>(p + rs), dr = *reinterpret_cast<const f32x4 *>(p + rs + C);
          v = ul * wt[0] + ur * wt[1] + dl * wt[2] + dr * wt[3];
        }
        cur[pw] = v;
      }
      if (ph > 0) {
#pragma unroll
        for (int pw = 0; pw < PS; pw++) {
          bf16x4_t hv, lv;
          f32x4 fv;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float sv = (((prev[pw][q] + prev[pw + 1][q]) + cur[pw][q]) + cur[pw + 1][q]) / 4.0f;
            const __bf16 hq = (__bf16)sv;
            hv[q] = hq;
            lv[q] = (__bf16)(sv - (float)hq);
            fv[q] = olo ? (float)hq + (float)lv[q] : (float)hq;   // exactly what merging the planes would give
          }
          if (of32) *reinterpret_cast<f32x4 *>(of32 + (long)n * PS * PS * C + ((ph - 1) * PS + pw) * (long)C + c) = fv;
          const long o = ob + ((ph - 1) * PS + pw) * (long)CS + (il ? ((c >> 5) << 6) + (c & 31) : c);
          *reinterpret_cast<bf16x4_t *>(ohi + o) = hv;
          if (olo) *reinterpret_cast<bf16x4_t *>(olo + o) = lv;
        }
      }
#pragma unroll
      for (int pw = 0; pw < AS; pw++) prev[pw] = cur[pw];
    }
  }
}

}  // namespace

extern "C" {

int nafae_rpn_decode(const float *head, const float *anchors, const float *im_info, float *scores, float *boxes,
                     int F, int H, int W, int A, int feat_stride, void *stream) {
  if (!head || !anchors || !im_info || !scores || !boxes || F <= 0 || H <= 0 || W <= 0 || A <= 0) return NAFAE_EINVAL;
  if (A & 1) return NAFAE_EINVAL;  // 16-byte aligned delta quads need 2A % 4 == 0
  const long total = (long)F * H * W * A;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), head, anchors,
                     im_info, scores, boxes, F, H, W, A, feat_stride);
  return launched();
}

int nafae_sort_desc(const float *scores, int32_t *order, int F, int n, void *stream) {
  if (!scores || !order || F <= 0 || n <= 0) return NAFAE_EINVAL;
  if (n > 16384) return NAFAE_ELIMIT;
  int P = 64;
  while (P < n) P <<= 1;
  const size_t lds = (size_t)P * sizeof(unsigned long long);
  if (lds > 64 * 1024) {
    if (nafae::allow_dynamic_lds(reinterpret_cast<const void *>(sort_kernel), 128 * 1024) != NAFAE_OK) return NAFAE_ELAUNCH;
  }
  hipLaunchKernelGGL(sort_kernel, dim3(F), dim3(P < 1024 ? P : 1024), lds, S(stream), scores, order, n, P);
  return launched();
}

static int launch_nms(const float *boxes, int box_stride, const float *scores, const int32_t *order, int F, int n,
                      int n_sorted, float thresh, int topN, int32_t *keep_out, int32_t *num_out, float *rois,
                      float *roi_scores, hipStream_t st) {
  if (topN > 8192) return NAFAE_ELIMIT;
  const size_t lds = (size_t)topN * sizeof(f32x4);
  if (lds > 64 * 1024) {
    if (nafae::allow_dynamic_lds(reinterpret_cast<const void *>(nms_kernel), 128 * 1024) != NAFAE_OK) return NAFAE_ELAUNCH;
  }
  hipLaunchKernelGGL(nms_kernel, dim3(F), dim3(64), lds, st, boxes, box_stride, scores, order, n, n_sorted, thresh, topN,
                     keep_out, num_out, rois, roi_scores);
  return launched();
}

int nafae_nms(int32_t *keep_out, int32_t *num_out, const float *boxes, int n, int dim, float thresh, void *stream) {
  if (!keep_out || !num_out || !boxes || n <= 0 || dim < 4) return NAFAE_EINVAL;
  return launch_nms(boxes, dim, nullptr, nullptr, 1, n, n, thresh, n, keep_out, num_out, nullptr, nullptr, S(stream));
}

int nafae_proposals(const float *boxes, const float *scores, const int32_t *order, int F, int n, int n_sorted,
                    float nms_thresh, int post_nms_topN, float *rois, float *roi_scores, int32_t *n_keep,
                    void *stream) {
  if (!boxes || !scores || !order || !rois || !roi_scores || !n_keep) return NAFAE_EINVAL;
  if (F <= 0 || n <= 0 || n_sorted <= 0 || n_sorted > n || post_nms_topN <= 0) return NAFAE_EINVAL;
  return launch_nms(boxes, 4, scores, order, F, n, n_sorted, nms_thresh, post_nms_topN, nullptr, n_keep, rois,
                    roi_scores, S(stream));
}

int nafae_roi_align_forward(int AH, int AW, float scale, const float *features, int B, int C, int H, int W,
                            const float *rois, int N, float *output, void *stream) {
  if (!features || !rois || !output || AH < 2 || AW < 2 || B <= 0 || C <= 0 || H < 2 || W < 2 || N <= 0)
    return NAFAE_EINVAL;
  const long total = (long)N * C * AH * AW;
  hipLaunchKernelGGL(roi_align_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), total,
                     features, scale, H, W, C, AH, AW, rois, output);
  return launched();
}

int nafae_roi_align_backward(int AH, int AW, float scale, const float *top_grad, const float *rois, int N,
                             float *bottom_grad, int B, int C, int H, int W, void *stream) {
  if (!top_grad || !rois || !bottom_grad || AH < 2 || AW < 2 || B <= 0 || C <= 0 || H < 2 || W < 2 || N <= 0)
    return NAFAE_EINVAL;
  const long total = (long)N * C * AH * AW;
  if ((total + 255) / 256 > 0x7fffffffL) return NAFAE_ELIMIT;
  hipLaunchKernelGGL(roi_align_bwd_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), total,
                     top_grad, scale, H, W, C, AH, AW, rois, bottom_grad);
  return launched();
}

int nafae_roi_align_avg_nhwc(const float *feat, int F, int H, int W, int C, const float *rois, int N, float scale,
                             float *out, void *stream) {
  if (!feat || !rois || !out || F <= 0 || H < 2 || W < 2 || C <= 0 || N <= 0) return NAFAE_EINVAL;
  if (C & 1) return NAFAE_EINVAL;
  const int nt = (C & 3) == 0 && C <= 512 ? 128 : 256;   // one pass over the channels with 4 per thread when they fit
  hipLaunchKernelGGL(roi_align_avg_nhwc_kernel, dim3(N), dim3(nt), 0, S(stream), feat, H, W, C, rois, scale, out);
  return launched();
}

int nafae_roi_align_avg_nhwc_bf16(const void *feat_hi, const void *feat_lo, int F, int H, int W, int C, const float *rois,
                                  int N, float spatial_scale, void *out_hi, void *out_lo, float *out_f32, void *stream) {
  if (!feat_hi || !rois || !out_hi || F <= 0 || H < 2 || W < 2 || C <= 0 || N <= 0) return NAFAE_EINVAL;
  if (C & 3) return NAFAE_EINVAL;
  if ((feat_lo == nullptr) != (out_lo == nullptr)) return NAFAE_EINVAL;
  hipLaunchKernelGGL(roi_align_avg_nhwc_bf16_kernel, dim3(N), dim3(128), 0, S(stream), (const __bf16 *)feat_hi,
                     (const __bf16 *)feat_lo, H, W, C, rois, spatial_scale, (__bf16 *)out_hi, (__bf16 *)out_lo, out_f32);
  return launched();
}

int nafae_roi_align_avg_nhwc_to_planes(const float *feat, int F, int H, int W, int C, const float *rois, int N,
                                       float spatial_scale, void *out_hi, void *out_lo, float *out_f32, void *stream) {
  if (!feat || !rois || !out_hi || F <= 0 || H < 2 || W < 2 || C <= 0 || N <= 0) return NAFAE_EINVAL;
  if (C & 3) return NAFAE_EINVAL;
  if ((long)H * W * C >= (1L << 31)) return NAFAE_ELIMIT;
  hipLaunchKernelGGL(roi_align_avg_to_planes_kernel, dim3(N), dim3(128), 0, S(stream), feat, H, W, C, rois, spatial_scale,
                     (__bf16 *)out_hi, (__bf16 *)out_lo, out_f32);
  return launched();
}

}  // extern "C"
