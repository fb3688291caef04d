/* ORACLE (test infrastructure; never linked into or called from the product path).
 *
 * Plain-C CPU restatement of the two CUDA-only native ops on NAFAE's hot path.  The reference has
 * no CPU implementation of either (lib/model/nms/nms_wrapper.py:18, lib/model/roi_align/functions/
 * roi_align.py:29) and ships no test vectors for them, so these follow the CUDA sources line by
 * line and are pinned by hand-checkable cases in tests/test_oracle_native.py
 * ("parity unpinned" by the reference itself -- see DESIGN.md).
 *
 * Floating-point contract (this file is compiled with -O2 -ffp-contract=off, no fast-math):
 *   every +,-,*,/ below is one IEEE-754 binary32 (NMS) or binary64 (ROI-Align interpolation)
 *   operation with no fused multiply-add, in the order written.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* lib/model/nms/src/nms_cuda_kernel.cu:31-39 (devIoU): +1 pixel convention, fp32. */
static inline float dev_iou(const float *a, const float *b) {
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float width = fmaxf(right - left + 1.0f, 0.f), height = fmaxf(bottom - top + 1.0f, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0] + 1.0f) * (a[3] - a[1] + 1.0f);
  float Sb = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
  return interS / (Sa + Sb - interS);
}

float oracle_iou(const float *a, const float *b) { return dev_iou(a, b); }

/* Greedy NMS over boxes already sorted by descending score.
 * nms_cuda_kernel.cu:41-85 builds, for every box i, the bitmask of later boxes j>i with
 * IoU(i,j) > thresh (strict, :78); :123-144 then sweeps i in order, keeping i iff no earlier
 * KEPT box suppressed it.  Equivalent direct form below.  boxes: [n, dim] with dim >= 4
 * (x1,y1,x2,y2[,score]).  keep_out: [n] ascending positions; returns n_keep via *num_out. */
void oracle_nms(int *keep_out, int *num_out, const float *boxes, int n, int dim, float thresh) {
  unsigned char *removed = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
  int nk = 0;
  for (int i = 0; i < n; i++) {
    if (removed[i]) continue;
    keep_out[nk++] = i;
    const float *bi = boxes + (size_t)i * dim;
    for (int j = i + 1; j < n; j++) {
      if (!removed[j] && dev_iou(bi, boxes + (size_t)j * dim) > thresh) removed[j] = 1;
    }
  }
  *num_out = nk;
  free(removed);
}

/* lib/model/roi_align/src/roi_align_kernel.cu:15-70 (ROIAlignForward), element by element.
 * NOTE the reference mixes float and double: literals like `1.` are double, so the ROI extent,
 * the bin size and the whole 4-tap interpolation are evaluated in double and rounded to float
 * on assignment (C usual arithmetic conversions) -- restated faithfully here.
 * features: [B, C, H, W] float; rois: [N,5] (batch_ind, x1,y1,x2,y2); out: [N, C, AH, AW]. */
void oracle_roi_align_forward(const float *bottom_data, float spatial_scale, int num_rois,
                              int height, int width, int channels, int aligned_height,
                              int aligned_width, const float *bottom_rois, float *top_data) {
  long nthreads = (long)num_rois * channels * aligned_height * aligned_width;
  for (long index = 0; index < nthreads; index++) {
    int pw = (int)(index % aligned_width);
    int ph = (int)((index / aligned_width) % aligned_height);
    int c = (int)((index / aligned_width / aligned_height) % channels);
    int n = (int)(index / aligned_width / aligned_height / channels);

    float roi_batch_ind = bottom_rois[n * 5 + 0];
    float roi_start_w = bottom_rois[n * 5 + 1] * spatial_scale;
    float roi_start_h = bottom_rois[n * 5 + 2] * spatial_scale;
    float roi_end_w = bottom_rois[n * 5 + 3] * spatial_scale;
    float roi_end_h = bottom_rois[n * 5 + 4] * spatial_scale;

    float roi_width = fmaxf((float)((double)(roi_end_w - roi_start_w) + 1.), 0.f);  /* :40 */
    float roi_height = fmaxf((float)((double)(roi_end_h - roi_start_h) + 1.), 0.f); /* :41 */
    float bin_size_h = (float)((double)roi_height / ((double)aligned_height - 1.)); /* :42 */
    float bin_size_w = (float)((double)roi_width / ((double)aligned_width - 1.));   /* :43 */

    float h = (float)(ph)*bin_size_h + roi_start_h; /* :45 */
    float w = (float)(pw)*bin_size_w + roi_start_w; /* :46 */

    int hstart = (int)fminf(floorf(h), (float)(height - 2)); /* :48 */
    int wstart = (int)fminf(floorf(w), (float)(width - 2));  /* :49 */

    int img_start = (int)(roi_batch_ind * (float)channels * (float)height * (float)width); /* :51 */

    if (h < 0 || h >= height || w < 0 || w >= width) { /* :54 */
      top_data[index] = 0.f;
    } else {
      float h_ratio = h - (float)(hstart);
      float w_ratio = w - (float)(wstart);
      int upleft = img_start + (c * height + hstart) * width + wstart;
      int upright = upleft + 1;
      int downleft = upleft + width;
      int downright = downleft + 1;
      double v = (double)bottom_data[upleft] * (1. - (double)h_ratio) * (1. - (double)w_ratio) +
                 (double)bottom_data[upright] * (1. - (double)h_ratio) * (double)w_ratio +
                 (double)bottom_data[downleft] * (double)h_ratio * (1. - (double)w_ratio) +
                 (double)bottom_data[downright] * (double)h_ratio * (double)w_ratio; /* :64-67 */
      top_data[index] = (float)v;
    }
  }
}

/* ROIAlignBackward, roi_align_kernel.cu:93-141 (launcher :143-160, binding roi_align_cuda.c:42-79): every top element
 * scatters its gradient to the four taps of its sample with the bilinear weights.  The reference adds with atomicAdd in
 * an unspecified order; this restatement adds in index order.  Float/double mixing as written there: the two upper
 * taps are evaluated in double (`1.` literals) and rounded once, the two lower taps entirely in float. */
void oracle_roi_align_backward(const float *top_diff, float spatial_scale, int num_rois, int height,
                               int width, int channels, int aligned_height, int aligned_width,
                               const float *bottom_rois, float *bottom_diff) {
  long nthreads = (long)num_rois * channels * aligned_height * aligned_width;
  for (long index = 0; index < nthreads; index++) {
    int pw = (int)(index % aligned_width);
    int ph = (int)((index / aligned_width) % aligned_height);
    int c = (int)((index / aligned_width / aligned_height) % channels);
    int n = (int)(index / aligned_width / aligned_height / channels);

    float roi_batch_ind = bottom_rois[n * 5 + 0];
    float roi_start_w = bottom_rois[n * 5 + 1] * spatial_scale;
    float roi_start_h = bottom_rois[n * 5 + 2] * spatial_scale;
    float roi_end_w = bottom_rois[n * 5 + 3] * spatial_scale;
    float roi_end_h = bottom_rois[n * 5 + 4] * spatial_scale;

    float roi_width = fmaxf((float)((double)(roi_end_w - roi_start_w) + 1.), 0.f);  /* :115 */
    float roi_height = fmaxf((float)((double)(roi_end_h - roi_start_h) + 1.), 0.f); /* :116 */
    float bin_size_h = (float)((double)roi_height / ((double)aligned_height - 1.)); /* :117 */
    float bin_size_w = (float)((double)roi_width / ((double)aligned_width - 1.));   /* :118 */

    float h = (float)(ph)*bin_size_h + roi_start_h;
    float w = (float)(pw)*bin_size_w + roi_start_w;
    int hstart = (int)fminf(floorf(h), (float)(height - 2));
    int wstart = (int)fminf(floorf(w), (float)(width - 2));
    int img_start = (int)(roi_batch_ind * (float)channels * (float)height * (float)width);

    if (!(h < 0 || h >= height || w < 0 || w >= width)) { /* :129 */
      float h_ratio = h - (float)(hstart);
      float w_ratio = w - (float)(wstart);
      int upleft = img_start + (c * height + hstart) * width + wstart;
      int upright = upleft + 1;
      int downleft = upleft + width;
      int downright = downleft + 1;
      float td = top_diff[index];
      bottom_diff[upleft] += (float)((double)td * (1. - (double)h_ratio) * (double)(1.f - w_ratio)); /* :137 */
      bottom_diff[upright] += (float)((double)td * (1. - (double)h_ratio) * (double)w_ratio);        /* :138 */
      bottom_diff[downleft] += (td * h_ratio) * (1.f - w_ratio);                                     /* :139 */
      bottom_diff[downright] += (td * h_ratio) * w_ratio;                                            /* :140 */
    }
  }
}

/* lib/model/roi_align/modules/roi_align.py:26-29: align to (P+1)x(P+1), then avg_pool2d(k=2,s=1).
 * torch's CPU avg_pool2d sums the window in (kh, kw) row-major order in fp32 and divides by 4.
 * out: [N, C, P, P]. */
void oracle_roi_align_avg(const float *bottom_data, float spatial_scale, int num_rois, int height,
                          int width, int channels, int pooled, const float *bottom_rois,
                          float *out) {
  int A = pooled + 1;
  size_t nel = (size_t)num_rois * channels * A * A;
  float *tmp = (float *)malloc(nel * sizeof(float));
  oracle_roi_align_forward(bottom_data, spatial_scale, num_rois, height, width, channels, A, A,
                           bottom_rois, tmp);
  for (long nc = 0; nc < (long)num_rois * channels; nc++) {
    const float *t = tmp + (size_t)nc * A * A;
    float *o = out + (size_t)nc * pooled * pooled;
    for (int y = 0; y < pooled; y++)
      for (int x = 0; x < pooled; x++) {
        float s = t[y * A + x];
        s = s + t[y * A + x + 1];
        s = s + t[(y + 1) * A + x];
        s = s + t[(y + 1) * A + x + 1];
        o[y * pooled + x] = s / 4.0f;
      }
  }
  free(tmp);
}
