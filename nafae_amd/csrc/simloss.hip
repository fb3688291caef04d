// simloss.hip -- region x query similarity, contextual-similarity (ranking) loss, visual-clustering loss and
// their backward, for gfx950.  Restates DVSA.forward (reference model.py:517-614) as five kernels:
//
//   sim_max_kernel     S_ = V W^T on fp32 MFMA, masked, reduced on the fly to per-frame max / arg-max over the Nb
//                      proposals (model.py:548-551, 580-583, 610-612).  S_ never reaches HBM.
//   loss_tail_kernel   O(F*Q) ranking term with min-max frame attention + its gradient wrt S_max
//                      (model.py:585-603); one workgroup, everything stays in L2/LDS.
//   cluster_kernel     visual-clustering term per (segment, entity) incl. the reference's frame-0 gather quirk
//                      (model.py:553-577) and its gradient wrt the gathered rows.
//   loss_final_kernel  dem / vis_loss / margin_loss (model.py:576-577, 606).
//   sim_bwd_dv/dw      dV (arg-max rows + clustering rows, optional fused tanh/dropout backward) and dW.
//
// All reductions run in a fixed order (no float atomics), so results are bit-reproducible run to run.
#include "mfma_tile.h"
#include "../../include/nafae_hip.h"
#include "hip_util.h"
#include "sim_common.h"

using namespace nafae;
using nafae_sim::keep_elem;

namespace {

constexpr float EPS = 1e-5f;  // model.py:33

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }
// launch failures (bad configuration, missing code object, wrong runtime) must be loud, never silent
inline int launched() { return hipGetLastError() == hipSuccess ? NAFAE_OK : NAFAE_ELAUNCH; }

__device__ __forceinline__ f32x4 ldg4(const float *p, bool ok) {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return ok ? *reinterpret_cast<const f32x4 *>(p) : z;
}

// ------------------------------------------------------------------------------------------------ sim + max
// grid (ceil(Q/32), F); block 256 = 4 waves x (32 proposals x 32 queries) MFMA tiles; proposals of the frame are
// walked in chunks of 128.
__global__ __launch_bounds__(NTHREADS) void sim_max_kernel(const float *__restrict__ V, const float *__restrict__ Wm,
                                                           const int32_t *__restrict__ ent_len, int Nb, int Ne, int Q,
                                                           int D, float *__restrict__ S_max,
                                                           int64_t *__restrict__ D_ind) {
  using E = Engine<128, 32, 4, 1>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *stile = smem + 2 * E::STAGE;                       // [128][33]
  float *pv = stile + 128 * 33;                             // [8][32] partial max
  int *pi = reinterpret_cast<int *>(pv + 8 * 32);           // [8][32] partial arg-max
  E e;
  e.init();
  const int f = blockIdx.y, q0 = blockIdx.x * 32;
  const int tid = threadIdx.x;
  const int nk = (D + BK - 1) / BK;
  const int col = tid & 31, grp = tid >> 5;
  float best = -INFINITY;
  int best_i = 0;

  const int qrow = q0 + e.srow;  // B tile has 32 rows: thread (srow, slot) stages exactly one chunk
  const bool vb = qrow < Q;
  const float *pb = Wm + (size_t)(vb ? qrow : 0) * D + e.slot * 4;

  for (int chunk = 0; chunk < Nb; chunk += 128) {
    e.zero_acc();
    const float *pa[E::NA];
    bool va[E::NA];
#pragma unroll
    for (int i = 0; i < E::NA; i++) {
      const int r = chunk + e.srow + 32 * i;
      va[i] = r < Nb;
      pa[i] = V + ((size_t)f * Nb + (va[i] ? r : 0)) * D + e.slot * 4;
    }
    f32x4 ra[E::NA], rb[E::NB];
    auto fetch = [&](int kt) {
      const bool kin = kt * BK + e.slot * 4 < D;
#pragma unroll
      for (int i = 0; i < E::NA; i++) ra[i] = ldg4(pa[i] + kt * BK, va[i] && kin);
      rb[0] = ldg4(pb + kt * BK, vb && kin);
    };
    fetch(0);
    e.store_stage(smem, ra, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
      float *cur = smem + (kt & 1) * E::STAGE;
      float *nxt = smem + ((kt + 1) & 1) * E::STAGE;
      if (kt + 1 < nk) fetch(kt + 1);
      e.compute(cur);
      if (kt + 1 < nk) e.store_stage(nxt, ra, rb);
      __syncthreads();
    }
    const int c = e.acc_col(0);
#pragma unroll
    for (int r = 0; r < 16; r++) stile[e.acc_row(0, r) * 33 + c] = e.acc[0][0][r];
    __syncthreads();
    // column scan: thread (grp, col) owns rows grp*16 .. grp*16+15 of the chunk, in ascending order
#pragma unroll 4
    for (int r = 0; r < 16; r++) {
      const int row = grp * 16 + r;
      if (chunk + row < Nb) {
        const float v = stile[row * 33 + col];
        if (v > best) {
          best = v;
          best_i = chunk + row;
        }
      }
    }
    __syncthreads();
  }
  pv[grp * 32 + col] = best;
  pi[grp * 32 + col] = best_i;
  __syncthreads();
  if (tid < 32) {
    float bv = pv[tid];
    int bi = pi[tid];
#pragma unroll
    for (int g = 1; g < 8; g++) {
      const float v = pv[g * 32 + tid];
      const int i = pi[g * 32 + tid];
      if (v > bv || (v == bv && i < bi)) {  // first maximal index, like torch.max(dim)
        bv = v;
        bi = i;
      }
    }
    const int q = q0 + tid;
    if (q < Q) {
      const int a = q / Ne, en = q - a * Ne;
      if (en >= ent_len[a]) {  // masked query slot: the whole S_ column is 0 (model.py:551)
        bv = 0.f;
        bi = 0;
      }
      S_max[(size_t)f * Q + q] = bv;
      D_ind[(size_t)f * Q + q] = (int64_t)bi;
    }
  }
}

// ------------------------------------------------------------------------------------------------ workspace
struct LossWs {
  // float offsets into the workspace
  size_t mn, mx, amin, amax;    // [Na*Q]  (amin/amax are int32)
  size_t Sf, dSf;               // [Na*Ns*Na]
  size_t fs, cs1, rs2;          // [Na*Ns]
  size_t scal;                  // [8]: rank, cscale, vis, dem
  size_t part_sum, part_cnt;    // [Na*Ne]
  size_t cidx;                  // [Na*Ne*Ns] int32
  size_t dgc;                   // [Na*Ne*Ns*D]
  size_t live;                  // int32 [4 + Q]: [0] = number of live query slots L, [4 + j] = query index of live slot j
  size_t total;
};

__host__ __device__ inline LossWs loss_ws(int Na, int Ns, int Nb, int Ne, int D) {
  LossWs w;
  size_t o = 0;
  const size_t Q = (size_t)Na * Ne;
  auto take = [&](size_t n) {
    size_t r = o;
    o += (n + 3) & ~(size_t)3;
    return r;
  };
  w.mn = take(Na * Q);
  w.mx = take(Na * Q);
  w.amin = take(Na * Q);
  w.amax = take(Na * Q);
  w.Sf = take((size_t)Na * Ns * Na);
  w.dSf = take((size_t)Na * Ns * Na);
  w.fs = take((size_t)Na * Ns);
  w.cs1 = take((size_t)Na * Ns);
  w.rs2 = take((size_t)Na * Ns);
  w.scal = take(8);
  w.part_sum = take(Q);
  w.part_cnt = take(Q);
  w.cidx = take(Q * Ns);
  w.dgc = take(Q * Ns * D);
  w.live = take(Q + 4);
  w.total = o;
  (void)Nb;
  return w;
}

// ------------------------------------------------------------------------------------------------ ranking term
// S is S_max viewed as [Na, Ns, Q].  One workgroup of 1024 threads; phases separated by __syncthreads().
__global__ __launch_bounds__(1024) void loss_tail_kernel(const float *__restrict__ Sm,
                                                         const int32_t *__restrict__ ent_len, int Na, int Ns, int Ne,
                                                         float Delta, float *__restrict__ dS, float *__restrict__ ws,
                                                         LossWs L) {
  const int Q = Na * Ne;
  const int tid = threadIdx.x, nt = blockDim.x;
  float *mn = ws + L.mn, *mx = ws + L.mx;
  int *amin = reinterpret_cast<int *>(ws + L.amin), *amax = reinterpret_cast<int *>(ws + L.amax);
  float *Sf = ws + L.Sf, *dSf = ws + L.dSf, *fs = ws + L.fs, *cs1 = ws + L.cs1, *rs2 = ws + L.rs2;
  float *scal = ws + L.scal;
  if (tid == 0) {               // live query slots, ascending (what sim_bwd_dv scans instead of all Q columns)
    int *live = reinterpret_cast<int *>(ws + L.live);
    int n = 0;
    for (int a = 0; a < Na; a++) {
      int l = ent_len[a];
      l = l < 0 ? 0 : (l > Ne ? Ne : l);
      for (int en = 0; en < l; en++) live[4 + n++] = a * Ne + en;
    }
    live[0] = n;
  }

  // A: per (a, q): min / max over the Ns frames of segment a, first-occurrence indices (model.py:587)
  for (int i = tid; i < Na * Q; i += nt) {
    const int a = i / Q, q = i - a * Q;
    float lo = INFINITY, hi = -INFINITY;
    int ilo = 0, ihi = 0;
    for (int s = 0; s < Ns; s++) {
      const float v = Sm[((size_t)a * Ns + s) * Q + q];
      if (v < lo) {
        lo = v;
        ilo = s;
      }
      if (v > hi) {
        hi = v;
        ihi = s;
      }
    }
    mn[i] = lo;
    mx[i] = hi;
    amin[i] = ilo;
    amax[i] = ihi;
  }
  __syncthreads();
  // B: Sf[a,s,j] = sum_e S*att / max(len_j,1)   (model.py:588-592)
  for (int i = tid; i < Na * Ns * Na; i += nt) {
    const int j = i % Na;
    const int as = i / Na;
    const int a = as / Ns;
    float acc = 0.f;
    for (int en = 0; en < Ne; en++) {
      const int q = j * Ne + en;
      const float v = Sm[(size_t)as * Q + q];
      const float lo = mn[a * Q + q], hi = mx[a * Q + q];
      acc += v * ((v - lo) / (hi - lo + EPS));
    }
    const int l = ent_len[j];
    Sf[i] = acc / (float)(l == 0 ? 1 : l);
  }
  __syncthreads();
  // C: frame_score (model.py:603) and the hinge-active counts its backward needs
  for (int i = tid; i < Na * Ns; i += nt) {
    const int a = i / Ns, s = i - a * Ns;
    const float diag = Sf[(a * Ns + s) * Na + a];
    float t1 = 0.f, t2 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int k = 0; k < Na; k++) {
      const float u1 = Sf[(k * Ns + s) * Na + a] - diag + Delta;  // column a, rows k
      const float u2 = Sf[(a * Ns + s) * Na + k] - diag + Delta;  // row a, columns k
      if (u1 > 0.f) {
        t1 += u1;
        c1 += 1.f;
      }
      if (u2 > 0.f) {
        t2 += u2;
        c2 += 1.f;
      }
    }
    fs[i] = t1 / (float)Na + t2 / (float)Na;
    cs1[i] = c1;
    rs2[i] = c2;
  }
  __syncthreads();
  if (tid == 0) {
    float acc = 0.f;
    for (int i = 0; i < Na * Ns; i++) acc += fs[i];
    scal[0] = acc / (float)(Na * Ns);
  }
  if (dS == nullptr) return;
  // E: d(10 * mean fs) / dSf
  const float cN = 10.0f / (float)(Na * Ns) / (float)Na;
  for (int i = tid; i < Na * Ns * Na; i += nt) {
    const int j = i % Na;
    const int as = i / Na;
    const int a = as / Ns, s = as - a * Ns;
    const float v = Sf[i];
    const float g1 = (v - Sf[(j * Ns + s) * Na + j] + Delta > 0.f) ? 1.f : 0.f;
    const float g2 = (v - Sf[(a * Ns + s) * Na + a] + Delta > 0.f) ? 1.f : 0.f;
    float g = cN * (g1 + g2);
    if (a == j) g -= cN * (cs1[a * Ns + s] + rs2[a * Ns + s]);
    dSf[i] = g;
  }
  __syncthreads();
  // F: back through S*att with the min / max paths (model.py:587-588)
  for (int i = tid; i < Na * Q; i += nt) {
    const int a = i / Q, q = i - a * Q;
    const int j = q / Ne, en = q - j * Ne;
    const bool masked = en >= ent_len[j];
    const int l = ent_len[j];
    const float inv_div = 1.0f / (float)(l == 0 ? 1 : l);
    const float lo = mn[i], hi = mx[i];
    const float den = hi - lo + EPS;
    float gmn = 0.f, gmx = 0.f;
    for (int s = 0; s < Ns; s++) {
      const float v = Sm[((size_t)a * Ns + s) * Q + q];
      const float dT = dSf[(a * Ns + s) * Na + j] * inv_div;
      gmn += dT * v * (v - hi - EPS) / (den * den);
      gmx -= dT * v * (v - lo) / (den * den);
    }
    const int ilo = amin[i], ihi = amax[i];
    for (int s = 0; s < Ns; s++) {
      const float v = Sm[((size_t)a * Ns + s) * Q + q];
      const float dT = dSf[(a * Ns + s) * Na + j] * inv_div;
      float g = dT * ((v - lo) / den + v / den);
      if (s == ilo) g += gmn;
      if (s == ihi) g += gmx;
      dS[((size_t)a * Ns + s) * Q + q] = masked ? 0.f : g;
    }
  }
}

// The same ranking term with everything on chip: only the L LIVE query slots (e < len_a) carry information -- a masked
// column of S_max is identically 0 and adds +0 to every sum below -- so the kernel compacts them (prefix scan of ent_len),
// copies S_max[:, live] into LDS once (F*L floats: 4 KB at C5 with histogram lengths) and runs all phases out of LDS instead
// of re-reading S_max from L2 in every phase (six dependent global round trips per entry in the kernel above: 70 us at C5).
// Same arithmetic, same summation order over the live slots.  dS is written for every (frame, slot): 0 for masked ones.
// Lcap = the host's upper bound on L (LDS is sized with it); L > Lcap is reported as NaN loss, never silently truncated.
__device__ __forceinline__ void loss_tail_lds_body(const float *__restrict__ Sm, const int32_t *__restrict__ ent_len,
                                                   int Na, int Ns, int Ne, float Delta, float *__restrict__ dS,
                                                   float *__restrict__ ws, LossWs L, int Lcap) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Q = Na * Ne, F = Na * Ns;
  const int tid = threadIdx.x, nt = blockDim.x;
  float *Sl = sm;                                   // [F][Lcap]
  float *mn = Sl + (size_t)F * Lcap, *mx = mn + Na * Lcap;
  int *amin = reinterpret_cast<int *>(mx + Na * Lcap), *amax = amin + Na * Lcap;
  float *Sf = reinterpret_cast<float *>(amax + Na * Lcap), *dSf = Sf + Na * Ns * Na;
  float *fs = dSf + Na * Ns * Na, *cs1 = fs + Na * Ns, *rs2 = cs1 + Na * Ns;
  int *prefix = reinterpret_cast<int *>(rs2 + Na * Ns);     // [Na + 1]
  int *qlist = prefix + Na + 1;                              // [Lcap]
  int *colmap = qlist + Lcap;                                // [Q]: live index of a query slot, or -1
  float *scal = ws + L.scal;

  if (tid < 64) {               // exclusive prefix of the clamped entity counts (one wave)
    int carry = 0;
    for (int base = 0; base < Na; base += 64) {
      const int a = base + tid;
      int x = a < Na ? ent_len[a] : 0;
      x = x < 0 ? 0 : (x > Ne ? Ne : x);
      int incl = x;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o);
        if (tid >= o) incl += y;
      }
      if (a < Na) prefix[a] = carry + incl - x;
      carry += __shfl(incl, 63);
    }
    if (tid == 0) prefix[Na] = carry;
  }
  __syncthreads();
  const int Lc = prefix[Na];
  if (Lc > Lcap) {              // the caller's bound was too small: fail loudly
    if (tid == 0) scal[0] = NAN;
    return;
  }
  for (int q = tid; q < Q; q += nt) {
    const int a = q / Ne, en = q - a * Ne;
    const int l = prefix[a + 1] - prefix[a];
    const int j = en < l ? prefix[a] + en : -1;
    colmap[q] = j;
    if (j >= 0) qlist[j] = q;
  }
  __syncthreads();
  {
    int *live = reinterpret_cast<int *>(ws + L.live);
    if (tid == 0) live[0] = Lc;
    for (int j = tid; j < Lc; j += nt) live[4 + j] = qlist[j];
  }
  for (int i = tid; i < F * Lc; i += nt) {
    const int f = i / Lc, j = i - f * Lc;
    Sl[f * Lcap + j] = Sm[(size_t)f * Q + qlist[j]];
  }
  __syncthreads();
  // A: per (a, live slot): min / max over the Ns frames of segment a, first-occurrence indices (model.py:587)
  for (int i = tid; i < Na * Lc; i += nt) {
    const int a = i / Lc, j = i - a * Lc;
    float lo = INFINITY, hi = -INFINITY;
    int ilo = 0, ihi = 0;
    for (int s = 0; s < Ns; s++) {
      const float v = Sl[(a * Ns + s) * Lcap + j];
      if (v < lo) {
        lo = v;
        ilo = s;
      }
      if (v > hi) {
        hi = v;
        ihi = s;
      }
    }
    mn[a * Lcap + j] = lo;
    mx[a * Lcap + j] = hi;
    amin[a * Lcap + j] = ilo;
    amax[a * Lcap + j] = ihi;
    ws[L.mn + (size_t)a * Q + qlist[j]] = lo;      // the clustering kernel reads its own-segment entries
    ws[L.mx + (size_t)a * Q + qlist[j]] = hi;
  }
  __syncthreads();
  // B: Sf[a,s,j] = sum_e S*att / max(len_j,1)   (model.py:588-592); masked slots add exactly +0
  for (int i = tid; i < Na * Ns * Na; i += nt) {
    const int j = i % Na;
    const int as = i / Na;
    const int a = as / Ns;
    float acc = 0.f;
    for (int c = prefix[j]; c < prefix[j + 1]; c++) {
      const float v = Sl[as * Lcap + c];
      const float lo = mn[a * Lcap + c], hi = mx[a * Lcap + c];
      acc += v * ((v - lo) / (hi - lo + EPS));
    }
    const int l = ent_len[j];
    Sf[i] = acc / (float)(l == 0 ? 1 : l);
  }
  __syncthreads();
  // C: frame_score (model.py:603) and the hinge-active counts its backward needs
  for (int i = tid; i < Na * Ns; i += nt) {
    const int a = i / Ns, s = i - a * Ns;
    const float diag = Sf[(a * Ns + s) * Na + a];
    float t1 = 0.f, t2 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int k = 0; k < Na; k++) {
      const float u1 = Sf[(k * Ns + s) * Na + a] - diag + Delta;
      const float u2 = Sf[(a * Ns + s) * Na + k] - diag + Delta;
      if (u1 > 0.f) {
        t1 += u1;
        c1 += 1.f;
      }
      if (u2 > 0.f) {
        t2 += u2;
        c2 += 1.f;
      }
    }
    fs[i] = t1 / (float)Na + t2 / (float)Na;
    cs1[i] = c1;
    rs2[i] = c2;
  }
  __syncthreads();
  if (tid == 0) {
    float acc = 0.f;
    for (int i = 0; i < Na * Ns; i++) acc += fs[i];
    scal[0] = acc / (float)(Na * Ns);
  }
  if (dS == nullptr) return;
  // E: d(10 * mean fs) / dSf
  const float cN = 10.0f / (float)(Na * Ns) / (float)Na;
  for (int i = tid; i < Na * Ns * Na; i += nt) {
    const int j = i % Na;
    const int as = i / Na;
    const int a = as / Ns, s = as - a * Ns;
    const float v = Sf[i];
    const float g1 = (v - Sf[(j * Ns + s) * Na + j] + Delta > 0.f) ? 1.f : 0.f;
    const float g2 = (v - Sf[(a * Ns + s) * Na + a] + Delta > 0.f) ? 1.f : 0.f;
    float g = cN * (g1 + g2);
    if (a == j) g -= cN * (cs1[a * Ns + s] + rs2[a * Ns + s]);
    dSf[i] = g;
  }
  __syncthreads();
  // F: back through S*att with the min / max paths (model.py:587-588), live slots
  for (int i = tid; i < Na * Lc; i += nt) {
    const int a = i / Lc, c = i - a * Lc;
    const int q = qlist[c];
    const int j = q / Ne;
    const int l = ent_len[j];
    const float inv_div = 1.0f / (float)(l == 0 ? 1 : l);
    const float lo = mn[a * Lcap + c], hi = mx[a * Lcap + c];
    const float den = hi - lo + EPS;
    float gmn = 0.f, gmx = 0.f;
    for (int s = 0; s < Ns; s++) {
      const float v = Sl[(a * Ns + s) * Lcap + c];
      const float dT = dSf[(a * Ns + s) * Na + j] * inv_div;
      gmn += dT * v * (v - hi - EPS) / (den * den);
      gmx -= dT * v * (v - lo) / (den * den);
    }
    const int ilo = amin[a * Lcap + c], ihi = amax[a * Lcap + c];
    for (int s = 0; s < Ns; s++) {
      const float v = Sl[(a * Ns + s) * Lcap + c];
      const float dT = dSf[(a * Ns + s) * Na + j] * inv_div;
      float g = dT * ((v - lo) / den + v / den);
      if (s == ilo) g += gmn;
      if (s == ihi) g += gmx;
      dS[((size_t)a * Ns + s) * Q + q] = g;
    }
  }
  // masked slots: exactly 0 (model.py:551 zero-fills the column, so no gradient flows)
  for (int i = tid; i < F * Q; i += nt)
    if (colmap[i % Q] < 0) dS[i] = 0.f;
}

// The ranking term for ANY number of live slots (C5 with all 512 slots live: F*L floats do not fit one workgroup's LDS, and the
// global-memory kernel above walks every phase through L2: 78 us).  The term couples the columns of S_max only through
// Sf[a,s,j] = sum over the slots e of segment j, so it splits by segment: loss_seg_fwd_kernel, one workgroup per segment j,
// brings S_max[:, slots of j] into LDS (F * len_j floats: 16 KB at most at C5), takes the min / max over the frames of every
// segment a and writes Sf[:, :, j]; loss_seg_bwd_kernel, again one workgroup per segment j, recomputes the O(F * Na) hinge terms
// from the complete Sf (every workgroup the same arithmetic in the same order), and turns them into dS for the slots of j.
// Same arithmetic and summation order as loss_tail_lds_kernel; the kernel boundary is the only synchronisation.
__device__ __forceinline__ int clamp_len(int l, int Ne) { return l < 0 ? 0 : (l > Ne ? Ne : l); }

__device__ __forceinline__ void loss_seg_fwd_body(int j, const float *__restrict__ Sm, const int32_t *__restrict__ ent_len, int Na,
                                                  int Ns, int Ne, float *__restrict__ ws, LossWs L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int Q = Na * Ne, F = Na * Ns;
  const int lraw = ent_len[j];
  const int l = clamp_len(lraw, Ne);
  int off = 0;
  for (int a = 0; a < j; a++) off += clamp_len(ent_len[a], Ne);
  {  // this segment's part of the compact live-slot list (what sim_bwd_dv scans)
    int *live = reinterpret_cast<int *>(ws + L.live);
    for (int e = tid; e < l; e += nt) live[4 + off + e] = j * Ne + e;
    if (j == Na - 1 && tid == 0) live[0] = off + l;
  }
  float *Sf = ws + L.Sf;
  if (l == 0) {
    for (int i = tid; i < F; i += nt) Sf[(size_t)i * Na + j] = 0.f;      // (0 / max(len, 1))
    return;
  }
  float *Sl = sm, *mn = Sl + (size_t)F * l, *mx = mn + (size_t)Na * l;
  for (int i = tid; i < F * l; i += nt) {
    const int f = i / l, e = i - f * l;
    Sl[i] = Sm[(size_t)f * Q + j * Ne + e];
  }
  __syncthreads();
  int *amin = reinterpret_cast<int *>(ws + L.amin), *amax = reinterpret_cast<int *>(ws + L.amax);
  for (int i = tid; i < Na * l; i += nt) {     // A: min / max over the Ns frames of segment a, first occurrences (model.py:587)
    const int a = i / l, e = i - a * l;
    float lo = INFINITY, hi = -INFINITY;
    int ilo = 0, ihi = 0;
    for (int s = 0; s < Ns; s++) {
      const float v = Sl[(a * Ns + s) * l + e];
      if (v < lo) {
        lo = v;
        ilo = s;
      }
      if (v > hi) {
        hi = v;
        ihi = s;
      }
    }
    mn[i] = lo;
    mx[i] = hi;
    const size_t g = (size_t)a * Q + j * Ne + e;
    ws[L.mn + g] = lo;
    ws[L.mx + g] = hi;
    amin[g] = ilo;
    amax[g] = ihi;
  }
  __syncthreads();
  for (int i = tid; i < F; i += nt) {          // B: Sf[a,s,j] = sum_e S*att / max(len_j,1)   (model.py:588-592)
    const int a = i / Ns;
    float acc = 0.f;
    for (int e = 0; e < l; e++) {
      const float v = Sl[i * l + e];
      const float lo = mn[a * l + e], hi = mx[a * l + e];
      acc += v * ((v - lo) / (hi - lo + EPS));
    }
    Sf[(size_t)i * Na + j] = acc / (float)(lraw == 0 ? 1 : lraw);
  }
}

__global__ __launch_bounds__(256) void loss_seg_bwd_kernel(const float *__restrict__ Sm, const int32_t *__restrict__ ent_len, int Na,
                                                           int Ns, int Ne, float Delta, float *__restrict__ dS,
                                                           float *__restrict__ ws, LossWs L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int j = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int Q = Na * Ne, F = Na * Ns;
  float *Sf = sm, *fs = Sf + (size_t)F * Na, *cs1 = fs + F, *rs2 = cs1 + F, *dcol = rs2 + F;
  for (int i = tid; i < F * Na; i += nt) Sf[i] = ws[L.Sf + i];
  __syncthreads();
  for (int i = tid; i < F; i += nt) {          // C: frame_score (model.py:603) and the hinge-active counts
    const int a = i / Ns, s = i - a * Ns;
    const float diag = Sf[(a * Ns + s) * Na + a];
    float t1 = 0.f, t2 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int k = 0; k < Na; k++) {
      const float u1 = Sf[(k * Ns + s) * Na + a] - diag + Delta;
      const float u2 = Sf[(a * Ns + s) * Na + k] - diag + Delta;
      if (u1 > 0.f) {
        t1 += u1;
        c1 += 1.f;
      }
      if (u2 > 0.f) {
        t2 += u2;
        c2 += 1.f;
      }
    }
    fs[i] = t1 / (float)Na + t2 / (float)Na;
    cs1[i] = c1;
    rs2[i] = c2;
  }
  __syncthreads();
  if (j == 0 && tid == 0) {
    float acc = 0.f;
    for (int i = 0; i < F; i++) acc += fs[i];
    ws[L.scal] = acc / (float)F;
  }
  if (dS == nullptr) return;
  const float cN = 10.0f / (float)F / (float)Na;
  for (int i = tid; i < F; i += nt) {          // E: d(10 * mean fs) / dSf[:, :, j]
    const int a = i / Ns, s = i - a * Ns;
    const float v = Sf[i * Na + j];
    const float g1 = (v - Sf[(j * Ns + s) * Na + j] + Delta > 0.f) ? 1.f : 0.f;
    const float g2 = (v - Sf[(a * Ns + s) * Na + a] + Delta > 0.f) ? 1.f : 0.f;
    float g = cN * (g1 + g2);
    if (a == j) g -= cN * (cs1[i] + rs2[i]);
    dcol[i] = g;
  }
  __syncthreads();
  const int lraw = ent_len[j];
  const int l = clamp_len(lraw, Ne);
  const float inv_div = 1.0f / (float)(lraw == 0 ? 1 : lraw);
  const int *amin = reinterpret_cast<const int *>(ws + L.amin), *amax = reinterpret_cast<const int *>(ws + L.amax);
  for (int i = tid; i < Na * l; i += nt) {     // F: back through S*att with the min / max paths (model.py:587-588)
    const int a = i / l, e = i - a * l;
    const int q = j * Ne + e;
    const size_t g = (size_t)a * Q + q;
    const float lo = ws[L.mn + g], hi = ws[L.mx + g];
    const float den = hi - lo + EPS;
    float gmn = 0.f, gmx = 0.f;
    for (int s = 0; s < Ns; s++) {
      const float v = Sm[((size_t)a * Ns + s) * Q + q];
      const float dT = dcol[a * Ns + s] * inv_div;
      gmn += dT * v * (v - hi - EPS) / (den * den);
      gmx -= dT * v * (v - lo) / (den * den);
    }
    const int ilo = amin[g], ihi = amax[g];
    for (int s = 0; s < Ns; s++) {
      const float v = Sm[((size_t)a * Ns + s) * Q + q];
      const float dT = dcol[a * Ns + s] * inv_div;
      float gg = dT * ((v - lo) / den + v / den);
      if (s == ilo) gg += gmn;
      if (s == ihi) gg += gmx;
      dS[((size_t)a * Ns + s) * Q + q] = gg;
    }
  }
  const int nm = Ne - l;                       // masked slots of this segment: exactly 0 (model.py:551)
  for (int i = tid; i < F * nm; i += nt) {
    const int f = i / nm, e = l + (i - f * nm);
    dS[(size_t)f * Q + j * Ne + e] = 0.f;
  }
}

inline size_t loss_tail_lds_bytes(int Na, int Ns, int Ne, int Lcap) {
  const size_t F = (size_t)Na * Ns, Q = (size_t)Na * Ne;
  return 4 * (F * Lcap + 4 * (size_t)Na * Lcap + 2 * (size_t)Na * Ns * Na + 3 * F + (Na + 1) + Lcap + Q) + 16;
}

// ------------------------------------------------------------------------------------------------ clustering term
__device__ __forceinline__ float block_sum(float v, float *red) {
  // 256 threads; result broadcast to all
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
// any number of waves (<= 16), fixed order
__device__ __forceinline__ float block_sum_n(float v, float *red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < nw; k++) t += red[k];
  return t;
}

// grid Na*Ne, block 512 (8 waves: one per frame row at Ns = 8).  Dynamic LDS: g[Ns][D] raw gathered rows, G[Ns][D]
// normalised * sn, tot[D] = sum_s G[s].
// All Ns gathered rows are requested in ONE pass (independent 16-B loads, one barrier) and the per-row reductions (norm,
// g . dG) are wave reductions -- a wave owns rows w, w+4, ... -- instead of Ns sequential block reductions with a global
// round trip each (the first version: 36 us, most of it that serialised gather).
// The min / max of the slot's OWN segment (what model.py:567 normalises with) are taken here from the Ns values the kernel loads
// anyway, with the loss tail's loop (first occurrence, `<` / `>`: the same bits) -- so the kernel does not depend on the tail and
// runs beside it in one launch (loss_lds_cluster_kernel / loss_segf_cluster_kernel below).
__device__ __forceinline__ void cluster_body(int ae, const float *__restrict__ Sm, const int64_t *__restrict__ D_ind,
                                             const float *__restrict__ V,
                                             const int32_t *__restrict__ ent_len, int Na, int Ns, int Ne,
                                             int D, float *__restrict__ ws, LossWs L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[16];
  __shared__ float s_sn[64], s_nrm[64], s_dot[64];
  __shared__ int s_idx[64];
  float *g = sm, *G = sm + (size_t)Ns * D, *tot = G + (size_t)Ns * D;
  const int a = ae / Ne, en = ae - a * Ne;
  const int Q = Na * Ne, q = a * Ne + en;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nt = blockDim.x, nw = nt >> 6;
  float *part_sum = ws + L.part_sum, *part_cnt = ws + L.part_cnt;
  int *cidx = reinterpret_cast<int *>(ws + L.cidx) + (size_t)ae * Ns;
  float *dgc = ws + L.dgc + (size_t)ae * Ns * D;
  if (en >= ent_len[a]) {  // padded entity slot: masked out of the loss (model.py:540-542)
    if (tid == 0) {
      part_sum[ae] = 0.f;
      part_cnt[ae] = 0.f;
    }
    for (int s = tid; s < Ns; s += nt) cidx[s] = -1;
    return;
  }
  for (int s = tid; s < Ns; s += nt) {
    const size_t o = ((size_t)a * Ns + s) * Q + q;
    s_dot[s] = Sm[o];                          // (parked here until the normalisation below)
    const int ix = (int)D_ind[o];              // in [0, Nb): indexes V WITHOUT a frame offset (model.py:562-569)
    s_idx[s] = ix;
    cidx[s] = ix;
  }
  __syncthreads();
  {
    float lo = INFINITY, hi = -INFINITY;
    for (int s = 0; s < Ns; s++) {
      const float v = s_dot[s];
      if (v < lo) lo = v;
      if (v > hi) hi = v;
    }
    for (int s = tid; s < Ns; s += nt) s_sn[s] = (s_dot[s] - lo) / (hi - lo + EPS);  // model.py:567 (no grad)
  }
  const int d4n = D >> 2;
  for (int i = tid; i < Ns * d4n; i += nt) {
    const int s = i / d4n, d4 = i - s * d4n;
    *reinterpret_cast<f32x4 *>(&g[s * D + d4 * 4]) = *reinterpret_cast<const f32x4 *>(V + (size_t)s_idx[s] * D + d4 * 4);
  }
  __syncthreads();
  for (int s = wave; s < Ns; s += nw) {
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) acc += g[s * D + d] * g[s * D + d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) s_nrm[s] = sqrtf(acc);
  }
  __syncthreads();
  for (int i = tid; i < Ns * D; i += nt) {
    const int s = i / D;
    G[i] = (g[i] / (s_nrm[s] + EPS)) * s_sn[s];  // (g / (norm + EPS)) * sn   (model.py:570-571)
  }
  __syncthreads();
  for (int d = tid; d < D; d += nt) {
    float t = 0.f;
    for (int s2 = 0; s2 < Ns; s2++) t += G[s2 * D + d];
    tot[d] = t;
  }
  // gram: 1 - G_s . G_s' for s != s' (model.py:574-575); sum and count of non-zeros (model.py:576)
  float lsum = 0.f, lcnt = 0.f;
  for (int p = wave; p < Ns * Ns; p += nw) {
    const int s1 = p / Ns, s2 = p - s1 * Ns;
    if (s1 == s2) continue;
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) acc += G[s1 * D + d] * G[s2 * D + d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const float m = 1.0f - acc;
    if (lane == 0) {
      lsum += m;
      if (m != 0.f) lcnt += 1.f;
    }
  }
  // (lane 0 of each wave holds its partials; everyone else 0)
  lsum = block_sum_n(lsum, red);
  lcnt = block_sum_n(lcnt, red);       // (its barriers also publish tot[])
  if (tid == 0) {
    part_sum[ae] = lsum;
    part_cnt[ae] = lcnt;
  }
  // gradient wrt the gathered rows, unscaled by 10*vis_lam/dem: dG_s = -2 (sum_s' G_s' - G_s)
  for (int s = wave; s < Ns; s += nw) {
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) acc += g[s * D + d] * (-2.0f * (tot[d] - G[s * D + d]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) s_dot[s] = acc;
  }
  __syncthreads();
  for (int i = tid; i < Ns * D; i += nt) {
    const int s = i / D, d = i - s * D;
    const float n = s_nrm[s], ne = n + EPS, sn = s_sn[s];
    // u = g/(n+EPS), G = sn*u;  dg = sn * ( dG/(n+EPS) - g * (g.dG) / (n * (n+EPS)^2) )
    const float k2 = s_dot[s] / (n * ne * ne);
    const float dG = -2.0f * (tot[d] - G[i]);
    dgc[i] = sn * (dG / ne - g[i] * k2);
  }
}

__global__ __launch_bounds__(512) void cluster_kernel(const float *__restrict__ Sm, const int64_t *__restrict__ D_ind,
                                                      const float *__restrict__ V,
                                                      const int32_t *__restrict__ ent_len, int Na, int Ns, int Ne,
                                                      int D, float *__restrict__ ws, LossWs L) {
  cluster_body(blockIdx.x, Sm, D_ind, V, ent_len, Na, Ns, Ne, D, ws, L);
}

__global__ __launch_bounds__(1024) void loss_tail_lds_kernel(const float *__restrict__ Sm, const int32_t *__restrict__ ent_len,
                                                             int Na, int Ns, int Ne, float Delta, float *__restrict__ dS,
                                                             float *__restrict__ ws, LossWs L, int Lcap) {
  loss_tail_lds_body(Sm, ent_len, Na, Ns, Ne, Delta, dS, ws, L, Lcap);
}

__global__ __launch_bounds__(256) void loss_seg_fwd_kernel(const float *__restrict__ Sm, const int32_t *__restrict__ ent_len, int Na,
                                                           int Ns, int Ne, float *__restrict__ ws, LossWs L) {
  loss_seg_fwd_body(blockIdx.x, Sm, ent_len, Na, Ns, Ne, ws, L);
}

// Ranking term and clustering term in ONE launch (training): they share their inputs and nothing else, so back to back on a
// stream the second only waited for the first (12 + 16 us at C5).  Workgroup 0 (resp. the first Na) = the loss tail, the
// others = one query slot of the clustering term each; the dynamic LDS is the larger of the two needs.
__global__ __launch_bounds__(1024) void loss_lds_cluster_kernel(const float *__restrict__ Sm, const int64_t *__restrict__ D_ind,
                                                                const float *__restrict__ V, const int32_t *__restrict__ ent_len,
                                                                int Na, int Ns, int Ne, int D, float Delta, float *__restrict__ dS,
                                                                float *__restrict__ ws, LossWs L, int Lcap) {
  if (blockIdx.x == 0)
    loss_tail_lds_body(Sm, ent_len, Na, Ns, Ne, Delta, dS, ws, L, Lcap);
  else
    cluster_body(blockIdx.x - 1, Sm, D_ind, V, ent_len, Na, Ns, Ne, D, ws, L);
}
__global__ __launch_bounds__(512) void loss_segf_cluster_kernel(const float *__restrict__ Sm, const int64_t *__restrict__ D_ind,
                                                                const float *__restrict__ V, const int32_t *__restrict__ ent_len,
                                                                int Na, int Ns, int Ne, int D, float *__restrict__ ws, LossWs L) {
  if ((int)blockIdx.x < Na)
    loss_seg_fwd_body(blockIdx.x, Sm, ent_len, Na, Ns, Ne, ws, L);
  else
    cluster_body(blockIdx.x - Na, Sm, D_ind, V, ent_len, Na, Ns, Ne, D, ws, L);
}

__global__ __launch_bounds__(64) void loss_final_kernel(float *__restrict__ ws, LossWs L, int Na, int Ne, float vis_lam, int train,
                                                        float *__restrict__ loss_out) {
  // one wave: lane-strided partial sums in a fixed order, then a fixed shuffle tree (deterministic)
  const int lane = threadIdx.x;
  float *scal = ws + L.scal;
  const float rank = scal[0];
  float vis = 0.f, dem = 0.f, cscale = 0.f;
  if (train) {
    float sum = 0.f;
    for (int i = lane; i < Na * Ne; i += 64) {
      sum += ws[L.part_sum + i];
      dem += ws[L.part_cnt + i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sum += __shfl_xor(sum, o);
      dem += __shfl_xor(dem, o);
    }
    vis = sum / dem;  // dem == 0 -> NaN/Inf exactly like the reference (model.py:577)
    cscale = 10.0f * vis_lam / dem;
  }
  if (lane != 0) return;
  scal[1] = cscale;
  scal[2] = vis;
  scal[3] = dem;
  loss_out[0] = train ? (rank + vis_lam * vis) * 10.0f : rank * 10.0f;  // model.py:606
  loss_out[1] = rank;
  loss_out[2] = vis;
  loss_out[3] = dem;
}

// ------------------------------------------------------------------------------------------------ backward
constexpr int MAXCH = 8;  // D <= 2048 : float4 chunks per lane

// one wave per region row r
// (Measured and NOT adopted, twice: a workgroup per (frame, 16-row slice) that stages the frame's (dS, arg-max, query) entries in
// LDS once and lets its waves scan them from there -- it takes the 115 MB of per-row entry re-reads at C5 with all slots live
// off the L2, but a wave then walks its four rows one after the other, each with its dependent W-row round trips: 157 us
// against 52 us for this form, whose 19 200 independent waves hide those round trips behind each other.)
// CP = 1: one wave per row r = 4 * blk + wave (the rows that carry no clustering gradient).  CP = 4: the WORKGROUP blk is row
// r = blk of frame 0, a row the clustering term lands on: wave 0 takes the slot hits and the first quarter of the clustering
// entries, waves 1-3 the other quarters; the partial rows are added in wave order through LDS (with all 512 slots of C5 live
// such a row has 4 096 entries to scan and ~14 rows to gather -- as one wave per row the 300 of them were the critical path).
template <int MC, int CP>
__device__ __forceinline__ void sim_bwd_dv_body(int blk, const float *__restrict__ dS, const int64_t *__restrict__ D_ind,
                                                const float *__restrict__ Wm, int R, int Nb, int Q, int D,
                                                int train, int n_centries, const float *__restrict__ ws,
                                                LossWs L, const float *__restrict__ pre_scale,
                                                const float *__restrict__ grad_scale,
                                                float *__restrict__ dV) {
  constexpr int NBT = MC <= 2 ? 4 : 2;         // hits whose W rows are in flight together
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  const int lane = threadIdx.x & 63;
  int *lq = reinterpret_cast<int *>(dyn) + (threadIdx.x >> 6) * 1024;   // this wave's hit list: [512] query slot,
  float *lw = reinterpret_cast<float *>(lq + 512);                      //                       [512] dS
  const int wave = threadIdx.x >> 6;
  const int r = CP == 1 ? blk * 4 + wave : blk;
  if (r >= R) return;
  if (CP == 1 && train && r < Nb) return;      // (a clustering row: a CP = 4 workgroup writes it)
  const int f = r / Nb, b = r - f * Nb;
  f32x4 acc[MC];
#pragma unroll
  for (int c = 0; c < MC; c++) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // a block of 64 candidates: the hits (row index `id` of the source matrix, weight `wt`) are appended to the wave's list in lane
  // order; returns the new list length
  auto push_hits = [&](bool hit, int id, float wt, int H) {
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const int pos = H + __popcll(m & ((1ull << lane) - 1ull));
      lq[pos] = id;
      lw[pos] = wt;
    }
    return H + __popcll(m);
  };
  // acc += wt * src[id] over the list, in list order, NBT source rows requested together
  auto gather_hits = [&](int H, const float *__restrict__ src) {
    for (int i = 0; i < H; i += NBT) {
      float w[NBT];
      const float *wr[NBT];
#pragma unroll
      for (int t = 0; t < NBT; t++) {
        const int k = i + t < H ? i + t : H - 1;
        w[t] = lw[k];
        wr[t] = src + (size_t)lq[k] * D;
      }
      f32x4 x[NBT][MC];
#pragma unroll
      for (int t = 0; t < NBT; t++)
#pragma unroll
        for (int c = 0; c < MC; c++) {
          const int d = lane * 4 + c * 256;
          if (d < D) x[t][c] = *reinterpret_cast<const f32x4 *>(wr[t] + d);
        }
#pragma unroll
      for (int t = 0; t < NBT; t++)
        if (i + t < H) {
#pragma unroll
          for (int c = 0; c < MC; c++) {
            const int d = lane * 4 + c * 256;
            if (d < D) acc[c] += w[t] * x[t][c];
          }
        }
    }
  };
  // only live query slots can carry gradient (dS of a masked slot is exactly 0): scan the compact list the loss tail left
  // in the workspace instead of all Q columns (C5: 17 of 512)
  const int *live = ws ? reinterpret_cast<const int *>(ws + L.live) : nullptr;
  const int nq = live ? live[0] : Q;
  const bool ident = !live || nq == Q;        // every slot live: the list is 0 .. Q-1, one dependent round trip fewer
  // The (dS, arg-max) entries of up to 512 live slots are requested together before the first ballot (one dependent L2 round
  // trip per 512 slots instead of one per 64: with all 512 slots of C5 live the row spent eight of them back to back).
  for (int qb0 = 0; qb0 < (CP == 1 || wave == 0 ? nq : 0); qb0 += 512) {
    float dsv[8];
    int qv[8];
    bool hv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int j = qb0 + u * 64 + lane;
      dsv[u] = 0.f;
      qv[u] = 0;
      hv[u] = false;
      if (j < nq) {
        qv[u] = ident ? j : live[4 + j];
        dsv[u] = dS[(size_t)f * Q + qv[u]];
        hv[u] = ((int)D_ind[(size_t)f * Q + qv[u]] == b) && dsv[u] != 0.f;
      }
    }
    // The row's hits, compacted in ascending slot order into this wave's LDS list, then taken NBT at a time with their W rows
    // requested TOGETHER (one by one, every hit was a dependent L2 round trip).
    int H = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) H = push_hits(hv[u], qv[u], dsv[u], H);
    gather_hits(H, Wm);
  }
  if (CP > 1) {           // clustering gradient lands on rows [0, Nb) only (reference quirk)
    const int *cidx = reinterpret_cast<const int *>(ws + L.cidx);
    const float cscale = ws[L.scal + 1];
    // entries t = (slot q, frame s) of live slots only: q*Ns + s (masked slots hold cidx = -1 and no gradient).  Scanned 512 at
    // a time like the slots above (64 at a time, each block a dependent round trip, the 4 096 entries of C5 with every slot live
    // made the 300 rows of frame 0 the kernel's critical path: 52 us).
    const int Nsc = n_centries / Q;
    const int ne = live ? nq * Nsc : n_centries;
    const int per = ((ne + CP - 1) / CP + 511) / 512 * 512;      // entries per wave, whole passes
    const int e_end = (wave + 1) * per < ne ? (wave + 1) * per : ne;
    for (int tb0 = wave * per; tb0 < e_end; tb0 += 512) {
      int tv[8];
      bool hv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = tb0 + u * 64 + lane;
        tv[u] = 0;
        hv[u] = false;
        if (e < e_end) {
          tv[u] = ident ? e : live[4 + e / Nsc] * Nsc + e % Nsc;
          hv[u] = cidx[tv[u]] == r;
        }
      }
      int H = 0;
#pragma unroll
      for (int u = 0; u < 8; u++) H = push_hits(hv[u], tv[u], cscale, H);
      gather_hits(H, ws + L.dgc);
    }
    // waves 1 .. 3 hand their partial rows to wave 0: ((w0 + w1) + w2) + w3
    float *part = dyn + 4 * 1024;                // [3][D], behind the four hit lists
    if (wave > 0) {
#pragma unroll
      for (int c = 0; c < MC; c++) {
        const int d = lane * 4 + c * 256;
        if (d < D) *reinterpret_cast<f32x4 *>(part + (size_t)(wave - 1) * D + d) = acc[c];
      }
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int u = 0; u < CP - 1; u++)
#pragma unroll
      for (int c = 0; c < MC; c++) {
        const int d = lane * 4 + c * 256;
        if (d < D) acc[c] += *reinterpret_cast<const f32x4 *>(part + (size_t)u * D + d);
      }
  }
#pragma unroll
  for (int c = 0; c < MC; c++) {
    const int d = lane * 4 + c * 256;
    if (d < D) {
      f32x4 v = acc[c];
      if (grad_scale) v *= grad_scale[0];
      if (pre_scale) v *= *reinterpret_cast<const f32x4 *>(pre_scale + (size_t)r * D + d);
      *reinterpret_cast<f32x4 *>(dV + (size_t)r * D + d) = v;
    }
  }
}

// one WORKGROUP per query column q: its 4 waves take the frames f = w, w+4, ... four at a time (the arg-max row gathers of
// four frames are in flight together), and the four partial rows are added in wave order through LDS -- a fixed order.
// (The first version walked the F frames of a column serially in one wave: F dependent L2 round trips, 63 us at C5.)
template <int MC>
__device__ __forceinline__ void sim_bwd_dw_body(int q, const float *__restrict__ dS, const int64_t *__restrict__ D_ind,
                                                const float *__restrict__ V, const int32_t *__restrict__ ent_len,
                                                int F, int Nb, int Ne, int Q, int D,
                                                const float *__restrict__ grad_scale, float *__restrict__ dW) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [4][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = q / Ne, en = q - a * Ne;
  if (en >= ent_len[a]) {                        // masked slot: dS is 0 for every frame
    for (int d = threadIdx.x * 4; d < D; d += 1024) *reinterpret_cast<f32x4 *>(dW + (size_t)q * D + d) = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  f32x4 acc[MC];
#pragma unroll
  for (int c = 0; c < MC; c++) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int f0 = wave; f0 < F; f0 += 16) {
    float ds[4];
    const float *vr[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int f = f0 + 4 * u;
      ds[u] = f < F ? dS[(size_t)f * Q + q] : 0.f;
      const int ix = f < F ? (int)D_ind[(size_t)f * Q + q] : 0;
      vr[u] = V + ((size_t)(f < F ? f : 0) * Nb + ix) * D;
    }
    // all four rows are requested before the first is used; a frame with ds == 0 (inactive hinge, or beyond F) adds +-0
#pragma unroll
    for (int c = 0; c < MC; c++) {
      const int d = lane * 4 + c * 256;
      if (d < D) {
        f32x4 r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = *reinterpret_cast<const f32x4 *>(vr[u] + d);
#pragma unroll
        for (int u = 0; u < 4; u++) acc[c] += ds[u] * r[u];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < MC; c++) {
    const int d = lane * 4 + c * 256;
    if (d < D) *reinterpret_cast<f32x4 *>(&red[wave * D + d]) = acc[c];
  }
  __syncthreads();
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    f32x4 t = *reinterpret_cast<const f32x4 *>(&red[d]);
    t += *reinterpret_cast<const f32x4 *>(&red[D + d]);
    t += *reinterpret_cast<const f32x4 *>(&red[2 * D + d]);
    t += *reinterpret_cast<const f32x4 *>(&red[3 * D + d]);
    *reinterpret_cast<f32x4 *>(dW + (size_t)q * D + d) = grad_scale ? t * grad_scale[0] : t;
  }
}

// dV and dW in ONE launch: neither needs the other (both read dS and the arg-max), so as two launches on a stream the second
// only waited for the first.  Roles by workgroup index: the rows the clustering gradient lands on (one each), the query
// columns (one each), the other region rows (four each).
template <int MC>
__global__ __launch_bounds__(256, MC <= 2 ? 6 : 2) void sim_bwd_kernel(const float *__restrict__ dS, const int64_t *__restrict__ D_ind,
                                                      const float *__restrict__ V, const float *__restrict__ Wm,
                                                      const int32_t *__restrict__ ent_len, int F, int R, int Nb, int Ne, int Q, int D,
                                                      int train, int n_centries, const float *__restrict__ ws, LossWs L,
                                                      const float *__restrict__ pre_scale, const float *__restrict__ grad_scale,
                                                      float *__restrict__ dV, float *__restrict__ dW) {
  const int ncl = train ? (Nb < R ? Nb : R) : 0;           // clustering rows: the first workgroups (the longest ones)
  const int bid = blockIdx.x;
  if (bid < ncl)
    sim_bwd_dv_body<MC, 4>(bid, dS, D_ind, Wm, R, Nb, Q, D, train, n_centries, ws, L, pre_scale, grad_scale, dV);
  else if (bid < ncl + Q)
    sim_bwd_dw_body<MC>(bid - ncl, dS, D_ind, V, ent_len, F, Nb, Ne, Q, D, grad_scale, dW);
  else
    sim_bwd_dv_body<MC, 1>(bid - ncl - Q, dS, D_ind, Wm, R, Nb, Q, D, train, n_centries, ws, L, pre_scale, grad_scale, dV);
}

// ------------------------------------------------------------------------------------------------ embedding tails
__global__ __launch_bounds__(256) void dropout_tanh_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                           float scale, float *__restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
    if (mask) {
      const uchar4 m = reinterpret_cast<const uchar4 *>(mask)[i];
      v[0] = v[0] * (float)m.x * scale;
      v[1] = v[1] * (float)m.y * scale;
      v[2] = v[2] * (float)m.z * scale;
      v[3] = v[3] * (float)m.w * scale;
    }
    f32x4 o = {tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3])};
    reinterpret_cast<f32x4 *>(y)[i] = o;
  }
}

__global__ __launch_bounds__(256) void dropout_tanh_bwd_kernel(const float *__restrict__ go, const float *__restrict__ y,
                                                               const uint8_t *__restrict__ mask, float scale,
                                                               float *__restrict__ gi, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 g = reinterpret_cast<const f32x4 *>(go)[i];
    const f32x4 t = reinterpret_cast<const f32x4 *>(y)[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = g[k] * (1.0f - t[k] * t[k]);
    if (mask) {
      const uchar4 m = reinterpret_cast<const uchar4 *>(mask)[i];
      o[0] = o[0] * (float)m.x * scale;
      o[1] = o[1] * (float)m.y * scale;
      o[2] = o[2] * (float)m.z * scale;
      o[3] = o[3] * (float)m.w * scale;
    }
    reinterpret_cast<f32x4 *>(gi)[i] = o;
  }
}

// (the seeded keep rule, keep_elem(seed, i, thresh), lives in sim_common.h: the plane-emitting forms in simplanes.hip share it)
__global__ __launch_bounds__(256) void dropout_tanh_seeded_kernel(const float *__restrict__ x, uint64_t seed, uint32_t thresh,
                                                                  float scale, float *__restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = keep_elem(seed, (uint64_t)i * 4 + k, thresh) ? tanhf(v[k] * scale) : 0.f;
    reinterpret_cast<f32x4 *>(y)[i] = o;
  }
}

__global__ __launch_bounds__(256) void dropout_tanh_bwd_seeded_kernel(const float *__restrict__ go, const float *__restrict__ y,
                                                                      uint64_t seed, uint32_t thresh, float scale,
                                                                      float *__restrict__ gi, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 g = reinterpret_cast<const f32x4 *>(go)[i];
    const f32x4 t = reinterpret_cast<const f32x4 *>(y)[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; k++)
      o[k] = keep_elem(seed, (uint64_t)i * 4 + k, thresh) ? g[k] * (1.0f - t[k] * t[k]) * scale : 0.f;
    reinterpret_cast<f32x4 *>(gi)[i] = o;
  }
}

// BatchNorm1d over the rows of x [Q, D].  Workgroup = 32 feature columns (one 128-B line per row) x 8 row lanes: lane g sums the
// rows q = g, g + 8, ... in ascending order, the 8 partial sums are added in lane order (every thread of a column adds the same
// values in the same order, so all agree bit for bit) -- a fixed order, deterministic.  (The first version walked the Q rows of
// a column serially in one thread, three dependent passes: 0.7 ms at Q = 256, more than the detector's RPN stage.)
constexpr int BN_RL = 8;

__device__ __forceinline__ float bn_col_sum(float part, float (*sh)[33], int g, int c) {
  __syncthreads();                 // (the previous use of sh is over)
  sh[g][c] = part;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < BN_RL; k++) t += sh[k][c];
  return t;
}

__global__ __launch_bounds__(256) void bn_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                     const float *__restrict__ b, float *__restrict__ rmean,
                                                     float *__restrict__ rvar, float *__restrict__ y,
                                                     float *__restrict__ smean, float *__restrict__ sinv, int Q, int D,
                                                     int training, float momentum, float eps) {
  __shared__ float sh[BN_RL][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + c;
  const bool ok = d < D;
  const int dd = ok ? d : D - 1;
  float mean, inv;
  if (training) {
    float s = 0.f;
    for (int q = g; q < Q; q += BN_RL) s += x[(size_t)q * D + dd];
    mean = bn_col_sum(s, sh, g, c) / (float)Q;
    float v = 0.f;
    for (int q = g; q < Q; q += BN_RL) {
      const float t = x[(size_t)q * D + dd] - mean;
      v += t * t;
    }
    v = bn_col_sum(v, sh, g, c);
    const float var_b = v / (float)Q;
    inv = 1.0f / sqrtf(var_b + eps);
    if (g == 0 && ok) {
      if (rmean) {
        const float var_u = Q > 1 ? v / (float)(Q - 1) : var_b;
        rmean[d] = (1.0f - momentum) * rmean[d] + momentum * mean;
        rvar[d] = (1.0f - momentum) * rvar[d] + momentum * var_u;
      }
      if (smean) {
        smean[d] = mean;
        sinv[d] = inv;
      }
    }
  } else {
    mean = rmean[dd];
    inv = 1.0f / sqrtf(rvar[dd] + eps);
  }
  if (!ok) return;
  const float gm = w[d], bb = b[d];
  for (int q = g; q < Q; q += BN_RL) y[(size_t)q * D + d] = (x[(size_t)q * D + d] - mean) * inv * gm + bb;
}

__global__ __launch_bounds__(256) void bn_bwd_kernel(const float *__restrict__ gy, const float *__restrict__ x,
                                                     const float *__restrict__ w, const float *__restrict__ smean,
                                                     const float *__restrict__ sinv, float *__restrict__ gx,
                                                     float *__restrict__ gw, float *__restrict__ gb, int Q, int D, int accumulate) {
  __shared__ float sh[BN_RL][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int d = blockIdx.x * 32 + c;
  const bool ok = d < D;
  const int dd = ok ? d : D - 1;
  const float mean = smean[dd], inv = sinv[dd], gm = w[dd];
  float sg = 0.f, sgx = 0.f;
  for (int q = g; q < Q; q += BN_RL) {
    const float dy = gy[(size_t)q * D + dd];
    const float xh = (x[(size_t)q * D + dd] - mean) * inv;
    sg += dy;
    sgx += dy * xh;
  }
  sg = bn_col_sum(sg, sh, g, c);
  sgx = bn_col_sum(sgx, sh, g, c);
  if (!ok) return;
  if (g == 0) {
    gw[d] = accumulate ? gw[d] + sgx : sgx;
    gb[d] = accumulate ? gb[d] + sg : sg;
  }
  const float invQ = 1.0f / (float)Q;
  for (int q = g; q < Q; q += BN_RL) {
    const float dy = gy[(size_t)q * D + d];
    const float xh = (x[(size_t)q * D + d] - mean) * inv;
    gx[(size_t)q * D + d] = gm * inv * (dy - sg * invQ - xh * sgx * invQ);
  }
}

// out[j] (+)= sum_i x[i][j].  Workgroup = 64 columns (a 256-B piece of every row) x 16 row lanes (1024 threads); a row lane sums
// its rows in ascending order, the 16 partials are added in lane order through LDS: deterministic.  `idx` / `count` (device)
// restrict the sum to a list of rows: the gradient of the visual embedding is exactly zero on >= 85 % of its rows
// (nafae_nonzero_rows), so its bias gradient reads ~1-2 k rows instead of R = 8 192 ... 19 200.
__global__ __launch_bounds__(1024) void colsum_kernel(const float *__restrict__ x, float *__restrict__ out, int rows, int cols,
                                                      const int32_t *__restrict__ idx, const int32_t *__restrict__ count,
                                                      int accumulate) {
  __shared__ float part[16][65];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int n = idx ? (count[0] < rows ? count[0] : rows) : rows;
  float acc = 0.f;
  if (c < cols)
    for (int r = g; r < n; r += 16) acc += x[(size_t)(idx ? idx[r] : r) * cols + c];
  part[g][cl] = acc;
  __syncthreads();
  if (g == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += part[k][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// ------------------------------------------------------------------------------------------------ non-zero rows
// The gradient wrt the visual embedding is non-zero only on the rows some (frame, query) picked as its arg-max (plus
// the clustering rows): >= 85 % of the R rows are exactly zero.  rowflag + compact build the ascending list of the
// others on the device, and the weight-gradient GEMM contracts over that list only (nafae_gemm_tn_rows).
__global__ __launch_bounds__(256) void rowflag_kernel(const float *__restrict__ x, int rows, int cols, int *__restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  bool nz = false;
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (size_t)r * cols + c);
    nz = nz || v[0] != 0.f || v[1] != 0.f || v[2] != 0.f || v[3] != 0.f;
  }
  const unsigned long long any = __ballot(nz);
  if (lane == 0) flag[r] = any != 0ull;
}

// ascending list of the flagged rows: thread t owns the rows [t * per, (t + 1) * per); one block-wide exclusive scan of the
// per-thread counts (wave shuffles + 16 wave totals through LDS), then every thread writes its own indices: one pass, two
// barriers.  (The first version looped over the rows 1 024 at a time with three barriers per trip: up to 0.5 ms at R = 16 384.)
__global__ __launch_bounds__(1024) void compact_kernel(const int *__restrict__ flag, int rows, int *__restrict__ idx,
                                                       int *__restrict__ count) {
  __shared__ int wsum[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (rows + 1023) / 1024;
  const int r0 = threadIdx.x * per;
  int r1 = r0 + per;
  r1 = r1 < rows ? r1 : rows;
  int cnt = 0;
  for (int r = r0; r < r1; r++) cnt += flag[r] != 0;
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int off = incl - cnt;
  for (int w = 0; w < wave; w++) off += wsum[w];
  for (int r = r0; r < r1; r++)
    if (flag[r] != 0) idx[off++] = r;
  if (threadIdx.x == 1023) count[0] = off;
}

// ------------------------------------------------------------------------------------------------ optimiser step
// clip_grad_norm_ + Adam (model.py:773-774, :1077-1082) over ONE flat buffer: pass 1 = per-block sums of squares in
// a fixed order, pass 2 = every block re-adds the partials in the same order (bit-identical total on all blocks),
// derives the clip coefficient and applies torch.optim.Adam's update (L2 weight decay folded into the gradient).
constexpr int ADAM_NB = 256;

__global__ __launch_bounds__(256) void sumsq_kernel(const float *__restrict__ g, long n, float *__restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += g[i] * g[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, long n, float lr, float beta1, float beta2,
                                                   float eps, float wd, float max_norm, float bc1, float bc2_sqrt,
                                                   const float *__restrict__ partial, float *__restrict__ norm_out) {
  float tot = 0.f;
  for (int k = 0; k < ADAM_NB; k++) tot += partial[k];
  const float total_norm = sqrtf(tot);
  float coef = max_norm / (total_norm + 1e-6f);   // torch.nn.utils.clip_grad_norm_
  coef = coef > 1.0f ? 1.0f : coef;
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = total_norm;
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gi = g[i] * coef;
    g[i] = gi;                                    // the clipped gradient stays observable, as in the reference
    const float pi = p[i];
    gi = gi + wd * pi;
    const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);      // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

}  // namespace

extern "C" {

int nafae_nonzero_rows(const float *x, int rows, int cols, int32_t *flag_ws, int32_t *idx_out, int32_t *count_out,
                       void *stream) {
  if (!x || !flag_ws || !idx_out || !count_out || rows <= 0 || cols <= 0 || (cols & 3)) return NAFAE_EINVAL;
  hipLaunchKernelGGL(rowflag_kernel, dim3((rows + 3) / 4), dim3(256), 0, S(stream), x, rows, cols, flag_ws);
  hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(1024), 0, S(stream), flag_ws, rows, idx_out, count_out);
  return launched();
}

int nafae_adam_step(float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, float max_norm, int step, float *workspace,
                    float *total_norm_out, void *stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !workspace || n <= 0 || step <= 0) return NAFAE_EINVAL;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(sumsq_kernel, dim3(ADAM_NB), dim3(256), 0, S(stream), grads, (long)n, workspace);
  hipLaunchKernelGGL(adam_kernel, dim3(ADAM_NB), dim3(256), 0, S(stream), params, grads, exp_avg, exp_avg_sq, (long)n, lr, beta1,
                     beta2, eps, weight_decay, max_norm, bc1, bc2_sqrt, workspace, total_norm_out);
  return launched();
}

int nafae_sim_max_fwd_frames(const float *V, const float *W, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D,
                             float *S_max, int64_t *D_ind, void *stream) {
  if (!V || !W || !ent_len || !S_max || !D_ind) return NAFAE_EINVAL;
  if (Na <= 0 || F <= 0 || Nb <= 0 || Ne <= 0 || D <= 0 || (D & 3)) return NAFAE_EINVAL;
  using E = Engine<128, 32, 4, 1>;
  const int Q = Na * Ne;
  if (F > 65535) return NAFAE_ELIMIT;
  const size_t lds = (2 * E::STAGE + 128 * 33 + 2 * 8 * 32) * sizeof(float);
  NAFAE_TAG("sim_max_kernel (exact-fp32 first generation)");
  hipLaunchKernelGGL(sim_max_kernel, dim3((Q + 31) / 32, F), dim3(NTHREADS), lds, S(stream), V, W, ent_len, Nb, Ne, Q, D,
                     S_max, D_ind);
  return launched();
}

int nafae_sim_max_fwd(const float *V, const float *W, const int32_t *ent_len, int Na, int Ns, int Nb, int Ne, int D,
                      float *S_max, int64_t *D_ind, void *stream) {
  if (Na <= 0 || Ns <= 0) return NAFAE_EINVAL;
  return nafae_sim_max_fwd_frames(V, W, ent_len, Na * Ns, Nb, Na, Ne, D, S_max, D_ind, stream);
}

int64_t nafae_loss_workspace_bytes(int Na, int Ns, int Nb, int Ne, int D) {
  if (Na <= 0 || Ns <= 0 || Nb <= 0 || Ne <= 0 || D <= 0) return NAFAE_EINVAL;
  return (int64_t)(loss_ws(Na, Ns, Nb, Ne, D).total * sizeof(float));
}

int nafae_loss_fwd_bwd_ex(const float *S_max, const int64_t *D_ind, const float *V, const int32_t *ent_len, int Na,
                          int Ns, int Nb, int Ne, int D, float Delta, float vis_lam, int train, int max_live_cols,
                          float *loss_out, float *dS, void *workspace, void *stream) {
  if (!S_max || !D_ind || !ent_len || !loss_out || !workspace) return NAFAE_EINVAL;
  if (Na <= 0 || Ns <= 0 || Nb <= 0 || Ne <= 0 || D <= 0 || (D & 3)) return NAFAE_EINVAL;
  if (train && !V) return NAFAE_EINVAL;
  const LossWs L = loss_ws(Na, Ns, Nb, Ne, D);
  float *ws = reinterpret_cast<float *>(workspace);
  const int Q = Na * Ne;
  int Lcap = (max_live_cols < 0 || max_live_cols > Q) ? Q : max_live_cols;
  if (Lcap < 1) Lcap = 1;
  const size_t tail_lds = loss_tail_lds_bytes(Na, Ns, Ne, Lcap);
  bool cluster_done = false;
  const char *seg_env = nafae::experiment_env("NAFAE_LOSS_SEG");   // experiments build: 1 = always the per-segment kernels
  if (tail_lds <= 150 * 1024 && !(seg_env && seg_env[0] == '1')) {     // everything on chip (see loss_tail_lds_kernel)
    if (tail_lds > 64 * 1024 &&
        nafae::allow_dynamic_lds(reinterpret_cast<const void *>(loss_tail_lds_kernel), 150 * 1024) != NAFAE_OK)
      return NAFAE_ELAUNCH;
    const size_t clds = (size_t)(2 * Ns + 1) * D * sizeof(float);
    if (train && Ns <= 64 && tail_lds <= 64 * 1024 && clds <= 64 * 1024) {     // both terms in one launch
      NAFAE_TAG("loss_lds_cluster (tail + clustering term, one launch)");
      hipLaunchKernelGGL(loss_lds_cluster_kernel, dim3(1 + Na * Ne), dim3(1024), tail_lds > clds ? tail_lds : clds, S(stream), S_max,
                         D_ind, V, ent_len, Na, Ns, Ne, D, Delta, dS, ws, L, Lcap);
      cluster_done = true;
    } else {
      NAFAE_TAG("loss_tail_lds%s", train ? " + cluster_kernel" : "");
      hipLaunchKernelGGL(loss_tail_lds_kernel, dim3(1), dim3(1024), tail_lds, S(stream), S_max, ent_len, Na, Ns, Ne, Delta, dS, ws,
                         L, Lcap);
    }
  } else {
    const size_t fwd_lds = ((size_t)Na * Ns * Ne + 2 * (size_t)Na * Ne) * sizeof(float);
    const size_t bwd_lds = ((size_t)Na * Ns * Na + 4 * (size_t)Na * Ns) * sizeof(float);
    if (fwd_lds <= 150 * 1024 && bwd_lds <= 150 * 1024) {       // one workgroup per segment, two launches
      if (fwd_lds > 64 * 1024 &&
          nafae::allow_dynamic_lds(reinterpret_cast<const void *>(loss_seg_fwd_kernel), 150 * 1024) != NAFAE_OK)
        return NAFAE_ELAUNCH;
      if (bwd_lds > 64 * 1024 &&
          nafae::allow_dynamic_lds(reinterpret_cast<const void *>(loss_seg_bwd_kernel), 150 * 1024) != NAFAE_OK)
        return NAFAE_ELAUNCH;
      const size_t clds = (size_t)(2 * Ns + 1) * D * sizeof(float);
      if (train && Ns <= 64 && fwd_lds <= 64 * 1024 && clds <= 64 * 1024) {     // forward of the ranking term + clustering term
        NAFAE_TAG("loss_segf_cluster + loss_seg_bwd (per-segment tail)");
        hipLaunchKernelGGL(loss_segf_cluster_kernel, dim3(Na + Na * Ne), dim3(512), fwd_lds > clds ? fwd_lds : clds, S(stream),
                           S_max, D_ind, V, ent_len, Na, Ns, Ne, D, ws, L);
        cluster_done = true;
      } else {
        NAFAE_TAG("loss_seg_fwd + loss_seg_bwd%s", train ? " + cluster_kernel" : "");
        hipLaunchKernelGGL(loss_seg_fwd_kernel, dim3(Na), dim3(256), fwd_lds, S(stream), S_max, ent_len, Na, Ns, Ne, ws, L);
      }
      hipLaunchKernelGGL(loss_seg_bwd_kernel, dim3(Na), dim3(256), bwd_lds, S(stream), S_max, ent_len, Na, Ns, Ne, Delta, dS, ws, L);
    } else {
      NAFAE_TAG("loss_tail (global memory)%s", train ? " + cluster_kernel" : "");
      hipLaunchKernelGGL(loss_tail_kernel, dim3(1), dim3(1024), 0, S(stream), S_max, ent_len, Na, Ns, Ne, Delta, dS, ws, L);
    }
  }
  if (train && !cluster_done) {
    if (Ns > 64) return NAFAE_ELIMIT;
    const size_t lds = (size_t)(2 * Ns + 1) * D * sizeof(float);
    if (lds > 64 * 1024) {
      if (lds > 144 * 1024) return NAFAE_ELIMIT;
      if (nafae::allow_dynamic_lds(reinterpret_cast<const void *>(cluster_kernel), 144 * 1024) != NAFAE_OK) return NAFAE_ELAUNCH;
    }
    hipLaunchKernelGGL(cluster_kernel, dim3(Na * Ne), dim3(512), lds, S(stream), S_max, D_ind, V, ent_len, Na, Ns, Ne, D,
                       ws, L);
  }
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, S(stream), ws, L, Na, Ne, vis_lam, train, loss_out);
  return launched();
}

int nafae_loss_fwd_bwd(const float *S_max, const int64_t *D_ind, const float *V, const int32_t *ent_len, int Na,
                       int Ns, int Nb, int Ne, int D, float Delta, float vis_lam, int train, float *loss_out,
                       float *dS, void *workspace, void *stream) {
  return nafae_loss_fwd_bwd_ex(S_max, D_ind, V, ent_len, Na, Ns, Nb, Ne, D, Delta, vis_lam, train, -1, loss_out, dS, workspace,
                               stream);
}

int nafae_sim_bwd_frames(const float *dS, const int64_t *D_ind, const float *V, const float *W, const int32_t *ent_len,
                         int F, int Na, int Ns, int Nb, int Ne, int D, int cluster_rows, const void *workspace,
                         const float *pre_scale, const float *grad_scale, float *dV, float *dW, void *stream) {
  if (!dS || !D_ind || !V || !W || !ent_len || !dV || !dW) return NAFAE_EINVAL;
  if (F <= 0 || Na <= 0 || Ns <= 0 || Nb <= 0 || Ne <= 0 || D <= 0 || (D & 3)) return NAFAE_EINVAL;
  if (D > 256 * MAXCH) return NAFAE_ELIMIT;
  if (cluster_rows && !workspace) return NAFAE_EINVAL;
  const LossWs L = loss_ws(Na, Ns, Nb, Ne, D);
  const int Q = Na * Ne, R = F * Nb;
  const size_t dv_lds = 16384 + (size_t)3 * D * sizeof(float), dw_lds = (size_t)4 * D * sizeof(float);   // dV: 4 hit lists + [3][D]; dW: [4][D]
  const size_t bwd_lds = dv_lds > dw_lds ? dv_lds : dw_lds;
  const int ncl = cluster_rows ? (Nb < R ? Nb : R) : 0;
  // (MC = float4 chunks per lane: 2 covers D <= 512 in 50-odd registers, i.e. 8 waves per SIMD instead of 5 -- the rows' dependent
  // round trips are hidden by the number of rows in flight)
  NAFAE_TAG("sim_bwd<%d> (dV and dW, one launch)", D <= 512 ? 2 : MAXCH);
  if (D <= 512)
    hipLaunchKernelGGL(sim_bwd_kernel<2>, dim3(ncl + Q + (R + 3) / 4), dim3(256), bwd_lds, S(stream), dS, D_ind, V,
                       W, ent_len, F, R, Nb, Ne, Q, D, cluster_rows, Q * Ns, reinterpret_cast<const float *>(workspace), L,
                       pre_scale, grad_scale, dV, dW);
  else
    hipLaunchKernelGGL(sim_bwd_kernel<MAXCH>, dim3(ncl + Q + (R + 3) / 4), dim3(256), bwd_lds, S(stream), dS, D_ind,
                       V, W, ent_len, F, R, Nb, Ne, Q, D, cluster_rows, Q * Ns, reinterpret_cast<const float *>(workspace), L,
                       pre_scale, grad_scale, dV, dW);
  return launched();
}

int nafae_sim_bwd(const float *dS, const int64_t *D_ind, const float *V, const float *W, const int32_t *ent_len,
                  int Na, int Ns, int Nb, int Ne, int D, int train, const void *workspace, const float *pre_scale,
                  const float *grad_scale, float *dV, float *dW, void *stream) {
  if (Na <= 0 || Ns <= 0) return NAFAE_EINVAL;
  return nafae_sim_bwd_frames(dS, D_ind, V, W, ent_len, Na * Ns, Na, Ns, Nb, Ne, D, train, workspace, pre_scale, grad_scale,
                              dV, dW, stream);
}

int nafae_dropout_tanh(const float *x, const uint8_t *mask, float scale, float *y, int64_t n, void *stream) {
  if (!x || !y || n <= 0 || (n & 3)) return NAFAE_EINVAL;
  const long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(dropout_tanh_kernel, dim3(blocks), dim3(256), 0, S(stream), x, mask, scale, y, n4);
  return launched();
}

int nafae_dropout_tanh_bwd(const float *g_out, const float *y, const uint8_t *mask, float scale, float *g_in, int64_t n,
                           void *stream) {
  if (!g_out || !y || !g_in || n <= 0 || (n & 3)) return NAFAE_EINVAL;
  const long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(dropout_tanh_bwd_kernel, dim3(blocks), dim3(256), 0, S(stream), g_out, y, mask, scale, g_in, n4);
  return launched();
}

static inline uint32_t drop_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
}

int nafae_dropout_tanh_seeded(const float *x, uint64_t seed, float p, float *y, int64_t n, void *stream) {
  if (!x || !y || n <= 0 || (n & 3) || !(p >= 0.f) || !(p < 1.f)) return NAFAE_EINVAL;
  const long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(dropout_tanh_seeded_kernel, dim3(blocks), dim3(256), 0, S(stream), x, seed, drop_threshold(p),
                     1.0f / (1.0f - p), y, n4);
  return launched();
}

int nafae_dropout_tanh_bwd_seeded(const float *g_out, const float *y, uint64_t seed, float p, float *g_in, int64_t n,
                                  void *stream) {
  if (!g_out || !y || !g_in || n <= 0 || (n & 3) || !(p >= 0.f) || !(p < 1.f)) return NAFAE_EINVAL;
  const long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(dropout_tanh_bwd_seeded_kernel, dim3(blocks), dim3(256), 0, S(stream), g_out, y, seed, drop_threshold(p),
                     1.0f / (1.0f - p), g_in, n4);
  return launched();
}

int nafae_batchnorm_fwd(const float *x, const float *weight, const float *bias, float *running_mean,
                        float *running_var, float *y, float *save_mean, float *save_invstd, int Q, int D, int training,
                        float momentum, float eps, void *stream) {
  if (!x || !weight || !bias || !y || Q <= 0 || D <= 0) return NAFAE_EINVAL;
  if (!training && (!running_mean || !running_var)) return NAFAE_EINVAL;
  hipLaunchKernelGGL(bn_fwd_kernel, dim3((D + 31) / 32), dim3(256), 0, S(stream), x, weight, bias, running_mean,
                     running_var, y, save_mean, save_invstd, Q, D, training, momentum, eps);
  return launched();
}

int nafae_batchnorm_bwd_acc(const float *g_y, const float *x, const float *weight, const float *save_mean,
                            const float *save_invstd, float *g_x, float *g_weight, float *g_bias, int Q, int D, int accumulate,
                            void *stream) {
  if (!g_y || !x || !weight || !save_mean || !save_invstd || !g_x || !g_weight || !g_bias || Q <= 0 || D <= 0)
    return NAFAE_EINVAL;
  hipLaunchKernelGGL(bn_bwd_kernel, dim3((D + 31) / 32), dim3(256), 0, S(stream), g_y, x, weight, save_mean, save_invstd, g_x,
                     g_weight, g_bias, Q, D, accumulate);
  return launched();
}

int nafae_batchnorm_bwd(const float *g_y, const float *x, const float *weight, const float *save_mean,
                        const float *save_invstd, float *g_x, float *g_weight, float *g_bias, int Q, int D,
                        void *stream) {
  return nafae_batchnorm_bwd_acc(g_y, x, weight, save_mean, save_invstd, g_x, g_weight, g_bias, Q, D, 0, stream);
}

int nafae_colsum_acc(const float *x, float *out, int rows, int cols, int accumulate, void *stream) {
  if (!x || !out || rows <= 0 || cols <= 0) return NAFAE_EINVAL;
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64), dim3(1024), 0, S(stream), x, out, rows, cols, nullptr, nullptr,
                     accumulate);
  return launched();
}

int nafae_colsum(const float *x, float *out, int rows, int cols, void *stream) {
  return nafae_colsum_acc(x, out, rows, cols, 0, stream);
}

int nafae_colsum_rows(const float *x, const int32_t *idx, const int32_t *count, int max_rows, int cols, float *out,
                      int accumulate, void *stream) {
  if (!x || !out || !idx || !count || max_rows <= 0 || cols <= 0) return NAFAE_EINVAL;
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64), dim3(1024), 0, S(stream), x, out, max_rows, cols, idx, count,
                     accumulate);
  return launched();
}

}  // extern "C"
