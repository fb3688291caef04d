// simplanes.hip -- region x query similarity with MANY live query slots (C5 with every slot live: 512 columns), fourth generation,
// gfx950: the operands arrive as matrix-core PLANES written by their producer, so nothing is converted inside the kernel.
// (DVSA.forward, reference model.py:548-551, 580-583, 610-612.)
//
// What round 3's sim_frame_kernel (simfused.hip) measured: its staging waves convert fp32 -> bf16 hi / lo and ds_write the planes
// next to the MFMA waves of the same SIMD; the two share the SIMD's vector issue, and a k-loop trip took 3 400 cycles beside
// 2 400 of MFMA time (33 us of a 60 us kernel at C5; 20.6 us with the conversion compiled out).  Each (frame, column group)
// workgroup repeated that conversion for its frame's 614 KB.  Here:
//
//   planes  V and W are split ONCE, in the epilogue of the kernel that produces them (VisEbd's / WordEbd's dropout + tanh:
//           nafae_dropout_tanh_planes / _seeded_planes; nafae_sim_planes is the stand-alone form for any fp32 matrix), into the
//           layout the kernel stages: per row and 128-byte line either [hi k 0..31 | lo k 0..31] as bf16 (kind BF16X3: the same
//           4 bytes per element as fp32) or k 0..63 as fp16 (kind F16: half the bytes), plus two fp32 statistics per row
//           (max |x|, l2 norm) that the filter margin needs.
//   stage   waves 4-7 move whole lines global -> LDS by LDS-DMA (global_load_lds_dwordx4: 8 rows x one line per instruction; the
//           XOR swizzle sits on the per-lane SOURCE slot because DMA destinations are wave-linear) -- no VGPR, no VALU, no ds_write.
//   filter  waves 0-3 (one per SIMD): RW x CW accumulator tiles of 32 x 32 per wave, bf16x3 (hi*hi + hi*lo + lo*hi) or ONE fp16
//           product per k-step.  The product only FILTERS: pass 1 takes the column maxima of the filter values, pass 2 lists every row
//           whose filter value lies within `margin` of its column's maximum (the maximum included), and all listed rows are
//           re-evaluated as exact fp32 dot products from the fp32 operands by the whole workgroup (a flat work list, BATCH rows in
//           flight per wave); torch.max's rules decide (ties -> smaller row, NaN first).  A column whose list overflows, or that
//           saw a NaN / Inf, goes on the slow list: exact fp32 over all rows.  The margin is rigorous for ANY operand values:
//             bf16x3  2^-14 D max|V_frame| |w|_inf + 2^-11 |score|          (round 3's bound with the per-column W statistic)
//             fp16    (2.0e-3 + 2.4e-7 D) |v|_2,max |w|_2 + 1.2e-7 sqrt(D) (|v|_2,max + |w|_2)
//                     -- twice { operand rounding 2^-11 |x| + 2^-25 on both sides (Cauchy-Schwarz over the row), fp32 accumulation
//                     D 2^-24 } plus the rounding of the two exact evaluations being compared; an operand beyond the fp16 range
//                     becomes Inf, Inf * 0 marks the column for the slow list.
//
// Grid, LDS image, fragment mapping and the exact / slow phases follow sim_frame_kernel; one workgroup = (frame, group of 64 CW
// live columns) streams all rows of its frame, so max, exact evaluation and output happen inside the launch.
// Algorithmic bytes: 4*D*(R+Q) + 12*F*Q.  No float atomics; every reduction runs in a fixed order (the LDS integer
// atomics only allocate list slots: the decision is independent of the list order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bf16_tile.h"
#include "hip_util.h"
#include "sim_common.h"

using namespace nafae;
using namespace nafae_sim;

#ifdef NAFAE_EXPERIMENTS
__device__ unsigned long long nafae_simplanes_stamps[8 * 8192];
#define PSTAMP(k)                                                                                    \
  do {                                                                                               \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                                \
      nafae_simplanes_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64();      \
  } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KIND_BF16X3 = NAFAE_SIMPLANES_BF16X3, KIND_F16 = NAFAE_SIMPLANES_F16;

// ---------------------------------------------------------------------------------------------------- producers
// One wave per row (grid-stride), lane l takes the float4s l, l + 64, ... of the row.  MODE 0: planes of X as it is; 1: y =
// tanh(x * mask * scale) (nafae_dropout_tanh); 2: y = keep(seed, i) ? tanh(x * scale) : 0 (nafae_dropout_tanh_seeded) -- the same
// element arithmetic as those kernels, so y is bit-identical to theirs.  stats[row] = (max |y|, sqrt(sum y^2)); the sum runs over
// the lane's elements in ascending order, then over the lanes by wave_sum: fixed order.
template <int MODE>
__global__ __launch_bounds__(256) void planes_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask, float scale,
                                                     uint64_t seed, uint32_t thresh, float *__restrict__ y, int rows, int D, int kind,
                                                     unsigned char *__restrict__ planes, float *__restrict__ stats,
                                                     const float *__restrict__ x2, int rows2, unsigned char *__restrict__ planes2,
                                                     float *__restrict__ stats2) {
  const int lane = threadIdx.x & 63;
  const int n4 = D >> 2;
  // (MODE 0 only: a second matrix x2 [rows2, D] -> planes2 / stats2 in the same launch: V and W of one similarity call)
  for (int row0 = blockIdx.x * 4 + (threadIdx.x >> 6); row0 < rows + rows2; row0 += gridDim.x * 4) {
    int row = row0;
    if (MODE == 0 && row0 >= rows) {
      row = row0 - rows;
      x = x2;
      planes = planes2;
      stats = stats2;
    }
    float amax = 0.f, ss = 0.f;
    for (int j = lane; j < n4; j += 64) {
      const size_t i4 = (size_t)row * n4 + j;
      f32x4 v = reinterpret_cast<const f32x4 *>(x)[i4];
      if (MODE == 1) {
        if (mask) {
          const uchar4 m = reinterpret_cast<const uchar4 *>(mask)[i4];
          v[0] = v[0] * (float)m.x * scale;
          v[1] = v[1] * (float)m.y * scale;
          v[2] = v[2] * (float)m.z * scale;
          v[3] = v[3] * (float)m.w * scale;
        }
        v = f32x4{tanhf(v[0]), tanhf(v[1]), tanhf(v[2]), tanhf(v[3])};
      } else if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = keep_elem(seed, (uint64_t)i4 * 4 + k, thresh) ? tanhf(v[k] * scale) : 0.f;
      }
      if (MODE != 0) reinterpret_cast<f32x4 *>(y)[i4] = v;
      amax = absmax4(amax, v);
      ss = fmaf(v[0], v[0], ss);
      ss = fmaf(v[1], v[1], ss);
      ss = fmaf(v[2], v[2], ss);
      ss = fmaf(v[3], v[3], ss);
      const int k0 = j * 4;
      if (kind == KIND_BF16X3) {
        bf16x4 hi, lo;
        split4(v, hi, lo);
        unsigned char *line = planes + (size_t)row * D * 4 + (size_t)(k0 >> 5) * 128 + (k0 & 31) * 2;
        *reinterpret_cast<bf16x4 *>(line) = hi;
        *reinterpret_cast<bf16x4 *>(line + 64) = lo;
      } else {
        // round-to-nearest-even conversions (v_cvt_f16_f32 in the default float mode; NOT the round-toward-zero pack form)
        const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<f16x4 *>(planes + (size_t)row * D * 2 + (size_t)k0 * 2) = h;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    ss = wave_sum(ss);
    if (lane == 0) {
      stats[(size_t)row * 2] = amax;
      stats[(size_t)row * 2 + 1] = sqrtf(ss);
    }
  }
}

// ---------------------------------------------------------------------------------------------------- the frame kernel
constexpr int PL_MAXC = 16;     // listed candidates per column before the column takes the slow path
constexpr int PL_LANEC = 4;     // ... per (lane, column) in one pass (byte-packed ids)
constexpr int FR_MAXT = 2;      // D <= 512: float4 pieces per lane of an exact dot product
constexpr int PL_BATCH = 8;     // exact dot products in flight per wave

// one accumulator element against the column's threshold: cnt += hit, ids = hit ? (ids << 8 | ID) : ids.  Four vector
// instructions, branch-free (as a ternary hipcc builds exec-masked branches around 160 of these).  Measured and NOT adopted: a
// scalar skip (v_cmp; s_cbranch_vccz over the other three -- four elements in five have no hit in any lane): the branch waits
// for the compare's VCC every time, the scan took 5.8 us instead of 3.3 at C5.
template <int ID>
__device__ __forceinline__ void cand_push(float v, float thr, int &cnt, unsigned &ids) {
  unsigned tmp;
  if constexpr (ID <= 64) {
    asm volatile(
        "v_lshl_or_b32 %2, %1, 8, %5\n\t"
        "v_cmp_ge_f32 vcc, %3, %4\n\t"
        "v_cndmask_b32_e32 %1, %1, %2, vcc\n\t"
        "v_addc_co_u32 %0, vcc, 0, %0, vcc"
        : "+v"(cnt), "+v"(ids), "=&v"(tmp)
        : "v"(v), "v"(thr), "n"(ID)
        : "vcc");
  } else {
    asm volatile(
        "v_lshl_or_b32 %2, %1, 8, %5\n\t"
        "v_cmp_ge_f32 vcc, %3, %4\n\t"
        "v_cndmask_b32_e32 %1, %1, %2, vcc\n\t"
        "v_addc_co_u32 %0, vcc, 0, %0, vcc"
        : "+v"(cnt), "+v"(ids), "=&v"(tmp)
        : "v"(v), "v"(thr), "s"(ID)
        : "vcc");
  }
}

// v = OFF < lim ? v : ninf (a row beyond the frame), two vector instructions
template <int OFF>
__device__ __forceinline__ void mask_row(float &v, int lim, float ninf) {
  asm volatile("v_cmp_gt_i32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %3, %0, vcc" : "+v"(v) : "v"(lim), "n"(OFF), "v"(ninf) : "vcc");
}

struct PlanesLds {
  int qmap, wst, wam, cmax, cflag, ccnt, clist, vst, prefix, total;
};
template <int RT, int GC, int WR, int NST>
__host__ __device__ inline PlanesLds planes_lds(int Na) {
  PlanesLds o;
  int p = NST * (RT + GC) * 128;               // [NST stages][(RT + GC) rows][128 B]; after the k-loops: the exact phase's lists
  o.qmap = p;   p += GC * 4;                   // live column -> query row (-1: beyond the live count)
  o.wst = p;    p += GC * 4;                   // the column's W statistic the margin uses
  o.wam = p;    p += GC * 4;                   // the column's max |w| (fp16 planes: range check)
  o.cmax = p;   p += WR * GC * 4;              // [row part][column] filter maximum of the current super-tile
  o.cflag = p;  p += GC * 4;                   // bit 0: NaN / Inf seen, bit 1: more than PL_LANEC hits in one lane
  o.ccnt = p;   p += GC * 4;                   // listed candidates
  o.clist = p;  p += GC * PL_MAXC * 4;         // their rows
  o.vst = p;    p += 16;                       // the frame's V statistics (max over its rows): [0] the margin's, [1] max |v|
  o.prefix = p; p += ((Na + 1) * 4 + 15) & ~15;
  o.total = p;
  return o;
}

// grid ceil(F/8)*8*G workgroups of 512 threads (one per CU); workgroup = (frame f, column group g); the G workgroups of a frame
// share an XCD (blockIdx % 8), so the second to G-th pass over the frame's planes can hit its L2.
// WR: how the four MFMA waves split the tile -- 2: two row halves x two column halves (many live columns: 64 CW per workgroup);
//     4: four row quarters x ONE 32-column block (few live columns, round 4: the frame's rows stream ONCE, against 32 columns).
// NST: stages of the LDS ring = lines of look-ahead + 1.  With two stages a trip costs DMA issue + the FULL latency of its line
//     (1.06 us per 57 KB line at C5); where the stage is small enough for more (the narrow form: 20 KB at 128 rows, 37 KB at 256),
//     the lines of the next NST - 2 trips are already in flight and the wait is counted (vmcnt).
template <int RW, int CW, int KIND, int WR = 2, int NST = 2>
__global__ __launch_bounds__(512) void sim_planes_kernel(const float *__restrict__ V, const float *__restrict__ Wm,
                                                         const unsigned char *__restrict__ Vp, const unsigned char *__restrict__ Wp,
                                                         const float *__restrict__ vstat, const float *__restrict__ wstat,
                                                         const int32_t *__restrict__ ent_len, int F, int Nb, int Na, int Ne, int D,
                                                         int G, float *__restrict__ S_max, int64_t *__restrict__ D_ind, int dbg) {
  constexpr int WC = 4 / WR;                 // column parts among the MFMA waves
  static_assert(WR == 2 || WR == 4, "row parts");
  constexpr int RT = WR * RW * 32;           // rows per super-tile
  constexpr int GC = WC * 32 * CW;           // live columns per workgroup
  constexpr int STAGE = (RT + GC) * 128;
  constexpr int CK = KIND == KIND_F16 ? 64 : 32;      // k per 128-byte line
  constexpr int ST = KIND == KIND_F16 ? 1 : 0;        // which row statistic the margin uses (1: l2 norm, 0: max |x|)
  (void)dbg;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const PlanesLds lo = planes_lds<RT, GC, WR, NST>(Na);
  int *qmap = reinterpret_cast<int *>(smem + lo.qmap);
  float *wst = reinterpret_cast<float *>(smem + lo.wst);
  float *wam = reinterpret_cast<float *>(smem + lo.wam);
  float *cmax = reinterpret_cast<float *>(smem + lo.cmax);
  int *cflag = reinterpret_cast<int *>(smem + lo.cflag);
  int *ccnt = reinterpret_cast<int *>(smem + lo.ccnt);
  int *clist = reinterpret_cast<int *>(smem + lo.clist);
  int *vsti = reinterpret_cast<int *>(smem + lo.vst);
  int *prefix = reinterpret_cast<int *>(smem + lo.prefix);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b8 = blockIdx.x >> 3;
  const int f = (b8 / G) * 8 + (blockIdx.x & 7), g = b8 % G;
  if (f >= F) return;
  const int Q = Na * Ne;
  const int nch = D / CK;
  const int rowbytes = nch * 128;

  // The staging waves request the V part of the first NST - 1 lines before anything else: it depends on the launch arguments only,
  // and the prologue below (prefix sums, column map, the frame's statistics: 1.0-1.3 us) then runs under the loads' latency.
  constexpr int NV = RT / 32, NW = GC / 32;    // DMAs per staging wave and line: instruction n = j * 4 + sw covers stage rows [8n, 8n + 8)
  constexpr int PER = NV + NW;
  const unsigned char *Vpf = Vp + (size_t)f * Nb * rowbytes;
  auto v_offsets = [&](int rt, unsigned (&voff)[NV]) {
    const int sw = wave - 4, lr8 = lane >> 3, ls = lane & 7;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int sr = (j * 4 + sw) * 8 + lr8;
      int row = rt * RT + sr;
      row = row < Nb ? row : Nb - 1;           // (a row beyond the frame re-reads its last row; masked before the scan)
      voff[j] = (unsigned)row * (unsigned)rowbytes + (unsigned)((ls ^ frame_swz(sr)) << 4);
    }
  };
  auto issue_v = [&](int ci, const unsigned (&voff)[NV]) {       // V rows of line ci -> stage ci % NST
    unsigned char *st = smem + (ci % NST) * STAGE + (wave - 4) * 1024;
#pragma unroll
    for (int j = 0; j < NV; j++) lds_dma16(Vpf + (size_t)ci * 128, voff[j], st + j * 4096);
  };
  if (wave >= 4) {
    unsigned voff[NV];
    v_offsets(0, voff);
    for (int ci = 0; ci < NST - 1 && ci < nch; ci++) issue_v(ci, voff);
  }

  build_prefix(ent_len, Na, Ne, prefix);
  if (tid < GC) {
    cflag[tid] = 0;
    ccnt[tid] = 0;
  }
  if (tid < 2) vsti[tid] = 0;
  __syncthreads();
  const int Ql = prefix[Na];
  if (g == 0) {      // masked slots of this frame: (0, 0) (model.py:551); live slots beyond the launch (bound too small): NaN
    for (int q = tid; q < Q; q += 512) {
      const int a = q / Ne, e = q - a * Ne;
      const int l = prefix[a + 1] - prefix[a];
      if (e >= l) {
        S_max[(size_t)f * Q + q] = 0.f;
        D_ind[(size_t)f * Q + q] = 0;
      } else if (prefix[a] + e >= G * GC) {
        S_max[(size_t)f * Q + q] = NAN;
        D_ind[(size_t)f * Q + q] = 0;
      }
    }
  }
  if (g * GC >= Ql) {                        // over-provisioned column group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (no LDS-DMA may be in flight when the workgroup's LDS is released)
    return;
  }
  if (tid < GC) {
    const int c = g * GC + tid;
    int q = -1;
    if (c < Ql) {
      const int a = find_seg(prefix, Na, c);
      q = a * Ne + (c - prefix[a]);
    }
    qmap[tid] = q;
    wst[tid] = q >= 0 ? wstat[(size_t)q * 2 + ST] : 0.f;
    wam[tid] = q >= 0 ? wstat[(size_t)q * 2] : 0.f;
  }
  {   // the frame's V statistic: statistics are >= 0 (or NaN, whose bit pattern compares above every number: the margin then is
      // NaN and every column takes the slow path), so an integer max over the bit patterns is the float max
    int m = 0, m2 = 0;
    for (int r = tid; r < Nb; r += 512) {
      const float2 st2 = *reinterpret_cast<const float2 *>(vstat + ((size_t)f * Nb + r) * 2);
      const int b = __float_as_int(ST ? st2.y : st2.x) & 0x7fffffff, b2 = __float_as_int(st2.x) & 0x7fffffff;
      m = b > m ? b : m;
      m2 = b2 > m2 ? b2 : m2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int y = __shfl_xor(m, o), y2 = __shfl_xor(m2, o);
      m = y > m ? y : m;
      m2 = y2 > m2 ? y2 : m2;
    }
    if (lane == 0) {
      atomicMax(vsti, m);
      atomicMax(vsti + 1, m2);
    }
  }
  __syncthreads();
  const float vst = __int_as_float(vsti[0]), vam = __int_as_float(vsti[1]);
  (void)vam;
  const float *Vf = V + (size_t)f * Nb * D;
  const int nsuper = (Nb + RT - 1) / RT;
  PSTAMP(0);

  if (wave >= 4) {
    // ================================================================ staging waves: LDS-DMA only
    // (Measured against register staging -- global_load_dwordx4 -> VGPRs -> ds_write_b128, two chunks of loads in flight, no
    // conversion: the k-loop at C5 took 9.9 us instead of 8.5 (fp16) and 23.3 instead of 21.7 (bf16x3): the 57 KB of ds_write_b128
    // per chunk compete with the MFMA waves' fragment reads for the LDS, and the loads' latency was not what bounded the DMA
    // form.  With the MFMAs compiled out the DMA loop alone runs at 0.7 us per chunk = 82 GB/s into one CU.)
    const int sw = wave - 4;
    const int lr8 = lane >> 3, ls = lane & 7;
    unsigned woff[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) {
      const int sc = (j * 4 + sw) * 8 + lr8;
      const int q = qmap[sc];
      woff[j] = (unsigned)(q >= 0 ? q : 0) * (unsigned)rowbytes + (unsigned)((ls ^ frame_swz(sc)) << 4);
    }
    for (int rt = 0; rt < nsuper; rt++) {
      unsigned voff[NV];
      v_offsets(rt, voff);
      auto issue_w = [&](int ci) {
        unsigned char *st = smem + (ci % NST) * STAGE + sw * 1024;
#pragma unroll
        for (int j = 0; j < NW; j++) lds_dma16(Wp + (size_t)ci * 128, woff[j], st + RT * 128 + j * 4096);
      };
      auto issue = [&](int ci) {               // line ci -> stage ci % NST
        issue_v(ci, voff);
        issue_w(ci);
      };
      // (the stages are free: their last readers passed the previous super-tile's last trip barrier)
      // (super-tile 0: the V rows of these lines were requested at the top of the kernel; all of them have to land with line 0's W
      // rows, as the counter sees them in issue order -- they have had the prologue's time)
      for (int ci = 0; ci < NST - 1 && ci < nch; ci++) {
        if (rt > 0) issue_v(ci, voff);
        issue_w(ci);
      }
      if (rt > 0) lds_barrier();               // (X1 of the previous super-tile: the MFMA waves exchange their column maxima)
      if (NST > 2 && nch >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();                           // line 0 has landed
      PSTAMP(1);
      for (int ci = 0; ci < nch; ci++) {       // trip ci: the MFMA waves compute line ci; request line ci + NST - 1, wait for line ci + 1
        if (ci + NST - 1 < nch) {
          issue(ci + NST - 1);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PER) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier();
      }
      PSTAMP(2);
    }
    lds_barrier();                             // X1 of the last super-tile
  } else {
    // ================================================================ MFMA waves
    const int lr = lane & 31, h = lane >> 5;
    const int rh = WR == 4 ? wave : (wave >> 1) & 1, ch = WR == 4 ? 0 : wave & 1;
    const int aswz = frame_swz(lr);
    const int a_base = (rh * RW * 32 + lr) * 128;
    const int b_base = (RT + ch * CW * 32 + lr) * 128;
    int fo[4];                                 // byte offset of this lane's 16-B piece inside its row, per slot pair
    if (KIND == KIND_F16) {
#pragma unroll
      for (int t = 0; t < 4; t++) fo[t] = ((2 * t + h) ^ aswz) << 4;                       // k-step t: k 16t + 8h ..
    } else {
#pragma unroll
      for (int pl = 0; pl < 2; pl++)
#pragma unroll
        for (int t = 0; t < 2; t++) fo[pl * 2 + t] = ((pl * 4 + 2 * t + h) ^ aswz) << 4;   // [plane][k-step]
    }
    float bmr[CW];                             // running column maximum of the filter values over the super-tiles
#pragma unroll
    for (int cb = 0; cb < CW; cb++) bmr[cb] = -INFINITY;
    for (int rt = 0; rt < nsuper; rt++) {
      f32x16 acc[RW][CW];
#pragma unroll
      for (int rb = 0; rb < RW; rb++)
#pragma unroll
        for (int cb = 0; cb < CW; cb++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[rb][cb][r] = 0.f;
      lds_barrier();                           // chunk 0 is in its stage
      PSTAMP(1);
      for (int ci = 0; ci < nch; ci++) {
        const unsigned char *st = smem + (ci % NST) * STAGE;
#ifdef NAFAE_EXPERIMENTS
        if (dbg & 2) {                         // timing experiment: no fragment reads, no MFMAs
          lds_barrier();
          continue;
        }
#endif
        if (KIND == KIND_F16) {
#pragma unroll
          for (int t = 0; t < 4; t++) {
            f16x8 b[CW];
#pragma unroll
            for (int cb = 0; cb < CW; cb++) b[cb] = *reinterpret_cast<const f16x8 *>(st + b_base + cb * 4096 + fo[t]);
#pragma unroll
            for (int rb = 0; rb < RW; rb++) {
              const f16x8 a = *reinterpret_cast<const f16x8 *>(st + a_base + rb * 4096 + fo[t]);
#pragma unroll
              for (int cb = 0; cb < CW; cb++) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[cb], acc[rb][cb], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < 2; t++) {
            bf16x8 bhi[CW], blo[CW];
#pragma unroll
            for (int cb = 0; cb < CW; cb++) {
              bhi[cb] = *reinterpret_cast<const bf16x8 *>(st + b_base + cb * 4096 + fo[t]);
              blo[cb] = *reinterpret_cast<const bf16x8 *>(st + b_base + cb * 4096 + fo[2 + t]);
            }
#pragma unroll
            for (int rb = 0; rb < RW; rb++) {
              const bf16x8 ahi = *reinterpret_cast<const bf16x8 *>(st + a_base + rb * 4096 + fo[t]);
              const bf16x8 alo = *reinterpret_cast<const bf16x8 *>(st + a_base + rb * 4096 + fo[2 + t]);
#pragma unroll
              for (int cb = 0; cb < CW; cb++) {
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi[cb], acc[rb][cb], 0, 0, 0);
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo[cb], acc[rb][cb], 0, 0, 0);
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi[cb], acc[rb][cb], 0, 0, 0);
              }
            }
          }
        }
        lds_barrier();
      }
      PSTAMP(2);
      // ---- pass 1: this lane's column (cb) holds 16 rows per 32-row block, element (rb, r) = row wbase + 32 rb + (r & 3) +
      // 8 (r >> 2) + 4 h.  Rows beyond the frame were staged as copies of its last row: where a wave's blocks reach beyond the
      // frame (uniform: the last super-tile's second row half) their elements become -inf first -- two instructions per element
      // through inline asm (as C++ hipcc hoists the 160 row-validity compares into SGPR masks and spills them).  Left in, a last row
      // that wins a column would be listed once per copy and overflow the lane's list: the slow path for a third of the workgroups
      // (measured: +30 us).
      const int wbase = rt * RT + rh * RW * 32;
      if (wbase + RW * 32 > Nb) {
        const float ninf = -INFINITY;
        unroll_blocks<RW>([&](auto rb_tag) {
          constexpr int rb = decltype(rb_tag)::value;
          const int lim = Nb - (wbase + rb * 32) - 4 * h;      // element r is a real row iff (r & 3) + 8 (r >> 2) < lim
          unroll_blocks<16>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
#pragma unroll
            for (int cb = 0; cb < CW; cb++) {
              float v = acc[rb][cb][r];
              mask_row<(r & 3) + 8 * (r >> 2)>(v, lim, ninf);
              acc[rb][cb][r] = v;
            }
          });
        });
      }
      // Column maximum of the filter values, and their SUM: NaN as soon as one element is NaN (or +Inf meets a masked -inf), +Inf
      // when one is +Inf -- the column then takes the exact slow path (so does a finite sum that overflows).  A -Inf filter value
      // alone is not flagged here (the masked rows are -inf): among real rows it only arises from an fp16 plane that overflowed,
      // which pass 2 catches from the operands' max |x| statistics.  (The order mask -> one loop matters to hipcc: with the NaN
      // test taken before the masking, or the sum as v_pk_add_f32 pairs, it spilled 380 registers.)
#pragma unroll
      for (int cb = 0; cb < CW; cb++) {
        float m = -INFINITY, sum = 0.f;
#pragma unroll
        for (int rb = 0; rb < RW; rb++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const float v = acc[rb][cb][r];
            sum += v;
            m = fmaxf(m, v);
          }
        m = fmaxf(m, __shfl_xor(m, 32));
        const int c = (ch * CW + cb) * 32 + lr;
        if (sum != sum || sum == INFINITY) atomicOr(&cflag[c], 1);
        if (h == 0) cmax[rh * GC + c] = m;
      }
      lds_barrier();                           // X1: both row halves have written their maxima
      // ---- pass 2: list the rows within the margin of the (running) column maximum
#pragma unroll
      for (int cb = 0; cb < CW; cb++) {
        const int c = (ch * CW + cb) * 32 + lr;
        float tmx = cmax[c];
#pragma unroll
        for (int p2 = 1; p2 < WR; p2++) tmx = fmaxf(tmx, cmax[p2 * GC + c]);
        bmr[cb] = fmaxf(bmr[cb], tmx);
        const float bm = bmr[cb];
        float margin;
        if (KIND == KIND_F16) {
          const float nw = wst[c];
          margin = (2.0e-3f + 2.4e-7f * (float)D) * vst * nw + 1.2e-7f * sqrtf((float)D) * (vst + nw);
          // an operand beyond the fp16 range is +-Inf in its plane: nothing the filter says about this column can be trusted
          if (!(vam <= 65504.f) || !(wam[c] <= 65504.f)) margin = INFINITY;
        } else {
          margin = 6.103515625e-05f * (float)D * vst * wst[c] + 4.8828125e-04f * fabsf(bm);
        }
        if (!(margin < INFINITY)) atomicOr(&cflag[c], 1);      // NaN / Inf statistics: exact over all rows
        const float thr = bm - margin;
        int cnt = 0;
        unsigned ids = 0;
        unroll_blocks<RW>([&](auto rb_tag) {
          constexpr int rb = decltype(rb_tag)::value;
          unroll_blocks<16>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            cand_push<rb * 16 + r>(acc[rb][cb][r], thr, cnt, ids);
          });
        });
        if (cnt > 0) {
          if (cnt > PL_LANEC) {
            atomicOr(&cflag[c], 2);
          } else {
            int rows_[PL_LANEC], nv = 0;
#pragma unroll
            for (int k = 0; k < PL_LANEC; k++) {
              const int id = (int)((ids >> (8 * k)) & 0xffu);
              const int row = wbase + (id >> 4) * 32 + (id & 3) + 8 * ((id & 15) >> 2) + 4 * h;
              rows_[k] = row;
              nv += (k < cnt && row < Nb) ? 1 : 0;             // (a copy of the frame's last row is not a row)
            }
            if (nv > 0) {
              int pos = atomicAdd(&ccnt[c], nv);
#pragma unroll
              for (int k = 0; k < PL_LANEC; k++)
                if (k < cnt && rows_[k] < Nb) {
                  if (pos < PL_MAXC) clist[c * PL_MAXC + pos] = rows_[k];
                  pos++;
                }
            }
          }
        }
      }
      PSTAMP(3);
    }
  }

  // ---- exact fp32.  Work items (column, row): item c < GC is column c's FIRST listed row (a fixed place: no scan), the other
  // listed rows (about one column in nine has a second one with the fp16 filter) are appended behind them.
  __syncthreads();                             // lists complete; the stages are free: reuse them as scratch
  int *itemc = reinterpret_cast<int *>(smem);                          // [GC + GC * PL_MAXC] column of an item (-1: none)
  int *itemr = itemc + GC + GC * PL_MAXC;                              // ... its row
  float *res = reinterpret_cast<float *>(itemr + GC + GC * PL_MAXC);   // ... its exact score
  int *nextra = reinterpret_cast<int *>(res + GC + GC * PL_MAXC);      // [1] items behind the first GC
  int *nslow = nextra + 1;                                             // [1]
  int *slowc = nslow + 1;                                              // [GC]
  float2 *sbest = reinterpret_cast<float2 *>(slowc + GC);              // [8] (8-byte aligned: the counts above sum to an even number)
  static_assert((3 * (GC + GC * PL_MAXC) + 2 + GC) % 2 == 0, "sbest alignment");
  if (tid == 0) {
    nextra[0] = 0;
    nslow[0] = 0;
  }
  __syncthreads();
  int my_n = 0;
  if (tid < GC) {
    const int c = tid;
    const int cn = ccnt[c];
    const bool dead = qmap[c] < 0;
    const bool slow = !dead && (cflag[c] != 0 || cn > PL_MAXC || cn == 0);
    my_n = (dead || slow) ? 0 : cn;
    itemc[c] = my_n > 0 ? c : -1;
    itemr[c] = my_n > 0 ? clist[c * PL_MAXC] : 0;
    if (my_n > 1) {
      const int o = GC + atomicAdd(nextra, my_n - 1);
      for (int k = 1; k < my_n; k++) {
        itemc[o + k - 1] = c;
        itemr[o + k - 1] = clist[c * PL_MAXC + k];
      }
    }
    if (slow) slowc[atomicAdd(nslow, 1)] = c;
  }
  __syncthreads();
  PSTAMP(4);
  {
    const int total = GC + nextra[0];
    // wave w: the first items of its GC / 8 columns, then every eighth batch of the appended ones
    auto batch = [&](int i0, int i_end) {
      f32x4 wf[PL_BATCH][FR_MAXT], xf[PL_BATCH][FR_MAXT];
      int ic[PL_BATCH];
#pragma unroll
      for (int u = 0; u < PL_BATCH; u++) {
        const int i = i0 + u < i_end ? i0 + u : i_end - 1;
        ic[u] = i0 + u < i_end ? itemc[i] : -1;
        const int q = qmap[ic[u] >= 0 ? ic[u] : 0];
        const int r = ic[u] >= 0 ? itemr[i] : 0;
#pragma unroll
        for (int k = 0; k < FR_MAXT; k++) {
          const int d = lane * 4 + 256 * k;
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          wf[u][k] = d < D ? *reinterpret_cast<const f32x4 *>(Wm + (size_t)(q >= 0 ? q : 0) * D + d) : z;
          xf[u][k] = d < D ? *reinterpret_cast<const f32x4 *>(Vf + (size_t)r * D + d) : z;
        }
      }
#pragma unroll
      for (int u = 0; u < PL_BATCH; u++) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < FR_MAXT; k++) {
          a = fmaf(xf[u][k][0], wf[u][k][0], a);
          a = fmaf(xf[u][k][1], wf[u][k][1], a);
          a = fmaf(xf[u][k][2], wf[u][k][2], a);
          a = fmaf(xf[u][k][3], wf[u][k][3], a);
        }
        a = wave_sum(a);
        if (lane == 0 && ic[u] >= 0) res[i0 + u] = a;
      }
    };
    constexpr int CPW = GC / 8;
    for (int i0 = wave * CPW; i0 < (wave + 1) * CPW; i0 += PL_BATCH) batch(i0, (wave + 1) * CPW);
    for (int i0 = GC + wave * PL_BATCH; i0 < total; i0 += 8 * PL_BATCH) batch(i0, total);
  }
  __syncthreads();
  PSTAMP(5);
  if (tid < GC && my_n > 0) {
    const int c = tid;
    float e1 = res[c];
    int ei = itemr[c];
    if (my_n > 1) {                            // this column's appended rows: a scan of the (short) appended list
      const int total = GC + nextra[0];
      for (int i = GC; i < total; i++)
        if (itemc[i] == c) {
          const float e = res[i];
          const int ix = itemr[i];
          if (better_nan(e, ix, e1, ei)) {
            e1 = e;
            ei = ix;
          }
        }
    }
    const int q = qmap[c];
    S_max[(size_t)f * Q + q] = e1;
    D_ind[(size_t)f * Q + q] = (int64_t)ei;
  }
  // ---- the slow list, one column at a time by the WHOLE workgroup: wave w evaluates the rows r = w (mod 8), eight rows in
  // flight, exactly; torch.max's rules decide (NaN first, ties -> smaller index)
  const int ns = nslow[0];
  for (int si = 0; si < ns; si++) {
    const int c = slowc[si];
    const int q = qmap[c];
    f32x4 wq[FR_MAXT];
#pragma unroll
    for (int k = 0; k < FR_MAXT; k++) {
      const int d = lane * 4 + 256 * k;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      wq[k] = d < D ? *reinterpret_cast<const f32x4 *>(Wm + (size_t)q * D + d) : z;
    }
    float eb = -INFINITY;
    int ei = 0x7fffffff;
    for (int r0 = wave; r0 < Nb; r0 += 64) {
      f32x4 xr[8][FR_MAXT];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int r = r0 + 8 * j < Nb ? r0 + 8 * j : Nb - 1;
#pragma unroll
        for (int k = 0; k < FR_MAXT; k++) {
          const int d = lane * 4 + 256 * k;
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          xr[j][k] = d < D ? *reinterpret_cast<const f32x4 *>(Vf + (size_t)r * D + d) : z;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int r = r0 + 8 * j;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < FR_MAXT; k++) {
          a = fmaf(xr[j][k][0], wq[k][0], a);
          a = fmaf(xr[j][k][1], wq[k][1], a);
          a = fmaf(xr[j][k][2], wq[k][2], a);
          a = fmaf(xr[j][k][3], wq[k][3], a);
        }
        a = wave_sum(a);
        if (r < Nb && better_nan(a, r, eb, ei)) {
          eb = a;
          ei = r;
        }
      }
    }
    if (lane == 0) sbest[wave] = make_float2(eb, __int_as_float(ei));
    __syncthreads();
    if (tid == 0) {
      float2 b = sbest[0];
      for (int w = 1; w < 8; w++) {
        const float2 o = sbest[w];
        if (better_nan(o.x, __float_as_int(o.y), b.x, __float_as_int(b.y))) b = o;
      }
      const int bi = __float_as_int(b.y);
      S_max[(size_t)f * Q + q] = b.x;
      D_ind[(size_t)f * Q + q] = (int64_t)((bi >= 0 && bi < Nb) ? bi : 0);
    }
    __syncthreads();
  }
  PSTAMP(6);
}

template <int RW, int CW, int KIND, int WR = 2, int NST = 2>
int launch_planes(const float *V, const float *W, const unsigned char *Vp, const unsigned char *Wp, const float *vstat,
                  const float *wstat, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int G, float *S_max,
                  int64_t *D_ind, hipStream_t st) {
  constexpr int RT = WR * RW * 32, GC = (4 / WR) * 32 * CW;
  const size_t lds = (size_t)planes_lds<RT, GC, WR, NST>(Na).total;
  static_assert((size_t)(3 * (GC + GC * PL_MAXC) + 2 + GC + 16) * 4 <= (size_t)NST * (RT + GC) * 128, "exact-phase scratch fits the stages");
  const void *k = reinterpret_cast<const void *>(sim_planes_kernel<RW, CW, KIND, WR, NST>);
  if (lds > 160 * 1024) return NAFAE_ELIMIT;
  if (lds > 64 * 1024) {
    const int rc = allow_dynamic_lds(k, 160 * 1024);
    if (rc != NAFAE_OK) return rc;
  }
  const int grid = ((F + 7) / 8) * 8 * G;
  int dbg = 0;
  if (const char *e = nafae::experiment_env("NAFAE_SIM_DBG")) dbg = atoi(e);
  NAFAE_TAG("sim_planes<%d,%d,%s%s>", RW, CW, KIND == KIND_F16 ? "f16" : "bf16x3", WR == 4 ? ",narrow" : "");
  hipLaunchKernelGGL((sim_planes_kernel<RW, CW, KIND, WR, NST>), dim3(grid), dim3(512), lds, st, V, W, Vp, Wp, vstat, wstat, ent_len, F, Nb,
                     Na, Ne, D, G, S_max, D_ind, dbg);
  return launch_status();
}

inline uint32_t drop_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
}

int planes_args_ok(const void *x, int rows, int D, int kind, const void *planes, const void *stats) {
  if (!x || !planes || !stats || rows <= 0 || D <= 0) return NAFAE_EINVAL;
  if (kind != KIND_BF16X3 && kind != KIND_F16) return NAFAE_EINVAL;
  if (D % (kind == KIND_F16 ? 64 : 32)) return NAFAE_EINVAL;      // whole 128-byte lines
  return NAFAE_OK;
}
inline int planes_grid(int rows) { return rows / 4 + 1 < 4096 ? rows / 4 + 1 : 4096; }

}  // namespace

namespace nafae_sim {

// L > 64 live columns with operand planes.  Same shape limits as launch_frames (simfused.hip): D <= 512, Nb > 64; D a multiple of
// the line's k (32 / 64).
int launch_planes_frames(const float *V, const float *W, const void *Vp, const void *Wp, const float *vstat, const float *wstat,
                         int kind, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int Qh, float *S_max,
                         int64_t *D_ind, hipStream_t st) {
  const int nrb = (Nb + 31) / 32;
  const int cw = ((long)F * ((Qh + 127) / 128) >= 200) ? 2 : 1;
  const int gc = 64 * cw;
  const int G = (Qh + gc - 1) / gc;
  int rw = (nrb + 1) / 2;                     // row blocks per wave so that one super-tile covers the frame, at most 5
  rw = rw > 5 ? 5 : (rw < 2 ? 2 : rw);
  const unsigned char *vp = reinterpret_cast<const unsigned char *>(Vp), *wp = reinterpret_cast<const unsigned char *>(Wp);
#define NAFAE_PL(RW_, CW_)                                                                                                     \
  if (rw == RW_ && cw == CW_)                                                                                                  \
    return kind == KIND_F16 ? launch_planes<RW_, CW_, KIND_F16>(V, W, vp, wp, vstat, wstat, ent_len, F, Nb, Na, Ne, D, G, S_max, D_ind, st) \
                            : launch_planes<RW_, CW_, KIND_BF16X3>(V, W, vp, wp, vstat, wstat, ent_len, F, Nb, Na, Ne, D, G, S_max, D_ind, st);
  NAFAE_PL(2, 1) NAFAE_PL(3, 1) NAFAE_PL(4, 1) NAFAE_PL(5, 1)
  NAFAE_PL(2, 2) NAFAE_PL(3, 2) NAFAE_PL(4, 2) NAFAE_PL(5, 2)
#undef NAFAE_PL
  return NAFAE_ELIMIT;
}

// FEW live columns (Qh <= 64) with fp16 operand planes: the narrow form -- four row quarters x one 32-column block per workgroup, the
// frame's rows stream once per 32 live columns, a ring as deep as the stage size allows.  Nb > 32.
int launch_planes_narrow(const float *V, const float *W, const void *Vp, const void *Wp, const float *vstat, const float *wstat,
                         const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int Qh, float *S_max, int64_t *D_ind,
                         hipStream_t st) {
  const int nrb = (Nb + 31) / 32;
  int rw = (nrb + 3) / 4;                     // row blocks per wave so that one super-tile covers the frame, at most 4 (two stages of
  rw = rw > 4 ? 4 : rw;                       // 544 rows = 139 KB of LDS; longer frames take several super-tiles)
  const int G = (Qh + 31) / 32;
  const unsigned char *vp = reinterpret_cast<const unsigned char *>(Vp), *wp = reinterpret_cast<const unsigned char *>(Wp);
#define NAFAE_PN(RW_, NST_) \
  if (rw == RW_) return launch_planes<RW_, 1, KIND_F16, 4, NST_>(V, W, vp, wp, vstat, wstat, ent_len, F, Nb, Na, Ne, D, G, S_max, D_ind, st);
  NAFAE_PN(1, 6) NAFAE_PN(2, 3) NAFAE_PN(3, 2) NAFAE_PN(4, 2)
#undef NAFAE_PN
  return NAFAE_ELIMIT;
}

// Callers that hold only the fp32 operands: split V and W here, in ONE pre-pass launch, into `scratch` (planes_scratch_bytes), then
// run the planes kernel.  fp16 planes where the row is a whole number of line PAIRS (D % 128 == 0), else bf16x3 ones (D % 64 == 0).
int64_t planes_scratch_bytes(int R, int Q, int D) {
  const int64_t pb = (int64_t)D * (D % 128 == 0 ? 2 : 4);
  return (((int64_t)R * pb + 255) & ~(int64_t)255) + (((int64_t)Q * pb + 255) & ~(int64_t)255) + (int64_t)(R + Q) * 8;
}
int launch_frames_prepass(const float *V, const float *W, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int Qh,
                          float *S_max, int64_t *D_ind, void *scratch, hipStream_t st) {
  const int kind = D % 128 == 0 ? KIND_F16 : KIND_BF16X3;
  const int R = F * Nb, Q = Na * Ne;
  const int64_t pb = (int64_t)D * (kind == KIND_F16 ? 2 : 4);
  unsigned char *vp = reinterpret_cast<unsigned char *>(scratch);
  unsigned char *wp = vp + (((int64_t)R * pb + 255) & ~(int64_t)255);
  float *vs = reinterpret_cast<float *>(wp + (((int64_t)Q * pb + 255) & ~(int64_t)255));
  float *ws = vs + (size_t)R * 2;
  hipLaunchKernelGGL(planes_kernel<0>, dim3(planes_grid(R + Q)), dim3(256), 0, st, V, (const uint8_t *)nullptr, 1.f, (uint64_t)0, 0u,
                     (float *)nullptr, R, D, kind, vp, vs, W, Q, wp, ws);
  const int rc = launch_planes_frames(V, W, vp, wp, vs, ws, kind, ent_len, F, Nb, Na, Ne, D, Qh, S_max, D_ind, st);
#ifdef NAFAE_EXPERIMENTS
  {   // (tag: the pre-pass in front of the kernel the call above named)
    char t[200];
    snprintf(t, sizeof t, "planes pre-pass + %s", nafae::last_kernel_buf());
    NAFAE_TAG("%s", t);
  }
#endif
  return rc;
}

}  // namespace nafae_sim

extern "C" {

int64_t nafae_sim_planes_bytes(int rows, int D, int kind) {
  if (rows <= 0 || D <= 0 || (kind != KIND_BF16X3 && kind != KIND_F16)) return NAFAE_EINVAL;
  return (int64_t)rows * D * (kind == KIND_F16 ? 2 : 4);
}

int nafae_sim_planes(const float *X, int rows, int D, int kind, void *planes, float *stats, void *stream) {
  const int rc = planes_args_ok(X, rows, D, kind, planes, stats);
  if (rc != NAFAE_OK) return rc;
  hipLaunchKernelGGL(planes_kernel<0>, dim3(planes_grid(rows)), dim3(256), 0, as_stream(stream), X, (const uint8_t *)nullptr, 1.f,
                     (uint64_t)0, 0u, (float *)nullptr, rows, D, kind, reinterpret_cast<unsigned char *>(planes), stats,
                     (const float *)nullptr, 0, (unsigned char *)nullptr, (float *)nullptr);
  return launch_status();
}

int nafae_dropout_tanh_planes(const float *x, const uint8_t *mask, float scale, float *y, int rows, int D, int kind, void *planes,
                              float *stats, void *stream) {
  const int rc = planes_args_ok(x, rows, D, kind, planes, stats);
  if (rc != NAFAE_OK || !y) return rc != NAFAE_OK ? rc : NAFAE_EINVAL;
  hipLaunchKernelGGL(planes_kernel<1>, dim3(planes_grid(rows)), dim3(256), 0, as_stream(stream), x, mask, scale, (uint64_t)0, 0u, y,
                     rows, D, kind, reinterpret_cast<unsigned char *>(planes), stats, (const float *)nullptr, 0, (unsigned char *)nullptr,
                     (float *)nullptr);
  return launch_status();
}

int nafae_dropout_tanh_seeded_planes(const float *x, uint64_t seed, float p, float *y, int rows, int D, int kind, void *planes,
                                     float *stats, void *stream) {
  const int rc = planes_args_ok(x, rows, D, kind, planes, stats);
  if (rc != NAFAE_OK || !y || !(p >= 0.f) || !(p < 1.f)) return rc != NAFAE_OK ? rc : NAFAE_EINVAL;
  hipLaunchKernelGGL(planes_kernel<2>, dim3(planes_grid(rows)), dim3(256), 0, as_stream(stream), x, (const uint8_t *)nullptr,
                     1.0f / (1.0f - p), seed, drop_threshold(p), y, rows, D, kind, reinterpret_cast<unsigned char *>(planes), stats,
                     (const float *)nullptr, 0, (unsigned char *)nullptr, (float *)nullptr);
  return launch_status();
}

}  // extern "C"

#ifdef NAFAE_EXPERIMENTS
extern "C" int nafae_simplanes_debug_stamps(unsigned long long *out_host, int n) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(nafae_simplanes_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -3;
}
#endif
