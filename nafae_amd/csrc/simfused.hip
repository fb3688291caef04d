// simfused.hip -- region x query similarity reduced to per-frame max / arg-max (DVSA.forward, reference model.py:548-551,
// 580-583, 610-612), third generation, gfx950.  Two kernels that replace simmax.hip's part / tile + finish pairs where they apply
// (simmax.hip's make_plan routes; its kernels stay as the fallback for the shapes these do not take):
//
//   sim_live_kernel   L <= 32 live query slots per column block -- every BASELINE configuration with entity lengths from
//       the data set's histogram (C2: 19 of 128 slots live, C5: 17 of 512).  The problem is then a pure stream of V (C5: 39 MB
//       against 0.7 GFLOP), so there is no filter and no second pass over V at all: every (proposal, live query) score is an
//       fp32 dot product on the fp32 matrix cores while the rows stream through (a wave = 32 rows x a quarter of K, see the
//       kernel).  A workgroup owns 32 consecutive rows of ONE frame and leaves its per-column (max, arg-max) in the workspace;
//       the LAST of a frame's workgroups to arrive takes the best of them (ties -> smaller index, NaN -> first NaN: torch.max's
//       rules; round 3 did that in a second launch) and the masked slots are zero-filled.  All of V is requested within the first two microseconds.  (Its first form, fp32 FMA
//       chains on the vector ALU with W as an LDS image, was VALU-bound at 17 us for C5 and is gone.)
//
//   sim_frame_kernel<RW, CW>   L > 32 (C5 with every slot live: 512 columns).  One workgroup = (frame, group of 64*CW live
//       columns): it streams ALL rows of its frame, so the per-frame max, the exact-fp32 re-evaluation of the winner and the
//       output happen inside the kernel -- no partials, no finish launch, no second pass over V from HBM (round 2: finish kernel
//       21 us re-gathering 2 x 35 MB).  Operands are staged global -> registers -> LDS: every thread converts the fp32 it
//       loaded into bf16 hi/lo ONCE and writes the planes in MFMA-fragment order (XOR-swizzled 128-B rows, conflict-free
//       ds_read_b128), so the k-loop has no LDS-DMA issue cost, no per-wave re-split of shared fragments and no W pre-pass
//       (round 2: sim_wprep 5 us).  4 waves (one per SIMD) = 2 row halves x 2 column halves, a wave holds RW x CW 32x32
//       accumulator tiles (RW = 5, CW = 2: 14 fragment reads per 30 MFMAs).  bf16x3 = hi*hi + hi*lo + lo*hi FILTERS: per column
//       the kernel keeps the top-2 (value, row) and the third-best value of each contributor; the winner and every listed
//       runner-up within `margin` of it are re-evaluated with exact fp32 dot products; if an unlisted row could lie within the
//       margin (third-best value too close), or a NaN was seen, the column takes the slow path: exact fp32 over all rows.
//       margin = 2^-14 * D * max|V_frame| * max|W_group| + 2^-11 * |score| -- the maxima are measured while staging, so the
//       bound holds for ANY embeddings, not only tanh outputs (round 2 assumed |V|, |W| <= 1).
//
// Algorithmic bytes: 4*D*(R+Q) + 12*F*Q.  No atomics; every reduction runs in a fixed order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "hip_util.h"
#include "sim_common.h"

using namespace nafae;
using namespace nafae_sim;

#ifdef NAFAE_EXPERIMENTS
// phase stamps (experiments build only; scripts/simfused_stamps.py): wall_clock64() = 100 MHz
__device__ unsigned long long nafae_simfused_stamps[8 * 8192];
#define FSTAMP(k)                                                                                    \
  do {                                                                                               \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                                \
      nafae_simfused_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64();       \
  } while (0)
#else
#define FSTAMP(k) do { } while (0)
#endif

namespace {

constexpr int FEW_MAXL = 32;    // live columns the kernel takes (one 32-column MFMA tile)
constexpr int FEW_ROWS = 32;    // rows per workgroup (one 32-row MFMA tile; its 4 waves split K)

// ---------------------------------------------------------------------------------------------------- few live columns, fp32 MFMA
// Workgroup (f, s) = rows [s*32, s*32+32) of frame f against the L <= 32 live columns; `parts` = the records of the in-launch merge.
// v_mfma_f32_32x32x2_f32 multiplies and accumulates in fp32, so every score is an fp32 dot product (no filter, no margin, no
// second pass) -- the 19 200 x 512 x 32 products of C5 cost 4 us of MFMA issue spread over the chip (as vector FMAs: 17 us).
// Wave q of the workgroup owns a quarter of K -- whole 128-B lines, [q*NL/4, (q+1)*NL/4) of a row's NL = D/32 -- of the 32 rows,
// in windows of one line per row:
//   V: a load instruction covers 8 rows x one full line (8 lanes x 16 B per row); the wave drops the window into its PRIVATE
//      LDS region (row pitch 144 B) and reads it back as A fragments -- lane (r = lane & 31, h = lane >> 5) takes the 16 B at
//      slot 2j + h of row r: both directions conflict-free, no barrier (the LDS executes one wave's instructions in order).
//      The MFMAs of a line start when that line has arrived (counted vmcnt), so what is left to do after the last byte of
//      the chip-wide stream has landed is a quarter of a wave's work, not all of it;
//   W: lane (r, h) loads the same k (slot 2j + h of the quarter) of live column r straight from global memory into the B
//      operand registers -- 64 KB of live W rows, L2 hits;
//   element e of step j multiplies k = (first k of the wave) + 8j + 4h + e on both sides: no transpose anywhere.
// The first V line is requested before the prologue (entity-length prefix, column lookup), W and the other lines behind it
// (vmcnt retires in order: waiting for W then also covers line 0, which is wanted first anyway).  The four K quarters are
// added in wave order in the epilogue, then one wave takes the column maxima over the 32 rows (torch.max's rules).
constexpr int LIVE_PITCH = 144;                 // bytes per row of a window: one 128-B line + 16
constexpr int LIVE_WIN = 32 * LIVE_PITCH;       // per wave (the wave's accumulator tile, 4 KB, goes there afterwards)
struct LiveLds {
  int win, prefix, total;
};
__host__ __device__ inline LiveLds live_lds(int Na) {
  LiveLds o;
  int p = 0;
  o.win = p;    p += 4 * LIVE_WIN;          // [wave] window; afterwards the wave's accumulator tile [register][lane]
  o.prefix = p; p += ((Na + 1) * 4 + 15) & ~15;
  o.total = p;
  return o;
}

__global__ __launch_bounds__(256, 3) void sim_live_kernel(const float *__restrict__ V, const float *__restrict__ Wm,
                                                          const int32_t *__restrict__ ent_len, int F, int Nb, int Na, int Ne,
                                                          int D, int S, int Lh, unsigned long long *__restrict__ parts,
                                                          int *__restrict__ arrived, float *__restrict__ S_max,
                                                          int64_t *__restrict__ D_ind, int dbg) {
  (void)dbg;   // timing experiment (experiments build): 1 = no MFMA (loads, transposition and the epilogue only)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const LiveLds lo = live_lds(Na);
  int *prefix = reinterpret_cast<int *>(smem + lo.prefix);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char *win = smem + lo.win + wave * LIVE_WIN;
  const int f = blockIdx.x / S, s = blockIdx.x - f * S;
  const int r31 = lane & 31, hi = lane >> 5;
  // this wave's share of K: whole 128-B lines, [l0, l1) of the D / 32 lines of a row (a quarter of them when D % 128 == 0; a
  // wave may get none -- it then reads a valid line and multiplies it as zeros)
  const int NLT = D >> 5;
  const int l0 = (wave * NLT) >> 2, l1 = ((wave + 1) * NLT) >> 2;
  const int nl = l1 - l0;                      // lines of this wave (<= 4)
  const int nj = nl * 4;                       // 8-k steps (<= 16)
  const int lb = l0 < NLT ? l0 : NLT - 1;      // first line to address
  const int nlc = nl > 0 ? nl - 1 : 0, njc = nj > 0 ? nj - 1 : 0;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  FSTAMP(0);
  // ---- V.  Lane (t = lane >> 3, sl = lane & 7): 16-B slot sl of the line; row group g holds the rows
  // (t & 1) * 8 + (t >> 1) + 4 * (g & 1) + 16 * (g >> 1), so that 16 consecutive lanes write rows a and a + 8 (disjoint banks).
  // A row beyond the frame reads the frame's last row (dropped in the epilogue), a line beyond D re-reads the last line
  // (multiplied as zeros).
  const int t8 = lane >> 3, sl = lane & 7;
  const int rl = (t8 & 1) * 8 + (t8 >> 1);
  const float *vq = V + (size_t)f * Nb * D + (size_t)lb * 32 + sl * 4;
  f32x4 x[16];
  auto v_request = [&](int li) {               // line li of the K quarter, all 32 rows
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int r = s * FEW_ROWS + rl + 4 * (g & 1) + 16 * (g >> 1);
      x[li * 4 + g] = *reinterpret_cast<const f32x4 *>(vq + (size_t)(r < Nb ? r : Nb - 1) * D + (li < nl ? li : nlc) * 32);
    }
  };
  // ---- The entity lengths are the FIRST load of the kernel: vmcnt retires in order, so waiting for them then leaves the V
  // window requested right behind them in flight (requested after it, they would be waited for together with it; as scalar
  // loads they queued behind the chip-wide flood of V requests: 10 us in the slowest workgroups).
  int el = 0;
  if (tid < 64 && tid < Na) el = ent_len[tid];
  v_request(0);
  if (Na <= 64) {
    if (wave == 0) {
      const int xl = el < 0 ? 0 : (el > Ne ? Ne : el);
      int incl = xl;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
      }
      if (lane < Na) prefix[lane] = incl - xl;
      if (lane == 63) prefix[Na] = incl;
    }
  } else {
    build_prefix(ent_len, Na, Ne, prefix);
  }
  __syncthreads();
  const int Ql = prefix[Na];
  const int CB = gridDim.y, cb = blockIdx.y;   // 32-column blocks of the live columns: this workgroup takes block cb
  int L = Ql < Lh ? Ql : Lh;                   // live columns the launch covers (those beyond the caller's bound: NaN, below)
  L = L < CB * FEW_MAXL ? L : CB * FEW_MAXL;
  int Lc = L - cb * FEW_MAXL;                  // ... of which in this block
  Lc = Lc < 0 ? 0 : (Lc > FEW_MAXL ? FEW_MAXL : Lc);
  // every lane looks up the query row of ITS column (no column map in LDS, no second barrier)
  int cq = 0;
  if (Lc > 0) {
    const int c = cb * FEW_MAXL + (r31 < Lc ? r31 : Lc - 1);
    const int a = find_seg(prefix, Na, c);
    cq = a * Ne + (c - prefix[a]);
  }
  if (Lc == 0 && !(s == 0 && cb == 0)) return;  // an over-provisioned column block (block (s = 0, cb = 0) still zero-fills)
  FSTAMP(1);
  // ---- W: a lane beyond the live columns reads the last live one (the columns of an MFMA tile are independent; dropped below)
  const float *wp = Wm + (size_t)cq * D + (size_t)lb * 32 + hi * 4;
  f32x4 w[16];
#pragma unroll
  for (int j = 0; j < 16; j++) w[j] = *reinterpret_cast<const f32x4 *>(wp + 8 * (j < nj ? j : njc));
  v_request(1);
  v_request(2);
  v_request(3);
  if (s == 0 && cb == 0) {   // this frame's masked slots: (0, 0) (model.py:551); live slots beyond the caller's bound: NaN, loud
    const int Q = Na * Ne;
    for (int q = tid; q < Q; q += 256) {
      const int a = q / Ne, e = q - a * Ne;
      const int l = prefix[a + 1] - prefix[a];
      if (e >= l) {
        S_max[(size_t)f * Q + q] = 0.f;
        D_ind[(size_t)f * Q + q] = 0;
      } else if (prefix[a] + e >= L) {
        S_max[(size_t)f * Q + q] = NAN;
        D_ind[(size_t)f * Q + q] = 0;
      }
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
  for (int li = 0; li < 4; li++) {             // window = one line of the 32 rows: 4 steps of 8 k
#pragma unroll
    for (int g = 0; g < 4; g++)
      *reinterpret_cast<f32x4 *>(win + (rl + 4 * (g & 1) + 16 * (g >> 1)) * LIVE_PITCH + sl * 16) = x[li * 4 + g];
    f32x4 a[4];
#pragma unroll
    for (int jw = 0; jw < 4; jw++) a[jw] = *reinterpret_cast<const f32x4 *>(win + r31 * LIVE_PITCH + (2 * jw + hi) * 16);
#ifdef NAFAE_EXPERIMENTS
    if (dbg & 1) {                             // timing experiment: consume the operands, no MFMA
#pragma unroll
      for (int jw = 0; jw < 4; jw++) acc[li * 4 + jw] = a[jw][0] + w[li * 4 + jw][0];
    } else
#endif
#pragma unroll
    for (int jw = 0; jw < 4; jw++) {
      const int j = li * 4 + jw;
      f32x4 av = a[jw], bv = w[j];
      if (j >= nj) {
        av = z4;
        bv = z4;
      }
#pragma unroll
      for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
    }
  }
  FSTAMP(2);
  // ---- K quarters: waves 1 .. 3 leave their tiles in their own LDS region for wave 0; ((q0 + q1) + q2) + q3
  if (wave > 0) {
    float *red = reinterpret_cast<float *>(win);
#pragma unroll
    for (int r = 0; r < 16; r++) red[r * 64 + lane] = acc[r];
  }
  __syncthreads();
  FSTAMP(3);
  if (wave == 0) {
#pragma unroll
    for (int u = 1; u < 4; u++) {
      const float *red = reinterpret_cast<const float *>(smem + lo.win + u * LIVE_WIN);
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] += red[r * 64 + lane];
    }
    // lane (column r31, half hi) holds rows (r & 3) + 8 * (r >> 2) + 4 * hi of the tile, ascending in r
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      // rows arrive in ascending order: the first valid one is taken, a later one wins only if strictly greater, or NaN while
      // the best is not NaN yet
      const int rr = s * FEW_ROWS + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float v = acc[r];
      const bool take = (rr < Nb) & ((bi == 0x7fffffff) | (!(bv != bv) & ((v != v) | (v > bv))));
      bv = take ? v : bv;
      bi = take ? rr : bi;
    }
    const float ov = __shfl_xor(bv, 32);
    const int oi = __shfl_xor(bi, 32);
    if (better_nan(ov, oi, bv, bi)) {
      bv = ov;
      bi = oi;
    }
    // ---- hand-off inside the launch (round 4; round 3 ran sim_few_merge_kernel as a second launch: 4 us of 18 at C5).  The S
    // row-block workgroups of (frame, column block) leave their (max, arg-max) records in the workspace and count themselves on
    // arrived[f][cb]; the one whose add returns S - 1 has seen every other arrive, reads the S records and writes the frame's
    // result, then puts the counter back to 0 for the next call (the workspace is zeroed ONCE, when it is created).  Visibility
    // across CUs / XCDs without fences (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores, the storing wave's
    // vmcnt(0), one agent-scope add by a lane of that wave, sc1 loads by the wave whose add came last): records are written and
    // read as 8-byte agent-scope relaxed atomics (global_store / global_load ... sc1), all by wave 0.
    if (Lc > 0) {
      unsigned long long *rec0 = parts + (((size_t)f * S) * CB + cb) * FEW_MAXL + r31;
      const size_t rstride = (size_t)CB * FEW_MAXL;
      if (lane < Lc)
        __hip_atomic_store(rec0 + (size_t)s * rstride, ((unsigned long long)(unsigned)bi << 32) | (unsigned long long)__float_as_uint(bv),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add(&arrived[f * CB + cb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      if (old == S - 1) {                      // every record of (f, cb) is in memory: lane (column r31, half hi) takes s = hi, hi + 2, ...
        asm volatile("" ::: "memory");
        float mv = -INFINITY;
        int mi = 0x7fffffff;
        constexpr int SB = 8;
        for (int s0 = hi; s0 < S; s0 += 2 * SB) {
          unsigned long long o[SB];
#pragma unroll
          for (int u = 0; u < SB; u++) {
            const int sx = s0 + 2 * u < S ? s0 + 2 * u : s0;
            o[u] = __hip_atomic_load(rec0 + (size_t)sx * rstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int u = 0; u < SB; u++) {
            const float v = __uint_as_float((unsigned)o[u]);
            const int ix = (int)(o[u] >> 32);
            if (s0 + 2 * u < S && better_nan(v, ix, mv, mi)) {
              mv = v;
              mi = ix;
            }
          }
        }
        const float pv = __shfl_xor(mv, 32);
        const int pi = __shfl_xor(mi, 32);
        if (better_nan(pv, pi, mv, mi)) {
          mv = pv;
          mi = pi;
        }
        if (lane < Lc) {
          mi = mi < 0 ? 0 : (mi >= Nb ? Nb - 1 : mi);
          const size_t o = (size_t)f * (Na * Ne) + cq;
          S_max[o] = mv;
          D_ind[o] = (int64_t)mi;
        }
        if (lane == 0) __hip_atomic_store(&arrived[f * CB + cb], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  FSTAMP(4);
}

// ---------------------------------------------------------------------------------------------------- many live columns
// per-column statistics of one contributor: best two (value, row) and the third-best value
struct Top {
  float m1, m2, m3;
  int i1, i2;
};
__device__ __forceinline__ Top top_empty() { return Top{-INFINITY, -INFINITY, -INFINITY, 0x7fffffff, 0x7fffffff}; }   // (index: none)
// One element into the running top-3.  Elements arrive in ascending row order, so strict `>` keeps the smaller row on ties.  The
// payload is the element's compile-time id (ID - 16 is an inline constant for ID < 80), decoded into a row afterwards: eight
// branch-free vector instructions.  (Written as ternaries hipcc turned the updates into exec-masked branches and hoisted all
// 160 row-validity compares of the tile into spilled SGPR masks.)
template <int ID>
__device__ __forceinline__ void top_push(Top &t, float v) {
  static_assert(ID >= 0 && ID < 80, "id must fit an inline constant");
  asm volatile(
      "v_med3_f32 %2, %1, %2, %5\n\t"
      "v_cmp_gt_f32 vcc, %5, %1\n\t"
      "v_cndmask_b32_e64 %4, %4, %6, vcc\n\t"
      "v_med3_f32 %1, %0, %1, %5\n\t"
      "v_cmp_gt_f32 vcc, %5, %0\n\t"
      "v_cndmask_b32_e32 %4, %4, %3, vcc\n\t"
      "v_cndmask_b32_e64 %3, %3, %6, vcc\n\t"
      "v_max_f32 %0, %0, %5"
      : "+v"(t.m1), "+v"(t.m2), "+v"(t.m3), "+v"(t.i1), "+v"(t.i2)
      : "v"(v), "n"(ID - 16)
      : "vcc");
}
__device__ __forceinline__ Top top_merge(Top a, Top b) {
  if (better(b.m1, b.i1, a.m1, a.i1)) {
    const Top t = a;
    a = b;
    b = t;
  }
  Top o;
  o.m1 = a.m1;
  o.i1 = a.i1;
  if (better(a.m2, a.i2, b.m1, b.i1)) {
    o.m2 = a.m2;
    o.i2 = a.i2;
    o.m3 = fmaxf(a.m3, b.m1);
  } else {
    o.m2 = b.m1;
    o.i2 = b.i1;
    o.m3 = fmaxf(a.m2, b.m2);
  }
  return o;
}

constexpr int FR_MAXT = 2;      // D <= 512: float4 pieces per lane of an exact dot product

// grid ceil(F/8)*8*G workgroups of 512 threads (one per CU: ~120 KB of LDS); workgroup = (frame f, column group g).
// Waves 0-3 (one per SIMD) issue the MFMAs: wave = (row half rh, column half ch), RW x CW accumulator tiles.  Waves 4-7 (their
// SIMD partners) STAGE: global_load_dwordx4 (8 lanes per 128-B line) -> split into bf16 hi / lo -> ds_write_b64 into the other
// LDS stage, two chunks of loads in flight.  The hardware interleaves the partner's vector work with the MFMA wave's matrix
// work; as one instruction stream hipcc ran the conversion, the MFMAs and the loads of a chunk one after the other.
// LDS: [2 stages][(RT + GC) rows][128 B: hi p0..p3 | lo p0..p3, 16-B slots XOR-swizzled by (row >> 1) & 7][qmap GC][prefix]
// Where a trip of the k-loop goes (dbg bit 256, scripts/simfused_trip.py; C5 all live, shader cycles per trip): MFMA waves issue
// reads + 60 MFMAs in 2 400; a staging wave's split + ds_write takes 1 500 alone and 3 400 beside a running MFMA wave (the two
// share the SIMD's vector issue and overlap by a third only); the loads are never waited for -- with the conversion
// compiled out the loop is MFMA-bound (20.6 us against 33), and a third register set of loads in flight changed nothing.
template <int RW, int CW>
__global__ __launch_bounds__(512) void sim_frame_kernel(const float *__restrict__ V, const float *__restrict__ Wm,
                                                        const int32_t *__restrict__ ent_len, int F, int Nb, int Na, int Ne,
                                                        int D, int G, float *__restrict__ S_max, int64_t *__restrict__ D_ind,
                                                        int dbg) {
  constexpr int RT = 2 * RW * 32;            // rows per super-tile (two row halves)
  (void)dbg;   // timing experiments (experiments build): 1 stop before the exact phase, 2 no MFMAs, 4 no global loads, 8 no conversion
  constexpr int GC = 64 * CW;                // live columns per workgroup (two column halves)
  constexpr int NSV = RT / 32, NSW = GC / 32;
  constexpr int STAGE = (RT + GC) * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *stage0 = smem;
  int *qmap = reinterpret_cast<int *>(smem + 2 * STAGE);
  int *prefix = qmap + GC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b8 = blockIdx.x >> 3;
  const int f = (b8 / G) * 8 + (blockIdx.x & 7), g = b8 % G;     // the G workgroups of a frame share an XCD (blockIdx % 8)
  if (f >= F) return;
  const int Q = Na * Ne;
  const int nch = D >> 5;

  build_prefix(ent_len, Na, Ne, prefix);
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
  if (g * GC >= Ql) return;                  // over-provisioned column group
  if (tid < GC) {
    const int c = g * GC + tid;
    int q = -1;
    if (c < Ql) {
      const int a = find_seg(prefix, Na, c);
      q = a * Ne + (c - prefix[a]);
    }
    qmap[tid] = q;
  }
  __syncthreads();
  const float *Vf = V + (size_t)f * Nb * D;
  const int nsuper = (Nb + RT - 1) / RT;
  FSTAMP(0);

  Top top[CW];
  bool nanf[CW];
#pragma unroll
  for (int cb = 0; cb < CW; cb++) {
    top[cb] = top_empty();
    nanf[cb] = false;
  }
  float mv = 0.f, mw = 0.f;                    // max |x| over the operand elements this thread staged
  const int lr = lane & 31, h = lane >> 5;
  const int rh = (wave >> 1) & 1, ch = wave & 1;

  if (wave >= 4) {
    // ================================================================ staging waves
#ifdef NAFAE_EXPERIMENTS
    if (dbg & 32) __builtin_amdgcn_s_setprio(1);       // timing experiment: staging waves win the issue arbitration
    if (dbg & 128) __builtin_amdgcn_s_setprio(3);
#endif
    const int ct = tid - 256;
    const int tr = ct >> 3, ts = ct & 7;       // thread (tr, ts) moves 16 B (4 k) of tile row tr + 32 i per slot
    const int swz = frame_swz(tr);
    const int wr_hi = tr * 128 + ((ts >> 1) ^ swz) * 16 + (ts & 1) * 8;
    const int wr_lo = tr * 128 + ((4 + (ts >> 1)) ^ swz) * 16 + (ts & 1) * 8;
    int woff[NSW];
#pragma unroll
    for (int j = 0; j < NSW; j++) {
      const int q = qmap[tr + 32 * j];
      woff[j] = (q >= 0 ? q : 0) * D + ts * 4;   // (a column beyond the live count reads query row 0; its results are never stored)
    }
    for (int rt = 0; rt < nsuper; rt++) {
      int voff[NSV];
#pragma unroll
      for (int i = 0; i < NSV; i++) {
        int row = rt * RT + tr + 32 * i;
        row = row < Nb ? row : Nb - 1;
        voff[i] = row * D + ts * 4;
      }
      f32x4 sv[2][NSV], sw[2][NSW];            // two register sets: chunk c travels in set c & 1
      auto issue = [&](int ci, auto set_tag) {
        constexpr int S_ = decltype(set_tag)::value;
#ifdef NAFAE_EXPERIMENTS
        if (dbg & 4) {                         // timing experiment: no global loads
#pragma unroll
          for (int i = 0; i < NSV; i++) sv[S_][i] = f32x4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
          for (int j = 0; j < NSW; j++) sw[S_][j] = f32x4{1.f, 2.f, 3.f, 4.f};
          return;
        }
#endif
#pragma unroll
        for (int i = 0; i < NSV; i++) sv[S_][i] = *reinterpret_cast<const f32x4 *>(Vf + voff[i] + ci * 32);
#pragma unroll
        for (int j = 0; j < NSW; j++) sw[S_][j] = *reinterpret_cast<const f32x4 *>(Wm + woff[j] + ci * 32);
      };
      auto convert = [&](unsigned char *st, auto set_tag) {
        constexpr int S_ = decltype(set_tag)::value;
#ifdef NAFAE_EXPERIMENTS
        if (dbg & 8) {                         // timing experiment: loads only (their values still have to arrive)
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < NSV; i++) t += sv[S_][i][0];
#pragma unroll
          for (int j = 0; j < NSW; j++) t += sw[S_][j][0];
          mv = fmaxf(mv, t);
          return;
        }
#endif
#pragma unroll
        for (int i = 0; i < NSV; i++) {
          bf16x4 hi, lo;
          const f32x4 x = sv[S_][i];
          split4(x, hi, lo);
          mv = absmax4(mv, x);
          *reinterpret_cast<bf16x4 *>(st + wr_hi + i * 4096) = hi;
          *reinterpret_cast<bf16x4 *>(st + wr_lo + i * 4096) = lo;
        }
#pragma unroll
        for (int j = 0; j < NSW; j++) {
          bf16x4 hi, lo;
          const f32x4 x = sw[S_][j];
          split4(x, hi, lo);
          mw = absmax4(mw, x);
          *reinterpret_cast<bf16x4 *>(st + RT * 128 + wr_hi + j * 4096) = hi;
          *reinterpret_cast<bf16x4 *>(st + RT * 128 + wr_lo + j * 4096) = lo;
        }
      };
      using S0 = std::integral_constant<int, 0>;
      using S1 = std::integral_constant<int, 1>;
      if (rt > 0) lds_barrier();               // (the MFMA waves are done reading the previous super-tile's last stage)
      // nch is even (the launcher checks D % 64 == 0).  Every convert / issue below is UNCONDITIONAL -- past the end a valid
      // chunk is re-read and converted into a stage nobody reads any more: with `if (ci + 3 < nch) issue(...)` hipcc's
      // wait-count pass had to assume the younger set of loads might not exist and waited vmcnt(13..0) in every convert,
      // i.e. for BOTH sets, which halves the prefetch distance.
      auto clampc = [&](int c) { return c < nch ? c : nch - 1; };
      {
        issue(0, S0{});
        issue(1, S1{});
        convert(stage0, S0{});
        issue(clampc(2), S0{});
        lds_barrier();
        FSTAMP(1);
#ifdef NAFAE_EXPERIMENTS
        if (dbg & 256) {                       // timing experiment: where a staging wave's trip goes (shader-clock cycles, summed)
          long long t_wait = 0, t_conv = 0, t_bar = 0;
          for (int ci = 0; ci < nch; ci += 2) {
            long long ta = clock64();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSV + NSW) : "memory");
            long long tb = clock64();
            convert(stage0 + STAGE, S1{});
            issue(clampc(ci + 3), S1{});
            long long tc = clock64();
            lds_barrier();
            long long td = clock64();
            t_wait += tb - ta; t_conv += tc - tb; t_bar += td - tc;
            ta = clock64();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSV + NSW) : "memory");
            tb = clock64();
            convert(stage0, S0{});
            issue(clampc(ci + 4), S0{});
            tc = clock64();
            lds_barrier();
            td = clock64();
            t_wait += tb - ta; t_conv += tc - tb; t_bar += td - tc;
          }
          if (lane == 0 && blockIdx.x < 1024) {
            unsigned long long *o = nafae_simfused_stamps + (blockIdx.x * 8 + wave) * 8;
            o[3] = t_wait; o[4] = t_conv; o[5] = t_bar;
          }
        } else
#endif
        // trip ci (the MFMA waves compute chunk ci): convert chunk ci + 1 into the other stage, request chunk ci + 3
        for (int ci = 0; ci < nch; ci += 2) {
          convert(stage0 + STAGE, S1{});
          issue(clampc(ci + 3), S1{});
          lds_barrier();
          convert(stage0, S0{});
          issue(clampc(ci + 4), S0{});
          lds_barrier();
        }
      }
      FSTAMP(2);
    }
  } else {
    // ================================================================ MFMA waves
#ifdef NAFAE_EXPERIMENTS
    if (dbg & 64) __builtin_amdgcn_s_setprio(1);        // timing experiment: MFMA waves win the issue arbitration
#endif
    const int aswz = frame_swz(lr);
    const int a_base = (rh * RW * 32 + lr) * 128;
    const int b_base = (RT + ch * CW * 32 + lr) * 128;
    int fo[2][2];                              // [plane][k-step]: byte offset of this lane's 16-B piece inside its row
#pragma unroll
    for (int pl = 0; pl < 2; pl++)
#pragma unroll
      for (int t = 0; t < 2; t++) fo[pl][t] = ((pl * 4 + 2 * t + h) ^ aswz) << 4;
    for (int rt = 0; rt < nsuper; rt++) {
      f32x16 acc[RW][CW];
#pragma unroll
      for (int rb = 0; rb < RW; rb++)
#pragma unroll
        for (int cb = 0; cb < CW; cb++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[rb][cb][r] = 0.f;
      if (rt > 0) lds_barrier();
      lds_barrier();                           // chunk 0 is in stage 0
      FSTAMP(1);
#ifdef NAFAE_EXPERIMENTS
      long long m_comp = 0, m_bar = 0, m_t0 = 0;
#endif
      for (int ci = 0; ci < nch; ci++) {
        const unsigned char *st = stage0 + (ci & 1) * STAGE;
#ifdef NAFAE_EXPERIMENTS
        if (dbg & 256) m_t0 = clock64();
#endif
#ifdef NAFAE_EXPERIMENTS
        if (dbg & 2) {                         // timing experiment: no fragment reads, no MFMAs
          lds_barrier();
          continue;
        }
#endif
#pragma unroll
        for (int t = 0; t < 2; t++) {
          bf16x8 bhi[CW], blo[CW];
#pragma unroll
          for (int cb = 0; cb < CW; cb++) {
            bhi[cb] = *reinterpret_cast<const bf16x8 *>(st + b_base + cb * 4096 + fo[0][t]);
            blo[cb] = *reinterpret_cast<const bf16x8 *>(st + b_base + cb * 4096 + fo[1][t]);
          }
#pragma unroll
          for (int rb = 0; rb < RW; rb++) {
            const bf16x8 ahi = *reinterpret_cast<const bf16x8 *>(st + a_base + rb * 4096 + fo[0][t]);
            const bf16x8 alo = *reinterpret_cast<const bf16x8 *>(st + a_base + rb * 4096 + fo[1][t]);
#pragma unroll
            for (int cb = 0; cb < CW; cb++) {
              acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi[cb], acc[rb][cb], 0, 0, 0);
              acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo[cb], acc[rb][cb], 0, 0, 0);
              acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi[cb], acc[rb][cb], 0, 0, 0);
            }
          }
        }
#ifdef NAFAE_EXPERIMENTS
        if (dbg & 256) {
          const long long t1 = clock64();
          lds_barrier();
          const long long t2 = clock64();
          m_comp += t1 - m_t0;
          m_bar += t2 - t1;
          continue;
        }
#endif
        lds_barrier();
      }
      FSTAMP(2);
#ifdef NAFAE_EXPERIMENTS
      if ((dbg & 256) && lane == 0 && blockIdx.x < 1024) {
        unsigned long long *o = nafae_simfused_stamps + (blockIdx.x * 8 + wave) * 8;
        o[3] = m_comp; o[4] = m_bar; o[5] = 0;
      }
#endif
      // ---- this lane's column: 16 rows per 32-row block, ascending.  Element id = rb * 16 + r; NaN / Inf anywhere in the column
      // makes nanacc NaN (x * 0), which sends the column to the exact slow path.
#pragma unroll
      for (int cb = 0; cb < CW; cb++) {
        Top loc = top_empty();
        float nanacc = 0.f;
        const int wbase = rt * RT + rh * RW * 32;        // first row of this wave's blocks
        unroll_blocks<RW>([&](auto rb_tag) {
          constexpr int rb = decltype(rb_tag)::value;
          const int lim = Nb - (wbase + rb * 32) - 4 * h;  // element r is a real row iff (r & 3) + 8 * (r >> 2) < lim
          unroll_blocks<16>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            float v = acc[rb][cb][r];
            nanacc = fmaf(v, 0.f, nanacc);
            if (wbase + rb * 32 + 32 > Nb) v = ((r & 3) + 8 * (r >> 2) < lim) ? v : -INFINITY;   // (uniform: the frame's last block only)
            top_push<rb * 16 + r>(loc, v);
          });
        });
        // payload -> row, then into the running statistics (an earlier super-tile's rows are smaller: better() keeps the order)
        auto row_of = [&](int id) {
          if (id > 63) return 0x7fffffff;             // (none)
          id += 16;
          return wbase + (id >> 4) * 32 + (id & 3) + 8 * ((id & 15) >> 2) + 4 * h;
        };
        loc.i1 = row_of(loc.i1);
        loc.i2 = row_of(loc.i2);
        top[cb] = top_merge(top[cb], loc);
        nanf[cb] = nanf[cb] || (nanacc != nanacc);
      }
    }
  }

  // ---- the four contributors of a column (row half x 16-row lane half) leave their statistics in LDS: 8 listed candidates
#ifdef NAFAE_EXPERIMENTS
  if (dbg & 256) return;                       // (trip-time experiment: slots 3..5 hold cycle sums)
#endif
  FSTAMP(3);
  __syncthreads();                             // the stages are free: reuse them as scratch
  Top *ctop = reinterpret_cast<Top *>(smem);                       // [4 contributors][GC]
  int *cnan = reinterpret_cast<int *>(smem + 4 * GC * sizeof(Top));   // [4][GC]
  float *red = reinterpret_cast<float *>(smem + 4 * GC * sizeof(Top) + 4 * GC * 4);   // [2][4 staging waves]
  float2 *sbest = reinterpret_cast<float2 *>(red + 8);             // [8 waves] per-wave result of a slow column (8-B aligned:
  static_assert((4 * GC * sizeof(Top) + 4 * GC * 4 + 32) % 8 == 0, "sbest must be 8-byte aligned");   // ds_read/write_b64)
  int *nslow = reinterpret_cast<int *>(sbest + 8);                 // [1] number of columns on the slow path
  int *slowc = nslow + 1;                                           // [GC] their column indices
  if (tid == 0) nslow[0] = 0;
  if (wave < 4) {
#pragma unroll
    for (int cb = 0; cb < CW; cb++) {
      const int c = (ch * CW + cb) * 32 + lr;
      ctop[(rh * 2 + h) * GC + c] = top[cb];
      cnan[(rh * 2 + h) * GC + c] = (int)nanf[cb];
    }
  } else {
    float a = mv, b = mw;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a = fmaxf(a, __shfl_xor(a, o));
      b = fmaxf(b, __shfl_xor(b, o));
    }
    if (lane == 0) {
      red[wave - 4] = a;
      red[wave] = b;
    }
  }
  __syncthreads();
#ifdef NAFAE_EXPERIMENTS
  if (dbg & 1) return;                         // timing experiment: k-loop + scan only
#endif
  const float mvw = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  const float mabs = 6.103515625e-05f * (float)D * mvw;          // 2^-14 * D * max|V| * max|W|

  // ---- one thread per column: winner, the other LISTED candidates within the margin of the best filter value, and whether an
  // UNLISTED row could lie within it (a contributor's third-best value too close) or a NaN was seen (-> slow list)
  int *rec = slowc + GC;                                           // [GC][10]: i1, ncand, cand[0..7]
  if (tid < GC) {
    const int c = tid;
    Top t[4];
    int nan = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      t[k] = ctop[k * GC + c];
      nan |= cnan[k * GC + c];
    }
    float bm = -INFINITY;
    int i1 = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (better(t[k].m1, t[k].i1, bm, i1)) {
        bm = t[k].m1;
        i1 = t[k].i1;
      }
    const float margin = mabs + 4.8828125e-04f * fabsf(bm);
    bool slow = nan != 0;
    int n = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      slow = slow || !(bm - t[k].m3 >= margin);
      if (t[k].i1 != i1 && t[k].i1 >= 0 && t[k].i1 < Nb && bm - t[k].m1 < margin) rec[c * 10 + 2 + n++] = t[k].i1;
      if (t[k].i2 != i1 && t[k].i2 >= 0 && t[k].i2 < Nb && bm - t[k].m2 < margin) rec[c * 10 + 2 + n++] = t[k].i2;
    }
    rec[c * 10] = (i1 >= 0 && i1 < Nb) ? i1 : 0;
    rec[c * 10 + 1] = n;
    if (slow && qmap[c] >= 0) slowc[atomicAdd(nslow, 1)] = c;
    if (slow) rec[c * 10 + 1] = -1;
  }
  __syncthreads();

  FSTAMP(4);
  // ---- exact fp32, phase A: wave w takes columns [w * GC/8, (w+1) * GC/8), BATCH at a time: the W row and the winner's V row of
  // the whole batch are requested together
  constexpr int CPW = GC / 8, BATCH = 8;
  for (int c0 = wave * CPW; c0 < (wave + 1) * CPW; c0 += BATCH) {
    f32x4 wf[BATCH][FR_MAXT], xf[BATCH][FR_MAXT];
    int qq[BATCH], i1[BATCH], nc[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; u++) {
      const int c = c0 + u;
      qq[u] = qmap[c];
      i1[u] = rec[c * 10];
      nc[u] = rec[c * 10 + 1];
      const int q = qq[u] >= 0 ? qq[u] : 0;
#pragma unroll
      for (int k = 0; k < FR_MAXT; k++) {
        const int d = lane * 4 + 256 * k;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        wf[u][k] = d < D ? *reinterpret_cast<const f32x4 *>(Wm + (size_t)q * D + d) : z;
        xf[u][k] = d < D ? *reinterpret_cast<const f32x4 *>(Vf + (size_t)i1[u] * D + d) : z;
      }
    }
    float eb[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; u++) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < FR_MAXT; k++) {
        acc = fmaf(xf[u][k][0], wf[u][k][0], acc);
        acc = fmaf(xf[u][k][1], wf[u][k][1], acc);
        acc = fmaf(xf[u][k][2], wf[u][k][2], acc);
        acc = fmaf(xf[u][k][3], wf[u][k][3], acc);
      }
      eb[u] = wave_sum(acc);
    }
#pragma unroll
    for (int u = 0; u < BATCH; u++) {
      if (qq[u] < 0 || nc[u] < 0) continue;    // (wave-uniform) column beyond the live count, or on the slow list
      const int c = c0 + u;
      float e1 = eb[u];
      int ei = i1[u];
      for (int k = 0; k < nc[u]; k++) {        // (rare) the other listed candidates: decided in fp32 where the filter cannot
        const int ix = rec[c * 10 + 2 + k];
        const float e = wave_dot<FR_MAXT>(Vf + (size_t)ix * D, wf[u], D, lane);
        if (better_nan(e, ix, e1, ei)) {
          e1 = e;
          ei = ix;
        }
      }
      if (lane == 0) {
        S_max[(size_t)f * Q + qq[u]] = e1;
        D_ind[(size_t)f * Q + qq[u]] = (int64_t)ei;
      }
    }
  }
  FSTAMP(5);
  // ---- phase B: the slow list, one column at a time by the WHOLE workgroup: wave w evaluates the rows r = w (mod 8), eight
  // rows (sixteen 16-B loads per lane) in flight, exactly; torch.max's rules decide (NaN first, ties -> smaller index)
  __syncthreads();
#ifdef NAFAE_EXPERIMENTS
  if (dbg & 16) return;                        // timing experiment: no slow list
#endif
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
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < FR_MAXT; k++) {
          acc = fmaf(xr[j][k][0], wq[k][0], acc);
          acc = fmaf(xr[j][k][1], wq[k][1], acc);
          acc = fmaf(xr[j][k][2], wq[k][2], acc);
          acc = fmaf(xr[j][k][3], wq[k][3], acc);
        }
        acc = wave_sum(acc);
        if (r < Nb && better_nan(acc, r, eb, ei)) {
          eb = acc;
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
  FSTAMP(6);
}

template <int RW, int CW>
int launch_frame(const float *V, const float *W, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int G,
                 float *S_max, int64_t *D_ind, hipStream_t st) {
  constexpr int RT = 2 * RW * 32, GC = 64 * CW;
  const size_t lds = 2 * (size_t)(RT + GC) * 128 + GC * 4 + (((size_t)(Na + 1) * 4 + 15) & ~(size_t)15);
  const void *k = reinterpret_cast<const void *>(sim_frame_kernel<RW, CW>);
  if (lds > 64 * 1024) {
    const int rc = allow_dynamic_lds(k, 160 * 1024);
    if (rc != NAFAE_OK) return rc;
  }
  const int grid = ((F + 7) / 8) * 8 * G;
  int dbg = 0;
  if (const char *e = nafae::experiment_env("NAFAE_SIM_DBG")) dbg = atoi(e);
  hipLaunchKernelGGL((sim_frame_kernel<RW, CW>), dim3(grid), dim3(512), lds, st, V, W, ent_len, F, Nb, Na, Ne, D, G, S_max, D_ind,
                     dbg);
  return launch_status();
}

}  // namespace

namespace nafae_sim {

// Any number of live columns Lh (the caller's bound) in blocks of 32, D % 32 == 0, D <= 512.  A block re-reads V (from L2 when
// the blocks of a row block run together), so beyond a few blocks launch_frames is the better route where it applies.
// workspace: FEW_NCNT arrival counters (int; one per (frame, column block)) at its START -- a fixed place and size, whatever the
// shape of the call, so that one workspace can serve calls of different shapes -- then the (max, arg-max) records [F][S][CB][32] of
// 8 bytes.  The counters must be ZERO when the first call on a workspace starts; every completed call leaves them zero.
int64_t few_workspace_bytes(int F, int Nb, int Q) {
  const int64_t CB = (Q + FEW_MAXL - 1) / FEW_MAXL;
  return (int64_t)FEW_NCNT * 4 + (int64_t)F * ((Nb + FEW_ROWS - 1) / FEW_ROWS) * CB * FEW_MAXL * 8;
}

int launch_few(const float *V, const float *W, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int Lh, float *S_max,
               int64_t *D_ind, void *workspace, hipStream_t st) {
  const int S = (Nb + FEW_ROWS - 1) / FEW_ROWS;
  const int CB = (Lh + FEW_MAXL - 1) / FEW_MAXL;
  if ((long)F * CB > FEW_NCNT) return NAFAE_ELIMIT;
  int *arrived = reinterpret_cast<int *>(workspace);
  unsigned long long *parts = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(workspace) + (size_t)FEW_NCNT * 4);
  const LiveLds lo = live_lds(Na);
  int dbg = 0;
  if (const char *e = nafae::experiment_env("NAFAE_SIM_DBG")) dbg = atoi(e);
  hipLaunchKernelGGL(sim_live_kernel, dim3(F * S, CB), dim3(256), lo.total, st, V, W, ent_len, F, Nb, Na, Ne, D, S, Lh, parts, arrived,
                     S_max, D_ind, dbg);
  return launch_status();
}

// L > 32 live columns: Qh = the caller's bound.  D % 64 == 0, D <= 512, Nb > 64.
int launch_frames(const float *V, const float *W, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int Qh,
                  float *S_max, int64_t *D_ind, hipStream_t st) {
  const int nrb = (Nb + 31) / 32;
  // 128-column groups when they still give every CU a workgroup, else 64-column groups
  const int cw = ((long)F * ((Qh + 127) / 128) >= 200) ? 2 : 1;
  const int gc = 64 * cw;
  const int G = (Qh + gc - 1) / gc;
  int rw = (nrb + 1) / 2;                     // row blocks per wave so that one super-tile covers the frame, at most 5
  rw = rw > 5 ? 5 : (rw < 2 ? 2 : rw);
#define NAFAE_FR(RW_, CW_) \
  if (rw == RW_ && cw == CW_) return launch_frame<RW_, CW_>(V, W, ent_len, F, Nb, Na, Ne, D, G, S_max, D_ind, st);
  NAFAE_FR(2, 1) NAFAE_FR(3, 1) NAFAE_FR(4, 1) NAFAE_FR(5, 1)
  NAFAE_FR(2, 2) NAFAE_FR(3, 2) NAFAE_FR(4, 2) NAFAE_FR(5, 2)
#undef NAFAE_FR
  return NAFAE_ELIMIT;
}

}  // namespace nafae_sim

#ifdef NAFAE_EXPERIMENTS
extern "C" int nafae_simfused_debug_stamps(unsigned long long *out_host, int n) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(nafae_simfused_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -3;
}
#endif
