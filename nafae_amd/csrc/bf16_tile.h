// bf16_tile.h -- bf16 MFMA tile engine with optional split-bf16 ("bf16x3") fp32 emulation, gfx950.
//
// Why: fp32 MFMA runs at 1/16 of the bf16 MFMA rate on CDNA4 and there is no xf32/TF32 path.  An fp32 value x is
// carried as two bf16 planes  hi = bf16(x), lo = bf16(x - hi)  (16-17 significant bits together, the same 4 bytes per
// element as fp32), and a product is evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  on the bf16 matrix cores with
// fp32 accumulation: 3 MFMAs instead of one fp32 MFMA that costs 16, relative error ~1e-5 per product (the dropped
// lo*lo term and the representation error are each <= 2^-18).  With SPLIT = false only the hi planes exist: plain bf16
// (BASELINE config C3).
//
// Mapping (cdna_hip_programming.md section 3):
//   * v_mfma_f32_32x32x16_bf16, 32 cycles; lane l = (r = l&31, h = l>>5) supplies A[row r][k = 8h+j], B[k = 8h+j][col r].
//   * OPERAND SWAP: the MFMA "A" operand is the WEIGHT side (rows = output channel n), the "B" operand is the
//     ACTIVATION side (cols = pixel / row m).  Accumulator register r of lane l is then
//     C[m = l&31][n = (r&3) + 8*(r>>2) + 4*(l>>5)]: 4 consecutive output channels of one row sit in 4 consecutive
//     registers of one lane, so the epilogue writes 8-byte (bf16x4) / 16-byte (f32x4) pieces instead of 2-byte scalars.
//   * both sides are K-contiguous in HBM; a k-tile is 32 bf16 = one 64-B row piece per plane; LDS image [rows][32]
//     per plane with the four 16-B slots XOR-swizzled by (row>>2)&3 -> conflict-free ds_read_b128 fragment reads
//     (16-lane groups cover rows that are distinct mod 16).
//   * 512 threads = 8 waves (2 per SIMD), tile 256(m) x 128(n): 21-31 B/clk/CU of L2->LDS staging per k-tile, the most
//     the L2 fabric sustains chip-wide at the bf16x3 MFMA rate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace nafae {

constexpr int BKH = 32;    // bf16 elements per k-tile
constexpr int NT16 = 512;  // threads per workgroup

__device__ __forceinline__ int lds_off16(int row, int slot) { return row * BKH + ((slot ^ ((row >> 2) & 3)) << 3); }

// Operand-tile layout policy.  Two layouts of a split (hi, lo) operand exist, in HBM and in LDS alike:
//   SEP  two separate planes, rows of 32 bf16 = 64 B each (also the only layout of plain bf16, PL = 1);
//   IL   "I32": hi and lo interleaved per 32-element piece, [.., C/32, 2, 32]: one 128-B line holds hi(32)|lo(32) of one
//        k-tile, so a staging instruction covers 8 rows x one FULL line instead of 16 rows x half a line -- half the
//        L1/L2 requests per byte (measured 12 % on the fc6 GEMM).  LDS rows are 128 B, 8 slots XOR-swizzled by
//        (row>>1)&7, the conflict-free pattern of the fp32 engine.
// A staging chunk c (16 B) always lands at LDS byte c*16 of its tile (LDS-DMA destinations are wave-linear).
template <bool IL, int PL>
struct TileLayout {
  // bf16-element offset of k-slot s (0..3) of plane p of row r inside a tile of NR rows
  template <int NR>
  static __device__ __forceinline__ int frag(int r, int p, int s) {
    if (IL) return r * (2 * BKH) + ((((p << 2) | s) ^ ((r >> 1) & 7)) << 3);
    return p * NR * BKH + lds_off16(r, s);
  }
  // staging chunk c of a tile of NR rows -> (row, plane, logical k-slot) it must fetch
  template <int NR>
  static __device__ __forceinline__ void decode(int c, int &row, int &plane, int &slot) {
    if (IL) {
      row = c >> 3;
      const int l = (c & 7) ^ ((row >> 1) & 7);
      plane = l >> 2;
      slot = l & 3;
    } else {
      const int rg = c >> 2;
      plane = rg / NR;
      row = rg - plane * NR;
      slot = (c & 3) ^ ((row >> 2) & 3);
    }
  }
  static constexpr int KTS = IL ? 2 * BKH : BKH;  // elements to advance per k-tile along a row
  static constexpr int RS = IL ? 2 : 1;           // row-stride multiplier (a row holds both planes)
};

// One LDS-DMA: 16 bytes per lane from the per-lane global address `src` to LDS byte (lds_dst + 16 * lane), lds_dst uniform.
// Issued through inline assembly rather than __builtin_amdgcn_global_load_lds: while a DMA that the compiler knows about is
// outstanding, its s_waitcnt insertion treats every LDS read conservatively and emits lgkmcnt(0) -- never a counted wait -- so a
// fragment prefetched for the NEXT group of MFMAs is waited for together with the current one (every lgkmcnt in the DMA kernels
// was 0).  Hidden from it, the DMAs cost no LDS-read ordering; correctness rests on the explicit vmcnt waits and barriers that
// the pipelines place anyway.  (The compiler's own vmcnt arithmetic for ordinary loads then under-counts what is outstanding,
// which only ever makes its waits stricter: vector-memory operations retire in order.)  NAFAE_ASM_DMA=0 restores the builtin.
#ifndef NAFAE_ASM_DMA
#define NAFAE_ASM_DMA 1
#endif
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16(const void *src, void *lds_dst) {
#if NAFAE_ASM_DMA
  const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(lds_dst));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(src) : "m0");
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)lds_dst,
                                   16, 0, 0);
#endif
}
// same with a uniform base and a 32-bit per-lane byte offset (the SGPR-base form of the instruction: one address register)
__device__ __forceinline__ void lds_dma16(const void *base, unsigned off, void *lds_dst) {
#if NAFAE_ASM_DMA
  const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(lds_dst));
  const uintptr_t b = reinterpret_cast<uintptr_t>(base);
  const unsigned bh = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)), bl = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned long long bs = ((unsigned long long)bh << 32) | bl;   // (the builtin returns int: widen as unsigned)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(off), "s"(bs) : "m0");
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(static_cast<const char *>(base) + off),
                                   (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
#endif
}
#pragma clang diagnostic pop

__device__ __forceinline__ void split_bf16(float v, __bf16 &hi, __bf16 &lo) {
  hi = (__bf16)v;
  lo = (__bf16)(v - (float)hi);
}

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool S16> struct AccType { typedef f32x16 type; };
template <> struct AccType<true> { typedef f32x4 type; };

// BX: rows of the activation side (m), BW: rows of the weight side (n); WX x WW waves.
// S16 selects the MFMA shape: false = v_mfma_f32_32x32x16_bf16 (two k-steps per 32-element k-tile), true =
// v_mfma_f32_16x16x32_bf16 (one k-step; lane l = (r = l&15, q = l>>4) supplies A[row r][k = 8q+j], B[k = 8q+j][col r];
// accumulator register r of lane l is C[m = l&15][n = 4*(l>>4) + r]).  Both shapes do the same FLOP per cycle and read
// the same LDS bytes per k-tile; the chip holds a higher clock under the 16x16x32 stream (MI355X_MICROARCH.md, DVFS
// give-back item 7), so the shape is chosen by measured wall time.  The accumulators are addressed through "pieces":
// NI x NJ MFMA tiles of MS x MS, each lane holding NG groups of 4 consecutive output channels per tile.
// PAIR: the two 32-element pieces of an interleaved row are not (hi, lo) of the same 32 k-values but 64 CONSECUTIVE k-values
// of a PLAIN bf16 operand (a row-major plain tensor IS that layout with K/2 'pairs'): the k-tile is 64 deep at the same LDS
// footprint, and a product is piece0*piece0 + piece1*piece1 -- 2 MFMAs per barrier-interval step instead of plain's 1.
template <int BX, int BW, int WX, int WW, bool SPLIT, bool IL = false, bool S16 = false, bool PAIR = false>
struct EngineH {
  static_assert(WX * WW == 8, "8 waves per workgroup");
  static_assert(!IL || SPLIT, "the interleaved layout is for split (hi, lo) operands");
  static_assert(!PAIR || (SPLIT && IL), "PAIR rides on the interleaved two-piece layout");
  static constexpr int TX = BX / WX / 32;
  static constexpr int TW = BW / WW / 32;
  static_assert(TX >= 1 && TW >= 1, "tile too small");
  static constexpr int MS = S16 ? 16 : 32;
  static constexpr int BXT = BX, BWT = BW;          // tile extents (rows of the activation side, of the weight side)
  static constexpr int NJ = BX / WX / MS, NI = BW / WW / MS, NG = S16 ? 1 : 4;
  static constexpr int NGRP = 4 * TW * TX / (S16 ? 1 : 2);  // `between` call sites per k-tile
  static constexpr int PL = SPLIT ? 2 : 1;
  using L = TileLayout<IL, PL>;
  static constexpr int CHUNKS = (BX + BW) * 4 * PL;  // 16-byte chunks per stage
  static constexpr int NCH = (CHUNKS + NT16 - 1) / NT16;  // per thread (the last one may be partial)
  static constexpr int STAGE = (BX + BW) * BKH * PL;  // bf16 elements per LDS stage
  static constexpr int XCH = BX * 4 * PL;             // chunks of the activation side

  using Acc = typename AccType<S16>::type;
  Acc acc[NI][NJ];
  int lane, wx, ww;

  __device__ __forceinline__ void init() {
    lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    wx = wave / WW;
    ww = wave % WW;
    zero_acc();
  }
  __device__ __forceinline__ void zero_acc() {
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 4 * NG; r++) acc[i][j][r] = 0.f;
  }

  // one accumulator tile: acc[i][j] += W-fragment(s) x X-fragment(s); small cross terms first, the dominant hi*hi last
  __device__ __forceinline__ void mma(int i, int j, const bf16x8 *w, const bf16x8 *x) {  // w[p], x[p]: planes
    if (PAIR) {
      acc[i][j] = mfma_bf16(w[0], x[0], acc[i][j]);
      acc[i][j] = mfma_bf16(w[PL - 1], x[PL - 1], acc[i][j]);
      return;
    }
    if (SPLIT) {
      acc[i][j] = mfma_bf16(w[PL - 1], x[0], acc[i][j]);
      acc[i][j] = mfma_bf16(w[0], x[PL - 1], acc[i][j]);
    }
    acc[i][j] = mfma_bf16(w[0], x[0], acc[i][j]);
  }
  // fragment coordinates of this lane: row inside an MS-row tile, and the k-slot it reads at k-step s
  __device__ __forceinline__ int frow() const { return S16 ? (lane & 15) : (lane & 31); }
  __device__ __forceinline__ int fslot(int s) const { return S16 ? (lane >> 4) : 2 * s + (lane >> 5); }
  static constexpr int KSTEPS = S16 ? 1 : 2;

  // chunk i of this thread -> (is weight side, plane, row, slot) and its LDS element offset inside a stage
  struct Chunk {
    bool w, valid;
    int plane, row, slot, lds;
  };
  __device__ __forceinline__ Chunk chunk(int i) const {
    Chunk c;
    int id = threadIdx.x + NT16 * i;
    c.valid = id < CHUNKS;
    if (!c.valid) id = 0;
    c.slot = id & 3;
    int rowg = id >> 2;
    c.w = id >= XCH;
    if (!c.w) {
      c.plane = rowg / BX;
      c.row = rowg - c.plane * BX;
      c.lds = c.plane * BX * BKH + lds_off16(c.row, c.slot);
    } else {
      rowg -= BX * PL;
      c.plane = rowg / BW;
      c.row = rowg - c.plane * BW;
      c.lds = BX * BKH * PL + c.plane * BW * BKH + lds_off16(c.row, c.slot);
    }
    return c;
  }

  // `between(g)` is called after the MFMAs of accumulator tile g (g = 0 .. NGRP-1 over the k-tile): the DMA
  // pipeline issues one staging instruction there, so the ~60-100 cycle issue cost of each global_load_lds overlaps
  // matrix work already queued on the pipe instead of serialising in front of it.
  template <class Fn>
  __device__ __forceinline__ void compute(const __bf16 *stage, Fn between) {
    const __bf16 *sX = stage;
    const __bf16 *sW = stage + BX * BKH * PL;
    const int fr = frow();
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
      const int sl = fslot(s);
      // all activation fragments of the k-step, then one weight fragment at a time (keeps the live set at NJ + 1..2
      // fragments: the 16x16x32 shape has one k-step per tile, i.e. twice the fragments of a 32x32x16 k-step)
      bf16x8 xa[NJ][PL];
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int p = 0; p < PL; p++)
          xa[j][p] = *reinterpret_cast<const bf16x8 *>(&sX[L::template frag<BX>(wx * (TX * 32) + j * MS + fr, p, sl)]);
#pragma unroll
      for (int i = 0; i < NI; i++) {
        bf16x8 wa[PL];
#pragma unroll
        for (int p = 0; p < PL; p++)
          wa[p] = *reinterpret_cast<const bf16x8 *>(&sW[L::template frag<BW>(ww * (TW * 32) + i * MS + fr, p, sl)]);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          mma(i, j, wa, xa[j]);
          between(s * NI * NJ + i * NJ + j);
        }
      }
    }
  }
  __device__ __forceinline__ void compute(const __bf16 *stage) {
    compute(stage, [](int) {});
  }

  // epilogue coordinates: group g (0..NG-1) of tile (i, j) holds C[pm(j)][pn(i, g) .. +3] in acc[i][j][4g .. 4g+3]
  __device__ __forceinline__ int pm(int j) const { return wx * (TX * 32) + j * MS + frow(); }
  __device__ __forceinline__ int pn(int i, int g) const {
    return ww * (TW * 32) + i * MS + (S16 ? 4 * (lane >> 4) : 8 * g + 4 * (lane >> 5));
  }
};

}  // namespace nafae
