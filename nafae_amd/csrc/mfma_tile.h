// mfma_tile.h -- fp32 MFMA tile engine shared by the GEMM, implicit-GEMM conv and similarity kernels.
//
// CDNA4 mapping (MI355X_MICROARCH.md / cdna_hip_programming.md section 3):
//   * v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 64 cycles per instruction per SIMD; a wave owns
//     TM x TN accumulator tiles of 32x32 (16 VGPRs each).
//   * operands: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; result register r of lane l
//     is C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
//   * both GEMM operands are K-contiguous in HBM ("NT"), staged as [rows][BK = 32 floats] LDS tiles whose
//     16-byte slots are XOR-swizzled by row, so one ds_read_b128 per lane fetches 4 consecutive k for its row
//     with no bank conflict in any 16-lane group; lanes 0-31 take k-slot 2*kk, lanes 32-63 slot 2*kk+1, and
//     element j of the two fragments feeds MFMA j (the k order inside a 8-wide group is permuted
//     identically for A and B, which a dot product does not care about).
//   * staging is global_load_dwordx4 -> registers -> ds_write_b128 one tile ahead of the MFMAs.  With 64-cycle
//     MFMAs (fp32 runs at 1/16 of the bf16 rate) the kernel is MFMA-issue bound and the 8 staging
//     instructions per 64 MFMAs are noise, so LDS-DMA would buy nothing here; predicated register loads also
//     give the zero fill that conv padding and ragged edges need.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace nafae {

constexpr int BK = 32;          // floats per k-tile (128 B rows)
constexpr int NTHREADS = 256;   // 4 waves, one per SIMD

// float offset of 16-byte slot `slot` (0..7) of row `row` inside a [rows][32] tile
__device__ __forceinline__ int lds_off(int row, int slot) { return row * BK + ((slot ^ ((row >> 1) & 7)) << 2); }

template <int BM, int BN, int WM, int WN>
struct Engine {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static constexpr int TM = BM / WM / 32;
  static constexpr int TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1, "tile too small");
  static constexpr int NA = BM / 32;  // 16-byte chunks per thread per A tile
  static constexpr int NB = BN / 32;
  static constexpr int STAGE = (BM + BN) * BK;  // floats per LDS stage

  f32x16 acc[TM][TN];
  int lane, wm, wn, srow, slot;

  __device__ __forceinline__ void init() {
    int tid = threadIdx.x;
    lane = tid & 63;
    int wave = tid >> 6;
    wm = wave / WN;
    wn = wave % WN;
    srow = tid >> 3;
    slot = tid & 7;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  }

  __device__ __forceinline__ void zero_acc() {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  }

  // registers -> LDS stage (A tile first, then B tile)
  __device__ __forceinline__ void store_stage(float *stage, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
    float *sA = stage;
    float *sB = stage + BM * BK;
#pragma unroll
    for (int i = 0; i < NA; i++) *reinterpret_cast<f32x4 *>(&sA[lds_off(srow + 32 * i, slot)]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; i++) *reinterpret_cast<f32x4 *>(&sB[lds_off(srow + 32 * i, slot)]) = rb[i];
  }

  // 16 * TM * TN MFMAs over one staged k-tile
  __device__ __forceinline__ void compute(const float *stage) {
    const float *sA = stage;
    const float *sB = stage + BM * BK;
    const int r31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; i++)
        a[i] = *reinterpret_cast<const f32x4 *>(&sA[lds_off(wm * (TM * 32) + i * 32 + r31, kk * 2 + hi)]);
#pragma unroll
      for (int j = 0; j < TN; j++)
        b[j] = *reinterpret_cast<const f32x4 *>(&sB[lds_off(wn * (TN * 32) + j * 32 + r31, kk * 2 + hi)]);
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
  }

  // row / col (inside the BM x BN block tile) of accumulator register r of tile (i, j)
  __device__ __forceinline__ int acc_row(int i, int r) const {
    return wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  }
  __device__ __forceinline__ int acc_col(int j) const { return wn * (TN * 32) + j * 32 + (lane & 31); }
};

// XCD-aware, L2-friendly tile order.  Workgroups b and b+8 share an XCD (round-robin dispatch), so first give
// every XCD a contiguous range of tile ids (bijective for any grid size), then walk tiles in groups of GM
// row-tiles so that the ~64 workgroups resident on one XCD cover an ~8x8 patch of the output and share their
// A row-panels / B column-panels through that XCD's 4 MiB L2.  Placement affects speed only.
__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int &tm, int &tn) {
  const int nwg = tiles_m * tiles_n;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  constexpr int GM = 8;
  const int per_group = GM * tiles_n;
  const int g = t / per_group;
  const int first = g * GM;
  const int gsz = min(GM, tiles_m - first);
  const int in_g = t - g * per_group;
  tm = first + in_g % gsz;
  tn = in_g / gsz;
}

}  // namespace nafae
