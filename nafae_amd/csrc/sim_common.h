// sim_common.h -- pieces of the similarity kernels (simfused.hip; limits used by the routing in simmax.hip): live-column
// bookkeeping, the arg-max ordering, the fp32 -> bf16 hi/lo split, wave reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace nafae_sim {

constexpr int NA_MAX = 2048;   // segments per batch the live-column prefix table holds (LDS)
constexpr int FEW_NCNT = 1 << 18;   // arrival counters at the start of the similarity workspace (sim_live_kernel): F * ceil(L / 32) of them are used

// (value desc, index asc): the order torch.max(dim) resolves ties in (first maximal index)
__device__ __forceinline__ bool better(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

// The same with torch.max's NaN rule: a NaN is the maximum (the first one wins).  For the exact-fp32 decisions.
__device__ __forceinline__ bool better_nan(float va, int ia, float vb, int ib) {
  const bool na = va != va, nb = vb != vb;
  if (na || nb) return na && (!nb || ia < ib);
  return va > vb || (va == vb && ia < ib);
}

// exclusive prefix of the clamped entity counts into LDS (prefix[Na] = number of live columns); wave 0 works, caller syncs
__device__ __forceinline__ void build_prefix(const int32_t *__restrict__ ent_len, int Na, int Ne, int *prefix) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < Na; base += 64) {
      const int a = base + lane;
      int x = 0;
      if (a < Na) {
        const int l = ent_len[a];
        x = l < 0 ? 0 : (l > Ne ? Ne : l);
      }
      int incl = x;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
      }
      if (a < Na) prefix[a] = carry + incl - x;
      carry += __shfl(incl, 63);
    }
    if (lane == 0) prefix[Na] = carry;
  }
}

// segment a with prefix[a] <= c < prefix[a+1]  (c < prefix[Na])
__device__ __forceinline__ int find_seg(const int *prefix, int Na, int c) {
  int lo = 0, hi = Na;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= c) lo = mid; else hi = mid;
  }
  return lo;
}

// hi = bf16(x), lo = bf16(x - hi), two elements per instruction: v_cvt_pk_bf16_f32 packs a pair, the pair's two halves widened
// back (shift / mask) feed one v_pk_add_f32 -- 10 vector instructions per float4 (element by element hipcc emitted 16: the
// conversions once per element to rebuild float(hi), once more per pair to pack).  Same bits as the element-wise form.
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const f32x4 x, bf16x4 &hi, bf16x4 &lo) {
  const f32x2_ x01 = {x[0], x[1]}, x23 = {x[2], x[3]};
  const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x01, bf16x2_));
  const unsigned p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x23, bf16x2_));
  const f32x2_ r01 = x01 - f32x2_{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)};
  const f32x2_ r23 = x23 - f32x2_{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
  const unsigned q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2_));
  const unsigned q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2_));
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  hi = __builtin_bit_cast(bf16x4, (u32x2_){p0, p1});
  lo = __builtin_bit_cast(bf16x4, (u32x2_){q0, q1});
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// sum over the 64 lanes of a wave (all lanes active), the same bits in every lane: four DPP steps inside each row of 16 lanes
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row sums through v_readlane, added as
// (r0 + r1) + (r2 + r3).  (__shfl_xor lowers to ds_bpermute: six dependent LDS round trips per sum.)
__device__ __forceinline__ float wave_sum(float x) {
  x += dpp_mov<0xB1>(x);
  x += dpp_mov<0x4E>(x);
  x += dpp_mov<0x141>(x);
  x += dpp_mov<0x140>(x);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
  return (r0 + r1) + (r2 + r3);
}

// exact fp32 dot product of two D-float rows by one wave (D % 4 == 0, D <= 1024): lane l takes the float4s at l*4 + 256*t, a
// 4-FMA chain per piece, then wave_sum.  `wf` = this lane's pieces of the query row (loaded once per column).
template <int MAXT>
__device__ __forceinline__ float wave_dot(const float *__restrict__ vrow, const f32x4 (&wf)[MAXT], int D, int lane) {
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < MAXT; t++) {
    const int d = lane * 4 + 256 * t;
    if (d < D) {
      const f32x4 x = *reinterpret_cast<const f32x4 *>(vrow + d);
      acc = fmaf(x[0], wf[t][0], acc);
      acc = fmaf(x[1], wf[t][1], acc);
      acc = fmaf(x[2], wf[t][2], acc);
      acc = fmaf(x[3], wf[t][3], acc);
    }
  }
  return wave_sum(acc);
}


// ---- pieces shared by the frame kernels (simfused.hip, simplanes.hip) ----------------------------------------------------------
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, typename Fn>
__device__ __forceinline__ void unroll_blocks(Fn &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    unroll_blocks<N, I + 1>(f);
  }
}

// max(m, |x0|, |x1|, |x2|, |x3|) in two v_max3_f32 (the |.| are source modifiers)
__device__ __forceinline__ float absmax4(float m, const f32x4 x) {
  asm("v_max3_f32 %0, |%1|, |%2|, %0\n\tv_max3_f32 %0, |%3|, |%4|, %0" : "+v"(m) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
  return m;
}

// workgroup barrier without the vmcnt(0) of __syncthreads(): waves keep global loads / LDS-DMAs in flight across it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// 16-B slot swizzle of a 128-B stage row.  u = (row >> 1) & 7 enumerates the 8 slots over 16 consecutive rows (8 even + 8 odd
// rows: the 16 lanes of a ds_read_b128 group hit all 64 banks once); its bits are ROTATED (u0 -> bit 2) so that the rows r and
// r + 2 a staging wave writes in one ds_write_b64 put their 64-B plane halves into different halves of the row -- with the
// plain value the four even rows of a write shared 16 banks (4 cycles per write instead of 2); the row's parity flips bit 2.
__device__ __forceinline__ int frame_swz(int row) {
  const int u = (row >> 1) & 7;
  return (((u & 1) << 2) | (u >> 1)) ^ ((row & 1) << 2);
}

// Seeded dropout: the keep decision of element i is a pure function of (seed, i) -- a counter-based generator with a 2 x 32-bit
// multiply-xorshift mix (not bit-compatible with torch's Philox stream, which nothing downstream depends on: the reference draws
// its masks from the device generator, model.py:627,641).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool keep_elem(uint64_t seed, uint64_t i, uint32_t thresh) {
  const uint32_t lo = (uint32_t)i, hi = (uint32_t)(i >> 32);
  uint32_t h = mix32(lo ^ (uint32_t)seed);
  h = mix32(h + hi * 0x9e3779b9U + (uint32_t)(seed >> 32));
  return h >= thresh;                 // P(drop) = thresh / 2^32
}

}  // namespace nafae_sim
