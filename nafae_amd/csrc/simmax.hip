// simmax.hip -- entry points and routing of the region x query similarity reduced to per-frame max / arg-max (DVSA.forward,
// reference model.py:548-551, 580-583, 610-612), gfx950.  The kernels: simfused.hip (sim_live_kernel: fp32 matrix cores, any number
// of live columns in blocks of 32, ONE launch), simplanes.hip (sim_planes_kernel: many live columns, the operands as matrix-core
// planes -- attached by their producer, or split here in a pre-pass -- as a filter with an exact fp32 finish per frame) and
// simloss.hip (sim_max_kernel: the exact-fp32 first-generation kernel, the fallback for the shapes those do not take).  In every
// route only the LIVE query slots (e < ent_len[a]) are contracted -- the reference zero-fills the S_ column of a padded slot after
// the product (model.py:551), ~85 % of the columns at 2.07 entities per segment -- S_ is never written, S_max is an fp32 dot
// product and D_ind follows torch.max (first maximal row, NaN first).
//
// (History: round 2's part / tile + finish kernels assumed |V|, |W| <= 1 in their filter margin and left in round 3; round 3's
// sim_frame_kernel converted fp32 -> bf16 hi / lo inside its k-loop, 59 us at C5 with every slot live, and left in round 4 when the
// planes kernel took its shapes: 31 us with fp16 planes from the producer, ~43 us including this file's pre-pass.)
//
// Algorithmic bytes: 4*D*(R+Q) + 12*F*Q.  Everything is deterministic (no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "hip_util.h"
#include "sim_common.h"

using namespace nafae;
using namespace nafae_sim;

namespace nafae_sim {          // simfused.hip
int64_t few_workspace_bytes(int F, int Nb, int Q);
int launch_few(const float *V, const float *W, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int Lh, float *S_max,
               int64_t *D_ind, void *workspace, hipStream_t st);
int launch_planes_narrow(const float *V, const float *W, const void *Vp, const void *Wp, const float *vstat, const float *wstat,
                         const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int Qh, float *S_max, int64_t *D_ind,
                         hipStream_t st);
int64_t planes_scratch_bytes(int R, int Q, int D);
int launch_frames_prepass(const float *V, const float *W, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int Qh,
                          float *S_max, int64_t *D_ind, void *scratch, hipStream_t st);
// simplanes.hip
int launch_planes_frames(const float *V, const float *W, const void *Vp, const void *Wp, const float *vstat, const float *wstat,
                         int kind, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D, int Qh, float *S_max,
                         int64_t *D_ind, hipStream_t st);
}  // namespace nafae_sim

namespace {
// Live columns up to which the fp32-MFMA kernel is taken when the frame kernel applies as well: every 32-column block of it is one
// more pass over V (out of L2) and 4-6 us of fp32 MFMA issue at C5, the frame kernel starts at ~30 us.  Measured (graph time per
// call, live / frame kernel): 33-64 columns C5 28-29 / 34 us, C4 23-24 / 27-28, C2 15 / 19-20; 96 columns C5 41 / 44, C4 32 / 28;
// 128 columns C5 49 / 44, C4 42 / 28, C2 23 / 20 (scripts/sim_sweep_live.py).
constexpr int LIVE_MAX_DEFAULT = 64;

// 0: the exact-fp32 fallback (simloss.hip), 1: sim_live_kernel, 2: sim_planes_kernel (simplanes.hip)
inline int fused_route(int F, int Nb, int Na, int Ne, int D, int Qh) {
  if (D % 32 != 0 || D > 512 || Na > NA_MAX || F < 1 || Nb < 1) return 0;
  if ((long)Nb * D >= (1L << 30) || (long)Na * Ne * D >= (1L << 30)) return 0;      // 32-bit element offsets inside the kernels
  int live_max = LIVE_MAX_DEFAULT;
  if (const char *e = nafae::experiment_env("NAFAE_SIM_LIVE_MAX")) live_max = atoi(e);
  const bool frame_ok = Nb > 64 && D % 64 == 0;      // (the frame kernel's staging loop takes the 32-k chunks two at a time)
  if (Qh <= live_max || !frame_ok) return ((Qh + 31) / 32 <= 65535 && (long)F * ((Qh + 31) / 32) <= FEW_NCNT) ? 1 : 0;
  return 2;
}
}  // namespace

extern "C" {

// exact-fp32 first-generation kernel (simloss.hip): the fallback for shapes this file does not take
int nafae_sim_max_fwd_frames(const float *V, const float *W, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D,
                             float *S_max, int64_t *D_ind, void *stream);

int64_t nafae_sim_max_workspace_bytes(int F, int Nb, int Na, int Ne, int D) {
  if (F <= 0 || Nb <= 0 || Na <= 0 || Ne <= 0 || D <= 0) return NAFAE_EINVAL;
  // the live-column route's counters + records, or -- many live columns, fp32 operands only -- the counters' MiB (kept: one
  // zeroed region whatever the route) + the operand planes of the pre-pass
  const int64_t few = nafae_sim::few_workspace_bytes(F, Nb, Na * Ne);
  const int64_t pre = (int64_t)FEW_NCNT * 4 + nafae_sim::planes_scratch_bytes(F * Nb, Na * Ne, D);
  return few > pre ? few : pre;
}

// Would nafae_sim_max_fwd_planes READ operand planes of `kind` for this shape (1), or ignore them and run the fp32 live-column /
// fallback kernels (0)?  The producers (VisEbd / WordEbd epilogues) ask before writing planes nobody reads.
int nafae_sim_planes_used(int F, int Nb, int Na, int Ne, int D, int max_live_cols, int kind) {
  if (Na <= 0 || F <= 0 || Nb <= 0 || Ne <= 0 || D <= 0 || (D & 3)) return 0;
  if (kind != NAFAE_SIMPLANES_F16 && kind != NAFAE_SIMPLANES_BF16X3) return 0;
  const int Q = Na * Ne;
  int Qh = (max_live_cols < 0 || max_live_cols > Q) ? Q : max_live_cols;
  if (Qh < 1) Qh = 1;
  const char *ne = nafae::experiment_env("NAFAE_SIM_NARROW");
  const bool narrow_on = ne ? ne[0] != '0' : Nb >= 224;
  if (narrow_on && kind == NAFAE_SIMPLANES_F16 && Qh <= 64 && Nb > 32 && D % 128 == 0 && D <= 512 && Na <= NA_MAX &&
      (long)Nb * D < (1L << 30) && (long)Q * D < (1L << 30))
    return 1;
  return fused_route(F, Nb, Na, Ne, D, Qh) == 2 && D % (kind == NAFAE_SIMPLANES_F16 ? 128 : 64) == 0 ? 1 : 0;
}

int nafae_sim_max_fwd_ws(const float *V, const float *W, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D,
                         int max_live_cols, float *S_max, int64_t *D_ind, void *workspace, int64_t workspace_bytes,
                         void *stream) {
  if (!V || !W || !ent_len || !S_max || !D_ind) return NAFAE_EINVAL;
  if (Na <= 0 || F <= 0 || Nb <= 0 || Ne <= 0 || D <= 0 || (D & 3)) return NAFAE_EINVAL;
  if ((long)F * Na * Ne > (1L << 31) - 256) return NAFAE_ELIMIT;
  const int Q = Na * Ne;
  int Qh = (max_live_cols < 0 || max_live_cols > Q) ? Q : max_live_cols;
  if (Qh < 1) Qh = 1;
  const int route = fused_route(F, Nb, Na, Ne, D, Qh);
  if (route == 1) {
    if (!workspace || workspace_bytes < nafae_sim::few_workspace_bytes(F, Nb, Qh)) return NAFAE_EINVAL;
    return nafae_sim::launch_few(V, W, ent_len, F, Nb, Na, Ne, D, Qh, S_max, D_ind, workspace, as_stream(stream));
  }
  if (route == 2) {   // fp32 operands only: split them into planes here (a pre-pass into the workspace), then the planes kernel
    if (!workspace || workspace_bytes < (int64_t)FEW_NCNT * 4 + nafae_sim::planes_scratch_bytes(F * Nb, Q, D)) return NAFAE_EINVAL;
    return nafae_sim::launch_frames_prepass(V, W, ent_len, F, Nb, Na, Ne, D, Qh, S_max, D_ind,
                                            reinterpret_cast<unsigned char *>(workspace) + (size_t)FEW_NCNT * 4, as_stream(stream));
  }
  return nafae_sim_max_fwd_frames(V, W, ent_len, F, Nb, Na, Ne, D, S_max, D_ind, stream);
}

int nafae_sim_max_fwd_planes(const float *V, const float *W, const int32_t *ent_len, int F, int Nb, int Na, int Ne, int D,
                             int max_live_cols, int kind, const void *V_planes, const float *V_stats, const void *W_planes,
                             const float *W_stats, float *S_max, int64_t *D_ind, void *workspace, int64_t workspace_bytes,
                             void *stream) {
  if (!V || !W || !ent_len || !S_max || !D_ind) return NAFAE_EINVAL;
  if (Na <= 0 || F <= 0 || Nb <= 0 || Ne <= 0 || D <= 0 || (D & 3)) return NAFAE_EINVAL;
  if (kind != NAFAE_SIMPLANES_BF16X3 && kind != NAFAE_SIMPLANES_F16) return NAFAE_EINVAL;
  if ((long)F * Na * Ne > (1L << 31) - 256) return NAFAE_ELIMIT;
  const int Q = Na * Ne;
  int Qh = (max_live_cols < 0 || max_live_cols > Q) ? Q : max_live_cols;
  if (Qh < 1) Qh = 1;
  const bool have = V_planes && V_stats && W_planes && W_stats;
  // few live columns + fp16 planes (what the embedding modules attach) + LONG frames: the narrow planes kernel.  Measured (hipGraph of
  // back-to-back calls, data set's entity histogram): C5 (300 proposals) 17.0 us against sim_live_kernel's 17.9, C4 (256) 13.8 / 15.4,
  // C2 (128) 13.5 / 11.4 -- one workgroup per frame streams half the bytes but has the whole frame's latency chain to itself, so
  // short frames stay on the fp32 live-column kernel, which splits a frame over row-block workgroups.  (NAFAE_SIM_NARROW=0/1 in the
  // experiments build forces one or the other.)
  {
    const char *ne = nafae::experiment_env("NAFAE_SIM_NARROW");
    const bool narrow_on = ne ? ne[0] != '0' : Nb >= 224;
    if (have && narrow_on && kind == NAFAE_SIMPLANES_F16 && Qh <= 64 && Nb > 32 && D % 128 == 0 && D <= 512 && Na <= NA_MAX &&
        (long)Nb * D < (1L << 30) && (long)Q * D < (1L << 30))
      return nafae_sim::launch_planes_narrow(V, W, V_planes, W_planes, V_stats, W_stats, ent_len, F, Nb, Na, Ne, D, Qh, S_max, D_ind,
                                             as_stream(stream));
  }
  // (the kernel takes the 128-byte lines of a row two at a time: D % 128 == 0 for fp16 planes, D % 64 == 0 for bf16x3 ones)
  if (have && fused_route(F, Nb, Na, Ne, D, Qh) == 2 && D % (kind == NAFAE_SIMPLANES_F16 ? 128 : 64) == 0)
    return nafae_sim::launch_planes_frames(V, W, V_planes, W_planes, V_stats, W_stats, kind, ent_len, F, Nb, Na, Ne, D, Qh, S_max,
                                           D_ind, as_stream(stream));
  return nafae_sim_max_fwd_ws(V, W, ent_len, F, Nb, Na, Ne, D, max_live_cols, S_max, D_ind, workspace, workspace_bytes, stream);
}

}  // extern "C"
