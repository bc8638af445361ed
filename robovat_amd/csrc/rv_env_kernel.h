// rv_env_kernel.h — k_env<MODE>: one wave64 per env, the whole reset / macro step / n substeps / settle loop
// out of LDS (rv_dev_env.h).  Included by BOTH translation units of librovat_hip.so:
//   rv_kernels.hip        the register-rich build (one env per SIMD: 1024 envs on an MI355X) and the C ABI
//   rv_kernels_occ2.hip   the same program compiled for two waves per SIMD (amdgpu_waves_per_eu(2, 2): at most 256
//                         registers; the env's LDS block is 20 KB, so eight env workgroups fit a CU) under the
//                         name k_env_occ2 -- launched when a world holds more envs than the GPU has SIMDs, where a
//                         second resident wave hides the LDS / dependent-issue latency of the first
//                         (config 5, 8192 envs: 79 k -> 116 k env-steps/s; profiles/r04_*)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rovat.h"
#include "rv_dev_env.h"
#include "rv_dev_obs.h"

using namespace rv;

enum { MODE_RESET = 0, MODE_MACRO = 1, MODE_SUB = 2, MODE_WAIT = 3, MODE_ROLLOUT = 4, MODE_PARTIAL = 5 };

struct EnvKernelArgs {
  const rv_config* cfg;
  const rv_scene* scene;
  DevEnv* envs;
  const uint8_t* mask;
  int n_envs;
  int n_substeps;
  float lin_thr, ang_thr;
  int check_after, min_stable, max_steps;
  int stop_after;   // profiling hook (env RV_DEBUG_STOP, MODE_SUB only)
  int first_index, auto_reset;   // MODE_ROLLOUT
  RolloutRec rec;                // MODE_ROLLOUT: what every env.step() returns
  int* budget;                   // MODE_ROLLOUT, asynchronous: shared pool of env.step() calls
  int32_t* steps_taken;          // optional [N]
  unsigned long long budget_clk; // MODE_PARTIAL: shader clocks this launch may spend per env (0: no limit)
  uint8_t* finished;             // MODE_PARTIAL: [N] 1 = the env.step() of this env completed in this launch
  int mode;                      // k_env<-1>: which of the modes this launch is
};

// TMODE = MODE_ROLLOUT: the single-launch rollout, a kernel of its own (it is the one bench.py times and the profiles
// name); TMODE = -1: every other mode, picked at run time from args.mode -- the substep loop is inlined once per
// kernel, so two instantiations instead of six keep the build at a minute
template <int TMODE>
#ifdef RV_WAVES_PER_EU      // experiment: cap the registers so that RV_WAVES_PER_EU waves fit a SIMD (tools/flag_variants.sh)
#define RV_ENV_OCC __attribute__((amdgpu_waves_per_eu(RV_WAVES_PER_EU, RV_WAVES_PER_EU)))
#else
#define RV_ENV_OCC
#endif
__global__ __launch_bounds__(64) RV_ENV_OCC void k_env(EnvKernelArgs args) {
  const int MODE = TMODE >= 0 ? TMODE : args.mode;
  Shared& S = g_shared;
  const int env = (int)blockIdx.x;
  if (env >= args.n_envs) return;
  DevEnv* g = args.envs + env;
  const int lane = (int)threadIdx.x;
  {
    // stage the launch constants in LDS
    const uint32_t* src = reinterpret_cast<const uint32_t*>(args.cfg);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&S.cfg);
    for (int i = lane; i < (int)(sizeof(rv_config) / 4); i += 64) dst[i] = src[i];
    src = reinterpret_cast<const uint32_t*>(&args.scene->arm);
    dst = reinterpret_cast<uint32_t*>(&S.arm);
    for (int i = lane; i < (int)(sizeof(rv_arm) / 4); i += 64) dst[i] = src[i];
  }
#ifdef RV_POISON_LDS      // debugging aid (tools/gpu.sh poison): the scratch part of the block starts as garbage that differs from launch
  {                       // to launch, so that a read of a field no phase of THIS launch wrote shows up as a parity failure
    uint32_t* dst = reinterpret_cast<uint32_t*>(&S.s);
    const uint32_t pat = (uint32_t)RV_POISON_LDS;
    for (int i = lane; i < (int)(sizeof(Scratch) / 4); i += 64) dst[i] = pat + (uint32_t)i * 2654435761u * (pat & 1u);
  }
#endif
  Consts K = lds_consts(args.scene, (MODE == MODE_SUB) ? args.stop_after : 0);
  constexpr int W = (int)(sizeof(DevEnv) / 4);
  bool skip = false;
  if (MODE == MODE_RESET) skip = (args.mask != nullptr) && (args.mask[env] == 0);
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(g);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&S.e);
    for (int i = lane; i < W; i += 64) dst[i] = src[i];
  }
  __syncthreads();
#ifdef RV_PROFILE
  if (lane == 0) { S.e.prof_t = __builtin_amdgcn_s_memtime(); for (int g2 = 0; g2 < 4; ++g2) S.e.prof_t2[g2] = S.e.prof_t; }
  __syncthreads();
#endif
  if (MODE == MODE_MACRO) skip = (S.e.done != 0);
  if (MODE == MODE_MACRO || MODE == MODE_ROLLOUT || MODE == MODE_SUB || MODE == MODE_WAIT) {
    // A lock-step entry point on an env that rv_step_poll left in the middle of an env.step() CANCELS that
    // step (include/rovat.h, rv_step_begin): its phase machine is not resumed by a later poll -- without
    // this the next poll would run a second env.step() with the stale action.
    if (S.e.in_step != 0) {
      if (lane == 0) { g->in_step = 0; g->step_stage = -1; }
      __syncthreads();
      if (lane == 0) { S.e.in_step = 0; S.e.step_stage = -1; }
      __syncthreads();
    }
  }
  if (MODE == MODE_PARTIAL) {
    skip = (S.e.in_step != 1) && !(S.e.in_step == 2 && args.auto_reset);
    if (skip && lane == 0) {
      // no step pending; a step that was begun on a finished episode is reported once, with
      // reward 0 and done (rv_step_macro skips such an env the same way)
      const int fin = S.e.in_step == 2;
      if (args.finished) args.finished[env] = (uint8_t)fin;
      if (fin) { g->in_step = 0; g->reward_valid = 0; rollout_record(args.rec, nullptr, (size_t)env, &S.cfg, &S.arm); }   // reward 0, done, zero rows
    }
  }
  if (MODE == MODE_ROLLOUT) skip = (S.e.done != 0) && !args.auto_reset;
  if (skip) {
    // (the counters are per launch: an env the launch skips contributes nothing to rv_get_stats; an
    // env that a macro launch does not step -- its episode is over -- has no step result any more,
    // while a reset that masks it out, rv_step_sub or rv_wait_until_stable leave its reward alone)
    if (lane == 0) launch_counters_zero(*g);
    if (lane == 0 && (MODE == MODE_MACRO || MODE == MODE_ROLLOUT)) g->reward_valid = 0;
    if (MODE == MODE_ROLLOUT && args.budget == nullptr)   // steps not taken: reward 0, done, zero rows
      for (int k = lane; k < args.n_substeps; k += 64) rollout_record(args.rec, nullptr, (size_t)k * args.n_envs + env, &S.cfg, &S.arm);
    return;
  }
  if (MODE != MODE_RESET) env_enter(S, K);
  ProgArgs pa;
  pa.gid = K.cfg->env_id_offset + env; pa.n_steps = args.n_substeps;
  pa.lin_thr = args.lin_thr; pa.ang_thr = args.ang_thr; pa.check_after = args.check_after; pa.min_stable = args.min_stable; pa.max_steps = args.max_steps;
  pa.first_index = args.first_index; pa.auto_reset = args.auto_reset; pa.rec = args.rec; pa.env = env; pa.n_envs = args.n_envs; pa.budget = args.budget;
  // ONE call of the env program per kernel (it holds the only copy of the substep loop)
  const int resetting = MODE == MODE_PARTIAL && RV_UNI(S.e.in_step == 2);   // rv_set_auto_reset: a step begun on a finished episode -> env.reset()
  if (MODE == MODE_MACRO || MODE == MODE_SUB || MODE == MODE_WAIT || (MODE == MODE_PARTIAL && !resetting)) {
    if (lane == 0) {
      launch_counters_zero(S.e);
      if (MODE == MODE_PARTIAL) { S.s.bud_sub = args.n_substeps; S.s.bud_sub0 = 0; S.s.bud_clk = args.budget_clk; S.s.bud_t0 = __builtin_amdgcn_s_memtime(); }
    }
    __syncthreads();
  }
  const int prog = MODE == MODE_RESET || resetting ? RV_PROG_RESET : (MODE == MODE_MACRO ? RV_PROG_MACRO : (MODE == MODE_ROLLOUT ? RV_PROG_ROLLOUT :
                   (MODE == MODE_PARTIAL ? RV_PROG_PARTIAL : (MODE == MODE_SUB ? RV_PROG_SUB : RV_PROG_WAIT))));
  const int fin = env_program(S, K, prog, pa);
  if (MODE == MODE_ROLLOUT) {
    if (lane == 0 && args.steps_taken) args.steps_taken[env] = S.e.stepped;
  } else if (MODE == MODE_PARTIAL) {
    if (resetting) {
      // the poll hands back what env.reset() returns
      __syncthreads();
      if (lane == 0) {
        S.e.in_step = 0; S.e.reward_valid = 0; S.e.last_reward = 0.0f;
        if (args.finished) args.finished[env] = 1;
        rollout_record(args.rec, &S.e, (size_t)env, &S.cfg, &S.arm);
      }
    } else if (lane == 0) {
      if (args.finished) args.finished[env] = (uint8_t)fin;
      if (fin) rollout_record(args.rec, &S.e, (size_t)env, &S.cfg, &S.arm);     // what env.step() returns, for the envs that finished
    }
  }
  __syncthreads();
  {
    uint32_t* dst = reinterpret_cast<uint32_t*>(g);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&S.e);
    for (int i = lane; i < W; i += 64) dst[i] = src[i];
  }
}


// launch k_env<mode> of THIS translation unit
static inline void rv_launch_k_env_here(int mode, const EnvKernelArgs& a0, int n_envs, hipStream_t stream) {
  const dim3 g((unsigned)n_envs), b(64);
  EnvKernelArgs a = a0;
  a.mode = mode;
  if (mode == MODE_ROLLOUT) hipLaunchKernelGGL(k_env<MODE_ROLLOUT>, g, b, 0, stream, a);
  else hipLaunchKernelGGL(k_env<-1>, g, b, 0, stream, a);
}
// ... and of the two-waves-per-SIMD build (rv_kernels_occ2.hip)
void rv_launch_k_env_occ2(int mode, const EnvKernelArgs& a, int n_envs, hipStream_t stream);
