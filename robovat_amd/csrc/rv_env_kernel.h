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
  // MODE_ROLLOUT through a task queue (worlds with more envs than the GPU has wave slots): a task is ONE env.step() of one
  // env; the launch has as many workgroups as fit the GPU, each takes the next task, runs it from and back to the env's block
  // in HBM, and puts the env back at the tail while it has steps left.  q_slots[t] = env of task t (-1: not yet published),
  // q_total = n_envs x n_steps tasks in all.  nullptr: one workgroup per env, all its steps (the plain launch)
  int* q_slots; unsigned* q_head; unsigned* q_tail; int* q_done; int q_total;
  int poison_lo, poison_hi;      // RV_POISON_LDS builds: the words of the scratch block that start as garbage (RV_POISON_LO / _HI: bisecting)
};

// TMODE = MODE_ROLLOUT: the single-launch rollout, a kernel of its own (it is the one bench.py times and the profiles
// name); TMODE = -1: every other mode, picked at run time from args.mode -- the substep loop is inlined once per
// kernel, so two instantiations instead of six keep the build at a minute
template <int TMODE>
__device__ __forceinline__ void rv_env_task(const EnvKernelArgs& args, const int MODE, const int env, Shared& S, const Consts& K, const int k0, const int k_stop);
template <int TMODE>
#ifdef RV_WAVES_PER_EU      // experiment: cap the registers so that RV_WAVES_PER_EU waves fit a SIMD (tools/flag_variants.sh)
#define RV_ENV_OCC __attribute__((amdgpu_waves_per_eu(RV_WAVES_PER_EU, RV_WAVES_PER_EU)))
#else
#define RV_ENV_OCC
#endif
__global__ __launch_bounds__(64) RV_ENV_OCC void k_env(EnvKernelArgs args) {
  const int MODE = TMODE >= 0 ? TMODE : args.mode;
  Shared& S = g_shared;
  const bool queued = TMODE < 0 && args.q_slots != nullptr;      // (the run-time-dispatched instantiation only)
  if (!queued && (int)blockIdx.x >= args.n_envs) return;
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
    for (int i = lane; i < (int)(sizeof(Scratch) / 4); i += 64) if (i >= args.poison_lo && i < args.poison_hi) dst[i] = pat + (uint32_t)i * 2654435761u * (pat & 1u);
  }
#endif
  Consts K = lds_consts(args.scene, (MODE == MODE_SUB) ? args.stop_after : 0);
  // ---- the task queue (MODE_ROLLOUT; a plain launch is ONE pass of this loop: its env is the workgroup's own, all steps).
  // Memory: a block is handed on with agent-scope ATOMIC stores and loads of its words (written through to memory / read
  // from memory, past the XCD's L2), the slot that names the next task is published after those stores have been
  // acknowledged (s_waitcnt vmcnt(0)), and the taker checks the block's step number and the sum of its words before it
  // believes it (rv_env_task).  What was tried before: agent-scope release / acquire fences -- correct, but they write back and
  // invalidate the XCD's whole L2 per task, everybody's scratch lines included (8192 envs: 121 k -> 116 k env-steps/s); and
  // uncached device memory for the blocks -- fast, but intermittently wrong or stalled for seconds, and a hipFree of such
  // memory corrupted later allocations of the process (ROCm 7.0).
  for (;;) {
    int env = (int)blockIdx.x, k0 = 0;
    if (queued) {
      __syncthreads();
      if (lane == 0) {
        int e = -1;
        const unsigned t = atomicAdd(args.q_head, 1u);
        if (t < (unsigned)args.q_total) {
          while ((e = __hip_atomic_load(&args.q_slots[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) __builtin_amdgcn_s_sleep(16);
        }
        S.s.loop_break = e;
      }
      __syncthreads();
      env = __builtin_amdgcn_readfirstlane(S.s.loop_break);
      if (env < 0) return;
      k0 = env / args.n_envs; env = env - k0 * args.n_envs;      // (a slot says whose turn it is AND which of its steps)
    }
    rv_env_task<TMODE>(args, MODE, env, S, K, k0, queued ? k0 + 1 : 0);      // (the ONE call site of the env program in this kernel)
    if (!queued) return;
    __syncthreads();                                   // (the env's block is written: every lane's stores are issued)
    // (the block has reached memory: every store of the wave acknowledged.  A workgroup-scope release fence is NOT that -- for
    // a workgroup of one wave it compiles to nothing --, and the hand-over lost a step now and then: 13 823 of 13 824)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    if (lane == 0) {
      if (k0 + 1 < args.n_substeps) {
        const unsigned p = atomicAdd(args.q_tail, 1u);
        __hip_atomic_store(&args.q_slots[p], env + args.n_envs * (k0 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// sum of the words of the env block in LDS, the checksum word itself left out (every lane gets the total)
__device__ __forceinline__ unsigned rv_block_sum(const Shared& S) {
  constexpr int W = (int)(sizeof(DevEnv) / 4), QS = (int)(offsetof(DevEnv, q_sum) / 4);
  const uint32_t* w = reinterpret_cast<const uint32_t*>(&S.e);
  unsigned acc = 0u;
  for (int i = (int)threadIdx.x; i < W; i += 64) acc += (i == QS) ? 0u : w[i];
  for (int o = 32; o > 0; o >>= 1) acc += (unsigned)__shfl_xor((int)acc, o);
  return acc;
}
// One env, one run of the env program: load its block into LDS, run, store it back.  k0 / k_stop: MODE_ROLLOUT as a task
// of the queue (the steps k0 .. k_stop - 1; 0 / 0: all steps).  Inlined at its single call site per branch of the kernel.
template <int TMODE>
__device__ __forceinline__ void rv_env_task(const EnvKernelArgs& args, const int MODE, const int env, Shared& S, const Consts& K, const int k0, const int k_stop) {
  DevEnv* g = args.envs + env;
  const int lane = (int)threadIdx.x;
  constexpr int W = (int)(sizeof(DevEnv) / 4);
  bool skip = false;
  if (MODE == MODE_RESET) skip = (args.mask != nullptr) && (args.mask[env] == 0);
  for (;;) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(g);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&S.e);
    // (a task of the queue: word by word with agent-scope atomic loads -- they go to memory, past the L2 of this XCD, which
    // may hold what this block was several steps ago)
    if (k_stop > 0) { for (int i = lane; i < W; i += 64) dst[i] = __hip_atomic_load(const_cast<uint32_t*>(&src[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else { for (int i = lane; i < W; i += 64) dst[i] = src[i]; }
    __syncthreads();
    // a task of the queue after the env's first: the block was stored by another workgroup a moment ago.  It is taken only
    // when it carries the number of this step and its words add up to the sum its last owner left -- a block that is still
    // on its way (the old one, or a mix) is read again.  (s_waitcnt vmcnt(0) before the hand-over made a stale read rare,
    // not impossible: one step in ~80 000 tasks was still lost in the stress run, tools/queue_stress.py)
    if (k_stop == 0 || k0 == 0) break;
    const unsigned sum = rv_block_sum(S);
    const int ok = __builtin_amdgcn_readfirstlane((int)(S.e.q_seq == k0 && S.e.q_sum == sum));
    if (ok) break;
    __syncthreads();
    __builtin_amdgcn_s_sleep(8);
  }
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
      if (lane == 0 && k_stop == 0) { g->in_step = 0; g->step_stage = -1; }      // (a task of the queue touches the block in HBM at its two ends only)
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
    // (a task of the queue hands the block on through LDS like any other task: the changes go there)
    DevEnv* tgt = k_stop > 0 ? &S.e : g;
    if (lane == 0 && k0 == 0) launch_counters_zero(*tgt);
    if (lane == 0 && (MODE == MODE_MACRO || MODE == MODE_ROLLOUT)) tgt->reward_valid = 0;
    if (MODE == MODE_ROLLOUT && args.budget == nullptr)   // steps not taken: reward 0, done, zero rows
      for (int k = k0 + lane; k < (k_stop > 0 ? k_stop : args.n_substeps); k += 64) rollout_record(args.rec, nullptr, (size_t)k * args.n_envs + env, &S.cfg, &S.arm);
    if (k_stop == 0) return;
  }
  if (!skip) {
  if (MODE != MODE_RESET) env_enter(S, K);
  ProgArgs pa;
  pa.gid = K.cfg->env_id_offset + env; pa.n_steps = args.n_substeps;
  pa.lin_thr = args.lin_thr; pa.ang_thr = args.ang_thr; pa.check_after = args.check_after; pa.min_stable = args.min_stable; pa.max_steps = args.max_steps;
  pa.first_index = args.first_index; pa.auto_reset = args.auto_reset; pa.rec = args.rec; pa.env = env; pa.n_envs = args.n_envs; pa.budget = args.budget;
  pa.k0 = k0; pa.k_stop = k_stop;
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
    // (through the queue the env's tasks run on different XCDs, each with an L2 of its own: plain stores of the same word
    // from two of them may reach memory in either order at the end of the kernel -- an atomic store goes there at once)
    if (lane == 0 && args.steps_taken) {
      if (k_stop > 0) __hip_atomic_store(&args.steps_taken[env], S.e.stepped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else args.steps_taken[env] = S.e.stepped;
    }
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
  }      // (!skip)
  __syncthreads();
  if (k_stop > 0) {      // a task of the queue: the block leaves with its step number and the sum of its words
    if (lane == 0) S.e.q_seq = k0 + 1;
    __syncthreads();
    const unsigned sum = rv_block_sum(S);
    if (lane == 0) S.e.q_sum = sum;
    __syncthreads();
  }
  {
    uint32_t* dst = reinterpret_cast<uint32_t*>(g);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&S.e);
    // (a task of the queue: agent-scope atomic stores -- written through to memory, where the next owner's loads look)
    if (k_stop > 0) { for (int i = lane; i < W; i += 64) __hip_atomic_store(&dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else { for (int i = lane; i < W; i += 64) dst[i] = src[i]; }
  }
}


// launch k_env<mode> of THIS translation unit
// (n_grid: one workgroup per env -- or, for a rollout through the task queue, as many as fit the GPU)
static inline void rv_launch_k_env_here(int mode, const EnvKernelArgs& a0, int n_grid, hipStream_t stream) {
  const dim3 g((unsigned)n_grid), b(64);
  EnvKernelArgs a = a0;
  a.mode = mode;
  static const bool force_dispatch = getenv("RV_FORCE_DISPATCH") != nullptr;      // (measurement aid)
  if (mode == MODE_ROLLOUT && a.q_slots == nullptr && !force_dispatch) hipLaunchKernelGGL(k_env<MODE_ROLLOUT>, g, b, 0, stream, a);
  else hipLaunchKernelGGL(k_env<-1>, g, b, 0, stream, a);
}
// workgroups of the run-time-dispatched kernel that are resident on one CU at a time (the queue's grid is this x CUs)
static inline int rv_k_env_blocks_per_cu_here() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_env<-1>, 64, 0) != hipSuccess) nb = 0;
  return nb;
}
// ... and of the two-waves-per-SIMD build (rv_kernels_occ2.hip)
void rv_launch_k_env_occ2(int mode, const EnvKernelArgs& a, int n_grid, hipStream_t stream);
int rv_k_env_occ2_blocks_per_cu();
