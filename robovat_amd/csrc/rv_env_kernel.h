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

// control words of the task queues (ints; every counter on a 128-byte line of its own)
enum { RV_Q_NQ = 8, RV_Q_TAKEN = 0, RV_Q_ERR = 64, RV_Q_CTL_WORDS = 96 + 96 * RV_Q_NQ + 64 };
// measurement words (RV_QUEUE_DEBUG): per XCD the tasks it ran, the envs it kept, when its last workgroup left (s_memtime >> 10)
#define RV_Q_DBG(x) (96 + 96 * RV_Q_NQ + 8 * (x))
#define RV_Q_HEAD(x) (96 + 96 * (x))
#define RV_Q_TAIL(x) (96 + 96 * (x) + 32)
#define RV_Q_FRESH(x) (96 + 96 * (x) + 64)      // first steps handed out of the envs x, x + RV_Q_NQ, x + 2 RV_Q_NQ, ...
// which XCD this wave runs on (0 .. 7)
__device__ __forceinline__ int rv_xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return (int)(x & (unsigned)(RV_Q_NQ - 1));
}

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
  // MODE_ROLLOUT through task queues (worlds with more envs than the GPU has wave slots): a task is ONE env.step() of one
  // env; the launch has as many workgroups as fit the GPU; each takes the next task, runs it from and back to the env's block
  // in HBM, and puts the env back at the tail of ITS XCD's queue while it has steps left.  One queue per XCD (HW_REG_XCC_ID),
  // so that an env's block is only ever handed over between workgroups behind the same L2 (see k_env).  q_ctl: counters
  // (RV_Q_*), q_slots: RV_Q_NQ rings of q_cap slots, slot = env + n_envs x step (-1: not yet published); q_total tasks in
  // all; q_launch: number of this launch (every block it hands over is stamped with it).  nullptr: one workgroup per env
  int* q_slots; int* q_ctl; int q_cap; int q_total; int q_pool; int q_launch;
  int q_steal;                   // 1: a workgroup whose own queue is dry serves the longest other one (k_env); needs q_wt
  int q_global;                  // RV_QUEUE_GLOBAL (measurement aid): every workgroup serves queue 0 -- blocks cross XCDs (needs q_wt)
  int q_debug;                   // RV_QUEUE_DEBUG: fill the measurement words
  int q_sticky;                  // 1: a workgroup keeps an env whose step was a slow one (k_env)
  int q_wt;                      // 1: the block goes out write-through and comes in past the L1 (16-byte sc1 stores / loads): it does not occupy the L2
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
  // ---- the task queues (MODE_ROLLOUT; a plain launch is ONE pass of this loop: its env is the workgroup's own, all steps).
  //
  // Protocol (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").  An env's block goes
  // from the workgroup that ran step k to the one that runs step k + 1.  Both sit on the SAME XCD: a workgroup reads the
  // XCD it runs on from the hardware (HW_REG_XCC_ID -- a fact about where it is, not an assumption about dispatch) and only
  // ever puts an env into, and takes one out of, the queue of that XCD; an env is bound to an XCD by whoever runs its first
  // step (its block was then written by an EARLIER kernel and is visible everywhere).  Behind one L2 the hand-over needs no
  // write-back of that L2 (an agent-scope release fence writes back every dirty line of it -- everybody's scratch included:
  // 8192 envs 121 k -> 116 k env-steps/s) and no write-through traffic:
  //   producer  plain stores of the block (the vector L1 writes through: they land in the XCD's L2) -> asm s_waitcnt vmcnt(0)
  //             (every store acknowledged by the L2; inline asm, which the compiler can neither drop nor move stores across)
  //             -> lane 0 publishes the slot with a relaxed agent-scope atomic store
  //   consumer  lane 0 polls ITS slot with relaxed agent-scope atomic loads (they bypass the L1) -> ONE agent-scope acquire
  //             fence (buffer_inv sc1: drops this CU's L1 lines, which may hold the block as it was steps ago) -> plain loads,
  //             served by the L2 the producer stored into.
  // No checksum, no retry: the step number and the launch number the block carries are ASSERTED (a mismatch raises
  // RV_Q_ERR, which rv_get_stats reports as an error) -- they have not fired since the queues are per XCD.
  // Termination: RV_Q_TAKEN counts the tasks that were begun; once it reaches q_total no task is left to publish, so a
  // workgroup that waits for a slot leaves.  (A work-conserving pool, q_pool: a task is counted when an env was found for
  // it -- an env is put back after every step, the pool ends the launch.)
  const int xcc = (queued && !args.q_global) ? rv_xcc_id() : 0;      // (q_global, a measurement aid: ONE queue for every XCD)
  int* const q_ring = queued ? args.q_slots + (size_t)xcc * (size_t)args.q_cap : nullptr;
  int* const q_head = queued ? args.q_ctl + RV_Q_HEAD(xcc) : nullptr;
  int* const q_tail = queued ? args.q_ctl + RV_Q_TAIL(xcc) : nullptr;
  unsigned fresh_mask = (1u << RV_Q_NQ) - 1u;      // pools (own XCD first) that may still hold an env: lane 0's
  // Which env goes back to the queue and which one does the workgroup keep?  A launch cannot end before the longest chain of
  // steps of ONE env has, so the envs on that chain must not wait in a queue between their steps, while all the others are
  // there to be balanced (list scheduling by the longest remaining chain).  After a task the workgroup compares what is left
  // of this env -- its remaining steps x the duration of the step it just took (a contact-rich state persists) -- with what is
  // left of the launch -- the tasks not yet begun x this workgroup's mean task duration / the number of workgroups --, and
  // keeps the env (runs its next step at once) when the env's rest is more than half of the launch's.  Measured (profiles/
  // r06_*queue_variants*): 8192 envs x 20 steps, throughput-bound, nothing is kept and the queues give + 14 %; 4096 concave
  // envs x 10 steps and 8192 envs without deactivation are bound by their slowest env, where plain FIFO lost 5 - 9 %.
  // keep >= 0: the task to run next without asking the queue.  Not in a pool (there the envs take turns)
  int keep = -1;
  unsigned long long clk_sum = 0ull; int clk_n = 0;
  for (;;) {
    int env = (int)blockIdx.x, k0 = 0;
    if (queued) {
      __syncthreads();
      if (lane == 0) {
        int e = keep;
        // an env nobody has stepped in this launch (its block was written by an earlier kernel: visible everywhere).  The envs
        // are dealt out evenly -- env i belongs to pool i mod 8, a workgroup serves the pool of its own XCD first -- because
        // every env brings the same number of steps: with ONE pool the XCDs ended up with 966 ... 1082 envs each, and the
        // launch with the busiest one.  The pools of the other XCDs are looked at only when the own one is empty (an XCD that
        // is faster takes a few more; a GPU partition without some XCD still runs every env)
        for (int tr = 0; e < 0 && tr < RV_Q_NQ; ++tr) {
          if (!(fresh_mask & (1u << tr))) continue;
          const int px = (xcc + tr) & (RV_Q_NQ - 1);
          const int f = atomicAdd(args.q_ctl + RV_Q_FRESH(px), 1);
          const int cand = px + RV_Q_NQ * f;
          if (cand < args.n_envs) e = cand; else fresh_mask &= ~(1u << tr);
        }
        if (e < 0) {               // the next env of this XCD's queue ...
          int* ring = q_ring; int* head = q_head;
          if (args.q_steal && !args.q_global) {
            // (RV_QUEUE_STEAL=1, off by default) ... unless nothing waits there while another XCD's queue is long: then its head is
            // served and the env moves to this XCD.  A block then crosses XCDs: stored write-through, loaded past the L1 (q_wt: the
            // R1 form of the guide, sc1 payload -> drained -> flag / poll -> sc1 loads; 0 differences in 50 stress runs with ONE
            // queue for all XCDs, where every hand-over crosses: profiles/r06_o_*).  Measured: it buys nothing -- an XCD runs dry
            // only at the very end of a launch, when the other queues are empty too (what is left are kept envs finishing their
            // chains): 100 - 700 steals per XCD, same 40 - 57 ms between the first and the last XCD, same rate (profiles/r06_p_*).
            // So the shipped hand-over never leaves an XCD.
            const int own = __hip_atomic_load(q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - __hip_atomic_load(q_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (own <= 0) {
              int best = 4, bx = -1;      // (a queue shorter than this is left to its own workgroups)
              for (int d = 1; d < RV_Q_NQ; ++d) {
                const int vx = (xcc + d) & (RV_Q_NQ - 1);
                const int len = __hip_atomic_load(args.q_ctl + RV_Q_TAIL(vx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                                __hip_atomic_load(args.q_ctl + RV_Q_HEAD(vx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (len > best) { best = len; bx = vx; }
              }
              if (bx >= 0) { ring = args.q_slots + (size_t)bx * (size_t)args.q_cap; head = args.q_ctl + RV_Q_HEAD(bx); if (args.q_debug) atomicAdd(args.q_ctl + RV_Q_DBG(xcc) + 5, 1); }
            }
          }
          const int t = atomicAdd(head, 1);
          if (t < args.q_cap) {
            while ((e = __hip_atomic_load(&ring[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) {
              if (__hip_atomic_load(args.q_ctl + RV_Q_TAKEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= args.q_total) break;
              __builtin_amdgcn_s_sleep(32);
            }
          }
        }
        if (e >= 0) {              // a task is begun (in a pool: if the pool still has one)
          const int t = atomicAdd(args.q_ctl + RV_Q_TAKEN, 1);
          if (t >= args.q_total) e = -1;
        }
        S.s.loop_break = e;
      }
      __syncthreads();
      env = __builtin_amdgcn_readfirstlane(S.s.loop_break);
      if (env < 0) {
        if (lane == 0 && args.q_debug) { atomicMax(args.q_ctl + RV_Q_DBG(xcc) + 2, (int)((__builtin_amdgcn_s_memrealtime() >> 4) & 0x7fffffffull)); atomicAdd(args.q_ctl + RV_Q_DBG(xcc) + 4, 1); }
        return;
      }
      k0 = env / args.n_envs; env = env - k0 * args.n_envs;      // (a slot says whose turn it is AND which of its steps)
      if (k0 > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (buffer_inv sc1, once per task, before the block is loaded)
    }
    const unsigned long long t_begin = queued ? __builtin_amdgcn_s_memtime() : 0ull;
    rv_env_task<TMODE>(args, MODE, env, S, K, k0, queued ? k0 + 1 : 0);      // (the ONE call site of the env program in this kernel)
    if (!queued) return;
    __syncthreads();                                   // (the env's block is written: every lane's stores are issued)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... and acknowledged by the L2 of this XCD
    const unsigned long long dur = __builtin_amdgcn_s_memtime() - t_begin;
    clk_sum += dur; ++clk_n;
    const bool more = args.q_pool || k0 + 1 < args.n_substeps;
    bool slow = false;
    if (more && !args.q_pool && args.q_sticky) {
      const int begun = __builtin_amdgcn_readfirstlane(__hip_atomic_load(args.q_ctl + RV_Q_TAKEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      const unsigned long long left = (unsigned long long)(begun < args.q_total ? args.q_total - begun : 0);
      const unsigned long long env_rest = (unsigned long long)(args.n_substeps - k0 - 1) * dur * (unsigned long long)gridDim.x;
      slow = 2ull * env_rest * (unsigned long long)clk_n > left * clk_sum;
    }
    keep = slow ? env + args.n_envs * (k0 + 1) : -1;
    if (lane == 0 && args.q_debug) { atomicAdd(args.q_ctl + RV_Q_DBG(xcc), 1); if (slow) atomicAdd(args.q_ctl + RV_Q_DBG(xcc) + 1, 1); if (k0 == 0) atomicAdd(args.q_ctl + RV_Q_DBG(xcc) + 3, 1); }
    if (lane == 0 && more && keep < 0) {
      const int p = atomicAdd(q_tail, 1);
      if (p < args.q_cap) __hip_atomic_store(&q_ring[p], env + args.n_envs * (k0 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else atomicOr(args.q_ctl + RV_Q_ERR, 2);
    }
  }
}

// The env block between HBM and LDS with `sc1` accesses (relaxed agent-scope atomics, one word per lane and access): the stores
// are written through -- the line does not stay in the XCD's L2, which the scratch of the eight resident waves of a CU wants
// for itself --, the loads bypass the vector L1 and are served by the L2 / memory.  (6.4 KB per block: the width of the access
// does not matter.)  Still the same-XCD protocol of k_env: one L2 orders the write-through and the later read of a word.
__device__ __forceinline__ void rv_block_load_sc1(uint32_t* lds, const uint32_t* glb, const int W, const int lane) {
  for (int i = lane; i < W; i += 64) lds[i] = __hip_atomic_load(const_cast<uint32_t*>(&glb[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rv_block_store_sc1(uint32_t* glb, const uint32_t* lds, const int W, const int lane) {
  for (int i = lane; i < W; i += 64) __hip_atomic_store(&glb[i], lds[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(g);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&S.e);
    if (k_stop > 0 && args.q_wt) rv_block_load_sc1(dst, src, W, lane);
    else for (int i = lane; i < W; i += 64) dst[i] = src[i];
    __syncthreads();
    // a task of a queue after the env's first: the block was stored a moment ago by another workgroup of this XCD.  It
    // carries the number of the step it is ready for and of the launch that stored it -- asserted, never repaired
    if (k_stop > 0 && k0 > 0 && lane == 0 && (S.e.q_seq != k0 || S.e.q_sum != (unsigned)args.q_launch)) atomicOr(args.q_ctl + RV_Q_ERR, 1);
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
    if (lane == 0 && args.steps_taken) {
      // (the tasks of an env all run behind one L2: a plain store.  RV_QUEUE_GLOBAL sends them to every XCD: two L2s that each
      // hold a dirty copy of the word write back in either order at the end of the kernel -- an atomic store goes to memory at once)
      if (k_stop > 0 && args.q_global) __hip_atomic_store(&args.steps_taken[env], S.e.stepped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  if (k_stop > 0) {      // a task of a queue: the block leaves with the number of its next step and of this launch
    if (lane == 0) { S.e.q_seq = k0 + 1; S.e.q_sum = (unsigned)args.q_launch; }
    __syncthreads();
  }
  {
    uint32_t* dst = reinterpret_cast<uint32_t*>(g);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&S.e);
    if (k_stop > 0 && args.q_wt) rv_block_store_sc1(dst, src, W, lane);
    else for (int i = lane; i < W; i += 64) dst[i] = src[i];
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
