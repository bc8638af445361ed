// rv_emu.cpp — host lane-emulation of the env kernel program (TEST AID ONLY).
//
// Compiles robovat_amd/csrc/rv_dev_env.h with -DRV_EMULATE: every lane phase
// becomes a loop over 64 lanes.  This lets the CPU-only test-suite exercise
// the *kernel's* decomposition (lanes, phases, LDS layout) against the oracle
// before a GPU is available.  It is never loaded by the product package.
#define RV_EMULATE 1
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../robovat_amd/csrc/rv_dev_env.h"

using namespace rv;

struct EmuWorld {
  rv_config cfg; rv_scene scene; int n; DevEnv* envs;
};

/* LDS is not zero-initialised on the GPU: the scratch block starts as garbage.  NaNs by default; RV_EMU_POISON = another word
   (a finite huge float finds other reads than a NaN does), on words [RV_EMU_POISON_LO, RV_EMU_POISON_HI) */
static void poison_scratch(Shared& S) {
  memset(&S.s, 0xFF, sizeof(S.s));
  if (const char* pat = getenv("RV_EMU_POISON")) {
    const uint32_t v = (uint32_t)strtoul(pat, nullptr, 0);
    const char* lo = getenv("RV_EMU_POISON_LO"); const char* hi = getenv("RV_EMU_POISON_HI");
    const long a = lo ? atol(lo) : 0, b = hi ? atol(hi) : (long)(sizeof(S.s) / 4);
    uint32_t* dst = (uint32_t*)&S.s;
    for (long k = 0; k < (long)(sizeof(S.s) / 4); ++k) if (k >= a && k < b) dst[k] = v + (uint32_t)k * 2654435761u * (v & 1u);   /* (an odd word is hashed per position, as in the kernel's RV_POISON_LDS) */
  }
}

static void run_env(EmuWorld* w, int i, int mode, int n_sub, float lin, float ang, int ca, int ms, int mx) {
  Shared& S = g_shared;
  poison_scratch(S);
  S.cfg = w->cfg; S.arm = w->scene.arm;
  Consts K = lds_consts(&w->scene, 0);
  memcpy(&S.e, &w->envs[i], sizeof(DevEnv));
  if (mode == 1 && S.e.done) { w->envs[i].substeps_last = 0; w->envs[i].awake_last = 0; w->envs[i].pairs_last = 0; w->envs[i].stepped = 0; return; }
  if (mode != 0) env_enter(S, K);
  ProgArgs pa; memset(&pa, 0, sizeof(pa));
  pa.gid = w->cfg.env_id_offset + i; pa.n_steps = n_sub; pa.lin_thr = lin; pa.ang_thr = ang; pa.check_after = ca; pa.min_stable = ms; pa.max_steps = mx;
  pa.env = i; pa.n_envs = w->n; pa.budget = nullptr;
  if (mode == 0) env_program(S, K, RV_PROG_RESET, pa);
  else if (mode == 1) { launch_counters_zero(S.e); env_program(S, K, RV_PROG_MACRO, pa); }
  else if (mode == 4) { pa.first_index = ca; pa.auto_reset = ms; env_program(S, K, RV_PROG_ROLLOUT, pa); }
  else if (mode == 2) { S.e.substeps_last = 0; S.e.awake_last = 0; S.e.pairs_last = 0; S.e.stepped = 0; env_program(S, K, RV_PROG_SUB, pa); }
  else { S.e.substeps_last = 0; S.e.awake_last = 0; S.e.pairs_last = 0; S.e.stepped = 0; env_program(S, K, RV_PROG_WAIT, pa); }
  memcpy(&w->envs[i], &S.e, sizeof(DevEnv));
}

extern "C" {
#ifdef RV_EMU_COUNT
void emu_get_counts(long* out) { for (int i = 0; i < 48; ++i) out[i] = rv_emu_cnt[i]; }
void emu_get_dbg(long* out) { for (int i = 0; i < 16; ++i) out[i] = rv_emu_dbg[i]; }
void emu_get_dbg2(long* out) { for (int i = 0; i < 48; ++i) out[i] = rv_emu_dbg2[i]; }
#endif
EmuWorld* emu_create(const rv_config* cfg, const rv_scene* scene) {
  EmuWorld* w = (EmuWorld*)calloc(1, sizeof(EmuWorld));
  w->cfg = *cfg; w->scene = *scene; w->n = cfg->n_envs;
  w->envs = (DevEnv*)calloc((size_t)w->n, sizeof(DevEnv));
  for (int i = 0; i < w->n; ++i) {
    for (int b = 0; b < RV_MAXB; ++b) w->envs[i].body[b][6] = 1.0f;
    for (int f = 0; f < RV_NFRAME; ++f) w->envs[i].fquat[f][3] = 1.0f;
    w->envs[i].done = 1;
    w->envs[i].mu_finger = cfg->arm_friction; w->envs[i].mu_table = cfg->table_friction;
  }
  return w;
}
void emu_destroy(EmuWorld* w) { if (w) { free(w->envs); free(w); } }
int emu_sizeof_env(void) { return (int)sizeof(DevEnv); }
int emu_sizeof_shared(void) { return (int)sizeof(Shared); }
void emu_reset(EmuWorld* w, const uint8_t* mask) {
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) {
    if (mask && !mask[i]) { w->envs[i].substeps_last = 0; w->envs[i].awake_last = 0; w->envs[i].pairs_last = 0; w->envs[i].stepped = 0; continue; }
    run_env(w, i, 0, 0, 0, 0, 0, 0, 0);
  }
}
void emu_step_macro(EmuWorld* w) {
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) run_env(w, i, 1, 0, 0, 0, 0, 0, 0);
}
void emu_rollout(EmuWorld* w, int n_steps, int first_index, int auto_reset) {
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) {
    if (w->envs[i].done && !auto_reset) { w->envs[i].substeps_last = 0; w->envs[i].awake_last = 0; w->envs[i].pairs_last = 0; w->envs[i].stepped = 0; continue; }
    run_env(w, i, 4, n_steps, 0, 0, first_index, auto_reset, 0);
  }
}
// rv_step_begin / rv_step_poll with a substep budget (the time budget needs the shader clock)
void emu_step_begin(EmuWorld* w, const float* a, const uint8_t* mask) {
  int G = w->cfg.num_goal_steps > 0 ? w->cfg.num_goal_steps : 1;
  for (int i = 0; i < w->n; ++i) {
    DevEnv& e = w->envs[i];
    if ((mask && !mask[i]) || e.in_step == 1) continue;
    for (int g = 0; g < G; ++g) for (int k = 0; k < 4; ++k) e.action[g][k] = a[((size_t)i * G + g) * 4 + k];
    e.in_step = e.done ? 2 : 1; e.step_stage = -1;
  }
}
void emu_step_poll(EmuWorld* w, int max_substeps, uint8_t* finished) {
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) {
    DevEnv& g = w->envs[i];
    g.substeps_last = 0; g.awake_last = 0; g.pairs_last = 0; g.stepped = 0;
    g.l_unsafe = 0; g.l_ineffective = 0; g.l_useful = 0; g.l_episodes = 0; g.l_successes = 0;
    if (g.in_step != 1) { finished[i] = g.in_step == 2; if (g.in_step == 2) { g.in_step = 0; g.reward_valid = 0; } continue; }
    Shared& S = g_shared;
    poison_scratch(S);
    S.cfg = w->cfg; S.arm = w->scene.arm;
    Consts K = lds_consts(&w->scene, 0);
    memcpy(&S.e, &g, sizeof(DevEnv));
    env_enter(S, K);
    S.s.bud_sub = max_substeps; S.s.bud_sub0 = 0; S.s.bud_clk = 0; S.s.bud_t0 = 0;
    { ProgArgs pa; memset(&pa, 0, sizeof(pa)); finished[i] = (uint8_t)env_program(S, K, RV_PROG_PARTIAL, pa); }
    memcpy(&g, &S.e, sizeof(DevEnv));
  }
}
void emu_step_sub(EmuWorld* w, int n) {
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) run_env(w, i, 2, n, 0, 0, 0, 0, 0);
}
void emu_wait_until_stable(EmuWorld* w, float lin, float ang, int ca, int ms, int mx) {
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) run_env(w, i, 3, 0, lin, ang, ca, ms, mx);
}
void emu_set_actions(EmuWorld* w, const float* a) {
  int G = w->cfg.num_goal_steps > 0 ? w->cfg.num_goal_steps : 1;
  for (int i = 0; i < w->n; ++i) for (int g = 0; g < G; ++g) for (int k = 0; k < 4; ++k) w->envs[i].action[g][k] = a[((size_t)i * G + g) * 4 + k];
}
void emu_get_body_state(EmuWorld* w, float* out) {
  for (int i = 0; i < w->n; ++i) for (int b = 0; b < RV_MAXB; ++b) for (int k = 0; k < 13; ++k) out[((size_t)i * RV_MAXB + b) * 13 + k] = w->envs[i].body[b][k];
}
void emu_set_body_state(EmuWorld* w, const float* in) {
  for (int i = 0; i < w->n; ++i) {
    for (int b = 0; b < RV_MAXB; ++b) for (int k = 0; k < 13; ++k) w->envs[i].body[b][k] = in[((size_t)i * RV_MAXB + b) * 13 + k];
    for (int m = 0; m < RV_NMAN; ++m) w->envs[i].man[m].n = 0;
    for (int b = 0; b < RV_MAXB; ++b) { w->envs[i].asleep[b] = 0; w->envs[i].sleep_count[b] = 0; w->envs[i].still_count[b] = 0; w->envs[i].undisturbed[b] = 0; }
  }
}
void emu_get_body_params(EmuWorld* w, float* out) {
  for (int i = 0; i < w->n; ++i) for (int b = 0; b < RV_MAXB; ++b) {
    const DevEnv& e = w->envs[i]; float* o = out + ((size_t)i * RV_MAXB + b) * 8;
    o[0] = (float)e.active[b]; o[1] = (float)e.shape[b]; o[2] = e.scale[b]; o[3] = e.mass[b]; o[4] = e.friction[b]; o[5] = (float)e.frozen[b]; o[6] = e.table_z; o[7] = (float)e.asleep[b];
  }
}
void emu_set_body_params(EmuWorld* w, const float* in) {
  Consts K; K.cfg = &w->cfg; K.arm = &w->scene.arm; K.scene = &w->scene; K.stop_after = 0;
  for (int i = 0; i < w->n; ++i) {
    DevEnv& e = w->envs[i]; int nb = 0;
    for (int b = 0; b < RV_MAXB; ++b) {
      const float* o = in + ((size_t)i * RV_MAXB + b) * 8;
      e.active[b] = (int)o[0]; e.shape[b] = (int)o[1]; e.scale[b] = o[2]; e.friction[b] = o[4]; e.frozen[b] = (int)o[5]; e.asleep[b] = 0; e.sleep_count[b] = 0; e.still_count[b] = 0; e.undisturbed[b] = 0;
      if (b == 0) e.table_z = o[6];
      if (e.active[b]) { body_set_mass(e, K, b, o[3]); nb++; }
    }
    e.n_bodies = nb;
  }
}
void emu_get_joint_state(EmuWorld* w, float* out) {
  for (int i = 0; i < w->n; ++i) for (int j = 0; j < RV_NJ; ++j) { out[((size_t)i * RV_NJ + j) * 2] = w->envs[i].q[j]; out[((size_t)i * RV_NJ + j) * 2 + 1] = w->envs[i].qd[j]; }
}
void emu_get_link_poses(EmuWorld* w, float* out) {
  for (int i = 0; i < w->n; ++i) for (int f = 0; f < RV_NFRAME; ++f) {
    float* o = out + ((size_t)i * RV_NFRAME + f) * 7;
    for (int k = 0; k < 3; ++k) o[k] = w->envs[i].fpos[f][k];
    for (int k = 0; k < 4; ++k) o[3 + k] = w->envs[i].fquat[f][k];
  }
}
void emu_get_env_counters(EmuWorld* w, int32_t* out) {
  for (int i = 0; i < w->n; ++i) {
    const DevEnv& e = w->envs[i]; int32_t* o = out + (size_t)i * RV_NCOUNTERS;
    o[8] = e.awake_last; o[9] = e.pairs_last;
    o[0] = e.sim_steps; o[1] = e.num_steps; o[2] = e.num_episodes; o[3] = e.phase; o[4] = e.done; o[5] = e.is_safe; o[6] = e.is_effective; o[7] = e.substeps_last;
  }
}
void emu_get_manifold_counts(EmuWorld* w, int32_t* out) {
  for (int i = 0; i < w->n; ++i) for (int m = 0; m < RV_NMAN; ++m) out[(size_t)i * RV_NMAN + m] = w->envs[i].man[m].n;
}
void emu_reward(EmuWorld* w, float* r, uint8_t* d) {
  for (int i = 0; i < w->n; ++i) { r[i] = w->envs[i].last_reward; d[i] = (uint8_t)w->envs[i].done; }
}
}
