// micro-benchmark: clocks per row step of the lane-per-row Gauss-Seidel sweep of solve_singles (rv_dev_env.h) on gfx950, one
// wave per SIMD as k_env runs.  16 lanes per body, lane 16 b + 3 p + k = row k of table point p of body b.
//   V = 0  round 4's row step: g-form  nl = med3(lam + (bias - g) invk, lo, hi); d = nl - lam; g += A[s] bcast(d)
//          with the alive / present masks as selects on the chain and the residual tracked per row step
//   V = 1  normalised residual form: rr = (bias - g) invk kept up to date with ONE fma per row step,
//          rr = fma(An[s], bcast(d), rr), An = -(A invk); inert rows are rows of zeros; the residual of a sweep is
//          |lam - lam at the start of the sweep|, taken once per sweep
//   V = 2  as 1, exit test every sweep through v_readlane of a DPP row-max (what the kernel would do)
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench/sweep_step.hip -o /tmp/sweep_step
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int S_> __device__ __forceinline__ float grp_bcast(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + S_, 0xf, 0xf, false));
}
template <int S_> __device__ __forceinline__ float grp_bcast_bc(float x) {
  const int xi = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x150 + S_, 0xf, 0xf, true));
}

template <int V>
__global__ __launch_bounds__(64) void k(const float* in, float* out, unsigned long long* cyc, int sweeps, int ntv) {
  const int lane = threadIdx.x;
  const int b = lane >> 4, r = lane & 15;
  const int rr_ = r < 12 ? r : 11;
  const int p = rr_ / 3, kq = rr_ - 3 * p;
  const int nt = ntv;                      // points per body (runtime)
  const bool act = r < 12 && p < nt;
  float A[12];
#pragma unroll
  for (int s = 0; s < 12; ++s) A[s] = in[64 * s + lane];
  float invk = act ? in[64 * 12 + lane] : 0.0f, bias = kq == 0 ? in[64 * 13 + lane] : 0.0f, mu = in[64 * 14 + lane];
  float lam = act ? in[64 * 15 + lane] : 0.0f, g = in[64 * 16 + lane];
  const float cap = 1e30f;
  const int nmax = nt;
  int done = in[64 * 17] > 100.0f ? 3 : 0;  // (runtime: no island is done)
  const int thr = (int)in[64 * 17 + 1];     // (runtime 0: a residual is never below it -- every sweep runs)
  int iters_done = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (V == 0) {
    for (int it = 0; it < sweeps && done != 15; ++it) {
      const bool alive = act && !((done >> b) & 1);
      const float invk_e = alive ? invk : 0.0f;
      int resv = 0;
#define ROW_STEP(pp_, kk_) { \
      constexpr int s_ = 3 * pp_ + kk_; \
      float nl; \
      if (kk_ == 0) nl = __builtin_amdgcn_fmed3f(lam + (bias - g) * invk_e, 0.0f, cap); \
      else nl = __builtin_amdgcn_fmed3f(lam + (-g * invk_e), -lim, lim); \
      const float d = nl - lam; \
      if (r == s_ && alive) lam = nl; \
      const float sd = grp_bcast<s_>(d); \
      if (kk_ == 0) lim = grp_bcast<s_>(mu * nl); \
      const int mag = __builtin_bit_cast(int, sd) & 0x7fffffff; \
      resv = resv > mag ? resv : mag; \
      if (pp_ < nt && alive) g = g + A[s_] * sd; }
#define POINT(pp_) if (pp_ < nmax) { float lim = 0.0f; ROW_STEP(pp_, 0) ROW_STEP(pp_, 1) ROW_STEP(pp_, 2) }
      POINT(0) POINT(1) POINT(2) POINT(3)
#undef POINT
#undef ROW_STEP
      const int res0 = __builtin_amdgcn_readlane(resv, 0), res1 = __builtin_amdgcn_readlane(resv, 16),
                res2 = __builtin_amdgcn_readlane(resv, 32), res3 = __builtin_amdgcn_readlane(resv, 48);
      done |= (res0 < thr ? 1 : 0) | (res1 < thr ? 2 : 0) | (res2 < thr ? 4 : 0) | (res3 < thr ? 8 : 0);
      ++iters_done;
    }
  } else {
    // normalised form.  Rows that are absent or inert: lam = 0, rr = 0, bounds 0, a row of zeros in An
    float An[12];
#pragma unroll
    for (int s = 0; s < 12; ++s) An[s] = act ? -(A[s] * invk) : 0.0f;
    float rr = act ? (bias - g) * invk : 0.0f;
    const float hi_n = act ? cap : 0.0f;
    const float muv = act ? mu : 0.0f;
    for (int it = 0; it < sweeps && done != 15; ++it) {
      const float lam0 = lam;
#define ROW_STEP(pp_, kk_) { \
      constexpr int s_ = 3 * pp_ + kk_; \
      float nl; \
      if (kk_ == 0) nl = __builtin_amdgcn_fmed3f(lam + rr, 0.0f, hi_n); \
      else nl = __builtin_amdgcn_fmed3f(lam + rr, -lim, lim); \
      const float d = nl - lam; \
      if (r == s_) lam = nl; \
      if (kk_ == 0) lim = grp_bcast_bc<s_>(muv * nl); \
      rr = __builtin_fmaf(An[s_], grp_bcast_bc<s_>(d), rr); }
#define POINT(pp_) if (pp_ < nmax) { float lim = 0.0f; ROW_STEP(pp_, 0) ROW_STEP(pp_, 1) ROW_STEP(pp_, 2) }
      POINT(0) POINT(1) POINT(2) POINT(3)
#undef POINT
#undef ROW_STEP
      if (V == 2) {
        // residual of the sweep: the largest |change| of a row of the island = max over the 16-lane group
        int m = __builtin_bit_cast(int, lam - lam0) & 0x7fffffff;
        m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x128, 0xf, 0xf, true));   // row_ror:8
        m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x124, 0xf, 0xf, true));   // row_ror:4
        m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x122, 0xf, 0xf, true));   // row_ror:2
        m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x121, 0xf, 0xf, true));   // row_ror:1
        const int res0 = __builtin_amdgcn_readlane(m, 0), res1 = __builtin_amdgcn_readlane(m, 16),
                  res2 = __builtin_amdgcn_readlane(m, 32), res3 = __builtin_amdgcn_readlane(m, 48);
        done |= (res0 < thr ? 1 : 0) | (res1 < thr ? 2 : 0) | (res2 < thr ? 4 : 0) | (res3 < thr ? 8 : 0);
      }
      ++iters_done;
    }
    g = rr;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + lane] = lam + g + iters_done;
}

// V = 3: "target form".  T = lam + (bias - g) invk is the unclamped new impulse of a row; a row's own step leaves its T where it
//        is (lam + rr is invariant under the row's own update up to 1 - A_ss invk_s), the other rows' T move by C[r][s] d,
//        C = -(A invk):  nl = med3(T, lo, hi); d = nl - lam; lam = nl; T = fma(C[s], bcast(d), T)  -- chain med3, sub, (dpp) fma
// V = 4: as 3 + per-sweep exit / stall test done by the lanes themselves (DPP row max, best / since in VGPRs, done groups leave
//        through exec), ONE ballot per sweep
template <int V>
__global__ __launch_bounds__(64) void k2(const float* in, float* out, unsigned long long* cyc, int sweeps, int ntv) {
  const int lane = threadIdx.x;
  const int r = lane & 15;
  const int rr_ = r < 12 ? r : 11;
  const int p = rr_ / 3, kq = rr_ - 3 * p;
  const int nt = ntv;
  const bool act = r < 12 && p < nt;
  float A[12];
#pragma unroll
  for (int s = 0; s < 12; ++s) A[s] = in[64 * s + lane];
  const float invk = act ? in[64 * 12 + lane] : 0.0f, bias = kq == 0 ? in[64 * 13 + lane] : 0.0f, mu = in[64 * 14 + lane];
  float lam = act ? in[64 * 15 + lane] : 0.0f; const float g = in[64 * 16 + lane];
  const float cap = 1e30f;
  const int nmax = nt;
  const int thr = (int)in[64 * 17 + 1];
  const int stall = 12 + (int)in[64 * 17 + 2];
  int iters_done = 0;
  float C[12];
#pragma unroll
  for (int s = 0; s < 12; ++s) C[s] = act ? (s == r ? 1.0f - A[s] * invk : -(A[s] * invk)) : 0.0f;
  float T = act ? lam + (bias - g) * invk : 0.0f;
  const float hi_n = act ? cap : 0.0f;
  const float muv = act ? mu : 0.0f;
  int best = 0x7f800000, since = 0;
  bool alive = in[64 * 17] < 100.0f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < sweeps; ++it) {
    if (alive) {
      const float lam0 = lam;
#define ROW_STEP(pp_, kk_) { \
      constexpr int s_ = 3 * pp_ + kk_; \
      float nl; \
      if (kk_ == 0) nl = __builtin_amdgcn_fmed3f(T, 0.0f, hi_n); \
      else nl = __builtin_amdgcn_fmed3f(T, -lim, lim); \
      const float d = nl - lam; \
      if (V == 6) asm volatile("v_cmp_eq_u32 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(lam) : "v"(nl), "n"(s_), "v"(r) : "vcc"); \
      else if (V == 7) { if (s_ == 11) lam = lam + nl; } \
      else if (r == s_) lam = nl; \
      if (kk_ == 0) lim = grp_bcast_bc<s_>(muv * nl); \
      if (V == 8) { T = __builtin_fmaf(C[s_], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), s_)), T); } \
      else if (V == 5) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(T) : "v"(d), "v"(C[s_]), "n"(s_)); \
      else T = __builtin_fmaf(C[s_], grp_bcast_bc<s_>(d), T); }
#define POINT(pp_) if (pp_ < nmax) { float lim = 0.0f; ROW_STEP(pp_, 0) ROW_STEP(pp_, 1) ROW_STEP(pp_, 2) }
      POINT(0) POINT(1) POINT(2) POINT(3)
#undef POINT
#undef ROW_STEP
      if (V == 4) {
        int m = __builtin_bit_cast(int, lam - lam0) & 0x7fffffff;
        m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x128, 0xf, 0xf, true));
        m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x124, 0xf, 0xf, true));
        m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x122, 0xf, 0xf, true));
        m = max(m, __builtin_amdgcn_update_dpp(m, m, 0x121, 0xf, 0xf, true));
        const bool better = m < best;
        best = better ? m : best;
        since = better ? 0 : since + 1;
        if (m < thr || since >= stall) alive = false;
      }
    }
    ++iters_done;
    if (V == 4) { if (__builtin_amdgcn_ballot_w64(alive) == 0) break; }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + lane] = lam + T + iters_done + since;
}

int main() {
  const int NF = 64 * 18;
  float* h = (float*)malloc(NF * sizeof(float));
  srand(1);
  for (int i = 0; i < NF; ++i) h[i] = 0.0f;
  for (int l = 0; l < 64; ++l) {
    for (int s = 0; s < 12; ++s) h[64 * s + l] = (s == (l & 15) ? 12.0f : 0.5f * ((rand() % 200) / 100.0f - 1.0f));
    h[64 * 17 + 2] = 1000000.0f;
    h[64 * 12 + l] = 1.0f / 12.0f; h[64 * 13 + l] = 0.01f; h[64 * 14 + l] = 0.6f; h[64 * 15 + l] = 0.002f; h[64 * 16 + l] = -0.003f;
  }
  float *din, *dout; unsigned long long* dc;
  const int nb = 1024, sweeps = 2000;
  hipMalloc(&din, NF * 4); hipMalloc(&dout, nb * 64 * 4); hipMalloc(&dc, nb * 8);
  hipMemcpy(din, h, NF * 4, hipMemcpyHostToDevice);
  const char* names[9] = {"round-4 row step (g form, selects on the chain, per-step residual, 4 readlanes per sweep)",
                          "normalised residual form, fma + row_newbcast, no exit test",
                          "normalised residual form + per-sweep exit test (DPP row max, 4 readlanes)",
                          "target form: chain med3, sub, fma(row_newbcast); no exit test",
                          "target form + exit / stall test by the lanes (DPP row max, exec), one ballot per sweep",
                          "target form, v_fmac_f32_dpp by inline asm (broadcast folded into the fma); no exit test",
                          "target form, lam update as v_cmp_eq + v_cndmask vcc (inline asm)", "target form, NO lam update (lower bound, wrong)", "target form, v_readlane broadcast (one island only)"};
  unsigned long long* hc = (unsigned long long*)malloc(nb * 8);
  for (int nt = 4; nt >= 3; --nt)
    for (int v = 0; v < 9; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        switch (v) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(nb), dim3(64), 0, 0, din, dout, dc, sweeps, nt); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(nb), dim3(64), 0, 0, din, dout, dc, sweeps, nt); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(nb), dim3(64), 0, 0, din, dout, dc, sweeps, nt); break;
          case 3: hipLaunchKernelGGL(k2<3>, dim3(nb), dim3(64), 0, 0, din, dout, dc, sweeps, nt); break;
          case 5: hipLaunchKernelGGL(k2<5>, dim3(nb), dim3(64), 0, 0, din, dout, dc, sweeps, nt); break;
          case 6: hipLaunchKernelGGL(k2<6>, dim3(nb), dim3(64), 0, 0, din, dout, dc, sweeps, nt); break;
          case 7: hipLaunchKernelGGL(k2<7>, dim3(nb), dim3(64), 0, 0, din, dout, dc, sweeps, nt); break;
          case 8: hipLaunchKernelGGL(k2<8>, dim3(nb), dim3(64), 0, 0, din, dout, dc, sweeps, nt); break;
          case 4: hipLaunchKernelGGL(k2<4>, dim3(nb), dim3(64), 0, 0, din, dout, dc, sweeps, nt); break;
        }
        hipDeviceSynchronize();
      }
      hipMemcpy(hc, dc, nb * 8, hipMemcpyDeviceToHost);
      double sum = 0; for (int i = 0; i < nb; ++i) sum += (double)hc[i];
      float o0; hipMemcpy(&o0, dout, 4, hipMemcpyDeviceToHost);
      printf("nt=%d V%d: %.1f clocks per row step, %.0f per sweep  (%s)  [out %g]\n", nt, v, sum / nb / sweeps / (3.0 * nt), sum / nb / sweeps, names[v], o0);
    }
  return 0;
}
