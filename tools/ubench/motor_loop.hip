// micro-benchmark: cycles per iteration of single-wave dependent chains on gfx950 (one wave per
// SIMD, as k_env runs).  Variants of the joint-motor substep of coast_fused.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench/motor_loop.hip -o tools/ubench/motor_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ float fclampr(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ float fabsr(float x) { return x < 0.0f ? -x : x; }
__device__ __forceinline__ float rdlane(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }

template <int V>
__global__ __launch_bounds__(64) void k(const float* in, float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x;
  const int j = lane < 9 ? lane : 8;
  float q = in[j], qd = in[16 + j];
  const float kp = in[32], mq = in[48 + j], vmax = in[64 + j], amax_dt = in[80 + j], lo = -3.0f, hi = 3.0f;
  const float dt = in[33], inv_dt = 1.0f / dt;
  float trav = 0.0f;
  float cf[9];
  for (int k2 = 0; k2 < 9; ++k2) cf[k2] = in[96 + k2];
  const bool limb = j < 7, mine = lane < 9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int stops = 0;
  for (int it = 0; it < iters; ++it) {
    float vd;
    if (V == 1 || V == 4 || V == 5) vd = kp * (mq - q) * inv_dt; else vd = kp * (mq - q) / dt;
    const float raw = fabsr(vd);
    float sync = 1.0f;
    if (V != 2 && V != 5) {
      const bool sat = limb && mine && raw > vmax;
      if (__builtin_amdgcn_ballot_w64(sat) != 0) {
        float ratio = 1.0f;
        if (sat) ratio = vmax / raw;
        int r = __builtin_bit_cast(int, ratio < 1.0f ? ratio : 1.0f);
        r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x128, 0xf, 0xf, false));
        r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x124, 0xf, 0xf, false));
        r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x122, 0xf, 0xf, false));
        r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x121, 0xf, 0xf, false));
        sync = __builtin_bit_cast(float, r);
      }
    }
    float vdd = vd;
    if (limb) vdd = vdd * sync;
    if (V == 4 || V == 5) vdd = __builtin_amdgcn_fmed3f(vdd, -vmax, vmax); else vdd = fclampr(vdd, -vmax, vmax);
    float dv;
    if (V == 4 || V == 5) dv = __builtin_amdgcn_fmed3f(vdd - qd, -amax_dt, amax_dt); else dv = fclampr(vdd - qd, -amax_dt, amax_dt);
    float qdn = qd + dv;
    float qn = q + qdn * dt;
    if (qn < lo) { qn = lo; qdn = 0.0f; }
    if (qn > hi) { qn = hi; qdn = 0.0f; }
    const float travn = trav + fabsr(qdn) * dt;
    if (V == 3) {
      float T = 0.0f;
#pragma unroll
      for (int k2 = 0; k2 < 9; ++k2) T = __builtin_fmaf(cf[k2], rdlane(travn, k2), T);
      if (__builtin_amdgcn_ballot_w64(T > 1e9f) != 0) { ++stops; break; }
    }
    q = qn; qd = qdn; trav = travn;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + lane] = q + qd + trav + stops;
}

int main() {
  float h[128];
  for (int i = 0; i < 128; ++i) h[i] = 0.0f;
  for (int j = 0; j < 9; ++j) { h[j] = 0.1f * j; h[16 + j] = 0.0f; h[48 + j] = 1.0f + 0.1f * j; h[64 + j] = 0.8f; h[80 + j] = 0.008f; h[96 + j] = 0.1f * (9 - j); }
  h[32] = 0.05f; h[33] = 1e-3f;
  float *din, *dout; unsigned long long* dc;
  const int nb = 1024, iters = 20000;
  hipMalloc(&din, sizeof(h)); hipMalloc(&dout, nb * 64 * 4); hipMalloc(&dc, nb * 8);
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  const char* names[6] = {"as coast_fused: division, ballot + DPP sync, ternary clamps", "x * (1/dt) instead of x / dt", "no saturation path (no ballot / division / DPP)",
                          "as 0 + the 9-readlane travel bound and its ballot", "multiply + v_med3 clamps", "multiply + v_med3, no saturation path"};
  for (int v = 0; v < 6; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (v) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(nb), dim3(64), 0, 0, din, dout, dc, iters); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(nb), dim3(64), 0, 0, din, dout, dc, iters); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(nb), dim3(64), 0, 0, din, dout, dc, iters); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(nb), dim3(64), 0, 0, din, dout, dc, iters); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(nb), dim3(64), 0, 0, din, dout, dc, iters); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(nb), dim3(64), 0, 0, din, dout, dc, iters); break;
      }
      hipDeviceSynchronize();
    }
    unsigned long long hc[1024];
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nb; ++i) s += (double)hc[i];
    printf("variant %d: %7.1f s_memtime ticks per iteration  (%s)\n", v, s / nb / iters, names[v]);
  }
  return 0;
}
