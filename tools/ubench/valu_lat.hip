// micro-benchmark: clocks per VALU instruction of ONE wave per SIMD on gfx950 (s_memtime), for the instruction kinds on the
// critical paths of k_env: dependent / independent v_add_f32, v_fma, v_med3, v_mov_dpp row_newbcast, v_readlane -> VALU use,
// v_cndmask, ds_read dependent chains.   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_lat.hip -o tools/ubench/valu_lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int V>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, int iters, float seed) {
  float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0001f, d = 0.25f, e2 = 3.0f, f = 5.0f;
  __shared__ float lds[256];
  lds[threadIdx.x] = threadIdx.x * 4.0f; lds[64 + threadIdx.x] = 0; lds[128 + threadIdx.x] = 0; lds[192 + threadIdx.x] = 0;
  __syncthreads();
  int idx = (threadIdx.x * 4) & 255;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (V == 0) asm volatile(REP64("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
    if (V == 1) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(c), "v"(b));
    if (V == 2) asm volatile(REP64("v_med3_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(c), "v"(b));
    if (V == 3) asm volatile(REP8("v_add_f32 %0, %0, %4\nv_add_f32 %1, %1, %4\nv_add_f32 %2, %2, %4\nv_add_f32 %3, %3, %4\nv_add_f32 %0, %0, %4\nv_add_f32 %1, %1, %4\nv_add_f32 %2, %2, %4\nv_add_f32 %3, %3, %4\n") : "+v"(a), "+v"(d), "+v"(e2), "+v"(f) : "v"(b));
    if (V == 4) asm volatile(REP8(REP8("v_add_f32 %0, %0, %1\ns_nop 1\nv_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) : "+v"(a) : "v"(b));
    if (V == 5) asm volatile(REP8(REP8("v_add_f32 %0, %0, %1\nv_readlane_b32 s20, %0, 3\nv_add_f32 %0, s20, %0\n")) : "+v"(a) : "v"(b) : "s20");
    if (V == 6) asm volatile(REP64("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(b) : "vcc");
    if (V == 7) asm volatile(REP8(REP8("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\n")) : "+v"(idx) : : "memory");
    if (V == 8) asm volatile(REP8(REP8("v_add_f32 %0, %0, %1\ns_nop 1\nv_fmac_f32_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) : "+v"(a) : "v"(b));
    if (V == 9) asm volatile(REP64("v_add_f32 %0, %0, %1\ns_add_u32 s20, s20, 1\n") : "+v"(a) : "v"(b) : "s20");
    if (V == 10) asm volatile(REP8(REP8("v_cmp_lt_f32 vcc, %0, %1\nv_cndmask_b32 %0, %0, %1, vcc\n")) : "+v"(a) : "v"(b) : "vcc");
    if (V == 11) asm volatile(REP8(REP8("v_cmp_lt_f32 s[20:21], %0, %1\ns_and_b64 s[20:21], s[20:21], exec\ns_cbranch_scc0 4\nv_add_f32 %0, %0, %1\n")) : "+v"(a) : "v"(b) : "s20", "s21", "scc");
    if (V == 12) asm volatile(REP8(REP8("v_pk_add_f32 %0, %0, %1\n")) : "+v"(*(double*)&a) : "v"(*(double*)&b));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = a + d + e2 + f + idx;
}
int main() {
  float* dout; unsigned long long* dc; const int nb = 8192, iters = 200;
  hipMalloc(&dout, nb * 64 * 4); hipMalloc(&dc, nb * 8);
  unsigned long long* hc = (unsigned long long*)malloc(nb * 8);
  const char* nm[13] = {"dependent v_add_f32", "dependent v_fma_f32", "dependent v_med3_f32", "4 independent v_add chains", "v_add -> s_nop 1 -> v_mov_dpp row_newbcast (pair)",
                        "v_add -> v_readlane -> v_add with the SGPR (triple)", "dependent v_cndmask (vcc)", "dependent ds_read_b32 + waitcnt", "v_add -> s_nop 1 -> v_fmac_dpp (pair)",
                        "dependent v_add interleaved with s_add (pair)", "v_cmp -> v_cndmask (pair)", "v_cmp sgpr -> s_and -> branch -> v_add (group)", "dependent v_pk_add_f32"};
  const int per[13] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};
  for (int nbv = 0; nbv < 4; ++nbv) {
    const int blocks = 1024 << nbv;      // 1, 2, 4, 8 waves per SIMD (256 CUs x 4 SIMDs)
    for (int v = 0; v < 13; ++v) {
      if (v == 9 || v == 11) continue;      // (these two clobber an SGPR the loop counter lives in: not measurements)
      for (int rep = 0; rep < 2; ++rep) {
#define L(n) case n: hipLaunchKernelGGL(k<n>, dim3(blocks), dim3(64), 0, 0, dout, dc, iters, 1.5f); break;
        switch (v) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) }
        hipDeviceSynchronize();
      }
      hipMemcpy(hc, dc, blocks * 8, hipMemcpyDeviceToHost);
      double sum = 0; for (int i = 0; i < blocks; ++i) sum += (double)hc[i];
      sum = sum * 1024 / blocks;      // (mean over the blocks, in the units of the print below)
      printf("blocks=%d V%-2d %6.2f clocks per unit per wave = %5.2f per SIMD  (%s)\n", blocks, v, sum / 1024 / iters / per[v], sum / 1024 / iters / per[v] / (blocks / 1024), nm[v]);
    }
  }
  return 0;
}
