// rv_kernels_occ2.hip — the env kernel compiled for two waves per SIMD (see rv_env_kernel.h).
// gfx950 only; built with hipcc --offload-arch=gfx950 into librovat_hip.so next to rv_kernels.hip.
#define RV_WAVES_PER_EU 2
#ifdef RV_OCC2_LOOP_OUT_OF_LINE     // (round 4's arrangement, kept as a build variant for measurements: tools/gpu.sh variants)
#define RV_SIM_RUN_NOINLINE 1
#else
#define RV_SEGMENTS_NOINLINE 1      // the loop inlined into the kernel, the segments of the env program out of line: see env_program
#endif
#ifdef RV_OCC2_COAST_OUT_OF_LINE   // (round 6, measured and rejected: the coasting run as a function of its own -- coast_run_fn.  Its loops then carry
#define RV_COAST_NOINLINE 1        // no reload, but the call costs more than they did: config 5 160 k -> 146 k, config 4 27.5 k -> 24.5 k; profiles/r06_k_*)
#endif
#define k_env k_env_occ2
#include "rv_env_kernel.h"

void rv_launch_k_env_occ2(int mode, const EnvKernelArgs& a, int n_grid, hipStream_t stream) {
  rv_launch_k_env_here(mode, a, n_grid, stream);
}
int rv_k_env_occ2_blocks_per_cu() { return rv_k_env_blocks_per_cu_here(); }
