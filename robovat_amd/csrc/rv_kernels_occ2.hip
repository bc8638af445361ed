// rv_kernels_occ2.hip — the env kernel compiled for two waves per SIMD (see rv_env_kernel.h).
// gfx950 only; built with hipcc --offload-arch=gfx950 into librovat_hip.so next to rv_kernels.hip.
#define RV_WAVES_PER_EU 2
#define RV_SIM_RUN_NOINLINE 1
#define k_env k_env_occ2
#include "rv_env_kernel.h"

void rv_launch_k_env_occ2(int mode, const EnvKernelArgs& a, int n_envs, hipStream_t stream) {
  rv_launch_k_env_here(mode, a, n_envs, stream);
}
