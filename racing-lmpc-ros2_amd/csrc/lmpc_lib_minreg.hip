// lmpc_lib_minreg.hip -- second translation unit of liblmpc_hip.so: the mixed-precision learning kernels
// lmpc_solve_kernel<float, 4, {2, 3}, double> (BASELINE configs[4]), compiled with the minimum-register instruction scheduler
// (Makefile: -mllvm -amdgpu-sched-strategy=iterative-minreg).  The flag is per compilation, so they get a compilation of their own;
// lmpc_lib.hip sees them as `extern template` and takes their address from here.
#define LMPC_MINREG_TU
#include "lmpc_solve_kernel.hip"
