// lmpc_lib.hip -- single translation unit of liblmpc_hip.so (kernels + C ABI), so that the device
// code is compiled without relocatable-device-code linking and -Rpass-analysis reports per kernel.
#include "lmpc_prep_kernels.hip"
#include "lmpc_solve_kernel.hip"
#include "lmpc_ss_kernel.hip"
#include "lmpc_reg_kernel.hip"
#include "lmpc_sqp_kernel.hip"
#include "lmpc_capi.hip"
