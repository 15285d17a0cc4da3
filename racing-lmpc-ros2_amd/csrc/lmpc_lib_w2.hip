// lmpc_lib_w2.hip -- third translation unit of liblmpc_hip.so: the two-wavefronts-per-problem kernels lmpc_solve_kernel_w2<7 | 11 | 14>
// (csrc/lmpc_solve_w2.hip.h; the fp64 tracking problem for N >= 24).  lmpc_lib.hip sees them as `extern template` and takes their
// address from here.
#define LMPC_W2_TU
#include "lmpc_solve_kernel.hip"
