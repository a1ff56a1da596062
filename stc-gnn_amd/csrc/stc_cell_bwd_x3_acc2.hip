// stc_cell_bwd_x3.hip's kernel in the form that accumulates into the gradient planes of BOTH sides (cell_bwd_x3_kernel<F, 32, 1, 1, 1>), compiled
// WITHOUT -mllvm -amdgpu-mfma-vgpr-form=1: the rewrite pass of this LLVM (AMDGPU Rewrite AGPR-Copy-MFMA) crashes on this one instantiation, and
// the other five forms are 5 % shorter with it (Makefile, EXTRA_*).
#define STC_CB_ACC2_UNIT
#include "stc_cell_bwd_x3.hip"
