// march_flat_bwdvol.hip -- the translation unit of B1, the brick-centric backward with a d_volume gradient box
// (brick_flat_kernel<bwd, vol[, tf]> and its work-item twin): march_flat.hip compiled a second time with only that kernel
// family's launch function, so that it can have its own scheduler strategy (Makefile: -mllvm
// -amdgpu-sched-strategy=iterative-minreg; B1 -2.5 % on the same device, everything else in march_flat.o is 2-4 % slower under it).
#define DR_FLAT_TU_BWDVOL 1
#include "march_flat.hip"
