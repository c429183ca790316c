// Second translation unit of the kernels: only twr::rom_kernel (and its host launcher), compiled with
// -mllvm -amdgpu-sched-strategy=max-memory-clause (see Makefile and kernels.hip).
#define TWR_TU_ROM
#include "kernels.hip"
