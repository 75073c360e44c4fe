// mtfjsp_gin_res.hip — the single-launch GIN kernel (mtfjsp_gin_resident.h) as a translation unit of its own: hipcc's scheduling strategy is a
// per-file flag, and this kernel gains from max-ILP where the other encoder kernels lose (e2e-mappo-for-mt-fjsp_amd/_build.py: SOURCE_FLAGS).
// mtfjsp_encoder.hip includes the same header with MTFJSP_GIN_RES_DECL_ONLY: the argument struct, the LDS size and the kernel's declaration.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "mtfjsp_enc_shared.h"
#include "mtfjsp_gin_resident.h"
