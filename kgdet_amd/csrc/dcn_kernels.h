// Kernel declarations shared between the .hip translation units and the C-ABI launchers.
#pragma once
#include "dcn_common.h"

namespace kgdet {

__global__ void dcn_fwd_mfma(const DcnProblem p, float *__restrict__ slabs);
__global__ void dcn_fwd_fixup(const DcnProblem p, const float *__restrict__ slabs, int G);
__global__ void dcn_pack_weight(const float *__restrict__ w, float *__restrict__ wpk, int Og, int Cg, int K,
                                int Cg_pad, int Og_pad);
__global__ void dcn_unpack_weight(const float *__restrict__ wpk, float *__restrict__ w, int Og, int Cg, int K,
                                  int Cg_pad, int Og_pad, int accumulate);

}  // namespace kgdet
