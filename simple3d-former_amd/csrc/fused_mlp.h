// norm2 + fc1 + GELU + fc2 + residual of a timm Block in one launch, D = 192 (fused_mlp.hip)
#pragma once
#include "common.h"
#include "fused_block.h"

bool s3d_fused_mlp_full_ok(long M, int D, int hidden);
// a: the fields of the norm2 + fc1 launch (FusedMlpArgs; hact_lo is not written: nothing reads it once fc2 runs inside); then fc2 and the output
int s3d_launch_fused_mlp_full(const FusedMlpArgs& a, const bf16_t* w2_hi, const bf16_t* w2_lo, const float* b2, float* x_out, int D, hipStream_t s);
