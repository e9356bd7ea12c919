// Attention launchers (see attention.hip).  Argument block = public C-ABI struct (include/s3d_hip.h):
//   qkv [rows][ld]: q | k | v column blocks of D, head-major inside; row(b, t) = b*sb + t*st
//   (timm: sb=N, st=1; seq-first encoder layer: sb=1, st=Nb); lse/delta [Bb*H][N].
#pragma once
#include "common.h"
#include "s3d_hip.h"
typedef S3dAttnArgs AttnArgs;
int s3d_launch_attention_fwd(const AttnArgs& a, bool split, hipStream_t s);
struct AdamFillQueue;        // adam_fill.h: optimizer shares riding on the N <= 32 one-launch backward as filler workgroups
int s3d_launch_attention_bwd(const AttnArgs& a, hipStream_t s, AdamFillQueue* fill = nullptr);
bool s3d_attention_pairs_packed(const AttnArgs& a);   // lse / delta of this problem use the pair-packed layout (see attention.hip)
