// Internal (not part of the C-ABI): the 32-rows-per-wave attention kernels of attn32.hip, dispatched to by sdlt_attn_fwd / sdlt_attn_bwd (attn.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/sdlt_kernels.h"
bool sdlt_attn32_ok(const sdlt_attn_params& p);
int sdlt_attn32_fwd(const sdlt_attn_params& p, int ks, hipStream_t s);
int sdlt_attn32_bwd_both(const sdlt_attn_params& p, int ks, hipStream_t s);
