// Error plumbing shared by all entry points of libsdlt_kernels.so.
#include <cstdarg>
#include <cstdio>
#include "../../include/sdlt_kernels.h"

static thread_local char g_err[512] = "";

void sdlt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sdlt_last_error(void) { return g_err; }
extern "C" int sdlt_abi_version(void) { return 1; }

// sizeof() of the parameter structs, so a binding can verify its mirror of the layout (tests/test_capi_symbols.py)
extern "C" int sdlt_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(sdlt_gemm_params);
    case 1: return (int)sizeof(sdlt_lora_grad_desc);
    case 2: return (int)sizeof(sdlt_attn_params);
    case 3: return (int)sizeof(sdlt_groupnorm_params);
    case 4: return (int)sizeof(sdlt_shadow_desc);
    case 5: return (int)sizeof(sdlt_gemm_batch_item);
    case 6: return (int)sizeof(sdlt_dora_desc);
    case 7: return (int)sizeof(sdlt_dora_wt_desc);
    case 8: return (int)sizeof(sdlt_dora_grad_desc);
    case 9: return (int)sizeof(sdlt_splitsum_desc);
    case 10: return (int)sizeof(sdlt_strip_params);
    case 11: return (int)sizeof(sdlt_ta_params);
  }
  return -1;
}
