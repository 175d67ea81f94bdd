// Error plumbing shared by all entry points of libsdlt_kernels.so.
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <hip/hip_runtime.h>
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
    case 12: return (int)sizeof(sdlt_ln_slabs_params);
    case 13: return (int)sizeof(sdlt_ta_group);
    case 14: return (int)sizeof(sdlt_affine_grad_item);
    case 15: return (int)sizeof(sdlt_wgrad_tr_item);
    case 16: return (int)sizeof(sdlt_ln_fold_desc);
    case 17: return (int)sizeof(sdlt_colsum_finish_desc);
    case 18: return (int)sizeof(sdlt_wsk_gemm_params);
  }
  return -1;
}

// hipFuncAttributeMaxDynamicSharedMemorySize applies to the device that is current when it is set: every kernel that needs more than the default
// 64 KB of dynamic LDS raises its limit once per (kernel, device).  Thread-safe (the header promises thread-compatible entry points; the table is
// only written under the mutex) and checked: a failed call returns -1 and is retried by the next launch instead of being remembered as done.
int sdlt_raise_smem(const void* fn, int bytes) {
  struct Entry { const void* fn; int dev; int bytes; };
  static Entry table[1024];
  static int n = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < n; ++i)
    if (table[i].fn == fn && table[i].dev == dev && table[i].bytes >= bytes) return 0;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
  if (n < 1024) table[n++] = Entry{fn, dev, bytes};
  return 0;
}
