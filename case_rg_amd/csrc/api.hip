// Error reporting and version for libcase_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

static thread_local char g_err[512] = "";

int case_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int case_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return CASE_OK;
  return case_set_error(CASE_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

// ABI generation: bumped with every struct or signature change (round 1: 100; round 2: +case_gemm_dw_bias, decode, optimizer; round 3:
// CaseOptTensor grew to 64 bytes, K16 / K17 entry points; round 4: K18 / K19, scratch / workspace queries, reserved CUs).
// case_rg_amd/_abi.py refuses a library whose generation differs from the one it was written against.
extern "C" int case_version(void) { return CASE_ABI_VERSION; }
extern "C" uint32_t case_abi_features(void) {
  return CASE_FEAT_GEMM_256 | CASE_FEAT_GEMM_SMALL | CASE_FEAT_ENCODER_CHAIN | CASE_FEAT_ATTN_SCORES | CASE_FEAT_ATTN_DECODE | CASE_FEAT_OPTIM |
         CASE_FEAT_ATTN_RESIDENT | CASE_FEAT_RESERVED_CUS | CASE_FEAT_GEMM_DW_SLABS | CASE_FEAT_ATTN_DECODE_MQA | CASE_FEAT_POINTER_DECODE | CASE_FEAT_POINTER_HEAD | CASE_FEAT_GEMM_LN | CASE_FEAT_STEP_STATE | CASE_FEAT_INTERACTION | CASE_FEAT_ATTN_DECODE_APPEND | CASE_FEAT_LINEAR_SKINNY;
}
extern "C" const char* case_last_error(void) { return g_err; }

// ---- the one piece of mutable library configuration: compute units the persistent kernels (gemm8w, K16, K17, K18, K19) leave free,
// so that RCCL's kernels can be resident beside them during backward when world_size > 1 (case_rg_amd.parallel.GradSync sets it) ----
static int g_reserved_cus = [] {
  const char* e = getenv("CASE_RESERVE_CUS");
  return e ? atoi(e) : 0;
}();
extern "C" int case_set_reserved_cus(int n) {
  if (n < 0 || n > 128) return case_set_error(CASE_E_ARG, "case_set_reserved_cus: %d out of range (0 .. 128)", n);
  g_reserved_cus = n;
  return CASE_OK;
}
extern "C" int case_get_reserved_cus(void) { return g_reserved_cus; }
int case_device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  return cus;
}
int case_persistent_cus() {  // whole XCD rounds (a multiple of 8), at least 8
  int c = case_device_cus() - g_reserved_cus;
  c = c / 8 * 8;
  return c > 8 ? c : 8;
}
extern "C" int case_sizeof_opt_tensor(void) { return (int)sizeof(CaseOptTensor); }
