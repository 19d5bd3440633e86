// pna_common.hip -- error reporting and version query of the C ABI (include/pna_amd.h).
#include <stdio.h>
#include <string.h>

#include "pna_amd.h"
#include "pna_internal.h"

static thread_local char g_err[512] = "";

int pna_set_error(int code, const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return code;
}

int pna_check_struct_size(const char* fn, unsigned got, unsigned long need) {
  if (got >= need) return PNA_OK;
  char msg[256];
  snprintf(msg, sizeof(msg), "%s: args.struct_size = %u, this library's struct has %lu bytes (a binding built against an older include/pna_amd.h, or struct_size not set)", fn, got, need);
  return pna_set_error(PNA_E_INVALID, msg);
}

extern "C" const char* pna_last_error(void) { return g_err; }
extern "C" int pna_abi_version(void) { return PNA_ABI_VERSION; }
