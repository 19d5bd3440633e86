// pna_common.hip -- error reporting and version query of the C ABI (include/pna_amd.h).
#include <string.h>

#include "pna_amd.h"
#include "pna_internal.h"

static thread_local char g_err[512] = "";

int pna_set_error(int code, const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return code;
}

extern "C" const char* pna_last_error(void) { return g_err; }
extern "C" int pna_abi_version(void) { return PNA_ABI_VERSION; }
