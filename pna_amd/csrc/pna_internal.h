// Internal helpers shared by the translation units of libpna_amd.so (not part of the C ABI).
#ifndef PNA_INTERNAL_H
#define PNA_INTERNAL_H
// Records `msg` as the calling thread's last error and returns `code`.
int pna_set_error(int code, const char* msg);
#endif
