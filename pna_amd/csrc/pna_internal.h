// Internal helpers shared by the translation units of libpna_amd.so (not part of the C ABI).
#ifndef PNA_INTERNAL_H
#define PNA_INTERNAL_H
// Records `msg` as the calling thread's last error and returns `code`.
int pna_set_error(int code, const char* msg);
// PNA_E_INVALID with both sizes in the message when a caller's args struct is shorter than this library's (0 = fine).
int pna_check_struct_size(const char* fn, unsigned got, unsigned long need);
#endif
