// pna_pack.hip -- row gather for the halo exchange of the destination-sharded multi-GPU path.
// Implements pna_pack_rows_f32 of include/pna_amd.h.
//
// Before the all-to-all every rank packs the rows its peers asked for, grouped by peer, into one contiguous send buffer.
// The reference has no distributed code; this replaces a generic index_select: rows are F floats wide (75, 128, ...), 4-byte
// aligned only, so a lane group of ceil(F/4) lanes moves one row with unaligned dwordx4 accesses and the last lane's window
// slid back to [F-4, F) -- the same trick as the gather kernel -- and rows are streamed with nontemporal stores (the send
// buffer is read once, by the copy engine / the peer).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pna_amd.h"
#include "pna_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { f4 v; };

// out[i, 0:F] = x[idx[i], 0:F]; one lane per 4-float chunk, L lanes per row
__global__ __launch_bounds__(256) void k_pack_rows(const float* x, long ldx, const int32_t* idx, long n, int F, int L, float* out, long ldo) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = gid / L;
  if (row >= n) return;
  const int c = (int)(gid - row * L);
  const long srow = idx[row];
  if (F >= 4) {
    const int off = min(c * 4, F - 4);
    const f4 v = reinterpret_cast<const f4u*>(x + srow * ldx + off)->v;
    typedef f4 f4a4 __attribute__((aligned(4)));
    __builtin_nontemporal_store(v, reinterpret_cast<f4a4*>(out + row * ldo + off));
  } else {
    if (c < F) out[row * ldo + c] = x[srow * ldx + c];
  }
}

}  // namespace

extern "C" int pna_pack_rows_f32(const float* x, int64_t ldx, const int32_t* idx, int64_t n, int32_t F, float* out, int64_t ldo,
                                 pna_stream_t stream) {
  if (n < 0 || F <= 0 || ldx < F || ldo < F) return pna_set_error(PNA_E_INVALID, "pna_pack_rows_f32: bad n / F / leading dimensions");
  if (n == 0) return PNA_OK;
  if (!x || !idx || !out) return pna_set_error(PNA_E_INVALID, "pna_pack_rows_f32: x / idx / out must be non-null");
  const int L = F >= 4 ? (F + 3) / 4 : F;
  const long threads = (long)n * L;
  const long blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffL) return pna_set_error(PNA_E_INVALID, "pna_pack_rows_f32: too many rows for one launch");
  hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, idx, (long)n, F, L, out, (long)ldo);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
