// pna_collate.hip -- batching + destination-sorted CSR on the device.
// Implements pna_collate_workspace_bytes / pna_collate_csr_i32 of include/pna_amd.h.
//
// The reference collates on the host: dgl.batch offsets and concatenates the member graphs' edge lists
// (realworld_benchmark/data/molecules.py:153-164) and DGL builds its in-edge CSR lazily inside update_all
// (models/dgl/pna_layer.py:64,:202).  For tiny-graph batches that index work, not the layer, is what is left once
// the layer is fast.  Here: one fused offset pass, one STABLE radix sort of (global destination id, edge id) limited
// to the bits the node count needs, one gather pass for the source ids and one boundary pass for rowptr.  Stability
// keeps the original edge order inside a destination's segment: the mailbox order DGL's degree-bucketed reduce sees,
// and the order that decides max / min ties and the fp32 summation order of the segment-reduce kernel.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "pna_amd.h"
#include "pna_internal.h"

namespace {

constexpr int kBlock = 256;

__host__ __device__ inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// keys[k] = global destination id, vals[k] = k, srcg[k] = global source id (member-graph offsets applied)
__global__ void k_keys(const int32_t* src, const int32_t* dst, const int32_t* edge_graph, const int32_t* node_offset,
                       int64_t E, int32_t* keys, int32_t* vals, int32_t* srcg) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < E; k += (int64_t)gridDim.x * blockDim.x) {
    const int32_t off = edge_graph ? node_offset[edge_graph[k]] : 0;
    keys[k] = dst[k] + off;
    vals[k] = (int32_t)k;
    srcg[k] = src[k] + off;
  }
}

// col[p] = source of the p-th CSR edge; rowptr[v] = first p with row[p] >= v (every row between two consecutive
// distinct destinations is empty and gets the same p)
__global__ void k_finish(const int32_t* row, const int32_t* eid, const int32_t* srcg, int64_t E, int32_t V,
                         int32_t* col, int32_t* rowptr) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p <= E; p += (int64_t)gridDim.x * blockDim.x) {
    if (p < E) col[p] = srcg[eid[p]];
    const int32_t lo = p == 0 ? 0 : row[p - 1] + 1;      // rows (row[p-1], row[p]] start at p
    const int32_t hi = p == E ? V : row[p];
    for (int32_t v = lo; v <= hi; ++v) rowptr[v] = (int32_t)p;
  }
}

int key_bits(int32_t V) {
  int b = 1;
  while (b < 31 && ((int64_t)1 << b) < (int64_t)V) ++b;
  return b;
}

size_t sort_temp_bytes(int64_t E, int bits) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                                           (int32_t*)nullptr, (int)E, 0, bits, (hipStream_t)0);
  return bytes;
}

}  // namespace

extern "C" int64_t pna_collate_workspace_bytes(int64_t n_edges, int32_t n_nodes) {
  if (n_edges < 0 || n_edges >= ((int64_t)1 << 31) || n_nodes < 0) return -1;
  return 3 * align256(n_edges * 4) + align256((int64_t)sort_temp_bytes(n_edges, key_bits(n_nodes)));
}

extern "C" int pna_collate_csr_i32(const int32_t* src, const int32_t* dst, int64_t n_edges, int32_t n_nodes,
                                   const int32_t* edge_graph, const int32_t* node_offset, int32_t* rowptr, int32_t* col,
                                   int32_t* eid, int32_t* row, void* workspace, int64_t workspace_bytes, pna_stream_t stream) {
  if (n_edges < 0 || n_edges >= ((int64_t)1 << 31) || n_nodes < 0) return pna_set_error(PNA_E_INVALID, "pna_collate_csr_i32: sizes out of the int32 range");
  if (!rowptr || (n_edges > 0 && (!src || !dst || !col || !eid || !row))) return pna_set_error(PNA_E_INVALID, "pna_collate_csr_i32: null pointer");
  if ((edge_graph == nullptr) != (node_offset == nullptr))
    return pna_set_error(PNA_E_INVALID, "pna_collate_csr_i32: edge_graph and node_offset come together");
  const int64_t need = pna_collate_workspace_bytes(n_edges, n_nodes);
  if (n_edges > 0 && (!workspace || workspace_bytes < need)) return pna_set_error(PNA_E_INVALID, "pna_collate_csr_i32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t E = n_edges;
  char* ws = (char*)workspace;
  int32_t* keys = (int32_t*)ws;
  int32_t* vals = (int32_t*)(ws + align256(E * 4));
  int32_t* srcg = (int32_t*)(ws + 2 * align256(E * 4));
  void* temp = ws + 3 * align256(E * 4);
  const int grid = (int)((E + kBlock) / kBlock > 65535 ? 65535 : (E + kBlock) / kBlock);
  if (E > 0) {
    hipLaunchKernelGGL(k_keys, dim3(grid), dim3(kBlock), 0, st, src, dst, edge_graph, node_offset, E, keys, vals, srcg);
    const int bits = key_bits(n_nodes);
    size_t temp_bytes = sort_temp_bytes(E, bits);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const int32_t*)keys, row, (const int32_t*)vals, eid, (int)E, 0, bits, st);
    if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k_finish, dim3(grid), dim3(kBlock), 0, st, (const int32_t*)row, (const int32_t*)eid, (const int32_t*)srcg, E, n_nodes, col, rowptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pna_set_error(PNA_E_LAUNCH, hipGetErrorString(e));
  return PNA_OK;
}
