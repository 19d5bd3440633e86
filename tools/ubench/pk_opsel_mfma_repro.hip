// pk_opsel_mfma_repro.hip -- reproducer of the packed-fp32 wrong sums.  MI355X (gfx950), ROCm 7.2.
//
// FINDING.  A VOP3P packed-fp32 instruction (v_pk_add_f32, v_pk_fma_f32) whose LOW lane selects the HIGH half of SRC1
// (op_sel[1] = 1, e.g. `v_pk_add_f32 v[a:a+1], v[a:a+1], v[m:m+1] op_sel:[0,1] op_sel_hi:[1,0]`: lo += m.hi, hi += m.lo)
// intermittently DROPS the update of its low-half result in lanes 48-63 while ANOTHER wavefront of the same SIMD issues MFMAs
// (v_mfma_f32_16x16x32_bf16).  The same arithmetic with the op_sel on src0, with the swap done by v_pk_mov_b32 in front of a plain
// packed add, with any op_sel_hi (high-lane) selection, or with single v_add_f32 is exact; a co-resident wavefront that only
// reads LDS or copies global -> LDS does not trigger it; s_nop around the instruction changes nothing; the register numbers
// (v16.. or v232..) and the kernel's VGPR count (42 or 248) do not matter.  hipcc emits exactly this form when it vectorises
// `s += m` over register pairs whose halves are crossed -- the round-2 fused kernel's fold (wrong sums in lanes 48-63 of one VGPR).
//
// One 512-thread workgroup per CU = two wavefronts per SIMD.  Wavefronts 0-3 (role M): mode bits 1 MFMAs, 2 ds_read_b128, 4
// LDS-DMA copies, 0 idle.  Wavefronts 4-7 (role F): per iteration one 16-byte load (x y z w), then with explicit registers
//   s0,s1 += (x, y)   plain v_pk_add_f32          q pairs: plain packed mul + add (q2,q3 behind a v_pk_mov_b32 swap)
//   s2,s3 += (w, z)   THE INSTRUCTION UNDER TEST (SWZ_INSTR)
// and the same sums again with single v_add_f32 / v_mul_f32 from the same loaded values; differing bits are counted per sum and
// per lane quarter, and the first bad lane's s2 is recomputed on the host (the single-instruction value is the right one).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w tools/ubench/pk_opsel_mfma_repro.hip -o tools/ubench/pk_opsel_mfma_repro
//   tools/ubench/pk_opsel_mfma_repro [iterations] [launches]
// Variants: -DSWZ_INSTR='"v_pk_add_f32 v[242:243], v[234:235], v[242:243] op_sel:[1,0] op_sel_hi:[0,1]\n"' (op_sel on src0: exact),
//   '"v_pk_mov_b32 v[236:237], v[234:235], v[234:235] op_sel:[1,0]\n v_pk_add_f32 v[242:243], v[242:243], v[236:237]\n"' (exact).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#ifndef SWZ_INSTR
#define SWZ_INSTR "v_pk_add_f32 v[242:243], v[242:243], v[234:235] op_sel:[0,1] op_sel_hi:[1,0]\n"
#endif
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short bf8 __attribute__((ext_vector_type(8)));
#define REGS "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247"

__global__ __launch_bounds__(512, 1) void k_repro(const float* __restrict__ x, int rows, int iters, int mode, float* sink, unsigned* bad,
                                                  const unsigned char* wimg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) {                                          // role M
    if (mode == 0) return;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf8 a, b[3];
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + lane + j); for (int r = 0; r < 3; ++r) b[r][j] = (short)(0x3f00 + r + j); }
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<f4*>(lds)[i] = (f4){1.f, 2.f, 3.f, 4.f};
    __builtin_amdgcn_s_waitcnt(0);
    for (int it = 0; it < iters; ++it) {
      if (mode & 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wimg + ((size_t)(it & 63) * 4096 + wave * 1024 + lane * 16)),
                                         (__attribute__((address_space(3))) void*)(lds + (it & 3) * 16384 + wave * 1024), 16, 0, 0);
      if (mode & 2)
        for (int r = 0; r < 3; ++r) b[r] = *reinterpret_cast<const bf8*>(lds + ((it & 3) * 16384 + r * 1024 + lane * 16));
      if (mode & 1) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[r % 3], acc[n], 0, 0, 0);
      } else {
        asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
        acc[0][0] += (float)b[0][0];
      }
    }
    sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    return;
  }
  float Sr[4] = {0, 0, 0, 0}, Qr[4] = {0, 0, 0, 0};         // role F: packed sums in v[240:243] (s) and v[244:247] (q), operands in v[232:239]
  asm volatile("v_mov_b32 v240, 0\n v_mov_b32 v241, 0\n v_mov_b32 v242, 0\n v_mov_b32 v243, 0\n v_mov_b32 v244, 0\n v_mov_b32 v245, 0\n v_mov_b32 v246, 0\n v_mov_b32 v247, 0" ::: REGS);
  unsigned row = (blockIdx.x * 8 + wave) * 64 + lane;
  for (int it = 0; it < iters; ++it) {
    row = (row * 1664525u + 1013904223u);
    const f4 v = *reinterpret_cast<const f4*>(x + (size_t)(row % (unsigned)rows) * 4);
    asm volatile("v_mov_b32 v232, %0\n v_mov_b32 v233, %1\n v_mov_b32 v234, %2\n v_mov_b32 v235, %3\n"
                 "v_pk_add_f32 v[240:241], v[240:241], v[232:233]\n"
                 "v_pk_mul_f32 v[236:237], v[232:233], v[232:233]\n"
                 "v_pk_add_f32 v[244:245], v[244:245], v[236:237]\n" SWZ_INSTR
                 "v_pk_mov_b32 v[238:239], v[234:235], v[234:235] op_sel:[1,0]\n"
                 "v_pk_mul_f32 v[236:237], v[238:239], v[238:239]\n"
                 "v_pk_add_f32 v[246:247], v[246:247], v[236:237]" : : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : REGS);
    const float mm[4] = {v.x, v.y, v.w, v.z};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s1, p1, q1;
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(Sr[j]), "v"(mm[j]));
      asm volatile("v_mul_f32 %0, %1, %1" : "=v"(p1) : "v"(mm[j]));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(q1) : "v"(Qr[j]), "v"(p1));
      Sr[j] = s1; Qr[j] = q1;
    }
  }
  unsigned ps[4], pq[4];
  asm volatile("v_mov_b32 %0, v240\n v_mov_b32 %1, v241\n v_mov_b32 %2, v242\n v_mov_b32 %3, v243\n v_mov_b32 %4, v244\n v_mov_b32 %5, v245\n v_mov_b32 %6, v246\n v_mov_b32 %7, v247"
               : "=v"(ps[0]), "=v"(ps[1]), "=v"(ps[2]), "=v"(ps[3]), "=v"(pq[0]), "=v"(pq[1]), "=v"(pq[2]), "=v"(pq[3]) : : REGS);
  unsigned d = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned ds = ps[j] != __builtin_bit_cast(unsigned, Sr[j]), dq = pq[j] != __builtin_bit_cast(unsigned, Qr[j]);
    if (ds) atomicAdd(bad + 8 + j, 1u);
    if (dq) atomicAdd(bad + 12 + j, 1u);
    d += ds + dq;
  }
  if (d) {
    atomicAdd(bad, d); atomicAdd(bad + 24 + (lane >> 4), 1u);
    if (atomicAdd(bad + 1, 1u) == 0) { bad[2] = blockIdx.x; bad[3] = threadIdx.x; bad[4] = ps[2]; bad[5] = __builtin_bit_cast(unsigned, Sr[2]); }
  }
  sink[blockIdx.x * 512 + threadIdx.x] = __builtin_bit_cast(float, ps[0]);
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 8192, launches = argc > 2 ? atoi(argv[2]) : 10, rows = 1 << 23;
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  float *x, *sink; unsigned* bad; unsigned char* wimg;
  hipMalloc(&x, (size_t)rows * 16); hipMalloc(&sink, (size_t)cus * 2048); hipMalloc(&bad, 256); hipMalloc(&wimg, 65 * 4096);
  hipMemset(wimg, 0x3c, 65 * 4096);
  std::vector<float> hx((size_t)rows * 4);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k_repro, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int modes[] = {7, 1, 2, 4, 0};
  const char* names[] = {"MFMA + ds_read + LDS-DMA", "MFMA only", "ds_read only", "LDS-DMA only", "idle"};
  for (int mi = 0; mi < 5; ++mi) {
    hipMemset(bad, 0, 256);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k_repro, dim3(cus), dim3(512), 65536, 0, x, rows, iters, modes[mi], sink, bad, wimg);
    hipDeviceSynchronize();
    unsigned hb[64];
    hipMemcpy(hb, bad, 256, hipMemcpyDeviceToHost);
    printf("other wavefront of the SIMD: %-26s %d launches x %d CUs x %d iterations: %u lanes with a differing sum%s\n", names[mi], launches, cus, iters,
           hb[1], hb[1] ? "  <-- MISMATCH" : "");
    if (!hb[1]) continue;
    printf("   by sum: s0..3 %u %u %u %u  q0..3 %u %u %u %u | by lane quarter: %u %u %u %u\n", hb[8], hb[9], hb[10], hb[11], hb[12], hb[13], hb[14], hb[15], hb[24],
           hb[25], hb[26], hb[27]);
    unsigned row = (hb[2] * 8 + (hb[3] >> 6)) * 64 + (hb[3] & 63);
    float e = 0.f;
    for (int it = 0; it < iters; ++it) { row = row * 1664525u + 1013904223u; e += hx[(size_t)(row % (unsigned)rows) * 4 + 3]; }
    printf("   first: block %u thread %u: s2 packed %08x, single %08x, host %08x -> the %s value is wrong\n", hb[2], hb[3], hb[4], hb[5],
           __builtin_bit_cast(unsigned, e), __builtin_bit_cast(unsigned, e) == hb[5] ? "PACKED" : __builtin_bit_cast(unsigned, e) == hb[4] ? "SINGLE" : "(neither matches)");
  }
  return 0;
}
