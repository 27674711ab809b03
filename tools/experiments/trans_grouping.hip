// trans_grouping.hip - does it matter HOW the transcendental instructions of the LIF step are spread among the packed ones?
// 64-instruction blocks of 56 packed (fma : mul clamp : add = 4 : 2 : 1) + 8 transcendental (rcp / log alternating), the
// transcendentals in runs of 1, 2, 4 or 8; all instructions independent (eight accumulators each).  One workgroup per CU,
// 2 / 3 / 4 waves per SIMD; SIMD time per wave64 instruction from the launch's wall time (as tools/valu_issue_rate.hip).
//   build: hipcc -O3 --offload-arch=gfx950 tools/experiments/trans_grouping.hip -o /tmp/trans_grouping
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PK(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i) & 7]) : "v"(q0), "v"(q1));
#define PM(i) asm volatile("v_pk_mul_f32 %0, %0, %1 clamp" : "+v"(p[(i) & 7]) : "v"(q0));
#define PA(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i) & 7]) : "v"(q1));
#define P7(i) PK(i) PM(i + 1) PK(i + 2) PA(i + 3) PK(i + 4) PM(i + 5) PK(i + 6)
#define TR(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[(i) & 7]));
#define TL(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[(i) & 7]));
template <int RUN>
__global__ __launch_bounds__(1024) void k(float* sink, int iters) {
  extern __shared__ unsigned char pad[];
  float a[8]; f32x2 p[8];
  const f32x2 q0 = {1.0000001f, 0.9999999f}, q1 = {1e-9f, -1e-9f};
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = 1.0f + threadIdx.x * 1e-6f + i; p[i] = (f32x2){a[i], a[i] + 0.5f}; }
  for (int it = 0; it < iters; ++it) {
    if constexpr (RUN == 1) { P7(0) TR(0) P7(7) TL(1) P7(14) TR(2) P7(21) TL(3) P7(28) TR(4) P7(35) TL(5) P7(42) TR(6) P7(49) TL(7) }
    else if constexpr (RUN == 2) { P7(0) P7(7) TR(0) TR(1) P7(14) P7(21) TL(2) TL(3) P7(28) P7(35) TR(4) TR(5) P7(42) P7(49) TL(6) TL(7) }
    else if constexpr (RUN == 4) { P7(0) P7(7) P7(14) P7(21) TR(0) TR(1) TR(2) TR(3) P7(28) P7(35) P7(42) P7(49) TL(4) TL(5) TL(6) TL(7) }
    else if constexpr (RUN == 8) { P7(0) P7(7) P7(14) P7(21) P7(28) P7(35) P7(42) P7(49) TR(0) TR(1) TR(2) TR(3) TL(4) TL(5) TL(6) TL(7) }
    else { P7(0) P7(7) P7(14) P7(21) P7(28) P7(35) P7(42) P7(49) PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) }      // RUN 0: 64 packed, no transcendental
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 123.456f) sink[0] = s;
}
template <int RUN>
static void run(int threads, int n_cu, float* sink) {
  const int iters = 20000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<RUN>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<RUN>), dim3(n_cu), dim3(threads), 96 * 1024, 0, sink, iters / 8);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<RUN>), dim3(n_cu), dim3(threads), 96 * 1024, 0, sink, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  printf("runs of %d transcendental(s) per 56 : 8 block   waves/SIMD %d   SIMD ns per wave-instruction %.3f\n", RUN, threads / 256, 1e6 * ms / ((double)iters * 64 * (threads / 256)));
}
int main() {
  hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  float* sink; hipMalloc(&sink, 64);
  for (int threads : {256, 512, 768, 1024}) {
    run<0>(threads, prop.multiProcessorCount, sink); run<1>(threads, prop.multiProcessorCount, sink); run<2>(threads, prop.multiProcessorCount, sink);
    run<4>(threads, prop.multiProcessorCount, sink); run<8>(threads, prop.multiProcessorCount, sink);
  }
  return 0;
}
