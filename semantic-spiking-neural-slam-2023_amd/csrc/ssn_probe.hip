// ssn_probe.hip - vector-ALU issue-rate probe behind ssn_probe_issue_rate() (include/ssn.h).
//
// Not on the step path: bench.py calls it next to the timed region so that the VALU-issue roofline of k_ens_block is priced
// with rates measured IN THE SAME RUN on the same GPU at the clock that GPU sustains under this kind of load (VERDICT r2,
// roofline item: the 2.4 GHz constant and the per-wave median of round 2's microbenchmark are gone).
//
// Every wave issues `iters` x 64 independent instructions of one kind (eight accumulators: dependent-issue latency never
// gates); one workgroup of 256 x waves_per_simd threads per CU, LDS-sized so that a second one does not fit.  Reported:
// nanoseconds of SIMD time per wave64 instruction = launch time (HIP events) / (instructions per wave x waves per SIMD) -
// whatever the arbiter does with the resident waves (at 3 waves per SIMD it serves two and the third runs afterwards: the
// per-wave median that round 2 used understates the SIMD's time by a third) and whatever the clock does under load.
// The standalone tools/valu_issue_rate.hip prints the same figure next to the per-wave stamps for more instruction kinds.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ssn.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(1024) void k_issue_stream(float* sink, unsigned long long* ticks, int iters) {
  extern __shared__ unsigned char pad[];          // only there to keep a second workgroup off the CU
  float a[8];
  f32x2 p[8];
  const float c0 = 1.0000001f, c1 = 1e-9f;
  const f32x2 q0 = {1.0000001f, 0.9999999f}, q1 = {1e-9f, -1e-9f};
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = 1.0f + threadIdx.x * 1e-6f + i; p[i] = (f32x2){a[i], a[i] + 0.5f}; }
  __builtin_amdgcn_s_barrier();
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (KIND == SSN_PROBE_PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q0), "v"(q1));
        REP8(X) REP8(X)
#undef X
      } else if constexpr (KIND == SSN_PROBE_PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q0));
        REP8(X) REP8(X)
#undef X
      } else if constexpr (KIND == SSN_PROBE_PK_ADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q1));
        REP8(X) REP8(X)
#undef X
      } else if constexpr (KIND == SSN_PROBE_TRANS) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        REP8(X) REP8(X)
#undef X
      } else if constexpr (KIND == SSN_PROBE_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
        REP8(X) REP8(X)
#undef X
      } else if constexpr (KIND == SSN_PROBE_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        REP8(X) REP8(X)
#undef X
      } else if constexpr (KIND == SSN_PROBE_DPP) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        REP8(X) REP8(X)
#undef X
      } else if constexpr (KIND == SSN_PROBE_MOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c1));
        REP8(X) REP8(X)
#undef X
      } else if constexpr (KIND == SSN_PROBE_READLANE) {
#define X(i) asm volatile("v_readlane_b32 s20, %0, 5" :: "v"(a[i]) : "s20");
        REP8(X) REP8(X)
#undef X
      } else if constexpr (KIND == SSN_PROBE_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c1));
        REP8(X) REP8(X)
#undef X
      } else {
#define X(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        REP8(X) REP8(X)
#undef X
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 123.456f) sink[0] = s;                 // keeps the accumulators alive
  if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }
}

template <int KIND>
hipError_t run_kind(int threads, int iters, int n_cu, float* sink, unsigned long long* d_ticks, float* ms) {
  constexpr int LDS = 96 * 1024;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_issue_stream<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (e != hipSuccess) return e;
  hipEvent_t e0, e1;
  if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
  if ((e = hipEventCreate(&e1)) != hipSuccess) { (void)hipEventDestroy(e0); return e; }
  hipLaunchKernelGGL((k_issue_stream<KIND>), dim3(n_cu), dim3(threads), LDS, 0, sink, d_ticks, iters / 8 + 1);      // warm-up
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_issue_stream<KIND>), dim3(n_cu), dim3(threads), LDS, 0, sink, d_ticks, iters);
  (void)hipEventRecord(e1, 0);
  e = hipEventSynchronize(e1);
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipEventElapsedTime(ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return e;
}

}  // namespace

namespace ssn {
int probe_fail(int code, const char* what, hipError_t e);      // ssn_host.hip: sets ssn_last_error
}

extern "C" int ssn_probe_issue_rate(int32_t device, int32_t kind, int32_t waves_per_simd, int32_t iters,
                                    double* ns_per_wave_instruction, double* shader_mhz) {
  if (!ns_per_wave_instruction || waves_per_simd < 1 || waves_per_simd > 4 || iters < 1 || iters > (1 << 22) || kind < 0 || kind >= SSN_PROBE_N_KINDS)
    return ssn::probe_fail(SSN_EINVAL, "ssn_probe_issue_rate: kind 0..10, waves_per_simd 1..4, iters 1..2^22", hipSuccess);
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return ssn::probe_fail(SSN_EHIP, "hipSetDevice", e);
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return ssn::probe_fail(SSN_EHIP, "hipGetDeviceProperties", e);
  const int n_cu = prop.multiProcessorCount, threads = 256 * waves_per_simd;
  float* sink = nullptr;
  unsigned long long* d_ticks = nullptr;
  if ((e = hipMalloc(&sink, 64)) != hipSuccess) return ssn::probe_fail(SSN_ENOMEM, "hipMalloc", e);
  if ((e = hipMalloc(&d_ticks, 16)) != hipSuccess) { (void)hipFree(sink); return ssn::probe_fail(SSN_ENOMEM, "hipMalloc", e); }
  float ms = 0.0f;
  switch (kind) {
#define K(k) case k: e = run_kind<k>(threads, iters, n_cu, sink, d_ticks, &ms); break;
    K(SSN_PROBE_PK_FMA) K(SSN_PROBE_PK_MUL) K(SSN_PROBE_PK_ADD) K(SSN_PROBE_TRANS) K(SSN_PROBE_FMA) K(SSN_PROBE_ADD)
    K(SSN_PROBE_DPP) K(SSN_PROBE_MOV) K(SSN_PROBE_READLANE) K(SSN_PROBE_CNDMASK) K(SSN_PROBE_OTHER)
#undef K
    default: e = hipErrorInvalidValue;
  }
  unsigned long long t[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(t, d_ticks, 16, hipMemcpyDeviceToHost);
  (void)hipFree(sink);
  (void)hipFree(d_ticks);
  if (e != hipSuccess) return ssn::probe_fail(SSN_EHIP, "issue-rate probe", e);
  const double n_inst = (double)iters * 64.0;              // per wave
  *ns_per_wave_instruction = 1e6 * (double)ms / (n_inst * waves_per_simd);
  if (shader_mhz) *shader_mhz = t[1] ? 100.0 * (double)t[0] / (double)t[1] : 0.0;      // s_memtime ticks per 100 MHz s_memrealtime tick
  return SSN_OK;
}
