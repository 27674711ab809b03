// Does a hipGraph captured from forked streams run its independent branches concurrently on MI355X / ROCm 7.2?
// Two (or four) chains of small kernels: captured on one stream vs forked over several streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(float* p, int iters) {      // one workgroup, ~iters dependent FMAs
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}

int main() {
  const int chains = 4, len = 12, reps = 200;
  float* buf; CK(hipMalloc(&buf, chains * 1024 * sizeof(float))); CK(hipMemset(buf, 0, chains * 1024 * sizeof(float)));
  hipStream_t s0; CK(hipStreamCreate(&s0));
  std::vector<hipStream_t> ss(chains);
  for (auto& s : ss) CK(hipStreamCreate(&s));
  hipEvent_t fork, t0, t1; CK(hipEventCreate(&fork)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  std::vector<hipEvent_t> joins(chains);
  for (auto& jv : joins) CK(hipEventCreate(&jv));
  for (int iters : {2000, 20000}) {
    for (int mode = 0; mode < 2; ++mode) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
      if (mode == 0) {
        for (int c = 0; c < chains; ++c)
          for (int i = 0; i < len; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, s0, buf + c * 1024, iters);
      } else {
        CK(hipEventRecord(fork, s0));
        for (int c = 0; c < chains; ++c) {
          CK(hipStreamWaitEvent(ss[c], fork, 0));
          for (int i = 0; i < len; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, ss[c], buf + c * 1024, iters);
          CK(hipEventRecord(joins[c], ss[c]));
          CK(hipStreamWaitEvent(s0, joins[c], 0));
        }
      }
      CK(hipStreamEndCapture(s0, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
      CK(hipEventRecord(t0, s0));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s0));
      CK(hipEventRecord(t1, s0)); CK(hipStreamSynchronize(s0));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      printf("iters %6d %-22s: %.1f us per graph (%d kernels) -> %.2f us per kernel slot\n", iters,
             mode ? "forked over 4 streams" : "one stream", ms / reps * 1e3, chains * len, ms / reps * 1e3 / (chains * len));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  // mode 2: the simulator's pattern - 16 steps, each: [main kernel] fork -> items on side streams with cross-stream
  // event waits -> join -> next step, side streams re-used across steps, non-blocking streams, no-timing events
  {
    const int steps = 16, items = 12;
    std::vector<hipStream_t> side(8);
    for (auto& st : side) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<hipEvent_t> evs(steps * (items + 1));
    for (auto& ev : evs) CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    size_t ne = 0;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    for (int st = 0; st < steps; ++st) {
      hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, s0, buf, 2000);
      hipEvent_t fk = evs[ne++];
      CK(hipEventRecord(fk, s0));
      std::vector<hipEvent_t> done(items);
      for (int i = 0; i < items; ++i) {
        hipStream_t q = side[i % 4];
        if (i < 4) CK(hipStreamWaitEvent(q, fk, 0));
        if (i >= 4 && (i % 4) != 0) CK(hipStreamWaitEvent(q, done[i - 1 - (i % 4 == 1 ? 0 : 0)], 0));   // wait for a neighbour stream's item
        hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, q, buf + (i % 4) * 1024, 2000);
        done[i] = evs[ne++];
        CK(hipEventRecord(done[i], q));
      }
      for (int c = 0; c < 4; ++c) CK(hipStreamWaitEvent(s0, done[items - 4 + c], 0));
    }
    hipError_t ee = hipStreamEndCapture(s0, &g);
    printf("mode 2 end capture: %s\n", hipGetErrorString(ee));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
    CK(hipEventRecord(t0, s0));
    for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge, s0));
    CK(hipEventRecord(t1, s0)); CK(hipStreamSynchronize(s0));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    printf("mode 2 (16 steps x (1 + 12 items on 4 re-used side streams)): %.1f us per graph\n", ms / 50 * 1e3);
  }
  return 0;
}
