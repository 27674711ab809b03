// Replays a capture trace ("OP L q" launch on stream q (-1 = origin), "OP R e q" record event e on q,
// "OP W q e" stream q waits for event e) to reproduce capture problems outside the simulator.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void nop(float* p) { p[threadIdx.x] += 1.0f; }
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "r");
  int limit = argc > 2 ? atoi(argv[2]) : 1 << 30;
  float* buf; CK(hipMalloc(&buf, 16 * 1024 * 4));
  hipStream_t s0; CK(hipStreamCreate(&s0));
  std::vector<hipStream_t> side(8);
  for (auto& st : side) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  std::vector<hipEvent_t> evs(4096);
  for (auto& ev : evs) CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  char line[256]; int n = 0;
  while (fgets(line, sizeof line, f) && n < limit) {
    char op; int a = 0, b = 0;
    if (strncmp(line, "OP ", 3)) continue;
    sscanf(line + 3, "%c %d %d", &op, &a, &b);
    ++n;
    if (op == 'L') hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a < 0 ? s0 : side[a], buf + (a + 1) * 1024);
    else if (op == 'R') CK(hipEventRecord(evs[a], b < 0 ? s0 : side[b]));
    else if (op == 'W') CK(hipStreamWaitEvent(a < 0 ? s0 : side[a], evs[b], 0));
  }
  printf("replayed %d ops\n", n); fflush(stdout);
  hipGraph_t g; hipGraphExec_t ge;
  hipError_t ee = hipStreamEndCapture(s0, &g);
  printf("end capture: %s\n", hipGetErrorString(ee)); fflush(stdout);
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
  printf("ok\n");
  return 0;
}
