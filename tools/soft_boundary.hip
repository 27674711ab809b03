// What does a SOFTWARE kernel boundary cost on MI355X?  
//
// A SLAM timestep is ~3.4 dependent k_round launches with ~4 us between the last block of one and the first block of the next.
// Candidate: the blocks of round r + 1 ride at the end of round r's grid (one launch for two rounds) - every block of r ends with an
// agent-scope release (buffer_wbl2 sc1 + s_waitcnt) and an increment of the round's counter, every block of r + 1 polls the counter
// and then does an agent-scope acquire (buffer_inv sc1) before it reads anything r wrote.  Blocks are dispatched in index order, so
// all of r's blocks are resident or done before the first block of r + 1 starts: the poll cannot starve its producers.
//
// The microbenchmark runs pairs of rounds shaped like SLAM's streaming rounds - N blocks of 256 threads, each streaming `kb` KB of
// its own input and writing 8 KB of state plus 1 KB of partial sums; the second round's block c reads the partial sums of block
// (7c + 3) mod N of the first - (a) as two launches, (b) as one launch with the software boundary per block, (c) with ONE write-back
// per XCD (the first consumer block that arrives on an XCD once every producer has finished issues it for all of them) and the
// invalidate per block, (d) with write-back and agent-scope invalidate once per XCD and a workgroup-scope invalidate per block -
// 64 pairs per hipGraph, every sum checked (a stale read shows as a wrong sum).
//
//   hipcc -O3 --offload-arch=gfx950 tools/soft_boundary.hip -o gpurun_out/soft_boundary && gpurun_out/soft_boundary [N] [kb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  const float4* in_a; const float4* in_b;     // [N][vecs] each
  float* state;                               // [N][2048]
  float* mid;                                 // [N][256]
  float* out;                                 // [N][256]
  unsigned* counter; unsigned* err;
  unsigned* elect;                            // [pair][8] per-XCD election tickets, [pair][8 + 0] .. ; flushed[pair] behind them (modes 3, 4)
  int pair;
  int N, vecs;                                // vecs: float4 per block
  int mode;                                   // 0: producers only, 1: consumers only, 2: both in one grid, release + acquire per block;
                                              // 3: release ONCE per XCD (the first consumer block there writes the XCD's L2 back once every producer has
                                              //    finished), acquire per block; 4: as 3 with the agent-scope invalidate once per XCD and sc0 per block
  unsigned expect;                            // counter value when every producer of this pair has finished
  float salt;
};

__device__ __forceinline__ float stream_sum(const float4* p, int vecs) {
  float s = 0.f;
  for (int v = threadIdx.x; v < vecs; v += 256) { const float4 q = p[v]; s += (q.x + q.y) + (q.z + q.w); }
  return s;
}

__global__ __launch_bounds__(256) void k_pair(Args a) {
  const int bx = blockIdx.x, tid = threadIdx.x;
  const bool producer = a.mode == 0 || (a.mode >= 2 && bx < a.N);
  if (producer) {
    const float s = stream_sum(a.in_a + (size_t)bx * a.vecs, a.vecs);
    for (int i = tid; i < 2048; i += 256) a.state[(size_t)bx * 2048 + i] = s + a.salt;       // 8 KB of dirty state
    a.mid[(size_t)bx * 256 + tid] = s + a.salt;
    if (a.mode == 2) {
      __threadfence();                          // agent-scope release: the block's stores leave the XCD's L2
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (a.mode >= 3) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the block's stores have reached the XCD's L2
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    const int c = a.mode >= 2 ? bx - a.N : bx;
    if (c < 0 || c >= a.N) return;               // (cannot happen: the grid is N or 2 N blocks)
    const float own = stream_sum(a.in_b + (size_t)c * a.vecs, a.vecs);     // the consumer's own stream does not wait
    if (a.mode == 2) {
      if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.expect) {
          if (++spins > (1u << 16) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {    // (a producer that never comes: no hang)
            __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __syncthreads();
      __threadfence();                          // agent-scope acquire: nothing stale from this XCD's caches
    } else if (a.mode >= 3) {
      if (tid == 0) {
        unsigned spins = 0;
        auto stuck = [&]() {
          if (++spins > (1u << 16) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true;
          }
          __builtin_amdgcn_s_sleep(2);
          return false;
        };
        while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.expect) if (stuck()) break;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        unsigned* const el = a.elect + (size_t)a.pair * 16;
        if (__hip_atomic_fetch_add(el + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          // first consumer block on this XCD: every producer has finished, so ONE write-back covers all their stores here
          asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
          if (a.mode == 4) asm volatile("buffer_inv sc1" ::: "memory");
          __hip_atomic_fetch_add(el + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(el + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u) if (stuck()) break;
      }
      __syncthreads();
      if (a.mode == 3) asm volatile("buffer_inv sc1" ::: "memory");
      else asm volatile("buffer_inv sc0" ::: "memory");
    }
    const int src = (int)(((long long)c * 7 + 3) % a.N);
    a.out[(size_t)c * 256 + tid] = a.mid[(size_t)src * 256 + tid] + own;
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2048;
  const int kb = argc > 2 ? atoi(argv[2]) : 48;
  const int vecs = kb * 1024 / 16;
  const int PAIRS = 64, REPS = 10;
  Args a{};
  a.N = N; a.vecs = vecs;
  float4 *in_a, *in_b;
  CHECK(hipMalloc(&in_a, (size_t)N * vecs * 16)); CHECK(hipMalloc(&in_b, (size_t)N * vecs * 16));
  {
    std::vector<float> h((size_t)N * vecs * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 20) & 7) * 0.125f;
    CHECK(hipMemcpy(in_a, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 40503u >> 9) & 3) * 0.25f;
    CHECK(hipMemcpy(in_b, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  a.in_a = in_a; a.in_b = in_b;
  CHECK(hipMalloc(&a.state, (size_t)N * 2048 * 4)); CHECK(hipMalloc(&a.mid, (size_t)N * 256 * 4)); CHECK(hipMalloc(&a.out, (size_t)N * 256 * 4));
  CHECK(hipMalloc(&a.counter, 4)); CHECK(hipMalloc(&a.err, 4));
  CHECK(hipMalloc(&a.elect, (size_t)64 * 16 * 4));
  CHECK(hipMemset(a.counter, 0, 4)); CHECK(hipMemset(a.err, 0, 4));
  hipStream_t st; CHECK(hipStreamCreate(&st));

  // reference sums of the last pair (salt PAIRS - 1) from a plain two-launch run
  std::vector<float> want((size_t)N * 256), got((size_t)N * 256);
  auto run_pairs = [&](int mode2, unsigned base) {           // enqueue PAIRS pairs on st
    for (int p = 0; p < PAIRS; ++p) {
      Args b = a; b.salt = (float)p;
      b.pair = p;
      if (mode2) { b.mode = mode2 == 1 ? 2 : mode2 + 1; b.expect = base + (unsigned)(p + 1) * (unsigned)N; hipLaunchKernelGGL(k_pair, dim3(2 * N), dim3(256), 0, st, b); }
      else { b.mode = 0; hipLaunchKernelGGL(k_pair, dim3(N), dim3(256), 0, st, b); b.mode = 1; hipLaunchKernelGGL(k_pair, dim3(N), dim3(256), 0, st, b); }
    }
  };
  run_pairs(0, 0);
  CHECK(hipStreamSynchronize(st));
  CHECK(hipMemcpy(want.data(), a.out, want.size() * 4, hipMemcpyDeviceToHost));

  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int mode2 = 0; mode2 < 4; ++mode2) {
    // one graph of PAIRS pairs; the ride's counter keeps counting over replays (expect is baked in per replay: re-capture per rep)
    float best = 1e30f, sum = 0.f;
    unsigned base = 0;
    CHECK(hipMemset(a.counter, 0, 4));
    for (int rep = 0; rep < REPS; ++rep) {
      hipGraph_t g; hipGraphExec_t ge;
      CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      run_pairs(mode2, base);
      CHECK(hipStreamEndCapture(st, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CHECK(hipMemsetAsync(a.out, 0, (size_t)N * 256 * 4, st));
      CHECK(hipMemsetAsync(a.elect, 0, (size_t)64 * 16 * 4, st));
      CHECK(hipEventRecord(e0, st));
      CHECK(hipGraphLaunch(ge, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipStreamSynchronize(st));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0) { best = ms < best ? ms : best; sum += ms; }
      base += (unsigned)PAIRS * (unsigned)N;
      CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    }
    unsigned err = 0; CHECK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(got.data(), a.out, got.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
    const double mb = 2.0 * N * vecs * 16 / 1e6;
    printf("%s: N = %d blocks per round, %d KB per block (%.0f MB per pair): best %.2f us per pair, mean %.2f (%.2f TB/s); wrong sums %zu of %zu, time-outs %u\n",
           mode2 == 0 ? "two launches per pair                          " : mode2 == 1 ? "one launch: release + acquire per block        " :
           mode2 == 2 ? "one launch: release per XCD, acquire per block " : "one launch: release + acquire per XCD, sc0/block", N, kb, mb,
           1e3 * best / PAIRS, 1e3 * sum / (REPS - 1) / PAIRS, mb / (1e3 * best / PAIRS), bad, got.size(), err);
  }
  return 0;
}
