// grid_barrier.hip - what a grid barrier costs on MI355X (8 XCDs, one L2 each, not coherent with each other), and which
// fences it needs.  A persistent grid of G workgroups runs R rounds: every workgroup writes its round number into its own
// 256-byte slot, passes the barrier, then reads the slot of a workgroup on ANOTHER XCD and counts stale values.
//
//   variant 0  thread 0 of every workgroup: __threadfence() - atomic - poll - __threadfence()      (what the removed persistent variant of k_round did)
//   variant 1  every workgroup: s_waitcnt vmcnt(0); only the LAST arrival of an XCD group writes the group's L2 back
//              (buffer_wbl2 sc1) and, once all groups are done, invalidates it (buffer_inv sc1) and releases its group;
//              the others then invalidate only their L1 (buffer_inv sc0)
//   variant 2  like 1, but every workgroup issues buffer_inv sc1 after the release
//   variant 3  no cache maintenance at all (shows that the stale-read detector works)
// Groups are blockIdx.x % 8; the kernel also reports how many workgroups found blockIdx.x % 8 != XCC_ID.
//
//   hipcc -O3 --offload-arch=gfx950 tools/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Ctl {
  unsigned int global_done;  unsigned int pad0[31];
  unsigned int arrived[8][32];
  unsigned int release[8][32];
  unsigned int stale, xcc_mismatch, timeout, pad1[29];
};

__device__ inline unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int V>
__device__ inline void barrier(Ctl* c, unsigned gen) {
  if (V != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = blockIdx.x & 7u, members = (gridDim.x + 7u - g) / 8u;
    unsigned spins = 0;
    if (V == 0) {
      __threadfence();
      const unsigned v = atomicAdd(&c->arrived[g][0], 1u);
      if (v + 1u == gen * members) atomicAdd(&c->global_done, 1u);
      while (ld(&c->global_done) < gen * 8u) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) { c->timeout = 1; break; } }
      __threadfence();
    } else {
      const unsigned v = atomicAdd(&c->arrived[g][0], 1u);
      if (v + 1u == gen * members) {                     // last arrival of this group: its L2 holds every member's stores
        if (V != 3) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        atomicAdd(&c->global_done, 1u);
        while (ld(&c->global_done) < gen * 8u) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) { c->timeout = 1; break; } }
        if (V != 3) asm volatile("buffer_inv sc1" ::: "memory");
        __hip_atomic_store(&c->release[g][0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (ld(&c->release[g][0]) < gen) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) { c->timeout = 1; break; } }
      }
      if (V == 1) asm volatile("buffer_inv sc0" ::: "memory");
      if (V == 2) asm volatile("buffer_inv sc1" ::: "memory");
    }
  }
  __syncthreads();
}

template <int V>
__global__ __launch_bounds__(256) void k_persist(Ctl* c, unsigned* slots, int rounds, unsigned long long* cycles) {
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((xcc & 7u) != (blockIdx.x & 7u)) atomicAdd(&c->xcc_mismatch, 1u);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned stale = 0;
  for (int r = 1; r <= rounds; ++r) {
    slots[(size_t)blockIdx.x * 64 + (threadIdx.x & 63)] = (unsigned)r;
    barrier<V>(c, (unsigned)(2 * r - 1));
    const unsigned other = (blockIdx.x + 37u) % gridDim.x;         // 37 % 8 = 5: another group
    const unsigned got = slots[(size_t)other * 64 + (threadIdx.x & 63)];
    if (got != (unsigned)r) stale += 1;
    // (the writer of `other` overwrites its slot only after the NEXT barrier's arrival of this workgroup... so a second barrier)
    barrier<V>(c, (unsigned)(2 * r));
  }
  if (stale) atomicAdd(&c->stale, stale);
  if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = __builtin_readcyclecounter() - t0;
}

template <int V>
void run(int per_cu, int rounds) {
  Ctl* c; unsigned* slots; unsigned long long* cyc;
  const int G = 256 * per_cu;
  (void)hipMalloc(&c, sizeof(Ctl)); (void)hipMemset(c, 0, sizeof(Ctl));
  (void)hipMalloc(&slots, (size_t)G * 256); (void)hipMemset(slots, 0, (size_t)G * 256);
  (void)hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_persist<V>), dim3(G), dim3(256), 0, 0, c, slots, rounds, cyc);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  Ctl h; (void)hipMemcpy(&h, c, sizeof h, hipMemcpyDeviceToHost);
  printf("variant %d  %4d workgroups  %.2f us per barrier  stale reads %u  timeouts %u  blockIdx%%8 != XCC_ID in %u workgroups\n",
         V, G, 1e3 * ms / (2.0 * rounds), h.stale, h.timeout, h.xcc_mismatch);
  (void)hipFree(c); (void)hipFree(slots); (void)hipFree(cyc);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  for (int per_cu : {1, 4, 6}) {
    run<0>(per_cu, rounds);
    run<1>(per_cu, rounds);
    run<2>(per_cu, rounds);
    run<3>(per_cu, rounds);
  }
  return 0;
}
