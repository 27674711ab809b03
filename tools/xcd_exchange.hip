// xcd_exchange.hip - what a per-timestep exchange of four partial sums among the P workgroups of a SPLIT VCO costs on
// MI355X when the members sit on CUs of ONE XCD (shared L2) - the question VERDICT r2 item 6 asks before the 1.9x
// strong-scaling floor of DESIGN 5.1 is accepted.  Round 2 only measured a 256-workgroup barrier through atomics on the
// memory side of eight non-coherent L2s (tools/grid_barrier.hip: 2.3 us); a split VCO needs far less: every member
// publishes four floats and reads the four floats of its P - 1 partners.
//
// Protocol (no atomics, no fences, no cache maintenance):
//   slot[buf][member][4] words, three buffers used round-robin by timestep; a word holds the SENTINEL between uses.
//   step t:  (1) the member resets ITS words of buffer (t + 1) % 3 to the sentinel (they were read for the last time at
//                step t - 2: a partner that has published step t - 1 has consumed everybody's step t - 2) and waits for
//                that store's acknowledgement (s_waitcnt vmcnt(0)) - the wait hides under the step's neuron arithmetic;
//            (2) ... neuron arithmetic (emulated: `work` packed FMAs per lane) ...
//            (3) lanes 0, 16, 32, 48 of wave 0 store the member's four partial sums to buffer t % 3 (plain stores: the
//                vector L1 is write-through, the data reaches the XCD's L2);
//            (4) every wave polls: lane l loads word (l >> 4) of member (l & 15) with an L1-bypassing load (sc0 - or sc1 /
//                sc0 sc1 for members on other XCDs) until no lane sees the sentinel; a 4-step DPP row sum then leaves the
//                total of value r in row r of every wave - the lane-distributed form k_ens_block already uses.
//   Each word is checked on its own, so nothing depends on 16-byte atomicity.
// Placement: blockIdx % 8 is the XCD (checked against HW_REG_XCC_ID); "same XCD" groups are blockIdx = x + 8 * (P * q + m),
// "spread" groups are P consecutive blockIdx values (P different XCDs).
// Every round's totals are verified (member m publishes (t * 8 + m) * (r + 1)).
//
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_exchange.hip -o tools/build/xcd_exchange && tools/build/xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned SENT = 0x7fc0dead;          // a NaN payload no sum produces

template <int BITS> __device__ inline unsigned ld_bypass(const unsigned* p) {
  unsigned v;
  if (BITS == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (BITS == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (BITS == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (BITS == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  // 4: invalidate this CU's vector L1, then a plain load: misses the L1, may hit the XCD's L2 (the time loop of k_ens_block
  //    holds nothing in the L1 that it would miss afterwards)
  if (BITS == 4) asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  // 5: a returning atomic OR with 0 at workgroup scope (no sc1): atomics execute in the XCD's L2, never in the L1
  if (BITS == 5) { const unsigned z = 0; asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory"); }
  return v;
}
template <int BITS> __device__ inline void st_word(unsigned* p, unsigned v) {
  if (BITS == 0 || BITS == 1 || BITS == 4 || BITS == 5) asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (BITS == 2) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (BITS == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

__device__ inline float row_sum_dpp(float v) {
  int x;
#define A(ctrl) x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false); v += __builtin_bit_cast(float, x);
  A(0xB1) A(0x4E) A(0x141) A(0x140)
#undef A
  return v;
}

struct Res { unsigned bad, xcc_mismatch, timeout, pad; unsigned long long cycles, real; };

// P members per group; SAME: members on one XCD; BITS: cache bits of the polling loads / publishing stores
template <int P, int SAME, int BITS, int W0>
__global__ __launch_bounds__(512) void k_exchange(unsigned* slots, int steps, int work, Res* res, float* sink) {
  __shared__ float red[64];
  __shared__ unsigned bc[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int group, member;
  if (SAME) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; group = (j / P) * 8 + x; member = j % P; }
  else { group = blockIdx.x / P; member = blockIdx.x % P; }
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((xcc & 7u) != (blockIdx.x & 7u)) atomicAdd(&res->xcc_mismatch, 1u);
  }
  unsigned* const gs = slots + (size_t)group * 3 * 16 * 4;      // [3][16 members][4 words]
  f32x2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x2){1.0f + tid * 1e-6f + i, 0.5f + i};
  const f32x2 q0 = {1.0000001f, 0.9999999f}, q1 = {1e-9f, -1e-9f};
  unsigned bad = 0, timeout = 0;
  unsigned long long t0, r0, t1, r1;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  for (int t = 0; t < steps; ++t) {
    const int buf = t % 3, nxt = (t + 1) % 3;
    // (1) own words of the next buffer back to the sentinel, acknowledged before anything of this step is published
    if (P > 1 && wave == 0 && (lane & 15) == 0) st_word<BITS>(gs + (nxt * 16 + member) * 4 + (lane >> 4), SENT);
    // (2) the step's arithmetic
    for (int w = 0; w < work; ++w) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(q0), "v"(q1));
    }
    // workgroup sums as k_ens_block forms them: wave sums -> LDS -> barrier -> row sums (values: (t * 8 + member) * (r + 1) / 8 per wave)
    if ((lane & 15) == 0) red[(lane >> 4) * 16 + wave] = (float)((t % 1000) * 8 + member) * (float)((lane >> 4) + 1) * 0.125f;
    if (P > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float v = row_sum_dpp((lane & 15) < 8 ? red[(lane >> 4) * 16 + (lane & 15)] : 0.0f);      // row r: this member's value r
    if (P > 1) {
      // (3) publish
      if (wave == 0 && (lane & 15) == 0) st_word<BITS>(gs + (buf * 16 + member) * 4 + (lane >> 4), __builtin_bit_cast(unsigned, v));
      // (4) poll (W0: only wave 0 polls and hands the partners' values on through LDS - one more workgroup barrier, an
      //     eighth of the polling traffic)
      const unsigned* src = gs + (buf * 16 + (lane & 15)) * 4 + (lane >> 4);
      unsigned got = 0, spins = 0;
      const bool mine = (lane & 15) < P;
      if (!W0 || wave == 0) {
        while (true) {
          if (mine) got = ld_bypass<BITS>(src);
          const bool wait = mine && got == SENT;
          if (!__any(wait)) break;
          if (++spins > (1u << 18)) { timeout = 1; break; }
        }
      }
      if (W0) {
        if (wave == 0) bc[lane] = got;
        __syncthreads();
        got = bc[lane];
      }
      v = row_sum_dpp(mine ? __builtin_bit_cast(float, got) : 0.0f);
    }
    // verify: total of value r = sum_m (t * 8 + m) * (r + 1)
    float want = 0.0f;
    for (int m = 0; m < P; ++m) want += (float)((t % 1000) * 8 + m) * (float)((lane >> 4) + 1);
    if (v != want) bad += 1;
    if (timeout) break;       // (the partners then time out as well: the launch ends)
    __syncthreads();      // (k_ens_block has this barrier too: red[] is double-buffered there; here it protects red[])
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  if (s == 123.456f) sink[0] = s;
  if (bad) atomicAdd(&res->bad, bad);
  if (timeout) atomicAdd(&res->timeout, 1u);
  if (blockIdx.x == 0 && tid == 0) { res->cycles = t1 - t0; res->real = r1 - r0; }
}

template <int P, int SAME, int BITS, int W0 = 0>
static void run(int steps, int work, int n_wg) {
  unsigned* slots; Res* res; float* sink;
  const int groups = n_wg;       // (upper bound)
  (void)hipMalloc(&slots, (size_t)groups * 3 * 16 * 4 * 4);
  std::vector<unsigned> init((size_t)groups * 3 * 16 * 4, SENT);
  (void)hipMemcpy(slots, init.data(), init.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&res, sizeof(Res)); (void)hipMemset(res, 0, sizeof(Res));
  (void)hipMalloc(&sink, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_exchange<P, SAME, BITS, W0>), dim3(n_wg), dim3(512), 0, 0, slots, steps, work, res, sink);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  Res h; (void)hipMemcpy(&h, res, sizeof h, hipMemcpyDeviceToHost);
  static const char* bits[] = {"plain", "sc0", "sc1", "sc0 sc1", "inv+ld", "atomic"};
  printf("P %d  %-7s  %s  loads/stores %-8s  work %4d  %4d workgroups  %.3f us per timestep (events)  %.0f shader cycles per timestep, clock %.0f MHz  "
         "wrong sums %u  timeouts %u  blockIdx%%8 != XCC_ID %u\n",
         P, SAME ? "one XCD" : "spread", W0 ? "wave 0 polls" : "all waves poll", bits[BITS], work, n_wg, 1e3 * ms / steps, (double)h.cycles / steps,
         h.real ? (double)h.cycles / (double)h.real * 100.0 : 0.0, h.bad, h.timeout, h.xcc_mismatch);
  fflush(stdout);
  (void)hipFree(slots); (void)hipFree(res); (void)hipFree(sink);
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 2000;
  // work = trips of 8 packed FMAs per lane: 0 = the bare exchange; 64 ~ a quarter VCO's timestep (512 instructions per wave)
  for (int work : {0, 64}) {
    run<1, 1, 1>(steps, work, 256);
    run<2, 1, 4>(steps, work, 256);
    run<4, 1, 4>(steps, work, 256);
    run<8, 1, 4>(steps, work, 256);
    run<2, 1, 5>(steps, work, 256);
    run<4, 1, 5>(steps, work, 256);
    run<8, 1, 5>(steps, work, 256);
    run<4, 1, 4, 1>(steps, work, 256);
    run<4, 1, 5, 1>(steps, work, 256);
    run<2, 1, 3>(steps, work, 256);
    run<4, 1, 3>(steps, work, 256);
    run<4, 1, 3, 1>(steps, work, 256);
    run<2, 0, 2>(steps, work, 256);
    run<4, 0, 2>(steps, work, 256);
    run<4, 0, 4>(steps, work, 256);     // L1 invalidate + plain load across XCDs: the other XCD's L2 is not coherent - expected stale (detector check)
    run<4, 0, 5>(steps, work, 256);     // workgroup-scope atomics across XCDs: likewise
  }
  // fewer workgroups than CUs (an 8-GPU shard holds 64 VCOs x 4 members = 256; a 4-GPU shard 127 x 2 = 254)
  run<4, 1, 4>(steps, 64, 64);
  run<4, 1, 5>(steps, 64, 64);
  run<2, 1, 4>(steps, 64, 128);
  run<1, 1, 1>(steps, 256, 256);
  run<2, 1, 4>(steps, 128, 256);
  run<2, 1, 5>(steps, 128, 256);
  return 0;
}
