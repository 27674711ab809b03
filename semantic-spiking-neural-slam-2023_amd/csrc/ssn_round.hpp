// ssn_round.hpp - k_round: every mutually independent operator of a timestep in ONE launch.
//
// A SLAM timestep is a chain of ~25 dependency levels (reference networks/slam.py:259-307 wires ensembles, products
// and memories through un-filtered connections, so most of a step is sequential), and each level holds a few big
// operators next to a dozen 1 000 - 2 000 element vector operators.  A launch costs ~5 us on this GPU whatever it
// does, so one launch per operator (or per operator kind) makes the step launch-bound.  k_round runs a whole level:
// the host planner (ssn_host.hip, plan_rounds) assigns every operator of the step the earliest round its data
// hazards allow; a round's operators become the entries of ONE grid - entry i owns virtual blocks
// [first, first + gx * gy) - and each 256-thread workgroup runs the body its block belongs to:
//
//   RK_GLUE     one chunk (GLUE_CHUNK elements / GLUE_ROWS reduction rows) of one micro-operator: fill, axpy, lowpass,
//               table row, block-row hand-off, probe sample, partial-sum reductions, step counter.  The single-workgroup
//               interpreter k_program ran these back to back (~0.4 us each, latency-bound); here they run side by side.
//   RK_GATE, RK_ARGMAX   whole-vector micro-operators (a dot product / an argmax decide what is written): one block
//   RK_MATVEC_*, RK_SPMV, RK_NEURONS, RK_DFT, RK_PES, RK_VOJA   the bodies of the stand-alone kernels (ssn_kernels.hpp)
//   RK_ENS_*    k_ensarray's body for the variants a SLAM network uses: the path integrator's oscillators (3 inputs, 4 or 5
//               decoded rows, spike-sparse decoders) and the product ensembles of the circular convolutions (1-D)
//
// Workgroup memory is one dynamic allocation sized for the round's hungriest body (no static LDS in any body: static
// allocations of all bodies would add up).
#pragma once
#include "ssn_kernels.hpp"

namespace ssn {

// One chunk of one micro-operator.  Element-wise kinds: elements [chunk * GLUE_CHUNK, ...) - four per thread,
// loads before stores; row kinds (reductions, ensemble finish, small matvec): rows [chunk * GLUE_ROWS, ...), one per thread.
template <typename T, bool ROWS = false>
__device__ __forceinline__ void glue_body(const MicroOp<T>& op, const int chunk, const int sub, T* sig, StepCtx* __restrict__ ctx) {
  // sub: timestep offset of this instance inside a pipelined launch sequence (the clock advances once per sequence)
  // ROWS: member of a chain - element-wise kinds handle element chunk * GLUE_ROWS + tid only, like the row kinds
  const int tid = threadIdx.x;
  constexpr int U = ROWS ? 1 : 4;
  auto ew = [&](auto load, auto store) {
    const long long base = (long long)chunk * (ROWS ? GLUE_ROWS : GLUE_CHUNK);
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long i = base + u * 256 + tid; if (i < op.len) v[u] = load(i); }
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long i = base + u * 256 + tid; if (i < op.len) store(i, v[u]); }
  };
  switch (op.kind) {
    case M_FILL:
      { T* const d = sig + op.dst; ew([&](long long) { return op.a; }, [&](long long i, T v) { d[i] = v; }); }
      break;
    case M_AXPY_INC:
      { T* const d = sig + op.dst; const T* const x = sig + op.src;
        ew([&](long long i) { return d[i] + op.a * x[i]; }, [&](long long i, T v) { d[i] = v; }); }
      break;
    case M_AXPY_SET:
      { T* const d = sig + op.dst; const T* const x = sig + op.src;
        ew([&](long long i) { return op.a * x[i]; }, [&](long long i, T v) { d[i] = v; }); }
      break;
    case M_LOWPASS:
      { T* const d = sig + op.dst; const T* const x = sig + op.src;
        ew([&](long long i) { return op.a * d[i] + op.b * x[i]; }, [&](long long i, T v) { d[i] = v; }); }
      break;
    case M_LINCOMB: {   // dst = a * dst + b * (c + sum_k alpha_k * sig[src_k + i]); p0 = LinTerm[i0], i1 bits of c
      const LinTerm<T>* const t = (const LinTerm<T>*)op.p0;
      const int nt = (int)op.i0;
      T* const d = sig + op.dst;
      const long long base = (long long)chunk * (ROWS ? GLUE_ROWS : GLUE_CHUNK);
      T acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = op.c;
      for (int k = 0; k < nt; ++k) {
        const T alpha = t[k].alpha;
        const T* const x = sig + t[k].src;
#pragma unroll
        for (int u = 0; u < U; ++u) { const long long i = base + u * 256 + tid; if (i < op.len) acc[u] += alpha * x[i]; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = base + u * 256 + tid;
        if (i < op.len) d[i] = (op.a != T(0) ? op.a * d[i] : T(0)) + op.b * acc[u];
      }
      break;
    }
    case M_TABLE: {
      const TableSlot* t = (const TableSlot*)op.p0;
      const long long rel = ctx->step + sub - t->first_step;
      int row = -1;
      if (rel >= 0 && rel < t->n_idx) row = t->idx[rel];
      const bool have = row >= 0 && row < t->n_rows;
      const T* const trow = (const T*)t->rows + (have ? (size_t)row * t->width : 0);
      T* const d = sig + op.dst;
      ew([&](long long i) { return have ? trow[i] : T(0); }, [&](long long i, T v) { d[i] = v; });
      break;
    }
    case M_ROW_IN: {
      const T* const x = (const T*)op.p0 + (size_t)(ctx->step + sub - ctx->block_start + 1) * op.i0 + op.i1;
      T* const d = sig + op.dst;
      ew([&](long long i) { return x[i]; }, [&](long long i, T v) { d[i] = v; });
      break;
    }
    case M_ROW_OUT: {
      T* const d = (T*)op.p0 + (size_t)(ctx->step + sub - ctx->block_start + 1) * op.i0 + op.i1;
      const T* const x = sig + op.src;
      ew([&](long long i) { return x[i]; }, [&](long long i, T v) { d[i] = v; });
      break;
    }
    case M_PROBE: {
      const ProbeSlot* ps = (const ProbeSlot*)op.p0;
      const long long s1 = ctx->step + sub + 1;
      if (s1 % ps->every == 0) {
        const long long slot = s1 / ps->every - 1 - ps->base_slot;
        if (slot >= 0 && slot < ps->capacity) {
          T* const out = (T*)ps->data + (size_t)slot * op.len;
          const T* const x = sig + op.src;
          ew([&](long long i) { return x[i]; }, [&](long long i, T v) { out[i] = v; });
        } else if (tid == 0 && chunk == 0) {
          ctx->probe_overflow = 1;
        }
      }
      break;
    }
    case M_MATVEC_INC:
    case M_MATVEC_SET: {
      const T* Wm = (const T*)op.p0;
      const long long r = (long long)chunk * GLUE_ROWS + tid;
      if (r < op.len) {
        T s = T(0);
        for (int c = 0; c < (int)op.i0; ++c) s += Wm[(size_t)r * op.i1 + c] * sig[op.src + c];
        if (op.kind == M_MATVEC_INC) sig[op.dst + r] += s; else sig[op.dst + r] = s;
      }
      break;
    }
    case M_ENS_FINISH: {
      const T* part = (const T*)op.p0;
      if (op.src > 0) part += (size_t)((ctx->step + sub) & 1) * (size_t)op.src;      // (the array keeps two sets of partial sums, by timestep parity)
      const int* didx = (const int*)op.p1;
      const int P = (int)op.i0, dout = (int)op.i1;
      const long long i = (long long)chunk * GLUE_ROWS + tid;
      if (i < op.len) {
        const long long k = i / dout, r = i - k * dout;
        const int dst = didx[i];                 // (fetched beside the partial sums, not behind them)
        T s = T(0);
        int p = 0;
        for (; p + 8 <= P; p += 8) {             // eight partial sums in flight, added in workgroup order
          T v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = part[((size_t)k * P + p + q) * dout + r];
#pragma unroll
          for (int q = 0; q < 8; ++q) s += v[q];
        }
        {
          T v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = p + q < P ? part[((size_t)k * P + p + q) * dout + r] : T(0);
#pragma unroll
          for (int q = 0; q < 8; ++q) if (p + q < P) s += v[q];
        }
        sig[dst] = s;
      }
      break;
    }
    case M_REDUCE_SET:
    case M_REDUCE_INC: {
      const T* part = (const T*)op.p0;
      const int nc = (int)op.i0;
      const long long r = (long long)chunk * GLUE_ROWS + tid;
      if (r < op.len) {
        T s = T(0);
        int c = 0;
        for (; c + 16 <= nc; c += 16) {          // sixteen chunk reads in flight (a reduction is one of the dependent trips of every
          T v[16];                               // population hop: 40 chunks are three trips this way, five with eight), added in chunk order
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] = part[(size_t)(c + q) * op.i1 + r];
#pragma unroll
          for (int q = 0; q < 16; ++q) s += v[q];
        }
        for (; c + 8 <= nc; c += 8) {
          T v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = part[(size_t)(c + q) * op.i1 + r];
#pragma unroll
          for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; c < nc; ++c) s += part[(size_t)c * op.i1 + r];
        if (op.kind == M_REDUCE_INC) sig[op.dst + r] += s; else sig[op.dst + r] = s;
      }
      break;
    }
    case M_STEP_END:
      if (tid == 0 && chunk == 0) ctx->step = ctx->step + 1;
      break;
    default:
      break;
  }
}

// GATE (reference slam.py:236-257's loop-closure correction): dst = open ? rate * (est - cur) : 0, open when the two
// d-vectors agree (dot product above the threshold) and the flag input is 0.
template <typename T>
__device__ __forceinline__ void gate_body(const MicroOp<T>& op, T* __restrict__ sig, unsigned char* smem) {
  T* sred = reinterpret_cast<T*>(smem);
  const int tid = threadIdx.x;
  T part = T(0);
  for (long long i = tid; i < op.len; i += 256) part += sig[op.src + i] * sig[op.src + op.len + i];
  part = wave_sum(part);
  if ((tid & 63) == 0) sred[tid >> 6] = part;
  __syncthreads();
  const T dot = ((sred[0] + sred[1]) + sred[2]) + sred[3];
  const T flag = sig[op.src + 2 * op.len];
  const bool open = (flag <= T(1e-3) && flag >= T(-1e-3)) && dot > op.a;
  for (long long i = tid; i < op.len; i += 256)
    sig[op.dst + i] = open ? op.b * (sig[op.src + i] - sig[op.src + op.len + i]) : T(0);
}

// ARGMAX + row gather of the clean-up memory: dst = table[first argmax of sims]
template <typename T>
__device__ __forceinline__ void argmax_gather_body(const MicroOp<T>& op, T* __restrict__ sig, unsigned char* smem) {
  T* sred = reinterpret_cast<T*>(smem);
  int* sidx = reinterpret_cast<int*>(smem + 4 * sizeof(T));
  const int tid = threadIdx.x;
  T best = T(-INFINITY);
  int bi = 0x7fffffff;
  const T* sims = (const T*)op.p1;
  const int n_cand = op.src > 0 ? (int)op.src : (int)op.i0;
  const int* cand_idx = op.src > 0 ? (const int*)(sims + op.src) : nullptr;
  for (int i0 = tid; i0 < n_cand; i0 += 2048) {       // eight reads in flight, examined in ascending order
    T v[8];
    int vi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256;
      v[u] = i < n_cand ? sims[i] : T(-INFINITY);
      vi[u] = (cand_idx && i < n_cand) ? cand_idx[i] : i;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (v[u] > best) { best = v[u]; bi = vi[u]; }     // first maximum within a thread (ascending i)
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T ov = __shfl_down(best, off, 64);
    const int oi = __shfl_down(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { sred[tid >> 6] = best; sidx[tid >> 6] = bi; }
  __syncthreads();
  best = sred[0]; bi = sidx[0];
  for (int w = 1; w < 4; ++w)
    if (sred[w] > best || (sred[w] == best && sidx[w] < bi)) { best = sred[w]; bi = sidx[w]; }
  if (bi == 0x7fffffff) bi = 0;
  const T* const trow = (const T*)op.p0 + (size_t)bi * op.i1;
  T* const d = sig + op.dst;
  for (long long i = tid; i < op.len; i += 256) d[i] = trow[i];
}

// A serial chain (RK_SOLO): single-workgroup units in program order, a workgroup barrier between them.  At most one transform
// per chain, and it runs OUTSIDE the member loops (table: {n, code_0, sub_0, ...}; code < 0: the transform): inside a loop its
// descriptor and addressing stay live around the back edge and cost every block of the round kernel ~20 VGPRs.
template <typename T>
__device__ __forceinline__ void solo_body(const RoundArgs<T>& ra, const int* __restrict__ ch, unsigned char* smem) {
  // The chain table and the members' descriptors go to LDS first, all loads in flight together: read one at a time behind the
  // barrier of the member before, each member would start with two dependent trips to memory (table entry -> descriptor)
  // before the first load of its data - measured ~4 us per member instead of ~2.
  __shared__ int s_tab[2 * SOLO_MAX_MEMBERS + 2];
  __shared__ __align__(16) unsigned char s_ops[SOLO_MAX_MEMBERS * sizeof(MicroOp<T>)];
  const int tid = threadIdx.x;
  const int n = __builtin_amdgcn_readfirstlane(min(ch[0], SOLO_MAX_MEMBERS));
  if (tid < 2 * n) s_tab[tid] = ch[1 + tid];
  {
    constexpr int DW = (int)(sizeof(MicroOp<T>) / 4);
    static_assert(sizeof(MicroOp<T>) % 4 == 0, "descriptor copied by dwords");
    for (int i = tid; i < n * DW; i += 256) {
      const int q = i / DW, w = i - q * DW;
      const int code = ch[1 + 2 * q];
      if (code >= 0) reinterpret_cast<int*>(s_ops)[i] = reinterpret_cast<const int*>(ra.mops + code)[w];
    }
  }
  __syncthreads();
  auto members = [&](int q0, int q1) {
#pragma unroll 1
    for (int q = q0; q < q1; ++q) {
      const int sub = __builtin_amdgcn_readfirstlane(s_tab[2 * q + 1]);
      if (q) __syncthreads();
      const MicroOp<T>& op = reinterpret_cast<const MicroOp<T>*>(s_ops)[q];
      if (op.kind == M_GATE) gate_body<T>(op, ra.sig, smem);
      else if (op.kind == M_ARGMAX_GATHER) argmax_gather_body<T>(op, ra.sig, smem);
      else {
        const bool rows = op.kind == M_MATVEC_INC || op.kind == M_MATVEC_SET || op.kind == M_ENS_FINISH || op.kind == M_REDUCE_SET || op.kind == M_REDUCE_INC;
        const int chunks = (int)((op.len + (rows ? GLUE_ROWS : GLUE_CHUNK) - 1) / (rows ? GLUE_ROWS : GLUE_CHUNK));
#pragma unroll 1
        for (int c = 0; c < chunks; ++c) glue_body<T>(op, c, sub, ra.sig, ra.ctx);
      }
    }
  };
  int qd = n;                     // position of the transform (n: none)
#pragma unroll 1
  for (int q = 0; q < n; ++q) if (s_tab[2 * q] < 0) qd = q;
  qd = __builtin_amdgcn_readfirstlane(qd);
  members(0, qd);
  if (qd < n) {
    if (qd) __syncthreads();
    // (the descriptor is read through the constant address space: behind the stores of the members before it the compiler
    //  would otherwise fetch its uniform fields with vector loads and hold them in VGPRs)
    typedef const __attribute__((address_space(4))) DftArgs DftArgsK;
    if constexpr (sizeof(T) == 4) dft_body<true, DftArgsK>(*(DftArgsK*)(uintptr_t)(ra.arena + (size_t)(-s_tab[2 * qd] - 1) * 16), smem);
    members(qd + 1, n);
  }
}

// One virtual block of a round.
template <typename T>
__device__ __forceinline__ void round_block(const RoundArgs<T>& ra, int vb, unsigned char* smem) {
  int ei = 0;
  for (int q = 1; q < ra.n; ++q) if (vb >= ra.e[q].first) ei = q;       // (entries in ascending `first` order; uniform)
  const RoundEntry e = ra.e[ei];
  vb -= e.first;
  const int g = vb + e.lo;                   // (an operator may run in pieces, one per round: blocks [lo, lo + cnt) here)
  const int bx = g % e.gx, by = g / e.gx;
  switch (e.kind) {
    case RK_GLUE: {
      const GlueBlock gb = ((const GlueBlock*)e.args)[vb];
      if (gb.op >= 0) {
        glue_body<T>(ra.mops[gb.op], gb.chunk & 0xffffff, gb.chunk >> 24, ra.sig, ra.ctx);
      } else {
        const int* ch = ra.chain + (-gb.op - 1);
        const int n = ch[0];
        for (int q = 0; q < n; ++q) glue_body<T, true>(ra.mops[ch[1 + 2 * q]], gb.chunk, ch[2 + 2 * q], ra.sig, ra.ctx);
      }
      break;
    }
    case RK_SOLO: solo_body<T>(ra, (const int*)e.args, smem); break;
    case RK_GATE: gate_body<T>(*(const MicroOp<T>*)e.args, ra.sig, smem); break;
    case RK_ARGMAX: argmax_gather_body<T>(*(const MicroOp<T>*)e.args, ra.sig, smem); break;
    case RK_MATVEC_R1: matvec_body<T, true, 1, 4>(*(const MatvecArgs<T>*)e.args, bx, smem); break;
    // (one copy of the 16-rows-per-workgroup product for both: a plain product's descriptor has nr.V == nullptr)
    case RK_MATVEC_R4: case RK_MATVEC_NEURONS: matvec_neurons_body<T>(*(const MatvecNeuronsArgs<T>*)e.args, bx, smem); break;
    case RK_SPMV: spmv_body<T>(*(const SpmvArgs<T>*)e.args, bx, by, smem); break;
    case RK_NEURONS: neurons_body<T>(*(const NeuronsArgs<T>*)e.args, bx, smem); break;
    case RK_DFT:
      if constexpr (sizeof(T) == 4) dft_body<true>(*(const DftArgs*)e.args, smem);
      break;
    case RK_ENS_3_4_S: ens_body<T, 3, 4, 1, true>(*(const EnsArgs<T>*)e.args, bx, smem); break;
    case RK_ENS_3_5_S: ens_body<T, 3, 5, 1, true>(*(const EnsArgs<T>*)e.args, bx, smem); break;
    case RK_ENS_1_1_D: ens_body<T, 1, 1, 2, true>(*(const EnsArgs<T>*)e.args, bx, smem); break;
    case RK_ENS_SMALL: ens_small_body<T>(*(const EnsArgs<T>*)e.args, bx); break;
    case RK_GRID_LHS: grid_lhs_body<T>(*(const GridLhsArgs<T>*)e.args, bx); break;
    case RK_GRID_DOT: {      // row block bx of the remaining-axis factors against left-operand row by: similarities [by * nn + 16 bx ...)
      const GridDotArgs<T> g = *(const GridDotArgs<T>*)e.args;
      const MatvecArgs<T> m{g.W, g.A + (size_t)by * g.lda, g.dst + (size_t)by * g.nn, g.nn, g.k2, g.ldw, 1};
      matvec_body<T, true, 4>(m, bx, smem);
      break;
    }
    case RK_PES: pes_body<T>(*(const PesArgs<T>*)e.args, bx, by); break;
    case RK_VOJA: voja_body<T>(*(const VojaArgs<T>*)e.args, bx); break;
    default: break;
  }
}

#ifdef SSN_ROUND_STAMPS
// diagnostic build only (make F32_EXTRA=-DSSN_ROUND_STAMPS, tools/round_stamps.py): every block of every k_round launch leaves
// {launch id (RoundArgs::pad) << 32 | body kind << 24 | block index, s_memrealtime at its start, at its end} (100 MHz ticks)
constexpr unsigned int ROUND_STAMP_CAP = 1u << 20;
__device__ unsigned long long g_round_stamps[ROUND_STAMP_CAP][3];
__device__ unsigned int g_round_stamp_n;
inline long long read_round_stamps(unsigned long long* out, long long cap, int reset) {
  unsigned int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_round_stamp_n), sizeof n) != hipSuccess) return -1;
  const long long m = std::min<long long>(std::min<long long>(n, ROUND_STAMP_CAP), cap);
  if (out && m > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_round_stamps), (size_t)m * 24) != hipSuccess) return -1;
  if (reset) { n = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_round_stamp_n), &n, sizeof n) != hipSuccess) return -1; }
  return m;
}
#endif

// (experiment switch: -DSSN_ROUND_WAVES=8 asks the compiler for 64 VGPRs - 8 workgroups per CU instead of 7 - at the price of spills)
#ifdef SSN_ROUND_WAVES
#define SSN_ROUND_ATTR __attribute__((amdgpu_waves_per_eu(SSN_ROUND_WAVES, SSN_ROUND_WAVES)))
#else
#define SSN_ROUND_ATTR
#endif
template <typename T>
__global__ __launch_bounds__(256) SSN_ROUND_ATTR void k_round(RoundArgs<T> ra) {
  extern __shared__ __align__(16) unsigned char ssn_round_smem[];
  int vb = (int)blockIdx.x;
  if (ra.stride > 1u && vb >= ra.head) {
    const unsigned int m = gridDim.x - (unsigned int)ra.head;
    vb = ra.head + (int)(((unsigned long long)(unsigned int)(vb - ra.head) * ra.stride) % m);
  }
#ifdef SSN_ROUND_STAMPS
  unsigned long long st0, st1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st0) :: "memory");
#endif
  // The latency-bound bodies at the head of the grid (transforms, serial chains, gate, argmax, glue: one workgroup each, and the
  // round lasts as long as the slowest of them) share their SIMDs with up to six waves of bandwidth-bound blocks: per-block
  // stamps (tools/round_stamps.py) show a transform that takes 9.5 us alone taking 25 - 50 us there.  Their waves issue first.
  if (vb < ra.prio) __builtin_amdgcn_s_setprio(3);
  round_block<T>(ra, vb, ssn_round_smem);
#ifdef SSN_ROUND_STAMPS
  if constexpr (sizeof(T) == 4) {
    __syncthreads();
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st1) :: "memory");
    if (threadIdx.x == 0) {
      int ei = 0;
      for (int q = 1; q < ra.n; ++q) if (vb >= ra.e[q].first) ei = q;
      const unsigned int i = atomicAdd(&g_round_stamp_n, 1u);
      if (i < ROUND_STAMP_CAP) {
        g_round_stamps[i][0] = ((unsigned long long)(unsigned int)ra.pad << 32) | ((unsigned long long)(ra.e[ei].kind & 255) << 24) | (unsigned long long)(blockIdx.x & 0xffffff);
        g_round_stamps[i][1] = st0;
        g_round_stamps[i][2] = st1;
      }
    }
  }
#endif
}

// (A persistent variant - all rounds of a step graph in one resident grid with grid barriers - was built, tested and
//  measured in round 2: 411 us per timestep against 142 us with one launch per round at SLAM config 3, because every
//  workgroup's release / acquire fence is a write-back + invalidate of its XCD's whole L2; the barrier measurements are
//  in tools/grid_barrier.hip and profiles/round2_grid_barrier.txt.  It was removed.)

template <typename T>
hipError_t launch_round(hipStream_t s, const RoundArgs<T>& ra, int n_blocks, size_t lds_bytes) {
  static std::atomic<uint64_t> configured{0};
  if (hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(&k_round<T>), 64 * 1024, configured); e != hipSuccess) return e;
  if (lds_bytes > 64 * 1024 || n_blocks <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_round<T>), dim3((unsigned)n_blocks), dim3(256), std::max<size_t>(lds_bytes, 64), s, ra);
  return hipGetLastError();
}

}  // namespace ssn
