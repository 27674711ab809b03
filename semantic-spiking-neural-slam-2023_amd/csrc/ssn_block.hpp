// ssn_block.hpp - k_ens_block: a recurrent ensemble array stepped through a whole time block in ONE launch.
//
// Applies when the per-timestep core is an array of K ensembles whose only recurrence is
// ensemble k -> Lowpass -> ensemble k (the path integrator's VCO array, reference
// sspslam/networks/pathintegration.py:150-164): the K oscillators are then independent of each other
// inside a block - everything else they read (the velocity terms) was produced for all B timesteps of
// the block by the time-batched pre stage.  So instead of streaming every neuron parameter from HBM once
// per timestep (k_ensarray, HBM-bound: (din + dout + 5) words per neuron-step), one workgroup takes one
// ensemble, loads its encoders, biases, decoders and LIF state words into REGISTERS once (10 words per
// neuron at din 3, dout 5 - the 256 CUs' vector register files hold 128 MiB), and runs all B timesteps
// with a single workgroup barrier per step.  HBM traffic drops by the block length (256x); the kernel is
// bound by VALU issue instead.
//
// Per timestep and workgroup:
//   x[d]   = pre-stage row (block buffer) + alpha[d] * filter_state[xrow[d]]           (uniform, scalar)
//   for each of the thread's NPT neurons: J = e.x + bias; LIF step on the packed state word; acc += spike*dec
//   acc[dout] -> DPP wave reduction -> LDS [parity][dout][wave] -> barrier -> every thread adds the wave sums
//   in fixed order (deterministic; all threads hold identical totals)
//   filter_state[r] = a[r]*filter_state[r] + b[r]*total[r]; thread r hands total[r] to the post stage's row.
// At the end the state words go back to HBM and thread r publishes decoded value / filter state r to the
// signal vector (what k_ens_finish does after every step of the per-step plan).
#pragma once
#include <type_traits>
#include "ssn_launch.hpp"

namespace ssn {

// sum over the 64 lanes of a wave, result valid in lane 63 (fixed order -> deterministic)
__device__ inline float wave_sum_dpp(float v) {
  int x;
#define SSN_DPP_ADD(ctrl, rmask)                                                                             \
  x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false);                    \
  v += __builtin_bit_cast(float, x);
  SSN_DPP_ADD(0xB1, 0xF)     // quad_perm [1,0,3,2]
  SSN_DPP_ADD(0x4E, 0xF)     // quad_perm [2,3,0,1]
  SSN_DPP_ADD(0x141, 0xF)    // row_half_mirror
  SSN_DPP_ADD(0x140, 0xF)    // row_mirror: every lane of a 16-lane row holds the row sum
  SSN_DPP_ADD(0x142, 0xA)    // row_bcast15 -> rows 1, 3
  SSN_DPP_ADD(0x143, 0xC)    // row_bcast31 -> rows 2, 3: lane 63 = total
#undef SSN_DPP_ADD
  return v;
}
// sum within each 16-lane row, result in every lane of the row
__device__ inline float row_sum_dpp(float v) {
  int x;
#define SSN_DPP_ADD(ctrl)                                                                                    \
  x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false);                      \
  v += __builtin_bit_cast(float, x);
  SSN_DPP_ADD(0xB1) SSN_DPP_ADD(0x4E) SSN_DPP_ADD(0x141) SSN_DPP_ADD(0x140)
#undef SSN_DPP_ADD
  return v;
}
// N independent wave sums, stage by stage (result of each in lane 63)
template <int N>
__device__ inline void wave_sum_dpp_n(float* v) {
#define SSN_DPP_STAGE(ctrl, rmask)                                                                            \
  _Pragma("unroll") for (int r = 0; r < N; ++r) {                                                              \
    const int x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[r]), ctrl, rmask, 0xF, false);      \
    v[r] += __builtin_bit_cast(float, x);                                                                      \
  }
  SSN_DPP_STAGE(0xB1, 0xF) SSN_DPP_STAGE(0x4E, 0xF) SSN_DPP_STAGE(0x141, 0xF) SSN_DPP_STAGE(0x140, 0xF)
  SSN_DPP_STAGE(0x142, 0xA) SSN_DPP_STAGE(0x143, 0xC)
#undef SSN_DPP_STAGE
}
__device__ inline double wave_sum_dpp(double v) {     // parity/test instantiation: plain shuffles, result in every lane
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <int N>
__device__ inline void wave_sum_dpp_n(double* v) {
#pragma unroll
  for (int r = 0; r < N; ++r) v[r] = wave_sum_dpp(v[r]);
}

// Branch-free f32 LIF step on the packed state word (s >= 0: voltage; s < 0: minus the remaining refractory
// time), TWO neurons per call in the two halves of 64-bit register pairs so that the multiplies / adds / FMAs
// issue as packed v_pk_*_f32 (2 flops per lane per issue slot - the kernel is VALU-issue bound).  Value for value
// the arithmetic of k_ensarray's fast path (-R' = min(s,0) + dt is the exact negation of R - dt, and so on),
// except expm1's Taylor polynomial stopping one term earlier (below).  The spike branch is evaluated for every
// lane and selected: at ~4 % spikes per step some lane of a wave takes it for 93 % of the neurons anyway.
// Requires dt/tau_rc <= 1/8 (Taylor range) and tau_ref >= dt; the host planner checks both.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
struct LifConstF32 { float dt, neg_dt, neg_inv_tau, tau_rc_ln2, tau_ref_dt; };

__device__ inline f32x2 lif_packed_step_f32x2(f32x2 J, f32x2& s, const LifConstF32& c) {
  // min / max against 0 on the bit pattern (negative floats are negative integers): one instruction each,
  // where fminf / fmaxf cost a NaN-quieting pre-pass in IEEE mode
  const i32x2 sb = __builtin_bit_cast(i32x2, s);
  const f32x2 m = __builtin_bit_cast(f32x2, __builtin_elementwise_min(sb, (i32x2)(0)));    // -(remaining refractory time)
  const f32x2 V0 = __builtin_bit_cast(f32x2, __builtin_elementwise_max(sb, (i32x2)(0)));
  const f32x2 negR = m + c.dt;                     // -(R - dt)
  f32x2 delta = negR + c.dt;                       // dt - (R - dt)
  delta.x = __builtin_amdgcn_fmed3f(delta.x, 0.0f, c.dt);
  delta.y = __builtin_amdgcn_fmed3f(delta.y, 0.0f, c.dt);
  const f32x2 x = delta * c.neg_inv_tau;
  f32x2 q = (f32x2)(1.0f / 120.0f);                // expm1(x)/x to x^4/120: next term < 4e-8 relative at |x| <= 1/8
  q = __builtin_elementwise_fma(q, x, (f32x2)(1.0f / 24.0f));
  q = __builtin_elementwise_fma(q, x, (f32x2)(1.0f / 6.0f));
  q = __builtin_elementwise_fma(q, x, (f32x2)(0.5f));
  q = __builtin_elementwise_fma(q, x, (f32x2)(1.0f));
  const f32x2 V = V0 - (J - V0) * (q * x);
  const f32x2 vm1 = V - 1.0f, jm1 = J - 1.0f;
  f32x2 rc;
  rc.x = __builtin_amdgcn_rcpf(jm1.x);
  rc.y = __builtin_amdgcn_rcpf(jm1.y);
  const f32x2 omu = 1.0f - vm1 * rc;               // 1 - (V - 1) / (J - 1)
  f32x2 lg2;
  lg2.x = __builtin_amdgcn_logf(omu.x);            // log2; used only where the neuron spiked
  lg2.y = __builtin_amdgcn_logf(omu.y);
  const f32x2 Rs = __builtin_elementwise_fma((f32x2)(c.tau_rc_ln2), lg2, (f32x2)(c.tau_ref_dt));   // tau_ref + dt + tau_rc ln(1-u)
  const f32x2 Vc = __builtin_bit_cast(f32x2, __builtin_elementwise_max(__builtin_bit_cast(i32x2, V), (i32x2)(0)));
  f32x2 spk;
  {
    const float s_ns = negR.x < c.neg_dt ? negR.x : Vc.x;      // still refractory after this step ? -R : V
    const bool sp = V.x > 1.0f;
    s.x = sp ? -Rs.x : s_ns;                                   // (tau_ref >= dt: a spike is always followed by a refractory step)
    spk.x = sp ? 1.0f : 0.0f;
  }
  {
    const float s_ns = negR.y < c.neg_dt ? negR.y : Vc.y;
    const bool sp = V.y > 1.0f;
    s.y = sp ? -Rs.y : s_ns;
    spk.y = sp ? 1.0f : 0.0f;
  }
  return spk;
}

// Neurons are dealt to threads in groups of PK adjacent neurons (PK = 2 for f32: one 64-bit register pair,
// 1 for f64): neuron index of (group g, thread tid, component c) = (g * nthr + tid) * PK + c - coalesced.
template <typename T, int DIN, int DOUT, int NPT, int TPB, bool ENC_LDS, bool CLUSTER>
__global__ __launch_bounds__(TPB) void k_ens_block(BlockArgs<T> a) {
  extern __shared__ __align__(16) unsigned char ssn_block_dyn[];
  T* const e_lds = reinterpret_cast<T*>(ssn_block_dyn);       // ENC_LDS: encoders [DIN][nthr * NPT]
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int PK = F32 ? 2 : 1;
  constexpr int NG = NPT / PK;
  static_assert(NPT % PK == 0, "f32 variants handle neuron pairs");
  using G = typename std::conditional<F32, f32x2, T>::type;   // one group of neurons
  constexpr int DP = DOUT <= 4 ? 4 : 8;
  // Cluster mode (a.P > 1, few ensembles per GPU): P workgroups share one ensemble's neurons and exchange their
  // partial sums once per timestep.  Members of a cluster get block indices that are equal mod 8, i.e. land on
  // one XCD under round-robin placement (a speed bonus for the exchange, never needed for correctness).
  int k = blockIdx.x, p = 0;
  if (CLUSTER && a.P > 1) {
    const int b8 = blockIdx.x / (8 * a.P), rem = blockIdx.x - b8 * 8 * a.P;
    k = b8 * 8 + (rem & 7);
    p = rem >> 3;
    if (k >= a.K) return;                                  // padding of the last group of 8 ensembles
  }
  const int tid = threadIdx.x;
  const int nthr = NG == 1 ? (int)blockDim.x : TPB;       // variants with more than one group always run full workgroups
  const int lane = tid & 63, wave = tid >> 6;
  const size_t row = (size_t)a.n_pad;
  const T* __restrict__ enc = a.enc + (size_t)k * DIN * row;
  const T* __restrict__ bias = a.bias + (size_t)k * row;
  T* __restrict__ Sp = a.S + (size_t)k * row;
  const NeuronParams<T> np = a.np;
  const LifMath<T> lm(np);
  const LifConstF32 lc{(float)np.dt, -(float)np.dt, -1.0f / (float)np.tau_rc, (float)np.tau_rc * 0.6931471805599453f,
                       (float)np.tau_ref + (float)np.dt};

  // ---- parameters and state of this thread's neurons -> registers ---------------------------------------------
  G e[ENC_LDS ? 1 : NG][DIN], b[NG], s[NG], dc[NG][DOUT];
  const int cap = nthr * NPT;
  auto comp = [](G& v, int c) -> T& { if constexpr (F32) return c == 0 ? reinterpret_cast<T*>(&v)[0] : reinterpret_cast<T*>(&v)[1]; else return v; };
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int c = 0; c < PK; ++c) {
      const int li = (g * nthr + tid) * PK + c;          // index inside this workgroup's share
      const int i = p * cap + li;
      const bool ok = i < a.n;
#pragma unroll
      for (int d = 0; d < DIN; ++d) {
        const T ev = ok ? enc[d * row + i] : T(0);
        if constexpr (ENC_LDS) e_lds[d * cap + li] = ev; else comp(e[g][d], c) = ev;
      }
      comp(b[g], c) = ok ? bias[i] : T(0);
      comp(s[g], c) = ok ? Sp[i] : T(0);
      if (a.dec_neuron_major) {
        const T* dp = a.dec + ((size_t)k * row + i) * DP;
#pragma unroll
        for (int r = 0; r < DOUT; ++r) comp(dc[g][r], c) = ok ? dp[r] : T(0);
      } else {
        const T* dp = a.dec + (size_t)k * DOUT * row + i;
#pragma unroll
        for (int r = 0; r < DOUT; ++r) comp(dc[g][r], c) = ok ? dp[r * row] : T(0);
      }
    }
  }

  // ---- per-row constants (uniform) -------------------------------------------------------------------------
  T fs[DOUT], la[DOUT], lb[DOUT], xa[DIN];
  int xr[DIN];
#pragma unroll
  for (int r = 0; r < DOUT; ++r) {
    const long long i = (long long)k * DOUT + r;
    const int st = a.lp_state[i];
    fs[r] = st >= 0 ? a.sig[st] : T(0);
    la[r] = st >= 0 ? a.lp_a[i] : T(0);
    lb[r] = st >= 0 ? a.lp_b[i] : T(0);
  }
#pragma unroll
  for (int d = 0; d < DIN; ++d) {
    xr[d] = a.xrow[(long long)k * DIN + d];
    xa[d] = a.xalpha[(long long)k * DIN + d];
  }

  // ---- LDS: wave sums (two parities), and the block-buffer traffic of CH timesteps at a time so that no global
  //      memory operation sits inside the time loop (a workgroup barrier waits for every outstanding one) ----
  constexpr int CH = 128;                    // timesteps per input/output chunk
  constexpr int XP = 4;                      // words per timestep of inputs (DIN <= 4): one 16-byte LDS read
  constexpr int RW = DOUT <= 4 ? 64 : 128;   // wave-sum slots per parity: [r][16 waves]
  __shared__ __align__(16) T red[2][RW];
  __shared__ __align__(16) T xs[CH][XP];
  __shared__ T os[CH][DOUT];
  __shared__ int s_dst[DOUT];
  __shared__ int s_out[DOUT];
  constexpr int PMAX = 4;                    // largest cluster
  constexpr int GV = sizeof(T) / 4;          // 32-bit payload words per value (exchange granule = {payload, tag})
  __shared__ unsigned int xt[2][CLUSTER ? PMAX * DOUT * GV : 1];
  for (int i = tid; i < 2 * RW; i += nthr) (&red[0][0])[i] = T(0);     // waves that do not exist add 0
  if (tid < DOUT) { s_dst[tid] = a.didx[(long long)k * DOUT + tid]; s_out[tid] = a.rowout[(long long)k * DOUT + tid] ? 1 : 0; }
  const T* __restrict__ xbase = a.xrows + (size_t)a.row0 * a.n_sig + a.x_off + (long long)k * DIN;
  T tot[DOUT];
#pragma unroll
  for (int r = 0; r < DOUT; ++r) tot[r] = T(0);

  G en[DIN];                                 // ENC_LDS: encoders of the group about to be processed
  if constexpr (ENC_LDS) {
#pragma unroll
    for (int d = 0; d < DIN; ++d) en[d] = *reinterpret_cast<const G*>(e_lds + d * cap + tid * PK);
  }
  for (int j0 = 0; j0 < a.B; j0 += CH) {
    const int cn = min(CH, a.B - j0);
    __syncthreads();                         // previous chunk: every wave is past its last xs read / os write
    if (j0 > 0) {                            // hand the previous chunk's decoded rows to the post stage
      for (int i = tid; i < CH * DOUT; i += nthr) {
        const int jj = i / DOUT, r = i - jj * DOUT;
        if (s_out[r] && p == 0) a.bsig[(size_t)(a.row0 + j0 - CH + jj) * a.n_sig + s_dst[r]] = os[jj][r];
      }
    }
    for (int i = tid; i < cn * DIN; i += nthr) {
      const int jj = i / DIN, d = i - jj * DIN;
      xs[jj][d] = xbase[(size_t)(j0 + jj) * a.n_sig + d];
    }
    __syncthreads();

    for (int jj = 0; jj < cn; ++jj) {
      T xin[XP];
      if constexpr (sizeof(T) == 4) *(float4*)xin = *(const float4*)xs[jj];
      else { *(double2*)xin = *(const double2*)xs[jj]; *(double2*)(xin + 2) = *(const double2*)(xs[jj] + 2); }
      T x[DIN];
#pragma unroll
      for (int d = 0; d < DIN; ++d) {
        T st = T(0);
#pragma unroll
        for (int r = 0; r < DOUT; ++r) st = xr[d] == r ? fs[r] : st;
        x[d] = xr[d] >= 0 ? xin[d] + xa[d] * st : xin[d];
      }
      G accg[DOUT];
#pragma unroll
      for (int r = 0; r < DOUT; ++r) accg[r] = G(0);
      // ENC_LDS: a thread reads back only the entries it wrote itself (no barrier needed); the next group's
      // encoders are requested one group ahead so the LDS latency hides under this group's arithmetic (with two
      // waves per SIMD an exposed ds_read stalls the SIMD)
      G nx[DIN];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if constexpr (ENC_LDS) {          // (the last group requests group 0 again: the next timestep's first operands)
          const int gn = g + 1 < NG ? g + 1 : 0;
#pragma unroll
          for (int d = 0; d < DIN; ++d) nx[d] = *reinterpret_cast<const G*>(e_lds + d * cap + (gn * nthr + tid) * PK);
        }
        G J = b[g];
#pragma unroll
        for (int d = 0; d < DIN; ++d) {
          if constexpr (ENC_LDS) J += en[d] * x[d];
          else J += e[g][d] * x[d];
        }
        if constexpr (ENC_LDS) {
#pragma unroll
          for (int d = 0; d < DIN; ++d) en[d] = nx[d];
        }
        G spk;
        if constexpr (F32) spk = lif_packed_step_f32x2(J, s[g], lc);
        else {
          // packed state word -> nengo's LIF step, operation for operation k_ensarray's fast path
          const T sw = s[g];
          T V = sw < T(0) ? T(0) : sw;
          T R = (sw < T(0) ? -sw : T(0)) - np.dt;
          T delta = np.dt - R;
          delta = delta < T(0) ? T(0) : (delta > np.dt ? np.dt : delta);
          V = V - (J - V) * lm.decay(delta);
          spk = T(0);
          if (V > T(1)) {
            const T t_spike = np.dt + np.tau_rc * lm.spike_time_term(V, J);
            R = np.tau_ref + t_spike;
            V = T(0);
            spk = T(1);
          } else if (V < T(0)) {
            V = T(0);
          }
          s[g] = R > np.dt ? -R : V;
        }
#pragma unroll
        for (int r = 0; r < DOUT; ++r) {           // spk is 0 or 1: exact add
          if constexpr (F32) accg[r] = __builtin_elementwise_fma(spk, dc[g][r], accg[r]);
          else accg[r] = fma(spk, dc[g][r], accg[r]);
        }
      }
      T acc[DOUT];
#pragma unroll
      for (int r = 0; r < DOUT; ++r) {
        if constexpr (F32) acc[r] = accg[r].x + accg[r].y; else acc[r] = accg[r];
      }
      const int par = jj & 1;
#if defined(SSN_BLOCK_EXPERIMENT) && SSN_BLOCK_EXPERIMENT == 1      /* timing bisection only: no reduction (wrong results) */
#pragma unroll
      for (int r = 0; r < DOUT; ++r) tot[r] = acc[r] * T(1e-30);
#else
      // the DOUT wave reductions are independent: all sums first (their DPP stages interleave and fill each
      // other's wait states), then one predicated store block
      T wsum[DOUT];
#pragma unroll
      for (int r = 0; r < DOUT; ++r) wsum[r] = acc[r];
      wave_sum_dpp_n<DOUT>(wsum);
      if (lane == 63) {
#pragma unroll
        for (int r = 0; r < DOUT; ++r) red[par][r * 16 + wave] = wsum[r];
      }
      __syncthreads();
      if constexpr (sizeof(T) == 4) {
        // slot r*16 + w sits in lane r*16 + w: a 16-lane DPP row reduction adds the waves, readlane makes the
        // totals scalar (identical in every wave: same inputs, same fixed order)
        float v0 = red[par][lane];
        v0 = row_sum_dpp(v0);
#pragma unroll
        for (int r = 0; r < (DOUT < 4 ? DOUT : 4); ++r) tot[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), r * 16));
        if constexpr (DOUT > 4) {
          float v1 = red[par][64 + lane];
          v1 = row_sum_dpp(v1);
#pragma unroll
          for (int r = 4; r < DOUT; ++r) tot[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v1), (r - 4) * 16));
        }
      } else {
#pragma unroll
        for (int r = 0; r < DOUT; ++r) {
          T t = T(0);
#pragma unroll
          for (int w = 0; w < 16; ++w) t += red[par][r * 16 + w];
          tot[r] = t;
        }
      }
#endif
      if (CLUSTER && a.P > 1) {
        // ---- cluster exchange: publish this workgroup's partial sums as 8-byte {payload, tag = step} granules
        //      (one atomic store each: data and flag arrive together), collect the P partials of the step, add
        //      them in member order - every member ends up with bit-identical totals.  Two parities of slots:
        //      a member can only publish step t+2 after every other member has read its step-t granules.
        const unsigned int tag = (unsigned int)(a.step0 + j0 + jj + 1);
        unsigned long long* slot = a.xch + (((size_t)k * 2 + par) * PMAX) * (DOUT * GV);
        const int nmine = DOUT * GV, nall = a.P * DOUT * GV;
        if (wave == 0) {
          if (lane < nmine) {
            const int r = lane / GV, w = lane - r * GV;
            T v = T(0);
#pragma unroll
            for (int q = 0; q < DOUT; ++q) v = r == q ? tot[q] : v;
            unsigned int bits;
            if constexpr (GV == 1) bits = __builtin_bit_cast(unsigned int, v);
            else { const unsigned long long u = __builtin_bit_cast(unsigned long long, v); bits = w ? (unsigned int)(u >> 32) : (unsigned int)u; }
            __hip_atomic_store(slot + p * nmine + lane, ((unsigned long long)tag << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (lane < nall) {
            unsigned long long gr = 0;
            int spins = 0;
            for (;;) {
              gr = __hip_atomic_load(slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((unsigned int)(gr >> 32) == tag) break;
              if (++spins > (1 << 22)) { *a.err = 1; break; }          // a missing member: give up, the host reports it
              __builtin_amdgcn_s_sleep(1);
            }
            xt[par][lane] = (unsigned int)gr;
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < DOUT; ++r) {
          T t = T(0);
          for (int q = 0; q < a.P; ++q) {
            T v;
            if constexpr (GV == 1) v = __builtin_bit_cast(T, xt[par][q * nmine + r]);
            else v = __builtin_bit_cast(T, ((unsigned long long)xt[par][q * nmine + r * GV + 1] << 32) | xt[par][q * nmine + r * GV]);
            t += v;
          }
          tot[r] = t;
        }
      }
#pragma unroll
      for (int r = 0; r < DOUT; ++r) fs[r] = la[r] * fs[r] + lb[r] * tot[r];    // rows without a filter: la = lb = 0
      if (tid < DOUT) {
        T v = T(0);
#pragma unroll
        for (int r = 0; r < DOUT; ++r) v = tid == r ? tot[r] : v;
        os[jj][tid] = v;
      }
    }
  }
  __syncthreads();
  {                                          // last chunk's decoded rows
    const int j0 = (a.B - 1) / CH * CH, cn = a.B - j0;
    for (int i = tid; i < cn * DOUT; i += nthr) {
      const int jj = i / DOUT, r = i - jj * DOUT;
      if (s_out[r] && p == 0) a.bsig[(size_t)(a.row0 + j0 + jj) * a.n_sig + s_dst[r]] = os[jj][r];
    }
  }

  // ---- state back to HBM; decoded values and filter states of the last step to the signal vector ------------
  int tid2 = tid;
  asm volatile("" : "+v"(tid2));      // fresh addresses: keeps NPT pointers from staying live across the time loop
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int c = 0; c < PK; ++c) {
      const int i = p * cap + (g * nthr + tid2) * PK + c;
      if (i < a.n) Sp[i] = comp(s[g], c);
    }
  }
  if (tid < DOUT && a.B > 0 && p == 0) {
    T v = T(0), f = T(0);
#pragma unroll
    for (int r = 0; r < DOUT; ++r) { v = tid == r ? tot[r] : v; f = tid == r ? fs[r] : f; }
    a.sig_w[s_dst[tid]] = v;
    const int st = a.lp_state[(long long)k * DOUT + tid];
    if (st >= 0) a.sig_w[st] = f;
  }
}

// (workgroup size, neurons per thread, encoders in LDS) variants.  The register budget of a wave is
// 512 / (waves per SIMD): 1024 threads -> 128 registers, 512 -> 256, 256 -> 512 (incl. AGPRs); a neuron needs
// din + dout + 2 persistent words (10 at din 3, dout 5), 7 with the encoders in LDS.
template <typename T, int DIN, int DOUT, int NPT, int TPB, bool ENC_LDS, bool CLUSTER = false>
static hipError_t launch_block_variant(hipStream_t s, const BlockArgs<T>& a) {
  // cluster mode is compiled only for the variants a shard of a big ensemble lands on (and the f64 test sizes)
  constexpr bool HAS_CLUSTER = !ENC_LDS && ((sizeof(T) == 8) || (TPB == 512 && (NPT == 6 || NPT == 10)) || (TPB == 1024 && NPT <= 4));
  if constexpr (!CLUSTER && HAS_CLUSTER) {
    if (a.P > 1) return launch_block_variant<T, DIN, DOUT, NPT, TPB, ENC_LDS, true>(s, a);
  }
  if (!CLUSTER && a.P > 1) return hipErrorInvalidValue;
  const int lds = ENC_LDS ? DIN * a.threads * NPT * (int)sizeof(T) : 0;
  static bool configured = false;
  if (ENC_LDS && !configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ens_block<T, DIN, DOUT, NPT, TPB, ENC_LDS, CLUSTER>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DIN * TPB * NPT * (int)sizeof(T));
    if (e != hipSuccess) return e;
    configured = true;
  }
  const unsigned grid = a.P > 1 ? (unsigned)((a.K + 7) / 8 * 8 * a.P) : (unsigned)a.K;
  hipLaunchKernelGGL((k_ens_block<T, DIN, DOUT, NPT, TPB, ENC_LDS, CLUSTER>), dim3(grid), dim3((unsigned)a.threads), lds, s, a);
  return hipGetLastError();
}

template <typename T, int DIN, int DOUT>
static hipError_t launch_block_npt(hipStream_t s, const BlockArgs<T>& a) {
  const int key = (a.tpb * 100 + a.npt) * 2 + (a.enc_lds ? 1 : 0);
  switch (key) {
#define SSN_CASE(TPB, N, L) case (TPB * 100 + N) * 2 + L: return launch_block_variant<T, DIN, DOUT, N, TPB, (L != 0)>(s, a);
    SSN_CASE(1024, 2, 0) SSN_CASE(1024, 4, 0)
#undef SSN_CASE
#define SSN_CASE(TPB, N, L) case (TPB * 100 + N) * 2 + L: if constexpr (sizeof(T) == 8) return launch_block_variant<T, DIN, DOUT, N, TPB, (L != 0)>(s, a); else return hipErrorInvalidValue;
    SSN_CASE(1024, 1, 0)
#undef SSN_CASE
#define SSN_CASE(TPB, N, L) case (TPB * 100 + N) * 2 + L: if constexpr (sizeof(T) == 4) return launch_block_variant<T, DIN, DOUT, N, TPB, (L != 0)>(s, a); else return hipErrorInvalidValue;
    SSN_CASE(1024, 6, 0) SSN_CASE(512, 6, 0) SSN_CASE(512, 10, 0) SSN_CASE(1024, 10, 1) SSN_CASE(768, 14, 1) SSN_CASE(512, 16, 0) SSN_CASE(512, 20, 0) SSN_CASE(512, 20, 1) SSN_CASE(256, 40, 0)
#undef SSN_CASE
    default: return hipErrorInvalidValue;
  }
}

template <typename T>
bool ens_block_supported(int din, int dout, int n, int* threads, int* tpb, int* npt, int* enc_lds) {
  if (!(din == 3 && dout >= 3 && dout <= 5)) return false;
  struct V { int tpb, npt, lds; };
  const V f32v[] = {{1024, 2, 0}, {1024, 4, 0}, {512, 6, 0}, {1024, 6, 0}, {512, 10, 0}, {512, 20, 1}, {1024, 10, 1}, {768, 14, 1}, {256, 40, 0}, {512, 16, 0}, {512, 20, 0}};
  const V f64v[] = {{1024, 1, 0}, {1024, 2, 0}, {1024, 4, 0}};
  const V* vs = sizeof(T) == 4 ? f32v : f64v;
  const int nv = sizeof(T) == 4 ? 11 : 3;
  int want_tpb = 0, want_npt = 0, want_lds = 0;
  if (const char* env = getenv("SSN_BLOCK_VARIANT")) sscanf(env, "%d,%d,%d", &want_tpb, &want_npt, &want_lds);   // tuning knob
  for (int i = 0; i < nv; ++i) {
    if (want_tpb && (vs[i].tpb != want_tpb || vs[i].npt != want_npt || vs[i].lds != want_lds)) continue;
    int th = vs[i].tpb;
    const int pk = sizeof(T) == 4 ? 2 : 1;         // neurons per group; single-group variants run n / pk threads
    if (vs[i].npt == pk && n < th * pk) th = std::max(64, ((n + pk - 1) / pk + 63) / 64 * 64);
    if ((int64_t)th * vs[i].npt >= n) { *threads = th; *tpb = vs[i].tpb; *npt = vs[i].npt; *enc_lds = vs[i].lds; return true; }
  }
  return false;
}

template <typename T>
hipError_t launch_ens_block(hipStream_t s, const BlockArgs<T>& a) {
  if (a.din == 3 && a.dout == 5) return launch_block_npt<T, 3, 5>(s, a);
  if (a.din == 3 && a.dout == 4) return launch_block_npt<T, 3, 4>(s, a);
  if (a.din == 3 && a.dout == 3) return launch_block_npt<T, 3, 3>(s, a);
  return hipErrorInvalidValue;
}

}  // namespace ssn
