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
//   x[d]   = pre-stage row (block buffer, staged in LDS) + alpha[d] * filter_state[xrow[d]]
//   for each of the thread's NPT neurons: J = e.x + bias; LIF step on the packed state word; acc += spike*dec
//   acc[dout] -> sums over the wave -> LDS [parity][dout][wave] -> ONE barrier -> every wave adds the wave sums in the
//   same fixed order (deterministic; all waves hold identical totals)
//   filter_state[r] = a[r]*filter_state[r] + b[r]*total[r]; total[r] is staged for the post stage's row.
// At the end the state words go back to HBM and the decoded values / filter states of the last step go to the
// signal vector (what k_ens_finish does after every step of the per-step plan).
// f32 state words hold -(R - dt) for a refractory neuron (see lif_input_part), f64 ones -R (k_ensarray's fast path).
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

// ---- packed f32 arithmetic with the VOP3P clamp modifier: result clamped to [0, 1], NaN -> 0 (DX10 clamp mode) ------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ inline f32x2 pk_clamp01(f32x2 x) {
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, 1.0 op_sel_hi:[1,0] clamp" : "=v"(r) : "v"(x));
  return r;
}
__device__ inline f32x2 pk_sub_clamp01(f32x2 x, f32x2 y) {          // clamp(x - y), y a uniform constant pair
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "v"(x), "s"(y));
  return r;
}
__device__ inline f32x2 pk_mul_clamp01(f32x2 x, f32x2 y) {
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(x), "s"(y));          // (y: a uniform constant pair)
  return r;
}
__device__ inline f32x2 pk_fma_clamp01_vsv(f32x2 x, f32x2 y, f32x2 z) {          // clamp(x * y + z), y a uniform constant pair
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(x), "s"(y), "v"(z));
  return r;
}
// clamp(x * y + z), z a uniform constant pair, x the RESULT OF A TRANSCENDENTAL (v_log_f32): gfx940-family hardware needs one
// wait state between a transcendental's write and a non-transcendental VALU read of that register, and the compiler's hazard
// recognizer does not look inside inline asm - so the wait state is part of the asm (without it the DOUT = 5 variants, whose
// schedule put the v_log right in front, read the stale register: caught by the sharded-runner f32 test).
__device__ inline f32x2 pk_fma_clamp01_trans_vvs(f32x2 x, f32x2 y, f32x2 z) {
  f32x2 r;
  asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(x), "v"(y), "s"(z));
  return r;
}

__device__ inline f32x2 pk_fma_clamp01_vs_negv(f32x2 x, f32x2 y, f32x2 z) {     // clamp(x * y - z), y a uniform constant pair
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1] clamp" : "=v"(r) : "v"(x), "s"(y), "v"(z));
  return r;
}

// Branch-free f32 LIF step, TWO neurons per call in the two halves of 64-bit register pairs, built ONLY from packed
// multiply / add / FMA (v_pk_*_f32: 2 flops per lane per issue slot) plus one v_rcp_f32 and one v_log_f32 per neuron:
// no compare, no select, no integer min / max.  On gfx950 every non-FMA vector instruction (v_cndmask, v_cmp, v_max_i32,
// v_med3 ...) costs a full issue slot of its own per NEURON where a packed one serves two (v_cndmask: five of them - measured,
// tools/valu_issue_rate.hip, profiles/round3_valu_issue_rate.txt), so the selects of nengo's LIF step (SURVEY
// Appendix A.4) are rewritten as exact arithmetic on {0, 1} indicators made with the clamp modifier.
//
// Inside the time loop the state word is w = 1 - V for a neuron that integrates (0 <= w <= 1: the distance to the threshold)
// and w = 1 + K (R - dt) for a refractory one (R = refractory time left at the start of the next step, > dt; K a power of two
// with K tau_ref <= 1: times in units of 1 / K so that they fit the clamp's [0, 1] without losing bits next to the 1).  The
// input current arrives as J - 1 (the bias registers hold bias - 1).  With everything measured from the threshold, V - 1 and
// J - 1 - the operands of the spike test and of the spike time - are what the update produces anyway: 15 packed operations
// per neuron pair against 19 with V and J (round 3's second cut; the words in HBM keep round 2's form, see the prologue).
//
//   W0 = clamp(w)                 1 - V0  (V0 = 0 while refractory)
//   dl = clamp(c - a w)           integration time / dt  (nengo: clip(dt - refractory, 0, dt)): a = 1 / (K dt), c = a + 1 rounded
//                                 up, so that every integrating neuron (w <= 1) gets exactly 1 and a refractory one 1 - (R - dt) / dt
//   nmt = clamp(w - (1 + K dt))   > 0 while the neuron stays refractory beyond the next step: its new word is 1 + nmt
//   em = dl * P(dl)               -expm1(-delta / tau_rc), P of degree 2: the interpolant of (1 - exp(-x)) / x through x = h,
//                                 h / 2, 0.067 h with h = dt / tau_rc <= 1/20 - EXACT for a full step (delta = dt, all steps
//                                 but the one in which a refractory period ends), within 2.3e-7 of the result for a partial
//                                 one (a 4-term Taylor polynomial, one operation more, was within 5.3e-8 everywhere)
//   U  = (Jm1 + W0) * em - W0     V - 1 = V0 + (J - V0) em - 1
//   spk = clamp(U * 2^100)        1 if V > 1, else 0 (a positive U is at least 2^-53: it is the rounded difference of a
//                                 product >= 2^-6 and a W0 next to it)
//   nu = clamp(K tau_ref + K tau_rc ln(1 - u)), u = U / Jm1      K (tau_ref + t_spike - dt) of a spiking neuron, inside [0, 1]
//                                 because dt <= tau_ref.  A neuron that crosses the threshold in this step has
//                                 0 <= u <= 1 - exp(-dt / tau_rc) <= 0.049 (the overshoot V - 1 is at most (J - 1) (1 - exp(-delta / tau_rc))),
//                                 and there ln(1 - u) = u phi(u) with phi = ln(1 - u) / u replaced by its quadratic interpolant through
//                                 three Chebyshev nodes of [0, u_max]: within 5e-8 of ln(1 - u) - what v_log_f32 of the ROUNDED 1 - u
//                                 gives as well - for three packed FMAs instead of an FMA and two transcendentals per neuron pair
//                                 (round 4: 40 -> 20 transcendentals per wave and timestep).  A silent neuron feeds rcp whatever
//                                 its operand is (0, negative: inf, NaN); the clamp turns any of it into a number in [0, 1]
//                                 (DX10 clamp: NaN -> 0), spk = 0 discards it
//   Wn = clamp(spk * 2^100 - U)   1 - Vn: voltage of a silent neuron clamped at min_voltage 0, Vn = 0 for a spiking one
//   w' = (Wn + nmt) + spk * nu    (spiking: Wn = 1, nmt = 0; silent: spk = 0 - exact selects, products with 0 / 1)
// Requires dt / tau_rc <= 1/20 and tau_ref >= dt (checked by the host planner).
struct LifConstV3 { float na, ca, m1, c1, c2, c3, p0, p1, p2, ktau_ref, K; };      // p: K tau_rc phi(u) ~ p0 + p1 u + p2 u^2

// K, the constants of dl / nmt and the coefficients of P for dl in units of dt (uniform; evaluated once per launch, in double)
__device__ inline LifConstV3 lif_const_v3(double dt, double tau_rc, double tau_ref) {
  double K = 1.0;
  while (2.0 * K * tau_ref <= 1.0 && K < 1048576.0) K *= 2.0;
  while (K * tau_ref > 1.0) K *= 0.5;
  const double h = dt / tau_rc;
  const double x0 = h, x1 = 0.5 * h, x2 = 0.0669872981077807 * h;
  auto f = [](double x) { return -expm1(-x) / x; };
  const double f0 = f(x0), f1 = f(x1), f2 = f(x2);
  // Newton form of the interpolant, expanded to monomials
  const double d01 = (f1 - f0) / (x1 - x0), d12 = (f2 - f1) / (x2 - x1), d012 = (d12 - d01) / (x2 - x0);
  const double q2 = d012, q1 = d01 - d012 * (x0 + x1), q0 = f0 - d01 * x0 + d012 * x0 * x1;
  LifConstV3 c;
  const float a = (float)(1.0 / (K * dt));
  float ca = a + 1.0f;
  if ((double)ca < (double)a + 1.0) ca = __builtin_bit_cast(float, __builtin_bit_cast(unsigned int, ca) + 1u);      // next float up: c - a >= 1 in exact arithmetic, a full step is 1, never 1 - ulp
  c.na = -a; c.ca = ca; c.m1 = (float)(1.0 + K * dt);
  c.c1 = (float)(q0 * h); c.c2 = (float)(q1 * h * h); c.c3 = (float)(q2 * h * h * h);       // x = dl * h
  {
    // phi(u) = ln(1 - u) / u on [0, u_max], u_max = 1 - exp(-h): interpolant through the Chebyshev nodes, times K tau_rc
    const double um = -expm1(-h);
    const double u0 = um * 0.9330127018922193, u1 = um * 0.5, u2 = um * 0.0669872981077807;
    auto phi = [](double u) { return log1p(-u) / u; };
    const double g0 = phi(u0), g1 = phi(u1), g2 = phi(u2);
    const double e01 = (g1 - g0) / (u1 - u0), e12 = (g2 - g1) / (u2 - u1), e012 = (e12 - e01) / (u2 - u0);
    const double r2 = e012, r1 = e01 - e012 * (u0 + u1), r0 = g0 - e01 * u0 + e012 * u0 * u1;
    c.p0 = (float)(K * tau_rc * r0); c.p1 = (float)(K * tau_rc * r1); c.p2 = (float)(K * tau_rc * r2);
  }
  c.ktau_ref = (float)(K * tau_ref); c.K = (float)K;
  return c;
}

// The step in two halves: lif_state_part needs only the state word, lif_input_part is what needs J.  (Measured and dropped in
// round 3: running the state half for all of a thread's neurons in the tail of the previous timestep, behind the workgroup
// barrier or - the older wave of each SIMD - in front of it, 40 more registers: 2.90 - 2.92 ms per 1000 steps either way against
// 2.91 - 2.92 with the halves back to back, same box; alternating s_setprio between the two waves of a SIMD: 3.21 ms.)
__device__ inline void lif_state_part(f32x2 w, const LifConstV3& c, f32x2& W0, f32x2& em) {
  W0 = pk_clamp01(w);
  const f32x2 dl = pk_fma_clamp01_vsv(w, (f32x2)(c.na), (f32x2)(c.ca));      // clamp(c - a w)
  // (uniform coefficients as scalar-register pairs: one constant-bus operand per instruction, the first FMA's second
  //  coefficient lives in a vector register pair)
  f32x2 P;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P) : "v"(dl), "v"((f32x2)(c.c3)), "s"((f32x2)(c.c2)));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P) : "v"(dl), "v"(P), "s"((f32x2)(c.c1)));
  em = dl * P;
}
// The input half in three pieces, so that a wave whose neurons are all silent in a timestep can leave out the spike-time
// arithmetic (round 4): lif_spike_test produces U = V - 1, the spike indicators and the stay-refractory term; lif_finish_spiking
// is the rest of lif_input_part; lif_finish_silent is what that rest computes when every indicator is 0 - bit for bit:
// Wn = clamp(0 * 2^100 - U) = clamp(-U), w' = fma(0, nu, Wn + nmt) = Wn + nmt (nu is a number in [0, 1] after its clamp, so
// 0 * nu = +0, and Wn + nmt >= 0).
__device__ inline f32x2 lif_spike_test(f32x2 Jm1, f32x2 w, f32x2 W0, f32x2 em, const LifConstV3& c, f32x2 big, f32x2& U, f32x2& nmt) {
  nmt = pk_sub_clamp01(w, (f32x2)(c.m1));                                    // clamp(w - 1 - K dt)
  U = __builtin_elementwise_fma(Jm1 + W0, em, -W0);
  return pk_mul_clamp01(U, big);
}
__device__ inline void lif_finish_spiking(f32x2 Jm1, f32x2& w, f32x2 U, f32x2 nmt, f32x2 spk, const LifConstV3& c, f32x2 big) {
  f32x2 rc;
  rc.x = __builtin_amdgcn_rcpf(Jm1.x);
  rc.y = __builtin_amdgcn_rcpf(Jm1.y);
  const f32x2 u = U * rc;                                                    // (V - 1) / (J - 1)
  f32x2 P, nu;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P) : "v"(u), "v"((f32x2)(c.p2)), "s"((f32x2)(c.p1)));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P) : "v"(u), "v"(P), "s"((f32x2)(c.p0)));
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(nu) : "v"(u), "v"(P), "s"((f32x2)(c.ktau_ref)));
  const f32x2 Wn = pk_fma_clamp01_vs_negv(spk, big, U);
  w = __builtin_elementwise_fma(spk, nu, Wn + nmt);
}
__device__ inline void lif_finish_silent(f32x2& w, f32x2 U, f32x2 nmt) {
  f32x2 Wn;
  asm("v_pk_mul_f32 %0, %1, -1.0 op_sel_hi:[1,0] clamp" : "=v"(Wn) : "v"(U));          // clamp(-U)
  w = Wn + nmt;
}
__device__ inline f32x2 lif_input_part(f32x2 Jm1, f32x2& w, f32x2 W0, f32x2 em, const LifConstV3& c, f32x2 big) {
  const f32x2 nmt = pk_sub_clamp01(w, (f32x2)(c.m1));                        // clamp(w - 1 - K dt)
  const f32x2 U = __builtin_elementwise_fma(Jm1 + W0, em, -W0);
  const f32x2 spk = pk_mul_clamp01(U, big);
  f32x2 rc;
  rc.x = __builtin_amdgcn_rcpf(Jm1.x);
  rc.y = __builtin_amdgcn_rcpf(Jm1.y);
  // Wn = clamp(spk * 2^100 - U) sits between the reciprocals and their first reader: the wait state a non-transcendental reader
  // of a transcendental's result needs is filled by an instruction that has to be issued anyway (the compiler's hazard recognizer
  // does not look inside inline asm, so the order is fixed by ONE asm statement: Wn, then u = U * rc)
  // nu = clamp(K tau_ref + u (p0 + u (p1 + u p2))): the spike time without v_log (see the table above)
  f32x2 Wn, u, P, nu;
  asm("v_pk_fma_f32 %0, %2, %3, %4 neg_lo:[0,0,1] neg_hi:[0,0,1] clamp\n\t"
      "v_pk_mul_f32 %1, %4, %5"
      : "=&v"(Wn), "=v"(u)
      : "v"(spk), "s"(big), "v"(U), "v"(rc));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P) : "v"(u), "v"((f32x2)(c.p2)), "s"((f32x2)(c.p1)));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P) : "v"(u), "v"(P), "s"((f32x2)(c.p0)));
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(nu) : "v"(u), "v"(P), "s"((f32x2)(c.ktau_ref)));
  w = __builtin_elementwise_fma(spk, nu, Wn + nmt);
  return spk;
}
// HBM state word (s >= 0: voltage; s < 0: -(R - dt)) <-> loop word
__device__ inline float lif_word_in(float s, float K) { return __builtin_fmaf(-s, s < 0.0f ? K : 1.0f, 1.0f); }
__device__ inline float lif_word_out(float w, float K) { return (1.0f - w) * (w > 1.0f ? 1.0f / K : 1.0f); }

__device__ inline float bits_f(unsigned int x) { return __builtin_bit_cast(float, x); }
__device__ inline unsigned int f_bits(float x) { return __builtin_bit_cast(unsigned int, x); }

#ifdef SSN_BLOCK_STAMPS
// diagnostic build only (make F32_EXTRA=-DSSN_BLOCK_STAMPS, tools/block_stamps.py): shader-clock stamps (s_memtime) between the
// sections of a timestep of the f32 time loop, summed per wave and added up over all waves at the end of the launch:
//   [0] input assembly (LDS row read, v_readlane of the filter states)   [1] neuron groups (encode, LIF step, decode)
//   [2] wave reduction + publication of the wave sums in LDS              [3] wait at the workgroup barrier
//   [4] totals over the waves, filter update, hand-off to the post stage  [5] wave-timesteps counted
//   [6] / [7] s_memtime / s_memrealtime ticks of workgroup 0's time loop (shader clock = 100 MHz * [6] / [7])
__device__ unsigned long long g_block_stamps[8];
inline hipError_t read_block_stamps(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_stamps), sizeof(unsigned long long) * 8);
  if (e == hipSuccess && reset) {
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_block_stamps), z, sizeof z);
  }
  return e;
}
#define SSN_BSTAMP(var)                                                                          \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory");                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
#define SSN_BSTAMP_ADD(i, a, b) bst[i] += (b) - (a)
#else
#define SSN_BSTAMP(var) do {} while (0)
#define SSN_BSTAMP_ADD(i, a, b) do {} while (0)
#endif

// Neurons are dealt to threads in groups of PK adjacent neurons (PK = 2 for f32: one 64-bit register pair,
// 1 for f64): neuron index of (group g, thread tid, component c) = (g * nthr + tid) * PK + c - coalesced.
//
// Cross-lane sums (f32).  Per timestep a wave owes DOUT sums over its 64 lanes, and every wave then needs the DOUT
// totals over all waves for the filter update and the next input.  Both steps are kept off the scalar unit and off
// per-row instruction sequences:
//   * in the wave: four values at once - two v_permlane32_swap + add fold lanes 32..63 onto 0..31 for two value pairs,
//     one v_permlane16_swap + add leaves the four values in the four 16-lane rows, four DPP steps finish all four rows
//     together (gfx950's swap instructions; a fifth value takes the plain six-step DPP sum): 17 instead of 30 steps;
//   * across waves: wave w leaves value r in LDS slot r * 16 + w; after the one barrier of the timestep lane l of every
//     wave reads slot l, so a four-step DPP row sum gives row r of EVERY wave the total of value r in fixed order
//     (identical in all waves: deterministic).  The totals stay "lane-distributed" (row r of a register = decoded row
//     r): the Lowpass update of the recurrent filter states is ONE v_fma on that register with per-lane constants, and
//     only the (at most DIN) states the next input needs are broadcast with v_readlane.
template <typename T, int DIN, int DOUT, int NPT, int TPB, int LDSW, int SPLIT = 0>      // LDSW: 0 all parameters in registers | DIN: the encoder rows in LDS
__global__ __launch_bounds__(TPB) void k_ens_block(BlockArgs<T> a) {   // SPLIT: an ensemble over a.P member workgroups (own instantiation: the plain kernel's loop stays as it is)
  constexpr bool ENC_LDS = LDSW == DIN;
  // (Encoders AND bias in LDS - rows as long as the padded neuron count, 160 000 of the CU's 163 840 bytes at n = 10 000 -
  //  were built and measured in round 2: (768, 14) 3.68 ms and (1024, 10) 3.64 ms per 1000 timesteps of config 2 against
  //  3.49 for (512, 20) with the bias in registers: removed.)
  static_assert(LDSW == 0 || LDSW == DIN, "LDSW");
  extern __shared__ __align__(16) unsigned char ssn_block_dyn[];
  T* const e_lds = reinterpret_cast<T*>(ssn_block_dyn);       // ENC_LDS: encoder rows [DIN][cap]
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int PK = F32 ? 2 : 1;
  constexpr int NG = NPT / PK;
  static_assert(NPT % PK == 0, "f32 variants handle neuron pairs");
  using G = typename std::conditional<F32, f32x2, T>::type;   // one group of neurons
  constexpr int DP = DOUT <= 4 ? 4 : 8;
  // split ensembles: P consecutive workgroups are the members of ensemble k (f32, DOUT <= 4 only - the planner sees to it)
  static_assert(!SPLIT || (sizeof(T) == 4 && DOUT <= 4), "split ensembles: f32, at most four decoded rows");
  const int P = SPLIT ? a.P : 1;
  const int k = P > 1 ? (int)blockIdx.x / P : (int)blockIdx.x;
  const int member = P > 1 ? (int)blockIdx.x - k * P : 0;
  const int noff = member * a.n_member;                        // first neuron of this member's slice
  const int n_loc = P > 1 ? max(0, min(a.n_member, a.n_pad - noff)) : a.n_pad;
  const int tid = threadIdx.x;
  const int nthr = NG == 1 ? (int)blockDim.x : TPB;       // variants with more than one group always run full workgroups
  const int lane = tid & 63, wave = tid >> 6;
  const size_t row = (size_t)a.n_pad;
  const T* __restrict__ enc = a.enc + (size_t)k * DIN * row + noff;
  const T* __restrict__ bias = a.bias + (size_t)k * row + noff;
  T* __restrict__ Sp = a.S + (size_t)k * row + noff;
  const NeuronParams<T> np = a.np;
  const LifMath<T> lm(np);
  // (uniform values: readfirstlane moves them to scalar registers, where the packed instructions of the time loop can take
  //  them as their one constant-bus operand)
  auto uni = [](float v) { return bits_f(__builtin_amdgcn_readfirstlane(f_bits(v))); };
  LifConstV3 lc = lif_const_v3((double)np.dt, (double)np.tau_rc, (double)np.tau_ref);
  lc.na = uni(lc.na); lc.ca = uni(lc.ca); lc.m1 = uni(lc.m1); lc.c1 = uni(lc.c1); lc.c2 = uni(lc.c2); lc.c3 = uni(lc.c3); lc.p0 = uni(lc.p0); lc.p1 = uni(lc.p1); lc.p2 = uni(lc.p2); lc.ktau_ref = uni(lc.ktau_ref);
  lc.K = uni(lc.K);
  const f32x2 big = {0x1p100f, 0x1p100f};

  // ---- parameters and state of this thread's neurons -> registers ---------------------------------------------
  G e[ENC_LDS ? 1 : NG][DIN], b[NG], s[NG], dc[NG][DOUT];
  // LDS rows: as long as the variant's capacity (compile-time strides).
  constexpr int cap = (NG == 1 ? 1024 : TPB) * NPT;
  // one LDS base address per parameter row; a group's entry is a compile-time offset from it (the opaque asm keeps the
  // compiler from hoisting one address register per (group, row) out of the time loop: 28 registers at 7 groups)
  int lrow[LDSW > 0 ? LDSW : 1];           // (element offsets, not pointers: a pointer that went through the asm would be
#pragma unroll                             //  dereferenced with flat loads - 64-bit addresses, and the LDS aperture check)
  for (int d = 0; d < (LDSW > 0 ? LDSW : 1); ++d) {
    lrow[d] = d * cap + tid * PK;
    asm volatile("" : "+v"(lrow[d]));
  }
  auto lds_group = [&](int d, int g) -> const G* {
    return reinterpret_cast<const G*>(e_lds + lrow[d] + g * nthr * PK);
  };
  // Straight-line loads, no per-neuron branches (which cost the register allocator its view of the long live ranges):
  // rows are zero-padded to n_pad (a multiple of 4 elements) on upload, so a whole group is read whenever it starts
  // inside the padded row; groups past it read group 0 and are multiplied by 0.
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int i0 = (g * nthr + tid) * PK;
    const bool in = i0 < n_loc;
    const int ii = in ? i0 : 0;
    const T msk = in ? T(1) : T(0);
    auto ld = [&](const T* p) -> G { if constexpr (F32) return *reinterpret_cast<const f32x2*>(p) * msk; else return *p * msk; };
#pragma unroll
    for (int d = 0; d < DIN; ++d) {
      const G ev = ld(enc + d * row + ii);
      if constexpr (ENC_LDS) *reinterpret_cast<G*>(e_lds + d * cap + i0) = ev; else e[g][d] = ev;
    }
    b[g] = ld(bias + ii);
    s[g] = ld(Sp + ii);
    if constexpr (F32) {                                   // the time loop's forms: J - 1, distance to the threshold (lif_state_part / lif_input_part)
      b[g] = b[g] - 1.0f;
      s[g] = (f32x2){lif_word_in(s[g].x, lc.K), lif_word_in(s[g].y, lc.K)};
    }
    if (a.dec_neuron_major) {
      const T* dp = a.dec + ((size_t)k * row + noff + ii) * DP;
#pragma unroll
      for (int r = 0; r < DOUT; ++r) {
        if constexpr (F32) dc[g][r] = (f32x2){dp[r], dp[DP + r]} * msk; else dc[g][r] = dp[r] * msk;
      }
    } else {
      const T* dp = a.dec + (size_t)k * DOUT * row + noff + ii;
#pragma unroll
      for (int r = 0; r < DOUT; ++r) dc[g][r] = ld(dp + r * row);
    }
  }

  // ---- LDS: wave sums (two parities), and the block-buffer traffic of CH timesteps at a time so that no global
  //      memory operation sits inside the time loop (a workgroup barrier waits for every outstanding one) ----
  constexpr int CH = 32;                     // timesteps per input/output chunk (static LDS stays under 2 KB)
  constexpr int XP = 4;                      // words per timestep of inputs (DIN <= 4): one 16-byte LDS read
  constexpr int RW = DOUT <= 4 ? 64 : 128;   // wave-sum slots per parity: [r][16 waves]
  __shared__ __align__(16) T red[2][RW];
  __shared__ __align__(16) T xs[CH][XP];
  __shared__ T os[CH][DOUT];
  __shared__ int s_dst[DOUT];
  __shared__ int s_out[DOUT];
  for (int i = tid; i < 2 * RW; i += nthr) (&red[0][0])[i] = T(0);     // waves that do not exist add 0
  if (tid < DOUT) { s_dst[tid] = a.didx[(long long)k * DOUT + tid]; s_out[tid] = a.rowout[(long long)k * DOUT + tid] ? 1 : 0; }
  const T* __restrict__ xbase = a.xrows + (size_t)a.row0 * a.n_sig + a.x_off + (long long)k * DIN;

  // ---- per-row constants.  f32: lane-distributed - register 0 holds decoded row (lane >> 4), register 1 row 4 -----
  //      f64 (parity / small ensembles): uniform arrays, as the per-timestep kernel computes them
  T fs[F32 ? 1 : DOUT], la[F32 ? 1 : DOUT], lb[F32 ? 1 : DOUT], tot[F32 ? 1 : DOUT];
  float F0 = 0.0f, F1 = 0.0f, la0 = 0.0f, lb0 = 0.0f, la1 = 0.0f, lb1 = 0.0f, v0 = 0.0f, v1 = 0.0f;
  T xa[DIN];
  int xr[DIN];
#pragma unroll
  for (int d = 0; d < DIN; ++d) {
    xr[d] = a.xrow[(long long)k * DIN + d];
    xa[d] = xr[d] >= 0 ? a.xalpha[(long long)k * DIN + d] : T(0);
  }
  if constexpr (F32) {
    const int r0 = lane >> 4;
    if (r0 < DOUT) {
      const long long i = (long long)k * DOUT + r0;
      const int st = a.lp_state[i];
      if (st >= 0) { F0 = a.sig[st]; la0 = a.lp_a[i]; lb0 = a.lp_b[i]; }
    }
    if (DOUT > 4) {
      const long long i = (long long)k * DOUT + 4;
      const int st = a.lp_state[i];
      if (st >= 0) { F1 = a.sig[st]; la1 = a.lp_a[i]; lb1 = a.lp_b[i]; }
    }
  } else {
#pragma unroll
    for (int r = 0; r < DOUT; ++r) {
      const long long i = (long long)k * DOUT + r;
      const int st = a.lp_state[i];
      fs[r] = st >= 0 ? a.sig[st] : T(0);
      la[r] = st >= 0 ? a.lp_a[i] : T(0);
      lb[r] = st >= 0 ? a.lp_b[i] : T(0);
      tot[r] = T(0);
    }
  }
  // f32: where this lane's wave sums go (slot r * 16 + wave) and which filter states feed the inputs
  int wslot = -1;
  if constexpr (F32) {
    if constexpr (DOUT >= 4) {
      // after the swap reduction the rows of the 4-value register hold values 0, 2, 1, 3; value 4 sits in lane 63
      const int rv = (lane >> 4) == 1 ? 2 : ((lane >> 4) == 2 ? 1 : (lane >> 4));
      if ((lane & 15) == 0) wslot = rv * 16 + wave;
      if (DOUT == 5 && lane == 63) wslot = 64 + wave;
    }
  }

  constexpr int ILE = F32 ? (SSN_BLOCK_IL < NG ? SSN_BLOCK_IL : NG) : 1;      // groups stepped side by side (see the time loop)
  G en[ILE][LDSW > 0 ? LDSW : 1];            // ENC_LDS: encoders (and bias) of the groups about to be processed
  if constexpr (ENC_LDS) {
#pragma unroll
    for (int u = 0; u < ILE; ++u)
#pragma unroll
      for (int d = 0; d < LDSW; ++d) en[u][d] = *lds_group(d, u);
  }
#ifdef SSN_BLOCK_STAMPS
  unsigned long long bst[5] = {0, 0, 0, 0, 0}, bt0 = 0, bt1 = 0, bt2 = 0, bt3 = 0, bt4 = 0, bt5 = 0, bm0 = 0, br0 = 0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(bm0), "=s"(br0) :: "memory");
#endif
  unsigned int n_slots = 0, n_silent = 0;      // (wave-uniform: scalar registers)
  for (int j0 = 0; j0 < a.B; j0 += CH) {
    const int cn = min(CH, a.B - j0);
    __syncthreads();                         // previous chunk: every wave is past its last xs read / os write
    if (j0 > 0 && member == 0) {             // hand the previous chunk's decoded rows to the post stage
      for (int i = tid; i < CH * DOUT; i += nthr) {
        const int jj = i / DOUT, r = i - jj * DOUT;
        if (s_out[r]) a.bsig[(size_t)(a.row0 + j0 - CH + jj) * a.n_sig + s_dst[r]] = os[jj][r];
      }
    }
    for (int i = tid; i < cn * DIN; i += nthr) {
      const int jj = i / DIN, d = i - jj * DIN;
      xs[jj][d] = xbase[(size_t)(j0 + jj) * a.n_sig + d];
    }
    __syncthreads();

    for (int jj = 0; jj < cn; ++jj) {
      SSN_BSTAMP(bt0);
      if constexpr (SPLIT != 0) {
        if (P > 1 && wave == 0 && (lane & 15) == 0) {        // split ensemble: own words of the NEXT exchange buffer back to the sentinel
          unsigned int* const rst = a.xslots + (size_t)k * 64 + (size_t)((j0 + jj + 1) % 3) * ((size_t)a.K * 64) + member * 4 + (lane >> 4);
          const unsigned int sent = BLOCK_XCHG_SENTINEL;
          asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(rst), "v"(sent) : "memory");
        }
      }
      T xin[XP];
      if constexpr (sizeof(T) == 4) *(float4*)xin = *(const float4*)xs[jj];
      else { *(double2*)xin = *(const double2*)xs[jj]; *(double2*)(xin + 2) = *(const double2*)(xs[jj] + 2); }
      T x[DIN];
      if constexpr (F32) {
        // the filter states the inputs read: row xr[d] of the lane-distributed registers (uniform lane index)
        float f1 = 0.0f;
        if constexpr (DOUT > 4) f1 = bits_f(__builtin_amdgcn_readlane(f_bits(F1), 0));
#pragma unroll
        for (int d = 0; d < DIN; ++d) {
          const float f0 = bits_f(__builtin_amdgcn_readlane(f_bits(F0), (xr[d] & 3) << 4));
          float st = f0;                                            // (a v_cndmask costs five issue slots: only where a fifth row exists)
          if constexpr (DOUT > 4) st = xr[d] >= 4 ? f1 : f0;
          x[d] = __builtin_fmaf(xa[d], st, xin[d]);                 // (xa = 0 where no state feeds the input)
        }
      } else {
#pragma unroll
        for (int d = 0; d < DIN; ++d) {
          T st = T(0);
#pragma unroll
          for (int r = 0; r < DOUT; ++r) st = xr[d] == r ? fs[r] : st;
          x[d] = xr[d] >= 0 ? xin[d] + xa[d] * st : xin[d];
        }
      }
      SSN_BSTAMP(bt1);
      G accg[DOUT];
#pragma unroll
      for (int r = 0; r < DOUT; ++r) accg[r] = G(0);
      // ENC_LDS: a thread reads back only the entries it wrote itself (no barrier needed); the next group's
      // encoders are requested one group ahead so the LDS latency hides under this group's arithmetic
      // Groups are stepped IL at a time: each group's LIF step is one long dependent chain (~30 packed operations, a
      // reciprocal and a logarithm), and with two or three waves per SIMD the hardware alone does not cover its latency -
      // measured 6.8 cycles per vector instruction with one chain per wave against 4.4 issue-bound.  Beyond IL the chains
      // are kept apart (scheduling barrier): more interleaving only costs registers.
      constexpr int IL = ILE;
#pragma unroll
      for (int g0 = 0; g0 < NG; g0 += IL) {
        G J[IL], spk[IL];
#pragma unroll
        for (int u = 0; u < IL; ++u) {
          const int g = g0 + u < NG ? g0 + u : NG - 1;       // (a last, partial round: the surplus slots repeat its last group and are discarded)
          J[u] = b[g];
#pragma unroll
          for (int d = 0; d < DIN; ++d) {
            if constexpr (ENC_LDS) J[u] += en[u][d] * x[d];
            else J[u] += e[g][d] * x[d];
          }
        }
        if constexpr (ENC_LDS) {
          // the groups' LDS operands are consumed: request the next groups' into the same registers now - they arrive
          // under this round's neuron arithmetic (the last round requests round 0: the next timestep's first operands)
#pragma unroll
          for (int u = 0; u < IL; ++u) {
            if (g0 + u >= NG) continue;
            asm volatile("" : "+v"(J[u]));
            const int gn = g0 + IL + u < NG ? g0 + IL + u : (g0 + IL + u) % IL;
#pragma unroll
            for (int d = 0; d < LDSW; ++d) en[u][d] = *lds_group(d, gn);
          }
        }
#ifndef SSN_BLOCK_SKIP
#define SSN_BLOCK_SKIP 0      // (1 and 2 were measured slower than the plain loop in round 4: see the comments below and DESIGN.md section 3.1)
#endif
        // f32: a wave's neurons of this round are often ALL silent - the host deals neurons to (wave, round) slots by the part
        // of the oscillator's cycle in which they can fire (Sim::reorder_block_neurons), so that whole slots fall silent together -
        // and a silent slot needs neither the spike time (two v_rcp, two v_log and two packed operations per neuron pair) nor
        // the decode: 6 of a pair's 22 packed and all 4 of its transcendental instructions.  The branch is wave-uniform.
        // SSN_BLOCK_SKIP == 2: slots AT REST are left out altogether.  A neuron whose input current is <= 0 and whose state word is
        // exactly 1 (V = 0, not refractory) leaves the step as it entered it, bit for bit: W0 = 1, the integration time is a full
        // step, U = J * em - 1 < 0, no spike, Wn = clamp(1 - J * em) = 1, nmt = 0, w' = 1 - and adds +0 to every decoded sum.  The
        // test costs 8 packed / plain instructions per round: r = clamp(J) + (w - 1)^2 over the slot's neurons is 0 only then.
        bool at_rest = false;
        if constexpr (F32 && SSN_BLOCK_SKIP == 2) {
          ++n_slots;
          f32x2 r2 = {0.0f, 0.0f};
#pragma unroll
          for (int u = 0; u < IL; ++u) {
            if (g0 + u >= NG) continue;
            f32x2 pj;
            asm("v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0] clamp" : "=v"(pj) : "v"(J[u]));      // clamp(J): J[] holds J - 1
            const f32x2 q = s[g0 + u] - 1.0f;
            r2 += __builtin_elementwise_fma(q, q, pj);
          }
          at_rest = __builtin_amdgcn_ballot_w64(__builtin_bit_cast(unsigned long long, r2) != 0ull) == 0ull;
          if (at_rest) ++n_silent;
        }
        if constexpr (F32 && SSN_BLOCK_SKIP == 1) {
          ++n_slots;
          f32x2 Uu[IL], nm[IL];
          f32x2 any2 = {0.0f, 0.0f};
#pragma unroll
          for (int u = 0; u < IL; ++u) {
            const int g = g0 + u < NG ? g0 + u : NG - 1;
            f32x2 W0, em;
            lif_state_part(s[g], lc, W0, em);
            spk[u] = lif_spike_test(J[u], s[g], W0, em, lc, big, Uu[u], nm[u]);
            if (g0 + u < NG) any2 += spk[u];
          }
          if (__builtin_amdgcn_ballot_w64(any2.x + any2.y != 0.0f) != 0ull) {
#pragma unroll
            for (int u = 0; u < IL; ++u)
              if (g0 + u < NG) lif_finish_spiking(J[u], s[g0 + u], Uu[u], nm[u], spk[u], lc, big);
            // the decode belongs to this branch (spk is 0 or 1: exact adds) - the scheduler interleaves its independent FMAs
            // with the rcp -> log chains above; behind the branch the compiler turned a skipped decode into 80 v_cndmask
#pragma unroll
            for (int u = 0; u < IL; ++u) {
              if (g0 + u >= NG) continue;
#pragma unroll
              for (int r = 0; r < DOUT; ++r) accg[r] = __builtin_elementwise_fma(spk[u], dc[g0 + u][r], accg[r]);
            }
          } else {
#pragma unroll
            for (int u = 0; u < IL; ++u)
              if (g0 + u < NG) lif_finish_silent(s[g0 + u], Uu[u], nm[u]);
            ++n_silent;
          }
        }
        if (!at_rest) {
#pragma unroll
        for (int u = 0; u < IL; ++u) {
          const int g = g0 + u;
          if (g >= NG) continue;
          if constexpr (F32 && SSN_BLOCK_SKIP == 1) {
            // (stepped above)
          } else if constexpr (F32) {
            f32x2 W0, em;
            lif_state_part(s[g], lc, W0, em);
            spk[u] = lif_input_part(J[u], s[g], W0, em, lc, big);
          } else {
            // packed state word -> nengo's LIF step, operation for operation k_ensarray's fast path
            const T sw = s[g];
            T V = sw < T(0) ? T(0) : sw;
            T R = (sw < T(0) ? -sw : T(0)) - np.dt;
            T delta = np.dt - R;
            delta = delta < T(0) ? T(0) : (delta > np.dt ? np.dt : delta);
            V = V - (J[u] - V) * lm.decay(delta);
            spk[u] = T(0);
            if (V > T(1)) {
              const T t_spike = np.dt + np.tau_rc * lm.spike_time_term(V, J[u]);
              R = np.tau_ref + t_spike;
              V = T(0);
              spk[u] = T(1);
            } else if (V < T(0)) {
              V = T(0);
            }
            s[g] = R > np.dt ? -R : V;
          }
        }
#pragma unroll
        for (int u = 0; u < IL; ++u) {
          const int g = g0 + u;
          if (g >= NG) continue;
          // the state word is final HERE: without this the compiler sinks the rcp / log half of every group's step to the
          // end of the timestep and keeps its operands live until then
          if constexpr (F32) asm volatile("" : "+v"(s[g]));
          if constexpr (F32 && SSN_BLOCK_SKIP == 1) continue;      // (decoded inside the spiking branch above)
#pragma unroll
          for (int r = 0; r < DOUT; ++r) {           // spk is 0 or 1: exact add
            if constexpr (F32) accg[r] = __builtin_elementwise_fma(spk[u], dc[g][r], accg[r]);
            else accg[r] = fma(spk[u], dc[g][r], accg[r]);
          }
        }
        }
#ifndef SSN_BLOCK_NO_SCHED_BARRIER
        if constexpr (F32) __builtin_amdgcn_sched_barrier(0);
#endif
      }
      const int par = jj & 1;
      SSN_BSTAMP(bt2);
      if constexpr (F32) {
        float acc[DOUT];
#pragma unroll
        for (int r = 0; r < DOUT; ++r) acc[r] = accg[r].x + accg[r].y;
        if constexpr (DOUT >= 4) {
          const auto p01 = __builtin_amdgcn_permlane32_swap(f_bits(acc[0]), f_bits(acc[1]), false, false);
          const auto p23 = __builtin_amdgcn_permlane32_swap(f_bits(acc[2]), f_bits(acc[3]), false, false);
          const float s01 = bits_f(p01[0]) + bits_f(p01[1]);       // lanes 0..31: value 0, 32..63: value 1
          const float s23 = bits_f(p23[0]) + bits_f(p23[1]);
          const auto q = __builtin_amdgcn_permlane16_swap(f_bits(s01), f_bits(s23), false, false);
          float t4 = bits_f(q[0]) + bits_f(q[1]);                  // rows: values 0, 2, 1, 3
          t4 = row_sum_dpp(t4);
          if constexpr (DOUT == 5) {
            const float w4 = wave_sum_dpp(acc[4]);                 // lane 63
            if (wslot >= 0) red[par][wslot] = lane == 63 ? w4 : t4;
          } else {
            if (wslot >= 0) red[par][wslot] = t4;
          }
        } else {
          wave_sum_dpp_n<DOUT>(acc);
          if (lane == 63) {
#pragma unroll
            for (int r = 0; r < DOUT; ++r) red[par][r * 16 + wave] = acc[r];
          }
        }
        SSN_BSTAMP(bt3);
        __syncthreads();
        SSN_BSTAMP(bt4);
        v0 = row_sum_dpp(red[par][lane]);                          // row r of every wave: total of decoded row r
        if constexpr (SPLIT != 0) {
          if (P > 1) {
            // Split ensemble: the members publish their four sums and read their partners' (tools/xcd_exchange.hip measured
            // this protocol: +0.5 - 0.6 us per timestep wherever the members sit).  Three rotating buffers of sentinel
            // words: step t publishes into buffer t % 3 and resets the member's own words of buffer (t + 1) % 3, which
            // its partners read for the last time at step t - 2 (a partner that has published step t - 1 has consumed
            // everybody's step t - 2); the reset is acknowledged (s_waitcnt vmcnt(0)) before this step's publication.
            // Row r of every wave sums the members in lane order: identical totals, bit for bit, in every member.
            const int tq = j0 + jj;
            unsigned int* const xb = a.xslots + (size_t)k * 64;                      // [16 members][4 words] of one buffer
            const size_t bstride = (size_t)a.K * 64;
            if (wave == 0 && (lane & 15) == 0) {
              unsigned int* const pub = xb + (size_t)(tq % 3) * bstride + member * 4 + (lane >> 4);
              const unsigned int val = f_bits(v0);
              // (the reset of buffer (t + 1) % 3 was issued at the top of the timestep: its acknowledgement has arrived long ago)
              asm volatile("s_waitcnt vmcnt(0)\n\tglobal_store_dword %0, %1, off sc0 sc1" :: "v"(pub), "v"(val) : "memory");
            }
            const unsigned int* const src = xb + (size_t)(tq % 3) * bstride + (lane & 15) * 4 + (lane >> 4);
            const bool mine = (lane & 15) < P;
            unsigned int got = 0, spins = 0;
            while (true) {
              if (mine) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(src) : "memory");
              if (!__any(mine && got == BLOCK_XCHG_SENTINEL)) break;
              if (++spins > (1u << 20)) { if (lane == 0) *a.xerr = 1; break; }        // (a partner that never comes: no hang)
            }
            v0 = row_sum_dpp(mine ? bits_f(got) : 0.0f);
          }
        }
        F0 = __builtin_fmaf(la0, F0, lb0 * v0);                    // rows without a filter: la = lb = 0
        if constexpr (DOUT > 4) {
          v1 = row_sum_dpp(red[par][64 + lane]);
          F1 = __builtin_fmaf(la1, F1, lb1 * v1);
        }
        if (wave == 0 && member == 0) {
          if ((lane & 15) == 0 && (lane >> 4) < DOUT) os[jj][lane >> 4] = v0;
          if (DOUT > 4 && lane == 0) os[jj][4] = v1;
        }
        SSN_BSTAMP(bt5);
        SSN_BSTAMP_ADD(0, bt0, bt1); SSN_BSTAMP_ADD(1, bt1, bt2); SSN_BSTAMP_ADD(2, bt2, bt3); SSN_BSTAMP_ADD(3, bt3, bt4);
        SSN_BSTAMP_ADD(4, bt4, bt5);
      } else {
        T acc[DOUT];
#pragma unroll
        for (int r = 0; r < DOUT; ++r) acc[r] = accg[r];
        wave_sum_dpp_n<DOUT>(acc);
        if (lane == 63) {
#pragma unroll
          for (int r = 0; r < DOUT; ++r) red[par][r * 16 + wave] = acc[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < DOUT; ++r) {
          T t = T(0);
#pragma unroll
          for (int w = 0; w < 16; ++w) t += red[par][r * 16 + w];
          tot[r] = t;
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
  }
#ifdef SSN_BLOCK_STAMPS
  if constexpr (F32) {
    unsigned long long bm1, br1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(bm1), "=s"(br1) :: "memory");
    if (lane == 0) {
      for (int i = 0; i < 5; ++i) atomicAdd(&g_block_stamps[i], bst[i]);
      atomicAdd(&g_block_stamps[5], (unsigned long long)a.B);
      if (k == 0 && wave == 0) { atomicAdd(&g_block_stamps[6], bm1 - bm0); atomicAdd(&g_block_stamps[7], br1 - br0); }
    }
  }
#endif
  if constexpr (F32 && SSN_BLOCK_SKIP != 0) {
    if (lane == 0 && a.slot_stats) {
      atomicAdd(a.slot_stats, (unsigned long long)n_slots);
      atomicAdd(a.slot_stats + 1, (unsigned long long)n_silent);
    }
  }
  __syncthreads();
  if (member == 0) {                         // last chunk's decoded rows
    const int j0 = (a.B - 1) / CH * CH, cn = a.B - j0;
    for (int i = tid; i < cn * DOUT; i += nthr) {
      const int jj = i / DOUT, r = i - jj * DOUT;
      if (s_out[r]) a.bsig[(size_t)(a.row0 + j0 + jj) * a.n_sig + s_dst[r]] = os[jj][r];
    }
  }

  // ---- state back to HBM; decoded values and filter states of the last step to the signal vector ------------
  int tid2 = tid;
  asm volatile("" : "+v"(tid2));      // fresh addresses: keeps NPT pointers from staying live across the time loop
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int i0 = (g * nthr + tid2) * PK;
    if constexpr (F32) s[g] = (f32x2){lif_word_out(s[g].x, lc.K), lif_word_out(s[g].y, lc.K)};
    if (i0 < n_loc) *reinterpret_cast<G*>(Sp + i0) = s[g];         // (padding elements of the row are never read back)
  }
  if (a.B > 0 && member == 0) {
    if constexpr (F32) {
      if (wave == 0 && (lane & 15) == 0 && (lane >> 4) < DOUT) {
        const int r = lane >> 4;
        a.sig_w[s_dst[r]] = v0;
        const int st = a.lp_state[(long long)k * DOUT + r];
        if (st >= 0) a.sig_w[st] = F0;
      }
      if (DOUT > 4 && tid == 0) {
        a.sig_w[s_dst[4]] = v1;
        const int st = a.lp_state[(long long)k * DOUT + 4];
        if (st >= 0) a.sig_w[st] = F1;
      }
    } else if (tid < DOUT) {
      T v = T(0), f = T(0);
#pragma unroll
      for (int r = 0; r < DOUT; ++r) { v = tid == r ? tot[r] : v; f = tid == r ? fs[r] : f; }
      a.sig_w[s_dst[tid]] = v;
      const int st = a.lp_state[(long long)k * DOUT + tid];
      if (st >= 0) a.sig_w[st] = f;
    }
  }
}

constexpr int BLOCK_LDS_BYTES = 160 * 1024, BLOCK_STATIC_LDS = 2560;     // CU capacity; bound on the kernel's static arrays

template <typename T, int DIN, int DOUT, int NPT, int TPB, int LDSW, int SPLIT>
static hipError_t launch_block_variant_s(hipStream_t s, const BlockArgs<T>& a) {
  const int lds = LDSW * a.threads * NPT * (int)sizeof(T);
  static std::atomic<uint64_t> configured{0};
  if (LDSW > 0) {
    hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(&k_ens_block<T, DIN, DOUT, NPT, TPB, LDSW, SPLIT>),
                                            BLOCK_LDS_BYTES - BLOCK_STATIC_LDS, configured);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((k_ens_block<T, DIN, DOUT, NPT, TPB, LDSW, SPLIT>), dim3((unsigned)(a.K * (SPLIT ? a.P : 1))), dim3((unsigned)a.threads), lds, s, a);
  return hipGetLastError();
}
template <typename T, int DIN, int DOUT, int NPT, int TPB, int LDSW>
static hipError_t launch_block_variant(hipStream_t s, const BlockArgs<T>& a) {
  if (a.P > 1) {
    if constexpr (sizeof(T) == 4 && DOUT <= 4) return launch_block_variant_s<T, DIN, DOUT, NPT, TPB, LDSW, 1>(s, a);
    else return hipErrorInvalidValue;
  }
  return launch_block_variant_s<T, DIN, DOUT, NPT, TPB, LDSW, 0>(s, a);
}

// (workgroup size, neurons per thread, parameter rows in LDS) variants, in the order the planner tries them.  A wave's
// register budget is 512 / (waves per SIMD): 1024 threads -> 128, 768 -> 168, 512 -> 256; a neuron keeps
// din + dout + 2 words (9 at din 3, dout 4) of which the LDS rows take 3.  Measured on MI355X at n = 10 000
// (tools/bench_block.py, profiles/round2_block_variants.txt): (512, 20, LDS) 3.49 ms per 1000 timesteps of config 2,
// (768, 14, LDS) 3.68 - more waves per SIMD do not pay: every wave repeats the per-timestep reductions.
struct BlockVariant { int tpb, npt, ldsw; };
template <typename T> struct BlockVariants;
template <> struct BlockVariants<float> {
  static constexpr int N = 7;
  static constexpr BlockVariant v[N] = {{1024, 2, 0}, {1024, 4, 0}, {512, 6, 0}, {1024, 6, 0}, {512, 10, 0}, {512, 20, 3}, {768, 14, 3}};
};
template <> struct BlockVariants<double> {
  static constexpr int N = 3;
  static constexpr BlockVariant v[N] = {{1024, 1, 0}, {1024, 2, 0}, {1024, 4, 0}};
};

template <typename T, int DIN, int DOUT>
static hipError_t launch_block_npt(hipStream_t s, const BlockArgs<T>& a) {
  const int key = (a.tpb * 100 + a.npt) * 8 + a.enc_lds;
  switch (key) {
#define SSN_CASE(TPB, N, L) case (TPB * 100 + N) * 8 + L: return launch_block_variant<T, DIN, DOUT, N, TPB, L>(s, a);
    SSN_CASE(1024, 2, 0) SSN_CASE(1024, 4, 0)
#undef SSN_CASE
#define SSN_CASE(TPB, N, L) case (TPB * 100 + N) * 8 + L: if constexpr (sizeof(T) == 8) return launch_block_variant<T, DIN, DOUT, N, TPB, L>(s, a); else return hipErrorInvalidValue;
    SSN_CASE(1024, 1, 0)
#undef SSN_CASE
#define SSN_CASE(TPB, N, L) case (TPB * 100 + N) * 8 + L: if constexpr (sizeof(T) == 4) return launch_block_variant<T, DIN, DOUT, N, TPB, L>(s, a); else return hipErrorInvalidValue;
    SSN_CASE(1024, 6, 0) SSN_CASE(512, 6, 0) SSN_CASE(512, 10, 0) SSN_CASE(512, 20, 3) SSN_CASE(768, 14, 3)
#undef SSN_CASE
    default: return hipErrorInvalidValue;
  }
}

template <typename T>
bool ens_block_supported(int din, int dout, int n, int* threads, int* tpb, int* npt, int* enc_lds) {
  if (!(din == 3 && dout >= 3 && dout <= 5)) return false;
  int want_tpb = 0, want_npt = 0, want_lds = 0;
  if (const char* env = getenv("SSN_BLOCK_VARIANT")) sscanf(env, "%d,%d,%d", &want_tpb, &want_npt, &want_lds);   // tuning knob
  const int pk = sizeof(T) == 4 ? 2 : 1;         // neurons per group; single-group variants run n / pk threads
  for (int i = 0; i < BlockVariants<T>::N; ++i) {
    const BlockVariant v = BlockVariants<T>::v[i];
    if (want_tpb && (v.tpb != want_tpb || v.npt != want_npt || v.ldsw != want_lds)) continue;
    int th = v.tpb;
    if (v.npt == pk && n < th * pk) th = std::max(64, ((n + pk - 1) / pk + 63) / 64 * 64);
    if ((int64_t)th * v.npt < n) continue;
    if ((int64_t)v.ldsw * th * v.npt * (int64_t)sizeof(T) > BLOCK_LDS_BYTES - BLOCK_STATIC_LDS) continue;
    *threads = th; *tpb = v.tpb; *npt = v.npt; *enc_lds = v.ldsw;
    return true;
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
