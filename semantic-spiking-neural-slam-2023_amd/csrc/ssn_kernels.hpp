// ssn_kernels.hpp - gfx950 (MI355X, CDNA4) kernels of the SSP-SLAM step loop, templated on the
// arithmetic type T (float = fast mode, double = parity mode).  Included by ssn_f32.hip / ssn_f64.hip,
// which explicitly instantiate the launchers declared in ssn_launch.hpp.
//
// Kernel             replaces (merged nengo operators)                       bound
// k_ensarray<T,..>   Reset+DotInc(enc)+SimNeurons(LIF)+DotInc(dec) of an EnsembleArray    HBM stream
// k_program<T>       every small vector op of a step (fill/table/axpy/lowpass/...)        latency
// k_matvec<T>        dense DotInc (one 64-lane wave per output row)                       HBM / L2
// k_neurons<T>       SimNeurons on a current vector                                       HBM
// k_pes<T>           SimPES + weight increment (fused outer-product update)               HBM
// k_voja<T>          SimVoja + encoder increment (rows of spiking neurons only)           spike-sparse
//
// All kernels are wave64: reductions use 64-lane shuffles, block sizes are multiples of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "ssn_launch.hpp"

namespace ssn {

template <typename T> struct VecT;
template <> struct VecT<float>  { using type = float4;  static constexpr int W = 4; };
template <> struct VecT<double> { using type = double2; static constexpr int W = 2; };

__device__ inline float  expm1_(float x)  { return expm1f(x); }
__device__ inline double expm1_(double x) { return expm1(x); }
__device__ inline float  log1p_(float x)  { return log1pf(x); }
__device__ inline double log1p_(double x) { return log1p(x); }

// LIF arithmetic of the FAST kernel variant.  double: the libm calls, operation for operation like the
// oracle.  float: the same formulas with cheap evaluations that are exact to f32 rounding on the
// argument ranges that occur - expm1(x) for x in [-dt/tau_rc, 0] by its Taylor polynomial (|x| <= ~0.05:
// the first neglected term is < 2e-13 relative), log1p(-u) for u in (0, 1) via the hardware log, and
// reciprocal-multiply divisions.  This cuts the per-neuron VALU work ~4x (the kernel was co-limited by it).
template <typename T> struct LifMath;
template <> struct LifMath<double> {
  const double tau_rc;
  __device__ explicit LifMath(const NeuronParams<double>& p) : tau_rc(p.tau_rc) {}
  __device__ double decay(double delta) const { return expm1(-delta / tau_rc); }
  __device__ double spike_time_term(double V, double J) const { return log1p(-(V - 1.0) / (J - 1.0)); }
};
template <> struct LifMath<float> {
  const float neg_inv_tau;
  __device__ explicit LifMath(const NeuronParams<float>& p) : neg_inv_tau(-1.0f / p.tau_rc) {}
  __device__ float decay(float delta) const {
    const float x = delta * neg_inv_tau;
    if (x < -0.125f) return expm1f(x);          // dt/tau_rc larger than the polynomial's range: library call
    float q = 1.0f / 720.0f;
    q = fmaf(q, x, 1.0f / 120.0f);
    q = fmaf(q, x, 1.0f / 24.0f);
    q = fmaf(q, x, 1.0f / 6.0f);
    q = fmaf(q, x, 0.5f);
    q = fmaf(q, x, 1.0f);
    return q * x;
  }
  __device__ float spike_time_term(float V, float J) const {
    const float u = (V - 1.0f) * __frcp_rn(J - 1.0f);
    return __logf(1.0f - u);
  }
};

// ---------------------------------------------------------------------------------------------
// neuron models.  Returns the unit-amplitude activity: spike 0/1 or rate.
// Operation order follows the oracle (oracle/stepper.py lif_step) so that the f64 build, compiled
// with -ffp-contract=off, tracks it to rounding.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ inline T neuron_step(const NeuronParams<T>& p, T J, T& V, T& R) {
  if (p.type == 0) {                                   // LIF
    R = R - p.dt;
    T delta = p.dt - R;
    delta = delta < T(0) ? T(0) : (delta > p.dt ? p.dt : delta);
    V = V - (J - V) * expm1_(-delta / p.tau_rc);
    if (V > T(1)) {
      T t_spike = p.dt + p.tau_rc * log1p_(-(V - T(1)) / (J - T(1)));
      V = T(0);
      R = p.tau_ref + t_spike;
      return T(1);
    }
    if (V < p.min_voltage) V = p.min_voltage;
    return T(0);
  } else if (p.type == 1) {                            // LIFRate
    T j = J - T(1);
    return j > T(0) ? T(1) / (p.tau_ref + p.tau_rc * log1p_(T(1) / j)) : T(0);
  }
  return J > T(0) ? J : T(0);                          // ReLU
}

template <typename T>
__device__ inline T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// k_ensarray: K equal ensembles of n neurons, din-dimensional, dout decoded rows each.
//   grid = K * P workgroups of 256 threads; workgroup (k, p) streams a contiguous chunk of
//   ensemble k's neurons once.  Every streamed access is a 16-byte-per-lane coalesced vector; spikes
//   never leave registers.  Each workgroup leaves dout partial sums; the P partials of an ensemble are
//   added in fixed order afterwards (deterministic, no atomics).
//
//   Two variants (template FAST):
//   * generic (any neuron type): enc[din] + bias + V + R in, V + R out, dec[dout] in:
//     (din + dout + 5) words per neuron-step (the 52 B of SURVEY 8d at din 3, dout 5, f32).
//   * FAST (spiking LIF with min_voltage = 0) moves fewer bytes for the same result, bit for bit:
//       - voltage and refractory time share ONE state word s: s >= 0 is the voltage of a
//         non-refractory neuron, s < 0 is minus the remaining refractory time (voltage is exactly 0
//         then).  A refractory time <= dt is not stored: the next step integrates for the full dt
//         whatever its value.  (-2 words)
//       - decoders are stored neuron-major ([n][DP], DP = 4 or 8 words) and fetched only for the
//         neurons that spiked this step (~10 % at SLAM rates): ~64 B per spike instead of dout words
//         per neuron.
// Layout: enc [K][DIN][n_pad], bias/V/R [K][n_pad], dec [K][DOUT][n_pad] (generic) or [K][n_pad][DP]
// (FAST with DOUT >= 3), n_pad % W == 0.
// ---------------------------------------------------------------------------------------------
template <int DOUT> struct DecPitch { static constexpr int value = DOUT <= 4 ? 4 : 8; };

template <typename T, int DIN, int DOUT, int MODE, bool RND = false>   // MODE 0 generic | 1 fast, spike-sparse decoders | 2 fast, dense decoders; RND: a body of the round grid
__device__ __forceinline__ void ens_body(const EnsArgs<T>& a, const int bx, unsigned char* smem) {   // smem: 4 * DOUT + DIN values of T
  if (bx >= a.K * a.P) return;          // (grid.x is sized for the larger array of a batch)
  using vec = typename VecT<T>::type;
  constexpr int W = VecT<T>::W;
  constexpr bool FAST = MODE != 0;
  constexpr bool SPARSE = MODE == 1;
  constexpr int DP = DecPitch<DOUT>::value;
  const int k = bx / a.P;
  const int p = bx - k * a.P;
  const size_t row = (size_t)a.n_pad;
  const T* __restrict__ enc = a.enc + (size_t)k * DIN * row;
  const T* __restrict__ bias = a.bias + (size_t)k * row;
  const T* __restrict__ dec = a.dec + (size_t)k * (SPARSE ? DP : DOUT) * row;
  T* __restrict__ Vp = a.V + (size_t)k * row;
  T* __restrict__ Rp = a.R + (size_t)k * row;

  const int n_vec = a.n_pad / W;
  const int v_begin = p * a.chunk_vec;
  const int v_end = min(n_vec, v_begin + a.chunk_vec);
  const NeuronParams<T> np = a.np;
  const LifMath<T> lm(np);

  // The streaming loads of the first sweep are issued before the (dependent, scalar) input assembly so
  // that its latency chain - step counter -> row address -> x - hides under them.
  T e[DIN][W], b[W], Vv[W], Rv[W];
  int v = v_begin + (int)threadIdx.x;
  auto load_sweep = [&](int vv) {
    const size_t o = (size_t)vv * W;
#pragma unroll
    for (int d = 0; d < DIN; ++d) *(vec*)e[d] = *(const vec*)(enc + d * row + o);
    *(vec*)b = *(const vec*)(bias + o);
    *(vec*)Vv = *(const vec*)(Vp + o);
    if constexpr (!FAST) *(vec*)Rv = *(const vec*)(Rp + o);
  };
  if (v < v_end) load_sweep(v);

  T x[DIN];
  long long step = 0;
  if (a.xrows || a.defer) step = a.ctx->step + a.sub;
  // (the self-finishing form exists in the round grid only: in the stand-alone kernel - config 4's - its branch cost 1 %)
  if (RND && a.defer == 2) {
    // Round plan (round 4): the array completes ITS OWN previous timestep.  Every workgroup of ensemble k sums the P partial
    // sums that the ensemble's workgroups left at timestep s - 1 for the rows that feed its inputs, advances the recurrent
    // filter states (identical values in every workgroup; workgroup p = 0 keeps them: fstate ping-pongs by timestep parity)
    // and adds them to the rest of the input, which a micro-operator assembled in the signal vector.  The four rounds
    // array -> finish -> Lowpass -> input -> array of every oscillator become one.
    T* s_x = reinterpret_cast<T*>(smem) + 4 * DOUT;      // [DIN]
    const int par = (int)(step & 1);
    const long long NR = (long long)a.K * DOUT;
    const T* prev = a.partials + (size_t)(par ^ 1) * a.partials_stride + (size_t)k * a.P * DOUT;
    const int tid = threadIdx.x;
    if (tid < DIN) {
      const long long e = (long long)k * DIN + tid;
      T xv = a.sig[a.x_off + e];
      const int xr = a.xrow[e];
      if (xr >= 0) {
        const long long i = (long long)k * DOUT + xr;
        T v = T(0);
        for (int q = 0; q < a.P; ++q) v += prev[(size_t)q * DOUT + xr];
        const T st = a.lp_a[i] * a.fstate[(size_t)(par ^ 1) * NR + i] + a.lp_b[i] * v;
        if (p == 0) a.fstate[(size_t)par * NR + i] = st;
        xv += a.xalpha[e] * st;
      }
      s_x[tid] = xv;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DIN; ++d) x[d] = s_x[d];
  } else if (!a.defer) {
    const T* xsrc = a.sig;
    if (a.xrows) xsrc = a.xrows + (size_t)(step - a.ctx->block_start + 1) * a.n_sig;
#pragma unroll
    for (int d = 0; d < DIN; ++d) {
      const long long xi = a.x_off + (long long)k * DIN + d;
      T xv = xsrc[xi];
      for (int j = 0; j < a.n_rec; ++j)
        if (xi >= a.rec_dst[j] && xi < a.rec_dst[j] + a.rec_len[j]) xv += a.rec_alpha[j] * a.sig[a.rec_src[j] + (xi - a.rec_dst[j])];
      x[d] = xv;
    }
  } else {
    // Deferred finish.  A handful of lanes do the scalar work and broadcast through LDS:
    //  * lane d < DIN: this step's input x[d] = pre-stage row + alpha * (filter state after step-1), where
    //    that state is completed here from step-1's partial sums if it is still pending;
    //  * workgroup p == 0, lanes 64 + r (r < DOUT): publish step-1's decoded row r (signal, filter state,
    //    hand-off to the post stage).  Every workgroup of an ensemble derives identical values from
    //    identical inputs, so nothing else needs to wait for the publisher.
    T* s_x = reinterpret_cast<T*>(smem) + 4 * DOUT;      // [DIN]
    const int par = (int)(step & 1);
    const long long NR = (long long)a.K * DOUT;
    const bool pending = (step - 1) > a.ctx->finished;
    const T* prev = a.partials + (size_t)(par ^ 1) * a.partials_stride + (size_t)k * a.P * DOUT;
    const int tid = threadIdx.x;
    if (tid < DIN) {
      const long long e = (long long)k * DIN + tid;
      const T* xsrc = a.xrows + (size_t)(step - a.ctx->block_start + 1) * a.n_sig;
      T xv = xsrc[a.x_off + e];
      const int xr = a.xrow[e];
      if (xr >= 0) {
        const long long i = (long long)k * DOUT + xr;
        T st;
        if (pending) {
          T v = T(0);
          for (int q = 0; q < a.P; ++q) v += prev[(size_t)q * DOUT + xr];
          st = a.lp_a[i] * a.fstate[(size_t)par * NR + i] + a.lp_b[i] * v;
        } else {
          st = a.fstate[(size_t)(par ^ 1) * NR + i];
        }
        xv += a.xalpha[e] * st;
      }
      s_x[tid] = xv;
    } else if (p == 0 && pending && tid >= 64 && tid < 64 + DOUT) {
      const int r = tid - 64;
      const long long i = (long long)k * DOUT + r;
      T v = T(0);
      for (int q = 0; q < a.P; ++q) v += prev[(size_t)q * DOUT + r];
      const int dst = a.didx[i];
      a.sig_w[dst] = v;
      if (a.lp_has[i]) a.fstate[(size_t)(par ^ 1) * NR + i] = a.lp_a[i] * a.fstate[(size_t)par * NR + i] + a.lp_b[i] * v;
      if (a.rowout[i]) a.bsig[(size_t)(step - a.ctx->block_start) * a.n_sig + dst] = v;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DIN; ++d) x[d] = s_x[d];
  }

  T acc[DOUT];
#pragma unroll
  for (int r = 0; r < DOUT; ++r) acc[r] = T(0);

  for (; v < v_end; v += 256) {
    const size_t o = (size_t)v * W;
    if constexpr (!FAST) {
      T dd[DOUT][W];
#pragma unroll
      for (int r = 0; r < DOUT; ++r) *(vec*)dd[r] = *(const vec*)(dec + r * row + o);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if ((int)o + j < a.n) {
          T J = b[j];
#pragma unroll
          for (int d = 0; d < DIN; ++d) J += e[d][j] * x[d];
          const T act = neuron_step(np, J, Vv[j], Rv[j]);
#pragma unroll
          for (int r = 0; r < DOUT; ++r) acc[r] += act * dd[r][j];
        }
      }
      *(vec*)(Vp + o) = *(vec*)Vv;
      *(vec*)(Rp + o) = *(vec*)Rv;
    } else {
      bool spiked[W];
#pragma unroll
      for (int j = 0; j < W; ++j) {
        spiked[j] = false;
        if ((int)o + j < a.n) {
          T J = b[j];
#pragma unroll
          for (int d = 0; d < DIN; ++d) J += e[d][j] * x[d];
          // unpack the state word, then nengo's LIF step operation for operation
          const T s = Vv[j];
          T V = s < T(0) ? T(0) : s;
          T R = (s < T(0) ? -s : T(0)) - np.dt;
          T delta = np.dt - R;
          delta = delta < T(0) ? T(0) : (delta > np.dt ? np.dt : delta);
          V = V - (J - V) * lm.decay(delta);
          if (V > T(1)) {
            const T t_spike = np.dt + np.tau_rc * lm.spike_time_term(V, J);
            R = np.tau_ref + t_spike;
            V = T(0);
            spiked[j] = true;
          } else if (V < T(0)) {
            V = T(0);
          }
          Vv[j] = R > np.dt ? -R : V;       // voltage is exactly 0 whenever R > dt
        }
      }
      *(vec*)(Vp + o) = *(vec*)Vv;
      if (v + 256 < v_end) load_sweep(v + 256);     // next sweep streams in under the decoder gather
      if constexpr (SPARSE) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
          if (spiked[j]) {
            T dd[DP];
            const T* dp = dec + (o + j) * DP;
#pragma unroll
            for (int q = 0; q < DP; q += W) *(vec*)(dd + q) = *(const vec*)(dp + q);
#pragma unroll
            for (int r = 0; r < DOUT; ++r) acc[r] += dd[r];
          }
        }
      } else {
        T dd[DOUT][W];
#pragma unroll
        for (int r = 0; r < DOUT; ++r) *(vec*)dd[r] = *(const vec*)(dec + r * row + o);
#pragma unroll
        for (int j = 0; j < W; ++j)
          if (spiked[j]) {
#pragma unroll
            for (int r = 0; r < DOUT; ++r) acc[r] += dd[r][j];
          }
      }
    }
    if constexpr (!FAST) { if (v + 256 < v_end) load_sweep(v + 256); }
  }

  T (*red)[DOUT] = reinterpret_cast<T (*)[DOUT]>(smem);      // [4][DOUT]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < DOUT; ++r) {
    const T s = wave_sum(acc[r]);
    if (lane == 0) red[wave][r] = s;
  }
  __syncthreads();
  if (threadIdx.x < DOUT) {
    const int r = threadIdx.x;
    const T total = (red[0][r] + red[1][r]) + (red[2][r] + red[3][r]);
    if (a.direct) {                      // the workgroup saw the whole ensemble: this IS the decoded value
      a.sig_w[a.didx[(size_t)k * DOUT + r]] = total;
    } else {
      T* out = a.partials + (a.defer ? (size_t)(step & 1) * a.partials_stride : (size_t)0);
      out[((size_t)k * a.P + p) * DOUT + r] = total;
    }
  }
}
// Arrays of MANY SMALL one-dimensional ensembles - the product ensembles of a circular convolution (reference binding.py:297-317:
// 2 x 2032 ensembles of 50 neurons per network at d = 1015) - as a body of the round grid.  ens_body gives every ensemble a
// 256-thread workgroup of which 13 threads have neurons (four each): 8 128 workgroups per SLAM timestep, each a chain of
// dependent loads, a wave reduction, an LDS round trip and a barrier for 50 neurons.  Here a WAVE owns an ensemble (lane =
// neuron, one coalesced 200-byte row per parameter), takes four ensembles with all their loads in flight before the first
// neuron step, reduces with six shuffles and writes the decoded value itself: no LDS, no barrier, 16 ensembles per
// workgroup (508 workgroups per timestep instead of 8 128).  LIF fast path (packed state word), dense decoders, n <= 64.
template <typename T>
__device__ __forceinline__ void ens_small_body(const EnsArgs<T>& a, const int bx) {
  constexpr int EPW = ENS_SMALL_PER_WAVE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = (bx * 4 + wave) * EPW;
  const NeuronParams<T> np = a.np;
  const LifMath<T> lm(np);
  const size_t row = (size_t)a.n_pad;
  T e[EPW], b[EPW], s[EPW], d[EPW], x[EPW];
  bool live[EPW];
#pragma unroll
  for (int u = 0; u < EPW; ++u) {
    const int k = k0 + u;
    live[u] = k < a.K && lane < a.n;
    const size_t o = (size_t)(k < a.K ? k : 0) * row + (size_t)(lane < a.n ? lane : 0);
    e[u] = a.enc[o]; b[u] = a.bias[o]; s[u] = a.V[o]; d[u] = a.dec[o];
    const long long xi = a.x_off + (long long)(k < a.K ? k : 0);
    T xv = a.sig[xi];
    for (int j = 0; j < a.n_rec; ++j)
      if (xi >= a.rec_dst[j] && xi < a.rec_dst[j] + a.rec_len[j]) xv += a.rec_alpha[j] * a.sig[a.rec_src[j] + (xi - a.rec_dst[j])];
    x[u] = xv;
  }
#pragma unroll
  for (int u = 0; u < EPW; ++u) {
    const int k = k0 + u;
    T act = T(0);
    if (live[u]) {
      // packed state word -> nengo's LIF step, operation for operation ens_body's fast path
      const T J = b[u] + e[u] * x[u];
      const T sw = s[u];
      T V = sw < T(0) ? T(0) : sw;
      T R = (sw < T(0) ? -sw : T(0)) - np.dt;
      T delta = np.dt - R;
      delta = delta < T(0) ? T(0) : (delta > np.dt ? np.dt : delta);
      V = V - (J - V) * lm.decay(delta);
      if (V > T(1)) {
        const T t_spike = np.dt + np.tau_rc * lm.spike_time_term(V, J);
        R = np.tau_ref + t_spike;
        V = T(0);
        act = d[u];
      } else if (V < T(0)) {
        V = T(0);
      }
      a.V[(size_t)k * row + lane] = R > np.dt ? -R : V;
    }
    const T total = wave_sum(act);
    if (lane == 0 && k < a.K) a.sig_w[a.didx[k]] = total;
  }
}

template <typename T, int DIN, int DOUT, int MODE>
__global__ __launch_bounds__(256) void k_ensarray(EnsBatch<T> batch) {
  __shared__ __align__(16) unsigned char smem[(4 * DOUT + DIN) * sizeof(T)];
  ens_body<T, DIN, DOUT, MODE>(batch.a[blockIdx.y], (int)blockIdx.x, smem);
}

// decoder re-layout [K][dout][n] (row-major, host order, ld = n_pad) <-> [K][n_pad][DP] (neuron-major)
template <typename T>
__global__ void k_dec_pack(const T* __restrict__ src, T* __restrict__ dst, int K, int dout, int n, int n_pad, int DP, int unpack) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)K * n_pad * DP) return;
  const int q = (int)(i % DP);
  const int64_t t = i / DP;
  const int nn = (int)(t % n_pad), k = (int)(t / n_pad);
  if (!unpack) dst[i] = (q < dout && nn < n) ? src[((size_t)k * dout + q) * n_pad + nn] : T(0);
  else if (q < dout && nn < n) dst[((size_t)k * dout + q) * n_pad + nn] = src[i];
}
template <typename T>
hipError_t launch_dec_pack(hipStream_t s, const T* src, T* dst, int K, int dout, int n, int n_pad, int DP, int unpack) {
  const int64_t total = (int64_t)K * n_pad * DP;
  hipLaunchKernelGGL((k_dec_pack<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, K, dout, n, n_pad, DP, unpack);
  return hipGetLastError();
}

// packed LIF state word -> (voltage, refractory time) for ssn_read_buffer; r_offset: what a negative word leaves out of
// the refractory time (0, or dt for the f32 whole-block kernel's words)
template <typename T>
__global__ void k_state_unpack(const T* __restrict__ s, T* __restrict__ out, int64_t n, int want_refractory, T r_offset) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const T v = s[i];
  out[i] = want_refractory ? (v < T(0) ? r_offset - v : T(0)) : (v < T(0) ? T(0) : v);
}
template <typename T>
hipError_t launch_state_unpack(hipStream_t s, const T* src, T* out, int64_t n, int want_refractory, T r_offset) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((k_state_unpack<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, out, n, want_refractory, r_offset);
  return hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void k_ens_finish(FinishArgs<T> f) {
  const long long step = f.ctx->step;           // mode 0: step being finished; modes 1, 2: first step not yet run
  const int i = blockIdx.x * 256 + threadIdx.x;
  const long long NR = (long long)f.K * f.dout;
  if (i < f.K * f.dout) {
    const int st = f.lp_state[i];
    if (f.mode == 2) {                           // begin: filter states of step-1 from the signal vector
      f.fstate[(size_t)((step - 1) & 1) * NR + i] = st >= 0 ? f.sig[st] : T(0);
    } else {
      const long long t = f.mode == 1 ? step - 1 : step;        // the step whose results are completed here
      const int k = i / f.dout, r = i - k * f.dout;
      const T* part = f.partials + (f.mode == 1 ? (size_t)(t & 1) * f.partials_stride : (size_t)0);
      T s = T(0);
      for (int p = 0; p < f.P; ++p) s += part[((size_t)k * f.P + p) * f.dout + r];
      const int dst = f.didx[i];
      f.sig[dst] = s;
      if (st >= 0) {
        if (f.mode == 1) f.sig[st] = f.lp_a[i] * f.fstate[(size_t)((t - 1) & 1) * NR + i] + f.lp_b[i] * s;
        else f.sig[st] = f.lp_a[i] * f.sig[st] + f.lp_b[i] * s;
      }
      if (f.rowout[i]) f.bsig[(size_t)(t - f.ctx->block_start + 1) * f.n_sig + dst] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int old = atomicAdd(f.ticket, 1u);
    if (old % (unsigned)f.n_blocks == (unsigned)f.n_blocks - 1u) {       // every block has read `step`
      if (f.mode == 0) f.ctx->step = step + 1;
      else f.ctx->finished = step - 1;
    }
  }
}

template <typename T>
hipError_t launch_ens_finish(hipStream_t s, const FinishArgs<T>& f) {
  hipLaunchKernelGGL((k_ens_finish<T>), dim3(f.n_blocks), dim3(256), 0, s, f);
  return hipGetLastError();
}

template <typename T, int DIN, int MODE>
static hipError_t launch_ens_dout(hipStream_t s, const EnsBatch<T>& b, int count) {
  const EnsArgs<T>& a = b.a[0];
  int wgs = 0;
  for (int i = 0; i < count; ++i) wgs = std::max(wgs, b.a[i].K * b.a[i].P);
  const dim3 grid((unsigned)wgs, (unsigned)count), block(256);
  switch (a.dout) {
#define SSN_CASE(D) case D: hipLaunchKernelGGL((k_ensarray<T, DIN, D, MODE>), grid, block, 0, s, b); break;
    SSN_CASE(1) SSN_CASE(2) SSN_CASE(3) SSN_CASE(4) SSN_CASE(5) SSN_CASE(6) SSN_CASE(7) SSN_CASE(8)
#undef SSN_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <typename T, int MODE>
static hipError_t launch_ens_din(hipStream_t s, const EnsBatch<T>& b, int count) {
  switch (b.a[0].din) {
    case 1: return launch_ens_dout<T, 1, MODE>(s, b, count);
    case 2: return launch_ens_dout<T, 2, MODE>(s, b, count);
    case 3: return launch_ens_dout<T, 3, MODE>(s, b, count);
    case 4: return launch_ens_dout<T, 4, MODE>(s, b, count);
    default: return hipErrorInvalidValue;
  }
}

template <typename T>
hipError_t launch_ensarray_batch(hipStream_t s, const EnsBatch<T>& b, int count) {
  if (b.a[0].fast == 1) return launch_ens_din<T, 1>(s, b, count);
  if (b.a[0].fast == 2) return launch_ens_din<T, 2>(s, b, count);
  return launch_ens_din<T, 0>(s, b, count);
}
template <typename T>
hipError_t launch_ensarray(hipStream_t s, const EnsArgs<T>& a) {
  EnsBatch<T> b{};
  b.a[0] = a;
  return launch_ensarray_batch<T>(s, b, 1);
}

// ---------------------------------------------------------------------------------------------
// k_dft: bind / unbind transforms of the circular-convolution networks.  The reference multiplies by dense
// real-DFT matrices (transform_in 4(d/2+1) x d, transform_out d x 4(d/2+1); 8.25 MB each at d = 1015, read
// every timestep).  d is never a power of two (55 = 5*11, 1015 = 5*7*29, 3751 = 11*11*31), so this is a
// mixed-radix Stockham autosort FFT with generic radix-r butterflies (r <= 32, O(r^2) each) on one workgroup:
// data, scratch and the twiddle table live in LDS (24 B per point), one output point per thread per stage:
//   y[q + s(r p + j)] = sum_k x[q + s(p + m k)] * W_N^{(j k N/r + p j N/n) mod N},  n -> n/r, s -> s r.
// Forward kinds write the 4-slot layout [Re,Im,Re,Im] (A) / [Re,Im,Im,Re] (B) per half-spectrum bin, conjugated
// for the inverted operand; the inverse kind recombines the four product slots (Re = s0 - s1, Im = s2 + s3),
// completes the Hermitian spectrum and transforms back (1/d included).  f32 only: the f64 parity mode multiplies
// by the matrix like the oracle.  A length with a prime factor > 32 (97, 1801, 2049 = 3 * 683) goes through Bluestein's
// chirp-z form: X_k = w_k sum_n (x_n w_n) conj(w_{k-n}), w_n = exp(-i pi n^2 / N) - a circular convolution of smooth
// length M >= 2N - 1, i.e. two Stockham transforms of length M around a pointwise product with a precomputed spectrum.
// ---------------------------------------------------------------------------------------------
// In-register butterflies of the Stockham passes with radix 2, 4 and 8 (round 3): one thread reads the r inputs of an output
// group ONCE, runs the r-point FFT on registers (r log2 r real operations instead of the generic pass's r^2 complex
// multiply-adds, each of which re-read its operand and a twiddle from LDS) and applies the group's r - 1 twiddles on the way
// out.  A 4096-point transform - the Bluestein length of d = 1801 - is four radix-8 passes: 0.26 MB of LDS traffic and ~40 k
// multiply-adds instead of 3 MB and 197 k with (16, 16, 16) generic butterflies.
__device__ inline float2 cmulf(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }
__device__ inline float2 caddf(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ inline float2 csubf(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ inline float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }                 // a * (-i)
__device__ inline void fft4_fwd(float2& c0, float2& c1, float2& c2, float2& c3) {             // outputs in natural order
  const float2 e0 = caddf(c0, c2), e1 = csubf(c0, c2), o0 = caddf(c1, c3), o1 = cmul_mi(csubf(c1, c3));
  c0 = caddf(e0, o0); c2 = csubf(e0, o0); c1 = caddf(e1, o1); c3 = csubf(e1, o1);
}
// r-point forward FFT on registers, r = 2, 4, 8; outputs in natural order
template <int R>
__device__ inline void fft_small(float2* v) {
  if constexpr (R == 2) {
    const float2 a = caddf(v[0], v[1]), b = csubf(v[0], v[1]);
    v[0] = a; v[1] = b;
  } else if constexpr (R == 4) {
    fft4_fwd(v[0], v[1], v[2], v[3]);
  } else {
    // radix 8: even outputs = FFT4 of the sums, odd outputs = FFT4 of the differences times W8^k
    float2 a0 = caddf(v[0], v[4]), a1 = caddf(v[1], v[5]), a2 = caddf(v[2], v[6]), a3 = caddf(v[3], v[7]);
    float2 b0 = csubf(v[0], v[4]), b1 = csubf(v[1], v[5]), b2 = csubf(v[2], v[6]), b3 = csubf(v[3], v[7]);
    const float h = 0.70710678118654752f;
    b1 = make_float2(h * (b1.x + b1.y), h * (b1.y - b1.x));            // * (1 - i) / sqrt 2
    b2 = cmul_mi(b2);                                                  // * -i
    b3 = make_float2(h * (b3.y - b3.x), -h * (b3.x + b3.y));           // * (-1 - i) / sqrt 2
    fft4_fwd(a0, a1, a2, a3);
    fft4_fwd(b0, b1, b2, b3);
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3; v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
  }
}
template <int R>
__device__ inline void dft_pass_small(const float2* x, float2* y, const float2* tw, int L, unsigned us, unsigned fn, int tid, int nthr) {
  const int groups = L / R;                       // output groups (q, p): inputs x[u + k L / R], outputs y[q + s (R p + j)]
  for (int u = tid; u < groups; u += nthr) {
    const unsigned q = (unsigned)u % us, p = (unsigned)u / us;
    float2 v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = x[u + k * groups];
    fft_small<R>(v);
    float2* out = y + q + us * (R * p);
    const unsigned t1 = p * fn;                   // twiddle of output j: W_L^(p j fn), p j fn < L
    out[0] = v[0];
#pragma unroll
    for (int j = 1; j < R; ++j) out[us * j] = cmulf(v[j], tw[t1 * j]);
  }
}

// One in-place pass over M points in LDS, span n (the sub-transforms' current length), radix R: the group (block, i) owns the
// points z[block n + i + k n / R].  DIF: butterfly, then output j times W_n^(i j); DIT (the mirror): input k times W_n^(i k),
// then the butterfly.  W_n^i = T[i M / n], T = the first M / 8 entries of the twiddle table, in LDS; the group's other powers
// by multiplication (six complex products instead of six more table reads).  Points are stored at pad(a) = a + a / 8: in the
// passes with short spans a wave's 64 groups would otherwise hit the same LDS banks eight at a time.
__device__ inline unsigned dft_pad(unsigned a) { return a + (a >> 3); }
template <int R, bool DIT>
__device__ inline void dft_pass_inplace(float2* z, const float2* T, int M, int n, int tid, int nthr) {
  const unsigned sub = (unsigned)n / R, f = (unsigned)(M / n);
  for (int u = tid; u < M / R; u += nthr) {
    const unsigned blk = (unsigned)u / sub, i = (unsigned)u - blk * sub;
    const unsigned g = blk * (unsigned)n + i;
    float2 v[R], w[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = z[dft_pad(g + k * sub)];
    w[1] = T[i * f];
    if constexpr (R >= 4) { w[2] = cmulf(w[1], w[1]); w[3] = cmulf(w[2], w[1]); }
    if constexpr (R == 8) { w[4] = cmulf(w[2], w[2]); w[5] = cmulf(w[4], w[1]); w[6] = cmulf(w[3], w[3]); w[7] = cmulf(w[4], w[3]); }
    if constexpr (DIT) {
#pragma unroll
      for (int k = 1; k < R; ++k) v[k] = cmulf(v[k], w[k]);
    }
    fft_small<R>(v);
    if constexpr (!DIT) {
#pragma unroll
      for (int k = 1; k < R; ++k) v[k] = cmulf(v[k], w[k]);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) z[dft_pad(g + k * sub)] = v[k];
  }
}
template <bool DIT>
__device__ inline void dft_pass_inplace_r(int r, float2* z, const float2* T, int M, int n, int tid, int nthr) {
  if (r == 8) dft_pass_inplace<8, DIT>(z, T, M, n, tid, nthr);
  else if (r == 4) dft_pass_inplace<4, DIT>(z, T, M, n, tid, nthr);
  else dft_pass_inplace<2, DIT>(z, T, M, n, tid, nthr);
}

// One mixed-radix Stockham pass structure over L points held in LDS (x -> result returned; y is scratch)
template <typename RP>      // (RP: pointer to the radix list - generic, or constant address space inside a serial chain of a round)
__device__ inline float2* dft_stockham(float2* x, float2* y, const float2* tw, int L, RP radix, int nr, int tid, int nthr) {
  // Generic radix: four output points per thread at a time: a radix-r butterfly is r dependent LDS round trips (value +
  // twiddle), and one wave per SIMD has nothing else to hide them behind - four independent chains do (a 256-thread workgroup
  // then runs a 1015-point transform as fast as 1024 threads with one point each).
  constexpr int U = 4;
  int n = L, s = 1;
  for (int st = 0; st < nr; ++st) {
    const unsigned r = (unsigned)radix[st], m = (unsigned)n / r, fn = (unsigned)(L / n), fr = (unsigned)L / r;
    const unsigned us = (unsigned)s, stride = us * m;
    if (r == 8 || r == 4 || r == 2) {
      if (r == 8) dft_pass_small<8>(x, y, tw, L, us, fn, tid, nthr);
      else if (r == 4) dft_pass_small<4>(x, y, tw, L, us, fn, tid, nthr);
      else dft_pass_small<2>(x, y, tw, L, us, fn, tid, nthr);
      __syncthreads();
      float2* t2 = x; x = y; y = t2;
      n = (int)m; s *= (int)r;
      continue;
    }
    for (int o0 = tid; o0 < L; o0 += nthr * U) {
      // (all products below stay under 2^32: p j fn < L * 32, L <= 6400, see plan_dft)
      int idx[U], stp[U];
      const float2* xi[U];
      float2 acc[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int o = o0 + u * nthr;
        ok[u] = o < L;
        const unsigned uo = ok[u] ? (unsigned)o : 0u;
        const unsigned q = uo % us, w = uo / us, j = w % r, p = w / r;
        idx[u] = (int)((p * j * fn) % (unsigned)L);
        stp[u] = (int)((j * fr) % (unsigned)L);
        xi[u] = x + q + us * p;
        acc[u] = make_float2(0.0f, 0.0f);
      }
      for (unsigned k = 0; k < r; ++k) {
        float2 v[U], t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = xi[u][stride * k]; t[u] = tw[idx[u]]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc[u].x = fmaf(v[u].x, t[u].x, fmaf(-v[u].y, t[u].y, acc[u].x));
          acc[u].y = fmaf(v[u].x, t[u].y, fmaf(v[u].y, t[u].x, acc[u].y));
          idx[u] += stp[u];
          if (idx[u] >= L) idx[u] -= L;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) if (ok[u]) y[o0 + u * nthr] = acc[u];
    }
    __syncthreads();
    float2* t2 = x; x = y; y = t2;
    n = (int)m; s *= (int)r;
  }
  return x;
}

// ---------------------------------------------------------------------------------------------
// dft4_fft: the same transforms as a FOUR-STEP FFT on the matrix cores (round 3).  L = N1 * N2, input index n = N2 n1 + n2
// (the natural array IS the N1 x N2 matrix), output index k = k1 + N1 k2:
//   step 1   A[k1, n2] = sum_n1 W_N1^(k1 n1) x[n1, n2]          one (2 N1 x 2 N1) . (2 N1 x N2) real product  (K = N1 for real input)
//   step 2   B[k1, n2] = A[k1, n2] W_L^(k1 n2)                   on the accumulators, in registers
//   step 3   X[k1, k2] = sum_n2 B[k1, n2] W_N2^(n2 k2)          one (N1 x 2 N2) . (2 N2 x 2 N2) real product
// with complex products written as real ones ([Ar; Ai] = [[Fr, -Fi], [Fi, Fr]] [xr; xi]).  The two small DFT matrices
// come precomputed in MFMA lane order (v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
// result col = l & 15, row = 4 (l >> 4) + reg), real and imaginary rows (columns) of the same 16 indices in adjacent
// tiles, so that ONE wave holds both parts of an element in the same lane and register: the twiddle multiply and the
// planar stores need no exchange.  A 1015-point transform (29 x 35) is ~0.4 Mflop - ~100 MFMAs per wave of a
// 256-thread workgroup - against 41 dependent LDS round trips per point in the Stockham passes (generic radix-29 / 7 / 5
// butterflies): the radix-r butterflies are what a dense r x r DFT matrix IS, and the matrix cores do them as such.
// Any factorisation works (no radix limit), so a prime length up to 181 is a single dense DFT (N1 = 1).
// LDS: four planar float arrays of L (16 B per point; the Stockham passes take 24).
typedef float f32x4v __attribute__((ext_vector_type(4)));
struct Dft4Io { float* xr; float* xi; float* br; float* bi; };

__device__ inline void dft4_fft(const DftArgs& a, const Dft4Io& io, const bool real_in, const bool need_imag, const int n_out,
                                const int tid, const int nthr) {
  const int N1 = a.N1, N2 = a.N2, L = N1 * N2;
  const int lane = tid & 63, wave = tid >> 6, n_waves = nthr >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int P1 = (N1 + 15) >> 4, C1 = (N2 + 15) >> 4;          // step 1: row-tile pairs x column tiles
  const int N1p = (N1 + 3) & ~3, N2p = (N2 + 3) & ~3;
  const int KS1_all = 2 * N1p / 4, KS1 = real_in ? N1p / 4 : KS1_all;
  // ---- step 1 + 2 -------------------------------------------------------------------------------------------------------
  for (int task = wave; task < P1 * C1; task += n_waves) {
    const int p = task / C1, ct = task - p * C1;
    f32x4v ar = {0.0f, 0.0f, 0.0f, 0.0f}, ai = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* gr = a.g1 + ((size_t)(p * 2 + 0) * KS1_all) * 64 + lane;
    const float* gi = a.g1 + ((size_t)(p * 2 + 1) * KS1_all) * 64 + lane;
    const int n2 = ct * 16 + li;
    for (int ks = 0; ks < KS1; ++ks) {
      const int k = 4 * ks + lk;                             // K index: [0, N1p) real input rows, [N1p, 2 N1p) imaginary ones
      const int n1 = k < N1p ? k : k - N1p;
      const float* src = k < N1p ? io.xr : io.xi;
      const float b = (n1 < N1 && n2 < N2) ? src[n1 * N2 + n2] : 0.0f;
      ar = __builtin_amdgcn_mfma_f32_16x16x4f32(gr[(size_t)ks * 64], b, ar, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f32_16x16x4f32(gi[(size_t)ks * 64], b, ai, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k1 = p * 16 + 4 * lk + v;
      if (k1 < N1 && n2 < N2) {
        const float2 w = a.tw[(unsigned)(k1 * n2) % (unsigned)L];          // exp(-2 pi i k1 n2 / L)
        io.br[k1 * N2 + n2] = ar[v] * w.x - ai[v] * w.y;
        io.bi[k1 * N2 + n2] = ar[v] * w.y + ai[v] * w.x;
      }
    }
  }
  __syncthreads();
  // ---- step 3: X[k1 + N1 k2]; only the column tiles that hold an index below n_out --------------------------------------
  const int R3 = P1, KS3 = 2 * N2p / 4;
  int Q3 = (N2 + 15) >> 4;
  while (Q3 > 1 && N1 * 16 * (Q3 - 1) >= n_out) --Q3;        // (k >= N1 k2: tile pair q starts at k2 = 16 q)
  for (int task = wave; task < R3 * Q3; task += n_waves) {
    const int rt = task / Q3, q = task - rt * Q3;
    f32x4v xr = {0.0f, 0.0f, 0.0f, 0.0f}, xi = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* gr = a.g2 + ((size_t)(q * 2 + 0) * KS3) * 64 + lane;
    const float* gi = a.g2 + ((size_t)(q * 2 + 1) * KS3) * 64 + lane;
    const int k1a = rt * 16 + li;                            // A operand row
    for (int ks = 0; ks < KS3; ++ks) {
      const int k = 4 * ks + lk;
      const int n2 = k < N2p ? k : k - N2p;
      const float* src = k < N2p ? io.br : io.bi;
      const float av = (k1a < N1 && n2 < N2) ? src[k1a * N2 + n2] : 0.0f;
      xr = __builtin_amdgcn_mfma_f32_16x16x4f32(av, gr[(size_t)ks * 64], xr, 0, 0, 0);
      if (need_imag) xi = __builtin_amdgcn_mfma_f32_16x16x4f32(av, gi[(size_t)ks * 64], xi, 0, 0, 0);
    }
    const int k2 = q * 16 + li;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k1 = rt * 16 + 4 * lk + v;
      const int kk = k1 + N1 * k2;
      if (k1 < N1 && k2 < N2 && kk < n_out) { io.xr[kk] = xr[v]; if (need_imag) io.xi[kk] = xi[v]; }
    }
  }
  __syncthreads();
}

// The transforms of dft_body through dft4_fft (planar LDS arrays).
__device__ __forceinline__ void dft4_body(const DftArgs& a, unsigned char* ssn_dft_dyn) {
  const int N = a.N, H = N / 2 + 1, tid = threadIdx.x, nthr = blockDim.x;
  const int L = a.M > 0 ? a.M : N;
  float* base = reinterpret_cast<float*>(ssn_dft_dyn);
  const Dft4Io io{base, base + L, base + 2 * L, base + 3 * L};
  const bool bluestein = a.M > 0;
  for (int i = tid; i < L; i += nthr) {
    float re = 0.0f, im = 0.0f;
    if (i < N) {
      if (a.kind != 5) re = a.src[i];
      else {
        const int w = i < H ? i : N - i;
        const float* p = a.src + 4 * w;
        re = p[0] - p[1];
        im = p[2] + p[3];
        if (w == 0 || 2 * w == N) im = 0.0f;            // purely real bins (their imaginary slots carry no signal)
        im = i < H ? -im : im;                          // Z[i] = Y_w or conj(Y_w); the inverse is conj(FFT(conj Z)): load conj Z
      }
      if (bluestein) {                                  // a_n = x_n w_n
        const float2 c = a.chirp[i];
        const float r2 = re * c.x - im * c.y;
        im = re * c.y + im * c.x;
        re = r2;
      }
    }
    io.xr[i] = re;
    io.xi[i] = im;
  }
  __syncthreads();
  if (!bluestein) {
    // forward kinds: real input, the half spectrum; inverse: complex (Hermitian) input, only the real part of every point
    dft4_fft(a, io, a.kind != 5, a.kind != 5, a.kind != 5 ? H : N, tid, nthr);
  } else {
    dft4_fft(a, io, false, true, L, tid, nthr);
    for (int i = tid; i < L; i += nthr) {               // spectrum of the convolution; conj for the transform back (1 / M folded into fb)
      const float vr = io.xr[i], vi = io.xi[i];
      const float2 f = a.fb[i];
      io.xr[i] = vr * f.x - vi * f.y;
      io.xi[i] = -(vr * f.y + vi * f.x);
    }
    __syncthreads();
    dft4_fft(a, io, false, true, N, tid, nthr);
    for (int i = tid; i < N; i += nthr) {
      const float vr = io.xr[i], vi = -io.xi[i];
      const float2 c = a.chirp[i];
      io.xr[i] = vr * c.x - vi * c.y;
      io.xi[i] = vr * c.y + vi * c.x;
    }
    __syncthreads();
  }
  if (a.kind != 5) {
    const bool conj = a.kind >= 3, slotB = a.kind == 2 || a.kind == 4;
    for (int w = tid; w < H; w += nthr) {
      const float re = io.xr[w], im = conj ? -io.xi[w] : io.xi[w];
      float* d = a.dst + 4 * w;
      const float v2 = slotB ? im : re, v3 = slotB ? re : im;
      if (a.set) { d[0] = re; d[1] = im; d[2] = v2; d[3] = v3; }
      else { d[0] += re; d[1] += im; d[2] += v2; d[3] += v3; }
    }
  } else {
    const float inv = 1.0f / (float)N;
    for (int i = tid; i < N; i += nthr) {
      const float v = io.xr[i] * inv;
      if (a.set) a.dst[i] = v; else a.dst[i] += v;
    }
  }
}

// (ROUND = true: the body inside k_round never runs the four-step engine - its MFMA accumulators would add 8 AGPRs to the
//  round kernel and take every body of every round from 7 to 6 waves per SIMD; such transforms are launched on their own)
// Bluestein's convolution with both transforms in place (DftArgs::inplace): one LDS array of M points.
template <typename A>
__device__ __forceinline__ void dft_bluestein_inplace(const A& a, unsigned char* ssn_dft_dyn) {
  const int N = a.N, H = N / 2 + 1, M = a.M, tid = threadIdx.x, nthr = blockDim.x;
  float2* z = reinterpret_cast<float2*>(ssn_dft_dyn);           // M + M / 8 points (padded), then the first M / (smallest radix) twiddles W_M^i
  float2* T = z + M + M / 8;
  int rmin = 8;
  for (int st = 0; st < a.nr; ++st) rmin = min(rmin, a.radix[st]);
  for (int i = tid; i < M / rmin; i += nthr) T[i] = a.tw[i];       // a pass of radix r reads W_M^e with e < M / r
  for (int i = tid; i < M; i += nthr) {
    float2 v = make_float2(0.0f, 0.0f);
    if (i < N) {
      if (a.kind != 5) v = make_float2(a.src[i], 0.0f);
      else {
        const int w = i < H ? i : N - i;
        const float* p = a.src + 4 * w;
        float re = p[0] - p[1], im = p[2] + p[3];
        if (w == 0 || 2 * w == N) im = 0.0f;            // purely real bins (their imaginary slots carry no signal)
        v = make_float2(re, i < H ? -im : im);          // conj Z (the inverse transform is conj . FFT . conj)
      }
      v = cmulf(v, a.chirp[i]);                         // a_n = x_n w_n
    }
    z[dft_pad(i)] = v;
  }
  __syncthreads();
  int n = M;
  for (int st = 0; st < a.nr; ++st) {                   // decimation in frequency: natural order in, digit-reversed out
    dft_pass_inplace_r<false>(a.radix[st], z, T, M, n, tid, nthr);
    __syncthreads();
    n /= a.radix[st];
  }
  for (int i = tid; i < M; i += nthr) {                 // spectrum of the convolution (fb in the same digit-reversed order, 1 / M folded in); conj for the way back
    const float2 v = cmulf(z[dft_pad(i)], a.fb[i]);
    z[dft_pad(i)] = make_float2(v.x, -v.y);
  }
  __syncthreads();
  n = 1;
  for (int st = a.nr - 1; st >= 0; --st) {              // the mirrored decimation-in-time passes: digit-reversed in, natural out
    n *= a.radix[st];
    dft_pass_inplace_r<true>(a.radix[st], z, T, M, n, tid, nthr);
    __syncthreads();
  }
  // X_k = w_k conj(c_k)
  if (a.kind != 5) {
    const bool conj = a.kind >= 3, slotB = a.kind == 2 || a.kind == 4;
    for (int w = tid; w < H; w += nthr) {
      const float2 zz = z[dft_pad(w)];
      const float2 x = cmulf(make_float2(zz.x, -zz.y), a.chirp[w]);
      const float re = x.x, im = conj ? -x.y : x.y;
      float* d = a.dst + 4 * w;
      const float v2 = slotB ? im : re, v3 = slotB ? re : im;
      if (a.set) { d[0] = re; d[1] = im; d[2] = v2; d[3] = v3; }
      else { d[0] += re; d[1] += im; d[2] += v2; d[3] += v3; }
    }
  } else {
    const float inv = 1.0f / (float)N;
    for (int i = tid; i < N; i += nthr) {
      const float2 zz = z[dft_pad(i)];
      const float2 x = cmulf(make_float2(zz.x, -zz.y), a.chirp[i]);
      const float v = x.x * inv;
      if (a.set) a.dst[i] = v; else a.dst[i] += v;
    }
  }
}

template <bool ROUND = false, typename A = DftArgs>
__device__ __forceinline__ void dft_body(const A& a, unsigned char* ssn_dft_dyn) {
  if constexpr (!ROUND) { if (a.N1 > 0) { dft4_body(a, ssn_dft_dyn); return; } }
  if (a.M > 0 && a.inplace) { dft_bluestein_inplace(a, ssn_dft_dyn); return; }
  const int N = a.N, H = N / 2 + 1, tid = threadIdx.x, nthr = blockDim.x;
  const int L = a.M > 0 ? a.M : N;              // length of the transforms actually run
  float2* x = reinterpret_cast<float2*>(ssn_dft_dyn);
  float2* y = x + L;
  float2* tw = y + L;
  for (int i = tid; i < L; i += nthr) {
    tw[i] = a.tw[i];
    float2 v = make_float2(0.0f, 0.0f);
    if (i < N) {
      if (a.kind != 5) v = make_float2(a.src[i], 0.0f);
      else {
        const int w = i < H ? i : N - i;
        const float* p = a.src + 4 * w;
        float re = p[0] - p[1], im = p[2] + p[3];
        if (w == 0 || 2 * w == N) im = 0.0f;            // purely real bins (their imaginary slots carry no signal)
        // Z[i] = Y_w (i <= N/2) or conj(Y_w) (i > N/2); the inverse transform is conj(FFT(conj Z)): load conj Z
        v = make_float2(re, i < H ? -im : im);
      }
      if (a.M > 0) {                                    // Bluestein: a_n = x_n w_n
        const float2 c = a.chirp[i];
        v = make_float2(v.x * c.x - v.y * c.y, v.x * c.y + v.y * c.x);
      }
    }
    x[i] = v;
  }
  __syncthreads();
  x = dft_stockham(x, y, tw, L, a.radix, a.nr, tid, nthr);
  if (a.M > 0) {
    // X_k = w_k * (a (*) conj(w))_k: multiply the spectra, transform back (conj . FFT . conj; 1 / M folded into fb)
    y = x == reinterpret_cast<float2*>(ssn_dft_dyn) ? x + L : reinterpret_cast<float2*>(ssn_dft_dyn);
    for (int i = tid; i < L; i += nthr) {
      const float2 v = x[i], f = a.fb[i];
      x[i] = make_float2(v.x * f.x - v.y * f.y, -(v.x * f.y + v.y * f.x));
    }
    __syncthreads();
    x = dft_stockham(x, y, tw, L, a.radix, a.nr, tid, nthr);
    for (int i = tid; i < N; i += nthr) {
      const float2 v = make_float2(x[i].x, -x[i].y), c = a.chirp[i];
      x[i] = make_float2(v.x * c.x - v.y * c.y, v.x * c.y + v.y * c.x);
    }
    __syncthreads();
  }
  if (a.kind != 5) {
    const bool conj = a.kind >= 3, slotB = a.kind == 2 || a.kind == 4;
    for (int w = tid; w < H; w += nthr) {
      const float re = x[w].x, im = conj ? -x[w].y : x[w].y;
      float* d = a.dst + 4 * w;
      const float v2 = slotB ? im : re, v3 = slotB ? re : im;
      if (a.set) { d[0] = re; d[1] = im; d[2] = v2; d[3] = v3; }
      else { d[0] += re; d[1] += im; d[2] += v2; d[3] += v3; }
    }
  } else {
    const float inv = 1.0f / (float)N;
    for (int i = tid; i < N; i += nthr) {
      const float v = x[i].x * inv;
      if (a.set) a.dst[i] = v; else a.dst[i] += v;
    }
  }
}
template <int DUMMY>
__global__ __launch_bounds__(1024) void k_dft(DftBatch batch) {
  extern __shared__ __align__(16) unsigned char ssn_dft_smem[];
  dft_body(batch.a[blockIdx.x], ssn_dft_smem);      // (a reference: the radix list is indexed dynamically - a copy would live in scratch)
}

template <typename T>
hipError_t launch_dft(hipStream_t s, const DftBatch& b, int count) {
  int N = 0;
  for (int i = 0; i < count; ++i) N = std::max(N, b.a[i].M > 0 ? b.a[i].M : b.a[i].N);
  int threads = std::min(1024, std::max(64, ((N + 3) / 4 + 63) / 64 * 64));      // four output points per thread (dft_stockham)
  if (b.a[0].inplace) threads = std::min(1024, std::max(64, (N / 8 + 63) / 64 * 64));                // one radix-8 group per thread and pass
  if (b.a[0].N1 > 0) {                // four-step: one wave per tile task of the larger step (at most 16 waves)
    const int tasks = ((b.a[0].N1 + 15) / 16) * ((b.a[0].N2 + 15) / 16);
    threads = std::min(1024, std::max(256, tasks * 64));
  }
  static std::atomic<uint64_t> configured{0};
  if (hipError_t e = set_max_dynamic_lds_once(reinterpret_cast<const void*>(&k_dft<0>), 6400 * 3 * (int)sizeof(float2), configured); e != hipSuccess) return e;
  hipLaunchKernelGGL((k_dft<0>), dim3(count), dim3(threads), (size_t)N * 3 * sizeof(float2), s, b);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_program: one workgroup interprets a list of small vector operators back to back, with a
// workgroup barrier only where the host scheduler found a dependency (level change).  This turns
// the dozen tiny nengo operators between two big kernels into a single launch.
// ---------------------------------------------------------------------------------------------
// for i = tid, tid + 1024, ... < len: store(i, load(i)), four elements per trip with all loads issued first
template <typename T, int U, typename L, typename S>
__device__ inline void vecn_loop(int tid, int len, L load, S store) {
  for (int i0 = tid; i0 < len; i0 += U * 1024) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = i0 + u * 1024; if (i < len) v[u] = load(i); }
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = i0 + u * 1024; if (i < len) store(i, v[u]); }
  }
}
// (long operators - the row hand-offs of a SLAM timestep move 10-20 k elements each - keep 16 loads in flight per thread;
//  32-bit element indices against per-operator base pointers: the interpreter is sensitive to its instruction count)
template <typename T, typename L, typename S>
__device__ inline void vec4_loop(int tid, long long len, L load, S store) {
  if (len > 8192) vecn_loop<T, 16>(tid, (int)len, load, store);
  else vecn_loop<T, 4>(tid, (int)len, load, store);
}

#ifdef SSN_PROGRAM_STAMPS
// diagnostic build only (make F32_EXTRA=-DSSN_PROGRAM_STAMPS): shader-clock stamp after every operator of every program,
// indexed like the micro-operator array; tools/experiments/plan_dump.py prints the per-operator cycles
__device__ unsigned long long g_prog_stamps[2048];
inline hipError_t read_program_stamps(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prog_stamps), sizeof(unsigned long long) * (size_t)std::min(n, 2048));
}
#define SSN_STAMP(i)                                                                                   \
  do {                                                                                                 \
    unsigned long long t__;                                                                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
    if (threadIdx.x == 0 && (i) < 2048) g_prog_stamps[i] = t__;                                        \
  } while (0)
#else
#define SSN_STAMP(i) do {} while (0)
#endif

template <typename T>
__global__ __launch_bounds__(1024) void k_program(const MicroOp<T>* __restrict__ all_ops, const ProgDesc* __restrict__ progs, int n_progs,
                                                  T* __restrict__ gsig, StepCtx* __restrict__ ctx) {
  __shared__ T sred[16];
  __shared__ int sidx[16];
  const int tid = threadIdx.x;
  long long step = ctx->step;          // steps completed before the one being executed
 for (int pi = 0; pi < n_progs; ++pi) {
  const ProgDesc pd = progs[pi];
  const MicroOp<T>* ops = all_ops + pd.op_begin;
  const int n_ops = pd.op_count;
  T* const sig = gsig;
  if (pi > 0) __syncthreads();
  SSN_STAMP(1024 + pd.op_begin);           // (program entry: stamps of the first operator are measured from here)
  for (int o = 0; o < n_ops; ++o) {
    const MicroOp<T> op = ops[o];
    if (op.barrier) __syncthreads();
    switch (op.kind) {
      // element-wise operators: four independent elements per thread and trip, loads before stores - a single
      // workgroup hides memory latency only through loads in flight (the head program of a SLAM timestep moves
      // ~120 k elements: 48 -> 14 us)
      case M_FILL:
        { T* const d = sig + op.dst; for (int i = tid; i < (int)op.len; i += 1024) d[i] = op.a; }
        break;
      case M_AXPY_INC:
        { T* const d = sig + op.dst; const T* const x = sig + op.src;
          vec4_loop<T>(tid, op.len, [&](int i) { return d[i] + op.a * x[i]; }, [&](int i, T v) { d[i] = v; }); }
        break;
      case M_AXPY_SET:
        { T* const d = sig + op.dst; const T* const x = sig + op.src;
          vec4_loop<T>(tid, op.len, [&](int i) { return op.a * x[i]; }, [&](int i, T v) { d[i] = v; }); }
        break;
      case M_LOWPASS:   // dst = a*dst + b*src, b = (1-a)*gain
        { T* const d = sig + op.dst; const T* const x = sig + op.src;
          vec4_loop<T>(tid, op.len, [&](int i) { return op.a * d[i] + op.b * x[i]; }, [&](int i, T v) { d[i] = v; }); }
        break;
      case M_LINCOMB: {   // dst = a * dst + b * (c + sum_k alpha_k * sig[src_k + i]); p0 = LinTerm[i0]
        const LinTerm<T>* const t = (const LinTerm<T>*)op.p0;
        T* const d = sig + op.dst;
        for (int i = tid; i < (int)op.len; i += 1024) {
          T acc = op.c;
          for (int k = 0; k < (int)op.i0; ++k) acc += t[k].alpha * sig[t[k].src + i];
          d[i] = (op.a != T(0) ? op.a * d[i] : T(0)) + op.b * acc;
        }
        break;
      }
      case M_TABLE: {   // p0 = TableSlot*
        const TableSlot* t = (const TableSlot*)op.p0;
        const long long rel = step - t->first_step;
        int row = -1;
        if (rel >= 0 && rel < t->n_idx) row = t->idx[rel];
        const T* rows = (const T*)t->rows;
        const bool have = row >= 0 && row < t->n_rows;
        const T* const trow = rows + (have ? (size_t)row * t->width : 0);
        T* const d = sig + op.dst;
        vec4_loop<T>(tid, op.len, [&](int i) { return have ? trow[i] : T(0); }, [&](int i, T v) { d[i] = v; });
        break;
      }
      case M_MATVEC_INC:
      case M_MATVEC_SET: {   // p0 = W (rows x ld), i0 = cols, i1 = ld; len = rows
        const T* Wm = (const T*)op.p0;
        for (long long r = tid; r < op.len; r += 1024) {
          T s = T(0);
          for (int c = 0; c < (int)op.i0; ++c) s += Wm[(size_t)r * op.i1 + c] * sig[op.src + c];
          if (op.kind == M_MATVEC_INC) sig[op.dst + r] += s; else sig[op.dst + r] = s;
        }
        break;
      }
      case M_ENS_FINISH: {   // p0 = partials [K][P][dout], p1 = dst_idx int32 [K*dout]; len = K*dout, i0 = P, i1 = dout
        const T* part = (const T*)op.p0;
        const int* didx = (const int*)op.p1;
        const int P = (int)op.i0, dout = (int)op.i1;
        for (long long i = tid; i < op.len; i += 1024) {
          const long long k = i / dout, r = i - k * dout;
          T s = T(0);
          for (int p = 0; p < P; ++p) s += part[((size_t)k * P + p) * dout + r];
          sig[didx[i]] = s;
        }
        break;
      }
      case M_GATE: {   // src = [est(d), cur(d), flag], len = d, a = thres, b = rate
        T part = T(0);
        for (long long i = tid; i < op.len; i += 1024) part += sig[op.src + i] * sig[op.src + op.len + i];
        part = wave_sum(part);
        if ((tid & 63) == 0) sred[tid >> 6] = part;
        __syncthreads();
        T dot = T(0);
        for (int w = 0; w < 16; ++w) dot += sred[w];
        const T flag = sig[op.src + 2 * op.len];
        const bool open = (flag <= T(1e-3) && flag >= T(-1e-3)) && dot > op.a;
        for (long long i = tid; i < op.len; i += 1024)
          sig[op.dst + i] = open ? op.b * (sig[op.src + i] - sig[op.src + op.len + i]) : T(0);
        __syncthreads();
        break;
      }
      case M_ARGMAX_GATHER: {   // p1 = sims (i0 rows, scratch), p0 = table (i0 x ld=i1), len = cols: dst = table[argmax]
        // src = P > 0: p1 holds the first maxima of P consecutive slices (k_argmax_partial): P values, then their P row indices
        T best = T(-INFINITY);
        int bi = 0x7fffffff;
        const T* sims = (const T*)op.p1;
        const int n_cand = op.src > 0 ? (int)op.src : (int)op.i0;
        const int* cand_idx = op.src > 0 ? (const int*)(sims + op.src) : nullptr;
        for (int i0 = tid; i0 < n_cand; i0 += 4096) {       // four reads in flight, examined in ascending order
          T v[4];
          int vi[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 1024;
            v[u] = i < n_cand ? sims[i] : T(-INFINITY);
            vi[u] = (cand_idx && i < n_cand) ? cand_idx[i] : i;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
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
        for (int w = 1; w < 16; ++w)
          if (sred[w] > best || (sred[w] == best && sidx[w] < bi)) { best = sred[w]; bi = sidx[w]; }
        if (bi == 0x7fffffff) bi = 0;
        const T* tab = (const T*)op.p0;
        { const T* const trow = tab + (size_t)bi * op.i1; T* const d = sig + op.dst;
          vec4_loop<T>(tid, op.len, [&](int i) { return trow[i]; }, [&](int i, T v) { d[i] = v; }); }
        __syncthreads();
        break;
      }
      case M_PROBE: {   // p0 = ProbeSlot*; src, len = width
        const ProbeSlot* ps = (const ProbeSlot*)op.p0;
        const long long s1 = step + 1;
        if (s1 % ps->every == 0) {
          const long long slot = s1 / ps->every - 1 - ps->base_slot;
          if (slot >= 0 && slot < ps->capacity) {
            T* out = (T*)ps->data + (size_t)slot * op.len;
            const T* const x = sig + op.src;
            vec4_loop<T>(tid, op.len, [&](int i) { return x[i]; }, [&](int i, T v) { out[i] = v; });
          } else if (tid == 0) {
            ctx->probe_overflow = 1;
          }
        }
        break;
      }
      case M_ROW_IN: {    // p0 = bsig, i0 = n_sig: sig[dst..] = bsig[row][dst..], row = step - block_start + 1
        const T* row = (const T*)op.p0 + (size_t)(step - ctx->block_start + 1) * op.i0;
        { const T* const x = row + op.i1; T* const d = sig + op.dst;                                  // i1 = offset in the signal vector
          vec4_loop<T>(tid, op.len, [&](int i) { return x[i]; }, [&](int i, T v) { d[i] = v; }); }
        break;
      }
      case M_ROW_OUT: {   // bsig[row][src..] = sig[src..]
        T* row = (T*)op.p0 + (size_t)(step - ctx->block_start + 1) * op.i0;
        { const T* const x = sig + op.src; T* const d = row + op.i1;
          vec4_loop<T>(tid, op.len, [&](int i) { return x[i]; }, [&](int i, T v) { d[i] = v; }); }
        break;
      }
      case M_REDUCE_SET:
      case M_REDUCE_INC: {   // p0 = partial [i0 chunks][i1 rows_pad]: dst[r] (+)= sum_c partial[c][r], fixed order
        const T* part = (const T*)op.p0;
        const int nc = (int)op.i0;
        for (long long r = tid; r < op.len; r += 1024) {
          T s = T(0);
          int c = 0;
          for (; c + 8 <= nc; c += 8) {            // eight chunk reads in flight, added in chunk order
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
        step += 1;
        if (tid == 0) ctx->step = step;
        break;
      default:
        break;
    }
    SSN_STAMP(pd.op_begin + o);
  }
 }
}

// k_vecops: the first level of a timestep's head program when it is long (SLAM config 3: ~100 k elements of resets
// and hand-offs from the pre stage) - independent element-wise operators, executed grid-wide instead of by the
// program's single workgroup.
template <typename T>
__global__ __launch_bounds__(256) void k_vecops(const MicroOp<T>* __restrict__ ops, int n_ops, T* __restrict__ sig, const StepCtx* __restrict__ ctx) {
  const long long step = ctx->step;
  const long long gtid = (long long)blockIdx.x * 256 + threadIdx.x, gsz = (long long)gridDim.x * 256;
  for (int o = 0; o < n_ops; ++o) {
    const MicroOp<T> op = ops[o];
    switch (op.kind) {
      case M_FILL:
        for (long long i = gtid; i < op.len; i += gsz) sig[op.dst + i] = op.a;
        break;
      case M_AXPY_INC:
        for (long long i = gtid; i < op.len; i += gsz) sig[op.dst + i] += op.a * sig[op.src + i];
        break;
      case M_AXPY_SET:
        for (long long i = gtid; i < op.len; i += gsz) sig[op.dst + i] = op.a * sig[op.src + i];
        break;
      case M_LOWPASS:
        for (long long i = gtid; i < op.len; i += gsz) sig[op.dst + i] = op.a * sig[op.dst + i] + op.b * sig[op.src + i];
        break;
      case M_ROW_IN: {
        const T* row = (const T*)op.p0 + (size_t)(step - ctx->block_start + 1) * op.i0;
        for (long long i = gtid; i < op.len; i += gsz) sig[op.dst + i] = row[op.i1 + i];
        break;
      }
      case M_TABLE: {
        const TableSlot* t = (const TableSlot*)op.p0;
        const long long rel = step - t->first_step;
        int row = -1;
        if (rel >= 0 && rel < t->n_idx) row = t->idx[rel];
        const bool have = row >= 0 && row < t->n_rows;
        const T* rows = (const T*)t->rows;
        for (long long i = gtid; i < op.len; i += gsz) sig[op.dst + i] = have ? rows[(size_t)row * t->width + i] : T(0);
        break;
      }
      default: break;
    }
  }
}
template <typename T>
hipError_t launch_vecops(hipStream_t s, const MicroOp<T>* ops, int n_ops, int wgs, T* sig, const StepCtx* ctx) {
  hipLaunchKernelGGL((k_vecops<T>), dim3(wgs), dim3(256), 0, s, ops, n_ops, sig, ctx);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_program(hipStream_t s, const MicroOp<T>* d_ops, const ProgDesc* progs, int n_progs, T* sig, StepCtx* ctx) {
  hipLaunchKernelGGL((k_program<T>), dim3(1), dim3(1024), 0, s, d_ops, progs, n_progs, sig, ctx);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_matvec: y (+)= W x, W row-major [rows][ld].  One wave per row, 16-byte vector loads of W,
// x re-read through L1/L2 (it is a few KB).  If x is identically zero (the correction input of the
// path integrator outside loop closures) the wave skips W entirely - exact, and it saves the
// whole matrix read.
// ---------------------------------------------------------------------------------------------
template <typename T, bool XLDS, int RW, int CUV = 8>
__device__ __forceinline__ void matvec_body(const MatvecArgs<T>& ma, const int bx, unsigned char* ssn_mv_dyn) {
  const T* __restrict__ Wm = ma.Wm;
  const T* __restrict__ sig_src = ma.src;
  T* __restrict__ sig_dst = ma.dst;
  const int rows = ma.rows, cols = ma.cols, ld = ma.ld, set = ma.set;
  if (bx * 4 * RW >= rows) return;      // (grid.x is sized for the tallest matrix of the batch)
  // y = W x, one wave per RW rows (4 RW rows per workgroup): the source vector is staged in LDS once per
  // workgroup and each lane keeps independent 16-byte row loads in flight per trip - one of each of four rows (RW = 4),
  // or CUV consecutive vectors of one row (RW = 1: short, wide matrices such as a learned decoder product, which
  // would otherwise fill a quarter of the CUs and need the extra loads per lane to cover the memory latency).  Per row the lane-strided accumulation and the wave reduction are
  // the same sequence in every variant.
  // XLDS = false: x is longer than the 48 KB stage (dense ensembles of more than 12 288 neurons) - it goes through
  // the stage a slab at a time; slabs hold a multiple of 512 vectors, so every lane still adds its terms in the same order.
  using vec = typename VecT<T>::type;
  constexpr int W = VecT<T>::W;
  constexpr int CU = RW == 1 ? CUV : 1;            // vectors of one row in flight per lane (RW = 1: few workgroups, more per lane)
  constexpr int SLAB = 48 * 1024 / (int)sizeof(T);
  T* xs = reinterpret_cast<T*>(ssn_mv_dyn);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (bx * 4 + wave) * RW;
  const bool active = r0 < rows;               // (a wave past the last row still takes part in the slab barriers)
  const T* wr[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) wr[q] = Wm + (size_t)min(r0 + q, rows - 1) * ld;
  T s[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) s[q] = T(0);
  const int n_vec = cols / W;
  // The first trip's row loads are issued before the source vector is staged: the stage (global -> LDS -> barrier,
  // ~2 us) would otherwise pass with no matrix load in flight - a quarter of a 1015-column row's time.  (An all-zero
  // source still skips the rest of the matrix.)
  constexpr bool PRE = XLDS && RW > 1;
  T w0[RW][W];
  const bool have0 = PRE && active && lane < n_vec;
  if (have0) {
#pragma unroll
    for (int q = 0; q < RW; ++q) *(vec*)w0[q] = *(const vec*)(wr[q] + (size_t)lane * W);
  }
  if (XLDS) {
    int nz = 0;
    for (int c = threadIdx.x; c < cols; c += 256) {
      const T v = sig_src[c];
      xs[c] = v;
      nz |= (v != T(0));
    }
    nz = __syncthreads_or(nz);
    if (!nz) {                                   // all-zero input (correction / init paths are zero most of the time)
      if (set && lane < RW && r0 + lane < rows) sig_dst[r0 + lane] = T(0);
      return;
    }
  }
  auto accumulate = [&](int c0, int nv, int v_first) {      // columns [c0, c0 + nv W) of the rows against xs[0, nv W)
    for (int v = v_first; v < nv; v += 64 * CU) {
      T w[RW][CU][W], xv[CU][W];
#pragma unroll
      for (int u = 0; u < CU; ++u)
        if (v + 64 * u < nv) {
#pragma unroll
          for (int q = 0; q < RW; ++q) *(vec*)w[q][u] = *(const vec*)(wr[q] + c0 + (size_t)(v + 64 * u) * W);
          *(vec*)xv[u] = *(const vec*)(xs + (size_t)(v + 64 * u) * W);
        }
#pragma unroll
      for (int u = 0; u < CU; ++u)
        if (v + 64 * u < nv) {
#pragma unroll
          for (int q = 0; q < RW; ++q)
#pragma unroll
            for (int j = 0; j < W; ++j) s[q] += w[q][u][j] * xv[u][j];
        }
    }
  };
  if (XLDS) {
    if (!active) return;
    if (have0) {
      T xv[W];
      *(vec*)xv = *(const vec*)(xs + (size_t)lane * W);
#pragma unroll
      for (int q = 0; q < RW; ++q)
#pragma unroll
        for (int j = 0; j < W; ++j) s[q] += w0[q][j] * xv[j];
    }
    accumulate(0, n_vec, PRE ? lane + 64 : lane);
  } else {
    for (int c0 = 0; c0 < n_vec * W; c0 += SLAB) {
      const int cn = min(SLAB, n_vec * W - c0);
      __syncthreads();
      int nz = 0;
      for (int c = threadIdx.x; c < cn; c += 256) { const T v = sig_src[c0 + c]; xs[c] = v; nz |= (v != T(0)); }
      nz = __syncthreads_or(nz);
      if (active && nz) accumulate(c0, cn / W, lane);      // an all-zero slab adds nothing
    }
    if (!active) return;
  }
  const T* __restrict__ x = XLDS ? xs : sig_src;
  for (int c = n_vec * W + lane; c < cols; c += 64)
#pragma unroll
    for (int q = 0; q < RW; ++q) s[q] += wr[q][c] * x[c];
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const T t = wave_sum(s[q]);
    if (lane == 0 && r0 + q < rows) { if (set) sig_dst[r0 + q] = t; else sig_dst[r0 + q] += t; }
  }
}
template <typename T, bool XLDS, int RW, int CUV = 8>
__global__ __launch_bounds__(256) void k_matvec(MatvecBatch<T> batch) {
  extern __shared__ __align__(16) unsigned char ssn_mv_smem[];
  matvec_body<T, XLDS, RW, CUV>(batch.a[blockIdx.y], (int)blockIdx.x, ssn_mv_smem);
}

// Encoder product of a dense population with its neuron update in the epilogue (a body of the round grid; round 4; with
// nr.V == nullptr the plain product of the round grid's 16-rows-per-workgroup variant - one copy of the code for both).  The
// product is matvec_body<T, true, 4>'s - the same loads, the same accumulation order per row - and instead of adding its 16 sums to
// the current vector J it steps the 16 neurons of those rows: lanes 0 - 3 of each wave take one neuron each (J = J[i] + sum, or
// the sum alone for a product that sets), write state and output, and the workgroup leaves the spikes as segment bx of a spike
// list in 16-neuron segments (ascending index; spmv_body, seg_len = 16).  One dependent round of a population hop - product ->
// neurons -> sparse decode -> reduction - less; the current vector is never written.
template <typename T>
__device__ __forceinline__ void matvec_neurons_body(const MatvecNeuronsArgs<T>& a, const int bx, unsigned char* ssn_mv_dyn) {
  const MatvecArgs<T>& ma = a.mv;
  const T* __restrict__ Wm = ma.Wm;
  const T* __restrict__ sig_src = ma.src;
  const int rows = ma.rows, cols = ma.cols, ld = ma.ld, set = ma.set;
  if (bx * 16 >= rows) return;
  using vec = typename VecT<T>::type;
  constexpr int W = VecT<T>::W;
  constexpr int RW = 4;
  T* xs = reinterpret_cast<T*>(ssn_mv_dyn);
  int* wmask = reinterpret_cast<int*>(ssn_mv_dyn + (((size_t)cols * sizeof(T) + 15) / 16) * 16);      // [4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (bx * 4 + wave) * RW;
  const bool active = r0 < rows;
  const T* wr[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) wr[q] = Wm + (size_t)min(r0 + q, rows - 1) * ld;
  T s[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) s[q] = T(0);
  const int n_vec = cols / W;
  T w0[RW][W];
  const bool have0 = active && lane < n_vec;
  if (have0) {
#pragma unroll
    for (int q = 0; q < RW; ++q) *(vec*)w0[q] = *(const vec*)(wr[q] + (size_t)lane * W);
  }
  int nz = 0;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const T v = sig_src[c];
    xs[c] = v;
    nz |= (v != T(0));
  }
  nz = __syncthreads_or(nz);
  if (nz && active) {                            // (an all-zero input adds nothing: the matrix is not read)
    if (have0) {
      T xv[W];
      *(vec*)xv = *(const vec*)(xs + (size_t)lane * W);
#pragma unroll
      for (int q = 0; q < RW; ++q)
#pragma unroll
        for (int j = 0; j < W; ++j) s[q] += w0[q][j] * xv[j];
    }
    for (int v = lane + 64; v < n_vec; v += 64) {
      T w[RW][W], xv[W];
#pragma unroll
      for (int q = 0; q < RW; ++q) *(vec*)w[q] = *(const vec*)(wr[q] + (size_t)v * W);
      *(vec*)xv = *(const vec*)(xs + (size_t)v * W);
#pragma unroll
      for (int q = 0; q < RW; ++q)
#pragma unroll
        for (int j = 0; j < W; ++j) s[q] += w[q][j] * xv[j];
    }
    for (int c = n_vec * W + lane; c < cols; c += 64)
#pragma unroll
      for (int q = 0; q < RW; ++q) s[q] += wr[q][c] * xs[c];
  }
  T t[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) t[q] = __shfl(wave_sum(s[q]), 0, 64);       // (lane 0 holds the sum: to every lane)
  const NeuronsArgs<T>& na = a.nr;
  if (!na.V) {                                   // a plain product (no population behind it): y (+)= W x, as matvec_body<T, true, 4>
    if (active && lane < RW && r0 + lane < rows && (set || nz)) {
      const T sum = lane == 0 ? t[0] : (lane == 1 ? t[1] : (lane == 2 ? t[2] : t[3]));
      T* const d = ma.dst + r0 + lane;
      if (set) *d = sum; else *d += sum;
    }
    return;
  }
  // the neurons of this wave's rows: lane q < 4 takes row r0 + q
  const int i = r0 + lane;
  T act = T(0);
  if (active && lane < RW && i < rows) {
    const T sum = lane == 0 ? t[0] : (lane == 1 ? t[1] : (lane == 2 ? t[2] : t[3]));
    const T J = set ? sum : na.J[i] + sum;
    T v = na.V[i], r = na.R[i];
    act = neuron_step(na.np, J, v, r);
    na.V[i] = v; na.R[i] = r;
    na.out[i] = na.amp * act;
  }
  if (na.seg_list) {
    const unsigned long long mask = __ballot(act != T(0));      // (bits 0 - 3 at most)
    if (lane == 0) wmask[wave] = (int)(mask & 15ull);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += __popc(wmask[w]);
    if (act != T(0)) na.seg_list[bx * 16 + base + __popc((int)(mask & ((1ull << lane) - 1ull)))] = i;
    if (threadIdx.x == 0) na.seg_cnt[bx] = __popc(wmask[0]) + __popc(wmask[1]) + __popc(wmask[2]) + __popc(wmask[3]);
  }
}

template <typename T>
hipError_t launch_matvec(hipStream_t s, const MatvecBatch<T>& b, int count) {
  int rows = 0, cols = 0;
  for (int i = 0; i < count; ++i) { rows = std::max(rows, b.a[i].rows); cols = std::max(cols, b.a[i].cols); }
  const size_t xb = (size_t)cols * sizeof(T);
  const dim3 block(256);
  if (rows <= 4096) {        // four rows per wave would leave most CUs without a workgroup
    const dim3 grid((rows + 3) / 4, count);
    // (vectors of a row in flight per lane: 4 measured better than 8 with the vector in LDS at 1015 x 1524; 8 for the slab variant)
    if (xb <= 48 * 1024) hipLaunchKernelGGL((k_matvec<T, true, 1, 4>), grid, block, xb, s, b);
    else hipLaunchKernelGGL((k_matvec<T, false, 1, 8>), grid, block, 48 * 1024, s, b);
  } else {
    const dim3 grid((rows + 15) / 16, count);
    if (xb <= 48 * 1024) hipLaunchKernelGGL((k_matvec<T, true, 4>), grid, block, xb, s, b);
    else hipLaunchKernelGGL((k_matvec<T, false, 4>), grid, block, 48 * 1024, s, b);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_spmv_partial: y = W s for a SPARSE s (the spike vector of a LIF ensemble: ~5-10 % non-zero), with W
// stored neuron-major, Wt[n][ldt] - one spiking neuron = one contiguous row of `rows` decoders.
//   grid (column tiles of 256, chunks): every workgroup compacts the spike vector into LDS (ascending
//   index order, blocked scan), takes its 1/chunks share of the spike list and accumulates its 256
//   outputs; partial[chunk][r] is reduced in fixed order by the following program (deterministic).
//   Traffic: (#spikes x rows) weights instead of (n x rows).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void spmv_body(const SpmvArgs<T>& sa, const int bx, const int by, unsigned char* smem_all) {
  const T* __restrict__ Wt = sa.Wt;
  const int ldt = sa.ldt;
  const T* __restrict__ spikes = sa.spikes;
  const int n = sa.n, rows = sa.rows;
  T* __restrict__ partial = sa.partial;
  const int rows_pad = sa.rows_pad, chunks = sa.chunks;
  const int* __restrict__ glist = sa.list;
  const int* __restrict__ gcount = sa.count;
  const int seg = sa.seg;
  if (bx * 256 >= rows || by >= chunks) return;      // (grid sized for the largest product of a batch)
  int* counts = reinterpret_cast<int*>(smem_all);       // [257] (+ pad), then the locally compacted spike list
  unsigned char* smem = smem_all + 272 * sizeof(int);
  const int tid = threadIdx.x;
  const int* list = glist;                      // spike list produced by k_neurons_compact ...
  int m = 0;
  if (seg > 0) {
  } else if (glist) {
    m = gcount[0];
  } else {                                      // ... or compacted here (ascending order, blocked scan)
    int* llist = (int*)smem;
    const int per = (n + 255) / 256;
    const int lo = min(n, tid * per), hi = min(n, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += spikes[i] != T(0);
    counts[tid + 1] = cnt;
    if (tid == 0) counts[0] = 0;
    __syncthreads();
    if (tid == 0) for (int t = 1; t <= 256; ++t) counts[t] += counts[t - 1];
    __syncthreads();
    int w = counts[tid];
    for (int i = lo; i < hi; ++i) if (spikes[i] != T(0)) llist[w++] = i;
    __syncthreads();
    m = counts[256];
    list = llist;
  }
  const int c = by;
  const int r = bx * 256 + tid;
  if (seg > 0) {
    // Segmented spike list: chunk c takes the 256-neuron spans [c * seg, (c + 1) * seg).  A span's spikes (ascending neuron index)
    // are copied to a compact LDS list - one trip for the count(s), one for the entries - and every thread walks that list with
    // eight row reads in flight.
    //   seg_len 256 (k_neurons: one segment per span): thread t copies entry t of the segment;
    //   seg_len 16 (matvec_neurons_body: 16 small segments per span): thread t looks at slot t % 16 of small segment t / 16;
    //     the 16 counts sit in lanes 0 - 15 of every wave and are scanned with shuffles (as uniform loads they would take 16
    //     scalar registers of a kernel - k_round - that has none to spare).
    T acc = T(0);
    int* llist = counts;                                // [256] (the scan counters of the list-less path: unused here)
    const int n_small = (n + 15) / 16;
    const int n_span = (n + 255) / 256;
    for (int span = c * seg; span < min(n_span, (c + 1) * seg); ++span) {
      int total, src = -1, at = 0;
      if (sa.seg_len == 16) {
        const int q = tid >> 4, pos = tid & 15, ln = tid & 63;
        int e = 0;
        if (ln < 16) { const int sgm = span * 16 + ln; e = sgm < n_small ? gcount[sgm] : 0; }
        int inc = e;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const int v = __shfl_up(inc, off, 64); if (ln >= off) inc += v; }
        total = __shfl(inc, 15, 64);
        const int mine = __shfl(e, q, 64);
        at = __shfl(inc, q, 64) - mine + pos;
        if (pos < mine) src = (span * 16 + q) * 16 + pos;
      } else {
        total = gcount[span];
        at = tid;
        if (tid < total) src = span * 256 + tid;
      }
      __syncthreads();                                     // (the list of the span before is no longer read)
      if (src >= 0) llist[at] = glist[src];
      __syncthreads();
      if (r < rows) {
        int i = 0;
        for (; i + 8 <= total; i += 8) {
          int j[8]; T w[8], sv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) j[u] = llist[i + u];
#pragma unroll
          for (int u = 0; u < 8; ++u) { w[u] = Wt[(size_t)j[u] * ldt + r]; sv[u] = spikes[j[u]]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += sv[u] * w[u];
        }
        for (; i < total; ++i) { const int j = llist[i]; acc += spikes[j] * Wt[(size_t)j * ldt + r]; }
      }
    }
    if (r < rows) partial[(size_t)c * rows_pad + r] = acc;
    return;
  }
  const int b = (int)((long long)m * c / chunks), e = (int)((long long)m * (c + 1) / chunks);
  T acc = T(0);
  if (r < rows) {
    int i = b;
    for (; i + 4 <= e; i += 4) {
      const int j0 = list[i], j1 = list[i + 1], j2 = list[i + 2], j3 = list[i + 3];
      const T w0 = Wt[(size_t)j0 * ldt + r], w1 = Wt[(size_t)j1 * ldt + r], w2 = Wt[(size_t)j2 * ldt + r], w3 = Wt[(size_t)j3 * ldt + r];
      acc += spikes[j0] * w0; acc += spikes[j1] * w1; acc += spikes[j2] * w2; acc += spikes[j3] * w3;
    }
    for (; i < e; ++i) { const int j = list[i]; acc += spikes[j] * Wt[(size_t)j * ldt + r]; }
    partial[(size_t)c * rows_pad + r] = acc;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_spmv_partial(SpmvBatch<T> batch) {
  extern __shared__ __align__(16) unsigned char ssn_spmv_smem[];
  spmv_body<T>(batch.a[blockIdx.z], (int)blockIdx.x, (int)blockIdx.y, ssn_spmv_smem);
}

template <typename T>
hipError_t launch_spmv_partial(hipStream_t s, const SpmvBatch<T>& b, int count) {
  int rows = 0, chunks = 0;
  size_t lds = 272 * sizeof(int);
  for (int i = 0; i < count; ++i) {
    rows = std::max(rows, b.a[i].rows); chunks = std::max(chunks, b.a[i].chunks);
    if (!b.a[i].list) lds = std::max(lds, 272 * sizeof(int) + (size_t)b.a[i].n * sizeof(int));
  }
  hipLaunchKernelGGL((k_spmv_partial<T>), dim3((rows + 255) / 256, chunks, count), dim3(256), lds, s, b);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_matvec_ordered: y[r] = sum_j Wt[j][r] * x[j] accumulated strictly in j order (one thread per row,
// transposed matrix so that consecutive threads read consecutive addresses).  Used by the clean-up in
// the f64 parity mode, where argmax ties are decided by rounding and the order must match the oracle.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_matvec_ordered(const T* __restrict__ Wt, const T* __restrict__ x,
                                                        T* __restrict__ y, int rows, int cols, int ldt) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  T s = T(0);
  for (int j = 0; j < cols; ++j) s = s + Wt[(size_t)j * ldt + r] * x[j];
  y[r] = s;
}

template <typename T>
__global__ void k_transpose(const T* __restrict__ src, T* __restrict__ dst, int rows, int cols, int ld, int ldt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
  dst[(size_t)c * ldt + r] = src[(size_t)r * ld + c];
}

template <typename T>
hipError_t launch_matvec_ordered(hipStream_t s, const T* Wt, const T* x, T* y, int rows, int cols, int ldt) {
  hipLaunchKernelGGL((k_matvec_ordered<T>), dim3((rows + 255) / 256), dim3(256), 0, s, Wt, x, y, rows, cols, ldt);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_transpose(hipStream_t s, const T* src, T* dst, int rows, int cols, int ld, int ldt) {
  const int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL((k_transpose<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, cols, ld, ldt);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void neurons_body(const NeuronsArgs<T>& na, const int bx, unsigned char* smem) {
  const NeuronParams<T> np = na.np;
  const T* __restrict__ J = na.J;
  T* __restrict__ out = na.out;
  T* __restrict__ V = na.V;
  T* __restrict__ R = na.R;
  const int n = na.n;
  const T amp = na.amp;
  int* __restrict__ seg_list = na.seg_list;
  int* __restrict__ seg_cnt = na.seg_cnt;
  if (bx * 256 >= n) return;        // (grid.x is sized for the largest population of the batch)
  const int i = bx * 256 + threadIdx.x;
  T a = T(0);
  if (i < n) {
    T v = V[i], r = R[i];
    a = neuron_step(np, J[i], v, r);
    V[i] = v; R[i] = r;
    out[i] = amp * a;
  }
  if (seg_list) {
    // this workgroup's spikes as an ascending index list (segment blockIdx.x of the ensemble's segmented spike
    // list): the spike-sparse decoder product walks segments in order and never has to scan the spike vector
    int* wcnt = reinterpret_cast<int*>(smem);      // [4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(a != T(0));
    if (lane == 0) wcnt[wave] = __popcll(mask);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wcnt[w];
    if (a != T(0)) seg_list[bx * 256 + base + __popcll(mask & ((1ull << lane) - 1ull))] = i;
    if (threadIdx.x == 0) seg_cnt[bx] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_neurons(NeuronsBatch<T> batch) {
  __shared__ __align__(16) unsigned char smem[16];
  neurons_body<T>(batch.a[blockIdx.y], (int)blockIdx.x, smem);
}

template <typename T>
hipError_t launch_neurons(hipStream_t s, const NeuronsBatch<T>& b, int count) {
  int n = 0;
  for (int i = 0; i < count; ++i) n = std::max(n, b.a[i].n);
  hipLaunchKernelGGL((k_neurons<T>), dim3((n + 255) / 256, count), dim3(256), 0, s, b);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_pes: W[r][c] += kappa * err[r] * act[c]   (fused delta + increment).
// One workgroup per (PES_ROWS rows, 1024-column tile).  With neuron-major weights (the learned decoders of the associative
// memory: a row per neuron, `err` = the filtered activities) ~1 % of the rows have a nonzero factor: a workgroup reads the
// factors of its eight rows at once and updates the rows that have one - 1 270 workgroups, all resident at once, instead of
// a sweep of 10 150 that each load one factor and leave (round 3; SLAM config 3).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void pes_body(const PesArgs<T>& a, const int bx, const int by) {
  // The PES_ROWS row factors of this workgroup come with ONE load per wave (lane l: row r0 + l); the rows whose factor is not zero -
  // 1.3 % of the memory population's filtered activities - are walked by ballot.  (Eight factors per workgroup as uniform loads
  // were 1 269 workgroups per SLAM timestep, most of which waited a memory trip for eight zeros: 12 ms of workgroup-slot time per
  // timestep, tools/round_stamps.py.)
  const int r0 = by * PES_ROWS;
  const int lane = threadIdx.x & 63;
  const T el = (lane < PES_ROWS && r0 + lane < a.rows) ? a.kappa * a.err[r0 + lane] : T(0);
  if (a.lp_dst) {
    // folded filter of the row factors (planned only where one column tile covers the matrix: this workgroup is the only reader
    // of its factors): every wave has its copies before thread q advances factor q
    __syncthreads();
    const int q = threadIdx.x;
    if (bx == 0 && q < PES_ROWS && r0 + q < a.rows) a.lp_dst[r0 + q] = a.lp_a * a.lp_dst[r0 + q] + a.lp_b * a.lp_src[r0 + q];
  }
  unsigned long long m = __ballot(el != T(0));
  const int c0 = bx * 1024;
  while (m) {
    const int q = __builtin_ctzll(m);
    m &= m - 1ull;
    const T e = __shfl(el, q, 64);
    T* wr = a.Wm + (size_t)(r0 + q) * a.ld;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + j * 256 + threadIdx.x;
      if (c < a.cols) wr[c] += e * a.act[c];
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_pes(T* __restrict__ Wm, const T* __restrict__ err, const T* __restrict__ act,
                                             int rows, int cols, int ld, T kappa) {
  pes_body<T>(PesArgs<T>{Wm, err, act, rows, cols, ld, kappa, nullptr, nullptr, T(0), T(0)}, (int)blockIdx.x, (int)blockIdx.y);
}

template <typename T>
hipError_t launch_pes(hipStream_t s, T* Wm, const T* err, const T* act, int rows, int cols, int ld, T kappa) {
  hipLaunchKernelGGL((k_pes<T>), dim3((cols + 1023) / 1024, (rows + PES_ROWS - 1) / PES_ROWS), dim3(256), 0, s, Wm, err, act, rows, cols, ld, kappa);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_voja: for each neuron i that spiked (a_i != 0):
//   E[i][:] += lr_dt * (1 + learn) * (scale_i * a_i * key[:] - a_i * E[i][:])      (Appendix A.8)
// One wave per neuron row; silent neurons exit after one load.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void voja_body(const VojaArgs<T>& v, const int bx) {
  // A wave owns VOJA_ROWS_PER_WAVE consecutive rows: their activities come with ONE load (lane l: row i0 + l), the rows that moved
  // - a percent of them in the reference's memory - are walked by ballot.  (One row per wave was 2 538 workgroups per SLAM
  // timestep that each waited a memory trip for one zero and left: 12 ms of workgroup-slot time per timestep, tools/round_stamps.py.)
  const int lane = threadIdx.x & 63;
  const int i0 = (bx * 4 + (int)(threadIdx.x >> 6)) * VOJA_ROWS_PER_WAVE;
  if (i0 >= v.rows) return;
  const int il = i0 + lane;
  const T al = (lane < VOJA_ROWS_PER_WAVE && il < v.rows) ? v.spk[il] : T(0);
  unsigned long long m = __ballot(al != T(0));
  if (m == 0ull) return;
  const T g = v.lr_dt * (T(1) + v.learn[0]);
  while (m) {
    const int r = __builtin_ctzll(m);
    m &= m - 1ull;
    const int i = i0 + r;
    const T a = __shfl(al, r, 64);
    const T sa = v.scale[i] * a;
    T* er = v.E + (size_t)i * v.ld;
    for (int c = lane; c < v.cols; c += 64) er[c] += g * (sa * v.key[c] - a * er[c]);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_voja(T* __restrict__ E, const T* __restrict__ spk, const T* __restrict__ key,
                                              const T* __restrict__ learn, const T* __restrict__ scale,
                                              int rows, int cols, int ld, T lr_dt) {
  voja_body<T>(VojaArgs<T>{E, spk, key, learn, scale, rows, cols, ld, lr_dt}, (int)blockIdx.x);
}

template <typename T>
hipError_t launch_voja(hipStream_t s, T* E, const T* spk, const T* key, const T* learn, const T* scale,
                       int rows, int cols, int ld, T lr_dt) {
  hipLaunchKernelGGL((k_voja<T>), dim3((rows + 4 * VOJA_ROWS_PER_WAVE - 1) / (4 * VOJA_ROWS_PER_WAVE)), dim3(256), 0, s, E, spk, key, learn, scale, rows, cols, ld, lr_dt);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Time-batched operators (feed-forward pre / post stages, stages.py): one launch runs an operator for
// all B timesteps of a block on the block buffer bsig[B+1][n_sig] (row r = signals after step
// block_start + r).  A per-step GEMV becomes one GEMM (the matrix is read once per block), a Lowpass
// becomes a scan along time.
// ---------------------------------------------------------------------------------------------
// element i of row t (the signals after step block_start + t) of one element-wise operator
template <typename T>
__device__ inline void kb_elementwise_at(const BatchOp<T>& o, long long t, long long i) {
  T* drow = o.bsig + (size_t)(t + 1) * o.n_sig;
  const T* srow = o.bsig + (size_t)(t + 1 - o.src_prev) * o.n_sig;
  switch (o.kind) {
    case M_FILL: drow[o.dst + i] = o.a; break;
    case M_AXPY_INC: drow[o.dst + i] += o.a * srow[o.src + i]; break;
    case M_AXPY_SET: drow[o.dst + i] = o.a * srow[o.src + i]; break;
    case M_TABLE: {
      const TableSlot* tb = (const TableSlot*)o.p0;
      const long long rel = o.step0 + t - tb->first_step;
      int row = -1;
      if (rel >= 0 && rel < tb->n_idx) row = tb->idx[rel];
      drow[o.dst + i] = (row >= 0 && row < tb->n_rows) ? ((const T*)tb->rows)[(size_t)row * tb->width + i] : T(0);
      break;
    }
    case M_PROBE: {
      const ProbeSlot* ps = (const ProbeSlot*)o.p0;
      const long long s1 = o.step0 + t + 1;
      if (s1 % ps->every == 0) {
        const long long slot = s1 / ps->every - 1 - ps->base_slot;
        if (slot >= 0 && slot < ps->capacity) ((T*)ps->data)[(size_t)slot * o.len + i] = drow[o.src + i];
      }
      break;
    }
    default: break;
  }
}
template <typename T>
__device__ inline void kb_elementwise_body(const BatchOp<T>& o) {
  const long long total = (long long)o.B * o.len;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long t = e / o.len;
    kb_elementwise_at<T>(o, t, e - t * o.len);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void kb_elementwise(BatchOp<T> o) { kb_elementwise_body<T>(o); }
// Several element-wise operators of a time-batched stage in one launch (a launch costs ~5 us + the gap behind it).  Thread
// (t, i) runs element i of row t of EVERY operator of the list, in list order: operators may depend on each other where
// both touch a signal element at the same index i of the same row (ssn_host.hip run_batch checks: equal range origins, no
// previous-row read in between) - a reset, the sums into it and the hand-off of the result are one launch.
constexpr int KB_MULTI_ROWS = 8;      // rows per workgroup of kb_elementwise_multi (independent iterations: eight accesses in flight per thread)
template <typename T>
__global__ __launch_bounds__(256) void kb_elementwise_multi(BatchOpList<T> l) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int B = l.op[0].B;
  for (long long t0 = (long long)blockIdx.y * KB_MULTI_ROWS; t0 < B; t0 += (long long)gridDim.y * KB_MULTI_ROWS)
    for (int q = 0; q < l.count; ++q) {
      if (i >= l.op[q].len) continue;
      const BatchOp<T> o = l.op[q];
      if (t0 + KB_MULTI_ROWS <= B && (o.kind == M_FILL || o.kind == M_AXPY_INC || o.kind == M_AXPY_SET)) {
        // full chunk of a reset / sum: all loads first (the compiler cannot move a load across a store to the same buffer)
        T* d = o.bsig + (size_t)(t0 + 1) * o.n_sig + o.dst + i;
        const T* u = o.bsig + (size_t)(t0 + 1 - o.src_prev) * o.n_sig + o.src + i;
        T x[KB_MULTI_ROWS], y[KB_MULTI_ROWS];
        if (o.kind != M_FILL) {
#pragma unroll
          for (int r = 0; r < KB_MULTI_ROWS; ++r) x[r] = u[(size_t)r * o.n_sig];
        }
        if (o.kind == M_AXPY_INC) {
#pragma unroll
          for (int r = 0; r < KB_MULTI_ROWS; ++r) y[r] = d[(size_t)r * o.n_sig];
        }
#pragma unroll
        for (int r = 0; r < KB_MULTI_ROWS; ++r) {
          T v = o.a;
          if (o.kind == M_AXPY_SET) v = o.a * x[r];
          if (o.kind == M_AXPY_INC) { const T p = o.a * x[r]; v = y[r] + p; }
          d[(size_t)r * o.n_sig] = v;
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < KB_MULTI_ROWS; ++r)
        if (t0 + r < B) kb_elementwise_at<T>(o, t0 + r, i);
    }
}
template <typename T>
hipError_t launch_batch_elementwise(hipStream_t s, const BatchOpList<T>& l) {
  long long len = 0;
  for (int q = 0; q < l.count; ++q) len = std::max(len, (long long)l.op[q].len);
  if (len <= 0 || l.count <= 0 || l.op[0].B <= 0) return hipSuccess;
  const int gy = std::min((l.op[0].B + KB_MULTI_ROWS - 1) / KB_MULTI_ROWS, 16384);
  hipLaunchKernelGGL((kb_elementwise_multi<T>), dim3((unsigned)((len + 255) / 256), (unsigned)gy), dim3(256), 0, s, l);
  return hipGetLastError();
}

// scan along time: y[t+1] = a*y[t] + b*u[t], one thread per signal element, rows read ahead in groups
template <typename T>
__global__ __launch_bounds__(64) void kb_lowpass(BatchOp<T> o) {
  const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
  if (i >= o.len) return;
  T y = o.bsig[o.dst + i];                                  // carry-in (row 0)
  const T* u = o.bsig + (size_t)(1 - o.src_prev) * o.n_sig + o.src + i;
  T* out = o.bsig + (size_t)o.n_sig + o.dst + i;
  int t = 0;
  for (; t + 8 <= o.B; t += 8) {
    T v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = u[(size_t)(t + j) * o.n_sig];
#pragma unroll
    for (int j = 0; j < 8; ++j) { y = o.a * y + o.b * v[j]; out[(size_t)(t + j) * o.n_sig] = y; }
  }
  for (; t < o.B; ++t) { y = o.a * y + o.b * u[(size_t)t * o.n_sig]; out[(size_t)t * o.n_sig] = y; }
}

// f32 scan along time, time split into C chunks per element (the f64 parity mode keeps the sequential recurrence
// of kb_lowpass): thread (element i, chunk c) first runs its chunk from a zero state to get the chunk's own
// contribution L_c, the chunk carries  Y_c = a^len_c * Y_{c-1} + L_c  are chained through LDS, then every thread
// re-runs its chunk from the right carry and writes the outputs.  2 * B / C dependent steps instead of B.
template <int C, int E>                       // C time chunks x E elements per workgroup (C * E threads)
__global__ __launch_bounds__(C * E) void kb_lowpass_chunked(BatchOp<float> o) {
  __shared__ float sl[C][E];
  __shared__ float sp[C][E];
  const int e = threadIdx.x % E, c = threadIdx.x / E;
  const long long i = (long long)blockIdx.x * E + e;
  const int per = (o.B + C - 1) / C;
  const int t0 = min(o.B, c * per), t1 = min(o.B, t0 + per);
  const bool ok = i < o.len;
  const float* u = o.bsig + (size_t)(1 - o.src_prev) * o.n_sig + o.src + (ok ? i : 0);
  float y = 0.0f, pw = 1.0f;
  if (ok) {
    int t = t0;
    for (; t + 8 <= t1; t += 8) {           // eight row reads in flight
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = u[(size_t)(t + j) * o.n_sig];
#pragma unroll
      for (int j = 0; j < 8; ++j) { y = o.a * y + o.b * v[j]; pw *= o.a; }
    }
    for (; t < t1; ++t) { y = o.a * y + o.b * u[(size_t)t * o.n_sig]; pw *= o.a; }
  }
  sl[c][e] = y; sp[c][e] = pw;
  __syncthreads();
  if (!ok) return;
  float carry = o.bsig[o.dst + i];                            // row 0: state before the block
  for (int q = 0; q < c; ++q) carry = sp[q][e] * carry + sl[q][e];
  float* out = o.bsig + (size_t)o.n_sig + o.dst + i;
  y = carry;
  int t = t0;
  for (; t + 8 <= t1; t += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = u[(size_t)(t + j) * o.n_sig];
#pragma unroll
    for (int j = 0; j < 8; ++j) { y = o.a * y + o.b * v[j]; out[(size_t)(t + j) * o.n_sig] = y; }
  }
  for (; t < t1; ++t) { y = o.a * y + o.b * u[(size_t)t * o.n_sig]; out[(size_t)t * o.n_sig] = y; }
}

// C[t][r] (+)= sum_c A[t][c] * W[r][c]   (A = block rows of the source signal, W row-major [rows][ld])
// 32 x 32 output tile per 256-thread workgroup, K staged through LDS in slabs of 32.
template <typename T>
__global__ __launch_bounds__(256) void kb_gemm(BatchOp<T> o) {
  __shared__ T As[32][33];
  __shared__ T Ws[32][33];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 x 16 threads, 2 x 2 outputs each
  const int t0 = blockIdx.y * 32, r0 = blockIdx.x * 32;
  const T* A = o.bsig + (size_t)(1 - o.src_prev) * o.n_sig + o.src;
  const T* Wm = (const T*)o.p0;
  T acc[2][2] = {{T(0), T(0)}, {T(0), T(0)}};
  const int rows = (int)o.len;
  for (int k0 = 0; k0 < o.cols; k0 += 32) {
    for (int e = threadIdx.x; e < 32 * 32; e += 256) {
      const int rr = e >> 5, cc = e & 31;
      const int t = t0 + rr, r = r0 + rr, c = k0 + cc;
      As[rr][cc] = (t < o.B && c < o.cols) ? A[(size_t)t * o.n_sig + c] : T(0);
      Ws[rr][cc] = (r < rows && c < o.cols) ? Wm[(size_t)r * o.ld + c] : T(0);
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      const T a0 = As[ty][kk], a1 = As[ty + 16][kk];
      const T w0 = Ws[tx][kk], w1 = Ws[tx + 16][kk];
      acc[0][0] += a0 * w0; acc[0][1] += a0 * w1;
      acc[1][0] += a1 * w0; acc[1][1] += a1 * w1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int t = t0 + ty + 16 * a, r = r0 + tx + 16 * b;
      if (t < o.B && r < rows) {
        T* d = o.bsig + (size_t)(t + 1) * o.n_sig + o.dst + r;
        if (o.kind == M_MATVEC_SET) *d = acc[a][b]; else *d += acc[a][b];
      }
    }
}

// f32 GEMMs on the matrix cores: the f32-input MFMAs take f32 operands and accumulate an exact f32 FMA chain
// (MI355X_MICROARCH.md: their rate equals the packed-FMA vector peak, but needs no register blocking to get there).
// 64 x 64 output tile per workgroup, K in slabs of 32 through LDS, next slab prefetched into registers while the
// current one is multiplied.  v_mfma_f32_32x32x2_f32 (k_gemm_nt_mfma_f32): A[i = l & 31][k = l >> 5],
// B[k = l >> 5][j = l & 31]; result col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)
// (cdna_hip_programming.md, MFMA layouts).
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Sixteen waves per 64 x 64 tile, one 16 x 16 accumulator each (v_mfma_f32_16x16x4_f32; operands
// A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; result col = l & 15, row = 4 (l >> 4) + reg): the grids of the
// time-batched stages are small (384 tiles at 1000 x 1524 on 256 CUs), so a CU holds one or two tiles - with four waves
// per tile (one 32 x 32 accumulator each, the first version) a SIMD had a single wave and nothing to hide its operand
// loads behind; four waves per SIMD measured 10 % faster (48 vs 54 us).
template <int BK>
__global__ __launch_bounds__(1024) void kb_gemm_mfma_f32(BatchOp<float> o) {
  __shared__ float As[64][BK + 4];          // row stride 36: (36 row + k) mod 64 is distinct for 16 rows x 4 k - conflict-free operand reads
  __shared__ float Ws[64][BK + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int rows = (int)o.len, cols = o.cols;
  const int t0 = blockIdx.y * 64, r0 = blockIdx.x * 64;
  const float* __restrict__ A = o.bsig + (size_t)(1 - o.src_prev) * o.n_sig + o.src;
  const float* __restrict__ Wm = (const float*)o.p0;
  constexpr int RS = 1024 / BK, RPT = 64 / RS;      // this thread stages rows lr + RS i, column lc of a slab
  const int lr = tid / BK, lc = tid % BK;
  float ra[RPT], rw[RPT];
  auto fetch = [&](int k0) {
    const int c = k0 + lc;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int t = t0 + lr + RS * i, r = r0 + lr + RS * i;
      ra[i] = (t < o.B && c < cols) ? A[(size_t)t * o.n_sig + c] : 0.0f;
      rw[i] = (r < rows && c < cols) ? Wm[(size_t)r * o.ld + c] : 0.0f;
    }
  };
  f32x4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
  fetch(0);
  for (int k0 = 0; k0 < cols; k0 += BK) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) { As[lr + RS * i][lc] = ra[i]; Ws[lr + RS * i][lc] = rw[i]; }
    __syncthreads();
    if (k0 + BK < cols) fetch(k0 + BK);
    const float* ap = &As[wm * 16 + (lane & 15)][lane >> 4];
    const float* wp = &Ws[wn * 16 + (lane & 15)][lane >> 4];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk], wp[kk], acc, 0, 0, 0);
    __syncthreads();
  }
  const int r = r0 + wn * 16 + (lane & 15);
  if (r < rows) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int t = t0 + wm * 16 + 4 * (lane >> 4) + v;
      if (t < o.B) {
        float* d = o.bsig + (size_t)(t + 1) * o.n_sig + o.dst + r;
        if (o.kind == M_MATVEC_SET) *d = acc[v]; else *d += acc[v];
      }
    }
  }
}

// k_argmax_partial: first maximum (lowest index on ties) of each of P consecutive slices of a long similarity
// vector; the program's argmax then only looks at the P candidates (a single workgroup scanning 10^6 similarities
// costs ~100 us).  Output: P values followed by P int row indices.
template <typename T>
__global__ __launch_bounds__(256) void k_argmax_partial(const T* __restrict__ sims, long long n, T* __restrict__ out, int P, int nsplit) {
  __shared__ T sred[4];
  __shared__ int sidx[4];
  const long long per = (n + P - 1) / P;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  const int tid = threadIdx.x;
  T best = T(-INFINITY);
  int bi = 0x7fffffff;
  for (long long i0 = lo + tid; i0 < hi; i0 += 1024) {
    T v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * 256;
      v[u] = i < hi ? sims[i] : T(-INFINITY);
      for (int sp = 1; sp < nsplit; ++sp) v[u] += i < hi ? sims[(size_t)sp * n + i] : T(0);     // K-split partial products, fixed order
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (v[u] > best) { best = v[u]; bi = (int)(i0 + u * 256); }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T ov = __shfl_down(best, off, 64);
    const int oi = __shfl_down(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { sred[tid >> 6] = best; sidx[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    best = sred[0]; bi = sidx[0];
    for (int w = 1; w < 4; ++w)
      if (sred[w] > best || (sred[w] == best && sidx[w] < bi)) { best = sred[w]; bi = sidx[w]; }
    out[blockIdx.x] = best;
    reinterpret_cast<int*>(out + P)[blockIdx.x] = bi == 0x7fffffff ? (int)lo : bi;
  }
}

template <typename T>
hipError_t launch_argmax_partial(hipStream_t s, const T* sims, long long n, T* out, int P, int nsplit) {
  hipLaunchKernelGGL((k_argmax_partial<T>), dim3((unsigned)P), dim3(256), 0, s, sims, n, out, P, nsplit);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Clean-up over a large sample grid without the pass over the table (sspspace.grid_factors): the similarity of
// x to grid point j = a * N + r is  sum_k Re(w_k conj(X_k) E1[a, k] . Erest[r, k]),  X = half spectrum of x.
// k_grid_lhs forms the (n_a x 2K) left operand from X and the axis-1 factors; k_gemm_nt_mfma_f32 multiplies it
// with the (N x 2K) factors of the remaining axes on the matrix cores: C[m][n] = sum_k A[m][k] W[n][k].
// 64 x 64 tile per 256-thread workgroup (4 waves x one 32 x 32 accumulator), K slabs of 32 through LDS, register prefetch.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_grid_lhs(const T* __restrict__ X, const T* __restrict__ E, int lde,
                                                  T* __restrict__ A, int lda, int na, int K) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= na * K) return;
  const int a = i / K, k = i - a * K;
  const T xr = X[2 * k], xi = X[2 * k + 1];
  const T er = E[(size_t)a * lde + 2 * k], ei = E[(size_t)a * lde + 2 * k + 1];
  A[(size_t)a * lda + 2 * k] = xr * er + xi * ei;              //  Re(conj(X) E)
  A[(size_t)a * lda + 2 * k + 1] = xi * er - xr * ei;          // -Im(conj(X) E)
}

// the same as a body of the round grid: 256 (a, k) pairs per block
template <typename T>
__device__ __forceinline__ void grid_lhs_body(const GridLhsArgs<T>& g, const int bx) {
  const int i = bx * 256 + (int)threadIdx.x;
  if (i >= g.na * g.K) return;
  const int a = i / g.K, k = i - a * g.K;
  const T xr = g.X[2 * k], xi = g.X[2 * k + 1];
  const T er = g.E[(size_t)a * g.lde + 2 * k], ei = g.E[(size_t)a * g.lde + 2 * k + 1];
  g.A[(size_t)a * g.lda + 2 * k] = xr * er + xi * ei;
  g.A[(size_t)a * g.lda + 2 * k + 1] = xi * er - xr * ei;
}

template <typename T>
hipError_t launch_grid_lhs(hipStream_t s, const T* X, const T* E, int lde, T* A, int lda, int na, int K) {
  hipLaunchKernelGGL((k_grid_lhs<T>), dim3((unsigned)((na * K + 255) / 256)), dim3(256), 0, s, X, E, lde, A, lda, na, K);
  return hipGetLastError();
}

template <int BK>
__global__ __launch_bounds__(256) void k_gemm_nt_mfma_f32(const float* __restrict__ A, int lda, const float* __restrict__ Wm, int ldw,
                                                          float* __restrict__ C, int ldc, int M, int N, int Kall, int kper) {
  // blockIdx.z = K split: columns [z kper, min(Kall, (z + 1) kper)) of both operands, partial product z of C (M x ldc each)
  const int kz = blockIdx.z * kper;
  const int K = min(Kall - kz, kper);
  A += kz; Wm += kz; C += (size_t)blockIdx.z * M * ldc;
  __shared__ float As[64][BK + 1];
  __shared__ float Ws[64][BK + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int t0 = blockIdx.y * 64, r0 = blockIdx.x * 64;
  const int lr = tid >> 5, lc = tid & 31;
  float ra[8], rw[8];
  auto fetch = [&](int k0) {
    const int c = k0 + lc;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = t0 + lr + 8 * i, r = r0 + lr + 8 * i;
      ra[i] = (t < M && c < K) ? A[(size_t)t * lda + c] : 0.0f;
      rw[i] = (r < N && c < K) ? Wm[(size_t)r * ldw + c] : 0.0f;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { As[lr + 8 * i][lc] = ra[i]; Ws[lr + 8 * i][lc] = rw[i]; }
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);
    const float* ap = &As[wm * 32 + (lane & 31)][lane >> 5];
    const float* wp = &Ws[wn * 32 + (lane & 31)][lane >> 5];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], wp[kk], acc, 0, 0, 0);
    __syncthreads();
  }
  const int r = r0 + wn * 32 + (lane & 31);
  if (r < N) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int t = t0 + wm * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
      if (t < M) C[(size_t)t * ldc + r] = acc[v];
    }
  }
}

template <typename T>
hipError_t launch_gemm_nt(hipStream_t s, const T* A, int lda, const T* Wm, int ldw, T* C, int ldc, int M, int N, int K, int splits) {
  if constexpr (sizeof(T) == 4) {
    const int kper = ((K + splits - 1) / splits + 31) / 32 * 32;
    hipLaunchKernelGGL((k_gemm_nt_mfma_f32<32>), dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)((K + kper - 1) / kper)),
                       dim3(256), 0, s, A, lda, Wm, ldw, C, ldc, M, N, K, kper);
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;      // f64 is the parity mode: it keeps the ordered pass over the table
  }
}

template <typename T>
hipError_t launch_batch_op(hipStream_t s, const BatchOp<T>& o) {
  if (o.B <= 0 || o.len <= 0) return hipSuccess;
  if (o.kind == M_LOWPASS) {
    if constexpr (sizeof(T) == 4) {
      if (o.B >= 256) {
        hipLaunchKernelGGL((kb_lowpass_chunked<32, 32>), dim3((unsigned)((o.len + 31) / 32)), dim3(1024), 0, s, o);
        return hipGetLastError();
      }
    }
    hipLaunchKernelGGL((kb_lowpass<T>), dim3((unsigned)((o.len + 63) / 64)), dim3(64), 0, s, o);
  } else if ((o.kind == M_MATVEC_INC || o.kind == M_MATVEC_SET) && sizeof(T) == 4 && o.cols >= 64 && o.len >= 32 && o.B >= 32) {
    if constexpr (sizeof(T) == 4)
      hipLaunchKernelGGL((kb_gemm_mfma_f32<32>), dim3((unsigned)((o.len + 63) / 64), (unsigned)((o.B + 63) / 64)), dim3(1024), 0, s, o);
  } else if (o.kind == M_MATVEC_INC || o.kind == M_MATVEC_SET) {
    hipLaunchKernelGGL((kb_gemm<T>), dim3((unsigned)((o.len + 31) / 32), (unsigned)((o.B + 31) / 32)), dim3(256), 0, s, o);
  } else {
    const long long total = (long long)o.B * o.len;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL((kb_elementwise<T>), dim3(grid), dim3(256), 0, s, o);
  }
  return hipGetLastError();
}

// conversion helpers for uploads / downloads (host double <-> device T), row-padded
template <typename T>
__global__ void k_convert_in(const double* __restrict__ src, T* __restrict__ dst, int64_t rows, int64_t cols, int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * ld) return;
  const int64_t r = i / ld, c = i - r * ld;
  dst[i] = c < cols ? (T)src[r * cols + c] : T(0);
}
template <typename T>
__global__ void k_convert_out(const T* __restrict__ src, double* __restrict__ dst, int64_t rows, int64_t cols, int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r = i / cols, c = i - r * cols;
  dst[i] = (double)src[r * ld + c];
}
template <typename T>
hipError_t launch_convert_in(hipStream_t s, const double* src, T* dst, int64_t rows, int64_t cols, int64_t ld) {
  const int64_t n = rows * ld;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((k_convert_in<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, cols, ld);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_convert_out(hipStream_t s, const T* src, double* dst, int64_t rows, int64_t cols, int64_t ld) {
  const int64_t n = rows * cols;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL((k_convert_out<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, cols, ld);
  return hipGetLastError();
}

}  // namespace ssn
#include "ssn_block.hpp"
#include "ssn_round.hpp"
namespace ssn {

#define SSN_INSTANTIATE(T)                                                                                   \
  template hipError_t launch_ens_block<T>(hipStream_t, const BlockArgs<T>&);                                 \
  template hipError_t launch_dft<T>(hipStream_t, const DftBatch&, int);                                            \
  template bool ens_block_supported<T>(int, int, int, int*, int*, int*, int*);                                         \
  template hipError_t launch_ensarray<T>(hipStream_t, const EnsArgs<T>&);                                    \
  template hipError_t launch_ensarray_batch<T>(hipStream_t, const EnsBatch<T>&, int);                        \
  template hipError_t launch_dec_pack<T>(hipStream_t, const T*, T*, int, int, int, int, int, int);           \
  template hipError_t launch_state_unpack<T>(hipStream_t, const T*, T*, int64_t, int, T);                    \
  template hipError_t launch_program<T>(hipStream_t, const MicroOp<T>*, const ProgDesc*, int, T*, StepCtx*); \
  template hipError_t launch_vecops<T>(hipStream_t, const MicroOp<T>*, int, int, T*, const StepCtx*);        \
  template hipError_t launch_ens_finish<T>(hipStream_t, const FinishArgs<T>&);                               \
  template hipError_t launch_matvec<T>(hipStream_t, const MatvecBatch<T>&, int);            \
  template hipError_t launch_matvec_ordered<T>(hipStream_t, const T*, const T*, T*, int, int, int);         \
  template hipError_t launch_spmv_partial<T>(hipStream_t, const SpmvBatch<T>&, int); \
  template hipError_t launch_transpose<T>(hipStream_t, const T*, T*, int, int, int, int);                   \
  template hipError_t launch_neurons<T>(hipStream_t, const NeuronsBatch<T>&, int); \
  template hipError_t launch_pes<T>(hipStream_t, T*, const T*, const T*, int, int, int, T);                 \
  template hipError_t launch_voja<T>(hipStream_t, T*, const T*, const T*, const T*, const T*, int, int, int, T); \
  template hipError_t launch_batch_op<T>(hipStream_t, const BatchOp<T>&);                                   \
  template hipError_t launch_grid_lhs<T>(hipStream_t, const T*, const T*, int, T*, int, int, int);          \
  template hipError_t launch_argmax_partial<T>(hipStream_t, const T*, long long, T*, int, int);             \
  template hipError_t launch_gemm_nt<T>(hipStream_t, const T*, int, const T*, int, T*, int, int, int, int, int); \
  template hipError_t launch_batch_elementwise<T>(hipStream_t, const BatchOpList<T>&);                      \
  template hipError_t launch_convert_in<T>(hipStream_t, const double*, T*, int64_t, int64_t, int64_t);      \
  template hipError_t launch_convert_out<T>(hipStream_t, const T*, double*, int64_t, int64_t, int64_t);      \
  template hipError_t launch_round<T>(hipStream_t, const RoundArgs<T>&, int, size_t);

}  // namespace ssn
