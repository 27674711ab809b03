// ssn_launch.hpp - argument structs and launcher declarations shared by the host executor
// (ssn_host.hip) and the kernel translation units (ssn_f32.hip, ssn_f64.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace ssn {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: a process that drives simulators on several GPUs
// (the `device` argument of ShardedPathIntegration / ShardedSLAM) has to make it once on each of them.  `done` is the
// launcher's own bit mask of devices already configured (one static per kernel instantiation; any thread may launch).
inline hipError_t set_max_dynamic_lds_once(const void* fn, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

template <typename T>
struct NeuronParams {
  int type;            // 0 LIF, 1 LIFRate, 2 ReLU
  T dt, tau_rc, tau_ref, min_voltage;
};

struct StepCtx;

template <typename T>
struct EnsArgs {
  const T* enc;        // [K][din][n_pad]
  const T* bias;       // [K][n_pad]
  const T* dec;        // [K][dout][n_pad]   (pre-scaled by amplitude/dt)
  T* V;                // [K][n_pad]
  T* R;                // [K][n_pad]
  const T* sig;        // signal vector (reads x)
  T* partials;         // [K][P][dout]
  int64_t x_off;
  int K, n, n_pad, din, dout;
  int P;               // chunks (workgroups) per ensemble
  int chunk_vec;       // 16-byte vectors per chunk
  NeuronParams<T> np;
  // fused input assembly (recurrent ensemble array): x = block-buffer row of this step (pre stage
  // output) + sum_j alpha_j * sig[rec_src_j + (xi - rec_dst_j)] for absolute x index xi inside term j
  const T* xrows;      // block buffer base or nullptr (then x comes from sig)
  long long n_sig;
  const StepCtx* ctx;
  int n_rec;
  long long rec_dst[4], rec_src[4], rec_len[4];
  T rec_alpha[4];
  int fast;            // LIF fast variant: packed state word (+ neuron-major spike-sparse decoders if dout >= 3)
  // deferred finish (one launch per timestep): the prologue of step t completes step t-1 for its own
  // ensemble - sums the previous partials, updates the recurrent filter states, hands results on - and
  // the partials / filter states ping-pong between two buffers by step parity.
  int direct;                      // P == 1 (one workgroup per ensemble): decoded rows go straight to sig_w[didx[...]] - no finish
  int defer;
  int sub;                         // step offset of this launch inside a captured graph (step = ctx->step + sub)
  long long partials_stride;       // elements between the two partial buffers
  const int* didx;                 // [K*dout] destination signal per row
  const int* lp_has;               // [K*dout] 1: row feeds a lowpass filter state
  const T* lp_a;                   // [K*dout]
  const T* lp_b;                   // [K*dout]
  T* fstate;                       // [2][K*dout] filter states, index parity = step parity
  const int* xrow;                 // [K*din] row (0..dout-1) of the same ensemble whose filter state feeds this input, or -1
  const T* xalpha;                 // [K*din]
  const unsigned char* rowout;     // [K*dout] 1: decoded value also goes to the post stage's block row
  T* bsig;
  T* sig_w;
};

// Finish of a fused recurrent ensemble array: one thread per decoded row (k, r):
//   v = sum_p partials[k][p][r];  sig[dst] = v;  optional lowpass state update;  optional hand-off to
//   the post stage (block-buffer row of this step).  The last workgroup to finish advances the step.
template <typename T>
struct FinishArgs {
  const T* partials;
  const int* didx;         // [K*dout] destination signal per row
  const int* lp_state;     // [K*dout] state signal updated from this row, or -1
  const T* lp_a;           // [K*dout]
  const T* lp_b;           // [K*dout]
  const unsigned char* rowout;   // [K*dout] 1: also written to the block buffer
  T* sig;
  T* bsig;
  long long n_sig;
  StepCtx* ctx;
  unsigned int* ticket;
  int K, P, dout, n_blocks;
  // deferred-finish cores: mode 0 = classic finish of this step; 1 = flush (finish the block's last step from
  // the ping-pong buffers, publish filter states to the signal vector); 2 = begin (load filter states)
  int mode;
  T* fstate;
  long long partials_stride;
  const int* lp_has;
};

// Whole-block launch of a recurrent ensemble array whose ensembles are independent inside a block
// (ssn_block.hpp): one workgroup per ensemble, parameters and state held in registers for B timesteps.
template <typename T>
struct BlockArgs {
  const T* enc;        // [K][din][n_pad]
  const T* bias;       // [K][n_pad]
  const T* dec;        // [K][n_pad][DP] (dec_neuron_major) or [K][dout][n_pad]
  T* S;                // [K][n_pad] packed LIF state words
  const T* xrows;      // block buffer (pre-stage rows)
  T* bsig;             // block buffer (rows handed to the post stage)
  const T* sig;        // signal vector: filter states in
  T* sig_w;            // signal vector: decoded values / filter states out
  const int* didx;     // [K*dout] destination signal per decoded row
  const int* lp_state; // [K*dout] filter-state signal fed by the row, or -1
  const T* lp_a;       // [K*dout]
  const T* lp_b;       // [K*dout]
  const int* xrow;     // [K*din] row of the same ensemble whose filter state feeds this input, or -1
  const T* xalpha;     // [K*din]
  const unsigned char* rowout;   // [K*dout] 1: decoded value goes to the post stage's block row
  long long n_sig, x_off;
  int K, n, n_pad, din, dout;
  int B;               // timesteps in this launch
  int row0;            // block-buffer row of the launch's first timestep
  int threads, tpb, npt;   // workgroup size, kernel variant (launch bound, neurons per thread): threads * npt >= n
  int dec_neuron_major;
  int enc_lds;         // kernel variant keeps the encoders in LDS instead of registers
  NeuronParams<T> np;
  // Split ensembles (round 3; f32, dout <= 4): P member workgroups per ensemble, member m steps neurons [m * n_member, ...)
  // and the members exchange their four partial sums every timestep (k_ens_block, ssn_block.hpp).  P = 1: one workgroup.
  int P;
  int n_member;        // neurons per member (a multiple of 4)
  unsigned int* xslots;   // [3 buffers][K][16 members][4 words], all words = the sentinel at launch (host memset)
  int* xerr;           // set to 1 by a member that waited too long for a partner (the launch then produced garbage)
  unsigned long long* slot_stats;   // [2]: (wave, round) slots stepped / silent among them (f32; one atomic pair per wave and launch)
};
constexpr unsigned int BLOCK_XCHG_SENTINEL = 0x7fc0deadu;      // a NaN payload no sum produces
// neuron groups of a thread that k_ens_block steps side by side (ssn_block.hpp; the host deals neurons to (wave, round) slots
// accordingly: Sim::reorder_block_neurons)
#ifndef SSN_BLOCK_IL
#define SSN_BLOCK_IL 2
#endif

// k_dft: the real-DFT maps of a circular-convolution network (reference binding.py:23-74) as a mixed-radix FFT.
struct DftArgs {
  const float* src;      // d reals (kinds 1-4) or 4*(d/2+1) slot products (kind 5)
  float* dst;
  const float2* tw;      // exp(-2 pi i k / N), k = 0..N-1
  int N;                 // transform length d
  int kind;              // 1 A, 2 B, 3 A conj, 4 B conj (forward, 4 slots per half-spectrum bin); 5 inverse
  int set;               // 1: dst = result, 0: dst += result
  int nr;                // number of radices
  int radix[12];         // product = N (or M), each <= 32
  // Bluestein (N has a prime factor > 32 - 97, 1801, 2049 = 3 * 683): the length-N DFT as a circular convolution of
  // length M >= 2N - 1 (M = 2^a 3^b 5^c, `radix` then factors M and `tw` has M entries)
  int M;                 // 0: N is transformed directly
  const float2* chirp;   // w_n = exp(-i pi n^2 / N), n = 0..N-1
  const float2* fb;      // FFT_M of the wrapped conjugate chirp, divided by M
  // inplace = 1 (M a product of radices 2, 4, 8 only): the two transforms of the convolution run IN PLACE on one LDS array of
  // M points - decimation in frequency forward (natural order in, digit-reversed out), `fb` stored in that digit-reversed
  // order, the mirrored decimation-in-time passes back (digit-reversed in, natural out): no reordering pass, 8 B of LDS per
  // point instead of 24, twiddles read from `tw` in global memory (dft_bluestein_inplace, ssn_kernels.hpp)
  int inplace;
  // Four-step transform on the matrix cores (dft4_fft, ssn_kernels.hpp), used instead of the Stockham passes when N1 > 0:
  // L = N1 * N2 (L = M for Bluestein), two small dense DFTs as f32 MFMA products around a twiddle multiply.
  int N1, N2;
  const float* g1;       // step-1 A operand in MFMA lane order: [ceil(N1 / 16) row-tile pairs][re rows | im rows][k steps][64]
  const float* g2;       // step-3 B operand: [ceil(N2 / 16) column-tile pairs][re cols | im cols][k steps][64]
};

// Independent operators of one kind that the scheduler placed next to each other share one launch
// (blockIdx.y selects the operator): every launch costs ~4-5 us on this GPU whatever its size.
constexpr int MAX_BATCH = 4;
template <typename T> struct MatvecArgs { const T* Wm; const T* src; T* dst; int rows, cols, ld, set; };
template <typename T> struct MatvecBatch { MatvecArgs<T> a[MAX_BATCH]; };
template <typename T> struct NeuronsArgs { NeuronParams<T> np; const T* J; T* out; T* V; T* R; int n; T amp; int* seg_list; int* seg_cnt; };
template <typename T> struct NeuronsBatch { NeuronsArgs<T> a[MAX_BATCH]; };
// A dense population's encoder product with its neuron update in the epilogue (matvec_neurons_body): the workgroup that owns 16
// rows of W steps those 16 neurons - J = (set ? 0 : J[i]) + W[i] . x never goes back to memory - and leaves their spikes as
// segment bx of a spike list in 16-neuron segments (SpmvArgs::seg_len = 16).
template <typename T> struct MatvecNeuronsArgs { MatvecArgs<T> mv; NeuronsArgs<T> nr; };
struct DftBatch { DftArgs a[MAX_BATCH]; };
template <typename T> struct SpmvArgs { const T* Wt; int ldt; const T* spikes; int n, rows; T* partial; int rows_pad, chunks; const int* list; const int* count;
                                       int seg; int seg_len; };      // seg_len: neurons per list segment (256: k_neurons; 16: matvec_neurons_body)
template <typename T> struct SpmvBatch { SpmvArgs<T> a[MAX_BATCH]; };
constexpr int MAX_ENS_BATCH = 2;
template <typename T> struct EnsBatch { EnsArgs<T> a[MAX_ENS_BATCH]; };

enum MicroKind {
  M_FILL = 1, M_AXPY_INC, M_AXPY_SET, M_LOWPASS, M_TABLE, M_MATVEC_INC, M_MATVEC_SET, M_ENS_FINISH,
  M_GATE, M_ARGMAX_GATHER, M_PROBE, M_STEP_END, M_ROW_IN, M_ROW_OUT, M_REDUCE_SET, M_REDUCE_INC,
  M_LINCOMB            // dst = a * dst + b * (c + sum_k alpha_k * sig[src_k + i]); p0 = LinTerm[i0]
};
template <typename T> struct LinTerm { long long src; T alpha; };

template <typename T>
struct MicroOp {
  int kind;
  int barrier;         // workgroup barrier before this op
  long long dst, src, len;
  T a, b;
  const void* p0;
  const void* p1;
  long long i0, i1;
  T c;
};

struct ProgDesc { int op_begin, op_count; };

struct TableSlot {     // lives in device memory; re-pointed by ssn_set_table without re-planning
  const void* rows;    // [n_rows][width] in simulator dtype
  const int* idx;      // [n_idx]
  long long n_rows, width, n_idx, first_step;
};

struct ProbeSlot {
  void* data;          // [capacity][width]
  long long every, base_slot, capacity;
};

struct StepCtx {
  long long step;        // steps completed
  int probe_overflow;
  int pad;
  long long block_start; // step number at which the current time-batched block began
  long long finished;    // deferred-finish cores: last step whose finish has been applied
};

// One operator of a time-batched stage, executed for rows t = 0..B-1 of the block.  Row r of the block
// buffer `bsig` ([B+1][n_sig]) holds the signals after step (block_start + r); row 0 is the carry-in.
template <typename T>
struct BatchOp {
  T* bsig;
  long long n_sig;       // row stride
  int B;
  int kind;              // MicroKind (M_FILL, M_AXPY_*, M_LOWPASS, M_TABLE, M_MATVEC_*, M_PROBE)
  long long dst, src, len;   // len = rows for matvec
  int cols, ld;
  int src_prev;          // 1: read row t (value before this step's update) instead of row t+1
  T a, b;
  const void* p0;        // W | TableSlot* | ProbeSlot*
  long long step0;       // = block_start
};

// ---------------------------------------------------------------------------------------------
// Rounds (k_round, ssn_round.hpp): all mutually independent operators of a timestep - big and small - share ONE
// launch.  A round is a list of entries; entry i owns the virtual blocks [first, first + gx * gy) of the grid and
// names the operator body that runs them.  The arguments of every body live in device memory (they do not change
// from timestep to timestep; the step counter is read from StepCtx).
// ---------------------------------------------------------------------------------------------
constexpr int PES_ROWS = 32;     // rows per workgroup of the PES update (pes_body: their factors are one load per wave; at most 64)
template <typename T> struct PesArgs { T* Wm; const T* err; const T* act; int rows, cols, ld; T kappa;
                                       // round plan (round 4): the Lowpass that filters the row factors (the memory population's activities, reference
                                       // associativememory.py:38-43: PES(pre_synapse)) folded into the update - row r's factor is advanced by the
                                       // workgroup that has just read it: lp_dst[r] = lp_a * lp_dst[r] + lp_b * lp_src[r]  (lp_dst == err; null: not folded)
                                       T* lp_dst; const T* lp_src; T lp_a, lp_b; };
template <typename T> struct VojaArgs { T* E; const T* spk; const T* key; const T* learn; const T* scale; int rows, cols, ld; T lr_dt; };

enum RoundKind {
  RK_GLUE = 0,        // element-wise / reduction micro-operators, one chunk per block (args: GlueBlock map)
  RK_GATE, RK_ARGMAX, // whole-vector micro-operators: one block each (args: MicroOp)
  RK_MATVEC_R1, RK_MATVEC_R4, RK_SPMV, RK_NEURONS, RK_DFT, RK_PES, RK_VOJA,
  RK_ENS_3_4_S, RK_ENS_3_5_S, RK_ENS_1_1_D,     // k_ensarray<din, dout, spike-sparse | dense decoders>
  RK_ENS_SMALL,                                 // arrays of many small 1-D ensembles: a wave per ensemble, 16 per block (ens_small_body)
  RK_GRID_LHS, RK_GRID_DOT,                     // clean-up over a sample grid from its factor tables (round 4): left operand; similarities
  RK_MATVEC_NEURONS,                            // encoder product + neuron update of a dense population (MatvecNeuronsArgs)
  RK_SOLO                                       // a serial chain of single-workgroup units, one block: args -> {n, code_0, sub_0, ...} in RoundArgs::chain;
                                                // code >= 0: micro-operator index, code < 0: DftArgs at RoundArgs::arena + 16 * (-code - 1)
};
// Clean-up similarities of a 2-D sample grid without the pass over its table (reference slam.py:209-215: 10^4 x d every timestep,
// 40 MB at d = 1015): sims[a * nn + r] = sum_k A[a][k] * W[r][k] with A = Re / -Im of conj(X) * lhs[a] (grid_lhs_body) and W the
// factors of the remaining axis - two 100 x 2K tables that stay in L2 instead of the table streamed from HBM.
template <typename T> struct GridLhsArgs { const T* X; const T* E; int lde; T* A; int lda; int na; int K; };
template <typename T> struct GridDotArgs { const T* A; int lda; const T* W; int ldw; T* dst; int nn; int k2; int na; };
struct GlueBlock { int op; int chunk; };        // micro-operator index (into RoundArgs::mops); chunk of it (low 24 bits), timestep offset (high 8)
// op < 0: a CHAIN of element-aligned micro-operators - RoundArgs::chain[-op - 1 ...] = {n, op_0, sub_0, ..., op_{n-1}, sub_{n-1}} -
// run back to back by this block on elements [chunk * GLUE_ROWS, ...): a dependent operator whose every shared element
// sits at the same index (a filter update behind the reduction it filters, next step's input hand-off behind that update)
// needs no barrier and no launch of its own when the same thread handles the same index in program order.
struct RoundEntry { int kind; int first; int gx; int gy; const void* args; int lo; int cnt; };   // blocks [lo, lo + cnt) of the gx x gy grid
constexpr int MAX_ROUND_ENTRIES = 96;
constexpr int ENS_SMALL_PER_WAVE = 4;         // small ensembles one wave of ens_small_body steps (16 per workgroup; 6: no gain - 98.3 vs 97.8 us per SLAM timestep; 8 costs k_round a 73rd VGPR)
constexpr int VOJA_ROWS_PER_WAVE = 8;         // rows of the encoder matrix one wave of the Voja body looks at (32 per workgroup)
constexpr int SOLO_MAX_MEMBERS = 24;         // members of a serial chain (their descriptors are staged in LDS)
template <typename T>
struct RoundArgs {
  int n;
  int head;                 // blocks [0, head) run the virtual block of their own index (latency-bound bodies first); see stride
  // Interleaved dispatch (round 3): block b >= head runs virtual block head + ((b - head) * stride) mod (gridDim - head), with
  // stride coprime to the modulus and ~0.618 of it.  The grid's blocks are dispatched in index order and the entries own
  // contiguous virtual ranges, so without this a round ran its bodies ONE AFTER THE OTHER (the oscillators' HBM-bound blocks,
  // then the Voja rows, then 8 000 latency-bound product-ensemble blocks ...: 70 us = the sum of the bodies' stand-alone times);
  // the stride deals every entry's blocks evenly over the whole launch, so bandwidth-bound and latency-bound bodies share the CUs.
  unsigned int stride;      // 0 / 1: identity
  int pad;                  // launch id (diagnostic stamps)
  int prio;                 // blocks [0, prio) - the latency-bound head of the grid - raise their waves' issue priority (s_setprio 3)
  int pad2;
  const MicroOp<T>* mops;
  const int* chain;
  const unsigned char* arena;   // body arguments of the plan (serial chains address their transforms through it)
  T* sig;
  StepCtx* ctx;
  RoundEntry e[MAX_ROUND_ENTRIES];
};
template <typename T> hipError_t launch_round(hipStream_t, const RoundArgs<T>&, int n_blocks, size_t lds_bytes);
// elements (rows for the reductions) of a micro-operator that one block of a round handles
constexpr int GLUE_CHUNK = 1024;
constexpr int GLUE_ROWS = 256;

constexpr int MAX_BATCH_OPS = 24;
template <typename T> struct BatchOpList { BatchOp<T> op[MAX_BATCH_OPS]; int count; };
template <typename T> hipError_t launch_batch_elementwise(hipStream_t, const BatchOpList<T>&);   // independent element-wise ops, one launch

template <typename T> hipError_t launch_ensarray(hipStream_t, const EnsArgs<T>&);
template <typename T> hipError_t launch_ensarray_batch(hipStream_t, const EnsBatch<T>&, int count);   // equal din / dout / variant
template <typename T> hipError_t launch_dec_pack(hipStream_t, const T* src, T* dst, int K, int dout, int n, int n_pad, int DP, int unpack);
template <typename T> hipError_t launch_state_unpack(hipStream_t, const T* src, T* out, int64_t n, int want_refractory, T r_offset);
template <typename T> hipError_t launch_ens_finish(hipStream_t, const FinishArgs<T>&);
template <typename T> hipError_t launch_dft(hipStream_t, const DftBatch&, int count);   // (T only selects the translation unit)
template <typename T> hipError_t launch_ens_block(hipStream_t, const BlockArgs<T>&);
template <typename T> bool ens_block_supported(int din, int dout, int n, int* threads, int* tpb, int* npt, int* enc_lds);
template <typename T> hipError_t launch_program(hipStream_t, const MicroOp<T>* ops, const ProgDesc* progs, int n_progs, T* sig, StepCtx* ctx);
template <typename T> hipError_t launch_vecops(hipStream_t, const MicroOp<T>* ops, int n_ops, int wgs, T* sig, const StepCtx* ctx);
template <typename T> hipError_t launch_matvec(hipStream_t, const MatvecBatch<T>&, int count);
template <typename T> hipError_t launch_matvec_ordered(hipStream_t, const T* Wt, const T* x, T* y, int rows, int cols, int ldt);
template <typename T> hipError_t launch_transpose(hipStream_t, const T* src, T* dst, int rows, int cols, int ld, int ldt);
template <typename T> hipError_t launch_spmv_partial(hipStream_t, const SpmvBatch<T>&, int count);
template <typename T> hipError_t launch_neurons(hipStream_t, const NeuronsBatch<T>&, int count);
template <typename T> hipError_t launch_pes(hipStream_t, T* W, const T* err, const T* act, int rows, int cols, int ld, T kappa);
template <typename T> hipError_t launch_voja(hipStream_t, T* E, const T* spk, const T* key, const T* learn, const T* scale,
                                            int rows, int cols, int ld, T lr_dt);
template <typename T> hipError_t launch_batch_op(hipStream_t, const BatchOp<T>&);
// clean-up over a factored sample grid: left operand from the half spectrum, then C[M x N] = A[M x K] . W[N x K]^T
template <typename T> hipError_t launch_argmax_partial(hipStream_t, const T* sims, long long n, T* out, int P, int nsplit);
template <typename T> hipError_t launch_grid_lhs(hipStream_t, const T* X, const T* E, int lde, T* A, int lda, int na, int K);
template <typename T> hipError_t launch_gemm_nt(hipStream_t, const T* A, int lda, const T* Wm, int ldw, T* C, int ldc, int M, int N, int K, int splits);
template <typename T> hipError_t launch_convert_in(hipStream_t, const double* src, T* dst, int64_t rows, int64_t cols, int64_t ld);
template <typename T> hipError_t launch_convert_out(hipStream_t, const T* src, double* dst, int64_t rows, int64_t cols, int64_t ld);

}  // namespace ssn
