// ssn_host.hip - C ABI (include/ssn.h) and step-loop executor of libssn_hip.so.
//
// ssn_create turns the host-built operator list into a launch plan:
//   * runs of small vector operators become "programs" (one k_program launch each),
//   * ensemble arrays, dense matvecs, neuron updates and learning rules become their own kernels,
//   * the tail program of step s and the head program of step s+1 are fused into one launch,
//   * `steps_per_graph` consecutive timesteps are captured into one hipGraph; ssn_run_steps
//     replays it, so the host issues one graph launch per `steps_per_graph` simulated steps.
// Every kernel reads the current step from device memory (StepCtx), which is what makes the
// captured graph step-invariant.  There is no CPU fallback: without a HIP device ssn_create fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ssn.h"
#include "ssn_launch.hpp"

namespace {

__global__ void k_set_block(ssn::StepCtx* ctx, long long block_start) { ctx->block_start = block_start; }
__global__ void k_advance(ssn::StepCtx* ctx, long long n) { ctx->step += n; }

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

}  // namespace
namespace ssn {
int probe_fail(int code, const char* what, hipError_t e) {
  return e == hipSuccess ? fail(code, "%s", what) : fail(code, "%s failed: %s", what, hipGetErrorString(e));
}
}  // namespace ssn
namespace {

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess)                                                                        \
      return fail(e__ == hipErrorOutOfMemory ? SSN_ENOMEM : SSN_EHIP, "%s failed: %s (%s:%d)",    \
                  #expr, hipGetErrorString(e__), __FILE__, __LINE__);                             \
  } while (0)

#define CHK(expr)                     \
  do {                                \
    int rc__ = (expr);                \
    if (rc__ != SSN_OK) return rc__;  \
  } while (0)

struct Buf {
  void* d = nullptr;
  int64_t count = 0;
  int kind = 0;
  int64_t rows = 1, cols = 0, ld = 0;     // device layout (row padded)
  bool shaped = false;
  bool keep = false;                      // keep host copy for reset (state / learned)
  std::vector<double> host;               // initial contents if keep
  // ensemble-array fast path (k_ensarray FAST): decoders neuron-major, (V, R) packed into the V buffer
  int dec_K = 0, dec_dout = 0, dec_n = 0, dec_DP = 0;   // dec_DP > 0: stored [K][ld][DP] instead of [K*dout][ld]
  int packed = 0;                         // 1: holds packed LIF state words; 2: refractory view of buffer `partner`
  int partner = -1;
  // spike-sparse decoders (k_spmv_partial): logical [rows][cols] stored neuron-major [cols][ldt]
  bool transposed = false;
  int64_t ldt = 0;
  // neuron order on the device (whole-block kernel, Sim::reorder_block_neurons): column c of the rows of ensemble k holds the
  // neuron perm[k * cols + c] of the caller's order; perm_rows = rows per ensemble.  Empty: the caller's order.
  std::shared_ptr<std::vector<int32_t>> perm;
  int perm_rows = 1;
};

enum ItemType { IT_PROGRAM = 0, IT_ENS, IT_MATVEC, IT_NEURONS, IT_PES, IT_VOJA, IT_MATVEC_ORDERED, IT_FINISH, IT_SPMV, IT_DFT, IT_VECOPS, IT_GRID_LHS, IT_GRID_GEMM, IT_ARGMAX_PART, IT_ROUND,
                IT_GRID_DOT,        // (a body of the round grid only)
                IT_MATVEC_NEURONS,  // (round plan) a dense population's encoder product with the neuron update in its epilogue
                IT_NONE };          // (round plan) an operator that was folded into another one

}  // namespace

struct ssn_sim {
  virtual ~ssn_sim() {}
  virtual int reset() = 0;
  virtual int set_table(int id, const double* rows, const void* rows_dev, int64_t n_rows, int64_t width,
                        const int32_t* idx, int64_t n_idx, int64_t first_step) = 0;
  virtual int stage_table(int id, const void* rows_t, int64_t n_rows, int64_t width, const int32_t* idx, int64_t n_idx, int64_t first_step) = 0;
  virtual int commit_tables() = 0;
  virtual int reserve_probes(int64_t n) = 0;
  virtual int run_steps(int64_t n, int profile) = 0;
  virtual int run_phase(int phase) = 0;
  virtual int64_t exchange_size() = 0;
  virtual int64_t cycle_len() = 0;
  virtual int exchange_copy(void* buf, bool pack) = 0;
  virtual int phase_async(int phase, void* buf, hipStream_t ext) = 0;
  virtual int phase_sync(hipStream_t ext) = 0;
  virtual int read_probe(int id, double* dst, void* dst_dev, int64_t first, int64_t count) = 0;
  virtual int64_t probe_count(int id) = 0;
  virtual int rw_signal(int64_t off, int64_t count, double* dst, const double* src) = 0;
  virtual int rw_buffer(int id, double* dst, const double* src, int64_t count) = 0;
  virtual int counters(ssn_counters* out) = 0;
  virtual int kernel_times(ssn_kernel_time* out, int capacity) = 0;
  virtual int64_t n_steps() = 0;
};

namespace {

template <typename T>
struct Sim final : ssn_sim {
  using MOp = ssn::MicroOp<T>;
  static constexpr int VW = 16 / sizeof(T);

  struct Item {
    int type = IT_PROGRAM;
    int op_begin = 0, op_count = 0;          // program: range in d_mops
    ssn::EnsArgs<T> ens;
    ssn::FinishArgs<T> fin;
    bool dominant = false;
    // matvec / pes / voja / neurons
    T* Wm = nullptr; const T* src = nullptr; T* dst = nullptr; const T* aux0 = nullptr; const T* aux1 = nullptr;
    const T* aux2 = nullptr; T* V = nullptr; T* R = nullptr;
    int rows = 0, cols = 0, ld = 0, set = 0, n = 0;
    T scalar = 0;
    ssn::NeuronParams<T> np;
    int* list = nullptr; int* count = nullptr;     // spike list (k_neurons_compact -> k_spmv_partial)
    int seg = 0;                                   // > 0: segmented spike list (k_neurons), segments per spmv chunk
    int seg_len = 256;                             // neurons per list segment (16: written by a fused product + neuron update)
    int level = -1;                                // scheduling round of the operator (builder): equal level = independent
    int phase = -1;                                // neuron-sharded models: 0 before the per-timestep exchange, 1 after (set by plan())
    T* lp_dst = nullptr; const T* lp_src = nullptr; T lp_a = 0, lp_b = 0;      // IT_PES: folded filter of the row factors (build_rounds)
    int batch = 1;                                 // this item and the next batch-1 items (same kind, independent) share one launch
    bool merged = false;                           // launched by the item that leads its batch
    ssn::DftArgs dft;
  };

  int device = 0;
  double dt = 0.001;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;          // downloads (see download())
  double* up_stage = nullptr;                 // device staging of upload() (float64 as it arrives from the host)
  int64_t up_cap = 0;
  void* dl_stage = nullptr;                   // pinned host staging of download()
  int64_t dl_cap = 0;
  std::mutex dl_mutex;
  int64_t n_sig = 0;
  T* sig = nullptr;
  std::vector<double> sig_init;
  std::vector<Buf> bufs;
  std::vector<Item> items;                   // one timestep of the core stage, unfused
  std::vector<void*> scratch_bufs;
  // time-batched pre / post stages
  T* bsig = nullptr;                          // [block+1][n_sig]
  int block = 0;                              // timesteps per block (0: staging off)
  bool core_empty = false;                    // no per-timestep work at all
  int flags = 0;                              // ssn_model_desc.flags
  bool fused_core = false;                    // core == one recurrent ensemble array: [k_ensarray, k_ens_finish]
  bool fused_defer = false;                   // ... with the finish deferred into the next step's prologue: [k_ensarray]
  ssn::FinishArgs<T> fin_begin, fin_flush;
  bool fused_block = false;                   // ... stepped a whole block per launch: [k_ens_block] (ssn_block.hpp)
  ssn::BlockArgs<T> blk;
  std::vector<void*> fused_bufs;
  std::set<int> sparse_w;                     // decoder buffers multiplied with a LIF spike vector
  std::set<int> learned_w;                    // matrices a learning rule (PES, Voja) updates
  std::vector<std::pair<int64_t, std::pair<int*, int*>>> spike_lists;   // spike signal offset -> (list, count)
  std::vector<int64_t> seg_spikes;                                       // spike signals whose list is segmented
  std::vector<std::pair<int64_t, float2*>> dft_tables;                  // dft_key(kind, sizes) -> table (twiddles, chirps, spectra, DFT matrices)
  std::vector<ssn::BatchOp<T>> pre_ops, post_ops;
  int batch_skipped = 0;                      // products of the batched stages whose whole input was zero over a block (run_batch)
  std::vector<ssn_range> pre_to_core, core_to_post;
  std::vector<unsigned char> batched_mask;    // signals owned by a batched stage (for ssn_read_signal)
  std::vector<MOp> mops;                     // [head][middle programs...][tail][head copy]
  MOp* d_mops = nullptr;
  std::vector<ssn::ProgDesc> prog_descs;     // one per program (+ a copy of the head behind the tail)
  ssn::ProgDesc* d_progs = nullptr;
  std::vector<std::pair<const void*, std::pair<const int32_t*, int64_t>>> host_idx;   // device idx ptr -> host copy
  int head_begin = 0, head_count = 0, tail_begin = 0, tail_count = 0;
  bool can_fuse = false;
  // generic plan: which earlier items of the timestep each item must wait for (data hazards on signal ranges,
  // buffers and scratch memory) - lets the step graph fork independent branches over several streams
  struct Rng { const void* space; int64_t lo, hi; bool w; };
  std::vector<std::vector<int>> item_deps;
  std::vector<MOp> vecops_host;               // operators of the grid-wide first level (IT_VECOPS), host copy
  std::map<const void*, std::vector<ssn::LinTerm<T>>> lin_terms;   // device term list of a M_LINCOMB operator -> host copy
  // round plan (k_round, ssn_round.hpp): every launch of a timestep, in order
  struct RoundLaunch { ssn::RoundArgs<T> args; int n_blocks = 0; size_t lds = 0; int round = 0; };
  struct Launch { int rl = -1; int item = -1; int phase = 0; };
  bool round_mode = false;
  std::vector<RoundLaunch> round_launches;
  std::vector<Launch> launch_list;            // one timestep
  std::vector<Launch> graph_list;             // steps_per_graph timesteps, software-pipelined (empty: replay launch_list)
  // neuron-sharded models, pipelined (round 4): cycle_steps timesteps as cycle_steps + 1 launch segments; the caller's exchange of
  // timestep k sits between segments k and k + 1 (ssn_run_phase / ssn_phase_async with phase 3 = "the next segment")
  std::vector<std::vector<Launch>> cycle_segs;
  int cycle_steps = 0, next_seg = 0;
  int n_fused_populations = 0, n_serial_chains = 0, n_folded_inputs = 0;      // (counters of the round plan)
  std::vector<hipGraph_t> cycle_graph;
  std::vector<hipGraphExec_t> cycle_exec;
  std::vector<Launch> phase2_list;            // neuron-sharded models: the updates of timestep s and timestep s + 1 up to its exchange,
                                              // planned as ONE set of rounds (empty: the two halves one after the other)
  int graph_rounds = 0;
  std::vector<void*> round_bufs;

  ssn::StepCtx* d_ctx = nullptr;
  std::vector<ssn::TableSlot> tables;
  // host copy of every table's row index per step, and where each table's rows land in the signal vector: a time-batched
  // product whose whole input is a table that is zero for the whole block (the init-SSP input of the path integrator after
  // its first 50 ms, reference run_pathint.py:136) is not multiplied out - its result rows are zero-filled instead
  std::vector<std::vector<int32_t>> table_idx_host;
  std::vector<std::pair<long long, long long>> table_dst;      // (signal offset, width) per table id; width 0: unknown
  ssn::TableSlot* d_tables = nullptr;
  std::vector<void*> table_rows;
  std::vector<int*> table_idx;
  std::vector<int64_t> table_rows_cap, table_idx_cap;
  // staged tables (ssn_stage_table / ssn_commit_tables): the NEXT chunk's inputs arrive in a second set of buffers by DMA on the
  // copy stream while the compute stream still reads the current ones; commit swaps the sets between two runs
  struct Staged { void* rows = nullptr; int* idx = nullptr; int64_t rows_cap = 0, idx_cap = 0; bool valid = false;
                  int64_t n_rows = 0, width = 0, n_idx = 0, first_step = 0; std::vector<int32_t> idx_host; };
  std::vector<Staged> staged;
  std::mutex stage_mutex;
  std::vector<ssn_probe_desc> probes;
  std::vector<ssn::ProbeSlot> pslots;
  ssn::ProbeSlot* d_pslots = nullptr;
  std::vector<int64_t> probe_cap_bytes;
  int64_t reserve_first = 0, reserve_n = 0;
  int steps_per_graph = 0;
  hipGraphExec_t graph_exec = nullptr;
  hipGraph_t graph = nullptr;
  int64_t steps_done = 0;
  int64_t device_bytes = 0;
  // timing
  hipEvent_t ev_run0 = nullptr, ev_run1 = nullptr;
  std::vector<hipEvent_t> ev_pool;
  int64_t dom_launches = 0;
  double dom_ms = 0.0, last_run_ms = 0.0;
  double dom_bytes = 0.0;
  int64_t dom_units = 0;
  int launches_per_step = 0;
  // neuron-sharded model: the timestep's items in two halves around the caller's all-reduce
  std::vector<ssn_range> exchange;
  bool phased = false;
  int next_phase = 0;
  hipGraphExec_t phase_exec[3] = {nullptr, nullptr, nullptr};      // [2]: phase 1 of a timestep followed by phase 0 of the next
  hipGraph_t phase_graph[3] = {nullptr, nullptr, nullptr};
  // stream-ordered stepping (ssn_phase_async): the same halves with the exchange copies inside the graph, for one exchange
  // buffer of the caller; launched on the caller's stream, no host synchronisation per timestep
  hipGraphExec_t async_exec[3] = {nullptr, nullptr, nullptr};
  hipGraph_t async_graph[3] = {nullptr, nullptr, nullptr};
  void* async_buf = nullptr;
  bool async_captured = false, async_active = false;
  hipStream_t async_stream = nullptr;         // the caller's stream a stream-ordered run is in flight on (while async_active)
  static constexpr int N_ITEM_TYPES = 18;
  double type_ms[N_ITEM_TYPES] = {};             // profile = 2: device time per plan-item type
  int64_t type_launches[N_ITEM_TYPES] = {};
  std::vector<double> item_ms;                   // ... and per plan item (printed under SSN_DEBUG_PLAN)
  std::vector<int64_t> item_n;

  ~Sim() override {
    hipSetDevice(device);
    if (async_active && async_stream) hipStreamSynchronize(async_stream);
    if (stream) hipStreamSynchronize(stream);
    if (graph_exec) hipGraphExecDestroy(graph_exec);
    if (graph) hipGraphDestroy(graph);
    for (int h = 0; h < 3; ++h) {
      if (phase_exec[h]) hipGraphExecDestroy(phase_exec[h]);
      if (phase_graph[h]) hipGraphDestroy(phase_graph[h]);
      if (async_exec[h]) hipGraphExecDestroy(async_exec[h]);
      if (async_graph[h]) hipGraphDestroy(async_graph[h]);
    }
    drop_cycle_graphs();
    for (auto& b : bufs) if (b.d) hipFree(b.d);
    for (auto p : table_rows) if (p) hipFree(p);
    for (auto p : table_idx) if (p) hipFree(p);
    for (auto& g : staged) { if (g.rows) hipFree(g.rows); if (g.idx) hipFree(g.idx); }
    for (auto& s : pslots) if (s.data) hipFree(s.data);
    for (auto& it : items) if (it.type == IT_ENS && it.ens.partials) hipFree(it.ens.partials);
    for (auto p : scratch_bufs) if (p) hipFree(p);
    for (auto p : fused_bufs) if (p) hipFree(p);
    for (auto p : round_bufs) if (p) hipFree(p);
    for (auto e : ev_pool) hipEventDestroy(e);
    if (ev_run0) hipEventDestroy(ev_run0);
    if (ev_run1) hipEventDestroy(ev_run1);
    if (sig) hipFree(sig);
    if (bsig) hipFree(bsig);
    if (d_mops) hipFree(d_mops);
    if (d_progs) hipFree(d_progs);
    if (d_ctx) hipFree(d_ctx);
    if (d_tables) hipFree(d_tables);
    if (d_pslots) hipFree(d_pslots);
    if (copy_stream) { hipStreamSynchronize(copy_stream); hipStreamDestroy(copy_stream); }
    if (dl_stage) hipHostFree(dl_stage);
    if (up_stage) hipFree(up_stage);
    if (stream) hipStreamDestroy(stream);
  }

  template <typename P>
  int dmalloc(P** p, int64_t bytes) {
    if (bytes <= 0) bytes = 16;
    HIPCHK(hipMalloc((void**)p, (size_t)bytes));
    device_bytes += bytes;
    return SSN_OK;
  }

  // host double[rows*cols] -> device T[rows*ld]
  int upload(const double* src, T* dst, int64_t rows, int64_t cols, int64_t ld) {
    const int64_t n = rows * cols;
    if (n == 0) return SSN_OK;
    // float64 staging on the device: uploads of up to 64 MB share one buffer that stays (the tables of a chunked run are
    // replaced at every chunk boundary - a hipMalloc / hipFree pair per table there cost more than the copy); larger ones
    // (weight matrices at build time, config 5's 14 GB clean-up table) take and release their own
    constexpr int64_t KEEP = (int64_t)8 << 20;                 // elements
    double* stage = nullptr;
    bool own = false;
    if (n <= KEEP) {
      if (n > up_cap) {
        if (up_stage) { hipFree(up_stage); up_stage = nullptr; up_cap = 0; }
        const int64_t want = std::min<int64_t>(KEEP, std::max<int64_t>(2 * n, 1 << 16));
        HIPCHK(hipMalloc((void**)&up_stage, (size_t)want * sizeof(double)));
        up_cap = want;
      }
      stage = up_stage;
    } else {
      HIPCHK(hipMalloc((void**)&stage, (size_t)n * sizeof(double)));
      own = true;
    }
    hipError_t e = hipMemcpyAsync(stage, src, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = ssn::launch_convert_in<T>(stream, stage, dst, rows, cols, ld);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (own) hipFree(stage);
    HIPCHK(e);
    return SSN_OK;
  }

  // Device -> host as float64.  The values travel in the simulator's own type into a pinned host buffer (a DMA copy on a
  // stream of its own: no kernel, so it does not queue behind a compute kernel that owns every CU, and no pageable
  // staging) and are widened on the host by a few threads.  A caller thread may therefore fetch the probe samples of the
  // timesteps that are done while another thread's ssn_run_steps steps the next ones (simulator.py does, chunk by
  // chunk).  Everything read here was completed by a call that synchronised the compute stream before it returned.
  int download(const T* src, double* dst, int64_t rows, int64_t cols, int64_t ld) {
    const int64_t n = rows * cols;
    if (n == 0) return SSN_OK;
    std::lock_guard<std::mutex> lock(dl_mutex);
    if (!copy_stream) HIPCHK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    if (n > dl_cap) {
      if (dl_stage) { HIPCHK(hipStreamSynchronize(copy_stream)); hipHostFree(dl_stage); dl_stage = nullptr; }
      const int64_t want = std::max<int64_t>(n, 1 << 20);
      HIPCHK(hipHostMalloc((void**)&dl_stage, (size_t)want * sizeof(T), hipHostMallocDefault));
      dl_cap = want;
    }
    T* h = (T*)dl_stage;
    if (ld == cols) HIPCHK(hipMemcpyAsync(h, src, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, copy_stream));
    else HIPCHK(hipMemcpy2DAsync(h, (size_t)cols * sizeof(T), src, (size_t)ld * sizeof(T), (size_t)cols * sizeof(T), (size_t)rows, hipMemcpyDeviceToHost, copy_stream));
    HIPCHK(hipStreamSynchronize(copy_stream));
    auto widen = [h, dst](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) dst[i] = (double)h[i]; };
    const int nt = n >= (1 << 21) ? 4 : 1;
    if (nt == 1) { widen(0, n); return SSN_OK; }
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(widen, n * t / nt, n * (t + 1) / nt);
    widen(0, n / nt);
    for (auto& th : pool) th.join();
    return SSN_OK;
  }

  bool ens_fast(const ssn_op_desc& o) const {
    // spiking LIF with min_voltage == 0: packed state word + (dout >= 3) spike-sparse neuron-major decoders
    return !(flags & 2) && o.i[11] == SSN_LIF && o.f[2] == 0.0;
  }

  bool ens_sparse(const ssn_op_desc& o) const { return ens_fast(o) && !(flags & 4) && o.i[4] >= 3; }

  // host [rows][cols] double -> device layout of buffer b
  int upload_buf(Buf& b, const double* src) {
    std::vector<double> ordered;
    if (b.perm) {                      // the device's neuron order (columns permuted within every ensemble's rows)
      ordered.resize((size_t)(b.rows * b.cols));
      const int32_t* pm = b.perm->data();
      for (int64_t r = 0; r < b.rows; ++r) {
        const int32_t* pk = pm + (r / b.perm_rows) * b.cols;
        const double* in = src + r * b.cols;
        double* out = ordered.data() + r * b.cols;
        for (int64_t c = 0; c < b.cols; ++c) out[c] = in[pk[c]];
      }
      src = ordered.data();
    }
    if (b.transposed) {
      T* tmp = nullptr;
      HIPCHK(hipMalloc((void**)&tmp, (size_t)(b.rows * b.ld) * sizeof(T)));
      int rc = upload(src, tmp, b.rows, b.cols, b.ld);
      hipError_t e = hipSuccess;
      if (rc == SSN_OK) e = hipMemsetAsync(b.d, 0, (size_t)(b.cols * b.ldt) * sizeof(T), stream);
      if (rc == SSN_OK && e == hipSuccess) e = ssn::launch_transpose<T>(stream, tmp, (T*)b.d, (int)b.rows, (int)b.cols, (int)b.ld, (int)b.ldt);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      hipFree(tmp);
      CHK(rc);
      HIPCHK(e);
      return SSN_OK;
    }
    if (!b.dec_DP) return upload(src, (T*)b.d, b.rows, b.cols, b.ld);
    T* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, (size_t)(b.rows * b.ld) * sizeof(T)));
    int rc = upload(src, tmp, b.rows, b.cols, b.ld);
    hipError_t e = hipSuccess;
    if (rc == SSN_OK) e = ssn::launch_dec_pack<T>(stream, tmp, (T*)b.d, b.dec_K, b.dec_dout, b.dec_n, (int)b.ld, b.dec_DP, 0);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(tmp);
    CHK(rc);
    HIPCHK(e);
    return SSN_OK;
  }

  int download_buf(Buf& b, double* dst) {
    if (b.perm) {                      // back to the caller's neuron order
      std::shared_ptr<std::vector<int32_t>> pm = b.perm;
      b.perm.reset();
      std::vector<double> dev((size_t)(b.rows * b.cols));
      const int rc = download_buf(b, dev.data());
      b.perm = pm;
      CHK(rc);
      for (int64_t r = 0; r < b.rows; ++r) {
        const int32_t* pk = pm->data() + (r / b.perm_rows) * b.cols;
        const double* in = dev.data() + r * b.cols;
        double* out = dst + r * b.cols;
        for (int64_t c = 0; c < b.cols; ++c) out[pk[c]] = in[c];
      }
      return SSN_OK;
    }
    if (b.transposed) {
      T* tmp = nullptr;
      HIPCHK(hipMalloc((void**)&tmp, (size_t)(b.rows * b.ld) * sizeof(T)));
      hipError_t e = ssn::launch_transpose<T>(stream, (const T*)b.d, tmp, (int)b.cols, (int)b.rows, (int)b.ldt, (int)b.ld);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);          // (download() copies on its own stream)
      int rc = e == hipSuccess ? download(tmp, dst, b.rows, b.cols, b.ld) : SSN_OK;
      hipFree(tmp);
      HIPCHK(e);
      return rc;
    }
    if (!b.dec_DP && !b.packed) return download((const T*)b.d, dst, b.rows, b.cols, b.ld);
    T* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, (size_t)(b.rows * b.ld) * sizeof(T)));
    hipError_t e = hipSuccess;
    if (b.dec_DP) {
      e = hipMemsetAsync(tmp, 0, (size_t)(b.rows * b.ld) * sizeof(T), stream);
      if (e == hipSuccess) e = ssn::launch_dec_pack<T>(stream, (const T*)b.d, tmp, b.dec_K, b.dec_dout, b.dec_n, (int)b.ld, b.dec_DP, 1);
    } else {
      const T* words = (const T*)(b.packed == 2 ? bufs[b.partner].d : b.d);
      // (the f32 whole-block kernel keeps -(R - dt) in the state word of a refractory neuron, every other kernel -R)
      e = ssn::launch_state_unpack<T>(stream, words, tmp, b.rows * b.ld, b.packed == 2, (fused_block && sizeof(T) == 4) ? (T)dt : T(0));
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);            // (download() copies on its own stream)
    int rc = e == hipSuccess ? download(tmp, dst, b.rows, b.cols, b.ld) : SSN_OK;
    hipFree(tmp);
    HIPCHK(e);
    return rc;
  }

  int shape(int id, int64_t rows, int64_t cols, bool pad, bool keep) {
    if (id < 0 || id >= (int)bufs.size()) return fail(SSN_EINVAL, "buffer id %d out of range", id);
    Buf& b = bufs[id];
    if (sparse_w.count(id)) { b.transposed = true; b.ldt = (rows + VW - 1) / VW * VW; pad = true; }
    if (b.kind != SSN_BUF_REAL) return fail(SSN_EINVAL, "buffer %d is not a real buffer", id);
    if (rows * cols != b.count) return fail(SSN_EINVAL, "buffer %d has %lld elements, operator expects %lld x %lld",
                                            id, (long long)b.count, (long long)rows, (long long)cols);
    const int64_t ld = pad ? (cols + VW - 1) / VW * VW : cols;
    if (b.shaped && (b.rows != rows || b.cols != cols || b.ld != ld))
      return fail(SSN_EINVAL, "buffer %d is used with two different layouts", id);
    b.rows = rows; b.cols = cols; b.ld = ld; b.shaped = true;
    b.keep = b.keep || keep;
    return SSN_OK;
  }

  static bool is_micro(const ssn_op_desc& o) {
    switch (o.kind) {
      case SSN_OP_FILL: case SSN_OP_TABLE: case SSN_OP_AXPY: case SSN_OP_LOWPASS: case SSN_OP_GATE: case SSN_OP_LINCOMB: return true;
      case SSN_OP_MATVEC: return o.i[3] <= 16 && o.i[2] <= 8192;
      default: return false;
    }
  }

  int check_range(int64_t off, int64_t len, const char* what) {
    if (off < 0 || len < 0 || off + len > n_sig) return fail(SSN_EINVAL, "%s range [%lld,+%lld) outside the %lld signals",
                                                             what, (long long)off, (long long)len, (long long)n_sig);
    return SSN_OK;
  }

  int create(const ssn_model_desc* m) {
    device = m->device;
    dt = m->dt;
    flags = m->flags;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&ev_run0));
    HIPCHK(hipEventCreate(&ev_run1));
    n_sig = m->n_signals;
    if (n_sig >= (1LL << 31)) return fail(SSN_EINVAL, "%lld signals: the vector operators index with 32 bits", (long long)n_sig);
    sig_init.assign(m->signal_init, m->signal_init + n_sig);
    CHK(dmalloc(&sig, (n_sig + 8) * (int64_t)sizeof(T)));
    bufs.resize(m->n_buffers);
    for (int i = 0; i < m->n_buffers; ++i) {
      bufs[i].count = m->buffers[i].count;
      bufs[i].kind = m->buffers[i].kind;
      bufs[i].cols = bufs[i].ld = bufs[i].count;
      if (!m->buffers[i].data && bufs[i].count) return fail(SSN_EINVAL, "buffer %d has no data", i);
    }
    for (int i = 0; i < m->n_ops; ++i)
      if (m->ops[i].kind == SSN_OP_PES || m->ops[i].kind == SSN_OP_VOJA) learned_w.insert((int)m->ops[i].i[0]);
    // decoders applied to the spike vector of a dense LIF ensemble are kept neuron-major and multiplied sparsely
    if (!(flags & 8))
      for (int i = 0; i < m->n_ops; ++i) {
        const ssn_op_desc& o = m->ops[i];
        if (o.kind != SSN_OP_MATVEC || o.stage != 1 || is_micro(o) || o.i[3] < 128) continue;
        if (o.i[3] > 15000 && (flags & 1024)) continue;      // (a spike list rebuilt inside k_spmv_partial lives in LDS)
        for (int j = 0; j < m->n_ops; ++j) {
          const ssn_op_desc& q = m->ops[j];
          if (q.kind == SSN_OP_NEURONS && q.i[5] == SSN_LIF && q.i[1] == o.i[1] && q.i[2] == o.i[3]) sparse_w.insert((int)o.i[4]);
        }
      }
    for (int i = 0; i < m->n_ops; ++i) {          // a buffer also used as a dense operand elsewhere stays dense
      const ssn_op_desc& o = m->ops[i];
      if (o.kind == SSN_OP_MATVEC && sparse_w.count((int)o.i[4])) {
        bool ok = false;
        for (int j = 0; j < m->n_ops; ++j) {
          const ssn_op_desc& q = m->ops[j];
          if (q.kind == SSN_OP_NEURONS && q.i[5] == SSN_LIF && q.i[1] == o.i[1] && q.i[2] == o.i[3] && o.stage == 1 && !is_micro(o)) ok = true;
        }
        if (!ok) sparse_w.erase((int)o.i[4]);
      }
      if ((o.kind == SSN_OP_VOJA || o.kind == SSN_OP_CLEANUP) && sparse_w.count((int)o.i[o.kind == SSN_OP_VOJA ? 0 : 4])) sparse_w.erase((int)o.i[o.kind == SSN_OP_VOJA ? 0 : 4]);
    }
    // pass 1: buffer layouts
    for (int i = 0; i < m->n_ops; ++i) {
      const ssn_op_desc& o = m->ops[i];
      switch (o.kind) {
        case SSN_OP_ENSARRAY: {
          const int64_t K = o.i[1], n = o.i[2], din = o.i[3], dout = o.i[4];
          if (din < 1 || din > 4 || dout < 1 || dout > 8)
            return fail(SSN_EUNSUPPORTED, "ensemble array with din=%lld dout=%lld (kernels cover din<=4, dout<=8)",
                        (long long)din, (long long)dout);
          CHK(shape((int)o.i[5], K * din, n, true, false));
          CHK(shape((int)o.i[6], K, n, true, false));
          CHK(shape((int)o.i[7], K * dout, n, true, false));
          CHK(shape((int)o.i[9], K, n, true, true));
          CHK(shape((int)o.i[10], K, n, true, true));
          if (ens_fast(o)) {
            Buf& vb = bufs[o.i[9]]; Buf& rb = bufs[o.i[10]]; Buf& db = bufs[o.i[7]];
            if (vb.packed == 2 || rb.packed == 1 || o.i[9] == o.i[10]) return fail(SSN_EINVAL, "ensemble array state buffers shared inconsistently");
            vb.packed = 1; rb.packed = 2; rb.partner = (int)o.i[9];
            if (ens_sparse(o)) { db.dec_K = (int)K; db.dec_dout = (int)dout; db.dec_n = (int)n; db.dec_DP = dout <= 4 ? 4 : 8; }
          }
          if (o.i[8] < 0 || o.i[8] >= m->n_buffers || bufs[o.i[8]].kind != SSN_BUF_I32 || bufs[o.i[8]].count != K * dout)
            return fail(SSN_EINVAL, "ensemble array dst_idx buffer must be int32 [K*dout]");
          const int32_t* di = (const int32_t*)m->buffers[o.i[8]].data;
          for (int64_t j = 0; j < K * dout; ++j)
            if (di[j] < 0 || di[j] >= n_sig) return fail(SSN_EINVAL, "ensemble array destination index out of range");
          CHK(check_range(o.i[0], K * din, "ensarray x"));
          break;
        }
        case SSN_OP_MATVEC:
          // (a matrix that a learning rule updates keeps that rule's padded rows also when its product is small enough to run as a
          //  glue micro-operator - those read it with the buffer's leading dimension: found by the random-network tests, round 4)
          CHK(shape((int)o.i[4], o.i[2], o.i[3], !is_micro(o) || learned_w.count((int)o.i[4]) != 0, false));
          CHK(check_range(o.i[0], o.i[2], "matvec dst"));
          CHK(check_range(o.i[1], o.i[3], "matvec src"));
          break;
        case SSN_OP_NEURONS:
          CHK(shape((int)o.i[3], 1, o.i[2], false, true));
          CHK(shape((int)o.i[4], 1, o.i[2], false, true));
          CHK(check_range(o.i[0], o.i[2], "neurons J"));
          CHK(check_range(o.i[1], o.i[2], "neurons out"));
          break;
        case SSN_OP_PES:
          CHK(shape((int)o.i[0], o.i[1], o.i[2], true, true));
          CHK(check_range(o.i[3], o.i[1], "pes err"));
          CHK(check_range(o.i[4], o.i[2], "pes act"));
          break;
        case SSN_OP_VOJA:
          CHK(shape((int)o.i[0], o.i[1], o.i[2], true, true));
          CHK(shape((int)o.i[6], 1, o.i[1], false, false));
          CHK(check_range(o.i[3], o.i[1], "voja spikes"));
          CHK(check_range(o.i[4], o.i[2], "voja key"));
          CHK(check_range(o.i[5], 1, "voja learn"));
          break;
        case SSN_OP_CLEANUP:
          CHK(shape((int)o.i[4], o.i[2], o.i[3], true, false));
          if (o.i[5] > 0) {       // factor tables of the sample grid: dft (2K x d), lhs (n_a x 2K), rhs (N x 2K), n_a * N = rows
            if (o.i[8] <= 0 || o.i[9] <= 0 || o.i[10] <= 0 || (o.i[10] & 1) || o.i[8] * o.i[9] != o.i[2])
              return fail(SSN_EINVAL, "cleanup grid factors %lld x %lld (2K = %lld) do not match the %lld-row table",
                          (long long)o.i[8], (long long)o.i[9], (long long)o.i[10], (long long)o.i[2]);
            CHK(shape((int)o.i[5] - 1, o.i[10], o.i[3], true, false));
            CHK(shape((int)o.i[6] - 1, o.i[8], o.i[10], true, false));
            CHK(shape((int)o.i[7] - 1, o.i[9], o.i[10], true, false));
          }
          CHK(check_range(o.i[0], o.i[3], "cleanup dst"));
          CHK(check_range(o.i[1], o.i[3], "cleanup src"));
          break;
        case SSN_OP_FILL: CHK(check_range(o.i[0], o.i[1], "fill")); break;
        case SSN_OP_TABLE:
          CHK(check_range(o.i[0], o.i[1], "table"));
          if (o.i[2] < 0 || o.i[2] >= m->n_tables) return fail(SSN_EINVAL, "table id out of range");
          break;
        case SSN_OP_AXPY: CHK(check_range(o.i[0], o.i[2], "axpy dst")); CHK(check_range(o.i[1], o.i[2], "axpy src")); break;
        case SSN_OP_LOWPASS: CHK(check_range(o.i[0], o.i[2], "lowpass dst")); CHK(check_range(o.i[1], o.i[2], "lowpass src")); break;
        case SSN_OP_GATE: CHK(check_range(o.i[0], o.i[2], "gate dst")); CHK(check_range(o.i[1], 2 * o.i[2] + 1, "gate src")); break;
        case SSN_OP_LINCOMB: {
          CHK(check_range(o.i[0], o.i[1], "lincomb dst"));
          if (o.stage != 1) return fail(SSN_EUNSUPPORTED, "lincomb operators belong to the per-timestep core");
          if (o.i[2] < 0 || o.i[3] < 0 || o.i[3] >= m->n_buffers || o.i[4] < 0 || o.i[4] >= m->n_buffers) return fail(SSN_EINVAL, "lincomb term buffers out of range");
          const ssn_buffer_desc& sb = m->buffers[o.i[3]]; const ssn_buffer_desc& ab = m->buffers[o.i[4]];
          if (sb.kind != SSN_BUF_I32 || ab.kind != SSN_BUF_REAL || sb.count != o.i[2] || ab.count != o.i[2]) return fail(SSN_EINVAL, "lincomb term buffers: int32 offsets and real coefficients of n_terms elements each");
          for (int64_t q = 0; q < o.i[2]; ++q) CHK(check_range(((const int32_t*)sb.data)[q], o.i[1], "lincomb src"));
          break;
        }
        default: return fail(SSN_EINVAL, "unknown operator kind %d", o.kind);
      }
    }
    // uploads
    for (int i = 0; i < m->n_buffers; ++i) {
      Buf& b = bufs[i];
      if (b.kind == SSN_BUF_I32) {
        CHK(dmalloc((int32_t**)&b.d, b.count * 4));
        HIPCHK(hipMemcpy(b.d, m->buffers[i].data, (size_t)b.count * 4, hipMemcpyHostToDevice));
      } else {
        const int64_t elems = b.dec_DP ? (int64_t)b.dec_K * b.ld * b.dec_DP : (b.transposed ? b.cols * b.ldt : b.rows * b.ld);
        CHK(dmalloc((T**)&b.d, elems * (int64_t)sizeof(T)));
        CHK(upload_buf(b, (const double*)m->buffers[i].data));
        if (b.keep) b.host.assign((const double*)m->buffers[i].data, (const double*)m->buffers[i].data + b.count);
      }
    }
    CHK(upload(sig_init.data(), sig, 1, n_sig, n_sig));
    CHK(dmalloc(&d_ctx, sizeof(ssn::StepCtx)));
    HIPCHK(hipMemset(d_ctx, 0, sizeof(ssn::StepCtx)));
    // tables and probes
    tables.assign(m->n_tables, ssn::TableSlot{nullptr, nullptr, 0, 0, 0, 0});
    table_rows.assign(m->n_tables, nullptr);
    table_idx.assign(m->n_tables, nullptr);
    table_rows_cap.assign(m->n_tables, 0);
    table_idx_cap.assign(m->n_tables, 0);
    staged.resize((size_t)m->n_tables);
    CHK(dmalloc(&d_tables, std::max<int64_t>(1, m->n_tables) * (int64_t)sizeof(ssn::TableSlot)));
    if (m->n_tables) HIPCHK(hipMemcpy(d_tables, tables.data(), tables.size() * sizeof(ssn::TableSlot), hipMemcpyHostToDevice));
    probes.assign(m->probes, m->probes + m->n_probes);
    for (auto& p : probes) {
      if (p.every < 1) return fail(SSN_EINVAL, "probe 'every' must be >= 1");
      CHK(check_range(p.src, p.width, "probe"));
    }
    pslots.assign(m->n_probes, ssn::ProbeSlot{nullptr, 1, 0, 0});
    probe_cap_bytes.assign(m->n_probes, 0);
    for (int i = 0; i < m->n_probes; ++i) pslots[i].every = probes[i].every;
    CHK(dmalloc(&d_pslots, std::max<int64_t>(1, m->n_probes) * (int64_t)sizeof(ssn::ProbeSlot)));
    if (m->n_probes) HIPCHK(hipMemcpy(d_pslots, pslots.data(), pslots.size() * sizeof(ssn::ProbeSlot), hipMemcpyHostToDevice));
    exchange.assign(m->exchange, m->exchange + (m->n_exchange > 0 ? m->n_exchange : 0));
    phased = !exchange.empty();
    for (auto& r : exchange) CHK(check_range(r.lo, r.hi - r.lo, "exchange"));
    // time-batched stages
    bool staged = false;
    for (int i = 0; i < m->n_ops; ++i) staged = staged || m->ops[i].stage != 1;
    for (auto& p : probes) staged = staged || p.stage != 1;
    batched_mask.assign((size_t)n_sig, 0);
    if (staged && phased) return fail(SSN_EINVAL, "a neuron-sharded model is stepped whole (build it unstaged)");
    if (staged) {
      // default: 1024 timesteps per block while the block buffer stays under 256 MiB, else 256
      block = m->block_steps > 0 ? m->block_steps : ((int64_t)1025 * n_sig * (int64_t)sizeof(T) <= (256ll << 20) ? 1024 : 256);
      CHK(dmalloc(&bsig, (int64_t)(block + 1) * n_sig * (int64_t)sizeof(T)));
      CHK(init_bsig());
      pre_to_core.assign(m->pre_to_core, m->pre_to_core + m->n_pre_to_core);
      core_to_post.assign(m->core_to_post, m->core_to_post + m->n_core_to_post);
      for (auto& r : pre_to_core) CHK(check_range(r.lo, r.hi - r.lo, "pre->core boundary"));
      for (auto& r : core_to_post) CHK(check_range(r.lo, r.hi - r.lo, "core->post boundary"));
    }
    // default: 64 timesteps per step graph where the time-batched blocks hold whole graphs (the software-pipelined round plan
    // fills and drains once per graph: SLAM config 3 106.3 us per timestep at 16, 105.9 at 32, 104.5 at 64 - round 4), else 16
    steps_per_graph = m->steps_per_graph != 0 ? m->steps_per_graph : ((block == 0 || block % 64 == 0) ? 64 : 16);
    if (steps_per_graph > 128) return fail(SSN_EINVAL, "steps_per_graph %d: at most 128 (a glue block carries its timestep offset in 8 signed bits)", steps_per_graph);
    CHK(plan(m));
    CHK(capture());
    HIPCHK(hipStreamSynchronize(stream));
    return SSN_OK;
  }

  // ---- planning ---------------------------------------------------------------------------
  ssn::NeuronParams<T> neuron_params(int64_t type, const double* f) const {
    ssn::NeuronParams<T> np;
    np.type = (int)type; np.dt = (T)dt; np.tau_rc = (T)f[0]; np.tau_ref = (T)f[1]; np.min_voltage = (T)f[2];
    return np;
  }

  void ens_chunking(ssn::EnsArgs<T>& a, int min_wgs = 4096) {
    // whole 256-thread sweeps per workgroup; keep >= ~4096 workgroups chip-wide while sweeps can grow (2048 inside a
    // round's grid, where other operators fill the chip as well: two sweeps per workgroup - the second streams in under
    // the first one's decoder gather - measured 148 -> 144 us per timestep at SLAM config 3)
    const int n_vec = a.n_pad / VW;
    int sweeps = 1;
    const int max_sweeps = (n_vec + 255) / 256;
    while (sweeps < max_sweeps && (int64_t)a.K * ((n_vec + 256 * (sweeps + 1) - 1) / (256 * (sweeps + 1))) >= min_wgs) ++sweeps;
    if (const char* env = getenv("SSN_ENS_SWEEPS")) sweeps = std::max(1, std::min(max_sweeps, atoi(env)));   // tuning knob
    a.chunk_vec = 256 * sweeps;
    a.P = (n_vec + a.chunk_vec - 1) / a.chunk_vec;
  }

  void fill_ens_args(const ssn_op_desc& o, ssn::EnsArgs<T>& a) {
    a = ssn::EnsArgs<T>{};
    a.K = (int)o.i[1]; a.n = (int)o.i[2]; a.din = (int)o.i[3]; a.dout = (int)o.i[4];
    a.n_pad = (int)bufs[o.i[5]].ld;
    a.enc = (const T*)bufs[o.i[5]].d; a.bias = (const T*)bufs[o.i[6]].d; a.dec = (const T*)bufs[o.i[7]].d;
    a.V = (T*)bufs[o.i[9]].d; a.R = (T*)bufs[o.i[10]].d;
    a.sig = sig; a.x_off = o.i[0];
    a.np = neuron_params(o.i[11], o.f);
    a.xrows = nullptr; a.n_sig = n_sig; a.ctx = d_ctx; a.n_rec = 0;
    a.fast = ens_fast(o) ? (ens_sparse(o) ? 1 : 2) : 0;
    a.defer = 0; a.sub = 0; a.partials_stride = 0;
    ens_chunking(a);
  }

  // FFT plan of a DFT-structured matvec: radices (<= 32 each; runs of small primes are merged into radices <= 16, one
  // stage and one workgroup barrier each) and the twiddle table.  A length with a prime factor > 32 (97, 1801,
  // 2049 = 3 * 683) is planned as Bluestein's convolution of smooth length M >= 2d - 1.  N = 0 when no plan fits
  // (LDS: 24 B per point of the longest transform, 32-bit index products) - the matrix is used then.
  static std::vector<int> smooth_radices(int L) {
    std::vector<int> primes;
    int rest = L, twos = 0;
    while (rest % 2 == 0) { ++twos; rest /= 2; }
    for (int p = 3; p <= 32 && rest > 1; ++p)
      while (rest % p == 0) { primes.push_back(p); rest /= p; }
    if (rest != 1) return {};
    std::sort(primes.begin(), primes.end());
    std::vector<int> rad;
    // factors of two: radix 8 / 4 / 2 passes run as in-register FFT butterflies (dft_pass_small) - eights, with a remainder
    // of 2^1 as one radix-2 pass and of 2^4 as two fours
    while (twos >= 3 && twos != 4) { rad.push_back(8); twos -= 3; }
    while (twos >= 2) { rad.push_back(4); twos -= 2; }
    if (twos) rad.push_back(2);
    for (size_t i = 0; i < primes.size();) {           // greedy merge of ascending odd primes while the product stays <= 16
      int r = primes[i++];
      while (i < primes.size() && r * primes[i] <= 16) r *= primes[i++];
      rad.push_back(r);
    }
    std::sort(rad.rbegin(), rad.rend());
    return rad;
  }
  // relative cost of the Stockham passes of a length: a generic radix-r pass does r complex multiply-adds per point out of
  // LDS, an in-register one (2, 4, 8) reads and writes each point once
  static double stockham_cost(int L) {
    const std::vector<int> rad = smooth_radices(L);
    if (rad.empty() || rad.size() > 12) return -1.0;
    double c = 0.0;
    for (int r : rad) c += (r == 8 || r == 4 || r == 2) ? 1.0 : 0.25 * r + 0.5;
    return c * L;
  }

  // Injective key of a table: what it is (kind) and every size its contents depend on - a network with circular convolutions
  // of two different lengths must never be handed the other length's table.
  enum { DK_TWIDDLE = 1, DK_CHIRP = 2, DK_CHIRP_SPECTRUM = 3, DK_CHIRP_SPECTRUM_INPLACE = 4, DK_DFT4_G1 = 5, DK_DFT4_G2 = 6 };
  static int64_t dft_key(int kind, int a, int b = 0) { return ((int64_t)kind << 56) | ((int64_t)(uint32_t)a << 28) | (int64_t)(uint32_t)b; }
  int dft_table(int64_t key, const std::vector<float2>& h, float2** out) {
    for (auto& t : dft_tables) if (t.first == key) { *out = t.second; return SSN_OK; }
    float2* d = nullptr;
    CHK(dmalloc(&d, (int64_t)h.size() * (int64_t)sizeof(float2)));
    HIPCHK(hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
    scratch_bufs.push_back(d);
    dft_tables.push_back({key, d});
    *out = d;
    return SSN_OK;
  }

  static std::vector<float2> twiddles(int L) {
    std::vector<float2> h((size_t)L);
    for (int k = 0; k < L; ++k) {
      const double ang = -2.0 * M_PI * (double)k / (double)L;
      h[(size_t)k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    return h;
  }

  // ---- four-step transform on the matrix cores (dft4_fft): the split L = N1 * N2 and the two DFT matrices in MFMA lane order ----
  // cost of a split in MFMA instructions (16 x 16 x 4, two per k step: real and imaginary rows / columns)
  static long long dft4_cost(int N1, int N2, bool complex_in) {
    const long long P1 = (N1 + 15) / 16, C1 = (N2 + 15) / 16, N1p = (N1 + 3) / 4 * 4, N2p = (N2 + 3) / 4 * 4;
    return P1 * C1 * ((complex_in ? 2 : 1) * N1p / 4) * 2 + P1 * C1 * (2 * N2p / 4) * 2;
  }
  // best (N1, N2) with N1 * N2 = L, N2 <= 192 (the step-3 matrix is (2 N2)^2 floats); {0, 0} if L has no such divisor pair
  static std::pair<int, int> dft4_split(int L, bool complex_in) {
    std::pair<int, int> best{0, 0};
    long long bc = -1;
    for (int n1 = 1; n1 <= L; ++n1) {
      if (L % n1) continue;
      const int n2 = L / n1;
      if (n2 > 192 || n1 > 192) continue;
      const long long c = dft4_cost(n1, n2, complex_in);
      if (bc < 0 || c < bc) { bc = c; best = {n1, n2}; }
    }
    return best;
  }
  int dft4_tables(int N1, int N2, const float** g1_out, const float** g2_out) {
    const int64_t key1 = dft_key(DK_DFT4_G1, N1, N2), key2 = dft_key(DK_DFT4_G2, N1, N2);
    for (auto& t : dft_tables) if (t.first == key1) *g1_out = (const float*)t.second;
    for (auto& t : dft_tables) if (t.first == key2) *g2_out = (const float*)t.second;
    if (*g1_out && *g2_out) return SSN_OK;
    const int P1 = (N1 + 15) / 16, Q2 = (N2 + 15) / 16, N1p = (N1 + 3) / 4 * 4, N2p = (N2 + 3) / 4 * 4;
    const int KS1 = 2 * N1p / 4, KS3 = 2 * N2p / 4;
    std::vector<float> g1((size_t)P1 * 2 * KS1 * 64, 0.0f), g2((size_t)Q2 * 2 * KS3 * 64, 0.0f);
    for (int p = 0; p < P1; ++p)
      for (int part = 0; part < 2; ++part)
        for (int ks = 0; ks < KS1; ++ks)
          for (int l = 0; l < 64; ++l) {
            const int k1 = 16 * p + (l & 15), k = 4 * ks + (l >> 4);
            const int n1 = k < N1p ? k : k - N1p;
            if (k1 >= N1 || n1 >= N1) continue;
            const double th = 2.0 * M_PI * (double)(((long long)k1 * n1) % N1) / (double)N1;
            // F = cos - i sin: re out = cos xr + sin xi, im out = -sin xr + cos xi
            const double v = part == 0 ? (k < N1p ? std::cos(th) : std::sin(th)) : (k < N1p ? -std::sin(th) : std::cos(th));
            g1[(((size_t)p * 2 + part) * KS1 + ks) * 64 + l] = (float)v;
          }
    for (int q = 0; q < Q2; ++q)
      for (int part = 0; part < 2; ++part)
        for (int ks = 0; ks < KS3; ++ks)
          for (int l = 0; l < 64; ++l) {
            const int k2 = 16 * q + (l & 15), k = 4 * ks + (l >> 4);
            const int n2 = k < N2p ? k : k - N2p;
            if (k2 >= N2 || n2 >= N2) continue;
            const double ph = 2.0 * M_PI * (double)(((long long)k2 * n2) % N2) / (double)N2;
            // Xr = Br cos + Bi sin, Xi = -Br sin + Bi cos
            const double v = part == 0 ? (k < N2p ? std::cos(ph) : std::sin(ph)) : (k < N2p ? -std::sin(ph) : std::cos(ph));
            g2[(((size_t)q * 2 + part) * KS3 + ks) * 64 + l] = (float)v;
          }
    float* d1 = nullptr; float* d2 = nullptr;
    CHK(dmalloc(&d1, (int64_t)g1.size() * 4));
    CHK(dmalloc(&d2, (int64_t)g2.size() * 4));
    HIPCHK(hipMemcpy(d1, g1.data(), g1.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d2, g2.data(), g2.size() * 4, hipMemcpyHostToDevice));
    scratch_bufs.push_back(d1); scratch_bufs.push_back(d2);
    dft_tables.push_back({key1, (float2*)d1}); dft_tables.push_back({key2, (float2*)d2});
    *g1_out = d1; *g2_out = d2;
    return SSN_OK;
  }

  int plan_dft(const ssn_op_desc& o, ssn::DftArgs* a) {
    const int kind = (int)o.i[6];
    const int d = kind == 5 ? (int)o.i[2] : (int)o.i[3];
    *a = ssn::DftArgs{};
    if (d < 8 || d > 6400) return SSN_OK;
    if (flags & 536870912) {
      // opt-in (round 3, measured slower than the Stockham passes as built - DESIGN.md): the four-step transform on the matrix
      // cores; a length without a usable divisor pair (a prime above 192, or 2 x such a prime ...) goes through Bluestein's
      // convolution of the cheapest length M in [2d - 1, 2d - 1 + 12 %]
      std::pair<int, int> sp = dft4_split(d, kind == 5);
      int M = 0;
      if (!sp.first) {
        long long bc = -1;
        for (int c = 2 * d - 1; c <= 6400 && c <= (2 * d - 1) + (2 * d - 1) / 8; ++c) {
          const std::pair<int, int> s2 = dft4_split(c, true);
          if (!s2.first) continue;
          const long long cost = dft4_cost(s2.first, s2.second, true);
          if (bc < 0 || cost < bc) { bc = cost; M = c; sp = s2; }
        }
      }
      if (sp.first) {
        const int L = M > 0 ? M : d;
        float2* tw = nullptr;
        CHK(dft_table(dft_key(DK_TWIDDLE, L), twiddles(L), &tw));
        if (M > 0) CHK(bluestein_tables(d, M, a));
        const float* g1 = nullptr; const float* g2 = nullptr;
        CHK(dft4_tables(sp.first, sp.second, &g1, &g2));
        a->src = (const float*)(sig + o.i[1]); a->dst = (float*)(sig + o.i[0]); a->tw = tw; a->N = d; a->kind = kind;
        a->set = (int)o.i[5]; a->nr = 0; a->N1 = sp.first; a->N2 = sp.second; a->g1 = g1; a->g2 = g2;
        return SSN_OK;
      }
    }
    std::vector<int> rad = smooth_radices(d);
    int L = d, M = 0;
    if (rad.empty()) {
      // the cheapest smooth length in [2d - 1, 2 (2d - 1)] (a power of two where one fits: four radix-8 passes at 4096)
      double best = -1.0;
      for (int c = 2 * d - 1; c <= 6400 && c <= 2 * (2 * d - 1); ++c) {
        const double cost = stockham_cost(c);
        if (cost > 0.0 && (best < 0.0 || cost < best)) { best = cost; M = c; }
      }
      if (best < 0.0) return SSN_OK;
      rad = smooth_radices(M);
      L = M;
    }
    if (rad.empty() || rad.size() > 12) return SSN_OK;
    float2* tw = nullptr;
    CHK(dft_table(dft_key(DK_TWIDDLE, L), twiddles(L), &tw));
    bool small_only = M > 0 && !getenv("SSN_DFT_NO_INPLACE");          // every pass an in-register one: both transforms in place
    for (int r : rad) small_only = small_only && (r == 8 || r == 4 || r == 2);
    if (M > 0) CHK(bluestein_tables(d, M, a, small_only ? &rad : nullptr));
    a->src = (const float*)(sig + o.i[1]); a->dst = (float*)(sig + o.i[0]); a->tw = tw; a->N = d; a->kind = kind;
    a->set = (int)o.i[5]; a->nr = (int)rad.size();
    for (size_t i = 0; i < rad.size(); ++i) a->radix[i] = rad[i];
    return SSN_OK;
  }

  // position of frequency k after the in-place decimation-in-frequency passes with radices rad[0], rad[1], ...
  static int dif_position(int k, int M, const std::vector<int>& rad) {
    int pos = 0, len = M;
    for (int r : rad) { len /= r; pos += (k % r) * len; k /= r; }
    return pos;
  }
  int bluestein_tables(int d, int M, ssn::DftArgs* a, const std::vector<int>* inplace_radices = nullptr) {
    {
      // chirp w_n = exp(-i pi n^2 / d) with the phase reduced exactly (n^2 mod 2d), and the spectrum of its wrapped conjugate
      std::vector<float2> w((size_t)d);
      std::vector<double> br((size_t)M, 0.0), bi((size_t)M, 0.0);
      for (int n = 0; n < d; ++n) {
        const long long q = ((long long)n * n) % (2LL * d);
        const double ang = -M_PI * (double)q / (double)d;
        w[(size_t)n] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        br[(size_t)n] = std::cos(ang); bi[(size_t)n] = -std::sin(ang);
        if (n) { br[(size_t)(M - n)] = std::cos(ang); bi[(size_t)(M - n)] = -std::sin(ang); }
      }
      std::vector<float2> fb((size_t)M);
      std::vector<double> cs((size_t)M), sn((size_t)M);
      for (int k = 0; k < M; ++k) { cs[(size_t)k] = std::cos(-2.0 * M_PI * k / M); sn[(size_t)k] = std::sin(-2.0 * M_PI * k / M); }
      std::vector<int> nzn;
      for (int n = 0; n < M; ++n) if (br[(size_t)n] != 0.0 || bi[(size_t)n] != 0.0) nzn.push_back(n);
      for (int k = 0; k < M; ++k) {                     // plain O(M d) double DFT, once per length at build time (M <= 6400)
        double sr = 0.0, si = 0.0;
        for (int n : nzn) {
          const size_t q = (size_t)(((long long)k * n) % M);
          sr += br[(size_t)n] * cs[q] - bi[(size_t)n] * sn[q];
          si += br[(size_t)n] * sn[q] + bi[(size_t)n] * cs[q];
        }
        fb[(size_t)k] = make_float2((float)(sr / M), (float)(si / M));
      }
      float2* dw = nullptr; float2* dfb = nullptr;
      CHK(dft_table(dft_key(DK_CHIRP, d), w, &dw));
      if (inplace_radices) {                             // the in-place engine multiplies in digit-reversed order
        std::vector<float2> fp((size_t)M);
        for (int k = 0; k < M; ++k) fp[(size_t)dif_position(k, M, *inplace_radices)] = fb[(size_t)k];
        fb.swap(fp);
      }
      CHK(dft_table(dft_key(inplace_radices ? DK_CHIRP_SPECTRUM_INPLACE : DK_CHIRP_SPECTRUM, d, M), fb, &dfb));     // (the spectrum depends on d, M and the engine)
      a->M = M; a->chirp = dw; a->fb = dfb; a->inplace = inplace_radices ? 1 : 0;
    }
    return SSN_OK;
  }

  // Neuron order for the whole-block kernel (f32).  k_ens_block leaves out the spike-time arithmetic and the decode of a
  // (wave, round) slot - 64 lanes x SSN_BLOCK_IL neuron pairs - in which no neuron spikes (ssn_block.hpp).  With neurons in
  // sampling order some neuron of every slot always spikes; so the neurons of each ensemble are dealt to the slots by WHEN they
  // can fire.  For the oscillator ensembles (reference pathintegration.py:162-166: 3-D, state on a circle in the plane of the
  // first two dimensions, third dimension the frequency input) neuron i is driven above threshold while
  //   rho_i cos(theta - phi_i) > 1 - bias_i      (rho_i, phi_i: its scaled encoder in that plane; theta: the oscillator's phase)
  // i.e. always (class 0), never (class 2), or for theta within alpha_i = acos((1 - bias_i) / rho_i) of phi_i (class 1).  Class 1
  // is cut into bands of similar alpha and sorted by phi inside a band: a slot then holds neurons of one sector and falls
  // silent when the phase is elsewhere.  Any order is a valid order - sums over an ensemble's neurons do not depend on it beyond
  // rounding - so the heuristic only decides how often the shortcut applies (other 3-D ensembles: rarely, harmlessly).  The
  // permutation lives in the buffers' upload / download (Buf::perm): callers keep seeing their own order.
  int reorder_block_neurons(const ssn_model_desc* m, const ssn_op_desc& eo, int nthr, int npt) {
    const int64_t K = eo.i[1], n = eo.i[2], din = eo.i[3], dout = eo.i[4];
    if (din != 3 || n < 256) return SSN_OK;
    const double* enc = (const double*)m->buffers[eo.i[5]].data;        // [K][din][n]
    const double* bias = (const double*)m->buffers[eo.i[6]].data;       // [K][n]
    // positions of the ensemble's row in slot order: slot (round j, wave w) = positions of the groups j * IL ... j * IL + IL - 1
    // at the wave's 128 columns, group after group (neuron index of (group g, thread t, component c) = (g * nthr + t) * 2 + c)
    const int ng = npt / 2, il = std::min<int>(SSN_BLOCK_IL, ng), waves = nthr / 64;
    std::vector<int32_t> pos;
    pos.reserve((size_t)n);
    for (int j = 0; j * il < ng; ++j)
      for (int w = 0; w < waves; ++w)
        for (int u = 0; u < il && j * il + u < ng; ++u)
          for (int q = 0; q < 128; ++q) {
            const int64_t p = ((int64_t)(j * il + u) * nthr + w * 64) * 2 + q;
            if (p < n) pos.push_back((int32_t)p);
          }
    if ((int64_t)pos.size() != n) return SSN_OK;                        // (a variant this layout does not describe: keep the order)
    auto perm = std::make_shared<std::vector<int32_t>>((size_t)(K * n));
    struct Key { int cls; double a, phi; int32_t i; };
    std::vector<Key> keys((size_t)n);
    const int slot = 128 * il;
    for (int64_t k = 0; k < K; ++k) {
      const double* e0 = enc + (k * din + 0) * n; const double* e1 = enc + (k * din + 1) * n; const double* b = bias + k * n;
      int n1 = 0;
      for (int64_t i = 0; i < n; ++i) {
        const double rho = std::hypot(e0[i], e1[i]), mth = 1.0 - b[i];
        Key& q = keys[(size_t)i];
        q.i = (int32_t)i; q.phi = std::atan2(e1[i], e0[i]);
        if (!(rho > 0.0) || mth >= rho) { q.cls = 2; q.a = mth - rho; }
        else if (mth <= -rho) { q.cls = 0; q.a = 0.0; }
        else { q.cls = 1; q.a = std::acos(mth / rho); ++n1; }
      }
      // class 0 | class 1 by descending alpha | class 2 by how far below threshold
      std::sort(keys.begin(), keys.end(), [](const Key& x, const Key& y) {
        if (x.cls != y.cls) return x.cls < y.cls;
        if (x.cls == 1 && x.a != y.a) return x.a > y.a;
        if (x.cls == 2 && x.a != y.a) return x.a < y.a;
        return x.i < y.i;
      });
      // bands of class 1: whole slots each, about four bands; inside a band by phi
      size_t lo = 0;
      while (lo < keys.size() && keys[lo].cls == 0) ++lo;
      const size_t hi = lo + (size_t)n1;
      const size_t band = std::max<size_t>(slot, ((size_t)n1 / 4 + slot - 1) / slot * slot);
      for (size_t s0 = lo; s0 < hi; s0 += band)
        std::sort(keys.begin() + (long)s0, keys.begin() + (long)std::min(hi, s0 + band), [](const Key& x, const Key& y) {
          return x.phi != y.phi ? x.phi < y.phi : x.i < y.i;
        });
      int32_t* pk = perm->data() + k * n;
      for (int64_t r = 0; r < n; ++r) pk[pos[(size_t)r]] = keys[(size_t)r].i;
    }
    struct Target { int64_t id; int rows; };
    for (const Target t : {Target{eo.i[5], (int)din}, Target{eo.i[6], 1}, Target{eo.i[7], (int)dout}, Target{eo.i[9], 1}, Target{eo.i[10], 1}}) {
      Buf& b = bufs[(size_t)t.id];
      if (b.cols != n) return fail(SSN_EINVAL, "ensemble array buffer %lld has %lld columns, expected %lld", (long long)t.id, (long long)b.cols, (long long)n);
      b.perm = perm; b.perm_rows = t.rows;
      if (b.packed == 2) continue;                                      // (the refractory view of the voltage buffer: no storage of its own to re-upload)
      CHK(upload_buf(b, (const double*)m->buffers[t.id].data));
    }
    return SSN_OK;
  }

  // Core stage == one recurrent ensemble array (the path integrator's VCO array): x assembled in the
  // kernel prologue, decoded rows / synapse update / hand-off / step counter in one barrier-free finish
  // kernel.  Returns false (and plans nothing) when the core does not have that shape.
  bool try_fused_core(const ssn_model_desc* m, int* rc) {
    *rc = SSN_OK;
    if (!bsig) return false;
    for (auto& p : probes) if (p.stage == 1) return false;
    int ens_i = -1;
    std::vector<int> axpys, lows;
    for (int i = 0; i < m->n_ops; ++i) {
      const ssn_op_desc& o = m->ops[i];
      if (o.stage != 1) continue;
      if (o.kind == SSN_OP_ENSARRAY) { if (ens_i >= 0) return false; ens_i = i; }
      else if (o.kind == SSN_OP_AXPY && o.i[3] == 0) axpys.push_back(i);
      else if (o.kind == SSN_OP_LOWPASS) lows.push_back(i);
      else return false;
    }
    if (ens_i < 0 || axpys.size() > 4) return false;
    const ssn_op_desc& eo = m->ops[ens_i];
    const int64_t K = eo.i[1], din = eo.i[3], dout = eo.i[4];
    const int64_t x0 = eo.i[0], x1 = eo.i[0] + K * din;
    auto inside = [](int64_t lo, int64_t hi, const std::vector<ssn_range>& rs) {
      for (auto& r : rs) if (lo >= r.lo && hi <= r.hi) return true;
      return false;
    };
    if (!inside(x0, x1, pre_to_core)) return false;
    for (int i : axpys) {                       // recurrent terms must land inside pre-stage-provided inputs
      const ssn_op_desc& o = m->ops[i];
      if (!inside(o.i[0], o.i[0] + o.i[2], pre_to_core)) return false;
    }
    const int32_t* didx = (const int32_t*)m->buffers[eo.i[8]].data;
    std::vector<int> row_of((size_t)n_sig, -1);
    for (int64_t j = 0; j < K * dout; ++j) row_of[(size_t)didx[j]] = (int)j;   // padded rows share a trash slot: harmless
    std::vector<int> lp_state((size_t)(K * dout), -1);
    std::vector<double> lp_a((size_t)(K * dout), 0.0), lp_b((size_t)(K * dout), 0.0);
    std::vector<unsigned char> rowout((size_t)(K * dout), 0);
    for (int i : lows) {
      const ssn_op_desc& o = m->ops[i];
      for (int64_t e = 0; e < o.i[2]; ++e) {
        const int row = row_of[(size_t)(o.i[1] + e)];
        if (row < 0) {
          // no local row feeds this filter input (VCO owned by another rank): input and state stay 0,
          // provided nothing else writes that signal
          if (inside(o.i[1] + e, o.i[1] + e + 1, pre_to_core) || sig_init[(size_t)(o.i[1] + e)] != 0.0) return false;
          continue;
        }
        if (lp_state[(size_t)row] >= 0) return false;
        lp_state[(size_t)row] = (int)(o.i[0] + e);
        lp_a[(size_t)row] = o.f[0];
        lp_b[(size_t)row] = (1.0 - o.f[0]) * o.f[1];
      }
    }
    for (auto& r : core_to_post)
      for (int64_t e = r.lo; e < r.hi; ++e) {
        const int row = row_of[(size_t)e];
        if (row >= 0) rowout[(size_t)row] = 1;      // (elements no local row writes stay at their initial value)
      }
    // ---- can the finish be deferred into the next step's prologue?  every recurrent term must read a
    //      filter state owned by a row of the SAME ensemble (ens_k -> Lowpass -> ens_k), one term per input
    bool defer = !(flags & 16);        // decided below: only worth it in the latency-bound regime (few workgroups)
    std::vector<int> xrow((size_t)(K * din), -1), lp_has((size_t)(K * dout), 0);
    std::vector<double> xalpha((size_t)(K * din), 0.0);
    {
      std::vector<int> owner((size_t)n_sig, -1);
      for (int64_t j = 0; j < K * dout; ++j)
        if (lp_state[(size_t)j] >= 0) { owner[(size_t)lp_state[(size_t)j]] = (int)j; lp_has[(size_t)j] = 1; }
      // filter states whose input no row writes (dropped all-zero decoder rows, other ranks' VCOs) stay 0
      std::vector<unsigned char> dead_zero((size_t)n_sig, 0);
      for (int i : lows) {
        const ssn_op_desc& o = m->ops[i];
        for (int64_t e = 0; e < o.i[2]; ++e) {
          const int64_t st = o.i[0] + e, src = o.i[1] + e;
          if (row_of[(size_t)src] < 0 && sig_init[(size_t)st] == 0.0 && sig_init[(size_t)src] == 0.0 &&
              !inside(src, src + 1, pre_to_core)) dead_zero[(size_t)st] = 1;
        }
      }
      for (int i : axpys) {
        const ssn_op_desc& o = m->ops[i];
        for (int64_t e = 0; e < K * din && defer; ++e) {
          const int64_t xi = x0 + e;
          if (xi < o.i[0] || xi >= o.i[0] + o.i[2]) continue;
          const int64_t st = o.i[1] + (xi - o.i[0]);
          const int row = owner[(size_t)st];
          if (row < 0 && dead_zero[(size_t)st]) continue;               // adds exactly 0
          if (row < 0 || row / dout != e / din || xrow[(size_t)e] >= 0) defer = false;
          else { xrow[(size_t)e] = (int)(row % dout); xalpha[(size_t)e] = o.f[0]; }
        }
      }
    }
    // ---- plan A: the ensembles are independent inside a block -> one k_ens_block launch per block -----
    int blk_threads = 0, blk_tpb = 0, blk_npt = 0, blk_lds = 0;
    // (One workgroup steps one ensemble.  Splitting an ensemble over P workgroups that exchange partial sums through L2
    //  every timestep was built and measured in round 1 - 3.3 us per timestep at P = 2 and 3.6 at P = 4 against 2.95 for
    //  one workgroup on 127- / 64-VCO shards: a cross-CU round trip costs ~1.6 us, more than halving the neuron work
    //  saves - and removed in round 2: DESIGN.md section 5.)
    // Split ensembles (flag 1073741824; f32, dout <= 4): when the array has fewer ensembles than the GPU has CUs - a 4- or
    // 8-GPU shard of config 2, a small model - P = 2 or 4 member workgroups step one ensemble and exchange their partial sums
    // every timestep (k_ens_block; the exchange costs 0.5 - 0.6 us per timestep, tools/xcd_exchange.hip).  All K * P
    // workgroups must be resident together, so P is only raised while K * P fits the CUs - and the caller only asks for
    // it when this process has the GPU to itself.
    int split_P = 1, n_member = 0;
    if ((flags & 1073741824) && sizeof(T) == 4 && dout <= 4) {
      hipDeviceProp_t prop;
      int n_cu = 256;
      if (hipGetDeviceProperties(&prop, device) == hipSuccess) n_cu = prop.multiProcessorCount;
      if (const char* env = getenv("SSN_BLOCK_SPLIT_CUS")) n_cu = atoi(env);            // (tests: pretend the GPU is smaller / larger)
      const int64_t n_ens = eo.i[2];
      for (int cand : {4, 2}) {
        if (K * cand <= n_cu && n_ens / cand >= 1024) { split_P = cand; break; }
      }
      if (const char* env = getenv("SSN_BLOCK_SPLIT")) { const int q = atoi(env); if (q == 1 || q == 2 || q == 4 || q == 8) split_P = q; }
      if (split_P > 1) n_member = (int)(((n_ens + split_P - 1) / split_P + 3) / 4 * 4);
    }
    if (defer && !(flags & 128) && ens_fast(eo) && (sizeof(T) == 8 || (dt <= 0.05 * eo.f[0] && eo.f[1] >= dt)) &&
        ssn::ens_block_supported<T>((int)din, (int)dout, split_P > 1 ? n_member : (int)eo.i[2], &blk_threads, &blk_tpb, &blk_npt, &blk_lds)) {
      const int64_t nr = K * dout;
      int* d_lp = nullptr; T* d_a = nullptr; T* d_b = nullptr; unsigned char* d_ro = nullptr; int* d_xrow = nullptr; T* d_xalpha = nullptr;
      if ((*rc = dmalloc(&d_lp, nr * 4)) != SSN_OK) return true;
      if ((*rc = dmalloc(&d_a, nr * (int64_t)sizeof(T))) != SSN_OK) return true;
      if ((*rc = dmalloc(&d_b, nr * (int64_t)sizeof(T))) != SSN_OK) return true;
      if ((*rc = dmalloc(&d_ro, nr)) != SSN_OK) return true;
      if ((*rc = dmalloc(&d_xrow, (int64_t)K * din * 4)) != SSN_OK) return true;
      if ((*rc = dmalloc(&d_xalpha, (int64_t)K * din * (int64_t)sizeof(T))) != SSN_OK) return true;
      fused_bufs.insert(fused_bufs.end(), {(void*)d_lp, (void*)d_a, (void*)d_b, (void*)d_ro, (void*)d_xrow, (void*)d_xalpha});
      hipMemcpy(d_lp, lp_state.data(), (size_t)nr * 4, hipMemcpyHostToDevice);
      hipMemcpy(d_ro, rowout.data(), (size_t)nr, hipMemcpyHostToDevice);
      hipMemcpy(d_xrow, xrow.data(), (size_t)(K * din) * 4, hipMemcpyHostToDevice);
      if ((*rc = upload(lp_a.data(), d_a, 1, nr, nr)) != SSN_OK) return true;
      if ((*rc = upload(lp_b.data(), d_b, 1, nr, nr)) != SSN_OK) return true;
      if ((*rc = upload(xalpha.data(), d_xalpha, 1, K * din, K * din)) != SSN_OK) return true;
      ssn::EnsArgs<T> ea;
      fill_ens_args(eo, ea);
      blk = ssn::BlockArgs<T>{};
      blk.enc = ea.enc; blk.bias = ea.bias; blk.dec = ea.dec; blk.S = ea.V;
      blk.xrows = bsig; blk.bsig = bsig; blk.sig = sig; blk.sig_w = sig;
      blk.didx = (const int*)bufs[eo.i[8]].d; blk.lp_state = d_lp; blk.lp_a = d_a; blk.lp_b = d_b;
      blk.xrow = d_xrow; blk.xalpha = d_xalpha; blk.rowout = d_ro;
      blk.n_sig = n_sig; blk.x_off = eo.i[0];
      blk.K = ea.K; blk.n = ea.n; blk.n_pad = ea.n_pad; blk.din = ea.din; blk.dout = ea.dout;
      blk.B = 0; blk.row0 = 1; blk.threads = blk_threads; blk.tpb = blk_tpb; blk.npt = blk_npt; blk.enc_lds = blk_lds;
      blk.dec_neuron_major = ea.fast == 1 ? 1 : 0;
      blk.np = ea.np;
      blk.P = split_P; blk.n_member = n_member; blk.xslots = nullptr; blk.xerr = nullptr;
      {
        unsigned long long* d_ss = nullptr;
        if ((*rc = dmalloc(&d_ss, 16)) != SSN_OK) return true;
        fused_bufs.push_back((void*)d_ss);
        hipMemset(d_ss, 0, 16);
        blk.slot_stats = d_ss;
      }
      if (split_P > 1) {
        unsigned int* d_x = nullptr; int* d_e = nullptr;
        if ((*rc = dmalloc(&d_x, (int64_t)3 * K * 64 * 4)) != SSN_OK) return true;
        if ((*rc = dmalloc(&d_e, 16)) != SSN_OK) return true;
        fused_bufs.insert(fused_bufs.end(), {(void*)d_x, (void*)d_e});
        hipMemset(d_e, 0, 16);
        blk.xslots = d_x; blk.xerr = d_e;
      }
      dom_units = (int64_t)ea.K * ea.n * block;
      dom_bytes = (double)dom_units * (ea.din + ea.dout + 5) * sizeof(T);
      fused_core = fused_block = true;
      // (only worth anything with a kernel built with SSN_BLOCK_SKIP = 1 or 2 - both measured slower than the plain loop, round 4 -
      //  so the neuron order is the caller's unless SSN_BLOCK_SORT=1 asks for the slot order)
      if (sizeof(T) == 4 && split_P == 1 && getenv("SSN_BLOCK_SORT") && atoi(getenv("SSN_BLOCK_SORT")) == 1)
        *rc = reorder_block_neurons(m, eo, blk_npt == 2 ? blk_threads : blk_tpb, blk_npt);
      return true;
    }
    // ---- plan B: [k_ensarray (fused prologue), k_ens_finish] ------------------------------------------
    Item it; it.type = IT_ENS;
    fill_ens_args(eo, it.ens);
    ssn::EnsArgs<T>& a = it.ens;
    a.xrows = bsig;
    // measured on MI355X (tools/bench_shard.py): 5080 workgroups (config 2 on one GPU) 35.4 us/step with the
    // separate finish kernel vs 37.4 deferred; 2540 / 1270 / 640 workgroups (2 / 4 / 8-GPU shards) 24.5 / 15.8 /
    // 13.1 vs 22.0 / 12.5 / 9.4 deferred
    // (one process, full size: 36.6 us/step deferred vs 37.4 with the finish kernel -> deferred whenever it applies)
    for (int i : axpys) {
      const ssn_op_desc& o = m->ops[i];
      a.rec_dst[a.n_rec] = o.i[0]; a.rec_src[a.n_rec] = o.i[1]; a.rec_len[a.n_rec] = o.i[2]; a.rec_alpha[a.n_rec] = (T)o.f[0];
      ++a.n_rec;
    }
    a.partials_stride = (int64_t)a.K * a.P * a.dout;
    if ((*rc = dmalloc(&a.partials, 2 * a.partials_stride * (int64_t)sizeof(T))) != SSN_OK) return true;
    hipMemset(a.partials, 0, (size_t)(2 * a.partials_stride) * sizeof(T));
    it.dominant = true;
    dom_units = (int64_t)a.K * a.n;
    dom_bytes = (double)dom_units * (a.din + a.dout + 5) * sizeof(T);
    Item fi; fi.type = IT_FINISH;
    ssn::FinishArgs<T>& f = fi.fin;
    f = ssn::FinishArgs<T>{};
    const int64_t nr = K * dout;
    int* d_lp = nullptr; T* d_a = nullptr; T* d_b = nullptr; unsigned char* d_ro = nullptr; unsigned int* d_ticket = nullptr;
    if ((*rc = dmalloc(&d_lp, nr * 4)) != SSN_OK) return true;
    if ((*rc = dmalloc(&d_a, nr * (int64_t)sizeof(T))) != SSN_OK) return true;
    if ((*rc = dmalloc(&d_b, nr * (int64_t)sizeof(T))) != SSN_OK) return true;
    if ((*rc = dmalloc(&d_ro, nr)) != SSN_OK) return true;
    if ((*rc = dmalloc(&d_ticket, 64)) != SSN_OK) return true;
    fused_bufs.insert(fused_bufs.end(), {(void*)d_lp, (void*)d_a, (void*)d_b, (void*)d_ro, (void*)d_ticket});
    hipMemcpy(d_lp, lp_state.data(), (size_t)nr * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ro, rowout.data(), (size_t)nr, hipMemcpyHostToDevice);
    hipMemset(d_ticket, 0, 64);
    if ((*rc = upload(lp_a.data(), d_a, 1, nr, nr)) != SSN_OK) return true;
    if ((*rc = upload(lp_b.data(), d_b, 1, nr, nr)) != SSN_OK) return true;
    f.partials = a.partials; f.didx = (const int*)bufs[eo.i[8]].d; f.lp_state = d_lp; f.lp_a = d_a; f.lp_b = d_b;
    f.rowout = d_ro; f.sig = sig; f.bsig = bsig; f.n_sig = n_sig; f.ctx = d_ctx; f.ticket = d_ticket;
    f.K = a.K; f.P = a.P; f.dout = a.dout; f.n_blocks = (int)((nr + 255) / 256);
    f.mode = 0; f.fstate = nullptr; f.partials_stride = a.partials_stride; f.lp_has = nullptr;
    if (defer) {
      int* d_has = nullptr; int* d_xrow = nullptr; T* d_xalpha = nullptr; T* d_fstate = nullptr;
      if ((*rc = dmalloc(&d_has, nr * 4)) != SSN_OK) return true;
      if ((*rc = dmalloc(&d_xrow, (int64_t)K * din * 4)) != SSN_OK) return true;
      if ((*rc = dmalloc(&d_xalpha, (int64_t)K * din * (int64_t)sizeof(T))) != SSN_OK) return true;
      if ((*rc = dmalloc(&d_fstate, 2 * nr * (int64_t)sizeof(T))) != SSN_OK) return true;
      fused_bufs.insert(fused_bufs.end(), {(void*)d_has, (void*)d_xrow, (void*)d_xalpha, (void*)d_fstate});
      hipMemcpy(d_has, lp_has.data(), (size_t)nr * 4, hipMemcpyHostToDevice);
      hipMemcpy(d_xrow, xrow.data(), (size_t)(K * din) * 4, hipMemcpyHostToDevice);
      hipMemset(d_fstate, 0, (size_t)(2 * nr) * sizeof(T));
      if ((*rc = upload(xalpha.data(), d_xalpha, 1, K * din, K * din)) != SSN_OK) return true;
      a.defer = 1; a.sub = 0; a.didx = f.didx; a.lp_has = d_has; a.lp_a = d_a; a.lp_b = d_b; a.fstate = d_fstate;
      a.xrow = d_xrow; a.xalpha = d_xalpha; a.rowout = d_ro; a.bsig = bsig; a.sig_w = sig;
      f.fstate = d_fstate; f.lp_has = d_has;
      fin_flush = f; fin_flush.mode = 1;
      fin_begin = f; fin_begin.mode = 2;
      items.push_back(it);
      fused_core = fused_defer = true;
      return true;
    }
    items.push_back(it);
    items.push_back(fi);
    fused_core = true;
    // the batched stages are planned by plan(); nothing else runs per timestep
    return true;
  }

  int plan(const ssn_model_desc* m) {
    int frc = SSN_OK;
    const bool fused = !(flags & 1) && try_fused_core(m, &frc);
    CHK(frc);
    std::vector<std::vector<MOp>> programs;   // programs in step order
    std::vector<int> item_prog;               // for IT_PROGRAM items: index into programs
    std::vector<MOp> cur;
    int prev_level = -1;
    bool force_barrier = false;
    // Element-wise micro operators are executed with element p of a range starting at `base` on thread
    // (p - base) mod 1024.  A dependency between two of them therefore needs no workgroup barrier when both map every
    // shared element to the same thread (equal range starts modulo 1024): program order inside the thread suffices.
    struct EwAcc { long long lo, len; bool w; };
    auto ew_access = [&](const MOp& m, std::vector<EwAcc>& out) -> bool {
      switch (m.kind) {
        case ssn::M_FILL: case ssn::M_TABLE: case ssn::M_ROW_IN: out.push_back({m.dst, m.len, true}); return true;
        case ssn::M_AXPY_SET: out.push_back({m.src, m.len, false}); out.push_back({m.dst, m.len, true}); return true;
        case ssn::M_AXPY_INC: case ssn::M_LOWPASS: out.push_back({m.src, m.len, false}); out.push_back({m.dst, m.len, true}); return true;
        case ssn::M_ROW_OUT: case ssn::M_PROBE: out.push_back({m.src, m.len, false}); return true;
        default: return false;
      }
    };
    auto level_change_needs_barrier = [&](const MOp& op) -> bool {
      std::vector<EwAcc> mine;
      if (!ew_access(op, mine)) return true;
      for (size_t q = cur.size(); q-- > 0;) {             // operators since the last barrier
        std::vector<EwAcc> theirs;
        if (!ew_access(cur[q], theirs)) return true;
        for (const EwAcc& x : mine)
          for (const EwAcc& y : theirs)
            if ((x.w || y.w) && x.lo < y.lo + y.len && y.lo < x.lo + x.len && ((x.lo - y.lo) % 1024) != 0) return true;
        if (cur[q].barrier) break;
      }
      return false;
    };
    auto push_micro = [&](MOp op, int level, bool exec_inserted) {
      if (cur.empty()) op.barrier = 0;
      else if (force_barrier || exec_inserted) op.barrier = 1;
      else if (level != prev_level) op.barrier = level_change_needs_barrier(op) ? 1 : 0;
      else op.barrier = 0;
      force_barrier = exec_inserted;
      prev_level = level;
      cur.push_back(op);
    };
    auto flush = [&]() {
      if (cur.empty()) return;
      Item it; it.type = IT_PROGRAM;
      item_prog.push_back((int)programs.size());
      programs.push_back(cur);
      items.push_back(it);
      cur.clear();
      force_barrier = false;
      prev_level = -1;
    };
    int64_t best_units = -1;
    int best_item = -1;
    int cur_phase = 0;
    for (auto& r : pre_to_core) {        // the pre stage's results for this timestep
      if (fused) break;
      MOp op{};
      op.kind = ssn::M_ROW_IN; op.dst = r.lo; op.len = r.hi - r.lo; op.p0 = bsig; op.i0 = n_sig; op.i1 = r.lo;
      push_micro(op, -10, false);
    }
    std::vector<std::pair<int, ssn::BatchOp<T>>> pre_sorted, post_sorted;
    std::vector<std::pair<MOp, int>> pending_reduce;
    auto drain_reduces = [&]() {
      bool first = true;
      for (auto& pr : pending_reduce) { push_micro(pr.first, pr.second, first); first = false; }
      pending_reduce.clear();
    };
    for (int i = 0; i < m->n_ops; ++i) {
      const ssn_op_desc& o = m->ops[i];
      MOp op{};
      const size_t items_before = items.size();
      if (phased && o.stage == 1 && o.phase != cur_phase) {      // the exchange: no program spans it
        if (o.phase < cur_phase) return fail(SSN_EINVAL, "operators of phase 0 after phase 1");
        flush();
        for (auto& it : items) if (it.phase < 0) it.phase = cur_phase;
        cur_phase = o.phase;
      }
      struct LevelTag { std::vector<Item>& v; size_t from; int level; ~LevelTag() { for (size_t q = from; q < v.size(); ++q) if (v[q].type != IT_PROGRAM && v[q].level < 0) v[q].level = level; } }
          level_tag{items, items_before, o.level};
      if (o.stage == 1 && !fused && !pending_reduce.empty() &&
          !(o.kind == SSN_OP_MATVEC && !is_micro(o) && bufs[o.i[4]].transposed && !(flags & 4096)))
        drain_reduces();
      if (o.stage != 1) {
        ssn::BatchOp<T> b{};
        b.bsig = bsig; b.n_sig = n_sig; b.src_prev = o.src_prev ? 1 : 0;
        int64_t wlo = 0, wlen = 0;
        switch (o.kind) {
          case SSN_OP_FILL: b.kind = ssn::M_FILL; b.dst = o.i[0]; b.len = o.i[1]; b.a = (T)o.f[0]; wlo = o.i[0]; wlen = o.i[1]; break;
          case SSN_OP_TABLE: b.kind = ssn::M_TABLE; b.dst = o.i[0]; b.len = o.i[1]; b.p0 = d_tables + o.i[2];
            tables[o.i[2]].width = o.i[1]; wlo = o.i[0]; wlen = o.i[1];
            if (table_dst.size() < tables.size()) table_dst.resize(tables.size(), {0, 0});
            table_dst[(size_t)o.i[2]] = {o.i[0], o.i[1]};
            break;
          case SSN_OP_AXPY: b.kind = o.i[3] ? ssn::M_AXPY_SET : ssn::M_AXPY_INC; b.dst = o.i[0]; b.src = o.i[1]; b.len = o.i[2];
            b.a = (T)o.f[0]; wlo = o.i[0]; wlen = o.i[2]; break;
          case SSN_OP_LOWPASS: b.kind = ssn::M_LOWPASS; b.dst = o.i[0]; b.src = o.i[1]; b.len = o.i[2];
            b.a = (T)o.f[0]; b.b = (T)((1.0 - o.f[0]) * o.f[1]); wlo = o.i[0]; wlen = o.i[2]; break;
          case SSN_OP_MATVEC: b.kind = o.i[5] ? ssn::M_MATVEC_SET : ssn::M_MATVEC_INC; b.dst = o.i[0]; b.src = o.i[1];
            b.len = o.i[2]; b.cols = (int)o.i[3]; b.ld = (int)bufs[o.i[4]].ld; b.p0 = bufs[o.i[4]].d; wlo = o.i[0]; wlen = o.i[2]; break;
          default: return fail(SSN_EUNSUPPORTED, "operator kind %d cannot run in a time-batched stage", o.kind);
        }
        for (int64_t j = wlo; j < wlo + wlen; ++j) batched_mask[(size_t)j] = 1;
        (o.stage == 0 ? pre_sorted : post_sorted).push_back({o.border, b});
        continue;
      }
      if (fused) continue;               // the core is [k_ensarray, k_ens_finish], planned above
      switch (o.kind) {
        case SSN_OP_FILL:
          op.kind = ssn::M_FILL; op.dst = o.i[0]; op.len = o.i[1]; op.a = (T)o.f[0];
          push_micro(op, o.level, false); break;
        case SSN_OP_TABLE:
          op.kind = ssn::M_TABLE; op.dst = o.i[0]; op.len = o.i[1]; op.p0 = d_tables + o.i[2];
          tables[o.i[2]].width = o.i[1];
          push_micro(op, o.level, false); break;
        case SSN_OP_AXPY:
          op.kind = o.i[3] ? ssn::M_AXPY_SET : ssn::M_AXPY_INC; op.dst = o.i[0]; op.src = o.i[1]; op.len = o.i[2]; op.a = (T)o.f[0];
          push_micro(op, o.level, false); break;
        case SSN_OP_LOWPASS:
          op.kind = ssn::M_LOWPASS; op.dst = o.i[0]; op.src = o.i[1]; op.len = o.i[2];
          op.a = (T)o.f[0]; op.b = (T)((1.0 - o.f[0]) * o.f[1]);
          push_micro(op, o.level, false); break;
        case SSN_OP_LINCOMB: {
          std::vector<ssn::LinTerm<T>> terms((size_t)o.i[2]);
          for (int64_t q = 0; q < o.i[2]; ++q)
            terms[(size_t)q] = ssn::LinTerm<T>{(long long)((const int32_t*)m->buffers[o.i[3]].data)[q], (T)((const double*)m->buffers[o.i[4]].data)[q]};
          ssn::LinTerm<T>* d_terms = nullptr;
          CHK(dmalloc(&d_terms, (int64_t)std::max<size_t>(1, terms.size()) * (int64_t)sizeof(ssn::LinTerm<T>)));
          scratch_bufs.push_back(d_terms);
          if (!terms.empty()) HIPCHK(hipMemcpy(d_terms, terms.data(), terms.size() * sizeof(ssn::LinTerm<T>), hipMemcpyHostToDevice));
          lin_terms[(const void*)d_terms] = terms;
          op.kind = ssn::M_LINCOMB; op.dst = o.i[0]; op.len = o.i[1]; op.i0 = o.i[2]; op.p0 = d_terms;
          op.a = (T)o.f[0]; op.b = T(1); op.c = (T)o.f[1];
          push_micro(op, o.level, false); break;
        }
        case SSN_OP_GATE:
          op.kind = ssn::M_GATE; op.dst = o.i[0]; op.src = o.i[1]; op.len = o.i[2]; op.a = (T)o.f[0]; op.b = (T)o.f[1];
          push_micro(op, o.level, true); break;
        case SSN_OP_MATVEC: {
          const Buf& w = bufs[o.i[4]];
          ssn::DftArgs da{};
          // (a chirp-z transform of M >= 2048 points is two long FFTs on ONE workgroup: 16.5 us per launch at d = 1801 with the
          //  in-register radix-8 passes of round 3 (38.5 us with generic radix-16 butterflies) against ~7 us for its 26 MB matrix
          //  spread over the chip - measured on config 5: 266 us per timestep with the matrices, 288 - 295 with the FFT as a body
          //  of the round grid; the matrix is used for such a length unless flag 268435456 asks for the FFT)
          if (o.i[6] && sizeof(T) == 4 && !(flags & 512) && plan_dft(o, &da) == SSN_OK && da.N &&
              (da.M < 2048 || (flags & 268435456))) {
            // the matrix is a real-DFT map of a circular-convolution network: mixed-radix FFT instead of a GEMV
            flush();
            Item it; it.type = IT_DFT; it.dft = da;
            it.src = sig + o.i[1]; it.dst = sig + o.i[0]; it.rows = (int)o.i[2]; it.cols = (int)o.i[3];
            items.push_back(it);
            break;
          }
          if (is_micro(o)) {
            op.kind = o.i[5] ? ssn::M_MATVEC_SET : ssn::M_MATVEC_INC;
            op.dst = o.i[0]; op.src = o.i[1]; op.len = o.i[2]; op.i0 = o.i[3]; op.i1 = w.ld; op.p0 = w.d;
            push_micro(op, o.level, false);
          } else if (w.transposed) {
            // spike-sparse decoders: partial sums per spike-list chunk, reduced in the following program
            flush();
            int chunks = 32;
            const int rows_pad = (int)w.ldt;
            int seg = 0;
            if (std::find(seg_spikes.begin(), seg_spikes.end(), o.i[1]) != seg_spikes.end()) {
              const int n_seg = ((int)o.i[3] + 255) / 256;       // segments of 256 neurons, ~32 chunks
              const int max_chunks = getenv("SSN_SPMV_MAX_CHUNKS") ? std::max(1, atoi(getenv("SSN_SPMV_MAX_CHUNKS"))) : 64;      // (A/B knob)
              seg = std::max(1, (n_seg + max_chunks - 1) / max_chunks);
              chunks = (n_seg + seg - 1) / seg;
            }
            T* partial = nullptr;
            CHK(dmalloc(&partial, (int64_t)chunks * rows_pad * (int64_t)sizeof(T)));
            scratch_bufs.push_back(partial);
            Item it; it.type = IT_SPMV; it.Wm = (T*)w.d; it.src = sig + o.i[1]; it.dst = partial;
            it.rows = (int)o.i[2]; it.cols = (int)o.i[3]; it.ld = (int)w.ldt; it.n = chunks; it.seg = seg;
            for (auto& sl : spike_lists) if (sl.first == o.i[1]) { it.list = sl.second.first; it.count = sl.second.second; }
            {
              // the chunk reduction is deferred past any sparse products that follow directly (they read spike vectors,
              // never a reduction's output): the products then sit next to each other and share a launch, and their
              // reductions share one program
              items.push_back(it);
              MOp r{};
              r.kind = o.i[5] ? ssn::M_REDUCE_SET : ssn::M_REDUCE_INC; r.dst = o.i[0]; r.len = o.i[2]; r.i0 = chunks; r.i1 = rows_pad; r.p0 = partial;
              pending_reduce.push_back({r, o.level});
            }
          } else {
            flush();
            Item it; it.type = IT_MATVEC; it.Wm = (T*)w.d; it.src = sig + o.i[1]; it.dst = sig + o.i[0];
            it.rows = (int)o.i[2]; it.cols = (int)o.i[3]; it.ld = (int)w.ld; it.set = (int)o.i[5];
            items.push_back(it);
          }
          break;
        }
        case SSN_OP_ENSARRAY: {
          flush();
          Item it; it.type = IT_ENS;
          ssn::EnsArgs<T>& a = it.ens;
          fill_ens_args(o, a);
          ens_chunking(a, 2048);
          CHK(dmalloc(&a.partials, (int64_t)a.K * a.P * a.dout * (int64_t)sizeof(T)));
          const int64_t units = (int64_t)a.K * a.n;
          if (units > best_units) { best_units = units; best_item = (int)items.size(); }
          host_idx.push_back({bufs[o.i[8]].d, {(const int32_t*)m->buffers[o.i[8]].data, (int64_t)a.K * a.dout}});
          if (a.P == 1 && !(flags & 8192)) {
            // one workgroup covers a whole ensemble (the 8128 product ensembles of 50 neurons): it writes the decoded
            // rows itself, no finish operator (and no program launch) behind the kernel
            a.direct = 1; a.didx = (const int*)bufs[o.i[8]].d; a.sig_w = sig;
            items.push_back(it);
            force_barrier = true;
            break;
          }
          items.push_back(it);
          MOp f{};
          f.kind = ssn::M_ENS_FINISH; f.len = (int64_t)a.K * a.dout; f.i0 = a.P; f.i1 = a.dout;
          f.p0 = a.partials; f.p1 = bufs[o.i[8]].d;
          push_micro(f, o.level, true);
          break;
        }
        case SSN_OP_NEURONS: {
          flush();
          Item it; it.type = IT_NEURONS; it.src = sig + o.i[0]; it.dst = sig + o.i[1]; it.n = (int)o.i[2];
          it.V = (T*)bufs[o.i[3]].d; it.R = (T*)bufs[o.i[4]].d; it.np = neuron_params(o.i[5], o.f); it.scalar = (T)o.f[3];
          bool feeds_sparse = false;     // does a spike-sparse decoder product read this ensemble's spikes?
          for (int j = 0; j < m->n_ops; ++j) {
            const ssn_op_desc& q = m->ops[j];
            if (q.kind == SSN_OP_MATVEC && q.stage == 1 && q.i[1] == o.i[1] && q.i[3] == o.i[2] && bufs[q.i[4]].transposed) feeds_sparse = true;
          }
          if (feeds_sparse && !(flags & 1024)) {
            // segmented spike list: every 256-neuron workgroup of k_neurons leaves its spikes as an ordered index list
            const int n_seg = ((int)o.i[2] + 255) / 256;
            CHK(dmalloc(&it.list, (int64_t)n_seg * 256 * 4));
            CHK(dmalloc(&it.count, (int64_t)n_seg * 4 + 64));
            HIPCHK(hipMemset(it.count, 0, (size_t)n_seg * 4 + 64));
            scratch_bufs.push_back(it.list); scratch_bufs.push_back(it.count);
            spike_lists.push_back({o.i[1], {it.list, it.count}});
            seg_spikes.push_back(o.i[1]);
          }
          items.push_back(it);
          break;
        }
        case SSN_OP_PES: {
          flush();
          const Buf& w = bufs[o.i[0]];
          Item it; it.type = IT_PES; it.Wm = (T*)w.d; it.rows = (int)o.i[1]; it.cols = (int)o.i[2]; it.ld = (int)w.ld;
          it.aux0 = sig + o.i[3]; it.aux1 = sig + o.i[4]; it.scalar = (T)o.f[0];
          if (w.transposed) {   // neuron-major weights: W^T[c][r] += kappa * act[c] * err[r]
            it.rows = (int)o.i[2]; it.cols = (int)o.i[1]; it.ld = (int)w.ldt; it.aux0 = sig + o.i[4]; it.aux1 = sig + o.i[3];
          }
          items.push_back(it);
          break;
        }
        case SSN_OP_VOJA: {
          flush();
          const Buf& w = bufs[o.i[0]];
          Item it; it.type = IT_VOJA; it.Wm = (T*)w.d; it.rows = (int)o.i[1]; it.cols = (int)o.i[2]; it.ld = (int)w.ld;
          it.src = sig + o.i[3]; it.aux0 = sig + o.i[4]; it.aux1 = sig + o.i[5]; it.aux2 = (const T*)bufs[o.i[6]].d;
          it.scalar = (T)o.f[0];
          items.push_back(it);
          break;
        }
        case SSN_OP_CLEANUP: {
          // similarities S @ x into a scratch vector (k_matvec), then argmax + row gather in the next program
          flush();
          const Buf& w = bufs[o.i[4]];
          const char* gmin = getenv("SSN_GRID_MIN_MB");
          const int64_t grid_min_bytes = (gmin ? atoll(gmin) : 64) << 20;
          bool grid_route = sizeof(T) == 4 && o.i[5] > 0 && !(flags & 524288) && o.i[2] * o.i[3] * (int64_t)sizeof(T) >= grid_min_bytes;
          // Round 4: a smaller grid (2-D, 10^4 rows: 40 MB of HBM stream per timestep at d = 1015) also goes through its factor
          // tables when the step runs as rounds - as bodies of the round grid (no MFMA product: 10^4 dot products of length 2K
          // from two L2-resident 100 x 2K tables), so nothing is launched on its own.  SSN_GRID_DOT_MIN_MB moves the threshold.
          const char* dmin = getenv("SSN_GRID_DOT_MIN_MB");
          const bool grid_dot = !grid_route && sizeof(T) == 4 && o.i[5] > 0 && !(flags & (524288 | 2097152)) && o.i[2] < 65536 &&
                                o.i[10] * (int64_t)sizeof(T) <= 48 * 1024 &&
                                o.i[2] * o.i[3] * (int64_t)sizeof(T) >= ((dmin ? atoll(dmin) : 16) << 20);
          grid_route = grid_route || grid_dot;
          // K-split of the grid product (partial products summed by the argmax's first stage): enough workgroups to hide latency
          int gsplit = grid_route && o.i[2] >= 65536 ? (int)std::min<int64_t>(4, std::max<int64_t>(1, o.i[10] / 448)) : 1;
          if (grid_route && o.i[2] >= 65536 && getenv("SSN_GRID_SPLIT")) gsplit = std::max(1, std::min(8, atoi(getenv("SSN_GRID_SPLIT"))));
          for (int it2 = 0; it2 < 4; ++it2) {            // the split count the launcher's 32-column slabs actually produce
            const int kper = (((int)o.i[10] + gsplit - 1) / gsplit + 31) / 32 * 32;
            gsplit = ((int)o.i[10] + kper - 1) / kper;
          }
          T* scratch = nullptr;
          CHK(dmalloc(&scratch, o.i[2] * gsplit * (int64_t)sizeof(T)));
          scratch_bufs.push_back(scratch);
          Item it; it.type = IT_MATVEC; it.Wm = (T*)w.d; it.src = sig + o.i[1]; it.dst = scratch;
          it.rows = (int)o.i[2]; it.cols = (int)o.i[3]; it.ld = (int)w.ld; it.set = 1;
          if (grid_route) {
            // big sample grid (10^6 points in 3-D): similarities from the grid's factor tables - half spectrum of x
            // (k_matvec with the DFT rows), left operand (k_grid_lhs), one MFMA product - instead of streaming the
            // table; the table itself is only read for the winning row
            const Buf& fd = bufs[o.i[5] - 1]; const Buf& fl = bufs[o.i[6] - 1]; const Buf& fr = bufs[o.i[7] - 1];
            const int na = (int)o.i[8], nn = (int)o.i[9], k2 = (int)o.i[10];
            T* X = nullptr; T* A = nullptr;
            CHK(dmalloc(&X, k2 * (int64_t)sizeof(T)));
            scratch_bufs.push_back(X);
            CHK(dmalloc(&A, (int64_t)na * k2 * (int64_t)sizeof(T)));
            scratch_bufs.push_back(A);
            Item sx; sx.type = IT_MATVEC; sx.Wm = (T*)fd.d; sx.src = sig + o.i[1]; sx.dst = X;
            sx.rows = k2; sx.cols = (int)o.i[3]; sx.ld = (int)fd.ld; sx.set = 1;
            items.push_back(sx);
            Item sl; sl.type = IT_GRID_LHS; sl.src = X; sl.aux0 = (const T*)fl.d; sl.ld = (int)fl.ld; sl.dst = A;
            sl.rows = na; sl.cols = k2;
            items.push_back(sl);
            it.type = grid_dot ? IT_GRID_DOT : IT_GRID_GEMM; it.src = A; it.Wm = (T*)fr.d; it.ld = (int)fr.ld; it.rows = na; it.n = nn; it.cols = k2;
            it.seg = gsplit;
          }
          if (sizeof(T) == 8) {
            // parity mode: ordered accumulation over a transposed copy (ties are decided by rounding)
            const int ldt = ((int)o.i[2] + VW - 1) / VW * VW;
            T* wt = nullptr;
            CHK(dmalloc(&wt, (int64_t)o.i[3] * ldt * (int64_t)sizeof(T)));
            scratch_bufs.push_back(wt);
            HIPCHK(ssn::launch_transpose<T>(stream, (const T*)w.d, wt, (int)o.i[2], (int)o.i[3], (int)w.ld, ldt));
            it.type = IT_MATVEC_ORDERED; it.Wm = wt; it.ld = ldt;
          }
          items.push_back(it);
          MOp g{};
          g.kind = ssn::M_ARGMAX_GATHER; g.dst = o.i[0]; g.len = o.i[3]; g.i0 = o.i[2]; g.i1 = w.ld; g.p0 = w.d; g.p1 = scratch;
          if (o.i[2] >= 65536) {
            // long similarity vector: first maxima of P slices in their own launch, the program picks among the P candidates
            const int P = (int)std::min<int64_t>(1024, (o.i[2] + 1023) / 1024);      // ~4 workgroups per CU at 10^6 rows
            T* part = nullptr;
            CHK(dmalloc(&part, P * (int64_t)(sizeof(T) + sizeof(int))));
            scratch_bufs.push_back(part);
            Item ap; ap.type = IT_ARGMAX_PART; ap.src = scratch; ap.dst = part; ap.rows = (int)o.i[2]; ap.n = P; ap.seg = gsplit;
            items.push_back(ap);
            g.p1 = part; g.src = P;
          }
          push_micro(g, o.level, true);
          break;
        }
        default: return fail(SSN_EINVAL, "unknown operator kind %d", o.kind);
      }
    }
    drain_reduces();
    auto by_order = [](const std::pair<int, ssn::BatchOp<T>>& a, const std::pair<int, ssn::BatchOp<T>>& b) { return a.first < b.first; };
    std::stable_sort(pre_sorted.begin(), pre_sorted.end(), by_order);
    std::stable_sort(post_sorted.begin(), post_sorted.end(), by_order);
    for (auto& x : pre_sorted) pre_ops.push_back(x.second);
    for (auto& x : post_sorted) post_ops.push_back(x.second);
    bool first_out = true;
    for (auto& r : core_to_post) {       // hand this timestep's results to the post stage
      if (fused) break;
      MOp op{};
      op.kind = ssn::M_ROW_OUT; op.src = r.lo; op.len = r.hi - r.lo; op.p0 = bsig; op.i0 = n_sig; op.i1 = r.lo;
      push_micro(op, -11, first_out);
      first_out = false;
    }
    bool first_probe = true;
    for (size_t p = 0; p < probes.size(); ++p) {
      if (probes[p].stage != 1) {         // sampled by the batched stages, after all their operators
        ssn::BatchOp<T> b{};
        b.bsig = bsig; b.n_sig = n_sig; b.kind = ssn::M_PROBE; b.src = probes[p].src; b.len = probes[p].width; b.p0 = d_pslots + p;
        post_ops.push_back(b);
        continue;
      }
      MOp op{};
      op.kind = ssn::M_PROBE; op.src = probes[p].src; op.len = probes[p].width; op.p0 = d_pslots + p;
      push_micro(op, -2, first_probe);
      first_probe = false;
    }
    optimise_batch(pre_ops);
    optimise_batch(post_ops);
    if (!fused) {
      MOp end{};
      end.kind = ssn::M_STEP_END;
      push_micro(end, -3, false);
    }
    force_barrier = true;
    flush();
    for (auto& it : items) if (it.phase < 0) it.phase = cur_phase;      // (the tail program - probes, step counter - closes phase 1)
    if (phased && cur_phase != 1) return fail(SSN_EINVAL, "a model with exchange ranges needs operators of phase 1");
    if (best_item >= 0) {
      items[best_item].dominant = true;
      const ssn::EnsArgs<T>& a = items[best_item].ens;
      dom_units = (int64_t)a.K * a.n;
      dom_bytes = (double)dom_units * (a.din + a.dout + 5) * sizeof(T);
    }
    round_mode = !fused && !(flags & 2097152);
    if (round_mode) {
      CHK(build_rounds(programs, item_prog));
      int n_core = 0;
      for (int i = 0; i < m->n_ops; ++i) n_core += m->ops[i].stage == 1;
      bool probe_in_core = false;
      for (auto& p : probes) probe_in_core = probe_in_core || p.stage == 1;
      core_empty = bsig && n_core == 0 && !probe_in_core;
      if (core_empty) launches_per_step = 0;
      return SSN_OK;
    }
    // Sink programs: a program none of the operators up to the next program depends on (e.g. the clean-up's
    // argmax + row gather, whose result is first used after the path integrator's ensembles; the chunk reductions
    // of sparse products whose sums are first used behind the next neuron populations) joins that next program -
    // same operators, same order among dependent ones, one launch (~8 us inside the step graph) fewer each.
    if (!fused && !phased) {
      for (bool changed = true; changed;) {
        changed = false;
        analyse_dependencies(programs, item_prog);
        int pi = 0;
        for (size_t i = 0; i < items.size() && !changed; ++i) {
          if (items[i].type != IT_PROGRAM) continue;
          const int my_prog = pi++;
          if (i == 0) continue;                                  // the head starts the timestep
          size_t j = i + 1;
          bool free_to_move = true;
          for (; j < items.size() && items[j].type != IT_PROGRAM; ++j)
            for (int dep : item_deps[j]) free_to_move = free_to_move && dep != (int)i;
          if (j >= items.size() || j == i + 1 || !free_to_move) continue;
          std::vector<MOp>& dst = programs[(size_t)my_prog + 1];
          if (!dst.empty()) dst[0].barrier = 1;
          dst.insert(dst.begin(), programs[(size_t)my_prog].begin(), programs[(size_t)my_prog].end());
          programs.erase(programs.begin() + my_prog);
          items.erase(items.begin() + (long)i);
          item_prog.pop_back();
          changed = true;
        }
      }
    }
    // micro-op storage: programs in order, then a copy of the head behind the tail for the fused launch.
    // Long first level of the head program -> its own grid-wide launch (k_vecops).  The head then no longer starts the
    // timestep, so the tail / head fusion is given up: worth it from ~16 k elements on (SLAM config 3: 42 -> ~23 us).
    std::vector<MOp>& vec_ops = vecops_host;
    vec_ops.clear();
    if (!fused && !phased && !items.empty() && items.front().type == IT_PROGRAM && !programs.empty()) {
      std::vector<MOp>& head = programs[(size_t)item_prog[0]];
      size_t lv = 0;
      long long elems = 0;
      // The prefix that moves: element-wise operators up to the first barrier that are INDEPENDENT of each other.  Inside
      // a program an operator may follow one it depends on without a barrier when both map every element to the same
      // thread (push_micro); k_vecops maps elements to threads of the whole grid, so such a chain must not move as one.
      auto reads_src = [](const MOp& m) { return m.kind == ssn::M_AXPY_INC || m.kind == ssn::M_AXPY_SET || m.kind == ssn::M_LOWPASS; };
      auto overlap = [](long long a, long long al, long long b, long long bl) { return a < b + bl && b < a + al; };
      for (; lv < head.size() && (lv == 0 || !head[lv].barrier); ++lv) {
        const MOp& q = head[lv];
        const int kd = q.kind;
        if (!(kd == ssn::M_FILL || kd == ssn::M_ROW_IN || kd == ssn::M_TABLE || kd == ssn::M_AXPY_INC ||
              kd == ssn::M_AXPY_SET || kd == ssn::M_LOWPASS)) break;
        bool hazard = false;
        for (size_t e = 0; e < lv && !hazard; ++e) {
          const MOp& pr = head[e];
          hazard = overlap(q.dst, q.len, pr.dst, pr.len) || (reads_src(pr) && overlap(q.dst, q.len, pr.src, pr.len)) ||
                   (reads_src(q) && overlap(q.src, q.len, pr.dst, pr.len));
        }
        if (hazard) break;
        elems += q.len;
      }
      if (elems >= 16384 && lv > 0 && lv < head.size()) {
        vec_ops.assign(head.begin(), head.begin() + (long)lv);
        head.erase(head.begin(), head.begin() + (long)lv);
        head[0].barrier = 0;
        Item vi; vi.type = IT_VECOPS; vi.n = (int)std::min<long long>(256, (elems + 2047) / 2048);
        items.insert(items.begin(), vi);
      }
    }
    int n_prog = (int)programs.size();
    mops.clear();
    prog_descs.clear();
    for (int p = 0; p < n_prog; ++p) {
      ssn::ProgDesc pd{};
      pd.op_begin = (int)mops.size(); pd.op_count = (int)programs[p].size();
      mops.insert(mops.end(), programs[p].begin(), programs[p].end());
      prog_descs.push_back(pd);
    }
    int pi = 0;
    for (auto& it : items)
      if (it.type == IT_PROGRAM) { it.op_begin = pi; it.op_count = 1; ++pi; }      // op_begin = program index
    can_fuse = !fused && !phased && items.size() >= 2 && items.front().type == IT_PROGRAM && items.back().type == IT_PROGRAM;
    if (can_fuse) {
      // the tail is the last program: a copy of the head's descriptor behind it makes [tail, head] one launch
      ssn::ProgDesc hd = prog_descs.front();
      const int hb = (int)mops.size();
      for (int j = 0; j < hd.op_count; ++j) mops.push_back(mops[(size_t)(hd.op_begin + j)]);
      mops[(size_t)hb].barrier = 1;
      hd.op_begin = hb;
      prog_descs.push_back(hd);
      tail_begin = n_prog - 1;
    }
    CHK(dmalloc(&d_progs, (int64_t)std::max<size_t>(1, prog_descs.size()) * (int64_t)sizeof(ssn::ProgDesc)));
    if (!prog_descs.empty()) HIPCHK(hipMemcpy(d_progs, prog_descs.data(), prog_descs.size() * sizeof(ssn::ProgDesc), hipMemcpyHostToDevice));
    if (!vec_ops.empty()) {
      items.front().op_begin = (int)mops.size(); items.front().op_count = (int)vec_ops.size();
      mops.insert(mops.end(), vec_ops.begin(), vec_ops.end());
    }
    CHK(dmalloc(&d_mops, (int64_t)mops.size() * (int64_t)sizeof(MOp)));
    HIPCHK(hipMemcpy(d_mops, mops.data(), mops.size() * sizeof(MOp), hipMemcpyHostToDevice));
    launches_per_step = (int)items.size() - (can_fuse ? 1 : 0);
    if (fused_defer) launches_per_step = 1;
    if (fused_block) launches_per_step = 0;      // one launch per block
    int n_core_ops = 0;
    for (int i = 0; i < m->n_ops; ++i) n_core_ops += m->ops[i].stage == 1;
    bool core_probe = false;
    for (auto& p : probes) core_probe = core_probe || p.stage == 1;
    core_empty = bsig && n_core_ops == 0 && !core_probe;
    if (core_empty) launches_per_step = 0;
    if (!fused) analyse_dependencies(programs, item_prog);
    if (!fused && !(flags & (256 | 4096))) merge_adjacent_items();
    if (getenv("SSN_DEBUG_PLAN")) {
      int pj = 0;
      for (size_t i = 0; i < items.size(); ++i) {
        if (items[i].type != IT_PROGRAM) { fprintf(stderr, "[ssn] plan item %2zu type %d rows %d cols %d n %d\n", i, items[i].type, items[i].rows, items[i].cols, items[i].n); continue; }
        const auto& pr = programs[(size_t)item_prog[(size_t)pj++]];
        long long elems = 0; int levels = 1;
        for (size_t q = 0; q < pr.size(); ++q) { elems += pr[q].len * (pr[q].kind == ssn::M_MATVEC_INC || pr[q].kind == ssn::M_MATVEC_SET ? pr[q].i0 : 1); if (q && pr[q].barrier) ++levels; }
        fprintf(stderr, "[ssn] plan item %2zu program: %zu ops, %d levels, %lld elements:", i, pr.size(), levels, elems);
        for (auto& o : pr) fprintf(stderr, " %d/%lld", o.kind, o.len);
        fprintf(stderr, "\n");
      }
    }
    return SSN_OK;
  }


  // ---- round plan ---------------------------------------------------------------------------------------------------
  // Every micro-operator and every big operator of the timestep is a unit; a unit's round is the earliest one its data
  // hazards (RAW, WAR, WAW against every earlier unit of the sequential order) allow.  A round is launched as one k_round
  // grid (ssn_round.hpp) plus one plain launch for each operator whose kernel has no body there (ensemble arrays, the
  // ordered / factored clean-up products).
  static bool glue_row_kind(int k) {
    return k == ssn::M_MATVEC_INC || k == ssn::M_MATVEC_SET || k == ssn::M_ENS_FINISH || k == ssn::M_REDUCE_SET || k == ssn::M_REDUCE_INC;
  }

  // k_round body of an ensemble array (-1: its variant has none, it is launched on its own)
  int ens_round_kind(const ssn::EnsArgs<T>& a) const {
    if ((a.defer && a.defer != 2) || a.xrows || (flags & 4194304)) return -1;
    if (a.fast == 1 && a.din == 3 && a.dout == 4) return ssn::RK_ENS_3_4_S;
    if (a.fast == 1 && a.din == 3 && a.dout == 5) return ssn::RK_ENS_3_5_S;
    if (a.fast == 2 && a.din == 1 && a.dout == 1) {
      // many small ensembles (the product arrays of a circular convolution): a wave per ensemble instead of a workgroup
      if (a.n <= 64 && a.P == 1 && a.direct && !no_small_ens) return ssn::RK_ENS_SMALL;
      return ssn::RK_ENS_1_1_D;
    }
    return -1;
  }
  const bool no_small_ens = getenv("SSN_SMALL_ENS") && atoi(getenv("SSN_SMALL_ENS")) == 0;      // A/B knob
  // blocks of an ensemble array's grid inside a round
  int ens_round_blocks(const ssn::EnsArgs<T>& a) const {
    return ens_round_kind(a) == ssn::RK_ENS_SMALL ? (a.K + 4 * ssn::ENS_SMALL_PER_WAVE - 1) / (4 * ssn::ENS_SMALL_PER_WAVE) : a.K * a.P;
  }

  // Cost model of the round balancer: device time a unit needs when bandwidth-bound (us at ~5 TB/s), the latency of a
  // single-workgroup unit (us), and the number of blocks of its grid (0: not splittable).
  // latency of the single-workgroup bodies as the balancer prices them (us): alone a transform takes 9.5 us and the argmax ~3; inside a
  // round that also streams they take 25 - 30 and ~13 (tools/round_stamps.py) - SSN_LAT_DFT / SSN_LAT_SOLO for A/B runs
  // (fractions of its matrix a learning rule's update is priced at: dense it reads and writes everything, in the reference's memory
  //  1.3 % of the filtered activities are nonzero - SSN_COST_PES / SSN_COST_VOJA for A/B runs)
  // tallest matrix that runs four rows per workgroup with four vectors of a row in flight per lane (one trip for 1015 columns)
  // instead of sixteen rows per workgroup with one vector of four rows in flight (four trips)
  int matvec_r1_max = getenv("SSN_MATVEC_R1_MAX") ? atoi(getenv("SSN_MATVEC_R1_MAX")) : 4096;
  double cost_pes = getenv("SSN_COST_PES") ? atof(getenv("SSN_COST_PES")) : 2.0;
  double cost_voja = getenv("SSN_COST_VOJA") ? atof(getenv("SSN_COST_VOJA")) : 0.2;
  double lat_dft = getenv("SSN_LAT_DFT") ? atof(getenv("SSN_LAT_DFT")) : 9.0;
  double lat_solo = getenv("SSN_LAT_SOLO") ? atof(getenv("SSN_LAT_SOLO")) : 3.0;
  void unit_cost(int mop, int item, double* us, double* lat, int* blocks) const {
    *us = 0.0; *lat = 0.0; *blocks = 0;
    const double per_us = 5.0e6;      // bytes per microsecond
    if (mop >= 0) {
      const MOp& op = mops[(size_t)mop];
      if (op.kind == ssn::M_GATE || op.kind == ssn::M_ARGMAX_GATHER) *lat = lat_solo;
      else *us = (double)op.len * 3.0 * sizeof(T) / per_us;
      return;
    }
    const Item& it = items[(size_t)item];
    switch (it.type) {
      case IT_ENS:
        *us = (double)it.ens.K * it.ens.n_pad * (it.ens.din + 4.5) * sizeof(T) / per_us;
        if (ens_round_kind(it.ens) >= 0) *blocks = ens_round_blocks(it.ens);
        break;
      case IT_MATVEC:
        *us = (double)it.rows * it.ld * sizeof(T) / per_us;
        if ((size_t)it.cols * sizeof(T) <= 48 * 1024) *blocks = it.rows <= matvec_r1_max ? (it.rows + 3) / 4 : (it.rows + 15) / 16;
        break;
      case IT_PES: *us = cost_pes * it.rows * it.ld * sizeof(T) / per_us; *blocks = ((it.cols + 1023) / 1024) * ((it.rows + ssn::PES_ROWS - 1) / ssn::PES_ROWS); break;
      case IT_VOJA: *us = cost_voja * it.rows * it.ld * sizeof(T) / per_us; break;
      case IT_SPMV: *us = 0.1 * (double)it.cols * it.ld * sizeof(T) / per_us; break;
      case IT_NEURONS: *us = (double)it.n * 5.0 * sizeof(T) / per_us; break;
      case IT_MATVEC_NEURONS: *us = (double)it.rows * it.ld * sizeof(T) / per_us; *blocks = (it.rows + 15) / 16; break;
      case IT_DFT: *lat = lat_dft; break;
      default: *us = 1.0; break;
    }
  }

  // Round balancer of the pipelined plan.  The earliest-round assignment stacks the bandwidth-bound operators of a
  // timestep in a few rounds and leaves others with one latency-bound workgroup (a DFT) and an idle chip.  Every block of
  // a grid is independent, so a heavy operator may run in pieces: any round from its own up to the last one before its
  // first conflicting successor is valid for any of its blocks (successors were placed later than that, predecessors
  // earlier; pieces only ever move later, which keeps every other window valid).  Heavy instances are poured, a
  // thirty-second at a time, into the round of their window where the quantum costs least (free under a latency-bound
  // round's critical workgroup; otherwise the least loaded).
  template <typename Inst, typename H, typename C>
  void balance_rounds(std::vector<Inst>& all, int nr, size_t per, H conflict, C cost) {
    const size_t N = all.size();
    std::vector<double> load((size_t)nr, 0.0), lat((size_t)nr, 0.0);
    std::vector<double> us(N, 0.0);
    std::vector<int> blocks(N, 0);
    std::map<int, double> serial;
    for (size_t i = 0; i < N; ++i) {
      double l = 0.0;
      cost(all[i].unit, &us[i], &l, &blocks[i]);
      load[(size_t)all[i].round] += us[i];
      if (all[i].solo && all[i].chain >= 0) { serial[all[i].chain] += std::max(l, 1.0 + us[i]); l = serial[all[i].chain]; }      // a serial chain: one block, members back to back
      lat[(size_t)all[i].round] = std::max(lat[(size_t)all[i].round], l);
    }
    std::vector<Inst> extra;
    for (size_t i = 0; i < N; ++i) {
      if (us[i] < 4.0 || blocks[i] < 64) continue;            // heavy: >= ~20 MB
      const int r0 = all[i].round;
      int r1 = nr - 1;
      const size_t stop = std::min(N, ((size_t)all[i].sub + 2) * per);
      for (size_t j = i + 1; j < stop; ++j)
        if (all[j].round - 1 < r1 && conflict(all[i].unit, all[j].unit)) r1 = all[j].round - 1;
      if (r1 <= r0) continue;
      load[(size_t)r0] -= us[i];
      const int Q = 32;
      const double q = us[i] / Q;
      std::vector<int> share((size_t)(r1 - r0 + 1), 0);
      for (int k = 0; k < Q; ++k) {
        int best = r0;
        double best_cost = 1e30, best_load = 1e30;
        for (int r = r0; r <= r1; ++r) {
          const double before = std::max(load[(size_t)r], lat[(size_t)r]), after = std::max(load[(size_t)r] + q, lat[(size_t)r]);
          const double c = after - before;
          if (c < best_cost - 1e-9 || (c < best_cost + 1e-9 && load[(size_t)r] < best_load)) { best = r; best_cost = c; best_load = load[(size_t)r]; }
        }
        share[(size_t)(best - r0)] += 1;
        load[(size_t)best] += q;
      }
      int lo = 0, used = 0;
      bool first = true;
      for (int r = r0; r <= r1; ++r) {
        const int sh = share[(size_t)(r - r0)];
        if (!sh) continue;
        used += sh;
        const int hi = used == Q ? blocks[i] : (int)((long long)blocks[i] * used / Q);
        if (hi <= lo) continue;
        if (first) { all[i].round = r; all[i].lo = lo; all[i].cnt = hi - lo; first = false; }
        else { Inst p = all[i]; p.round = r; p.lo = lo; p.cnt = hi - lo; extra.push_back(p); }
        lo = hi;
      }
    }
    all.insert(all.end(), extra.begin(), extra.end());
    if (getenv("SSN_DEBUG_PLAN")) {          // what the balancer believes the rounds cost (us): bandwidth-bound load | longest single-workgroup body
      fprintf(stderr, "[ssn] balanced rounds (load | latency, us):");
      double total = 0.0;
      for (int r = 0; r < nr; ++r) { fprintf(stderr, " %.0f|%.0f", load[(size_t)r], lat[(size_t)r]); total += std::max(load[(size_t)r], lat[(size_t)r]); }
      fprintf(stderr, "\n[ssn] sum of max(load, latency) over %d rounds: %.0f us\n", nr, total);
    }
  }

  // Round plan, recurrent ensemble arrays (the SLAM network's oscillators, reference pathintegration.py:180-182: ensemble k ->
  // Lowpass -> ensemble k).  As separate operators the recurrence costs four dependent rounds per timestep: array (partial sums
  // per 1024-neuron chunk) -> M_ENS_FINISH (sums) -> M_LOWPASS (filter state) -> M_LINCOMB (input) -> array.  When every input
  // element of the array takes its recurrent term from a filter fed by a decoded row of the SAME ensemble - the proof
  // try_fused_core makes for the whole-block kernel, here on the micro-operators of the round plan - the array does it all in
  // its prologue (ens_body, defer == 2): the filter and the recurrent term of the lincomb leave the plan, the finish stays for
  // the other readers of the decoded rows (the to_SSP read-out).
  std::vector<std::pair<void*, size_t>> zero_on_reset;      // scratch that holds state across timesteps (partial sums, filter states)
  int fold_recurrent_filter(std::vector<std::vector<MOp>>& programs, const std::vector<int>& item_prog) {
    for (size_t ei = 0; ei < items.size(); ++ei) {
      Item& E = items[ei];
      if (E.type != IT_ENS || E.ens.direct || E.ens.defer || E.ens.xrows || E.ens.P < 2 || ens_round_kind(E.ens) < 0) continue;
      ssn::EnsArgs<T>& a = E.ens;
      const int64_t K = a.K, din = a.din, dout = a.dout;
      // the finish operator and the host copy of the destination indices
      MOp* F = nullptr;
      for (auto& pr : programs) for (MOp& op : pr) if (op.kind == ssn::M_ENS_FINISH && op.p0 == (const void*)a.partials) F = &op;
      if (!F) continue;
      const int32_t* didx = nullptr;
      for (auto& h : host_idx) if (h.first == F->p1 && h.second.second == K * dout) didx = h.second.first;
      if (!didx) continue;
      std::map<long long, int> row_of;                          // decoded signal -> row index k * dout + r
      for (int64_t j = 0; j < K * dout; ++j) row_of[didx[j]] = (int)j;
      // filters fed by decoded rows of this array whose state is read by lincombs into the array's input range only
      const long long x0 = a.x_off, x1 = a.x_off + K * din;
      std::vector<int> xrow((size_t)(K * din), -1);
      std::vector<double> xalpha((size_t)(K * din), 0.0), lpa((size_t)(K * dout), 0.0), lpb((size_t)(K * dout), 0.0);
      struct Hit { size_t prog, op; };
      std::vector<Hit> filters;
      std::vector<std::pair<Hit, size_t>> terms;                // (lincomb, term index) to remove
      bool ok = true;
      for (size_t pi = 0; pi < programs.size() && ok; ++pi)
        for (size_t oi = 0; oi < programs[pi].size() && ok; ++oi) {
          const MOp& L = programs[pi][oi];
          if (L.kind != ssn::M_LOWPASS) continue;
          bool fed = false, foreign = false;
          for (long long q = 0; q < L.len; ++q) { if (row_of.count(L.src + q)) fed = true; }
          if (!fed) continue;
          // every reader of the state range must be a lincomb into [x0, x1) with the state as one of its terms, element for element
          std::vector<std::pair<Hit, size_t>> my_terms;
          for (size_t pj = 0; pj < programs.size() && !foreign; ++pj)
            for (size_t oj = 0; oj < programs[pj].size() && !foreign; ++oj) {
              const MOp& X = programs[pj][oj];
              if (&X == &L) continue;
              std::vector<Rng> acc;
              micro_access(acc, X, false);
              bool touches = false;
              for (const Rng& r : acc) if (r.space == (const void*)sig && r.lo < L.dst + L.len && L.dst < r.hi) touches = true;
              if (!touches) continue;
              if (X.kind != ssn::M_LINCOMB || X.dst < x0 || X.dst + X.len > x1) { foreign = true; break; }
              auto lt = lin_terms.find(X.p0);
              if (lt == lin_terms.end()) { foreign = true; break; }
              for (size_t t = 0; t < lt->second.size(); ++t) {
                const long long ts = lt->second[t].src;
                if (ts + X.len <= L.dst || L.dst + L.len <= ts) continue;
                if (ts < L.dst || ts + X.len > L.dst + L.len) { foreign = true; break; }
                my_terms.push_back({Hit{pj, oj}, t});
              }
            }
          for (auto& it2 : items) if (it2.type != IT_PROGRAM) { std::vector<Rng> acc; item_access(acc, it2); for (const Rng& r : acc) if (r.space == (const void*)sig && r.lo < L.dst + L.len && L.dst < r.hi) foreign = true; }
          for (auto& pb : probes) if (pb.src < L.dst + L.len && L.dst < pb.src + pb.width) foreign = true;
          if (foreign || my_terms.empty()) continue;             // somebody else needs this state: it stays an operator
          for (long long q = 0; q < L.len && ok; ++q) if (sig_init[(size_t)(L.dst + q)] != 0.0) ok = false;      // (x of the first timestep uses the state as it stands)
          // map every input element to the row that feeds it
          for (auto& mt : my_terms) {
            const MOp& X = programs[mt.first.prog][mt.first.op];
            const auto& tt = lin_terms[X.p0][mt.second];
            for (long long q = 0; q < X.len && ok; ++q) {
              const long long e = X.dst + q - x0, st = tt.src + q;
              auto f = row_of.find(L.src + (st - L.dst));
              if (f == row_of.end()) continue;                   // a state no row feeds (dropped all-zero decoder row): stays 0
              if (f->second / dout != e / din || xrow[(size_t)e] >= 0) { ok = false; break; }
              xrow[(size_t)e] = (int)(f->second % dout); xalpha[(size_t)e] = (double)tt.alpha * (double)X.b;
              lpa[(size_t)f->second] = (double)L.a; lpb[(size_t)f->second] = (double)L.b;
            }
            terms.push_back(mt);
          }
          filters.push_back(Hit{pi, oi});
        }
      if (!ok || filters.empty()) continue;
      // device tables, the second set of partial sums, the filter states
      const int64_t nr = K * dout, ps = (int64_t)K * a.P * dout;
      int* d_xrow = nullptr; T* d_xalpha = nullptr; T* d_a = nullptr; T* d_b = nullptr; T* d_f = nullptr; T* d_part = nullptr;
      CHK(dmalloc(&d_xrow, K * din * 4)); CHK(dmalloc(&d_xalpha, K * din * (int64_t)sizeof(T)));
      CHK(dmalloc(&d_a, nr * (int64_t)sizeof(T))); CHK(dmalloc(&d_b, nr * (int64_t)sizeof(T)));
      CHK(dmalloc(&d_f, 2 * nr * (int64_t)sizeof(T))); CHK(dmalloc(&d_part, 2 * ps * (int64_t)sizeof(T)));
      for (void* q : {(void*)d_xrow, (void*)d_xalpha, (void*)d_a, (void*)d_b, (void*)d_f}) scratch_bufs.push_back(q);
      HIPCHK(hipMemcpy(d_xrow, xrow.data(), (size_t)(K * din) * 4, hipMemcpyHostToDevice));
      CHK(upload(xalpha.data(), d_xalpha, 1, K * din, K * din));
      CHK(upload(lpa.data(), d_a, 1, nr, nr)); CHK(upload(lpb.data(), d_b, 1, nr, nr));
      HIPCHK(hipMemset(d_f, 0, (size_t)(2 * nr) * sizeof(T))); HIPCHK(hipMemset(d_part, 0, (size_t)(2 * ps) * sizeof(T)));
      zero_on_reset.push_back({(void*)d_f, (size_t)(2 * nr) * sizeof(T)}); zero_on_reset.push_back({(void*)d_part, (size_t)(2 * ps) * sizeof(T)});
      hipFree(a.partials);                                      // (the single set planned before; ~Sim frees the new one through the item)
      F->p0 = d_part; F->src = ps;
      a.partials = d_part; a.partials_stride = ps; a.defer = 2; a.sub = 0;
      a.xrow = d_xrow; a.xalpha = d_xalpha; a.lp_a = d_a; a.lp_b = d_b; a.fstate = d_f;
      // the recurrent terms leave their lincombs (new term lists), the filters leave the plan
      std::map<const void*, std::vector<size_t>> drop;          // lincomb term list -> term indices to drop
      for (auto& mt : terms) drop[programs[mt.first.prog][mt.first.op].p0].push_back(mt.second);
      for (auto& pr : programs)
        for (MOp& op : pr) {
          if (op.kind != ssn::M_LINCOMB) continue;
          auto d = drop.find(op.p0);
          if (d == drop.end()) continue;
          std::vector<ssn::LinTerm<T>> tt;
          const auto& old = lin_terms[op.p0];
          for (size_t t = 0; t < old.size(); ++t) if (std::find(d->second.begin(), d->second.end(), t) == d->second.end()) tt.push_back(old[t]);
          ssn::LinTerm<T>* d_terms = nullptr;
          CHK(dmalloc(&d_terms, (int64_t)std::max<size_t>(1, tt.size()) * (int64_t)sizeof(ssn::LinTerm<T>)));
          scratch_bufs.push_back(d_terms);
          if (!tt.empty()) HIPCHK(hipMemcpy(d_terms, tt.data(), tt.size() * sizeof(ssn::LinTerm<T>), hipMemcpyHostToDevice));
          lin_terms[(const void*)d_terms] = tt;
          op.p0 = d_terms; op.i0 = (long long)tt.size();
        }
      std::sort(filters.begin(), filters.end(), [](const Hit& x, const Hit& y) { return x.prog != y.prog ? x.prog > y.prog : x.op > y.op; });
      for (const Hit& h : filters) programs[h.prog].erase(programs[h.prog].begin() + (long)h.op);
      if (getenv("SSN_DEBUG_PLAN")) fprintf(stderr, "[ssn] recurrent array of %lld ensembles completes its own previous timestep: %zu filter operator(s) and %zu lincomb term(s) folded\n", (long long)K, filters.size(), terms.size());
    }
    (void)item_prog;
    return SSN_OK;
  }

  int build_rounds(const std::vector<std::vector<MOp>>& programs_in, const std::vector<int>& item_prog) {
    struct Unit { int mop = -1; int item = -1; int phase = 0; bool writes = false; std::vector<Rng> acc; };
    std::vector<Unit> units;
    mops.clear();
    std::vector<std::vector<MOp>> programs = programs_in;
    // Round 4: the critical recurrence of a SLAM timestep - memory spikes -> sparse decode -> PES (may not touch W before the
    // decode has read it) -> filter of the activities PES reads (may not run before PES) -> next timestep's spikes (may not
    // overwrite what that filter reads) - is four rounds of WAR hazards.  The filter update folded into the PES body (the
    // workgroup that has read row r's factor advances it) takes one of them out.  Round 2 measured this fold slower because the
    // oscillators' own four-round recurrence kept the period at four; it pays together with the oscillator array completing
    // its previous timestep itself (below).  SSN_PES_FOLD=0: off.
    if ((!phased || (getenv("SSN_PHASED_PES_FOLD") && atoi(getenv("SSN_PHASED_PES_FOLD")) == 1)) && !(getenv("SSN_PES_FOLD") && atoi(getenv("SSN_PES_FOLD")) == 0))
      for (size_t i = 0; i < items.size(); ++i) {
        Item& pe = items[i];
        if (pe.type != IT_PES || pe.cols > 1024 || !(pe.aux0 >= sig && pe.aux0 < sig + n_sig)) continue;
        const long long f0 = pe.aux0 - sig;
        bool done = false;
        int pj = 0;
        for (size_t j = 0; j < items.size() && !done; ++j) {
          if (items[j].type != IT_PROGRAM) continue;
          std::vector<MOp>& pr = programs[(size_t)item_prog[(size_t)pj++]];
          if (j < i) continue;                                  // (the filter update follows the rule in program order: reads before updates)
          for (size_t q = 0; q < pr.size(); ++q)
            if (pr[q].kind == ssn::M_LOWPASS && pr[q].dst == f0 && pr[q].len == pe.rows) {
              pe.lp_dst = sig + pr[q].dst; pe.lp_src = sig + pr[q].src; pe.lp_a = pr[q].a; pe.lp_b = pr[q].b;
              pr.erase(pr.begin() + (long)q);
              done = true;
              break;
            }
        }
      }
    // Round 4: the neuron update of a dense population in the epilogue of its encoder product.  A population hop is four
    // dependent rounds - product (J += W x) -> neurons -> sparse decode over the spike list -> reduction (+ filter) - and a
    // neuron-sharded timestep adds the exchange and the filter behind it: six rounds between two exchanges.  Where the product is
    // the last writer of the current vector J before the neurons, nothing else ever reads J, and nothing between the two touches
    // the product's operands, the workgroup that owns 16 rows steps their 16 neurons itself (matvec_neurons_body).  The fused
    // operator takes the neurons' place in program order (whatever reads the previous timestep's spikes between the two still
    // comes first).  SSN_FUSE_NEURONS=0: off.
    if (!(getenv("SSN_FUSE_NEURONS") && atoi(getenv("SSN_FUSE_NEURONS")) == 0)) {
      // accesses of every operator in program order: (item index, accesses)
      std::vector<std::pair<int, std::vector<Rng>>> seq;
      {
        int pj = 0;
        for (size_t i = 0; i < items.size(); ++i) {
          if (items[i].type == IT_PROGRAM) {
            for (const MOp& op : programs[(size_t)item_prog[(size_t)pj]]) { seq.push_back({(int)i, {}}); micro_access(seq.back().second, op, false); }
            ++pj;
          } else {
            seq.push_back({(int)i, {}});
            item_access(seq.back().second, items[i]);
          }
        }
      }
      for (size_t jn = 0; jn < items.size(); ++jn) {
        Item& N = items[jn];
        if (N.type != IT_NEURONS || !(N.src >= sig && N.src < sig + n_sig)) continue;
        const int64_t j0 = N.src - sig, j1 = j0 + N.n;
        int jm = -1;
        for (size_t j = 0; j < jn; ++j)
          if (items[j].type == IT_MATVEC && items[j].dst == N.src && items[j].rows == N.n) jm = (int)j;
        if (jm < 0) continue;
        const Item& M = items[(size_t)jm];
        // (smaller populations keep the four-rows-per-workgroup product, which fills more of the chip; SSN_FUSE_MIN_ROWS for tests)
        const int fuse_min_rows = getenv("SSN_FUSE_MIN_ROWS") ? atoi(getenv("SSN_FUSE_MIN_ROWS")) : 4097;
        if ((size_t)M.cols * sizeof(T) > 48 * 1024 || M.rows < fuse_min_rows || M.phase != N.phase) continue;
        std::vector<Rng> accM;
        item_access(accM, M);
        bool ok = true;
        for (const auto& e : seq) {
          if (e.first == jm || e.first == (int)jn) continue;
          for (const Rng& r : e.second) {
            if (r.space != (const void*)sig || r.lo >= j1 || j0 >= r.hi) continue;
            if (!r.w) ok = false;                                        // another reader of J
            else if (e.first > jm && e.first < (int)jn) ok = false;      // another writer between the product and the neurons
            else if (e.first > (int)jn) ok = false;                      // (a writer behind the neurons: not the pattern)
          }
          if (e.first > jm && e.first < (int)jn && hazard(e.second, accM)) ok = false;      // the product's operands change on the way
          if (!ok) break;
        }
        if (!ok) continue;
        if (N.list) {                        // spike list in 16-neuron segments: counts per small segment, the spans of the products stay
          int* cnt = nullptr;
          const int n_small = (N.n + 15) / 16;
          CHK(dmalloc(&cnt, (int64_t)n_small * 4 + 64));
          HIPCHK(hipMemset(cnt, 0, (size_t)n_small * 4 + 64));
          scratch_bufs.push_back(cnt);
          for (Item& sp : items) if (sp.type == IT_SPMV && sp.list == N.list) { sp.count = cnt; sp.seg_len = 16; }
          N.count = cnt;
        }
        N.type = IT_MATVEC_NEURONS;
        N.Wm = M.Wm; N.aux0 = M.src; N.rows = M.rows; N.cols = M.cols; N.ld = M.ld; N.set = M.set;
        items[(size_t)jm].type = IT_NONE;
        n_fused_populations += 1;
        if (getenv("SSN_DEBUG_PLAN")) fprintf(stderr, "[ssn] encoder product %d x %d and the update of its %d neurons fused (items %d, %zu)\n", N.rows, N.cols, N.n, jm, jn);
      }
    }
    // Round 4: the input of an ensemble array assembled by the array itself.  The product ensembles of a circular convolution read
    // x = alpha_1 A + alpha_2 B of the two transforms' outputs (reference binding.py:297-317); as a lincomb operator of its own that
    // sum is one more dependent round on the network's long loop, between the transforms and the ensembles.  EnsArgs carries up to
    // four such terms already (the per-timestep PathIntegration kernel's input assembly): where a lincomb that SETS exactly the
    // array's input range from at most four terms is the range's only writer, the array its only reader, the range starts at zero and
    // nothing between the two writes a term's source, the terms move into the array and the operator leaves the plan.
    // Measured at SLAM config 3 (four lincombs fold, 217 -> 204 rounds per 64 timesteps): 101.5 us per timestep against 100.3 without,
    // twice on one box - like the transforms in serial chains, fewer rounds on the loop and no gain; opt-in (SSN_FOLD_ENS_INPUT=1).
    if (getenv("SSN_FOLD_ENS_INPUT") && atoi(getenv("SSN_FOLD_ENS_INPUT")) == 1) {
      for (size_t ei = 0; ei < items.size(); ++ei) {
        Item& E = items[ei];
        if (E.type != IT_ENS || E.ens.n_rec != 0 || E.ens.xrows || E.ens.defer || ens_round_kind(E.ens) < 0) continue;
        const int64_t x0 = E.ens.x_off, xn = (int64_t)E.ens.K * E.ens.din;
        // program-order positions: (item index, program, op index)
        int pj = 0, lp = -1, lo = -1, li = -1;
        bool ok = true;
        std::vector<std::pair<int, std::vector<Rng>>> seq;           // (item index, accesses) of every other operator
        for (size_t i = 0; i < items.size(); ++i) {
          if (items[i].type == IT_PROGRAM) {
            const int pr = item_prog[(size_t)pj++];
            for (size_t q = 0; q < programs[(size_t)pr].size(); ++q) {
              const MOp& op = programs[(size_t)pr][q];
              if (op.kind == ssn::M_LINCOMB && op.dst == x0 && op.len == xn && lp < 0 && i < ei) { lp = pr; lo = (int)q; li = (int)i; continue; }
              seq.push_back({(int)i, {}});
              micro_access(seq.back().second, op, false);
            }
          } else if (i != ei) {
            seq.push_back({(int)i, {}});
            item_access(seq.back().second, items[i]);
          }
        }
        if (lp < 0) continue;
        const MOp L = programs[(size_t)lp][(size_t)lo];
        auto lt = lin_terms.find(L.p0);
        if (lt == lin_terms.end() || lt->second.empty() || lt->second.size() > 4 || L.a != T(0) || L.b != T(1) || L.c != T(0)) continue;
        for (int64_t q = 0; q < xn && ok; ++q) if (sig_init[(size_t)(x0 + q)] != 0.0) ok = false;
        for (const auto& e : seq) {
          if (!ok) break;
          for (const Rng& r : e.second) {
            if (r.space != (const void*)sig) continue;
            if (r.lo < x0 + xn && x0 < r.hi) ok = false;                                  // someone else touches the input range
            if (r.w && e.first >= li && e.first <= (int)ei)
              for (const auto& t : lt->second) if (r.lo < t.src + xn && t.src < r.hi) ok = false;      // a source changes between the two
          }
        }
        if (!ok) continue;
        E.ens.n_rec = (int)lt->second.size();
        for (int j = 0; j < E.ens.n_rec; ++j) {
          E.ens.rec_dst[j] = x0; E.ens.rec_len[j] = xn; E.ens.rec_src[j] = lt->second[(size_t)j].src; E.ens.rec_alpha[j] = lt->second[(size_t)j].alpha;
        }
        programs[(size_t)lp].erase(programs[(size_t)lp].begin() + lo);
        n_folded_inputs += 1;
        if (getenv("SSN_DEBUG_PLAN")) fprintf(stderr, "[ssn] input lincomb of %d terms folded into ensemble array %zu (K %d)\n", E.ens.n_rec, ei, E.ens.K);
      }
    }
    // Element-wise micro-operators are cut at every range endpoint of the other operators.  The builder merges
    // neighbouring resets / hand-offs into one long operator (one fill over all accumulators of a network); as a unit
    // it would inherit the hazards of every signal it spans - the reset of an accumulator that is read in the last
    // round of step s would hold back the whole head of step s + 1.
    std::vector<int64_t> cuts;
    {
      std::vector<Rng> all_acc;
      int pj = 0;
      for (size_t i = 0; i < items.size(); ++i) {
        if (items[i].type == IT_PROGRAM) { for (const MOp& op : programs[(size_t)item_prog[(size_t)pj]]) micro_access(all_acc, op, false); ++pj; }
        else item_access(all_acc, items[i]);
      }
      for (const Rng& r : all_acc) if (r.space == (const void*)sig) { cuts.push_back(r.lo); cuts.push_back(r.hi); }
      std::sort(cuts.begin(), cuts.end());
      cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    }
    auto split_micro = [&](const MOp& op, std::vector<MOp>& out) -> int {
      const bool ew = op.kind == ssn::M_FILL || op.kind == ssn::M_AXPY_INC || op.kind == ssn::M_AXPY_SET || op.kind == ssn::M_LOWPASS ||
                      op.kind == ssn::M_LINCOMB || op.kind == ssn::M_ROW_IN || op.kind == ssn::M_ROW_OUT;
      if (!ew || op.len <= 1 || (flags & 33554432)) { out.push_back(op); return SSN_OK; }
      std::vector<int64_t> bases;                       // signal-space origins of the operator's operands
      if (op.kind != ssn::M_ROW_OUT) bases.push_back(op.dst);
      if (op.kind == ssn::M_AXPY_INC || op.kind == ssn::M_AXPY_SET || op.kind == ssn::M_LOWPASS || op.kind == ssn::M_ROW_OUT) bases.push_back(op.src);
      const std::vector<ssn::LinTerm<T>>* terms = nullptr;
      if (op.kind == ssn::M_LINCOMB) { auto f = lin_terms.find(op.p0); if (f != lin_terms.end()) { terms = &f->second; for (auto& t : *terms) bases.push_back(t.src); } }
      std::vector<int64_t> offs{0, op.len};
      for (int64_t b : bases) {
        auto lo = std::upper_bound(cuts.begin(), cuts.end(), b), hi = std::lower_bound(cuts.begin(), cuts.end(), b + op.len);
        for (auto it = lo; it < hi; ++it) offs.push_back(*it - b);
      }
      std::sort(offs.begin(), offs.end());
      offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
      if (offs.size() > 64) { out.push_back(op); return SSN_OK; }       // (pathological fragmentation: keep the operator whole)
      for (size_t q = 0; q + 1 < offs.size(); ++q) {
        const int64_t o = offs[q];
        MOp pc = op;
        pc.len = offs[q + 1] - o;
        pc.dst = op.dst + o;
        if (op.kind == ssn::M_AXPY_INC || op.kind == ssn::M_AXPY_SET || op.kind == ssn::M_LOWPASS || op.kind == ssn::M_ROW_OUT) pc.src = op.src + o;
        if (op.kind == ssn::M_ROW_IN || op.kind == ssn::M_ROW_OUT) pc.i1 = op.i1 + o;
        if (terms && o != 0) {
          std::vector<ssn::LinTerm<T>> tt = *terms;
          for (auto& t : tt) t.src += o;
          ssn::LinTerm<T>* d_terms = nullptr;
          CHK(dmalloc(&d_terms, (int64_t)std::max<size_t>(1, tt.size()) * (int64_t)sizeof(ssn::LinTerm<T>)));
          scratch_bufs.push_back(d_terms);
          if (!tt.empty()) HIPCHK(hipMemcpy(d_terms, tt.data(), tt.size() * sizeof(ssn::LinTerm<T>), hipMemcpyHostToDevice));
          lin_terms[(const void*)d_terms] = tt;
          pc.p0 = d_terms;
        }
        out.push_back(pc);
      }
      return SSN_OK;
    };
    // (opt-in, SSN_ENS_SELF_FINISH=1: measured in round 4 at SLAM config 3 - 117.9 us per timestep alone, 111.0 together with the PES
    //  fold, against 106.4 for the PES fold alone: the prologue's dependent loads lengthen every workgroup of the bandwidth-bound
    //  round, and the greedy placement does not turn the freed recurrence into a shorter cycle)
    if (!phased && getenv("SSN_ENS_SELF_FINISH") && atoi(getenv("SSN_ENS_SELF_FINISH")) == 1) CHK(fold_recurrent_filter(programs, item_prog));
    int prog_i = 0;
    for (size_t i = 0; i < items.size(); ++i) {
      const Item& it = items[i];
      if (it.type == IT_PROGRAM) {
        std::vector<MOp> pieces;
        for (const MOp& op0 : programs[(size_t)item_prog[(size_t)prog_i]]) CHK(split_micro(op0, pieces));
        for (const MOp& op0 : pieces) {
          MOp op = op0;
          op.barrier = 0;
          Unit u; u.mop = (int)mops.size(); u.phase = std::max(0, it.phase);
          micro_access(u.acc, op, true);
          mops.push_back(op);
          units.push_back(std::move(u));
        }
        ++prog_i;
      } else if (it.type != IT_NONE) {
        Unit u; u.item = (int)i; u.phase = std::max(0, it.phase);
        item_access(u.acc, it);
        units.push_back(std::move(u));
      }
    }
    for (Unit& u : units) for (const Rng& r : u.acc) u.writes = u.writes || r.w;
    // device copies: micro-operators, the block -> (operator, chunk) maps of the glue entries, body arguments
    CHK(dmalloc(&d_mops, (int64_t)std::max<size_t>(1, mops.size()) * (int64_t)sizeof(MOp)));
    if (!mops.empty()) HIPCHK(hipMemcpy(d_mops, mops.data(), mops.size() * sizeof(MOp), hipMemcpyHostToDevice));
    std::vector<ssn::GlueBlock> glue_map;
    std::vector<unsigned char> arena;
    auto put = [&](const void* src, size_t bytes) { const size_t off = (arena.size() + 15) / 16 * 16; arena.resize(off + bytes); memcpy(arena.data() + off, src, bytes); return off; };
    std::vector<long long> unit_arg(units.size(), -1);      // arena offset of a big operator's body arguments (shared by its instances)
    std::map<int, std::pair<long long, long long>> ens_parity_arg;      // ... of a self-finishing array: one copy per timestep parity
    struct Fix { size_t rl; int entry; int what; size_t off; };     // what: 0 arena, 1 glue map, 2 micro-operator
    std::vector<Fix> fixes;
    round_launches.clear();

    // One instance of a unit: (unit, timestep offset inside the launch sequence, round).
    struct Inst { int unit; int sub; int round; int lo = 0; int cnt = -1; int chain = -1; bool solo = false; };      // blocks [lo, lo + cnt) of the unit's grid (cnt < 0: all); solo: member of a serial chain
    // Chains: an element-wise micro-operator whose only hazards inside a round are with other element-wise micro-operators
    // on the SAME elements (equal ranges: a filter update behind the reduction it filters, the next step's input hand-off
    // behind that update) joins their round - one block then runs the chain's operators back to back on 256 elements, the
    // same thread on the same index in program order, with no barrier and no launch between them.
    auto chainable = [&](const Unit& u) {
      if (u.mop < 0 || (flags & 134217728)) return false;
      switch (mops[(size_t)u.mop].kind) {
        case ssn::M_FILL: case ssn::M_AXPY_INC: case ssn::M_AXPY_SET: case ssn::M_LOWPASS: case ssn::M_LINCOMB: case ssn::M_TABLE:
        case ssn::M_ROW_IN: case ssn::M_ROW_OUT: case ssn::M_PROBE: case ssn::M_REDUCE_SET: case ssn::M_REDUCE_INC: return true;
        default: return false;
      }
    };
    // Serial chains (round 4, second half).  The long loop of a SLAM timestep - clean-up -> binding -> memory -> unbinding -> gate ->
    // oscillator input, one synapse delay per hop - is partly made of operators that ONE workgroup executes: glue over 1 - 2
    // thousand elements, the gate, the argmax + gather, the transforms of the circular convolutions.  A single-workgroup unit
    // whose only predecessors in the round before its own are members of one such chain (or one such unit) joins it instead of
    // opening a round: the chain's block (RK_SOLO, solo_body) runs its members in program order with a workgroup barrier
    // between them - results pass through global memory inside one workgroup, workgroup-scope visibility is all that needs.
    // Measured at SLAM config 3 (profiles/round4_serial_chains.txt): a member costs ~2 us (one trip for its data, one for its
    // store to land before the barrier; ~4 us before the members' descriptors were staged in LDS), a transform ~9.5 us.
    //   * glue / gate / argmax members only (default): 102.7 vs 104.7 us per timestep, any cap between 8 and 24 us;
    //   * with the transforms as members (SSN_SOLO_DFT=1) the plan drops from 3.78 to 3.0 - 3.3 rounds per timestep and gets
    //     SLOWER - 110.8 us at a cap of 12 us, 114.8 at 16, 129 at 22, 139 at 26: a round now lasts as long as its longest
    //     chain (transform + members: 20 - 45 us) and the bandwidth-bound work does not fill the time under it;
    //   * neuron-sharded plans: nothing for the per-timestep phases (157.8 vs 155.3 us, ten launches either way); with the
    //     plan pipelined over the exchange 123.0 vs 125.6 us at world 1 (72 vs 76 launches per 16 timesteps) - on there too
    //     (SSN_PHASED_SOLO=0: off for sharded plans only).
    // SSN_SOLO_CHAINS=0: off; SSN_SOLO_CAP_US: longest serial chain in estimated microseconds (default 16); SSN_SOLO_MAX_LEN:
    // longest glue operator a single workgroup takes (default 2048 elements).
    // LDS bytes of a transform as a round body
    auto dft_lds = [&](const Item& it) {
      size_t lds = (size_t)(it.dft.M > 0 ? it.dft.M : it.dft.N) * (it.dft.N1 > 0 ? 4 * sizeof(float) : 3 * sizeof(float2));
      if (it.dft.inplace) {                         // in place: M + M / 8 points + M / (smallest radix) twiddles
        int rmin = 8;
        for (int q = 0; q < it.dft.nr; ++q) rmin = std::min(rmin, it.dft.radix[q]);
        lds = (size_t)(it.dft.M + it.dft.M / 8 + it.dft.M / rmin) * sizeof(float2);
      }
      return lds;
    };
    const bool solo_on = !(phased && getenv("SSN_PHASED_SOLO") && atoi(getenv("SSN_PHASED_SOLO")) == 0) && !(flags & 134217728) && !(getenv("SSN_SOLO_CHAINS") && atoi(getenv("SSN_SOLO_CHAINS")) == 0);
    const double solo_cap = getenv("SSN_SOLO_CAP_US") ? atof(getenv("SSN_SOLO_CAP_US")) : 16.0;
    const bool solo_dft = getenv("SSN_SOLO_DFT") && atoi(getenv("SSN_SOLO_DFT")) == 1;      // transforms as chain members: measured slower (below)
    const long long solo_max_len = getenv("SSN_SOLO_MAX_LEN") ? atoll(getenv("SSN_SOLO_MAX_LEN")) : 2048;
    auto solo_lat = [&](const Unit& u) -> double {        // estimated serial time of the unit on one workgroup; < 0: not eligible
      if (!solo_on) return -1.0;
      if (u.mop >= 0) {
        const MOp& op = mops[(size_t)u.mop];
        switch (op.kind) {
          // (measured on the chains of SLAM config 3: ~2 us per glue member - one trip for its data, one for the store to land
          //  before the barrier - with its descriptor already in LDS; a transform ~9.5 us)
          case ssn::M_GATE: case ssn::M_ARGMAX_GATHER: return 3.5;
          case ssn::M_FILL: case ssn::M_AXPY_INC: case ssn::M_AXPY_SET: case ssn::M_LOWPASS: case ssn::M_TABLE:
          case ssn::M_ROW_IN: case ssn::M_ROW_OUT: case ssn::M_PROBE:
            return op.len <= solo_max_len ? 1.7 + 0.5 * (double)((op.len + ssn::GLUE_CHUNK - 1) / ssn::GLUE_CHUNK) : -1.0;
          case ssn::M_LINCOMB:
            return op.len <= solo_max_len ? 1.7 + (0.5 + 0.1 * (double)op.i0) * (double)((op.len + ssn::GLUE_CHUNK - 1) / ssn::GLUE_CHUNK) : -1.0;
          case ssn::M_REDUCE_SET: case ssn::M_REDUCE_INC:
            return op.len <= solo_max_len ? 1.7 + (double)((op.len + ssn::GLUE_ROWS - 1) / ssn::GLUE_ROWS) * (0.6 + 0.15 * (double)((op.i0 + 7) / 8)) : -1.0;
          default: return -1.0;
        }
      }
      if (u.item < 0) return -1.0;            // (the exchange of a sharded cycle plan)
      const Item& it = items[(size_t)u.item];
      if (solo_dft && it.type == IT_DFT && it.dft.N1 == 0 && sizeof(T) == 4 && dft_lds(it) <= 60 * 1024) return 9.5;
      return -1.0;
    };
    // 0: no hazard, 1: hazards only on identical signal ranges (element-aligned), 2: any other hazard
    auto conflict_kind = [&](const std::vector<Rng>& x, const std::vector<Rng>& y) {
      int kind = 0;
      for (const Rng& p : x)
        for (const Rng& q : y)
          if (p.space == q.space && p.lo < q.hi && q.lo < p.hi && (p.w || q.w)) {
            if (p.space == (const void*)sig && p.lo == q.lo && p.hi == q.hi) kind = std::max(kind, 1);
            else return 2;
          }
      return kind;
    };
    // round of a new instance given the instances it may conflict with; joins a chain where it can
    auto place = [&](std::vector<Inst>& placed, size_t from, int unit, const std::vector<std::vector<Rng>>& accs, int base_round,
                     std::vector<std::vector<int>>& chains) {
      const Unit& uu = units[(size_t)unit];
      const bool can = chainable(uu);
      const long long len = can ? (long long)mops[(size_t)uu.mop].len : 0;
      int r_hard = base_round, r_soft = -1;
      std::vector<size_t> soft, hard;
      for (size_t v = from; v < placed.size(); ++v) {
        if (placed[v].round < r_hard - 1 && placed[v].round < r_soft) continue;      // can neither raise a bound nor be a chain partner
        const Unit& vv = units[(size_t)placed[v].unit];
        const bool both = can && chainable(vv) && (long long)mops[(size_t)vv.mop].len == len;
        const int k = both ? conflict_kind(accs[(size_t)unit], accs[(size_t)placed[v].unit])
                           : (hazard(accs[(size_t)unit], accs[(size_t)placed[v].unit]) ? 2 : 0);
        // (the exchange of a sharded cycle plan is no kernel: it happens in front of its round's launch, so what waits for it may
        //  run in that very round)
        if (k == 2) { r_hard = std::max(r_hard, placed[v].round + ((vv.mop < 0 && vv.item < 0) ? 0 : 1)); hard.push_back(v); }
        else if (k == 1) { r_soft = std::max(r_soft, placed[v].round); soft.push_back(v); }
      }
      // serial chain: every hard predecessor in round r_hard - 1 is a single-workgroup unit of ONE chain (or one such unit alone)
      const double my_lat = solo_lat(uu);
      if (my_lat >= 0.0 && r_hard - 1 >= base_round && r_hard - 1 > r_soft) {
        int chain = -2;               // -2: none seen, -3: not joinable, -1: one unchained instance (lone), >= 0: a chain
        size_t lone = 0;
        for (size_t v : hard) {
          if (placed[v].round != r_hard - 1) continue;
          if (solo_lat(units[(size_t)placed[v].unit]) < 0.0 || placed[v].cnt >= 0) { chain = -3; break; }
          if (placed[v].chain >= 0) {
            if (chain == -2) chain = placed[v].chain;
            else if (chain != placed[v].chain) { chain = -3; break; }
          } else {
            if (chain == -2) { chain = -1; lone = v; }
            else { chain = -3; break; }
          }
        }
        if (chain >= 0 || chain == -1) {
          double total = my_lat;
          bool ok = true;
          int transforms = uu.mop < 0 ? 1 : 0;        // (at most one per chain: solo_body runs it outside its member loops)
          if (chain >= 0 && chains[(size_t)chain].size() + 1 > (size_t)ssn::SOLO_MAX_MEMBERS) ok = false;
          if (chain == -1) { total += solo_lat(units[(size_t)placed[lone].unit]); transforms += units[(size_t)placed[lone].unit].mop < 0 ? 1 : 0; }
          else
            for (int mi : chains[(size_t)chain]) {
              const double l = solo_lat(units[(size_t)placed[(size_t)mi].unit]);
              if (l < 0.0 || placed[(size_t)mi].round != r_hard - 1) { ok = false; break; }
              total += l;
              transforms += units[(size_t)placed[(size_t)mi].unit].mop < 0 ? 1 : 0;
            }
          if (ok && transforms <= 1 && total <= solo_cap) {
            if (chain == -1) { chain = (int)chains.size(); placed[lone].chain = chain; chains.push_back({(int)lone}); }
            for (int mi : chains[(size_t)chain]) placed[(size_t)mi].solo = true;
            Inst in{unit, 0, r_hard - 1, 0, -1, chain, true};
            chains[(size_t)chain].push_back((int)placed.size());
            placed.push_back(in);
            return in.round;
          }
        }
      }
      Inst in{unit, 0, std::max(r_hard, r_soft), 0, -1, -1};
      if (r_soft >= 0 && r_soft >= r_hard) {
        int chain = -2;
        for (size_t v : soft) {
          if (placed[v].round != r_soft) continue;
          if (placed[v].chain < 0) { placed[v].chain = (int)chains.size(); chains.push_back({(int)v}); }
          if (chain == -2) chain = placed[v].chain;
          else if (chain != placed[v].chain) chain = -3;
        }
        if (chain >= 0 && placed[(size_t)chains[(size_t)chain][0]].solo && chains[(size_t)chain].size() + 1 > (size_t)ssn::SOLO_MAX_MEMBERS) chain = -3;
        if (chain >= 0) { in.chain = chain; in.solo = placed[(size_t)chains[(size_t)chain][0]].solo; chains[(size_t)chain].push_back((int)placed.size()); }
        else in.round = r_soft + 1;            // element-aligned with members of two chains: a round of its own
      }
      placed.push_back(in);
      return in.round;
    };
    // Launch sequence of a set of instances grouped by round (instances of one round are mutually independent).
    const bool no_interleave = getenv("SSN_ROUND_INTERLEAVE") && atoi(getenv("SSN_ROUND_INTERLEAVE")) == 0;      // A/B knob
    const bool head_prio = !(getenv("SSN_HEAD_PRIO") && atoi(getenv("SSN_HEAD_PRIO")) == 0);                       // A/B knob (ssn_round.hpp, k_round)
    std::vector<int> chain_tab;
    auto emit = [&](const std::vector<Inst>& insts, int n_rounds, std::vector<Launch>& out, const std::vector<std::vector<int>>& chains) {
      std::vector<std::vector<const Inst*>> by_round((size_t)n_rounds);
      for (const Inst& in : insts) by_round[(size_t)in.round].push_back(&in);
      auto in_chain = [&](const Inst* in) { return in->chain >= 0 && chains[(size_t)in->chain].size() >= 2; };
      for (int r = 0; r < n_rounds; ++r) {
        RoundLaunch rl;
        rl.round = r;
        rl.args = ssn::RoundArgs<T>{};
        rl.args.mops = d_mops; rl.args.sig = sig; rl.args.ctx = d_ctx;
        int phase = 0;
        std::vector<Launch> plain;
        auto close = [&]() {
          if (rl.args.n == 0) return;
          // interleaved dispatch of the big grids behind the latency-bound head (RoundArgs::stride)
          rl.args.head = 0; rl.args.stride = 0;
          int head = 0;
          for (int q = 0; q < rl.args.n; ++q) {
            const int k = rl.args.e[q].kind;
            if (k == ssn::RK_DFT || k == ssn::RK_SOLO || k == ssn::RK_GATE || k == ssn::RK_ARGMAX || k == ssn::RK_GLUE) head = rl.args.e[q].first + rl.args.e[q].cnt;
            else break;
          }
          rl.args.prio = (head_prio && head < rl.n_blocks) ? head : 0;      // (blocks [0, prio) raise their waves' issue priority)
          if (!no_interleave) {
            const long long m = (long long)rl.n_blocks - head;
            if (m >= 64) {
              long long st = (long long)(0.6180339887 * (double)m);
              auto gcd = [](long long a, long long b) { while (b) { const long long t = a % b; a = b; b = t; } return a; };
              while (st > 1 && gcd(st, m) != 1) --st;
              if (st > 1) { rl.args.head = head; rl.args.stride = (unsigned int)st; }
            }
          }
          Launch l; l.rl = (int)round_launches.size(); l.phase = phase;
          rl.args.pad = l.rl;                    // (launch id: read by the diagnostic stamps of k_round, SSN_ROUND_STAMPS)
          round_launches.push_back(rl);
          out.push_back(l);
          rl.args.n = 0; rl.n_blocks = 0; rl.lds = 0;
        };
        int part_lo = 0, part_cnt = -1;      // block range of the instance being emitted (set per instance below)
        // The grid's blocks are dispatched in order: single-workgroup bodies whose latency bounds the round (a DFT: ~9 us)
        // go first so that they run beside the bandwidth-bound blocks instead of behind them; then the small glue; then
        // the big grids, longest first.
        struct Pending { int kind, gx, gy, what, lo, cnt; size_t lds, off; };
        std::vector<Pending> pending;
        auto entry = [&](int kind, int gx, int gy, size_t lds, int what, size_t off) {
          Pending q{kind, std::max(1, gx), std::max(1, gy), what, 0, 0, lds, off};
          q.lo = part_cnt < 0 ? 0 : part_lo;
          q.cnt = part_cnt < 0 ? q.gx * q.gy : std::min(part_cnt, q.gx * q.gy - q.lo);
          pending.push_back(q);
        };
        auto flush_entries = [&]() {
          auto prio = [](const Pending& q) { return (q.kind == ssn::RK_DFT || q.kind == ssn::RK_SOLO) ? 0 : (q.kind == ssn::RK_GATE || q.kind == ssn::RK_ARGMAX) ? 1 : q.kind == ssn::RK_GLUE ? 2 : 3; };
          std::stable_sort(pending.begin(), pending.end(), [&](const Pending& a, const Pending& b) {
            if (prio(a) != prio(b)) return prio(a) < prio(b);
            return prio(a) == 3 && a.cnt > b.cnt;
          });
          for (const Pending& q : pending) {
            if (rl.args.n == ssn::MAX_ROUND_ENTRIES) close();
            ssn::RoundEntry& e = rl.args.e[rl.args.n];
            e.kind = q.kind; e.first = rl.n_blocks; e.gx = q.gx; e.gy = q.gy; e.args = nullptr; e.lo = q.lo; e.cnt = q.cnt;
            fixes.push_back(Fix{round_launches.size(), rl.args.n, q.what, q.off});
            rl.n_blocks += e.cnt;
            rl.lds = std::max(rl.lds, q.lds);
            rl.args.n += 1;
          }
          pending.clear();
        };
        // the exchange of a neuron-sharded cycle plan (a unit without operator): the caller's all-reduce comes before this round
        for (const Inst* in : by_round[(size_t)r])
          if (units[(size_t)in->unit].mop < 0 && units[(size_t)in->unit].item < 0) { Launch x; x.phase = -2; out.push_back(x); }
        // serial chains: one block each, members in program order behind workgroup barriers (RK_SOLO; see solo_lat)
        part_lo = 0; part_cnt = -1;
        for (const Inst* in : by_round[(size_t)r]) {
          if (!(in_chain(in) && in->solo)) continue;
          const std::vector<int>& members = chains[(size_t)in->chain];
          if (&insts[(size_t)members[0]] != in) continue;
          phase = units[(size_t)in->unit].phase;
          const int ofs = (int)chain_tab.size();
          size_t lds = 64;
          chain_tab.push_back((int)members.size());
          if (getenv("SSN_DEBUG_PLAN")) fprintf(stderr, "[ssn]   serial chain in round %d:", r);
          for (int mi : members) {
            const Inst& m = insts[(size_t)mi];
            const Unit& mu = units[(size_t)m.unit];
            if (mu.mop >= 0) {
              chain_tab.push_back(mu.mop);
              if (getenv("SSN_DEBUG_PLAN")) fprintf(stderr, " k%d/%lld(s%d)", mops[(size_t)mu.mop].kind, (long long)mops[(size_t)mu.mop].len, m.sub);
            } else {
              const Item& it = items[(size_t)mu.item];      // (a transform: solo_lat admits nothing else)
              long long& ao = unit_arg[(size_t)m.unit];
              if (ao < 0) ao = (long long)put(&it.dft, sizeof it.dft);
              chain_tab.push_back(-(int)(ao / 16) - 1);
              lds = std::max(lds, dft_lds(it));
              if (getenv("SSN_DEBUG_PLAN")) fprintf(stderr, " dft%d(s%d)", it.dft.kind, m.sub);
            }
            chain_tab.push_back(m.sub);
          }
          if (getenv("SSN_DEBUG_PLAN")) fprintf(stderr, "\n");
          entry(ssn::RK_SOLO, 1, 1, lds, 3, (size_t)ofs);
        }
        // glue: one entry for all chunked micro-operators of the round
        const size_t map_begin = glue_map.size();
        for (const Inst* in : by_round[(size_t)r]) {
          const Unit& u = units[(size_t)in->unit];
          phase = u.phase;
          if (u.mop < 0) continue;
          if (in_chain(in) && in->solo) continue;
          const MOp& op = mops[(size_t)u.mop];
          if (op.kind == ssn::M_GATE || op.kind == ssn::M_ARGMAX_GATHER) continue;
          if (in_chain(in)) {
            const std::vector<int>& members = chains[(size_t)in->chain];
            if (&insts[(size_t)members[0]] != in) continue;                 // emitted with its chain's first member
            const int ofs = (int)chain_tab.size();
            if (getenv("SSN_DEBUG_PLAN")) {
              fprintf(stderr, "[ssn]   chain in round %d (len %lld):", r, (long long)op.len);
              for (int mi : members) fprintf(stderr, " k%d", mops[(size_t)units[(size_t)insts[(size_t)mi].unit].mop].kind);
              fprintf(stderr, "\n");
            }
            chain_tab.push_back((int)members.size());
            for (int mi : members) { chain_tab.push_back(units[(size_t)insts[(size_t)mi].unit].mop); chain_tab.push_back(insts[(size_t)mi].sub); }
            const int blocks = (int)std::max<long long>(1, (op.len + ssn::GLUE_ROWS - 1) / ssn::GLUE_ROWS);
            for (int c = 0; c < blocks; ++c) glue_map.push_back(ssn::GlueBlock{-(ofs + 1), c});
            continue;
          }
          const long long per = glue_row_kind(op.kind) ? ssn::GLUE_ROWS : ssn::GLUE_CHUNK;
          const int chunks = (int)std::max<long long>(1, (op.len + per - 1) / per);
          for (int c = 0; c < chunks; ++c) glue_map.push_back(ssn::GlueBlock{u.mop, c | (in->sub << 24)});
        }
        if (glue_map.size() > map_begin) entry(ssn::RK_GLUE, (int)(glue_map.size() - map_begin), 1, 64, 1, map_begin);
        for (const Inst* in : by_round[(size_t)r]) {
          const Unit& u = units[(size_t)in->unit];
          part_lo = in->lo; part_cnt = in->cnt;
          if (in_chain(in) && in->solo) continue;
          if (u.mop < 0 && u.item < 0) continue;        // (the exchange marker)
          if (u.mop >= 0) {
            const MOp& op = mops[(size_t)u.mop];
            if (op.kind == ssn::M_GATE) entry(ssn::RK_GATE, 1, 1, 64, 2, (size_t)u.mop);
            else if (op.kind == ssn::M_ARGMAX_GATHER) entry(ssn::RK_ARGMAX, 1, 1, 64, 2, (size_t)u.mop);
            continue;
          }
          const Item& it = items[(size_t)u.item];
          const size_t xb = (size_t)it.cols * sizeof(T);
          long long& ao = unit_arg[(size_t)in->unit];
          switch (it.type) {
            case IT_MATVEC:
              if (xb <= 48 * 1024) {
                const bool r1 = it.rows <= matvec_r1_max;
                if (r1) {
                  ssn::MatvecArgs<T> a{it.Wm, it.src, it.dst, it.rows, it.cols, it.ld, it.set};
                  if (ao < 0) ao = (long long)put(&a, sizeof a);
                } else {                 // (the 16-rows-per-workgroup body is matvec_neurons_body's: no population behind this product)
                  ssn::MatvecNeuronsArgs<T> a{};
                  a.mv = ssn::MatvecArgs<T>{it.Wm, it.src, it.dst, it.rows, it.cols, it.ld, it.set};
                  if (ao < 0) ao = (long long)put(&a, sizeof a);
                }
                entry(r1 ? ssn::RK_MATVEC_R1 : ssn::RK_MATVEC_R4, r1 ? (it.rows + 3) / 4 : (it.rows + 15) / 16, 1, (xb + 15) / 16 * 16 + 64, 0, (size_t)ao);
                continue;
              }
              break;
            case IT_ENS: {
              const ssn::EnsArgs<T>& a = it.ens;
              const int kind = ens_round_kind(a);
              if (kind >= 0) {
                if (a.defer == 2) {          // the timestep offset of the instance decides which set of partial sums it writes: two copies
                  auto f = ens_parity_arg.find(in->unit);
                  if (f == ens_parity_arg.end()) {
                    ssn::EnsArgs<T> c0 = a, c1 = a;
                    c0.sub = 0; c1.sub = 1;
                    const long long o0 = (long long)put(&c0, sizeof c0), o1 = (long long)put(&c1, sizeof c1);
                    f = ens_parity_arg.insert({in->unit, {o0, o1}}).first;
                  }
                  entry(kind, ens_round_blocks(a), 1, 512, 0, (size_t)((in->sub & 1) ? f->second.second : f->second.first));
                  continue;
                }
                if (ao < 0) ao = (long long)put(&a, sizeof a);
                entry(kind, ens_round_blocks(a), 1, 512, 0, (size_t)ao);
                continue;
              }
              break;
            }
            case IT_SPMV: {
              ssn::SpmvArgs<T> a{it.Wm, it.ld, it.src, it.cols, it.rows, it.dst, it.ld, it.n, it.list, it.count, it.seg, it.seg_len};
              const size_t lds = 272 * sizeof(int) + (it.list ? 0 : (size_t)it.cols * sizeof(int));
              if (lds <= 60 * 1024) {
                if (ao < 0) ao = (long long)put(&a, sizeof a);
                entry(ssn::RK_SPMV, (it.rows + 255) / 256, it.n, lds, 0, (size_t)ao);
                continue;
              }
              break;
            }
            case IT_NEURONS: {
              ssn::NeuronsArgs<T> a{it.np, it.src, it.dst, it.V, it.R, it.n, it.scalar, it.list, it.count};
              if (ao < 0) ao = (long long)put(&a, sizeof a);
              entry(ssn::RK_NEURONS, (it.n + 255) / 256, 1, 64, 0, (size_t)ao);
              continue;
            }
            case IT_MATVEC_NEURONS: {
              ssn::MatvecNeuronsArgs<T> a{ssn::MatvecArgs<T>{it.Wm, it.aux0, nullptr, it.rows, it.cols, it.ld, it.set},
                                          ssn::NeuronsArgs<T>{it.np, it.src, it.dst, it.V, it.R, it.n, it.scalar, it.list, it.count}};
              if (ao < 0) ao = (long long)put(&a, sizeof a);
              entry(ssn::RK_MATVEC_NEURONS, (it.rows + 15) / 16, 1, (xb + 15) / 16 * 16 + 64, 0, (size_t)ao);
              continue;
            }
            case IT_DFT: {
              const size_t lds = dft_lds(it);
              if (lds <= 64 * 1024 && it.dft.N1 == 0) {      // (the four-step engine is launched on its own: see dft_body<ROUND>)
                if (ao < 0) ao = (long long)put(&it.dft, sizeof it.dft);
                entry(ssn::RK_DFT, 1, 1, lds, 0, (size_t)ao);
                continue;
              }
              break;
            }
            case IT_GRID_LHS: {
              ssn::GridLhsArgs<T> a{it.src, it.aux0, it.ld, it.dst, it.cols, it.rows, it.cols / 2};
              if (ao < 0) ao = (long long)put(&a, sizeof a);
              entry(ssn::RK_GRID_LHS, (it.rows * (it.cols / 2) + 255) / 256, 1, 64, 0, (size_t)ao);
              continue;
            }
            case IT_GRID_DOT: {
              ssn::GridDotArgs<T> a{it.src, it.cols, it.Wm, it.ld, it.dst, it.n, it.cols, it.rows};
              if (ao < 0) ao = (long long)put(&a, sizeof a);
              entry(ssn::RK_GRID_DOT, (it.n + 15) / 16, it.rows, xb, 0, (size_t)ao);
              continue;
            }
            case IT_PES: {
              ssn::PesArgs<T> a{it.Wm, it.aux0, it.aux1, it.rows, it.cols, it.ld, it.scalar, it.lp_dst, it.lp_src, it.lp_a, it.lp_b};
              if (ao < 0) ao = (long long)put(&a, sizeof a);
              entry(ssn::RK_PES, (it.cols + 1023) / 1024, (it.rows + ssn::PES_ROWS - 1) / ssn::PES_ROWS, 64, 0, (size_t)ao);
              continue;
            }
            case IT_VOJA: {
              ssn::VojaArgs<T> a{it.Wm, it.src, it.aux0, it.aux1, it.aux2, it.rows, it.cols, it.ld, it.scalar};
              if (ao < 0) ao = (long long)put(&a, sizeof a);
              entry(ssn::RK_VOJA, (it.rows + 4 * ssn::VOJA_ROWS_PER_WAVE - 1) / (4 * ssn::VOJA_ROWS_PER_WAVE), 1, 64, 0, (size_t)ao);
              continue;
            }
            default: break;
          }
          Launch l; l.item = u.item; l.phase = u.phase;
          plain.push_back(l);
        }
        for (const Launch& l : plain) out.push_back(l);      // the plain launches first (ensemble arrays: the long ones), then the round's grid
        flush_entries();
        close();
      }
    };

    // (diagnostic) the chain of binding hazards behind the last instance of a pipelined plan: what its period is made of
    auto print_critical_chain = [&](const std::vector<Inst>& all, const std::vector<std::vector<Rng>>& accs) {
      size_t cur = 0;
      for (size_t i = 0; i < all.size(); ++i) if (all[i].round >= all[cur].round) cur = i;
      fprintf(stderr, "[ssn] critical chain (backwards from the last round):\n");
      for (int hop = 0; hop < 70 && all[cur].round > 0; ++hop) {
        const Unit& u = units[(size_t)all[cur].unit];
        if (u.mop >= 0) fprintf(stderr, "[ssn]   round %3d step %2d micro %d/%lld dst %lld\n", all[cur].round, all[cur].sub, mops[(size_t)u.mop].kind, (long long)mops[(size_t)u.mop].len, (long long)mops[(size_t)u.mop].dst);
        else if (u.item < 0) fprintf(stderr, "[ssn]   round %3d step %2d EXCHANGE\n", all[cur].round, all[cur].sub);
        else fprintf(stderr, "[ssn]   round %3d step %2d item %d type %d rows %d cols %d\n", all[cur].round, all[cur].sub, u.item, items[(size_t)u.item].type, items[(size_t)u.item].rows, items[(size_t)u.item].cols);
        size_t prev = cur;
        for (size_t v = cur; v-- > 0;)
          if (all[v].round == all[cur].round - 1 && hazard(accs[(size_t)all[cur].unit], accs[(size_t)all[v].unit])) { prev = v; break; }
        if (prev == cur)          // a chain member: its predecessor sits in the same round, in the same block
          for (size_t v = cur; v-- > 0;)
            if (all[v].round == all[cur].round && all[v].chain >= 0 && all[v].chain == all[cur].chain) { prev = v; break; }
        if (prev == cur) break;
        cur = prev;
      }
    };

    // ---- one timestep: the eager / profiled / phased sequence --------------------------------------------------------
    std::vector<Inst> one;
    int n_rounds = 0, phase1_base = 0;
    bool in_phase1 = false;
    std::vector<std::vector<Rng>> acc_all(units.size());
    for (size_t u = 0; u < units.size(); ++u) acc_all[u] = units[u].acc;
    std::vector<std::vector<int>> chains_one;
    size_t phase1_from = 0;
    for (size_t u = 0; u < units.size(); ++u) {
      if (units[u].phase == 1 && !in_phase1) { in_phase1 = true; phase1_base = n_rounds; phase1_from = one.size(); }
      // (no chain across the exchange: a phase-1 instance only sees soft hazards with phase-1 instances; hard ones with all)
      int r = place(one, 0, (int)u, acc_all, in_phase1 ? phase1_base : 0, chains_one);
      if (in_phase1 && one.back().chain >= 0 && (size_t)chains_one[(size_t)one.back().chain][0] < phase1_from) {
        // joined a chain that started before the exchange: undo, take the first round of phase 1 instead
        chains_one[(size_t)one.back().chain].pop_back();
        const int r_own = r + (one.back().solo ? 1 : 0);        // (a serial member sat one round before its hard bound)
        one.back().chain = -1; one.back().solo = false; one.back().round = std::max(r_own, phase1_base);
        r = one.back().round;
      }
      n_rounds = std::max(n_rounds, r + 1);
    }
    launch_list.clear();
    emit(one, n_rounds, launch_list, chains_one);
    n_serial_chains = 0;
    for (size_t c = 0; c < chains_one.size(); ++c) if (chains_one[c].size() >= 2 && one[(size_t)chains_one[c][0]].solo) n_serial_chains += 1;
    launches_per_step = (int)launch_list.size();
    const int launches_unpipelined = launches_per_step;

    // ---- G timesteps software-pipelined: the sequence a step graph replays -------------------------------------------
    // Instance (u, s) takes the earliest round that its hazards against the instances of timesteps s and s - 1 allow
    // (a writer conflicts with itself one step earlier, so older steps are ordered transitively).  The recurrence of a
    // SLAM step runs through synapse states only: the long un-filtered chain of step s (two circular convolutions in
    // series) ends in filter updates that step s + 1 reads only to update other filters, so the head of step s + 1 -
    // the bandwidth-bound oscillators and encoder products - runs in the rounds where step s has one latency-bound
    // workgroup left.  Clock readers take the step number from StepCtx plus their own offset; the clock itself
    // advances once, behind the last round (k_advance).
    graph_list.clear();
    graph_rounds = 0;
    const int G = steps_per_graph;
    if (!phased && G > 1 && !(flags & 8388608) && units.size() * (size_t)G <= 40000) {
      std::vector<int> keep;
      for (size_t u = 0; u < units.size(); ++u)
        if (!(units[u].mop >= 0 && mops[(size_t)units[u].mop].kind == ssn::M_STEP_END)) keep.push_back((int)u);
      std::vector<std::vector<Rng>> acc_nc(units.size());     // access lists without the clock
      for (int u : keep) { if (units[(size_t)u].mop >= 0) micro_access(acc_nc[(size_t)u], mops[(size_t)units[(size_t)u].mop], false); else acc_nc[(size_t)u] = units[(size_t)u].acc; }
      std::vector<Inst> all;
      std::vector<std::vector<int>> chains_all;
      all.reserve(keep.size() * (size_t)G);
      int nr = 0;
      const size_t per = keep.size();
      for (int st = 0; st < G; ++st)
        for (size_t q = 0; q < per; ++q) {
          const int u = keep[q];
          const size_t lo = st > 0 ? (size_t)(st - 1) * per : 0;
          const int r = place(all, lo, u, acc_nc, 0, chains_all);
          all.back().sub = st;
          nr = std::max(nr, r + 1);
        }
      if (getenv("SSN_DEBUG_PLAN")) print_critical_chain(all, acc_nc);
      if (getenv("SSN_DEBUG_DEPS")) {
        // what every instance of the steady-state rounds waits for in the round before it (diagnostic for in-launch dependencies)
        auto name = [&](const Inst& in) {
          const Unit& u = units[(size_t)in.unit];
          char buf[160];
          if (u.mop >= 0) snprintf(buf, sizeof buf, "micro k%d len %lld dst %lld", mops[(size_t)u.mop].kind, (long long)mops[(size_t)u.mop].len, (long long)mops[(size_t)u.mop].dst);
          else snprintf(buf, sizeof buf, "item %d type %d rows %d cols %d n %d", u.item, items[(size_t)u.item].type, items[(size_t)u.item].rows, items[(size_t)u.item].cols, items[(size_t)u.item].n);
          return std::string(buf);
        };
        const int r_lo = atoi(getenv("SSN_DEBUG_DEPS")), r_hi = r_lo + 4;
        for (size_t i = 0; i < all.size(); ++i) {
          if (all[i].round < r_lo || all[i].round >= r_hi) continue;
          fprintf(stderr, "[ssn] deps: round %d step %d %s\n", all[i].round, all[i].sub, name(all[i]).c_str());
          for (size_t v = 0; v < all.size(); ++v) {
            if (v == i || all[v].round != all[i].round - 1) continue;
            const std::vector<Rng>& x = acc_nc[(size_t)all[i].unit];
            const std::vector<Rng>& y = acc_nc[(size_t)all[v].unit];
            bool raw = false, war = false, waw = false;
            for (const Rng& p : x) for (const Rng& q : y)
              if (p.space == q.space && p.lo < q.hi && q.lo < p.hi) { if (!p.w && q.w) raw = true; if (p.w && !q.w) war = true; if (p.w && q.w) waw = true; }
            if ((raw || war || waw) && (v < i)) fprintf(stderr, "[ssn] deps:     <- step %d %s%s%s%s\n", all[v].sub, name(all[v]).c_str(), raw ? " RAW" : "", war ? " WAR" : "", waw ? " WAW" : "");
          }
        }
      }
      if (!(flags & 16777216)) balance_rounds(all, nr, per, [&](int a, int b) { return hazard(acc_nc[(size_t)a], acc_nc[(size_t)b]); },
                                              [&](int u, double* us, double* lat, int* blocks) { unit_cost(units[(size_t)u].mop, units[(size_t)u].item, us, lat, blocks); });
      emit(all, nr, graph_list, chains_all);
      graph_rounds = nr;
      launches_per_step = ((int)graph_list.size() + G - 1) / G;       // (average of the replayed sequence)
    }

    // ---- neuron-sharded models: what runs between two exchanges - the updates of timestep s (phase 1) and timestep s + 1 up to
    // its exchange (phase 0) - as ONE set of rounds.  A phase-0 operator of s + 1 waits only for the phase-1 operators whose
    // results it reads: table rows, the encoder products of tabulated inputs and whatever else does not hang on a filter state
    // start beside the filter updates, PES and Voja instead of behind them.  The clock is read with an offset (phase-0
    // instances: + 1) and advanced once behind the last round, as in the pipelined step graph.
    phase2_list.clear();
    if (phased && !(flags & 8388608) && !getenv("SSN_PHASE2_SEQUENTIAL")) {
      std::vector<std::vector<Rng>> acc_p(units.size());
      for (size_t u = 0; u < units.size(); ++u) { if (units[u].mop >= 0) micro_access(acc_p[u], mops[(size_t)units[u].mop], false); else acc_p[u] = units[u].acc; }
      std::vector<Inst> comb;
      std::vector<std::vector<int>> chains_comb;
      int nr = 0;
      for (int ph = 1; ph >= 0; --ph)
        for (size_t u = 0; u < units.size(); ++u) {
          if (units[u].phase != ph) continue;
          if (units[u].mop >= 0 && mops[(size_t)units[u].mop].kind == ssn::M_STEP_END) continue;
          const int r = place(comb, 0, (int)u, acc_p, 0, chains_comb);
          comb.back().sub = ph == 0 ? 1 : 0;
          nr = std::max(nr, r + 1);
        }
      emit(comb, nr, phase2_list, chains_comb);
      if (getenv("SSN_DEBUG_PLAN")) fprintf(stderr, "[ssn] phased plan: %zu launches per timestep apart (phase 0 + phase 1), %zu with the updates of s and the head of s + 1 planned together\n", launch_list.size(), phase2_list.size());
      launches_per_step = (int)phase2_list.size();
    }

    // ---- neuron-sharded models, pipelined over the exchange (round 4).  The plan above still starts every
    // timestep from an empty chip behind its exchange: ten rounds per timestep, none of them shared with another timestep.  Here
    // the exchange of timestep s is a unit of the plan like any other - it reads and writes the exchanged signal ranges, so the
    // operators that complete the partial sums come before it and their consumers (synapse filters, probes) behind it - and C
    // timesteps are placed together as in the unsharded step graph: what does not hang on the exchange of s (the head of
    // s + 1: tabulated inputs, encoder products, neuron updates of populations whose input is ready) shares rounds with the
    // tail of s.  The launch sequence is cut at the exchanges into C + 1 segments; the caller runs segment, all-reduce,
    // segment, ... (phase 3), the clock advances by C behind the last one.  SSN_CYCLE_STEPS: C (default 16; 0: off).
    cycle_segs.clear();
    cycle_steps = 0;
    {
      const int C = getenv("SSN_CYCLE_STEPS") ? atoi(getenv("SSN_CYCLE_STEPS")) : 16;
      if (phased && C > 1 && C <= 64 && !(flags & 8388608) && (units.size() + 1) * (size_t)C <= 40000) {
        const int xu = (int)units.size();
        {
          Unit x; x.phase = 0; x.writes = true;
          for (auto& r : exchange) x.acc.push_back(Rng{(const void*)sig, (int64_t)r.lo, (int64_t)r.hi, true});
          units.push_back(std::move(x));
          unit_arg.push_back(-1);
        }
        std::vector<std::vector<Rng>> acc_c(units.size());
        std::vector<int> order;              // program order of one timestep with the exchange between its phases
        bool seen1 = false;
        for (int u = 0; u < xu; ++u) {
          if (units[(size_t)u].mop >= 0) micro_access(acc_c[(size_t)u], mops[(size_t)units[(size_t)u].mop], false); else acc_c[(size_t)u] = units[(size_t)u].acc;
          if (units[(size_t)u].mop >= 0 && mops[(size_t)units[(size_t)u].mop].kind == ssn::M_STEP_END) continue;
          if (units[(size_t)u].phase == 1 && !seen1) { seen1 = true; order.push_back(xu); }
          order.push_back(u);
        }
        acc_c[(size_t)xu] = units[(size_t)xu].acc;
        if (seen1) {
          std::vector<Inst> all;
          std::vector<std::vector<int>> chains_c;
          int nr = 0;
          const size_t per = order.size();
          for (int st = 0; st < C; ++st)
            for (size_t q = 0; q < per; ++q) {
              const size_t lo = st > 0 ? (size_t)(st - 1) * per : 0;
              const int r = place(all, lo, order[q], acc_c, 0, chains_c);
              all.back().sub = st;
              nr = std::max(nr, r + 1);
            }
          if (!(flags & 16777216))
            balance_rounds(all, nr, per, [&](int a, int b) { return hazard(acc_c[(size_t)a], acc_c[(size_t)b]); },
                           [&](int u, double* us, double* lat, int* blocks) {
                             if (u == xu) { *us = 0.0; *lat = 0.0; *blocks = 0; return; }
                             unit_cost(units[(size_t)u].mop, units[(size_t)u].item, us, lat, blocks);
                           });
          if (getenv("SSN_DEBUG_PLAN")) print_critical_chain(all, acc_c);
          std::vector<Launch> seq;
          emit(all, nr, seq, chains_c);
          cycle_segs.emplace_back();
          for (const Launch& l : seq) {
            if (l.phase == -2) { cycle_segs.emplace_back(); continue; }
            cycle_segs.back().push_back(l);
          }
          if ((int)cycle_segs.size() == C + 1) {
            cycle_steps = C;
            launches_per_step = ((int)(seq.size() - (size_t)C) + C - 1) / C;       // (average over the cycle)
            if (getenv("SSN_DEBUG_PLAN")) {
              fprintf(stderr, "[ssn] sharded plan pipelined over the exchange: %d timesteps in %d rounds, %zu launches, segments:", C, nr, seq.size() - (size_t)C);
              for (auto& sg : cycle_segs) fprintf(stderr, " %zu", sg.size());
              fprintf(stderr, "\n");
            }
          } else {
            cycle_segs.clear();
          }
        }
        units.pop_back();
        unit_arg.pop_back();
      }
    }

    T* d_arena = nullptr; ssn::GlueBlock* d_map = nullptr;
    CHK(dmalloc(&d_arena, (int64_t)arena.size() + 16));
    round_bufs.push_back(d_arena);
    CHK(dmalloc(&d_map, (int64_t)(glue_map.size() + 1) * (int64_t)sizeof(ssn::GlueBlock)));
    round_bufs.push_back(d_map);
    if (!arena.empty()) HIPCHK(hipMemcpy(d_arena, arena.data(), arena.size(), hipMemcpyHostToDevice));
    if (!glue_map.empty()) HIPCHK(hipMemcpy(d_map, glue_map.data(), glue_map.size() * sizeof(ssn::GlueBlock), hipMemcpyHostToDevice));
    int* d_chain = nullptr;
    CHK(dmalloc(&d_chain, (int64_t)(chain_tab.size() + 1) * 4));
    round_bufs.push_back(d_chain);
    if (!chain_tab.empty()) HIPCHK(hipMemcpy(d_chain, chain_tab.data(), chain_tab.size() * 4, hipMemcpyHostToDevice));
    for (RoundLaunch& rl : round_launches) { rl.args.chain = d_chain; rl.args.arena = (const unsigned char*)d_arena; }
    for (const Fix& f : fixes) {
      ssn::RoundEntry& e = round_launches[f.rl].args.e[f.entry];
      if (f.what == 0) e.args = (const unsigned char*)d_arena + f.off;
      else if (f.what == 1) e.args = d_map + f.off;
      else if (f.what == 3) e.args = d_chain + f.off;
      else e.args = d_mops + f.off;
    }
    if (getenv("SSN_DEBUG_PLAN")) {
      fprintf(stderr, "[ssn] round plan: %zu units in %d rounds, %d launches per timestep; %d timesteps pipelined: %d rounds, %zu launches\n",
              units.size(), n_rounds, launches_unpipelined, G, graph_rounds, graph_list.size());
      for (const Launch& l : launch_list) {
        if (l.rl < 0) { const Item& it = items[(size_t)l.item]; fprintf(stderr, "[ssn]   plain item %d type %d rows %d cols %d n %d\n", l.item, it.type, it.rows, it.cols, it.n); continue; }
        const RoundLaunch& rl = round_launches[(size_t)l.rl];
        fprintf(stderr, "[ssn]   round %2d phase %d: %d blocks, %zu B LDS:", rl.round, l.phase, rl.n_blocks, rl.lds);
        for (int q = 0; q < rl.args.n; ++q) fprintf(stderr, " %d[%dx%d]", rl.args.e[q].kind, rl.args.e[q].gx, rl.args.e[q].gy);
        fprintf(stderr, "\n");
        for (const Inst& in : one)
          if (in.round == rl.round && units[(size_t)in.unit].mop >= 0) {
            const MOp& o = mops[(size_t)units[(size_t)in.unit].mop];
            fprintf(stderr, "[ssn]       micro %d/%lld dst %lld src %lld\n", o.kind, (long long)o.len, (long long)o.dst, (long long)o.src);
          }
      }
      for (const Launch& l : graph_list) {
        if (l.rl < 0) { fprintf(stderr, "[ssn]   pipelined: plain item %d type %d\n", l.item, items[(size_t)l.item].type); continue; }
        const RoundLaunch& rl = round_launches[(size_t)l.rl];
        fprintf(stderr, "[ssn]   pipelined round %3d: %6d blocks, %5zu B LDS:", rl.round, rl.n_blocks, rl.lds);
        for (int q = 0; q < rl.args.n; ++q) fprintf(stderr, " %d[%d]", rl.args.e[q].kind, rl.args.e[q].cnt);
        fprintf(stderr, "\n");
      }
    }
    return SSN_OK;
  }

  hipError_t launch_one(const Launch& l) {
    if (l.rl >= 0) { const RoundLaunch& rl = round_launches[(size_t)l.rl]; return ssn::launch_round<T>(stream, rl.args, rl.n_blocks, rl.lds); }
    return launch_item(items[(size_t)l.item], nullptr, nullptr);
  }

  // Adjacent items of one kind with no data hazard between them share a launch (blockIdx.y selects the item).
  void merge_adjacent_items() {
    const int n = (int)items.size();
    auto xlds = [&](const Item& it) { return (size_t)it.cols * sizeof(T) <= 48 * 1024; };
    for (int i = 0; i < n; ++i) {
      Item& lead = items[(size_t)i];
      if (lead.merged || !(lead.type == IT_MATVEC || lead.type == IT_NEURONS || lead.type == IT_DFT || lead.type == IT_ENS || lead.type == IT_SPMV)) continue;
      if (lead.type == IT_ENS && (lead.dominant || lead.ens.defer)) continue;
      const int cap = lead.type == IT_ENS ? ssn::MAX_ENS_BATCH : ssn::MAX_BATCH;
      int k = 1;
      while (i + k < n && k < cap && items[(size_t)(i + k)].type == lead.type && items[(size_t)(i + k)].phase == lead.phase) {
        const Item& nx = items[(size_t)(i + k)];
        if (lead.type == IT_MATVEC && xlds(nx) != xlds(lead)) break;
        if (lead.type == IT_ENS && (nx.dominant || nx.ens.din != lead.ens.din || nx.ens.dout != lead.ens.dout || nx.ens.fast != lead.ens.fast)) break;
        bool indep = true;
        for (int d : item_deps[(size_t)(i + k)]) if (d >= i && d < i + k) indep = false;
        if (!indep) break;
        ++k;
      }
      lead.batch = k;
      for (int q = 1; q < k; ++q) items[(size_t)(i + q)].merged = true;
    }
    int launches = 0;
    for (auto& it : items) launches += it.merged ? 0 : 1;
    if (!core_empty) launches_per_step = launches - (can_fuse ? 1 : 0);
  }

  // Read / write sets of micro-operators and plan items (signal ranges, buffers and scratch memory by base pointer)
  void acc_sig(std::vector<Rng>& a, int64_t lo, int64_t len, bool w) const { if (len > 0) a.push_back(Rng{(const void*)sig, lo, lo + len, w}); }
  static void acc_ptr(std::vector<Rng>& a, const void* p, bool w) { if (p) a.push_back(Rng{p, 0, 1, w}); }
  // scattered destinations (decoded rows of an ensemble array) as runs of consecutive signals
  void acc_index_list(std::vector<Rng>& a, const void* dev_idx, bool w) const {
    for (auto& h : host_idx)
      if (h.first == dev_idx) {
        std::vector<int32_t> v(h.second.first, h.second.first + h.second.second);
        std::sort(v.begin(), v.end());
        for (size_t j = 0; j < v.size();) {
          size_t e = j + 1;
          while (e < v.size() && v[e] <= v[e - 1] + 1) ++e;
          acc_sig(a, v[j], (int64_t)v[e - 1] - v[j] + 1, w);
          j = e;
        }
      }
  }
  void micro_access(std::vector<Rng>& a, const MOp& op, bool with_clock = false) const {
    const void* clock = (const void*)d_ctx;
    switch (op.kind) {
      case ssn::M_FILL: acc_sig(a, op.dst, op.len, true); break;
      case ssn::M_TABLE: case ssn::M_ROW_IN: acc_sig(a, op.dst, op.len, true); if (with_clock) acc_ptr(a, clock, false); break;
      case ssn::M_AXPY_INC: case ssn::M_LOWPASS: acc_sig(a, op.src, op.len, false); acc_sig(a, op.dst, op.len, true); break;
      case ssn::M_AXPY_SET: acc_sig(a, op.src, op.len, false); acc_sig(a, op.dst, op.len, true); break;
      case ssn::M_LINCOMB: {
        auto it = lin_terms.find(op.p0);
        if (it != lin_terms.end()) for (auto& t : it->second) acc_sig(a, t.src, op.len, false);
        acc_sig(a, op.dst, op.len, true);
        break;
      }
      case ssn::M_MATVEC_INC: case ssn::M_MATVEC_SET: acc_sig(a, op.src, op.i0, false); acc_sig(a, op.dst, op.len, true); acc_ptr(a, op.p0, false); break;
      case ssn::M_GATE: acc_sig(a, op.src, 2 * op.len + 1, false); acc_sig(a, op.dst, op.len, true); break;
      case ssn::M_ARGMAX_GATHER: acc_ptr(a, op.p1, false); acc_ptr(a, op.p0, false); acc_sig(a, op.dst, op.len, true); break;
      case ssn::M_PROBE: case ssn::M_ROW_OUT: acc_sig(a, op.src, op.len, false); if (with_clock) acc_ptr(a, clock, false); break;
      case ssn::M_REDUCE_SET: case ssn::M_REDUCE_INC: acc_ptr(a, op.p0, false); acc_sig(a, op.dst, op.len, true); break;
      case ssn::M_ENS_FINISH: acc_ptr(a, op.p0, false); acc_index_list(a, op.p1, true); break;
      case ssn::M_STEP_END: if (with_clock) acc_ptr(a, clock, true); break;      // (item plan: the tail program is a join point anyway)
      default: break;
    }
  }
  void item_access(std::vector<Rng>& a, const Item& it) const {
    switch (it.type) {
      case IT_ENS:
        if (it.ens.direct) acc_index_list(a, (const void*)it.ens.didx, true);
        acc_sig(a, it.ens.x_off, (int64_t)it.ens.K * it.ens.din, false);
        for (int j = 0; j < it.ens.n_rec; ++j) acc_sig(a, it.ens.rec_src[j], it.ens.rec_len[j], false);      // (folded input terms)
        acc_ptr(a, it.ens.partials, true); acc_ptr(a, it.ens.V, true); acc_ptr(a, it.ens.R, true);
        acc_ptr(a, it.ens.enc, false); acc_ptr(a, it.ens.bias, false); acc_ptr(a, it.ens.dec, false);
        break;
      case IT_DFT:
        acc_sig(a, it.src - sig, it.cols, false); acc_sig(a, it.dst - sig, it.rows, true); acc_ptr(a, it.dft.tw, false);
        break;
      case IT_MATVEC: case IT_MATVEC_ORDERED:
        acc_sig(a, it.src - sig, it.cols, false); acc_ptr(a, it.Wm, false);
        if (it.dst >= sig && it.dst < sig + n_sig) acc_sig(a, it.dst - sig, it.rows, true); else acc_ptr(a, it.dst, true);
        break;
      case IT_GRID_LHS:
        acc_ptr(a, it.src, false); acc_ptr(a, it.aux0, false); acc_ptr(a, it.dst, true);
        break;
      case IT_GRID_GEMM: case IT_GRID_DOT:
        acc_ptr(a, it.src, false); acc_ptr(a, it.Wm, false); acc_ptr(a, it.dst, true);
        break;
      case IT_ARGMAX_PART:
        acc_ptr(a, it.src, false); acc_ptr(a, it.dst, true);
        break;
      case IT_SPMV:
        acc_sig(a, it.src - sig, it.cols, false); acc_ptr(a, it.Wm, false); acc_ptr(a, it.list, false); acc_ptr(a, it.count, false); acc_ptr(a, it.dst, true);
        break;
      case IT_NEURONS:
        acc_sig(a, it.src - sig, it.n, false); acc_sig(a, it.dst - sig, it.n, true); acc_ptr(a, it.V, true); acc_ptr(a, it.R, true);
        acc_ptr(a, it.list, true); acc_ptr(a, it.count, true);
        break;
      case IT_MATVEC_NEURONS:     // src: the current vector J (read unless the product sets), aux0: the product's input
        acc_sig(a, it.aux0 - sig, it.cols, false); acc_ptr(a, it.Wm, false);
        if (!it.set) acc_sig(a, it.src - sig, it.n, false);
        acc_sig(a, it.dst - sig, it.n, true); acc_ptr(a, it.V, true); acc_ptr(a, it.R, true);
        acc_ptr(a, it.list, true); acc_ptr(a, it.count, true);
        break;
      case IT_PES:
        acc_ptr(a, it.Wm, true); acc_sig(a, it.aux0 - sig, it.rows, false); acc_sig(a, it.aux1 - sig, it.cols, false);
        if (it.lp_dst) { acc_sig(a, it.lp_dst - sig, it.rows, true); acc_sig(a, it.lp_src - sig, it.rows, false); }
        break;
      case IT_VOJA:
        acc_ptr(a, it.Wm, true); acc_sig(a, it.src - sig, it.rows, false); acc_sig(a, it.aux0 - sig, it.cols, false); acc_sig(a, it.aux1 - sig, 1, false);
        acc_ptr(a, it.aux2, false);
        break;
      default: break;
    }
  }
  static bool hazard(const std::vector<Rng>& x, const std::vector<Rng>& y) {
    for (const Rng& p : x)
      for (const Rng& q : y)
        if (p.space == q.space && p.lo < q.hi && q.lo < p.hi && (p.w || q.w)) return true;
    return false;
  }

  // item_deps of the item plan (RAW, WAR and WAW hazards on earlier items).
  void analyse_dependencies(const std::vector<std::vector<MOp>>& programs, const std::vector<int>& item_prog) {
    std::vector<std::vector<Rng>> acc(items.size());
    int prog_i = 0;
    for (size_t i = 0; i < items.size(); ++i) {
      const Item& it = items[i];
      if (it.type == IT_PROGRAM) {
        for (const MOp& op : programs[(size_t)item_prog[(size_t)prog_i]]) micro_access(acc[i], op);
        ++prog_i;
      } else if (it.type == IT_VECOPS) {
        for (const MOp& op : vecops_host) micro_access(acc[i], op);
      } else {
        item_access(acc[i], it);
      }
    }
    item_deps.assign(items.size(), {});
    for (size_t i = 0; i < items.size(); ++i)
      for (size_t j = 0; j < i; ++j)
        if (hazard(acc[i], acc[j])) item_deps[i].push_back((int)j);
  }

  int init_bsig() {
    if (!bsig) return SSN_OK;
    CHK(upload(sig_init.data(), bsig, 1, n_sig, n_sig));
    for (int r = 1; r <= block; ++r)
      HIPCHK(hipMemcpyAsync(bsig + (size_t)r * n_sig, bsig, (size_t)n_sig * sizeof(T), hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return SSN_OK;
  }

  // Element-wise operators of a time-batched stage: what they write / read (column ranges; rows are handled alike).
  static bool batch_elementwise(const ssn::BatchOp<T>& o) {
    return o.kind == ssn::M_FILL || o.kind == ssn::M_AXPY_INC || o.kind == ssn::M_AXPY_SET || o.kind == ssn::M_TABLE || o.kind == ssn::M_PROBE;
  }
  static bool batch_writes(const ssn::BatchOp<T>& o) { return o.kind != ssn::M_PROBE; }
  static bool batch_reads(const ssn::BatchOp<T>& o) {
    return o.kind == ssn::M_AXPY_INC || o.kind == ssn::M_AXPY_SET || o.kind == ssn::M_PROBE || o.kind == ssn::M_LOWPASS ||
           o.kind == ssn::M_MATVEC_SET || o.kind == ssn::M_MATVEC_INC;
  }
  static long long batch_read_len(const ssn::BatchOp<T>& o) { return (o.kind == ssn::M_MATVEC_SET || o.kind == ssn::M_MATVEC_INC) ? o.cols : o.len; }
  // Hazard between two operators of a stage, `a` the earlier one.  0: none; 1: only on elements that both touch at the SAME
  // index of the same row (equal range origins, no previous-row read of what the other writes) - two element-wise operators
  // may then share a launch, the same thread runs them in order (kb_elementwise_multi); 2: anything else.
  static int batch_hazard(const ssn::BatchOp<T>& a, const ssn::BatchOp<T>& b) {
    auto ov = [](long long x, long long xl, long long y, long long yl) { return x < y + yl && y < x + xl; };
    int kind = 0;
    auto pair = [&](long long x, long long xl, long long y, long long yl, bool prev) {
      if (!ov(x, xl, y, yl)) return;
      kind = std::max(kind, (x == y && !prev) ? 1 : 2);
    };
    if (batch_writes(a) && batch_writes(b)) pair(a.dst, a.len, b.dst, b.len, false);
    if (batch_writes(a) && batch_reads(b)) pair(a.dst, a.len, b.src, batch_read_len(b), b.src_prev != 0);
    if (batch_reads(a) && batch_writes(b)) pair(a.src, batch_read_len(a), b.dst, b.len, a.src_prev != 0);
    if (kind == 1 && !(batch_elementwise(a) && batch_elementwise(b))) kind = 2;
    return kind;
  }

  // Order of a time-batched stage: the builder hands its operators over in one valid order (stages.py `border`), in which
  // independent operators of different kinds alternate (a scan, the sum that reads it, the next scan ...) and every one is
  // a launch.  Here: resets and sums are cut at the range endpoints of the other operators (the reset of all accumulators
  // of a network then lines up with each consumer), every operator takes the earliest dependency level its hazards allow,
  // operators are sorted by level and kind, and neighbouring scans with equal coefficients over adjacent ranges are joined
  // again.  run_batch then packs consecutive element-wise operators - also dependent ones, see batch_hazard - into launches.
  void optimise_batch(std::vector<ssn::BatchOp<T>>& ops) {
    if (ops.size() < 2 || getenv("SSN_NO_BATCH_REORDER")) return;
    typedef ssn::BatchOp<T> Op;
    std::vector<long long> cuts;
    for (const Op& o : ops) {
      if (batch_writes(o)) { cuts.push_back(o.dst); cuts.push_back(o.dst + o.len); }
      if (batch_reads(o)) { cuts.push_back(o.src); cuts.push_back(o.src + batch_read_len(o)); }
    }
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    std::vector<Op> cut;
    for (const Op& o : ops) {
      const bool axpy = o.kind == ssn::M_AXPY_INC || o.kind == ssn::M_AXPY_SET;
      if (!(axpy || o.kind == ssn::M_FILL) || o.len <= 1) { cut.push_back(o); continue; }
      std::vector<long long> offs{0, o.len};
      for (long long base : {o.dst, axpy ? o.src : o.dst}) {
        auto lo = std::upper_bound(cuts.begin(), cuts.end(), base), hi = std::lower_bound(cuts.begin(), cuts.end(), base + o.len);
        for (auto it = lo; it < hi; ++it) offs.push_back(*it - base);
      }
      std::sort(offs.begin(), offs.end());
      offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
      if (offs.size() > 32) { cut.push_back(o); continue; }
      for (size_t q = 0; q + 1 < offs.size(); ++q) {
        Op pc = o;
        pc.dst = o.dst + offs[q]; pc.len = offs[q + 1] - offs[q];
        if (axpy) pc.src = o.src + offs[q];
        cut.push_back(pc);
      }
    }
    const size_t n = cut.size();
    std::vector<int> level(n, 0);
    for (size_t j = 0; j < n; ++j)
      for (size_t i = 0; i < j; ++i)
        if (batch_hazard(cut[i], cut[j])) level[j] = std::max(level[j], level[i] + 1);
    auto cls = [](const Op& o) { return batch_elementwise(o) ? 0 : o.kind == ssn::M_LOWPASS ? 1 : 2; };
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
      if (level[x] != level[y]) return level[x] < level[y];
      if (cls(cut[x]) != cls(cut[y])) return cls(cut[x]) < cls(cut[y]);
      return cut[x].kind == ssn::M_LOWPASS && cut[y].kind == ssn::M_LOWPASS && cut[x].dst < cut[y].dst;
    });
    std::vector<Op> out;
    int last_level = -1;
    for (size_t k : order) {
      const Op& o = cut[k];
      if (!out.empty() && level[k] == last_level && o.kind == ssn::M_LOWPASS && out.back().kind == ssn::M_LOWPASS && out.back().a == o.a && out.back().b == o.b &&
          out.back().src_prev == o.src_prev && out.back().dst + out.back().len == o.dst && out.back().src + out.back().len == o.src) {
        out.back().len += o.len;
        continue;
      }
      out.push_back(o);
      last_level = level[k];
    }
    if (getenv("SSN_DEBUG_PLAN")) {
      fprintf(stderr, "[ssn] time-batched stage: %zu operators -> %zu\n", ops.size(), out.size());
      for (const Op& o : out) fprintf(stderr, "[ssn]   batch op kind %d dst %lld src %lld len %lld cols %d prev %d\n", o.kind, (long long)o.dst, (long long)o.src, (long long)o.len, o.cols, o.src_prev);
    }
    ops.swap(out);
  }

  hipError_t run_batch(std::vector<ssn::BatchOp<T>>& ops, int B, int64_t step0) {
    auto elementwise = [](const ssn::BatchOp<T>& o) { return batch_elementwise(o); };
    // Signal elements known to be ZERO in every row of this block: a table without an entry for any timestep of the block
    // (the init-SSP input of the path integrator after its first 50 ms, reference run_pathint.py:136), and what fills /
    // copies / products make of zeros.  A product whose whole input is zero is not multiplied out (PathIntegration config 2:
    // the 1000 x 1015 x 1524 to_Fourier GEMM of every block but the first - 47 of a block's 3480 us).
    std::vector<char> zero((size_t)n_sig, 0);
    const bool track = !getenv("SSN_NO_ZERO_SKIP");
    auto all_zero = [&](long long lo, long long n) { for (long long q = lo; q < lo + n; ++q) if (!zero[(size_t)q]) return false; return n > 0; };
    auto set_zero = [&](long long lo, long long n, bool v) { for (long long q = lo; q < lo + n; ++q) zero[(size_t)q] = v ? 1 : 0; };
    auto note = [&](const ssn::BatchOp<T>& o) {              // effect of an operator that IS executed on the zero map
      if (!track) return;
      switch (o.kind) {
        case ssn::M_TABLE: {
          bool z = false;
          for (size_t id = 0; id < table_dst.size() && id < table_idx_host.size(); ++id)
            if (table_dst[id].second > 0 && table_dst[id].first == o.dst && table_dst[id].second == o.len) z = table_is_zero((int)id, B, step0);
          set_zero(o.dst, o.len, z); break;
        }
        case ssn::M_FILL: set_zero(o.dst, o.len, o.a == T(0)); break;
        case ssn::M_AXPY_SET: set_zero(o.dst, o.len, !o.src_prev && all_zero(o.src, o.len)); break;
        case ssn::M_AXPY_INC: if (o.src_prev || !all_zero(o.src, o.len)) set_zero(o.dst, o.len, false); break;
        case ssn::M_PROBE: break;
        default: set_zero(o.dst, o.len, false); break;          // products, filters: not known
      }
    };
    for (size_t i = 0; i < ops.size();) {
      ops[i].B = B; ops[i].step0 = step0;
      if (elementwise(ops[i]) && !(flags & 262144)) {
        // consecutive element-wise operators share one launch unless a hazard between them crosses threads (batch_hazard)
        ssn::BatchOpList<T> l{};
        l.op[0] = ops[i];
        l.count = 1;
        size_t j = i + 1;
        for (; j < ops.size() && l.count < ssn::MAX_BATCH_OPS && elementwise(ops[j]); ++j) {
          ops[j].B = B; ops[j].step0 = step0;
          bool clash = false;
          for (int q = 0; q < l.count; ++q) clash = clash || batch_hazard(l.op[q], ops[j]) == 2;
          if (clash) break;
          l.op[l.count++] = ops[j];
        }
        if (l.count > 1) {
          hipError_t e = ssn::launch_batch_elementwise<T>(stream, l);
          if (e != hipSuccess) return e;
          for (int q = 0; q < l.count; ++q) note(l.op[q]);      // (list order = program order of every thread)
          i = j;
          continue;
        }
      }
      if (track && (ops[i].kind == ssn::M_MATVEC_SET || ops[i].kind == ssn::M_MATVEC_INC) && !ops[i].src_prev && all_zero(ops[i].src, ops[i].cols)) {
        ++batch_skipped;
        if (ops[i].kind == ssn::M_MATVEC_SET) {        // W @ 0: the result rows of the block are zero (an increment adds nothing)
          hipError_t e = hipMemset2DAsync(bsig + n_sig + ops[i].dst, (size_t)n_sig * sizeof(T), 0, (size_t)ops[i].len * sizeof(T), (size_t)B, stream);
          if (e != hipSuccess) return e;
          set_zero(ops[i].dst, ops[i].len, true);
        }
        ++i;
        continue;
      }
      hipError_t e = ssn::launch_batch_op<T>(stream, ops[i]);
      if (e != hipSuccess) return e;
      note(ops[i]);
      ++i;
    }
    return hipSuccess;
  }

  // Has the table no entry (index -1, or outside its range) for every timestep of this block?  Its rows are zeros then.
  bool table_is_zero(int id, int B, int64_t step0) const {
    const ssn::TableSlot& t = tables[(size_t)id];
    const std::vector<int32_t>& ix = table_idx_host[(size_t)id];
    for (int64_t st = step0; st < step0 + B; ++st) {
      const int64_t rel = st - t.first_step;
      if (rel >= 0 && rel < (int64_t)ix.size() && ix[(size_t)rel] >= 0) return false;
    }
    return true;
  }

  // ---- launching --------------------------------------------------------------------------
  hipError_t launch_item(const Item& it, hipEvent_t e0, hipEvent_t e1) {
    if (it.merged) return hipSuccess;               // its batch leader launched it
    const Item* g = &it;                            // (batch members are adjacent in `items`)
    switch (it.type) {
      case IT_PROGRAM: return ssn::launch_program<T>(stream, d_mops, d_progs + it.op_begin, 1, sig, d_ctx);
      case IT_VECOPS: return ssn::launch_vecops<T>(stream, d_mops + it.op_begin, it.op_count, it.n, sig, d_ctx);
      case IT_ENS: {
        if (e0) { hipError_t e = hipEventRecord(e0, stream); if (e != hipSuccess) return e; }
        hipError_t e;
        if (it.batch > 1) {
          ssn::EnsBatch<T> b{};
          for (int q = 0; q < it.batch; ++q) b.a[q] = g[q].ens;
          e = ssn::launch_ensarray_batch<T>(stream, b, it.batch);
        } else {
          e = ssn::launch_ensarray<T>(stream, it.ens);
        }
        if (e != hipSuccess) return e;
        if (e1) return hipEventRecord(e1, stream);
        return hipSuccess;
      }
      case IT_MATVEC: {
        ssn::MatvecBatch<T> b{};
        for (int q = 0; q < it.batch; ++q) b.a[q] = ssn::MatvecArgs<T>{g[q].Wm, g[q].src, g[q].dst, g[q].rows, g[q].cols, g[q].ld, g[q].set};
        return ssn::launch_matvec<T>(stream, b, it.batch);
      }
      case IT_MATVEC_ORDERED: return ssn::launch_matvec_ordered<T>(stream, it.Wm, it.src, it.dst, it.rows, it.cols, it.ld);
      case IT_FINISH: return ssn::launch_ens_finish<T>(stream, it.fin);
      case IT_DFT: {
        ssn::DftBatch b{};
        for (int q = 0; q < it.batch; ++q) b.a[q] = g[q].dft;
        return ssn::launch_dft<T>(stream, b, it.batch);
      }
      case IT_ARGMAX_PART: return ssn::launch_argmax_partial<T>(stream, it.src, (long long)it.rows, it.dst, it.n, std::max(1, it.seg));
      case IT_GRID_LHS: return ssn::launch_grid_lhs<T>(stream, it.src, it.aux0, it.ld, it.dst, it.cols, it.rows, it.cols / 2);
      case IT_GRID_DOT: return hipErrorInvalidValue;      // (planned only where the step runs as rounds)
      case IT_GRID_GEMM: return ssn::launch_gemm_nt<T>(stream, it.src, it.cols, it.Wm, it.ld, it.dst, it.n, it.rows, it.n, it.cols, std::max(1, it.seg));
      case IT_SPMV: {
        ssn::SpmvBatch<T> b{};
        for (int q = 0; q < it.batch; ++q)
          b.a[q] = ssn::SpmvArgs<T>{g[q].Wm, g[q].ld, g[q].src, g[q].cols, g[q].rows, g[q].dst, g[q].ld, g[q].n, g[q].list, g[q].count, g[q].seg, g[q].seg_len};
        return ssn::launch_spmv_partial<T>(stream, b, it.batch);
      }
      case IT_NEURONS: {
        ssn::NeuronsBatch<T> b{};
        for (int q = 0; q < it.batch; ++q)
          b.a[q] = ssn::NeuronsArgs<T>{g[q].np, g[q].src, g[q].dst, g[q].V, g[q].R, g[q].n, g[q].scalar, g[q].list, g[q].count};
        return ssn::launch_neurons<T>(stream, b, it.batch);
      }
      case IT_PES: return ssn::launch_pes<T>(stream, it.Wm, it.aux0, it.aux1, it.rows, it.cols, it.ld, it.scalar);
      case IT_VOJA: return ssn::launch_voja<T>(stream, it.Wm, it.src, it.aux0, it.aux1, it.aux2, it.rows, it.cols, it.ld, it.scalar);
    }
    return hipErrorInvalidValue;
  }

  // `count` consecutive steps; fused = tail(s)+head(s+1) in one launch
  hipError_t launch_steps(int count, bool fused) {
    if (fused_defer) {                          // one k_ensarray per timestep; the clock advances once for the group
      for (int s = 0; s < count; ++s) {
        items[0].ens.sub = s;
        hipError_t e = launch_item(items[0], nullptr, nullptr);
        if (e != hipSuccess) return e;
      }
      items[0].ens.sub = 0;
      hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, stream, d_ctx, (long long)count);
      return hipGetLastError();
    }
    if (round_mode) {
      if (count == steps_per_graph && !graph_list.empty()) {
        for (const Launch& l : graph_list) { hipError_t e = launch_one(l); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, stream, d_ctx, (long long)count);
        return hipGetLastError();
      }
      for (int s = 0; s < count; ++s)
        for (const Launch& l : launch_list) { hipError_t e = launch_one(l); if (e != hipSuccess) return e; }
      return hipSuccess;
    }
    const int n_items = (int)items.size();
    for (int s = 0; s < count; ++s) {
      for (int i = 0; i < n_items; ++i) {
        const Item& it = items[i];
        hipError_t e = hipSuccess;
        if (fused && can_fuse && i == 0 && s > 0) continue;               // head already ran with the previous tail
        if (fused && can_fuse && i == n_items - 1 && s + 1 < count)
          e = ssn::launch_program<T>(stream, d_mops, d_progs + tail_begin, 2, sig, d_ctx);
        else
          e = launch_item(it, nullptr, nullptr);
        if (e != hipSuccess) return e;
      }
    }
    return hipSuccess;
  }

  void drop_cycle_graphs() {
    for (auto e : cycle_exec) if (e) hipGraphExecDestroy(e);
    for (auto g : cycle_graph) if (g) hipGraphDestroy(g);
    cycle_exec.clear(); cycle_graph.clear();
  }
  // segment k of the pipelined sharded cycle (k = 0 .. cycle_steps); the clock advances behind the last one
  hipError_t launch_segment(int k) {
    for (const Launch& l : cycle_segs[(size_t)k]) { hipError_t e = launch_one(l); if (e != hipSuccess) return e; }
    if (k == cycle_steps) {
      hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, stream, d_ctx, (long long)cycle_steps);
      return hipGetLastError();
    }
    return hipSuccess;
  }

  hipError_t launch_phase(int phase) {
    if (phase == 2) {          // updates of timestep s + timestep s + 1 up to its exchange
      if (round_mode && !phase2_list.empty()) {
        for (const Launch& l : phase2_list) { hipError_t e = launch_one(l); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, stream, d_ctx, 1LL);
        return hipGetLastError();
      }
      hipError_t e = launch_phase(1);
      return e == hipSuccess ? launch_phase(0) : e;
    }
    if (round_mode) {
      for (const Launch& l : launch_list) {
        if (l.phase != phase) continue;
        hipError_t e = launch_one(l);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    }
    for (const Item& it : items) {
      if (it.phase != phase) continue;
      hipError_t e = launch_item(it, nullptr, nullptr);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }

  int capture() {
    if (phased) {             // one graph per half of the timestep
      if (fused_core || core_empty) return fail(SSN_EUNSUPPORTED, "neuron-sharded models run on the generic per-timestep plan");
      for (int h = 0; h < 3; ++h) {        // [2]: phase 1 of a timestep + phase 0 of the next one in one launch
        HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        hipError_t e = launch_phase(h);
        hipError_t e2 = hipStreamEndCapture(stream, &phase_graph[h]);
        HIPCHK(e);
        HIPCHK(e2);
        HIPCHK(hipGraphInstantiate(&phase_exec[h], phase_graph[h], nullptr, nullptr, 0));
      }
      return SSN_OK;
    }
    if (steps_per_graph <= 1 || fused_block) return SSN_OK;
    HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    hipError_t e = launch_steps(steps_per_graph, true);
    hipError_t e2 = hipStreamEndCapture(stream, &graph);
    HIPCHK(e);
    HIPCHK(e2);
    HIPCHK(hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0));
    return SSN_OK;
  }

  int run_phase(int phase) override {
    HIPCHK(hipSetDevice(device));
    if (!phased) return fail(SSN_EINVAL, "ssn_run_phase: the model has no exchange ranges (use ssn_run_steps)");
    if (phase == 3) {            // the next segment of a pipelined cycle (eager launches; the caller exchanges between two segments)
      if (!cycle_steps || !round_mode) return fail(SSN_EINVAL, "ssn_run_phase(3): this model has no pipelined cycle plan (ssn_cycle_steps() == 0)");
      if (next_seg == 0 && next_phase != 0) return fail(SSN_EINVAL, "ssn_run_phase(3): a cycle starts at a timestep boundary (phase %d is due)", next_phase);
      HIPCHK(launch_segment(next_seg));
      HIPCHK(hipStreamSynchronize(stream));
      if (++next_seg > cycle_steps) {
        next_seg = 0;
        steps_done += cycle_steps;
        ssn::StepCtx ctx;
        HIPCHK(hipMemcpy(&ctx, d_ctx, sizeof ctx, hipMemcpyDeviceToHost));
        if (ctx.step != steps_done) return fail(SSN_EHIP, "device step counter %lld != host %lld", (long long)ctx.step, (long long)steps_done);
        if (ctx.probe_overflow) return fail(SSN_EINVAL, "probe storage overflow: call ssn_reserve_probes before stepping");
      }
      return SSN_OK;
    }
    if (next_seg != 0) return fail(SSN_EINVAL, "ssn_run_phase(%d): a pipelined cycle is under way (segment %d of %d is due)", phase, next_seg, cycle_steps + 1);
    if (phase < 0 || phase > 2) return fail(SSN_EINVAL, "ssn_run_phase(%d): 0, 1 or 2 (= 1 followed by the next timestep's 0)", phase);
    if ((phase == 2 ? 1 : phase) != next_phase) return fail(SSN_EINVAL, "ssn_run_phase(%d): phase %d is due", phase, next_phase);
    HIPCHK(hipGraphLaunch(phase_exec[phase], stream));
    HIPCHK(hipStreamSynchronize(stream));
    next_phase = phase == 0 ? 1 : (phase == 1 ? 0 : 1);
    if (phase >= 1) {
      steps_done += 1;
      if (steps_done % 64 == 0 || steps_done == reserve_first + reserve_n) {      // (a device -> host round trip: not on every timestep)
        ssn::StepCtx ctx;
        HIPCHK(hipMemcpy(&ctx, d_ctx, sizeof ctx, hipMemcpyDeviceToHost));
        if (ctx.step != steps_done) return fail(SSN_EHIP, "device step counter %lld != host %lld", (long long)ctx.step, (long long)steps_done);
        if (ctx.probe_overflow) return fail(SSN_EINVAL, "probe storage overflow: call ssn_reserve_probes before stepping");
      }
    }
    return SSN_OK;
  }

  int64_t cycle_len() override { return round_mode ? cycle_steps : 0; }

  int64_t exchange_size() override {
    int64_t n = 0;
    for (auto& r : exchange) n += r.hi - r.lo;
    return n;
  }

  // signal ranges of the exchange <-> one contiguous device buffer (a handful of ranges: one device copy each)
  int exchange_copy(void* buf, bool pack) override {
    HIPCHK(hipSetDevice(device));
    T* b = (T*)buf;
    for (auto& r : exchange) {
      const size_t bytes = (size_t)(r.hi - r.lo) * sizeof(T);
      if (pack) HIPCHK(hipMemcpyAsync(b, sig + r.lo, bytes, hipMemcpyDeviceToDevice, stream));
      else HIPCHK(hipMemcpyAsync(sig + r.lo, b, bytes, hipMemcpyDeviceToDevice, stream));
      b += r.hi - r.lo;
    }
    HIPCHK(hipStreamSynchronize(stream));
    return SSN_OK;
  }

  hipError_t exchange_copy_async(void* buf, bool pack, hipStream_t st) {
    T* b = (T*)buf;
    for (auto& r : exchange) {
      const size_t bytes = (size_t)(r.hi - r.lo) * sizeof(T);
      hipError_t e = pack ? hipMemcpyAsync(b, sig + r.lo, bytes, hipMemcpyDeviceToDevice, st)
                          : hipMemcpyAsync(sig + r.lo, b, bytes, hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return e;
      b += r.hi - r.lo;
    }
    return hipSuccess;
  }

  // (Re)capture the three phase graphs with the exchange copies of `buf` inside: [unpack] -> kernels -> [pack].
  int capture_async(void* buf) {
    for (int h = 0; h < 3; ++h) {
      if (async_exec[h]) { hipGraphExecDestroy(async_exec[h]); async_exec[h] = nullptr; }
      if (async_graph[h]) { hipGraphDestroy(async_graph[h]); async_graph[h] = nullptr; }
    }
    for (int h = 0; h < 3; ++h) {
      HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      hipError_t e = hipSuccess;
      if (h >= 1 && buf) e = exchange_copy_async(buf, false, stream);
      if (e == hipSuccess) e = launch_phase(h);
      if (h != 1 && buf && e == hipSuccess) e = exchange_copy_async(buf, true, stream);
      hipError_t e2 = hipStreamEndCapture(stream, &async_graph[h]);
      HIPCHK(e);
      HIPCHK(e2);
      HIPCHK(hipGraphInstantiate(&async_exec[h], async_graph[h], nullptr, nullptr, 0));
    }
    drop_cycle_graphs();
    if (cycle_steps && round_mode) {          // segment k: [unpack the sums of timestep k - 1] -> rounds -> [pack the partial sums of timestep k]
      cycle_graph.assign((size_t)cycle_steps + 1, nullptr);
      cycle_exec.assign((size_t)cycle_steps + 1, nullptr);
      for (int k = 0; k <= cycle_steps; ++k) {
        HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipSuccess;
        if (k >= 1 && buf) e = exchange_copy_async(buf, false, stream);
        if (e == hipSuccess) e = launch_segment(k);
        if (k < cycle_steps && buf && e == hipSuccess) e = exchange_copy_async(buf, true, stream);
        hipError_t e2 = hipStreamEndCapture(stream, &cycle_graph[(size_t)k]);
        HIPCHK(e);
        HIPCHK(e2);
        HIPCHK(hipGraphInstantiate(&cycle_exec[(size_t)k], cycle_graph[(size_t)k], nullptr, nullptr, 0));
      }
    }
    async_buf = buf;
    async_captured = true;
    return SSN_OK;
  }

  int phase_async(int phase, void* buf, hipStream_t ext) override {
    HIPCHK(hipSetDevice(device));
    if (!phased) return fail(SSN_EINVAL, "ssn_phase_async: the model has no exchange ranges (use ssn_run_steps)");
    if (phase == -1) {              // capture only: the graphs for this exchange buffer are built now, nothing is launched
      if (async_active) return fail(SSN_EINVAL, "ssn_phase_async(-1): a run is in flight (call ssn_phase_sync first)");
      HIPCHK(hipStreamSynchronize(stream));
      if (!async_captured || buf != async_buf) CHK(capture_async(buf));
      return SSN_OK;
    }
    if (phase == 3) {
      if (!cycle_steps || !round_mode) return fail(SSN_EINVAL, "ssn_phase_async(3): this model has no pipelined cycle plan (ssn_cycle_steps() == 0)");
      if (next_seg == 0 && next_phase != 0) return fail(SSN_EINVAL, "ssn_phase_async(3): a cycle starts at a timestep boundary (phase %d is due)", next_phase);
    } else {
      if (next_seg != 0) return fail(SSN_EINVAL, "ssn_phase_async(%d): a pipelined cycle is under way (segment %d of %d is due)", phase, next_seg, cycle_steps + 1);
      if (phase < 0 || phase > 2) return fail(SSN_EINVAL, "ssn_phase_async(%d): 0, 1, 2 (= 1 followed by the next timestep's 0) or 3 (next segment of a pipelined cycle)", phase);
      if ((phase == 2 ? 1 : phase) != next_phase) return fail(SSN_EINVAL, "ssn_phase_async(%d): phase %d is due", phase, next_phase);
    }
    if (!async_active) {
      HIPCHK(hipStreamSynchronize(stream));          // uploads / table updates issued on the simulator's own stream come first
      if (!async_captured || buf != async_buf) CHK(capture_async(buf));
      async_active = true;
    } else if (buf != async_buf) {
      return fail(SSN_EINVAL, "ssn_phase_async: the exchange buffer changed inside a run (call ssn_phase_sync first)");
    }
    async_stream = ext ? ext : stream;
    if (phase == 3) {
      HIPCHK(hipGraphLaunch(cycle_exec[(size_t)next_seg], ext ? ext : stream));
      if (++next_seg > cycle_steps) { next_seg = 0; steps_done += cycle_steps; }
      return SSN_OK;
    }
    HIPCHK(hipGraphLaunch(async_exec[phase], ext ? ext : stream));
    next_phase = phase == 0 ? 1 : (phase == 1 ? 0 : 1);
    if (phase >= 1) steps_done += 1;
    return SSN_OK;
  }

  int phase_sync(hipStream_t ext) override {
    HIPCHK(hipSetDevice(device));
    if (!phased) return fail(SSN_EINVAL, "ssn_phase_sync: the model has no exchange ranges");
    HIPCHK(hipStreamSynchronize(ext ? ext : stream));
    async_active = false;
    if (next_seg != 0) { const int due = next_seg; next_seg = 0; return fail(SSN_EINVAL, "ssn_phase_sync inside a pipelined cycle (segment %d of %d was due): the simulator's state is part-way through %d timesteps - reset it", due, cycle_steps + 1, cycle_steps); }
    ssn::StepCtx ctx;
    HIPCHK(hipMemcpy(&ctx, d_ctx, sizeof ctx, hipMemcpyDeviceToHost));
    if (ctx.step != steps_done) return fail(SSN_EHIP, "device step counter %lld != host %lld", (long long)ctx.step, (long long)steps_done);
    if (ctx.probe_overflow) return fail(SSN_EINVAL, "probe storage overflow: call ssn_reserve_probes before stepping");
    return SSN_OK;
  }

  int run_steps(int64_t n, int profile) override {
    HIPCHK(hipSetDevice(device));
    if (phased) return fail(SSN_EINVAL, "a neuron-sharded model is stepped with ssn_run_phase (the caller exchanges between the phases)");
    if (n < 0) return fail(SSN_EINVAL, "negative step count");
    if (n == 0) return SSN_OK;
    int n_dom = 0;
    for (auto& it : items) n_dom += it.dominant ? 1 : 0;
    size_t ev_used = 0;
    std::vector<int> ev_types, ev_items;        // profile = 2: plan-item type / index of each event pair
    if (profile == 2 && (fused_block || core_empty)) profile = 1;
    if (profile) {
      const size_t need = fused_block ? (size_t)(2 * (n / std::max(1, block) + 2))
                                      : (size_t)(2 * n * (profile == 2 ? (int)(items.size() + launch_list.size()) : std::max(1, n_dom)));
      if (need > 400000) return fail(SSN_EINVAL, "profile run too long (%lld steps): at most 200000 timed launches", (long long)n);
      while (ev_pool.size() < need) {
        hipEvent_t ev;
        HIPCHK(hipEventCreate(&ev));
        ev_pool.push_back(ev);
      }
    }
    HIPCHK(hipEventRecord(ev_run0, stream));
    int64_t done = 0;
    while (done < n) {
      // one block: [pre stage, batched] -> core, one timestep at a time -> [post stage, batched]
      const int64_t B = bsig ? std::min<int64_t>(block, n - done) : (n - done);
      const int64_t step0 = steps_done + done;
      if (bsig) {
        hipLaunchKernelGGL(k_set_block, dim3(1), dim3(1), 0, stream, d_ctx, (long long)step0);
        HIPCHK(hipGetLastError());
        HIPCHK(run_batch(pre_ops, (int)B, step0));
      }
      if (fused_defer) HIPCHK(ssn::launch_ens_finish<T>(stream, fin_begin));
      if (fused_block) {
        const bool timed = profile && B == block && ev_used + 2 <= ev_pool.size();
        blk.B = (int)B;
        if (blk.P > 1)      // every exchange word back to the sentinel: the kernel boundary is the members' only common barrier
          HIPCHK(hipMemsetD32Async((hipDeviceptr_t)blk.xslots, (int)ssn::BLOCK_XCHG_SENTINEL, (size_t)3 * blk.K * 64, stream));
        if (timed) HIPCHK(hipEventRecord(ev_pool[ev_used], stream));
        HIPCHK(ssn::launch_ens_block<T>(stream, blk));
        if (timed) { HIPCHK(hipEventRecord(ev_pool[ev_used + 1], stream)); ev_used += 2; }
        hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, stream, d_ctx, (long long)B);
        HIPCHK(hipGetLastError());
      } else if (core_empty) {
        // nothing is stepped one timestep at a time (purely feed-forward model): just advance the clock
        hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, stream, d_ctx, (long long)B);
        HIPCHK(hipGetLastError());
      } else if (profile == 2) {
        // every launch of every timestep between its own event pair (plain eager launches, no graph)
        int64_t s_piped = 0;
        for (; round_mode && !graph_list.empty() && s_piped + steps_per_graph <= B; s_piped += steps_per_graph) {
          // the pipelined sequence a step graph replays, launch by launch
          for (const Launch& l : graph_list) {
            HIPCHK(hipEventRecord(ev_pool[ev_used], stream));
            HIPCHK(launch_one(l));
            HIPCHK(hipEventRecord(ev_pool[ev_used + 1], stream));
            ev_types.push_back(l.rl >= 0 ? (int)IT_ROUND : items[(size_t)l.item].type);
            ev_items.push_back(l.rl >= 0 ? (int)items.size() + l.rl : l.item);
            ev_used += 2;
          }
          hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, stream, d_ctx, (long long)steps_per_graph);
          HIPCHK(hipGetLastError());
        }
        for (int64_t s = s_piped; s < B && round_mode; ++s)
          for (const Launch& l : launch_list) {
            if (l.rl < 0 && items[(size_t)l.item].merged) continue;
            HIPCHK(hipEventRecord(ev_pool[ev_used], stream));
            HIPCHK(launch_one(l));
            HIPCHK(hipEventRecord(ev_pool[ev_used + 1], stream));
            ev_types.push_back(l.rl >= 0 ? (int)IT_ROUND : items[(size_t)l.item].type);
            ev_items.push_back(l.rl >= 0 ? (int)items.size() + l.rl : l.item);
            ev_used += 2;
          }
        for (int64_t s = 0; s < B && !round_mode; ++s) {
          for (auto& it : items) {
            if (it.merged) continue;
            HIPCHK(hipEventRecord(ev_pool[ev_used], stream));
            HIPCHK(launch_item(it, nullptr, nullptr));
            HIPCHK(hipEventRecord(ev_pool[ev_used + 1], stream));
            ev_types.push_back(it.type);
            ev_items.push_back((int)(&it - items.data()));
            ev_used += 2;
          }
          if (fused_defer) { hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, stream, d_ctx, 1LL); HIPCHK(hipGetLastError()); }
        }
      } else if (profile && round_mode) {
        for (int64_t s = 0; s < B; ++s)
          for (const Launch& l : launch_list) {
            if (l.rl < 0 && items[(size_t)l.item].dominant) { HIPCHK(launch_item(items[(size_t)l.item], ev_pool[ev_used], ev_pool[ev_used + 1])); ev_used += 2; }
            else HIPCHK(launch_one(l));
          }
      } else if (profile) {
        for (int64_t s = 0; s < B; ++s) {
          for (auto& it : items) {
            if (it.dominant) { HIPCHK(launch_item(it, ev_pool[ev_used], ev_pool[ev_used + 1])); ev_used += 2; }
            else HIPCHK(launch_item(it, nullptr, nullptr));
          }
          if (fused_defer) { hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, stream, d_ctx, 1LL); HIPCHK(hipGetLastError()); }
        }
      } else {
        int64_t left = B;
        if (graph_exec)
          for (; left >= steps_per_graph; left -= steps_per_graph) HIPCHK(hipGraphLaunch(graph_exec, stream));
        if (left > 0) HIPCHK(launch_steps((int)left, true));
      }
      if (fused_defer) HIPCHK(ssn::launch_ens_finish<T>(stream, fin_flush));
      if (bsig) {
        HIPCHK(run_batch(post_ops, (int)B, step0));
        HIPCHK(hipMemcpyAsync(bsig, bsig + (size_t)B * n_sig, (size_t)n_sig * sizeof(T), hipMemcpyDeviceToDevice, stream));
      }
      done += B;
    }
    HIPCHK(hipEventRecord(ev_run1, stream));
    HIPCHK(hipStreamSynchronize(stream));
    for (size_t j = 0; j < ev_used; j += 2) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, ev_pool[j], ev_pool[j + 1]));
      if (profile == 2) {
        const int ty = ev_types[j / 2];
        type_ms[ty] += ms;
        type_launches[ty] += 1;
        if (item_ms.size() != items.size() + round_launches.size()) { item_ms.assign(items.size() + round_launches.size(), 0.0); item_n.assign(items.size() + round_launches.size(), 0); }
        item_ms[(size_t)ev_items[j / 2]] += ms; item_n[(size_t)ev_items[j / 2]] += 1;
        continue;
      }
      dom_ms += ms;
      dom_launches += 1;
    }
    if (profile == 2 && getenv("SSN_DEBUG_PLAN")) {
      // diagnostic build (F32_EXTRA=-DSSN_PROGRAM_STAMPS): shader cycles each operator of each program took in the last timestep
      typedef int (*stamp_fn)(unsigned long long*, int);
      if (stamp_fn fn = (stamp_fn)dlsym(RTLD_DEFAULT, "ssn_debug_program_stamps")) {
        std::vector<unsigned long long> st(2048, 0);
        if (fn(st.data(), 2048) == 0)
          for (size_t p = 0; p < prog_descs.size(); ++p) {
            const ssn::ProgDesc& pd = prog_descs[p];
            if (pd.op_begin + pd.op_count > 1024) continue;
            unsigned long long prev = st[(size_t)(1024 + pd.op_begin)];
            fprintf(stderr, "[ssn] program %zu (%d ops) cycles per operator [kind/len:cycles]:", p, pd.op_count);
            for (int o = 0; o < pd.op_count; ++o) {
              const unsigned long long t = st[(size_t)(pd.op_begin + o)];
              if (!t) { fprintf(stderr, " %d/%lld:-", mops[(size_t)(pd.op_begin + o)].kind, (long long)mops[(size_t)(pd.op_begin + o)].len); continue; }
              fprintf(stderr, " %d/%lld:%lld", mops[(size_t)(pd.op_begin + o)].kind, (long long)mops[(size_t)(pd.op_begin + o)].len, (long long)(t - prev));
              prev = t;
            }
            fprintf(stderr, "\n");
          }
      }
    }
    if (profile == 2 && getenv("SSN_DEBUG_PLAN"))
      for (size_t i = 0; i < item_ms.size(); ++i)
        if (item_n[i]) {
          if (i < items.size()) fprintf(stderr, "[ssn] item %2zu type %2d batch %d: %8.2f us avg over %lld launches\n", i, items[i].type, items[i].batch,
                                        1e3 * item_ms[i] / item_n[i], (long long)item_n[i]);
          else fprintf(stderr, "[ssn] round launch %2zu (round %d, %d blocks): %8.2f us avg over %lld launches\n", i - items.size(),
                       round_launches[i - items.size()].round, round_launches[i - items.size()].n_blocks, 1e3 * item_ms[i] / item_n[i], (long long)item_n[i]);
        }
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev_run0, ev_run1));
    last_run_ms = ms;
    steps_done += n;
    ssn::StepCtx ctx;
    HIPCHK(hipMemcpy(&ctx, d_ctx, sizeof ctx, hipMemcpyDeviceToHost));
    if (ctx.step != steps_done) return fail(SSN_EHIP, "device step counter %lld != host %lld", (long long)ctx.step, (long long)steps_done);
    if (ctx.probe_overflow) return fail(SSN_EINVAL, "probe storage overflow: call ssn_reserve_probes before ssn_run_steps");
    if (fused_block && blk.P > 1) {
      int xe = 0;
      HIPCHK(hipMemcpy(&xe, blk.xerr, sizeof xe, hipMemcpyDeviceToHost));
      if (xe) return fail(SSN_EHIP, "split ensembles: a member workgroup waited too long for a partner's sums (are all %d workgroups "
                                    "resident together? flag 1073741824 needs the GPU for this process alone)", blk.K * blk.P);
    }
    return SSN_OK;
  }

  int reset() override {
    HIPCHK(hipSetDevice(device));
    if (async_active) {          // phase graphs may still be queued on the caller's stream: they come first
      HIPCHK(hipStreamSynchronize(async_stream ? async_stream : stream));
      async_active = false;
    }
    HIPCHK(hipStreamSynchronize(stream));
    CHK(upload(sig_init.data(), sig, 1, n_sig, n_sig));
    for (auto& b : bufs)
      if (b.keep) CHK(upload_buf(b, b.host.data()));
    HIPCHK(hipMemset(d_ctx, 0, sizeof(ssn::StepCtx)));
    CHK(init_bsig());
    steps_done = 0;
    next_phase = 0;
    next_seg = 0;
    reserve_first = reserve_n = 0;
    for (auto& s : pslots) { s.base_slot = 0; s.capacity = 0; }
    if (!pslots.empty()) HIPCHK(hipMemcpy(d_pslots, pslots.data(), pslots.size() * sizeof(ssn::ProbeSlot), hipMemcpyHostToDevice));
    dom_launches = 0; dom_ms = 0.0;
    for (int t = 0; t < N_ITEM_TYPES; ++t) { type_ms[t] = 0.0; type_launches[t] = 0; }
    if (fused_block && blk.slot_stats) HIPCHK(hipMemset(blk.slot_stats, 0, 16));
    for (auto& z : zero_on_reset) HIPCHK(hipMemset(z.first, 0, z.second));
    return SSN_OK;
  }

  int set_table(int id, const double* rows, const void* rows_dev, int64_t n_rows, int64_t width,
                const int32_t* idx, int64_t n_idx, int64_t first_step) override {
    HIPCHK(hipSetDevice(device));
    if (async_active) return fail(SSN_EINVAL, "a stream-ordered run is in flight on the caller's stream: call ssn_phase_sync first");
    if (id < 0 || id >= (int)tables.size()) return fail(SSN_EINVAL, "table id %d out of range", id);
    if (width != tables[id].width) return fail(SSN_EINVAL, "table %d is %lld wide, got %lld", id, (long long)tables[id].width, (long long)width);
    if (n_rows < 0 || n_idx < 0 || (!idx && n_idx)) return fail(SSN_EINVAL, "bad table arguments");
    for (int64_t i = 0; i < n_idx; ++i)
      if (idx[i] >= n_rows) return fail(SSN_EINVAL, "table row index %d >= n_rows %lld", idx[i], (long long)n_rows);
    HIPCHK(hipStreamSynchronize(stream));
    const int64_t need = std::max<int64_t>(1, n_rows * width);
    if (need > table_rows_cap[id]) {
      if (table_rows[id]) hipFree(table_rows[id]);
      table_rows[id] = nullptr;
      CHK(dmalloc((T**)&table_rows[id], need * (int64_t)sizeof(T)));
      table_rows_cap[id] = need;
    }
    if (n_idx > table_idx_cap[id]) {
      if (table_idx[id]) hipFree(table_idx[id]);
      table_idx[id] = nullptr;
      CHK(dmalloc(&table_idx[id], n_idx * 4));
      table_idx_cap[id] = n_idx;
    }
    if (rows_dev) HIPCHK(hipMemcpy(table_rows[id], rows_dev, (size_t)(n_rows * width) * sizeof(T), hipMemcpyDeviceToDevice));
    else if (n_rows) CHK(upload(rows, (T*)table_rows[id], n_rows, width, width));
    if (n_idx) HIPCHK(hipMemcpy(table_idx[id], idx, (size_t)n_idx * 4, hipMemcpyHostToDevice));
    if (table_idx_host.size() < tables.size()) table_idx_host.resize(tables.size());
    table_idx_host[(size_t)id].assign(idx, idx + n_idx);
    tables[id] = ssn::TableSlot{table_rows[id], table_idx[id], n_rows, width, n_idx, first_step};
    HIPCHK(hipMemcpy(d_tables + id, &tables[id], sizeof(ssn::TableSlot), hipMemcpyHostToDevice));
    return SSN_OK;
  }

  // Rows already in the simulator's type (the caller converted them): pure DMA into the inactive buffer set on the copy
  // stream - no kernel, so it proceeds while a compute kernel owns every CU, and any host thread may call it during ssn_run_steps.
  int stage_table(int id, const void* rows_t, int64_t n_rows, int64_t width, const int32_t* idx, int64_t n_idx, int64_t first_step) override {
    HIPCHK(hipSetDevice(device));
    if (id < 0 || id >= (int)tables.size()) return fail(SSN_EINVAL, "table id %d out of range", id);
    if (width != tables[id].width) return fail(SSN_EINVAL, "table %d is %lld wide, got %lld", id, (long long)tables[id].width, (long long)width);
    if (n_rows < 0 || n_idx < 0 || (!idx && n_idx) || (!rows_t && n_rows)) return fail(SSN_EINVAL, "bad table arguments");
    for (int64_t i = 0; i < n_idx; ++i)
      if (idx[i] >= n_rows) return fail(SSN_EINVAL, "table row index %d >= n_rows %lld", idx[i], (long long)n_rows);
    std::lock_guard<std::mutex> lock(stage_mutex);
    if (!copy_stream) HIPCHK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    Staged& g = staged[(size_t)id];
    g.valid = false;
    const int64_t need = std::max<int64_t>(1, n_rows * width);
    if (need > g.rows_cap) {
      if (g.rows) hipFree(g.rows);
      g.rows = nullptr;
      CHK(dmalloc((T**)&g.rows, need * (int64_t)sizeof(T)));
      g.rows_cap = need;
    }
    if (n_idx > g.idx_cap) {
      if (g.idx) hipFree(g.idx);
      g.idx = nullptr;
      CHK(dmalloc(&g.idx, n_idx * 4));
      g.idx_cap = n_idx;
    }
    if (n_rows) HIPCHK(hipMemcpyAsync(g.rows, rows_t, (size_t)(n_rows * width) * sizeof(T), hipMemcpyHostToDevice, copy_stream));
    if (n_idx) HIPCHK(hipMemcpyAsync(g.idx, idx, (size_t)n_idx * 4, hipMemcpyHostToDevice, copy_stream));
    HIPCHK(hipStreamSynchronize(copy_stream));          // (the host arrays are the caller's again)
    g.n_rows = n_rows; g.width = width; g.n_idx = n_idx; g.first_step = first_step;
    g.idx_host.assign(idx, idx + n_idx);
    g.valid = true;
    return SSN_OK;
  }

  int commit_tables() override {
    HIPCHK(hipSetDevice(device));
    if (async_active) return fail(SSN_EINVAL, "a stream-ordered run is in flight on the caller's stream: call ssn_phase_sync first");
    std::lock_guard<std::mutex> lock(stage_mutex);
    bool any = false;
    for (auto& g : staged) any = any || g.valid;
    if (!any) return SSN_OK;
    HIPCHK(hipStreamSynchronize(stream));              // the run that read the current set has ended
    if (table_idx_host.size() < tables.size()) table_idx_host.resize(tables.size());
    for (size_t id = 0; id < staged.size(); ++id) {
      Staged& g = staged[id];
      if (!g.valid) continue;
      std::swap(g.rows, table_rows[id]); std::swap(g.idx, table_idx[id]);
      std::swap(g.rows_cap, table_rows_cap[id]); std::swap(g.idx_cap, table_idx_cap[id]);
      table_idx_host[id].swap(g.idx_host);
      tables[id] = ssn::TableSlot{table_rows[id], table_idx[id], g.n_rows, g.width, g.n_idx, g.first_step};
      g.valid = false;
    }
    HIPCHK(hipMemcpy(d_tables, tables.data(), tables.size() * sizeof(ssn::TableSlot), hipMemcpyHostToDevice));
    return SSN_OK;
  }

  int reserve_probes(int64_t n) override {
    HIPCHK(hipSetDevice(device));
    if (async_active) return fail(SSN_EINVAL, "a stream-ordered run is in flight on the caller's stream: call ssn_phase_sync first");
    HIPCHK(hipStreamSynchronize(stream));
    reserve_first = steps_done;
    reserve_n = n;
    for (size_t p = 0; p < pslots.size(); ++p) {
      const int64_t every = probes[p].every;
      const int64_t base = steps_done / every;
      const int64_t cap = (steps_done + n) / every - base;
      const int64_t bytes = std::max<int64_t>(1, cap * probes[p].width) * (int64_t)sizeof(T);
      if (bytes > probe_cap_bytes[p]) {
        if (pslots[p].data) hipFree(pslots[p].data);
        pslots[p].data = nullptr;
        CHK(dmalloc((T**)&pslots[p].data, bytes));
        probe_cap_bytes[p] = bytes;
      }
      pslots[p].base_slot = base;
      pslots[p].capacity = cap;
    }
    if (!pslots.empty()) HIPCHK(hipMemcpy(d_pslots, pslots.data(), pslots.size() * sizeof(ssn::ProbeSlot), hipMemcpyHostToDevice));
    return SSN_OK;
  }

  int64_t probe_count(int id) override {
    if (id < 0 || id >= (int)pslots.size()) return -1;
    const int64_t every = probes[id].every;
    return std::min<int64_t>(pslots[id].capacity, steps_done / every - pslots[id].base_slot);
  }

  int read_probe(int id, double* dst, void* dst_dev, int64_t first, int64_t count) override {
    HIPCHK(hipSetDevice(device));
    if (async_active) return fail(SSN_EINVAL, "a stream-ordered run is in flight on the caller's stream: call ssn_phase_sync first");
    if (id < 0 || id >= (int)pslots.size()) return fail(SSN_EINVAL, "probe id %d out of range", id);
    if (first < 0 || count < 0 || first + count > probe_count(id))
      return fail(SSN_EINVAL, "probe %d holds %lld samples, asked for [%lld,+%lld)", id, (long long)probe_count(id), (long long)first, (long long)count);
    const int64_t w = probes[id].width;
    const T* src = (const T*)pslots[id].data + first * w;
    if (dst_dev) {
      HIPCHK(hipMemcpyAsync(dst_dev, src, (size_t)(count * w) * sizeof(T), hipMemcpyDeviceToDevice, stream));
      HIPCHK(hipStreamSynchronize(stream));
      return SSN_OK;
    }
    return download(src, dst, count, w, w);
  }

  int rw_signal(int64_t off, int64_t count, double* dst, const double* src) override {
    HIPCHK(hipSetDevice(device));
    if (async_active) return fail(SSN_EINVAL, "a stream-ordered run is in flight on the caller's stream: call ssn_phase_sync first");
    CHK(check_range(off, count, "signal access"));
    if (!dst) {
      CHK(upload(src, sig + off, 1, count, count));
      return bsig ? upload(src, bsig + off, 1, count, count) : SSN_OK;
    }
    CHK(download(sig + off, dst, 1, count, count));
    if (bsig) {                       // signals of the batched stages live in the block buffer (row 0 = latest)
      std::vector<double> tmp((size_t)count);
      CHK(download(bsig + off, tmp.data(), 1, count, count));
      for (int64_t i = 0; i < count; ++i) if (batched_mask[(size_t)(off + i)]) dst[i] = tmp[(size_t)i];
    }
    return SSN_OK;
  }

  int rw_buffer(int id, double* dst, const double* src, int64_t count) override {
    HIPCHK(hipSetDevice(device));
    if (async_active) return fail(SSN_EINVAL, "a stream-ordered run is in flight on the caller's stream: call ssn_phase_sync first");
    if (id < 0 || id >= (int)bufs.size()) return fail(SSN_EINVAL, "buffer id %d out of range", id);
    Buf& b = bufs[id];
    if (b.kind != SSN_BUF_REAL || count != b.count) return fail(SSN_EINVAL, "buffer %d: real buffer of %lld elements expected", id, (long long)b.count);
    if (dst) return download_buf(b, dst);
    if (b.packed == 2) return fail(SSN_EUNSUPPORTED, "buffer %d: the refractory times of a LIF ensemble array are packed with its voltages; write the voltage buffer", id);
    return upload_buf(b, src);
  }

  int counters(ssn_counters* out) override {
    out->n_steps = steps_done;
    out->launches_per_step = launches_per_step;
    out->dominant_launches = dom_launches;
    out->dominant_ms_total = dom_ms;
    out->dominant_bytes_per_launch = dom_bytes;
    out->dominant_units_per_launch = dom_units;
    out->last_run_ms = last_run_ms;
    out->device_bytes = device_bytes;
    out->block_tpb = fused_block ? blk.tpb : 0;
    out->block_npt = fused_block ? blk.npt : 0;
    out->block_enc_lds = fused_block ? blk.enc_lds : 0;
    out->block_threads = fused_block ? blk.threads : 0;
    out->block_members = fused_block ? std::max(1, blk.P) : 0;
    out->batch_products_skipped = batch_skipped;
    out->block_slots = 0; out->block_slots_silent = 0;
    out->fused_populations = n_fused_populations; out->serial_chains = n_serial_chains;
    if (fused_block && blk.slot_stats) {
      unsigned long long ss[2] = {0, 0};
      hipSetDevice(device);
      if (hipMemcpy(ss, blk.slot_stats, sizeof ss, hipMemcpyDeviceToHost) == hipSuccess) { out->block_slots = (int64_t)ss[0]; out->block_slots_silent = (int64_t)ss[1]; }
    }
    out->fft_transforms = 0; out->fft_bluestein = 0;
    for (auto& it : items) if (it.type == IT_DFT) { out->fft_transforms += 1; out->fft_bluestein += it.dft.M > 0 ? 1 : 0; }
    return SSN_OK;
  }

  int kernel_times(ssn_kernel_time* out, int capacity) override {
    static const char* names[N_ITEM_TYPES] = {"k_program", "k_ensarray", "k_matvec", "k_neurons", "k_pes", "k_voja", "k_matvec_ordered",
                                              "k_ens_finish", "k_spmv_partial", "k_dft", "k_vecops", "k_grid_lhs", "k_gemm_nt_mfma_f32",
                                              "k_argmax_partial", "k_round", "", "", ""};
    int n = 0;
    for (int t = 0; t < N_ITEM_TYPES; ++t) {
      if (!type_launches[t]) continue;
      if (n < capacity) {
        snprintf(out[n].name, sizeof out[n].name, "%s", names[t]);
        out[n].launches = type_launches[t];
        out[n].ms_total = type_ms[t];
      }
      ++n;
    }
    return n;
  }

  int64_t n_steps() override { return steps_done; }
};

template <typename T>
int create_sim(const ssn_model_desc* desc, ssn_sim** out) {
  auto s = std::make_unique<Sim<T>>();
  int rc = s->create(desc);
  if (rc != SSN_OK) return rc;
  *out = s.release();
  return SSN_OK;
}

}  // namespace

extern "C" {

int ssn_create(const ssn_model_desc* desc, ssn_sim** out) {
  if (!desc || !out) return fail(SSN_EINVAL, "null argument");
  *out = nullptr;
  if (desc->abi_version != SSN_ABI_VERSION) return fail(SSN_EINVAL, "ABI version %d != %d", desc->abi_version, SSN_ABI_VERSION);
  if (desc->n_signals <= 0 || !desc->signal_init || desc->n_ops <= 0 || !desc->ops || desc->dt <= 0)
    return fail(SSN_EINVAL, "empty or inconsistent model description");
  int ndev = 0;
  const hipError_t dev_err = hipGetDeviceCount(&ndev);
  if (dev_err != hipSuccess || ndev <= 0)
    return fail(SSN_EHIP, "no HIP device available (there is no CPU fallback): hipGetDeviceCount -> %s, %d devices", hipGetErrorString(dev_err), ndev);
  if (desc->device < 0 || desc->device >= ndev) return fail(SSN_EINVAL, "device %d out of range (%d devices)", desc->device, ndev);
  if (desc->dtype == SSN_F32) return create_sim<float>(desc, out);
  if (desc->dtype == SSN_F64) return create_sim<double>(desc, out);
  return fail(SSN_EINVAL, "unknown dtype %d", desc->dtype);
}

void ssn_destroy(ssn_sim* sim) { delete sim; }
int ssn_reset(ssn_sim* sim) { return sim ? sim->reset() : fail(SSN_EINVAL, "null simulator"); }
int ssn_set_table(ssn_sim* sim, int32_t id, const double* rows, int64_t n_rows, int64_t width, const int32_t* idx,
                  int64_t n_idx, int64_t first_step) {
  if (!sim || (!rows && n_rows)) return fail(SSN_EINVAL, "null argument");
  return sim->set_table(id, rows, nullptr, n_rows, width, idx, n_idx, first_step);
}
int ssn_set_table_device(ssn_sim* sim, int32_t id, const void* rows_dev, int64_t n_rows, int64_t width,
                         const int32_t* idx, int64_t n_idx, int64_t first_step) {
  if (!sim || !rows_dev) return fail(SSN_EINVAL, "null argument");
  return sim->set_table(id, nullptr, rows_dev, n_rows, width, idx, n_idx, first_step);
}
int ssn_stage_table(ssn_sim* sim, int32_t id, const void* rows_t, int64_t n_rows, int64_t width, const int32_t* idx, int64_t n_idx, int64_t first_step) {
  if (!sim) return fail(SSN_EINVAL, "null simulator");
  return sim->stage_table(id, rows_t, n_rows, width, idx, n_idx, first_step);
}
int ssn_commit_tables(ssn_sim* sim) { return sim ? sim->commit_tables() : fail(SSN_EINVAL, "null simulator"); }
int ssn_reserve_probes(ssn_sim* sim, int64_t n) { return sim ? sim->reserve_probes(n) : fail(SSN_EINVAL, "null simulator"); }
int ssn_run_steps(ssn_sim* sim, int64_t n, int32_t profile) { return sim ? sim->run_steps(n, profile) : fail(SSN_EINVAL, "null simulator"); }
int ssn_read_probe(ssn_sim* sim, int32_t id, double* dst, int64_t first, int64_t count) {
  if (!sim || (!dst && count)) return fail(SSN_EINVAL, "null argument");
  return sim->read_probe(id, dst, nullptr, first, count);
}
int ssn_read_probe_device(ssn_sim* sim, int32_t id, void* dst_dev, int64_t first, int64_t count) {
  if (!sim || !dst_dev) return fail(SSN_EINVAL, "null argument");
  return sim->read_probe(id, nullptr, dst_dev, first, count);
}
int64_t ssn_probe_count(ssn_sim* sim, int32_t id) { return sim ? sim->probe_count(id) : -1; }
int ssn_read_signal(ssn_sim* sim, int64_t off, int64_t count, double* dst) {
  if (!sim || !dst) return fail(SSN_EINVAL, "null argument");
  return sim->rw_signal(off, count, dst, nullptr);
}
int ssn_write_signal(ssn_sim* sim, int64_t off, int64_t count, const double* src) {
  if (!sim || !src) return fail(SSN_EINVAL, "null argument");
  return sim->rw_signal(off, count, nullptr, src);
}
int ssn_read_buffer(ssn_sim* sim, int32_t id, double* dst, int64_t count) {
  if (!sim || !dst) return fail(SSN_EINVAL, "null argument");
  return sim->rw_buffer(id, dst, nullptr, count);
}
int ssn_write_buffer(ssn_sim* sim, int32_t id, const double* src, int64_t count) {
  if (!sim || !src) return fail(SSN_EINVAL, "null argument");
  return sim->rw_buffer(id, nullptr, src, count);
}
int ssn_get_counters(ssn_sim* sim, ssn_counters* out) {
  if (!sim || !out) return fail(SSN_EINVAL, "null argument");
  return sim->counters(out);
}
int ssn_run_phase(ssn_sim* sim, int32_t phase) { return sim ? sim->run_phase(phase) : fail(SSN_EINVAL, "null simulator"); }
int64_t ssn_exchange_size(ssn_sim* sim) { return sim ? sim->exchange_size() : -1; }
int64_t ssn_cycle_steps(ssn_sim* sim) { return sim ? sim->cycle_len() : -1; }
int ssn_exchange_pack(ssn_sim* sim, void* dst_dev) {
  if (!sim || !dst_dev) return fail(SSN_EINVAL, "null argument");
  return sim->exchange_copy(dst_dev, true);
}
int ssn_exchange_unpack(ssn_sim* sim, const void* src_dev) {
  if (!sim || !src_dev) return fail(SSN_EINVAL, "null argument");
  return sim->exchange_copy(const_cast<void*>(src_dev), false);
}
int ssn_phase_async(ssn_sim* sim, int32_t phase, void* exchange_buf, void* hip_stream) {
  return sim ? sim->phase_async(phase, exchange_buf, (hipStream_t)hip_stream) : fail(SSN_EINVAL, "null simulator");
}
int ssn_phase_sync(ssn_sim* sim, void* hip_stream) { return sim ? sim->phase_sync((hipStream_t)hip_stream) : fail(SSN_EINVAL, "null simulator"); }
int ssn_get_kernel_times(ssn_sim* sim, ssn_kernel_time* out, int32_t capacity) {
  if (!sim || (!out && capacity > 0) || capacity < 0) return fail(SSN_EINVAL, "null argument");
  return sim->kernel_times(out, capacity);
}
int64_t ssn_n_steps(ssn_sim* sim) { return sim ? sim->n_steps() : -1; }
int ssn_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
const char* ssn_last_error(void) { return g_err.c_str(); }
const char* ssn_version(void) { return "libssn_hip 0.9 (gfx950, ABI 8)"; }

}  // extern "C"
