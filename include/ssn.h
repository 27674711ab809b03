/* ssn.h - C ABI of libssn_hip.so: the MI355X (gfx950) step-loop backend for SSP-SLAM spiking networks.
 *
 * Drop-in boundary.  The reference has no FFI of its own: its hot path is the
 * Python call  `sim = nengo.Simulator(model); with sim: sim.run(T); sim.data[probe]`
 * (reference experiments/run_pathint.py:147-165,171-181; experiments/run_slam.py:198-235,250-268),
 * whose per-timestep arithmetic is executed by the third-party `nengo` / `nengo_ocl` simulators.
 * This library replaces that simulator core.  A thin ctypes wrapper (sspslam_amd/simulator.py)
 * exposes the same Simulator API on top of the entry points below; each entry point names the
 * Simulator member it serves.
 *
 * Model = one flat signal vector + parameter/state buffers + an ordered operator list (the frozen
 * "BuiltModel", produced on the host by sspslam_amd/builder.py).  All host arrays are float64 (or
 * int32 for index buffers), borrowed for the duration of the call only and converted to the
 * simulator's arithmetic type on upload.  The library owns all device memory.
 *
 * Every call returns SSN_OK (0) or a negative error code; ssn_last_error() gives the message
 * (thread-local).  One host thread drives a simulator; nothing calls back into the host during
 * ssn_run_steps.
 */
#ifndef SSN_H
#define SSN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSN_ABI_VERSION 8

enum ssn_status {
  SSN_OK = 0,
  SSN_EINVAL = -1,        /* bad argument / inconsistent model description        */
  SSN_EHIP = -2,          /* a HIP runtime call failed (message has the HIP error) */
  SSN_ERCCL = -3,         /* reserved                                             */
  SSN_ENOMEM = -4,        /* host or device allocation failed                     */
  SSN_EUNSUPPORTED = -5   /* operator shape outside what the kernels implement    */
};

enum ssn_dtype { SSN_F32 = 0, SSN_F64 = 1 };           /* arithmetic + state type of a simulator   */
enum ssn_buffer_kind { SSN_BUF_REAL = 0, SSN_BUF_I32 = 1 };
enum ssn_neuron { SSN_LIF = 0, SSN_LIFRATE = 1, SSN_RELU = 2 };

/* Operator kinds; field use per kind is listed next to ssn_op_desc. */
enum ssn_op_kind {
  SSN_OP_FILL = 1, SSN_OP_TABLE = 2, SSN_OP_AXPY = 3, SSN_OP_MATVEC = 4, SSN_OP_LOWPASS = 5,
  SSN_OP_ENSARRAY = 6, SSN_OP_NEURONS = 7, SSN_OP_PES = 8, SSN_OP_VOJA = 9, SSN_OP_CLEANUP = 10,
  SSN_OP_GATE = 11, SSN_OP_LINCOMB = 12
};

typedef struct ssn_buffer_desc {
  const void* data;     /* float64 (SSN_BUF_REAL) or int32 (SSN_BUF_I32) host array, C order */
  int64_t count;        /* number of elements                                               */
  int32_t kind;         /* ssn_buffer_kind                                                  */
  int32_t reserved;
} ssn_buffer_desc;

/* One operator.  Offsets are element offsets into the signal vector; "buf" = index into buffers[].
 *  FILL     i0 dst  i1 len                                  f0 value
 *  TABLE    i0 dst  i1 width i2 table_id                                      (ssn_set_table)
 *  AXPY     i0 dst  i1 src   i2 len  i3 mode(0 inc,1 set)   f0 alpha
 *  MATVEC   i0 dst  i1 src   i2 rows i3 cols i4 W buf i5 mode i6 dft          W is rows x cols; dft != 0: W is
 *           the real-DFT map of a circular-convolution network (1-4 transform_in A / B / conj A / conj B, 5 transform_out,
 *           reference binding.py:23-74) - the f32 core may then run k_dft (mixed-radix FFT) instead of W
 *  LINCOMB  i0 dst  i1 len   i2 n_terms i3 src buf (int32 [n_terms]: signal offsets) i4 alpha buf (real [n_terms])
 *                                                           f0 self f1 const  dst=self*dst+(const+sum_k alpha_k*sig[src_k+i])
 *           (the folded linear glue between two big operators - chains of nengo Reset / ElementwiseInc / copy operators
 *           collapsed at build time; per-timestep core only)
 *  LOWPASS  i0 dst  i1 src   i2 len                         f0 a  f1 gain     dst=a*dst+(1-a)*gain*src
 *  ENSARRAY i0 x    i1 K i2 n i3 din i4 dout i5 enc buf [K][din][n] i6 bias buf [K][n]
 *           i7 dec buf [K][dout][n] i8 dst_idx buf (int32 [K][dout]) i9 V buf i10 R buf
 *           i11 neuron                                      f0 tau_rc f1 tau_ref f2 min_voltage
 *  NEURONS  i0 J    i1 out   i2 n i3 V buf i4 R buf i5 neuron  f0 tau_rc f1 tau_ref f2 min_voltage f3 amp
 *  PES      i0 W buf i1 rows i2 cols i3 err i4 act          f0 kappa          W += kappa*outer(err,act)
 *  VOJA     i0 E buf i1 rows i2 cols i3 spk i4 key i5 learn i6 scale buf  f0 lr*dt
 *  CLEANUP  i0 dst  i1 src   i2 rows i3 cols i4 table buf                     dst = T[argmax(T@src)]
 *           optional factor tables of a sample grid (i5 != 0; buffer ids + 1): i5 dft buf (2K x cols: Re / Im rows of the
 *           half spectrum), i6 lhs buf (i8 x 2K), i7 rhs buf (i9 x 2K), i8 * i9 = rows, i10 = 2K, (Re, Im) interleaved, with
 *           T[a * i9 + r] . x = sum_k Re(conj(X_k) lhs[a, k] rhs[r, k]) - the f32 core then forms the similarities of a
 *           large table (>= 64 MB) as one MFMA product instead of a pass over the table (reference slam.py:209-215)
 *  GATE     i0 dst  i1 src   i2 d                           f0 thres f1 rate  (reference slam.py:233-237)
 * level: scheduling round from the host builder; vector ops of equal level touch disjoint data. */
typedef struct ssn_op_desc {
  int32_t kind;
  int32_t level;
  int32_t stage;        /* 0 pre, 1 core (stepped one timestep at a time), 2 post; pre/post run time-batched  */
  int32_t border;       /* position in the time-batched order of its stage (pre/post), else -1              */
  int32_t src_prev;     /* batched: source operand is a synapse state read before its update (previous row) */
  int32_t phase;        /* neuron-sharded models: 0 = before the per-timestep exchange, 1 = after it (the updates)  */
  int64_t i[12];
  double f[4];
} ssn_op_desc;

typedef struct ssn_probe_desc {
  int64_t src;          /* signal offset  */
  int64_t width;
  int64_t every;        /* sample when (step % every) == 0, steps counted from 1 */
  int64_t stage;        /* 1: sampled by the core every step; 0/2: sampled by a time-batched stage */
} ssn_probe_desc;

typedef struct ssn_range { int64_t lo, hi; } ssn_range;   /* signal range [lo, hi) */

typedef struct ssn_model_desc {
  int32_t abi_version;  /* SSN_ABI_VERSION */
  int32_t dtype;        /* ssn_dtype       */
  int32_t device;       /* HIP device index */
  int32_t n_tables;
  double dt;
  int64_t n_signals;
  const double* signal_init;          /* n_signals values */
  int32_t n_buffers;
  int32_t n_ops;
  int32_t n_probes;
  int32_t steps_per_graph;            /* timesteps captured per hipGraph (at most 128); 0 = library default (64 where blocks hold whole graphs, else 16), 1 = no graph */
  const ssn_buffer_desc* buffers;
  const ssn_op_desc* ops;
  const ssn_probe_desc* probes;
  /* stage boundaries (host builder, stages.py): signals the pre stage hands to the core at the start of
   * each timestep, and signals the core hands to the post stage at the end of each timestep */
  int32_t n_pre_to_core;
  int32_t n_core_to_post;
  const ssn_range* pre_to_core;
  const ssn_range* core_to_post;
  /* Neuron-sharded model (one rank of a SLAMNetwork split over several GPUs): the partial sums in these signal
   * ranges are completed by an all-reduce between the two phases of every timestep.  The library does not communicate: the
   * caller steps with ssn_run_phase(0), exchanges (ssn_exchange_pack -> all-reduce -> ssn_exchange_unpack), ssn_run_phase(1). */
  int32_t n_exchange;
  int32_t reserved;
  const ssn_range* exchange;
  int32_t block_steps;                /* timesteps per time-batched block; 0 = library default (1024, or 256 for very wide models) */
  int32_t flags;                      /* debug / A-B switches (default 0; tests check that every alternative plan gives the same results):
                                         1 = no fused recurrent-array core (generic programs),
                                         2 = no LIF fast path (unpacked state, dense row-major decoders),
                                         4 = LIF fast path with dense decoders (no spike-sparse gather),
                                         8 = dense decoder products for dense ensembles (no k_spmv_partial),
                                         16 = fused recurrent-array core always with a separate finish kernel
                                              (default: finish deferred into the next step's prologue, 1 launch per step),
                                         128 = no whole-block kernel for a recurrent array of independent ensembles
                                              (k_ens_block): step it once per timestep (k_ensarray) instead,
                                         512 = no FFT kernel for DFT-structured matvecs (always multiply by the matrix),
                                         1024 = k_spmv_partial rebuilds the spike list itself (no segmented list from k_neurons),
                                         4096 = one launch per operator (adjacent independent operators of one kind are not batched),
                                         8192 = ensemble arrays always leave partial sums for a finish operator (no direct write
                                              when one workgroup covers an ensemble),
                                         262144 = one launch per element-wise operator of the time-batched stages (no batching),
                                         524288 = clean-up similarities always from the pass over the table (no factored grid),
                                         2097152 = no rounds: one launch per big operator and one k_program launch per run of
                                              small ones (the round-1 plan; flag 4096 only acts
                                              together with this one).  Default: every operator takes the earliest round its data
                                              hazards allow and a round is ONE heterogeneous grid (k_round),
                                         4194304 = ensemble arrays are launched on their own, not as bodies of the round's grid,
                                         8388608 = the rounds of one timestep at a time (default: the steps_per_graph timesteps of a
                                              step graph are software-pipelined - an operator of step s + 1 may share a round with
                                              operators of step s),
                                         16777216 = heavy operators stay whole (default: the pipelined plan may run the blocks of a
                                              bandwidth-bound operator in pieces over the rounds of its slack window),
                                         33554432 = merged element-wise operators stay whole (default: cut at the range endpoints of
                                              the other operators, so that each piece has its own hazards),
                                         134217728 = no chains (default: an element-wise micro-operator whose only hazards inside a
                                              round are on identical element ranges joins that round and runs behind its
                                              predecessor in the same block),
                                         268435456 = k_dft also for chirp-z (Bluestein) transforms of 2048 points and more
                                              (default: their dense matrix - one workgroup needs 40 us for such a transform).
                                         536870912 = the four-step FFT on the matrix cores (two small dense DFTs as f32 MFMA products around a
                                              twiddle multiply, round 3; any factorisation, primes up to 192 as one dense DFT) instead of
                                              the Stockham passes (generic radix-r butterflies through LDS).  Correct for every length of the
                                              tests, but measured no faster (its operand loads are latency-bound): opt-in.
                                         1073741824 = split ensembles in the whole-block kernel (f32, at most 4 decoded rows): an array with fewer
                                              ensembles than the GPU has CUs (a 4- or 8-GPU shard of config 2) is stepped by 2 or 4
                                              member workgroups per ensemble that exchange their partial sums every timestep; needs every
                                              workgroup of the launch resident at once, i.e. the GPU for this process alone.
                                         (Round 1's opt-in experiments 32, 64, 2048, 16384, 32768 the multi-stream step graph 256 and the program-planner switches 65536, 131072, 1048576 of
                                          the round-1 plan - all measured slower - were removed.) */
} ssn_model_desc;

typedef struct ssn_counters {
  int64_t n_steps;                  /* steps executed since create/reset                          */
  int64_t launches_per_step;        /* kernel launches in one timestep                            */
  int64_t dominant_launches;        /* timed launches of the dominant (ensemble-array) kernel      */
  double dominant_ms_total;         /* sum of their durations, HIP events on the launching stream */
  double dominant_bytes_per_launch; /* algorithmic bytes of one such launch           */
  int64_t dominant_units_per_launch;/* neuron-steps per launch                                    */
  double last_run_ms;               /* device time of the last ssn_run_steps (events)             */
  int64_t device_bytes;             /* device memory held by this simulator                       */
  /* whole-block kernel (k_ens_block) variant the planner picked; all 0 when the core is stepped per timestep */
  int32_t block_tpb;                /* workgroup size the variant is compiled for                  */
  int32_t block_npt;                /* neurons per thread                                         */
  int32_t block_enc_lds;            /* 1: encoders live in LDS, 0: in registers                   */
  int32_t block_threads;            /* threads actually launched per workgroup                    */
  int32_t fft_transforms;           /* DFT-structured matvecs of a timestep that run as k_dft (FFT) instead of the matrix */
  int32_t fft_bluestein;            /* ... of which through Bluestein's convolution (a prime factor > 32)           */
  int32_t block_members;            /* member workgroups per ensemble of the whole-block kernel (flag 1073741824; 1 = not split, 0 = no block kernel) */
  int32_t batch_products_skipped; /* time-batched products not multiplied out because their whole input was zero over the block (cumulative) */
  /* whole-block kernel (f32): (wave, round) slots stepped since create / reset, and how many of them were silent - no neuron of the
   * slot spiked in the timestep, so the spike-time arithmetic and the decode were left out (ABI 7; the VALU roofline of bench.py
   * prices the two paths with these) */
  int64_t block_slots;
  int64_t block_slots_silent;
  /* round plan (ABI 8) */
  int32_t fused_populations;        /* dense populations whose neuron update runs in the epilogue of their encoder product      */
  int32_t serial_chains;            /* serial chains of single-workgroup operators in one timestep's rounds (the eager plan)    */
} ssn_counters;

/* Per-kernel device time of the generic (one launch per operator) plan, collected by ssn_run_steps(profile = 2):
 * every launch of the timed timesteps is bracketed with HIP events on the simulator's stream. */
typedef struct ssn_kernel_time {
  char name[32];                    /* kernel of the plan item: "k_program", "k_matvec", "k_ensarray", ... */
  int64_t launches;
  double ms_total;
} ssn_kernel_time;

typedef struct ssn_sim ssn_sim;

/* Simulator(network): upload the built model, plan kernels, capture the step graph. */
int ssn_create(const ssn_model_desc* desc, ssn_sim** out);
/* Simulator.close() / __exit__ */
void ssn_destroy(ssn_sim* sim);
/* Simulator.reset(): restore initial signals, neuron state and learned buffers; step = 0. */
int ssn_reset(ssn_sim* sim);

/* Pre-tabulated output of a t-only Node (the host evaluates the user's closure once per step of
 * the coming run): rows[n_rows][width] unique rows, idx[n_idx] row per step (-1 = zeros) for steps
 * first_step .. first_step+n_idx-1 (0-based).  Serves Node(lambda t: ...) at run_pathint.py:134-136. */
int ssn_set_table(ssn_sim* sim, int32_t table_id, const double* rows, int64_t n_rows, int64_t width,
                  const int32_t* idx, int64_t n_idx, int64_t first_step);
/* Same, rows already on the device in the simulator's dtype (multi-GPU exchange path). */
int ssn_set_table_device(ssn_sim* sim, int32_t table_id, const void* rows_dev, int64_t n_rows,
                         int64_t width, const int32_t* idx, int64_t n_idx, int64_t first_step);

/* The same in two halves for chunked runs (Simulator.run(T) over many chunks: the node closures of chunk k + 1 are evaluated
 * while the device steps chunk k, reference run_pathint.py:160-165 times the whole of it): ssn_stage_table takes rows ALREADY in
 * the simulator's dtype (float for SSN_F32, double for SSN_F64) and copies them by DMA into a second set of device buffers - it
 * may be called from another host thread while ssn_run_steps is in flight; ssn_commit_tables, called between two runs, makes
 * every staged table the current one.  (ABI 7) */
int ssn_stage_table(ssn_sim* sim, int32_t table_id, const void* rows_typed, int64_t n_rows, int64_t width,
                    const int32_t* idx, int64_t n_idx, int64_t first_step);
int ssn_commit_tables(ssn_sim* sim);

/* Make room for the probe samples of the next n_steps (drops samples already read). */
int ssn_reserve_probes(ssn_sim* sim, int64_t n_steps);
/* Simulator.run_steps(n): blocking. profile = 1 times every dominant-kernel launch with events; profile = 2 times
 * every launch of the per-timestep plan (eager launches instead of graph replay) for ssn_get_kernel_times. */
int ssn_run_steps(ssn_sim* sim, int64_t n, int32_t profile);
/* Simulator.data[probe]: samples [first, first+count) of the current reservation -> dst[count][width]. */
int ssn_read_probe(ssn_sim* sim, int32_t probe_id, double* dst, int64_t first, int64_t count);
/* Same into device memory (simulator dtype), no host round trip. */
int ssn_read_probe_device(ssn_sim* sim, int32_t probe_id, void* dst_dev, int64_t first, int64_t count);
int64_t ssn_probe_count(ssn_sim* sim, int32_t probe_id);

/* Signals and buffers (learned weights, encoders, neuron state): checkpointing, tests, weight probes. */
int ssn_read_signal(ssn_sim* sim, int64_t off, int64_t count, double* dst);
int ssn_write_signal(ssn_sim* sim, int64_t off, int64_t count, const double* src);
int ssn_read_buffer(ssn_sim* sim, int32_t buffer_id, double* dst, int64_t count);
int ssn_write_buffer(ssn_sim* sim, int32_t buffer_id, const double* src, int64_t count);

/* Neuron-sharded models: one half of a timestep (phase 0: everything up to the exchange; phase 1: the updates, probe
 * sampling, step counter; phase 2 = phase 1 followed by phase 0 of the next timestep in one launch, for the inside of a
 * run: 0, x, 2, x, 2, ..., x, 1 with x the caller's exchange).  Blocking.  ssn_run_steps refuses such models. */
int ssn_run_phase(ssn_sim* sim, int32_t phase);
/* Pipelined over the exchange (ABI 8): C = ssn_cycle_steps() timesteps planned together as C + 1 launch segments, the exchange
 * of timestep k between segments k and k + 1 - what does not hang on that exchange (the head of timestep k + 1) shares rounds
 * with what does.  phase = 3 (ssn_run_phase and ssn_phase_async alike) runs the next segment; a cycle is
 *   3, x, 3, x, ..., x, 3   (C + 1 segments, C exchanges x)
 * it starts at a timestep boundary, and phases 0 / 1 / 2 are refused until its last segment has run (they serve the timesteps that
 * do not fill a cycle).  0: the model has no such plan (not neuron-sharded, or planned with flag 8388608 / SSN_CYCLE_STEPS=0). */
int64_t ssn_cycle_steps(ssn_sim* sim);
/* Elements of the exchange (sum of the range lengths); copy them to / from one contiguous device buffer of the simulator's dtype. */
int64_t ssn_exchange_size(ssn_sim* sim);
int ssn_exchange_pack(ssn_sim* sim, void* dst_dev);
int ssn_exchange_unpack(ssn_sim* sim, const void* src_dev);

/* Stream-ordered stepping of a neuron-sharded model: no host synchronisation per timestep (SURVEY 8b "no callbacks into Python
 * from the step loop", 8e; the loop it serves closes every timestep: reference networks/slam.py:259,306-307).  Enqueues, as ONE
 * graph launch on `hip_stream` (a hipStream_t of the caller - the stream its collective is ordered on, e.g. the current stream of
 * torch.distributed's RCCL backend; NULL = the simulator's own stream), and returns without waiting:
 *   [phase 1 / 2: exchange_buf (the all-reduced sums) -> the exchange ranges]  the phase's kernels
 *   [phase 0 / 2: the exchange ranges (this rank's partial sums) -> exchange_buf]
 * exchange_buf: device buffer of ssn_exchange_size() elements of the simulator's dtype, the same pointer throughout a run (it is
 * captured into the graphs); NULL when the caller does not exchange (a single rank).  A run is
 *   ssn_phase_async(0), [collective on hip_stream], ssn_phase_async(2), [collective], ..., ssn_phase_async(1), ssn_phase_sync.
 * phase = -1 only builds the graphs for exchange_buf (otherwise built by the first call of a run) and launches nothing;
 * phase = 3: the next segment of a pipelined cycle (ssn_cycle_steps above), [unpack] / [pack] around it as for phases 2 / 0. */
int ssn_phase_async(ssn_sim* sim, int32_t phase, void* exchange_buf, void* hip_stream);
/* Waits for everything ssn_phase_async enqueued on that stream; checks the device step counter and the probe-overflow flag. */
int ssn_phase_sync(ssn_sim* sim, void* hip_stream);

/* Measurement aid for the roofline of the VALU-bound whole-block kernel (bench.py; not on the step path): `iters` x 64 independent
 * wave64 instructions of one kind per wave, one workgroup of 256 x waves_per_simd threads on every CU; reports the nanoseconds of
 * SIMD time per wave instruction from the launch's HIP-event time (and the shader clock during the launch, s_memtime ticks per
 * 100 MHz s_memrealtime tick).  csrc/ssn_probe.hip. */
enum ssn_probe_kind {
  SSN_PROBE_PK_FMA = 0, SSN_PROBE_PK_MUL = 1, SSN_PROBE_PK_ADD = 2,   /* v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32            */
  SSN_PROBE_TRANS = 3,                                                /* v_rcp_f32 (v_log_f32 issues at the same rate)        */
  SSN_PROBE_FMA = 4, SSN_PROBE_ADD = 5,                               /* v_fma_f32; v_add_f32 (v_mul_f32, v_sub_f32)          */
  SSN_PROBE_DPP = 6, SSN_PROBE_MOV = 7, SSN_PROBE_READLANE = 8,       /* v_add_f32_dpp; v_mov_b32; v_readlane_b32             */
  SSN_PROBE_CNDMASK = 9, SSN_PROBE_OTHER = 10,                        /* v_cndmask_b32 (vcc); v_max_i32 (every other VALU op) */
  SSN_PROBE_N_KINDS = 11
};
int ssn_probe_issue_rate(int32_t device, int32_t kind, int32_t waves_per_simd, int32_t iters,
                         double* ns_per_wave_instruction, double* shader_mhz);

int ssn_get_counters(ssn_sim* sim, ssn_counters* out);
/* Fills at most `capacity` entries; returns the number of kernels with timed launches (or a negative status). */
int ssn_get_kernel_times(ssn_sim* sim, ssn_kernel_time* out, int32_t capacity);
int64_t ssn_n_steps(ssn_sim* sim);
int ssn_device_count(void);
const char* ssn_last_error(void);
const char* ssn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SSN_H */
