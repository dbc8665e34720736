/* bayesic_hip.h -- C ABI of libbayesic_hip.so, the MI355X (gfx950) backend for
 * Bayesic's ELBO-gradient / mean-field update path.
 *
 * The reference (mjwillson/Bayesic) has no FFI: its numeric boundary is the
 * Python hook set that emits Theano ops.  Every entry
 * point below names the reference interface it stands behind.  All functions:
 *
 *   - are extern "C", take plain pointers and sizes, and return 0 on success or
 *     a negative bsc_status; bsc_last_error() gives the thread-local message;
 *   - take DEVICE pointers unless the parameter is named host_*;
 *   - are asynchronous on the context's HIP stream (no host sync, no
 *     allocation) unless documented otherwise, so a caller may capture a
 *     sequence of them into a hipGraph;
 *   - never fall back to a CPU implementation.
 *
 * Arithmetic: data operands float32 (the reference default dtype,
 * bayesic/algebra.py:109); statistics, gradients and variational parameters
 * float64 unless stated.
 */
#ifndef BAYESIC_HIP_H
#define BAYESIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSC_VERSION 100 /* 0.1.0 */
#define BSC_MAX_RANK 6  /* strided-tensor entry points accept up to 6 axes */

typedef enum bsc_status {
    BSC_OK = 0,
    BSC_ERR_INVALID = -1,     /* bad argument (shape, alignment, null pointer) */
    BSC_ERR_HIP = -2,         /* a HIP runtime call failed */
    BSC_ERR_UNSUPPORTED = -3, /* valid request outside what the kernels cover */
    BSC_ERR_NOMEM = -4
} bsc_status;

typedef struct bsc_ctx bsc_ctx;

/* ---- context, memory, timing (build decision; reference has none) ------- */

/* Binds to `device`; `stream` is a hipStream_t (or NULL for the null stream)
 * owned by the caller -- e.g. torch.cuda.current_stream().cuda_stream. */
int bsc_ctx_create(int device, void* stream, bsc_ctx** out);
int bsc_ctx_destroy(bsc_ctx* ctx);
int bsc_ctx_set_stream(bsc_ctx* ctx, void* stream);
/* Pre-size the context workspace (partial-sum slabs).  Synchronous.  Entry
 * points grow it on demand; call this first when capturing into a graph. */
int bsc_ctx_reserve(bsc_ctx* ctx, size_t bytes);
int bsc_ctx_sync(bsc_ctx* ctx);

/* ---- a launch sequence as one hipGraph ------------------------------------------------------
 * The reference re-walks its expression tree on every evaluation (bayesic/algebra.py:34-40, 60-61,
 * 767-768); a device update built from a few dozen short launches is then bound by the host walk,
 * not by the GPU.  Between bsc_capture_begin and bsc_capture_end every entry point called on the
 * context records its launches into a graph instead of running them (hipStreamBeginCapture on the
 * context's stream -- which must not be the null stream); bsc_graph_launch replays them in one
 * submission.  The graph holds the POINTERS the calls were given: the caller keeps every buffer
 * alive and at the same address, and refreshes inputs by writing into them.  Entry points that
 * synchronise (bsc_ctx_sync, bsc_d2h, bsc_h2d, a workspace that has to grow) fail inside a capture:
 * run the sequence once eagerly first.  bsc_capture_end returns an error, and leaves the stream
 * usable, when something inside could not be captured. */
typedef struct bsc_graph bsc_graph;
int bsc_capture_begin(bsc_ctx* ctx);
int bsc_capture_end(bsc_ctx* ctx, bsc_graph** out);
int bsc_graph_launch(bsc_ctx* ctx, bsc_graph* g);
int bsc_graph_destroy(bsc_graph* g);
/* info[0]=CU count, [1]=wavefront size, [2]=max LDS bytes per workgroup,
 * [3]=clock kHz, [4]=L2 bytes, [5]=gcn arch number (950 for gfx950). */
int bsc_device_info(bsc_ctx* ctx, int64_t info[8]);
const char* bsc_last_error(void);
int bsc_version(void);

int bsc_malloc(bsc_ctx* ctx, size_t bytes, void** out); /* synchronous */
int bsc_free(bsc_ctx* ctx, void* ptr);                  /* synchronous */
int bsc_h2d(bsc_ctx* ctx, void* dst, const void* host_src, size_t bytes);
int bsc_d2h(bsc_ctx* ctx, void* host_dst, const void* src, size_t bytes); /* syncs */
int bsc_memset(bsc_ctx* ctx, void* dst, int value, size_t bytes);

/* Arithmetic of the MFMA-bound contractions.  terms = 0 (default; the dtype of every reported figure): f32
 * operands on the f32 MFMA, products and sums exactly those of an fmaf chain.  terms = 2 or 3: every f32
 * operand is written as a sum of that many bf16 terms (round to nearest, repeatedly) and the product is
 * taken on the bf16 MFMA -- 16 x the f32 rate per instruction -- from the 3 resp. 6 leading cross products,
 * accumulated in f32: relative error per product <= ~2^-17 resp. ~2^-23 (the f32 class).  Honoured by
 * bsc_lda_sstats / bsc_lda_sstats_bound at K = 128 (2 and 3), bsc_logreg_bbvi_loglik for 36 <= S <= 64, S % 4 == 0
 * (2; with 3 it computes as with 0) and bsc_mog_estep (either value: its forward product -- differences of large
 * terms -- always takes three terms, its backward two) and by bsc_gemm_strided_batched / bsc_gemm_epilogue when both
 * operands are the SAME row-major matrix of <= 256 columns, a multiple of 32 (X^T X: 2); every other entry computes as
 * with 0.  Also the option
 * "mfma_split" of bsc_ctx_set_option.  Replaces nothing in the reference (bayesic/algebra.py:1347-1383
 * contracts in the dtype of its operands); SURVEY.md 8(d) config 4 leaves the operand-split variant open. */
int bsc_ctx_set_mfma_split(bsc_ctx* ctx, int terms);

/* Kernel selection is a property of the CONTEXT, set by explicit calls: the library reads no environment
 * variable.  Every A/B switch and tuning knob is a named integer option (csrc/bsc_api.hip OPTIONS: "blr_q",
 * "gemm_dma", "lda_stream", "mfma_split", ...; bsc_ctx_option_name enumerates them, *host_key = NULL past the
 * last); a value outside an option's accepted set is an error, nothing is coerced.  Options that select
 * deletion builds (kernels with parts removed, for timing: their RESULTS ARE WRONG) are refused until the
 * option "profiling_builds" has been set to 1 on the same context, and then warn on stderr.  Defaults are the
 * measured best; a drop-in caller never needs these.  (The reference has no counterpart: its one backend
 * switch is Theano's own configuration, bayesic/algebra.py:54.) */
int bsc_ctx_set_option(bsc_ctx* ctx, const char* key, int64_t value);
int bsc_ctx_get_option(bsc_ctx* ctx, const char* key, int64_t* host_value);
int bsc_ctx_option_name(int32_t index, const char** host_key);

/* Per-kernel timing of the dominant kernel of each entry point, with hipEvents
 * recorded on the ctx stream immediately around that one launch.  enable = 0: off
 * (default); enable = n >= 1: time every n-th such launch (an event pair costs a
 * few microseconds of stream time, so a sampling period keeps the perturbation of
 * a timed region small).  bsc_ctx_profile_read synchronises, returns the summed
 * duration and the number of TIMED launches since the last read, and resets both. */
int bsc_ctx_profile(bsc_ctx* ctx, int enable);
int bsc_ctx_profile_read(bsc_ctx* ctx, double* host_total_ms, int64_t* host_launches);
/* The same reading per timing slot: 0 = the dominant kernel (what bsc_ctx_profile_read
 * returns), 1 = the collective of bsc_allreduce_sum, 2 = the finish kernel of
 * bsc_blr_fused_update.  Every slot is sampled with the period given to bsc_ctx_profile. */
int bsc_ctx_profile_read_slot(bsc_ctx* ctx, int slot, double* host_total_ms,
                              int64_t* host_launches);

/* ---- the exchange step of the data-parallel update: RCCL over xGMI -------------------
 * (ABSENT in reference; README.md:69-79 mini-batch SVI)  Rows are iid
 * (bayesic/distribution/base.py:146-172), every per-batch quantity is a SUM over rows, so
 * rank r of p holds a row block and the ranks exchange ONE vector per update: the
 * all-reduce(sum) of the concatenated statistic / gradient vector, after which every rank
 * applies the identical update.  The communicator is an ncclComm_t owned by the context;
 * the collective is enqueued on the context's own stream (pass -> all-reduce -> finish is
 * one in-order queue).  librccl is bound at run time by the first bsc_comm_* call, so a
 * single-GPU caller never maps it.
 *
 * bsc_comm_unique_id: rank 0 fills host_id[BSC_COMM_ID_BYTES] (ncclGetUniqueId) and hands
 *   the bytes to the other ranks over any host channel (a file, MPI, torch.distributed's
 *   store).  bsc_comm_init_rank: collective over all `world` ranks, synchronous; one
 *   communicator per context; world == 1 is allowed (the RCCL path at world size 1).
 *   bsc_comm_destroy is implied by bsc_ctx_destroy.
 * bsc_allreduce_sum / _max: in place over buf[n] (BSC_F32 | BSC_F64) on the ctx stream;
 *   on a context without a communicator they are the identity (a world of one). */
#define BSC_COMM_ID_BYTES 128
int bsc_comm_unique_id(void* host_id);
int bsc_comm_init_rank(bsc_ctx* ctx, const void* host_id, int32_t rank, int32_t world);
int bsc_comm_destroy(bsc_ctx* ctx);
int bsc_comm_info(bsc_ctx* ctx, int32_t* host_rank, int32_t* host_world,
                  int32_t* host_rccl_version);
int bsc_allreduce_sum(bsc_ctx* ctx, void* buf, int64_t n, int dtype);
int bsc_allreduce_max(bsc_ctx* ctx, void* buf, int64_t n, int dtype);
/* The same collective OVERLAPPED with the kernels that follow on the context's stream: _begin orders the
 * all-reduce behind everything enqueued so far (an event) and issues it on a second stream of the context;
 * kernels enqueued afterwards run beside it; _end(slot) makes the context's stream wait for that collective
 * (call it before the first kernel that reads `buf`).  slot < 16 names the collective in flight; each _begin
 * is ended once.  A context without a communicator (a world of one) does nothing in either; inside a graph
 * capture the collective stays on the captured stream.  Used where a statistic is finished piece by piece
 * (svi/lda.py: config 4's 51.2 MB all-reduce) -- only the last piece's collective is
 * exposed.  bsc_ctx_profile's slot 1 times every piece. */
int bsc_allreduce_sum_begin(bsc_ctx* ctx, void* buf, int64_t n, int dtype, int32_t slot);
int bsc_allreduce_sum_end(bsc_ctx* ctx, int32_t slot);

/* hipEvent wrappers so a ctypes caller can time the ctx stream. */
/* Measurement aid: best pure streaming-read rate (GB/s) over `buf` on this device, from
 * kernels that do nothing but read it: 16-byte loads into registers (two launch shapes) and
 * LDS-DMA of whole 1-KiB runs (one and two workgroups per CU; the first 4 GiB of `buf`) --
 * best of `reps` timed launches each.  Synchronises.  bench.py quotes the data pass against
 * this as well as against the spec-sheet peak. */
int bsc_hbm_read_probe(bsc_ctx* ctx, const void* buf, size_t bytes, int reps, double* host_gbps);

/* ---- mini-batch streaming: host memory -> HBM slots on a copy stream --------------
 * (README.md:69-79 "stochastic updates applied via subsampled minibatches"; the
 * reference has no loader -- SURVEY.md 8(f) rank 4.)  While the update of batch t
 * runs on the context's stream, batch t+1 crosses PCIe; the update kernels only ever
 * see device-resident batches.  submit queues the copy (after the kernels that last
 * read the slot), acquire makes the context's stream wait for the oldest submitted
 * batch and returns its device pointers, release marks the point on the context's
 * stream after which the slot may be overwritten.  At most n_slots batches may be
 * between submit and release.  Host buffers should be page-locked BY THE RUNTIME: allocated
 * with bsc_host_alloc (hipHostMalloc; torch pin_memory tensors qualify).  Memory that malloc
 * owns is never shown to the device -- there is no register-in-place entry point (round 2 had
 * one; a heap page it had registered, or the runtime had locked on the fly, was the address of a
 * later GPU memory fault: DESIGN.md section 10): a PAGEABLE source is copied by
 * the host, inside submit, into a page-locked bounce buffer of the slot, at a fraction of the
 * PCIe rate; it is free again when submit returns.  bsc_host_free drains the device before it
 * frees.  HOST BUFFER LIFETIME of a page-locked source: it
 * must stay valid and unchanged until submission k + n_slots has returned (submit
 * host-synchronises on the copy that last targeted the slot it is about to reuse), or
 * until bsc_loader_destroy.  Not thread-safe. */
typedef struct bsc_loader bsc_loader;
int bsc_host_alloc(size_t bytes, void** out);      /* hipHostMalloc: page-locked memory to stream from */
int bsc_host_free(void* host_ptr);
int bsc_loader_create(bsc_ctx* ctx, int64_t max_rows, int32_t D, int32_t n_slots, bsc_loader** out);
int bsc_loader_destroy(bsc_loader* loader);
int bsc_loader_submit(bsc_loader* loader, const float* host_X, int64_t ldx, const float* host_y,
                      int64_t rows);
int bsc_loader_acquire(bsc_loader* loader, const float** dX, const float** dy, int64_t* rows);
int bsc_loader_release(bsc_loader* loader);

int bsc_event_create(void** event);
int bsc_event_destroy(void* event);
int bsc_event_record(bsc_ctx* ctx, void* event);
int bsc_event_elapsed_ms(void* start, void* stop, float* host_ms); /* syncs on stop */

/* ---- reparameterisation sampler (ABSENT in reference; README.md:51) ----- */

/* Philox4x32-10 keyed standard normals, float64:
 *   eps[s*n_params + d], counter=(d/4, s, stream, step), key=(seed lo, hi). */
int bsc_philox_normal(bsc_ctx* ctx, uint64_t seed, uint32_t stream, uint32_t step,
                      int32_t n_samples, int32_t n_params, double* eps);

/* Bayesian linear regression q(w)=N(m,diag e^{2rho}), q(xi)=N(a,e^{2b});
 * lam=[m(D),rho(D),a,b] float64.  Writes eps[S,(D+1)] f64, W[S,D] f32
 * (w_s = m + e^rho * eps_s, rounded), xi[S] f64. */
int bsc_blr_sample(bsc_ctx* ctx, const double* lam, int32_t D, int32_t S,
                   uint64_t seed, uint32_t step, double* eps, float* W, double* xi);

/* The same noise for n_steps consecutive Philox steps [step0, step0+n_steps) in
 * one launch: eps[(k*S + s)*(D+1) + d].  The noise does not depend on the
 * parameters, so a driver produces it ahead of time (off the update's latency
 * chain) and hands it to bsc_blr_fused_update with eps_next_ready = 1. */
int bsc_blr_noise(bsc_ctx* ctx, int32_t D, int32_t S, uint64_t seed, uint32_t step0,
                  int32_t n_steps, double* eps);

/* ---- the mini-batch data pass (ABSENT in reference; README.md:51,69-79;
 *      likelihood decomposition bayesic/distribution/base.py:47-69) --------
 *
 * One streaming pass over X[B,D] (row-major, leading dimension ldx floats)
 * and y[B]:
 *      r[n,s] = y[n] - sum_d X[n,d] W[s,d]
 *      Q[s]   = sum_n r[n,s]^2            (float64 out)
 *      G[s,d] = sum_n r[n,s] X[n,d]       (float64 out, [S,D])
 * Requires D % 4 == 0, D <= 256, 1 <= S <= 64, X 16-byte aligned, ldx % 4 == 0.
 * Deterministic: fixed partition, fixed-order float64 finish.
 * S > 8: X is read once per SIXTEEN draws at D == 256 (both contractions on the MFMA pipe:
 * 1M x 256, S = 16 in 202-232 us, S = 64 in 751-775 us), once per eight otherwise. */
int bsc_blr_data_pass(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y,
                      int64_t B, int32_t D, const float* W, int32_t S,
                      double* Q, double* G);

/* Same pass, but leaves the per-workgroup float32 partials in the context
 * workspace for bsc_blr_fused_update(stats = NULL) to finish -- saves the
 * separate reduction launch on the single-GPU path.  Requires S <= 8.  The
 * partials stay valid until the next call that uses the workspace. */
int bsc_blr_data_pass_partial(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y,
                              int64_t B, int32_t D, const float* W, int32_t S);

/* The same two passes with the SWEEP ORDER chosen by the caller -- for a mini-batch that is
 * read again by the next pass (the following update of a resident batch; the next group of 8
 * draws when S > 8).  MI355X keeps the last ~256 MiB it has read in the Infinity Cache unless a
 * load says otherwise, so passes that alternate BSC_SWEEP_FORWARD_KEEP / BSC_SWEEP_BACKWARD_KEEP
 * start in the rows the previous pass ended with: 0.25 of a 1M x 256 mini-batch is then served
 * on-die (164 -> 158 us per pass), all of a shard below 256 MiB.  BSC_SWEEP_STREAM is what the
 * plain entry points do: forward, every load non-temporal, nothing left behind -- right for a batch
 * that is read once.  The statistics are the same sums taken in another order (float32 block
 * partials, so the last bits differ between orders; each order is deterministic).  When S > 8 the
 * sample groups alternate, starting from `sweep`.  D == 256 only; other widths always stream. */
#define BSC_SWEEP_STREAM 0
#define BSC_SWEEP_FORWARD_KEEP 1
#define BSC_SWEEP_BACKWARD_KEEP 2
int bsc_blr_data_pass_sweep(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y,
                            int64_t B, int32_t D, const float* W, int32_t S,
                            double* Q, double* G, int32_t sweep);
int bsc_blr_data_pass_partial_sweep(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y,
                                    int64_t B, int32_t D, const float* W, int32_t S,
                                    int32_t sweep);

/* Measurement aid (option "blr_stamps" = 1): the D = 256, S <= 8 pass leaves per workgroup
 * {start, end} in s_memrealtime ticks (100 MHz), the XCD it ran on and its HW_ID; this copies the
 * stamps of the LAST such launch to host_stamps[8 * rows] (synchronises; entries 4..7: the folded finish's hand-off, or with
 * option "blr_steal" when wave 0 left its static share / its queue and how many queued tiles it took).  Shows how evenly a static
 * partition of the mini-batch finishes (tools/ab_q.py). */
int bsc_blr_read_stamps(bsc_ctx* ctx, uint64_t* host_stamps, int32_t capacity_rows, int32_t* host_rows);

/* How many launches of the pass kernel bsc_blr_data_pass[_sweep] makes for S draws (eight per pass, or
 * sixteen while more than eight are left at D = 256): what a caller that alternates sweep directions
 * per update needs to know, answered by the library that decides it. */
int bsc_blr_pass_count(bsc_ctx* ctx, const float* y, int32_t D, int32_t S, int32_t* count);

/* Monte-Carlo ELBO and pathwise gradient from the (all-reduced) pass outputs.
 * batch_rows = global mini-batch rows, scale = N_total / batch_rows.
 * Writes elbo[1], grad[2D+2] (float64). */
int bsc_blr_elbo_grad(bsc_ctx* ctx, const double* lam, const double* eps,
                      const float* W, const double* xi, const double* Q,
                      const double* G, int32_t D, int32_t S, double batch_rows,
                      double scale, double alpha0, double beta0, double* elbo,
                      double* grad);

/* Fused finish of one update, one launch: (float64 reduction of the pending
 * partials when stats == NULL, else stats = [Q (S) | G (S*D)] as all-reduced)
 * -> ELBO + pathwise gradient (as bsc_blr_elbo_grad) -> Adam ascent step t
 * (as bsc_adam_ascent) written to lam_out (lam_in is not modified; m1, m2
 * updated in place) -> when the *_next buffers are non-NULL, the reparameterised
 * draws of Philox step `next_step` from lam_out (as bsc_blr_sample); with
 * eps_next_ready != 0 eps_next is an INPUT already holding that step's noise
 * (bsc_blr_noise) and only W_next / xi_next are written.  The caller
 * double-buffers lam and the draws: *_next must not alias eps/W/xi. */
int bsc_blr_fused_update(bsc_ctx* ctx, const double* stats, const double* lam_in,
                         double* lam_out, double* m1, double* m2, const double* eps,
                         const float* W, const double* xi, int32_t D, int32_t S,
                         double batch_rows, double scale, double alpha0, double beta0,
                         int64_t t, double lr, double beta1, double beta2, double adam_eps,
                         uint64_t seed, uint32_t next_step, double* eps_next,
                         int32_t eps_next_ready, float* W_next, double* xi_next, double* elbo,
                         double* grad);

/* The same fused finish for any log-joint per draw of the family
 *     f(w, xi; Q) = c0 + c_xi xi + e^{-xi} (-s_q Q / 2 - k_w |w|^2 / 2 - beta),   Q = sum_n (y_n - x_n.w)^2,
 *     d f / d w = e^{-xi} (s_q G - k_w w),   d f / d xi = c_xi + e^{-xi} (s_q Q / 2 + k_w |w|^2 / 2 + beta)
 * -- a Gaussian-linear likelihood with log noise variance xi, a zero-mean Gaussian prior on w of precision
 * k_w e^{-xi} and a log-variance prior linear in xi and e^{-xi} (an inverse-Gamma on the variance, a Gamma
 * on the precision, flat).  bsc_blr_fused_update is the member c0 = -(scale B + D)/2 log 2 pi + alpha0 log
 * beta0 - lnGamma(alpha0), c_xi = -(scale B + D)/2 - alpha0, s_q = scale, k_w = 1, beta = beta0.  This is what
 * bayesic_amd.inference.ReparamVI launches when it recognises such a model written with Distribution nodes
 * and bayesic.algebra expressions (bayesic/distribution/base.py:47-69 decomposition; coefficient extraction
 * with match, bayesic/algebra.py:1037-1063). */
/* ONE launch per update (round 4): the data pass of bsc_blr_data_pass_partial_sweep with the finish of
 * bsc_blr_fused_update[_general] (stats = NULL) carried in its tail -- after its block partial a workgroup
 * takes an arrival ticket, and the last (D + 7) / 8 + 1 arrivals do what the finish kernel's workgroups do once
 * every partial has been written (csrc/bsc_blr.hip FoldArgs: write-through partials, one arrival counter,
 * sc1 reads; a role's arithmetic is fixed whichever workgroup performs it, so the update stays reproducible).
 * Taken when D = 256, S <= 8, the mini-batch gives the grid >= 66 workgroups AND the option "blr_fold" is 1.
 * Its default is 0: measured on MI355X the fold is break-even with the two launches (the hand-off takes ~1 us,
 * but 33 role workgroups of four waves take 4-8 us over the slab where the finish kernel's sixteen-wave
 * workgroups take 6 us behind a 1.7 us boundary; profiles/r04_fold_timeline.txt).  Without the option, and for
 * any other shape, the entry point issues the two launches, with the same results up to the order of the float64 sum over
 * the partials.  bsc_blr_data_pass[_sweep] folds its float64 reduction the same way (the N > 1 structure:
 * pass -> all-reduce -> finish is then two launches and the collective).  Arguments: those of the two entry
 * points it replaces.  README.md:51, 69-79. */
int bsc_blr_pass_update(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B, int32_t D, int32_t sweep,
                        const double* lam_in, double* lam_out, double* m1, double* m2, const double* eps, const float* W,
                        const double* xi, int32_t S, double batch_rows, double scale, double alpha0, double beta0,
                        int64_t t, double lr, double beta1, double beta2, double adam_eps, uint64_t seed,
                        uint32_t next_step, double* eps_next, int32_t eps_next_ready, float* W_next, double* xi_next,
                        double* elbo, double* grad);
int bsc_blr_pass_update_general(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y, int64_t B, int32_t D,
                                int32_t sweep, const double* lam_in, double* lam_out, double* m1, double* m2,
                                const double* eps, const float* W, const double* xi, int32_t S, double c0, double c_xi,
                                double s_q, double k_w, double beta, int64_t t, double lr, double beta1, double beta2,
                                double adam_eps, uint64_t seed, uint32_t next_step, double* eps_next,
                                int32_t eps_next_ready, float* W_next, double* xi_next, double* elbo, double* grad);
int bsc_blr_fused_update_general(bsc_ctx* ctx, const double* stats, const double* lam_in, double* lam_out,
                                 double* m1, double* m2, const double* eps, const float* W, const double* xi,
                                 int32_t D, int32_t S, double c0, double c_xi, double s_q, double k_w, double beta,
                                 int64_t t, double lr, double beta1, double beta2, double adam_eps, uint64_t seed,
                                 uint32_t next_step, double* eps_next, int32_t eps_next_ready, float* W_next,
                                 double* xi_next, double* elbo, double* grad);

/* ---- parameter updates --------------------------------------------------- */

/* Adam ascent on a flat float64 vector; t is the 1-based step count. */
int bsc_adam_ascent(bsc_ctx* ctx, double* lam, const double* grad, double* m1,
                    double* m2, int64_t n, int64_t t, double lr, double beta1,
                    double beta2, double eps);

/* SVI natural-gradient / VMP step (README.md:36,75-77; Hoffman et al. [4]):
 *   eta <- (1-rho) eta + rho (eta0 + scale * message),  all float64[n]. */
int bsc_natgrad_update(bsc_ctx* ctx, double* eta, const double* eta0,
                       const double* message, int64_t n, double scale, double rho);

/* Dirichlet expectation out[r,c] = exp(digamma(lam[r,c]) - digamma(sum_c lam[r,c]))
 * (float32 in/out, float64 inside) and the float32 form of the natural-gradient
 * step with a scalar prior, eta <- (1-rho) eta + rho (eta0 + scale * message):
 * the K x V global parameter of the LDA-style Dirichlet-Multinomial model
 * (BASELINE config 4; its sufficient statistics are the two contractions
 * Bt * dot(Th.T, C / dot(Th, Bt)) evaluated through bsc_gemm_strided_batched). */
int bsc_dirichlet_expectation(bsc_ctx* ctx, const float* lam, int64_t rows, int64_t cols,
                              int64_t ld, float* out);
int bsc_natgrad_update_f32(bsc_ctx* ctx, float* eta, float eta0, const float* message, int64_t n,
                           float scale, float rho);

/* Expectation of a Categorical node: out[r, :] = softmax(in[r, :]) over the LAST axis and, when lse
 * is not NULL, lse[r] = log sum_c exp in[r, c] (the responsibilities of a mixture's discrete latent
 * and the bound's normaliser; README.md:43 -- in the derived mean-field engine the data-sized
 * assignments stay on the device).  float32, rows ld_in / ld_out elements apart, cols <= 1024. */
int bsc_softmax_rows(bsc_ctx* ctx, const float* in, int64_t rows, int64_t cols, int64_t ld_in, float* out,
                     int64_t ld_out, float* lse);

/* The same expectation when the logits are a tall-skinny product that nothing else needs:
 *   L = alpha * A . B,  R = softmax_rows(L),  lse[r] = log sum_c exp L[r, c],  cross[r] = sum_c R[r, c] L[r, c]
 * with A [rows, K] (unit column stride, rows lda apart), B [K, N] (strides ldbk, ldbn), R [rows, N]
 * (rows ldr apart) -- features(x_n) . coefficients of a mixture with exponential-family components
 * (README.md:43; alpha = the 1 / (N / B) of a local latent under mini-batch scaling, README.md:69-79).  One pass: A is read once and R written once; the logits never reach memory (as two
 * launches they are written, read and written again).  `cross` (may be NULL) is what the factor's
 * entropy sum_r (lse[r] - cross[r]) needs in place of the logits.  float32; K <= 64 and a multiple of
 * 8, N <= 64 and a multiple of 4, A and R 16-byte aligned, lda % 4 == ldr % 4 == 0; anything else is
 * BSC_ERR_UNSUPPORTED (callers then take bsc_gemm_strided_batched + bsc_softmax_rows). */
int bsc_gemm_softmax_rows(bsc_ctx* ctx, const float* A, int64_t lda, int64_t rows, int32_t K,
                          const float* B, int64_t ldbk, int64_t ldbn, int32_t N, float alpha, float* R,
                          int64_t ldr, float* lse, float* cross);

/* ... and the statistics every neighbour of that latent asks for, in the same pass:
 *   stats[c, f] = sum_r R[r, c] A[r, f]          (float32 [N, K], rows ldst apart),   lse_sum[0] = sum_r lse[r]
 * -- R^T . A with A = [T_1(x) | T_2(x) | .. | 1] the feature matrix the logits were formed from: the
 * responsibility-weighted sufficient statistics of a mixture with exponential-family components (statistics
 * of iid draws add up, bayesic/distribution/base.py:329-332; marginalisation by summation README.md:43,72) and,
 * through the ones column, the responsibilities' column sums.  The tile's responsibilities go from the softmax
 * registers through wave-private LDS into the backward MFMAs; R (may be NULL) is written only when something
 * else reads it.  csrc/bsc_mog.hip's E-step for any feature matrix of up to 64 columns; float64 fixed-order
 * finish.  `bias` (may be NULL; N values ldbn apart): L = alpha * (A . B + bias) -- the feature matrix's ones
 * column taken out of the product; its statistic, the column sums sum_r R[r, c], is then written to
 * stats[c, K] (ldst >= K + 1, K <= 56).  Same limits as bsc_gemm_softmax_rows otherwise. */
int bsc_gemm_softmax_stats(bsc_ctx* ctx, const float* A, int64_t lda, int64_t rows, int32_t K,
                           const float* B, int64_t ldbk, int64_t ldbn, int32_t N, float alpha, const float* bias,
                           float* R, int64_t ldr, float* stats, int64_t ldst, double* lse_sum);

/* Fixed-gamma local step of the LDA-style Dirichlet-Multinomial model (BASELINE
 * config 4): sstats[k,v] = Bt[k,v] * sum_d Th[d,k] C[d,v] / (sum_k' Th[d,k'] Bt[k',v]),
 * the algebra expression Bt * dot(Th.T, C / dot(Th, Bt)) (lowered by
 * bayesic/algebra.py:553-765 to two _tensordot GEMMs around a division) in ONE pass
 * over the count matrix C[docs, V]: neither docs x V intermediate is written.
 * Th[docs, K] = exp(E[log theta]), Bt[K, V] = exp(E[log beta]) (bsc_dirichlet_expectation);
 * all float32, row-major with leading dimensions in elements; fp32 MFMA.
 * K must be 32, 64, 96 or 128 (else BSC_ERR_UNSUPPORTED -- the executor path is general). */
int bsc_lda_sstats(bsc_ctx* ctx, const float* C, int64_t ldc, int64_t docs, int64_t V, int32_t K,
                   const float* Th, int64_t ldth, const float* Bt, int64_t ldb, float* sstats,
                   int64_t ldo);

/* The same statistic from SPARSE counts in compressed-sparse-column form (column v =
 * word v: colptr[V+1] offsets into rowidx (document ids) / vals (counts)): one pass over
 * the nonzeros, no atomics, fixed summation order.  Real bag-of-words data is ~1 % dense
 * (SURVEY.md 8(f) rank 4); explicit zeros contribute nothing.  Same K limits. */
int bsc_lda_sstats_csc(bsc_ctx* ctx, const int64_t* colptr, const int32_t* rowidx, const float* vals,
                       int64_t docs, int64_t V, int32_t K, const float* Th, int64_t ldth,
                       const float* Bt, int64_t ldb, float* sstats, int64_t ldo);

/* The evidence lower bound of the same model (ABSENT in the reference; README.md:30-37, mini-batch
 * form README.md:69-79; Hoffman, Blei, Bach 2010 eq. 7 with the per-word assignments at their
 * optimum; oracle.svi.lda_elbo):
 *   ELBO = scale * [ sum_dv C_dv log(phinorm_dv) - sum_d KL(Dir(gamma_d) || Dir(alpha)) ]
 *          - sum_k KL(Dir(lambda_k) || Dir(eta)),          phinorm = Th Bt.
 * bsc_lda_sstats_bound / bsc_lda_sstats_csc_bound are the statistic kernels above that ALSO return
 * ll[0] = sum_dv C_dv log(phinorm_dv) (float64, device): phinorm exists only in their registers, so
 * the sum is taken there -- a v_log_f32 and an fma per element, float32 per-wave partials, added in
 * float64 in a fixed order; no extra pass over C.
 * bsc_dirichlet_expectation_bound is bsc_dirichlet_expectation that also returns
 * bound[0] = sum_r -KL(Dir(lam_r) || Dir(prior)) (symmetric scalar prior; float64, fixed order), from
 * the digamma values it computes anyway plus one lnGamma per element (both factors' terms: rows =
 * topics for lambda, rows = documents for gamma).
 * bsc_natgrad_update_f32_elbo is bsc_natgrad_update_f32 that first writes
 * elbo[0] = scale * (ll[0] + local_bound[0]) + global_bound[0]: the bound at the lambda the step
 * starts from (ll and local_bound all-reduced when data-parallel). */
int bsc_lda_sstats_bound(bsc_ctx* ctx, const float* C, int64_t ldc, int64_t docs, int64_t V, int32_t K,
                         const float* Th, int64_t ldth, const float* Bt, int64_t ldb, float* sstats,
                         int64_t ldo, double* ll);
int bsc_lda_sstats_csc_bound(bsc_ctx* ctx, const int64_t* colptr, const int32_t* rowidx, const float* vals,
                             int64_t docs, int64_t V, int32_t K, const float* Th, int64_t ldth,
                             const float* Bt, int64_t ldb, float* sstats, int64_t ldo, double* ll);
int bsc_dirichlet_expectation_bound(bsc_ctx* ctx, const float* lam, int64_t rows, int64_t cols, int64_t ld,
                                    double prior, float* out, double* bound);
int bsc_natgrad_update_f32_elbo(bsc_ctx* ctx, float* eta, float eta0, const float* message, int64_t n,
                                float scale, float rho, const double* ll, const double* local_bound,
                                const double* global_bound, double* elbo);
/* The same step on a [rows, cols] block: eta with leading dimension ld_eta, the message with ld_msg -- the
 * piece of lambda [K, V] that one column range of the statistic, staged contiguously for its collective,
 * belongs to.  elbo != NULL: elbo[0] = scale * (ll[0] + .. + ll[n_ll - 1] + local_bound[0]) + global_bound[0]
 * first (one words' term per piece).  The same arithmetic per element as bsc_natgrad_update_f32.
 * bsc_lda_sstats_round_columns: the columns of ONE whole round of the persistent statistic kernel on this
 * context (128 per resident workgroup); a statistic taken in ranges that start at multiples of it is
 * bit-identical to the statistic taken in one call (every column block then has the schedule it has there). */
int bsc_natgrad_update_f32_2d(bsc_ctx* ctx, float* eta, int64_t ld_eta, float eta0, const float* message,
                              int64_t ld_msg, int64_t rows, int64_t cols, float scale, float rho, const double* ll,
                              int32_t n_ll, const double* local_bound, const double* global_bound, double* elbo);
int bsc_lda_sstats_round_columns(bsc_ctx* ctx, int32_t K, int64_t* host_cols);

/* ---- summed sufficient statistics of iid draws ---------------------------
 * ExpFamIndependentObservations.sufficient_statistics,
 * bayesic/distribution/base.py:328-332; Normal t(x)=(x,x^2),
 * bayesic/distribution/core.py:16-17.
 * stats[0]=n, stats[1]=sum x, stats[2]=sum x^2 (float64). */
int bsc_suffstats_normal(bsc_ctx* ctx, const float* x, int64_t n, double* stats);

/* ---- mixture of Gaussians: discrete latent marginalised by summation ------
 * (ABSENT in reference; README.md:43,72; statistics per
 * bayesic/distribution/base.py:329-332).  For every row of X[N,D]:
 *     logit_k = c[k] + sum_d (Wmat[k,d] x_d + Wmat[k,D+d] x_d^2)
 *     r_k     = softmax_k(logit)
 * and, summed over rows in float64 (fixed order):
 *     stats[k] = [sum r_k | sum r_k x (D) | sum r_k x^2 (D)]   ([K, 1+2D])
 *     lse[0]   = sum_n logsumexp_k(logit_nk)
 * Both contractions run on fp32 MFMA.  Limits: D <= 16, K <= 64 (one tile). */
int bsc_mog_estep(bsc_ctx* ctx, const float* X, int64_t ldx, int64_t N, int32_t D, int32_t K,
                  const float* Wmat, const float* c, double* stats, double* lse);

/* Responsibility-weighted second moments (full-covariance mixture statistic
 * sum_n r_nk x_n x_n^T; t(x) = (x, x x^T) of bayesic/distribution/core.py:41-44 summed over
 * the iid axis, bayesic/distribution/base.py:329-332):
 *     out[k,d,e] = scale * sum_n R[n,k] X[n,d] Y[n,e]          (float32 [K, D, E], contiguous)
 * in ONE pass over R[N,K], X[N,D], Y[N,E] (row-major, leading dimensions in floats).  The
 * reference lowers this einsum to a _tensordot over a materialised K x D x N product
 * (bayesic/algebra.py:632-636).  Y == X (same pointer, ld and extent) computes only d <= e and
 * mirrors.  fp32 MFMA, float64 fixed-order finish.  Limits: K <= 64, D, E <= 32, all
 * multiples of 4, 16-byte aligned operands, ld % 4 == 0 (else BSC_ERR_UNSUPPORTED -- the
 * executor then takes the general route). */
int bsc_weighted_outer(bsc_ctx* ctx, const float* R, int64_t ldr, const float* X, int64_t ldx,
                       const float* Y, int64_t ldy, int64_t N, int32_t K, int32_t D, int32_t E,
                       double scale, float* out);

/* Mean-field global parameters of the mixture: Dirichlet over weights and a
 * Normal-Gamma per (component, column), as one natural-parameter vector
 *   eta = [alpha-1 (K) | kappa*m (K*D) | kappa (K*D) | 2a-1 (K*D) | 2b+kappa*m^2 (K*D)].
 * bsc_mog_expected_params writes the float32 logit coefficients Wmat [K, 2D] and
 * c [K] (expectations under q; digamma in float64).  bsc_mog_natgrad applies
 *   eta <- (1-rho) eta + rho (eta0 + scale * message(stats))
 * (README.md:36,75-77) straight from the [K, 1+2D] statistics. */
int bsc_mog_expected_params(bsc_ctx* ctx, const double* eta, int32_t K, int32_t D, float* Wmat,
                            float* c);
int bsc_mog_natgrad(bsc_ctx* ctx, double* eta, const double* eta0, const double* stats, int32_t K,
                    int32_t D, double scale, double rho);

/* The evidence lower bound of the mixture (ABSENT in the reference; README.md:30-37 "maximise a
 * lower bound on the model evidence", mini-batch form README.md:69-79; every factor decomposed as
 * bayesic/distribution/base.py:47-69).  With the assignments marginalised by summation
 * (README.md:43,72) the bound at q(theta) = eta is
 *     ELBO = scale * sum_n logsumexp_k(logit_nk) + bound,
 *     bound = E_q[log p(pi,mu,tau)] - E_q[log q(pi,mu,tau)] = sum_factors <eta0-eta, E_q[T]> - A(eta0) + A(eta)
 * (oracle.svi.mog_elbo / mog_global_bound).  bsc_mog_log_normalizer writes A(eta) (one float64, device):
 * called once with the prior's eta0, it gives the constant `prior_A` of the model.
 * bsc_mog_expected_params_bound is bsc_mog_expected_params that also writes `bound` (one float64,
 * device) from the same digamma / log values; bsc_mog_natgrad_elbo is bsc_mog_natgrad that first writes
 * elbo[0] = scale * lse[0] + bound[0] (lse: what bsc_mog_estep returned for THIS eta, all-reduced
 * when data-parallel) -- the bound at the parameters the step starts from.  No extra launch, no
 * extra pass.  K <= 1024. */
int bsc_mog_log_normalizer(bsc_ctx* ctx, const double* eta, int32_t K, int32_t D, double* A_out);
int bsc_mog_expected_params_bound(bsc_ctx* ctx, const double* eta, const double* eta0, const double* prior_A,
                                  int32_t K, int32_t D, float* Wmat, float* c, double* bound);
int bsc_mog_natgrad_elbo(bsc_ctx* ctx, double* eta, const double* eta0, const double* stats, int32_t K,
                         int32_t D, double scale, double rho, const double* lse, const double* bound,
                         double* elbo);

/* ---- black-box VI, score-function gradient with control variate ------------
 * (ABSENT in reference; README.md:52 -> ref [3]; config 5: hierarchical logistic
 * regression y_n ~ Bernoulli(sigmoid(x_n.w + b_{g_n})), z = [w (D) | b (G) | log tau],
 * q(z) = N(mu, diag e^{2 rho}), lam = [mu (P) | rho (P)], P = D+G+1.)
 *
 * bsc_bbvi_sample: z_s = mu + e^rho eps_s (Philox stream 2) -> eps [S,P] f64,
 *   Wz [S,D] f32, Bz [G,S] f32 (transposed), zeta [S] f64.
 * bsc_logreg_bbvi_loglik: ONE pass over X[N,D], y[N] (0/1 as float), g[N] (int32):
 *   ell[s] = sum_n ( y_n l_ns - softplus(l_ns) ),  l_ns = x_n.Wz[s] + Bz[g_n,s]
 *   (float64 out, fixed order).  fp32 MFMA.  Requires S == 64, D % 4 == 0, D <= 256.
 * bsc_bbvi_grad: f_s = scale*ell_s + log p(z_s) - log q(z_s); control variate
 *   a = sum_i Cov(f h_i, h_i) / sum_i Var(h_i); grad = mean_s (f_s - a) h_s with
 *   h_s = [eps/sigma | eps^2 - 1]; elbo = mean_s f_s.  f_out (may be NULL): f [S]. */
int bsc_bbvi_sample(bsc_ctx* ctx, const double* lam, int32_t D, int32_t G, int32_t S, uint64_t seed,
                    uint32_t step, double* eps, float* Wz, float* Bz, double* zeta);
int bsc_logreg_bbvi_loglik(bsc_ctx* ctx, const float* X, int64_t ldx, const float* y,
                           const int32_t* g, int64_t N, int32_t D, int32_t n_groups,
                           const float* Wz, const float* Bz, int32_t S, double* ell);
int bsc_bbvi_grad(bsc_ctx* ctx, const double* lam, const double* eps, const double* ell, int32_t D,
                  int32_t G, int32_t S, double scale, double a0, double b0, double* elbo,
                  double* grad, double* f_out);
/* bsc_bbvi_update: everything after the pass in three launches -- bsc_bbvi_grad's f and
 * control-variate moments, then ONE kernel for the gradient, the Adam ascent step of
 * bsc_adam_ascent (step count t >= 1, in place on lam, m1, m2) and the draws of the NEXT update
 * (bsc_bbvi_sample with `next_step`, from the new lam: eps is overwritten, Wz / Bz / zeta
 * refreshed).  Same results as the three separate calls up to float64 summation order. */
int bsc_bbvi_update(bsc_ctx* ctx, double* lam, double* eps, const double* ell, int32_t D, int32_t G,
                    int32_t S, double scale, double a0, double b0, double* m1, double* m2, int64_t t,
                    double lr, double beta1, double beta2, double eps_adam, uint64_t seed,
                    uint32_t next_step, float* Wz, float* Bz, double* zeta, double* elbo,
                    double* grad, double* f_out);

/* ---- executable primitives of the algebra front end -----------------------
 * The five-op IR that Einsum lowering emits plus element-wise nodes:
 * _sum bayesic/algebra.py:1284-1294, _mul :1297-1309, _dimshuffle :1312-1326 and
 * _diagonal :1398-1414 (both are stride views -- no kernel), _tensordot
 * :1329-1383, elemwise/add :195-233, log/exp/pow/abs_ :1435-1448, eye :236-258.
 * Tensors are (pointer, shape, strides-in-elements); a stride of 0 broadcasts a
 * size-1 axis (the 'x' axes of dimshuffle).  rank <= BSC_MAX_RANK. */

typedef enum bsc_dtype { BSC_F32 = 0, BSC_F64 = 1 } bsc_dtype;

typedef enum bsc_op {
    BSC_OP_ADD = 0, /* n-ary, 1..8 inputs */
    BSC_OP_MUL = 1, /* n-ary, 1..8 inputs */
    BSC_OP_LOG = 2,
    BSC_OP_EXP = 3,
    BSC_OP_POW = 4, /* in0 ** in1 */
    BSC_OP_ABS = 5,
    BSC_OP_COPY = 6, /* materialise a strided view */
    BSC_OP_LGAMMA = 7,  /* log Gamma(x): unary, bsc_map_reduce only (log-normalisers of the */
    BSC_OP_DIGAMMA = 8, /* Gamma / Dirichlet / Wishart nodes and their expectations)        */
    BSC_OP_SCALE = 9    /* x * arg: unary, bsc_map_reduce only -- a per-operand coefficient, so that
                         * (1 - rho) eta + rho m (the damped natural-gradient step, README.md:75-77) is one launch */
} bsc_op;

/* out[i] = op(in_0[i], ..., in_{n-1}[i]) over the index space `shape`;
 * in_strides is [n_in][rank] row-major; `in` is a HOST array of n_in device
 * pointers. */
int bsc_elemwise(bsc_ctx* ctx, int op, int dtype, int rank, const int64_t* host_shape, void* out,
                 const int64_t* host_out_strides, int n_in, const void* const* host_in,
                 const int64_t* host_in_strides);

/* dst = (dst_dtype) src, both strided (dtype conversion + layout change). */
int bsc_convert(bsc_ctx* ctx, int src_dtype, int dst_dtype, int rank, const int64_t* host_shape,
                const void* src, const int64_t* host_src_strides, void* dst,
                const int64_t* host_dst_strides);

/* out[keep...] = sum over red... of in[keep..., red...]; float64 accumulation,
 * fixed summation order (deterministic).  out is contiguous over keep_shape. */
int bsc_sum(bsc_ctx* ctx, int dtype, int rank_keep, const int64_t* host_keep_shape,
            const int64_t* host_in_keep_strides, int rank_red, const int64_t* host_red_shape,
            const int64_t* host_in_red_strides, const void* in, void* out);

/* Fused pointwise map with an optional trailing sum: one pass over the operands
 * for a chain elemwise -> add/_mul -> elemwise -> _sum (bayesic/algebra.py:195-233,
 * 1284-1309, 1435-1448), which the reference leaves to Theano's graph optimiser.
 *   v(keep, red) = post( scale * COMBINE_i pre_i( in_i[keep, red] ) + shift )
 *   out[keep]    = sum over red of v(keep, red)     (rank_red == 0: out[keep] = v(keep))
 * combine is BSC_OP_ADD or BSC_OP_MUL; pre_op[i] and post_op are BSC_OP_COPY, SCALE, LOG,
 * EXP, ABS, LGAMMA, DIGAMMA or POW (x ** arg, arg taken from pre_arg[i] / post_arg).  in_keep_strides
 * is [n_in][rank_keep], in_red_strides [n_in][rank_red] (0 broadcasts); sums
 * accumulate in float64 in a fixed order. */
int bsc_map_reduce(bsc_ctx* ctx, int dtype, int combine, int rank_keep,
                   const int64_t* host_keep_shape, int rank_red, const int64_t* host_red_shape,
                   int n_in, const void* const* host_in, const int64_t* host_in_keep_strides,
                   const int64_t* host_in_red_strides, const int32_t* host_pre_op,
                   const double* host_pre_arg, double scale, double shift, int post_op,
                   double post_arg, void* out, const int64_t* host_out_strides);

/* C[b,m,n] = sum_k A[b,m,k] * B[b,k,n], every stride free (so transposed and
 * broadcast operands cost nothing).  float32 runs on v_mfma_f32_32x32x2_f32 with
 * a deterministic split-K when M*N is small against K (XtX: M=N=256, K=1e6);
 * float64 is a plain VALU kernel. */
int bsc_gemm_strided_batched(bsc_ctx* ctx, int dtype, int64_t batch, int64_t M, int64_t N,
                             int64_t K, const void* A, int64_t sa_b, int64_t sa_m, int64_t sa_k,
                             const void* B, int64_t sb_b, int64_t sb_k, int64_t sb_n, void* C,
                             int64_t sc_b, int64_t sc_m, int64_t sc_n);

/* The same product with its consumer folded into the store (what Theano's graph optimiser would do
 * for `B * dot(X, Y)` and `C / dot(X, Y)`, bayesic/algebra.py:741-765 + 1297-1309):
 *     C[b,m,n] = scale * (sum_k A[b,m,k] B[b,k,n]) ^ power * E[b,m,n],   power = 1 | -1,
 * E strided like C (a stride of 0 broadcasts; E == NULL: no factor).  float32, K > 0. */
int bsc_gemm_epilogue(bsc_ctx* ctx, int dtype, int64_t batch, int64_t M, int64_t N, int64_t K, const void* A,
                      int64_t sa_b, int64_t sa_m, int64_t sa_k, const void* B, int64_t sb_b, int64_t sb_k,
                      int64_t sb_n, void* C, int64_t sc_b, int64_t sc_m, int64_t sc_n, int power, double scale,
                      const void* E, int64_t se_b, int64_t se_m, int64_t se_n);
/* bsc_gemm_epilogue with an element-wise PRODUCER folded into the operand fragments as well:
 *   C = scale * dot(pre_a(A), pre_b(B))^power * E,   pre = 0 none | 1 square | 2 exp | 3 abs   (power 0: no epilogue)
 * -- `_mul` / `elemwise` feeding `_tensordot` (bayesic/algebra.py:741-765, 1297-1309 -> :1347) without the
 * intermediate array: dot(exp(X), Y), dot(X * X, A.T).  Only the persistent stream kernel applies prologues; when the
 * shape takes another path (matrix-vector, skinny, float64, short contractions), or both sides ask for exp (padded
 * zeros would no longer cancel), NOTHING is launched and *handled = 0: the caller materialises the operand and calls
 * bsc_gemm_strided_batched / bsc_gemm_epilogue.  *handled = 1: C is written. */
int bsc_gemm_fused(bsc_ctx* ctx, int dtype, int64_t batch, int64_t M, int64_t N, int64_t K, const void* A, int64_t sa_b,
                   int64_t sa_m, int64_t sa_k, int pre_a, const void* B, int64_t sb_b, int64_t sb_k, int64_t sb_n, int pre_b,
                   void* C, int64_t sc_b, int64_t sc_m, int64_t sc_n, int power, double scale, const void* E, int64_t se_b,
                   int64_t se_m, int64_t se_n, int32_t* handled);

/* Host-only: the schedule the persistent kernels (GEMM, LDA statistic) use for `tiles` tiles of `n_kt`
 * units on `slots` resident workgroups; out = {n_wg, rounds, tail_tiles, sk_stream, sk_q, sk_r} --
 * workgroup w takes tiles w, w + n_wg, ... of `rounds` rounds, then units [u(w), u(w + 1)) of the
 * tail tiles' unit list, u(w) = sk_stream ? w sk_q + min(w, sk_r) : min(w, tail_tiles) n_kt. */
int bsc_stream_plan(int64_t tiles, int32_t n_kt, int64_t slots, int32_t out[6]);

/* out[b] = log det A[b] for symmetric positive-definite A[b] (n x n, strides in
 * elements), by float64 Cholesky -- the T.logdet of MultivariateNormal's
 * log-normaliser, bayesic/distribution/core.py:49-52.  NaN when not SPD. */
int bsc_logdet_spd(bsc_ctx* ctx, int dtype, int64_t batch, int64_t n, const void* A, int64_t s_b,
                   int64_t s_r, int64_t s_c, void* out);

/* out[b] (n x n, contiguous) = the inverse of the symmetric positive-definite A[b] (strides in elements; the mean of
 * the two triangles is taken), by float64 Cholesky; logdet[b] = log det A[b] when logdet is not null.  What turns the
 * natural parameters of a multivariate-normal or Wishart factor (precision, V^-1) into its expectations on the
 * device: the variational-message-passing update of README.md:36 for the (t(x) = (x, x x^T)) family of
 * bayesic/distribution/core.py:41-56, whose host-side form calls numpy.linalg.inv.  NaN when not SPD. */
int bsc_inverse_spd(bsc_ctx* ctx, int dtype, int64_t batch, int64_t n, const void* A, int64_t s_b, int64_t s_r,
                    int64_t s_c, void* out, void* logdet);

/* out[n,n] = identity, contiguous. */
int bsc_eye(bsc_ctx* ctx, int dtype, void* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* BAYESIC_HIP_H */
