/*
 * qpnet_hip.h -- C ABI of libqpnet_hip.so: the MI355X (gfx950) QPNet hot path.
 *
 * The reference (bigpon/QPNet) has no native boundary: its hot path is the Python module
 * src/nets/qpnet.py calling stock torch ops.  The boundary that module would bind if its
 * hot path were native is defined here; each entry point cites the
 * reference interface it replaces.  The host-side mirror of the reference's Python surface
 * (class QPNet etc.) lives in qpnet_amd/qpnet.py and calls ONLY these functions.
 *
 * Conventions
 *   - plain C types; every `d_*` pointer is a DEVICE pointer owned by the caller
 *     (e.g. torch tensors' data_ptr()); `h_*` pointers are host memory.
 *   - return 0 on success, a negative QPN_E* code otherwise; nothing throws across the ABI.
 *     qpn_last_error() returns a static, human-readable description of the last failure.
 *   - the library owns only its handle and the workspaces it allocates in it.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).
 *   - a handle is bound to the device current at qpn_create(); not thread-safe per handle.
 *   - there is NO CPU fallback: every compute entry point fails with QPN_ENODEV when no
 *     HIP device is usable.
 */
#ifndef QPNET_HIP_H
#define QPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPN_OK 0
#define QPN_EINVAL (-1)      /* bad argument / unsupported geometry            */
#define QPN_ENODEV (-2)      /* no usable HIP device / HIP runtime error       */
#define QPN_ENOMEM (-3)      /* workspace allocation failed                    */
#define QPN_ERANGE (-4)      /* pitch-dependent gather left its ring (ref: assert qpnet.py:294,417) */
#define QPN_ESTATE (-5)      /* call order (e.g. decode before set_weights)    */

/* Constructor kwargs of QPNet (reference src/nets/qpnet.py:174-178). */
typedef struct {
    int n_quantize, n_aux, n_resch, n_skipch;
    int dilationF_depth, dilationF_repeat;
    int dilationA_depth, dilationA_repeat;
    int kernel_size, upsampling_factor;
} qpn_config;

typedef struct qpn_handle qpn_handle;

/* library / ABI version (major*1000 + minor) */
int qpn_version(void);

/* Last error message of this thread's most recent failing call. */
const char* qpn_last_error(void);

/* Number of fp32 parameters of the geometry = length of the flat parameter vector, which is
 * the concatenation of state_dict() tensors in registration order, each in its native
 * PyTorch layout (reference QPNet.__init__, src/nets/qpnet.py:200-235). */
int64_t qpn_param_count(const qpn_config* cfg);

/* QPNet.__init__ (src/nets/qpnet.py:174-237): validate geometry, build the decode program. */
int qpn_create(const qpn_config* cfg, qpn_handle** out);
void qpn_destroy(qpn_handle* h);

/* load_state_dict / .cuda() (reference bin/qpnet_decode.py:286-293): bind the flat fp32
 * parameter vector (device memory, n == qpn_param_count) and (re)pack the decode tiles.
 * The vector is read again by every later call; call again after the values change.
 * Stream-ordered, no host synchronisation. */
int qpn_set_weights(qpn_handle* h, const float* d_flat, size_t n, void* stream);

/* mode of qpn_decode */
#define QPN_MODE_ARGMAX 0
#define QPN_MODE_SAMPLING 1

/*
 * QPNet.batch_fast_generate (reference src/nets/qpnet.py:314-559), all batch rows at once: persistent kernels that loop
 * over all samples of their utterances, launched once per call.  The launch plan is sized from the device's CU count
 * (hipDeviceAttributeMultiprocessorCount), nothing assumes a whole 256-CU chip:
 *   paper-size geometry   (CUs / 40) * 8 resident groups of five workgroups (decode_pipe.hip), one utterance per group; more
 *                         rows than groups: a group steps two (three) utterances alternately, the shortest rows share
 *                         (rows are assigned longest first); beyond three per group, equal-sized launches;
 *   n_resch > 128         G workgroups per utterance (decode_coop.hip), G * rows <= CUs per launch;
 *   other geometries      one workgroup (one CU) per utterance.
 * Every wait between workgroups is bounded; when a multi-workgroup launch gives up (its workgroups were not co-resident:
 * CU-masked or shared GPU) qpn_decode / qpn_decode_finish re-run the call once with a smaller footprint (one-CU kernels,
 * or half the workgroups per utterance) before returning QPN_ENODEV.
 *   B          batch rows                         n_x   seed samples per row (x is B x n_x)
 *   F          frames of h (h is B x n_aux x F; if upsampling_factor==0, F = samples)
 *   Td         columns of the dilated factors (B x Td), float64 or float32 (d_is_f32)
 *   h_n_samples[B]  samples to generate per row (n_samples_list)
 *   maxd       int(nanmax(ceil(dilated_factors))) over the batch (qpnet.py:347-350)
 *   d_teacher  optional (B x max_n) int64: fed back instead of the pick (teacher forcing)
 *   d_out      (B x max_n) int64 picks, row-major, max_n = max(h_n_samples)
 *   d_logits   optional (B x max_n x n_quantize) fp32 per-step logits
 * Rows are returned in INPUT order; completion-order / list-mutation semantics of the
 * reference are applied by the Python mirror.  Blocks until the stream has finished.
 */
int qpn_decode(qpn_handle* h, int B, int n_x, int64_t F, int64_t Td,
               const int64_t* d_x, const float* d_h, const void* d_dfac, int d_is_f32,
               const int64_t* h_n_samples, int maxd, int mode, uint64_t seed,
               const int64_t* d_teacher, int64_t* d_out, float* d_logits, void* stream);

/* Same, split for stream use: enqueue only enqueues (no host synchronisation: the utterance descriptors are staged in
 * pinned memory owned by the handle) and returns; finish synchronises the stream and returns the device-side status
 * (QPN_ERANGE ...).  One decode in flight per handle: a second enqueue before finish returns QPN_ESTATE; the caller's
 * buffers must stay valid until finish (it may re-run the call, see above). */
int qpn_decode_enqueue(qpn_handle* h, int B, int n_x, int64_t F, int64_t Td,
                       const int64_t* d_x, const float* d_h, const void* d_dfac, int d_is_f32,
                       const int64_t* h_n_samples, int maxd, int mode, uint64_t seed,
                       const int64_t* d_teacher, int64_t* d_out, float* d_logits, void* stream);
int qpn_decode_finish(qpn_handle* h, void* stream);

/* Device time (ms) of the persistent decode kernel of the last finished qpn_decode call,
 * measured with HIP events on the launch stream (bench.py roofline). */
float qpn_last_decode_kernel_ms(qpn_handle* h);

/* The launch plan of the last decode call as text, e.g. "pipe rows=96 waves=1 x 96 (2 per group); one-cu rows=0"
 * (diagnostics / bench.py; owned by the handle, valid until the next decode call). */
const char* qpn_last_decode_plan(qpn_handle* h);

/*
 * QPNet.forward (reference src/nets/qpnet.py:239-312), teacher forced, fused fp32-MFMA kernels.
 *   d_flat   flat fp32 parameters (state_dict order), read at call time (training updates them)
 *   d_x      (B x T) int64 samples         d_h  (B x n_aux x F) fp32 frame-rate features
 *   d_dfac   (B x Td) fp32 dilated factors BL   batch_length (same for every row, qpnet.py:253)
 *   maxd     int(max(ceil(dilated_factors))) over the whole tensor (qpnet.py:255)
 *   d_logits (B x BL x n_quantize) fp32 out.  Activations needed by backward stay in the
 *   handle's workspace until the next qpn_train_forward.  Asynchronous on `stream`.
 */
int qpn_train_forward(qpn_handle* h, const float* d_flat, int B, int64_t T, int64_t F, int64_t Td, int BL, int maxd,
                      const int64_t* d_x, const float* d_h, const float* d_dfac, float* d_logits, void* stream);

/* autograd of the above (reference: loss.backward(), src/bin/qpnet_train.py:529-530):
 * d_dlogits (B x BL x n_quantize) -> d_flatgrad (n_params, overwritten), same layout as d_flat. */
int qpn_train_backward(qpn_handle* h, const float* d_dlogits, float* d_flatgrad, void* stream);

/* Same with the data-parallel hooks of the one gradient exchange per step (replaces the reference's dead DataParallel wrapper,
 * src/bin/qpnet_train.py:416-423): the flat gradient is multiplied by grad_scale (the rank's row count B*BL) inside the
 * reduction kernel; with append_scale != 0, d_flatgrad must hold n_params + 4 floats and receives {grad_scale, flagged, 0, 0} behind
 * the gradient, so ONE all-reduce(SUM) carries sum_r n_r g_r, sum_r n_r and the number of ranks whose device-side status word is set
 * (flagged = 1.0 then: qpn_adam_step_ex skips the update on EVERY rank).  append_scale = 1: the backward's last launch writes the flag (it then covers the
 * backward's own status bits too); 2: the trailer leaves with the early exchange bucket (qpn_train_early_bucket) and carries the forward's bits only. */
int qpn_train_backward_ex(qpn_handle* h, const float* d_dlogits, float* d_flatgrad, float grad_scale, int append_scale, void* stream);

/* Number of qpn_train_forward calls on this handle so far.  The activations backward needs live in the handle's workspace
 * (one outstanding forward per handle): a caller that defers backward (torch autograd) records the value after forward and
 * compares before backward (reference: autograd keeps per-call activations, src/bin/qpnet_train.py:520-530). */
int64_t qpn_train_generation(qpn_handle* h);

/* Synchronise and report the device-side status of the last training call (QPN_ERANGE when a
 * pitch-dependent tap left its layer input: reference assert qpnet.py:294). */
int qpn_train_status(qpn_handle* h, void* stream);

/* The same check without draining the stream: _enqueue copies the status word (sticky: kernels OR into it, only a read
 * clears it) to pinned memory behind the work enqueued so far and returns; _collect waits for that copy alone and
 * reports it.  The module calls _collect at the start of a forward and _enqueue behind it: an out-of-range tap of
 * step i is raised at step i+1 (reference: assert at step i, qpnet.py:294), and no step is serialised for it. */
int qpn_train_status_enqueue(qpn_handle* h, void* stream);
int qpn_train_status_collect(qpn_handle* h);
/* ... every enqueued check EXCEPT the newest one: never waits for the work the device still has queued (a fused training loop calls it
 * at the start of every step and reports a bad chunk two steps late at most; reference: the in-line asserts of qpnet.py:294, qpnet_train.py:525). */
int qpn_train_status_collect_lagged(qpn_handle* h);
/* The same without ever waiting: only the checks whose copies have already landed are reported (hipEventQuery); *pending (may be NULL)
 * receives the number still in flight.  For callers that must not stall the host behind queued work (QPNet.status_check = "lazy"). */
int qpn_train_status_poll(qpn_handle* h, int* pending);

/* torch.nn.CrossEntropyLoss() (mean) on the logits above and, optionally, its gradient
 * (reference src/bin/qpnet_train.py:430,526-528; a target outside [0, n_quantize) is clamped and flagged: qpn_train_status
 * returns QPN_ERANGE, the reference asserts at :525).  d_targets is the (B x tgt_stride) int64 target
 * tensor whose LAST BL columns are used (batch_t[:, -batch_length:]).  h_loss (optional, host)
 * receives the loss (synchronises); d_dlogits (optional) receives dL/dlogits. */
int qpn_ce_loss(qpn_handle* h, const float* d_logits, const int64_t* d_targets, int64_t tgt_stride, int B, int BL,
                float* d_dlogits, double* h_loss, void* stream);

/* Forward + loss in one call: qpn_train_forward followed by qpn_ce_loss (reference src/bin/qpnet_train.py:520-528:
 * `batch_output = model(...)`, `criterion(batch_output[i], batch_t[i])`), with the cross entropy and its gradient computed
 * inside the post-net kernel while a tile's logits are still in LDS (paper-size stacks; wider ones run the separate kernel
 * behind the forward -- same results either way).  d_targets / tgt_stride / d_dlogits as in qpn_ce_loss.  d_logits must be a
 * (B x BL x n_quantize) buffer; with want_logits bit 0 clear the implementation may leave it unwritten.  want_logits bit 1
 * (QPN_FWD_BACKWARD_FOLLOWS = 2): the caller promises to run qpn_train_backward[_ex] of this forward next, with this same
 * d_dlogits untouched -- the post-net's backward may then run inside the forward's launch sequence (one kernel for both
 * directions of a row tile); a backward with another buffer, or a repeated one, still computes everything itself.  The loss
 * stays on the device until qpn_train_loss (synchronises) is called. */
#define QPN_FWD_BACKWARD_FOLLOWS 2
int qpn_train_forward_loss(qpn_handle* h, const float* d_flat, int B, int64_t T, int64_t F, int64_t Td, int BL, int maxd,
                           const int64_t* d_x, const float* d_h, const float* d_dfac, const int64_t* d_targets, int64_t tgt_stride,
                           float* d_logits, int want_logits, float* d_dlogits, void* stream);
int qpn_train_loss(qpn_handle* h, double* h_loss, void* stream);
/* The same number without draining the stream at every step (the reference reads loss.item() per step, src/bin/qpnet_train.py:533, and
 * only ever uses the values summed over a reporting interval, :536-541): _enqueue copies this step's loss to pinned memory behind the
 * step's kernels; _collect(newest = 0) returns the one enqueued ONE call earlier (finished long ago: no wait in practice), newest = 1 the
 * last one (waits for it).  *h_valid = 0 when there is no such copy (first step, or already collected). */
int qpn_train_loss_enqueue(qpn_handle* h, void* stream);
int qpn_train_loss_collect(qpn_handle* h, int newest, double* h_loss, int* h_valid);

/* torch.optim.Adam step (reference src/bin/qpnet_train.py:426-429,531), fp32, in place:
 * d_flat, d_m, d_v (n floats each) updated from d_grad; `step` is the 1-based step count. */
int qpn_adam_step(qpn_handle* h, float* d_flat, const float* d_grad, float* d_m, float* d_v, int64_t n,
                  int step, float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* Same, reading the gradient as d_grad[i] / d_grad_denominator[0] (device scalar, e.g. the summed row count behind an
 * all-reduced gradient; NULL = 1): no host read-back and no extra elementwise launch in a data-parallel step.  Non-NULL: d_grad_denominator[1] is the
 * exchanged trailer's flag count (qpn_train_backward_ex): > 0 skips the update, as the handle's own status word does. */
int qpn_adam_step_ex(qpn_handle* h, float* d_flat, const float* d_grad, float* d_m, float* d_v, int64_t n,
                     int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                     const float* d_grad_denominator, void* stream);

/* Adam updates APPLIED on this handle so far, counted on the device: a k_adam launch that finds the sticky status word set -- or, behind a data-parallel
 * exchange, a peer rank's flag in the trailer (d_grad_denominator[1] > 0) -- applies nothing (every rank skips a step ANY rank flagged; the others report
 * "a peer rank flagged ..." with QPN_ERANGE).  Drains `stream`.  A caller that keeps the bias correction's step number on the host (the reference:
 * torch.optim.Adam's state["step"], src/bin/qpnet_train.py:531) re-bases it on this count after a status error. */
int qpn_train_applied_updates(qpn_handle* h, int64_t* applied, void* stream);

/* One optimisation step of the reference's training loop (src/bin/qpnet_train.py:517-531: forward, CrossEntropyLoss, backward, Adam.step, loss.item())
 * behind ONE call: qpn_train_forward_loss + qpn_train_backward + qpn_adam_step and the loss / status bookkeeping, in that order, so that the host side of
 * a step is a single foreign call.  d_logits / d_dlogits: caller-owned B*BL*n_quantize floats (the logits are not written); d_grad: n (+ 4) floats.
 * loss_mode 0: none; 1: "lagged" -- this step's loss is copied out behind its kernels, *h_loss receives the previous step's (*h_valid = 0 when there is none;
 * qpn_train_loss_collect(newest = 1) fetches the last one), the stream is never drained; 2: this step's loss, read in the call.  The device-side status word
 * (a tap or target out of range) is reported two steps late at most in modes 0 / 1, in the call in mode 2; flagged steps do not touch the parameters. */
int qpn_train_step(qpn_handle* h, float* d_flat, int B, int64_t T, int64_t F, int64_t Td, int BL, int maxd,
                   const int64_t* d_x, const float* d_h, const float* d_dfac, const int64_t* d_targets, int64_t tgt_stride,
                   float* d_logits, float* d_dlogits, float* d_grad, float* d_m, float* d_v, int64_t n,
                   int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                   int loss_mode, double* h_loss, int* h_valid, void* stream);

/* Per-launch-group device timings of the training calls issued between begin and end (HIP events on `stream`; used by bench.py for
 * the roofline).  h_ms[QPN_PG_*] receives milliseconds.  While a profile is being taken a step runs on ONE stream, and every heavy
 * kernel is a group of its own: LAYER_FWD / LAYER_BWD = the residual stack (one work-queue launch each at n_resch 64), WGRAD = the
 * gate contraction's weight gradient, WGRAD_WR / _SKIP / _POST / _CAUSAL = the residual 1x1's, the skip 1x1's, the post-net pair's and
 * the causal table's. */
#define QPN_PG_PREP 0
#define QPN_PG_LAYER_FWD 1
#define QPN_PG_POST_FWD 2
#define QPN_PG_CE 3
#define QPN_PG_POST_BWD 4
#define QPN_PG_WGRAD 5
#define QPN_PG_LAYER_BWD 6
#define QPN_PG_GRAD_TAIL 7
#define QPN_PG_ADAM 8
#define QPN_PG_ALLREDUCE 9     /* marked by the caller after its gradient all-reduce (qpn_train_profile_mark) */
#define QPN_PG_WGRAD_WR 10
#define QPN_PG_WGRAD_SKIP 11
#define QPN_PG_WGRAD_POST 12
#define QPN_PG_WGRAD_CAUSAL 13
#define QPN_PG_COUNT 14
/* Diagnostics of the one-launch residual stack (csrc/train_stack.hip; no reference counterpart): the first n <= 1024 control words of
 * the work queues as the last step left them -- [1] abort raised; forward [4..6] / backward [8..10]: escalations (a workgroup published
 * everything it held before waiting without a bound), polls spent waiting, waves that did not find their producers' flags at first look.  Synchronises the stream. */
int qpn_train_stack_stats(qpn_handle* h, unsigned* h_out, int n, void* stream);
/* Data-parallel step, two buckets (no reference counterpart: its DataParallel wrap, src/bin/qpnet_train.py:416-423, never runs with more than one
 * GPU).  After qpn_train_backward_ex(append_scale = 1) has been enqueued: [*first, *first + *count) is the tail of the exchange buffer -- the
 * post-net gradient blocks and the 4-float row-count trailer -- which the side stream completes while the layer backward is still running, and
 * `stream` is made to wait for exactly that (an event, no host wait).  The caller all-reduces that range on `stream`, the rest [0, *first) on
 * the stream the backward was enqueued on, and joins the two before qpn_adam_step_ex.  *count = 0: nothing finished early, exchange the whole buffer. */
int qpn_train_early_bucket(qpn_handle* h, int64_t* first, int64_t* count, void* stream);
/* First element of that tail (a property of the parameter layout; -1 if the layout has no such tail or the handle has not trained yet): ranks
 * agree on it ONCE before they split the exchange, so that every rank issues the same collectives in every step. */
int64_t qpn_train_early_first(qpn_handle* h);
int qpn_train_profile_begin(qpn_handle* h, void* stream);
/* The same with the step left as the timed loop runs it (the skip / post-net weight gradients, the early slab reduction, the aux tail and dWr on the
 * handle's side stream next to the stack backward and dW1): every launch is timed on the stream it runs on -- what a kernel costs INSIDE the
 * overlapped step (bench.py: roofline.overlapped, and the kernel the headline fraction is quoted for). */
int qpn_train_profile_begin_overlapped(qpn_handle* h, void* stream);
int qpn_train_profile_mark(qpn_handle* h, int group, void* stream);   /* attribute the work enqueued since the previous mark to `group` */
int qpn_train_profile_end(qpn_handle* h, float* h_ms, int n, void* stream);

/* _dilated_index (src/nets/qpnet.py:592-604, tensor path) and _generate_dilated_index
 * (src/nets/qpnet.py:613-618): d (B x L) float32 -> int64 (B x L), NOT replicated over
 * channels (the reference's .repeat over n_ch is redundant). */
int qpn_dilated_index_train(const float* d_d, int B, int64_t L, int dilation, int64_t* d_out, void* stream);
int qpn_dilated_index_gen_f32(const float* d_d, int64_t n, int dilation, int64_t* d_out, void* stream);
int qpn_dilated_index_gen_f64(const double* d_d, int64_t n, int dilation, int32_t* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QPNET_HIP_H */
