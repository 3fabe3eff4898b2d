/*
 * mjx.h -- C ABI of libmjx.so: the MI355X (gfx950) implementation of mjrl's
 * NPG / TRPO policy-update hot path, GAE and baseline fitting.
 *
 * The reference (aravindr93/mjrl) is pure Python and has no FFI of its own; the
 * seam this library sits behind is Agent.train_from_paths + process_samples.* +
 * Baseline.{fit,predict}.  Each entry point below names the reference code it
 * replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success, <0 = mjx error, >0 = hipError_t;
 *     mjx_last_error() returns a thread-local message for the last failure.
 *   - all `const float*` / `double*` DATA pointers are DEVICE pointers unless the
 *     parameter name ends in `_host`.  The caller owns them (e.g. torch tensors);
 *     the context owns only its internal workspace.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     work is enqueued asynchronously; nothing synchronises unless stated.
 *   - flat parameter order = [W1 (h1 x n) row-major, b1, W2 (h2 x h1), b2, ...,
 *     W_out (m x h_last), b_out, log_std (m)]
 *     (mjrl/policies/gaussian_mlp.py:37,50-52,60-63).
 *   - transforms are packed as [in_shift(n), in_scale(n), out_shift(m), out_scale(m)]
 *     (mjrl/utils/fc_network.py:27-37).
 *   - one context per process / per GPU; calls on one context are not re-entrant.
 */
#ifndef MJX_H
#define MJX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mjx_ctx mjx_ctx;

#define MJX_OK 0
#define MJX_ERR_ARG (-1)
#define MJX_ERR_STATE (-2)
#define MJX_ERR_UNSUPPORTED (-3)
#define MJX_ERR_NOGPU (-4)

/* ---- lifecycle ----------------------------------------------------------- */
const char* mjx_last_error(void);
int mjx_version(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int mjx_device_count(void);

/* Fork guard.  The reference's sampler forks its workers from the training process (mjrl/samplers/core.py:189-210, mp.Pool) and
 * pickles the policy into them (:196); a child forked from a process that holds HIP state cannot use the device.  out2[0] = number
 * of device-touching entries (mjx_device_count / mjx_create / mjx_malloc / mjx_stage_async) THIS process has made -- zeroed in a
 * forked child; out2[1] = 1 when this process was forked from one that had made any.  In such a child mjx_create / mjx_malloc /
 * mjx_stage_async fail with MJX_ERR_STATE and mjx_device_count returns 0 (nothing hangs inside the runtime). */
int mjx_process_state(int64_t* out2);

/* Create a context for a tanh-MLP Gaussian policy obs(n) -> hidden[...] -> act(m)
 * on HIP device `device`.  Replaces the torch modules built in
 * mjrl/policies/gaussian_mlp.py:8-56 (MLP) and gaussian_linear.py:9-56
 * (n_hidden = 0).  max_samples bounds the per-call batch (workspace sizing for
 * the layer-wise path); pass 0 to size lazily at mjx_bind_batch. */
int mjx_create(mjx_ctx** out, int device, int n, int m, const int* hidden, int n_hidden);
void mjx_destroy(mjx_ctx* ctx);
/* number of flat parameters d (gaussian_mlp.py:53) */
int64_t mjx_num_params(const mjx_ctx* ctx);
/* 1 if the fused single-kernel path serves this network shape, 0 if the
 * layer-wise path does */
int mjx_uses_fused_path(const mjx_ctx* ctx);

/* ---- plain device-memory helpers (so a C / cgo / JNI caller needs no torch; tests/c/c_caller.c) -- */
/* hipMalloc of bytes + 16: an observation block allocated here can never fault on the up-to-12-byte tail read documented at
 * mjx_bind_batch */
int mjx_malloc(void** dev_ptr, int64_t bytes);
int mjx_free(void* dev_ptr);
int mjx_memcpy_h2d(void* dst_dev, const void* src_host, int64_t bytes, void* stream);
int mjx_memcpy_d2h(void* dst_host, const void* src_dev, int64_t bytes, void* stream);
int mjx_stream_sync(void* stream);

/* ---- binding the inputs of one update ----------------------------------- */
/* The (N_local, n) observation, (N_local, m) action and (N_local) advantage
 * blocks of THIS rank's trajectory shard, fp32 row-major -- the arrays
 * process_paths concatenates (mjrl/algos/batch_reinforce.py:178-185).
 * N_global = total samples over all ranks (means are taken over N_global).
 * obs must be 16-byte aligned (it is read with 16-byte loads, and up to 12 bytes past its
 * end may be read -- always inside the allocation for hipMalloc / torch memory).
 * act / adv may be NULL when only mjx_fvp is used.
 * Every call drops what earlier calls cached for the previous batch (see mjx_surr_vpg). */
int mjx_bind_batch(mjx_ctx* ctx, const float* obs, const float* act, const float* adv,
                   int64_t N_local, int64_t N_global);

/* Re-bind the leading N_local rows of the batch given to the last mjx_bind_batch (same storage, same
 * contents), optionally with another advantage vector (NULL keeps the current one): DAPG evaluates the
 * gradient on [on-policy ; demonstrations] and the Fisher / surrogate / KL on the on-policy prefix
 * (mjrl/algos/dapg.py:92-106).  Unlike mjx_bind_batch this keeps what mjx_surr_vpg cached for the batch
 * (forward activations for mjx_fvp, old-policy outputs for mjx_eval_surr_kl).  Changing the CONTENTS of
 * the bound buffers requires a new mjx_bind_batch. */
int mjx_bind_rows(mjx_ctx* ctx, int64_t N_local, int64_t N_global, const float* adv);
/* theta_new / theta_old: flat parameter vectors (d floats each) of policy.model /
 * policy.old_model (+log_std); tr_new / tr_old: packed transforms (2n+2m floats)
 * or NULL for identity (parameter vectors 16-byte aligned).  old_is_new != 0 asserts both describe the same function
 * (always true at entry to train_from_paths, gaussian_mlp.py:44-45, npg_cg.py:142). */
int mjx_bind_policy(mjx_ctx* ctx, const float* theta_new, const float* theta_old,
                    const float* tr_new, const float* tr_old, int old_is_new);

/* ---- K1: CPI surrogate + vanilla policy gradient ------------------------- */
/* grad_out[d] = sum over local samples of d/dtheta_new [LR_i * adv_i] / N_global
 * (BatchREINFORCE.flat_vpg, batch_reinforce.py:54-58);
 * scal_out[0] = sum_i LR_i*adv_i (local, NOT divided; CPI_surrogate :40-46),
 * scal_out[1] = number of local samples.  scal_out is 4 doubles.
 * With old_is_new the call also keeps, per sample of the bound batch, the hidden activations, the normalised
 * observations and the policy's mean / log-likelihood (about 630 B per sample of device memory, owned by the context):
 * the mjx_fvp calls of the same update then skip the forward pass, mjx_eval_surr_kl skips the old policy's (the latter
 * after checking in the kernel that theta_old / tr_old still hold the values they had here).  mjx_bind_policy drops
 * the activations, mjx_bind_batch drops everything, mjx_bind_rows keeps both. */
int mjx_surr_vpg(mjx_ctx* ctx, float* grad_out, double* scal_out, void* stream);

/* ---- K2: Fisher-vector product ------------------------------------------- */
/* out[d] = local share of (Hessian of mean_kl wrt theta_new at theta_new==theta_old) * v,
 * WITHOUT the damping term (NPG.HVP, mjrl/algos/npg_cg.py:62-81 minus `regu_coef*vector`).
 * Summing `out` over ranks gives the full product.  Requires old_is_new.
 * Reproducibility: with the forward-activation cache of mjx_surr_vpg in place the product walks the sample tiles alternately
 * back-to-front and front-to-back (a counter the context resets in mjx_surr_vpg and mjx_cg_init: product k of a solve always
 * walks the same way, so whole solves / updates are bit-reproducible).  Two stand-alone calls with the same `v` therefore sum
 * the per-workgroup partials of DIFFERENT tile sets and may differ in the last bits (~1e-7 relative); call mjx_cg_init between
 * them -- or export MJX_FVP_SWEEP=0 (always front-to-back; costs ~3.5 % at 1M samples, the activation cache then defeats the
 * memory-side cache) -- when bit-identical repeats of single products are needed. */
int mjx_fvp(mjx_ctx* ctx, const float* v, float* out, void* stream);

/* ---- K3: surrogate + KL evaluation (no gradient) ------------------------- */
/* scal_out[0] = sum_i LR_i*adv_i, scal_out[1] = sum_i KL_i(new||old) (local sums,
 * divide by N_global after the cross-rank sum) -- CPI_surrogate + kl_old_new,
 * batch_reinforce.py:40-52; the four post-step forwards of npg_cg.py:140-141 and
 * the TRPO line search trpo.py:107-126. */
int mjx_eval_surr_kl(mjx_ctx* ctx, double* scal_out, void* stream);

/* ---- K4: conjugate gradient (mjrl/utils/cg_solve.py:3-22) ----------------- */
/* Optional cross-rank sum hook, called on the stream between the local FVP and
 * the CG vector update (buf holds `count` floats).  NULL = single rank. */
typedef int (*mjx_allreduce_fn)(void* user, float* buf, int64_t count, void* stream);
/* x_out = CG(A, b) with A p = allreduce(mjx_fvp(p)) + damping*p, x0 = 0 (the
 * reference ignores its x_0 argument), `iters` iterations, early stop once
 * r.r < tol (the remaining launches become no-ops on device: no host sync).
 * allreduce == NULL: the attached communicator's ncclAllReduce (mjx_comm_init) when there is one, else single rank.
 * Also writes bdotx_out[0] = b . x  (double) for the step-size rule. */
int mjx_cg_solve(mjx_ctx* ctx, const float* b, int iters, float damping, double tol,
                 float* x_out, double* bdotx_out, mjx_allreduce_fn allreduce, void* user, void* stream);
/* the three pieces, for callers that interleave their own collective: */
int mjx_cg_init(mjx_ctx* ctx, const float* b, void* stream);
const float* mjx_cg_p(mjx_ctx* ctx);                       /* current search direction (device) */
int mjx_cg_step(mjx_ctx* ctx, const float* Ap_nodamp, float damping, double tol, void* stream);
int mjx_cg_finish(mjx_ctx* ctx, const float* b, float* x_out, double* bdotx_out, void* stream);

/* ---- multi-rank: one process per GPU, RCCL over xGMI -------------------------------------------- */
/* The batch shards by whole trajectories over the ranks; every sample sum is formed locally with N_global and summed
 * over the ranks: d floats after K1 and after every Fisher-vector product of the solve, 4 doubles after K1 / K3.  With a
 * communicator attached, mjx_cg_solve (allreduce == NULL) and mjx_npg_update issue those sums as ncclAllReduce calls on
 * the launch stream between their kernels: no host round trip and no cross-stream event per CG iteration.  The
 * reference has nothing to replace here (its only parallelism is the sampler pool, mjrl/samplers/core.py:189-210).
 * librccl is bound at run time (dlopen; the copy the process already mapped -- PyTorch-ROCm's -- is preferred). */
#define MJX_COMM_ID_BYTES 128
/* rank 0: a fresh communicator id (ncclGetUniqueId); the caller hands the 128 bytes to the other ranks */
int mjx_comm_unique_id(char* id_out_host);
/* every rank: join (ncclCommInitRank on the context's device; collective, blocks until all ranks joined).
 * world == 1 is allowed: the collectives still run on a 1-rank communicator (rehearsals). */
int mjx_comm_init(mjx_ctx* ctx, int rank, int world, const char* id_host);
int mjx_comm_destroy(mjx_ctx* ctx);
/* A transport hook in place of RCCL (tests that put two ranks on one GPU, which RCCL refuses; other fabrics): called on
 * the host between the kernel launches of mjx_cg_solve / mjx_npg_update; it must leave the sum over the ranks in buf
 * (count elements, dtype 0 = fp32 / 1 = fp64) ordered after the work already enqueued on `stream` and before what
 * follows.  fn == NULL detaches.  Not allowed while an RCCL communicator is attached. */
typedef int (*mjx_reduce_fn)(void* user, void* buf, int64_t count, int dtype, void* stream);
int mjx_comm_set_callback(mjx_ctx* ctx, mjx_reduce_fn fn, void* user, int world);
/* number of ranks of the attached communicator / transport hook, 0 = none */
int mjx_comm_world(const mjx_ctx* ctx);
/* in-place sum over the ranks on `stream`; dtype 0 = fp32, 1 = fp64 */
int mjx_comm_allreduce(mjx_ctx* ctx, void* buf, int64_t count, int dtype, void* stream);

/* Peer exchange: a third transport for the same rank sums, built on HIP IPC and stream memory operations instead of RCCL --
 * for the 5-23 KB vectors of this path a collective library's launch + protocol latency is most of the cost.  Every rank owns one
 * uncached device buffer [2 parities][world slots of d floats + 4 doubles] + one arrival flag per source rank; mjx_peer_export allocates it and returns its
 * hipIpcMemHandle_t (MJX_PEER_HANDLE_BYTES bytes), the caller gathers the world's handles over any side channel (rank order) and
 * hands them to mjx_peer_connect, which maps the peers' buffers.  From then on every rank sum of mjx_comm_allreduce /
 * mjx_cg_solve / mjx_npg_update / mjx_trpo_update / mjx_dapg_update is: store the local vector into slot `rank` of EVERY rank's
 * buffer, then -- once those stores are acknowledged -- store the exchange number into this rank's flag in every peer's buffer
 * (in the CG loop the Fisher product's reduction kernel does both itself); the consuming kernel (in the loop: the CG vector
 * update) waits on the flags of its own buffer -- local memory, one polling thread per source rank -- and sums the local slots
 * in rank order.  (r06: in the one-call updates on the fused kernels the gradient and K1's four sums travel in ONE exchange -- the
 * sums at the end of the slot -- raised by the gradient's reduction kernel and consumed by the solve's first kernel; K3's sums are
 * pushed by their own reduction kernel.  A rank without samples sends zeros through the same exchanges.)
 * A flag and the data it announces come from the same rank over the same path; nothing is assumed about the
 * relative order of different peers' traffic.  All on the launch stream: no host synchronisation, no extra launch in the loop,
 * bit-identical results on all ranks.  The wait is bounded: a peer that has not delivered within MJX_PEER_TIMEOUT_MS
 * (environment, read by mjx_peer_export; default 5000) turns the result into NaN instead of hanging the GPU and is counted:
 * mjx_peer_status returns (and clears) the number of timed-out waits since the last call -- a host that reads back a
 * non-finite update asks it and raises (mjrl_amd.engine does).  After a timeout the ranks' exchange sequences are out of step:
 * tear the transport down (mjx_comm_destroy).  One process per rank, 2 <= world <= 16; the ranks may own different GPUs of a node (peer access over
 * xGMI) or share one (tests).  Every rank must issue the same sequence of rank sums.  Replaces nothing in the reference (its
 * only parallelism is the sampler pool).  mjx_comm_destroy tears it down.
 * handles == NULL: loop-back rehearsal -- every "peer" is this rank's own buffer, so the stores, counter updates, the stream
 * wait and the `world`-slot sums of a world-rank exchange all execute while the sums see this rank's vector only (a timing
 * diagnostic: bench.py --rehearse-world R --rehearse-transport peer). */
#define MJX_PEER_HANDLE_BYTES 64
int mjx_peer_export(mjx_ctx* ctx, int rank, int world, char* handle_out);
int mjx_peer_connect(mjx_ctx* ctx, const char* handles /* world x MJX_PEER_HANDLE_BYTES, rank order */);
/* *timeouts_out = waits of this rank's consumer kernels that gave up since the last call (0 without a peer transport); clears the
 * count.  Synchronises with the device (a 4-byte read): call it after an update's own read-back, not inside the loop. */
int mjx_peer_status(mjx_ctx* ctx, int* timeouts_out);

/* ONE device-resident NPG update enqueued without a host round trip -- what NPG.train_from_paths does between
 * process_paths and the parameter read-back (mjrl/algos/npg_cg.py:108-142):
 *   K1   grad = flat_vpg, surrogate sums                                   [rank sum: d floats + 4 doubles]
 *   K4   x = CG(F + damping I, grad): iters x (K2 [rank sum: d floats] + vector update), early stop below tol
 *        alpha = sqrt(|step_size / (grad.x + 1e-20)|)   (:133), or alpha = const_alpha when const_alpha is not NaN (:128-130; any finite
 *        value, zero and negative included, is applied as given -- NAN selects the normalised step)
 *        theta_out = theta_old + alpha x, log_std = max(log_std, min_log_std)   (:137-139, gaussian_mlp.py:73-75)
 *   K3   surrogate / KL sums of theta_out against theta_old                [rank sum: 4 doubles]
 * (r06, d <= 8192: b.x, x_out and theta_out are formed by the kernel of the solve's LAST vector update; K1's and K3's sums are
 *  reduced by the vector reduction / pushed to the peers by their own reduction -- one launch each around the loop on one rank,
 *  two on a peer rank; the arithmetic and its order are those of the call-by-call sequence, bit for bit.)
 * Requires mjx_bind_batch and mjx_bind_policy(old_is_new = 1).  theta_out (d floats) may be the bound theta_new
 * buffer (it is rewritten in place) but must not alias theta_old.  On return the context's NEW parameters are
 * theta_out, as after mjx_bind_policy(theta_out, theta_old, tr_new, tr_old, 0).
 * results (device, 16 doubles): [0] sum LR*adv and [1] sum KL at theta_out (K3); [4] sum LR*adv at theta_old and
 * [5] local sample count (K1); [8] grad.x; [9] alpha (left untouched under const_alpha).  Sums are over all ranks;
 * divide by N_global.  grad_out / x_out: d floats each. */
int mjx_npg_update(mjx_ctx* ctx, int iters, float damping, double tol, double step_size, double const_alpha,
                   float min_log_std, float* grad_out, float* x_out, float* theta_out, double* results, void* stream);

/* The TRPO update with its backtracking line search on the device (mjrl/algos/trpo.py:100-126): K1, CG and
 * alpha = sqrt(|step_size / (grad.x + 1e-20)|) (step_size = 2 kl_dist, :103-104) as in mjx_npg_update when `first`, then
 * n_trials (1..24) line-search trials enqueued back to back: theta_out = theta_old + alpha x, K3 (rank sums as above), accept
 * if the mean KL is below kl_dist, otherwise alpha <- 0.9 alpha (:107-118).  Once a trial is accepted the remaining trials of
 * the call leave theta_out alone (their evaluations repeat the accepted one).  The caller reads `results` once per call and
 * calls again with first = 0 while results[10] == 0 and results[11] < 100 (the reference gives up after 100 trials, :119-120).
 * results (device, 64 doubles): as for mjx_npg_update, and [9] step length of the last trial performed, [10] 1 if a trial was
 * accepted, [11] trials performed so far, [12] step length the next trial would use, [16 + 2k] / [17 + 2k] sum LR*adv /
 * sum KL of trial k at ring position k % 24 (a call performs at most 24 trials, so nothing is overwritten before it is read).
 * After 100 rejected trials the caller applies the zero step (mjx_apply_step with alpha = 0, :119-126). */
int mjx_trpo_update(mjx_ctx* ctx, int iters, float damping, double tol, double step_size, double kl_dist, int n_trials, int first,
                    float min_log_std, float* grad_out, float* x_out, float* theta_out, double* results, void* stream);

/* The whole DAPG update (mjrl/algos/dapg.py:92-121) enqueued by one call.  At entry the batch bound with mjx_bind_batch is
 * the block [on-policy rows ; demonstration rows] with the advantages of dapg.py:65-70 (N_global = all rows over all
 * ranks) and theta_new == theta_old.  Sequence: K1 over all rows (rank sum), gradient x N_all / N_on (:97-98, one fp32
 * product per element); mjx_bind_rows(rows_on, N_on_global, adv_on) -- the Fisher metric, surrogate and KL use the on-policy
 * prefix with the whitened on-policy advantages adv_on (device, rows_on floats) --; K3 = surr_before (:92); CG (:103-106);
 * alpha = sqrt(|step_size / (grad.x + 1e-20)|) with step_size = 2 kl_dist (:111-112); theta_out = theta_old + alpha x with
 * the log_std clamp; K3 at theta_out (:117-118).  On return the context is bound to the on-policy prefix and to theta_out
 * as after mjx_bind_rows + mjx_bind_policy(theta_out, theta_old, tr_new, tr_old, 0).
 * results as for mjx_npg_update, with [4] = sum LR*adv over the ON-POLICY rows at theta_old (surr_before x N_on_global). */
int mjx_dapg_update(mjx_ctx* ctx, int iters, float damping, double tol, double step_size, float min_log_std, int64_t rows_on,
                    int64_t N_on_global, const float* adv_on, float* grad_out, float* x_out, float* theta_out, double* results,
                    void* stream);

/* theta_out = theta + alpha * x, then log_std = max(log_std, min_log_std)
 * (npg_cg.py:137-139 + gaussian_mlp.py:73-75). */
int mjx_apply_step(mjx_ctx* ctx, const float* theta, const float* x, float alpha,
                   float min_log_std, float* theta_out, void* stream);

/* The same with the normalised NPG step length formed on the device from the solve's g.x (bdotx_out of mjx_cg_solve /
 * mjx_cg_finish): alpha = sqrt(|step_size / (g.x + 1e-20)|) in fp64 (npg_cg.py:133), written to alpha_out (device,
 * optional).  Lets a caller enqueue surrogate -> solve -> step -> evaluation without reading anything back in between. */
int mjx_apply_npg_step(mjx_ctx* ctx, const float* theta, const float* x, const double* gdotx, double step_size,
                       float min_log_std, float* theta_out, double* alpha_out, void* stream);

/* ---- K5: returns / GAE over ragged trajectories --------------------------- */
/* y[t] = x[t] + gamma*y[t+1] within each trajectory [offsets[i], offsets[i+1]),
 * terminal value 0 (process_samples.discount_sum :37-44, compute_returns :3-5). fp64. */
int mjx_discount_scan(const double* x, const int64_t* offsets, int64_t n_traj, double gamma,
                      double* y, void* stream);
/* tpos[s] = index of sample s inside its trajectory, for all trajectories [offsets[i], offsets[i+1]) (device pointers; int32
 * out): the np.arange(len(path)) of the baselines' time features (quadratic_baseline.py:28, mlp_baseline.py:47), formed where
 * the feature kernels (mjx_bl_*) read it. */
int mjx_time_index(const int64_t* offsets, int64_t n_traj, int32_t* tpos, void* stream);
/* GAE branch of compute_advantages (process_samples.py:21-29):
 *   b1 = [b, terminated ? 0 : b[-1]]; td = r + gamma*b1[1:] - b1[:-1];
 *   adv = discount_scan(td, gamma*lam).
 * lam outside [0,1] selects the non-GAE branch adv = returns - baseline (:10-13),
 * in which case `rewards` must hold the returns. */
int mjx_gae(const double* rewards, const double* baseline, const int64_t* offsets,
            const uint8_t* terminated, int64_t n_traj, double gamma, double lam,
            double* adv, void* stream);
/* Advantage whitening of process_paths (batch_reinforce.py:185) fused with the
 * fp64->fp32 cast the reference performs per call (gaussian_mlp.py:102-109):
 * out32 = (adv - mean)/(std + eps) with mean/std supplied by the caller (the
 * cross-rank values); stats_out = [sum, sum of squares about `shift`, count]. */
int mjx_sum_stats(const double* x, int64_t N, double shift, double* stats_out, void* stream);
int mjx_whiten_cast(const double* adv, int64_t N, double mean, double std, double eps,
                    float* out32, void* stream);
int mjx_cast_f64_f32(const double* x, int64_t count, float* out32, void* stream);

/* Batched policy inference: mean_out[i] = out_scale * MLP_theta((obs[i] - in_shift) / (in_scale + 1e-8))
 * + out_shift for N observations resident in HBM -- FCNetwork.forward (mjrl/utils/fc_network.py:39-52) on a
 * (N, n) block, as the model-based rollouts and evaluation sweeps call it (mjrl/algos/model_accel/sampling.py:66-89).
 * theta: flat parameters, tr: packed transforms (NULL = identity); neither has to be the bound policy.
 * Runs on the MFMA GEMM chain of the layer-wise path (fused bias + tanh / output-affine epilogues). */
int mjx_policy_forward(mjx_ctx* ctx, const float* obs, int64_t N, const float* theta, const float* tr,
                       float* mean_out, void* stream);

/* Minibatch Adam on the policy parameters: the torch-optimizer loops of behaviour cloning
 * (mjrl/algos/behavior_cloning.py:107-136, loss 0 = MSE, 1 = MLE) and PPO (mjrl/algos/ppo_clip.py:85-95, loss 2 =
 * clipped surrogate against theta_old / tr_old) with the whole batch resident in HBM.  idx holds steps x B row
 * indices (int32) -- the caller draws them (np.random.choice) so the random stream matches the reference.  theta is
 * updated in place; adam_m / adam_v (d floats each) and step0 (steps already taken) are the torch.optim.Adam
 * state (betas 0.9 / 0.999, eps 1e-8, no weight decay).  MSE leaves the log_std block and its Adam state alone
 * (it has no gradient there).  loss_trace (optional, steps doubles) receives every minibatch loss.
 * adv, theta_old, tr_old are only read for loss 2; tr / tr_old NULL = identity.
 * old_tracks_new (loss 2): 0 = the old policy stays fixed during the epochs (the algorithm as published);
 * 1 = the old NETWORK is evaluated with the current weights and only the old log_std stays fixed -- what the
 * reference computes once policy.set_param_values has been called with a float32 array (its new and old network
 * tensors then alias the same memory, mjrl/policies/gaussian_mlp.py:65-87), i.e. from the second iteration on. */
int mjx_policy_minibatch_adam(mjx_ctx* ctx, int loss, const float* obs, const float* act, const float* adv, const int32_t* idx,
                              int64_t steps, int B, float* theta, const float* tr, const float* theta_old, const float* tr_old,
                              int old_tracks_new, float* adam_m, float* adam_v, int64_t step0, float lr, float clip,
                              double* loss_trace, void* stream);

/* Host-side gather for rollout ingestion: copies blocks [first, first + count) of a list of
 * per-trajectory arrays -- src[i] is rows(i) x row_bytes, C-contiguous -- to their place in one staging block,
 * dst + offsets[i] * row_bytes (offsets = cumulative row counts, n_blocks + 1 entries), with n_threads worker
 * threads.  Replaces np.concatenate over the path list (mjrl/algos/batch_reinforce.py:180-181); dst is
 * normally page-locked memory that is then sent with one asynchronous copy per group.  No device work. */
int mjx_host_gather(void* dst, const void* const* src, const int64_t* offsets, int64_t first, int64_t count,
                    int64_t row_bytes, int n_threads);
/* The same gather for fp64 blocks that the device only needs in fp32 (the policy's observations and actions;
 * mjrl/algos/batch_reinforce.py:180-181 concatenates fp64, mjrl/policies/gaussian_mlp.py casts to float32 on entry): converts
 * while copying -- round to nearest even, the bits of ndarray.astype(float32) and of mjx_cast_f64_f32 -- so the block leaves the
 * host at half its size and needs no device-side cast.  src[i] is rows(i) x row_elems doubles, dst the fp32 staging block. */
int mjx_host_gather_f64_f32(float* dst, const double* const* src, const int64_t* offsets, int64_t first, int64_t count,
                            int64_t row_elems, int n_threads);
/* The whole staging of one block of a batch as ONE asynchronous job (r04): src[i] is lens[i] rows x row_elems items of
 * src_itemsize bytes (8: float64, 4: float32).  A native thread gathers the trajectories group by group (>= group_rows rows, whole
 * trajectories) into `pinned` -- converting fp64 -> fp32 on the way when hostcast != 0 -- and queues each group's host-to-device
 * copy on `stream` right behind its gather, so gather k + 1 overlaps transfer k; with hostcast == 0, float64 sources and
 * device_f32 != NULL the fp32 image is cast from the raw block on the device (mjx_cast_f64_f32 on the same stream).  Returns at
 * once: *job_out is joined (and freed) by mjx_stage_wait, after which everything is QUEUED on `stream` (order consumers after
 * it).  src / lens are copied; the trajectories themselves, `pinned` and the device blocks must stay alive until the wait.
 * Replaces np.concatenate + the per-call float32 casts of mjrl/algos/batch_reinforce.py:180-181 / mjrl/policies/gaussian_mlp.py:102-109
 * without a Python thread (no interpreter lock is held while the 184 MB of a 1M-timestep batch move). */
int mjx_stage_async(void** job_out, const void* const* src, const int64_t* lens, int64_t count, int64_t row_elems, int src_itemsize,
                    int hostcast, void* pinned, void* device_raw, float* device_f32, int64_t group_rows, int n_threads,
                    int device_index, void* stream);
int mjx_stage_wait(void* job);
/* Per-trajectory sums of a 1-D fp64 quantity: out[i] = src[i][0] + src[i][1] + ... + src[i][lens[i] - 1], added in that order --
 * the path returns of process_paths, `path_returns = [sum(p["rewards"]) for p in paths]` (mjrl/algos/batch_reinforce.py:187; Python's
 * sum() adds left to right, so the results carry the reference's bits).  Trajectories are spread over n_threads.  No device work. */
int mjx_host_segment_sums(const double* const* src, const int64_t* lens, int64_t count, double* out, int n_threads);
/* np.random.permutation(n) of NumPy's legacy (RandomState / global) MT19937 stream, bit for bit, as int32: what MLPBaseline.fit draws
 * once per epoch (mjrl/utils/optimize_model.py:22).  key624 / pos_io: the generator state as np.random.get_state() returns it
 * (624 words + position), advanced in place -- hand it back with np.random.set_state.  n < 2^31.  Host only, any thread. */
int mjx_host_mt19937_permutation(uint32_t* key624_host, int32_t* pos_io_host, int64_t n, int32_t* out_host);
/* `epochs` consecutive permutations of that stream, out[e * n .. (e + 1) * n): the bits of `epochs` calls of the function above (one
 * fit of MLPBaseline draws one per epoch), with the generator and the swaps running on two threads (r06) */
int mjx_host_mt19937_permutations(uint32_t* key624_host, int32_t* pos_io_host, int64_t n, int epochs, int32_t* out_host);
/* `count` draws of np.random.choice(n, size=...) (with replacement; == np.random.randint(0, n, size=...)) of the same stream: the
 * minibatch row indices BC and PPO draw once per Adam step (mjrl/algos/behavior_cloning.py:113, ppo_clip.py:77); draws of any
 * sizes concatenate, so steps x minibatch indices are ONE call.  1 <= n < 2^31. */
int mjx_host_mt19937_randint(uint32_t* key624_host, int32_t* pos_io_host, int64_t n, int64_t count, int32_t* out_host);

/* ---- K6: value baselines --------------------------------------------------- */
/* Feature maps of the reference baselines over the concatenated fp64 observation block
 * (N x n) with tpos[s] = time index of sample s inside its trajectory:
 *   kind 0  MLP       clip(obs,-10,10)/10, tau^1..4                 F = n + 4   (mlp_baseline.py:36-58)
 *   kind 1  LINEAR    clip/10, 1, tau^1..4                          F = n + 5   (linear_baseline.py:11-35)
 *   kind 2  QUADRATIC clip/10, o_i*o_j (i<=j), 1, tau^1..4          F = n + n(n+1)/2 + 5 (quadratic_baseline.py:11-41)
 * with tau = t / 1000. */
int mjx_bl_num_features(int kind, int n);
/* fp32 feature matrix (N x (n+4)) of the MLP baseline (computed in fp64, cast like mlp_baseline.py:65). */
int mjx_bl_features_f32(const double* obs, const int32_t* tpos, int64_t N, int n, float* out, void* stream);
/* Augmented normal equations in fp64 without materialising the feature matrix A (N x F):
 * G_aug ((F+1) x (F+1), row-major) = [A y]^T [A y], i.e. A^T A, A^T y and y^T y
 * (featmat.T.dot(featmat), featmat.T.dot(returns): quadratic_baseline.py:57-60, linear_baseline.py:48-51). */
int mjx_bl_gram(int kind, const double* obs, const int32_t* tpos, const double* y, int64_t N, int n,
                double* G_aug_out, void* stream);
/* out[s] = features(s) . coef   (fp64; quadratic_baseline.py:71-74) */
int mjx_bl_predict(int kind, const double* obs, const int32_t* tpos, int64_t N, int n, const double* coef,
                   double* out, void* stream);
/* ReLU MLP regressor d_in -> hidden... -> 1 (mlp_baseline.py:21-28), flat params [W1,b1,...]:
 * out[N] = model(feat)                                     (MLPBaseline.predict :97-105) */
int mjx_mlp_predict(const float* feat, int64_t N, int d_in, const int* hidden, int n_hidden, const float* params,
                    float* out, void* stream);
/* `epochs` x (N/batch - 1) minibatch steps of torch.optim.Adam(lr, weight_decay=wd) on the MSE loss
 * (utils/optimize_model.py:7-36); perm holds epochs*N row indices (np.random.permutation per epoch,
 * drawn by the caller to keep NumPy's RNG stream); params / m / v are updated in place, step0 = number
 * of Adam steps already taken; epoch_loss_out[e] = sum of minibatch losses of epoch e (device).
 * The reference's shape (two hidden layers of 128, batch 64) runs as ONE persistent launch: one workgroup up to 55 inputs,
 * ceil(d_in / 48) workgroups with a grid barrier per half-step beyond (up to 768 inputs; csrc/mlp_fit.h).  A wait of ~2 s for
 * another workgroup poisons epoch_loss_out with NaN instead of hanging.  The launch uses per-host-thread scratch (moment pairs,
 * the uncached exchange block): fits issued by one host thread must be stream-ordered with respect to each other. */
int mjx_mlp_fit_adam(const float* feat, const float* y, int64_t N, int d_in, const int* hidden, int n_hidden,
                     float* params, float* m, float* v, int64_t step0, const int32_t* perm, int epochs, int batch,
                     float lr, float wd, double* epoch_loss_out, void* stream);

/* ---- in-library kernel timing (bench.py roofline) ------------------------- */
/* While enabled (on = k >= 1), every k-th launch of the dominant Fisher-vector-product kernel
 * (fused k_fused MODE_FVP, or the whole layer-wise FVP chain) is bracketed by hipEvents recorded
 * on the launch stream.  An event pair costs ~10 us of dispatch serialisation (its markers carry
 * release fences), so a timed run samples with a stride that is coprime to the CG iteration count
 * instead of bracketing every launch.  mjx_profile_read synchronises and returns out[0] = total
 * milliseconds, out[1] = number of launches measured since the last mjx_profile_enable(ctx, k);
 * on = 0 switches the events off.  on = -k (r06): every k-th ITERATION of mjx_cg_solve's loop as a whole instead -- product,
 * reduction (+ peer exchange) and vector update between one pair of events (the last iteration, which also forms the step, is
 * left out); what bench.py's 8-rank rehearsal reports next to the product's own time. */
int mjx_profile_enable(mjx_ctx* ctx, int on);
int mjx_profile_read(mjx_ctx* ctx, double* out_host);
/* every bracketed launch's HIP-event time on its own (milliseconds, launch order; at most `cap` of them are written,
 * *count_out = how many were measured): the distribution behind mjx_profile_read's total */
int mjx_profile_samples(mjx_ctx* ctx, double* ms_out_host, int cap, int* count_out);

/* ---- debugging aid (tests only) ------------------------------------------ */
/* When non-NULL, the fused kernels dump the first tile's intermediates here. */
int mjx_set_debug_buffer(mjx_ctx* ctx, float* dbg, int64_t floats);
/* When non-NULL (device, 4 int64), workgroup 0 of every fused-kernel launch leaves its shader-cycle counter and the 100 MHz
 * real-time counter at entry ([0], [1]) and exit ([2], [3]) of the PRODUCTION kernels: cycles of a launch and the shader clock
 * the chip sustained under it (tools/fvp_time.py; the part trades clock for issue density, so both are needed). */
int mjx_set_clock_buffer(mjx_ctx* ctx, int64_t* clk);

#ifdef __cplusplus
}
#endif
#endif /* MJX_H */
