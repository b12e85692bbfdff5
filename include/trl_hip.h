/*
 * trl_hip.h -- C ABI of libtrl_hip.so, the MI355X (gfx950) learner hot path behind
 * the activatedgeek/torchrl Python surface.
 *
 * The reference has no FFI layer (it is pure Python on torch); every entry point
 * below therefore cites the Python code whose arithmetic it replaces.  The Python
 * classes in torchrl_amd/ keep the reference names and call these through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative TRL_E* code on failure;
 *     trl_last_error() gives the thread's last message.
 *   - all `*_d` / unqualified data pointers are DEVICE pointers owned by the caller
 *     (torch tensors' data_ptr()); `*_h` pointers are HOST pointers.
 *   - `stream` is a hipStream_t (void* here so the header needs no HIP include);
 *     NULL is the legacy default stream.  Nothing here allocates, frees or
 *     synchronises: work is enqueued on `stream` and the call returns.  ONE exception, switched
 *     off by trl_offpolicy_set_chain(0): the one-launch form of trl_ddpg_learn_fused allocates 1 KB
 *     of device memory and a ring of pinned host memory at its first call on a stream, and
 *     waits for its own launch of 16 calls back before reusing that launch's ring slot (never in
 *     practice: that launch finished long before).  Its device-wide barrier needs every workgroup of the launch
 *     resident at once: the grid is sized from hipOccupancyMaxActiveBlocksPerMultiprocessor, every wait is bounded,
 *     and a wait that runs out (the card is shared and a workgroup never got a CU) aborts the launch -- NaN losses,
 *     no further update phase -- and is REPORTED: the next trl_ddpg_learn_fused call on that stream returns TRL_EHIP
 *     naming the phase, clears the barrier block and runs that stream's later steps as one launch per phase.
 *   - thread-compatible.  Process-wide state is limited to: the thread-local error text; the
 *     mutex-guarded jump-ahead polynomial cache of trl_randperm_prefix (a pure function of the
 *     skip length, so it never changes a result); the per-device address of the static scratch
 *     array of the no-allocation entry point at the end of this header; the speculation worker of
 *     trl_randperm_prefix_ahead (one helper thread, created on first use; a fork()ed child starts its own);
 *     the chain switch and per-stream chain state above (barrier block, table ring, the pinned fault word a
 *     timed-out barrier reports through).  Nothing reads the environment: tuning / ablation switches exist only
 *     in the -DTRL_TUNING diagnostic build (torchrl_amd/_build.py --tuning -> lib/libtrl_hip_tune.so, used by
 *     tools/ and by three tests' child processes); the Python layer's switches are attributes set in code
 *     (torchrl_amd/config.py), not environment variables.
 */
#ifndef TRL_HIP_H
#define TRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRL_OK 0
#define TRL_EINVAL (-1) /* bad argument (shape, alignment, null pointer, unsupported size) */
#define TRL_EHIP (-2)   /* a HIP runtime call or kernel launch failed */

#define TRL_MAX_COLS 8

/* storage dtype of a replay column / output dtype of a gathered column */
#define TRL_F32 0
#define TRL_U8 1

/* GAE modes */
#define TRL_GAE_REFERENCE 0 /* done ignored inside the trajectory (reference behaviour) */
#define TRL_GAE_MASKED 1    /* (1-done) masks bootstrap and carry at every step (extension) */

/* output layout of the GAE kernel */
#define TRL_LAYOUT_TN 0 /* [T, N] like the inputs */
#define TRL_LAYOUT_NT 1 /* [N, T] env-major: the order examples/ppo.py:61-64 concatenates in */
#define TRL_LAYOUT_BLOCKED 2 /* [N / E][t_stride][E]: E = 8 or 16 consecutive envs x all steps contiguous */

/* loss kinds for the TD kernel */
#define TRL_LOSS_MSE 0   /* reference: dqn_controller.py:77-78, ddpg_controller.py:112 */
#define TRL_LOSS_HUBER 1 /* north-star variant, no reference counterpart */

int trl_version(void);
const char *trl_last_error(void);

/* ------------------------------------------------------------------ replay ring (a-1)
 * The ring is ONE device byte buffer of `capacity` records, `stride` bytes each
 * (stride % 4 == 0).  Column c of a record lives at byte offset col_off[c] and is
 * col_bytes[c] bytes wide, stored as col_dtype[c] (TRL_F32 or TRL_U8).  Logical row i
 * (0 = oldest, the order torchrl/utils/storage.py:7-16 keeps by cat + slice) is the
 * physical record (head + i) % capacity.
 */
typedef struct {
  int32_t ncols;
  int32_t col_off[TRL_MAX_COLS];   /* byte offset inside a record, multiple of 4 */
  int32_t col_bytes[TRL_MAX_COLS]; /* bytes of the column inside a record          */
  int32_t col_dtype[TRL_MAX_COLS]; /* TRL_F32 / TRL_U8                              */
} trl_ring_layout;

/* Replaces TensorTupleDataset.extend -> truncated_cat (storage.py:7-16, 27-37): writes
 * k new rows at physical records (tail + j) % capacity, j in [0, k), where
 * tail = (head + count) % capacity.  src[c] is a dense device array [k, col_bytes[c]]
 * of the column's storage dtype.  Requires 0 < k <= capacity (the caller drops the
 * rows that would not survive, storage.py:12-14, and advances head/count itself). */
int trl_ring_push(void *ring, int64_t capacity, int64_t stride, const trl_ring_layout *layout,
                  int64_t tail, const void *const *src, int64_t k, void *stream);

/* Replaces TensorTupleDataset.__getitem__ with a LongTensor (storage.py:44-45, called
 * from examples/dqn.py:40-41, examples/ddpg.py:40-41): dst[c][b] = column c of logical
 * row idx[b].  idx is a device int64 array [B]; negative values wrap once (idx + count)
 * like torch indexing.  out_dtype[c] may be TRL_F32 for a TRL_U8 column (exact cast,
 * what torch.Tensor(uint8 ndarray) gives at storage.py:65-67); otherwise it must equal
 * the storage dtype.  dst[c] is dense [B, width]. */
int trl_gather_rows(const void *ring, int64_t capacity, int64_t stride,
                    const trl_ring_layout *layout, int64_t head, int64_t count,
                    const int64_t *idx, int64_t B, void *const *dst, const int32_t *out_dtype,
                    void *stream);

/* Host-drawn indices -> device: dst[i] = src_pinned[i], where src_pinned is PINNED (device-mapped) host
 * memory that the kernel reads directly.  The `.to(self.device)` of the sampled indices at
 * examples/dqn.py:40-41 / examples/ddpg.py:40-41, without a DMA-engine copy in the middle of the
 * learner's stream. */
int trl_stage_indices(const int64_t *src_pinned, int64_t *dst, int64_t n, void *stream);

/* Same as trl_gather_rows for the contiguous logical range [start, start+B) -- the
 * `buffer[:]` form of examples/a2c.py:41 and examples/ppo.py:50 -- with no index array. */
int trl_gather_range(const void *ring, int64_t capacity, int64_t stride,
                     const trl_ring_layout *layout, int64_t head, int64_t count, int64_t start,
                     int64_t B, void *const *dst, const int32_t *out_dtype, void *stream);

/* ------------------------------------------------------------------ sampler (a-1, host)
 * Replaces `torch.randperm(len(buffer))[:batch_size]` (examples/dqn.py:40,
 * examples/ddpg.py:40) bit for bit: torch's CPU randperm is a forward Fisher-Yates over
 * raw 32-bit MT19937 draws, z = draw % (n - i), swap(r[i], r[i + z]) for i in [0, n-1).
 * `mt` is the 624-word MT19937 state, `*left`/`*next` torch's position counters (the
 * fields of at::CPUGeneratorImpl's legacy state).  Writes the first B entries to out_h
 * (host int64) and advances the state by exactly n-1 draws so that the stream stays in
 * lock-step with a reference process that called torch.randperm(n).  O(n) time, O(B) memory. */
int trl_mt19937_seed(uint32_t *mt /*[624]*/, int32_t *left, int32_t *next, uint64_t seed);
int trl_randperm_prefix(uint32_t *mt /*[624]*/, int32_t *left, int32_t *next, int64_t n, int64_t B,
                        int64_t *out_h);
/* The same result by the plain block-by-block skip only (never the GF(2) jump-ahead, which
 * trl_randperm_prefix switches to for rings of more than ~2.6 M rows): the yardstick the jump is
 * tested against, and the form for callers that want no process-wide cache touched. */
int trl_randperm_prefix_direct(uint32_t *mt /*[624]*/, int32_t *left, int32_t *next, int64_t n, int64_t B,
                               int64_t *out_h);

/* trl_randperm_prefix with the NEXT draw computed ahead of time, for the learner that samples every step from a
 * full ring (`torch.randperm(len(buffer))[:batch_size]` once per step, examples/dqn.py:40, examples/ddpg.py:40).
 * After returning this call's indices it starts, on a helper thread of the library, the draw of (next_n, B) from
 * the state it leaves in mt / left / next (next_n <= 0: none).  The next call compares the state it is given with
 * the one speculated from and, when they are equal (nobody else drew from the generator) and n == that next_n,
 * copies the finished result instead of regenerating ~n / 624 MT19937 blocks on the caller's thread; anything
 * else is computed as trl_randperm_prefix computes it.  Indices and final state are identical either way.
 * *hit (may be NULL) is set to 1 when the speculated result was used.  One speculation per process. */
int trl_randperm_prefix_ahead(uint32_t *mt /*[624]*/, int32_t *left, int32_t *next, int64_t n, int64_t B,
                              int64_t *out_h, int64_t next_n, int32_t *hit);

/* Throughput-mode sampler, on the device: B distinct indices uniform over [0, n) written to
 * idx_d (device int64 [B]), ready for trl_gather_rows -- the same distribution as
 * `torch.randperm(n)[:B]` (examples/dqn.py:40) but NOT the same indices (bit-exact indices are
 * trl_randperm_prefix above, O(n) host work per call; this is O(B) device work and needs no
 * host round trip, so a whole sample -> gather -> learn step can be queued or graph-replayed).
 * Counter-based (Philox-4x32-10 keyed by `seed`); *call_counter_d (device u64) numbers the calls and
 * is incremented by the launch; the result depends on (seed, call number, n, B) only.
 * 0 <= B <= min(n, 4096). */
int trl_sample_indices(int64_t n, int64_t B, uint64_t seed, uint64_t *call_counter_d, int64_t *idx_d,
                       void *stream);

/* ------------------------------------------------------------------ returns / GAE (a-3)
 * Replaces the Python reverse loop of A2CController.compute_return
 * (a2c_controller.py:38-43) and PPOController.compute_return (ppo_controller.py:41-46)
 * for N envs at once.  reward, value: [T, N] f32 (env index contiguous); bootstrap: [N]
 * = V(next_obs[T-1]) or 0 where done[T-1] (a2c_controller.py:32-36); done: [T, N] f32,
 * read only when mode == TRL_GAE_MASKED (may be NULL otherwise).
 *   delta = (r[t] + gamma * v[t+1]) - v[t];  gae = delta + gamma_lambda * gae;
 *   returns[t] = gae + v[t];  adv[t] = returns[t] - v[t]          (examples/ppo.py:54)
 * with gamma -> (float)gamma and gamma_lambda -> (float)(gamma * lmbda) (the product is
 * formed in double first, as Python does), every operation rounded to f32 separately (no
 * FMA contraction), so the result is bit-identical to the reference loop.  adv may be NULL.
 * Hyper-parameters are doubles throughout this header because the reference holds them
 * as Python floats and torch narrows them to f32 at the point of use. */
int trl_gae_reverse_scan(const float *reward, const float *value, const float *done,
                         const float *bootstrap, int64_t T, int64_t N, double gamma,
                         double lmbda, int32_t mode, int32_t out_layout, float *returns,
                         float *adv, void *stream);

/* The same scan over a rollout kept in the BLOCKED layout: element (t, env) of reward / value / done
 * is at ((env / E) * in_t_stride + t) * E + env % E floats, E = block_envs (8 or 16), N % E == 0,
 * in_t_stride >= T the rows allocated per env group (0 = T).  A workgroup's E envs x T steps are then
 * one contiguous run per plane (flat-stream geometry) instead of a 4 * E-byte segment of every [T, N]
 * row.  This is the layout torchrl_amd's RolloutBuffer(layout='blocked') appends env steps in and the
 * forward-only value kernel writes its values in; there is no reference counterpart (the reference
 * keeps one [T, .] buffer per env, examples/ppo.py:28-35).  Arithmetic and results are those of
 * trl_gae_reverse_scan element for element.  out_layout: TRL_LAYOUT_BLOCKED (same indexing with
 * out_t_stride, 0 = T) or TRL_LAYOUT_NT (env-major [N, T], the reference's minibatch order).
 * bootstrap stays [N].  Every plane must be 16-byte aligned. */
int trl_gae_reverse_scan_blocked(const float *reward, const float *value, const float *done,
                                 const float *bootstrap, int64_t T, int64_t N, int32_t block_envs,
                                 int64_t in_t_stride, double gamma, double lmbda, int32_t mode,
                                 int32_t out_layout, int64_t out_t_stride, float *returns, float *adv,
                                 void *stream);

/* ------------------------------------------------------------------ fused losses (a-4)
 * Workspace: every loss kernel reduces over the batch in two fixed-order stages, so
 * results are run-to-run reproducible.  `ws` is a caller-owned device scratch of at
 * least trl_loss_workspace_bytes() bytes. */
int64_t trl_loss_workspace_bytes(void);

/* DQN / Double-DQN TD target + loss + dL/dQ (dqn_controller.py:64-78).
 *   q [B, A] online Q(s, .); q_next_target [B, A]; q_next_online [B, A] or NULL
 *   (non-NULL selects the Double-DQN argmax of dqn_controller.py:68-69);
 *   action/reward/done [B] f32 (the reference stores them as f32, storage.py:65-67).
 *   y = reward + gamma * nq * (1 - done); loss = mean(l(y - q[b, a_b]))
 *   loss_out[0] = loss; dq [B, A] = dloss/dq (zeros off the taken action), may be NULL. */
int trl_td_loss(const float *q, const float *q_next_target, const float *q_next_online,
                const float *action, const float *reward, const float *done, int64_t B, int32_t A,
                double gamma, int32_t kind, double huber_delta, float *loss_out, float *dq, void *ws,
                void *stream);

/* DDPG critic target + MSE + dL/dcurrent_q (ddpg_controller.py:107-112):
 *   y = reward + (1 - done) * gamma * next_q;  loss = mean(l(cur_q - y)). */
int trl_ddpg_critic_loss(const float *cur_q, const float *next_q, const float *reward,
                         const float *done, int64_t B, double gamma, int32_t kind,
                         double huber_delta, float *loss_out, float *dcur_q, void *ws, void *stream);

/* PPO clipped surrogate + value MSE + Normal entropy (ppo_controller.py:55-68) and the
 * backward seeds of  loss = actor + alpha * critic - beta * entropy.
 *   mean [B, A], log_std [A], value [B], action [B, A], old_logp [B, A], adv [B],
 *   returns [B].  losses_out[0..2] = actor, critic, entropy; losses_out[3] = loss.
 *   dmean [B, A], dvalue [B], dlog_std [A] (reduced over the batch); NULL skips all three. */
int trl_ppo_loss(const float *mean, const float *log_std, const float *value, const float *action,
                 const float *old_logp, const float *adv, const float *returns, int64_t B, int32_t A,
                 double clip, double alpha, double beta, float *losses_out, float *dmean,
                 float *dlog_std, float *dvalue, void *ws, void *stream);

/* Normal log-prob per action dimension, the `dist.log_prob(action)` of
 * ppo_controller.py:48 (shape [B, A], not summed). */
int trl_normal_logp(const float *mean, const float *log_std, const float *action, int64_t B,
                    int32_t A, float *logp, void *stream);

/* A2C loss (a2c_controller.py:48-59): probs [B, A] is the actor's softmax output,
 * value/action/returns [B].  losses_out as for PPO; dprobs [B, A], dvalue [B]. */
int trl_a2c_loss(const float *probs, const float *value, const float *action, const float *returns,
                 int64_t B, int32_t A, double alpha, double beta, float *losses_out, float *dprobs,
                 float *dvalue, void *ws, void *stream);

/* ------------------------------------------------------------------ update (a-4)
 * One pass over a flat f32 parameter buffer: optional scale (1/world_size after the
 * all-reduce), optional elementwise clamp (torch.nn.utils.clip_grad_value_,
 * ppo_controller.py:72-73), then torch.optim.Adam's default update
 * (dqn_controller.py:39, ppo_controller.py:15: betas (0.9, 0.999), eps 1e-8, no decay)
 * in torch's single-tensor op order.  `step` is the 1-based step count.  The scaled /
 * clamped gradient is written back to `grad` (clip_grad_value_ works in place). */
int trl_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                  double lr, double beta1, double beta2, double eps, int64_t step, double grad_scale,
                  double clip_value /* <= 0: no clamp */, void *stream);

/* The same update with its step-dependent scalars kept ON THE DEVICE, for callers that replay
 * the learner step from a captured HIP graph (kernel arguments are frozen at capture, the step
 * count is not).  `state` is a device struct: the launch increments state->step, derives the
 * bias corrections from it in double precision and applies the update; the host writes the
 * struct once (and again whenever it changes lr / scale / clip). */
typedef struct {
  int64_t step;      /* steps taken so far; incremented by every launch */
  double lr, beta1, beta2, eps;
  double grad_scale; /* 1 = none */
  double clip_value; /* <= 0 = none */
  float derived[8];  /* scratch of the launch (the per-step scalars) */
} trl_adam_state;

int trl_adam_step_dev(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                      trl_adam_state *state, void *stream);

/* target <- tau * source + (1 - tau) * target (ddpg_controller.py:42-57); tau == 1 is
 * the hard sync of dqn_controller.py:85-86. */
int trl_polyak(float *target, const float *source, int64_t n, double tau, void *stream);

/* ------------------------------------------------------------------ MLP layers (a-2)
 * One Linear (+ ReLU) layer of the small policy / value networks (torchrl/contrib/models.py:
 * QNet :14-20, DDPGActorNet :37-44, DDPGCriticNet :66-72, A2CNet :92-103, ActorCriticNet
 * :120-130) on f32 MFMA, for the large-batch learner step where the BLAS picks are poor.
 *   fwd:  y[B, N] = act(x[B, K] @ w[N, K]^T + b[N]);   act: 0 = none, 1 = ReLU.  N <= 256.
 *   bwd:  g = dy (* (y > 0) for ReLU);  dx[B, K] = g @ w (dx may be NULL);
 *         dw[N, K] = g^T @ x;  db[N] = column sums of g   (both OVERWRITTEN, not accumulated).
 * `ws` is a caller-owned device scratch of trl_linear_bwd_workspace_bytes(B, K, N) bytes; the
 * weight-gradient partials are summed in a fixed order (reproducible, no float atomics). */
int trl_linear_fwd(const float *x, const float *w, const float *b, int64_t B, int32_t K, int32_t N,
                   int32_t act, float *y, void *stream);
int64_t trl_linear_bwd_workspace_bytes(int64_t B, int32_t K, int32_t N);
int trl_linear_bwd(const float *dy, const float *y, const float *x, const float *w, int64_t B, int32_t K,
                   int32_t N, int32_t act, float *dx, float *dw, float *db, void *ws, void *stream);

/* The same layer on uint8 input (frames kept uint8 in the replay ring): y = act(float(x) @ w^T + b),
 * x [B, K] uint8.  Replaces `QNet.net[0]` of torchrl/contrib/models.py:12-18 as DQNController.learn
 * calls it on stored frames (dqn_controller.py:64-72), which the reference stores as float32
 * (torchrl/utils/storage.py:65-67): uint8 -> float32 is exact, so the values are the same and the
 * batch never exists as float32.  N % 128 == 0, K % 32 == 0; `ws` is a caller-owned device scratch of
 * trl_linear_u8_fwd_workspace_bytes(B, N, K) bytes (split-K partial tiles, summed in a fixed order).
 * trl_linear_u8_wgrad: dw[N, K] = g[B, N]^T @ float(x)  (OVERWRITTEN; B % 32 == 0, K % 4 == 0).
 * Arithmetic: the uint8 operand is exactly a bf16 and the f32 operand goes in as its three exact bf16 parts, so every
 * partial product is exact and the sums are f32 (the MFMA's accumulators): f32 results, no reduced precision. */
int64_t trl_linear_u8_fwd_workspace_bytes(int64_t B, int32_t N, int32_t K);
int trl_linear_u8_fwd(const uint8_t *x, const float *w, const float *b, int64_t B, int32_t K, int32_t N,
                      int32_t act, float *y, void *ws, void *stream);
int trl_linear_u8_wgrad(const float *g, const uint8_t *x, int64_t B, int32_t K, int32_t N, float *dw,
                        void *stream);

/* A whole MLP (Linear -> ReLU -> ... -> Linear, 1..3 Linear layers, every fan-in / fan-out
 * <= 64, e.g. the "MLP 64x64" of the PPO benchmark) in ONE forward launch and ONE backward
 * launch (+ a small reduce): the activations go from layer to layer through LDS.
 *   dims[0] = fan-in of the net, dims[l+1] = fan-out of layer l; w[l] is [dims[l+1], dims[l]].
 *   fwd: acts[l] ([B, dims[l+1]], caller-owned) receives the output of layer l (post-ReLU for the
 *        hidden layers, plain for the last): acts[depth-1] is the network output, the others
 *        are what the backward needs.
 *   bwd: dy [B, dims[depth]] is the gradient w.r.t. the network output; `grads` receives the
 *        parameter gradients densely in the order dW_0, db_0, dW_1, db_1, ... (OVERWRITTEN).
 *        Alternatively `param_grads` (2 * depth pointers in that same order, e.g. the .grad
 *        views of a flat gradient buffer) receives each parameter's gradient directly; pass
 *        one of the two, the other NULL.
 *        No gradient w.r.t. x is produced (the nets' inputs are observations).
 * `ws`: trl_mlp_bwd_workspace_bytes(net, B) bytes of device scratch. */
typedef struct {
  int32_t depth;
  int32_t dims[4];
  const float *w[3];
  const float *b[3];
} trl_mlp_desc;

int trl_mlp_fwd(const trl_mlp_desc *net, const float *x, int64_t B, float *const *acts, void *stream);
int64_t trl_mlp_bwd_workspace_bytes(const trl_mlp_desc *net, int64_t B);
int trl_mlp_bwd(const trl_mlp_desc *net, const float *x, const float *const *acts, const float *dy, int64_t B,
                float *grads, float *const *param_grads, void *ws, void *stream);

/* The same for a group of G <= 4 independent networks that see the same batch size (the critic
 * and the actor of an actor-critic; online and target Q-networks) in one launch pair:
 * xs[g], dys[g]; acts[3 * g + l]; grads[g] or param_grads[6 * g + j]; the workspace is the sum of
 * the members' trl_mlp_bwd_workspace_bytes(). */
int trl_mlp_fwd_group(const trl_mlp_desc *nets, int32_t G, const float *const *xs, int64_t B, float *const *acts,
                      void *stream);
int trl_mlp_bwd_group(const trl_mlp_desc *nets, int32_t G, const float *const *xs, const float *const *acts,
                      const float *const *dys, int64_t B, float *const *grads, float *const *param_grads, void *ws,
                      void *stream);

/* ------------------------------------------------------------------ fused PPO learner step (a-2 + a-4)
 * The device work of PPOController.learn up to the gradients (ppo_controller.py:55-70: forward of
 * ActorCriticNet's critic and actor, models.py:120-141; the clipped-surrogate / value / entropy
 * loss; loss.backward()) for networks whose layers are all at most 64 wide, as ONE launch plus a
 * small fixed-order reduction: a wavefront carries 32 samples from the observation to the
 * parameter-gradient partials with the activations in registers (csrc/trl_learn.hip).
 *   critic, actor: depth 2 or 3 (one or two hidden layers, ReLU between), same depth and fan-in;
 *   a single hidden layer over a fan-in of at most 9 may be up to 256 wide with A <= 2 (walked as
 *   64-wide blocks inside the wave; with A = 2 and a fan-in of 8 or 9: up to 128 wide), every other width at most 64;
 *   critic fan-out 1, actor fan-out A (1 <= A <= 4);  log_std [A];  obs [B, fan-in];
 *   action, old_logp [B, A];  returns, adv [B].
 *   losses_out[0..3] = actor, critic, entropy, actor + alpha * critic - beta * entropy.
 *   critic_grads / actor_grads: 2 * depth destinations (dW_0, db_0, dW_1, ...), e.g. the .grad
 *   views of a flat gradient buffer, OVERWRITTEN with d loss / d parameter;  dlog_std [A] likewise.
 * `ws`: trl_ppo_learn_workspace_bytes() bytes of device scratch.
 * `adam` (may be NULL): also apply clip_grad_value_ + Adam (ppo_controller.py:72-74; the update of
 * trl_adam_step, same arithmetic) to every parameter inside the reduce launch, for the single-rank
 * learner.  It names four parallel flat f32 buffers of `n` floats; every gradient destination above
 * must lie inside `grad`, and the parameter / moments of a gradient element are the elements at the
 * same offset of `param` / `exp_avg` / `exp_avg_sq` (the layout of one flat buffer per quantity).
 * With an all-reduce between backward and update, pass NULL and call trl_adam_step afterwards. */
typedef struct {
  float *param, *grad, *exp_avg, *exp_avg_sq;
  int64_t n;
  double lr, beta1, beta2, eps;
  int64_t step;      /* 1-based count of this update */
  double grad_scale; /* 1 = none */
  double clip_value; /* <= 0 = none */
  /* Graph replay (trl_ddpg_learn_fused, trl_dqn_learn_fused[_u8] only; NULL otherwise): the step count
   * and hyper-parameters live in this DEVICE struct instead of the fields above.  The chain's first
   * launch increments dev_state->step and derives the bias corrections on the device (as
   * trl_adam_step_dev does), the update launches read them from there, so every kernel argument of
   * the chain is the same from step to step and the chain can be captured once and replayed. */
  trl_adam_state *dev_state;
} trl_adam_args;

int64_t trl_ppo_learn_workspace_bytes(const trl_mlp_desc *critic, const trl_mlp_desc *actor, int64_t B);
int trl_ppo_learn_fused(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                        const float *obs, const float *action, const float *returns, const float *old_logp,
                        const float *adv, int64_t B, int32_t A, double clip, double alpha, double beta,
                        float *losses_out, float *const *critic_grads, float *const *actor_grads,
                        float *dlog_std, const trl_adam_args *adam, void *ws, void *stream);

/* The same for A2CController.learn (a2c_controller.py:47-69) with A2CNet's critic and softmax
 * actor (models.py:84-109; `actor` describes the Linear layers, the softmax is applied inside):
 *   action [B] (the index, as f32), returns [B], adv [B] = returns - V(obs) DETACHED -- computed
 *   by the caller with trl_ppo_forward_fused(critic), which evaluates V bit-identically to this
 *   kernel's own critic pass.  losses_out as above (actor, critic, entropy, total). */
int trl_a2c_learn_fused(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *obs,
                        const float *action, const float *returns, const float *adv, int64_t B, int32_t A,
                        double alpha, double beta, float *losses_out, float *const *critic_grads,
                        float *const *actor_grads, const trl_adam_args *adam, void *ws, void *stream);

/* trl_ppo_learn_fused / trl_a2c_learn_fused with the step's 64 x 64 products (layer 1 forward, its weight gradient, its
 * input gradient: ppo_controller.py:55-70 / a2c_controller.py:47-66 through models.py:112-147's hidden layers) on the
 * bf16 matrix instruction.  Both operands of a product are split EXACTLY into three bf16 parts (8 + 8 + 8 significant
 * bits) and six of the nine partial products are kept, each exact, summed in f32 accumulators: the result is within
 * 1.2e-7 x sum |a_k| |b_k| of the exact dot product plus the f32 rounding of the sum -- the accuracy class of an f32 dot
 * product in another summation order ("f32-equivalent"), NOT the bits of trl_ppo_learn_fused, which stays the parity
 * mode.  Same arguments, workspace and outputs.  Shapes: the role-split kernel's only -- depth 3, hidden widths <= 64,
 * at most 4 observations, A <= 2 (any B; the Python layer
 * takes it from 32768 rows on, where the role-split kernel is the exact path's choice too); anything else is TRL_EINVAL.  Everything
 * outside the three products (layer 0, last layer, loss, layer-0 gradient, reduction, clip, Adam) is the exact entry
 * point's arithmetic.  No counterpart in the reference (which computes in f32 on the host). */
int trl_ppo_learn_fused_bx6(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                            const float *obs, const float *action, const float *returns, const float *old_logp,
                            const float *adv, int64_t B, int32_t A, double clip, double alpha, double beta,
                            float *losses_out, float *const *critic_grads, float *const *actor_grads,
                            float *dlog_std, const trl_adam_args *adam, void *ws, void *stream);
int trl_a2c_learn_fused_bx6(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *obs,
                            const float *action, const float *returns, const float *adv, int64_t B, int32_t A,
                            double alpha, double beta, float *losses_out, float *const *critic_grads,
                            float *const *actor_grads, const trl_adam_args *adam, void *ws, void *stream);

/* trl_ppo_forward_fused with layer 1's product as in the two entry points above (same staging, split and accumulation
 * order: a row's value / mean / log-prob is the bit pattern THEIR forward computes for it).  Same shapes as they take. */
int trl_ppo_forward_fused_bx6(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                              const float *obs, const float *action, int64_t B, int32_t A, float *value,
                              float *mean, float *logp, void *stream);

/* Test surface of the two entry points above: y[64][n] = w[64][64] x[64][n] (row-major device arrays) computed with
 * exactly their operand split, partial products and accumulation order, so that the product's error can be bounded
 * against float64 on adversarial inputs without a network around it.  No counterpart in the reference. */
int trl_bx6_product_probe(const float *w, const float *x, float *y, int64_t n, void *stream);

/* The forward alone of the same networks over B rows, only the outputs written: the value /
 * mean / log-prob pass of PPOController.compute_return (ppo_controller.py:34, 48) and act().
 *   value [B] (needs critic), mean [B, A], logp [B, A] = Normal(mean, exp(log_std)).log_prob(action)
 *   per dimension (need actor; logp also action [B, A] and log_std [A]); any output may be NULL.
 * Depth 2 or 3, widths <= 64, 1 <= A <= 4; one hidden layer (depth 2) over at most 9 observations
 * may be up to 256 wide (the width ppo_controller.py:14 builds).
 * A row's value / mean / log-prob is the bit pattern trl_ppo_learn_fused / trl_a2c_learn_fused compute for
 * it on the same networks (a first-epoch PPO ratio is exactly 1).  Pass the actor's action size in A even
 * when only `value` is asked for: it selects the kernel variant whose forward is the learner's. */
int trl_ppo_forward_fused(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                          const float *obs, const float *action, int64_t B, int32_t A, float *value, float *mean,
                          float *logp, void *stream);

/* The reference's PPO minibatch loop in ONE launch: `for epoch in range(n_epochs): for minibatch in loader:
 * controller.learn(*minibatch)` (examples/ppo.py:66-73 over ppo_controller.py:53-80), i.e. K = n_epochs *
 * ceil(n_rows / minibatch) consecutive learn() steps -- forward, clipped-surrogate / value / entropy loss, backward,
 * clip_grad_value_, Adam -- each on the weights the step before it left.  The minibatches are the consecutive row
 * ranges [j * minibatch, (j + 1) * minibatch) of the columns, the last one of an epoch short if n_rows is not a
 * multiple (DataLoader's shuffle=False, drop_last=False, ppo.py:68).  At the reference's minibatch sizes (32 / 64) a
 * step is a few microseconds of dependent arithmetic, so the loop runs as two persistent workgroups (the critic's
 * and the actor's chains of updates are independent) with parameters and Adam moments resident on the CU between
 * updates (csrc/trl_epoch.hip).
 *   critic, actor: depth 2 (one hidden layer, a multiple of 16 up to 256) or depth 3 (two hidden layers, multiples of
 *   16 up to 64), same fan-in <= 15; critic fan-out 1, actor fan-out A (1 <= A <= 4); log_std [A].  All parameters
 *   must lie inside adam->param (the flat buffer); moments at the same offsets of adam->exp_avg / exp_avg_sq.
 *   obs [n_rows, fan-in]; action, old_logp [n_rows, A]; returns, adv [n_rows].  1 <= minibatch <= 4096.
 *   adam: required; adam->step = the 1-based count of the FIRST update (the caller advances its count by K);
 *   clip_value as in trl_adam_step; grad_scale must be 1, dev_state NULL (single rank, eager).
 *   losses_out [K][4] = actor, critic, entropy, actor + alpha * critic - beta * entropy of every step, in order.
 *   On return (stream order) param / exp_avg / exp_avg_sq hold the state after K updates and adam->grad the clipped
 *   gradients of the last one, as K calls of trl_ppo_learn_fused(..., adam) would leave them -- the same arithmetic
 *   (exact f32 products, torch's Adam op order) up to the order of the f32 sums.
 * `ws`: trl_ppo_learn_epoch_workspace_bytes() bytes of device scratch (the K steps' bias corrections). */
int64_t trl_ppo_learn_epoch_workspace_bytes(int64_t n_rows, int64_t minibatch, int32_t n_epochs);
/* 1 when trl_ppo_learn_epoch takes networks of these shapes (the limits above and the LDS the two workgroups need: a
 * 256-wide hidden layer with four actions does not fit), 0 otherwise.  Only depth and dims of the descriptors are read. */
int trl_ppo_learn_epoch_supported(const trl_mlp_desc *critic, const trl_mlp_desc *actor, int32_t A);
int trl_ppo_learn_epoch(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std, const float *obs,
                        const float *action, const float *returns, const float *old_logp, const float *adv,
                        int64_t n_rows, int64_t minibatch, int32_t n_epochs, int32_t A, double clip, double alpha,
                        double beta, float *losses_out, const trl_adam_args *adam, void *ws, void *stream);

/* ONE PPOController.learn step (ppo_controller.py:53-80) over a batch of any size through the tile pipeline of
 * trl_ppo_learn_epoch, spread over the chip: slabs of the batch x two kinds of workgroup (critic, actor), each leaving
 * its partial gradient sums, then one reduce launch that adds the slabs in order, writes every gradient to its place in
 * adam->grad and (update != 0) applies clip_grad_value_ + Adam as trl_adam_step does.  128 slabs and both kinds in one
 * launch; from 32,768 rows on 256 slabs and one launch per kind (the actor's workgroups take longer than the critic's:
 * in one launch half the chip would wait for the other).  The general whole-step path: it takes every shape
 * trl_ppo_learn_epoch takes (fan-in <= 15, 1..4 actions, one hidden layer up to 256 wide or two up to 64) and, with ONE
 * hidden layer, fan-in up to 31 and up to six actions (ActorCriticNet(17, 6, 256): a second k round in the layer-0
 * products) -- in particular the shapes trl_ppo_learn_fused's register-chained kernels do not take (more than two
 * actions, a 256-wide layer over more than 9 observations) and that would otherwise run layer by layer.  Arguments as
 * trl_ppo_learn_fused; losses_out[0..3]; adam->step = the 1-based count of this update when update != 0.
 * `ws`: trl_ppo_learn_tiles_workspace_bytes(adam->n, B) bytes. */
int64_t trl_ppo_learn_tiles_workspace_bytes(int64_t n_flat, int64_t B);
int trl_ppo_learn_tiles(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std, const float *obs,
                        const float *action, const float *returns, const float *old_logp, const float *adv, int64_t B,
                        int32_t A, double clip, double alpha, double beta, float *losses_out, const trl_adam_args *adam,
                        int32_t update, void *ws, void *stream);

/* The same for ONE A2CController.learn step (a2c_controller.py:47-69) with A2CNet's critic and softmax actor
 * (models.py:84-109): arguments as trl_a2c_learn_fused (action [B] = the index as f32), 2 <= A <= 6, the shapes above --
 * A2CNet(8, 4, 256), the reference's default width with four actions over eight observations, is one of them.
 * adv [B] = returns - V(obs) detached (a2c_controller.py:49), computed by the caller -- or NULL: then the critic's
 * workgroups, launched ahead of the actor's, compute it from their own forward pass (no separate value pass).
 * Same workspace. */
int trl_a2c_learn_tiles(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *obs, const float *action,
                        const float *returns, const float *adv, int64_t B, int32_t A, double alpha, double beta,
                        float *losses_out, const trl_adam_args *adam, int32_t update, void *ws, void *stream);

/* The forward passes of compute_return (ppo_controller.py:32-51: values = critic(obs), dist = Normal(actor(obs),
 * exp(log_std)), log_probs = dist.log_prob(action); a2c_controller.py:30: values) through the same tile pipeline, for
 * the shapes trl_ppo_forward_fused does not take (those of trl_ppo_learn_tiles): the tiles stop behind the loss phase,
 * whose waves write value_out [B], mean_out [B, A] (or NULL) and logp_out [B, A] (or NULL; needs `action` [B, A]).
 * actor == NULL: the critic alone (A2CNet's values, the bootstrap values).  No workspace. */
int trl_ac_forward_tiles(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std, const float *obs,
                         const float *action, int64_t B, int32_t A, float *value_out, float *mean_out, float *logp_out,
                         void *stream);

/* ------------------------------------------------------------------ fused off-policy learner steps (a-2 + a-4)
 * DDPGController.learn (ddpg_controller.py:101-128) as 10 phases of one tile kernel (csrc/trl_offpolicy.hip;
 * one launch, or one per phase: trl_offpolicy_set_chain): the five
 * forward passes (actor(obs), critic(obs, actor(obs)), critic(obs, action), target_critic(next_obs,
 * target_actor(next_obs)); models.py:27-81), y = reward + (1 - done) * gamma * target_q, the two
 * losses, the backward of -mean Q through the critic into the actor's parameters only (the reference
 * discards what that pass leaves in the critic, :116-118) and of the critic loss into the critic's,
 * both Adam steps, and the Polyak averages of both targets (:42-57, :120-121) on the updated weights.
 *   Networks: two hidden ReLU layers (depth 3), hidden widths multiples of 16 up to 1024;
 *   actor obs -> A (tanh), critic [obs | action] -> 1; targets shaped like their online nets.
 *   obs, next_obs [B, S]; action [B, A]; reward, done [B]; B a multiple of 16, <= 4096.
 *   losses_out[0] = actor loss, [1] = critic loss.
 *   actor_opt / critic_opt: the flat buffers of each optimizer (see trl_adam_args); the networks'
 *   parameters must be views into `param`, their gradients are written at the same offsets of `grad`.
 *   step > 0: Adam is applied where each gradient tile is produced, then
 *   target <- tau * param + (1 - tau) * target into target_*_flat (same layout as `param`).
 *   step == 0 (both): gradients only -- for an all-reduce between backward and update.
 * `ws`: trl_ddpg_learn_workspace_bytes() bytes of device scratch, 16 B aligned. */
int64_t trl_ddpg_learn_workspace_bytes(const trl_mlp_desc *actor, const trl_mlp_desc *critic, int64_t B);
int trl_ddpg_learn_fused(const trl_mlp_desc *actor, const trl_mlp_desc *critic, const trl_mlp_desc *target_actor,
                         const trl_mlp_desc *target_critic, const float *obs, const float *action,
                         const float *reward, const float *next_obs, const float *done, int64_t B, double gamma,
                         int32_t kind, double huber_delta, float *losses_out, const trl_adam_args *actor_opt,
                         const trl_adam_args *critic_opt, float *target_actor_flat, float *target_critic_flat,
                         double tau, void *ws, void *stream);

/* No counterpart in the reference (a switch for tests and measurements).  trl_ddpg_learn_fused issues its chain of
 * phases as ONE launch (every workgroup walks the phases, a device-wide barrier between two of them; data handed
 * from phase to phase goes through agent-scope accesses) unless the optimizer state lives on the device (graph
 * capture, or `stream` is hipStreamPerThread).  on = 0: one launch per phase, as before; the two forms give the same
 * bits.  The one-launch form keeps, per (device, stream), 1 KB of device memory and a small ring of pinned host memory
 * for the phase tables, allocated at the first call on that stream and kept until the process ends (launches on one
 * stream run one after the other, so they share the barrier block; the count is bounded by the number of streams,
 * not of learners).  Two host threads must not call it on the SAME stream at the same time.  Process-wide; default 1.
 * on = 2: trl_dqn_learn_fused_u8[_ring] also issues the four small steps between its first layer's forward and weight
 * gradient (layer 1, layer 2, the TD loss as a one-workgroup member of the chain, the input gradients) as one launch:
 * same bits for B <= 256 (beyond: the loss value is another fixed-order sum), measured SLOWER than the four launches
 * (0.127 - 0.129 against 0.122 - 0.125 ms per step at B = 256 on one box: that step is device-bound), hence not part of the default. */
void trl_offpolicy_set_chain(int32_t on);

/* DQNController.learn up to the optimizer step (dqn_controller.py:61-83; the periodic hard sync
 * of :85-86 stays trl_polyak with tau = 1) as 6 launches (the uint8 variant below: 9): q_net(obs), target_net(next_obs) and, for
 * Double-DQN, q_net(next_obs) (QNet, models.py:6-24); the TD target and loss (trl_td_loss); the
 * backward into q_net; Adam.  Same network / batch limits as above; action [B] holds the index as
 * f32.  loss_out[0] = td loss.  opt->step == 0: gradients only. */
int64_t trl_dqn_learn_workspace_bytes(const trl_mlp_desc *q_net, int64_t B);
int trl_dqn_learn_fused(const trl_mlp_desc *q_net, const trl_mlp_desc *target_net, const float *obs,
                        const float *action, const float *reward, const float *next_obs, const float *done,
                        int64_t B, double gamma, int32_t double_dqn, int32_t kind, double huber_delta,
                        float *loss_out, const trl_adam_args *opt, void *ws, void *stream);

/* The same step on uint8 frames: obs / next_obs are uint8 [B, dims[0]] as the replay ring stores them
 * (the reference holds the same values as float32, torchrl/utils/storage.py:65-67).  Layer 0 of all
 * passes (dqn_controller.py:64-72) is one split-K launch + one reduce launch (trl_linear_u8_fwd's
 * kernels), its weight gradient one launch and its Adam step one streaming launch over the layer's slice of
 * the flat buffers: 9 launches per step.  dims[0] % 32
 * == 0 (any width up to 2^20), dims[1] % 128 == 0, B a multiple of 32. */
int64_t trl_dqn_learn_u8_workspace_bytes(const trl_mlp_desc *q_net, int64_t B);
int trl_dqn_learn_fused_u8(const trl_mlp_desc *q_net, const trl_mlp_desc *target_net, const uint8_t *obs,
                           const float *action, const float *reward, const uint8_t *next_obs, const float *done,
                           int64_t B, double gamma, int32_t double_dqn, int32_t kind, double huber_delta,
                           float *loss_out, const trl_adam_args *opt, void *ws, void *stream);

/* The same step with the sampled frames read IN PLACE: `[b.to(device) for b in buffer[b_idx]]` + `controller.learn(...)`
 * (examples/dqn.py:40-42) without the copy -- the first layer of all passes and its weight gradient read row b's two
 * uint8 frames at their place in the replay ring (record (head + idx[b]) % capacity), so the 2 x 28 KB per sampled
 * transition move once, as the layer's operand fetch, instead of being gathered into a dense batch first and read
 * again.  A small prologue launch resolves the records and gathers the 12 B action / reward / done tail.
 *   ring / capacity / stride / layout / head / count: the ring as trl_gather_rows takes it; the layout must be a
 *   Transition (storage.py:51-57): columns obs, action, reward, next_obs, done with obs / next_obs stored TRL_U8,
 *   dims[0] bytes wide at 16 B-aligned offsets, the other three one TRL_F32 each; stride % 16 == 0; capacity < 2^31.
 *   idx: int64 [B] logical rows (0 = oldest), device memory or pinned host memory the device can read; values are
 *   wrapped / clamped as trl_gather_rows does.  Everything else as trl_dqn_learn_fused_u8; same results bit for bit as
 *   trl_gather_rows + trl_dqn_learn_fused_u8 on the same indices.
 * `ws`: trl_dqn_learn_u8_ring_workspace_bytes() bytes. */
int64_t trl_dqn_learn_u8_ring_workspace_bytes(const trl_mlp_desc *q_net, int64_t B);
int trl_dqn_learn_fused_u8_ring(const trl_mlp_desc *q_net, const trl_mlp_desc *target_net, const void *ring,
                                int64_t capacity, int64_t stride, const trl_ring_layout *layout, int64_t head,
                                int64_t count, const int64_t *idx, int64_t B, double gamma, int32_t double_dqn,
                                int32_t kind, double huber_delta, float *loss_out, const trl_adam_args *opt, void *ws,
                                void *stream);

/* ------------------------------------------------------------------ env-step side: rollout push, exploration ("next" rows f2-f4)
 * trl_rollout_push: one env step's transitions -> the on-policy rollout columns, ONE launch.  Replaces the per-env
 * `buffer.extend([transition])` of examples/ppo.py:28-35 / examples/a2c.py:26-28 (one torch.cat per column per env
 * per step, torchrl/utils/storage.py:7-16).  rows_pinned: PINNED host memory [N, sum(width)] float32, env n's
 * packed row = its columns side by side; the kernel reads it directly (no DMA-engine copy in the stream).
 * Column i of env n goes to dst[i] + (n / block_envs[i]) * group_stride[i] + (n % block_envs[i]) * width[i]
 * (floats): block_envs = 1, group_stride = width for a dense [N, width] destination (row t of a [T, N, width]
 * rollout); block_envs = E, group_stride = capacity * E * width for the blocked layout [N / E, capacity, E, width]
 * with dst[i] already offset to step t. */
typedef struct {
  int32_t ncols;
  int32_t width[TRL_MAX_COLS];
  int32_t block_envs[TRL_MAX_COLS];
  int64_t group_stride[TRL_MAX_COLS];
  float *dst[TRL_MAX_COLS];
} trl_rollout_cols;
int trl_rollout_push(const float *rows_pinned, int64_t N, const trl_rollout_cols *cols, void *stream);

/* Exploration drawn on the device, one launch per act() (the reference draws on the host from numpy's global
 * generator after a device -> host copy of the network output: dqn_controller.py:57-59, ddpg_controller.py:92-94;
 * a2c_controller.py:21-27 and ppo_controller.py:24-30 sample with torch and copy back).  Counter-based generator
 * (Philox-4x32-10): the result is a function of (seed, call, row) alone, `call` numbered by the caller.  Same action
 * DISTRIBUTIONS as the reference, not the same draws.
 *   epsilon_greedy: q [n, A] -> actions [n] int64: argmax with probability 1 - eps + eps / A, any other action with
 *                   eps / A (the categorical of dqn_controller.py:16-22).
 *   normal:         actions [n, A] = mean + exp(log_std[a]) * N(0, 1)      (ActorCriticNet's Normal, models.py:139-141)
 *   categorical:    probs [n, A] -> actions [n] int64 by inverse CDF        (A2CNet's Categorical, models.py:105-107)
 *   ou_noise:       x <- x + theta * (mu - x) + sigma * N(0, 1) on the ONE state of A values all envs share
 *                   (ddpg_controller.py:29-36; state_in -> state_out, two buffers), actions [n, A] =
 *                   clip(actor_out + x, low, high) (:93).  A <= 64. */
int trl_act_epsilon_greedy(const float *q, int64_t n, int32_t A, double eps, uint64_t seed, uint64_t call,
                           int64_t *actions, void *stream);
int trl_act_normal(const float *mean, const float *log_std, int64_t n, int32_t A, uint64_t seed, uint64_t call,
                   float *actions, void *stream);
int trl_act_categorical(const float *probs, int64_t n, int32_t A, uint64_t seed, uint64_t call, int64_t *actions,
                        void *stream);
int trl_act_ou_noise(const float *actor_out, int64_t n, int32_t A, const float *state_in, float *state_out, double theta,
                     double mu, double sigma, double low, double high, uint64_t seed, uint64_t call, float *actions,
                     void *stream);

/* ------------------------------------------------------------------ no-allocation entry point
 * trl_ppo_loss for callers that cannot allocate or bring a workspace: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  It has NO counterpart in the reference (which has no device code at
 * all); it is the same computation, kernels and results as trl_ppo_loss above, with the reduction scratch (the
 * per-block partial sums of one launch) in a file-scope device array of the library instead of the `ws` argument:
 * 1024 x (4 + 64) floats = 272 KiB per GPU, held by every GPU the library runs on until the process ends.
 *   Limits: 1 <= A <= 64, 1 <= B <= TRL_PPO_LOSS_STATIC_MAX_B; anything else is refused with TRL_EINVAL before
 *   anything is enqueued.  The call allocates nothing and synchronises nothing; the first call on a device looks the
 *   array's device address up (hipGetSymbolAddress), later calls reuse it.
 *   Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race;
 *   calls on one stream are ordered and safe.  Nothing in the array has to be cleared between calls. */
#define TRL_PPO_LOSS_STATIC_MAX_B (1ll << 30)
int trl_ppo_loss_static(const float *mean, const float *log_std, const float *value, const float *action,
                        const float *old_logp, const float *adv, const float *returns, int64_t B, int32_t A,
                        double clip, double alpha, double beta, float *losses_out, float *dmean, float *dlog_std,
                        float *dvalue, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRL_HIP_H */
