/*
 * mzlearner.h -- C ABI of the MI355X-native MuZero learner step (libmzlearner_hip.so): the MLP nets (MuZeroMLPNet, network.py:236-267)
 * and -- round 5 -- the conv nets (MuZeroBoardGameNet, network.py:540-574, and MuZeroAtariNet, :501-537: residual towers with train-mode BatchNorm).
 *
 * Row f2 of SURVEY.md section 8: what the reference does per training step in `run_training` (pipeline.py:238-255) --
 * `calc_loss` (:541-612) + `loss.backward()` + optional `clip_grad_norm_` (:246-247) + `optimizer.step()` (torch.optim.Adam with
 * L2 weight decay, classic/run_training.py:94) -- as hand-written gfx950 kernels (muzero_amd/csrc/mz_learn.h).  The batch is read
 * straight from the HBM-resident replay ring (muzero_amd.replay.PrioritizedReplay(device='cuda')) through a vector of row indices;
 * nothing crosses PCIe.  Plain C: pointers and sizes, status codes (0 = ok, <0 = error, text via mzl_last_error()), caller-allocated
 * buffers, no torch types.  All d_* pointers are device pointers on the learner's GPU; `stream` is a hipStream_t (NULL = the
 * null stream): every call only ENQUEUES work on it and returns.
 *
 * The gradient and the update are separate calls so that a data-parallel learner can all-reduce the flat gradient vector
 * (RCCL, muzero_amd.learner.allreduce_gradients) between them.
 */
#ifndef MZLEARNER_H
#define MZLEARNER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZL_OK 0
#define MZL_E_INVALID (-1)
#define MZL_E_HIP (-2)
#define MZL_E_STATE (-3)

#define MZL_NET_MLP 0    /* MuZeroMLPNet (network.py:236-267); kernels: muzero_amd/csrc/mz_learn.h */
#define MZL_NET_BOARD 1  /* MuZeroBoardGameNet (network.py:540-574); kernels: muzero_amd/csrc/mz_learn_conv.h */
#define MZL_NET_ATARI 2  /* MuZeroAtariNet (network.py:501-537): the same kernels; the 96 x 96 representation (network.py:312-353) on 12 x 12 / 12 x 16 tiles */

/* The network's constructor arguments (MuZeroMLPNet network.py:239-247 | MuZeroBoardGameNet :543-549) + the batch geometry of calc_loss
 * (pipeline.py:541-575). */
typedef struct {
    int32_t in_dim;               /* flattened observation (network.py:153-154); conv nets: in_channels * board_h * board_w */
    int32_t num_actions;
    int32_t num_planes;
    int32_t hidden_dim;
    int32_t value_support_size;   /* 1: squared-error value head (network.py:126-134) */
    int32_t reward_support_size;
    int32_t unroll_steps;         /* K, config.py:87 */
    int32_t max_batch;            /* capacity: mzl_grad accepts any batch <= max_batch */
    int32_t grad_slices;          /* >= 1: the weight-gradient kernel splits its reduction over this many workgroup rows (large batches); conv nets: 1 */
    int32_t net_kind;             /* MZL_NET_MLP | MZL_NET_BOARD | MZL_NET_ATARI */
    /* conv nets only (hidden_dim is unused there; the board net's support sizes are 1: squared-error heads, network.py:551): */
    int32_t in_channels;          /* planes of the observation (network.py:549 / :508 input_shape[0]) */
    int32_t board_h, board_w;     /* MZL_NET_BOARD: the board, board_h * board_w <= 240; MZL_NET_ATARI: the frame, 96 x 96 (hidden state 6 x 6) */
    int32_t num_res_blocks;       /* residual blocks of each of the three towers (network.py:546) */
} mzl_config;

/* One batch of `Transition`s (replay.py:27-32) addressed inside the replay ring's storages. */
typedef struct {
    const void* d_state;      /* [capacity][in_dim] float32, or int8 when state_is_int8 (board games) */
    const void* d_action;     /* [capacity][K] int8, or int16 when action_bytes == 2 (num_actions > 128) */
    const float* d_pi_prob;   /* [capacity][K][num_actions] */
    const float* d_value;     /* [capacity][K] */
    const float* d_reward;    /* [capacity][K] */
    const int64_t* d_index;   /* [batch] ring rows of the sampled items (0 .. batch-1 for a stacked batch) */
    const float* d_weights;   /* [batch] importance-sampling weights (pipeline.py:597) */
    float* d_loss;            /* [1]   out: the reported loss (pipeline.py:594-597) */
    float* d_priorities;      /* [batch] out: |v_0 - z_0| (pipeline.py:603-609) */
    int32_t batch;
    int32_t state_is_int8;
    int32_t action_bytes;     /* 1 or 2 */
} mzl_batch;

typedef struct mz_learner mz_learner;

const char* mzl_last_error(void);

/* DIAGNOSTIC ENVIRONMENT SWITCHES (none needed in production; every alternative computes the same update -- the tests prove it -- and all are
 * read once, at mzl_create, except MZL_NO_HEADS3 which is read once per process):
 *   MLP nets:  MZL_GENERIC=1 (shape-generic <F = false> kernels), MZL_NO_SLICE=1 (no plane-sliced stages at small batches),
 *              MZL_NO_HEADS3=1 (heads' launch two instead of three workgroups per CU), MZL_FAST_MAX_TILES / MZL_CHAIN_FAST_MAX_TILES /
 *              MZL_CHAIN_MIN_TILES / MZL_CHAIN_MAX_TILES=n (batch thresholds between the register-resident, streaming and persistent-chain
 *              builds), MZL_STAMPS=1 (allocate the cycle-stamp buffer read by tools/dev/learn_stamps.py)
 *   conv nets: MZLC_NO_PAIR=1 (one tower job per launch instead of the dynamics / prediction towers of a step paired), MZLC_NO_SIDE=1 (the
 *              generic conv build instead of the 15 x 15 one), MZLC_NO_FUSE_APPLY=1 (every block output by its own elementwise kernel instead
 *              of the next conv's staging), MZLC_NO_XCD_REMAP=1 (weight-gradient workgroups in launch order instead of one image chunk per XCD),
 *              MZLC_DENSE_TILING=1 (the towers' pixel tiling by densest packing instead of by launch time at max_batch), MZLC_NO_WGRAD_STACK=1 (one small
 *              image per weight-gradient staging round), MZLC_STACK_ROWS=1 (the round's images stacked vertically instead of side by side),
 *              MZLC_ACT_MFMA=1 (the action planes' weight gradient inside the MFMA kernel instead of k_lc_wgrad_act's gather), MZLC_NO_TAPSETS=1 (the Atari net's stride-2 parity planes as nine-tap convolutions with zero
 *              weights), MZLC_NO_WIDE_TILES=1 (12 x 12 tiles at 48 x 48 too), MZLC_NO_HALO_IN=1 (the Atari net's tiled stride-1 convolutions over whole haloed
 *              tiles instead of the tiles' inner positions only: round 5's form), MZLC_NO_OUT_PLANE=1 (those convolutions write inner-only tiles that k_lc_tile_scatter
 *              moves into the plane and takes the BatchNorm statistics of, instead of doing both in their own epilogue), MZLC_NO_FUSE_ENTRY=1 (their data gradients leave a gradient plane for k_lc_entry_plain to mask and sum instead of
 *              masking and summing in the epilogue), MZLC_KEEP_H1=1 (a tiled block's inner activation as a plane too, not only as the next conv's input tiles),
 *              MZLC_NO_RING_ROWS=1 (their weight gradients reduce over every row of the haloed tile,
 *              ring zeroed, instead of the inner rows), MZLC_NO_ROW_STEPS=1 (16 flat positions per reduction step of those instead of one row of a wide tile's inner columns), MZLC_NO_QUAD_STEPS=1 (the 12-wide tiles' weight gradients step over whole pitch rows, pad quad included), MZLC_NO_KEEP_TILES=1 (the tiled stages' input tiles gathered again for the weight gradient instead of kept
 *              from the forward pass), MZLC_WGRAD_MIN_IPW=n (least images per weight-gradient workgroup), MZLC_DEFER_WGRAD=0 / 1 (the shared towers' block layers take their weight
 *              gradient per unroll step / in one launch per layer over all steps; default: one launch where a step's batch is at most four staging rounds per
 *              workgroup), MZLC_WGRAD_SG=n (at most n images per staging round) */

int mzl_create(const mzl_config* cfg, int device_id, mz_learner** out);
int mzl_destroy(mz_learner* h);

/* Number of float32 parameters (the 20 tensors of MuZeroMLPNet.state_dict() -- SURVEY 8b -- concatenated in that order, each in
 * torch layout) and the size in floats of the gradient buffer the caller must provide (grad_slices * parameters). */
int64_t mzl_num_params(const mz_learner* h);
int64_t mzl_grad_floats(const mz_learner* h);
/* Number of parameter tensors (MLP nets: 20), and the i-th tensor of the flat vector: name (state_dict key), float offset, rows, columns
 * (columns == 0: a vector; conv weights [cout][cin][3][3] report rows = cout, columns = cin * 9) */
int32_t mzl_num_tensors(const mz_learner* h);
int mzl_tensor_info(const mz_learner* h, int32_t i, const char** name, int64_t* offset, int32_t* rows, int32_t* cols);
/* Conv nets: the BatchNorm2d buffers (network.py:283-291; train mode updates them on every forward, pipeline.py:575-582).  Buffer i is the
 * layer `name` (state_dict prefix): name.running_mean = d_running[offset .. offset + count), name.running_var = the `count` floats behind it,
 * name.num_batches_tracked = d_num_batches[i].  mzl_num_running = floats of d_running.  MLP nets: 0 buffers. */
int32_t mzl_num_buffers(const mz_learner* h);
int64_t mzl_num_running(const mz_learner* h);
int mzl_buffer_info(const mz_learner* h, int32_t i, const char** name, int64_t* offset, int32_t* count);
/* Caller-owned device buffers of the BatchNorm statistics (float32 [mzl_num_running], int64 [mzl_num_buffers]); required before mzl_grad
 * for conv nets.  Replaces: the module buffers that network.state_dict() carries (pipeline.py:224-230). */
int mzl_bind_buffers(mz_learner* h, float* d_running, int64_t* d_num_batches);

/* Caller-owned flat device buffers: master weights, gradients, Adam's exp_avg / exp_avg_sq (torch.optim.Adam state), all float32.
 * Replaces: network.parameters() / optimizer.state (pipeline.py:224-230).  The caller keeps them alive while bound. */
int mzl_bind(mz_learner* h, float* d_params, float* d_grads, float* d_exp_avg, float* d_exp_avg_sq);
/* (Re)build the MFMA operand copies from d_params (after loading a checkpoint into it).  Replaces: network.load_state_dict. */
int mzl_commit(mz_learner* h, void* stream);

/* loss + backward (pipeline.py:241-244): fills d_grads (slice 0 holds the complete gradient), d_loss, d_priorities. */
int mzl_grad(mz_learner* h, const mzl_batch* batch, void* stream);
/* clip_grad_norm_ when max_grad_norm > 0 (pipeline.py:246-247), then optimizer.step() (:249) for Adam step number `step` (1-based),
 * learning rate `lr` as MultiStepLR gives it for this step (:250), and the refresh of the operand copies. */
int mzl_apply(mz_learner* h, double lr, double beta1, double beta2, double eps, double weight_decay, double max_grad_norm, int64_t step,
              void* stream);

/* ---- Replay sampling and priority updates on the device (replay.py:81-113) ----------------------------------------------------
 * For a replay ring whose bookkeeping lives in HBM (muzero_amd.replay.PrioritizedReplay with a device writer attached: the planner's epilogue
 * publishes the committed item count and the priorities on the GPU): the learner draws its batch and writes the new priorities back without a
 * host read of either.  Uniform (priority_exponent == 0: every launcher's default, replay.py:87-89): index = floor(u * size), weights 1.
 * Proportional (replay.py:90-98): inverse-CDF picks on float64 prefix sums of priority ^ alpha, importance weights ((1 / size) / p) ^ beta divided by
 * their batch maximum.  Uniforms: Philox4x32-10 keyed by (seed; draw, sample) -- pass a new `draw` number per batch. */
typedef struct {
    const float* d_priority;      /* [capacity]; may be NULL for uniform draws */
    const int64_t* d_num_added;   /* [1] committed item count (replay.py:115-118); size = min(count, capacity) */
    int64_t capacity;
    double priority_exponent;             /* alpha */
    double importance_sampling_exponent;  /* beta */
    uint64_t seed, draw;
    int32_t batch;
    int64_t* d_index;             /* out [batch] */
    float* d_weights;             /* out [batch]; may be NULL for uniform draws */
    double* d_scratch;            /* [mzl_replay_scratch_doubles(capacity)], proportional draws only */
} mzl_replay_draw;
int64_t mzl_replay_scratch_doubles(int64_t capacity);
/* Optional (round 6): a device int32[2] that the replay kernels bump where the reference would have raised -- [0] a draw from an EMPTY replay
 * (replay.py:83-84; the draw returns slot 0), [1] a non-finite or negative priority in mzl_replay_update_priorities (replay.py:106-110; the value
 * is skipped).  The kernels cannot raise: the caller reads the counters at a point where it synchronises anyway.  NULL (default): no counting.
 * Process-wide; the pointer is read when a call is enqueued. */
int mzl_replay_set_error_counters(int32_t* d_counters);
int mzl_replay_sample(const mzl_replay_draw* draw, void* stream);
/* priority[d_index[b]] = d_new[b] (replay.py:106-113; of a repeated index the last b wins).  d_owner: int32 [capacity] scratch. */
int mzl_replay_update_priorities(float* d_priority, int64_t capacity, const int64_t* d_index, const float* d_new, int32_t batch, int32_t* d_owner,
                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif
