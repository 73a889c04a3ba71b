/*
 * mzplanner.h -- C ABI of the MI355X-native MuZero self-play planner (libmzplanner_hip.so).
 *
 * This is the drop-in boundary for the planning hot path of michaelnny/muzero.  Each entry point names the
 * reference interface (file:line under /root/reference/muzero/) it replaces.  Plain C: pointers and sizes only,
 * status-code returns (0 = ok, <0 = error, text via mz_last_error()), no exceptions cross the boundary,
 * caller-allocated outputs.  One planner handle per GPU; a handle is NOT thread-safe (one host thread per
 * handle); the handle owns its HIP stream and all device memory.
 *
 * Pointer arguments named h_* are host pointers; d_* are device (HBM) pointers on the planner's GPU.
 */
#ifndef MZPLANNER_H
#define MZPLANNER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZ_OK 0
#define MZ_E_INVALID (-1)     /* bad argument / unsupported configuration */
#define MZ_E_HIP (-2)         /* a HIP runtime call failed */
#define MZ_E_STATE (-3)       /* call out of order (e.g. search before weights were committed) */
#define MZ_E_TIES (-4)        /* injected tie-break stream exhausted (parity mode only) */
#define MZ_E_NOMEM (-5)

#define MZ_NET_MLP 0   /* MuZeroMLPNet        network.py:236-267 */
#define MZ_NET_BOARD 1 /* MuZeroBoardGameNet  network.py:540-574 */
#define MZ_NET_ATARI 2 /* MuZeroAtariNet      network.py:501-537 */

#define MZ_ENV_NONE 0
#define MZ_ENV_CARTPOLE 1  /* CartPole-v1 + StackFrameAndAction(4) + PlayerIdAndActionMaskWrapper, gym_env.py:271-365,436-459 */
#define MZ_ENV_TICTACTOE 2 /* TicTacToeEnv, games/tictactoe.py + games/env.py */
#define MZ_ENV_GOMOKU 3    /* GomokuEnv (board = obs_h, stack 4, five in a row), games/gomoku.py + games/env.py */
#define MZ_ENV_SYNTHETIC 4 /* stand-in for the Atari emulator (absent dependency): fresh U[0,1) frames, reward 0, 1000-step episodes */

/* Everything uct_search reads from MuZeroConfig (config.py:51-103) and from the network constructors
 * (network.py:239-247, 504-512, 543-549), plus planner-only sizing knobs. */
typedef struct {
    /* network */
    int32_t net_kind;            /* MZ_NET_* */
    int32_t obs_c, obs_h, obs_w; /* observation shape; MLP nets flatten it (network.py:153-154) */
    int32_t num_actions;
    int32_t num_planes;
    int32_t hidden_dim;          /* MLP only */
    int32_t num_res_blocks;      /* conv nets only */
    int32_t value_support_size;
    int32_t reward_support_size;
    /* search: config.py:58-78 */
    int32_t num_simulations;
    double discount;
    double pb_c_base;
    double pb_c_init;
    int32_t is_board_game;
    int32_t has_known_bounds;
    double known_bounds_min, known_bounds_max;
    double root_dirichlet_alpha;
    double root_exploration_eps;
    /* planner */
    int32_t num_envs;  /* capacity B: environments searched in lock-step on this GPU */
    int32_t max_ties;  /* length of the injected tie-break stream per env (parity mode) */
    uint64_t seed;     /* Philox key for on-device randomness (production mode) */
    /* child_U's product in searches WITHOUT root noise (evaluators: deterministic = True, pipeline.py:374,468).  There `child.prior` is an
     * np.float32 scalar multiplied by a Python float (mcts.py:189-197): numpy >= 2 (NEP 50) keeps the product in float32, numpy 1.x -- the
     * reference pins 1.21.6 (requirements.txt:21) -- promotes it to float64 and rounds once.  0 (default): the numpy-2 form, the one every
     * recorded fixture of this repo was produced under; 1: the numpy-1.21 form.  Self-play (float64 prior after the noise) is the same in both. */
    int32_t legacy_scalar_promotion;
} mz_config;

/* Injected randomness for a batch of searches: replaces the reference's global numpy RNG
 * (np.random.dirichlet mcts.py:245, np.random.choice mcts.py:124 and :404).  All host pointers.
 * Passing NULL for the whole struct selects on-device Philox randomness. */
typedef struct {
    const double* h_noise;   /* [B, A] Dirichlet samples; NULL => draw on device */
    const double* h_u_tie;   /* [B, max_ties] uniforms in [0,1): k-th real tie among n candidates picks cand[floor(u*n)] */
    const double* h_u_final; /* [B] uniform for the final inverse-CDF sample of the play policy */
} mz_rng_inputs;

typedef struct mz_planner mz_planner;

const char* mz_last_error(void);
const char* mz_version(void);

/* Which kernel build this handle's LAST search launch dispatched to (e.g. "k_search_fast<planes=512, TR=2, TV=2, FUSE=true, AC=2, ...>",
 * or the conv-tower / HBM-tree sequence), followed by every diagnostic switch below as this handle read it.  The string lives until
 * the calling thread's next mz_planner_describe.  A bench line prints the instantiation it ran, not a hard-coded name.
 *
 * DIAGNOSTIC ENVIRONMENT SWITCHES.  None is needed in production; each selects an alternative, bit-identical path for A/B measurements and
 * for the tests that prove the paths agree.  They are read ONCE -- the first group when a handle is created (mz_planner_create), the second
 * once per process at the first planner that needs it -- never between two calls on a handle, so a C-ABI caller cannot be handed a
 * different kernel from one call to the next.
 *   per handle, at mz_planner_create:
 *     MZ_FORCE_GENERIC=1    the shape-generic k_search instead of the tuned k_search_fast builds
 *     MZ_FUSE_ENV=0|1       device self-play as three launches per move (0) or one (1, default for LDS-resident MLP searches)
 *     MZ_GTREE_WAVE=0|1     HBM trees: select with 16 lanes per env (0) or one wave per env (1, default up to 256 actions)
 *     MZ_HWX=0..3           work split of k_search_fast's helper waves (default by head kinds)
 *     MZ_TREE_OLD=1         evaluate every level of every descent (no selection cache; the anchor of the tree parity tests)
 *     MZ_HBM_TREE=1         MLP nets: trees in HBM around batched k_infer launches even where they fit LDS
 *     MZ_NO_FAST_LAYOUT=1   never give k_search_fast its own LDS carve-out (the LunarLander-shaped search then runs the generic kernel)
 *     MZ_FAST_AC4=0         the general-action-count build instead of the four-action one
 *   per process, at first use (conv nets):
 *     MZ_ACTION_SPARSE=0    evaluate the dynamics net's action planes densely
 *     MZ_ACTION_FUSE=0      add the sparse action terms in their own kernel instead of the first conv's epilogue
 *     MZ_CONV_SPEC=0        no shape-specialised conv / tower builds
 *     MZ_TOWER=0            one launch per conv instead of the persistent residual tower
 *     MZ_CONV_TILE=th*100+tw, MZ_CONV_G=n, MZ_CONV_NCT=n   force the tiled conv kernel's output tile / images per workgroup / channel tiles per wave */
const char* mz_planner_describe(mz_planner* p);

/* Lifetime.  Replaces: network construction + .to(device) in the launchers (classic/run_training.py:83-99) and the
 * per-search allocations of Node objects (mcts.py:75-102). */
int mz_planner_create(const mz_config* cfg, int device_id, mz_planner** out);
int mz_planner_destroy(mz_planner* p);

/* Weights.  `name` is a state_dict key of the reference module (SURVEY 8b lists them; e.g.
 * "dynamics_net.transition_net.0.weight"); data is a host float32 tensor in torch layout; it is copied.
 * Replaces: actor_network.load_state_dict (pipeline.py:266).  Call mz_planner_commit_params after the last tensor:
 * it packs weights into MFMA fragment order and folds eval-mode BatchNorm. */
int mz_planner_set_param(mz_planner* p, const char* name, const float* h_data, const int64_t* shape, int32_t ndim);
int mz_planner_commit_params(mz_planner* p);

/* MuZeroNet.initial_inference (network.py:62-84), batched.  obs float32 [batch, obs_c*obs_h*obs_w];
 * outputs hidden [batch, hidden_size], pi [batch, A], value [batch]; reward is identically 0 (network.py:76). */
int mz_planner_initial_inference(mz_planner* p, int32_t batch, const float* h_obs, float* h_hidden, float* h_pi, float* h_value);

/* MuZeroNet.recurrent_inference (network.py:86-111), batched.  action int32 [batch]. */
int mz_planner_recurrent_inference(mz_planner* p, int32_t batch, const float* h_hidden, const int32_t* h_action, float* h_hidden_out,
                                   float* h_reward, float* h_pi, float* h_value);
int32_t mz_planner_hidden_size(const mz_planner* p);

/* uct_search (mcts.py:302-407) for `batch` independent roots in lock-step.
 *   h_obs float32 [batch, obs]; h_mask uint8 [batch, A] or NULL (actions_mask=None); players int32 [batch];
 *   h_temperature float64 [batch]; deterministic as mcts.py:311; rng NULL => on-device randomness.
 *   outputs: action int32 [batch], pi float64 [batch, A], root_value float64 [batch], visits int32 [batch, A] (may be NULL). */
int mz_planner_search(mz_planner* p, int32_t batch, const float* h_obs, const uint8_t* h_mask, const int32_t* h_current_player,
                      const int32_t* h_opponent_player, const double* h_temperature, int32_t deterministic, const mz_rng_inputs* rng,
                      int32_t* h_action, double* h_pi, double* h_root_value, int32_t* h_visits);

/* Tree-only parity entry: the same search with the network replaced by scripted outputs
 * (h_pi0 float32 [batch, A]; h_values / h_rewards float32 [batch, num_simulations]: simulation s receives
 * value[s], reward[s]).  Also returns the (parent node, action) expanded by every simulation.  Test hook for the
 * tree kernels: mcts.py:369-389 with network outputs injected. */
int mz_planner_search_scripted(mz_planner* p, int32_t batch, const float* h_pi0, const float* h_values, const float* h_rewards,
                               const uint8_t* h_mask, const int32_t* h_current_player, const int32_t* h_opponent_player,
                               const double* h_temperature, int32_t deterministic, const mz_rng_inputs* rng, int32_t* h_action,
                               double* h_pi, double* h_root_value, int32_t* h_visits, int32_t* h_trace_parent, int32_t* h_trace_action);

/* Device-resident self-play: run_self_play's inner loop (pipeline.py:91-113) for num_envs on-device environments.
 * mz_selfplay_reset seeds/initialises the envs (h_init_state: CartPole float64 [B,4] or NULL => U(-0.05,0.05) from Philox).
 * mz_selfplay_step performs ONE lock-step move for all envs: search (num_simulations) -> sample action -> env.step ->
 * record (obs, action, reward, pi, root_value, player) -> auto-reset finished episodes.  Nothing crosses PCIe.
 * temperature >= 0: the same value for every env (classic / Atari schedules depend on training steps only, config.py:252-267);
 * temperature < 0: the board game's own per-env schedule by episode step (config.py:236-249). */
int mz_selfplay_reset(mz_planner* p, int32_t env_kind, const double* h_init_state);
int mz_selfplay_step(mz_planner* p, double temperature, int32_t n_moves);
/* Copy out the records of the last `n_moves` moves (newest last): arrays [n_moves, B, ...]; any pointer may be NULL.
 * The records live in a ring of R moves (R = 64; 16 when one move of observations exceeds 64 MiB; at least an open
 * trajectory when a replay is attached): mz_selfplay_step may play more than R moves between reads -- older records are
 * overwritten -- and mz_selfplay_read returns MZ_E_INVALID when asked for more moves than the ring holds. */
int mz_selfplay_read(mz_planner* p, int32_t n_moves, float* h_obs, int32_t* h_action, float* h_reward, double* h_pi,
                     double* h_root_value, int32_t* h_player, uint8_t* h_done);
/* counters since reset: [0] env steps, [1] simulations, [2] finished episodes, [3] sum of finished episode lengths */
int mz_selfplay_counters(mz_planner* p, int64_t out[4]);

/* Device epilogue of run_self_play: what the reference does per episode on the host -- compute_n_step_target /
 * compute_mc_return_target (pipeline.py:632-707), priorities (:129,156), make_unroll_sequence (:710-767), including the
 * mid-episode flush every acc_seq_length steps (:118-142), then data_queue.put -> PrioritizedReplay.add (replay.py:67-75)
 * -- done on the GPU after every lock-step move, written straight into a caller-owned replay ring in device memory.
 * All pointers are DEVICE pointers (e.g. the storages of muzero_amd.replay.PrioritizedReplay(device='cuda')); the caller
 * keeps them alive while attached.  SINGLE WRITER: a ring may be attached to ONE planner at a time -- the planner reserves slots from
 * its own cursor (seeded from *num_added at attach) and publishes *num_added by overwriting it; give each planner its own ring.  Items of one environment appear in step order; environments interleave (the
 * reference's actors are independent processes).  Call BEFORE mz_selfplay_reset (the record ring is sized to hold an open
 * trajectory); ring == NULL detaches.  Attach and detach drain the planner's stream, so after a detach the counter and the
 * priorities are final.  mz_selfplay_read then returns at most the moves the record ring holds. */
typedef struct {
    int64_t capacity;     /* ring slots; slot of the i-th item ever added = i % capacity */
    float* state;         /* [capacity, obs_c*obs_h*obs_w] */
    void* action;         /* [capacity, unroll_steps] int8 when num_actions <= 128, int16 otherwise (the reference's int8, pipeline.py:753,
                           * cannot hold Gomoku 15x15's 226 actions: numpy 2 raises OverflowError there) */
    float* pi_prob;       /* [capacity, unroll_steps, num_actions] */
    float* value;         /* [capacity, unroll_steps] */
    float* reward;        /* [capacity, unroll_steps] */
    float* priority;      /* [capacity] */
    int64_t* num_added;   /* one counter, read at attach (the write cursor starts there) and from then on only PUBLISHED by the
                           * device: it advances after each lock-step move, once every slot below the new value is completely
                           * written (state, windows, priority), so a reader on another stream never sees a half-filled slot.
                           * The caller must not write it while attached. */
    int32_t* origin;      /* optional [capacity]: index of the environment that produced the item, or NULL */
    int32_t acc_seq_length, unroll_steps, td_steps;  /* MuZeroConfig fields (config.py:58-94) */
} mz_replay_ring;
int mz_selfplay_attach_replay(mz_planner* p, const mz_replay_ring* ring);

/* Measurement hooks (bench.py): HIP-event timing on the planner's own stream.
 * mz_profile_begin/end bracket a region; mz_profile_end returns elapsed milliseconds and the number of
 * search-kernel launches inside it (the dominant kernel of the path). */
int mz_profile_begin(mz_planner* p);
int mz_profile_end(mz_planner* p, double* elapsed_ms, double* search_kernel_ms, int64_t* search_kernel_launches);
int mz_planner_synchronize(mz_planner* p);

#ifdef __cplusplus
}
#endif
#endif
