/*
 * mz_oracle.h -- CPU ORACLE for the MuZero self-play planning path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (michaelnny/muzero:
 * muzero/mcts.py, muzero/network.py, muzero/util.py, muzero/games/*, muzero/gym_env.py,
 * muzero/pipeline.py target builders).  Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this
 * library.  The product (libmzplanner_hip.so, muzero_amd/) never does.
 *
 * Parity status: PINNED.  The oracle is checked against golden vectors recorded from the
 * reference itself (oracle/gen_golden.py -> tests/golden/*.npz) by tests/test_oracle_*.py,
 * including the reference's own known-answer tests (tests/pipeline_test.py:24-53,
 * tests/util_test.py:25-48, tests/games/*_test.py win lines).  Unpinned upstream pieces:
 * gym 0.23.1 CartPole physics (package absent everywhere; restated from its published equations).
 */
#ifndef MZ_ORACLE_H
#define MZ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- search configuration: the MuZeroConfig fields read by mcts.py (config.py:51-103) ---- */
typedef struct {
    int32_t num_actions;
    int32_t num_simulations;
    double discount;
    double pb_c_base; /* 19652  config.py:71 */
    double pb_c_init; /* 1.25   config.py:72 */
    int32_t is_board_game;
    int32_t has_known_bounds;
    double kb_min, kb_max;
    double dirichlet_alpha;  /* root_dirichlet_alpha */
    double exploration_eps;  /* root_exploration_eps = 0.25 config.py:68 */
    int32_t legacy_scalar_promotion; /* 1: child_U's np.float32-scalar x Python-float product as numpy 1.21 computes it (float64, one rounding) */
} mzo_search_config;

/* injected randomness for one search (replaces the global numpy RNG, mcts.py:124,245,404) */
typedef struct {
    const double* noise; /* [A] Dirichlet sample, or NULL => no noise is mixed in */
    const double* u_tie; /* uniforms in [0,1): k-th tie-break with n>1 candidates picks cand[floor(u*n)] */
    int32_t n_tie;       /* length of u_tie */
    double u_final;      /* uniform for the final inverse-CDF action sample */
} mzo_rng_inputs;

typedef struct {
    int32_t action;
    double root_value;
    int32_t n_tie_used;
    int32_t status; /* 0 ok, <0 error (see MZO_E_*) */
} mzo_search_result;

#define MZO_E_TIES_EXHAUSTED (-2)
#define MZO_E_BAD_ARG (-1)

/* ---- networks ---- */
typedef struct mzo_net mzo_net;

/* MuZeroMLPNet (network.py:236-267).  params: 20 pointers in state_dict order:
 * represent_net.net.{0,2}.{weight,bias}, dynamics_net.transition_net.{0,2}.*, dynamics_net.reward_net.{0,2}.*,
 * prediction_net.policy_net.{0,2}.*, prediction_net.value_net.{0,2}.*  (weights row-major [out][in]). */
mzo_net* mzo_net_create_mlp(int32_t input_dim, int32_t num_actions, int32_t num_planes, int32_t hidden_dim,
                            int32_t value_support, int32_t reward_support, const float* const* params);

/* MuZeroBoardGameNet (network.py:540-574) kind=0, MuZeroAtariNet (network.py:501-537) kind=1.
 * params: state_dict order with the num_batches_tracked entries removed. */
mzo_net* mzo_net_create_conv(int32_t kind, int32_t in_c, int32_t in_h, int32_t in_w, int32_t num_actions,
                             int32_t num_res_blocks, int32_t num_planes, int32_t value_support, int32_t reward_support,
                             const float* const* params, int32_t n_params);

/* scripted network for tree-only parity: recurrent call s returns (values[s], rewards[s]); initial returns pi0 */
mzo_net* mzo_net_create_scripted(int32_t num_actions, const float* pi0, const float* values, const float* rewards, int32_t n);

void mzo_net_destroy(mzo_net*);
int32_t mzo_net_hidden_size(const mzo_net*);
int32_t mzo_net_obs_size(const mzo_net*);

/* network.py:62-84 / 86-111 (batch of one).  pi may be NULL. */
void mzo_initial_inference(mzo_net*, const float* obs, float* hidden_out, float* pi_out, float* value_out);
void mzo_recurrent_inference(mzo_net*, const float* hidden_in, int32_t action, float* hidden_out, float* reward_out,
                             float* pi_out, float* value_out);

/* ---- the search: mcts.py:302-407 ---- */
mzo_search_result mzo_uct_search(const mzo_search_config* cfg, mzo_net* net, const float* obs, const uint8_t* mask /*[A] or NULL*/,
                                 int32_t current_player, int32_t opponent_player, double temperature, int32_t deterministic,
                                 const mzo_rng_inputs* rng, double* out_pi /*[A]*/, int32_t* out_visits /*[A] or NULL*/,
                                 int32_t* trace_parent /*[S] or NULL*/, int32_t* trace_action /*[S] or NULL*/,
                                 double* out_minmax /*[2] or NULL*/);

/* B independent searches, OpenMP over envs (the CPU baseline of bench.py; one env per thread, batch-1 inference
 * exactly like the reference's one-actor-per-process layout, classic/run_training.py:168-186). */
int32_t mzo_uct_search_batch(const mzo_search_config* cfg, mzo_net* net, int32_t batch, const float* obs, const uint8_t* mask,
                             const int32_t* cur_player, const int32_t* opp_player, const double* temperature, int32_t deterministic,
                             const double* noise /*[B,A] or NULL*/, const double* u_tie /*[B,n_tie]*/, int32_t n_tie,
                             const double* u_final /*[B]*/, int32_t* out_action, double* out_pi, double* out_root_value,
                             int32_t* out_visits, int32_t num_threads);

/* ---- helpers that are part of the path (mcts.py:220-299) ---- */
void mzo_prepare_root_prior(const float* pi0, int32_t A, const double* noise, double eps, const uint8_t* mask,
                            int32_t deterministic, double* prior64, float* prior32);
void mzo_generate_play_policy(const int32_t* visits, int32_t A, double temperature, double* pi);
int32_t mzo_sample_action(const double* pi, int32_t A, double u);

/* util.py */
float mzo_signed_parabolic(float x);
float mzo_logits_to_value(const float* logits, int32_t S);
void mzo_normalize_hidden(float* h, int32_t channels, int32_t spatial);
float mzo_expf(float x);

/* ---- environments ---- */
/* CartPole-v1, gym 0.23.1 equations (not vendored upstream).  state: x, x_dot, theta, theta_dot (float64 like gym) */
typedef struct {
    double s[4];
    int32_t steps;
    int32_t done;
} mzo_cartpole;
void mzo_cartpole_reset(mzo_cartpole*, const double init[4]);
/* returns reward; sets done (termination or TimeLimit 500) */
double mzo_cartpole_step(mzo_cartpole*, int32_t action, float obs_out[4]);

/* StackFrameAndAction for vector observations (gym_env.py:271-353): obs [stack][dim+1], newest first */
void mzo_stack_reset(float* stacked, int32_t stack, int32_t dim, const float* obs, int32_t num_actions);
void mzo_stack_push(float* stacked, int32_t stack, int32_t dim, const float* obs, int32_t action, int32_t num_actions);

/* BoardGameEnv (games/env.py:24-381) with the TicTacToe / Gomoku win rule (games/tictactoe.py:33-77, games/gomoku.py:72-116) */
#define MZO_MAX_BOARD 19
#define MZO_MAX_STACK 8
typedef struct {
    int32_t board_size, stack, num_to_win, num_actions;
    int8_t board[MZO_MAX_BOARD * MZO_MAX_BOARD];
    uint8_t mask[MZO_MAX_BOARD * MZO_MAX_BOARD + 1];
    /* per-player history of own-stone planes, newest first (games/env.py:294-310) */
    int8_t planes[2][MZO_MAX_STACK][MZO_MAX_BOARD * MZO_MAX_BOARD];
    int32_t current_player; /* 1 black, 2 white */
    int32_t steps;
    int32_t winner; /* 0 none */
    int32_t last_action[2];
} mzo_board;
void mzo_board_reset(mzo_board*, int32_t board_size, int32_t stack, int32_t num_to_win);
/* returns status 0 ok / <0 invalid (games/env.py:119-124) */
int32_t mzo_board_step(mzo_board*, int32_t action, double* reward, int32_t* done);
void mzo_board_observation(const mzo_board*, int8_t* obs /*[2*stack+1][N][N]*/);
int32_t mzo_board_game_over(const mzo_board*);

/* pipeline.py:632-707 */
void mzo_n_step_target(const double* rewards, const double* root_values, int32_t T, int32_t td_steps, double discount, double* out);
void mzo_mc_return_target(const double* rewards, const int32_t* player_ids, int32_t T, double* out);

#ifdef __cplusplus
}
#endif
#endif
