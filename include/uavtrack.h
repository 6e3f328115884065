/*
 * uavtrack.h -- C ABI of libuavtrack.so, the MI355X (gfx950) batched
 * multi-UAV target-tracking environment.
 *
 * The reference (tjuDavidWang/MARL-UAVs-Targets-Tracking) has no FFI: its hot
 * path is the duck-typed Python class `Environment` (src/environment.py:12).
 * Each entry point below names the reference interface it replaces; the Python
 * binding a maintainer adds on the reference side is in INTEGRATION.md and is
 * what marl-uavs-targets-tracking_amd/uavtrack/_lib.py implements (ctypes).
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure; the message is
 *    in uavtrack_last_error() (thread-local).  No exceptions cross the ABI.
 *  - all array arguments are DEVICE pointers (HIP), caller-owned, contiguous,
 *    fp32 / int32 / uint8 as declared.  The library keeps no reference to them
 *    after the stream work it enqueued has run.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *    The stepping and reset calls are asynchronous on that stream and never
 *    synchronise the host or allocate: they can be captured into a HIP graph.
 *    (MAAC-R scratch is sized for cfg.horizon steps when the weights are set;
 *    only a longer call grows it -- a synchronisation -- and under capture
 *    that call is refused instead.)  The calls that DO synchronise say so:
 *    the weight uploads, the info / accounting queries, uavtrack_step_host.
 *  - a handle is owned by one host thread at a time (like the reference's
 *    single-threaded Environment); one handle per GPU, one process per GPU.
 *  - batch layout is struct-of-arrays: UAV arrays are [n_envs][n_uav], target
 *    arrays [n_envs][m_targets], row-major, environment-major.
 *  - there is no CPU fallback: on a machine without a gfx950 device
 *    uavtrack_create fails.
 */
#ifndef UAVTRACK_H
#define UAVTRACK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVTRACK_ABI_VERSION 1
#define UAVTRACK_OBS_DIM 12      /* environment.py:28  state_dim = (4+1) + 4 + (2+1) */
#define UAVTRACK_MAX_CLIMB 8

/* Which cooperative reward runs in Environment.calculate_rewards
 * (environment.py:222-226 -> uav.py:312-322). */
enum uavtrack_reward_mode {
    UAVTRACK_REWARD_RAW  = 0,   /* MAAC:   cooperative == 0, reward = raw      (uav.py:270,300) */
    UAVTRACK_REWARD_MEAN = 1,   /* MAAC-G: pmi is None, neighbour mean         (uav.py:293-310) */
    UAVTRACK_REWARD_PMI  = 2    /* MAAC-R: PMI-softmax weighted neighbours     (uav.py:262-291) */
};

/* Everything Environment.__init__ (environment.py:13-43), Environment.reset
 * (environment.py:87-107) and the per-step `config` dict reads
 * (environment.py:207-224) -- captured once, as a POD. */
typedef struct uavtrack_config {
    uint32_t struct_size;       /* = sizeof(uavtrack_config), ABI check */
    int32_t  n_envs;            /* B: independent Environment instances on this GPU */
    int32_t  n_uav;             /* environment.n_uav      (environment.py:32) */
    int32_t  m_targets;         /* environment.m_targets  (environment.py:33) */
    int32_t  dim;               /* 2 (reference) or 3 (our own spec) */
    int32_t  na;                /* environment.na, turn-rate actions (uav.py:73-81) */
    int32_t  nc;                /* climb-angle actions, 1 in 2-D; action = a_turn + na * a_climb */
    int32_t  norm_n_uav;        /* config['environment']['n_uav'] of the clip   (environment.py:210) */
    int32_t  norm_m_targets;    /* config['environment']['m_targets'] of the clip (environment.py:208) */
    int32_t  reward_mode;       /* enum uavtrack_reward_mode */
    int32_t  horizon;           /* done[b] = step_count[b] >= horizon (train.py:160 num_steps); 0 = never */
    int32_t  device_id;         /* HIP device ordinal */
    int64_t  env_offset;        /* global id of env 0 of this shard (keys the reset RNG; multi-GPU) */
    double   x_max, y_max, z_max;
    double   dt;                /* uav.dt */
    double   u_v_max;           /* uav.v_max */
    double   u_h_max;           /* radians: pi / yaml uav.h_max (environment.py:100) */
    double   u_g_max;           /* radians: max climb angle (3-D only) */
    double   dc, dp;            /* uav.dc, uav.dp */
    double   t_v_max;           /* target.v_max */
    double   alpha, beta, gamma;/* uav.alpha/beta/gamma (environment.py:219-220) */
    double   cooperative;       /* config['cooperative'] (environment.py:224) */
} uavtrack_config;

typedef struct uavtrack_env uavtrack_env;   /* opaque handle */

/* ABI version of the loaded library. */
int uavtrack_version(void);

/* Message of the last failure on this thread ("" if none). */
const char *uavtrack_last_error(void);

/* Replaces Environment.__init__ (environment.py:13-43).  Allocates the
 * internal SoA state on cfg->device_id.  Fails if no gfx950 device. */
int uavtrack_create(const uavtrack_config *cfg, uavtrack_env **out);

/* Frees the handle and its device state. */
int uavtrack_destroy(uavtrack_env *env);

/* Replaces Environment.reset (environment.py:87-107): UAV i (1-based) at
 * x = i*x_max/(n_uav+1), y = y_max/2 [, z = z_max/2], heading ~ U(-pi,pi),
 * previous action ~ U{0..na*nc-1}; targets uniform in the box, heading
 * ~ U(-pi,pi).  The reference draws from Python's MT19937; here the stream is
 * Philox4x32-10 keyed by (seed, env_offset + b, episode, agent), so a shard
 * reproduces the same envs as the unsharded batch.  step_count <- 0.
 * obs (nullable) [B][N][12] receives get_states() of the fresh state:
 * [-1]*9 + [x/dc, y/dc, a/Na] (uav.py:174,186). */
int uavtrack_reset(uavtrack_env *env, uint64_t seed, uint32_t episode,
                   float *obs, void *stream);

/* State injection / extraction (no reference equivalent: the reference pokes
 * uav.x / target.x attributes directly; used for parity tests and as the env
 * checkpoint).  uz/tz are ignored (may be NULL) when dim == 2.  step_count
 * (nullable) is int32[B]. */
int uavtrack_set_state(uavtrack_env *env,
                       const float *ux, const float *uy, const float *uz, const float *uh,
                       const int32_t *ua,
                       const float *tx, const float *ty, const float *tz, const float *th,
                       const int32_t *step_count, void *stream);
int uavtrack_get_state(uavtrack_env *env,
                       float *ux, float *uy, float *uz, float *uh, int32_t *ua,
                       float *tx, float *ty, float *tz, float *th,
                       int32_t *step_count, void *stream);

/* The rest of the checkpoint: episode [B] int32 (device pointer) = the number of each environment's last reset
 * (uavtrack_reset's `episode` argument, advanced by every automatic reset).  It keys the Philox counter of the next
 * automatic reset (uavtrack_step_many_autoreset), so a handle restored with uavtrack_set_state + uavtrack_set_episodes
 * resets into the same states as the run the checkpoint was taken from.  (A separate pair of entry points: the
 * signatures of uavtrack_set_state / uavtrack_get_state stay as they are.) */
int uavtrack_set_episodes(uavtrack_env *env, const int32_t *episode, void *stream);
int uavtrack_get_episodes(uavtrack_env *env, int32_t *episode, void *stream);

/* Replaces the `pmi` argument of Environment.step (environment.py:120;
 * PMINetwork.inference PMINet.py:64-72, eval mode).  `folded` is a HOST
 * pointer to the BatchNorm-folded fp32 blob, layout (H = hidden):
 *   Wc[5][H] bc[H]  Wo[4][H] bo[H]  Wb[3][H] bb[H]  W1[3H][H] b1[H]  w2[H] b2[1]
 * (input-major so consecutive lanes read consecutive outputs).  n_floats must
 * equal 12H + 3H + 3H*H + H + H + 1.  folded == NULL disables PMI. */
int uavtrack_set_pmi_weights(uavtrack_env *env, const float *folded, size_t n_floats,
                             int32_t hidden, void *stream);

/* Which arithmetic scores the neighbour pairs (the 3H x H layer of PMINet.py:59, 98 % of the network's work).  All
 * three evaluate PMINetwork.forward at fp32 accuracy (tested against an fp64 forward); they differ in speed and range:
 *   F16X3   three f16 MFMAs per fp32 product on block-scaled operand planes -- the default for hidden 64 / 96 / 128 when
 *           the weights and the activation bounds fit f16's range; it watches its inputs, and a chunk in which an operand
 *           could saturate is scored again by BF16X6 (stream-ordered, counted in uavtrack_pmi_info);
 *   BF16X6  six bf16 MFMAs per fp32 product: fp32's exponent range -- networks the range guard turns away;
 *   FP32    fp32-input MFMA: every width up to 256 (the only kernel for 32 and for widths past 128). */
enum uavtrack_pmi_scheme {
    UAVTRACK_PMI_AUTO   = 0,    /* the fastest one the weights allow (the default) */
    UAVTRACK_PMI_F16X3  = 1,
    UAVTRACK_PMI_BF16X6 = 2,
    UAVTRACK_PMI_FP32   = 3
};

/* Pins the scorer (A/B measurements, parity tests of every dispatchable kernel).  Fails when the loaded weights
 * cannot run on that scheme (width, or f16 range); uavtrack_set_pmi_weights fails likewise while a scheme is pinned. */
int uavtrack_set_pmi_scheme(uavtrack_env *env, int32_t scheme);

/* out[0] = the scheme the next MAAC-R step / uavtrack_pmi_inference will launch (enum uavtrack_pmi_scheme, never AUTO;
 * 0 without weights), out[1] = the hidden width after padding to a multiple of 32, out[2] = 1 if the weights passed the
 * host-side f16 range guard, out[3] = chunks the wide-range kernel has re-scored since the handle was created because an
 * operand left f16's range at run time.  Synchronises `stream`. */
int uavtrack_pmi_info(uavtrack_env *env, int64_t out[4], void *stream);

/* Replaces PMINetwork.inference (PMINet.py:64-72: eval mode, no grad) on a batch: x [n][12] (device; row k is what
 * uav.py:281 builds, la_i * la_j) -> scores [n] (device), with the weights of uavtrack_set_pmi_weights and on the very
 * kernels that score the neighbour pairs of a MAAC-R step (f16 x 3 on block-scaled planes, bf16 x 6 or fp32 MFMA by width and weight range) -- the network alone, for
 * callers that hold pair inputs of their own and for accuracy tests of the scorer.  Stream-ordered; must not run
 * concurrently with a MAAC-R step of the same handle.  (Not counted by uavtrack_pmi_pairs_scored.) */
int uavtrack_pmi_inference(uavtrack_env *env, const float *x, int64_t n, float *scores, void *stream);

/* Replaces Environment.step (environment.py:120-164) for the whole batch.
 *   actions [B][N] int32 in [0, na*nc)      (train.py:173-176 action_list)
 *   obs     [B][N][12]  next_states          (environment.py:144)
 *   reward  [B][N]      reward['rewards']    (environment.py:158)
 *   terms   [3][B][N]   target_tracking_reward, boundary_punishment,
 *                       duplicate_tracking_punishment (environment.py:159-161); nullable
 *   covered [B] int32   covered_targets      (environment.py:146); nullable
 *   done    [B] uint8   step_count >= horizon; nullable */
int uavtrack_step(uavtrack_env *env, const int32_t *actions,
                  float *obs, float *reward, float *terms,
                  int32_t *covered, uint8_t *done, void *stream);

/* T consecutive steps in ONE launch with the state resident on chip
 * (replaces the `for i in range(num_steps)` loop of train.py:160-185 for
 * open-loop / pre-sampled actions).  Bitwise identical to T uavtrack_step
 * calls.  Outputs gain a leading [T] axis: actions [T][B][N], obs
 * [T][B][N][12], reward [T][B][N], terms [T][3][B][N], covered [T][B],
 * done [T][B].  ep_sums (nullable) [B][5] receives the episode accumulators of
 * train.py:181-192 over the T steps: sum_t mean_i reward, sum_t mean_i of the
 * three terms, sum_t covered. obs/terms/covered/done/ep_sums are nullable. */
int uavtrack_step_many(uavtrack_env *env, int32_t T, const int32_t *actions,
                       float *obs, float *reward, float *terms,
                       int32_t *covered, uint8_t *done, float *ep_sums, void *stream);

/* uavtrack_step_many with automatic episode turnover (rollouts of T steps "with auto-reset at done"): an
 * environment whose done flag fires at step t (step_count reached cfg.horizon) is reset right behind that step, inside
 * the launch -- exactly uavtrack_reset(reset_seed, e + 1) for that environment, e being the episode number of its
 * previous reset (uavtrack_reset stores its `episode` argument per environment) -- so one launch may span episodes
 * and the driver's per-episode reset launch disappears.  Row t of the outputs is the terminal step, as without the
 * reset; row t + 1 is the first step of the new episode (the reset state's own observation, [-1]*9 + [x/dc, y/dc, a/Na],
 * is not emitted; an in-kernel policy sees it).  Bitwise identical to: uavtrack_step_many up to each done step,
 * uavtrack_reset, continue.  ep_sums runs over the whole call.  Needs cfg.horizon >= 1. */
int uavtrack_step_many_autoreset(uavtrack_env *env, int32_t T, uint64_t reset_seed, const int32_t *actions,
                                 float *obs, float *reward, float *terms,
                                 int32_t *covered, uint8_t *done, float *ep_sums, void *stream);

/* uavtrack_step that also ADDS this step's contribution to the caller's running episode
 * accumulators ep_sums [B][5] (same five sums as uavtrack_step_many, train.py:181-192), so a
 * closed-loop driver needs no reduction kernels of its own.  ep_sums must not be NULL. */
int uavtrack_step_accumulate(uavtrack_env *env, const int32_t *actions,
                             float *obs, float *reward, float *terms,
                             int32_t *covered, uint8_t *done, float *ep_sums, void *stream);

/* Replaces UAV.get_action_by_direction (uav.py:324-369), the C-METHOD greedy baseline policy, for
 * every UAV on the current state: actions [B][N] int32 out.  score_t = 1/d(u,t) - 0.8 * #{other UAVs
 * within dc of t}, first best target, angle = atan2 - heading, epsilon 0.25 random action, 0.3
 * keep-straight.  The reference calls an undefined find_closest_a_idx (uav.py:368); here it is the
 * turn rate of uav.py:73-81 nearest to the wrapped angle (lowest index on ties).  Draws are Philox
 * keyed by (seed, env_offset + b, step_count[b], uav).  2-D only. */
int uavtrack_greedy_actions(uavtrack_env *env, uint64_t seed, int32_t *actions, void *stream);

/* The whole C-METHOD evaluation loop of train.run_epoch (train.py:326-370) in ONE launch: per step the
 * baseline policy above picks the actions from the current state, then Environment.step runs -- T
 * closed-loop steps with the state on chip.  Bitwise identical to T x (uavtrack_greedy_actions,
 * uavtrack_step).  actions_out (nullable) [T][B][N] receives the chosen actions; the other outputs are
 * those of uavtrack_step_many.  Reward modes RAW / MEAN, 2-D only. */
int uavtrack_run_greedy(uavtrack_env *env, int32_t T, uint64_t seed, int32_t *actions_out,
                        float *obs, float *reward, float *terms,
                        int32_t *covered, uint8_t *done, float *ep_sums, void *stream);

/* ---- the learner's shared actor, run on the device next to the environment ----
 * The reference's train.operate_epoch calls ActorCritic.take_action once per UAV per step
 * (train.py:165-172 -> actor_critic.py:138-148): a batch-1 forward of FnnPolicyNet
 * (actor_critic.py:85-98: Linear(12,H) - ReLU - Linear(H,na) - softmax) and Categorical(probs).sample(),
 * with a host round trip each.  These three entry points keep that loop on the GPU. */
enum { UAVTRACK_ACTOR_SAMPLE = 0,   /* Categorical(probs).sample(): inverse CDF at a Philox uniform */
       UAVTRACK_ACTOR_ARGMAX = 1 }; /* deterministic evaluation: most probable action, lowest index on ties */

/* Uploads FnnPolicyNet's parameters (HOST pointers, fp32, torch layouts): w1 [hidden][12] = fc1.weight,
 * b1 [hidden] = fc1.bias, w2 [na*nc][hidden] = fc2.weight, b2 [na*nc] = fc2.bias.  na <= 12 in 2-D (the
 * reference's action space, configs: na = 12), na*nc <= 48 in 3-D.  w1 = NULL removes the actor.
 * Synchronises `stream`. */
int uavtrack_set_actor_weights(uavtrack_env *env, const float *w1, const float *b1,
                               const float *w2, const float *b2, int32_t hidden, void *stream);

/* take_action for every UAV: obs [B][N][12] (what get_local_state returned, i.e. the obs output of the
 * previous step / reset) -> actions [B][N] int32, and, if probs != NULL, the policy's probabilities
 * probs [B][N][na*nc].  Draws: Philox4x32-10 keyed by seed, counter (env_offset + b, step_count[b] >> 2,
 * uav); the uniform of step s is word s & 3 of that block (24 bits), the action the first index whose
 * cumulative probability exceeds it. */
int uavtrack_actor_actions(uavtrack_env *env, const float *obs, uint64_t seed, int32_t mode,
                           int32_t *actions, float *probs, void *stream);

/* The rollout half of train.operate_epoch (train.py:160-192) in ONE launch: per step every UAV's action
 * comes from the actor applied to its own previous observation (held in registers), then
 * Environment.step runs -- T closed-loop steps with the state on chip.  obs_in [B][N][12] is the
 * observation the policy sees at the first step.  Bitwise identical to T x (uavtrack_actor_actions,
 * uavtrack_step).  actions_out (nullable) [T][B][N]; the other outputs are those of uavtrack_step_many,
 * i.e. the (state, action, reward, next_state) transitions of train.py:176-180 land in
 * obs[t-1] / actions_out[t] / reward[t] / obs[t].  With reward_mode PMI the launch is chunked like
 * uavtrack_step_many (rollout, pair scorer, softmax mix per chunk); the policy never reads rewards. */
int uavtrack_run_actor(uavtrack_env *env, int32_t T, uint64_t seed, int32_t mode, const float *obs_in,
                       int32_t *actions_out, float *obs, float *reward, float *terms,
                       int32_t *covered, uint8_t *done, float *ep_sums, void *stream);

/* Optional extra output of every stepping entry point (uavtrack_step, _step_accumulate, _step_many, _run_greedy,
 * _run_actor): the target positions after each step, tpos [T][B][M][2] = (x, y) -- what Environment.step appends to
 * position['all_target_xs'/'all_target_ys'] (environment.py:150-153) and Environment.save_position writes to
 * t_xy<ep>.csv (environment.py:232-238).  (UAV positions are already in the observations: obs[..., 9:11] * dc,
 * uav.py:154.)  The buffer, a device pointer like the others, is written by every later stepping call -- row t of the
 * call's T steps at tpos[t] -- until it is replaced; it must hold capacity_steps >= the largest T passed while it
 * is set (checked).  tpos = NULL switches the output off (the default). */
int uavtrack_set_target_trace(uavtrack_env *env, float *tpos, int32_t capacity_steps);

/* Optional extra output of every stepping entry point, like the target trace: raw [T][B][N] = uav.raw_reward of every
 * UAV after each step -- alpha * tracking + beta * boundary + gamma * duplicate of the clipped, normalised terms
 * (environment.py:211-219), BEFORE the cooperative sharing of environment.py:222-226.  The reference keeps it as a public
 * attribute of each UAV (uav.py:50).  Row t of a call's T steps at raw + t * B * N; capacity_steps >= the largest T passed
 * while the buffer is set (checked); raw = NULL switches the output off (the default). */
int uavtrack_set_raw_reward_output(uavtrack_env *env, float *raw, int32_t capacity_steps);

/* Environment.step (environment.py:120-164) for a caller that lives on the HOST, the way the reference's own training
 * loop calls it (train.py:160-185: a Python list of actions in; next_states, the reward dict and the covered count out,
 * one environment, one step at a time).  actions_host [B][N] int32 is a HOST pointer; the results are left in a
 * library-owned, page-locked host block that the kernels write directly through its device mapping (no device-to-host
 * copy call), together with a copy of the state as it stands behind the step (what Environment.step appends to
 * `position`, environment.py:150-155, and what callers read as uav.x / target.x).  `out` receives HOST pointers into that
 * block; the block is allocated once per handle, so the pointers are the same on every call and stay valid -- their
 * contents overwritten by the next uavtrack_step_host -- until uavtrack_destroy.
 * This entry point SYNCHRONISES `stream` before it returns (the one stepping call that does): it is meant for batches
 * of one or a few environments -- the drop-in adapter -- where a step is bound by the launch and the synchronisation,
 * not by the kernel.  Batched rollouts use uavtrack_step / uavtrack_step_many on device buffers. */
typedef struct uavtrack_host_step {
    const float   *obs;        /* [B][N][12] next_states                         (environment.py:144) */
    const float   *reward;     /* [B][N]     reward['rewards']                   (environment.py:158) */
    const float   *terms;      /* [3][B][N]  the three normalised terms          (environment.py:159-161) */
    const float   *raw;        /* [B][N]     uav.raw_reward                      (environment.py:219) */
    const int32_t *covered;    /* [B]        covered_targets                     (environment.py:146) */
    const uint8_t *done;       /* [B]        step_count >= horizon */
    const float   *ux, *uy, *uz, *uh;   /* [B][N] UAV poses behind the step (uz NULL in 2-D) */
    const int32_t *ua;                  /* [B][N] the actions just applied (uav.a) */
    const float   *tx, *ty, *tz, *th;   /* [B][M] target poses behind the step (tz NULL in 2-D) */
    const int32_t *step_count;          /* [B] */
} uavtrack_host_step;
int uavtrack_step_host(uavtrack_env *env, const int32_t *actions_host, uavtrack_host_step *out, void *stream);

/* MAAC-R accounting for reports: out[0] = neighbour pairs the stepping entry points have handed to the PMI network
 * since the handle was created (each unordered pair once per step).  A pair of UAVs that are each other's ONLY
 * neighbour is not among them: the softmax over a single neighbour is 1 whatever its score (uav.py:287-288), so such a
 * pair is never scored.  Synchronises `stream`. */
int uavtrack_pmi_pairs_scored(uavtrack_env *env, uint64_t *out, void *stream);

/* Measurement hook (bench.py's roofline legs): with profiling on, every kernel launch of the stepping entry points is
 * bracketed by a HIP event pair on the launch stream.  uavtrack_get_profile synchronises `stream`, adds the elapsed
 * milliseconds up per kernel class into ms[UAVTRACK_PROF_CLASSES] and the launch counts into launches[...] (either may
 * be NULL), and forgets the recorded pairs.  Off by default: no events, no cost. */
enum { UAVTRACK_PROF_ROLLOUT = 0,   /* rollout_kernel: the fused environment step(s) */
       UAVTRACK_PROF_SCORER  = 1,   /* MAAC-R: pmi_score_x6_kernel / pmi_score_kernel */
       UAVTRACK_PROF_MIX     = 2,   /* MAAC-R: pmi_mix_kernel (softmax mix + final clip) */
       UAVTRACK_PROF_EPSUMS  = 3,   /* MAAC-R: ep_reward_kernel (episode return from the per-step means) */
       UAVTRACK_PROF_CLASSES = 4 };
int uavtrack_set_profiling(uavtrack_env *env, int32_t on);
int uavtrack_get_profile(uavtrack_env *env, double *ms, int64_t *launches, void *stream);

/* Launch geometry of the step kernel, for reports: out[0] = workgroup size,
 * out[1] = envs per workgroup, out[2] = workgroups, out[3] = LDS bytes per
 * workgroup, out[4] = 1 if a compile-time-specialised (N, M) variant is used. */
int uavtrack_kernel_info(uavtrack_env *env, int64_t out[5]);

/* Geometry of the most recent rollout launch of a stepping entry point (MAAC-R picks it per launch: the single-wavefront
 * variant exists for launches with every output and no extras only): out[0] = workgroup size, out[1] = envs per
 * workgroup, out[2] = workgroups, out[3] = 1 if the single-wavefront (pooled pair-list slots) kernel variant ran.
 * All zero before the first launch. */
int uavtrack_launch_info(uavtrack_env *env, int64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* UAVTRACK_H */
