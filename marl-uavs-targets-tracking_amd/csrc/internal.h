// internal.h -- shared between the C-ABI translation unit and the kernel files.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "uavtrack.h"

namespace uavtrack {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr int kMaxWorkgroup = 512;   // __launch_bounds__ of the rollout kernel
constexpr size_t kLdsSoft = 64 * 1024;    // dynamic LDS a launch gets without asking
constexpr size_t kLdsMax = 160 * 1024;    // LDS of a gfx950 CU: what one workgroup may take (hipFuncAttributeMaxDynamicSharedMemorySize)
constexpr int kSymBits = 20;         // fixed-point bits of the shared duplicate term (step_kernel.hip, sym_dup): e * 2^20 < 2^22

// Kernel arguments of one rollout launch (T >= 1 steps).  All pointers are device
// pointers.  Derived constants are computed once on the host in fp64, then cast.
// The state arrays live in ONE device slab at offsets that follow from (B, N, M, dim), so the kernel
// carries a single base pointer (2 SGPRs instead of 20 through the whole step loop) and derives the
// rest with scalar adds -- no dependent pointer load at the start of a launch.  Slab order (4-byte
// units): ux uy uh ua [uz] (B*N each), tx ty th [tz] (B*M each), step_count (B), episode (B), climb_c climb_s (8 each).
struct StateBlock {
    // state, SoA over (env, uav) and (env, target); updated in place
    float *ux, *uy, *uz, *uh;
    int32_t *ua;
    float *tx, *ty, *tz, *th;
    int32_t *step_count;
    int32_t *episode;           // episode number of each environment's last reset (keys the next automatic one)
    float *climb_c, *climb_s;   // cos/sin of the climb angles (UAVTRACK_MAX_CLIMB each)
};

__host__ __device__ inline StateBlock state_view(float *slab, int B, int N, int M, bool z3)
{
    const size_t BN = (size_t)B * N, BM = (size_t)B * M;
    StateBlock v;
    float *f = slab;
    v.ux = f; f += BN;
    v.uy = f; f += BN;
    v.uh = f; f += BN;
    v.ua = reinterpret_cast<int32_t *>(f); f += BN;
    v.uz = z3 ? f : nullptr; if (z3) f += BN;
    v.tx = f; f += BM;
    v.ty = f; f += BM;
    v.th = f; f += BM;
    v.tz = z3 ? f : nullptr; if (z3) f += BM;
    v.step_count = reinterpret_cast<int32_t *>(f); f += B;
    v.episode = reinterpret_cast<int32_t *>(f); f += B;
    v.climb_c = f; f += UAVTRACK_MAX_CLIMB;
    v.climb_s = f;
    return v;
}
inline size_t state_slab_floats(int B, int N, int M, bool z3)
{
    return ((size_t)B * N) * (z3 ? 5 : 4) + ((size_t)B * M) * (z3 ? 4 : 3) + 2 * (size_t)B + 2 * UAVTRACK_MAX_CLIMB;
}

struct StepParams {
    float *slab;             // state slab, see state_view()
    // per-step I/O, leading [T] axis
    const int32_t *actions;
    float *obs, *reward, *terms;
    float2 *tpos;            // optional target trace [T][B][M] (x, y) after each step (uavtrack_set_target_trace)
    float *raw;              // optional raw rewards [T][B][N]: uav.raw_reward of environment.py:219 (uavtrack_set_raw_reward_output)
    float *state_copy;       // optional second copy of the state slab as it stands behind the launch (uavtrack_step_host: the host block)
    uint32_t *nbrec;         // MAAC-R: neighbour record per agent-step, read by the deferred softmax mix (nbrec_words())
    int32_t *covered;
    uint8_t *done;
    float *ep_sums;
    uint2 *pairs;            // MAAC-R: neighbour pair list {flat [t][b][i] index of i, that of j}, i < j (0xFFFFFFFF: slot-pool dummy)
    unsigned *pair_count;
    unsigned long long *pair_total;   // accounting: neighbour pairs emitted so far (uavtrack_pmi_pairs_scored)
    // geometry
    int32_t B, N, M, E, T, na, na_total, horizon;
    int32_t ep_accumulate;   // 1: ep_sums += (uavtrack_step_accumulate), 0: ep_sums = sums of this launch
    // automatic episode turnover (uavtrack_step_many_autoreset): an environment whose done flag fires is reset in place
    int32_t auto_reset;
    uint32_t reset_k0, reset_k1;     // reset seed (Philox key), as uavtrack_reset
    double x_max_d, y_max_d, z_max_d;
    // fused greedy rollout (uavtrack_run_greedy): actions come from the in-kernel baseline policy
    int32_t *actions_out;    // [T][B][N], nullable
    int64_t env_offset;
    uint32_t greedy_k0, greedy_k1;   // policy seed (greedy and actor rollouts)
    // fused actor rollout (uavtrack_run_actor): actions come from the in-kernel policy network (actor.h)
    const float *obs_in;     // [B][N][12] observation seen at the first step
    const float *actor_w;    // packed weight blob
    int32_t actor_hblocks, actor_mode;   // 32-unit tiles of the hidden layer
    // constants
    float x_max, y_max, z_max;
    float dtv_u, dtv_t;          // dt * v_max of UAVs / targets
    float turn_unit;             // dt * h_max / (na - 1)        (uav.py:81,96)
    float inv_dc, inv_dp, dp, dp2, dc2, two_dp2;
    float le_neg_scale, le_dp2, le_dc2, le_two_dp2;   // pk_le_mask: -S and nextafter(K) * S per threshold
    float lt_dp2;                                     // strict form: K * S  (d2 < K)
    float vratio;                // target.v_max / uav.v_max      (uav.py:116)
    float inv_na_total;
    float inv_na;                // 1 / na: the climb index of a 3-D action is floor((a + 0.5) / na)
    float act_bias, inv_act_bias;   // step_kernel.hip act_bias_shape(): K, a power of two above n_uav * na * nc, and 1 / K
    float exp_k0, exp_k1;        // exp((2dp-d)/(2dp)) = exp2(k0 - k1*d)   (uav.py:226)
    float sym_k0;                // step_kernel.hip sym_dup: k0 + b, b = kSymBits the fixed-point bits of the shared duplicate term
    float tt_ceil, inv_tt_ceil;  // 2*m_targets                   (environment.py:208)
    float dup_k, sym_dup_k;      // the duplicate term's clip and normalisation folded (environment.py:210,217): dup = clamp(k * sum g, -1, 0)
                                 // with k = -0.5 / (e/2 * n_uav) for a float sum of g, times 2^-kSymBits for sym_dup's fixed-point sum
    float alpha, beta, gamma, coop;
};

// MAAC-R neighbour record of one agent-step, 32-bit words: [0 .. W) neighbour bit mask (d <= dp on post-move poses,
// uav.py:278; bit j = UAV j, self excluded), [W] index of the first pair this UAV emitted (its neighbours j > i, in
// ascending j, occupy consecutive slots of the pair list / score array; isolated pairs are not emitted).  W = 1 up to 32
// UAVs (one 8-byte record), 2 up to 64, else ceil(N / 32).  The UAV's raw reward travels in the REWARD output slot of the
// step (step_kernel.hip, pmi_reward_slot): raw while it has neighbours, the final reward when it has none.
__host__ __device__ inline int nbrec_mask_words(int N) { return N <= 32 ? 1 : (N <= 64 ? 2 : (N + 31) / 32); }
__host__ __device__ inline int nbrec_words(int N) { return nbrec_mask_words(N) + 1; }

struct PmiWeights {
    float *blob = nullptr;   // device, folded layout of uavtrack_set_pmi_weights; behind it the bf16 planes of fc1
    const void *x6 = nullptr; // -> into blob: fc1 as three bf16 planes in MFMA operand order (pack_pmi_x6), or null
    const void *l1 = nullptr; // -> into blob: the branch layers as f16 planes in MFMA A-operand order (pack_pmi_l1), with t3
    const void *t3 = nullptr; // -> into blob: fc1 block-scaled as two f16 planes (f16(T w), remainder) for pmi_score_t3_kernel,
                              //    or null when the network's weights / activation bounds do not fit f16's range
    float t3_s1 = 1.0f, t3_t = 1.0f;   // the powers of two folded into the branch layers (S1) and fc1 (T) of the t3 planes
    float rng_inv[3] = {0.0f, 0.0f, 0.0f};   // 1 / the largest |x| per branch input the f16 planes take (pmi_kernel.hip, PmiParams)
    int32_t hidden = 0;
    size_t n_floats = 0;
};

struct Geometry {
    int wgs = 0;          // threads per workgroup
    int envs_per_wg = 0;  // E
    int groups = 0;       // workgroups
    size_t lds_bytes = 0;
    int specialised = 0;
    int lone = 0;         // single-wave groups on a grid of at most two waves per SIMD: the LONE kernel variant
};

}  // namespace uavtrack

struct uavtrack_env {
    uavtrack_config cfg;
    uavtrack::StepParams base;   // constants filled at create
    float *slab = nullptr;       // the one device allocation behind `state`
    uavtrack::StateBlock state;  // pointers into the slab (host-side view)
    uavtrack::Geometry geo;
    // MAAC-R launches of fewer than kPmiShortLaunch steps: the 4-wave geometry (one pair-list reservation per workgroup-step
    // on a quarter of the workgroups; the single-wavefront variant's block reservations pay off over many steps, and a
    // launch that starts with an empty pool waits for its first one)
    uavtrack::Geometry geo_short;
    uavtrack::Geometry last_launch;   // geometry of the most recent rollout launch (uavtrack_launch_info)
    uavtrack::PmiWeights pmi;
    int32_t n_cus = 0;           // compute units of the device (grid of the persistent scorer)
    int32_t pmi_scheme = 0;      // uavtrack_set_pmi_scheme: UAVTRACK_PMI_AUTO or a pinned scorer
    unsigned *pmi_flags = nullptr;   // device [2]: the f16 scorer's range flag, chunks re-scored by the wide-range kernel
    float *actor_w = nullptr;    // device blob of uavtrack_set_actor_weights (actor.h layout)
    int32_t actor_hidden = 0;
    // MAAC-R scratch for `pmi_steps_cap` steps of deferred scoring (rewards never feed back into the
    // dynamics, so a chunk of steps is simulated first and all its pairs are scored in one launch):
    // pair list + counter, one score per pair, neighbour records [steps][B][N], and
    // observation / term buffers for callers that pass NULL
    int32_t pmi_steps_cap = 0;
    uint2 *pairs = nullptr;
    unsigned *pair_count = nullptr;
    unsigned long long *pair_total = nullptr;
    float *scores = nullptr, *obs_tmp = nullptr;   // scores: one per emitted pair
    uint32_t *nbrec = nullptr;
    float2 *tpos = nullptr;           // caller's target-trace buffer (not owned), capacity in steps
    int32_t tpos_steps = 0;
    float *raw_out = nullptr;         // caller's raw-reward buffer (not owned), capacity in steps (uavtrack_set_raw_reward_output)
    int32_t raw_steps = 0;
    float *state_copy_out = nullptr;  // (during uavtrack_step_host) where the launch leaves a second copy of the state slab
    // uavtrack_step_host: one pinned, device-mapped host block (actions in, every output and a copy of the state out)
    void *host_blk = nullptr;         // host address (hipHostMalloc)
    void *host_blk_dev = nullptr;     // the same block as the device sees it (hipHostGetDevicePointer)
    size_t host_blk_bytes = 0;
    float *rsum = nullptr;            // [steps][B] per-step mean of the final reward (mix kernel -> episode return)
    // uavtrack_set_profiling: a HIP event pair on the launch stream around every kernel launch of the stepping entry points,
    // by kernel class (UAVTRACK_PROF_*); read and cleared by uavtrack_get_profile
    struct ProfRec { int cls; hipEvent_t a, b; };
    bool profiling = false;
    std::vector<ProfRec> prof;
    // uavtrack_pmi_inference scratch (grow-only): [2 n][12] scorer inputs and the n pair records
    float *inf_obs = nullptr;
    uint2 *inf_pairs = nullptr;
    size_t inf_cap = 0;
};

namespace uavtrack {

// step_kernel.hip
Geometry plan_geometry(const uavtrack_config &cfg, int n_simd, bool allow_small_grid = true);
enum { kPolicyGiven = 0, kPolicyGreedy = 1, kPolicyActor = 2 };   // where a rollout's actions come from
// (geo: the launch geometry, default the handle's own; MAAC-R launches of a few steps use env->geo_short)
hipError_t launch_rollout(uavtrack_env *env, const StepParams &p, hipStream_t stream, int policy = kPolicyGiven,
                          const Geometry *geo = nullptr);
size_t rollout_lds_bytes(const Geometry &g, int policy);   // dynamic LDS of a rollout launch with that policy

// pmi_kernel.hip
constexpr int kPmiShortLaunch = 16;
constexpr int kPmiMaxHidden = 256;                  // widest PMINetwork hidden layer the scorer is instantiated for
inline int pmi_padded_hidden(int hidden) { return (hidden + 31) / 32 * 32; }   // the scorer's column-block granule
void pack_pmi_blob(const float *abi_blob, float *device_order, int hidden);
constexpr int kPmiX6MaxHidden = 128;                // widest layer whose three bf16 planes stay register-resident (4 waves, one per SIMD)
constexpr int kPmiX6MinHidden = 64;                 // (narrower layers have fewer k-steps than the producer has pairs to hide)
inline size_t pmi_x6_floats(int hp) { return hp >= kPmiX6MinHidden && hp <= kPmiX6MaxHidden ? (size_t)3 * hp * hp * 3 / 2 : 0; }   // 3 planes x 2 B
void pack_pmi_x6(const float *abi_blob, uint16_t *planes, int hidden);
inline size_t pmi_t3_floats(int hp) { return hp >= kPmiX6MinHidden && hp <= kPmiX6MaxHidden ? (size_t)3 * hp * hp : 0; }   // 2 planes x 2 B
void pack_pmi_t3(const float *abi_blob, uint16_t *planes, int hidden, float T);
inline size_t pmi_l1_floats(int hp) { return pmi_t3_floats(hp) ? (size_t)(hp / 32) * 3 * 3 * 64 * 8 / 2 : 0; }   // [w][branch][3 planes][lane][8] x 2 B
void pack_pmi_l1(const float *abi_blob, uint16_t *planes, int hidden, float S1);
// (pairs / scores / n_uav default to the handle's MAAC-R scratch and swarm size; uavtrack_pmi_inference passes its own)
hipError_t launch_pmi_score(const uavtrack_env *env, const float *obs, hipStream_t stream, const uint2 *pairs = nullptr,
                            float *scores = nullptr, int n_uav = 0);
int pmi_effective_scheme(const uavtrack_env *env);              // enum uavtrack_pmi_scheme, never AUTO
bool pmi_scheme_available(const uavtrack_env *env, int scheme);
bool pmi_scheme_fits(int hidden_padded, bool f16_range_ok, int scheme);   // the same test for weights that are not loaded yet
hipError_t launch_pmi_counters_reset(const uavtrack_env *env, hipStream_t stream);
hipError_t launch_pmi_inference_prep(const float *x, float *obs2, uint2 *pairs, unsigned n, hipStream_t stream);
hipError_t launch_pmi_finalize(const uavtrack_env *env, int steps, float *reward, float *rsum, hipStream_t stream);
hipError_t launch_ep_reward(const uavtrack_env *env, int steps, const float *rsum, float *ep_sums, bool add, hipStream_t stream);

// policy_kernel.hip
hipError_t launch_greedy(const uavtrack_env *env, uint64_t seed, int32_t *actions, hipStream_t stream);
hipError_t launch_actor(const uavtrack_env *env, const float *obs, uint64_t seed, int mode, int32_t *actions,
                        float *probs, hipStream_t stream);

// reset_kernel.hip
hipError_t launch_reset(const uavtrack_env *env, uint64_t seed, uint32_t episode, float *obs,
                        hipStream_t stream);

}  // namespace uavtrack
