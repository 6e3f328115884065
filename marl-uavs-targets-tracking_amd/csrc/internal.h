// internal.h -- shared between the C-ABI translation unit and the kernel files.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "uavtrack.h"

namespace uavtrack {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr int kMaxWorkgroup = 512;   // __launch_bounds__ of the rollout kernel

// Kernel arguments of one rollout launch (T >= 1 steps).  All pointers are device
// pointers.  Derived constants are computed once on the host in fp64, then cast.
// Per-handle block that lives in device memory: the kernel reaches the state arrays through
// one pointer instead of carrying 10 pointers (20 SGPRs) through the whole step loop.
struct StateBlock {
    // state, SoA over (env, uav) and (env, target); updated in place
    float *ux, *uy, *uz, *uh;
    int32_t *ua;
    float *tx, *ty, *tz, *th;
    int32_t *step_count;
    float climb_c[UAVTRACK_MAX_CLIMB], climb_s[UAVTRACK_MAX_CLIMB];   // cos/sin of the climb angles
};

struct StepParams {
    const StateBlock *st;
    // per-step I/O, leading [T] axis
    const int32_t *actions;
    float *obs, *reward, *terms;
    float4 *pose_out;        // MAAC-R: (x, y, z, raw reward) per agent-step, read by the deferred softmax mix
    int32_t *covered;
    uint8_t *done;
    float *ep_sums;
    uint2 *pairs;            // MAAC-R: neighbour pair list {flat [t][b][i] index of i, j}, i < j
    unsigned *pair_count;
    // geometry
    int32_t B, N, M, E, T, na, na_total, horizon;
    int32_t ep_accumulate;   // 1: ep_sums += (uavtrack_step_accumulate), 0: ep_sums = sums of this launch
    // fused greedy rollout (uavtrack_run_greedy): actions come from the in-kernel baseline policy
    int32_t *actions_out;    // [T][B][N], nullable
    int64_t env_offset;
    uint32_t greedy_k0, greedy_k1;
    // constants
    float x_max, y_max, z_max;
    float dtv_u, dtv_t;          // dt * v_max of UAVs / targets
    float turn_unit;             // dt * h_max / (na - 1)        (uav.py:81,96)
    float inv_dc, inv_dp, dp, dp2, dc2, two_dp2;
    float vratio;                // target.v_max / uav.v_max      (uav.py:116)
    float inv_na_total;
    float exp_k0, exp_k1;        // exp((2dp-d)/(2dp)) = exp2(k0 - k1*d)   (uav.py:226)
    float tt_ceil, inv_tt_ceil;  // 2*m_targets                   (environment.py:208)
    float dup_floor, inv_dup;    // -e/2*n_uav and 1/(e/2*n_uav)  (environment.py:210)
    float alpha, beta, gamma, coop;
};

struct PmiWeights {
    float *blob = nullptr;   // device, folded layout of uavtrack_set_pmi_weights
    int32_t hidden = 0;
    size_t n_floats = 0;
};

struct Geometry {
    int wgs = 0;          // threads per workgroup
    int envs_per_wg = 0;  // E
    int groups = 0;       // workgroups
    size_t lds_bytes = 0;
    int specialised = 0;
};

}  // namespace uavtrack

struct uavtrack_env {
    uavtrack_config cfg;
    uavtrack::StepParams base;   // constants filled at create
    uavtrack::StateBlock state;  // host copy of the device-resident block
    uavtrack::StateBlock *d_state = nullptr;
    uavtrack::Geometry geo;
    uavtrack::PmiWeights pmi;
    // MAAC-R scratch for `pmi_steps_cap` steps of deferred scoring (rewards never feed back into the
    // dynamics, so a chunk of steps is simulated first and all its pairs are scored in one launch):
    // pair list + counter, dense score matrix [steps][B][N][N], pose/raw [steps][B][N], and
    // observation / term buffers for callers that pass NULL
    int32_t pmi_steps_cap = 0;
    uint2 *pairs = nullptr;
    unsigned *pair_count = nullptr;
    unsigned long long *pair_total = nullptr;
    float *scores = nullptr, *obs_tmp = nullptr, *terms_tmp = nullptr;
    float4 *pose = nullptr;
};

namespace uavtrack {

// step_kernel.hip
Geometry plan_geometry(const uavtrack_config &cfg);
hipError_t launch_rollout(const uavtrack_env *env, const StepParams &p, hipStream_t stream, bool greedy = false);

// pmi_kernel.hip
bool pmi_hidden_supported(int hidden);
void pack_pmi_blob(const float *abi_blob, float *device_order, int hidden);
hipError_t launch_pmi_score(const uavtrack_env *env, const float *obs, hipStream_t stream);
hipError_t launch_pmi_finalize(const uavtrack_env *env, int steps, float *reward, hipStream_t stream);
hipError_t launch_ep_sums(const uavtrack_env *env, int steps, const float *reward, const float *terms,
                          const int32_t *covered, float *ep_sums, bool add, hipStream_t stream);

// policy_kernel.hip
hipError_t launch_greedy(const uavtrack_env *env, uint64_t seed, int32_t *actions, hipStream_t stream);

// reset_kernel.hip
hipError_t launch_reset(const uavtrack_env *env, uint64_t seed, uint32_t episode, float *obs,
                        hipStream_t stream);

}  // namespace uavtrack
