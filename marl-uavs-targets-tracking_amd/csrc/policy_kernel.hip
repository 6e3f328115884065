// policy_kernel.hip -- stand-alone policy kernels: (1) the C-METHOD greedy baseline policy,
// UAV.get_action_by_direction (reference src/agent/uav.py:324-369), and (2) the learner's shared actor,
// ActorCritic.take_action (src/models/actor_critic.py:138-148; see actor.h), for every UAV of every
// environment.  Both also exist fused into the rollout kernel; these are the per-step forms.
//
// (1) greedy baseline.
// Per UAV: score_t = 1 / d(u, t) - 0.8 * #{other UAVs (compared by position, uav.py:351) within dc
// of t}; the first best target wins; angle = atan2(ty - y, tx - x) - heading; with probability 0.25
// a uniformly random action instead (uav.py:338-339), with probability 0.3 the angle is zeroed
// (uav.py:365-366).  The reference then calls an undefined find_closest_a_idx (uav.py:368); it is
// defined here as the turn rate of uav.py:73-81 nearest to the wrapped angle, lowest index on ties.
// Random draws: Philox keyed by (seed, global env, step_count, uav).
//
// The reference's O(N M N) loop collapses: the number of UAVs within dc of a target is computed once
// per (env, target) by the target's lane; a UAV subtracts itself and the UAVs that share its exact
// position when it is itself within dc.  Same geometry as the step kernel (E whole envs per workgroup,
// one lane per UAV, poses staged in LDS).

#include "actor.h"
#include "greedy.h"

namespace uavtrack {

namespace {

struct GreedyParams {
    const float *ux, *uy, *uh, *tx, *ty;
    const int32_t *step_count;
    int32_t *actions;
    int32_t B, N, M, E, na;
    float dc2, turn_unit;
    int64_t env_offset;
    uint32_t k0, k1;
};

__global__ void __launch_bounds__(kMaxWorkgroup) greedy_policy_kernel(const GreedyParams p)
{
    extern __shared__ float2 gsm[];
    const int N = p.N, M = p.M, E = p.E;
    float2 *upos = gsm;                               // [E][N]
    float2 *tpos = upos + E * N;                      // [E][M]
    int *near_cnt = reinterpret_cast<int *>(tpos + E * M);   // [E][M] UAVs within dc of the target
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int env0 = blockIdx.x * E;
    const int envs_here = min(E, p.B - env0);
    const int e = tid / N, i = tid - e * N;
    const bool active = tid < E * N && e < envs_here;
    const size_t g = (size_t)(env0 + e) * N + i;
    float x = 0, y = 0, h = 0;
    if (active) {
        x = p.ux[g]; y = p.uy[g]; h = p.uh[g];
        upos[tid] = make_float2(x, y);
    }
    for (int q = tid; q < envs_here * M; q += nthreads)
        tpos[q] = make_float2(p.tx[(size_t)env0 * M + q], p.ty[(size_t)env0 * M + q]);
    __syncthreads();
    for (int q = tid; q < envs_here * M; q += nthreads) {
        const int te = q / M;
        near_cnt[q] = greedy_near_count(tpos[q], N, p.dc2, [&](int j) { return upos[te * N + j]; });
    }
    __syncthreads();
    if (!active) return;
    p.actions[g] = greedy_pick(x, y, h, i, N, M, p.na, p.dc2, p.turn_unit, (uint64_t)(p.env_offset + env0 + e),
                               (uint32_t)p.step_count[env0 + e], p.k0, p.k1,
                               [&](int j) { return upos[e * N + j]; },
                               [&](int k) { return tpos[e * M + k]; },
                               [&](int k) { return near_cnt[e * M + k]; });
}

struct ActorParams {
    const float *obs;          // [B][N][12]
    const float *weights;
    const int32_t *step_count;
    int32_t *actions;
    float *probs;              // [B][N][A], nullable
    int32_t B, N, A, hblocks, mode;
    int64_t env_offset;
    uint32_t k0, k1;
};

// one lane per UAV, 64 UAVs per wavefront (the MFMA tile), flat over (env, uav); whole wavefronts stay
// in step through actor_pick, lanes past the end carry zeros
template <int MT>
__global__ void __launch_bounds__(256) actor_policy_kernel(const ActorParams p)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = g < (size_t)p.B * p.N;
    float o[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int b = 0, i = 0, step = 0;
    if (valid) {
        b = (int)(g / p.N); i = (int)(g - (size_t)b * p.N);
        const float4 *ip = reinterpret_cast<const float4 *>(p.obs + g * UAVTRACK_OBS_DIM);
        const float4 q0 = ip[0], q1 = ip[1], q2 = ip[2];
        o[0] = q0.x; o[1] = q0.y; o[2] = q0.z; o[3] = q0.w; o[4] = q1.x; o[5] = q1.y;
        o[6] = q1.z; o[7] = q1.w; o[8] = q2.x; o[9] = q2.y; o[10] = q2.z; o[11] = q2.w;
        step = p.step_count[b];
    }
    float *pr = (valid && p.probs) ? p.probs + g * p.A : nullptr;
    ActorRng rng;
    rng.valid = false; rng.block = 0;
    const int act = actor_pick<true, MT>(o, p.weights, p.hblocks, p.A, (uint64_t)(p.env_offset + b),
                                     (uint32_t)step, i, p.k0, p.k1, p.mode, pr, rng);
    if (valid) p.actions[g] = act;
}

}  // namespace

hipError_t launch_actor(const uavtrack_env *env, const float *obs, uint64_t seed, int mode, int32_t *actions,
                        float *probs, hipStream_t stream)
{
    const uavtrack_config &c = env->cfg;
    ActorParams p;
    p.obs = obs; p.weights = env->actor_w; p.step_count = env->state.step_count;
    p.actions = actions; p.probs = probs;
    p.B = c.n_envs; p.N = c.n_uav; p.A = c.na * c.nc; p.hblocks = actor_blocks(env->actor_hidden); p.mode = mode;
    p.env_offset = c.env_offset;
    p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32);
    const size_t rows = (size_t)c.n_envs * c.n_uav;
    if (actor_tiles(c.dim == 3) == 1)
        hipLaunchKernelGGL(actor_policy_kernel<1>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(actor_policy_kernel<2>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_greedy(const uavtrack_env *env, uint64_t seed, int32_t *actions, hipStream_t stream)
{
    const uavtrack_config &c = env->cfg;
    GreedyParams p;
    p.ux = env->state.ux; p.uy = env->state.uy; p.uh = env->state.uh;
    p.tx = env->state.tx; p.ty = env->state.ty;
    p.step_count = env->state.step_count;
    p.actions = actions;
    p.B = c.n_envs; p.N = c.n_uav; p.M = c.m_targets; p.na = c.na;
    p.dc2 = env->base.dc2; p.turn_unit = env->base.turn_unit;
    p.env_offset = c.env_offset;
    p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32);
    // this kernel's own geometry: one lane per UAV, whole environments per workgroup (the rollout kernel's
    // lanes hold UAV pairs)
    const int wgs = p.N <= 256 ? 256 : kMaxWorkgroup;
    auto lds_of = [&](int E) { return (size_t)E * (p.N * 8 + p.M * 8 + p.M * 4); };
    int E = wgs / p.N;
    while (E > 1 && lds_of(E) > 64 * 1024) --E;
    p.E = E;
    const unsigned groups = (unsigned)((p.B + E - 1) / E);
    hipLaunchKernelGGL(greedy_policy_kernel, dim3(groups), dim3(wgs), lds_of(E), stream, p);
    return hipGetLastError();
}

}  // namespace uavtrack
