// reset_kernel.hip -- Environment.reset (reference src/environment.py:45-107) for the
// whole batch: UAV i (1-based) at x = i*x_max/(n_uav+1), y = y_max/2, heading
// ~ U(-pi, pi), previous action ~ U{0..Na-1}; targets uniform in the box with heading
// ~ U(-pi, pi).  One thread per agent; every draw is a pure function of
// (seed, global env id, episode, agent index), so any sharding of the batch over GPUs
// reproduces the same environments.

#include "internal.h"
#include "philox.h"

namespace uavtrack {

namespace {

struct ResetParams {
    float *ux, *uy, *uz, *uh;
    int32_t *ua;
    float *tx, *ty, *tz, *th;
    int32_t *step_count, *episode_out;
    float *obs;
    int32_t B, N, M, dim, na_total;
    int64_t env_offset;
    uint32_t k0, k1, episode;
    double x_max, y_max, z_max;
    float x_max_f, y_max_f, z_max_f, inv_dc, inv_na_total;
};

__global__ void __launch_bounds__(256) reset_kernel(const ResetParams p)
{
    const int per_env = p.N + p.M;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)p.B * per_env) return;
    const int b = (int)(gid / per_env);
    const int idx = (int)(gid - (long long)b * per_env);
    const uint64_t genv = (uint64_t)(p.env_offset + b);
    const Philox4 r = philox4x32_10((uint32_t)genv, p.episode, (uint32_t)idx,
                                    0x55415631u ^ (uint32_t)(genv >> 32), p.k0, p.k1);
    if (idx < p.N) {
        const size_t g = (size_t)b * p.N + idx;
        const float x = (float)((double)(idx + 1) * p.x_max / (double)(p.N + 1));   // environment.py:105
        const float y = (float)(p.y_max / 2.0);                                      // environment.py:107
        const float h = fmaf(u01(r.v[0]), kTwoPi, -kPi);
        const int a = (int)(((uint64_t)r.v[1] * (uint32_t)p.na_total) >> 32);
        p.ux[g] = x; p.uy[g] = y; p.uh[g] = h; p.ua[g] = a;
        if (p.dim == 3) p.uz[g] = (float)(p.z_max / 2.0);
        if (idx == 0) { p.step_count[b] = 0; p.episode_out[b] = (int32_t)p.episode; }
        if (p.obs) {   // get_states() with empty observation lists (uav.py:174,186)
            float4 *o = reinterpret_cast<float4 *>(p.obs + g * UAVTRACK_OBS_DIM);
            o[0] = make_float4(-1.f, -1.f, -1.f, -1.f);
            o[1] = make_float4(-1.f, -1.f, -1.f, -1.f);
            o[2] = make_float4(-1.f, x * p.inv_dc, y * p.inv_dc, (float)a * p.inv_na_total);
        }
    } else {
        const int k = idx - p.N;
        const size_t g = (size_t)b * p.M + k;
        p.tx[g] = u01(r.v[0]) * p.x_max_f;
        p.ty[g] = u01(r.v[1]) * p.y_max_f;
        p.th[g] = fmaf(u01(r.v[2]), kTwoPi, -kPi);
        if (p.dim == 3) p.tz[g] = u01(r.v[3]) * p.z_max_f;
    }
}

}  // namespace

hipError_t launch_reset(const uavtrack_env *env, uint64_t seed, uint32_t episode, float *obs,
                        hipStream_t stream)
{
    const uavtrack_config &c = env->cfg;
    const StateBlock &s = env->state;
    const StepParams &k = env->base;
    ResetParams p;
    p.ux = s.ux; p.uy = s.uy; p.uz = s.uz; p.uh = s.uh; p.ua = s.ua;
    p.tx = s.tx; p.ty = s.ty; p.tz = s.tz; p.th = s.th;
    p.step_count = s.step_count; p.episode_out = s.episode;
    p.obs = obs;
    p.B = c.n_envs; p.N = c.n_uav; p.M = c.m_targets; p.dim = c.dim;
    p.na_total = k.na_total;
    p.env_offset = c.env_offset;
    p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32); p.episode = episode;
    p.x_max = c.x_max; p.y_max = c.y_max; p.z_max = c.z_max;
    p.x_max_f = (float)c.x_max; p.y_max_f = (float)c.y_max; p.z_max_f = (float)c.z_max;
    p.inv_dc = k.inv_dc; p.inv_na_total = k.inv_na_total;
    const long long total = (long long)c.n_envs * (c.n_uav + c.m_targets);
    const int threads = 256;
    const unsigned blocks = (unsigned)((total + threads - 1) / threads);
    hipLaunchKernelGGL(reset_kernel, dim3(blocks), dim3(threads), 0, stream, p);
    return hipGetLastError();
}

}  // namespace uavtrack
