// api.hip -- the C ABI of libuavtrack.so (include/uavtrack.h): handle lifetime,
// argument validation, constant folding, and stream-ordered launches.  No compute
// happens on the host and there is no CPU fallback.

#include "internal.h"
#include "actor.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace uavtrack;

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e__));   \
    } while (0)

// Makes the handle's device current for the duration of one ABI call and gives the caller's device back on
// return: a process that drives several shards, or keeps PyTorch on another GPU, must not find its current
// device changed behind its back.
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            changed = (err == hipSuccess);
        }
    }
    ~DeviceGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define ON_DEVICE(dev)            \
    DeviceGuard guard__(dev);     \
    HIP_TRY(guard__.err)

template <typename T>
hipError_t dmalloc(T **p, size_t n)
{
    return hipMalloc(reinterpret_cast<void **>(p), (n ? n : 1) * sizeof(T));
}

void drop_profile(uavtrack_env *env)
{
    for (auto &r : env->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    env->prof.clear();
}

// One kernel launch of a stepping entry point, bracketed by an event pair when profiling is on (uavtrack_set_profiling).
template <typename F>
hipError_t timed_launch(uavtrack_env *env, int cls, hipStream_t st, F &&launch)
{
    if (!env->profiling || env->prof.size() >= 65536) return launch();      // (bounded: a caller that never reads the profile leaks nothing further)
    uavtrack_env::ProfRec r{cls, nullptr, nullptr};
    hipError_t e = hipEventCreate(&r.a);
    if (e == hipSuccess) e = hipEventCreate(&r.b);
    if (e == hipSuccess) e = hipEventRecord(r.a, st);
    if (e != hipSuccess) {          // no events to be had: the step still runs, this launch just goes untimed
        if (r.a) (void)hipEventDestroy(r.a);
        if (r.b) (void)hipEventDestroy(r.b);
        (void)hipGetLastError();
        return launch();
    }
    e = launch();
    if (e == hipSuccess) e = hipEventRecord(r.b, st);
    env->prof.push_back(r);
    return e;
}

void free_state(uavtrack_env *env)
{
    drop_profile(env);
    void *ptrs[] = {env->slab, env->pmi.blob, env->actor_w, env->pairs, env->pair_count, env->pair_total, env->scores, env->nbrec,
                    env->obs_tmp, env->rsum, env->inf_obs, env->inf_pairs, env->pmi_flags};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (env->host_blk) (void)hipHostFree(env->host_blk);
    env->host_blk = env->host_blk_dev = nullptr;
}

int validate(const uavtrack_config &c)
{
    if (c.struct_size != sizeof(uavtrack_config))
        return fail("uavtrack_config.struct_size %u != %zu (ABI mismatch)", c.struct_size, sizeof(uavtrack_config));
    if (c.n_envs < 1) return fail("n_envs must be >= 1 (got %d)", c.n_envs);
    if (c.n_uav < 1 || c.n_uav > kMaxWorkgroup) return fail("n_uav must be in [1, %d] (got %d)", kMaxWorkgroup, c.n_uav);
    if (c.m_targets < 1 || c.m_targets > 4096) return fail("m_targets must be in [1, 4096] (got %d)", c.m_targets);
    if (c.dim != 2 && c.dim != 3) return fail("dim must be 2 or 3 (got %d)", c.dim);
    if (c.na < 2) return fail("na must be >= 2 (got %d)", c.na);
    if (c.nc < 1 || c.nc > UAVTRACK_MAX_CLIMB) return fail("nc must be in [1, %d] (got %d)", UAVTRACK_MAX_CLIMB, c.nc);
    if (c.dim == 2 && c.nc != 1) return fail("nc must be 1 when dim == 2 (got %d)", c.nc);
    if (c.norm_n_uav < 1 || c.norm_m_targets < 1) return fail("norm_n_uav / norm_m_targets must be >= 1");
    if (c.reward_mode < UAVTRACK_REWARD_RAW || c.reward_mode > UAVTRACK_REWARD_PMI)
        return fail("reward_mode must be 0 (raw), 1 (mean) or 2 (pmi) (got %d)", c.reward_mode);
    if (c.horizon < 0) return fail("horizon must be >= 0");
    if (!(c.dc > 0) || !(c.dp > 0) || !(c.u_v_max > 0) || !(c.dt > 0)) return fail("dc, dp, u_v_max, dt must be > 0");
    // the step kernel addresses a step's outputs as uniform base + 32-bit lane offset (largest row: obs, 48 B per agent)
    if ((int64_t)c.n_envs * c.n_uav * UAVTRACK_OBS_DIM * 4 >= ((int64_t)1 << 32))
        return fail("batch too large: n_envs * n_uav must stay below %lld agents per GPU (got %lld)",
                    (long long)(((int64_t)1 << 32) / (UAVTRACK_OBS_DIM * 4)), (long long)c.n_envs * c.n_uav);
    if ((int64_t)c.n_envs * c.m_targets >= ((int64_t)1 << 30))
        return fail("batch too large: n_envs * m_targets must stay below %lld targets per GPU (got %lld)",
                    (long long)((int64_t)1 << 30), (long long)c.n_envs * c.m_targets);
    return 0;
}

// MAAC-R scratch for deferred scoring of up to `steps` steps per chunk (grow-only).  Per step: the pair
// list and its score array at their worst case (every pair within dp), the neighbour records, and observation /
// term buffers for callers that pass NULL.  UAVTRACK_PMI_SCRATCH_MB bounds it (default 8192 MiB; INTEGRATION.md: footprint).
int ensure_pmi_scratch(uavtrack_env *env, int32_t steps, hipStream_t st)
{
    const uavtrack_config &c = env->cfg;
    const size_t BN = (size_t)c.n_envs * c.n_uav;
    // slots per step: every pair within dp at worst -- twice that, because the single-wavefront rollout variant hands out
    // pair-list slots in blocks and what a block has left when a step does not fit goes to dummies (less than that step's
    // pairs each time: never more than one dummy per real pair), plus a block per workgroup for the end of the launch
    // (only the pooled geometry pays the doubling: the 4-wave one reserves exactly what a step emits)
    const size_t pairs_step = env->geo.lone ? BN * (c.n_uav - 1) + (size_t)env->geo.groups * 64 + 1 : BN * (c.n_uav - 1) / 2 + 1;
    const size_t rec_bytes = (size_t)nbrec_words(c.n_uav) * 4;
    const size_t per_step = pairs_step * (sizeof(uint2) + 4) + BN * rec_bytes + BN * UAVTRACK_OBS_DIM * 4 + (size_t)c.n_envs * 4;
    size_t budget = (size_t)8192 << 20;        // (of 288 GB: a 200-step rollout of the reference shape stays one chunk)
    if (const char *s = getenv("UAVTRACK_PMI_SCRATCH_MB")) budget = (size_t)atoll(s) << 20;
    int64_t cap = (int64_t)(budget / per_step);
    const int64_t idx_cap = (int64_t)(0xFFFFFFFFull / BN);     // pair records carry a 32-bit flat [step][b][i] index
    if (cap > idx_cap) cap = idx_cap;
    if (cap < 1) cap = 1;
    if (cap > steps) cap = steps;
    // the counters first, each guarded on its own: a failure between two of them must not leave a later call with a null one
    const bool fresh_counters = !env->pair_count || !env->pair_total || !env->pmi_flags;
    if (cap <= env->pmi_steps_cap && !fresh_counters) return 0;
    {   // Growing means a host synchronisation and device allocations: neither may happen while `st` is being captured into a
        // HIP graph.  uavtrack_set_pmi_weights sizes the scratch for cfg.horizon steps, so only a call longer than an
        // episode can get here; under capture it is refused with a message of its own instead of invalidating the capture.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail("MAAC-R scratch must grow to %lld steps per chunk (it holds %d) but the stream is being captured into a graph: "
                        "issue one call of this length outside the capture first, or keep T <= cfg.horizon (sized at "
                        "uavtrack_set_pmi_weights)", (long long)cap, env->pmi_steps_cap);
        (void)hipGetLastError();
    }
    if (!env->pair_count) { HIP_TRY(dmalloc(&env->pair_count, 2)); HIP_TRY(hipMemsetAsync(env->pair_count, 0, 2 * sizeof(unsigned), st)); }
    if (!env->pair_total) { HIP_TRY(dmalloc(&env->pair_total, 1)); HIP_TRY(hipMemsetAsync(env->pair_total, 0, sizeof(unsigned long long), st)); }
    if (!env->pmi_flags) { HIP_TRY(dmalloc(&env->pmi_flags, 2)); HIP_TRY(hipMemsetAsync(env->pmi_flags, 0, 2 * sizeof(unsigned), st)); }
    if (cap <= env->pmi_steps_cap) return 0;
    HIP_TRY(hipStreamSynchronize(st));
    // the larger buffers first, the old ones released only once all of them exist: a failed allocation leaves the handle
    // as it was (still good for chunks of the old size)
    const size_t S = (size_t)cap;
    uint2 *n_pairs = nullptr; float *n_scores = nullptr, *n_obs = nullptr, *n_rsum = nullptr; uint32_t *n_rec = nullptr;
    hipError_t e = dmalloc(&n_pairs, S * pairs_step);
    if (e == hipSuccess) e = dmalloc(&n_scores, S * pairs_step);
    if (e == hipSuccess) e = dmalloc(&n_rec, S * BN * nbrec_words(c.n_uav));
    if (e == hipSuccess) e = dmalloc(&n_obs, S * BN * UAVTRACK_OBS_DIM);
    if (e == hipSuccess) e = dmalloc(&n_rsum, S * (size_t)c.n_envs);
    if (e != hipSuccess) {
        void *fresh[] = {n_pairs, n_scores, n_rec, n_obs, n_rsum};
        for (void *q : fresh)
            if (q) (void)hipFree(q);
        (void)hipGetLastError();
        return fail("MAAC-R scratch for %lld steps per chunk (%.1f MiB) could not be allocated: %s; lower UAVTRACK_PMI_SCRATCH_MB",
                    (long long)cap, (double)(S * per_step) / 1048576.0, hipGetErrorString(e));
    }
    void *old[] = {env->pairs, env->scores, env->nbrec, env->obs_tmp, env->rsum};
    for (void *q : old)
        if (q) (void)hipFree(q);
    env->pairs = n_pairs; env->scores = n_scores; env->nbrec = n_rec; env->obs_tmp = n_obs; env->rsum = n_rsum;
    env->pmi_steps_cap = (int32_t)cap;
    return 0;
}

void fold_constants(const uavtrack_config &c, StepParams &p, float *climb_c, float *climb_s)
{
    p.B = c.n_envs; p.N = c.n_uav; p.M = c.m_targets;
    p.na = c.na; p.na_total = c.na * c.nc; p.horizon = c.horizon;
    p.x_max = (float)c.x_max; p.y_max = (float)c.y_max; p.z_max = (float)c.z_max;
    p.x_max_d = c.x_max; p.y_max_d = c.y_max; p.z_max_d = c.z_max;
    p.dtv_u = (float)(c.dt * c.u_v_max);
    p.dtv_t = (float)(c.dt * c.t_v_max);
    p.turn_unit = (float)(c.dt * c.u_h_max / (double)(c.na - 1));
    p.inv_dc = (float)(1.0 / c.dc);
    p.inv_dp = (float)(1.0 / c.dp);
    p.dp = (float)c.dp;
    p.dp2 = (float)(c.dp * c.dp);
    p.dc2 = (float)(c.dc * c.dc);
    p.two_dp2 = (float)(4.0 * c.dp * c.dp);
    {   // pk_le_mask constants: S = 2^k with ulp(smallest K) * S >= 1
        const float kmin = fminf(p.dp2, fminf(p.dc2, p.two_dp2));
        int e = 0;
        std::frexp(kmin, &e);                       // kmin = m * 2^e, m in [0.5, 1): ulp = 2^(e - 24)
        int k = 25 - e;                             // 2 / ulp: the strict form's smallest positive value is ulp/2 * S
        if (k < 0) k = 0;
        const float S = std::ldexp(1.0f, k);
        p.le_neg_scale = -S;
        p.le_dp2 = std::nextafterf(p.dp2, INFINITY) * S;
        p.le_dc2 = std::nextafterf(p.dc2, INFINITY) * S;
        p.le_two_dp2 = std::nextafterf(p.two_dp2, INFINITY) * S;
        p.lt_dp2 = p.dp2 * S;
    }
    p.vratio = (float)(c.t_v_max / c.u_v_max);
    p.inv_na_total = (float)(1.0 / (double)(c.na * c.nc));
    p.inv_na = (float)(1.0 / (double)c.na);
    {   // act_bias_shape() (step_kernel.hip): K = 2^k > n_uav * na * nc; validate() keeps n_uav * (K + na * nc) < 2^24 where it is used
        int k = 1;
        while ((int64_t)1 << k <= (int64_t)c.n_uav * c.na * c.nc && k < 30) ++k;
        p.act_bias = std::ldexp(1.0f, k);
        p.inv_act_bias = std::ldexp(1.0f, -k);
    }
    const double log2e = 1.4426950408889634;
    p.exp_k0 = (float)log2e;
    p.exp_k1 = (float)(log2e / (2.0 * c.dp));
    {   // sym_dup (step_kernel.hip) accumulates round(g * 2^b) as the low mantissa bits of 1.5 * 2^23 + g * 2^b: each term
        // (at most e * 2^b) must stay below 2^22, and n_uav of them below 2^32
        const int b = kSymBits;
        p.sym_k0 = (float)(log2e + b);
    }
    p.tt_ceil = (float)(2.0 * c.norm_m_targets);
    p.inv_tt_ceil = (float)(1.0 / (2.0 * c.norm_m_targets));
    // duplicate punishment (uav.py:214-229: -0.5 sum g) clipped to [-e/2 N, 0] and normalised to [-1, 0] (environment.py:210,
    // data_util.py choice -1: (v - floor) / (0 - floor) - 1 = v / |floor|), as ONE multiply and a clamp: clamp(k sum g, -1, 0)
    const double dup_k = -0.5 / (M_E / 2.0 * c.norm_n_uav);
    p.dup_k = (float)dup_k;
    p.sym_dup_k = (float)std::ldexp(dup_k, -kSymBits);
    p.alpha = (float)c.alpha; p.beta = (float)c.beta; p.gamma = (float)c.gamma;
    p.coop = (float)c.cooperative;
    for (int k = 0; k < UAVTRACK_MAX_CLIMB; ++k) {
        double g = 0.0;
        if (c.nc > 1 && k < c.nc) g = (2.0 * k - (c.nc - 1)) * c.u_g_max / (double)(c.nc - 1);
        climb_c[k] = (float)std::cos(g);
        climb_s[k] = (float)std::sin(g);
    }
}

// ---- Environment.step for a HOST caller (uavtrack_step_host) ----------------------------------------------------
struct HostLayout {      // offsets into the host block, in 4-byte units
    size_t actions, obs, reward, terms, raw, covered, done, state, total;
};
HostLayout host_layout(const uavtrack_config &c)
{
    const size_t BN = (size_t)c.n_envs * c.n_uav, B = (size_t)c.n_envs;
    HostLayout L;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += (n + 3) & ~(size_t)3; return at; };      // 16-byte aligned pieces
    L.actions = take(BN);
    L.obs = take(BN * UAVTRACK_OBS_DIM);
    L.reward = take(BN);
    L.terms = take(3 * BN);
    L.raw = take(BN);
    L.covered = take(B);
    L.done = take((B + 3) / 4);
    L.state = take(state_slab_floats(c.n_envs, c.n_uav, c.m_targets, c.dim == 3));
    L.total = o;
    return L;
}

}  // namespace

extern "C" {

int uavtrack_version(void) { return UAVTRACK_ABI_VERSION; }

const char *uavtrack_last_error(void) { return g_err.c_str(); }

int uavtrack_create(const uavtrack_config *cfg, uavtrack_env **out)
{
    if (!cfg || !out) return fail("uavtrack_create: null argument");
    *out = nullptr;
    if (validate(*cfg)) return 1;

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail("uavtrack_create: no HIP device visible (%s); libuavtrack has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail("uavtrack_create: device_id %d out of range [0, %d)", cfg->device_id, ndev);
    ON_DEVICE(cfg->device_id);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("uavtrack_create: device %d is %s; this library is built for gfx950 only", cfg->device_id,
                    prop.gcnArchName);

    uavtrack_env *env = new (std::nothrow) uavtrack_env();
    if (!env) return fail("uavtrack_create: out of host memory");
    env->cfg = *cfg;
    memset(&env->base, 0, sizeof env->base);
    float climb[2 * UAVTRACK_MAX_CLIMB];
    fold_constants(*cfg, env->base, climb, climb + UAVTRACK_MAX_CLIMB);
    if (!std::isfinite(env->base.le_dp2) || !std::isfinite(env->base.le_dc2) || !std::isfinite(env->base.le_two_dp2)) {
        delete env;
        return fail("uavtrack_create: dp=%g and dc=%g are too far apart for the fp32 range tests", cfg->dp, cfg->dc);
    }
    env->geo = plan_geometry(*cfg, prop.multiProcessorCount * 4);
    env->n_cus = prop.multiProcessorCount;
    if (env->geo.wgs == 0) {
        delete env;
        return fail("uavtrack_create: no workgroup geometry for n_uav=%d, m_targets=%d: one environment's tables must fit the %zu KiB of "
                    "LDS of a CU", cfg->n_uav, cfg->m_targets, kLdsMax / 1024);
    }
    // specialised swarms of up to 64 UAVs count neighbours in the high part of an exact fp32 integer sum (step_kernel.hip
    // act_bias_shape): n_uav * (K + na * nc) must stay below 2^24 there (the generic kernel keeps a separate count)
    if (env->geo.specialised && cfg->n_uav <= 64 && (int64_t)cfg->n_uav * cfg->n_uav * cfg->na * cfg->nc * 4 >= ((int64_t)1 << 24)) {
        delete env;
        return fail("uavtrack_create: na * nc = %d actions is beyond what the specialised kernel of a %d-UAV swarm supports "
                    "(n_uav^2 * na * nc must stay below 2^22)", cfg->na * cfg->nc, cfg->n_uav);
    }
    env->base.E = env->geo.envs_per_wg;
    env->geo_short = cfg->reward_mode == UAVTRACK_REWARD_PMI ? plan_geometry(*cfg, prop.multiProcessorCount * 4, false) : env->geo;
    if (env->geo_short.wgs == 0 || env->geo_short.lds_bytes > kLdsMax) env->geo_short = env->geo;
    if (env->geo.lds_bytes > kLdsMax) {      // (plan_geometry already turns such shapes away: a second line of defence)
        const size_t need = env->geo.lds_bytes;
        delete env;
        return fail("uavtrack_create: one environment's tables need %zu B of LDS, a gfx950 CU has %zu (n_uav=%d, m_targets=%d)", need,
                    kLdsMax, cfg->n_uav, cfg->m_targets);
    }

    const size_t nfl = state_slab_floats(cfg->n_envs, cfg->n_uav, cfg->m_targets, cfg->dim == 3);
    hipError_t err = dmalloc(&env->slab, nfl);
    if (err == hipSuccess) err = hipMemset(env->slab, 0, nfl * sizeof(float));
    if (err == hipSuccess) {
        env->state = state_view(env->slab, cfg->n_envs, cfg->n_uav, cfg->m_targets, cfg->dim == 3);
        err = hipMemcpy(env->state.climb_c, climb, sizeof climb, hipMemcpyHostToDevice);
        env->base.slab = env->slab;
    }
    if (err != hipSuccess) {
        free_state(env);
        delete env;
        return fail("uavtrack_create: device allocation failed: %s", hipGetErrorString(err));
    }
    *out = env;
    return 0;
}

int uavtrack_destroy(uavtrack_env *env)
{
    if (!env) return 0;
    DeviceGuard guard(env->cfg.device_id);
    free_state(env);
    delete env;
    return 0;
}

int uavtrack_reset(uavtrack_env *env, uint64_t seed, uint32_t episode, float *obs, void *stream)
{
    if (!env) return fail("uavtrack_reset: null handle");
    ON_DEVICE(env->cfg.device_id);
    HIP_TRY(launch_reset(env, seed, episode, obs, static_cast<hipStream_t>(stream)));
    return 0;
}

int uavtrack_set_state(uavtrack_env *env, const float *ux, const float *uy, const float *uz, const float *uh,
                       const int32_t *ua, const float *tx, const float *ty, const float *tz, const float *th,
                       const int32_t *step_count, void *stream)
{
    if (!env) return fail("uavtrack_set_state: null handle");
    if (!ux || !uy || !uh || !ua) return fail("uavtrack_set_state: null UAV array");
    const uavtrack_config &c = env->cfg;
    if (c.m_targets > 0 && (!tx || !ty || !th)) return fail("uavtrack_set_state: null target array");
    if (c.dim == 3 && (!uz || (c.m_targets > 0 && !tz))) return fail("uavtrack_set_state: dim == 3 needs uz and tz");
    ON_DEVICE(c.device_id);
    hipStream_t st = static_cast<hipStream_t>(stream);
    StateBlock &s = env->state;
    const size_t BN = (size_t)c.n_envs * c.n_uav * 4, BM = (size_t)c.n_envs * c.m_targets * 4;
    HIP_TRY(hipMemcpyAsync(s.ux, ux, BN, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.uy, uy, BN, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.uh, uh, BN, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(s.ua, ua, BN, hipMemcpyDeviceToDevice, st));
    if (BM) {
        HIP_TRY(hipMemcpyAsync(s.tx, tx, BM, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(s.ty, ty, BM, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(s.th, th, BM, hipMemcpyDeviceToDevice, st));
    }
    if (c.dim == 3) {
        HIP_TRY(hipMemcpyAsync(s.uz, uz, BN, hipMemcpyDeviceToDevice, st));
        if (BM) HIP_TRY(hipMemcpyAsync(s.tz, tz, BM, hipMemcpyDeviceToDevice, st));
    }
    if (step_count)
        HIP_TRY(hipMemcpyAsync(s.step_count, step_count, (size_t)c.n_envs * 4, hipMemcpyDeviceToDevice, st));
    else
        HIP_TRY(hipMemsetAsync(s.step_count, 0, (size_t)c.n_envs * 4, st));
    return 0;
}

int uavtrack_get_state(uavtrack_env *env, float *ux, float *uy, float *uz, float *uh, int32_t *ua, float *tx,
                       float *ty, float *tz, float *th, int32_t *step_count, void *stream)
{
    if (!env) return fail("uavtrack_get_state: null handle");
    const uavtrack_config &c = env->cfg;
    ON_DEVICE(c.device_id);
    hipStream_t st = static_cast<hipStream_t>(stream);
    StateBlock &s = env->state;
    const size_t BN = (size_t)c.n_envs * c.n_uav * 4, BM = (size_t)c.n_envs * c.m_targets * 4;
    if (ux) HIP_TRY(hipMemcpyAsync(ux, s.ux, BN, hipMemcpyDeviceToDevice, st));
    if (uy) HIP_TRY(hipMemcpyAsync(uy, s.uy, BN, hipMemcpyDeviceToDevice, st));
    if (uh) HIP_TRY(hipMemcpyAsync(uh, s.uh, BN, hipMemcpyDeviceToDevice, st));
    if (ua) HIP_TRY(hipMemcpyAsync(ua, s.ua, BN, hipMemcpyDeviceToDevice, st));
    if (BM) {
        if (tx) HIP_TRY(hipMemcpyAsync(tx, s.tx, BM, hipMemcpyDeviceToDevice, st));
        if (ty) HIP_TRY(hipMemcpyAsync(ty, s.ty, BM, hipMemcpyDeviceToDevice, st));
        if (th) HIP_TRY(hipMemcpyAsync(th, s.th, BM, hipMemcpyDeviceToDevice, st));
    }
    if (c.dim == 3) {
        if (uz) HIP_TRY(hipMemcpyAsync(uz, s.uz, BN, hipMemcpyDeviceToDevice, st));
        if (tz && BM) HIP_TRY(hipMemcpyAsync(tz, s.tz, BM, hipMemcpyDeviceToDevice, st));
    }
    if (step_count)
        HIP_TRY(hipMemcpyAsync(step_count, s.step_count, (size_t)c.n_envs * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

int uavtrack_set_episodes(uavtrack_env *env, const int32_t *episode, void *stream)
{
    if (!env) return fail("uavtrack_set_episodes: null handle");
    if (!episode) return fail("uavtrack_set_episodes: episode is null");
    ON_DEVICE(env->cfg.device_id);
    HIP_TRY(hipMemcpyAsync(env->state.episode, episode, (size_t)env->cfg.n_envs * 4, hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    return 0;
}

int uavtrack_get_episodes(uavtrack_env *env, int32_t *episode, void *stream)
{
    if (!env) return fail("uavtrack_get_episodes: null handle");
    if (!episode) return fail("uavtrack_get_episodes: episode is null");
    ON_DEVICE(env->cfg.device_id);
    HIP_TRY(hipMemcpyAsync(episode, env->state.episode, (size_t)env->cfg.n_envs * 4, hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    return 0;
}

int uavtrack_set_pmi_weights(uavtrack_env *env, const float *folded, size_t n_floats, int32_t hidden, void *stream)
{
    if (!env) return fail("uavtrack_set_pmi_weights: null handle");
    ON_DEVICE(env->cfg.device_id);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!folded) {
        HIP_TRY(hipStreamSynchronize(st));
        if (env->pmi.blob) (void)hipFree(env->pmi.blob);
        env->pmi = PmiWeights();
        return 0;
    }
    if (hidden < 1 || hidden > kPmiMaxHidden)
        return fail("uavtrack_set_pmi_weights: hidden %d outside [1, %d] (PMINetwork takes any hidden_dim, PMINet.py:19; the "
                    "MFMA scorer is built for widths up to %d)", hidden, kPmiMaxHidden, kPmiMaxHidden);
    const size_t H = (size_t)hidden;
    const size_t want = 12 * H + 3 * H + 3 * H * H + H + H + 1;
    if (n_floats != want)
        return fail("uavtrack_set_pmi_weights: n_floats %zu != %zu for hidden %d", n_floats, want, hidden);
    // The scorer tiles the hidden layer in blocks of 32 columns: other widths are padded with units whose weights
    // and biases are zero (relu(0) = 0 adds nothing to any sum), so every hidden_dim runs on the same kernels.
    const int hp = pmi_padded_hidden(hidden);
    const size_t HP = (size_t)hp, n_dev = 12 * HP + 3 * HP + 3 * HP * HP + HP + HP + 1;
    const size_t x6_off = (n_dev + 3) & ~(size_t)3, x6_len = pmi_x6_floats(hp);     // the bf16 planes, 16-B aligned
    const size_t l1_off = x6_off + x6_len, l1_len = pmi_l1_floats(hp);               // the branch layers' f16 planes behind them
    const size_t t3_off = l1_off + l1_len, t3_len = pmi_t3_floats(hp);               // ... and fc1 block-scaled (pmi_score_t3_kernel)

    // ---- everything the host can decide comes FIRST: a call that fails leaves the handle's previous weights in place
    bool h3_ok = t3_len != 0;
    float s1 = 1.0f, tw = 1.0f;
    float rng_inv[3] = {0.0f, 0.0f, 0.0f};
    std::vector<float> padded(n_dev, 0.0f), packed(n_dev);
    {
        const float *src = folded;
        float *dst = padded.data();
        auto rows = [&](size_t nrows, size_t in_w, size_t out_w) {      // nrows rows of in_w floats -> rows of out_w
            for (size_t r = 0; r < nrows; ++r) memcpy(dst + r * out_w, src + r * in_w, in_w * sizeof(float));
            src += nrows * in_w;
            dst += nrows * out_w;
        };
        rows(5 + 1, H, HP);                       // Wc[5][H] bc[H]
        rows(4 + 1, H, HP);                       // Wo[4][H] bo[H]
        rows(3 + 1, H, HP);                       // Wb[3][H] bb[H]
        for (int br = 0; br < 3; ++br) {          // W1[3H][H]: the three branch blocks of its input, each padded to HP rows
            rows(H, H, HP);
            dst += (HP - H) * HP;
        }
        rows(1, H, HP);                           // b1[H]
        rows(1, H, HP);                           // w2[H]
        rows(1, 1, 1);                            // b2
    }
    // The f16 x 3 scorer (pmi_score_t3_kernel) needs every MFMA operand inside f16's range (65504): the fc1 weights
    // are known here, the branch activations are bounded from the ranges of the observation products x = la_i * la_j
    // (uav.py:156-197: normalised offsets and action differences within [-1, 1], heading terms within +-(1 + v_t/v_u),
    // positions / dc taken up to three field lengths outside the box).  A network beyond half that range keeps the
    // bf16 x 6 kernel (bf16 has fp32's exponent).
    if (h3_ok) {
        const uavtrack_config &c = env->cfg;
        const double vr = 1.0 + c.t_v_max / c.u_v_max, pos = 4.0 * std::fmax(c.x_max, c.y_max) / c.dc;
        const double xb[12] = {1, 1, 4, 4, 1, 1, 1, vr * vr, vr * vr, pos * pos, pos * pos, 1};
        const float *pw = padded.data();
        double act_max = 0.0, w_max = 0.0, w1_max = 0.0;
        double gain[3] = {0.0, 0.0, 0.0}, bias[3] = {0.0, 0.0, 0.0};      // per branch: max_u sum_k |w_uk|, max_u |b_u|
        const int fan[3] = {5, 4, 3};
        int k0 = 0;
        for (int br = 0; br < 3; ++br) {              // W[fan][HP] then b[HP]
            for (size_t u = 0; u < HP; ++u) {
                double a = std::fabs(pw[(size_t)fan[br] * HP + u]), gsum = 0.0;
                bias[br] = std::fmax(bias[br], a);
                for (int k = 0; k < fan[br]; ++k) {
                    a += std::fabs(pw[(size_t)k * HP + u]) * xb[k0 + k];
                    gsum += std::fabs(pw[(size_t)k * HP + u]);
                }
                gain[br] = std::fmax(gain[br], gsum);
                act_max = std::fmax(act_max, a);
            }
            pw += (size_t)(fan[br] + 1) * HP;
            k0 += fan[br];
        }
        for (size_t k = 0; k < 3 * HP * HP; ++k) w1_max = std::fmax(w1_max, std::fabs(pw[k]));
        w_max = w1_max;
        for (size_t k = 0; k < 15 * HP; ++k) w_max = std::fmax(w_max, std::fabs(padded[k]));      // the branch layers (MFMA operands of pmi_score_t3_kernel)
        w_max = std::fmax(w_max, pos * pos);                                                         // ... and their inputs
        h3_ok = std::isfinite(act_max) && act_max < 32000.0 && w_max < 32000.0;
        // Block scales of the t3 planes, powers of two: T * max |fc1 weight| just below 32000 (the weights are known
        // exactly); S1 * (activation bound) below 512 -- the bound comes from nominal observation ranges, and the
        // uav.py:165 weight lets a UAV next to the origin exceed them, so the activations keep a factor 128 of
        // headroom to f16's 65504 (the unscaled h3 planes have 65504 / bound).  Remainders x - f16(x) of values within
        // 2^-12 (activations) / 2^-18 (weights) of those sizes are normal f16 numbers.
        auto scale_for = [](double bound, double target) {
            int e = 15;
            if (bound > 0.0) e = (int)std::floor(std::log2(target / bound));
            return std::ldexp(1.0f, e < -6 ? -6 : (e > 15 ? 15 : e));
        };
        s1 = scale_for(act_max, 512.0);
        tw = scale_for(w1_max, 32000.0);
        // The run-time watch of the f16 kernel (pmi_kernel.hip, PmiParams::rng_inv).  The bounds above come from NOMINAL
        // observation ranges; the uav.py:165 weight 1 / min(d, 1) lets a UAV next to the origin exceed them without
        // limit.  An activation of branch br stays below 60000 / S1 while |x| <= (60000 / S1 - max|b|) / max_u sum_k|w_uk|
        // over the branch's inputs, and an input splits into normal f16 planes below 30000 (its remainder is scaled by 2^11):
        // the kernel compares the largest |x| of a tile with the smaller of the two and has the chunk re-scored by the
        // bf16 kernel when it is exceeded.
        for (int br = 0; br < 3; ++br) {
            double lim = 30000.0;
            if (gain[br] > 0.0) lim = std::fmin(lim, (60000.0 / (double)s1 - bias[br]) / gain[br]);
            rng_inv[br] = lim > 0.0 ? (float)(1.0 / lim) : INFINITY;
        }
    }
    // a pinned scorer (uavtrack_set_pmi_scheme) that cannot take these weights: refused before anything is replaced
    if (!pmi_scheme_fits(hp, h3_ok, env->pmi_scheme))
        return fail("uavtrack_set_pmi_weights: these weights (hidden %d%s) cannot run on the pinned scorer scheme %d "
                    "(uavtrack_set_pmi_scheme); pin UAVTRACK_PMI_AUTO or a scheme that takes them; the previous weights stay loaded",
                    hidden, h3_ok ? "" : ", beyond f16's range", env->pmi_scheme);
    pack_pmi_blob(padded.data(), packed.data(), hp);
    std::vector<uint16_t> planes(x6_len * 2), planes1(h3_ok ? l1_len * 2 : 0), planes3t(h3_ok ? t3_len * 2 : 0);
    if (x6_len) pack_pmi_x6(padded.data(), planes.data(), hp);
    if (h3_ok) {
        pack_pmi_l1(padded.data(), planes1.data(), hp, s1);
        pack_pmi_t3(padded.data(), planes3t.data(), hp, tw);
    }

    // ---- device side: a blob of another size is allocated BEFORE the old one goes; the uploads are complete before the
    //      host vectors die and before the new weights are published in the handle
    float *blob = env->pmi.blob;
    const bool fresh = env->pmi.n_floats != n_dev || !blob;
    HIP_TRY(hipStreamSynchronize(st));          // (launches that still read the current weights)
    if (fresh) {
        blob = nullptr;
        HIP_TRY(dmalloc(&blob, t3_off + t3_len));
    }
    hipError_t e = hipMemcpyAsync(blob, packed.data(), n_dev * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && x6_len) e = hipMemcpyAsync(blob + x6_off, planes.data(), x6_len * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && h3_ok) e = hipMemcpyAsync(blob + l1_off, planes1.data(), l1_len * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && h3_ok) e = hipMemcpyAsync(blob + t3_off, planes3t.data(), t3_len * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        if (fresh) (void)hipFree(blob);          // (an in-place upload that failed half way cannot be undone: the handle loses its weights)
        else { (void)hipFree(env->pmi.blob); env->pmi = PmiWeights(); }
        (void)hipGetLastError();
        return fail("uavtrack_set_pmi_weights: upload failed: %s", hipGetErrorString(e));
    }
    if (fresh && env->pmi.blob) (void)hipFree(env->pmi.blob);
    env->pmi.blob = blob;
    env->pmi.x6 = x6_len ? blob + x6_off : nullptr;
    env->pmi.l1 = h3_ok ? blob + l1_off : nullptr;
    env->pmi.t3 = h3_ok ? blob + t3_off : nullptr;
    env->pmi.t3_s1 = s1;
    env->pmi.t3_t = tw;
    for (int k = 0; k < 3; ++k) env->pmi.rng_inv[k] = rng_inv[k];
    env->pmi.hidden = hp;
    env->pmi.n_floats = n_dev;
    // scratch for an episode's worth of deferred scoring (bounded by UAVTRACK_PMI_SCRATCH_MB): the stepping calls then never
    // allocate or synchronise, whatever their length up to cfg.horizon -- which also makes them capturable into a HIP graph
    if (ensure_pmi_scratch(env, env->cfg.horizon > 0 ? env->cfg.horizon : 1, st)) return 1;
    return 0;
}

int uavtrack_pmi_inference(uavtrack_env *env, const float *x, int64_t n, float *scores, void *stream)
{
    if (!env) return fail("uavtrack_pmi_inference: null handle");
    if (n < 0 || n >= ((int64_t)1 << 31)) return fail("uavtrack_pmi_inference: n must be in [0, 2^31) (got %lld)", (long long)n);
    if (n == 0) return 0;
    if (!x || !scores) return fail("uavtrack_pmi_inference: x and scores must not be null");
    if (!env->pmi.blob) return fail("uavtrack_pmi_inference: needs uavtrack_set_pmi_weights first");
    ON_DEVICE(env->cfg.device_id);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((size_t)n > env->inf_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (env->inf_obs) (void)hipFree(env->inf_obs);
        if (env->inf_pairs) (void)hipFree(env->inf_pairs);
        env->inf_obs = nullptr; env->inf_pairs = nullptr; env->inf_cap = 0;
        HIP_TRY(dmalloc(&env->inf_obs, (size_t)n * 2 * UAVTRACK_OBS_DIM));
        HIP_TRY(dmalloc(&env->inf_pairs, (size_t)n));
        env->inf_cap = (size_t)n;
    }
    // the scorer reads its pair count from the device counter the rollout kernel normally fills (zero between calls)
    HIP_TRY(launch_pmi_inference_prep(x, env->inf_obs, env->inf_pairs, (unsigned)n, st));
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(env->pair_count), (int)n, 1, st));
    HIP_TRY(launch_pmi_score(env, env->inf_obs, st, env->inf_pairs, scores, 2));
    HIP_TRY(launch_pmi_counters_reset(env, st));
    return 0;
}

int uavtrack_set_pmi_scheme(uavtrack_env *env, int32_t scheme)
{
    if (!env) return fail("uavtrack_set_pmi_scheme: null handle");
    if (scheme < UAVTRACK_PMI_AUTO || scheme > UAVTRACK_PMI_FP32)
        return fail("uavtrack_set_pmi_scheme: scheme %d is not one of enum uavtrack_pmi_scheme", scheme);
    if (env->pmi.blob && !pmi_scheme_available(env, scheme))
        return fail("uavtrack_set_pmi_scheme: the loaded weights (hidden %d padded%s) cannot run on scheme %d: F16X3 and BF16X6 take "
                    "widths 64 / 96 / 128, F16X3 only networks inside f16's range", env->pmi.hidden, env->pmi.t3 ? "" : ", beyond f16's range", scheme);
    env->pmi_scheme = scheme;
    return 0;
}

int uavtrack_pmi_info(uavtrack_env *env, int64_t out[4], void *stream)
{
    if (!env || !out) return fail("uavtrack_pmi_info: null argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!env->pmi.blob) return 0;
    out[0] = pmi_effective_scheme(env);
    out[1] = env->pmi.hidden;
    out[2] = env->pmi.t3 ? 1 : 0;
    if (env->pmi_flags) {
        ON_DEVICE(env->cfg.device_id);
        hipStream_t st = static_cast<hipStream_t>(stream);
        unsigned v[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(v, env->pmi_flags, sizeof v, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        out[3] = v[1];
    }
    return 0;
}

// Where a rollout's actions come from: the caller's tensor, or the in-kernel actor (uavtrack_run_actor).
struct PolicyArgs {
    bool auto_reset = false;           // uavtrack_step_many_autoreset: reset seed
    uint64_t reset_seed = 0;
    int policy = kPolicyGiven;
    const float *obs_in = nullptr;     // actor: observation seen at the first step
    int32_t *actions_out = nullptr;    // actor: chosen actions [T][B][N], nullable
    uint64_t seed = 0;
    int32_t mode = 0;
};

static int run_steps(uavtrack_env *env, int32_t T, const int32_t *actions, float *obs, float *reward, float *terms,
                     int32_t *covered, uint8_t *done, float *ep_sums, void *stream, const char *who,
                     bool accumulate = false, const PolicyArgs &pol = PolicyArgs())
{
    if (!env) return fail("%s: null handle", who);
    if (T < 1) return fail("%s: T must be >= 1 (got %d)", who, T);
    if (pol.policy == kPolicyGiven && !actions) return fail("%s: actions is null", who);
    if (!reward) return fail("%s: reward is null", who);
    ON_DEVICE(env->cfg.device_id);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (env->tpos && T > env->tpos_steps)
        return fail("%s: T = %d exceeds the %d steps the target-trace buffer holds (uavtrack_set_target_trace)", who, T, env->tpos_steps);
    if (env->raw_out && T > env->raw_steps)
        return fail("%s: T = %d exceeds the %d steps the raw-reward buffer holds (uavtrack_set_raw_reward_output)", who, T, env->raw_steps);
    StepParams p = env->base;
    p.actions = actions;
    p.tpos = env->tpos;
    p.raw = env->raw_out;
    p.state_copy = env->state_copy_out;
    p.obs = obs; p.reward = reward; p.terms = terms; p.nbrec = nullptr;
    p.covered = covered; p.done = done; p.ep_sums = ep_sums;
    p.pairs = nullptr; p.pair_count = nullptr; p.pair_total = nullptr;
    p.ep_accumulate = accumulate ? 1 : 0;
    p.actions_out = pol.actions_out;
    p.env_offset = env->cfg.env_offset;
    p.auto_reset = pol.auto_reset ? 1 : 0;
    p.reset_k0 = (uint32_t)pol.reset_seed; p.reset_k1 = (uint32_t)(pol.reset_seed >> 32);
    if (pol.policy == kPolicyActor) {
        p.greedy_k0 = (uint32_t)pol.seed; p.greedy_k1 = (uint32_t)(pol.seed >> 32);
        p.obs_in = pol.obs_in; p.actor_w = env->actor_w; p.actor_hblocks = actor_blocks(env->actor_hidden);
        p.actor_mode = pol.mode;
    }
    if (env->cfg.reward_mode != UAVTRACK_REWARD_PMI) {
        p.T = T;
        HIP_TRY(timed_launch(env, UAVTRACK_PROF_ROLLOUT, st, [&] { return launch_rollout(env, p, st, pol.policy); }));
        return 0;
    }
    // MAAC-R.  Rewards never feed back into the dynamics, so scoring is deferred: a chunk of steps is
    // simulated by ONE fused launch (observations, raw rewards, poses, and the neighbour pairs of every
    // step of the chunk), then ONE launch of the MFMA scorer over all those pairs, then ONE launch of the
    // softmax mix.  T = 1 (closed loop) is a chunk of one.  Everything is stream-ordered.
    if (!env->pmi.blob) return fail("%s: reward_mode PMI needs uavtrack_set_pmi_weights first", who);
    if (ensure_pmi_scratch(env, T, st)) return 1;
    const uavtrack_config &c = env->cfg;
    const size_t BN = (size_t)c.n_envs * c.n_uav;
    // episode sums: the three terms and the coverage are summed by the rollout kernel (registers, as in the other modes),
    // the return by the mix stage (per-step means) and a one-lane-per-environment reduction over the chunk's steps
    p.ep_sums = ep_sums;
    p.nbrec = env->nbrec;
    p.pairs = env->pairs;
    p.pair_count = env->pair_count;      // zero: set at allocation, re-zeroed by the mix kernel
    p.pair_total = env->pair_total;
    bool add = accumulate;
    for (int32_t t0 = 0; t0 < T; t0 += env->pmi_steps_cap) {
        const int32_t n = (T - t0 < env->pmi_steps_cap) ? T - t0 : env->pmi_steps_cap;
        float *obs_t = obs ? obs + (size_t)t0 * BN * UAVTRACK_OBS_DIM : env->obs_tmp;
        float *terms_t = terms ? terms + (size_t)t0 * 3 * BN : nullptr;
        float *reward_t = reward + (size_t)t0 * BN;
        int32_t *covered_t = covered ? covered + (size_t)t0 * c.n_envs : nullptr;
        p.T = n;
        p.ep_accumulate = add ? 1 : 0;
        p.actions = actions ? actions + (size_t)t0 * BN : nullptr;
        p.actions_out = pol.actions_out ? pol.actions_out + (size_t)t0 * BN : nullptr;
        p.obs = obs_t; p.reward = reward_t; p.terms = terms_t;
        p.covered = covered_t;
        p.done = done ? done + (size_t)t0 * c.n_envs : nullptr;
        p.tpos = env->tpos ? env->tpos + (size_t)t0 * c.n_envs * c.m_targets : nullptr;
        p.raw = env->raw_out ? env->raw_out + (size_t)t0 * BN : nullptr;
        // (short launches: 4-wave groups; long ones: the handle's geometry where its kernel variant exists -- launch_rollout)
        const Geometry *geo = n < kPmiShortLaunch ? &env->geo_short : nullptr;
        HIP_TRY(timed_launch(env, UAVTRACK_PROF_ROLLOUT, st, [&] { return launch_rollout(env, p, st, pol.policy, geo); }));
        // the actor of the next chunk starts from this chunk's last observation (a lane reads its own row
        // once, at launch start, before it writes anything: the scratch buffer may be reused in place)
        p.obs_in = obs_t + (size_t)(n - 1) * BN * UAVTRACK_OBS_DIM;
        HIP_TRY(timed_launch(env, UAVTRACK_PROF_SCORER, st, [&] { return launch_pmi_score(env, obs_t, st); }));
        HIP_TRY(timed_launch(env, UAVTRACK_PROF_MIX, st, [&] { return launch_pmi_finalize(env, n, reward_t, ep_sums ? env->rsum : nullptr, st); }));
        if (ep_sums) {
            HIP_TRY(timed_launch(env, UAVTRACK_PROF_EPSUMS, st, [&] { return launch_ep_reward(env, n, env->rsum, ep_sums, add, st); }));
            add = true;
        }
    }
    return 0;
}

int uavtrack_step(uavtrack_env *env, const int32_t *actions, float *obs, float *reward, float *terms,
                  int32_t *covered, uint8_t *done, void *stream)
{
    return run_steps(env, 1, actions, obs, reward, terms, covered, done, nullptr, stream, "uavtrack_step");
}

int uavtrack_step_accumulate(uavtrack_env *env, const int32_t *actions, float *obs, float *reward, float *terms,
                             int32_t *covered, uint8_t *done, float *ep_sums, void *stream)
{
    if (!ep_sums) return fail("uavtrack_step_accumulate: ep_sums is null");
    return run_steps(env, 1, actions, obs, reward, terms, covered, done, ep_sums, stream,
                     "uavtrack_step_accumulate", true);
}

int uavtrack_step_many(uavtrack_env *env, int32_t T, const int32_t *actions, float *obs, float *reward,
                       float *terms, int32_t *covered, uint8_t *done, float *ep_sums, void *stream)
{
    return run_steps(env, T, actions, obs, reward, terms, covered, done, ep_sums, stream, "uavtrack_step_many");
}

int uavtrack_step_many_autoreset(uavtrack_env *env, int32_t T, uint64_t reset_seed, const int32_t *actions, float *obs,
                                 float *reward, float *terms, int32_t *covered, uint8_t *done, float *ep_sums, void *stream)
{
    if (env && env->cfg.horizon < 1) return fail("uavtrack_step_many_autoreset: the configuration has no horizon (done never fires)");
    PolicyArgs pol;
    pol.auto_reset = true; pol.reset_seed = reset_seed;
    return run_steps(env, T, actions, obs, reward, terms, covered, done, ep_sums, stream, "uavtrack_step_many_autoreset", false, pol);
}

int uavtrack_run_greedy(uavtrack_env *env, int32_t T, uint64_t seed, int32_t *actions_out, float *obs, float *reward,
                        float *terms, int32_t *covered, uint8_t *done, float *ep_sums, void *stream)
{
    if (!env) return fail("uavtrack_run_greedy: null handle");
    if (T < 1) return fail("uavtrack_run_greedy: T must be >= 1 (got %d)", T);
    if (!reward) return fail("uavtrack_run_greedy: reward is null");
    if (env->cfg.dim != 2) return fail("uavtrack_run_greedy: the reference baseline is planar (dim must be 2)");
    if (env->cfg.reward_mode == UAVTRACK_REWARD_PMI)
        return fail("uavtrack_run_greedy: the C-METHOD baseline runs with the MAAC / MAAC-G rewards (C-METHOD.yaml: cooperative 0)");
    ON_DEVICE(env->cfg.device_id);
    if (env->tpos && T > env->tpos_steps)
        return fail("uavtrack_run_greedy: T = %d exceeds the %d steps the target-trace buffer holds", T, env->tpos_steps);
    if (env->raw_out && T > env->raw_steps)
        return fail("uavtrack_run_greedy: T = %d exceeds the %d steps the raw-reward buffer holds", T, env->raw_steps);
    StepParams p = env->base;
    p.T = T;
    p.tpos = env->tpos;
    p.raw = env->raw_out;
    p.actions = nullptr; p.actions_out = actions_out;
    p.obs = obs; p.reward = reward; p.terms = terms; p.nbrec = nullptr;
    p.covered = covered; p.done = done; p.ep_sums = ep_sums;
    p.pairs = nullptr; p.pair_count = nullptr; p.ep_accumulate = 0;
    p.env_offset = env->cfg.env_offset;
    p.greedy_k0 = (uint32_t)seed; p.greedy_k1 = (uint32_t)(seed >> 32);
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(timed_launch(env, UAVTRACK_PROF_ROLLOUT, st, [&] { return launch_rollout(env, p, st, kPolicyGreedy); }));
    return 0;
}

int uavtrack_greedy_actions(uavtrack_env *env, uint64_t seed, int32_t *actions, void *stream)
{
    if (!env) return fail("uavtrack_greedy_actions: null handle");
    if (!actions) return fail("uavtrack_greedy_actions: actions is null");
    if (env->cfg.dim != 2) return fail("uavtrack_greedy_actions: the reference baseline is planar (dim must be 2)");
    ON_DEVICE(env->cfg.device_id);
    HIP_TRY(launch_greedy(env, seed, actions, static_cast<hipStream_t>(stream)));
    return 0;
}

int uavtrack_set_actor_weights(uavtrack_env *env, const float *w1, const float *b1, const float *w2, const float *b2,
                               int32_t hidden, void *stream)
{
    if (!env) return fail("uavtrack_set_actor_weights: null handle");
    ON_DEVICE(env->cfg.device_id);
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipStreamSynchronize(st));
    if (!w1) {
        if (env->actor_w) (void)hipFree(env->actor_w);
        env->actor_w = nullptr; env->actor_hidden = 0;
        return 0;
    }
    if (!b1 || !w2 || !b2) return fail("uavtrack_set_actor_weights: b1, w2 and b2 must not be null");
    const int A = env->cfg.na * env->cfg.nc;
    const int mt = actor_tiles(env->cfg.dim == 3);
    if (A > actor_slots(mt))
        return fail("uavtrack_set_actor_weights: na*nc = %d actions; the device actor holds up to %d in %d-D "
                    "(12 = the reference's action space; 48 for the 3-D action space)", A, actor_slots(mt), env->cfg.dim);
    if (hidden < 1 || hidden > 4096) return fail("uavtrack_set_actor_weights: hidden %d out of range [1, 4096]", hidden);
    const size_t n = actor_blob_floats(hidden, mt);
    std::vector<float> blob(n, 0.0f);
    // nominal bounds of the observation entries (uav.py:156-197: normalised offsets and action differences within [-1, 1],
    // heading terms within +-2 / +-(1 + v_t / v_u), positions / dc taken up to four field lengths): they size the block
    // scale of the hidden layer, which keeps a factor 128 of headroom above them and saturates beyond that
    const uavtrack_config &c = env->cfg;
    const double vr = 1.0 + c.t_v_max / c.u_v_max, pos = 4.0 * std::fmax(c.x_max, c.y_max) / c.dc;
    const double xb[12] = {1, 1, 2, 2, 1, 1, 1, vr, vr, pos, pos, 1};
    pack_actor_blob(w1, b1, w2, b2, hidden, A, mt, xb, blob.data());
    if (env->actor_hidden != hidden) {
        if (env->actor_w) (void)hipFree(env->actor_w);
        env->actor_w = nullptr; env->actor_hidden = 0;
        HIP_TRY(dmalloc(&env->actor_w, n));
    }
    HIP_TRY(hipMemcpyAsync(env->actor_w, blob.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    env->actor_hidden = hidden;
    return 0;
}

int uavtrack_actor_actions(uavtrack_env *env, const float *obs, uint64_t seed, int32_t mode, int32_t *actions,
                           float *probs, void *stream)
{
    if (!env) return fail("uavtrack_actor_actions: null handle");
    if (!obs || !actions) return fail("uavtrack_actor_actions: obs and actions must not be null");
    if (!env->actor_w) return fail("uavtrack_actor_actions: needs uavtrack_set_actor_weights first");
    if (mode != UAVTRACK_ACTOR_SAMPLE && mode != UAVTRACK_ACTOR_ARGMAX)
        return fail("uavtrack_actor_actions: mode %d is neither UAVTRACK_ACTOR_SAMPLE nor UAVTRACK_ACTOR_ARGMAX", mode);
    ON_DEVICE(env->cfg.device_id);
    HIP_TRY(launch_actor(env, obs, seed, mode, actions, probs, static_cast<hipStream_t>(stream)));
    return 0;
}

int uavtrack_run_actor(uavtrack_env *env, int32_t T, uint64_t seed, int32_t mode, const float *obs_in,
                       int32_t *actions_out, float *obs, float *reward, float *terms, int32_t *covered,
                       uint8_t *done, float *ep_sums, void *stream)
{
    if (!env) return fail("uavtrack_run_actor: null handle");
    if (T < 1) return fail("uavtrack_run_actor: T must be >= 1 (got %d)", T);
    if (!reward) return fail("uavtrack_run_actor: reward is null");
    if (!obs_in) return fail("uavtrack_run_actor: obs_in is null (the observation the policy sees at the first step)");
    if (!env->actor_w) return fail("uavtrack_run_actor: needs uavtrack_set_actor_weights first");
    if (mode != UAVTRACK_ACTOR_SAMPLE && mode != UAVTRACK_ACTOR_ARGMAX)
        return fail("uavtrack_run_actor: mode %d is neither UAVTRACK_ACTOR_SAMPLE nor UAVTRACK_ACTOR_ARGMAX", mode);
    if (obs_in == obs && T > 1)
        return fail("uavtrack_run_actor: obs_in must not alias obs when T > 1 (pass the previous launch's last rows, "
                    "or a copy)");
    PolicyArgs pol;
    pol.policy = kPolicyActor; pol.obs_in = obs_in; pol.actions_out = actions_out; pol.seed = seed; pol.mode = mode;
    return run_steps(env, T, nullptr, obs, reward, terms, covered, done, ep_sums, stream, "uavtrack_run_actor", false, pol);
}

int uavtrack_set_target_trace(uavtrack_env *env, float *tpos, int32_t capacity_steps)
{
    if (!env) return fail("uavtrack_set_target_trace: null handle");
    if (tpos && capacity_steps < 1) return fail("uavtrack_set_target_trace: capacity_steps must be >= 1 (got %d)", capacity_steps);
    env->tpos = reinterpret_cast<float2 *>(tpos);
    env->tpos_steps = tpos ? capacity_steps : 0;
    return 0;
}

int uavtrack_set_raw_reward_output(uavtrack_env *env, float *raw, int32_t capacity_steps)
{
    if (!env) return fail("uavtrack_set_raw_reward_output: null handle");
    if (raw && capacity_steps < 1) return fail("uavtrack_set_raw_reward_output: capacity_steps must be >= 1 (got %d)", capacity_steps);
    env->raw_out = raw;
    env->raw_steps = raw ? capacity_steps : 0;
    return 0;
}

int uavtrack_step_host(uavtrack_env *env, const int32_t *actions_host, uavtrack_host_step *out, void *stream)
{
    if (!env) return fail("uavtrack_step_host: null handle");
    if (!actions_host || !out) return fail("uavtrack_step_host: actions_host and out must not be null");
    const uavtrack_config &c = env->cfg;
    ON_DEVICE(c.device_id);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const HostLayout L = host_layout(c);
    if (!env->host_blk) {
        void *h = nullptr, *d = nullptr;
        HIP_TRY(hipHostMalloc(&h, L.total * 4, hipHostMallocMapped));
        hipError_t e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) {
            (void)hipHostFree(h);
            return fail("uavtrack_step_host: hipHostGetDevicePointer: %s", hipGetErrorString(e));
        }
        memset(h, 0, L.total * 4);
        env->host_blk = h; env->host_blk_dev = d; env->host_blk_bytes = L.total * 4;
    }
    uint32_t *hb = static_cast<uint32_t *>(env->host_blk), *db = static_cast<uint32_t *>(env->host_blk_dev);
    const size_t BN = (size_t)c.n_envs * c.n_uav;
    memcpy(hb + L.actions, actions_host, BN * 4);          // the kernel reads them through the mapping: no copy call
    // the raw rewards are this call's own extra output; a buffer the caller installed comes back afterwards
    float *const raw_was = env->raw_out;
    const int32_t raw_steps_was = env->raw_steps;
    env->raw_out = reinterpret_cast<float *>(db + L.raw); env->raw_steps = 1;
    env->state_copy_out = reinterpret_cast<float *>(db + L.state);      // the rollout kernel leaves the state there itself
    const int rc = run_steps(env, 1, reinterpret_cast<const int32_t *>(db + L.actions), reinterpret_cast<float *>(db + L.obs),
                             reinterpret_cast<float *>(db + L.reward), reinterpret_cast<float *>(db + L.terms),
                             reinterpret_cast<int32_t *>(db + L.covered), reinterpret_cast<uint8_t *>(db + L.done), nullptr, stream,
                             "uavtrack_step_host");
    env->raw_out = raw_was; env->raw_steps = raw_steps_was;
    env->state_copy_out = nullptr;
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    const float *hf = reinterpret_cast<const float *>(hb);
    const StateBlock sv = state_view(const_cast<float *>(hf + L.state), c.n_envs, c.n_uav, c.m_targets, c.dim == 3);
    out->obs = hf + L.obs; out->reward = hf + L.reward; out->terms = hf + L.terms; out->raw = hf + L.raw;
    out->covered = reinterpret_cast<const int32_t *>(hb + L.covered);
    out->done = reinterpret_cast<const uint8_t *>(hb + L.done);
    out->ux = sv.ux; out->uy = sv.uy; out->uz = sv.uz; out->uh = sv.uh; out->ua = sv.ua;
    out->tx = sv.tx; out->ty = sv.ty; out->tz = sv.tz; out->th = sv.th;
    out->step_count = sv.step_count;
    return 0;
}

int uavtrack_pmi_pairs_scored(uavtrack_env *env, uint64_t *out, void *stream)
{
    if (!env || !out) return fail("uavtrack_pmi_pairs_scored: null argument");
    *out = 0;
    if (!env->pair_total) return 0;
    ON_DEVICE(env->cfg.device_id);
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, env->pair_total, sizeof v, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *out = v;
    return 0;
}

int uavtrack_set_profiling(uavtrack_env *env, int32_t on)
{
    if (!env) return fail("uavtrack_set_profiling: null handle");
    ON_DEVICE(env->cfg.device_id);
    drop_profile(env);
    env->profiling = on != 0;
    return 0;
}

int uavtrack_get_profile(uavtrack_env *env, double *ms, int64_t *launches, void *stream)
{
    if (!env) return fail("uavtrack_get_profile: null handle");
    ON_DEVICE(env->cfg.device_id);
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    for (int k = 0; k < UAVTRACK_PROF_CLASSES; ++k) {
        if (ms) ms[k] = 0.0;
        if (launches) launches[k] = 0;
    }
    for (auto &r : env->prof) {
        float t = 0.0f;
        if (r.a && r.b && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && r.cls >= 0 && r.cls < UAVTRACK_PROF_CLASSES) {
            if (ms) ms[r.cls] += (double)t;
            if (launches) launches[r.cls] += 1;
        }
    }
    drop_profile(env);
    return 0;
}

int uavtrack_launch_info(uavtrack_env *env, int64_t out[4])
{
    if (!env || !out) return fail("uavtrack_launch_info: null argument");
    out[0] = env->last_launch.wgs;
    out[1] = env->last_launch.envs_per_wg;
    out[2] = env->last_launch.groups;
    out[3] = env->last_launch.lone;
    return 0;
}

int uavtrack_kernel_info(uavtrack_env *env, int64_t out[5])
{
    if (!env || !out) return fail("uavtrack_kernel_info: null argument");
    out[0] = env->geo.wgs;
    out[1] = env->geo.envs_per_wg;
    out[2] = env->geo.groups;
    out[3] = (int64_t)env->geo.lds_bytes;
    out[4] = env->geo.specialised;
    return 0;
}

}  // extern "C"
