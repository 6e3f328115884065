/* Plain-C client of the C ABI in include/uavtrack.h: no Python, no torch, no C++.
 * Builds with `gcc` against libamdhip64 (device memory + stream only) and libuavtrack.so.
 * Creates B environments, resets them, takes T single steps with a fixed action pattern and prints
 * FNV-1a checksums of the last observation / reward buffers plus the covered-target total, which the
 * GPU test compares with the Python host layer driving the same library (tests/test_hip_parity.py); then one step through
 * uavtrack_step_host (host pointers in and out), whose checksums go on a second line.
 * usage: abi_roundtrip [B N M T seed]   (without a GPU it exits 3 after printing uavtrack_last_error) */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uavtrack.h"

static uint64_t fnv1a(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define UAV_OK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, uavtrack_last_error()); return 3; } } while (0)

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 20, M = argc > 3 ? atoi(argv[3]) : 10;
    const int T = argc > 4 ? atoi(argv[4]) : 5;
    const uint64_t seed = argc > 5 ? strtoull(argv[5], NULL, 10) : 42;
    if (uavtrack_version() != UAVTRACK_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }

    uavtrack_config c;
    memset(&c, 0, sizeof c);
    c.struct_size = sizeof c;
    c.n_envs = B; c.n_uav = N; c.m_targets = M; c.dim = 2; c.na = 12; c.nc = 1;
    c.norm_n_uav = N; c.norm_m_targets = M;
    c.reward_mode = UAVTRACK_REWARD_RAW; c.horizon = 200; c.device_id = 0; c.env_offset = 0;
    c.x_max = 2000.0; c.y_max = 2000.0; c.z_max = 0.0; c.dt = 1.0;
    c.u_v_max = 20.0; c.u_h_max = 3.14159265358979323846 / 6.0; c.u_g_max = 0.0;
    c.dc = 500.0; c.dp = 200.0; c.t_v_max = 5.0;
    c.alpha = 0.6; c.beta = 0.2; c.gamma = 0.2; c.cooperative = 0.0;

    uavtrack_env *env = NULL;
    UAV_OK(uavtrack_create(&c, &env));

    const size_t BN = (size_t)B * N;
    int32_t *d_act, *d_cov; float *d_obs, *d_rew, *d_terms; uint8_t *d_done;
    HIP_OK(hipMalloc((void **)&d_act, BN * 4));
    HIP_OK(hipMalloc((void **)&d_obs, BN * UAVTRACK_OBS_DIM * 4));
    HIP_OK(hipMalloc((void **)&d_rew, BN * 4));
    HIP_OK(hipMalloc((void **)&d_terms, 3 * BN * 4));
    HIP_OK(hipMalloc((void **)&d_cov, (size_t)B * 4));
    HIP_OK(hipMalloc((void **)&d_done, (size_t)B));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    int32_t *h_act = (int32_t *)malloc(BN * 4);
    float *h_obs = (float *)malloc(BN * UAVTRACK_OBS_DIM * 4), *h_rew = (float *)malloc(BN * 4);
    int32_t *h_cov = (int32_t *)malloc((size_t)B * 4);
    long covered_total = 0;

    UAV_OK(uavtrack_reset(env, seed, 0, d_obs, st));
    for (int t = 0; t < T; ++t) {
        for (size_t g = 0; g < BN; ++g) h_act[g] = (int32_t)((g * 7 + (size_t)t * 3) % 12);
        HIP_OK(hipMemcpyAsync(d_act, h_act, BN * 4, hipMemcpyHostToDevice, st));
        UAV_OK(uavtrack_step(env, d_act, d_obs, d_rew, d_terms, d_cov, d_done, st));
        HIP_OK(hipMemcpyAsync(h_cov, d_cov, (size_t)B * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        for (int b = 0; b < B; ++b) covered_total += h_cov[b];
    }
    HIP_OK(hipMemcpy(h_obs, d_obs, BN * UAVTRACK_OBS_DIM * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_rew, d_rew, BN * 4, hipMemcpyDeviceToHost));
    printf("obs %016llx reward %016llx covered %ld\n", (unsigned long long)fnv1a(h_obs, BN * UAVTRACK_OBS_DIM * 4),
           (unsigned long long)fnv1a(h_rew, BN * 4), covered_total);
    /* one more step the way the reference's own loop calls Environment.step: HOST actions in, HOST results out
     * (uavtrack_step_host: no device buffer, no copy call on the caller's side) -- second line of output */
    {
        uavtrack_host_step hs;
        long cov = 0;
        for (size_t g = 0; g < BN; ++g) h_act[g] = (int32_t)((g * 5 + 1) % 12);
        UAV_OK(uavtrack_step_host(env, h_act, &hs, st));
        for (int b = 0; b < B; ++b) cov += hs.covered[b];
        printf("host obs %016llx reward %016llx raw %016llx ux %016llx covered %ld ua0 %d\n",
               (unsigned long long)fnv1a(hs.obs, BN * UAVTRACK_OBS_DIM * 4), (unsigned long long)fnv1a(hs.reward, BN * 4),
               (unsigned long long)fnv1a(hs.raw, BN * 4), (unsigned long long)fnv1a(hs.ux, BN * 4), cov, (int)hs.ua[0]);
    }
    UAV_OK(uavtrack_destroy(env));
    return 0;
}
