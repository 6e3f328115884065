// greedy.h -- the per-UAV part of UAV.get_action_by_direction (reference src/agent/uav.py:324-369),
// shared by the stand-alone policy kernel and the fused greedy rollout so that both produce the same
// bits.  `upos(j)` / `tpos(k)` return the (x, y) of UAV j / target k of this UAV's environment,
// `near(k)` the number of UAVs within dc of target k (computed once per target).
#pragma once
#include "internal.h"
#include "philox.h"

namespace uavtrack {

// number of UAVs of the environment strictly within dc of a target (uav.py:352-354)
template <typename UPos>
__device__ __forceinline__ int greedy_near_count(float2 t, int N, float dc2, UPos upos)
{
    int c = 0;
    for (int j = 0; j < N; ++j) {
        const float2 u = upos(j);
        c += (fmaf(u.y - t.y, u.y - t.y, (u.x - t.x) * (u.x - t.x)) < dc2) ? 1 : 0;
    }
    return c;
}

template <typename UPos, typename TPos, typename Near>
__device__ __forceinline__ int greedy_pick(float x, float y, float h, int i, int N, int M, int na, float dc2,
                                           float turn_unit, uint64_t genv, uint32_t step, uint32_t k0, uint32_t k1,
                                           UPos upos, TPos tpos, Near near)
{
    const Philox4 r = philox4x32_10((uint32_t)genv, step, (uint32_t)i, 0x47524459u ^ (uint32_t)(genv >> 32), k0, k1);
    if (u01(r.v[0]) < 0.25f)                                           // uav.py:338-339
        return (int)(((uint64_t)r.v[1] * (uint32_t)na) >> 32);
    int same = 0;                                                      // UAVs at my exact position (me included):
    for (int j = 0; j < N; ++j) {                                      // `(uav_x, uav_y) != (self.x, self.y)`, uav.py:351
        const float2 u = upos(j);
        same += (u.x == x && u.y == y) ? 1 : 0;
    }
    float best = -INFINITY, best_dx = 1.0f, best_dy = 0.0f;
    for (int k = 0; k < M; ++k) {                                      // uav.py:341-362, first best wins
        const float2 t = tpos(k);
        const float dx = t.x - x, dy = t.y - y;
        const float d2 = fmaf(dy, dy, dx * dx);
        const int others = near(k) - (d2 < dc2 ? same : 0);
        const float score = __builtin_amdgcn_rsqf(d2) - 0.8f * (float)others;      // 1 / d (v_rsq_f32: ~1 ulp, one instruction where an IEEE sqrt and divide took ~25)
        if (score > best) { best = score; best_dx = dx; best_dy = dy; }
    }
    float angle = atan2f(best_dy, best_dx) - h;
    if (u01(r.v[2]) < 0.3f) angle = 0.0f;                              // uav.py:365-366
    // The reference calls an undefined find_closest_a_idx (uav.py:368).  Defined here: wrap the angle to
    // [-pi, pi) and take the nearest of the na turn rates (2a + 1 - na) * turn_unit, lowest index on ties.
    angle -= kTwoPi * floorf((angle + kPi) * (1.0f / kTwoPi));
    int besta = 0;
    float bestd = INFINITY;
    for (int a = 0; a < na; ++a) {
        const float dd = fabsf(angle - (float)(2 * a + 1 - na) * turn_unit);
        if (dd < bestd) { bestd = dd; besta = a; }
    }
    return besta;
}

}  // namespace uavtrack
