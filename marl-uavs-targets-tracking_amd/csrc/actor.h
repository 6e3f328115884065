// actor.h -- the shared policy network of the reference's learner, FnnPolicyNet
// (reference src/models/actor_critic.py:85-98: Linear(12, H) - ReLU - Linear(H, na) - softmax), and
// ActorCritic.take_action (actor_critic.py:138-148: Categorical(probs).sample()) for one UAV, shared by
// the stand-alone policy kernel and the fused actor rollout so that both produce the same bits.
//
// One lane = one UAV: its 12-float local state sits in registers; the weights are the same for every
// lane, so they arrive as *scalar* operands (uniform loads from the constant address space -> s_load,
// SGPR pairs feeding v_pk_fma_f32) and cost no vector register, LDS or vector-memory traffic.  Per
// hidden unit: 6 packed FMAs for the 12 inputs, a horizontal add + bias + ReLU, 6 packed FMAs into the
// 12 logits; H = 128 (configs/MAAC.yaml:34) is ~1.8 k VALU per UAV-step.
//
// Device weight blob (uavtrack_set_actor_weights packs it): kActorRow = 26 floats per hidden unit h --
// W1[h][0..11], b1[h], pad, W2[0..11][h] -- followed by one row holding b2[0..11].  Rows are packed
// tightly on purpose: at H = 128 the blob is 13.4 KB and stays resident in the 16 KB scalar cache that
// every wave of the CU streams it through.
// Action slots >= na*nc are zero-padded and masked out of the softmax.
//
// Sampling: torch's Categorical draws from torch's own generator, which has no place inside a kernel;
// here the draw is the inverse CDF of the same probabilities at a Philox uniform keyed by
// (seed, global env, step_count, uav) -- reproducible, shard-independent, restated by the oracle.
#pragma once
#include "internal.h"
#include "philox.h"

namespace uavtrack {

#ifndef UAVTRACK_ACTOR_ROW
#define UAVTRACK_ACTOR_ROW 26
#endif
constexpr int kActorRow = UAVTRACK_ACTOR_ROW;          // floats per hidden unit (even: rows are read as float2)
constexpr int kActorW2 = kActorRow >= 32 ? 16 : 14;    // offset of the W2 column inside a row
constexpr int kActorMaxActions = 12;
constexpr int kActorObs = UAVTRACK_OBS_DIM;   // 12

typedef float actor_v2 __attribute__((ext_vector_type(2)));
typedef const actor_v2 __attribute__((address_space(4))) *actor_cptr;   // constant address space: scalar loads

inline size_t actor_blob_floats(int hidden) { return (size_t)(hidden + 1) * kActorRow; }

// mode: UAVTRACK_ACTOR_SAMPLE (inverse-CDF draw) or UAVTRACK_ACTOR_ARGMAX (lowest index on ties)
template <bool WANT_PROBS>
__device__ __forceinline__ int actor_pick(const float (&o)[kActorObs], const float *weights, int H, int A,
                                          uint64_t genv, uint32_t step, int i, uint32_t k0, uint32_t k1, int mode,
                                          float *probs)
{
    const actor_cptr w = (actor_cptr)(uintptr_t)weights;
    actor_v2 x[6], l[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        x[k] = (actor_v2){o[2 * k], o[2 * k + 1]};
        l[k] = (actor_v2){0.0f, 0.0f};
    }
#pragma unroll 2
    for (int h = 0; h < H; ++h) {
        const actor_cptr row = w + h * (kActorRow / 2);
        actor_v2 acc = row[0] * x[0];
#pragma unroll
        for (int k = 1; k < 6; ++k) acc = __builtin_elementwise_fma(row[k], x[k], acc);
        const float a = fmaxf(acc.x + acc.y + row[6].x, 0.0f);
        const actor_v2 a2 = {a, a};
#pragma unroll
        for (int j = 0; j < 6; ++j) l[j] = __builtin_elementwise_fma(row[kActorW2 / 2 + j], a2, l[j]);
    }
    const actor_cptr b2 = w + H * (kActorRow / 2);
    float lg[kActorMaxActions];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const actor_v2 v = l[j] + b2[j];
        lg[2 * j] = (2 * j < A) ? v.x : -INFINITY;
        lg[2 * j + 1] = (2 * j + 1 < A) ? v.y : -INFINITY;
    }
    float m = lg[0];
    int am = 0;
#pragma unroll
    for (int j = 1; j < kActorMaxActions; ++j)
        if (lg[j] > m) { m = lg[j]; am = j; }
    float ex[kActorMaxActions], S = 0.0f;
#pragma unroll
    for (int j = 0; j < kActorMaxActions; ++j) {
        ex[j] = __builtin_amdgcn_exp2f((lg[j] - m) * 1.44269504088896340736f);   // masked slots: exp2(-inf) = 0
        S += ex[j];
    }
    if (WANT_PROBS && probs) {
        const float inv = 1.0f / S;
        for (int j = 0; j < A; ++j) probs[j] = ex[j] * inv;
    }
    if (mode == UAVTRACK_ACTOR_ARGMAX) return am;
    const Philox4 r = philox4x32_10((uint32_t)genv, step, (uint32_t)i, 0x4143544Fu ^ (uint32_t)(genv >> 32), k0, k1);
    const float target = u01(r.v[0]) * S;
    float c = 0.0f;
    int pick = A - 1;                       // rounding can leave the last cumulative sum <= target
    bool found = false;
#pragma unroll
    for (int j = 0; j < kActorMaxActions; ++j) {
        c += ex[j];
        if (!found && c > target && j < A) { pick = j; found = true; }
    }
    return pick;
}

}  // namespace uavtrack
