// actor.h -- the shared policy network of the reference's learner, FnnPolicyNet
// (reference src/models/actor_critic.py:85-98: Linear(12, H) - ReLU - Linear(H, na) - softmax), and
// ActorCritic.take_action (actor_critic.py:138-148: Categorical(probs).sample()) for the 64 UAVs of one
// wavefront, shared by the stand-alone policy kernel and the fused actor rollout so that both produce
// the same bits.
//
// One lane = one UAV, but the two Linear layers are GEMM-shaped (64 samples x 12 x H and 64 x H x 16
// per wavefront-step) and every lane needs every weight, so they run on the matrix cores, transposed:
//     Ht [H x 64]  = relu(W1 [H x 12] . Xt [12 x 64] + b1)        3 k-steps of v_mfma_f32_16x16x4_f32
//     Lt [16 x 64] = W2 [16 x H] . Ht [H x 64] + b2               H/4 k-steps
// In the transposed form the weights are the A operand (streamed from L1/L2 as ready-made fragments,
// one coalesced 256-B load per fragment), the samples are the B/N side, and -- the point of the layout --
// the accumulator fragment of the first GEMM *is* the B fragment of the second: D[i = 4*(l/16) + v][j =
// l%16] in register v of lane l is B[k = l/16][j = l%16] for the k-step made of hidden units
// {v, 4 + v, 8 + v, 12 + v} of the 16-unit block, so the hidden layer never leaves its registers (the
// k-order of a sum is free; W2's fragments are packed in that order).  Only the 12 inputs and the 16
// logits of a sample cross lanes, through 5 KB of LDS private to the wavefront.
// fp32 MFMA throughout (fp32 parity with the torch actor, 1e-5 on the probabilities); measured against
// a first version that fed v_pk_fma_f32 from scalar loads (weights as SGPR operands), which was bound
// by scalar-cache bandwidth: 15.6 -> see DESIGN.md for the closed-loop numbers.
//
// Device weight blob (uavtrack_set_actor_weights packs it), in units of one fragment = 64 floats, lane
// l at offset l; MT action tiles of 16 rows (1 in 2-D, 3 for the 3-D action space); HB = ceil(H/16) blocks
// of 7 + 4*MT fragments, then 4*MT fragments of b2:
//   block a:  W1 s=0..2        : W1[16a + l%16][4s + l/16]
//             b1 v=0..3        : b1[16a + 4*(l/16) + v]
//             W2 tile t, v=0..3: W2[16t + l%16][16a + 4*(l/16) + v]   (rows >= na*nc and units >= H are zero)
//   tail:     b2 tile t, v=0..3: b2[16t + 4*(l/16) + v]
//
// Sampling: torch's Categorical draws from torch's own generator, which has no place inside a kernel;
// here the draw is the inverse CDF of the same probabilities at a Philox uniform keyed by
// (seed, global env, step_count, uav) -- reproducible, shard-independent, restated by the oracle.
#pragma once
#include "internal.h"
#include "philox.h"

namespace uavtrack {

constexpr int kActorObs = UAVTRACK_OBS_DIM;            // 12
constexpr int kActorLdsFloats = 64 * 20;               // per wavefront: logits at a 20-float stride (conflict-free b128)
// Action tiles of 16 rows in the second GEMM: MT = 1 serves the reference's action space (na = 12), MT = 3 the
// 3-D action space of our own spec (na * nc = 36, up to 48).  Per-lane softmax slots: 12 / 48.
constexpr int actor_tiles(bool z3) { return z3 ? 3 : 1; }
constexpr int actor_slots(int mt) { return mt == 1 ? 12 : 16 * mt; }
constexpr int actor_frags_per_block(int mt) { return 7 + 4 * mt; }   // W1 x3, b1 x4, W2 x4 per tile

typedef float actor_v4 __attribute__((ext_vector_type(4)));

inline int actor_blocks(int hidden) { return (hidden + 15) / 16; }
inline size_t actor_blob_floats(int hidden, int mt)
{
    return ((size_t)actor_blocks(hidden) * actor_frags_per_block(mt) + 4 * mt) * 64;
}

// Host side: torch layouts (w1 [H][12], b1 [H], w2 [A][H], b2 [A]) -> fragment order above (tile t of W2 / b2
// holds actions 16t .. 16t+15; its fragments follow tile t-1's).
inline void pack_actor_blob(const float *w1, const float *b1, const float *w2, const float *b2, int H, int A, int mt, float *blob)
{
    const int HB = actor_blocks(H), FB = actor_frags_per_block(mt);
    for (int a = 0; a < HB; ++a) {
        float *blk = blob + (size_t)a * FB * 64;
        for (int l = 0; l < 64; ++l) {
            const int j = l & 15, g = l >> 4;
            for (int s = 0; s < 3; ++s) {
                const int u = 16 * a + j;
                blk[s * 64 + l] = u < H ? w1[(size_t)u * kActorObs + 4 * s + g] : 0.0f;
            }
            for (int v = 0; v < 4; ++v) {
                const int u = 16 * a + 4 * g + v;
                blk[(3 + v) * 64 + l] = u < H ? b1[u] : 0.0f;
                for (int t = 0; t < mt; ++t) {
                    const int act = 16 * t + j;
                    blk[(7 + 4 * t + v) * 64 + l] = (u < H && act < A) ? w2[(size_t)act * H + u] : 0.0f;
                }
            }
        }
    }
    float *tail = blob + (size_t)HB * FB * 64;
    for (int l = 0; l < 64; ++l)
        for (int t = 0; t < mt; ++t)
            for (int v = 0; v < 4; ++v) {
                const int act = 16 * t + 4 * (l >> 4) + v;
                tail[(4 * t + v) * 64 + l] = act < A ? b2[act] : 0.0f;
            }
}

// EVERY lane of the wavefront must reach this call together (MFMA ignores EXEC); lanes without a UAV pass
// zeros and ignore the result.  lds: kActorLdsFloats floats private to this wavefront.
// mode: UAVTRACK_ACTOR_SAMPLE (inverse-CDF draw) or UAVTRACK_ACTOR_ARGMAX (lowest index on ties).
struct ActorRng {          // Philox block cache of one UAV (see the draw below)
    Philox4 r;
    uint32_t block;
    bool valid;
};

template <bool WANT_PROBS, int MT>
__device__ __forceinline__ int actor_pick(const float (&o)[kActorObs], float *lds, const float *__restrict__ weights,
                                          int HB, int A, uint64_t genv, uint32_t step, int i, uint32_t k0, uint32_t k1,
                                          int mode, float *probs, ActorRng &rng)
{
    constexpr int FB = actor_frags_per_block(MT), SLOTS = actor_slots(MT);
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    // ---- the 12 inputs of the 64 samples, regrouped into B fragments: lane l <- obs[16n + l%16][4s + l/16]
    {
        float4 *xs = reinterpret_cast<float4 *>(lds + lane * kActorObs);
        xs[0] = make_float4(o[0], o[1], o[2], o[3]);
        xs[1] = make_float4(o[4], o[5], o[6], o[7]);
        xs[2] = make_float4(o[8], o[9], o[10], o[11]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float xf[4][3];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int s = 0; s < 3; ++s) xf[n][s] = lds[(16 * n + j) * kActorObs + 4 * s + g];

    const float *wl = weights + lane;
    actor_v4 d2[MT][4];
    {
        const float *t = wl + (size_t)HB * FB * 64;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const actor_v4 b2f = {t[(4 * mt) * 64], t[(4 * mt + 1) * 64], t[(4 * mt + 2) * 64], t[(4 * mt + 3) * 64]};
#pragma unroll
            for (int n = 0; n < 4; ++n) d2[mt][n] = b2f;
        }
    }
#pragma unroll 2
    for (int a = 0; a < HB; ++a) {
        const float *wa = wl + (size_t)a * FB * 64;
        float wf[FB];
#pragma unroll
        for (int f = 0; f < FB; ++f) wf[f] = wa[f * 64];
        const actor_v4 b1f = {wf[3], wf[4], wf[5], wf[6]};
        actor_v4 d1[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) d1[n] = b1f;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int n = 0; n < 4; ++n) d1[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xf[n][s], d1[n], 0, 0, 0);
        // ReLU as one v_med3_f32 per element (fmaxf on an MFMA result costs a canonicalising v_max first)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v) d1[n][v] = __builtin_amdgcn_fmed3f(d1[n][v], 0.0f, 3.0e38f);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    d2[mt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[7 + 4 * mt + v], d1[n][v], d2[mt][n], 0, 0, 0);
    }
    // ---- logits back to their own lane, one action tile at a time through the same 5 KB: lane l holds actions
    //      16t + 4*(l/16) .. +3 of samples 16n + l%16
    float lg[SLOTS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n = 0; n < 4; ++n)
            *reinterpret_cast<float4 *>(lds + (16 * n + j) * 20 + 4 * g) =
                make_float4(d2[mt][n][0], d2[mt][n][1], d2[mt][n][2], d2[mt][n][3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr int Q = (MT == 1) ? 3 : 4;               // float4 per tile this lane needs (12 of 16 at MT = 1)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(lds + lane * 20 + 4 * q);
            lg[16 * mt * (MT > 1) + 4 * q] = v.x; lg[16 * mt * (MT > 1) + 4 * q + 1] = v.y;
            lg[16 * mt * (MT > 1) + 4 * q + 2] = v.z; lg[16 * mt * (MT > 1) + 4 * q + 3] = v.w;
        }
    }
    __builtin_amdgcn_wave_barrier();                 // the next call's input staging must not overtake these reads
    if (A < SLOTS) {                                 // uniform; the reference's na = 12 fills every slot of MT = 1
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) lg[q] = (q < A) ? lg[q] : -INFINITY;
    }
    float m = lg[0];
#pragma unroll
    for (int q = 1; q < SLOTS; ++q) m = fmaxf(m, lg[q]);
    float ex[SLOTS], S = 0.0f;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        ex[q] = __builtin_amdgcn_exp2f((lg[q] - m) * 1.44269504088896340736f);   // masked slots: exp2(-inf) = 0
        S += ex[q];
    }
    if (WANT_PROBS && probs) {
        const float inv = 1.0f / S;
#pragma unroll
        for (int q = 0; q < SLOTS; ++q)
            if (q < A) probs[q] = ex[q] * inv;
    }
    if (mode == UAVTRACK_ACTOR_ARGMAX) {
        int am = 0;
#pragma unroll
        for (int q = SLOTS - 1; q >= 0; --q) am = (lg[q] == m) ? q : am;   // lowest index on ties
        return am;
    }
    // One Philox block serves four consecutive steps (its four words): the generator is the costly part of
    // the draw (40 quarter-rate integer multiplies), and a rollout caches the block across steps.
    const uint32_t blk = step >> 2;
    if (rng.block != blk || !rng.valid) {
        rng.r = philox4x32_10((uint32_t)genv, blk, (uint32_t)i, 0x4143544Fu ^ (uint32_t)(genv >> 32), k0, k1);
        rng.block = blk;
        rng.valid = true;
    }
    const uint32_t word = (step & 2) ? ((step & 1) ? rng.r.v[3] : rng.r.v[2]) : ((step & 1) ? rng.r.v[1] : rng.r.v[0]);
    const float target = u01(word) * S;
    // first q with cumsum_q > target == number of q with cumsum_q <= target (the sums never decrease);
    // rounding can leave even the last one <= target, hence the clamp
    float c = 0.0f;
    int pick = 0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        c += ex[q];
        pick += (c <= target) ? 1 : 0;
    }
    return min(pick, A - 1);
}

}  // namespace uavtrack
