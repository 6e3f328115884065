// actor.h -- the shared policy network of the reference's learner, FnnPolicyNet
// (reference src/models/actor_critic.py:85-98: Linear(12, H) - ReLU - Linear(H, na) - softmax), and
// ActorCritic.take_action (actor_critic.py:138-148: Categorical(probs).sample()) for the 64 UAVs of one
// wavefront, shared by the stand-alone policy kernel and the fused actor rollout so that both produce
// the same bits.
//
// One lane = one UAV, but the two Linear layers are GEMM-shaped (64 samples x 12 x H and 64 x H x na per
// wavefront-step) and every lane needs every weight, so they run on the matrix cores, transposed (weights are the
// A operand, the 64 samples the columns):
//     Ht [H x 64]  = relu(W1 [H x 16] . Xt [16 x 64])       inputs 0..11, input 12 = 1 carries b1, 13..15 = 0
//     Lt [32 x 64] = W2 [32 x H] . Ht [H x 64]              rows >= na are zero
// on the 16-BIT matrix cores at fp32 accuracy (round 4; rounds 1-3 ran 224 fp32-input MFMAs per wavefront-step, 1/16 of the
// 16-bit rate): an fp32 value is the two-term f16 sum v = hi + lo, hi = f16(v) toward zero, lo = f16(v - hi) -- 22
// significant bits, the remainder representable down to 2^-24 of a block scale the host folds into each layer (powers of
// two: exact) -- and a product is THREE v_mfma_f32_32x32x16_f16 with exact products and fp32 accumulation,
//     w v = wl vh + wh vl + wh vh      (small terms first; wl vl < 2^-22 |w v| is dropped),
// 72 of them per wavefront-step at H = 128 (24 for layer 1, 48 for layer 2) against 224 of twice the cycles before.
// The scorer of the MAAC-R reward (pmi_kernel.hip) carries the error analysis and the numpy emulation of this split.
//
// No value crosses lanes through memory:
//   * a 32x32x16 B operand holds, in lane l, eight consecutive k of column l % 32 (k = 0..7 in lanes 0..31, 8..15 in
//     lanes 32..63), and the two column tiles of a wavefront are samples 0..31 and 32..63: lane l owns sample l, so tile 0
//     needs inputs 8..15 of sample l in lane l + 32 and tile 1 inputs 0..7 of sample l + 32 in lane l -- ONE
//     v_permlane32_swap_b32 per register pair builds both tiles' operand registers;
//   * the accumulator of layer 1 (lane l: column l % 32, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)) IS layer 2's B operand
//     once converted: registers r = 8 h .. 8 h + 7 of the two half-wavefronts are the 16 k of k-step (tile, h) -- the
//     k-order of a sum is free, W2's fragments are packed in that order -- so the hidden layer never leaves its lane;
//   * layer 2's accumulator spreads a sample's logits over lanes l and l + 32 of either column tile; the same swap
//     instruction (tile 0's register r against tile 1's) hands every lane the logits of its own sample.
// ReLU is the v_med3_f32 that also caps a value at the f16 range (an observation next to the origin can be arbitrarily
// large, uav.py:165; saturation instead of inf/NaN), so remainders are non-negative and need no clamp of their own.
//
// Device weight blob (uavtrack_set_actor_weights packs it): 128 header floats { [0] 1 / (T1 T2), [16 .. 16 + 32 AT) b2 },
// then per tile a of 32 hidden units 2 + 4 AT fragments of 64 lanes x 8 f16 (16 B per lane, lane-major):
//   W1 hi, W1 lo        lane l, element j: T1 * W1[32 a + l % 32][k = 8 (l >> 5) + j]   (k = 12: b1, k > 12: 0)
//   per action tile t (AT = 1: the reference's 12 actions; AT = 2: the 3-D action space, up to 48) and half h = 0, 1:
//   W2 hi, W2 lo        lane l, element j: T2 * W2[32 t + l % 32][unit 32 a + (r & 3) + 8 (r >> 2) + 4 (j >> 3 ... see pack)]
// T1, T2 powers of two: T1 x (bound of |pre-activation| over nominal observation ranges) <= 512 and T x max |w| <= 16384.
//
// Sampling: torch's Categorical draws from torch's own generator, which has no place inside a kernel;
// here the draw is the inverse CDF of the same probabilities at a Philox uniform keyed by
// (seed, global env, step_count, uav) -- reproducible, shard-independent, restated by the oracle.
#pragma once
#include "internal.h"
#include "philox.h"

#include <cmath>
#include <cstring>
#include <utility>

namespace uavtrack {

constexpr int kActorObs = UAVTRACK_OBS_DIM;            // 12
constexpr int kActorHeaderFloats = 128;
constexpr int kActorB2Offset = 16;
// 32-row action tiles of the second GEMM: 1 serves the reference's action space (na = 12), 2 the 3-D action space of our
// own spec (na * nc = 36, up to 48).  Per-lane softmax slots: 12 / 48.
constexpr int actor_tiles(bool z3) { return z3 ? 2 : 1; }
constexpr int actor_slots(int at) { return at == 1 ? 12 : 48; }
constexpr int actor_frags_per_tile(int at) { return 2 + 4 * at; }      // W1 hi, lo; per action tile W2 (half 0, 1) x (hi, lo)

inline int actor_blocks(int hidden) { return (hidden + 31) / 32; }     // 32-unit tiles of the hidden layer
inline size_t actor_blob_floats(int hidden, int at)
{
    return kActorHeaderFloats + (size_t)actor_blocks(hidden) * actor_frags_per_tile(at) * 64 * 4;
}

// hidden unit held by accumulator register r of a lane in half-wavefront kh (32x32 MFMA C/D layout)
inline int actor_unit_of(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// Host side: torch layouts (w1 [H][12], b1 [H], w2 [A][H], b2 [A]) -> the blob above.  xb[12]: nominal bounds of |obs_k|
// (they size T1; a value beyond them saturates at 60000 / T1 inside the kernel instead of overflowing).
inline void pack_actor_blob(const float *w1, const float *b1, const float *w2, const float *b2, int H, int A, int at,
                            const double *xb, float *blob)
{
    const int HT = actor_blocks(H), FT = actor_frags_per_tile(at);
    double act = 0.0, w1max = 0.0, w2max = 0.0;
    for (int u = 0; u < H; ++u) {
        double a = std::fabs((double)b1[u]);
        w1max = std::fmax(w1max, a);
        for (int k = 0; k < kActorObs; ++k) {
            a += std::fabs((double)w1[(size_t)u * kActorObs + k]) * xb[k];
            w1max = std::fmax(w1max, std::fabs((double)w1[(size_t)u * kActorObs + k]));
        }
        act = std::fmax(act, a);
    }
    for (size_t k = 0; k < (size_t)A * H; ++k) w2max = std::fmax(w2max, std::fabs((double)w2[k]));
    auto pow2_below = [](double bound, double target) {         // largest 2^e with 2^e * bound <= target, e in [-24, 24]
        int e = 24;
        if (bound > 0.0 && std::isfinite(bound)) e = (int)std::floor(std::log2(target / bound));
        return std::ldexp(1.0, e < -24 ? -24 : (e > 24 ? 24 : e));
    };
    const double T1 = std::fmin(pow2_below(act, 512.0), pow2_below(w1max, 16384.0));
    const double T2 = pow2_below(w2max, 16384.0);
    memset(blob, 0, actor_blob_floats(H, at) * sizeof(float));
    blob[0] = (float)(1.0 / (T1 * T2));
    for (int q = 0; q < A; ++q) blob[kActorB2Offset + q] = b2[q];
    uint16_t *frag = reinterpret_cast<uint16_t *>(blob + kActorHeaderFloats);
    auto put = [&](int a, int f, int lane, int j, double v) {                  // hi into fragment f, lo into f + 1
        const float s = (float)v;
        const _Float16 hi = (_Float16)s;
        const _Float16 lo = (_Float16)(s - (float)hi);
        uint16_t bh, bl;
        memcpy(&bh, &hi, 2);
        memcpy(&bl, &lo, 2);
        frag[(((size_t)a * FT + f) * 64 + lane) * 8 + j] = bh;
        frag[(((size_t)a * FT + f + 1) * 64 + lane) * 8 + j] = bl;
    };
    for (int a = 0; a < HT; ++a)
        for (int lane = 0; lane < 64; ++lane) {
            const int row = lane & 31, kh = lane >> 5;
            for (int j = 0; j < 8; ++j) {
                // layer 1: row = hidden unit 32 a + row, k = 8 kh + j: inputs 0..11, then the bias on the constant input
                const int u = 32 * a + row, k = 8 * kh + j;
                double v = 0.0;
                if (u < H) v = k < kActorObs ? (double)w1[(size_t)u * kActorObs + k] : (k == kActorObs ? (double)b1[u] : 0.0);
                put(a, 0, lane, j, T1 * v);
                // layer 2, k-step (a, half): row = action 32 t + row, k = 8 kh + j <-> the unit the B operand's lane holds there
                for (int t = 0; t < at; ++t)
                    for (int half = 0; half < 2; ++half) {
                        const int unit = 32 * a + actor_unit_of(8 * half + j, kh), q = 32 * t + row;
                        const double w = (unit < H && q < A) ? (double)w2[(size_t)q * H + unit] : 0.0;
                        put(a, 2 + 4 * t + 2 * half, lane, j, T2 * w);
                    }
            }
        }
}

struct ActorRng {          // Philox block cache of one UAV (see the draw below)
    Philox4 r;
    uint32_t block;
    bool valid;
};

typedef _Float16 actor_h8 __attribute__((ext_vector_type(8)));
typedef float actor_f16v __attribute__((ext_vector_type(16)));
typedef unsigned actor_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ actor_h8 actor_as_h8(actor_u4 v) { return __builtin_bit_cast(actor_h8, v); }
// two floats -> one word of two f16 (first in the low half), toward zero: saturates at 65504, never inf
__device__ __forceinline__ unsigned actor_pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
// f16 pair of the remainders (a - hi.lo, b - hi.hi): one mixed-precision FMA per value (f16 operand * -1 + fp32 operand,
// rounded once to f16); the remainder of a toward-zero conversion is below one f16 ulp of the value: no overflow
__device__ __forceinline__ unsigned actor_rem(unsigned hi, float a, float b)
{
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(hi), "v"(b));
    return r;
}
// lanes 32..63 of `lo_keeps` trade places with lanes 0..31 of `hi_keeps` (v_permlane32_swap_b32): afterwards
//   lo_keeps: lanes 0..31 unchanged, lane l >= 32 holds what hi_keeps held in lane l - 32
//   hi_keeps: lanes 32..63 unchanged, lane l < 32 holds what lo_keeps held in lane l + 32
__device__ __forceinline__ void actor_swap32(unsigned &lo_keeps, unsigned &hi_keeps)
{
    const auto r = __builtin_amdgcn_permlane32_swap(lo_keeps, hi_keeps, false, false);
    lo_keeps = r[0];
    hi_keeps = r[1];
}
__device__ __forceinline__ void actor_swap32(float &lo_keeps, float &hi_keeps)
{
    unsigned a = __float_as_uint(lo_keeps), b = __float_as_uint(hi_keeps);
    actor_swap32(a, b);
    lo_keeps = __uint_as_float(a);
    hi_keeps = __uint_as_float(b);
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N - 1>)
template <int... I, class F>
__device__ __forceinline__ void actor_static_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void actor_static_for(F &&f) { actor_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

constexpr float kActorCap = 60000.0f;       // inside f16's 65504 with room for the toward-zero conversion

// EVERY lane of the wavefront must reach this call together (MFMA and the lane swaps ignore EXEC); lanes without a UAV
// pass zeros and ignore the result.  weights: the blob; HT = actor_blocks(hidden).
// mode: UAVTRACK_ACTOR_SAMPLE (inverse-CDF draw) or UAVTRACK_ACTOR_ARGMAX (lowest index on ties).
// HT_ > 0: the number of hidden tiles is known at compile time, and even (the caller checks HT == HT_): the tile loop takes two
// tiles per trip with the fragment buffers taking turns (no copy of 24 registers per tile).  (A fully straight layout --
// accumulators started from the MFMA's inline zero, immediate fragment addresses -- spilled: 208 bytes of scratch.)
template <bool WANT_PROBS, int AT, int HT_ = 0>
__device__ __forceinline__ int actor_pick(const float (&o)[kActorObs], const float *__restrict__ weights,
                                          int HT, int A, uint64_t genv, uint32_t step, int i, uint32_t k0, uint32_t k1,
                                          int mode, float *probs, ActorRng &rng)
{
    constexpr int FT = actor_frags_per_tile(AT), SLOTS = actor_slots(AT);
    const int lane = threadIdx.x & 63;
    // a tile's fragments: one coalesced 1-KiB load each, L2-resident; the first tile's are requested here, ahead of the
    // input conversion, every later tile's while its predecessor is being multiplied
    const actor_u4 *wl = reinterpret_cast<const actor_u4 *>(weights + kActorHeaderFloats) + lane;
    actor_u4 wf[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) wf[f] = wl[(size_t)f * 64];
    // ---- the 12 inputs as f16 planes, then as the B operands of the two column tiles (samples 0..31 / 32..63)
    unsigned xh[2][4], xl[2][4];
    {
        unsigned ph[6], pl[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float a = __builtin_amdgcn_fmed3f(o[2 * j], -kActorCap, kActorCap), b = __builtin_amdgcn_fmed3f(o[2 * j + 1], -kActorCap, kActorCap);
            ph[j] = actor_pk(a, b);
            pl[j] = actor_rem(ph[j], a, b);
        }
        // registers 0..3 of a tile's operand: k = 0..7 (inputs 0..7) in lanes 0..31, k = 8..15 (inputs 8..11, the constant 1
        // that carries b1, zeros) in lanes 32..63
        unsigned one = 0x00003C00u, zero = 0u, zl0 = 0u, zl1 = 0u;       // (f16 1.0 in the low half; the lo plane's constant is 0)
        xh[0][0] = ph[0]; xh[1][0] = ph[4]; actor_swap32(xh[0][0], xh[1][0]);
        xh[0][1] = ph[1]; xh[1][1] = ph[5]; actor_swap32(xh[0][1], xh[1][1]);
        xh[0][2] = ph[2]; xh[1][2] = one;   actor_swap32(xh[0][2], xh[1][2]);
        xh[0][3] = ph[3]; xh[1][3] = zero;  actor_swap32(xh[0][3], xh[1][3]);
        xl[0][0] = pl[0]; xl[1][0] = pl[4]; actor_swap32(xl[0][0], xl[1][0]);
        xl[0][1] = pl[1]; xl[1][1] = pl[5]; actor_swap32(xl[0][1], xl[1][1]);
        xl[0][2] = pl[2]; xl[1][2] = zl0;   actor_swap32(xl[0][2], xl[1][2]);
        xl[0][3] = pl[3]; xl[1][3] = zl1;   actor_swap32(xl[0][3], xl[1][3]);
    }
    actor_h8 bxh[2], bxl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bxh[t] = actor_as_h8((actor_u4){xh[t][0], xh[t][1], xh[t][2], xh[t][3]});
        bxl[t] = actor_as_h8((actor_u4){xl[t][0], xl[t][1], xl[t][2], xl[t][3]});
    }

    actor_f16v zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
    actor_f16v d2[AT][2];
    // The two layers as a software pipeline over the units (hidden tile a, column tile c): a wavefront issues in order, an
    // MFMA occupies the matrix pipe for 32 cycles but the issue port for 8, and the conversion of a finished layer-1 tile
    // into layer-2 operands is ~20 VALU instructions per half -- left to the compiler the MFMAs came in dependent groups of
    // three with the conversions between them, matrix pipe and VALU taking turns.  Here every MFMA is followed by its share
    // of VALU work that does NOT depend on it, pinned with sched_barrier:
    //   phase A   layer 1 of the NEXT unit        |  conversion of this unit's registers 0..7   (half 0)
    //   phase B   layer 2, k-step (a, 0)          |  conversion of registers 8..15              (half 1)
    //   phase C   layer 2, k-step (a, 1)          |  --
    // (layer 1: T1 (W1 x + b1), three products in one accumulator, small terms first; conversion: ReLU and the cap as one
    //  v_med3_f32 per value, hi = f16 toward zero, lo = f16(v - hi) >= 0.)
    auto layer1 = [&](actor_f16v &d, const actor_u4 (&w)[FT], int c, auto kc) {
        constexpr int k = decltype(kc)::value;
        const actor_h8 w1h = actor_as_h8(w[0]), w1l = actor_as_h8(w[1]);
        if constexpr (k == 0) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l, bxh[c], zero16, 0, 0, 0);
        else if constexpr (k == 1) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, bxl[c], d, 0, 0, 0);
        else d = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, bxh[c], d, 0, 0, 0);
        asm volatile("" : "+v"(d));           // (pins the MFMA in program order: see layer2)
    };
    struct Conv { float v[8]; unsigned hh[4], hl[4]; };
    // 16 atoms: 0..7 clamp value j, 8..11 pack pair j, 12..15 remainders of pair j
    auto conv_atom = [&](Conv &cv, const actor_f16v &d, int half, auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 8) cv.v[k] = __builtin_amdgcn_fmed3f(d[8 * half + k], 0.0f, kActorCap);
        else if constexpr (k < 12) cv.hh[k - 8] = actor_pk(cv.v[2 * (k - 8)], cv.v[2 * (k - 8) + 1]);
        else cv.hl[k - 12] = actor_rem(cv.hh[k - 12], cv.v[2 * (k - 12)], cv.v[2 * (k - 12) + 1]);
    };
    auto layer2 = [&](const actor_u4 (&w)[FT], const Conv &cv, int c, int half, auto tc, auto kc, auto firstc) {
        constexpr int t = decltype(tc)::value, k = decltype(kc)::value;
        constexpr bool first = decltype(firstc)::value;      // the very first product of this accumulator: C = 0, nothing to clear
        const actor_h8 bh = actor_as_h8((actor_u4){cv.hh[0], cv.hh[1], cv.hh[2], cv.hh[3]}), bl = actor_as_h8((actor_u4){cv.hl[0], cv.hl[1], cv.hl[2], cv.hl[3]});
        const actor_h8 w2h = actor_as_h8(w[2 + 4 * t + 2 * half]), w2l = actor_as_h8(w[3 + 4 * t + 2 * half]);
        if constexpr (k == 0 && first) d2[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l, bh, zero16, 0, 0, 0);
        else if constexpr (k == 0) d2[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l, bh, d2[t][c], 0, 0, 0);
        else if constexpr (k == 1) d2[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h, bl, d2[t][c], 0, 0, 0);
        else d2[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h, bh, d2[t][c], 0, 0, 0);
        // (an MFMA is a pure value to the compiler and would sink to its only reader -- the end of the loop -- leaving the
        //  conversions alone between the barriers; the empty statement pins it here)
        asm volatile("" : "+v"(d2[t][c]));
    };
    // one unit: `dn` receives layer 1 of the next unit (fragments wn, column tile cn) while `dc` (this unit's) is converted
    auto unit = [&](actor_f16v &dn, const actor_u4 (&wn)[FT], int cn, const actor_f16v &dc, const actor_u4 (&w)[FT], int c, auto firstc) {
        Conv c0, c1;
        actor_static_for<3>([&](auto kc) {                        // phase A
            constexpr int k = decltype(kc)::value;
            layer1(dn, wn, cn, kc);
            actor_static_for<16>([&](auto ac) {
                constexpr int q = decltype(ac)::value;
                if constexpr (q >= (k == 0 ? 0 : k == 1 ? 6 : 11) && q < (k == 0 ? 6 : k == 1 ? 11 : 16)) conv_atom(c0, dc, 0, ac);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        actor_static_for<AT>([&](auto tc) {                       // phase B (the conversion of half 1 rides behind action tile 0's MFMAs)
            actor_static_for<3>([&](auto kc) {
                constexpr int k = decltype(kc)::value, t = decltype(tc)::value;
                layer2(w, c0, c, 0, tc, kc, firstc);
                if constexpr (t == 0)
                    actor_static_for<16>([&](auto ac) {
                        constexpr int q = decltype(ac)::value;
                        if constexpr (q >= (k == 0 ? 0 : k == 1 ? 6 : 11) && q < (k == 0 ? 6 : k == 1 ? 11 : 16)) conv_atom(c1, dc, 1, ac);
                    });
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        actor_static_for<AT>([&](auto tc) {                       // phase C
            actor_static_for<3>([&](auto kc) {
                layer2(w, c1, c, 1, tc, kc, std::false_type{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    actor_f16v da, db;                       // layer-1 accumulators of column tile 0 / 1 (they take turns)
    actor_static_for<3>([&](auto kc) { layer1(da, wf, 0, kc); });
    if constexpr (HT_ > 0) {
        // HT_ tiles, an even number: the tile loop runs two tiles per trip and the two fragment buffers take turns -- no copy of
        // the 24 fragment registers per tile, and the trip count is known (no zero-trip path with its own register moves)
        static_assert(HT_ % 2 == 0, "the two-tiles-per-trip layout needs an even tile count");
#pragma unroll
        for (int t = 0; t < AT; ++t) { d2[t][0] = zero16; d2[t][1] = zero16; }
        actor_u4 nx[FT];
#pragma unroll 1
        for (int a = 0; a < HT_; a += 2) {
#pragma unroll
            for (int f = 0; f < FT; ++f) nx[f] = wl[((size_t)(a + 1) * FT + f) * 64];
            __builtin_amdgcn_sched_barrier(0);
            unit(db, wf, 1, da, wf, 0, std::false_type{});
            unit(da, nx, 0, db, wf, 1, std::false_type{});
            const int an = a + 2 < HT_ ? a + 2 : a + 1;      // (the last tile looks ahead at itself: that layer 1 lands in an accumulator nobody reads)
#pragma unroll
            for (int f = 0; f < FT; ++f) wf[f] = wl[((size_t)an * FT + f) * 64];
            __builtin_amdgcn_sched_barrier(0);
            unit(db, nx, 1, da, nx, 0, std::false_type{});
            unit(da, wf, 0, db, nx, 1, std::false_type{});
        }
    } else {
#pragma unroll
        for (int t = 0; t < AT; ++t) { d2[t][0] = zero16; d2[t][1] = zero16; }
#pragma unroll 1
        for (int a = 0; a < HT; ++a) {
            actor_u4 nx[FT];
            const int an = a + 1 < HT ? a + 1 : a;          // (the last iteration re-requests its own tile: no branch around loads,
#pragma unroll                                              //  and its look-ahead layer 1 lands in an accumulator nobody reads)
            for (int f = 0; f < FT; ++f) nx[f] = wl[((size_t)an * FT + f) * 64];
            __builtin_amdgcn_sched_barrier(0);
            unit(db, wf, 1, da, wf, 0, std::false_type{});
            unit(da, nx, 0, db, wf, 1, std::false_type{});
#pragma unroll
            for (int f = 0; f < FT; ++f) wf[f] = nx[f];
        }
    }
    // ---- logits to their own lane: register r of tile 0 against register r of tile 1; afterwards every lane holds, of ITS
    //      sample, action 32 t + (r & 3) + 8 (r >> 2) in the first and that action + 4 in the second
    float lg[SLOTS];
    {
        const float inv_scale = weights[0];
        const float *b2 = weights + kActorB2Offset;
#pragma unroll
        for (int t = 0; t < AT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q0 = 32 * t + (r & 3) + 8 * (r >> 2), q1 = q0 + 4;
                if (q0 >= SLOTS) continue;
                float u = d2[t][0][r], v = d2[t][1][r];
                actor_swap32(u, v);
                lg[q0] = fmaf(u, inv_scale, b2[q0]);
                if (q1 < SLOTS) lg[q1] = fmaf(v, inv_scale, b2[q1]);
            }
    }
    if (A < SLOTS) {                                 // uniform; the reference's na = 12 fills every slot of AT = 1
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) lg[q] = (q < A) ? lg[q] : -INFINITY;
    }
    float m = lg[0];
#pragma unroll
    for (int q = 1; q < SLOTS; ++q) m = fmaxf(m, lg[q]);
    float ex[SLOTS], S = 0.0f;
    const float mneg = -m * 1.44269504088896340736f;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        ex[q] = __builtin_amdgcn_exp2f(fmaf(lg[q], 1.44269504088896340736f, mneg));   // masked slots: exp2(-inf) = 0
        S += ex[q];
    }
    if (WANT_PROBS && probs) {
        const float inv = __builtin_amdgcn_rcpf(S);
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
    // word (step & 3) of the block, picked with bit selects: written as ternaries the compiler made it an indexed read of the
    // block -- which put the block into scratch memory, one dependent scratch load per step in front of the draw
    const uint32_t m1 = 0u - (step & 1u), m2 = 0u - ((step >> 1) & 1u);
    const uint32_t w01 = (rng.r.v[0] & ~m1) | (rng.r.v[1] & m1), w23 = (rng.r.v[2] & ~m1) | (rng.r.v[3] & m1);
    const uint32_t word = (w01 & ~m2) | (w23 & m2);
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
