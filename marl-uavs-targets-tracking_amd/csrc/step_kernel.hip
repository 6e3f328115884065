// step_kernel.hip -- the fused environment step for gfx950 (CDNA4, wave64).
//
// One launch advances every environment of the batch by T >= 1 steps.  It is the
// batched, fused restatement of Environment.step (reference
// src/environment.py:120-164) and of everything it calls:
//   TARGET.update_position          agent/target.py:27-60
//   UAV.update_position             agent/uav.py:83-99   (+ discrete_action :73-81)
//   UAV.observe_target / observe_uav agent/uav.py:101-147 (sequential-update view)
//   UAV.get_local_state             agent/uav.py:156-197
//   raw reward terms                agent/uav.py:199-260
//   clip_and_normalize + weights    utils/data_util.py:43-56, environment.py:206-220
//   cooperative reward (mean)       agent/uav.py:293-310
//   calculate_covered_target        environment.py:246-253
//
// Mapping.  One lane per (env, uav).  A workgroup owns E = floor(WGS / N) whole
// environments, so an environment never spans workgroups and the only
// synchronisation is the workgroup barrier.  Pose, heading sin/cos and action of the
// lane's own UAV live in registers across the T steps; HBM sees the state once per
// launch and, per step, only the action read and the output writes.
//
// LDS tables are laid out for packed fp32 (v_pk_add/mul/fma_f32 do two values per lane
// in one 4-cycle VALU slot, and the kernel is VALU-issue-bound): agents are stored in
// PAIRS, structure-of-arrays inside the pair --
//   UAV pair row    (x0,x1,y0,y1) (cos0,cos1,sin0,sin1) (a0,a1,z0,z1)    48 B
//   target pair row (x0,x1,y0,y1) (cos0,cos1,sin0,sin1)                  32 B
// so one packed instruction handles the same quantity of two peers.  The UAV table has
// two copies per pair (this step's post-move poses and the previous step's): the
// reference moves UAVs one after the other (environment.py:133-138), so UAV i sees peers
// j < i after their move and peers j > i before it.  Lane i picks the copy per PAIR with
// one address select (post-move iff 2*jp < i); the one pair that holds i itself then
// contains a known self term (own post-move pose for odd i, own pre-move pose for even
// i), which is subtracted after the sweep instead of testing j != i per pair.  Lanes of
// one environment read the same row, which the LDS serves as a broadcast.
//
// No MFMA: there is no dense contraction on this path.

#include "actor.h"
#include "greedy.h"
#include "philox.h"

#include <cstdlib>

// Partial unrolling of the pair sweeps (in agent pairs): full unrolling lets the scheduler
// hoist every LDS table read of the environment into registers (256 VGPRs + scratch).
#define UAVTRACK_UNROLL_U 5
#define UAVTRACK_UNROLL_T 5

namespace uavtrack {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

struct Acc {
    // peers (uav.py:124-147 rows, folded into sums)
    float cntU, iwU, sxU, syU, scU, ssU, saU;
    // targets (uav.py:101-122 rows)
    float cntT, iwT, sxT, syT, scT, ssT;
    float trk;   // uav.py:199-212
    float dup;   // uav.py:214-229
};

constexpr int kLoneActorTiles = 4;         // hidden width 128 (configs/MAAC.yaml): the tile count the single-wavefront actor rollout is laid out for
constexpr float kSymMagic = 12582912.0f;   // 1.5 * 2^23 = 0x4B400000
constexpr float kFar = 1.0e18f;   // padding agent of an odd-sized pair: every range test fails, 0 * kFar = 0

__device__ __forceinline__ float fast_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float fast_exp2(float v) { return __builtin_amdgcn_exp2f(v); }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float v) { return (v2f){v, v}; }
// Range test of two squared distances in one instruction: 1.0 where d2 <= K else 0.0, as
// clamp(fma(d2, -S, nextafter(K) * S)) with S a power of two >= 1 / ulp(K) (fold_constants).  The
// FMA rounds the exact value once, so its sign is exact; the smallest positive value is
// ulp(K) * S >= 1 and clamps to exactly 1; overflow gives -inf -> 0; NaN clamps to 0 (DX10 clamp),
// the same as the compare it replaces.  There is no packed compare/select, so this replaces
// 2 v_cmp + 2 v_cndmask.
// a wave-uniform value parked in a vector register (opaque to the compiler: it stays there)
__device__ __forceinline__ float vreg(float x)
{
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
    return v;
}
template <bool VC = false>
__device__ __forceinline__ v2f pk_le_mask(v2f d2, v2f neg_scale, float c)
{
    v2f r;
    const v2f c2 = splat(c);
    if constexpr (VC) {
        asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(d2), "v"(neg_scale), "v"(c2));
        return r;
    }
    // (clang treats inline asm as convergent in HIP, which blocks partial unrolling of the sweeps: the
    // Makefile builds this file with -fno-convergent-functions; every cross-lane operation here is a
    // builtin that carries its own convergence)
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(d2), "v"(neg_scale), "s"(c2));
    return r;
}

// The squared distance every range test of the path uses (also in pmi_kernel.hip): dx*dx
// rounded, then one fma.  Scalar and packed forms give identical bits.
__device__ __forceinline__ float dist2(float dx, float dy) { return fmaf(dy, dy, dx * dx); }

// Inclusive prefix sum of a small integer over the 64 lanes of a wavefront, in lane order, on the DPP paths: Kogge-Stone inside
// each row of 16 (shifts that leave the row bring in 0), then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into
// rows 2 and 3.  Every lane must be active (inactive lanes would contribute whatever their register holds).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_add_i(int v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_inclusive_sum(int v)
{
    v = dpp_add_i<0x111>(v);           // row_shr:1
    v = dpp_add_i<0x112>(v);           // row_shr:2
    v = dpp_add_i<0x114>(v);           // row_shr:4
    v = dpp_add_i<0x118>(v);           // row_shr:8
    v = dpp_add_i<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add_i<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}

// neighbour masks are 32-bit words up to 32 UAVs (64-bit arithmetic costs two to four instructions per operation)
__device__ __forceinline__ int mpop(unsigned v) { return __popc(v); }
__device__ __forceinline__ int mpop(unsigned long long v) { return __popcll(v); }
__device__ __forceinline__ int mffs(unsigned v) { return __ffs((int)v); }
__device__ __forceinline__ int mffs(unsigned long long v) { return __ffsll((long long)v); }
template <int N_> struct NbMaskOf { using type = unsigned long long; };
template <> struct NbMaskOf<20> { using type = unsigned; };
template <> struct NbMaskOf<10> { using type = unsigned; };
template <> struct NbMaskOf<5> { using type = unsigned; };

// Workgroups are dealt round-robin over the 8 XCDs; give consecutive environment
// groups to one XCD so neighbouring output spans land in the same L2 (speed only).
__device__ __forceinline__ int xcd_group(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    return xcd < r ? xcd * (q + 1) + slot : r * (q + 1) + (xcd - r) * q + slot;
}

// (h + pi) % (2 pi) - pi (uav.py:97) as h - 2 pi * rint(h / (2 pi)): three instructions, no branch, any magnitude, and no
// mantissa bit is spent when no wrap is needed (rint gives 0 and the FMA returns h).  The result lies in [-pi, pi]; Python's
// % gives [-pi, pi): the two differ only for a heading of exactly +pi (kept here, -pi there) -- the same direction.
__device__ __forceinline__ float wrap_heading(float h)
{
    return fmaf(-rintf(h * (1.0f / kTwoPi)), kTwoPi, h);
}

// sin/cos for |h| <= pi (headings are kept wrapped) on the hardware's own v_sin_f32 / v_cos_f32 (argument in revolutions).
// Measured on gfx950 over [-pi, pi]: the instructions themselves are good to 1.25e-7; the rounding of t = h / (2 pi) adds up
// to 1.9e-7 rad at |h| = pi, so that error -- exact from one FMA, plus the low part of 1 / (2 pi) -- goes back in as a
// first-order correction (sin += d cos, cos -= d sin): |err| <= 1.4e-7 for both, what the 22-instruction quadrant reduction
// + minimax polynomials gave that these eight instructions replace.  (It matters next to the origin, where the uav.py:165
// weights multiply an observation row by up to 1e5.)
__device__ __forceinline__ void sincos_wrapped(float h, float *s, float *c)
{
    constexpr float kInvHi = 0.15915494309189535f;                                      // fl32(1 / (2 pi))
    constexpr float kInvLo = (float)(0.15915494309189533576888 - (double)kInvHi);      // 1 / (2 pi) - kInvHi
    const float t = h * kInvHi;
    const float s0 = __builtin_amdgcn_sinf(t), c0 = __builtin_amdgcn_cosf(t);
    const float d = fmaf(h, kInvLo, fmaf(h, kInvHi, -t)) * kTwoPi;                      // 2 pi * (h / (2 pi) - t)
    *s = fmaf(d, c0, s0);
    *c = fmaf(-d, s0, c0);
}

// (ocml's sincosf, with its large-argument path, only for injected state whose heading is out of range)
__device__ __forceinline__ void sincos_any(float h, float *s, float *c)
{
    if (__builtin_expect(fabsf(h) <= kPi, 1)) sincos_wrapped(h, s, c);
    else sincosf(h, s, c);
}

// The two barriers of a step.  A single-wavefront workgroup (LONE) needs neither the barrier nor the LDS drain in
// front of it: a wavefront's LDS operations execute in program order, so its reads see its own lanes' earlier writes;
// only the compiler must be kept from moving them across.
// (A macro, not a function: this file is built with -fno-convergent-functions, and a barrier wrapped in a function of
// ours would lose its convergence -- the optimiser then moved it relative to divergent code in one kernel variant.)
#define UAVTRACK_STEP_BARRIER()                                          \
    do {                                                                 \
        if (LONE) {                                                      \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
            __builtin_amdgcn_wave_barrier();                             \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
        } else {                                                         \
            __syncthreads();   /* (an LDS-only barrier, no vmcnt drain, measured the same: tools/experiments) */ \
        }                                                                \
    } while (0)

// uniform base + zero-extended 32-bit per-lane byte offset: the form the global_load/store `saddr` encoding takes
template <typename T>
__device__ __forceinline__ T *at(T *base, unsigned byte_off)
{
    return (T *)((char *)base + byte_off);   // (char-pointer arithmetic keeps the global address space; an integer round trip would not)
}

// Outputs are written once and never read back by the launch that writes them.  The arrays a wavefront writes in whole
// lines -- reward, the three terms, actions, coverage, the MAAC-R records and pairs: 4 or 8 contiguous bytes per lane -- go
// out as streaming (non-temporal) stores: left as ordinary stores they linger dirty in the L2 until evicted, and the
// single-wavefront headline rollout, whose one wave per SIMD has nothing to hide a stalled store queue behind, ran 9 % slower
// (0.467-0.478 vs 0.429-0.432 ms per 200 steps, 30 launches each, A/B on one box).  The observation rows stay ordinary
// stores: a lane's 48 bytes leave as three 16-byte pieces, the L2 merges them into whole lines, and streamed they reach
// the memory as partial writes (WRITE_SIZE 1.45 GB per launch instead of 1.07) -- 0.42 ms at best, but 0.54-0.62 ms
// median once the write queues fill.
typedef unsigned v2u_nt __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void out_store(T *ptr, T v)
{
    __builtin_nontemporal_store(v, ptr);
}
__device__ __forceinline__ void out_store(uint2 *ptr, unsigned a, unsigned b)
{
    const v2u_nt q = {a, b};
    __builtin_nontemporal_store(q, reinterpret_cast<v2u_nt *>(ptr));
}

// ---- LDS table geometry (float4 units) --------------------------------------------------
// observation staging: one 64 x 3 float4 block per wavefront of the E * N lanes (kObsStage)
__device__ __host__ __forceinline__ int obs_stage_f4(int lanes) { return ((lanes + 63) / 64) * 192; }
__device__ __host__ __forceinline__ int pairs_of(int n) { return (n + 1) >> 1; }
// coverage words per environment: 12 target PAIRS per word (two interleaved base-4 digit strings of 12 digits each,
// accumulated exactly in fp32 by the packed sweep: see sweep_fast)
constexpr int kCovPairs = 12;
__device__ __host__ __forceinline__ int cov_words(int m) { return (pairs_of(m) + kCovPairs - 1) / kCovPairs; }
// env stride of the UAV table: pairs * 2 copies * 3 float4, plus one float4 when the stride would
// be a multiple of 64 dwords (rows of different envs read by one lane group would share banks)
__device__ __host__ __forceinline__ int ustride_of(int n)
{
    const int s = pairs_of(n) * 6;
    return (s * 4) % 64 == 0 ? s + 1 : s;
}
__device__ __host__ __forceinline__ int tstride_of(int m)
{
    const int s = pairs_of(m) * 2;
    return (s * 4) % 64 == 0 ? s + 1 : s;
}

// ---- symmetric duplicate term (MAAC reward mode): per environment the post-move poses once more as plain arrays of
// length N + N/2 (entry N + j repeats entry j, so "peer i + k" needs no wrap-around), and 2 N fixed-point accumulators
__device__ __host__ __forceinline__ int sym_pose_len(int n) { return n + n / 2 + 1; }   // (+1: the packed evaluation of the last peer reads its neighbour)
__device__ __host__ __forceinline__ int sym_words(int n, bool z3) { return sym_pose_len(n) * (z3 ? 3 : 2) + 2 * n; }

// scalar views of the pair rows (slow paths, table fills)
struct UavRow { float x, y, c, s, a, z; };
__device__ __forceinline__ UavRow uav_elem(const float4 *rows, int j)   // rows: env base + copy * 3
{
    const float *f = reinterpret_cast<const float *>(rows + (j >> 1) * 6);
    const int b = j & 1;
    return UavRow{f[b], f[2 + b], f[4 + b], f[6 + b], f[8 + b], f[10 + b]};
}
__device__ __forceinline__ void uav_store(float4 *rows, int j, float x, float y, float c, float s, float a, float z)
{
    float *f = reinterpret_cast<float *>(rows + (j >> 1) * 6);
    const int b = j & 1;
    f[b] = x; f[2 + b] = y; f[4 + b] = c; f[6 + b] = s; f[8 + b] = a; f[10 + b] = z;
}

// Specialised swarms of up to 64 UAVs keep `action + K` in the table's action slot, K = StepParams::act_bias a power of
// two above n_uav * na * nc: the peer sweep's sum of (mask * slot) then carries the neighbour COUNT in its high part and the
// action sum in its low part, both exact integers in fp32, and the count needs no accumulator of its own (one packed add
// less per peer pair).  fold_constants / uavtrack_create keep n_uav * (K + na * nc) below 2^24 for these shapes.
__device__ __host__ constexpr bool act_bias_shape(int n_spec) { return n_spec > 0 && n_spec <= 64; }

// ---------------------------------------------------------------------------------------
// Pair sweeps of one UAV, fast path: two agents per packed instruction, no per-agent
// `j != i` test (self terms are subtracted afterwards).  The uav.py:165/179 weight is 1
// here (see sweep_weighted).
template <int N_, int M_, bool Z3, bool NB, int PF, bool SYM = false, bool VC = false, bool NBSEQ = false>
__device__ __forceinline__ void sweep_fast(const StepParams &p, int N, int M, int i,
                                           const float4 *__restrict__ rowNew, const float4 *__restrict__ rowOld,
                                           const float4 *__restrict__ trow, const v2f *__restrict__ tzrow,
                                           unsigned *__restrict__ covw, int covbase,
                                           float xi, float yi, float zi, float ci, float si, float ai,
                                           float xo, float yo, float zo, float co, float so, float ao, Acc &a,
                                           unsigned long long &nbmask, const float4 *const *__restrict__ rsel = nullptr,
                                           const float4 *__restrict__ ubase = nullptr, unsigned copyw = 0)   // (copy_row)
{
    const v2f xi2 = splat(xi), yi2 = splat(yi), zi2 = splat(zi);
    const v2f nscale = splat(p.le_neg_scale);
    constexpr bool kActBias = act_bias_shape(N_);
    const int NP = pairs_of(N_ > 0 ? N_ : N), MP = pairs_of(M_ > 0 ? M_ : M);
    // Up to 10 pair rows (N <= 20) the whole peer sweep is unrolled: measured 4-7 % faster than
    // unrolling by 5 and, with the pair layout, still 127 VGPRs; larger N keeps the partial unroll.
    constexpr int UU = (N_ > 0 && (N_ + 1) / 2 <= 10) ? (N_ + 1) / 2 : UAVTRACK_UNROLL_U;

    // ---- targets: observe_target (<= dp), tracking reward (<= dp), coverage (< dp)
    v2f sx = splat(0.f), sy = splat(0.f), sc = splat(0.f), ss = splat(0.f), cnt = splat(0.f), trk = splat(0.f);
    v2f cov = splat(0.f);   // coverage (d < dp, strict) of the pair's two targets as base-4 digits: cov = 4 * cov + {0, 1}
    // PF > 0: table rows are requested PF peer rows ahead of their use and pinned there (sched_barrier lets ALU work
    // cross, LDS reads not): a wavefront that runs alone on its SIMD -- the small-grid geometry -- has nothing else to
    // cover an LDS round trip with (-6 % at 4096 envs; at a chip-filling batch the extra registers cost 5 %, so the
    // 4-wave groups keep PF = 0).  Specialised planar shapes only.
    constexpr int QM = M_ > 0 ? (M_ + 1) / 2 : 1, QN = N_ > 0 ? (N_ + 1) / 2 : 1;
    constexpr bool kPrefetch = PF > 0 && N_ > 0 && M_ > 0 && QN <= 10 && QM <= 5 && !Z3;
    float4 Q0[QM], Q1[QM], PN0[QN], PM0[QN], PM1[QN];
    v2f PM2[QN];
    // rsel (single-wavefront headline variant, even / odd steps as two copies of the step loop): the selected copy's row
    // base per pair, computed once per launch -- which copy is post-move alternates with the step, the choice per pair and
    // lane does not
    // The copy of pair jp this lane reads (one select serves pose, heading, action and z).  Up to 32 pairs the choice comes
    // from a per-lane bit word (copyw: bit jp = which copy holds what this lane sees of pair jp, this step): a bit-field
    // extract and a multiply-add on the row address, one instruction less than compare + select + add.
    auto copy_row = [&](int jp) -> const float4 * {
        if constexpr (N_ > 0 && N_ <= 64)       // (every caller of a specialised shape passes ubase and copyw)
            return reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(ubase) + __builtin_amdgcn_ubfe(copyw, (unsigned)jp, 1u) * 48u);
        return (2 * jp < i) ? rowNew : rowOld;
    };
    auto fetch_row = [&](int jq) {
        const float4 *rq = rsel ? rsel[jq] : copy_row(jq);
        PN0[jq] = rowNew[jq * 6]; PM0[jq] = rq[jq * 6]; PM1[jq] = rq[jq * 6 + 1];
        PM2[jq] = *reinterpret_cast<const v2f *>(&rq[jq * 6 + 2]);
    };
    if (kPrefetch) {
#pragma unroll
        for (int kp = 0; kp < QM; ++kp) { Q0[kp] = trow[kp * 2]; Q1[kp] = trow[kp * 2 + 1]; }
#pragma unroll
        for (int jq = 0; jq < PF && jq < QN; ++jq) fetch_row(jq);
        __builtin_amdgcn_sched_barrier(0x407);
    }
    // One coverage word (12 target pairs) at a time: an inner loop of constant trip count that unrolls without a
    // remainder, and the flush of the word's bits behind it (a `kp % 12` test inside a partially unrolled loop cost a scalar
    // division and a branch per pair, and cut the sweep into one basic block per pair).
    auto target_pair = [&](int kp) {
        const float4 q0 = kPrefetch ? Q0[kPrefetch ? kp : 0] : trow[kp * 2], q1 = kPrefetch ? Q1[kPrefetch ? kp : 0] : trow[kp * 2 + 1];
        const v2f dx = (v2f){q0.x, q0.y} - xi2, dy = (v2f){q0.z, q0.w} - yi2;
        v2f d2 = pk_fma(dy, dy, dx * dx);
        if (Z3) { const v2f dz = tzrow[kp] - zi2; d2 = pk_fma(dz, dz, d2); }
        const v2f dist = {fast_sqrt(d2.x), fast_sqrt(d2.y)};
        const v2f mm = pk_le_mask<VC>(d2, nscale, p.le_dp2);
        sx = pk_fma(mm, dx, sx);
        sy = pk_fma(mm, dy, sy);
        sc = pk_fma(mm, (v2f){q1.x, q1.y}, sc);
        ss = pk_fma(mm, (v2f){q1.z, q1.w}, ss);
        cnt += mm;
        trk = pk_fma(mm, dist, trk);                 // sum m d: tracking reward sum m (1 + (dp - d)/dp) = 2 cnt - (sum m d)/dp
        cov = pk_fma(cov, splat(4.0f), pk_le_mask<VC>(d2, nscale, p.lt_dp2));
    };
    auto flush_word = [&](int w) {                   // 12 digits < 2^24: exact in fp32
        const unsigned bits = (unsigned)cov.x | ((unsigned)cov.y << 1);
        if (bits) atomicOr(&covw[covbase + w], bits);
        cov = splat(0.f);
    };
    if constexpr (M_ > 0) {
        constexpr int MPc = (M_ + 1) / 2, CWc = (MPc + kCovPairs - 1) / kCovPairs;
#pragma unroll
        for (int w = 0; w < CWc; ++w) {
            constexpr int kInner = MPc <= UAVTRACK_UNROLL_T ? MPc : 6;      // (12 = 2 x 6; a last word shorter than 6 unrolls in full)
            const int k1 = (w + 1) * kCovPairs < MPc ? (w + 1) * kCovPairs : MPc;
#pragma unroll kInner
            for (int kp = w * kCovPairs; kp < k1; ++kp) target_pair(kp);
            flush_word(w);
        }
    } else {
        for (int w = 0; w * kCovPairs < MP; ++w) {
            const int k1 = min((w + 1) * kCovPairs, MP);
#pragma unroll 4
            for (int kp = w * kCovPairs; kp < k1; ++kp) target_pair(kp);
            flush_word(w);
        }
    }
    a.cntT = cnt.x + cnt.y; a.iwT = a.cntT;
    a.sxT = sx.x + sx.y; a.syT = sy.x + sy.y; a.trk = fmaf(-(trk.x + trk.y), p.inv_dp, 2.0f * a.cntT);
    // sum_k m (c_k r - c_i) = r sum_k m c_k - c_i cnt  (r = target speed / uav speed, uav.py:116)
    a.scT = fmaf(sc.x + sc.y, p.vratio, -ci * a.cntT);
    a.ssT = fmaf(ss.x + ss.y, p.vratio, -si * a.cntT);

    // ---- peers: duplicate punishment on post-move poses (<= 2dp), observe_uav on the
    //      sequential view (<= dc)
    sx = splat(0.f); sy = splat(0.f); sc = splat(0.f); ss = splat(0.f); cnt = splat(0.f);
    v2f sa = splat(0.f), dup = splat(0.f);
    constexpr bool NBF = NB && N_ > 0 && (N_ + 1) / 2 <= kCovPairs;
    v2f nbf = splat(0.f);
    const float4 *ub = ubase;
#pragma unroll UU
    for (int jp = 0; jp < NP; ++jp) {
        // (the row of pair jp in the copy this lane reads; with the bit word the pair's offset rides in a pointer that is
        //  bumped per pair -- after unrolling: immediate offsets and one add per loop body, not one per pair)
        const float4 *rp;
        if (rsel && kPrefetch) rp = rsel[kPrefetch ? jp : 0] + jp * 6;
        else if constexpr (N_ > 0 && N_ <= 64) rp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(ub) + __builtin_amdgcn_ubfe(copyw, (unsigned)jp, 1u) * 48u);
        else rp = copy_row(jp) + jp * 6;
        ub += 6;
        if (kPrefetch) {
            if (jp + PF < NP) fetch_row(jp + PF);
            __builtin_amdgcn_sched_barrier(0x407);
        }
        const float4 n0 = kPrefetch ? PN0[kPrefetch ? jp : 0] : rowNew[jp * 6];
        const float4 m0 = kPrefetch ? PM0[kPrefetch ? jp : 0] : rp[0], m1 = kPrefetch ? PM1[kPrefetch ? jp : 0] : rp[1];
        const float4 m2 = kPrefetch ? make_float4(PM2[kPrefetch ? jp : 0].x, PM2[kPrefetch ? jp : 0].y, 0.f, 0.f) : rp[2];
        const v2f dxn = (v2f){n0.x, n0.y} - xi2, dyn = (v2f){n0.z, n0.w} - yi2;
        const v2f dxm = (v2f){m0.x, m0.y} - xi2, dym = (v2f){m0.z, m0.w} - yi2;
        v2f d2n = pk_fma(dyn, dyn, dxn * dxn), d2m = pk_fma(dym, dym, dxm * dxm);
        if (Z3) {
            const v2f dzm = (v2f){m2.z, m2.w} - zi2;
            d2m = pk_fma(dzm, dzm, d2m);
            // (the post-move distance serves the duplicate term -- unless that is shared, SYM -- and the neighbour mask of the
            //  cooperative modes: MAAC-R has both SYM and the mask.  Round 4 shared MAAC-R's duplicate term and left the mask on
            //  the planar distance in 3-D; found by tests/fuzz_api.py in round 5)
            if (!SYM || NB) {
                const float4 n2 = rowNew[jp * 6 + 2];
                const v2f dzn = (v2f){n2.z, n2.w} - zi2;
                d2n = pk_fma(dzn, dzn, d2n);
            }
        }
        if (NBF)       // <= 12 pairs: the same packed mask + base-4 digits as the coverage bits (bit order fixed up below)
            // (NBSEQ: on the sequential view, whose lower half -- j < i -- is the post-move poses; the caller keeps that half
            //  and takes the upper one from the partners, see nb_exchange: the post-move rows are not read at all)
            nbf = pk_fma(nbf, splat(4.0f), pk_le_mask<VC>(NBSEQ ? d2m : d2n, nscale, p.le_dp2));
        else if (NB)   // cooperative modes (N <= 64): neighbours (d <= dp on post-move poses, uav.py:278) as a bit mask
            nbmask |= ((unsigned long long)(d2n.x <= p.dp2 ? 1u : 0u) | (unsigned long long)(d2n.y <= p.dp2 ? 2u : 0u)) << (2 * jp);
        if (!SYM) {   // (SYM: the duplicate term is shared between the two UAVs of a pair, see sym_dup)
            const v2f w = pk_fma((v2f){fast_sqrt(d2n.x), fast_sqrt(d2n.y)}, splat(-p.exp_k1), splat(p.exp_k0));
            dup = pk_fma(pk_le_mask<VC>(d2n, nscale, p.le_two_dp2), (v2f){fast_exp2(w.x), fast_exp2(w.y)}, dup);
        }
        const v2f mm = pk_le_mask<VC>(d2m, nscale, p.le_dc2);
        sx = pk_fma(mm, dxm, sx);
        sy = pk_fma(mm, dym, sy);
        sc = pk_fma(mm, (v2f){m1.x, m1.y}, sc);
        ss = pk_fma(mm, (v2f){m1.z, m1.w}, ss);
        sa = pk_fma(mm, (v2f){m2.x, m2.y}, sa);
        if (!kActBias) cnt += mm;
    }
    if (NBF) {   // Horner left pair jp at digit NP-1-jp: slot 1 into the even bits, reverse, align -> bit j = UAV j
        const unsigned w = (unsigned)nbf.y | ((unsigned)nbf.x << 1);
        nbmask = (unsigned long long)(__brev(w) >> (32 - 2 * NP));
    }
    // ---- self terms.  Post-move table: distance 0, exp2(k0).  Sequential view: odd i shares its pair with j = i - 1 < i,
    //      so it saw its own post-move pose: d = 0, always in range, every difference to itself 0 -- for those lanes
    //      sum_j m (c_j - c_i) over the OTHER peers is just (sum over all m c) - c_i * (count over all), no correction term.
    //      Even i saw its own pre-move pose, in range or not (mse): the same expression with the count over all, then
    //      - mse * (c_old - c_i), and the pre-move offset taken out of the position sums.
    const bool odd = i & 1;
    const float dxs = xo - xi, dys = yo - yi;
    float d2s = dist2(dxs, dys);
    if (Z3) { const float dzs = zo - zi; d2s = fmaf(dzs, dzs, d2s); }
    const float mse = (!odd && d2s <= p.dc2) ? 1.0f : 0.0f;
    a.dup = dup.x + dup.y - fast_exp2(p.exp_k0);
    float cnt_all = cnt.x + cnt.y, sa_all = sa.x + sa.y;
    if (kActBias) {   // the table's action slot holds a + K: sum m (a + K) = K * count + sum m a, both exact small integers
        cnt_all = floorf(sa_all * p.inv_act_bias);
        sa_all = fmaf(-p.act_bias, cnt_all, sa_all);
    }
    a.cntU = cnt_all - (odd ? 1.0f : mse); a.iwU = a.cntU;
    a.sxU = fmaf(-mse, dxs, sx.x + sx.y);
    a.syU = fmaf(-mse, dys, sy.x + sy.y);
    a.scU = fmaf(-mse, co - ci, fmaf(-ci, cnt_all, sc.x + sc.y));
    a.ssU = fmaf(-mse, so - si, fmaf(-si, cnt_all, ss.x + ss.y));
    a.saU = fmaf(-mse, ao - ai, fmaf(-ai, cnt_all, sa_all));
}

// The duplicate-tracking term (uav.py:214-229) is symmetric on the post-move poses: g(d_ij) enters UAV i's sum and UAV j's.
// Each lane evaluates only the peers i + 1 .. i + N/2 (cyclically: a balanced half of the pair matrix), adds the value to its
// own running sum and, through the LDS, to the partner's accumulator -- half the square roots and exponentials of the full
// sweep.  The accumulators are FIXED-POINT integers (g * 2^kSymBits, at most e * N * 2^kSymBits < 2^32): integer addition
// commutes, so the LDS atomics leave the same bits whatever order the wavefronts arrive in (a float accumulation would
// not), and the quantisation (2^-21 per term) is below the fp32 rounding of the sum it replaces.  The
// accumulator array has 2 N entries: lane i adds to entry i + k without wrapping and reads entries i and i + N.
// Even N: the opposite peer (k = N/2) is evaluated by both ends, each keeping it for itself.
template <int N_, bool Z3, bool VC = false>
__device__ __forceinline__ unsigned sym_dup(const StepParams &p, int N, int i, const float *__restrict__ sx, const float *__restrict__ sy,
                                            const float *__restrict__ sz, unsigned *__restrict__ dacc, float xi, float yi, float zi)
{
    const int n = N_ > 0 ? N_ : N;
    const int KP = (n - 1) / 2;                    // peers whose value is shared with the partner
    const v2f xi2 = splat(xi), yi2 = splat(yi), zi2 = splat(zi), nscale = splat(p.le_neg_scale);
    unsigned own = 0;
    struct P2 { v2f x, y, z; };
    auto load2 = [&](int k) {                      // poses of peers i + k and i + k + 1
        P2 q;
        q.x = (v2f){sx[i + k], sx[i + k + 1]};
        q.y = (v2f){sy[i + k], sy[i + k + 1]};
        q.z = Z3 ? (v2f){sz[i + k], sz[i + k + 1]} : splat(0.f);
        return q;
    };
    auto eval2 = [&](const P2 &q) {                // fixed-point g of each
        const v2f dx = q.x - xi2, dy = q.y - yi2;
        v2f d2 = pk_fma(dy, dy, dx * dx);
        if (Z3) { const v2f dz = q.z - zi2; d2 = pk_fma(dz, dz, d2); }
        const v2f w = pk_fma((v2f){fast_sqrt(d2.x), fast_sqrt(d2.y)}, splat(-p.exp_k1), splat(p.sym_k0));
        // float -> fixed point inside the FMA that applies the mask: 1.5 * 2^23 + v holds round(v) in its low mantissa bits
        // (0 <= v < 2^22), and the integer sums carry kSymMagic once per term -- N - 1 terms per UAV, taken off at the end
        const v2f g = pk_fma(pk_le_mask<VC>(d2, nscale, p.le_two_dp2), (v2f){fast_exp2(w.x), fast_exp2(w.y)}, splat(kSymMagic));
        return make_uint2(__float_as_uint(g.x), __float_as_uint(g.y));
    };
    // (the poses of the next two peers are requested before the current two are evaluated: the LDS round trip, the
    // square roots and the exponentials of consecutive iterations overlap)
    constexpr int UK = (N_ > 0 && (N_ - 1) / 2 <= 10) ? 5 : 4;     // pairs of peers per unrolled body
    const int last = KP + ((n & 1) == 0 ? 1 : 0);  // last peer index that is evaluated at all
    P2 cur = load2(1);
#pragma unroll UK
    for (int k = 1; k + 1 <= KP; k += 2) {
        P2 nxt = cur;
        if (k + 2 <= last) nxt = load2(k + 2);
        const uint2 q = eval2(cur);
        own += q.x + q.y;
        atomicAdd(&dacc[i + k], q.x);
        atomicAdd(&dacc[i + k + 1], q.y);
        cur = nxt;
    }
    if (KP & 1) {                                  // one shared peer left (`cur` holds it); with even N its neighbour is the opposite peer
        const uint2 q = eval2(cur);                // (entry i + KP + 1 <= i + N/2 exists in the doubled arrays either way)
        own += q.x;
        atomicAdd(&dacc[i + KP], q.x);
        if ((n & 1) == 0) own += q.y;              // k = N/2: kept, not shared
    } else if ((n & 1) == 0) {
        const uint2 q = eval2(cur);                // k = N/2 alone (its pair partner k + 1 is past the half: ignored)
        own += q.x;
    }
    return own;
}

// Literal form with the uav.py:165/179 weight min(dist((rel_x, rel_y), (abs_x, abs_y)), 1).
// It differs from 1 only when the UAV sits within ~2.5 m of the origin, so this path runs
// for the few wavefronts that hold such a UAV and favours clarity over speed: one agent at
// a time, the reference's own j != i test and per-agent j < i copy choice.
template <bool Z3>
__device__ __forceinline__ void sweep_weighted(const StepParams &p, int N, int M, int i,
                                               const float4 *__restrict__ rowNew, const float4 *__restrict__ rowOld,
                                               const float4 *__restrict__ trow, const v2f *__restrict__ tzrow,
                                               unsigned *__restrict__ covw, int covbase,
                                               float xi, float yi, float zi, float ci, float si, float ai, float abias, Acc &a)
{
    a = Acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned bits = 0;
#pragma unroll 1
    for (int k = 0; k < M; ++k) {
        const float *f = reinterpret_cast<const float *>(trow + (k >> 1) * 2);
        const int b = k & 1;
        const float dx = f[b] - xi, dy = f[2 + b] - yi, tc = f[4 + b], ts = f[6 + b];
        float d2 = dist2(dx, dy);
        if (Z3) { const float dz = reinterpret_cast<const float *>(tzrow)[k] - zi; d2 = fmaf(dz, dz, d2); }
        const float d = fast_sqrt(d2);
        const bool in = d2 <= p.dp2;
        const float m = in ? 1.0f : 0.0f;
        const float rx = dx * p.inv_dp - xi, ry = dy * p.inv_dp - yi;
        const float iw = in ? 1.0f / fminf(sqrtf(rx * rx + ry * ry), 1.0f) : 0.0f;
        a.scT = fmaf(iw, tc * p.vratio - ci, a.scT);
        a.ssT = fmaf(iw, ts * p.vratio - si, a.ssT);
        a.cntT += m;
        a.iwT += iw;
        a.sxT = fmaf(iw, dx, a.sxT);
        a.syT = fmaf(iw, dy, a.syT);
        a.trk = fmaf(m, fmaf(-d, p.inv_dp, 2.0f), a.trk);
        {   // the bit sweep_fast gives this target: digit (pairs in the word - 1 - pair index), slot k & 1
            const int kp = k >> 1, word = kp / kCovPairs, MPw = pairs_of(M);
            const int in_word = min(kCovPairs, MPw - word * kCovPairs);
            bits |= (d2 < p.dp2) ? (1u << (2 * (in_word - 1 - kp % kCovPairs) + (k & 1))) : 0u;
            if ((kp % kCovPairs == kCovPairs - 1 && (k & 1)) || k == M - 1) {
                if (bits) atomicOr(&covw[covbase + word], bits);
                bits = 0;
            }
        }
    }
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
        const bool other = (j != i);
        const UavRow nw = uav_elem(rowNew, j);
        const UavRow mx = uav_elem((j < i) ? rowNew : rowOld, j);
        const float dxn = nw.x - xi, dyn = nw.y - yi, dxm = mx.x - xi, dym = mx.y - yi;
        float d2n = dist2(dxn, dyn), d2m = dist2(dxm, dym);
        if (Z3) {
            const float dzn = nw.z - zi, dzm = mx.z - zi;
            d2n = fmaf(dzn, dzn, d2n);
            d2m = fmaf(dzm, dzm, d2m);
        }
        const float ex = fast_exp2(fmaf(fast_sqrt(d2n), -p.exp_k1, p.exp_k0));
        a.dup += (other && d2n <= p.two_dp2) ? ex : 0.0f;
        const bool inm = other && d2m <= p.dc2;
        const float m = inm ? 1.0f : 0.0f;
        const float rx = dxm * p.inv_dc - xi, ry = dym * p.inv_dc - yi;
        const float iw = inm ? 1.0f / fminf(sqrtf(rx * rx + ry * ry), 1.0f) : 0.0f;
        a.scU = fmaf(iw, mx.c - ci, a.scU);
        a.ssU = fmaf(iw, mx.s - si, a.ssU);
        a.saU = fmaf(iw, (mx.a - abias) - ai, a.saU);
        a.cntU += m;
        a.iwU += iw;
        a.sxU = fmaf(iw, dxm, a.sxU);
        a.syU = fmaf(iw, dym, a.syU);
    }
}

// ALLOUT: the caller passed every per-step output (obs, reward, terms, covered, done) -- the rollout a learner
// consumes and the benchmark workload; the nullable-pointer tests and their SGPR flags then fold away.
// EXTRAS: the launch uses the rarely wanted per-step extras (target trace, raw rewards, automatic reset); compiled out otherwise --
// their tests and parameters cost the plain rollout ~10 % when they sat in the same instantiation.
// LONE: built for single-wavefront workgroups on a grid of at most a few waves per SIMD (plan_geometry's small grid).
template <int N_, int M_, int MODE, bool Z3, int POLICY, bool ALLOUT = false, bool EXTRAS = false, bool LONE = false>
__global__ void __launch_bounds__(kMaxWorkgroup) rollout_kernel(const StepParams p_in)
{
    // Single-wavefront variants: a step uses ~40 float constants of StepParams beside ~20 pointers and offsets -- more than
    // the 102 scalar registers hold, so the compiler parks some in lanes of a vector register and fetches them back with a
    // v_readlane (a VALU instruction and a wait) wherever they are needed.  These variants run one or two wavefronts per
    // SIMD and use well under half of their vector registers: the constants that only ever feed per-lane arithmetic
    // move there for good.
    // (all single-wavefront variants, the actor one included since its matrix work moved to the 16-bit pipe: two
    //  wavefronts per SIMD either way, 1.448 -> 1.424 ms per 200-step actor rollout; not the 4-wave variants: no registers to spare)
    constexpr bool kVConst = LONE;
    StepParams p = p_in;
    if (kVConst) {
#define UAVTRACK_V(f) p.f = vreg(p_in.f)
        UAVTRACK_V(inv_dp); UAVTRACK_V(inv_dc); UAVTRACK_V(dp2); UAVTRACK_V(dc2); UAVTRACK_V(dup_k); UAVTRACK_V(sym_dup_k);
        UAVTRACK_V(vratio); UAVTRACK_V(inv_na_total); UAVTRACK_V(exp_k0); UAVTRACK_V(exp_k1); UAVTRACK_V(x_max); UAVTRACK_V(y_max);
        UAVTRACK_V(tt_ceil); UAVTRACK_V(inv_tt_ceil); UAVTRACK_V(alpha); UAVTRACK_V(beta); UAVTRACK_V(gamma);
        UAVTRACK_V(sym_k0); UAVTRACK_V(act_bias); UAVTRACK_V(inv_act_bias);
        UAVTRACK_V(le_neg_scale); UAVTRACK_V(le_dp2); UAVTRACK_V(lt_dp2); UAVTRACK_V(le_dc2); UAVTRACK_V(le_two_dp2);
        UAVTRACK_V(dtv_u); UAVTRACK_V(dtv_t); UAVTRACK_V(turn_unit); UAVTRACK_V(inv_na); UAVTRACK_V(two_dp2); UAVTRACK_V(dp); UAVTRACK_V(coop);
#undef UAVTRACK_V
    }
    constexpr bool GREEDY = POLICY == kPolicyGreedy;
    constexpr bool ACTOR = POLICY == kPolicyActor;
    constexpr bool GIVEN = POLICY == kPolicyGiven;
    extern __shared__ float4 smem4[];
    if (ALLOUT) __builtin_assume(p.obs != nullptr && p.reward != nullptr && p.terms != nullptr && p.covered != nullptr && p.done != nullptr);
    if (POLICY == kPolicyGiven) __builtin_assume(p.actions != nullptr);
    const int N = N_ > 0 ? N_ : p.N;
    const int M = M_ > 0 ? M_ : p.M;
    const int E = p.E;
    const int EN = E * N;
    const int CW = cov_words(M);
    const int MP = pairs_of(M);
    const float abias = act_bias_shape(N_) ? p.act_bias : 0.0f;     // what the table's action slot holds beside the action
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    using NbMask = typename NbMaskOf<N_>::type;      // (specialised shapes up to 32 UAVs: one word)
    constexpr int kNbBits = 8 * (int)sizeof(NbMask);

    // ---- LDS carve (float4 first: the dynamic base is 16-B aligned)
    const int ustride = ustride_of(N), tstride = tstride_of(M);
    float4 *utab = smem4;                        // [E][pairs][2 copies][3]
    float4 *ttab = utab + E * ustride;           // [E][pairs][2]
    // (multi-wave groups) staging of a wavefront's observation rows, 64 lanes x 3 float4 each: see kObsStage
    float4 *ostg = ttab + E * tstride;
    float *fb = reinterpret_cast<float *>(ostg + obs_stage_f4(E * N));
    float *thd = fb;   fb += E * M;              // [E][M]      target heading
    float *rawl = fb;  fb += E * (N + 1);        // [E][N + 1]  raw reward (cooperative modes), pair-padded
    float *tzf = fb;   if (Z3) fb += E * MP * 2; // [E][pairs] (z0, z1)
    int *ncnt = reinterpret_cast<int *>(fb);  fb += E * M;   // [E][M] UAVs within dc of a target (greedy policy)
    unsigned *covw = reinterpret_cast<unsigned *>(fb);   // [2][E * CW] (+ 2 words, MAAC-R pair emission; + E words, automatic reset)
    unsigned *rstw = covw + 2 * E * CW + 2;              // [E] 0, or 1 + the episode number an environment is being reset to
    float *climb_l = reinterpret_cast<float *>(rstw + E); // (3-D) [2][UAVTRACK_MAX_CLIMB] cos / sin of the climb angles
    // symmetric duplicate term (MAAC reward mode, sym_dup): [E][x | y | (z) | 2 N accumulators]
    // (MAAC and MAAC-R: the raw reward is needed only behind the second barrier -- in the output, in the neighbour record;
    //  MAAC-G reads its neighbours' raw rewards there and would need a third barrier)
    constexpr bool kSym = MODE != UAVTRACK_REWARD_MEAN;
    float *symbase = climb_l + (Z3 ? 2 * UAVTRACK_MAX_CLIMB : 0);
    const int symlen = sym_pose_len(N), symstride = sym_words(N, Z3);
    // MAAC-R, single-wavefront variants: the neighbour relation (d <= dp on post-move poses) is symmetric and the sequential
    // view of the peers j < i IS their post-move pose, so each lane tests only those (on the rows the observation sweep
    // reads anyway) and hands the bit to the partner through one word per UAV: [E][N]
    constexpr bool kNbSeq = LONE && MODE == UAVTRACK_REWARD_PMI && N_ > 0 && N_ <= 24 && !Z3;
    unsigned *nbt = reinterpret_cast<unsigned *>(symbase + (kSym ? E * symstride : 0));

    const int grp = xcd_group(blockIdx.x, gridDim.x);
    const int env0 = grp * E;
    const int envs_here = min(E, p.B - env0);
    const int e = tid / N;
    const int i = tid - e * N;
    const int b = env0 + e;
    const bool active = (tid < EN) && (e < envs_here);
    const size_t g = (size_t)b * N + i;          // flat (env, uav)
    const size_t BN = (size_t)p.B * N;
    float4 *const uenv = utab + e * ustride;     // this lane's environment
    float4 *const tenv = ttab + e * tstride;

    float x = 0, y = 0, z = 0, h = 0, c = 1, s = 0;
    int a_prev = 0, count = 0;
    float er = 0, ett = 0, ebp = 0, edup = 0;    // episode accumulators (train.py:181-192)
    int ecov = 0;
    int pn = 0;                                   // which table copy is "post-move" this step
    unsigned cov_pending = 0;                     // coverage word of the previous step (deferred read-back)
    int epi = 0;                                  // episode number of this environment's last reset

    // ---- load state once
    const StateBlock S = state_view(p.slab, p.B, N, M, Z3);
    if (Z3 && tid < 2 * UAVTRACK_MAX_CLIMB) climb_l[tid] = S.climb_c[tid];      // (climb_s follows climb_c in the slab)
    if (active) {
        x = S.ux[g]; y = S.uy[g]; h = S.uh[g]; a_prev = S.ua[g];
        if (Z3) z = S.uz[g];
        sincos_any(h, &s, &c);
        count = S.step_count[b];
        if (EXTRAS) epi = S.episode[b];
        uav_store(uenv + 3, i, x, y, c, s, (float)a_prev + abias, z);    // "previous" copy (1) for step 0
        if ((N & 1) && i == N - 1) {                                      // padding agent of the last pair
            uav_store(uenv, N, kFar, kFar, 0.f, 0.f, 0.f, 0.f);
            uav_store(uenv + 3, N, kFar, kFar, 0.f, 0.f, 0.f, 0.f);
        }
        if (i == 0) {
            rawl[e * (N + 1) + N] = 0.0f;
            for (int w = 0; w < CW; ++w) covw[e * CW + w] = covw[E * CW + e * CW + w] = 0;
            if (M & 1) {                                                  // padding target
                float *f = reinterpret_cast<float *>(tenv + (M >> 1) * 2);
                f[1] = kFar; f[3] = kFar; f[5] = 0.f; f[7] = 0.f;
                if (Z3) tzf[e * MP * 2 + M] = 0.f;
            }
        }
    }
    for (int q = tid; q < envs_here * M; q += nthreads) {
        const int te = q / M, k = q - te * M;
        const size_t gt = (size_t)env0 * M + q;
        const float th = S.th[gt];
        float ts, tc;
        sincos_any(th, &ts, &tc);
        float *f = reinterpret_cast<float *>(ttab + te * tstride + (k >> 1) * 2);
        const int bb = k & 1;
        f[bb] = S.tx[gt]; f[2 + bb] = S.ty[gt]; f[4 + bb] = tc; f[6 + bb] = ts;
        thd[q] = th;
        if (Z3) tzf[te * MP * 2 + k] = S.tz[gt];
    }
    // Per-step I/O addressing: every array is [t][...] with a uniform per-step stride, so the step loop
    // advances uniform base pointers (SGPRs) and each lane adds one 32-bit offset (validate() bounds
    // B * N * 48 below 4 GiB) -- `global_store ... v_off, s[base]` instead of 64-bit vector address math.
    const unsigned g32 = (unsigned)g;
    // (kObsStage) byte offset of this lane's float4 in the first of its wavefront's three 1 KB runs of an observation row block
    constexpr bool kObsStage = !LONE;
    constexpr bool kObsEarly = kObsStage && POLICY != kPolicyActor;
    const unsigned ostg_off = (((unsigned)env0 * (unsigned)N + (unsigned)(tid & ~63)) * 3u + (unsigned)(tid & 63)) * 16u;
    unsigned lane_off4 = g32 * 4u, lane_off48 = g32 * (unsigned)(UAVTRACK_OBS_DIM * 4);     // byte offsets of this lane in a [b][i] row of floats / of observations
    size_t row = 0;              // t * B * N: start of this step's [b][i] row in the per-step arrays (uniform)
    size_t rowb = 0;             // t * B
    int act_next = 0;
    int act = 0;
    if (GIVEN && active) act = min(max(*at(p.actions, g32 * 4u), 0), p.na_total - 1);
    ActorRng arng;
    arng.valid = false; arng.block = 0;
    float o[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // this UAV's local state; the actor reads last step's
    if (ACTOR && active) {
        const float4 *ip = reinterpret_cast<const float4 *>(p.obs_in + g * UAVTRACK_OBS_DIM);
        const float4 q0 = ip[0], q1 = ip[1], q2 = ip[2];
        o[0] = q0.x; o[1] = q0.y; o[2] = q0.z; o[3] = q0.w; o[4] = q1.x; o[5] = q1.y;
        o[6] = q1.z; o[7] = q1.w; o[8] = q2.x; o[9] = q2.y; o[10] = q2.z; o[11] = q2.w;
    }
    // Per-lane constants of the step loop: this lane's own slots in both table copies, and (when the
    // workgroup has at least one lane per target, as in every benchmark shape) its target's slot.
    float *const own0 = reinterpret_cast<float *>(uenv + (i >> 1) * 6) + (i & 1);
    // (register-resident targets cost five registers: at 50 x 25 that is the difference between three and four waves per
    // SIMD -- 16.6 vs 17.4 G agent-steps/s at 8192 envs in 3-D -- so the larger shapes keep the LDS read-modify-write)
    constexpr bool kRegTargets = LONE || (N_ > 0 && N_ <= 20);
    const bool one_target_per_lane = kRegTargets && E * M <= nthreads;
    const bool my_target = tid < envs_here * M;
    float *tgt = reinterpret_cast<float *>(ttab);
    float ttx = 0, tty = 0, ttc = 1, tts = 0, tth = 0;   // the lane's own target, resident across the T steps
    if (my_target) {
        const int te = tid / M, k = tid - te * M;
        tgt = reinterpret_cast<float *>(ttab + te * tstride + (k >> 1) * 2) + (k & 1);
    }
    __syncthreads();
    if (one_target_per_lane && my_target) { ttx = tgt[0]; tty = tgt[2]; ttc = tgt[4]; tts = tgt[6]; tth = thd[tid]; }

    // MAAC-R pair emission, one step behind (N <= 64, specialised shapes): a workgroup reserves its slots of the pair
    // list with ONE returning global atomic per step; waiting for it on the spot put a trip to L2 on every step's
    // critical path, so the reservation made at step t is consumed at step t + 1 (and behind the loop): these hold
    // what step t's lanes need to write their records and pairs then.
    // Single-wavefront workgroups (LONE) take their pair-list slots from a private POOL instead: four times the workgroups
    // would mean four times the reservations on the one counter (~10 ns each, serialised: 2.7 ms per 200 steps at 4096
    // envs).  A wavefront reserves a block worth a few steps with one atomic, hands the slots out itself, and asks for the
    // next block a step or two before the pool runs dry (the result is collected when it is needed); what a block has left
    // when a step does not fit, or at the end of the launch, is filled with dummy records the scorer skips.
    constexpr bool kPoolEmit = MODE == UAVTRACK_REWARD_PMI && N_ > 0 && N_ <= 64 && LONE;
    constexpr bool kPipeEmit = MODE == UAVTRACK_REWARD_PMI && N_ > 0 && N_ <= 64 && !LONE;
    constexpr unsigned kPoolBlock = 64;
    unsigned pool_base = 0, pool_left = 0, pend_size = 0, pend_base_v = 0, real_pairs = 0;   // (wave-uniform but for pend_base_v: lane 0's)
    bool pend = false, pend_got = false;      // a block has been asked for / its base has been collected into pend_base_s
    unsigned pend_base_s = 0;
    auto pool_dummies = [&](unsigned base, unsigned n) {
        for (unsigned k = (unsigned)tid; k < n; k += (unsigned)nthreads) out_store(p.pairs + base + k, 0xFFFFFFFFu, 0u);
    };
    unsigned pe_base = 0, pe_tg = 0;        // (thread 0) the reservation in flight; flat [t][b][i] of the pending step
    int pe_mine = 0, pe_slot = 0;
    NbMask pe_nball = 0, pe_later = 0;
    auto pmi_reward_slot = [&](bool has_neighbours, float raw_i) {
        return has_neighbours ? raw_i : fminf(fmaxf((1.0f - p.coop) * raw_i, -1.0f), 1.0f);
    };
    auto store_record = [&](unsigned tg, unsigned long long nball, unsigned first) {      // N <= 64 (internal.h, nbrec_words)
        if (N <= 32) out_store(reinterpret_cast<uint2 *>(p.nbrec + (size_t)tg * 2), (unsigned)nball, first);
        else {
            uint32_t *rec = p.nbrec + (size_t)tg * 3;
            rec[0] = (unsigned)nball; rec[1] = (unsigned)(nball >> 32); rec[2] = first;
        }
    };
    auto commit_pairs = [&](unsigned base) {
        const unsigned first = base + (unsigned)pe_slot;
        store_record(pe_tg, pe_nball, first);
        if (pe_mine) {
            uint2 *dst = p.pairs + first;
            NbMask later = pe_later;
            while (later) {                      // ascending j
                const int j = mffs(later) - 1;
                later &= later - 1;
                out_store(dst++, pe_tg, pe_tg - (unsigned)i + (unsigned)j);       // {flat index of i, flat index of j}
            }
        }
    };

    // An ISOLATED pair -- two UAVs that are each other's only neighbour -- needs no score: the softmax over a single
    // neighbour is exactly 1 whatever s_ij is (uav.py:287-288), so both rewards are (1 - a) raw + a raw_partner and the mix
    // kernel never reads the slot.  Such pairs are not emitted (a third of all pairs in the reference's 2000 m box).  `nball`:
    // this UAV's neighbours, `later`: those with a higher index; the neighbours' set sizes were left in LDS ahead of the barrier.
    auto drop_isolated = [&](auto nball, decltype(nball) later) {
        if (mpop(nball) == 1 && later != 0 &&
            reinterpret_cast<const int *>(rawl)[e * (N + 1) + (mffs(later) - 1)] == 1)
            later = 0;
        return later;
    };

    // The single-wavefront plain rollout runs even and odd steps as two copies of the loop body: the table copy that is
    // "post-move" alternates with the step, so per copy every per-pair row address is a per-lane constant.
    // (an inner loop of constant trip count that is unrolled in full, not `#pragma unroll 2` on the step loop: a loop with
    //  convergent operations in it is not unrolled when that needs a remainder loop)
    constexpr int kStepsPerIter = (LONE && MODE == UAVTRACK_REWARD_RAW && POLICY == kPolicyGiven && !EXTRAS) ? 2 : 1;
    constexpr int kSelPairs = (kStepsPerIter == 2 && N_ > 0) ? (N_ + 1) / 2 : 1;
    const float4 *selA[kSelPairs], *selB[kSelPairs];     // (kStepsPerIter == 2) row bases of the sequential view, post-move copy 0 / 1
    if (kStepsPerIter == 2) {
#pragma unroll
        for (int jq = 0; jq < kSelPairs; ++jq) {
            selA[jq] = (2 * jq < i) ? uenv : uenv + 3;
            selB[jq] = (2 * jq < i) ? uenv + 3 : uenv;
        }
    }
    // sequential view as a bit word: bit jp set = this lane sees pair jp BEFORE its move (2 jp >= i); the copy that holds it
    // is pn ^ bit (sweep_fast, copyw)
    const unsigned oldw = (i + 1) / 2 < 32 ? ~0u << ((i + 1) / 2) : 0u;
    for (int t0 = 0; t0 < p.T; t0 += kStepsPerIter) {
#pragma unroll
    for (int tpar = 0; tpar < kStepsPerIter; ++tpar) {
        const int t = t0 + tpar;
        if (kStepsPerIter > 1 && t >= p.T) break;
        const unsigned tg_off = (unsigned)t * (unsigned)BN + g32;   // flat [t][b][i] (MAAC-R pair records; < 2^32 by ensure_pmi_scratch)
        const int cbuf = (t & 1) * E * CW;

        // ---- P0 (fused greedy rollout only): the C-METHOD baseline policy (uav.py:324-369) on the state as it
        //      stands before this step -- UAV poses are the previous step's post-move copy, targets not yet moved
        if (GREEDY) {
            const int pc = pn ^ 1;
            auto upos_of = [&](const float4 *rows, int j) {
                const float *f = reinterpret_cast<const float *>(rows + (j >> 1) * 6);
                return make_float2(f[j & 1], f[2 + (j & 1)]);
            };
            auto tpos_of = [&](const float4 *rows, int k) {
                const float *f = reinterpret_cast<const float *>(rows + (k >> 1) * 2);
                return make_float2(f[k & 1], f[2 + (k & 1)]);
            };
            for (int q = tid; q < envs_here * M; q += nthreads) {
                const int te = q / M, k = q - te * M;
                const float4 *rows = utab + te * ustride + pc * 3;
                ncnt[q] = greedy_near_count(tpos_of(ttab + te * tstride, k), N, p.dc2,
                                            [&](int j) { return upos_of(rows, j); });
            }
            __syncthreads();
            if (active) {
                const float4 *rows = uenv + pc * 3;
                act = greedy_pick(x, y, h, i, N, M, p.na, p.dc2, p.turn_unit, (uint64_t)(p.env_offset + b),
                                  (uint32_t)count, p.greedy_k0, p.greedy_k1,
                                  [&](int j) { return upos_of(rows, j); },
                                  [&](int k) { return tpos_of(tenv, k); },
                                  [&](int k) { return ncnt[e * M + k]; });
                if (p.actions_out) out_store(at(p.actions_out + row, g32 * 4u), act);
            }
            __syncthreads();   // the policy has read the target table; now it may move
        }

        // ---- P0 (fused actor rollout only): take_action (actor_critic.py:138-148) on the UAV's own previous
        //      observation, still in registers -- no table, no barrier
        if (ACTOR) {     // whole wavefronts: the two layers run on the matrix cores (actor.h)
            // (the single-wavefront variant -- the closed loop at the reference shape -- carries the two-tiles-per-trip layout of
            //  the tile loop, for the reference's hidden width ONLY: 128 = kLoneActorTiles tiles; launch_rollout sends other
            //  widths to the general variants -- both layouts in one kernel spilled)
            if constexpr (LONE)
                act = actor_pick<false, actor_tiles(Z3), kLoneActorTiles>(o, p.actor_w, kLoneActorTiles, p.na_total, (uint64_t)(p.env_offset + b),
                                        (uint32_t)count, i, p.greedy_k0, p.greedy_k1, p.actor_mode, nullptr, arng);
            else
                act = actor_pick<false, actor_tiles(Z3)>(o, p.actor_w, p.actor_hblocks, p.na_total, (uint64_t)(p.env_offset + b),
                                        (uint32_t)count, i, p.greedy_k0, p.greedy_k1, p.actor_mode, nullptr, arng);
            if (active && p.actions_out) out_store(at(p.actions_out + row, g32 * 4u), act);
        }

        // ---- P1a: targets (target.py:27-60); straight flight, mirror at the walls
        auto advance_target = [&](float *f, int q) {      // f -> x slot of the target; y, cos, sin at +2, +4, +6
            float tx = f[0], ty = f[2];
            tx = fmaf(p.dtv_t, f[4], tx);
            ty = fmaf(p.dtv_t, f[6], ty);
            f[0] = tx; f[2] = ty;
            bool turned = false;
            float th = 0.0f;
            if (0.0f > ty || ty > p.y_max) {
                th = -thd[q];
                turned = true;
            } else if (tx < 0.0f || tx > p.x_max) {
                th = thd[q];
                th = (th > 0.0f) ? kPi - th : -kPi - th;
                turned = true;
            }
            if (turned) {   // rare: recompute so that T fused steps == T single steps bit for bit
                float ns, nc;
                sincos_any(th, &ns, &nc);
                f[4] = nc; f[6] = ns;
                thd[q] = th;
            }
        };
        if (one_target_per_lane) {
            // the owner lane holds (x, y, cos, sin, heading) in registers: no LDS read on the step's critical path
            if (my_target) {
                ttx = fmaf(p.dtv_t, ttc, ttx);
                tty = fmaf(p.dtv_t, tts, tty);
                tgt[0] = ttx; tgt[2] = tty;
                bool turned = false;
                float th = 0.0f;
                if (0.0f > tty || tty > p.y_max) {
                    th = -tth;
                    turned = true;
                } else if (ttx < 0.0f || ttx > p.x_max) {
                    th = (tth > 0.0f) ? kPi - tth : -kPi - tth;
                    turned = true;
                }
                if (turned) {   // rare: recompute so that T fused steps == T single steps bit for bit
                    sincos_any(th, &tts, &ttc);
                    tgt[4] = ttc; tgt[6] = tts;
                    tth = th;
                }
                if (EXTRAS && p.tpos) p.tpos[(rowb + env0) * M + tid] = make_float2(ttx, tty);   // optional trace, environment.py:150-153
            }
        } else {
            for (int q = tid; q < envs_here * M; q += nthreads) {
                const int te = q / M, k = q - te * M;
                float *f = reinterpret_cast<float *>(ttab + te * tstride + (k >> 1) * 2) + (k & 1);
                advance_target(f, q);
                if (EXTRAS && p.tpos) p.tpos[(rowb + env0) * M + q] = make_float2(f[0], f[2]);
            }
        }

        // ---- P1b: own kinematics (uav.py:83-99); position uses the OLD heading
        int a_now = 0;
        float ai = 0;
        const float xo = x, yo = y, zo = z, co = c, so = s, ao = (float)a_prev;   // pre-move pose: even-i self term
        if (active) {
            a_now = GIVEN ? act : min(max(act, 0), p.na_total - 1);   // (given actions are clamped where they are loaded)
            int a_turn = a_now, a_climb = 0;
            // (a / na through the float reciprocal: exact for these small integers, no integer-division sequence)
            if (Z3) { a_climb = (int)(((float)a_now + 0.5f) * p.inv_na); a_turn = a_now - a_climb * p.na; }
            float step_xy = p.dtv_u;
            if (Z3) {   // the climb-angle table sits in LDS (copied at launch start): no trip to L2 on every step
                step_xy = p.dtv_u * climb_l[a_climb];
                z = fmaf(p.dtv_u, climb_l[UAVTRACK_MAX_CLIMB + a_climb], z);
            }
            x = fmaf(step_xy, c, x);
            y = fmaf(step_xy, s, y);
            ai = (float)a_now;
            // turn rate (2 a + 1 - na) * unit (uav.py:81): the small integer built in floats (exact), one conversion serves both
            h = wrap_heading(fmaf(fmaf(Z3 ? (float)a_turn : ai, 2.0f, (float)(1 - p.na)), p.turn_unit, h));
            sincos_wrapped(h, &s, &c);
            {
                float *f = own0 + pn * 12;                   // copy pn of this lane's pair row
                f[0] = x; f[2] = y; f[4] = c; f[6] = s; f[8] = ai + abias; f[10] = z;
            }
            if (GIVEN && t + 1 < p.T) act_next = *at(p.actions + row + BN, g32 * 4u);   // prefetch next step's action
            if (i == 0)
                for (int w = 0; w < CW; ++w) covw[cbuf + e * CW + w] = 0;
            if (kSym) {      // post-move pose into the plain arrays (entry i + N repeats entry i), accumulators cleared
                float *sp = symbase + e * symstride;
                unsigned *dq = reinterpret_cast<unsigned *>(sp + (Z3 ? 3 : 2) * symlen);
                sp[i] = x; sp[symlen + i] = y;
                if (Z3) sp[2 * symlen + i] = z;
                if (2 * i <= N) {
                    sp[N + i] = x; sp[symlen + N + i] = y;
                    if (Z3) sp[2 * symlen + N + i] = z;
                }
                dq[i] = 0; dq[N + i] = 0;
            }
            if (kNbSeq) nbt[e * N + i] = 0;
        }
        UAVTRACK_STEP_BARRIER();

        // ---- P2: pair sweeps
        float tt = 0, bp = 0, dupn = 0, raw = 0;
        const float4 *rowNew = uenv + pn * 3;
        const float4 *rowOld = uenv + (pn ^ 1) * 3;
        const v2f *tzrow = reinterpret_cast<const v2f *>(tzf + e * MP * 2);
        // cooperative modes with N <= 64: the neighbour set (d <= dp on post-move poses) falls out of the peer
        // sweep as one 64-bit mask
        constexpr bool kMask = (MODE != UAVTRACK_REWARD_RAW) && N_ > 0 && N_ <= 64;
        unsigned long long nbmask = 0;
        unsigned sym_own = 0;                       // (kSym) this lane's own half of the duplicate-term pairs, fixed point
        // LDS reads of the sweeps run this many peer rows ahead of their use: 2 in the single-wavefront MAAC and MAAC-R rollouts
        // (a lone wave has nothing else to cover an LDS round trip with; MAAC-R since the end of round 4, when the kernel had
        // shed enough registers: 256 without scratch, 0.631 -> 0.607 ms per 200 steps).  The MAAC-G and in-kernel-policy forms
        // have more live state (raw rewards of the neighbours, the actor's fragments) and spill with any prefetch
        // (MAAC-G: 0.97 ms per 200 steps with it, 0.54 without)
        constexpr int kPrefetchRows = (LONE && MODE != UAVTRACK_REWARD_MEAN && POLICY == kPolicyGiven) ? 2 : 0;
        if (active) {
            Acc acc;
            // weight of uav.py:165 can be < 1 only near the origin (|rel| <= 1, so |abs| < 2 is necessary).  The branch
            // is per lane, not per wavefront: a lane's result must depend on its own environment only -- with a
            // wave-uniform branch the (differently rounded) literal path would also be taken by whichever other
            // environments happen to share the wavefront, and a shard of a batch would no longer reproduce the
            // unsharded batch bit for bit.  Wavefronts that hold such a UAV (rare) run both paths.
            const bool near0 = fmaxf(fabsf(x), fabsf(y)) < 2.5f;
            if (__builtin_expect(near0, 0)) {
                sweep_weighted<Z3>(p, N, M, i, rowNew, rowOld, tenv, tzrow, covw, cbuf + e * CW,
                                   x, y, z, c, s, ai, abias, acc);
                if (kMask)
                    for (int j = 0; j < N; ++j) {
                        const UavRow nw = uav_elem(rowNew, j);
                        float d2 = dist2(nw.x - x, nw.y - y);
                        if (Z3) { const float dz = nw.z - z; d2 = fmaf(dz, dz, d2); }
                        nbmask |= (unsigned long long)(d2 <= p.dp2 ? 1u : 0u) << j;
                    }
            } else {
                sweep_fast<N_, M_, Z3, kMask, kPrefetchRows, kSym, kVConst, kNbSeq>(p, N, M, i, rowNew, rowOld, tenv, tzrow, covw, cbuf + e * CW,
                                              x, y, z, c, s, ai, xo, yo, zo, co, so, ao, acc, nbmask,
                                              kStepsPerIter == 2 ? (tpar == 0 ? selA : selB) : nullptr,   // (pn == tpar: t0 is even)
                                              uenv, pn ? ~oldw : oldw);
            }
            if (kNbSeq) {    // lower half kept, each of its bits handed to the partner; the upper half is what the partners hand in
                const unsigned lower = (unsigned)nbmask & ((1u << i) - 1u);
                for (unsigned m = lower; m; m &= m - 1u) atomicOr(&nbt[e * N + __builtin_ctz(m)], 1u << i);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                nbmask = lower | *reinterpret_cast<volatile unsigned *>(&nbt[e * N + i]);
            }
            if (kSym) {      // every active lane, whichever sweep it took: its partners count on its half of the pairs
                const float *sp = symbase + e * symstride;
                unsigned *dq = const_cast<unsigned *>(reinterpret_cast<const unsigned *>(sp + (Z3 ? 3 : 2) * symlen));
                sym_own = sym_dup<N_, Z3, kVConst>(p, N, i, sp, sp + symlen, sp + 2 * symlen, dq, x, y, z);
            }

            // ---- P3: local state (uav.py:156-190)
            {   // empty list -> -1 (uav.py:174,186), branch-free and with two selects per list instead of one per entry: the
                // reciprocal of an empty list's count becomes 0 and the addend -1, so every entry is one FMA
                // (sum * scale) * rc + base -- the same products, in the same order, as the selected form
                const bool anyU = acc.cntU > 0.0f, anyT = acc.cntT > 0.0f;
                const float rcU = anyU ? fast_rcp(acc.cntU) : 0.0f, rcT = anyT ? fast_rcp(acc.cntT) : 0.0f;
                const float baseU = anyU ? 0.0f : -1.0f, baseT = anyT ? 0.0f : -1.0f;
                o[0] = fmaf(acc.sxU * p.inv_dc, rcU, baseU);
                o[1] = fmaf(acc.syU * p.inv_dc, rcU, baseU);
                o[2] = fmaf(acc.scU, rcU, baseU);
                o[3] = fmaf(acc.ssU, rcU, baseU);
                o[4] = fmaf(acc.saU * p.inv_na_total, rcU, baseU);
                o[5] = fmaf(acc.sxT * p.inv_dp, rcT, baseT);
                o[6] = fmaf(acc.syT * p.inv_dp, rcT, baseT);
                o[7] = fmaf(acc.scT, rcT, baseT);
                o[8] = fmaf(acc.ssT, rcT, baseT);
            }
            o[9] = x * p.inv_dc;
            o[10] = y * p.inv_dc;
            o[11] = ai * p.inv_na_total;
            // (kObsStage) the row goes into the wavefront's staging block as soon as it exists: twelve registers fewer are
            // live across the barrier and the reward phase (the in-kernel actor reads o[] at the next step and keeps them)
            if (kObsEarly && p.obs) {
                float4 *sp = ostg + (tid >> 6) * 192 + (tid & 63) * 3;
                sp[0] = make_float4(o[0], o[1], o[2], o[3]);
                sp[1] = make_float4(o[4], o[5], o[6], o[7]);
                sp[2] = make_float4(o[8], o[9], o[10], o[11]);
            }

            // ---- raw reward terms, clipped and normalised (environment.py:207-220)
            float d_bdr = fminf(fminf(x, p.x_max - x), fminf(y, p.y_max - y));
            if (Z3) d_bdr = fminf(d_bdr, fminf(z, p.z_max - z));
            // boundary punishment (uav.py:231-260) and its normalisation (environment.py:209,216), folded:
            //   d < 0: -0.5 -> -1;  0 <= d < dp: -0.5 (dp - d) / dp -> d / dp - 1;  d >= dp: 0 -> 0   ==  clamp(d / dp, 0, 1) - 1
            bp = __builtin_amdgcn_fmed3f(d_bdr * p.inv_dp, 0.0f, 1.0f) - 1.0f;
            tt = __builtin_amdgcn_fmed3f(acc.trk, 0.0f, p.tt_ceil) * p.inv_tt_ceil;
            if (!kSym) dupn = __builtin_amdgcn_fmed3f(acc.dup * p.dup_k, -1.0f, 0.0f);
            if (!kSym) raw = fmaf(p.gamma, dupn, fmaf(p.beta, bp, p.alpha * tt));      // (kSym: behind the barrier, once the partners' halves are in)
            if (MODE == UAVTRACK_REWARD_MEAN) rawl[e * (N + 1) + i] = raw;
            // MAAC-R: the size of this UAV's neighbour set, for its neighbours to see behind the barrier (isolated pairs, below)
            if (MODE == UAVTRACK_REWARD_PMI && kMask) reinterpret_cast<int *>(rawl)[e * (N + 1) + i] = mpop((NbMask)((NbMask)nbmask & ~((NbMask)1 << i)));
        }
        if (kPipeEmit && tid == 0) {
            // MAAC-R: the reservation made a step ago is taken HERE, ahead of this step's output stores -- its wait
            // drains the vector-memory counter, and behind the stores that would be a wait for all of them
            unsigned *wg_cnt = covw + 2 * E * CW;
            wg_cnt[0] = 0;
            if (t > 0) wg_cnt[1] = pe_base;
        }
        UAVTRACK_STEP_BARRIER();

        // (MAAC-R slot pool) a block asked for at an earlier step is collected HERE, ahead of this step's output stores: the
        // wait for the returning atomic drains the vector-memory counter, and behind the stores it would wait for all of them
        if (kPoolEmit && pend && !pend_got) {
            pend_base_s = (unsigned)__builtin_amdgcn_readfirstlane((int)pend_base_v);
            pend_got = true;
        }

        // ---- P4: cooperative reward, coverage, outputs
        if (active) {
            if (kSym) {      // duplicate term: own half + what the partners added to this UAV's two accumulator entries
                const unsigned *dq = reinterpret_cast<const unsigned *>(symbase + e * symstride + (Z3 ? 3 : 2) * symlen);
                // (every UAV's three sums hold N - 1 terms between them: its own half and one from each partner)
                const float dsum = (float)(sym_own + dq[i] + dq[N + i] - (unsigned)(N - 1) * 0x4B400000u);      // sum g * 2^kSymBits
                dupn = __builtin_amdgcn_fmed3f(dsum * p.sym_dup_k, -1.0f, 0.0f);
                raw = fmaf(p.gamma, dupn, fmaf(p.beta, bp, p.alpha * tt));
            }
            float r = raw;
            if (MODE == UAVTRACK_REWARD_MEAN) {
                if (p.coop != 0.0f && kMask) {   // uav.py:293-310: walk the set bits of the neighbour mask, ascending j
                    NbMask nb = (NbMask)nbmask & ~((NbMask)1 << i);
                    float sum = 0.0f;
                    const int ncnt_i = mpop(nb);
                    while (nb) {
                        const int j = mffs(nb) - 1;
                        nb &= nb - 1;
                        sum += rawl[e * (N + 1) + j];
                    }
                    r = ncnt_i ? fmaf(p.coop * sum, fast_rcp((float)ncnt_i), (1.0f - p.coop) * raw) : 0.0f;      // (v_rcp_f32 for the mean's divide)
                } else if (p.coop != 0.0f) {   // any N: neighbours re-derived from the post-move poses, self subtracted after
                    v2f sum = splat(0.f), cnt = splat(0.f);
                    const v2f xi2 = splat(x), yi2 = splat(y), zi2 = splat(z);
                    const int NP = pairs_of(N_ > 0 ? N_ : N);
#pragma unroll UAVTRACK_UNROLL_U
                    for (int jp = 0; jp < NP; ++jp) {
                        const float4 n0 = rowNew[jp * 6];
                        const v2f dx = (v2f){n0.x, n0.y} - xi2, dy = (v2f){n0.z, n0.w} - yi2;
                        v2f d2 = pk_fma(dy, dy, dx * dx);
                        if (Z3) {
                            const float4 n2 = rowNew[jp * 6 + 2];
                            const v2f dz = (v2f){n2.z, n2.w} - zi2;
                            d2 = pk_fma(dz, dz, d2);
                        }
                        const v2f mm = {d2.x <= p.dp2 ? 1.0f : 0.0f, d2.y <= p.dp2 ? 1.0f : 0.0f};
                        const float *rw = rawl + e * (N + 1) + 2 * jp;    // (N + 1 may be odd: no 8-B alignment)
                        sum = pk_fma(mm, (v2f){rw[0], rw[1]}, sum);
                        cnt += mm;
                    }
                    const float nsum = sum.x + sum.y - raw, ncnt = cnt.x + cnt.y - 1.0f;   // minus self (d = 0)
                    r = (ncnt > 0.0f) ? fmaf(p.coop * nsum, fast_rcp(ncnt), (1.0f - p.coop) * raw) : 0.0f;
                }
            }
            r = fminf(fmaxf(r, -1.0f), 1.0f);   // clip_and_normalize(reward, -1, 1), environment.py:225
            // MAAC-R: the reward slot carries what the mix stage needs of this UAV -- its RAW reward while it has neighbours
            // (they read it, and the mix overwrites it with the final value), its FINAL reward (1 - a) raw, clipped
            // (uav.py:290, environment.py:225), when it has none: the mix then neither recomputes nor rewrites it.
            // (unspecialised shapes: the neighbour set is only known in the emission block below, which stores it again)
            if (MODE == UAVTRACK_REWARD_PMI) r = pmi_reward_slot(kMask ? (NbMask)((NbMask)nbmask & ~((NbMask)1 << i)) != 0 : true, raw);

            ++count;
            // the prefetched action is consumed HERE, ahead of this step's stores: vector-memory operations
            // retire in order, so a first use at the top of the next step would wait for these stores too
            if (GIVEN) act = min(max(act_next, 0), p.na_total - 1);
            if (i == 0) {
                // Coverage count.  With one coverage word per environment (M <= 24) the LDS read-back is
                // deferred by a step: this step's word is requested now and popcounted at the next step's
                // P4 (or behind the loop), so its latency never sits on the critical path.
                if (CW == 1) {
                    if (t > 0) {
                        const int cov = __popc(cov_pending);
                        ecov += cov;
                        if (p.covered) out_store(p.covered + (rowb - p.B + b), cov);          // row t - 1
                    }
                    cov_pending = covw[cbuf + e];
                } else {
                    int cov = 0;
                    for (int w = 0; w < CW; ++w) cov += __popc(covw[cbuf + e * CW + w]);
                    ecov += cov;
                    if (p.covered) out_store(p.covered + (rowb + b), cov);
                }
                if (p.done) out_store(p.done + (rowb + b), (uint8_t)((p.horizon > 0 && count >= p.horizon) ? 1 : 0));
            }
            er += r; ett += tt; ebp += bp; edup += dupn;

            // (the per-lane byte offsets pass through an empty asm every step: left loop-invariant, the compiler adds it to each
            //  output's base ONCE, outside the loop, and then advances 64-bit vector addresses by the uniform row stride with two
            //  v_mad_u64_u32 and a v_lshl_add_u64 per array and step -- this way the uniform row base stays in scalar registers
            //  and the stores take the `saddr + 32-bit voffset` form)
            asm volatile("" : "+v"(lane_off4), "+v"(lane_off48));
            if (kObsEarly) {      // (staged in P3)
            } else if (kObsStage) {
                if (p.obs) {      // this lane's row into its wavefront's staging block; written out as whole lines behind this block
                    float4 *sp = ostg + (tid >> 6) * 192 + (tid & 63) * 3;
                    sp[0] = make_float4(o[0], o[1], o[2], o[3]);
                    sp[1] = make_float4(o[4], o[5], o[6], o[7]);
                    sp[2] = make_float4(o[8], o[9], o[10], o[11]);
                }
            } else if (p.obs) {
                float4 *op = at(reinterpret_cast<float4 *>(p.obs + row * UAVTRACK_OBS_DIM), lane_off48);
                op[0] = make_float4(o[0], o[1], o[2], o[3]);       // (ordinary stores: see out_store)
                op[1] = make_float4(o[4], o[5], o[6], o[7]);
                op[2] = make_float4(o[8], o[9], o[10], o[11]);
            }
            if (p.reward) out_store(at(p.reward + row, lane_off4), r);
            // optional: uav.raw_reward (environment.py:219), the weighted sum of the three terms before any sharing
            if (EXTRAS && p.raw) out_store(at(p.raw + row, lane_off4), raw);
            if (p.terms) {                                    // [t][3][b][i]
                float *tp = p.terms + 3 * row;
                out_store(at(tp, lane_off4), tt);
                out_store(at(tp + BN, lane_off4), bp);
                out_store(at(tp + 2 * BN, lane_off4), dupn);
            }
            a_prev = a_now;
        }
        if (kObsStage && p.obs) {
            // A lane's 48-byte row as three 16-byte stores at a 48-byte stride touches 48 lines per instruction, each in part:
            // three times the requests the L2 needs for the same bytes, and at a chip-filling batch the observation stores
            // cost 23 % of the launch (knock-out: 5.5 -> 4.24 ms at 65 536 envs; 12 % at 8192 x 50 x 25 in 3-D).  The rows of
            // a wavefront are contiguous in the output ([b][i] order = lane order), so they go through a 3 KB staging block:
            // written row-wise above (stride 12 words: conflict-free per 16 lanes), read back as consecutive float4 and
            // stored as three contiguous 1 KB runs.  Wavefront-private: no workgroup barrier.  Every lane of the wavefront
            // takes part (the last wavefront's idle lanes copy rows of its active ones).
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int w0 = tid & ~63;                                           // first lane of this wavefront
            const int nrow4 = 3 * max(0, min(64, min(EN, envs_here * N) - w0)); // float4 this wavefront's active lanes wrote
            const float4 *sp = ostg + (tid >> 6) * 192;
            unsigned oo = ostg_off;
            asm volatile("" : "+v"(oo));              // (as lane_off4 above: keeps the row base in scalar registers)
            float4 *op = at(reinterpret_cast<float4 *>(p.obs + row * UAVTRACK_OBS_DIM), oo);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k * 64 + (tid & 63) < nrow4) op[k * 64] = sp[k * 64 + (tid & 63)];
            // (the block is rewritten a step later, behind two workgroup barriers)
        }

        // ---- MAAC-R only: emit the neighbour pairs (i < j, d <= dp on post-move poses, uav.py:278) this
        //      workgroup owns into the global pair list the PMI scoring kernel consumes.  s_ij = s_ji
        //      (the input is la_i * la_j), so unordered pairs halve the work.
        if (kPoolEmit) {
            unsigned *wg_cnt = covw + 2 * E * CW;          // (a wavefront's LDS operations execute in program order)
            int mine = 0;
            NbMask nball = 0, later = 0;                    // all neighbours / neighbours j > i
            if (active) {
                nball = (NbMask)nbmask & ~((NbMask)1 << i);
                later = drop_isolated(nball, (NbMask)((i + 1 < kNbBits) ? (nball >> (i + 1)) << (i + 1) : 0));
                mine = mpop(later);
            }
            // every lane's run of slots: an inclusive prefix sum over the wavefront on the cross-lane (DPP) paths -- no trip to the
            // LDS (an LDS atomic per lane and the read-back of the total were three dependent round trips per step)
            const int incl = wave_inclusive_sum(mine);
            const int slot = incl - mine;
            const unsigned total = (unsigned)__builtin_amdgcn_readlane(incl, 63);
            unsigned first = pool_base + (unsigned)slot;
            if (total > pool_left) {
                // This step does not fit.  UAVs whose run of slots still fits keep the old block (they are a prefix in slot
                // order: runs are disjoint and consecutive), the others move to the next block as one piece; what the old
                // block has left behind the prefix -- less than one UAV's run -- goes to dummies.
                const bool fits = (unsigned)slot + (unsigned)mine <= pool_left;
                if (tid == 0) wg_cnt[1] = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (mine && !fits) atomicAdd(&wg_cnt[1], (unsigned)mine);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const unsigned moved = (unsigned)__builtin_amdgcn_readfirstlane((int)wg_cnt[1]), cut = total - moved;
                pool_dummies(pool_base + cut, pool_left - cut);
                if (pend && pend_size < moved) {            // (a step far above the recent ones: the block asked for is too small as well)
                    pool_dummies(pend_got ? pend_base_s : (unsigned)__builtin_amdgcn_readfirstlane((int)pend_base_v), pend_size);
                    pend = false;
                }
                if (!pend) {     // (the launch's last step takes exactly what it needs: nothing left over to fill with dummies)
                    pend_size = (moved > kPoolBlock || t + 1 == p.T) ? moved : kPoolBlock;
                    if (tid == 0) pend_base_v = atomicAdd(p.pair_count, pend_size);
                    pend_got = false;
                }
                // (as a rule the block was asked for a step or more ago and collected at the top of this step's P4; else wait here)
                const unsigned nbase = pend_got ? pend_base_s : (unsigned)__builtin_amdgcn_readfirstlane((int)pend_base_v);
                if (!fits) first = nbase + ((unsigned)slot - cut);
                pool_base = nbase + moved;
                pool_left = pend_size - moved;
                pend = false; pend_got = false;
            } else {
                pool_base += total;
                pool_left -= total;
            }
            real_pairs += total;
            if (active) {
                store_record(tg_off, nball, first);
                uint2 *dst = p.pairs + first;
                while (later) {                      // ascending j
                    const int j = mffs(later) - 1;
                    later &= later - 1;
                    out_store(dst++, tg_off, tg_off - (unsigned)i + (unsigned)j);
                }
            }
            if (!pend && pool_left < 2 * total && t + 2 < p.T) {   // about to run dry: ask for the next block now, collect it later
                pend_size = 4 * total > kPoolBlock ? 4 * total : kPoolBlock;
                if (tid == 0) pend_base_v = atomicAdd(p.pair_count, pend_size);
                pend = true; pend_got = false;
            }
        } else
        if (kPipeEmit) {
            unsigned *wg_cnt = covw + 2 * E * CW;          // two extra words behind the coverage masks
            // (thread 0 zeroed wg_cnt[0] and published the previous step's reservation in wg_cnt[1] ahead of the
            // P3 -> P4 barrier: one barrier of its own is all this block needs)
            if (t > 0 && active) commit_pairs(wg_cnt[1]);
            int mine = 0, slot = 0;
            NbMask nball = 0, later = 0;                    // all neighbours / neighbours j > i
            if (active) {
                nball = (NbMask)nbmask & ~((NbMask)1 << i);
                later = drop_isolated(nball, (NbMask)((i + 1 < kNbBits) ? (nball >> (i + 1)) << (i + 1) : 0));
                mine = mpop(later);
                if (mine) slot = (int)atomicAdd(&wg_cnt[0], (unsigned)mine);
            }
            __syncthreads();
            if (tid == 0) { pe_base = wg_cnt[0] ? atomicAdd(p.pair_count, wg_cnt[0]) : 0u; real_pairs += wg_cnt[0]; }
            pe_mine = mine; pe_slot = slot; pe_nball = nball; pe_later = later; pe_tg = tg_off;
        } else
        if (MODE == UAVTRACK_REWARD_PMI) {
            unsigned *wg_cnt = covw + 2 * E * CW;          // two extra words behind the coverage masks
            if (tid == 0) wg_cnt[0] = 0;
            // One record per agent-step for the mix kernel (internal.h, nbrec_words): the neighbour mask, where this
            // UAV's pairs start in the pair list (their scores land in the same slots), and the raw reward.
            const int W = nbrec_mask_words(N);
            int mine = 0, slot = 0;
            unsigned long long nball = 0, later = 0;        // all neighbours / neighbours j > i   (N <= 64)
            auto is_neighbour = [&](int j) {
                const UavRow nw = uav_elem(rowNew, j);
                float d2 = dist2(nw.x - x, nw.y - y);
                if (Z3) { const float dz = nw.z - z; d2 = fmaf(dz, dz, d2); }
                return j != i && d2 <= p.dp2;
            };
            if (active && N <= 64) {                        // (the post-move table is stable from the P2 barrier to the next step's P1)
                for (int j = 0; j < N; ++j) nball |= (unsigned long long)(is_neighbour(j) ? 1u : 0u) << j;
                reinterpret_cast<int *>(rawl)[e * (N + 1) + i] = __popcll(nball);      // for drop_isolated
            }
            __syncthreads();
            if (active) {
                if (N <= 64) {
                    later = drop_isolated(nball, (i + 1 < 64) ? (nball >> (i + 1)) << (i + 1) : 0ull);
                    mine = __popcll(later);
                } else {
                    for (int j = i + 1; j < N; ++j) mine += is_neighbour(j) ? 1 : 0;
                }
                if (mine) slot = (int)atomicAdd(&wg_cnt[0], (unsigned)mine);
            }
            __syncthreads();
            if (tid == 0) { wg_cnt[1] = wg_cnt[0] ? atomicAdd(p.pair_count, wg_cnt[0]) : 0u; real_pairs += wg_cnt[0]; }
            __syncthreads();
            if (active) {
                const unsigned first = wg_cnt[1] + (unsigned)slot;
                bool any = nball != 0ull;
                if (N <= 64) {
                    store_record(tg_off, nball, first);
                } else {
                    uint32_t *rec = p.nbrec + (size_t)tg_off * (W + 1);
                    for (int w = 0; w < W; ++w) {
                        unsigned bits = 0;
                        for (int j = 32 * w; j < min(N, 32 * w + 32); ++j) bits |= (is_neighbour(j) ? 1u : 0u) << (j - 32 * w);
                        rec[w] = bits;
                        any = any || bits != 0u;
                    }
                    rec[W] = first;
                }
                out_store(at(p.reward + row, g32 * 4u), pmi_reward_slot(any, raw));      // (the reward slot's MAAC-R content, see P4)
                if (mine) {
                    uint2 *dst = p.pairs + first;
                    if (N <= 64) {
                        while (later) {                      // ascending j
                            const int j = __ffsll((long long)later) - 1;
                            later &= later - 1;
                            *dst++ = make_uint2(tg_off, tg_off - (unsigned)i + (unsigned)j);
                        }
                    } else {
                        for (int j = i + 1; j < N; ++j)
                            if (is_neighbour(j)) *dst++ = make_uint2(tg_off, tg_off - (unsigned)i + (unsigned)j);
                    }
                }
            }
        }
        // ---- automatic episode turnover (uavtrack_step_many_autoreset): an environment whose done flag fired this step
        //      becomes uavtrack_reset(seed, episode + 1) -- reset_kernel.hip's formulas -- before the next one.  The barrier
        //      also keeps the table writes behind every read of this step.
        if (EXTRAS && p.auto_reset) {
            const bool fire = active && p.horizon > 0 && count >= p.horizon;
            if (active && i == 0) rstw[e] = fire ? (unsigned)(epi + 1) + 1u : 0u;
            if (__syncthreads_or(fire ? 1 : 0)) {
                if (fire) {
                    ++epi;
                    const uint64_t genv = (uint64_t)(p.env_offset + b);
                    const Philox4 r = philox4x32_10((uint32_t)genv, (uint32_t)epi, (uint32_t)i, 0x55415631u ^ (uint32_t)(genv >> 32),
                                                    p.reset_k0, p.reset_k1);
                    x = (float)((double)(i + 1) * p.x_max_d / (double)(N + 1));      // environment.py:105
                    y = (float)(p.y_max_d / 2.0);                                    // environment.py:107
                    if (Z3) z = (float)(p.z_max_d / 2.0);
                    h = fmaf(u01(r.v[0]), kTwoPi, -kPi);
                    a_prev = (int)(((uint64_t)r.v[1] * (uint32_t)p.na_total) >> 32);
                    sincos_any(h, &s, &c);
                    count = 0;
                    uav_store(uenv + pn * 3, i, x, y, c, s, (float)a_prev + abias, z);   // this step's copy: the next step's "previous" one
                    o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = o[6] = o[7] = o[8] = -1.0f;   // get_states() of a fresh state (uav.py:174,186)
                    o[9] = x * p.inv_dc; o[10] = y * p.inv_dc; o[11] = (float)a_prev * p.inv_na_total;
                }
                auto reset_target = [&](int q, float &tx, float &ty, float &th, float &tc, float &ts) {
                    const int te = q / M, k = q - te * M;
                    const unsigned ep1 = rstw[te];
                    if (!ep1) return false;
                    const uint64_t genv = (uint64_t)(p.env_offset + env0 + te);
                    const Philox4 r = philox4x32_10((uint32_t)genv, ep1 - 1u, (uint32_t)(N + k), 0x55415631u ^ (uint32_t)(genv >> 32),
                                                    p.reset_k0, p.reset_k1);
                    tx = u01(r.v[0]) * p.x_max;
                    ty = u01(r.v[1]) * p.y_max;
                    th = fmaf(u01(r.v[2]), kTwoPi, -kPi);
                    sincos_any(th, &ts, &tc);
                    float *f = reinterpret_cast<float *>(ttab + te * tstride + (k >> 1) * 2) + (k & 1);
                    f[0] = tx; f[2] = ty; f[4] = tc; f[6] = ts;
                    if (Z3) tzf[te * MP * 2 + k] = u01(r.v[3]) * p.z_max;
                    return true;
                };
                if (one_target_per_lane) {
                    if (my_target) reset_target(tid, ttx, tty, tth, ttc, tts);
                } else {
                    for (int q = tid; q < envs_here * M; q += nthreads) {
                        float tx, ty, th, tc, ts;
                        if (reset_target(q, tx, ty, th, tc, ts)) thd[q] = th;
                    }
                }
            }
        }
        pn ^= 1;
        row += BN;
        rowb += (size_t)p.B;
    }
    }
    if (kPoolEmit) {                         // what the pool and an uncollected block have left
        pool_dummies(pool_base, pool_left);
        if (pend) pool_dummies(pend_got ? pend_base_s : (unsigned)__builtin_amdgcn_readfirstlane((int)pend_base_v), pend_size);
    }
    if (MODE == UAVTRACK_REWARD_PMI && tid == 0 && real_pairs && p.pair_total)      // accounting (uavtrack_pmi_pairs_scored)
        atomicAdd(p.pair_total, (unsigned long long)real_pairs);
    if (kPipeEmit && p.T > 0) {              // the last step's records and pairs
        unsigned *wg_cnt = covw + 2 * E * CW;
        if (tid == 0) wg_cnt[1] = pe_base;
        __syncthreads();
        if (active) commit_pairs(wg_cnt[1]);
    }
    if (CW == 1 && active && i == 0) {       // the last step's deferred coverage count
        const int cov = __popc(cov_pending);
        ecov += cov;
        if (p.covered) out_store(p.covered + (rowb - p.B + b), cov);
    }

    // ---- store state once
    if (active) {
        S.ux[g] = x; S.uy[g] = y; S.uh[g] = h; S.ua[g] = a_prev;
        if (Z3) S.uz[g] = z;
        if (i == 0) {
            S.step_count[b] = count;
            if (EXTRAS) S.episode[b] = epi;
        }
    }
    // (uavtrack_step_host) the same state once more into the caller-visible copy of the slab: Environment.step's `position`
    // rows and the uav.x / target.x a host caller reads come out of the launch itself, no snapshot launch behind it
    if (EXTRAS && p.state_copy) {
        const StateBlock C2 = state_view(p.state_copy, p.B, N, M, Z3);
        if (active) {
            C2.ux[g] = x; C2.uy[g] = y; C2.uh[g] = h; C2.ua[g] = a_prev;
            if (Z3) C2.uz[g] = z;
            if (i == 0) { C2.step_count[b] = count; C2.episode[b] = epi; }
        }
        for (int q = tid; q < envs_here * M; q += nthreads) {
            const int te = q / M, k = q - te * M;
            const size_t gt = (size_t)env0 * M + q;
            const float *f = reinterpret_cast<const float *>(ttab + te * tstride + (k >> 1) * 2);
            // (register-resident targets keep their LDS slots current except for the heading, which lives in thd / tth)
            C2.tx[gt] = f[k & 1]; C2.ty[gt] = f[2 + (k & 1)];
            C2.th[gt] = (one_target_per_lane && q == tid) ? tth : thd[q];
            if (Z3) C2.tz[gt] = tzf[te * MP * 2 + k];
        }
    }
    if (EXTRAS && p.auto_reset && Z3)      // (target altitudes only ever change at a reset)
        for (int q = tid; q < envs_here * M; q += nthreads) {
            const int te = q / M, k = q - te * M;
            S.tz[(size_t)env0 * M + q] = tzf[te * MP * 2 + k];
        }
    if (one_target_per_lane) {
        if (my_target) {
            const size_t gt = (size_t)env0 * M + tid;
            S.tx[gt] = ttx; S.ty[gt] = tty; S.th[gt] = tth;
        }
    } else {
        for (int q = tid; q < envs_here * M; q += nthreads) {
            const int te = q / M, k = q - te * M;
            const size_t gt = (size_t)env0 * M + q;
            const float *f = reinterpret_cast<const float *>(ttab + te * tstride + (k >> 1) * 2);
            S.tx[gt] = f[k & 1]; S.ty[gt] = f[2 + (k & 1)]; S.th[gt] = thd[q];
        }
    }
    if (p.ep_sums) {
        __syncthreads();                       // everyone is done with the tables
        float4 *eps = utab;                    // E * ustride >= E * N float4 (see lds_bytes_for)
        if (active) eps[tid] = make_float4(er, ett, ebp, edup);
        __syncthreads();
        if (active && i == 0) {
            float4 sum = make_float4(0, 0, 0, 0);
            for (int j = 0; j < N; ++j) {       // fixed order: bitwise reproducible
                const float4 v = eps[e * N + j];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            const float inv_n = 1.0f / (float)N;
            float *ep = p.ep_sums + (size_t)b * 5;
            // (MAAC-R: the reward is final only behind the scorer; its episode sum comes from the mix stage, pmi_kernel.hip.
            // The three terms and the coverage do not depend on the scores and are summed here as in the other modes.)
            constexpr bool kReward = MODE != UAVTRACK_REWARD_PMI;
            if (p.ep_accumulate) {
                if (kReward) ep[0] += sum.x * inv_n;
                ep[1] += sum.y * inv_n; ep[2] += sum.z * inv_n; ep[3] += sum.w * inv_n;
                ep[4] += (float)ecov;
            } else {
                if (kReward) ep[0] = sum.x * inv_n;
                ep[1] = sum.y * inv_n; ep[2] = sum.z * inv_n; ep[3] = sum.w * inv_n;
                ep[4] = (float)ecov;
            }
        }
    }
}

// (the symmetric duplicate term's pose arrays and accumulators exist in the MAAC and MAAC-R reward modes: kSym)
size_t lds_bytes_for(int E, int N, int M, bool z3, int reward_mode)
{
    const size_t CW = cov_words(M), MP = pairs_of(M);
    const size_t f4 = (size_t)E * ustride_of(N) + (size_t)E * tstride_of(M) + obs_stage_f4(E * N);
    const size_t f = (size_t)E * M + (size_t)E * (N + 1) + (z3 ? (size_t)E * MP * 2 : 0) + (size_t)E * M + 2 * E * CW + 2 + E +
                     (z3 ? 2 * UAVTRACK_MAX_CLIMB : 0) + (reward_mode != UAVTRACK_REWARD_MEAN ? (size_t)E * sym_words(N, z3) : 0) +
                     (reward_mode == UAVTRACK_REWARD_PMI ? (size_t)E * N : 0);   // (kNbSeq)
    return (f4 * 16 + f * 4 + 15) & ~(size_t)15;   // ustride >= 3 N float4 per env, so ep_sums staging (E N float4) fits
}

using KernelFn = void (*)(const StepParams);

template <int N_, int M_, int POLICY, bool ALLOUT, bool EXTRAS, bool LONE = false>
KernelFn pick_reward(int mode, bool z3)
{
    constexpr bool PLANAR_ONLY = POLICY == kPolicyGreedy;      // the baseline policy is planar and runs with MAAC / MAAC-G
    if (z3 && !PLANAR_ONLY) {
        switch (mode) {
        case UAVTRACK_REWARD_MEAN: return rollout_kernel<N_, M_, UAVTRACK_REWARD_MEAN, !PLANAR_ONLY, POLICY, ALLOUT, EXTRAS, LONE>;
        case UAVTRACK_REWARD_PMI:  return rollout_kernel<N_, M_, PLANAR_ONLY ? UAVTRACK_REWARD_RAW : UAVTRACK_REWARD_PMI, !PLANAR_ONLY, POLICY, ALLOUT, EXTRAS, LONE>;
        default:                   return rollout_kernel<N_, M_, UAVTRACK_REWARD_RAW, !PLANAR_ONLY, POLICY, ALLOUT, EXTRAS, LONE>;
        }
    }
    switch (mode) {
    case UAVTRACK_REWARD_MEAN: return rollout_kernel<N_, M_, UAVTRACK_REWARD_MEAN, false, POLICY, ALLOUT, EXTRAS, LONE>;
    case UAVTRACK_REWARD_PMI:  return rollout_kernel<N_, M_, PLANAR_ONLY ? UAVTRACK_REWARD_RAW : UAVTRACK_REWARD_PMI, false, POLICY, ALLOUT, EXTRAS, LONE>;
    default:                   return rollout_kernel<N_, M_, UAVTRACK_REWARD_RAW, false, POLICY, ALLOUT, EXTRAS, LONE>;
    }
}

// The launches a single-wavefront (LONE) kernel variant exists for: pre-sampled actions with every output, or the fused
// actor rollout; no extras, planar, a specialised shape up to 20 x 10 (what sweep_fast's prefetch is written for).
constexpr bool lone_shape(int n_spec, int m_spec) { return n_spec > 0 && n_spec <= 20 && m_spec <= 10; }
bool lone_variant_exists(int N, int M, bool z3, int policy, bool allout, bool extras, int actor_hblocks)
{
    const bool shape = (N == 20 && M == 10) || (N == 10 && M == 10) || (N == 5 && M == 3);      // the specialised shapes that pass lone_shape()
    return shape && !z3 && !extras && ((policy == kPolicyGiven && allout) || (policy == kPolicyActor && actor_hblocks == kLoneActorTiles));
}

// Instantiations: pre-sampled actions with every output and no extras (the learner's rollout, the benchmark);
// any policy without extras; any policy with them.
template <int N_, int M_>
KernelFn pick_mode(int mode, bool z3, int policy, bool allout, bool extras, bool lone)
{
    if constexpr (lone_shape(N_, M_)) {
        if (policy == kPolicyGiven && allout && !extras && lone && !z3)
            return pick_reward<N_, M_, kPolicyGiven, true, false, true>(mode, false);
        // the fused actor rollout on single-wavefront groups: wave fences for barriers and, under MAAC-R, pair-list slots
        // from the pool (the 4-wave emission path would make one reservation per workgroup-step on four times the groups)
        if (policy == kPolicyActor && !extras && lone && !z3)
            return pick_reward<N_, M_, kPolicyActor, false, false, true>(mode, false);
    }
    if (policy == kPolicyGreedy)
        return extras ? pick_reward<N_, M_, kPolicyGreedy, false, true>(mode, z3) : pick_reward<N_, M_, kPolicyGreedy, false, false>(mode, z3);
    if (policy == kPolicyActor)
        return extras ? pick_reward<N_, M_, kPolicyActor, false, true>(mode, z3) : pick_reward<N_, M_, kPolicyActor, false, false>(mode, z3);
    if (extras) return pick_reward<N_, M_, kPolicyGiven, false, true>(mode, z3);
    return allout ? pick_reward<N_, M_, kPolicyGiven, true, false>(mode, z3) : pick_reward<N_, M_, kPolicyGiven, false, false>(mode, z3);
}

KernelFn pick_kernel(int N, int M, int mode, bool z3, int *specialised, int policy = kPolicyGiven, bool allout = false,
                     bool extras = false, bool lone = false)
{
    *specialised = 1;
    if (N == 20 && M == 10) return pick_mode<20, 10>(mode, z3, policy, allout, extras, lone);
    if (N == 50 && M == 25) return pick_mode<50, 25>(mode, z3, policy, allout, extras, lone);
    if (N == 10 && M == 10) return pick_mode<10, 10>(mode, z3, policy, allout, extras, lone);
    if (N == 5 && M == 3) return pick_mode<5, 3>(mode, z3, policy, allout, extras, lone);
    *specialised = 0;
    return pick_mode<0, 0>(mode, z3, policy, allout, extras, lone);
}

}  // namespace

Geometry plan_geometry(const uavtrack_config &cfg, int n_simd, bool allow_small_grid)
{
    Geometry g;
    const int N = cfg.n_uav;
    int best = 0;
    int forced = 0;
    if (const char *s = getenv("UAVTRACK_WGS")) forced = atoi(s);
    // Whole multiples of 4 waves (or fewer than 4) keep the four SIMDs of a CU evenly loaded:
    // 320-thread groups (100 % lane use at N = 20) measured 26 % slower than 256 at a
    // chip-filling batch.  Take the best lane utilisation (5 % band).  Inside the band the
    // batch decides (measured, N = 20 M = 10): up to ~3 waves per SIMD single-wave groups win
    // (no cross-wave barrier on a step whose waves mostly run alone on their SIMD: 4096 envs
    // 0.689 ms vs 0.751 ms per 200 steps); past that 4-wave groups win (65 536 envs 39.6 vs 33.9 G).
    static const int kLarge[] = {256, 128, 512, 64};
    static const int kSmall[] = {64, 128, 256, 512};
    // whole environments per workgroup: as many as there are lanes for, fewer when their tables would not fit
    // the 64 KB of LDS a workgroup has by default (few UAVs, many targets: N = 1, M = 70 fits 58 environments, not 64).
    auto lds_need = [&](int wgs, int E) {
        (void)wgs;
        return lds_bytes_for(E, N, cfg.m_targets, cfg.dim == 3, cfg.reward_mode);
    };
    // (kLdsSoft = the 64 KiB a launch gets without asking: several environments per workgroup stay inside it, so two or more
    //  workgroups share a CU; a SINGLE environment whose tables need more -- hundreds of UAVs with thousands of targets --
    //  may take up to the CU's whole 160 KiB, which launch_rollout requests per kernel, hipFuncSetAttribute)
    auto envs_of = [&](int wgs) {
        int E = wgs / N;
        while (E > 1 && lds_need(wgs, E) > kLdsSoft) --E;
        return E;
    };
    auto feasible = [&](int wgs) {
        const int E = envs_of(wgs);
        return E >= 1 && lds_need(wgs, E) <= kLdsMax;
    };
    auto util_of = [&](int wgs) {
        const int E = envs_of(wgs);
        const int Euse = E < cfg.n_envs ? E : cfg.n_envs;
        return (double)Euse * N / wgs;
    };
    // UAVTRACK_WGS (experiments) is honoured when that geometry is feasible, else ignored
    if (forced >= 64 && forced <= kMaxWorkgroup && forced % 64 == 0 && feasible(forced)) {
        best = forced;
    } else {
        double best_util = -1.0;
        for (int wgs : kLarge)
            if (feasible(wgs) && util_of(wgs) > best_util) best_util = util_of(wgs);
        bool small_grid = false;
        long waves64 = 0;
        // (MAAC-R keeps the larger groups: its pair emission costs one global atomic per workgroup-step, and
        // four times the workgroups measured 15.8 instead of 5.2 us per step at 4096 envs)
        // (MAAC-R used to keep the larger groups -- one pair-list reservation per workgroup-step on ONE counter; the
        // single-wavefront variant now reserves per block, see kPoolEmit -- so it follows the same rule, for the swarm
        // sizes that variant is built for)
        int spec_shape = 0;
        pick_kernel(N, cfg.m_targets, cfg.reward_mode, cfg.dim == 3, &spec_shape);
        if (allow_small_grid && feasible(64) &&
            (cfg.reward_mode != UAVTRACK_REWARD_PMI || (spec_shape && N <= 20 && cfg.m_targets <= 10 && cfg.dim == 2))) {
            const long waves = (cfg.n_envs + envs_of(64) - 1) / envs_of(64);
            small_grid = waves <= 3L * (n_simd > 0 ? n_simd : 1024);
            waves64 = waves;
        }
        // (the band is a little wider on a small grid: N = 10 at 4096 envs measured 0.409 ms with
        // 60/64 lanes in use against 0.435 ms with 250/256)
        for (int wgs : small_grid ? kSmall : kLarge)
            if (feasible(wgs) && util_of(wgs) >= best_util - (small_grid ? 0.07 : 0.05)) { best = wgs; break; }
        // at most two waves per SIMD: the LONE kernel variant (deep LDS prefetch, ~215 registers: two resident waves)
        g.lone = small_grid && best == 64 && waves64 <= 2L * (n_simd > 0 ? n_simd : 1024);
    }
    if (!best) return g;
    g.wgs = best;
    g.envs_per_wg = envs_of(best);
    g.groups = (cfg.n_envs + g.envs_per_wg - 1) / g.envs_per_wg;
    g.lds_bytes = lds_bytes_for(g.envs_per_wg, N, cfg.m_targets, cfg.dim == 3, cfg.reward_mode);
    pick_kernel(N, cfg.m_targets, cfg.reward_mode, cfg.dim == 3, &g.specialised);
    return g;
}

size_t rollout_lds_bytes(const Geometry &g, int policy)
{
    (void)policy;               // (the in-kernel actor needs no LDS since round 4: its operands move by lane swaps)
    return g.lds_bytes;
}

hipError_t launch_rollout(uavtrack_env *env, const StepParams &p, hipStream_t stream, int policy, const Geometry *geo)
{
    int spec = 0;
    const bool allout = p.obs && p.reward && p.terms && p.covered && p.done;
    const bool extras = p.auto_reset || p.tpos || p.raw || p.state_copy;
    // MAAC-R: the single-wavefront geometry only pays with the kernel variant written for it (pair-list slots from a pool).
    // Every other launch -- an output not requested, the target trace, the automatic reset -- would run the 4-wave emission
    // path (one pair-list reservation per workgroup-step on ONE counter) on four times the workgroups: measured 15.8
    // against 5.2 us per step at 4096 envs.  Those launches keep the 256-thread geometry.
    const bool lone_ok = lone_variant_exists(p.N, p.M, env->cfg.dim == 3, policy, allout, extras, p.actor_hblocks);
    if (!geo && env->cfg.reward_mode == UAVTRACK_REWARD_PMI && env->geo.lone && !lone_ok)
        geo = &env->geo_short;
    const Geometry &g = geo ? *geo : env->geo;
    env->last_launch = g;
    env->last_launch.lone = g.lone && lone_ok;
    // (an actor of another width than the single-wavefront variant is laid out for runs the general variant on the same geometry)
    const bool lone_kernel = g.lone != 0 && !(policy == kPolicyActor && p.actor_hblocks != kLoneActorTiles);
    KernelFn fn = pick_kernel(p.N, p.M, env->cfg.reward_mode, env->cfg.dim == 3, &spec, policy, allout, extras, lone_kernel);
    StepParams q = p;
    q.E = g.envs_per_wg;
    const size_t lds = rollout_lds_bytes(g, policy);
    if (lds > kLdsSoft) {      // beyond the default limit of a launch: raise this kernel's (a host-side attribute, no stream work)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3(g.groups), dim3(g.wgs), lds, stream, q);
    return hipGetLastError();
}

}  // namespace uavtrack
