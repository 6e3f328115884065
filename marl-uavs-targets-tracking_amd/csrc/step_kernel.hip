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
// synchronisation is the workgroup barrier.  Per environment the workgroup keeps in
// LDS: the UAV pose table in two interleaved copies (this step's post-move pose and
// the previous one -- the reference moves UAVs one after the other, so UAV i sees
// peers j < i after their move and peers j > i before it), the matching action
// table, and the target table.  Lanes of one environment read the same table row in
// the pair sweeps, which the LDS serves as a broadcast.  Pose, heading sin/cos and
// action of the lane's own UAV live in registers across the T steps; HBM sees the
// state once per launch and, per step, only the action read and the output writes.
//
// No MFMA: there is no dense contraction on this path.  The bound is HBM for the
// outputs (48 B of observation per agent-step) against ~1e3 VALU lane-ops per
// agent-step for the all-pairs sweeps.

#include "internal.h"

#include <cstdlib>

// Partial unrolling of the pair sweeps: full unrolling lets the scheduler hoist every
// LDS table read of the environment into registers (256 VGPRs + scratch spills).
#ifndef UAVTRACK_UNROLL_U
#define UAVTRACK_UNROLL_U 5
#endif
#ifndef UAVTRACK_UNROLL_T
#define UAVTRACK_UNROLL_T 5
#endif

namespace uavtrack {

namespace {

struct Acc {
    // peers (uav.py:124-147 rows, folded into sums)
    float cntU, iwU, sxU, syU, scU, ssU, saU;
    // targets (uav.py:101-122 rows)
    float cntT, iwT, sxT, syT, scT, ssT;
    float trk;   // uav.py:199-212
    float dup;   // uav.py:214-229
};

__device__ __forceinline__ float fast_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float fast_exp2(float v) { return __builtin_amdgcn_exp2f(v); }

// Workgroups are dealt round-robin over the 8 XCDs; give consecutive environment
// groups to one XCD so neighbouring output spans land in the same L2 (speed only).
__device__ __forceinline__ int xcd_group(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    return xcd < r ? xcd * (q + 1) + slot : r * (q + 1) + (xcd - r) * q + slot;
}

// (h + pi) % (2 pi) - pi with Python's sign convention (uav.py:97), without
// spending mantissa bits when no wrap is needed.
__device__ __forceinline__ float wrap_heading(float h)
{
    if (h >= kPi) {
        h -= kTwoPi;
        if (h >= kPi) {   // far out of range (injected state): general reduction
            float t = h + kPi;
            t -= kTwoPi * floorf(t * (1.0f / kTwoPi));
            h = t - kPi;
        }
    } else if (h < -kPi) {
        h += kTwoPi;
        if (h < -kPi) {
            float t = h + kPi;
            t -= kTwoPi * floorf(t * (1.0f / kTwoPi));
            h = t - kPi;
        }
    }
    return h;
}

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// sin/cos for |h| <= pi (headings are kept wrapped): quadrant reduction with a two-term
// pi/2 and the Cephes single-precision minimax polynomials on [-pi/4, pi/4] (|err| ~ 1e-7).
// ocml's sincosf carries a large-argument (Payne-Hanek) path the kernel never needs inside
// the step loop; it is only used for injected state whose heading is out of range.
__device__ __forceinline__ void sincos_wrapped(float h, float *s, float *c)
{
    const float q = rintf(h * 0.63661977236758134308f);
    float r = fmaf(-q, 1.57079637050628662109375f, h);       // fl32(pi/2); q in {-2..2}: q * hi is exact
    r = fmaf(-q, -4.371139006309477e-08f, r);                // pi/2 - fl32(pi/2)
    const float z = r * r;
    const float sp = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float cp = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                          fmaf(z, -0.5f, 1.0f));
    const int qi = (int)q;
    const float ss = (qi & 1) ? cp : sp;
    const float cc = (qi & 1) ? sp : cp;
    *s = (qi & 2) ? -ss : ss;
    *c = ((qi + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ void sincos_any(float h, float *s, float *c)
{
    if (__builtin_expect(fabsf(h) <= kPi, 1)) sincos_wrapped(h, s, c);
    else sincosf(h, s, c);
}

// ---------------------------------------------------------------------------------------
// Pair sweeps of one UAV, fast path: packed fp32 (v_pk_*) on (x, y) / (cos, sin) pairs,
// no per-pair `j != i` test.  The j == i iteration adds a known self term (distance 0 on
// the post-move table, own pre-move pose on the sequential table); the accumulators start
// at minus that term.  The uav.py:165/179 weight is 1 here (see sweep_weighted).
template <int N_, int M_, bool Z3>
__device__ __forceinline__ void sweep_fast(const StepParams &p, int N, int M, int ebaseT, int i,
                                           const float4 *__restrict__ rowNew, const float4 *__restrict__ rowOld,
                                           const float4 *__restrict__ ttab,
                                           const float *__restrict__ tzt, unsigned *__restrict__ covw, int covbase,
                                           float xi, float yi, float zi, float ci, float si, float ai,
                                           float xo, float yo, float zo, float co, float so, float ao, Acc &a)
{
    const v2f pos = {xi, yi};

    // ---- targets: observe_target (<= dp), tracking reward (<= dp), coverage (< dp)
    v2f sxyT = {0.f, 0.f}, scsT = {0.f, 0.f};
    float cntT = 0.f, trk = 0.f;
    unsigned bits = 0;
#pragma unroll UAVTRACK_UNROLL_T
    for (int k = 0; k < (M_ > 0 ? M_ : M); ++k) {
        const float4 tg = ttab[ebaseT + k];
        const v2f d = (v2f){tg.x, tg.y} - pos;
        const v2f sq = d * d;
        float d2 = sq.x + sq.y;
        if (Z3) {
            const float dz = tzt[ebaseT + k] - zi;
            d2 = fmaf(dz, dz, d2);
        }
        const float dist = fast_sqrt(d2);
        const bool in = d2 <= p.dp2;
        const float m = in ? 1.0f : 0.0f;
        const v2f mm = {m, m};
        sxyT = pk_fma(mm, d, sxyT);
        scsT = pk_fma(mm, (v2f){tg.z, tg.w}, scsT);
        cntT += m;
        trk = fmaf(m, fmaf(-dist, p.inv_dp, 2.0f), trk);   // 1 + (dp - d)/dp
        bits |= (d2 < p.dp2) ? (1u << (k & 31)) : 0u;
        if ((k & 31) == 31 || k == (M_ > 0 ? M_ : M) - 1) {
            if (bits) atomicOr(&covw[covbase + (k >> 5)], bits);
            bits = 0;
        }
    }

    // ---- peers.  Self terms first.
    const float dup_self = fast_exp2(p.exp_k0);            // exp2(k0 - k1 * 0), as the loop computes it
    const v2f dself = (v2f){xo, yo} - pos;
    const v2f sqs = dself * dself;
    float d2self = sqs.x + sqs.y;
    if (Z3) d2self = fmaf(zo - zi, zo - zi, d2self);
    const float mself = (d2self <= p.dc2) ? 1.0f : 0.0f;   // own pre-move pose, seen at j == i
    const v2f nself = {-mself, -mself};
    v2f sxyU = nself * dself, scsU = nself * (v2f){co, so}, sacU = nself * (v2f){ao, 1.0f};
    float dup = -dup_self;
#pragma unroll UAVTRACK_UNROLL_U
    for (int j = 0; j < (N_ > 0 ? N_ : N); ++j) {
        const float4 *rs = (j < i) ? rowNew : rowOld;       // one select serves pose, action and z
        const float4 nw = rowNew[j * 4];
        const float4 mx = rs[j * 4];
        const float4 ax = rs[j * 4 + 1];                    // (action, 1, z, -)
        const v2f am = {ax.x, ax.y};
        const v2f dn = (v2f){nw.x, nw.y} - pos;
        const v2f dm = (v2f){mx.x, mx.y} - pos;
        const v2f sqn = dn * dn, sqm = dm * dm;
        float d2n = sqn.x + sqn.y, d2m = sqm.x + sqm.y;
        if (Z3) {
            const float dzn = rowNew[j * 4 + 1].z - zi, dzm = ax.z - zi;
            d2n = fmaf(dzn, dzn, d2n);
            d2m = fmaf(dzm, dzm, d2m);
        }
        const float ex = fast_exp2(fmaf(fast_sqrt(d2n), -p.exp_k1, p.exp_k0));
        dup += (d2n <= p.two_dp2) ? ex : 0.0f;
        const float m = (d2m <= p.dc2) ? 1.0f : 0.0f;
        const v2f mm = {m, m};
        sxyU = pk_fma(mm, dm, sxyU);
        scsU = pk_fma(mm, (v2f){mx.z, mx.w}, scsU);
        sacU = pk_fma(mm, am, sacU);
    }
    // sum_j m (c_j - c_i) = sum_j m c_j - c_i cnt; targets carry the speed ratio
    a.cntT = cntT; a.iwT = cntT; a.sxT = sxyT.x; a.syT = sxyT.y; a.trk = trk;
    a.scT = fmaf(scsT.x, p.vratio, -ci * cntT);
    a.ssT = fmaf(scsT.y, p.vratio, -si * cntT);
    a.cntU = sacU.y; a.iwU = sacU.y; a.sxU = sxyU.x; a.syU = sxyU.y; a.dup = dup;
    a.scU = fmaf(-ci, sacU.y, scsU.x);
    a.ssU = fmaf(-si, sacU.y, scsU.y);
    a.saU = fmaf(-ai, sacU.y, sacU.x);
}

// Literal form with the uav.py:165/179 weight min(dist((rel_x, rel_y), (abs_x, abs_y)), 1).
// It differs from 1 only when the UAV sits within ~2.5 m of the origin, so this path runs
// for the few wavefronts that hold such a UAV and favours clarity over speed.
template <int N_, int M_, bool Z3>
__device__ __forceinline__ void sweep_weighted(const StepParams &p, int N, int M, int ebaseT, int i,
                                            const float4 *__restrict__ rowNew, const float4 *__restrict__ rowOld,
                                            const float4 *__restrict__ ttab,
                                            const float *__restrict__ tzt, unsigned *__restrict__ covw, int covbase,
                                            float xi, float yi, float zi, float ci, float si, float ai, Acc &a)
{
    a = Acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned bits = 0;
#pragma unroll 1
    for (int k = 0; k < M; ++k) {
        const float4 tg = ttab[ebaseT + k];
        const float dx = tg.x - xi, dy = tg.y - yi;
        const v2f sq = (v2f){dx, dy} * (v2f){dx, dy};        // same products as the fast path
        float d2 = sq.x + sq.y;
        if (Z3) {
            const float dz = tzt[ebaseT + k] - zi;
            d2 = fmaf(dz, dz, d2);
        }
        const float d = fast_sqrt(d2);
        const bool in = d2 <= p.dp2;
        const float m = in ? 1.0f : 0.0f;
        const float rx = dx * p.inv_dp - xi, ry = dy * p.inv_dp - yi;
        const float iw = in ? 1.0f / fminf(sqrtf(rx * rx + ry * ry), 1.0f) : 0.0f;
        a.scT = fmaf(iw, tg.z * p.vratio - ci, a.scT);
        a.ssT = fmaf(iw, tg.w * p.vratio - si, a.ssT);
        a.cntT += m;
        a.iwT += iw;
        a.sxT = fmaf(iw, dx, a.sxT);
        a.syT = fmaf(iw, dy, a.syT);
        a.trk = fmaf(m, fmaf(-d, p.inv_dp, 2.0f), a.trk);
        bits |= (d2 < p.dp2) ? (1u << (k & 31)) : 0u;
        if ((k & 31) == 31 || k == M - 1) {
            if (bits) atomicOr(&covw[covbase + (k >> 5)], bits);
            bits = 0;
        }
    }
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
        const bool other = (j != i);
        const float4 *rs = (j < i) ? rowNew : rowOld;
        const float4 nw = rowNew[j * 4];
        const float4 mx = rs[j * 4];
        const float4 ax = rs[j * 4 + 1];
        const float am = ax.x;
        const float dxn = nw.x - xi, dyn = nw.y - yi, dxm = mx.x - xi, dym = mx.y - yi;
        const v2f sqn = (v2f){dxn, dyn} * (v2f){dxn, dyn}, sqm = (v2f){dxm, dym} * (v2f){dxm, dym};
        float d2n = sqn.x + sqn.y, d2m = sqm.x + sqm.y;
        if (Z3) {
            const float dzn = rowNew[j * 4 + 1].z - zi, dzm = ax.z - zi;
            d2n = fmaf(dzn, dzn, d2n);
            d2m = fmaf(dzm, dzm, d2m);
        }
        const float ex = fast_exp2(fmaf(fast_sqrt(d2n), -p.exp_k1, p.exp_k0));
        a.dup += (other && d2n <= p.two_dp2) ? ex : 0.0f;
        const bool inm = other && d2m <= p.dc2;
        const float m = inm ? 1.0f : 0.0f;
        const float rx = dxm * p.inv_dc - xi, ry = dym * p.inv_dc - yi;
        const float iw = inm ? 1.0f / fminf(sqrtf(rx * rx + ry * ry), 1.0f) : 0.0f;
        a.scU = fmaf(iw, mx.z - ci, a.scU);
        a.ssU = fmaf(iw, mx.w - si, a.ssU);
        a.saU = fmaf(iw, am - ai, a.saU);
        a.cntU += m;
        a.iwU += iw;
        a.sxU = fmaf(iw, dxm, a.sxU);
        a.syU = fmaf(iw, dym, a.syU);
    }
}

template <int N_, int M_, int MODE, bool Z3>
__global__ void __launch_bounds__(kMaxWorkgroup) rollout_kernel(const StepParams p)
{
    extern __shared__ float4 smem4[];
    const int N = N_ > 0 ? N_ : p.N;
    const int M = M_ > 0 ? M_ : p.M;
    const int E = p.E;
    const int EN = E * N, EM = E * M;
    const int CW = (M + 31) >> 5;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;

    // ---- LDS carve (float4 first: the dynamic base is 16-B aligned)
    // UAV table: per (env, uav) two 32-B rows (copy 0 / copy 1, one of them "post-move of this
    // step"), each (x, y, cos h, sin h | action, 1, z, -).  The env stride carries 32 B of padding so
    // that rows of different envs read by one lane group start in different banks.
    const int ustride = N * 4 + 2;        // float4 units per env
    float4 *utab = smem4;                 // [E][N][2][2]
    float4 *ttab = utab + E * ustride;    // [EM]     (x, y, cos h, sin h)
    float *fb = reinterpret_cast<float *>(ttab + EM);
    float *thd = fb;   fb += EM;          // [EM]     target heading
    float *rawl = fb;  fb += EN;          // [EN]     raw reward (cooperative modes)
    float *tzt = fb;   if (Z3) fb += EM;
    unsigned *covw = reinterpret_cast<unsigned *>(fb);   // [2][E * CW]

    const int grp = xcd_group(blockIdx.x, gridDim.x);
    const int env0 = grp * E;
    const int envs_here = min(E, p.B - env0);
    const int e = tid / N;
    const int i = tid - e * N;
    const int b = env0 + e;
    const bool active = (tid < EN) && (e < envs_here);
    const size_t g = (size_t)b * N + i;          // flat (env, uav)
    const size_t BN = (size_t)p.B * N;
    const int ebaseU = e * N, ebaseT = e * M;

    float x = 0, y = 0, z = 0, h = 0, c = 1, s = 0;
    int a_prev = 0, count = 0;
    float er = 0, ett = 0, ebp = 0, edup = 0;    // episode accumulators (train.py:181-192)
    int ecov = 0;
    int pn = 0;                                   // which table copy is "post-move" this step

    // ---- load state once
    const StateBlock &S = *p.st;
    if (active) {
        x = S.ux[g]; y = S.uy[g]; h = S.uh[g]; a_prev = S.ua[g];
        if (Z3) z = S.uz[g];
        sincos_any(h, &s, &c);
        count = S.step_count[b];
        float4 *own = utab + e * ustride + i * 4 + 2;           // "previous" copy for step 0
        own[0] = make_float4(x, y, c, s);
        own[1] = make_float4((float)a_prev, 1.0f, z, 0.0f);
    }
    for (int q = tid; q < envs_here * M; q += nthreads) {
        const size_t gt = (size_t)env0 * M + q;
        const float th = S.th[gt];
        float ts, tc;
        sincos_any(th, &ts, &tc);
        ttab[q] = make_float4(S.tx[gt], S.ty[gt], tc, ts);
        thd[q] = th;
        if (Z3) tzt[q] = S.tz[gt];
    }
    if (active && i == 0)
        for (int w = 0; w < CW; ++w) covw[e * CW + w] = covw[E * CW + e * CW + w] = 0;
    int act = 0;
    if (active) act = p.actions[g];
    __syncthreads();

    for (int t = 0; t < p.T; ++t) {
        const size_t tg_off = (size_t)t * BN + g;          // [t][b][i]
        const int cbuf = (t & 1) * E * CW;

        // ---- P1a: targets (target.py:27-60); straight flight, mirror at the walls
        for (int q = tid; q < envs_here * M; q += nthreads) {
            float4 tg = ttab[q];
            float th = thd[q];
            tg.x = fmaf(p.dtv_t, tg.z, tg.x);
            tg.y = fmaf(p.dtv_t, tg.w, tg.y);
            bool turned = false;
            if (0.0f > tg.y || tg.y > p.y_max) {
                th = -th;
                turned = true;
            } else if (tg.x < 0.0f || tg.x > p.x_max) {
                th = (th > 0.0f) ? kPi - th : -kPi - th;
                turned = true;
            }
            if (turned) {   // rare: recompute so that T fused steps == T single steps bit for bit
                sincos_any(th, &tg.w, &tg.z);
                thd[q] = th;
            }
            ttab[q] = tg;
        }

        // ---- P1b: own kinematics (uav.py:83-99); position uses the OLD heading
        int a_now = 0;
        float ai = 0;
        float xo = x, yo = y, zo = z, co = c, so = s, ao = (float)a_prev;   // pre-move pose: the j == i self term
        if (active) {
            a_now = min(max(act, 0), p.na_total - 1);
            int a_turn = a_now, a_climb = 0;
            if (Z3) { a_climb = a_now / p.na; a_turn = a_now - a_climb * p.na; }
            float step_xy = p.dtv_u;
            if (Z3) {
                step_xy = p.dtv_u * S.climb_c[a_climb];
                z = fmaf(p.dtv_u, S.climb_s[a_climb], z);
            }
            x = fmaf(step_xy, c, x);
            y = fmaf(step_xy, s, y);
            h = wrap_heading(fmaf((float)(2 * a_turn + 1 - p.na), p.turn_unit, h));
            sincos_wrapped(h, &s, &c);
            ai = (float)a_now;
            float4 *own = utab + e * ustride + i * 4 + pn * 2;
            own[0] = make_float4(x, y, c, s);
            own[1] = make_float4(ai, 1.0f, z, 0.0f);
            if (t + 1 < p.T) act = p.actions[tg_off + BN];   // prefetch next step's action
        }
        if (active && i == 0)
            for (int w = 0; w < CW; ++w) covw[cbuf + e * CW + w] = 0;
        __syncthreads();

        // ---- P2: pair sweeps
        float o[12], tt = 0, bp = 0, dupn = 0, raw = 0;
        if (active) {
            Acc acc;
            // weight of uav.py:165 can be < 1 only near the origin; wave-uniform branch
            const bool near0 = fabsf(x) < 2.5f && fabsf(y) < 2.5f;
            const float4 *rowNew = utab + e * ustride + pn * 2;
            const float4 *rowOld = utab + e * ustride + (pn ^ 1) * 2;
            if (__builtin_expect(__any(near0), 0))
                sweep_weighted<N_, M_, Z3>(p, N, M, ebaseT, i, rowNew, rowOld, ttab, tzt, covw,
                                           cbuf + e * CW, x, y, z, c, s, ai, acc);
            else
                sweep_fast<N_, M_, Z3>(p, N, M, ebaseT, i, rowNew, rowOld, ttab, tzt, covw,
                                       cbuf + e * CW, x, y, z, c, s, ai, xo, yo, zo, co, so, ao, acc);

            // ---- P3: local state (uav.py:156-190)
            if (acc.cntU > 0.0f) {
                const float rc = fast_rcp(acc.cntU);
                o[0] = acc.sxU * p.inv_dc * rc;
                o[1] = acc.syU * p.inv_dc * rc;
                o[2] = acc.scU * rc;
                o[3] = acc.ssU * rc;
                o[4] = acc.saU * p.inv_na_total * rc;
            } else {
                o[0] = o[1] = o[2] = o[3] = o[4] = -1.0f;
            }
            if (acc.cntT > 0.0f) {
                const float rc = fast_rcp(acc.cntT);
                o[5] = acc.sxT * p.inv_dp * rc;
                o[6] = acc.syT * p.inv_dp * rc;
                o[7] = acc.scT * rc;
                o[8] = acc.ssT * rc;
            } else {
                o[5] = o[6] = o[7] = o[8] = -1.0f;
            }
            o[9] = x * p.inv_dc;
            o[10] = y * p.inv_dc;
            o[11] = ai * p.inv_na_total;

            // ---- raw reward terms, clipped and normalised (environment.py:207-220)
            float d_bdr = fminf(fminf(x, p.x_max - x), fminf(y, p.y_max - y));
            if (Z3) d_bdr = fminf(d_bdr, fminf(z, p.z_max - z));
            float bpun = (d_bdr >= 0.0f) ? ((d_bdr < p.dp) ? -0.5f * (p.dp - d_bdr) * p.inv_dp : 0.0f) : -0.5f;
            tt = fminf(fmaxf(acc.trk, 0.0f), p.tt_ceil) * p.inv_tt_ceil;
            bp = (fminf(fmaxf(bpun, -0.5f), 0.0f) + 0.5f) * 2.0f - 1.0f;
            dupn = (fminf(fmaxf(acc.dup * -0.5f, p.dup_floor), 0.0f) - p.dup_floor) * p.inv_dup - 1.0f;
            raw = p.alpha * tt + p.beta * bp + p.gamma * dupn;
            if (MODE != UAVTRACK_REWARD_RAW) rawl[tid] = raw;
        }
        __syncthreads();

        // ---- P4: cooperative reward, coverage, outputs
        if (active) {
            float r = raw;
            if (MODE == UAVTRACK_REWARD_MEAN) {
                if (p.coop != 0.0f) {   // uav.py:293-310
                    float sum = 0, cnt = 0;
                    const float4 *rowNew = utab + e * ustride + pn * 2;
#pragma unroll UAVTRACK_UNROLL_U
                    for (int j = 0; j < (N_ > 0 ? N_ : N); ++j) {
                        const float4 nw = rowNew[j * 4];
                        const v2f dd = (v2f){nw.x, nw.y} - (v2f){x, y};
                        const v2f sq = dd * dd;                  // same products as the sweeps
                        float d2 = sq.x + sq.y;
                        if (Z3) { const float dz = rowNew[j * 4 + 1].z - z; d2 = fmaf(dz, dz, d2); }
                        const bool nb = (j != i) && d2 <= p.dp2;
                        sum += nb ? rawl[ebaseU + j] : 0.0f;
                        cnt += nb ? 1.0f : 0.0f;
                    }
                    r = (cnt > 0.0f) ? (1.0f - p.coop) * raw + p.coop * sum / cnt : 0.0f;
                }
            }
            r = fminf(fmaxf(r, -1.0f), 1.0f);   // clip_and_normalize(reward, -1, 1), environment.py:225

            ++count;
            if (i == 0) {
                int cov = 0;
                for (int w = 0; w < CW; ++w) cov += __popc(covw[cbuf + e * CW + w]);
                ecov += cov;
                const size_t tb = (size_t)t * p.B + b;
                if (p.covered) p.covered[tb] = cov;
                if (p.done) p.done[tb] = (p.horizon > 0 && count >= p.horizon) ? 1 : 0;
            }
            er += r; ett += tt; ebp += bp; edup += dupn;

            if (p.obs) {
                float4 *op = reinterpret_cast<float4 *>(p.obs + tg_off * UAVTRACK_OBS_DIM);
                op[0] = make_float4(o[0], o[1], o[2], o[3]);
                op[1] = make_float4(o[4], o[5], o[6], o[7]);
                op[2] = make_float4(o[8], o[9], o[10], o[11]);
            }
            if (p.reward) p.reward[tg_off] = r;
            if (p.raw_out) p.raw_out[tg_off] = raw;
            if (p.terms) {
                float *tp = p.terms + (size_t)t * 3 * BN + g;   // [t][3][b][i]
                tp[0] = tt;
                tp[BN] = bp;
                tp[2 * BN] = dupn;
            }
            a_prev = a_now;
        }

        // ---- MAAC-R only: emit the neighbour pairs (i < j, d <= dp on post-move poses, uav.py:278) this
        //      workgroup owns into the global pair list the PMI scoring kernel consumes.  s_ij = s_ji
        //      (the input is la_i * la_j), so unordered pairs halve the work.
        if (MODE == UAVTRACK_REWARD_PMI) {
            unsigned *wg_cnt = covw + 2 * E * CW;          // two extra words behind the coverage masks
            if (tid == 0) wg_cnt[0] = 0;
            __syncthreads();
            int mine = 0, slot = 0;
            const float4 *rowNew = utab + e * ustride + pn * 2;
            if (active) {
                for (int j = i + 1; j < N; ++j) {
                    const float4 nw = rowNew[j * 4];
                    const v2f dd = (v2f){nw.x, nw.y} - (v2f){x, y};
                    const v2f sq = dd * dd;
                    float d2 = sq.x + sq.y;
                    if (Z3) { const float dz = rowNew[j * 4 + 1].z - z; d2 = fmaf(dz, dz, d2); }
                    mine += (d2 <= p.dp2) ? 1 : 0;
                }
                if (mine) slot = (int)atomicAdd(&wg_cnt[0], (unsigned)mine);
            }
            __syncthreads();
            if (tid == 0) wg_cnt[1] = wg_cnt[0] ? atomicAdd(p.pair_count, wg_cnt[0]) : 0u;
            __syncthreads();
            if (active && mine) {
                uint2 *dst = p.pairs + wg_cnt[1] + slot;
                for (int j = i + 1; j < N; ++j) {
                    const float4 nw = rowNew[j * 4];
                    const v2f dd = (v2f){nw.x, nw.y} - (v2f){x, y};
                    const v2f sq = dd * dd;
                    float d2 = sq.x + sq.y;
                    if (Z3) { const float dz = rowNew[j * 4 + 1].z - z; d2 = fmaf(dz, dz, d2); }
                    if (d2 <= p.dp2) *dst++ = make_uint2((unsigned)g, (unsigned)j);
                }
            }
        }
        pn ^= 1;
    }

    // ---- store state once
    if (active) {
        S.ux[g] = x; S.uy[g] = y; S.uh[g] = h; S.ua[g] = a_prev;
        if (Z3) S.uz[g] = z;
        if (i == 0) S.step_count[b] = count;
    }
    for (int q = tid; q < envs_here * M; q += nthreads) {
        const size_t gt = (size_t)env0 * M + q;
        const float4 tg = ttab[q];
        S.tx[gt] = tg.x; S.ty[gt] = tg.y; S.th[gt] = thd[q];
    }
    if (p.ep_sums) {
        __syncthreads();                       // everyone is done with utab
        float4 *eps = utab;
        if (active) eps[tid] = make_float4(er, ett, ebp, edup);
        __syncthreads();
        if (active && i == 0) {
            float4 sum = make_float4(0, 0, 0, 0);
            for (int j = 0; j < N; ++j) {       // fixed order: bitwise reproducible
                const float4 v = eps[ebaseU + j];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            const float inv_n = 1.0f / (float)N;
            float *ep = p.ep_sums + (size_t)b * 5;
            ep[0] = sum.x * inv_n; ep[1] = sum.y * inv_n; ep[2] = sum.z * inv_n; ep[3] = sum.w * inv_n;
            ep[4] = (float)ecov;
        }
    }
}

size_t lds_bytes_for(int E, int N, int M, bool z3)
{
    const size_t EN = (size_t)E * N, EM = (size_t)E * M, CW = (M + 31) / 32;
    size_t f4 = (size_t)E * (N * 4 + 2) + EM;
    size_t f = EM + EN + (z3 ? EM : 0) + 2 * E * CW + 2;
    return f4 * 16 + f * 4;
}

using KernelFn = void (*)(const StepParams);

template <int N_, int M_>
KernelFn pick_mode(int mode, bool z3)
{
    if (z3) {
        switch (mode) {
        case UAVTRACK_REWARD_MEAN: return rollout_kernel<N_, M_, UAVTRACK_REWARD_MEAN, true>;
        case UAVTRACK_REWARD_PMI:  return rollout_kernel<N_, M_, UAVTRACK_REWARD_PMI, true>;
        default:                   return rollout_kernel<N_, M_, UAVTRACK_REWARD_RAW, true>;
        }
    }
    switch (mode) {
    case UAVTRACK_REWARD_MEAN: return rollout_kernel<N_, M_, UAVTRACK_REWARD_MEAN, false>;
    case UAVTRACK_REWARD_PMI:  return rollout_kernel<N_, M_, UAVTRACK_REWARD_PMI, false>;
    default:                   return rollout_kernel<N_, M_, UAVTRACK_REWARD_RAW, false>;
    }
}

KernelFn pick_kernel(int N, int M, int mode, bool z3, int *specialised)
{
    *specialised = 1;
    if (N == 20 && M == 10) return pick_mode<20, 10>(mode, z3);
    if (N == 50 && M == 25) return pick_mode<50, 25>(mode, z3);
    if (N == 10 && M == 10) return pick_mode<10, 10>(mode, z3);
    if (N == 5 && M == 3) return pick_mode<5, 3>(mode, z3);
    *specialised = 0;
    return pick_mode<0, 0>(mode, z3);
}

}  // namespace

Geometry plan_geometry(const uavtrack_config &cfg)
{
    Geometry g;
    const int N = cfg.n_uav;
    int best = 0;
    double best_util = -1.0;
    int forced = 0;
    if (const char *s = getenv("UAVTRACK_WGS")) forced = atoi(s);
    // Whole multiples of 4 waves (or fewer than 4) keep the four SIMDs of a CU evenly loaded:
    // 320-thread groups (100 % lane use at N = 20) measured 26 % slower than 256 at a
    // chip-filling batch.  Among the candidates take the best lane utilisation; ties go to
    // the size that measured fastest (256, then 128, 512, 64).
    static const int kCandidates[] = {256, 128, 512, 64};
    for (int wgs : kCandidates) {
        if (forced && wgs != forced) continue;
        const int E = wgs / N;
        if (E < 1) continue;
        if (lds_bytes_for(E, N, cfg.m_targets, cfg.dim == 3) > 64 * 1024) continue;   // e.g. N = 1 with many targets
        const int Euse = E < cfg.n_envs ? E : cfg.n_envs;
        const double util = (double)Euse * N / wgs;
        if (util > best_util + 0.05) { best_util = util; best = wgs; }
    }
    if (!best && forced >= 64 && forced <= kMaxWorkgroup && forced % 64 == 0 && forced / N >= 1) best = forced;
    if (!best) return g;
    g.wgs = best;
    g.envs_per_wg = best / N;
    g.groups = (cfg.n_envs + g.envs_per_wg - 1) / g.envs_per_wg;
    g.lds_bytes = lds_bytes_for(g.envs_per_wg, N, cfg.m_targets, cfg.dim == 3);
    pick_kernel(N, cfg.m_targets, cfg.reward_mode, cfg.dim == 3, &g.specialised);
    return g;
}

hipError_t launch_rollout(const uavtrack_env *env, const StepParams &p, hipStream_t stream)
{
    int spec = 0;
    KernelFn fn = pick_kernel(p.N, p.M, env->cfg.reward_mode, env->cfg.dim == 3, &spec);
    const Geometry &g = env->geo;
    hipLaunchKernelGGL(fn, dim3(g.groups), dim3(g.wgs), g.lds_bytes, stream, p);
    return hipGetLastError();
}

}  // namespace uavtrack
