// pmi_kernel.hip -- the MAAC-R cooperative reward (reference src/agent/uav.py:262-291),
// i.e. PMINetwork.inference (src/models/PMINet.py:41-72, eval mode) on every neighbour
// pair, then the per-UAV softmax mix.
//
// This is the one dense contraction on the path: per pair a 12 -> 3H -> H -> 1 MLP whose
// 3H x H layer is 98 % of the work (2*3H*H = 98 304 FLOP at H = 128).  The reference runs it
// in fp32 torch, so it is kept in exact fp32 on the matrix cores:
// v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain.
//
// pmi_score_kernel<H>: weight-stationary.  A workgroup has H/32 wavefronts; wavefront w owns
// the 32 output columns [32w, 32w+32) of the folded fc1 matrix and keeps its whole
// 3H x 32 slice in registers (3H/2 VGPRs per lane -- 192 at H = 128) for the lifetime of the
// (persistent, grid-striding) workgroup, so in steady state no weight byte moves.  Per tile
// of 32 pairs: all threads gather the two 12-float observations of their pair from HBM/L2,
// form x = la_i * la_j, run the three small branch layers (5/4/3 -> H, BatchNorm folded,
// ReLU) on the VALU from LDS-resident weights and write the 32 x 3H activation tile to LDS
// in MFMA A-operand order; then each wavefront runs 3H/2 back-to-back MFMAs reading one
// ds_read_b128 per four of them.  Epilogue: ReLU, times fc2, fixed-order lane reduction on the DPP paths,
// fixed-order sum over the H/32 column blocks -> s_ij (= s_ji), stored once, in the pair's slot of the pair list.
//
// pmi_mix_kernel: one thread per UAV-step: neighbours in index order, max-shifted softmax of
// their scores (scipy.special.softmax, uav.py:287), reward = (1-a) raw_i + a sum_j w_j raw_j,
// (1-a) raw_i without neighbours (uav.py:290), final clip (environment.py:225).  It finds the
// score of (i, j) through the neighbour records the rollout kernel wrote (internal.h: nbrec_words):
// UAV min(i, j) emitted the pair, in the slot "its first slot + rank of max(i, j) among its later neighbours".

#include "internal.h"

#include <cstring>
#include <utility>

namespace uavtrack {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// Sum of a value over the 32 lanes of each wavefront half in a fixed order, on the VALU's cross-lane (DPP) paths --
// no LDS crossbar traffic beside the MFMA operand reads.  The total of lanes 0..31 lands in lanes 16..31, that of
// lanes 32..63 in lanes 48..63.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float half_wave_sum(float v)
{
    v = dpp_add<0xB1>(v);            // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);            // quad_perm [2,3,0,1]: every lane of a quad holds the quad's sum
    v = dpp_add<0x141>(v);           // row_half_mirror: + the other quad of the 8
    v = dpp_add<0x140>(v);           // row_mirror: + the other 8 of the row -> every lane holds its row's 16-lane sum
    v = dpp_add<0x142, 0xa>(v);      // row_bcast15 into rows 1 and 3: + the previous row's sum
    return v;
}

struct PmiParams {
    const float *blob;       // folded weights, layout of uavtrack_set_pmi_weights
    const void *x6;          // fc1 as three bf16 planes in MFMA B-operand order (pack_pmi_x6)
    const void *l1;          // the three branch layers as f16 planes in MFMA A-operand order (pack_pmi_l1), with t3
    const void *t3;          // fc1 as two f16 planes of T * w (hi, unscaled remainder) for pmi_score_t3_kernel (pack_pmi_t3), or null
    float t3_scale;          // S1 * T: the power-of-two scale the t3 kernel's layer-2 accumulators carry (b1 goes in times this)
    float t3_inv_scale;      // ... and its reciprocal (folded into w2)
    const float *obs;        // [S][B][N][12] local states of the chunk's steps
    const uint2 *pairs;      // {flat [step][b][i] index of i within the chunk, that of j} (0xFFFFFFFF: a dummy of the slot pool)
    const unsigned *pair_count;
    float *scores;           // one per pair, in pair-list order
    unsigned long long *pair_total;
    int32_t N;
    // f16 range watch of pmi_score_t3_kernel: 1 / (largest |x| each branch's inputs may reach before an f16 operand could
    // saturate; uavtrack_set_pmi_weights); a tile that exceeds one raises *range_flag, and the wide-range kernel launched
    // behind it (gate != null: it returns at once while *gate == 0) scores the chunk again
    float rng_inv[3];
    unsigned *range_flag;
    const unsigned *gate;
};

template <int H>
__global__ void __launch_bounds__(H * 2, 2) pmi_score_kernel(const PmiParams q)
{
    constexpr int K = 3 * H;             // fc1 input width
    constexpr int KH = K / 2;            // MFMA k-steps (32x32x2)
    constexpr int ROW = KH + 4;          // padded LDS row: 32 rows of one k-parity never share a bank slot
    constexpr int NW = H / 32;           // wavefronts = column blocks
    constexpr int NT = NW * 64;          // threads = 2 H
    constexpr int PPT = 32 / (NT / H);   // pairs per thread in the branch layers (16)

    __shared__ float4 lds4[(2 * 32 * ROW + 2 * 32 * 12 + 2 * NW * 32 + 20 * H) / 4 + 2];
    float *h0s = reinterpret_cast<float *>(lds4);          // [2][32][ROW]  (k parity, pair, k/2)
    float *xs = h0s + 2 * 32 * ROW;                        // [2 tiles][32 pairs][12]  x = la_i * la_j
    float *part = xs + 2 * 32 * 12;                        // [2 tiles][NW][32] per-column-block partial scores
    float *bw = part + 2 * NW * 32;                        // [H][20] branch-layer weights of output o: wc[5] bc wo[4] bo wb[3] bb, pad

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int col = w * 32 + (lane & 31);
    const int kh = lane >> 5;

    // ---- stationary operands
    // fc1 is stored on the device pre-packed in register order (pack_pmi_blob): per column block and
    // group of four k-steps one float4 per lane, so the stationary slice arrives in KH/4 fully
    // coalesced 1-KiB loads per wavefront instead of KH strided dword loads.
    const float *W1 = q.blob + 15 * H;
    const float *b1 = W1 + (size_t)K * H;
    const float *w2 = b1 + H;
    const float b2 = w2[H];
    float breg[KH];
    const float4 *wp = reinterpret_cast<const float4 *>(W1) + (size_t)w * (KH / 4) * 64 + lane;
#pragma unroll
    for (int t4 = 0; t4 < KH / 4; ++t4) {                  // breg[t] = B[k = 2t + kh][col]
        const float4 v = wp[t4 * 64];
        breg[4 * t4 + 0] = v.x; breg[4 * t4 + 1] = v.y; breg[4 * t4 + 2] = v.z; breg[4 * t4 + 3] = v.w;
    }
    const float bias1 = b1[col], wout = w2[col];
    // Branch layers (PMINet.py:50-55): thread (o, half) owns output o of each of the three branches for
    // 16 of the tile's 32 pairs; its 15 folded weights stay in registers, only the pair inputs come
    // from LDS (three broadcast ds_read_b128 per pair).
    // Their 15 folded weights and 3 biases per output live in LDS and are fetched at the head of every tile (five
    // ds_read_b128): held in registers for the whole kernel they pushed the register-stationary fc1 slice into
    // scratch spills.
    const int o = tid % H, phalf = tid / H;
    if (tid < H) {
        float *d = bw + tid * 20;
#pragma unroll
        for (int v = 0; v < 5; ++v) d[v] = q.blob[v * H + tid];
        d[5] = q.blob[5 * H + tid];
#pragma unroll
        for (int v = 0; v < 4; ++v) d[6 + v] = q.blob[6 * H + v * H + tid];
        d[10] = q.blob[10 * H + tid];
#pragma unroll
        for (int v = 0; v < 3; ++v) d[11 + v] = q.blob[11 * H + v * H + tid];
        d[14] = q.blob[14 * H + tid];
        d[15] = d[16] = d[17] = d[18] = d[19] = 0.0f;
    }
    // where this thread's three outputs live in the activation tile: concat order (PMINet.py:58)
    // comm | obs | boundary_state; k -> (parity k & 1, step k >> 1)
    const int k0 = o, k1 = H + o, k2 = 2 * H + o;
    float *const hc0 = h0s + ((k0 & 1) * 32) * ROW + (k0 >> 1);
    float *const hc1 = h0s + ((k1 & 1) * 32) * ROW + (k1 >> 1);
    float *const hc2 = h0s + ((k2 & 1) * 32) * ROW + (k2 >> 1);

    const unsigned npairs = *q.pair_count;
    const unsigned ntiles = (npairs + 31) >> 5;

    // gather of one pair by the first 32 threads: pair record + x = la_i * la_j (uav.py:281)
    auto gather = [&](unsigned tile, uint2 &pr, float4 (&x)[3]) {
        const unsigned pi = tile * 32 + tid;
        pr = make_uint2(0, 0);
        x[0] = x[1] = x[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 32 && tile < ntiles && pi < npairs) pr = q.pairs[pi];
        if (tid < 32 && tile < ntiles && pi < npairs && pr.x != 0xFFFFFFFFu) {      // (skips the dummies of the rollout kernel's slot pool)
            const unsigned gi = pr.x, gj = pr.y;
            const float4 *oi = reinterpret_cast<const float4 *>(q.obs + (size_t)gi * 12);
            const float4 *oj = reinterpret_cast<const float4 *>(q.obs + (size_t)gj * 12);
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float4 a = oi[v], b = oj[v];
                x[v] = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
            }
        }
    };

    uint2 pr_cur, pr_next;
    float4 x_next[3];
    {
        float4 x0[3];
        gather(blockIdx.x, pr_cur, x0);
        if (tid < 32) {
            float4 *dst = reinterpret_cast<float4 *>(xs + tid * 12);
            dst[0] = x0[0]; dst[1] = x0[1]; dst[2] = x0[2];
        }
    }
    gather(blockIdx.x + gridDim.x, pr_next, x_next);
    __syncthreads();

    int cur = 0;
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ---- branch layers of this tile, from xs[cur]
        const float *xt = xs + cur * 32 * 12;
        float wc[5], wo[4], wb[3], bc, bo, bb;
        {
            const float4 *wq = reinterpret_cast<const float4 *>(bw + o * 20);
            const float4 w0 = wq[0], w1 = wq[1], w2q = wq[2], w3 = wq[3];
            wc[0] = w0.x; wc[1] = w0.y; wc[2] = w0.z; wc[3] = w0.w; wc[4] = w1.x; bc = w1.y;
            wo[0] = w1.z; wo[1] = w1.w; wo[2] = w2q.x; wo[3] = w2q.y; bo = w2q.z;
            wb[0] = w2q.w; wb[1] = w3.x; wb[2] = w3.y; bb = w3.z;
        }
#pragma unroll 4
        for (int pp = 0; pp < PPT; ++pp) {
            const int pr_i = phalf * PPT + pp;
            const float4 *xp = reinterpret_cast<const float4 *>(xt + pr_i * 12);
            const float4 xa = xp[0], xb = xp[1], xc = xp[2];
            float c = bc;
            c = fmaf(wc[0], xa.x, c); c = fmaf(wc[1], xa.y, c); c = fmaf(wc[2], xa.z, c);
            c = fmaf(wc[3], xa.w, c); c = fmaf(wc[4], xb.x, c);
            float ob = bo;
            ob = fmaf(wo[0], xb.y, ob); ob = fmaf(wo[1], xb.z, ob); ob = fmaf(wo[2], xb.w, ob);
            ob = fmaf(wo[3], xc.x, ob);
            float bs = bb;
            bs = fmaf(wb[0], xc.y, bs); bs = fmaf(wb[1], xc.z, bs); bs = fmaf(wb[2], xc.w, bs);
            hc0[pr_i * ROW] = fmaxf(c, 0.0f);
            hc1[pr_i * ROW] = fmaxf(ob, 0.0f);
            hc2[pr_i * ROW] = fmaxf(bs, 0.0f);
        }
        // next tile's inputs go to the other xs buffer; the one after that is fetched under the MFMAs
        if (tid < 32) {
            float4 *dst = reinterpret_cast<float4 *>(xs + (cur ^ 1) * 32 * 12 + tid * 12);
            dst[0] = x_next[0]; dst[1] = x_next[1]; dst[2] = x_next[2];
        }
        pr_cur = pr_next;
        gather(tile + 2 * gridDim.x, pr_next, x_next);
        __syncthreads();                                    // activation tile and xs[cur ^ 1] complete

        // ---- fc1 (+ folded bn1) on the matrix cores: [32 pairs x 3H] x [3H x 32 cols]
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias1;
        const float4 *arow = reinterpret_cast<const float4 *>(h0s + (kh * 32 + (lane & 31)) * ROW);
#pragma unroll
        for (int t4 = 0; t4 < KH / 4; ++t4) {
            const float4 a4 = arow[t4];                     // A[pair][k = 2t + kh], t = 4 t4 .. 4 t4 + 3
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, breg[4 * t4 + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, breg[4 * t4 + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, breg[4 * t4 + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, breg[4 * t4 + 3], acc, 0, 0, 0);
        }

        // ---- ReLU, fc2 (PMINet.py:60-61): per row, sum over this block's 32 columns in a fixed
        //      butterfly order (bitwise reproducible), then over the column blocks in order
        float *pc = part + cur * NW * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = half_wave_sum(fmaxf(acc[r], 0.0f) * wout);
            // C/D layout of 32x32 MFMA: row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
            if ((lane & 31) == 16) pc[w * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh] = v;
        }
        __syncthreads();                                    // partials complete; everyone is past the MFMA reads
        if (tid < 32 && tile * 32 + tid < npairs) {
            float sc = b2;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) sc += pc[ww * 32 + tid];
            q.scores[tile * 32 + tid] = sc;                 // the pair's own slot: s_ij = s_ji is stored once
        }
        cur ^= 1;   // `part` and `xs` are double-buffered by tile parity: two barriers per tile suffice
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pmi_score_x6_kernel<H>: the same scorer with the 3H x H layer on the BF16 matrix cores at fp32 accuracy.
// gfx950 has no reduced-precision-but-wide fp32 MFMA (no xf32); its fp32-input MFMA runs at 1/16 of the bf16 rate.
// An fp32 value is exactly the sum of three bf16 values (8 + 8 + 8 significant bits, split by truncation:
// x = hi + mid + lo with hi = x & 0xFFFF0000, mid likewise of the exact remainder x - hi, ...), so
//     x * w = xh*wh + (xh*wm + xm*wh) + (xh*wl + xl*wh + xm*wm) + O(2^-24 |x w|)
// -- six bf16 MFMAs (v_mfma_f32_32x32x16_bf16: products exact, fp32 accumulate) replace eight fp32 ones of the same
// tile at 1/16 the cycles each: 2.67x the fp32 matrix rate, with an error below that of an fp32 fmaf chain
// (3.7e-6 against 1.1e-5 max over 384-term sums of O(1) data; tests/test_host_cpu.py::test_x6_split_error_model_on_adversarial_sums).  Operands cannot overflow or
// lose range: bf16 has fp32's exponent.
//
// Weight-stationary as above, but the three planes of a wavefront's 3H x 32 slice are 2.25 H registers (288 at
// H = 128): one workgroup per CU, one wavefront per SIMD, 512 registers per lane (the planes are MFMA-only operands
// and live in the accumulation-register half of the file).  The branch layers produce two adjacent k per thread,
// split them on the VALU (v_and / v_sub / v_perm: 5.5 instructions per activation) and write the three planes of
// the 32 x 3H activation tile to LDS in A-operand order, rows padded to a 16-byte-odd pitch (conflict-free
// ds_read_b128 fragments).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
// Workgroup barrier that orders LDS traffic only (outstanding global loads stay in flight; the compiler still waits for
// them where their registers are first used).
#define UAVTRACK_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N - 1>).  (The staged main loop below indexes register
// arrays and picks pieces of work by loop position; `#pragma unroll` leaves inner loops whose bounds depend on an outer
// induction variable rolled, and the arrays then land in scratch.)
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// Pieces of the three-way split of a float pair: its bf16 (truncated) parts as floats, those parts packed as
// (low half: .x, high half: .y), and the exact remainder v - hi.
// Everything beside the MFMAs is SCALAR fp32 on purpose: a v_pk_fma_f32 / v_pk_add_f32 issued in an MFMA's shadow costs
// ~11 cycles more than the two plain instructions it replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs"),
// and this file is built with -fno-slp-vectorize so that the compiler does not re-pack them.
constexpr __host__ __device__ int min_c(int a, int b) { return a < b ? a : b; }
struct f2 { float x, y; };
__device__ __forceinline__ f2 bf16_part(f2 v)
{
    return f2{__uint_as_float(__float_as_uint(v.x) & 0xFFFF0000u), __uint_as_float(__float_as_uint(v.y) & 0xFFFF0000u)};
}
__device__ __forceinline__ unsigned pack_hi16(f2 v)
{
    return __builtin_amdgcn_perm(__float_as_uint(v.y), __float_as_uint(v.x), 0x07060302u);
}
__device__ __forceinline__ f2 sub2(f2 v, f2 hi) { return f2{v.x - hi.x, v.y - hi.y}; }
__device__ __forceinline__ f2 relu2(f2 v) { return f2{fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f)}; }
__device__ __forceinline__ f2 fma2(f2 w, float x, f2 c) { return f2{fmaf(w.x, x, c.x), fmaf(w.y, x, c.y)}; }

template <int H>
__global__ void __launch_bounds__(H * 2, 1) pmi_score_x6_kernel(const PmiParams q)
{
    // launched behind pmi_score_t3_kernel as its wide-range stand-by: nothing to do unless that kernel met an operand
    // beyond f16's range (uniform over the grid)
    if (q.gate && *q.gate == 0u) return;
    constexpr int K = 3 * H;             // fc1 input width
    constexpr int KS = K / 16;           // MFMA k-steps (32x32x16)
    constexpr int PITCH = K * 2 + 16;    // bytes per activation row of one plane
    constexpr int PLANE = 32 * PITCH;    // bytes per plane of a tile
    constexpr int NW = H / 32;           // wavefronts = column blocks
    constexpr int NT = NW * 64;          // threads = 2 H
    constexpr int OP = H / 2;            // adjacent-output pairs per branch
    constexpr int PG = NT / OP;          // pair groups (4)
    constexpr int PPT = 32 / PG;         // pairs per thread in the branch layers (8)
    static_assert(KS >= PPT, "every produced pair needs at least one k-step to hide behind");

    __shared__ float4 lds4[(2 * 3 * PLANE + (2 * 32 * 12 + 2 * NW * 64 + NW * 96) * 4) / 16 + 2];
    // H = 64 runs two workgroups per CU (launch_pmi_score): both must fit the CU's 160 KiB of LDS
    static_assert(H != 64 || 2 * sizeof(lds4) <= 160 * 1024, "two H = 64 workgroups no longer share a CU: shrink xs / part / sink");
    unsigned char *aplanes = reinterpret_cast<unsigned char *>(lds4);               // [2 tiles][3 planes][32 pairs][PITCH]
    float *xs = reinterpret_cast<float *>(aplanes + 2 * 3 * PLANE);                // [2 tiles][32 pairs][12]
    float *part = xs + 2 * 32 * 12;                                                // [2 tiles][NW][2 column halves][32]
    float *sink = part + 2 * NW * 64;                                              // [NW][96] where non-writer lanes' stores go

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int col = w * 32 + (lane & 31);
    const int kh = lane >> 5;

    // ---- stationary operands: the three bf16 planes of fc1's slice, in MFMA B-operand order (pack_pmi_x6)
    const float *W1 = q.blob + 15 * H;
    const float *b1 = W1 + (size_t)K * H;
    const float *w2 = b1 + H;
    const float b2 = w2[H];
    u32x4 Bh[KS], Bm[KS], Bl[KS];
    {
        const u32x4 *bp = reinterpret_cast<const u32x4 *>(q.x6) + (size_t)w * 3 * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Bh[s] = bp[(0 * KS + s) * 64];
            Bm[s] = bp[(1 * KS + s) * 64];
            Bl[s] = bp[(2 * KS + s) * 64];
        }
        // hi and mid planes into the accumulation-register file for good (192 of its 256): the MFMAs read them from there
        // directly -- left to itself the allocator parks them there too, but copies each fragment back before its use
#pragma unroll
        for (int s = 0; s < KS; ++s) { asm volatile("" : "+a"(Bh[s])); asm volatile("" : "+a"(Bm[s])); }
    }
    const float bias1 = b1[col], wout = w2[col];
    f32x16 biasv;                            // fc1's bias as the first MFMA's srcC: no per-tile accumulator initialisation
#pragma unroll
    for (int r = 0; r < 16; ++r) biasv[r] = bias1;
    asm volatile("" : "+a"(biasv));

    // Branch layers (PMINet.py:50-55): thread (o2, pg) owns outputs 2 o2 and 2 o2 + 1 of each of the three branches
    // for PPT of the tile's 32 pairs; its 30 folded weights and 6 biases stay in registers.
    const int o2 = tid % OP, pg = tid / OP;
    f2 wc[5], wo[4], wb[3], bc, bo, bb;
    {
        const int o = 2 * o2;
#pragma unroll
        for (int v = 0; v < 5; ++v) wc[v] = f2{q.blob[v * H + o], q.blob[v * H + o + 1]};
        bc = f2{q.blob[5 * H + o], q.blob[5 * H + o + 1]};
#pragma unroll
        for (int v = 0; v < 4; ++v) wo[v] = f2{q.blob[6 * H + v * H + o], q.blob[6 * H + v * H + o + 1]};
        bo = f2{q.blob[10 * H + o], q.blob[10 * H + o + 1]};
#pragma unroll
        for (int v = 0; v < 3; ++v) wb[v] = f2{q.blob[11 * H + v * H + o], q.blob[11 * H + v * H + o + 1]};
        bb = f2{q.blob[14 * H + o], q.blob[14 * H + o + 1]};
    }
    // where this thread's outputs live in a plane's row: concat order comm | obs | boundary_state (PMINet.py:58)
    const int arow0 = (pg * PPT) * PITCH + 4 * o2;
    const int afrag0 = (lane & 31) * PITCH + kh * 16;

    const unsigned npairs = *q.pair_count;
    const unsigned ntiles = (npairs + 31) >> 5;
    const unsigned G = gridDim.x;

    // Inputs of a tile, by the first 32 threads: pair record -> the two observations -> x = la_i * la_j (uav.py:281).
    // Two dependent trips to memory, so the main loop runs them as a pipeline one tile deep each: a record is
    // requested three tiles ahead, its observations two tiles ahead, and neither request is ever waited for in the
    // iteration that issued it (a wait here would hold every wavefront at the tile's barrier).
    auto load_rec = [&](unsigned tile, uint2 &pr, bool &ok) {
        const unsigned pi = tile * 32 + tid;
        ok = tid < 32 && tile < ntiles && pi < npairs;
        pr = make_uint2(0, 0);
        if (ok) pr = q.pairs[pi];
    };
    auto load_obs = [&](uint2 pr, bool ok, float4 (&a)[3], float4 (&b)[3]) {
        a[0] = a[1] = a[2] = b[0] = b[1] = b[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && pr.x != 0xFFFFFFFFu) {     // (tested HERE, where the record is needed anyway: 0xFFFFFFFF = a dummy of the rollout kernel's slot pool)
            const unsigned gi = pr.x, gj = pr.y;
            const float4 *oi = reinterpret_cast<const float4 *>(q.obs + (size_t)gi * 12);
            const float4 *oj = reinterpret_cast<const float4 *>(q.obs + (size_t)gj * 12);
#pragma unroll
            for (int v = 0; v < 3; ++v) { a[v] = oi[v]; b[v] = oj[v]; }
        }
    };
    auto stash = [&](int buf, const float4 (&a)[3], const float4 (&b)[3]) {
        if (tid < 32) {
            float4 *dst = reinterpret_cast<float4 *>(xs + buf * 32 * 12 + tid * 12);
#pragma unroll
            for (int v = 0; v < 3; ++v) dst[v] = make_float4(a[v].x * b[v].x, a[v].y * b[v].y, a[v].z * b[v].z, a[v].w * b[v].w);
        }
    };
    // One pair of the branch layers -- 12 packed FMAs, ReLU, three-way split, nine 4-byte LDS writes -- as a list of
    // kMicro atoms of one or two instructions, so that the main loop can deal them out evenly behind its MFMAs (see
    // there).  The inputs of the next pair are requested by atom kLoadAtom, once this pair's FMAs have consumed theirs.
    constexpr int kMicro = 36, kLoadAtom = 20;
    struct Prod { float4 x[3]; f2 v[3], hi[3]; };      // inputs; the three branches' output pairs / remainders; their bf16 parts
    auto load_x = [&](int xbuf, int pp, Prod &P) {
        const float4 *xp = reinterpret_cast<const float4 *>(xs + xbuf * 32 * 12 + (pg * PPT + pp) * 12);
        P.x[0] = xp[0]; P.x[1] = xp[1]; P.x[2] = xp[2];
    };
    auto micro = [&](Prod &P, auto opc, int xbuf, int abuf, int pp) {
        constexpr int op = decltype(opc)::value;
        unsigned char *dst = aplanes + abuf * 3 * PLANE + arow0 + pp * PITCH;
        if constexpr (op < 12) {
            // FMA number `op`, the three branches taking turns (a branch's chain is dependent: its next FMA comes two
            // atoms later).  Input i of x = (la_i * la_j)[0..11]: branch comm takes 0..4, obs 5..8, boundary 9..11.
            constexpr int br = op < 9 ? op % 3 : op < 11 ? op - 9 : 0;          // c o b  c o b  c o b  c o  c
            constexpr int kk = op < 9 ? op / 3 : op < 11 ? 3 : 4;               // position inside the branch's chain
            constexpr int i = br == 0 ? kk : br == 1 ? 5 + kk : 9 + kk;
            const float4 &f = P.x[i / 4];
            const float xi = i % 4 == 0 ? f.x : i % 4 == 1 ? f.y : i % 4 == 2 ? f.z : f.w;
            const f2 wgt = br == 0 ? wc[min_c(kk, 4)] : br == 1 ? wo[min_c(kk, 3)] : wb[min_c(kk, 2)];
            const f2 bias = br == 0 ? bc : br == 1 ? bo : bb;
            P.v[br] = fma2(wgt, xi, kk == 0 ? bias : P.v[br]);
        } else if constexpr (op < 15) {
            P.v[op - 12] = relu2(P.v[op - 12]);
        } else if constexpr (op < 33) {
            // planes 0 (hi) and 1 (mid): bf16 parts, packed store, exact remainder -- each step for the three branches
            // in turn, so that no atom waits for the one just before it
            constexpr int q = op - 15, pl = q / 9, step = (q % 9) / 3, br = q % 3;
            if constexpr (step == 0) P.hi[br] = bf16_part(P.v[br]);
            else if constexpr (step == 1) *reinterpret_cast<unsigned *>(dst + pl * PLANE + br * 2 * H) = pack_hi16(P.v[br]);
            else P.v[br] = sub2(P.v[br], P.hi[br]);
        } else if constexpr (op < 36) {
            constexpr int br = op - 33;     // plane 2 (lo): what is left, truncated
            *reinterpret_cast<unsigned *>(dst + 2 * PLANE + br * 2 * H) = pack_hi16(P.v[br]);
        }
        if constexpr (op == kLoadAtom) { if (pp + 1 < PPT) load_x(xbuf, pp + 1, P); }
    };
    auto produce_pair = [&](Prod &P, int xbuf, int abuf, int pp) {       // all of it at once (prologue)
        static_for<kMicro>([&](auto opc) { micro(P, opc, xbuf, abuf, pp); });
    };

    // ---- prologue: the first tile's activation planes; the second tile's inputs in xs[1]; the third's observations
    //      and the fourth's pair record in flight
    uint2 rec_n;
    bool rec_ok;
    float4 oa[3], ob[3];
    {
        load_rec(blockIdx.x, rec_n, rec_ok);
        load_obs(rec_n, rec_ok, oa, ob);
        stash(0, oa, ob);
        load_rec(blockIdx.x + G, rec_n, rec_ok);
        load_obs(rec_n, rec_ok, oa, ob);
        __syncthreads();
        Prod P0;
        load_x(0, 0, P0);
#pragma unroll 2
        for (int pp = 0; pp < PPT; ++pp) produce_pair(P0, 0, 0, pp);
        stash(1, oa, ob);
        load_rec(blockIdx.x + 2 * G, rec_n, rec_ok);
        load_obs(rec_n, rec_ok, oa, ob);
        load_rec(blockIdx.x + 3 * G, rec_n, rec_ok);
        __syncthreads();
    }

    // ReLU, fc2 (PMINet.py:60-61) of a finished tile, row r of this wavefront's 16: sum over the block's 32 columns in
    // a fixed butterfly order (bitwise reproducible).  Three pieces, so the main loop can hide them too.
    auto epi_piece = [&](const f32x16 &a, float (&ev)[2], auto ec, float *pc) {
        // piece e: step e % 6 of rows 2 (e / 6) and 2 (e / 6) + 1 -- two independent chains per piece.  The butterfly
        // stops at the 16-lane row: the two rows of a wavefront half keep separate partials (summed with the column
        // blocks at the end), which saves the row_bcast step (a v_mov_dpp + v_add pair per row).
        constexpr int e = decltype(ec)::value, st = e % 6;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = 2 * (e / 6) + u;
            // ReLU on the bit pattern (max as integers: negatives are negative, positives keep their order): one
            // instruction, where fmaxf on an MFMA result first re-canonicalises it
            if constexpr (st == 0) ev[u] = __int_as_float(max(__float_as_int(a[r]), 0)) * wout;
            else if constexpr (st == 1) ev[u] = dpp_add<0xB1>(ev[u]);
            else if constexpr (st == 2) ev[u] = dpp_add<0x4E>(ev[u]);
            else if constexpr (st == 3) ev[u] = dpp_add<0x141>(ev[u]);
            else if constexpr (st == 4) ev[u] = dpp_add<0x140>(ev[u]);
            else {
                // C/D layout of 32x32 MFMA: row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).  Every lane stores: `pc` is
                // the writer lane's slot base and a private sink for the others (no exec-mask juggling between the MFMAs).
                pc[(r & 3) + 8 * (r >> 2)] = ev[u];
            }
        }
    };
    auto partial_base = [&](int buf) {       // writer: lane 0 of each 16-lane row
        return (lane & 15) == 0 ? part + buf * NW * 64 + w * 64 + ((lane >> 4) & 1) * 32 + 4 * kh : sink + w * 96 + lane;
    };
    // (by the LAST wavefront's first 32 lanes: the first wavefront already carries the input pipeline, and whoever is
    // late holds the others at the tile barrier)
    const int ft = tid - (NW - 1) * 64;
    auto final_sum = [&](unsigned tile, const float *pc) {      // over the column blocks, in order
        if (ft >= 0 && ft < 32 && tile * 32 + ft < npairs) {
            float sc = b2;
#pragma unroll
            for (int ww = 0; ww < 2 * NW; ++ww) sc += pc[ww * 32 + ft];
            q.scores[tile * 32 + ft] = sc;                      // the pair's own slot: s_ij = s_ji is stored once
        }
    };

    int cur = 0;
    bool have_prev = false;
    f32x16 accp;                             // the previous tile's accumulators: their epilogue runs under this tile's MFMAs
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[r] = 0.0f;
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += G) {
        // ---- fc1 (+ folded bn1) of this tile: [32 pairs x 3H] x [3H x 32 cols], six bf16 MFMAs per k-step (small
        //      terms first), interleaved in program order with the branch layers of the NEXT tile: a wavefront issues
        //      in order and an MFMA holds its SIMD's issue for 8 of its 32 cycles, so the VALU / LDS instructions
        //      placed between two MFMAs of the one accumulation chain run in the shadow of the first
        f32x16 acc = biasv;
        const unsigned char *afrag = aplanes + cur * 3 * PLANE + afrag0;
        auto load_a = [&](int s, u32x4 &h, u32x4 &m, u32x4 &l) {
            h = *reinterpret_cast<const u32x4 *>(afrag + 0 * PLANE + s * 32);
            m = *reinterpret_cast<const u32x4 *>(afrag + 1 * PLANE + s * 32);
            l = *reinterpret_cast<const u32x4 *>(afrag + 2 * PLANE + s * 32);
        };
        Prod P;
        load_x(cur ^ 1, 0, P);
        float ev[2] = {0.0f, 0.0f};
        float *pcp = partial_base(cur ^ 1);              // the previous tile's partial scores (this lane's store base)
        u32x4 fh, fm, fl, gh, gm, gl;            // this k-step's fragments, the next one's
        load_a(0, fh, fm, fl);
        __builtin_amdgcn_sched_barrier(0);
        static_for<KS>([&](auto sc) {
            // k-step s belongs to pair pp's share; its six MFMA slots carry the micro-ops [lo, hi) of that pair
            constexpr int s = decltype(sc)::value;
            constexpr int pp = s * PPT / KS;
            constexpr int s0 = (pp * KS + PPT - 1) / PPT, s1 = ((pp + 1) * KS + PPT - 1) / PPT;   // first k-step of pp, of pp + 1
            constexpr int nslot = 6 * (s1 - s0);
            static_for<6>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int slot = 6 * (s - s0) + t;
                // Which atoms ride behind this MFMA.  Three k-steps per pair (H = 128): a hand-balanced deal -- the slot
                // that also issues the next k-step's three fragment reads (t = 0) takes two atoms, the two that carry an
                // epilogue piece (t = 1, 4) one, the others three (two in the pair's last k-step) -- so that every gap holds five to seven
                // plain instructions: a gap shorter than the MFMA's 32 cycles idles the wavefront, a longer one idles
                // the matrix core.  Other widths: evenly by count.
                constexpr bool kTuned = KS == 3 * PPT;
                // atoms per slot: k-steps 0 and 1 of a pair {2,1,3,3,1,3} (13 each), k-step 2 {2,1,2,2,1,2} (the last 10)
                constexpr int pre[7] = {0, 2, 3, 6, 9, 10, 13}, pre2[7] = {26, 28, 29, 31, 33, 34, 36};
                constexpr int lo = !kTuned ? slot * kMicro / nslot : slot < 12 ? (slot / 6) * 13 + pre[slot % 6] : pre2[slot - 12];
                constexpr int hi = !kTuned ? (slot + 1) * kMicro / nslot : slot < 12 ? (slot / 6) * 13 + pre[slot % 6 + 1] : pre2[slot - 11];
                const bf16x8 a = as_bf16x8(t == 1 ? fl : (t == 2 || t == 4) ? fm : fh);
                const bf16x8 bq = as_bf16x8(t == 0 ? Bl[s] : (t == 2 || t == 3) ? Bm[s] : Bh[s]);
                if constexpr (s == 0 && t == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, biasv, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc, 0, 0, 0);
                if constexpr (t == 0 && s + 1 < KS) load_a(s + 1, gh, gm, gl);
                static_for<kMicro>([&](auto opc) {
                    if constexpr (decltype(opc)::value >= lo && decltype(opc)::value < hi) micro(P, opc, cur ^ 1, cur ^ 1, pp);
                });
                // the previous tile's epilogue: 48 pieces over the 6 KS slots
                constexpr int gslot = 6 * s + t;
                constexpr int elo = kTuned ? (t == 1 ? 2 * s : t == 4 ? 2 * s + 1 : 0) : gslot * 48 / (6 * KS);
                constexpr int ehi = kTuned ? (t == 1 || t == 4 ? elo + 1 : 0) : (gslot + 1) * 48 / (6 * KS);
                static_for<48>([&](auto ec) {
                    if constexpr (decltype(ec)::value >= elo && decltype(ec)::value < ehi) epi_piece(accp, ev, ec, pcp);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            fh = gh; fm = gm; fl = gl;
        });

        // xs[cur] fed this tile's branch layers during the previous iteration: free for the tile after the next
        stash(cur, oa, ob);                                   // x of tile + 2 G (observations requested an iteration ago)
        load_obs(rec_n, rec_ok, oa, ob);                      // tile + 3 G (record requested an iteration ago)
        load_rec(tile + 4 * G, rec_n, rec_ok);
        // next tile's planes, the previous tile's partials and xs[cur] are complete.  An LDS-only barrier: __syncthreads()
        // also drains the vector-memory counter, i.e. it would wait out the two requests the first wavefront has just
        // made -- a full trip to memory per tile, with every other wavefront parked at the barrier meanwhile.
        UAVTRACK_LDS_BARRIER();
        if (have_prev) final_sum(tile - G, part + (cur ^ 1) * NW * 64);
        accp = acc;
        have_prev = true;
        cur ^= 1;
    }
    if (have_prev) {                         // the last tile's epilogue has nothing left to hide behind
        float ev[2] = {0.0f, 0.0f};
        float *pcp = partial_base(cur ^ 1);
        static_for<48>([&](auto ec) { epi_piece(accp, ev, ec, pcp); });
        __syncthreads();
        // (the last tile of this workgroup: blockIdx.x + G * (its tile count - 1))
        const unsigned last = blockIdx.x + ((ntiles - 1 - blockIdx.x) / G) * G;
        final_sum(last, part + (cur ^ 1) * NW * 64);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// f16 helpers of pmi_score_t3_kernel.  f16 keeps 11 significant bits, so an fp32 value is a two-term sum to 2^-22:
//     x = xh + xl * 2^-11,   xh = f16(x),  xl = f16((x - xh) * 2^11)       (the remainder is exact in fp32; scaled so
// that it stays a NORMAL f16 number whatever the magnitude of x), and x * w = xh*wh + 2^-11 (xh*wl + xl*wh) + O(2^-22 |x w|):
// THREE v_mfma_f32_32x32x16_f16 per fp32 product (exact products, fp32 accumulation) where the bf16 split needs six.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
constexpr float kLoScale = 2048.0f;
// two floats -> one 32-bit word of two f16 (.x in the low half), round toward zero: ONE instruction for the pair; the
// remainder below is taken against the value that was really stored, so the rounding mode does not enter the result
__device__ __forceinline__ unsigned pk_f16(f2 v) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.x, v.y)); }
__device__ __forceinline__ f2 unpk_f16(unsigned w)
{
    const f16x2 h = __builtin_bit_cast(f16x2, w);
    return f2{(float)h.x, (float)h.y};
}
__device__ __forceinline__ f2 scale2(f2 v, float k) { return f2{v.x * k, v.y * k}; }
// ReLU of both halves of a packed f16 word, one instruction
__device__ __forceinline__ unsigned relu_h2(unsigned w)
{
    unsigned r;
    asm("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(w));
    return r;
}
// f16 pair of the exact remainders (a0 - hi.lo, a1 - hi.hi): a mixed-precision FMA per value (f16 operand * -1 + fp32 operand,
// rounded once to f16) -- no unpack, no subtract, no pack
__device__ __forceinline__ unsigned rem_h2(unsigned hi, float a0, float a1)
{
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(a0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(hi), "v"(a1));
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// pmi_score_t3_kernel<H>: the f16 x 3 scorer with the PAIRS ON THE LANES (every product transposed).  In the two kernels
// above a wavefront's VALU work beside the MFMAs -- branch layers, ReLU, operand split, DPP reductions -- outweighs the
// matrix pipe once the products are three per fp32 product.  Transposed, most of it goes away:
//   * layer 1 (12 -> 3H, PMINet.py:50-55) runs on the matrix cores too: H1^T[32 units x 32 pairs] = Wb^T[32 x 16] .
//     X^T[16 x 32], the bias riding as input 12 (x_12 = 1) -- three MFMAs per block of 32 units, three blocks (one per
//     branch) per wavefront and tile, instead of 24 scalar FMAs per pair and thread;
//   * a 32x32 accumulator tile holds, per lane, 16 rows of ONE column: with the pairs as columns a lane owns 16 hidden
//     units of its pair in groups of four adjacent ones, so ReLU and the f16 split stay in the lane and leave as two
//     ds_write_b64 per group into the activation planes (row = pair, unit index contiguous) -- the layout layer 2 reads;
//   * layer 2 (3H -> H) is the same MFMA with the operands swapped (weights as A, packed by pack_pmi_t3;
//     activations as B: the very LDS addresses the A fragments came from), which leaves H2^T in the accumulators: fc2
//     (PMINet.py:61) is then 16 FMAs down the lane's registers, no cross-lane reduction at all -- the two halves of a
//     wavefront and the column blocks are added by the thread that stores the score, as before.
// The instructions BESIDE the MFMAs set this kernel's time (one wavefront per SIMD: an MFMA gap hides ~5 of them,
// tools/microbench/mfma_fillers.hip; every further one costs its 4 cycles), and most of them were the operand split of
// the activations: combine two accumulators, ReLU, pack, unpack, subtract, scale, pack = 6 per value.  Block scaling
// removes the multiplications: the host folds a power of two S1 into the branch layers and T into fc1 (both exact)
// such that S1 * activation bound and T * max |w| sit just inside f16's range; a value of that size has a remainder
// x - f16(x) that is a NORMAL f16 number as it stands (an unscaled remainder would need a 2^11 factor to stay normal), for every value
// above 2^-18 of the bound -- smaller ones lose remainder bits, an absolute error below 2^-25 of the bound's scale, far
// under the fp32 rounding of the sums they enter (emulated in numpy against fp64: tests/test_host_cpu.py).  Then
//   hi = f16(acc) (toward zero), ReLU on the packed pair, lo = f16(acc - hi) by ONE mixed-precision FMA per value
//   (v_fma_mixlo/hi_f16: f16 operand, fp32 operand, f16 result), ReLU on the packed pair        = 2.5 per value;
// layer 1 accumulates its three products in one accumulator (a third plane, hi * 2^-11, pairs with the 2^11-scaled
// remainder of the inputs, whose magnitudes have no useful bound from below), layer 2 keeps two accumulators of equal
// scale S1 T; b1 enters as S1 T b1 and w2 as w2 / (S1 T).  Range: uavtrack_set_pmi_weights (bf16 x 6 kernel otherwise).
template <int H>
__global__ void __launch_bounds__(H * 2, 1) pmi_score_t3_kernel(const PmiParams q)
{
    constexpr int K = 3 * H;             // fc1 input width
    constexpr int KS = K / 16;           // MFMA k-steps of layer 2 (32x32x16)
    constexpr int PITCH = K * 2 + 16;    // bytes per activation row of one plane
    constexpr int PLANE = 32 * PITCH;    // bytes per plane of a tile
    constexpr int NP = 2;                // planes: hi, lo * 2^11
    constexpr int NW = H / 32;           // wavefronts = column blocks of layer 2 = unit blocks per branch of layer 1
    constexpr int XROW = 16;             // floats per row of the x staging buffer: x_0..11, 1, 0, 0, 0
    constexpr int kDuty0 = 2, kDutyStep = 2, kDutyItems = 12;     // the twelve items of the input duty go behind MFMAs 2, 4, .. 24 of the tile
    // the tile barrier stands behind this many of the tile's MFMAs (see tile_body; H = 64 -- two workgroups per CU cover each other's barrier -- measured best with 4 behind it, 128 indifferent between 2 and 8)
    constexpr int kBarrierSlot = 3 * (3 * H / 16) - (H <= 64 ? 4 : 6);

    __shared__ float4 lds4[(2 * NP * PLANE + (2 * 32 * XROW + 2 * NW * 64) * 4) / 16 + 2];
    static_assert(H != 64 || 2 * sizeof(lds4) <= 160 * 1024, "two H = 64 workgroups no longer share a CU");
    static_assert(kDuty0 + (kDutyItems - 1) * kDutyStep < kBarrierSlot && kBarrierSlot > 3 * (3 * H / 16 - 3),
                  "LDS work behind the tile barrier / a fragment request behind it");
    unsigned char *aplanes = reinterpret_cast<unsigned char *>(lds4);               // [2 tiles][2 planes][32 pairs][PITCH]
    float *xs = reinterpret_cast<float *>(aplanes + 2 * NP * PLANE);               // [2 tiles][32 pairs][XROW]
    float *part = xs + 2 * 32 * XROW;                                              // [2 tiles][NW][64 lanes]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: the duty / final-sum tests below are wave-uniform branches)
    const int pr32 = lane & 31;          // the pair (column) this lane holds in every accumulator tile
    const int kh = lane >> 5;

    // ---- stationary operands
    const float *b1 = q.blob + 15 * H + (size_t)K * H;
    const float *w2 = b1 + H;
    const float b2 = w2[H];
    u32x4 Ah[KS], Al[KS];                 // layer 2: W1^T fragments (pack_pmi_t3)
    {
        const u32x4 *bp = reinterpret_cast<const u32x4 *>(q.t3) + (size_t)w * NP * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Ah[s] = bp[(0 * KS + s) * 64];
            Al[s] = bp[(1 * KS + s) * 64];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) { asm volatile("" : "+a"(Ah[s])); asm volatile("" : "+a"(Al[s])); }
    }
    // layer 1: this wavefront's block of 32 units of each branch (pack_pmi_l1): S1 * w as hi, unscaled remainder, hi * 2^-11
    constexpr int NP1 = 3;
    u32x4 L1h[3], L1l[3], L1s[3];
    {
        const u32x4 *lp = reinterpret_cast<const u32x4 *>(q.l1) + (size_t)w * 3 * NP1 * 64 + lane;
#pragma unroll
        for (int j = 0; j < 3; ++j) { L1h[j] = lp[(j * NP1 + 0) * 64]; L1l[j] = lp[(j * NP1 + 1) * 64]; L1s[j] = lp[(j * NP1 + 2) * 64]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { asm volatile("" : "+a"(L1h[j])); asm volatile("" : "+a"(L1l[j])); asm volatile("" : "+a"(L1s[j])); }
    }
    // accumulator row r of a lane is column (unit) m(r) = (r & 3) + 8 (r >> 2) + 4 kh of the wavefront's 32
    f32x16 biasv, w2r;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * kh;
        biasv[r] = b1[m] * q.t3_scale;          // (powers of two: exact)
        w2r[r] = w2[m] * q.t3_inv_scale;
    }

    // where this lane's units of layer-1 block j land in an activation row: byte (j H + 32 w + 16 kh + r) * 2
    // A lane's 16 units of a block -- accumulator rows r, units (r & 3) + 8 (r >> 2) + 4 kh -- are stored CONTIGUOUSLY, at
    // k-positions 16 kh + r of the wavefront's 32: two 16-byte stores per plane and block (conflict-free: eight lanes x four
    // dwords cover the 32 banks at this pitch; the 8-byte stores of unit order were 2-way).  pack_pmi_t3 lays the fc1 rows
    // out in the same order; the MFMA does not care which unit sits at which k.
    const int obase = pr32 * PITCH + 2 * (32 * w + 16 * kh);
    const int bfrag0 = pr32 * PITCH + kh * 16;           // B fragments of layer 2: row = pair, 8 consecutive k per half-wave
    const int xoff = pr32 * XROW + kh * 8;               // this lane's eight inputs of the layer-1 B operand

    const unsigned npairs = *q.pair_count;
    if (npairs == 0) return;             // (uniform over the grid)
    const unsigned ntiles = (npairs + 31) >> 5;
    const unsigned G = gridDim.x;

    // ---- the tile inputs.  Both half-wavefronts carry the same 32 pairs (lanes l and l + 32 request the same addresses --
    // one transaction -- and store the same values) and indices are clamped rather than masked, so this is straight-line
    // code for all 64 lanes that the main loop can deal into its MFMA gaps without touching EXEC.
    auto load_rec = [&](unsigned tile) -> uint2 {
        return q.pairs[min(tile * 32 + (unsigned)pr32, npairs - 1)];      // (past the end: a valid record whose score nobody stores)
    };
    struct ObsAddr { const float4 *oi, *oj; };
    auto obs_addr = [&](uint2 pr) -> ObsAddr {
        const bool dummy = pr.x == 0xFFFFFFFFu;                            // a dummy of the rollout kernel's slot pool: any row will do
        const unsigned gi = dummy ? 0u : pr.x, gj = dummy ? 0u : pr.y;
        return ObsAddr{reinterpret_cast<const float4 *>(q.obs + (size_t)gi * 12), reinterpret_cast<const float4 *>(q.obs + (size_t)gj * 12)};
    };
    auto prod4 = [](const float4 &a, const float4 &b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); };   // x = la_i * la_j (uav.py:281)
    auto xs_row = [&](int buf) { return reinterpret_cast<float4 *>(xs + buf * 32 * XROW + pr32 * XROW); };
    // f16 range watch (see PmiParams): the largest |x| of each branch's inputs -- comm x_0..4, obs x_5..8, boundary x_9..11
    auto max3a = [](float a, float b, float c) { return fmaxf(fmaxf(fabsf(a), fabsf(b)), fabsf(c)); };
    auto range_raise = [&](float rc, float ro, float rb) {
        if (fmaxf(fmaxf(rc * q.rng_inv[0], ro * q.rng_inv[1]), rb * q.rng_inv[2]) > 1.0f) *q.range_flag = 1u;
    };

    // ---- the producer of one tile's activation planes, as a list of items of one or two instructions each that the main
    //      loop deals out behind its MFMAs: X (operand of layer 1 from the staged inputs), M (its nine MFMAs), P (ReLU,
    //      split and stores of a block), in the order X M0 M1 P0 M2 P1 P2 -- a block's MFMAs are issued a block ahead of
    //      the items that read their result, and at most two blocks' accumulators are live.
    struct Prod {
        float4 xa, xb;                    // the lane's eight inputs
        unsigned xh[4], xl[4];            // ... as f16 pairs: hi plane, lo plane (B operand of layer 1)
        f32x16 ah[2];                     // accumulators of two blocks in flight (block j uses slot j & 1)
        f2 t[2];                          // remainder of the input pair in flight
        unsigned pk[4], lo[4];            // the eight values in flight: hi words, remainder words
    };
    constexpr int NX = 2 + 4 * 5;                         // two loads; per input pair: pack, convert back, subtract, scale, pack
    constexpr int NPB = 2 * 12;                           // per block: per half of the lane's 16 units twelve items (20 VALU, 2 stores)
    constexpr int I_M0 = NX, I_M1 = I_M0 + 3, I_P0 = I_M1 + 3, I_M2 = I_P0 + NPB, I_P1 = I_M2 + 3, I_P2 = I_P1 + NPB;
    constexpr int NITEM = I_P2 + NPB;
    auto post_item = [&](Prod &P, auto jc, auto ic, int abuf) {
        constexpr int j = decltype(jc)::value, i = decltype(ic)::value, sl = j & 1;
        constexpr int hf = i / 12, st = i % 12;          // half of the lane's 16 units (eight values = one 16-byte store per plane)
        unsigned char *dst = aplanes + abuf * NP * PLANE + obase + 2 * j * H + 16 * hf;
        auto a = [&](int k) { return P.ah[sl][8 * hf + k]; };
        // S1 * z of eight units -> ReLU -> hi = f16 (toward zero), lo = f16(value - hi): hi is clamped first, so a negative
        // value leaves hi = 0 and a negative remainder, which its own clamp removes
        if constexpr (st == 0) { P.pk[0] = pk_f16(f2{a(0), a(1)}); P.pk[1] = pk_f16(f2{a(2), a(3)}); }
        else if constexpr (st == 1) { P.pk[2] = pk_f16(f2{a(4), a(5)}); P.pk[3] = pk_f16(f2{a(6), a(7)}); }
        else if constexpr (st == 2) { P.pk[0] = relu_h2(P.pk[0]); P.pk[1] = relu_h2(P.pk[1]); }
        else if constexpr (st == 3) { P.pk[2] = relu_h2(P.pk[2]); P.pk[3] = relu_h2(P.pk[3]); }
        else if constexpr (st == 4) *reinterpret_cast<uint4 *>(dst) = make_uint4(P.pk[0], P.pk[1], P.pk[2], P.pk[3]);
        else if constexpr (st >= 5 && st <= 8) P.lo[st - 5] = rem_h2(P.pk[st - 5], a(2 * (st - 5)), a(2 * (st - 5) + 1));
        else if constexpr (st == 9) { P.lo[0] = relu_h2(P.lo[0]); P.lo[1] = relu_h2(P.lo[1]); }
        else if constexpr (st == 10) { P.lo[2] = relu_h2(P.lo[2]); P.lo[3] = relu_h2(P.lo[3]); }
        else *reinterpret_cast<uint4 *>(dst + PLANE) = make_uint4(P.lo[0], P.lo[1], P.lo[2], P.lo[3]);
    };
    auto mfma_item = [&](Prod &P, auto jc, auto tc) {
        constexpr int j = decltype(jc)::value, t = decltype(tc)::value, sl = j & 1;
        const u32x4 xh = {P.xh[0], P.xh[1], P.xh[2], P.xh[3]}, xl = {P.xl[0], P.xl[1], P.xl[2], P.xl[3]};
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
        // S1 (w . x + b) in ONE accumulator, small terms first: wl xh, (wh 2^-11)(xl 2^11), wh xh
        if constexpr (t == 0) P.ah[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(L1l[j]), as_f16x8(xh), zero, 0, 0, 0);
        else if constexpr (t == 1) P.ah[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(L1s[j]), as_f16x8(xl), P.ah[sl], 0, 0, 0);
        else P.ah[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(L1h[j]), as_f16x8(xh), P.ah[sl], 0, 0, 0);
    };
    auto item = [&](Prod &P, auto ic, int xbuf, int abuf) {
        constexpr int I = decltype(ic)::value;
        if constexpr (I == 0) P.xa = *reinterpret_cast<const float4 *>(xs + xbuf * 32 * XROW + xoff);
        else if constexpr (I == 1) P.xb = *reinterpret_cast<const float4 *>(xs + xbuf * 32 * XROW + xoff + 4);
        else if constexpr (I < NX) {
            constexpr int pi = (I - 2) / 5, st = (I - 2) % 5;      // input pair pi = inputs 2 pi, 2 pi + 1 of this lane's eight
            const float4 &f = pi < 2 ? P.xa : P.xb;
            const f2 v = (pi & 1) ? f2{f.z, f.w} : f2{f.x, f.y};
            if constexpr (st == 0) P.xh[pi] = pk_f16(v);
            else if constexpr (st == 1) P.t[0] = unpk_f16(P.xh[pi]);
            else if constexpr (st == 2) P.t[0] = sub2(v, P.t[0]);
            else if constexpr (st == 3) P.t[0] = scale2(P.t[0], kLoScale);
            else P.xl[pi] = pk_f16(P.t[0]);
        }
        else if constexpr (I < I_M1) mfma_item(P, std::integral_constant<int, 0>{}, std::integral_constant<int, I - I_M0>{});
        else if constexpr (I < I_P0) mfma_item(P, std::integral_constant<int, 1>{}, std::integral_constant<int, I - I_M1>{});
        else if constexpr (I < I_M2) post_item(P, std::integral_constant<int, 0>{}, std::integral_constant<int, I - I_P0>{}, abuf);
        else if constexpr (I < I_P1) mfma_item(P, std::integral_constant<int, 2>{}, std::integral_constant<int, I - I_M2>{});
        else if constexpr (I < I_P2) post_item(P, std::integral_constant<int, 1>{}, std::integral_constant<int, I - I_P1>{}, abuf);
        else post_item(P, std::integral_constant<int, 2>{}, std::integral_constant<int, I - I_P2>{}, abuf);
    };

    // ---- prologue: the first tile's activation planes; the second tile's inputs in xs[1]; the third's observations
    //      and the fourth's pair record in flight
    // The tile inputs (pair record -> the two observations -> x) are two dependent trips to memory, and the observation
    // rows are a random gather from hundreds of megabytes: ~2 us under load, a whole tile time of this kernel.  The duty
    // therefore rotates over the wavefronts -- the inputs of iteration j are staged by wavefront j mod NW, which requested
    // the observations NW iterations and the record 2 NW iterations ahead: every request has NW tile times to arrive.
    uint2 rec_n;
    float4 oa[3], ob[3];
    auto tile_of = [&](unsigned j) { return blockIdx.x + j * G; };
    {
        auto fetch_obs = [&]() {
            const ObsAddr A = obs_addr(rec_n);
#pragma unroll
            for (int v = 0; v < 3; ++v) { oa[v] = A.oi[v]; ob[v] = A.oj[v]; }
        };
        if (w == 0) {                                      // iterations 0 and 1: staged here, synchronously
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                rec_n = load_rec(tile_of(b));
                fetch_obs();
                float4 *dst = xs_row(b);
#pragma unroll
                for (int v = 0; v < 3; ++v) dst[v] = prod4(oa[v], ob[v]);
                dst[3] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);            // input 12 = 1 (the bias rides on it), 13..15 = 0: written once
                const float4 x0 = prod4(oa[0], ob[0]), x1 = prod4(oa[1], ob[1]), x2 = prod4(oa[2], ob[2]);
                range_raise(fmaxf(max3a(x0.x, x0.y, x0.z), fmaxf(fabsf(x0.w), fabsf(x1.x))),
                            fmaxf(max3a(x1.y, x1.z, x1.w), fabsf(x2.x)), max3a(x2.y, x2.z, x2.w));
            }
        }
        const unsigned j0 = 2u + (unsigned)((w + NW - 2 % NW) % NW);     // this wavefront's first duty: the iteration j0 >= 2 with j0 mod NW == w
        rec_n = load_rec(tile_of(j0));
        fetch_obs();
        rec_n = load_rec(tile_of(j0 + NW));
        __syncthreads();
        Prod P0;
        static_for<NITEM>([&](auto ic) { item(P0, ic, 0, 0); });
        __syncthreads();
    }
    u32x4 nfh[2], nfl[2];                    // the B fragments of the coming tile's k-steps 0 and 1
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        nfh[k] = *reinterpret_cast<const u32x4 *>(aplanes + bfrag0 + 0 * PLANE + k * 32);
        nfl[k] = *reinterpret_cast<const u32x4 *>(aplanes + bfrag0 + 1 * PLANE + k * 32);
    }

    // fc2 (PMINet.py:60-61) of a finished tile: ReLU(H2) . w2 down the lane's 16 rows; 17 items
    auto epi_item = [&](const f32x16 &ah, const f32x16 &al, float &sum, auto ec, float *pc) {
        constexpr int e = decltype(ec)::value;
        if constexpr (e < 16) {
            const float v = __int_as_float(max(__float_as_int(al[e] + ah[e]), 0));     // (both accumulators carry S1 * T; w2r its reciprocal)
            sum = e == 0 ? v * w2r[0] : fmaf(v, w2r[e], sum);
        } else {
            pc[lane] = sum;
        }
    };
    auto final_sum = [&](unsigned tile, const float *pc) {      // over the two half-wavefronts of every column block, in order
        if (tile * 32 + pr32 < npairs) {                        // (both halves of the wavefront compute and store the same value)
            float sc = b2;
#pragma unroll
            for (int ww = 0; ww < 2 * NW; ++ww) sc += pc[ww * 32 + pr32];
            q.scores[tile * 32 + pr32] = sc;
        }
    };

    int cur = 0;
    unsigned it = 0;
    // One tile: layer 2 of iteration `it` into (acch, accl) -- 3 KS MFMAs, the "slots" -- and dealt behind them: the producer of
    // iteration it + 1's planes, fc2 of iteration it - 1 (accph, accpl), on the wavefront whose turn it is (DUTY) the inputs
    // of iterations it + 2 .. it + 2 + 2 NW, and on the last wavefront, behind the barrier, the final sum of iteration it - 1.
    auto tile_body = [&](auto dutyc, f32x16 &acch, f32x16 &accl, const f32x16 &accph, const f32x16 &accpl) {
        constexpr bool DUTY = decltype(dutyc)::value;
        acch = biasv;
#pragma unroll
        for (int r = 0; r < 16; ++r) accl[r] = 0.0f;
        const unsigned char *bfrag = aplanes + cur * NP * PLANE + bfrag0;
        auto load_b = [&](int s, u32x4 &h, u32x4 &l) {
            h = *reinterpret_cast<const u32x4 *>(bfrag + 0 * PLANE + s * 32);
            l = *reinterpret_cast<const u32x4 *>(bfrag + 1 * PLANE + s * 32);
        };
        Prod P;
        float esum = 0.0f;
        float *pcp = part + (cur ^ 1) * NW * 64 + w * 64;         // the previous tile's partial scores of this wavefront
        // B fragments of k-step s >= 2 live in ring slot s % 3 and are requested TWO k-steps (six MFMAs) ahead: a wavefront's
        // LDS operations complete in order, so a read queues behind the producer's stores of all four wavefronts, and one
        // k-step of lead (~100 cycles) did not cover that.  Those of k-steps 0 and 1 (nfh, nfl) were requested right behind
        // the previous tile's barrier.
        constexpr int RING = 3, AHEAD = 2;        // (rings of 4 and 5 slots, requests 3 or 4 k-steps ahead: all within 0.3 %)
        static_assert(AHEAD >= 2 && AHEAD < RING + 1 && AHEAD <= 4, "fragment ring");
        u32x4 fh[RING], fl[RING];
        float4 dm;                            // duty: the product row in flight
        float drc = 0.0f, dro = 0.0f, drb = 0.0f;   // ... and the largest |x| per branch so far (f16 range watch)
        ObsAddr dA;
        float4 *const dxs = xs_row(cur);      // xs[cur] fed the producer during the previous iteration: free for iteration it + 2
        static_for<KS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            static_for<3>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int slot = 3 * s + t;
                // S1 T H2^T += (T W1)^T (S1 H1)^T, small terms first: lo += Wl Hh, lo += Wh Hl, hi += Wh Hh
                const f16x8 a = as_f16x8(t == 0 ? Al[s] : Ah[s]);
                const u32x4 bh = s < 2 ? nfh[s < 2 ? s : 0] : fh[s % RING], bl = s < 2 ? nfl[s < 2 ? s : 0] : fl[s % RING];
                const f16x8 bq = as_f16x8(t == 1 ? bl : bh);
                if constexpr (t == 0) asm volatile("" :: "v"(bl));      // (one counted wait per k-step, for both fragments, not two)
                if constexpr (t == 2) acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, acch, 0, 0, 0);
                else accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, accl, 0, 0, 0);
                // (k-steps 2 .. AHEAD - 1 are requested at the top of the tile, the others AHEAD k-steps before their use)
                if constexpr (t == 0 && s == 0)
                    static_for<AHEAD - 2>([&](auto kc) { load_b(2 + decltype(kc)::value, fh[(2 + decltype(kc)::value) % RING], fl[(2 + decltype(kc)::value) % RING]); });
                if constexpr (t == 0 && s + AHEAD < KS) load_b(s + AHEAD, fh[(s + AHEAD) % RING], fl[(s + AHEAD) % RING]);
                if constexpr (slot < kBarrierSlot) {           // everything that touches the LDS images sits in front of the barrier
                    constexpr int lo = slot * NITEM / kBarrierSlot, hi = (slot + 1) * NITEM / kBarrierSlot;
                    static_for<NITEM>([&](auto ic) {
                        if constexpr (decltype(ic)::value >= lo && decltype(ic)::value < hi) item(P, ic, cur ^ 1, cur ^ 1);
                    });
                    constexpr int elo = slot * 17 / kBarrierSlot, ehi = (slot + 1) * 17 / kBarrierSlot;
                    static_for<17>([&](auto ec) {
                        if constexpr (decltype(ec)::value >= elo && decltype(ec)::value < ehi) epi_item(accph, accpl, esum, ec, pcp);
                    });
                }
                if constexpr (DUTY && slot >= kDuty0 && (slot - kDuty0) % kDutyStep == 0 && (slot - kDuty0) / kDutyStep < kDutyItems) {
                    constexpr int d = (slot - kDuty0) / kDutyStep;
                    // x of iteration it + 2 (observations requested NW iterations ago), then the observations of iteration
                    // it + 2 + NW (record requested NW iterations ago), then the record of iteration it + 2 + 2 NW; the f16
                    // range watch rides along
                    if constexpr (d == 0 || d == 2 || d == 4) dm = prod4(oa[d / 2], ob[d / 2]);
                    else if constexpr (d == 1) { dxs[0] = dm; drc = fmaxf(max3a(dm.x, dm.y, dm.z), fabsf(dm.w)); }
                    else if constexpr (d == 3) { dxs[1] = dm; drc = fmaxf(drc, fabsf(dm.x)); dro = max3a(dm.y, dm.z, dm.w); }
                    else if constexpr (d == 5) { dxs[2] = dm; dro = fmaxf(dro, fabsf(dm.x)); drb = max3a(dm.y, dm.z, dm.w); }
                    else if constexpr (d == 11) range_raise(drc, dro, drb);
                    else if constexpr (d == 6) dA = obs_addr(rec_n);
                    else if constexpr (d < 10) { oa[d - 7] = dA.oi[d - 7]; ob[d - 7] = dA.oj[d - 7]; }
                    else rec_n = load_rec(tile_of(it + 2 + 2 * NW));
                }
                if constexpr (slot == kBarrierSlot - 1) {
                    // THE tile barrier, six MFMAs before the tile's end: every LDS access of the tile has been issued (the
                    // last fragment request went out at slot nslot - 9) and is complete behind the wait, so the images
                    // change hands here -- and the next tile's first two fragment pairs travel under the remaining MFMAs
                    // instead of in front of an idle matrix pipe
                    UAVTRACK_LDS_BARRIER();
                    const unsigned char *nb = aplanes + (cur ^ 1) * NP * PLANE + bfrag0;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        nfh[k] = *reinterpret_cast<const u32x4 *>(nb + 0 * PLANE + k * 32);
                        nfl[k] = *reinterpret_cast<const u32x4 *>(nb + 1 * PLANE + k * 32);
                    }
                    // the previous tile's partial scores are complete behind the barrier: their sum rides in the gaps of the
                    // remaining MFMAs, which carry nothing else.  (The buffer is next written two tiles on, by wavefronts that
                    // have passed the next barrier -- which this wavefront reaches after these reads.)
                    if (w == NW - 1 && it >= 1) final_sum(tile_of(it - 1), part + (cur ^ 1) * NW * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        ++it;
        cur ^= 1;
    };
    auto run_tile = [&](f32x16 &acch, f32x16 &accl, const f32x16 &accph, const f32x16 &accpl) {
        if ((it + 2) % NW == (unsigned)w) tile_body(std::true_type{}, acch, accl, accph, accpl);
        else tile_body(std::false_type{}, acch, accl, accph, accpl);
    };
    // two accumulator sets take turns (the finished tile's fc2 runs under the next tile's MFMAs): no copies between them
    f32x16 acc0h, acc0l, acc1h, acc1l;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1h[r] = 0.0f; acc1l[r] = 0.0f; }
    bool last_in_0 = false;
    for (unsigned tile = blockIdx.x; tile < ntiles;) {
        run_tile(acc0h, acc0l, acc1h, acc1l);
        tile += G;
        last_in_0 = true;
        if (tile >= ntiles) break;
        run_tile(acc1h, acc1l, acc0h, acc0l);
        tile += G;
        last_in_0 = false;
    }
    if (it > 0) {                            // the last tile's epilogue has nothing left to hide behind
        float esum = 0.0f;
        float *pcp = part + (cur ^ 1) * NW * 64 + w * 64;
        if (last_in_0) static_for<17>([&](auto ec) { epi_item(acc0h, acc0l, esum, ec, pcp); });
        else static_for<17>([&](auto ec) { epi_item(acc1h, acc1l, esum, ec, pcp); });
        __syncthreads();
        if (w == NW - 1) final_sum(tile_of(it - 1), part + (cur ^ 1) * NW * 64);
    }
}

struct MixParams {
    const uint32_t *nbrec;       // [S][B][N][W + 1] neighbour records of the chunk's S steps (internal.h, nbrec_words)
    const float *scores;         // one per emitted pair
    float *reward;               // [S][B][N]: in, what the rollout kernel left (raw reward of a UAV with neighbours, final reward of one without); out, final
    float *rsum;                 // [S][B] mean over the UAVs of the final reward (nullable): what the episode sum adds up
    unsigned *pair_count;        // reset here for the next chunk's pair emission
    unsigned *flags;             // [0] the scorer's f16 range flag, cleared here; [1] chunks the wide-range kernel re-scored so far
    int32_t SB, N, E;            // SB = S * B "virtual environments"
    float coop;
};

// One lane per UAV-step, E whole (step, environment) instances per workgroup.  The instances' records and reward slots are
// staged in LDS; a lane with neighbours walks the set bits of its own mask twice (max, then exp-sum), reads one score per
// neighbour and rewrites its reward slot; a lane without neighbours found its final reward there and touches nothing.
constexpr int mix_batch(int ws) { return ws ? 4 : 1; }     // (2: 0.130, 4: 0.114, 8: 0.123, 16: 0.181 ms per 200 steps at 4096 x 20)     // instance groups per workgroup (swarms up to 64 UAVs): their loads are all in flight before the first is used
template <int WS>                // mask words of a record when N <= 64 (1: N <= 32, 2: N <= 64), 0: any N (word loops)
__global__ void __launch_bounds__(kMaxWorkgroup) pmi_mix_kernel(const MixParams f)
{
    extern __shared__ uint32_t mix_lds[];
    constexpr int kMixBatch = mix_batch(WS);
    const int tid = threadIdx.x;
    const int N = f.N;
    const int W = WS ? WS : nbrec_mask_words(N), RS = W + 1;
    const int EN = f.E * N;                                 // lanes of one instance group
    const int e = tid / N, i = tid - e * N;
    float *rv = reinterpret_cast<float *>(mix_lds + (size_t)kMixBatch * EN * RS);      // the reward slots as the rollout kernel left them
    if (blockIdx.x == 0 && tid == 0) {
        *f.pair_count = 0;
        if (f.flags[0]) { f.flags[0] = 0; f.flags[1] += 1; }
    }
    // A workgroup takes kMixBatch consecutive groups of E instances (a group of one or a few instances per pass over the
    // lanes): with one group per workgroup the launch was bound by workgroup turnover (270 000 single-wavefront groups of
    // two dependent memory trips each), not by its 13 bytes per UAV-step.
    bool active[kMixBatch];
    size_t gid[kMixBatch];
#pragma unroll
    for (int k = 0; k < kMixBatch; ++k) {
        const int env0 = (blockIdx.x * kMixBatch + k) * f.E;
        active[k] = tid < EN && env0 + e < f.SB;
        gid[k] = (size_t)(env0 + e) * N + i;
    }
    {   // every load of the batch in flight, then the stores to LDS
        uint2 rec2[kMixBatch];
        float rw[kMixBatch];
#pragma unroll
        for (int k = 0; k < kMixBatch; ++k)
            if (active[k]) {
                if (WS == 1) rec2[k] = reinterpret_cast<const uint2 *>(f.nbrec)[gid[k]];
                rw[k] = f.reward[gid[k]];
            }
#pragma unroll
        for (int k = 0; k < kMixBatch; ++k)
            if (active[k]) {
                if (WS == 1) reinterpret_cast<uint2 *>(mix_lds)[k * EN + tid] = rec2[k];
                else
                    for (int w = 0; w < RS; ++w) mix_lds[(size_t)(k * EN + tid) * RS + w] = f.nbrec[gid[k] * RS + w];
                rv[k * EN + tid] = rw[k];
            }
    }
    __syncthreads();
    float r[kMixBatch];
#pragma unroll
    for (int k = 0; k < kMixBatch; ++k) {
    r[k] = 0.0f;
    if (active[k]) {
    const uint32_t *env_rec = mix_lds + (size_t)(k * EN + e * N) * RS;
    const uint32_t *me = env_rec + (size_t)i * RS;
    const float *raw_of = rv + k * EN + e * N;
    float rr = raw_of[i];                                              // no neighbours: already (1 - a) raw_i, clipped (uav.py:290)
    if (WS) {
        auto mask_of = [&](const uint32_t *rec) {
            return WS == 1 ? (unsigned long long)rec[0] : ((unsigned long long)rec[0] | ((unsigned long long)rec[1] << 32));
        };
        const unsigned long long mask = mask_of(me);
        if (mask) {
            rr = (1.0f - f.coop) * rr;
            if (!(mask & (mask - 1))) {
                // ONE neighbour: its softmax weight is exp(s - s) / exp(s - s) = 1 whatever the score (uav.py:287-288), so the
                // score is not read -- and where both UAVs of a pair are each other's only neighbour the rollout kernel never
                // emitted it (step_kernel.hip, drop_isolated).  fmaf(1, raw_j, 0) / 1 = raw_j: the bits of the general form.
                rr = fmaf(f.coop, raw_of[__ffsll((long long)mask) - 1], rr);
            } else {
                auto slot_of = [&](int j) {                // where s_ij lives: emitted by the lower index
                    const int lo = j > i ? i : j, hi = j > i ? j : i;
                    const uint32_t *rl = env_rec + (size_t)lo * RS;
                    const unsigned long long ml = mask_of(rl);
                    const unsigned long long later = (lo + 1 < 64) ? (ml >> (lo + 1)) << (lo + 1) : 0ull;
                    return rl[W] + (unsigned)__popcll(later & ((1ull << hi) - 1ull));
                };
                float mx = -INFINITY;
                for (unsigned long long m = mask; m; m &= m - 1) mx = fmaxf(mx, f.scores[slot_of(__ffsll((long long)m) - 1)]);
                float den = 0.0f, num = 0.0f;
                for (unsigned long long m = mask; m; m &= m - 1) {          // ascending j: scipy softmax, uav.py:287
                    const int j = __ffsll((long long)m) - 1;
                    const float ew = __builtin_amdgcn_exp2f((f.scores[slot_of(j)] - mx) * 1.44269504088896340736f);      // exp on v_exp_f32
                    den += ew;
                    num = fmaf(ew, raw_of[j], num);
                }
                rr = fmaf(f.coop, num / den, rr);                            // uav.py:288
            }
            rr = fminf(fmaxf(rr, -1.0f), 1.0f);                              // environment.py:225
            f.reward[gid[k]] = rr;
        }
    } else {
        auto bit = [&](const uint32_t *rec, int j) { return (rec[j >> 5] >> (j & 31)) & 1u; };
        auto slot_of = [&](int j) {
            const int lo = j > i ? i : j, hi = j > i ? j : i;
            const uint32_t *rl = env_rec + (size_t)lo * RS;
            unsigned rank = 0;
            for (int q = lo + 1; q < hi; ++q) rank += bit(rl, q);
            return rl[W] + rank;
        };
        float mx = -INFINITY;
        int cnt = 0;
        for (int j = 0; j < N; ++j)
            if (bit(me, j)) { mx = fmaxf(mx, f.scores[slot_of(j)]); ++cnt; }
        if (cnt) {
            float den = 0.0f, num = 0.0f;
            for (int j = 0; j < N; ++j)
                if (bit(me, j)) {
                    const float ew = __builtin_amdgcn_exp2f((f.scores[slot_of(j)] - mx) * 1.44269504088896340736f);      // exp on v_exp_f32
                    den += ew;
                    num = fmaf(ew, raw_of[j], num);
                }
            rr = fmaf(f.coop, num / den, (1.0f - f.coop) * rr);
            rr = fminf(fmaxf(rr, -1.0f), 1.0f);
            f.reward[gid[k]] = rr;
        }
    }
    r[k] = rr;
    }
    }
    if (f.rsum) {      // mean over the instance's UAVs, fixed order (train.py:181: the episode return adds these up over t)
        float *rl = rv + (size_t)kMixBatch * EN;      // (its own region behind the staged slots: neighbours may still be reading those)
#pragma unroll
        for (int k = 0; k < kMixBatch; ++k)
            if (active[k]) rl[k * EN + tid] = r[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kMixBatch; ++k)
            if (active[k] && i == 0) {
                float sum = 0.0f;
                for (int j = 0; j < N; ++j) sum += rl[k * EN + e * N + j];
                f.rsum[(blockIdx.x * kMixBatch + k) * f.E + e] = sum * (1.0f / (float)N);
            }
    }
}

// MAAC-R episode return (train.py:181-182): ep_sums[b][0] (+)= sum over the chunk's steps of the per-step mean reward
// the mix kernel left in rsum [S][B].  16 lanes per environment, each summing every 16th step (coalesced across the
// environments); the slices are then added in a fixed order (bitwise reproducible).  (The other four sums are the
// rollout kernel's, as in the other reward modes.)
constexpr int kEpSlices = 16;
__global__ void __launch_bounds__(256) ep_reward_kernel(const float *__restrict__ rsum, float *__restrict__ ep_sums, int S, int B, int add)
{
    __shared__ float red[256];
    const int el = threadIdx.x % 16, sl = threadIdx.x / 16;          // environment within the block, step slice
    const int b = blockIdx.x * 16 + el;
    float s = 0.0f;
    if (b < B)
        for (int t = sl; t < S; t += kEpSlices) s += rsum[(size_t)t * B + b];
    red[sl * 16 + el] = s;
    __syncthreads();
    if (sl == 0 && b < B) {
        float tot = 0.0f;
#pragma unroll
        for (int k = 0; k < kEpSlices; ++k) tot += red[k * 16 + el];
        float *ep = ep_sums + (size_t)b * 5;
        if (add) ep[0] += tot; else ep[0] = tot;
    }
}

}  // namespace

// Host-side repack of the fc1 block of the ABI blob (W1[3H][H], input-major) into the order the
// scorer's lanes load it: [column block w][k-step group t4][lane][4], element q of lane l being
// W1[2 (4 t4 + q) + (l >> 5)][32 w + (l & 31)].
void pack_pmi_blob(const float *abi_blob, float *device_order, int H)
{
    const int K = 3 * H, KH = K / 2, NW = H / 32;
    const size_t w1_off = (size_t)15 * H, w1_len = (size_t)K * H;
    const size_t total = w1_off + w1_len + H + H + 1;
    for (size_t k = 0; k < total; ++k) device_order[k] = abi_blob[k];
    const float *W1 = abi_blob + w1_off;
    float *dst = device_order + w1_off;
    for (int w = 0; w < NW; ++w)
        for (int t4 = 0; t4 < KH / 4; ++t4)
            for (int l = 0; l < 64; ++l)
                for (int q = 0; q < 4; ++q)
                    dst[(((size_t)w * (KH / 4) + t4) * 64 + l) * 4 + q] =
                        W1[(size_t)(2 * (4 * t4 + q) + (l >> 5)) * H + w * 32 + (l & 31)];
}

// fc1 as three bf16 planes (x = hi + mid + lo by truncation, exact) in the B-operand order of
// v_mfma_f32_32x32x16_bf16: per column block w, plane p, k-step s and lane l the eight values
// k = 16 s + 8 (l >> 5) + j, j = 0..7, of column 32 w + (l & 31).
void pack_pmi_x6(const float *abi_blob, uint16_t *planes, int H)
{
    const int K = 3 * H, KS = K / 16, NW = H / 32;
    const float *W1 = abi_blob + (size_t)15 * H;
    for (int w = 0; w < NW; ++w)
        for (int s = 0; s < KS; ++s)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    float v = W1[(size_t)(16 * s + 8 * (l >> 5) + j) * H + w * 32 + (l & 31)];
                    for (int p = 0; p < 3; ++p) {
                        uint32_t u;
                        memcpy(&u, &v, 4);
                        u &= 0xFFFF0000u;
                        float hi;
                        memcpy(&hi, &u, 4);
                        planes[((((size_t)w * 3 + p) * KS + s) * 64 + l) * 8 + j] = (uint16_t)(u >> 16);
                        v -= hi;                     // exact: the remainder has at most 16 significant bits left
                    }
                }
}

// x [n][12] -> the scorer's inputs for n stand-alone evaluations (uavtrack_pmi_inference): "environments" of two UAVs
// whose observations are x_k and a row of ones (x * 1 = x exactly), and the pair (0, 1) of each.
__global__ void __launch_bounds__(256) pmi_inference_prep_kernel(const float4 *__restrict__ x, float4 *__restrict__ obs2,
                                                                 uint2 *__restrict__ pairs, unsigned n)
{
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        obs2[(size_t)k * 6 + v] = x[(size_t)k * 3 + v];
        obs2[(size_t)k * 6 + 3 + v] = one;
    }
    pairs[k] = make_uint2(2u * k, 2u * k + 1u);
}

// fc1 for pmi_score_t3_kernel, block-scaled -- plane 0 = f16(T w), plane 1 =
// f16(T w - plane 0), T a power of two (uavtrack_set_pmi_weights).
void pack_pmi_t3(const float *abi_blob, uint16_t *planes, int H, float T)
{
    const int K = 3 * H, KS = K / 16, NW = H / 32;
    const float *W1 = abi_blob + (size_t)15 * H;
    for (int w = 0; w < NW; ++w)
        for (int s = 0; s < KS; ++s)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    // k-position -> fc1 input: inside a block of 32, position q = 16 kh + r holds unit (r & 3) + 8 (r >> 2) + 4 kh,
                    // the order in which pmi_score_t3_kernel's lanes store their activations
                    const int kp = 16 * s + 8 * (l >> 5) + j, q = kp & 31, r = q & 15;
                    const int row = (kp & ~31) + (r & 3) + 8 * (r >> 2) + 4 * (q >> 4);
                    const float v = T * W1[(size_t)row * H + w * 32 + (l & 31)];
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    uint16_t bh, bl;
                    memcpy(&bh, &hi, 2);
                    memcpy(&bl, &lo, 2);
                    planes[((((size_t)w * 2 + 0) * KS + s) * 64 + l) * 8 + j] = bh;
                    planes[((((size_t)w * 2 + 1) * KS + s) * 64 + l) * 8 + j] = bl;
                }
}

// The branch layers (PMINet.py:50-55, BatchNorm folded) for pmi_score_t3_kernel: per wavefront w and branch j the block of
// 32 units [32 w, 32 w + 32) as the A operand of v_mfma_f32_32x32x16_f16 over the 16 "inputs" x_0..x_11, 1, 0, 0, 0 --
// row = unit, k = input: the branch's own inputs carry its weights, input 12 its bias, the rest zeros.  Three planes of
// the block-scaled value v = S1 w: f16(v), f16(v - plane 0), plane 0 * 2^-11 (the partner of the inputs' 2^11-scaled
// remainder); lane l holds k = 8 (l >> 5) .. + 7 of row l & 31.
void pack_pmi_l1(const float *abi_blob, uint16_t *planes, int H, float S1)
{
    const int NW = H / 32;
    const int k0[3] = {0, 5, 9}, fan[3] = {5, 4, 3};
    const size_t woff[3] = {0, (size_t)6 * H, (size_t)11 * H};
    for (int w = 0; w < NW; ++w)
        for (int j = 0; j < 3; ++j)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 8; ++jj) {
                    const int unit = 32 * w + (l & 31), k = 8 * (l >> 5) + jj;
                    float v = 0.0f;
                    if (k >= k0[j] && k < k0[j] + fan[j]) v = abi_blob[woff[j] + (size_t)(k - k0[j]) * H + unit];
                    else if (k == 12) v = abi_blob[woff[j] + (size_t)fan[j] * H + unit];
                    v *= S1;
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const _Float16 hs = (_Float16)((float)hi * (1.0f / 2048.0f));
                    uint16_t bh, bl, bs;
                    memcpy(&bh, &hi, 2);
                    memcpy(&bl, &lo, 2);
                    memcpy(&bs, &hs, 2);
                    planes[((((size_t)w * 3 + j) * 3 + 0) * 64 + l) * 8 + jj] = bh;
                    planes[((((size_t)w * 3 + j) * 3 + 1) * 64 + l) * 8 + jj] = bl;
                    planes[((((size_t)w * 3 + j) * 3 + 2) * 64 + l) * 8 + jj] = bs;
                }
}

// behind a stand-alone scorer launch (uavtrack_pmi_inference): what the mix kernel does to the counters of a MAAC-R chunk
__global__ void pmi_counters_reset_kernel(unsigned *pair_count, unsigned *flags)
{
    *pair_count = 0;
    if (flags[0]) { flags[0] = 0; flags[1] += 1; }
}

hipError_t launch_pmi_counters_reset(const uavtrack_env *env, hipStream_t stream)
{
    hipLaunchKernelGGL(pmi_counters_reset_kernel, dim3(1), dim3(1), 0, stream, env->pair_count, env->pmi_flags);
    return hipGetLastError();
}

hipError_t launch_pmi_inference_prep(const float *x, float *obs2, uint2 *pairs, unsigned n, hipStream_t stream)
{
    hipLaunchKernelGGL(pmi_inference_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(obs2), pairs, n);
    return hipGetLastError();
}

int pmi_effective_scheme(const uavtrack_env *env)
{
    const bool split = pmi_x6_floats(env->pmi.hidden) != 0;       // widths whose planes stay register-resident: 64, 96, 128
    if (env->pmi_scheme != UAVTRACK_PMI_AUTO) return env->pmi_scheme;
    return split && env->pmi.t3 ? UAVTRACK_PMI_F16X3 : split ? UAVTRACK_PMI_BF16X6 : UAVTRACK_PMI_FP32;
}

bool pmi_scheme_fits(int hidden_padded, bool f16_range_ok, int scheme)
{
    const bool split = pmi_x6_floats(hidden_padded) != 0;
    switch (scheme) {
    case UAVTRACK_PMI_AUTO:   return true;
    case UAVTRACK_PMI_F16X3:  return split && f16_range_ok;     // (not ok: a weight or an activation bound beyond f16's range)
    case UAVTRACK_PMI_BF16X6: return split;
    case UAVTRACK_PMI_FP32:   return true;
    default:                  return false;
    }
}

bool pmi_scheme_available(const uavtrack_env *env, int scheme)
{
    return pmi_scheme_fits(env->pmi.hidden, env->pmi.t3 != nullptr, scheme);
}

// One scorer launch over the pair list (the scheme: pmi_effective_scheme).  All three kernels are persistent workgroups
// grid-striding over 32-pair tiles and are built for every multiple of 32 up to kPmiMaxHidden (uavtrack_set_pmi_weights
// pads other widths); tuned at the reference's two: 128 (configs/MAAC-R.yaml) and 64 (the class default).
hipError_t launch_pmi_score(const uavtrack_env *env, const float *obs, hipStream_t stream, const uint2 *pairs, float *scores, int n_uav)
{
    PmiParams q;
    q.blob = env->pmi.blob;
    q.x6 = env->pmi.x6;
    q.l1 = env->pmi.l1;
    q.t3 = env->pmi.t3;
    q.t3_scale = env->pmi.t3_s1 * env->pmi.t3_t;
    q.t3_inv_scale = 1.0f / q.t3_scale;
    q.obs = obs;
    q.pairs = pairs ? pairs : env->pairs;
    q.pair_count = env->pair_count;
    q.scores = scores ? scores : env->scores;
    q.pair_total = env->pair_total;
    q.N = n_uav > 0 ? n_uav : env->cfg.n_uav;
    for (int k = 0; k < 3; ++k) q.rng_inv[k] = env->pmi.rng_inv[k];
    q.range_flag = env->pmi_flags;
    q.gate = nullptr;
    const int cus = env->n_cus > 0 ? env->n_cus : 256;
    // split kernels: one workgroup (H / 32 wavefronts, one per SIMD) per CU; at H = 64 a workgroup is two wavefronts and
    // two of them share a CU
    const int grid_split = cus * (env->pmi.hidden <= 64 ? 2 : 1);
    const int scheme = pmi_effective_scheme(env);
    if (!pmi_scheme_available(env, scheme)) return hipErrorInvalidValue;
    auto launch_x6 = [&]() -> hipError_t {
        switch (env->pmi.hidden) {
#define UAVTRACK_PMI_CASE(HH) case HH: hipLaunchKernelGGL(pmi_score_x6_kernel<HH>, dim3(grid_split), dim3(2 * HH), 0, stream, q); break;
            UAVTRACK_PMI_CASE(64) UAVTRACK_PMI_CASE(96) UAVTRACK_PMI_CASE(128)
#undef UAVTRACK_PMI_CASE
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    };
    if (scheme == UAVTRACK_PMI_F16X3) {
        switch (env->pmi.hidden) {
#define UAVTRACK_PMI_CASE(HH) case HH: hipLaunchKernelGGL(pmi_score_t3_kernel<HH>, dim3(grid_split), dim3(2 * HH), 0, stream, q); break;
            UAVTRACK_PMI_CASE(64) UAVTRACK_PMI_CASE(96) UAVTRACK_PMI_CASE(128)
#undef UAVTRACK_PMI_CASE
        default: return hipErrorInvalidValue;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        // The stand-by: the host-side range guard of uavtrack_set_pmi_weights is built on nominal observation ranges, and the
        // uav.py:165 weight 1 / min(d, 1) is unbounded next to the origin.  The f16 kernel watches its inputs; when a tile
        // could saturate an operand it raises a flag and the bf16 x 6 kernel (fp32's exponent range) scores the chunk
        // again, stream-ordered, no host round trip.  With the flag down this launch returns at once (~3 us).
        q.gate = env->pmi_flags;
        return launch_x6();
    }
    if (scheme == UAVTRACK_PMI_BF16X6) return launch_x6();
    const int grid = 512;            // fp32 MFMA, two workgroups per CU (__launch_bounds__(2H, 2))
    switch (env->pmi.hidden) {
#define UAVTRACK_PMI_CASE(HH) case HH: hipLaunchKernelGGL(pmi_score_kernel<HH>, dim3(grid), dim3(2 * HH), 0, stream, q); break;
        UAVTRACK_PMI_CASE(32) UAVTRACK_PMI_CASE(64) UAVTRACK_PMI_CASE(96) UAVTRACK_PMI_CASE(128)
        UAVTRACK_PMI_CASE(160) UAVTRACK_PMI_CASE(192) UAVTRACK_PMI_CASE(224) UAVTRACK_PMI_CASE(256)
#undef UAVTRACK_PMI_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_pmi_finalize(const uavtrack_env *env, int steps, float *reward, float *rsum, hipStream_t stream)
{
    const uavtrack_config &c = env->cfg;
    MixParams f;
    f.nbrec = env->nbrec; f.scores = env->scores; f.reward = reward; f.rsum = rsum;
    f.pair_count = env->pair_count;
    f.flags = env->pmi_flags;
    // one lane per UAV-step (the rollout kernel's geometry is its own)
    // single-wavefront groups where whole instances fill >= 90 % of a wavefront (measured 0.141 vs 0.150 ms per 200 steps at 4096 x 20)
    const int dflt = (c.n_uav <= 64 && (64 / c.n_uav) * c.n_uav * 10 >= 64 * 9) ? 64 : (c.n_uav <= 256 ? 256 : kMaxWorkgroup);
    const int wgs = dflt;
    f.SB = steps * c.n_envs; f.N = c.n_uav; f.E = wgs / c.n_uav;
    f.coop = env->base.coop;
    const int kMixBatch = mix_batch(c.n_uav <= 64 ? 1 : 0);
    const unsigned groups = (unsigned)((f.SB + f.E * kMixBatch - 1) / (f.E * kMixBatch));
    const size_t lds = (size_t)kMixBatch * f.E * f.N * (nbrec_words(c.n_uav) + 2) * 4;     // records + reward slots + one float per lane (per-step mean)
    if (c.n_uav <= 32)
        hipLaunchKernelGGL(pmi_mix_kernel<1>, dim3(groups), dim3(wgs), lds, stream, f);
    else if (c.n_uav <= 64)
        hipLaunchKernelGGL(pmi_mix_kernel<2>, dim3(groups), dim3(wgs), lds, stream, f);
    else
        hipLaunchKernelGGL(pmi_mix_kernel<0>, dim3(groups), dim3(wgs), lds, stream, f);
    return hipGetLastError();
}

hipError_t launch_ep_reward(const uavtrack_env *env, int steps, const float *rsum, float *ep_sums, bool add, hipStream_t stream)
{
    const int B = env->cfg.n_envs;
    hipLaunchKernelGGL(ep_reward_kernel, dim3((unsigned)((B + 15) / 16)), dim3(256), 0, stream, rsum, ep_sums, steps, B, add ? 1 : 0);
    return hipGetLastError();
}

}  // namespace uavtrack
