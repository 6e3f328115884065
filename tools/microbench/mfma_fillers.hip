// mfma_fillers.hip -- what one filler instruction costs in the shadow of a v_mfma_f32_32x32x16_f16, ONE wave per SIMD
// (the pair scorer's situation: pmi_score_t3_kernel is bound by the instructions beside its MFMAs).  Each test is a loop
// of 32 gaps (straight-line code, as in the scorer; a two-gap loop body showed fetch artefacts): one MFMA (two accumulators alternate, as in the scorer) followed by NF fillers of one kind; reported:
// cycles per gap.  Diagnostic only; not part of the library.
//   hipcc --offload-arch=gfx950 -O2 mfma_fillers.hip -o mfma_fillers && ./mfma_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define MFMA0 "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n"
#define MFMA1 "v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n"
// fillers act on s0..s7 = %4..%11 (independent registers), constants %12 (f32), %13 (packed f16 word / f32 pair low)
#define F_FMA(n) "v_fma_f32 %" #n ", %" #n ", %12, %12\n"
#define F_MIX(n) "v_fma_mix_f32 %" #n ", %13, -1.0, %" #n " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define F_MIXC(n) "v_fma_mix_f32 %" #n ", %13, -1.0, %" #n " op_sel:[0,0,0] op_sel_hi:[1,0,0] clamp\n"
#define F_PKMAX16(n) "v_pk_max_f16 %" #n ", %" #n ", %13\n"
#define F_CVTPK(n) "v_cvt_pkrtz_f16_f32 %" #n ", %" #n ", %12\n"
#define F_CVT(n) "v_cvt_f32_f16 %" #n ", %" #n "\n"
#define F_CVTS(n) "v_cvt_f32_f16_sdwa %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define F_MAXI(n) "v_max_i32 %" #n ", %" #n ", %13\n"
#define F_MUL(n) "v_mul_f32 %" #n ", %" #n ", %12\n"
#define F_AND(n) "v_and_b32 %" #n ", %" #n ", %13\n"
#define F_NOP(n) "s_nop 0\n"

#define GAPS(F)                                                                                                            \
    if (NF == 4) asm volatile(MFMA0 F(4) F(5) F(6) F(7) MFMA1 F(8) F(9) F(10) F(11)                                        \
                              : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "v"(k), "v"(w)); \
    else if (NF == 5) asm volatile(MFMA0 F(4) F(5) F(6) F(7) F(8) MFMA1 F(9) F(10) F(11) F(4) F(5)                         \
                              : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "v"(k), "v"(w)); \
    else if (NF == 6) asm volatile(MFMA0 F(4) F(5) F(6) F(7) F(8) F(9) MFMA1 F(10) F(11) F(4) F(5) F(6) F(7)               \
                              : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "v"(k), "v"(w)); \
    else if (NF == 7) asm volatile(MFMA0 F(4) F(5) F(6) F(7) F(8) F(9) F(10) MFMA1 F(11) F(4) F(5) F(6) F(7) F(8) F(9)     \
                              : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "v"(k), "v"(w)); \
    else if (NF == 8) asm volatile(MFMA0 F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) MFMA1 F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) \
                              : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "v"(k), "v"(w)); \
    else asm volatile(MFMA0 MFMA1 : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "v"(k), "v"(w));

#define DEP3(F) asm volatile(MFMA0 F(4) F(5) F(6) F(7) MFMA0 F(8) F(9) F(10) F(11) MFMA1 F(4) F(5) F(6) F(7)  \
                              : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "v"(k), "v"(w));
#define DEP3N asm volatile(MFMA0 MFMA0 MFMA1 : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "v"(k), "v"(w));
#define ALT3(F) asm volatile(MFMA0 F(4) F(5) F(6) F(7) MFMA1 F(8) F(9) F(10) F(11) MFMA0 F(4) F(5) F(6) F(7) MFMA1 F(8) F(9) F(10) F(11) MFMA0 F(4) F(5) F(6) F(7) MFMA1 F(8) F(9) F(10) F(11) \
                              : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7), "v"(k), "v"(w));
// the scorer's k-step: three MFMAs; behind the first two b128 reads (fragments two k-steps ahead, ring of three) and a b64
// store, behind all of them VALU fillers; the MFMA that consumes a fragment waits for it with a counted lgkmcnt
#define KSTEP(RD0, RD1, USE, WR)                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(3)\n v_mfma_f32_32x32x16_f16 %0, %2, " USE ", %0\n"                                       \
                 "ds_read_b128 " RD0 ", %13\n ds_read_b128 " RD1 ", %13 offset:25088\n" WR                                     \
                 "v_fma_f32 %4, %4, %12, %12\n v_fma_f32 %5, %5, %12, %12\n"                                                  \
                 "v_mfma_f32_32x32x16_f16 %0, %2, " USE ", %0\n"                                                               \
                 "v_fma_f32 %6, %6, %12, %12\n v_fma_f32 %7, %7, %12, %12\n v_fma_f32 %8, %8, %12, %12\n v_fma_f32 %9, %9, %12, %12\n v_fma_f32 %10, %10, %12, %12\n" \
                 "v_mfma_f32_32x32x16_f16 %1, %2, " USE ", %1\n"                                                               \
                 "v_fma_f32 %6, %6, %12, %12\n v_fma_f32 %7, %7, %12, %12\n v_fma_f32 %8, %8, %12, %12\n v_fma_f32 %9, %9, %12, %12\n v_fma_f32 %10, %10, %12, %12\n" \
                 : "+v"(c0), "+v"(c1), "+v"(a), "+v"(b), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7)                 \
                 : "v"(k), "v"(raddr), "v"(waddr), "v"(wdata), "v"(f0), "v"(f1), "v"(f2) : "memory");
template <int KIND, int NF>
__global__ void __launch_bounds__(256) bench(float *out, unsigned long long *ticks, int iters)
{
    const int tid = threadIdx.x;
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    f16x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(0.001f * (tid & 7)); b[r] = (_Float16)0.5f; }
    float s0 = 1.0f + tid, s1 = 2.f, s2 = 3.f, s3 = 4.f, s4 = 5.f, s5 = 6.f, s6 = 7.f, s7 = 8.f;
    const float k = 0.999f;
    const unsigned w = 0x3c003c00u;
    __shared__ float4 ldsbuf[5000];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u4 f0 = {0, 0, 0, 0}, f1 = f0, f2 = f0;
    const u2 wdata = {w, w};
    const int lane = tid & 63;
    // the scorer's addressing: fragment rows of 784 B per pair (conflict-free b128 reads); stores 8 B at row * 784 + ...
    const unsigned raddr = (lane & 31) * 784 + (lane >> 5) * 16;
    const unsigned waddr = (KIND == 16 ? (lane & 31) * 784 : (lane & 31) * 776) + (lane >> 5) * 8 + (tid >> 6) * 64;   // 784: 2-way store conflict, 776: none
    if (tid == 0) ldsbuf[0] = make_float4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define R16(x) x x x x x x x x x x x x x x x x
        if (KIND == 0) { R16(GAPS(F_FMA)) }
        else if (KIND == 1) { R16(GAPS(F_MIX)) }
        else if (KIND == 2) { R16(GAPS(F_MIXC)) }
        else if (KIND == 3) { R16(GAPS(F_PKMAX16)) }
        else if (KIND == 4) { R16(GAPS(F_CVTPK)) }
        else if (KIND == 5) { R16(GAPS(F_CVT)) }
        else if (KIND == 6) { R16(GAPS(F_CVTS)) }
        else if (KIND == 7) { R16(GAPS(F_MAXI)) }
        else if (KIND == 8) { R16(GAPS(F_MUL)) }
        else if (KIND == 9) { R16(GAPS(F_AND)) }
        else if (KIND == 10) { R16(GAPS(F_NOP)) }
        else if (KIND >= 14 && KIND <= 16) {
#define WRS "ds_write_b64 %14, %15 offset:50432\n"
            if (KIND == 14) { R16(KSTEP("%16", "%17", "%3", "") KSTEP("%16", "%17", "%3", "")) }
            else { R16(KSTEP("%16", "%17", "%3", WRS) KSTEP("%16", "%17", "%3", WRS)) }
        }
        else if (KIND == 11) { R16(DEP3(F_FMA)) R16(DEP3(F_FMA)) }     // 96 gaps: same accumulator twice in a row, then the other
        else if (KIND == 12) { R16(DEP3N) R16(DEP3N) }
        else if (KIND == 13) { R16(ALT3(F_FMA)) }                      // 96 gaps, alternating accumulators
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r];
    out[blockIdx.x * 256 + tid] = acc + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    if ((tid & 63) == 0) ticks[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int KIND, int NF>
int run(const char *name)
{
    const int grid = 256;                   // one 4-wave workgroup per CU: one wave per SIMD
    float *out; unsigned long long *ticks;
    CHK(hipMalloc(&out, (size_t)grid * 256 * 4)); CHK(hipMalloc(&ticks, (size_t)grid * 4 * 8));
    const int iters = 400;
    hipLaunchKernelGGL((bench<KIND, NF>), dim3(grid), dim3(256), 0, 0, out, ticks, 200);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((bench<KIND, NF>), dim3(grid), dim3(256), 0, 0, out, ticks, iters);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * 4);
    CHK(hipMemcpy(h.data(), ticks, (size_t)grid * 4 * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += (double)v;
    // s_memtime counts at 100 MHz; wall time -> ns per gap; cycles at the clock the MFMA-only loop implies (32 cycles per gap)
    const int gaps = KIND >= 11 ? 96 : 32;   // (KIND 14-16: 32 k-steps of three MFMAs)
    printf("%-28s NF %d: %7.2f ns per gap (wall)  %7.2f memtime ticks per gap\n", name, NF, ms * 1e6 / iters / gaps, s / (grid * 4) / iters / gaps);
    CHK(hipFree(out)); CHK(hipFree(ticks));
    return 0;
}

#define ALLNF(K, name) run<K, 4>(name); run<K, 5>(name); run<K, 6>(name); run<K, 7>(name); run<K, 8>(name);
int main()
{
    for (int rep = 0; rep < 2; ++rep) {     // (first pass also warms the clocks)
        run<0, 0>("MFMA only");
        ALLNF(0, "v_fma_f32")
        ALLNF(1, "v_fma_mix_f32 (hi half)")
        ALLNF(2, "v_fma_mix_f32 clamp")
        ALLNF(3, "v_pk_max_f16")
        ALLNF(4, "v_cvt_pkrtz_f16_f32")
        ALLNF(5, "v_cvt_f32_f16")
        ALLNF(6, "v_cvt_f32_f16_sdwa")
        ALLNF(7, "v_max_i32")
        ALLNF(8, "v_mul_f32")
        ALLNF(9, "v_and_b32")
        ALLNF(10, "s_nop 0")
        run<11, 4>("c0 c0 c1 + 4 v_fma_f32");
        run<12, 0>("c0 c0 c1 bare");
        run<13, 4>("c0 c1 c0 c1 + 4 v_fma_f32");
        run<14, 4>("k-step: 2 b128 reads, 12 fma");
        run<15, 4>("k-step: + b64 store (no conflict)");
        run<16, 4>("k-step: + b64 store (2-way)");
    }
    return 0;
}
