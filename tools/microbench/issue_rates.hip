// issue_rates.hip -- what ONE wavefront alone on its SIMD (and two sharing one) sustains on gfx950, per instruction
// class: the numbers the step kernel's lane mapping is designed around.  Diagnostic only; not part of the library.
//   hipcc --offload-arch=gfx950 -O2 issue_rates.hip -o issue_rates && ./issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int TEST>
__global__ void __launch_bounds__(64) bench(float *out, unsigned long long *ticks, int iters)
{
    __shared__ float4 lds[256];
    const int tid = threadIdx.x;
    lds[tid] = make_float4(tid, 1.0f, 2.0f, 3.0f);
    lds[tid + 64] = make_float4(tid, 1.0f, 2.0f, 3.0f);
    __syncthreads();
    v2f a0 = {1.0f + tid, 2.0f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const v2f m = {0.999f, 1.001f}, c = {0.5f, 0.25f};
    float s0 = 1.0f + tid, s1 = 2.f, s2 = 3.f, s3 = 4.f, s4 = 5.f, s5 = 6.f, s6 = 7.f, s7 = 8.f;
    int addr = (tid & 63) * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (TEST == 0) {          // 8 independent v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (TEST == 1) {   // one dependent v_pk_fma_f32 chain
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                              "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                              : "+v"(a0) : "v"(m), "v"(c));)
        } else if (TEST == 2) {   // 8 independent v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(m.x), "v"(c.x));)
        } else if (TEST == 3) {   // one dependent v_fma_f32 chain
            REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                              "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                              : "+v"(s0) : "v"(m.x), "v"(c.x));)
        } else if (TEST == 4) {   // 8 independent v_sqrt_f32
            REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));)
        } else if (TEST == 5) {   // dependent v_sqrt_f32
            REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %0, %0\n"
                              : "+v"(s0));)
        } else if (TEST == 6) {   // dependent ds_read_b32 chain (address from data): LDS latency
            REP8(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xfc, %0\n" : "+v"(addr));)
        } else if (TEST == 7) {   // 8 independent ds_read_b128 then one wait
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                         "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr));
            s0 += r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
        } else if (TEST == 8) {   // v_cmp -> v_cndmask pairs (SGPR mask hazard), 4 independent pairs
            REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %4\n v_cndmask_b32 %0, %0, %5, s[20:21]\n v_cmp_lt_f32 s[22:23], %1, %4\n v_cndmask_b32 %1, %1, %5, s[22:23]\n"
                              "v_cmp_lt_f32 s[24:25], %2, %4\n v_cndmask_b32 %2, %2, %5, s[24:25]\n v_cmp_lt_f32 s[26:27], %3, %4\n v_cndmask_b32 %3, %3, %5, s[26:27]\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(m.x), "v"(c.x) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if (TEST == 9) {   // 8 independent v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));)
        } else if (TEST == 10) {  // 2 interleaved dependent pk_fma chains
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n"
                              "v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n"
                              : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));)
        } else if (TEST == 11) {  // 2 interleaved dependent v_fma chains
            REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                              "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                              : "+v"(s0), "+v"(s1) : "v"(m.x), "v"(c.x));)
        } else if (TEST == 12) {  // 8 SALU (s_add_u32) independent-ish
            REP8(asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n"
                              ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");)
        } else if (TEST == 13) {  // alternating SALU / VALU
            REP8(asm volatile("s_add_u32 s20, s20, 1\n v_fma_f32 %0, %0, %4, %5\n s_add_u32 s21, s21, 1\n v_fma_f32 %1, %1, %4, %5\n s_add_u32 s22, s22, 1\n v_fma_f32 %2, %2, %4, %5\n s_add_u32 s23, s23, 1\n v_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(m.x), "v"(c.x) : "s20", "s21", "s22", "s23", "scc");)
        } else if (TEST == 14) {  // v_readlane + dependent SALU use
            REP8(asm volatile("v_readlane_b32 s20, %0, 3\n s_add_u32 s21, s20, 1\n v_readlane_b32 s22, %0, 5\n s_add_u32 s23, s22, 1\n v_readlane_b32 s24, %0, 7\n s_add_u32 s25, s24, 1\n v_readlane_b32 s26, %0, 9\n s_add_u32 s27, s26, 1\n"
                              :: "v"(s0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");)
        } else if (TEST == 15) {  // ds_write_b128 x8 then wait
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f wv = {s0, s1, s2, s3};
            asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n"
                         "ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n s_waitcnt lgkmcnt(0)\n"
                         :: "v"(addr), "v"(wv) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + tid] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + addr;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int TEST>
int run(const char *name, int per_iter, int grid, int lanes = 64)
{
    float *out; unsigned long long *ticks;
    CHK(hipMalloc(&out, (size_t)grid * 64 * 4)); CHK(hipMalloc(&ticks, (size_t)grid * 8));
    const int iters = 2000;
    hipLaunchKernelGGL(bench<TEST>, dim3(grid), dim3(lanes), 0, 0, out, ticks, 10);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<TEST>, dim3(grid), dim3(lanes), 0, 0, out, ticks, iters);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CHK(hipMemcpy(h.data(), ticks, (size_t)grid * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-52s grid %5d lanes %2d: %7.2f ticks/instr (s_memtime), %7.2f ns/instr wall\n", name, grid, lanes, s / grid / iters / per_iter, ms * 1e6 / iters / per_iter);
    CHK(hipFree(out)); CHK(hipFree(ticks));
    return 0;
}

int main()
{
    for (int grid : {1024, 2048, 4096}) {   // ~1, 2, 4 single-wave workgroups per SIMD
        run<0>("8 independent v_pk_fma_f32", 64, grid);
        run<1>("dependent v_pk_fma_f32 chain", 64, grid);
        run<10>("2 interleaved dependent v_pk_fma_f32 chains", 64, grid);
        run<2>("8 independent v_fma_f32", 64, grid);
        run<3>("dependent v_fma_f32 chain", 64, grid);
        run<11>("2 interleaved dependent v_fma_f32 chains", 64, grid);
        run<4>("8 independent v_sqrt_f32", 64, grid);
        run<5>("dependent v_sqrt_f32 chain", 64, grid);
        run<9>("8 independent v_exp_f32", 64, grid);
        run<6>("dependent ds_read_b32 (+wait +v_and) round trip", 8, grid);
        run<7>("8 ds_read_b128 + one wait (per read)", 8, grid);
        run<15>("8 ds_write_b128 + one wait (per write)", 8, grid);
        run<8>("v_cmp -> v_cndmask (4 pairs), per instruction", 64, grid);
        run<12>("s_add_u32 x8 (SALU)", 64, grid);
        run<13>("alternating s_add_u32 / v_fma_f32", 64, grid);
        run<14>("v_readlane -> s_add pairs, per instruction", 64, grid);
    }
    // Is the second 32-lane pass of a wave64 instruction skipped when EXEC[63:32] == 0?  Same loops, 32 and 20 active lanes.
    for (int lanes : {64, 32, 20})
        for (int grid : {4096, 8192}) {
            run<0>("8 independent v_pk_fma_f32", 64, grid, lanes);
            run<2>("8 independent v_fma_f32", 64, grid, lanes);
            run<4>("8 independent v_sqrt_f32", 64, grid, lanes);
            run<7>("8 ds_read_b128 + one wait (per read)", 8, grid, lanes);
        }
    return 0;
}
