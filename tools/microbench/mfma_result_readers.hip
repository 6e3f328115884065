// mfma_result_readers.hip -- does a VALU instruction beside an MFMA cost more when it READS registers that an earlier MFMA
// wrote (the pair scorer's producer reads layer-1 accumulators while layer-2 MFMAs are in flight) than when it reads
// registers a VALU instruction wrote?  One wave per SIMD; per gap one v_mfma_f32_32x32x16_f16 and NF fillers
// (v_cvt_pkrtz_f16_f32 of two sources); the sources are (A) plain registers, (B) the accumulator the MFMA TWO gaps back
// wrote (complete long ago), (C) the accumulator the PREVIOUS MFMA wrote.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-mfma-vgpr-form=1 mfma_result_readers.hip -o mfma_result_readers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NF>
__global__ void __launch_bounds__(256) bench(float *out, int iters)
{
    const int tid = threadIdx.x;
    f32x16 acc[3];
    for (int k = 0; k < 3; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    f16x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(0.001f * (tid & 7)); b[r] = (_Float16)0.5f; }
    float t[16];
    for (int r = 0; r < 16; ++r) t[r] = 1.0f + r + tid;
    unsigned s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 24; ++g) {             // 24 gaps of straight-line code per iteration; accumulators 0, 1, 2 in turn
            const int cur = g % 3, prev = (g + 2) % 3, prev2 = (g + 1) % 3;
            acc[cur] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[cur], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                float x, y;
                if (MODE == 0) { x = t[(2 * j) % 16]; y = t[(2 * j + 1) % 16]; }
                else if (MODE == 1) { x = acc[prev2][(2 * j) % 16]; y = acc[prev2][(2 * j + 1) % 16]; }
                else { x = acc[prev][(2 * j) % 16]; y = acc[prev][(2 * j + 1) % 16]; }
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(s[j % 8]) : "v"(x), "v"(y));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
    for (int k = 0; k < 3; ++k) for (int q = 0; q < 16; ++q) r += acc[k][q];
    for (int j = 0; j < 8; ++j) r += (float)s[j];
    out[blockIdx.x * 256 + tid] = r;
}

template <int MODE, int NF>
int run(const char *name)
{
    const int grid = 256;
    float *out;
    CHK(hipMalloc(&out, (size_t)grid * 256 * 4));
    const int iters = 600;
    hipLaunchKernelGGL((bench<MODE, NF>), dim3(grid), dim3(256), 0, 0, out, 300);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((bench<MODE, NF>), dim3(grid), dim3(256), 0, 0, out, iters);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s NF %d: %7.2f ns per gap\n", name, NF, ms * 1e6 / iters / 24);
    CHK(hipFree(out));
    return 0;
}
#define ALL(M, name) run<M, 0>(name); run<M, 2>(name); run<M, 4>(name); run<M, 5>(name); run<M, 6>(name);
int main()
{
    for (int rep = 0; rep < 2; ++rep) {
        ALL(0, "sources: plain registers")
        ALL(1, "sources: accumulator written two MFMAs back")
        ALL(2, "sources: accumulator of the previous MFMA")
    }
    return 0;
}
