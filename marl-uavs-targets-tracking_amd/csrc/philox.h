// philox.h -- Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as
// easy as 1, 2, 3", SC'11), the counter-based stream behind uavtrack_reset.  The
// reference seeds Python's MT19937 (environment.py:54-83); a sequential generator has
// no place on a GPU, so reset draws are keyed by (seed, global env id, episode, agent).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uavtrack {

struct Philox4 { uint32_t v[4]; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// 24 random bits -> [0, 1), exactly representable in fp32
__host__ __device__ inline float u01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-08f; }

}  // namespace uavtrack
