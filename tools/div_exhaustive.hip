// Exhaustive check of candidate correctly rounded f32 divisions a / b that share the work on the
// divisor between several dividends: every pair of mantissas (a, b in [1, 2): 2^46 pairs; powers of
// two scale every intermediate exactly while nothing leaves the normal range) against the
// compiler's IEEE division.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/div_exhaustive.hip -o build/div_exhaustive
// run:   build/div_exhaustive [chunks-of-2^17-dividends, default all 64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void check(uint32_t a_first, uint32_t a_count, unsigned long long *bad) {
    const uint32_t bi = blockIdx.x * blockDim.x + threadIdx.x;  // 2^23 divisors
    const float b = __uint_as_float(0x3F800000u + bi);
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float yrn = 1.0f / b;  // the correctly rounded reciprocal, however it was obtained
    unsigned n1 = 0, n1b = 0, n2 = 0, n2b = 0, n3 = 0;
    for (uint32_t i = 0; i < a_count; i++) {
        const float a = __uint_as_float(0x3F800000u + a_first + i);
        const float ref = a / b;
        {   // refined reciprocal, one correction of the quotient
            float q = a * y1;
            float r = __builtin_fmaf(-b, q, a);
            q = __builtin_fmaf(r, y1, q);
            n1 += q != ref;
            r = __builtin_fmaf(-b, q, a);  // ... and a second one
            q = __builtin_fmaf(r, y1, q);
            n2 += q != ref;
        }
        {   // the correctly rounded reciprocal, one correction of the quotient
            float q = a * yrn;
            float r = __builtin_fmaf(-b, q, a);
            q = __builtin_fmaf(r, yrn, q);
            n3 += q != ref;
        }
        {   // the hardware's reciprocal as it is
            float q = a * y0;
            float r = __builtin_fmaf(-b, q, a);
            q = __builtin_fmaf(r, y0, q);
            n1b += q != ref;
            r = __builtin_fmaf(-b, q, a);
            q = __builtin_fmaf(r, y0, q);
            n2b += q != ref;
        }
    }
    if (n1) atomicAdd(&bad[0], (unsigned long long)n1);
    if (n2) atomicAdd(&bad[1], (unsigned long long)n2);
    if (n1b) atomicAdd(&bad[2], (unsigned long long)n1b);
    if (n2b) atomicAdd(&bad[3], (unsigned long long)n2b);
    if (n3) atomicAdd(&bad[4], (unsigned long long)n3);
}

int main(int argc, char **argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 64;
    unsigned long long *bad, h[5];
    CHECK(hipMalloc(&bad, sizeof(h)));
    CHECK(hipMemset(bad, 0, sizeof(h)));
    const uint32_t per = 1u << 17;
    for (int c = 0; c < chunks; c++) {
        check<<<(1u << 23) / 256, 256>>>(c * per, per, bad);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
        printf("dividend chunk %d of 64: mismatches so far  rcp+newton,1 corr %llu  rcp+newton,2 corr %llu  rcp,1 corr %llu  rcp,2 corr %llu  RN(1/b),1 corr %llu\n",
               c + 1, h[0], h[1], h[2], h[3], h[4]);
        fflush(stdout);
    }
    return 0;
}
