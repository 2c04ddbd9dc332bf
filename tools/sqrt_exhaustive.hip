// Exhaustive check of candidate correctly rounded f32 square roots on the GPU: every float in
// [2^-96, inf) against (float)sqrt((double)x) (a double square root rounded to float is the correctly
// rounded float square root: 53 >= 2*24 + 2 bits).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/sqrt_exhaustive.hip -o /tmp/sqrt_exhaustive
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ float sqrt_fixup(float x) {  // v_sqrt_f32 and the two residuals of its neighbours
    float s = __builtin_amdgcn_sqrtf(x);
    float sd = __int_as_float(__float_as_int(s) - 1);
    float su = __int_as_float(__float_as_int(s) + 1);
    float rd = __builtin_fmaf(-sd, s, x);
    float ru = __builtin_fmaf(-su, s, x);
    float r = (rd <= 0.0f) ? sd : s;
    return (ru > 0.0f) ? su : r;
}

__device__ float sqrt_newton_rsq(float x) {  // one residual, one fused correction with h = rsq(x)/2
    float s = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rsqf(x);
    float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(h, d, s);
}

__device__ float sqrt_newton_rcp(float x) {  // the same with h = rcp(s)/2
    float s = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rcpf(s);
    float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(h, d, s);
}

__global__ void check(uint32_t first, uint32_t count, unsigned long long *bad) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < count; i += gridDim.x * blockDim.x) {
        float x = __uint_as_float(first + i);
        float ref = (float)sqrt((double)x);
        if (sqrt_fixup(x) != ref) atomicAdd(&bad[0], 1ull);
        if (sqrt_newton_rsq(x) != ref) atomicAdd(&bad[1], 1ull);
        if (sqrt_newton_rcp(x) != ref) atomicAdd(&bad[2], 1ull);
        if (sqrtf(x) != ref) atomicAdd(&bad[3], 1ull);
        // the reciprocal of the root, seeded with the reciprocal square root that the root itself uses
        // (x < 2^96 only: the root's callers that divide by it keep to that range)
        if (x < 0x1p96f) {
            const float inv = 1.0f / ref;
            const float y0 = __builtin_amdgcn_rsqf(x);
            const float e = __builtin_fmaf(-ref, y0, 1.0f);
            const float y1 = __builtin_fmaf(e, y0, y0);
            const float r = __builtin_fmaf(-ref, y1, 1.0f);
            if (y1 != inv) atomicAdd(&bad[4], 1ull);
            if (__builtin_fmaf(r, y1, y1) != inv) atomicAdd(&bad[5], 1ull);
            const float z0 = __builtin_amdgcn_rcpf(ref);
            const float e2 = __builtin_fmaf(-ref, z0, 1.0f);
            const float z1 = __builtin_fmaf(e2, z0, z0);
            const float r2 = __builtin_fmaf(-ref, z1, 1.0f);
            if (z1 != inv) atomicAdd(&bad[6], 1ull);
            if (__builtin_fmaf(r2, z1, z1) != inv) atomicAdd(&bad[7], 1ull);
        }
    }
}

int main() {
    unsigned long long *bad, h[8];
    if (hipMalloc(&bad, sizeof(h)) != hipSuccess) return 1;
    if (hipMemset(bad, 0, sizeof(h)) != hipSuccess) return 1;
    const uint32_t first = 0x0F800000u, last = 0x7F800000u;  // 2^-96 .. inf (exclusive)
    check<<<4096, 256>>>(first, last - first, bad);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    if (hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 3;
    printf("inputs %u\nfixup mismatches      %llu\nnewton(rsq) mismatches %llu\nnewton(rcp) mismatches %llu\nsqrtf mismatches      %llu\n",
           last - first, h[0], h[1], h[2], h[3]);
    printf("1/sqrt(x), x < 2^96: rsq seed + newton %llu, + one correction %llu; rcp seed + newton %llu, + one correction %llu\n", h[4], h[5], h[6], h[7]);
    return 0;
}
