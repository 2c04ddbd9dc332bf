// Issue-rate microbenchmark for the gfx950 vector ALU: cycles per wave64
// instruction per SIMD for the opcodes the ray-marching kernels are made of.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITERS = 2000;
constexpr int OPS = 16;  // independent instances per iteration

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, ASM2, ASM1)                                                                     \
    __global__ void k_##NAME(float *out, unsigned long long *cycles, float seed) {                   \
        float a[OPS], b[OPS];                                                                        \
        for (int i = 0; i < OPS; i++) { a[i] = seed + threadIdx.x + i; b[i] = seed * 0.5f + i; }   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int it = 0; it < ITERS; it++) {                                                         \
            _Pragma("unroll") for (int i = 0; i < OPS; i++) { ASM2; ASM1; }                          \
        }                                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        float s = 0; for (int i = 0; i < OPS; i++) s += a[i] + b[i];                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                              \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    }

#define NOP1 (void)0
KERNEL(v_add_f32, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])), NOP1)
KERNEL(v_mul_f32, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])), NOP1)
KERNEL(v_fma_f32, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i])), NOP1)
KERNEL(v_min_f32, asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])), NOP1)
KERNEL(v_min3_f32, asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i])), NOP1)
KERNEL(v_sqrt_f32, asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i])), NOP1)
KERNEL(v_rcp_f32, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i])), NOP1)
KERNEL(v_rsq_f32, asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i])), NOP1)
KERNEL(v_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])), NOP1)
KERNEL(v_cmp_cndmask, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc"), NOP1)
KERNEL(v_cmp_only, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc"), NOP1)
KERNEL(v_cndmask_only, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i])), NOP1)
KERNEL(v_ldexp_f32, asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(a[i])), NOP1)
KERNEL(v_div_fixup, asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i])), NOP1)
KERNEL(v_mov_b32, asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i])), NOP1)
KERNEL(v_readlane, { int s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(a[i])); asm volatile("" ::"s"(s_)); }, NOP1)
KERNEL(v_floor_f32, asm volatile("v_floor_f32 %0, %0" : "+v"(a[i])), NOP1)
KERNEL(v_med3_f32, asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i])), NOP1)
KERNEL(v_exp_f32, asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])), NOP1)
KERNEL(v_log_f32, asm volatile("v_log_f32 %0, %0" : "+v"(a[i])), NOP1)
KERNEL(add_mul_mix, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])), asm volatile("v_sqrt_f32 %0, %0" : "+v"(b[i])))

// packed: operate on register pairs
#define KERNEL_PK(NAME, INSTR)                                                                       \
    __global__ void k_##NAME(float *out, unsigned long long *cycles, float seed) {                   \
        typedef float f2 __attribute__((ext_vector_type(2)));                                        \
        f2 a[OPS], b[OPS];                                                                           \
        for (int i = 0; i < OPS; i++) { a[i] = f2{seed + threadIdx.x + i, seed}; b[i] = f2{seed * 0.5f + i, 1.0f}; } \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int it = 0; it < ITERS; it++) {                                                         \
            _Pragma("unroll") for (int i = 0; i < OPS; i++) asm volatile(INSTR : "+v"(a[i]) : "v"(b[i])); \
        }                                                                                            \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        float s = 0; for (int i = 0; i < OPS; i++) s += a[i].x + a[i].y;                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                              \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    }
KERNEL_PK(v_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL_PK(v_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL_PK(v_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")

typedef void (*kern_t)(float *, unsigned long long *, float);
struct Entry { const char *name; kern_t fn; int ops_per_iter; };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    Entry entries[] = {
#define E(N) {#N, k_##N, OPS},
        E(v_add_f32) E(v_mul_f32) E(v_fma_f32) E(v_min_f32) E(v_min3_f32) E(v_med3_f32) E(v_add_u32) E(v_mov_b32)
        E(v_cmp_only) E(v_cndmask_only) E(v_cmp_cndmask) E(v_floor_f32) E(v_ldexp_f32) E(v_div_fixup)
        E(v_sqrt_f32) E(v_rcp_f32) E(v_rsq_f32) E(v_exp_f32) E(v_log_f32) E(v_readlane)
        E(v_pk_add_f32) E(v_pk_mul_f32) E(v_pk_fma_f32) E(add_mul_mix)
    };
    float *out; unsigned long long *cyc;
    const int max_waves = cus * 4 * 8;
    CHECK(hipMalloc(&out, sizeof(float) * max_waves * 64));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * max_waves));
    printf("%-16s", "op");
    const int wps_list[] = {1, 2, 4, 8};
    for (int w : wps_list) printf("  wps=%d cyc/instr/SIMD  ", w);
    printf("\n");
    for (auto &e : entries) {
        printf("%-16s", e.name);
        for (int wps : wps_list) {
            // wps waves per SIMD: blocks of 256 threads (4 waves, one per SIMD), wps blocks per CU
            int blocks = cus * wps;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);  // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(blocks * 4);
            CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            double med = (double)h[h.size() / 2];
            int mult = (std::string(e.name) == "add_mul_mix") ? 2 : 1;
            double n_instr = (double)ITERS * OPS * mult;
            // s_memtime ticks at 100 MHz on this part?  report both raw ticks and wall-derived cycles
            double wall_cyc = ms * 1e-3 * 2.4e9;
            printf("  %6.2f (wall %6.2f)  ", med / n_instr / wps, wall_cyc / n_instr / wps);
        }
        printf("\n");
    }
    return 0;
}
