// Device-side arithmetic of the ray-marching hot path (gfx950).
//
// Compiled with -ffp-contract=off: every +,-,*,/ and sqrt below is one IEEE
// binary32 operation in the order the reference evaluates it, so results match
// a strict CPU evaluation bit for bit.  powf and atan2f are the only library
// calls with a platform-defined last bit.
//
// Citations are file:line in the reference tree (src/...).
#pragma once
#include <hip/hip_runtime.h>
#include "rm_types.hpp"

namespace rm {

#define RM_DEV __device__ __forceinline__

// Diagnostic builds (-DRM_ABL=bits) switch parts of the pixel path off to time the rest:
// 1 soft shadow, 2 reflections, 4 lightmap sampling, 8 everything after the primary march,
// 16 tone mapping, 32 surface normal, 64 object colour, 128 Phong terms.
// The shipped library is built with 0.
#ifndef RM_ABL
#define RM_ABL 0
#endif

// RM_FAST = 1 builds the "fast" tier of this arithmetic (rm_fast.hip, compiled with
// -ffp-contract=fast-honor-pragmas).  A pixel's chain of sight rays stays strict in that tier too:
// pixel direction, march, hit object, surface normal, mirror direction.  The normal is a difference
// of distances 0.0002 apart, which turns a last-bit change of a hit position on a sphere into a change
// of 3e-4 in the normal and, through the mirror, into other texels - the reference's image is a
// chaotic function of those bits, and only the same operations reproduce it.  What a hit's colour is
// made of is continuous in its inputs and relaxed: texture and lightmap coordinates, light terms,
// Phong, the soft-shadow ray's own arithmetic, the sums that put a pixel together, the tone map.  There
// multiplications fuse with additions, roots, reciprocals and quotients come from the hardware's
// one-ulp instructions, powf is exp2(y * log2(x)), and a pixel always packs from the tone map's
// estimate.  Frames of the fast tier are not bit-identical to a strict CPU evaluation; they stay
// within 2/255 per channel with at least 99.9 % of the pixels identical (tests/test_fast_tier_gpu.py).
//
// This file is strict by default in both tiers; the relaxed parts stand between RM_FAST_BEGIN and
// RM_FAST_END and compute with VC, a vector type whose operators may fuse (in the strict tier VC is V3).
#ifndef RM_FAST
#define RM_FAST 0
#endif
#ifndef RM_FAST_PRIMARY
#define RM_FAST_PRIMARY 0
#endif
#if RM_FAST
#pragma clang fp contract(off)
#define RM_FAST_BEGIN _Pragma("clang fp contract(fast)")
#define RM_FAST_END _Pragma("clang fp contract(off)")
#else
#define RM_FAST_BEGIN
#define RM_FAST_END
#endif

// Diagnostic builds (-DRM_DIAG) count march-loop iterations per wave and per lane in LDS; the
// render kernel adds them to the spare statistics slots 4..6.
#ifdef RM_DIAG
RM_DEV unsigned *diag_mem() {
    __shared__ unsigned m[65];
    return m;
}
RM_DEV unsigned *diag_kinds() {
    __shared__ unsigned kinds[8];
    return kinds;
}
RM_DEV unsigned *diag_lane_sums() {  // [1]: lanes in use summed over march-loop iterations, [2]: over shadow-loop iterations
    __shared__ unsigned sums[4];
    return sums;
}
RM_DEV void diag_iter(bool active, int kind = 0) {
    unsigned *m = diag_mem();
    m[threadIdx.x] += active ? 1u : 0u;
    const unsigned in_use = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(active));
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) == threadIdx.x) {
        m[64] += 1u;
        diag_kinds()[kind] += 1u;
        diag_lane_sums()[kind & 3] += in_use;
    }
}
RM_DEV void diag_event(int kind) {  // once per wave however many lanes get here
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) == threadIdx.x) diag_kinds()[kind] += 1u;
}
// The cycles since the previous lap go to section k.
enum { DG_SETUP = 0, DG_MARCH_LOOP, DG_CLOSEST, DG_SHADE, DG_SHADOW, DG_REFLECT, DG_PACK, DG_SECTIONS };
RM_DEV unsigned *diag_laps() {
    __shared__ unsigned t[2 + DG_SECTIONS];
    return t;
}
RM_DEV void diag_lap(int k) {
    unsigned *t = diag_laps();
    const unsigned now = (unsigned)__builtin_readcyclecounter();
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) == threadIdx.x) {  // the first lane still running
        t[2 + k] += now - t[0];
        t[0] = (unsigned)__builtin_readcyclecounter();
    }
}
#else
RM_DEV void diag_iter(bool, int = 0) {}
RM_DEV void diag_event(int) {}
RM_DEV void diag_lap(int) {}
#define DG_SETUP 0
#define DG_MARCH_LOOP 0
#define DG_CLOSEST 0
#define DG_SHADE 0
#define DG_SHADOW 0
#define DG_REFLECT 0
#define DG_PACK 0
#endif

// Wave votes on a predicate that already lives in a lane mask: no detour through a vector register.
RM_DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
RM_DEV bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0; }

// The set of lanes a loop still works for, held as a 64-bit mask in scalar registers: comparisons
// deliver such masks directly, the scalar unit combines them, and the loop's exit test is a scalar
// compare with zero.  lanes_in() turns a mask back into a per-lane predicate for selects.
typedef unsigned long long LaneMask;
RM_DEV LaneMask lanes_here() { return __builtin_amdgcn_ballot_w64(true); }
RM_DEV bool lanes_in(LaneMask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
RM_DEV LaneMask lanes_lt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 4); }  // ordered a < b
RM_DEV LaneMask lanes_gt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 2); }  // ordered a > b

struct V3 {
    float x, y, z;
};

RM_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RM_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RM_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RM_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RM_DEV V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
RM_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RM_DEV V3 emul(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }

// vector.rs:24: ordered lane sum of squares; the fourth lane is 0 whenever the
// other three are not already NaN, so it never changes the sum.
RM_DEV float mag_sqd(V3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
RM_DEV float sqrt_rn(float x);
RM_DEV float mag(V3 a) { return sqrt_rn(mag_sqd(a)); }  // vector.rs:22
// vector.rs:27-30: reciprocal then multiply
RM_DEV bool sqrt_fast_range(float x);
RM_DEV float sqrt_rn_in_range(float x);
RM_DEV float recip_rn_in_range(float b);
RM_DEV V3 normalized(V3 a) {
    const float m = mag_sqd(a);
    // (a root in [2^-48, 2^48) is within the range of the shared-work reciprocal)
    const float inv = wave_all(sqrt_fast_range(m)) ? recip_rn_in_range(sqrt_rn_in_range(m)) : 1.0f / sqrtf(m);
    return a * inv;
}
RM_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }  // vector.rs:32
RM_DEV V3 reflect(V3 v, V3 n) { return v - n * (2.0f * dot(n, v)); }          // vector.rs:38
// vector.rs:40-48; `ior` is the ratio of the two refractive indices.  Only the refraction block of
// engine.rs:196-228 (commented out in the reference) calls it.
RM_DEV V3 refract(V3 v, V3 n, float ior) {
    const float c = dot(v, n);
    const float k = 1.0f - (ior * ior) * (1.0f - c * c);
    if (k < 0.0f) return V3{0.0f, 0.0f, 0.0f};
    return v * ior - n * (ior * c + sqrt_rn(k));
}

// The vector type of the colour arithmetic (see RM_FAST above).
#if RM_FAST
RM_FAST_BEGIN
struct VC {
    float x, y, z;
    VC() = default;
    RM_DEV VC(V3 v) : x(v.x), y(v.y), z(v.z) {}
    RM_DEV operator V3() const { return V3{x, y, z}; }
};
RM_DEV VC vc(float x, float y, float z) {
    VC r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
RM_DEV VC operator+(VC a, VC b) { return vc(a.x + b.x, a.y + b.y, a.z + b.z); }
RM_DEV VC operator-(VC a, VC b) { return vc(a.x - b.x, a.y - b.y, a.z - b.z); }
RM_DEV VC operator*(VC a, float s) { return vc(a.x * s, a.y * s, a.z * s); }
RM_DEV VC operator/(VC a, float s) {
    const float y = __builtin_amdgcn_rcpf(s);
    return vc(a.x * y, a.y * y, a.z * y);
}
RM_DEV VC operator-(VC a) { return vc(-a.x, -a.y, -a.z); }
RM_DEV VC emul(VC a, VC b) { return vc(a.x * b.x, a.y * b.y, a.z * b.z); }
RM_DEV float mag_sqd(VC a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
RM_DEV float dot(VC a, VC b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RM_DEV VC normalized(VC a) { return a * __builtin_amdgcn_rsqf(mag_sqd(a)); }
RM_DEV VC reflect(VC v, VC n) { return v - n * (2.0f * dot(n, v)); }
RM_FAST_END
#else
typedef V3 VC;
RM_DEV VC vc(float x, float y, float z) { return V3{x, y, z}; }
#endif

// Rust f32::clamp keeps a NaN input; f32::min/max drop a NaN operand.
RM_DEV float rs_clamp(float x, float lo, float hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
RM_DEV float rs_max(float a, float b) { return fmaxf(a, b); }
RM_DEV float rs_min(float a, float b) { return fminf(a, b); }
// The same minimum as one bare v_min_f32.  Where an operand reaches fminf across a branch the
// compiler first re-quiets it (v_max_f32 x, x: it cannot see that no signalling NaN exists here,
// arithmetic never produces one); the instruction itself already drops a quiet NaN operand.
RM_DEV float rs_min_bare(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
RM_DEV float rs_max_bare(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

RM_DEV float powi5(float x) {
    float sq = x * x;
    float q = sq * sq;
    return x * q;
}

// Saturating float -> int casts with NaN -> 0, as Rust's `as`.
RM_DEV int sat_i32(float f) {
    if (!(f == f)) return 0;
    if (f <= 0.0f) return 0;
    if (f >= 1073741824.0f) return 1073741824;
    return (int)f;
}

// objects.rs:334, 295, 275, 315, 261; T_BOX: the commented-out box of objects.rs:370-380 with
// centre and half extents as parameters (an extension, evaluated in the order written there)
RM_DEV float sdf_eval(int type, const Geom &g, V3 p) {
    switch (type) {
        case T_SPHERE: return mag(p - v3(g.p[0], g.p[1], g.p[2])) - g.p[3];
        case T_XPLANE: return g.p[1] * (p.x - g.p[0]);
        case T_YPLANE: return g.p[1] * (p.y - g.p[0]);
        case T_ZPLANE: return g.p[1] * (p.z - g.p[0]);
        case T_PLANE: return ((p.x * g.p[0] + p.y * g.p[1]) + p.z * g.p[2]) - g.p[3];
        default: {
            V3 q = v3(fabsf(p.x - g.p[0]) - g.p[3], fabsf(p.y - g.p[1]) - g.p[4], fabsf(p.z - g.p[2]) - g.p[5]);
            V3 r = v3(rs_max(q.x, 0.0f), rs_max(q.y, 0.0f), rs_max(q.z, 0.0f));
            return mag(r) + rs_min(rs_max(q.x, rs_max(q.y, q.z)), 0.0f);
        }
    }
}

// Polynomial smooth minimum for "smooth unions" (an extension: the reference's union is the hard
// minimum of ray.rs:23-33).
// The blend width travels with its reciprocal: h = max(k - gap, 0) / k is then the quotient with
// shared work on the divisor (div_rn_in_range: four operations, the same correctly rounded value as
// the IEEE division's ten).  That form needs every intermediate to stay normal: the dividend lies
// in [0, k] and, where it is not zero, is at least one unit in the last place of k, so widths between
// 2^-60 and 2^60 qualify; any other width keeps the plain division (kr = 0).
struct BlendK {
    float k, kr;
    RM_DEV operator float() const { return k; }
};
RM_DEV float recip_rn_in_range(float b);
RM_DEV float div_rn_in_range(float a, float b, float y);
RM_DEV BlendK blend_k(float k) {
    BlendK b;
    b.k = k;
    b.kr = (k >= 0x1p-60f && k <= 0x1p60f) ? recip_rn_in_range(k) : 0.0f;
    return b;
}
RM_DEV float smooth_min(float a, float b, BlendK k) {
    // |a - b| >= k makes h = 0 / k = 0 and the result min(a, b) - 0: no division needed then
    // (the bare minimum: a and b reach this function across branches and loop back-edges, where fminf
    // would first re-quiet each of them - two instructions more per call, a tenth of a large scene's fold)
    const float gap = fabsf(a - b);
    const float lower = rs_min_bare(a, b);
    if (gap >= k.k) return lower;
    const float num = rs_max(k.k - gap, 0.0f);
    float h = k.kr != 0.0f ? div_rn_in_range(num, k.k, k.kr) : num / k.k;
    return lower - h * h * k.k * (1.0f / 4.0f);
}

// Correctly rounded square root.  v_sqrt_f32 is within one ulp; its residual x - s*s (exact in one
// fused operation) times half the reciprocal root, added to s in a second fused operation, rounds
// to the correctly rounded root: checked against the double-precision root for every float in
// [2^-96, inf) on the GPU (tools/sqrt_exhaustive.hip: no mismatch among 1,879,048,192 inputs).
// Smaller inputs, zeros, infinities, negatives and NaNs take the compiler's full IEEE expansion.
RM_DEV bool sqrt_fast_range(float x) {  // 2^-96 <= x < 2^96, as one unsigned compare of the bit pattern
    return (unsigned)(__float_as_int(x) - 0x0F800000) < (unsigned)(0x6F800000 - 0x0F800000);
}
RM_DEV float sqrt_rn_in_range(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(h, d, s);
}
RM_DEV float sqrt_rn(float x) {
    // One scalar branch for the whole wave: the special inputs practically never
    // occur, and a per-lane branch would cost more mask bookkeeping than it saves.
    if (wave_all(sqrt_fast_range(x))) return sqrt_rn_in_range(x);
    return sqrtf(x);
}

// The root of the colour arithmetic: the hardware's in the fast tier.
RM_DEV float sqrt_colour(float x) { return RM_FAST ? __builtin_amdgcn_sqrtf(x) : sqrt_rn(x); }

// Correctly rounded division for operands in a known range, with the work on the divisor shared
// between dividends.  v_rcp_f32 refined by one Newton step is the correctly rounded reciprocal, and
// a * y corrected once by its exact residual a - b*q is the correctly rounded quotient: both checked
// on the GPU against the compiler's IEEE division for every pair of mantissas (tools/div_exhaustive.hip:
// no mismatch among 2^46 pairs; tools/sqrt_exhaustive.hip for the reciprocal alone).  Powers of two
// scale every intermediate exactly as long as none leaves the normal range, which the callers ensure:
//   recip_rn_in_range(b):    2^-100 <= |b| <= 2^100
//   div_rn_in_range(a, b, y): y = recip_rn_in_range(b), b > 0; a = +-0, or |a| >= 2^-100 with
//                            2^-120 <= |a / b| <= 2^120 (the residual of a smaller dividend is not
//                            representable).  A zero dividend keeps its sign, as in a / b with b > 0:
//                            the correction step alone would turn -0 into +0.
RM_DEV bool float_bits_within(float x, int lo_bits, int hi_bits) {  // lo <= x < hi for positive bounds, one unsigned compare
    return (unsigned)(__float_as_int(x) - lo_bits) < (unsigned)(hi_bits - lo_bits);
}
RM_DEV float recip_rn_in_range(float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
RM_DEV float div_rn_in_range(float a, float b, float y) {
    const float q0 = a * y;
    const float r = __builtin_fmaf(-b, q0, a);
    const float q = __builtin_fmaf(r, y, q0);
    return __builtin_copysignf(q, a);
}
// true where x, y and z are each +-0 or at least 2^-90 in magnitude: twice the magnitude bits
// minus 2 wraps a zero around to the top of the unsigned range
RM_DEV bool zero_or_above_2m90(float x, float y, float z) {
    const unsigned tx = ((unsigned)__float_as_int(x) << 1) + 0xfffffffeu;
    const unsigned ty = ((unsigned)__float_as_int(y) << 1) + 0xfffffffeu;
    const unsigned tz = ((unsigned)__float_as_int(z) << 1) + 0xfffffffeu;
    const unsigned lowest = min(tx, min(ty, tz));
    return lowest >= (0x12800000u << 1) - 2u;  // 0x12800000 = 2^-90
}

// ---------------------------------------------------------------------------
// Scene access for the march loops: objects grouped by type.  `Small` is indexed
// by unrolled constants out of the kernel arguments (scalar registers); `Large`
// walks global arrays with wave-uniform addresses (scalar loads).
// ---------------------------------------------------------------------------

template <class G>
struct SceneSmallAcc {
    const G &g;
    template <class F>
    RM_DEV void spheres(F &&f) const {
#pragma unroll
        for (int i = 0; i < G::kSpheres; i++)
            if (G::kExact || i < g.n_sph) f(i, g.idx_sph[i], g.sph[i][0], g.sph[i][1], g.sph[i][2], g.sph[i][3]);
    }
    template <int AXIS, class F>
    RM_DEV void axis(F &&f) const {
        const int n = AXIS == 0 ? g.n_x : (AXIS == 1 ? g.n_y : g.n_z);
#pragma unroll
        for (int i = 0; i < G::kAxis; i++)
            if (G::kExact || i < n) f(g.idx_ax[AXIS][i], g.ax[AXIS][i][0], g.ax[AXIS][i][1]);
    }
    template <class F>
    RM_DEV void planes(F &&f) const {
#pragma unroll
        for (int i = 0; i < G::kPlanes; i++)
            if (G::kExact || i < g.n_gp) f(g.idx_gp[i], g.gp[i][0], g.gp[i][1], g.gp[i][2], g.gp[i][3]);
    }
    // the four groups of planes in the order of the fold
    template <class FX, class FY, class FZ, class FP>
    RM_DEV void all_planes(FX &&fx, FY &&fy, FZ &&fz, FP &&fp) const {
        axis<0>(fx);
        axis<1>(fy);
        axis<2>(fz);
        planes(fp);
    }
    template <class F>
    RM_DEV void boxes(F &&) const {}  // scenes with boxes or smooth unions use the large layout
    static constexpr bool kSmooth = false;
    static constexpr bool kGrid = false;
    static constexpr int kSphereSlots = G::kSpheres;  // spheres() hands out indices below this
    RM_DEV float sphere_extent() const { return g.sph_extent; }
    static constexpr bool kUnitAxis = G::kExact;  // the exact-fill tier also promises |dir| = 1 on every axis plane
    RM_DEV float smooth_k() const { return 0.0f; }
};

// Scene geometry does not change while a kernel runs.  Reading it through the constant address
// space tells the compiler so: loads at wave-uniform addresses become scalar loads (one fetch per
// wave, into scalar registers) even where the kernel's own atomics and stores would otherwise make
// it assume the data might have changed.
typedef const __attribute__((address_space(4))) float *ConstFloats;
typedef const __attribute__((address_space(4))) int *ConstInts;
typedef float Float4 __attribute__((ext_vector_type(4)));  // 16-byte loads
typedef int Int4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) Float4 *ConstFloat4s;
typedef const __attribute__((address_space(4))) Int4 *ConstInt4s;
RM_DEV ConstFloats constant(const float *p) { return (ConstFloats)p; }
RM_DEV ConstInts constant(const int *p) { return (ConstInts)p; }

struct SceneLargeAcc {
    const GroupedLarge &g;
    template <class F>
    RM_DEV void spheres(F &&f) const {
        ConstInts idx = constant(g.idx_sph);
        ConstFloats s = constant(g.sph);
        for (int i = 0; i < g.n_sph; i++) f(i, idx[i], s[4 * i], s[4 * i + 1], s[4 * i + 2], s[4 * i + 3]);
    }
    template <int AXIS, class F>
    RM_DEV void axis(F &&f) const {
        const int n = AXIS == 0 ? g.n_x : (AXIS == 1 ? g.n_y : g.n_z);
        ConstInts idx = constant(g.idx_ax[AXIS]);
        ConstFloats a = constant(g.ax[AXIS]);
        for (int i = 0; i < n; i++) f(idx[i], a[2 * i], a[2 * i + 1]);
    }
    template <class F>
    RM_DEV void planes(F &&f) const {
        ConstInts idx = constant(g.idx_gp);
        ConstFloats gp = constant(g.gp);
        for (int i = 0; i < g.n_gp; i++) f(idx[i], gp[4 * i], gp[4 * i + 1], gp[4 * i + 2], gp[4 * i + 3]);
    }
    // the same four groups out of the one array of records (a scalar load and a scalar branch per plane)
    template <class FX, class FY, class FZ, class FP>
    RM_DEV void all_planes(FX &&fx, FY &&fy, FZ &&fz, FP &&fp) const {
        typedef float Float8 __attribute__((ext_vector_type(8)));
        typedef const __attribute__((address_space(4))) Float8 *ConstFloat8s;
        const ConstFloat8s rec = (ConstFloat8s)g.planes_all;
        for (int i = 0; i < g.n_planes_all; i++) {
            const Float8 r = rec[i];
            const int kind = __float_as_int(r[0]), oi = __float_as_int(r[1]);
            if (kind == 0)
                fx(oi, r[2], r[3]);
            else if (kind == 1)
                fy(oi, r[2], r[3]);
            else if (kind == 2)
                fz(oi, r[2], r[3]);
            else
                fp(oi, r[2], r[3], r[4], r[5]);
        }
    }
    template <class F>
    RM_DEV void boxes(F &&f) const {
        ConstInts idx = constant(g.idx_box);
        for (int i = 0; i < g.n_box; i++) {
            ConstFloats b = constant(g.box) + 8 * i;
            f(idx[i], v3(b[0], b[1], b[2]), v3(b[3], b[4], b[5]));
        }
    }
    static constexpr bool kSmooth = true;
    static constexpr bool kGrid = true;
    static constexpr int kSphereSlots = 0;
    RM_DEV const GridView &grid() const { return g.grid; }
    RM_DEV float sphere_extent() const { return g.sph_extent; }
    static constexpr bool kUnitAxis = false;
    RM_DEV float smooth_k() const { return g.smooth_k; }
};

template <int NS, int NA, int NG, bool EX>
RM_DEV SceneSmallAcc<GroupedSmall<NS, NA, NG, EX>> acc_of(const GroupedSmall<NS, NA, NG, EX> &g) {
    return SceneSmallAcc<GroupedSmall<NS, NA, NG, EX>>{g};
}
RM_DEV SceneLargeAcc acc_of(const GroupedLarge &g) { return SceneLargeAcc{g}; }

// Which object a march leaves out (`ignore_object` of ray.rs:15,88).
struct NoIgnore {
    RM_DEV bool skip(int) const { return false; }
    RM_DEV float mask(int, float s) const { return s; }
};
struct UniformIgnore {  // the same object in every lane: left out by a scalar branch
    int u;
    RM_DEV bool skip(int oi) const { return oi == u; }
    RM_DEV float mask(int, float s) const { return s; }
};
struct LaneIgnore {  // differs between lanes: evaluated, then masked
    int v;
    RM_DEV bool skip(int) const { return false; }
    RM_DEV float mask(int oi, float s) const { return oi == v ? __builtin_inff() : s; }
};

// A lower bound on the distance to the nearest sphere, carried along one march.
//
// Distances are 1-Lipschitz: after moving a path of length L, every sphere is still at least
// S - L away, S being the smallest sphere distance where it was last evaluated.  While the running
// minimum d over the other objects stays below that, no sphere can change it and the whole group
// is left out of the step.  With t the distance travelled as the march itself sums it, the group
// is skipped while
//
//     d + t * dn  <=  T,      T = (S - M) + t_e * dn    (fixed at the evaluation, t_e = t there)
//
// dn bounds the path length per unit of t, roundings included.  Every step that is followed by
// another evaluation is at least SMALL_DISTANCE long, so per-step errors are at most 1e3 times
// the step: with P >= every |coordinate| reached,
//   * the step vector dir*d is at most 1.00001 |dir| d long;
//   * a position update leaves the ideal segment by at most 3.1e-7 P        <= 3.1e-4 P d;
//   * an addition to t rounds by at most 2^-24 * 64                         <= 3.8e-3 d,
// so L <= (t - t_e) * 1.004 * (1.00001 |dir| + 3.1e-4 P) <= (t - t_e) * dn with
// dn = 1.0041 |dir| + 3.2e-4 P.  M takes the errors that do not grow with the path: two computed
// sphere distances, each within 1.1e-6 (P + C) of the true one (C the largest |centre coordinate|
// + radius; this includes the one-ulp square root the bound is built from), and the two fused
// multiply-adds above, 4e-6 dn each: M = 1e-5 (P + C) + 1e-4 (1 + dn).
// A NaN anywhere makes the comparison fail, which means "evaluate".
struct SphereBound {
    float T;    // per lane
    float dn;   // per ray
    float M;
    float eps;  // 2e-6 (P + C): more than any sphere distance differs from its one-ulp estimate
    // The lane's last evaluation left the spheres out: each is then strictly farther than the value
    // found (M exceeds the rounding it covers), so none of them can be the closest object either.
    LaneMask spheres_out;
};

template <class SC>
RM_DEV SphereBound make_bound(const SC &sc, V3 o, V3 dir, float tmax) {
    SphereBound b;
    b.T = -__builtin_inff();
    b.spheres_out = 0;
    const float amax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
    const float dmax = fmaxf(fmaxf(fabsf(dir.x), fabsf(dir.y)), fabsf(dir.z));
    const float P = (amax + tmax * dmax) * 1.001f + 1.0f;
    const float dd = (dir.x * dir.x + dir.y * dir.y) + dir.z * dir.z;
    b.dn = __builtin_amdgcn_sqrtf(dd) * 1.0041f + 3.2e-4f * P;
    b.M = 1e-5f * (P + sc.sphere_extent()) + 1e-4f * (1.0f + b.dn);
    b.eps = 2e-6f * (P + sc.sphere_extent());
    return b;
}

// One sphere (box) of a block, fetched by a lane of its own.
struct SphereEntry {
    Float4 g;
    int oi;
};
RM_DEV SphereEntry grid_sphere(const GridView &gv, int s_at, int j) {
    const ConstInts group = constant(gv.pool) + s_at + kGridSphereGroup * (j >> 2);
    SphereEntry e;
    e.g = *(ConstFloat4s)(group + 4 * (j & 3));
    e.oi = group[16 + (j & 3)];
    return e;
}
struct BoxEntry {
    Float4 g0, g1;  // cx, cy, cz, hx | hy, hz, object index, -
};
RM_DEV BoxEntry grid_box(const GridView &gv, int b_at, int j) {
    const ConstFloat4s rec = (ConstFloat4s)(constant(gv.pool) + b_at + 8 * j);
    BoxEntry e;
    e.g0 = rec[0];
    e.g1 = rec[1];
    return e;
}

typedef float Float16 __attribute__((ext_vector_type(16)));
// (scalar loads need dword alignment only; the element type says so, lest a wider alignment be assumed)
typedef Float16 __attribute__((aligned(4))) Float16u;
typedef Int4 __attribute__((aligned(4))) Int4u;
typedef const __attribute__((address_space(4))) Float16u *ConstFloat16s;
typedef const __attribute__((address_space(4))) Int4u *ConstInt4us;
typedef int Int8 __attribute__((ext_vector_type(8)));
typedef Int8 __attribute__((aligned(4))) Int8u;
typedef const __attribute__((address_space(4))) Int8u *ConstInt8us;

// The march family: strict here, and once more with relaxed arithmetic for the fast tier.
constexpr bool kRelaxedMarch = false;
#include "rm_march.inc"
#if RM_FAST
RM_FAST_BEGIN
namespace relaxed {
typedef VC V3;  // (inside this namespace the march family computes with the vector type whose operators may fuse)
RM_DEV VC v3(float x, float y, float z) { return vc(x, y, z); }
RM_DEV float sqrt_rn(float x) { return __builtin_amdgcn_sqrtf(x); }
RM_DEV float smooth_min(float a, float b, BlendK bk) {
    const float k = bk.k;
    const float gap = fabsf(a - b);
    const float lower = rs_min_bare(a, b);
    if (gap >= k) return lower;
    const float h = rs_max(k - gap, 0.0f) * __builtin_amdgcn_rcpf(k);
    return lower - h * h * k * (1.0f / 4.0f);
}
constexpr bool kRelaxedMarch = true;
#include "rm_march.inc"
// (the callers hold rm::V3 positions)
template <class SC>
RM_DEV int march(const SC &sc, rm::V3 &pos, rm::V3 dir, int ignore, int cap, int &steps, bool &capped, float first = __builtin_nanf(""),
                 float first_sph = __builtin_nanf("")) {
    VC p = pos;
    const int idx = march(sc, p, VC(dir), ignore, cap, steps, capped, first, first_sph);
    pos = p;
    return idx;
}
}  // namespace relaxed
RM_FAST_END
#else
namespace relaxed = ::rm;  // one arithmetic in the strict tier
#endif

// objects.rs:135-141 for a per-lane object.  The evaluations sit inside one dispatch on the type,
// so a wave runs them only for the types its lanes hold.  `tetra` (the same in every lane) selects
// the four-point form the reference carries commented out (objects.rs:18-24, 142-149): the sum of
// step vector times distance over the corners of a tetrahedron of half edge 0.0015.
constexpr float TETRA_STEP = 0.0015f;
template <int TYPE>
RM_DEV V3 sdf_gradient(const Geom &g, V3 p, bool tetra) {
    if (tetra) {
        const float T = TETRA_STEP;
        const float sa = sdf_eval(TYPE, g, v3(p.x + T, p.y + -T, p.z + -T));
        const float sb = sdf_eval(TYPE, g, v3(p.x + -T, p.y + -T, p.z + T));
        const float sc = sdf_eval(TYPE, g, v3(p.x + -T, p.y + T, p.z + -T));
        const float sd = sdf_eval(TYPE, g, v3(p.x + T, p.y + T, p.z + T));
        return ((v3(T, -T, -T) * sa + v3(-T, -T, T) * sb) + v3(-T, T, -T) * sc) + v3(T, T, T) * sd;
    }
    float gx = sdf_eval(TYPE, g, v3(p.x + STEP_SIZE, p.y + 0.0f, p.z + 0.0f)) -
               sdf_eval(TYPE, g, v3(p.x - STEP_SIZE, p.y - 0.0f, p.z - 0.0f));
    float gy = sdf_eval(TYPE, g, v3(p.x + 0.0f, p.y + STEP_SIZE, p.z + 0.0f)) -
               sdf_eval(TYPE, g, v3(p.x - 0.0f, p.y - STEP_SIZE, p.z - 0.0f));
    float gz = sdf_eval(TYPE, g, v3(p.x + 0.0f, p.y + 0.0f, p.z + STEP_SIZE)) -
               sdf_eval(TYPE, g, v3(p.x - 0.0f, p.y - 0.0f, p.z - STEP_SIZE));
    return v3(gx, gy, gz);
}
RM_DEV V3 calculate_normal(int type, const Geom &g, V3 p, bool tetra) {
    V3 grad;
    switch (type) {
        case T_SPHERE: grad = sdf_gradient<T_SPHERE>(g, p, tetra); break;
        case T_XPLANE: grad = sdf_gradient<T_XPLANE>(g, p, tetra); break;
        case T_YPLANE: grad = sdf_gradient<T_YPLANE>(g, p, tetra); break;
        case T_ZPLANE: grad = sdf_gradient<T_ZPLANE>(g, p, tetra); break;
        case T_PLANE: grad = sdf_gradient<T_PLANE>(g, p, tetra); break;
        default: grad = sdf_gradient<T_BOX>(g, p, tetra); break;
    }
    return normalized(grad);
}

// ---- what a hit's colour is made of: relaxed in the fast tier (see RM_FAST) ----
RM_FAST_BEGIN

// objects.rs:285-290, 305-310, 325-330, 353-358
RM_DEV void sample_uv_from_pos(int type, const Geom &g, VC pos, float ms, float &u, float &v) {
    const float half = ms / 2.0f;
    switch (type) {
        case T_SPHERE: {
            VC n = normalized(pos - vc(g.p[0], g.p[1], g.p[2]));
            float uu = 0.5f + atan2f(n.x, -n.z) / 6.28318530717958647692f;
            float vv = 0.5f + (n.y / 2.0f);
            u = uu * ms;
            v = vv * ms;
            return;
        }
        case T_XPLANE:
            u = pos.y + half - 0.5f;
            v = pos.z + half - 0.5f;
            return;
        case T_YPLANE:
            u = pos.x + half - 0.5f;
            v = pos.z + half - 0.5f;
            return;
        case T_ZPLANE:
            u = pos.x + half - 0.5f;
            v = pos.y + half - 0.5f;
            return;
        default: {  // T_PLANE (no counterpart in the reference, objects.rs:42): p = normal, distance, tangent, sign of the bitangent
            const VC n = vc(g.p[0], g.p[1], g.p[2]), t = vc(g.p[4], g.p[5], g.p[6]);
            const VC b = vc(n.y * t.z - n.z * t.y, n.z * t.x - n.x * t.z, n.x * t.y - n.y * t.x) * g.p[7];
            const VC rel = pos - n * g.p[3];
            u = dot(rel, t) + half - 0.5f;
            v = dot(rel, b) + half - 0.5f;
            return;
        }
    }
}

// How the kernels read the SceneView: through the constant address space.  Nothing writes the view
// while a kernel runs, and a field read this way arrives by a scalar load (once per wave, through the
// scalar cache, held in a scalar register) instead of a vector load that every march and every
// shading pass would wait for; the pointers inside it, coming out of constant memory, are known to
// point to global memory, so the per-pixel records, luxels and texels behind them are fetched by
// global loads, not by generic ones.
typedef const __attribute__((address_space(4))) SceneView &SceneRef;
RM_DEV SceneRef scene_ref(const SceneView *p) { return *(const __attribute__((address_space(4))) SceneView *)p; }
// The same view through a copy of its address that the compiler cannot trace back: the fields read
// after this point are fetched here (a scalar load each, issued together), not at the top of the kernel
// and carried in scalar registers through the march loops, which have none to spare.
RM_DEV SceneRef scene_here(SceneRef sv) {
    const __attribute__((address_space(4))) SceneView *p = &sv;
    asm volatile("" : "+s"(p));
    return *p;
}

// texture.rs:31-37
RM_DEV VC texture_sample(const TextureRec &t, float u, float v) {
    float ax = floorf(fabsf(u * t.uscale)), ay = floorf(fabsf(v * t.vscale));
    float fw = (float)t.w, fh = (float)t.h;
    float fx, fy;
    if (t.inv_w != 0.0f) {
        // x mod 2^k for a non-negative integer-valued x: every step below is exact
        // (the quotient by a power of two is the product with its reciprocal, exactly: ax and ay are
        // whole numbers, infinite or NaN)
        fx = ax - fw * floorf(ax * t.inv_w);
        fy = ay - fh * floorf(ay * t.inv_h);
    } else {
        fx = fmodf(ax, fw);
        fy = fmodf(ay, fh);
    }
    int x = sat_i32(fx), y = sat_i32(fy);
    x = min(x, t.w - 1);  // fmod keeps both below w/h; NaN maps to 0
    y = min(y, t.h - 1);
    const float *px = t.texels + 4 * (size_t)(x + y * t.w);
    return vc(px[0], px[1], px[2]);
}

// objects.rs:260, 267-274, 294, 314, 336
RM_DEV VC object_colour(SceneRef sv, const ShadeRec &sr, const Geom &g, VC pos) {
    if (sr.textured) {
        float u, v;
        sample_uv_from_pos(T_YPLANE, g, pos, (float)sv.map_size, u, v);
        return texture_sample(sv.tex[sr.texture], u, v);
    }
    return vc(sr.cr, sr.cg, sr.cb);
}

// The four luxels around a sample.  Indices outside the map (the reference aborts on them) are
// clamped and reported; the check is made once for the four, the clamps only when it fails.
RM_DEV void luxels4(SceneRef sv, const float *lm, int u0, int u1, int v0, int v1, bool &oob, VC &s00, VC &s01,
                    VC &s10, VC &s11) {
    const int M = sv.map_size;
    const unsigned m = (unsigned)M;
    if (!(((unsigned)u0 < m) & ((unsigned)u1 < m) & ((unsigned)v0 < m) & ((unsigned)v1 < m))) {
        oob = true;
        u0 = min(max(u0, 0), M - 1);
        u1 = min(max(u1, 0), M - 1);
        v0 = min(max(v0, 0), M - 1);
        v1 = min(max(v1, 0), M - 1);
    }
    const float *r0 = lm + 3 * (u0 * M), *r1 = lm + 3 * (u1 * M);
    const float *p00 = r0 + 3 * v0, *p01 = r0 + 3 * v1, *p10 = r1 + 3 * v0, *p11 = r1 + 3 * v1;
    s00 = vc(p00[0], p00[1], p00[2]);
    s01 = vc(p01[0], p01[1], p01[2]);
    s10 = vc(p10[0], p10[1], p10[2]);
    s11 = vc(p11[0], p11[1], p11[2]);
}

// `mag_sqd() == 0.0` (objects.rs:78-113) without the squares: the sum of three squares is zero exactly
// when each square rounds to zero, which it does up to |x| = 2^-75 (x * x = 2^-150 lies halfway
// between 0 and the smallest denormal and goes to the even neighbour, 0).  A NaN fails both forms.
RM_DEV bool no_light(VC s) {
    const float t = 0x1p-75f;
    return fabsf(s.x) <= t && fabsf(s.y) <= t && fabsf(s.z) <= t;
}

// objects.rs:45-133 with colour.rs:47-71 (the two sorts there leave the four
// corners in the order they are passed in, because u0 < u1 and v0 < v1 always).
RM_DEV VC sample_lightmap(SceneRef sv, const ShadeRec &sr, const Geom &g, VC pos) {
    const int M = sv.map_size;
    const float *lm = sv.lightmaps + (size_t)sr.lm_slot * M * M * 3;
    float u, v;
    sample_uv_from_pos(sr.type, g, pos, (float)M, u, v);

    int u0 = sat_i32(rs_max(floorf(u), 0.0f));
    int v0 = sat_i32(rs_max(floorf(v), 0.0f));
    int u1 = u0 + 1, v1 = v0 + 1;
    u0 = (u0 >= M) ? u0 - 1 : u0;
    v0 = (v0 >= M) ? v0 - 1 : v0;
    u1 = (u1 >= M) ? u0 : u1;
    v1 = (v1 >= M) ? v0 : v1;

    bool oob = false;
    VC s00, s01, s10, s11;
    luxels4(sv, lm, u0, u1, v0, v1, oob, s00, s01, s10, s11);

    bool resample = false;
    if (no_light(s00)) {
        resample = true;
        if (no_light(s10)) {
            v0 += 1;
            v1 += 1;
        } else if (no_light(s11)) {
            v0 -= 1;
            v1 -= 1;
        }
        if (no_light(s01)) {
            u0 += 1;
            u1 += 1;
        } else if (no_light(s11)) {
            u0 -= 1;
            u1 -= 1;
        }
    } else {
        if (no_light(s10)) {
            resample = true;
            u1 -= 1;
        }
        if (no_light(s01)) {
            resample = true;
            v1 -= 1;
        }
    }
    if (resample) luxels4(sv, lm, u0, u1, v0, v1, oob, s00, s01, s10, s11);
    if (oob) atomicAdd(&sv.stats[2], 1ULL);
    if (u0 == u1) u1 += 1;
    if (v0 == v1) v1 += 1;

    // colour.rs:52-70 with (x1,y1)=(u0,v0): q11=s00, q12=s01, q21=s10, q22=s11
    float x1 = (float)u0, x2 = (float)u1, y1 = (float)v0, y2 = (float)v1;
    VC acc = (((s00 * (x2 - u)) * (y2 - v)) + ((s10 * (u - x1)) * (y2 - v))) + ((s01 * (x2 - u)) * (v - y1));
    acc = acc + ((s11 * (u - x1)) * (v - y1));
    // the cell is one luxel wide and high, so the divisor is exactly 1 (x / 1 = x); other values can
    // only come from indices the reference would have aborted on
    const float denom = (x2 - x1) * (y2 - y1);
    return denom == 1.0f ? acc : acc / denom;
}

// colour.rs:74-98
// (`li` = intensity / (dist * dist), colour.rs:83, formed by the caller)
RM_DEV void phong_ds(VC n, VC l, float li, const ShadeRec &m, VC view, float &diffuse, float &specular) {
    VC r = reflect(l, n);
    diffuse = m.diffuse * li * rs_max(dot(l, n), 0.0f);
    specular = 0.0f;
    if (diffuse > 0.0f) {
        float base = rs_max(dot(r, view), 0.0f);
        // 0 * li * powf(base, shininess) is +0 whenever li and the power are finite
        // and positive; only then is the call left out.
        bool zero = (m.specular == 0.0f) && (m.shininess >= 0.0f) && (li > 0.0f) && (li < __builtin_inff()) &&
                    (base < __builtin_inff());
        if (!zero) {
            // (fast tier: exp2(y * log2(x)) from the hardware's one-ulp functions; 0^0 = 1 as powf has it)
            const float p = RM_FAST ? (m.shininess == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(m.shininess * __builtin_amdgcn_logf(base)))
                                    : powf(base, m.shininess);
            specular = m.specular * li * p;
        }
    }
}

// engine.rs:171-173 and colour.rs:83: dist = |to_light|, to_light / dist (three divisions by one
// divisor), li = intensity / (dist * dist).  A wave whose lanes all hold values in the usual range
// (distances between 2^-20 and 2^20, a light intensity between 2^-40 and 2^40, no vector component
// below 2^-90 other than zero) shares the work on each divisor; any other wave takes the compiler's
// IEEE divisions.  Both deliver the same correctly rounded values.
RM_DEV void light_terms(VC &to_light, float intensity, float &dist, float &li) {
    const float m = mag_sqd(to_light);
    if (RM_FAST) {
        const float y = __builtin_amdgcn_rsqf(m);
        dist = m * y;
        to_light = to_light * y;
        li = intensity * (y * y);
        return;
    }
    // (bitwise on purpose: three cheap tests, no short-circuit branches)
    const bool usual = (int)float_bits_within(m, 0x2B800000, 0x53800000) &            // 2^-40 <= m < 2^40
                       (int)float_bits_within(intensity, 0x2B800000, 0x53800000) &    // (the same in every lane)
                       (int)zero_or_above_2m90(to_light.x, to_light.y, to_light.z);
    if (wave_all(usual)) {
        dist = sqrt_rn_in_range(m);
        const float y = recip_rn_in_range(dist);
        to_light = vc(div_rn_in_range(to_light.x, dist, y), div_rn_in_range(to_light.y, dist, y), div_rn_in_range(to_light.z, dist, y));
        const float d2 = dist * dist;
        li = div_rn_in_range(intensity, d2, recip_rn_in_range(d2));
    } else {
        dist = sqrt_rn(m);
        to_light = to_light / dist;
        li = intensity / (dist * dist);
    }
}

struct ShadeResult {
    VC final_colour;  // engine.rs:194
    VC object_colour;
    V3 n;  // (strict in both tiers: the mirror ray is built from it)
    float reflectivity;
};

// engine.rs:147-194: everything of shade_object before the reflection branch.
template <class SC>
RM_DEV ShadeResult shade_local(const SC &sc, SceneRef sv_of_kernel, int idx, V3 &pos, V3 &dir) {  // (pos and dir come back as they went in: by reference so that a caller need not keep copies across the shadow ray)
    SceneRef sv = scene_here(sv_of_kernel);
    const ShadeRec sr = sv.shade[idx];
    const Geom g = sv.geom[idx];
    ShadeResult out;
    VC oc = (RM_ABL & 64) ? vc(sr.cr, sr.cg, sr.cb) : object_colour(sv, sr, g, pos);
    V3 n = (RM_ABL & 32) ? v3(0.0f, 1.0f, 0.0f) : calculate_normal(sr.type, g, pos, sv.tetra_normals != 0);

    VC ambient;
    if ((RM_ABL & 4) || sr.lm_slot < 0)
        ambient = oc * sr.ambient;  // engine.rs:158
    else
        ambient = sample_lightmap(sv, sr, g, pos);  // engine.rs:159
    if (mag_sqd(ambient) < sr.ambient * sr.ambient) ambient = normalized(ambient) * sr.ambient;  // engine.rs:163-165
    ambient = emul(ambient, oc);                                                                  // engine.rs:168

    VC light = vc(sv.light[0], sv.light[1], sv.light[2]);
    VC to_light = light - VC(pos);
    float dist, li;
    light_terms(to_light, sv.intensity, dist, li);

    float diffuse = 0.5f, specular = 0.0f;
    if (!(RM_ABL & 128)) phong_ds(n, to_light, li, sr, dir, diffuse, specular);
    diag_lap(DG_SHADE);
    // engine.rs:194 multiplies the shade by diffuse + specular: where that is +-0 (the surface faces
    // away from the light) the product is that zero for any shade - a finite value in [0, 1] - and the
    // shadow ray is not cast
    const bool lit = !(diffuse + specular == 0.0f);
    // (the shade is a continuous function of the ray's arithmetic: the fast tier marches it relaxed)
    float ds = diffuse + specular;
    float reflectivity = sr.reflectivity;
#if !defined(RM_NO_STASH)
    // Large scenes: the march loops need every register the kernel has at seven waves per SIMD, and
    // what this function holds across the shadow ray would otherwise be spilled to private memory -
    // HBM traffic of about 100 bytes per pixel.  These values wait in LDS instead (a wave is a
    // workgroup of its own and nothing else uses LDS: 4 KiB per wave, 112 KiB per compute unit).
    constexpr bool kStash = SC::kGrid;
    __shared__ float stash[kStash ? 16 : 1][64];
    const int stash_lane = threadIdx.x & 63;
    if constexpr (kStash) {
        stash[0][stash_lane] = oc.x;
        stash[1][stash_lane] = oc.y;
        stash[2][stash_lane] = oc.z;
        stash[3][stash_lane] = ambient.x;
        stash[4][stash_lane] = ambient.y;
        stash[5][stash_lane] = ambient.z;
        stash[6][stash_lane] = n.x;
        stash[7][stash_lane] = n.y;
        stash[8][stash_lane] = n.z;
        stash[9][stash_lane] = ds;
        stash[10][stash_lane] = dir.x;
        stash[11][stash_lane] = dir.y;
        stash[12][stash_lane] = dir.z;
        stash[13][stash_lane] = pos.x;
        stash[14][stash_lane] = pos.y;
        stash[15][stash_lane] = pos.z;
        asm volatile("" ::: "memory");
    }
#endif
    float shade = (RM_ABL & 1) ? 1.0f : relaxed::smooth_shadow_march(sc, pos, to_light, idx, dist, SHADOW_K, lit, sv.light_clear);
#if !defined(RM_NO_STASH)
    if constexpr (kStash) {
        asm volatile("" ::: "memory");
        oc = vc(stash[0][stash_lane], stash[1][stash_lane], stash[2][stash_lane]);
        ambient = vc(stash[3][stash_lane], stash[4][stash_lane], stash[5][stash_lane]);
        n = v3(stash[6][stash_lane], stash[7][stash_lane], stash[8][stash_lane]);
        ds = stash[9][stash_lane];
        dir = v3(stash[10][stash_lane], stash[11][stash_lane], stash[12][stash_lane]);
        pos = v3(stash[13][stash_lane], stash[14][stash_lane], stash[15][stash_lane]);
    }
#endif

    diag_lap(DG_SHADOW);
    out.final_colour = ambient + oc * (shade * ds);
#if defined(RM_DIAG_PROOF)
    {   // picture of the shadow proof: green where the ray is proven open, blue where the whole wave is, red where marched, grey where no ray is wanted
        const bool open = relaxed::shadow_stays_open(sc, pos, to_light, LaneIgnore{idx}, dist, SHADOW_K, sv.light_clear);
        const bool all = wave_all(open || !lit);
        out.final_colour = !lit ? vc(0.2f, 0.2f, 0.2f) : all ? vc(0.0f, 0.2f, 1.0f) : open ? vc(0.0f, 1.0f, 0.0f) : vc(1.0f, 0.0f, 0.0f);
        reflectivity = 0.0f;
        out.reflectivity = 0.0f;
    }
#endif
    out.object_colour = oc;
    out.n = n;
    out.reflectivity = reflectivity;
    return out;
}

// engine.rs:129-145 + 231-241 with the recursion turned into an explicit stack;
// frames unwind innermost first, which keeps every rounding of the recursion.
// `depth0` = the number of reflective hits already above this ray (0 for a ray from the camera): the
// chain is cut where the recursion as a whole is MAX_DEPTH deep.
template <class SC>
RM_DEV V3 cast_sight_ray(const SC &sc, SceneRef sv, V3 pos, V3 dir, float first = __builtin_nanf(""),
                         float first_sph = __builtin_nanf(""), int depth0 = 0) {
    // (The frames, 28 bytes each, live in private memory from the third on.  Keeping the first frame in
    // registers was measured: 80 instead of 72 vector registers, 6 instead of 7 waves per SIMD, and the
    // frame 2 % slower.)
    float stk[MAX_DEPTH][7];
    // The first two frames of the stack - all that most reflecting pixels ever open - wait in LDS
    // (a wave is a workgroup of its own: 3.5 KiB for two frames of 64 lanes, 112 KiB per compute unit at
    // eight waves per SIMD): the kernel's HBM traffic beyond the pixels it writes was these frames'
    // (3840x2160: 100 -> 42 MB per launch for 33 MB of pixels).  Large scenes keep their LDS for the
    // values held across the shadow ray.
#if defined(RM_NO_LDS_STACK)
    constexpr int kLdsFrames = 0;
#else
    constexpr int kLdsFrames = SC::kGrid ? 0 : 2;
#endif
    __shared__ float lds_stk[kLdsFrames > 0 ? kLdsFrames : 1][7][64];
    const int stk_lane = threadIdx.x & 63;
    int depth = 0;
    VC colour;
    for (;;) {
        int steps;
        bool capped;
        diag_lap(DG_REFLECT);
        diag_event(5);
        int idx;
#if RM_FAST && RM_FAST_PRIMARY
        // Fast tier: the ray from the camera is first marched with the relaxed arithmetic.  Where it ends
        // on a matt axis plane - whose normal is the axis whatever the last bits of the hit position
        // are - everything that follows is a continuous function of that position, and the relaxed
        // result stands.  Any other hit (a sphere, a box, a generic plane: normals that amplify the
        // position's last bits; anything that reflects: the chain goes on from there) is marched
        // again with the strict operations, for those lanes alone.  Mirror rays are always strict.
        if (depth == 0 && depth0 == 0) {
            const V3 start = pos;
            idx = relaxed::march(sc, pos, dir, -1, sv.march_cap, steps, capped, first, first_sph);
            bool again = false;
            if (idx >= 0) {
                const int type = sv.shade[idx].type;
                again = !(type == T_XPLANE || type == T_YPLANE || type == T_ZPLANE) || sv.shade[idx].reflectivity > 1e-3f;
            }
            if (wave_any(again)) {
                if (again) {
                    V3 p = start;
                    idx = march(sc, p, dir, -1, sv.march_cap, steps, capped, first, first_sph);
                    pos = p;
                }
            }
        } else
#endif
            idx = march<true>(sc, pos, dir, -1, sv.march_cap, steps, capped, first, first_sph);
        first = __builtin_nanf("");  // known for the ray from the camera only
        if (capped) atomicAdd(&sv.stats[1], 1ULL);
        if (idx < 0) {
            colour = vc(sv.sky[0], sv.sky[1], sv.sky[2]);
            break;
        }
        if (RM_ABL & 8) {
            colour = vc(pos.x * 0.1f + 0.5f, pos.y * 0.1f + 0.5f, (float)idx * 0.1f);
            break;
        }
        ShadeResult r = shade_local(sc, sv, idx, pos, dir);
        if (!(RM_ABL & 2) && r.reflectivity > 1e-3f) {
            if (depth + depth0 >= MAX_DEPTH) {
                atomicAdd(&sv.stats[0], 1ULL);
                colour = r.final_colour;
                break;
            }
            float fresnel = powi5(rs_clamp(1.0f - dot(r.n, -dir), 0.0f, 1.0f));
            V3 rv = reflect(dir, r.n);
            float kk = rs_clamp(fresnel + r.reflectivity, 0.0f, 1.0f);
            const float frame[7] = {r.final_colour.x, r.final_colour.y, r.final_colour.z, r.object_colour.x, r.object_colour.y, r.object_colour.z, kk};
            if (kLdsFrames > 0 && depth < kLdsFrames) {
#pragma unroll
                for (int w = 0; w < 7; w++) lds_stk[depth][w][stk_lane] = frame[w];
            } else {
#pragma unroll
                for (int w = 0; w < 7; w++) stk[depth][w] = frame[w];
            }
            depth++;
            pos = pos + ((rv * 3.0f) * SMALL_DISTANCE);  // engine.rs:237
            dir = rv;
            continue;
        }
        colour = r.final_colour;
        break;
    }
    while (depth > 0) {
        depth--;
        float frame[7];
        if (kLdsFrames > 0 && depth < kLdsFrames) {
#pragma unroll
            for (int w = 0; w < 7; w++) frame[w] = lds_stk[depth][w][stk_lane];
        } else {
#pragma unroll
            for (int w = 0; w < 7; w++) frame[w] = stk[depth][w];
        }
        VC f = vc(frame[0], frame[1], frame[2]);
        VC oc = vc(frame[3], frame[4], frame[5]);
        colour = f + emul(colour, oc) * frame[6];  // engine.rs:239-240
    }
    return colour;
}
RM_FAST_END

// ray.rs:44-56: walks a ray that starts inside an object to where it leaves it, by the object's own
// distance alone, 20 steps at most.
RM_DEV V3 internal_march(int type, const Geom &g, V3 pos, V3 dir) {
    for (int it = 0; it < 20; it++) {
        const float d = sdf_eval(type, g, pos);
        if (d > SMALL_DISTANCE) break;
        pos = pos + dir * fabsf(d);
    }
    return pos;
}

// cast_sight_ray with the refraction block of engine.rs:196-228 switched on (the reference carries
// it commented out).  A reflective hit then has two children - the ray refracted through the object
// (index ratio 2.5, shaded only if it hits something) and the mirror ray - so the recursion is a tree.
// It is walked depth first with one frame per open hit: the refracted subtree is finished and added
// first, then the mirror subtree, which is the order the recursion adds them in.  `first_hit` >= 0
// starts at shade_object for that object at `pos` instead of marching (engine.rs:147).
constexpr float REFRACTION_IOR = 2.5f;
template <class SC>
RM_DEV V3 cast_sight_tree(const SC &sc, SceneRef sv, V3 pos, V3 dir, int first_hit = -1,
                          float first = __builtin_nanf(""), float first_sph = __builtin_nanf("")) {
    struct Frame {
        V3 acc, oc;       // the hit's colour so far (engine.rs:194, then :225 added); its object colour
        V3 rpos, rdir;    // the mirror ray, cast once the refracted subtree is done (engine.rs:235-237)
        float kk;         // engine.rs:240
        int refracted;    // 0: the refracted subtree is being walked; 1: the mirror subtree
    };
    Frame stk[MAX_DEPTH];
    int depth = 0;
    int ignore = -1;          // engine.rs:222: the refracted ray leaves out the object it comes from
    bool sky_on_miss = true;  // engine.rs:138 for sight rays; a refracted ray that hits nothing adds nothing (:224)
    for (;;) {
        int idx = first_hit;
        if (idx < 0) {
            int steps;
            bool capped;
            idx = march(sc, pos, dir, ignore, sv.march_cap, steps, capped, first, first_sph);
            if (capped) atomicAdd(&sv.stats[1], 1ULL);
        }
        first_hit = -1;
        first = __builtin_nanf("");  // known for the ray from the camera only
        bool have = true;
        V3 val = v3(sv.sky[0], sv.sky[1], sv.sky[2]);
        if (idx < 0) {
            have = sky_on_miss;
        } else {
            const ShadeResult r = shade_local(sc, sv, idx, pos, dir);
            val = r.final_colour;
            if (r.reflectivity > 1e-3f) {
                if (depth >= MAX_DEPTH) {
                    atomicAdd(&sv.stats[0], 1ULL);
                } else {
                    const Geom g = sv.geom[idx];
                    const int type = sv.shade[idx].type;
                    const V3 in = refract(dir, r.n, 1.0f / REFRACTION_IOR);                          // engine.rs:200
                    const V3 exit_pos = internal_march(type, g, pos + in * (3.0f * SMALL_DISTANCE), in);  // :201-206
                    const V3 exit_n = calculate_normal(type, g, exit_pos, sv.tetra_normals != 0);   // :208
                    V3 out = refract(in, -exit_n, REFRACTION_IOR);                                   // :209
                    if (mag_sqd(out) == 0.0f) out = reflect(in, -exit_n);                            // :213-215
                    const float fresnel = powi5(rs_clamp(1.0f - dot(r.n, -dir), 0.0f, 1.0f));        // :233
                    const V3 rv = reflect(dir, r.n);
                    Frame &f = stk[depth++];
                    f.acc = r.final_colour;
                    f.oc = r.object_colour;
                    f.rpos = pos + ((rv * 3.0f) * SMALL_DISTANCE);  // :237
                    f.rdir = rv;
                    f.kk = rs_clamp(fresnel + r.reflectivity, 0.0f, 1.0f);
                    f.refracted = 0;
                    pos = exit_pos;
                    dir = out;
                    ignore = idx;
                    sky_on_miss = false;
                    continue;
                }
            }
        }
        // a subtree is finished: hand its colour to the open hits above it
        bool done = true;
        while (depth > 0) {
            Frame &f = stk[depth - 1];
            if (f.refracted == 0) {
                if (have) f.acc = f.acc + val;  // engine.rs:225
                f.refracted = 1;
                pos = f.rpos;
                dir = f.rdir;
                ignore = -1;
                sky_on_miss = true;
                done = false;
                break;
            }
            val = f.acc + emul(val, f.oc) * f.kk;  // engine.rs:239-240
            have = true;
            depth--;
        }
        if (done) return val;
    }
}

// engine.rs:24-32
RM_DEV V3 rot(V3 a, float sx, float cx, float sy, float cy) {
    float x = a.x, y = a.y, z = a.z;
    float ny = y * cy - z * sy, nz = y * sy + z * cy;
    y = ny;
    z = nz;
    float nx = x * cx - z * sx;
    nz = x * sx + z * cx;
    return v3(nx, y, nz);
}

// engine.rs:79-84
// The screen coordinates u = 2 (x - W/2) / H of a column and v = 2 ((H - y_inv) - H/2) / H of a row
// depend on the frame size alone: the host tabulates them once per engine in the reference's f32
// operations, which saves every pixel two divisions.
RM_DEV V3 pixel_direction(const Pose &p, int x, int y_inv, const float *table_u, const float *table_v) {
    const float u = table_u[x], v = table_v[y_inv];
    return normalized(rot(v3(u, v, p.zdepth), p.sx, p.cx, p.sy, p.cy));
}

// colour.rs:33-44
RM_DEV float aces(float x) { return (x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f); }

// engine.rs:21-22
RM_DEV unsigned to_pixel_range(float i) {
    float v = rs_clamp(roundf(255.0f * i), 0.0f, 255.0f);
    return (v == v) ? (unsigned)v : 0u;
}

// engine.rs:109-125 for one pixel: linear colour -> packed B,G,R,0 bytes
RM_DEV unsigned pack_pixel_exact(V3 c) {
    float r = sqrt_rn(aces(c.x)), g = sqrt_rn(aces(c.y)), b = sqrt_rn(aces(c.z));
    return to_pixel_range(b) | (to_pixel_range(g) << 8) | (to_pixel_range(r) << 16);
}

// One channel of the same through the hardware's one-ulp reciprocal and square root: y estimates
// 255 * sqrt(ACES(x)) to within 7e-5 (three roundings and two one-ulp functions on a value of at most
// 260), and so does the value the exact operations produce.  Where y is a finite non-negative number
// (`sane`) at least 2e-4 away from every k + 1/2 (`sure`), both round to the same byte.
RM_FAST_BEGIN
RM_DEV unsigned channel_estimate(float x, bool &sure, bool &sane) {
    const float num = x * (2.51f * x + 0.03f), den = x * (2.43f * x + 0.59f) + 0.14f;
    const float y = 255.0f * __builtin_amdgcn_sqrtf(num * __builtin_amdgcn_rcpf(den));
    const float fl = floorf(y), fr = y - fl;
    sane = sane && y >= 0.0f && y < 1e6f;
    sure = sure && fabsf(fr - 0.5f) > 2e-4f;
    const float rounded = fl + (fr > 0.5f ? 1.0f : 0.0f);
    return (unsigned)fminf(rounded, 255.0f);
}

RM_DEV unsigned pack_pixel(VC linear, float exposure) {
    VC c = linear * exposure;
    if (RM_ABL & 16) return to_pixel_range(c.x) | (to_pixel_range(c.y) << 8) | (to_pixel_range(c.z) << 16);
    bool sure = true, sane = true;
    const unsigned r = channel_estimate(c.x, sure, sane), g = channel_estimate(c.y, sure, sane), b = channel_estimate(c.z, sure, sane);
    // (fast tier: the estimate alone - a value within 2e-4 of a rounding boundary may land on the other
    // byte - unless something in the wave is negative or not a number)
    if (wave_all(RM_FAST ? sane : (sure && sane))) return b | (g << 8) | (r << 16);
    return pack_pixel_exact(c);  // a lane close to a rounding boundary, or not a number: the whole wave takes the exact operations
}
RM_FAST_END

}  // namespace rm
