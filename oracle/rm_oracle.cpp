// CPU oracle for the ray-marching hot path.  TEST INFRASTRUCTURE ONLY.
//
// This file is a from-scratch C++ restatement of the reference renderer's
// per-pixel and per-patch arithmetic (joshua-jose/RayMarching, Rust).  It is
// the checker that the HIP kernels are compared with; nothing in the shipped
// library links, loads or calls it.  Only tests/, __graft_entry__.smoke() and
// the cpu_baseline leg of bench.py may use it.
//
// PARITY UNPINNED: the reference holds no tests, golden images or known-answer
// vectors, and it cannot be compiled in this environment (nightly Rust, no
// toolchain, no network for its crates).  The oracle is therefore pinned only
// by hand-derived known answers (tests/test_oracle_kat.py) and by line-by-line
// review against the cited source.
//
// Build: g++ -O2 -ffp-contract=off -fno-fast-math -fopenmp (see oracle/Makefile).
// All arithmetic is IEEE binary32 with no contraction, as rustc emits it.
//
// Citations are file:line in the reference tree (src/...).

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

// ---------------------------------------------------------------------------
// Rust float semantics used by the reference
// ---------------------------------------------------------------------------

// f32::max / f32::min: if one operand is NaN the other is returned.
static inline float rs_max(float a, float b) { return std::fmax(a, b); }
static inline float rs_min(float a, float b) { return std::fmin(a, b); }

// f32::clamp: comparisons only, so a NaN input stays NaN.
static inline float rs_clamp(float x, float lo, float hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

// `as u8` / `as u32` / `as usize`: truncating, saturating, NaN -> 0.
static inline uint8_t as_u8(float f) {
    if (!(f == f)) return 0;
    if (f <= 0.0f) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}
static inline uint32_t as_u32(float f) {
    if (!(f == f)) return 0;
    if (f <= 0.0f) return 0;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
static inline uint64_t as_usize(float f) {
    if (!(f == f)) return 0;
    if (f <= 0.0f) return 0;
    if (f >= 18446744073709551616.0f) return ~(uint64_t)0;
    return (uint64_t)f;
}

// powi with a constant exponent is expanded by LLVM by binary decomposition.
static inline float powi2(float x) { return x * x; }
static inline float powi3(float x) { return x * (x * x); }
static inline float powi5(float x) {
    float sq = x * x;
    float q = sq * sq;
    return x * q;
}

// ---------------------------------------------------------------------------
// Event counters (compiled in with -DORC_COUNT; used for the flop numerator)
// ---------------------------------------------------------------------------

struct Counters {
    uint64_t march_calls = 0;       // Ray::march / radiosity_march invocations
    uint64_t march_steps = 0;       // outer-loop iterations of march
    uint64_t march_sphere_sdf = 0;  // sphere sdf evaluations inside march
    uint64_t march_plane_sdf = 0;   // axis-plane sdf evaluations inside march
    uint64_t march_gplane_sdf = 0;  // generic-plane sdf evaluations inside march
    uint64_t shadow_calls = 0;
    uint64_t shadow_steps = 0;
    uint64_t shadow_sphere_sdf = 0;
    uint64_t shadow_plane_sdf = 0;
    uint64_t shadow_gplane_sdf = 0;
    uint64_t normals_sphere = 0;
    uint64_t normals_plane = 0;
    uint64_t normals_gplane = 0;
    uint64_t shade_calls = 0;
    uint64_t lightmap_samples = 0;
    uint64_t lightmap_resamples = 0;
    uint64_t lightmap_oob = 0;  // indices the reference would have aborted on
    uint64_t texture_samples = 0;
    uint64_t powf_calls = 0;
    uint64_t atan2_calls = 0;
    uint64_t reflection_rays = 0;
    uint64_t sky_hits = 0;
    uint64_t max_depth = 0;        // deepest reflection recursion seen
    uint64_t max_march_steps = 0;  // longest single march seen
    uint64_t pixels = 0;
    uint64_t march_box_sdf = 0;   // EXTENSION: box distance evaluations (sums, like the other sdf counts)
    uint64_t shadow_box_sdf = 0;
    static constexpr int N = 27;
    void add(const Counters &o) {
        const uint64_t *s = reinterpret_cast<const uint64_t *>(&o);
        uint64_t *d = reinterpret_cast<uint64_t *>(this);
        for (int i = 0; i < N; i++) {
            if (i == 22 || i == 23)
                d[i] = std::max(d[i], s[i]);
            else
                d[i] += s[i];
        }
    }
};
static_assert(sizeof(Counters) == Counters::N * sizeof(uint64_t), "Counters::N out of date");

static thread_local Counters tl_counters;
#ifdef ORC_COUNT
#define CNT(field, n) (tl_counters.field += (n))
#define CNT_MAX(field, v) (tl_counters.field = std::max<uint64_t>(tl_counters.field, (v)))
#else
#define CNT(field, n) ((void)0)
#define CNT_MAX(field, v) ((void)0)
#endif
// The out-of-bounds tally is always kept: it is a correctness signal.
#define CNT_ALWAYS(field, n) (tl_counters.field += (n))

// ---------------------------------------------------------------------------
// vector.rs:6-101  Vec3 = f32x4 {x, y, z, 0}
// ---------------------------------------------------------------------------

struct Vec3 {
    float l[4];

    Vec3() : l{0.0f, 0.0f, 0.0f, 0.0f} {}
    Vec3(float x, float y, float z) : l{x, y, z, 0.0f} {}  // vector.rs:12

    float x() const { return l[0]; }
    float y() const { return l[1]; }
    float z() const { return l[2]; }

    // vector.rs:24  (self.0 * self.0).reduce_sum(): ordered lane sum from 0.
    float mag_sqd() const {
        float s = 0.0f;
        for (int i = 0; i < 4; i++) s = s + l[i] * l[i];
        return s;
    }
    // vector.rs:22
    float mag() const { return std::sqrt(mag_sqd()); }
    // vector.rs:27-30  reciprocal, then multiply
    Vec3 normalized() const {
        float inv_mag = 1.0f / mag();
        return *this * inv_mag;
    }
    // vector.rs:32
    float dot(const Vec3 &o) const { return l[0] * o.l[0] + l[1] * o.l[1] + l[2] * o.l[2]; }
    // vector.rs:34
    Vec3 sqrt() const {
        Vec3 r;
        for (int i = 0; i < 4; i++) r.l[i] = std::sqrt(l[i]);
        return r;
    }
    // vector.rs:36
    Vec3 element_mul(const Vec3 &o) const {
        Vec3 r;
        for (int i = 0; i < 4; i++) r.l[i] = l[i] * o.l[i];
        return r;
    }
    // vector.rs:38
    Vec3 reflect(const Vec3 &normal) const { return *this - normal * (2.0f * normal.dot(*this)); }
    // vector.rs:40-48 (dead in the live render path; kept for completeness)
    Vec3 refract(const Vec3 &normal, float ior) const {
        float k = 1.0f - powi2(ior) * (1.0f - powi2(dot(normal)));
        if (k < 0.0f) return Vec3(0.0f, 0.0f, 0.0f);
        return (*this * ior) - (normal * (ior * dot(normal) + std::sqrt(k)));
    }

    Vec3 operator+(const Vec3 &o) const {
        Vec3 r;
        for (int i = 0; i < 4; i++) r.l[i] = l[i] + o.l[i];
        return r;
    }
    Vec3 &operator+=(const Vec3 &o) {
        for (int i = 0; i < 4; i++) l[i] = l[i] + o.l[i];
        return *this;
    }
    Vec3 operator-(const Vec3 &o) const {
        Vec3 r;
        for (int i = 0; i < 4; i++) r.l[i] = l[i] - o.l[i];
        return r;
    }
    // vector.rs:73-77  splat add (touches the w lane too)
    Vec3 add_scalar(float s) const {
        Vec3 r;
        for (int i = 0; i < 4; i++) r.l[i] = l[i] + s;
        return r;
    }
    Vec3 operator*(float s) const {
        Vec3 r;
        for (int i = 0; i < 4; i++) r.l[i] = l[i] * s;
        return r;
    }
    Vec3 operator/(float s) const {
        Vec3 r;
        for (int i = 0; i < 4; i++) r.l[i] = l[i] / s;
        return r;
    }
    Vec3 operator-() const {
        Vec3 r;
        for (int i = 0; i < 4; i++) r.l[i] = -l[i];
        return r;
    }
};
typedef Vec3 Colour;

// colour.rs:7-15  rgb![r,g,b] = ((c/255)^2)
static inline Colour rgb(float r, float g, float b) {
    return Colour((r / 255.0f) * (r / 255.0f), (g / 255.0f) * (g / 255.0f), (b / 255.0f) * (b / 255.0f));
}

// engine.rs:9-15
static const float MAX_MARCH_DISTANCE = 50.0f;
static const float SMALL_DISTANCE = 0.001f;
static const uint32_t MAX_SHAD_IT = 64;
static inline Colour sky_colour() { return rgb(135.0f, 206.0f, 235.0f); }

// objects.rs:13-16
static const float STEP_SIZE = 0.0001f;

// ---------------------------------------------------------------------------
// material.rs:2-21
// ---------------------------------------------------------------------------

struct Material {
    float ambient, diffuse, specular, shininess, reflectivity, emissive;
};

// ---------------------------------------------------------------------------
// texture.rs:4-37
// ---------------------------------------------------------------------------

struct Texture {
    std::vector<Colour> image;
    uint32_t width = 0, height = 0;
    float uscale = 1.0f, vscale = 1.0f;

    // texture.rs:13-29, fed with the decoded top-down RGB8 rows.
    static Texture from_rgb8(const uint8_t *rgb_bytes, uint32_t w, uint32_t h, float us, float vs) {
        Texture t;
        t.width = w;
        t.height = h;
        t.uscale = us;
        t.vscale = vs;
        t.image.reserve((size_t)w * h);
        for (size_t i = 0; i < (size_t)w * h; i++)
            t.image.push_back(rgb((float)rgb_bytes[3 * i], (float)rgb_bytes[3 * i + 1], (float)rgb_bytes[3 * i + 2]));
        return t;
    }
    // texture.rs:31-37
    Colour sample(float u, float v) const {
        CNT(texture_samples, 1);
        uint32_t x = as_u32(std::fmod(std::floor(std::fabs(u * uscale)), (float)width));
        uint32_t y = as_u32(std::fmod(std::floor(std::fabs(v * vscale)), (float)height));
        uint32_t i = x + (y * width);
        if (i >= image.size()) {  // the reference would abort on the bounds check
            CNT_ALWAYS(lightmap_oob, 1);
            return Colour();
        }
        return image[i];
    }
};

// ---------------------------------------------------------------------------
// radiosity.rs:12-17  Lightmap = [[Colour; MAP_SIZE]; MAP_SIZE], index [u][v]
// MAP_SIZE is a run-time value here (8 in the reference).
// ---------------------------------------------------------------------------

struct Lightmap {
    int m = 8;
    std::vector<Colour> s;
    explicit Lightmap(int m_ = 8) : m(m_), s((size_t)m_ * m_) {}
    Colour &at(uint64_t u, uint64_t v) { return s[(size_t)(u * m + v)]; }
    const Colour &at(uint64_t u, uint64_t v) const { return s[(size_t)(u * m + v)]; }
};

enum ObjType { T_SPHERE = 0, T_XPLANE = 1, T_YPLANE = 2, T_ZPLANE = 3, T_PLANE = 4, T_BOX = 5 };

// colour.rs:47-71
static Colour bilinear_interpolation(float x, float y, std::array<std::tuple<float, float, Colour>, 4> &points) {
    // two stable sorts: by y, then by x
    std::stable_sort(points.begin(), points.end(),
                     [](const auto &a, const auto &b) { return std::get<1>(a) < std::get<1>(b); });
    std::stable_sort(points.begin(), points.end(),
                     [](const auto &a, const auto &b) { return std::get<0>(a) < std::get<0>(b); });
    float x1 = std::get<0>(points[0]), y1 = std::get<1>(points[0]);
    Colour q11 = std::get<2>(points[0]);
    float y2 = std::get<1>(points[1]);
    Colour q12 = std::get<2>(points[1]);
    float x2 = std::get<0>(points[2]);
    Colour q21 = std::get<2>(points[2]);
    Colour q22 = std::get<2>(points[3]);
    // colour.rs:57-60 asserts the rectangle; a violation aborts the reference.
    if (x1 != std::get<0>(points[1]) || x2 != std::get<0>(points[3]) || y1 != std::get<1>(points[2]) ||
        y2 != std::get<1>(points[3]))
        CNT_ALWAYS(lightmap_oob, 1);
    return (q11 * (x2 - x) * (y2 - y) + q21 * (x - x1) * (y2 - y) + q12 * (x2 - x) * (y - y1) +
            q22 * (x - x1) * (y - y1)) /
           ((x2 - x1) * (y2 - y1));
}

// ---------------------------------------------------------------------------
// objects.rs:26-151  trait EngineObject
// ---------------------------------------------------------------------------

struct EngineObject {
    Material material{};
    Colour colour_const;
    Lightmap lightmap;
    bool has_lightmap = true;
    bool tetra_normals = false;  // EXTENSION switch: the commented-out four-point normal of objects.rs:142-149
    int type = T_SPHERE;

    explicit EngineObject(int map_size) : lightmap(map_size) {}
    virtual ~EngineObject() {}

    virtual float sdf(const Vec3 &position) const = 0;
    virtual Colour colour(const Vec3 &) const { return colour_const; }
    virtual bool radiosity_collide() const { return false; }
    // objects.rs:39,42 default to unimplemented!(): signalled through `ok`.
    virtual Vec3 get_sample_pos(uint64_t, uint64_t, bool &ok) const {
        ok = false;
        return Vec3();
    }
    virtual void sample_uv_from_pos(const Vec3 &, float &u, float &v, bool &ok) const {
        ok = false;
        u = v = 0.0f;
    }

    void clear_lightmap() { lightmap = Lightmap(lightmap.m); }

    // Fetch with the reference's bounds check made non-fatal and counted.
    const Colour &luxel(uint64_t u, uint64_t v) const {
        if (u >= (uint64_t)lightmap.m || v >= (uint64_t)lightmap.m) {
            CNT_ALWAYS(lightmap_oob, 1);
            // Deterministic stand-in for the abort: clamp as signed indices.
            int64_t su = (int64_t)u, sv = (int64_t)v;
            su = std::min<int64_t>(std::max<int64_t>(su, 0), lightmap.m - 1);
            sv = std::min<int64_t>(std::max<int64_t>(sv, 0), lightmap.m - 1);
            return lightmap.at((uint64_t)su, (uint64_t)sv);
        }
        return lightmap.at(u, v);
    }

    // objects.rs:45-133
    Colour sample_lightmap(const Vec3 &pos) const {
        CNT(lightmap_samples, 1);
        const uint64_t MAP_SIZE = (uint64_t)lightmap.m;
        float u, v;
        bool ok = true;
        sample_uv_from_pos(pos, u, v, ok);

        uint64_t u0 = as_usize(rs_max(std::floor(u), 0.0f));
        uint64_t v0 = as_usize(rs_max(std::floor(v), 0.0f));
        uint64_t u1 = u0 + 1, v1 = v0 + 1;

        u0 = (u0 >= MAP_SIZE) ? u0 - 1 : u0;
        v0 = (v0 >= MAP_SIZE) ? v0 - 1 : v0;
        u1 = (u1 >= MAP_SIZE) ? u0 : u1;
        v1 = (v1 >= MAP_SIZE) ? v0 : v1;

        Colour sample00 = luxel(u0, v0);
        Colour sample01 = luxel(u0, v1);
        Colour sample10 = luxel(u1, v0);
        Colour sample11 = luxel(u1, v1);

        bool resample = false;
        if (sample00.mag_sqd() == 0.0f) {
            resample = true;
            if (sample10.mag_sqd() == 0.0f) {
                v0 += 1;
                v1 += 1;
            } else if (sample11.mag_sqd() == 0.0f) {
                v0 -= 1;  // wrapping, as release-mode usize arithmetic
                v1 -= 1;
            }
            if (sample01.mag_sqd() == 0.0f) {
                u0 += 1;
                u1 += 1;
            } else if (sample11.mag_sqd() == 0.0f) {
                u0 -= 1;
                u1 -= 1;
            }
        } else {
            if (sample10.mag_sqd() == 0.0f) {
                resample = true;
                u1 -= 1;
            }
            if (sample01.mag_sqd() == 0.0f) {
                resample = true;
                v1 -= 1;
            }
        }
        if (resample) {
            CNT(lightmap_resamples, 1);
            sample00 = luxel(u0, v0);
            sample01 = luxel(u0, v1);
            sample10 = luxel(u1, v0);
            sample11 = luxel(u1, v1);
        }
        if (u0 == u1) u1 += 1;
        if (v0 == v1) v1 += 1;

        std::array<std::tuple<float, float, Colour>, 4> points = {
            std::make_tuple((float)u0, (float)v0, sample00),
            std::make_tuple((float)u0, (float)v1, sample01),
            std::make_tuple((float)u1, (float)v0, sample10),
            std::make_tuple((float)u1, (float)v1, sample11),
        };
        return bilinear_interpolation(u, v, points);
    }

    // objects.rs:135-141
    Vec3 calculate_normal(const Vec3 &position) const {
        const Vec3 X_STEP(STEP_SIZE, 0.0f, 0.0f), Y_STEP(0.0f, STEP_SIZE, 0.0f), Z_STEP(0.0f, 0.0f, STEP_SIZE);
        if (type == T_SPHERE)
            CNT(normals_sphere, 1);
        else if (type == T_PLANE)
            CNT(normals_gplane, 1);
        else
            CNT(normals_plane, 1);
        if (tetra_normals) {
            // objects.rs:18-24 and 142-149 (commented out in the reference)
            const float TETRA_STEP = 0.0015f;
            const Vec3 STEP_A(TETRA_STEP, -TETRA_STEP, -TETRA_STEP), STEP_B(-TETRA_STEP, -TETRA_STEP, TETRA_STEP);
            const Vec3 STEP_C(-TETRA_STEP, TETRA_STEP, -TETRA_STEP), STEP_D(TETRA_STEP, TETRA_STEP, TETRA_STEP);
            Vec3 norm = STEP_A * sdf(position + STEP_A) + STEP_B * sdf(position + STEP_B) + STEP_C * sdf(position + STEP_C) +
                        STEP_D * sdf(position + STEP_D);
            return norm.normalized();
        }
        float gx = sdf(position + X_STEP) - sdf(position - X_STEP);
        float gy = sdf(position + Y_STEP) - sdf(position - Y_STEP);
        float gz = sdf(position + Z_STEP) - sdf(position - Z_STEP);
        return Vec3(gx, gy, gz).normalized();
    }
};

// objects.rs:158-165, 333-359
struct Sphere : EngineObject {
    Vec3 position;
    float radius = 1.0f;
    explicit Sphere(int m) : EngineObject(m) { type = T_SPHERE; }
    float sdf(const Vec3 &p) const override { return (p - position).mag() - radius; }  // :334
    bool radiosity_collide() const override { return true; }                          // :343
    Vec3 get_sample_pos(uint64_t u, uint64_t v, bool &) const override {              // :345-352
        const float PI = 3.14159265358979323846f;
        const float MS = (float)lightmap.m;
        float theta = ((((float)u / MS) * 2.0f) - 1.0f) * PI;
        float phi = (((float)v / MS) - 0.5f) * PI;
        Vec3 n(std::sin(theta), std::sin(phi), -std::cos(theta));
        return position + n * radius;
    }
    void sample_uv_from_pos(const Vec3 &pos, float &u, float &v, bool &) const override {  // :353-358
        const float TAU = 6.28318530717958647692f;
        const float MS = (float)lightmap.m;
        Vec3 n = (pos - position).normalized();
        CNT(atan2_calls, 1);
        float uu = 0.5f + std::atan2(n.x(), -n.z()) / TAU;
        float vv = 0.5f + (n.y() / 2.0f);
        u = uu * MS;
        v = vv * MS;
    }
};

// objects.rs:167-175, 259-264.  plane_funcs!() gives Plane a lightmap (objects.rs:216), but
// get_sample_pos / sample_uv_from_pos are NOT implemented for it in the reference: lighting or
// shading one aborts (unimplemented!(), objects.rs:39,42).
//
// EXTENSION, defined here and in the HIP library alike: the lightmap of a Plane is laid out over two
// tangent directions through the plane's point nearest the origin, chosen so that a Plane with an
// axis normal gets the parameterisation of the matching XPlane / YPlane / ZPlane:
//   a  = the axis of the largest |normal component| (lowest on ties); e1 < e2 the other two axes
//   t  = normalized(unit(e1) - normal * normal[e1])
//   b  = (normal x t), negated if its e2 component is negative
//   origin = normal * distance
//   get_sample_pos(u, v)   = (origin + t * ((u - M/2) + 0.5)) + b * ((v - M/2) + 0.5)
//   sample_uv_from_pos(p)  = ((p - origin).t + M/2 - 0.5, (p - origin).b + M/2 - 0.5)
// all in f32, in the order written.
struct Plane : EngineObject {
    Vec3 normal;
    float distance = 0.0f;
    Vec3 tangent, bitangent, origin;
    explicit Plane(int m) : EngineObject(m) { type = T_PLANE; }
    float sdf(const Vec3 &p) const override { return p.dot(normal) - distance; }  // :261
    bool radiosity_collide() const override { return true; }
    void set_up_lightmap_axes() {
        const float n[3] = {normal.x(), normal.y(), normal.z()};
        int a = 0;
        if (std::fabs(n[1]) > std::fabs(n[a])) a = 1;
        if (std::fabs(n[2]) > std::fabs(n[a])) a = 2;
        const int e1 = a == 0 ? 1 : 0, e2 = a == 2 ? 1 : 2;
        const float d1 = n[e1];
        Vec3 raw((e1 == 0 ? 1.0f : 0.0f) - n[0] * d1, (e1 == 1 ? 1.0f : 0.0f) - n[1] * d1, (e1 == 2 ? 1.0f : 0.0f) - n[2] * d1);
        tangent = raw.normalized();
        const float t[3] = {tangent.x(), tangent.y(), tangent.z()};
        float c[3] = {n[1] * t[2] - n[2] * t[1], n[2] * t[0] - n[0] * t[2], n[0] * t[1] - n[1] * t[0]};
        const float sgn = c[e2] < 0.0f ? -1.0f : 1.0f;
        bitangent = Vec3(c[0] * sgn, c[1] * sgn, c[2] * sgn);
        origin = normal * distance;
    }
    Vec3 get_sample_pos(uint64_t u, uint64_t v, bool &) const override {
        const float MS = (float)lightmap.m;
        const float a = ((float)u) - (MS / 2.0f) + 0.5f, b = ((float)v) - (MS / 2.0f) + 0.5f;
        return (origin + tangent * a) + bitangent * b;
    }
    void sample_uv_from_pos(const Vec3 &pos, float &u, float &v, bool &) const override {
        const float MS = (float)lightmap.m;
        const Vec3 rel = pos - origin;
        u = rel.dot(tangent) + (MS / 2.0f) - 0.5f;
        v = rel.dot(bitangent) + (MS / 2.0f) - 0.5f;
    }
};

// EXTENSION (not live code in the reference): the box distance of the commented-out block
// objects.rs:370-380, with the centre and the half extents as parameters (the text has the
// half extents fixed at 0.7).  Evaluation order as written there.  It has no lightmap
// parameterisation, so it is shaded with the no-lightmap ambient of engine.rs:158.
struct Box : EngineObject {
    Vec3 position, half;
    explicit Box(int m) : EngineObject(m) {
        type = T_BOX;
        has_lightmap = false;
    }
    float sdf(const Vec3 &pos) const override {
        Vec3 p = pos - position;
        Vec3 abs_p(std::fabs(p.x()), std::fabs(p.y()), std::fabs(p.z()));
        Vec3 q = abs_p - half;
        Vec3 r(rs_max(q.x(), 0.0f), rs_max(q.y(), 0.0f), rs_max(q.z(), 0.0f));
        return r.mag() + rs_min(rs_max(q.x(), rs_max(q.y(), q.z())), 0.0f);
    }
    bool radiosity_collide() const override { return true; }
};

// EXTENSION: polynomial smooth minimum used for "smooth unions" (the reference's union is the
// hard minimum of ray.rs:23-33).  k = 0 is the hard minimum.
static inline float smooth_min(float a, float b, float k) {
    float h = rs_max(k - std::fabs(a - b), 0.0f) / k;
    return rs_min(a, b) - h * h * k * (1.0f / 4.0f);
}

// objects.rs:177-184, 293-311
struct XPlane : EngineObject {
    float x = 0.0f, dir = 1.0f;
    explicit XPlane(int m) : EngineObject(m) { type = T_XPLANE; }
    float sdf(const Vec3 &p) const override { return dir * (p.x() - x); }  // :295
    bool radiosity_collide() const override { return true; }
    Vec3 get_sample_pos(uint64_t u, uint64_t v, bool &) const override {  // :298-304
        const float MS = (float)lightmap.m;
        return Vec3(x, ((float)u) - (MS / 2.0f) + 0.5f, ((float)v) - (MS / 2.0f) + 0.5f);
    }
    void sample_uv_from_pos(const Vec3 &pos, float &u, float &v, bool &) const override {  // :305-310
        const float MS = (float)lightmap.m;
        u = pos.y() + (MS / 2.0f) - 0.5f;
        v = pos.z() + (MS / 2.0f) - 0.5f;
    }
};

// objects.rs:186-194, 266-291
struct YPlane : EngineObject {
    float y = 0.0f, dir = 1.0f;
    std::shared_ptr<Texture> texture;
    explicit YPlane(int m) : EngineObject(m) { type = T_YPLANE; }
    Colour colour(const Vec3 &position) const override {  // :267-274
        if (y > 0.0f) return colour_const;
        float u, v;
        bool ok = true;
        sample_uv_from_pos(position, u, v, ok);
        return texture->sample(u, v);
    }
    float sdf(const Vec3 &p) const override { return dir * (p.y() - y); }  // :275
    bool radiosity_collide() const override { return true; }
    Vec3 get_sample_pos(uint64_t u, uint64_t v, bool &) const override {  // :278-284
        const float MS = (float)lightmap.m;
        return Vec3(((float)u) - (MS / 2.0f) + 0.5f, y, ((float)v) - (MS / 2.0f) + 0.5f);
    }
    void sample_uv_from_pos(const Vec3 &pos, float &u, float &v, bool &) const override {  // :285-290
        const float MS = (float)lightmap.m;
        u = pos.x() + (MS / 2.0f) - 0.5f;
        v = pos.z() + (MS / 2.0f) - 0.5f;
    }
};

// objects.rs:196-203, 313-331
struct ZPlane : EngineObject {
    float z = 0.0f, dir = 1.0f;
    explicit ZPlane(int m) : EngineObject(m) { type = T_ZPLANE; }
    float sdf(const Vec3 &p) const override { return dir * (p.z() - z); }  // :315
    bool radiosity_collide() const override { return true; }
    Vec3 get_sample_pos(uint64_t u, uint64_t v, bool &) const override {  // :318-324
        const float MS = (float)lightmap.m;
        return Vec3(((float)u) - (MS / 2.0f) + 0.5f, ((float)v) - (MS / 2.0f) + 0.5f, z);
    }
    void sample_uv_from_pos(const Vec3 &pos, float &u, float &v, bool &) const override {  // :325-330
        const float MS = (float)lightmap.m;
        u = pos.x() + (MS / 2.0f) - 0.5f;
        v = pos.y() + (MS / 2.0f) - 0.5f;
    }
};

typedef std::vector<std::unique_ptr<EngineObject>> Objects;
static const uint64_t NONE = ~(uint64_t)0;

static inline void count_sdf(int type, bool shadow) {
#ifdef ORC_COUNT
    if (shadow) {
        if (type == T_SPHERE) tl_counters.shadow_sphere_sdf++;
        else if (type == T_PLANE) tl_counters.shadow_gplane_sdf++;
        else if (type == T_BOX) tl_counters.shadow_box_sdf++;
        else tl_counters.shadow_plane_sdf++;
    } else {
        if (type == T_SPHERE) tl_counters.march_sphere_sdf++;
        else if (type == T_PLANE) tl_counters.march_gplane_sdf++;
        else if (type == T_BOX) tl_counters.march_box_sdf++;
        else tl_counters.march_plane_sdf++;
    }
#else
    (void)type;
    (void)shadow;
#endif
}

// ---------------------------------------------------------------------------
// ray.rs:9-118
// ---------------------------------------------------------------------------

struct Ray {
    Vec3 position, direction;
    uint64_t last_steps = 0;
    // EXTENSION: > 0 blends the object distances with smooth_min.  The smooth minimum does not
    // associate, so the fold order is part of the definition: objects grouped by type (X, Y, Z
    // planes, generic planes, spheres, boxes), by index inside a group - `order` holds that
    // permutation.  With smooth_k = 0 the loops below are the reference's, in object order.
    float smooth_k = 0.0f;
    const std::vector<uint64_t> *order = nullptr;

    // ray.rs:15-42 and ray.rs:58-85 (radiosity=true also skips non-colliders)
    uint64_t march(const Objects &objects, uint64_t ignore_index, bool radiosity = false) {
        CNT(march_calls, 1);
        float distance_travelled = 0.0f;
        uint64_t steps = 0;
        while (distance_travelled < MAX_MARCH_DISTANCE) {
            steps++;
            CNT(march_steps, 1);
            float distance = std::numeric_limits<float>::infinity();
            float raw_best = std::numeric_limits<float>::infinity();
            uint64_t closest_object = NONE;
            for (uint64_t n = 0; n < objects.size(); n++) {
                const uint64_t i = (smooth_k > 0.0f && order) ? (*order)[n] : n;
                if (ignore_index == i) continue;
                if (radiosity && !objects[i]->radiosity_collide()) continue;
                count_sdf(objects[i]->type, false);
                float obj_distance = objects[i]->sdf(position);
                if (smooth_k > 0.0f) {
                    // EXTENSION: the step length is the blended distance; the hit object is still
                    // the one with the smallest own distance (lowest index on ties)
                    if (obj_distance < raw_best || (obj_distance == raw_best && i < closest_object)) {
                        raw_best = obj_distance;
                        closest_object = i;
                    }
                    distance = smooth_min(distance, obj_distance, smooth_k);
                } else if (obj_distance < distance) {
                    distance = obj_distance;
                    closest_object = i;
                }
            }
            if (distance < SMALL_DISTANCE) {
                last_steps = steps;
                CNT_MAX(max_march_steps, steps);
                return closest_object;
            }
            distance_travelled += distance;
            position += direction * distance;
        }
        last_steps = steps;
        CNT_MAX(max_march_steps, steps);
        return NONE;
    }

    // ray.rs:44-56 (dead in the live render path)
    void internal_march(const EngineObject &object) {
        uint16_t iterations = 0;
        while (iterations < 20) {
            float distance = object.sdf(position);
            if (distance > SMALL_DISTANCE) break;
            position += direction * std::fabs(distance);
            iterations += 1;
        }
    }

    // ray.rs:87-118
    float smooth_shadow_march(const Objects &objects, uint64_t ignore_obj_index, float light_dist, float shading_k) {
        CNT(shadow_calls, 1);
        float distance_travelled = 0.0f;
        float shade = 1.0f;
        auto smoothstep = [](float x) { return 3.0f * powi2(x) - 2.0f * powi3(x); };
        for (uint32_t it = 0; it < MAX_SHAD_IT; it++) {
            CNT(shadow_steps, 1);
            float distance = std::numeric_limits<float>::infinity();
            for (uint64_t n = 0; n < objects.size(); n++) {
                const uint64_t i = (smooth_k > 0.0f && order) ? (*order)[n] : n;
                if (ignore_obj_index == i) continue;
                count_sdf(objects[i]->type, true);
                float obj_distance = objects[i]->sdf(position);
                if (smooth_k > 0.0f)
                    distance = smooth_min(distance, obj_distance, smooth_k);
                else if (obj_distance < distance)
                    distance = obj_distance;
            }
            shade = rs_min(shade, smoothstep(rs_clamp(shading_k * distance / distance_travelled, 0.0f, 1.0f)));
            distance_travelled += distance;
            position += direction * distance;
            if (distance < SMALL_DISTANCE || distance_travelled > light_dist) break;
        }
        return rs_clamp(shade, 0.0f, 1.0f);
    }
};

// colour.rs:33-44
static Colour aces_film(Colour col) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    auto map_colours = [&](float x) { return (x * (a * x + b)) / (x * (c * x + d) + e); };
    col.l[0] = map_colours(col.x());
    col.l[1] = map_colours(col.y());
    col.l[2] = map_colours(col.z());
    return col;
}

// colour.rs:74-98
static void phong_ds(const Vec3 &n, const Vec3 &vector_to_light, float distance_to_light, float light_intensity,
                     const Material &mat, const Vec3 &view_direction, float &diffuse, float &specular) {
    Vec3 light_reflection_vector = vector_to_light.reflect(n);
    float li = light_intensity / powi2(distance_to_light);
    diffuse = mat.diffuse * li * rs_max(vector_to_light.dot(n), 0.0f);
    if (diffuse > 0.0f) {
        CNT(powf_calls, 1);
        specular = mat.specular * li * std::pow(rs_max(light_reflection_vector.dot(view_direction), 0.0f), mat.shininess);
    } else {
        specular = 0.0f;
    }
}

// engine.rs:21-22
static inline uint8_t to_pixel_range(float i) { return as_u8(rs_clamp(std::round(255.0f * i), 0.0f, 255.0f)); }

// engine.rs:24-32
static Vec3 rot(const Vec3 &vector, float sx, float cx, float sy, float cy) {
    float x = vector.x(), y = vector.y(), z = vector.z();
    {
        float ny = y * cy - z * sy, nz = y * sy + z * cy;
        y = ny;
        z = nz;
    }
    {
        float nx = x * cx - z * sx, nz = x * sx + z * cx;
        x = nx;
        z = nz;
    }
    return Vec3(x, y, z);
}

struct PointLight {
    Vec3 position;
    float intensity = 1.0f;
};

// ---------------------------------------------------------------------------
// engine.rs:37-333
// ---------------------------------------------------------------------------

struct Engine {
    Objects objects;
    Vec3 camera_position;
    PointLight light;
    int map_size = 8;
    float smooth_k = 0.0f;  // EXTENSION, see Ray::smooth_k
    int ssaa = 1;           // 4 = the four-ray supersampling of engine.rs:95-108 (commented out there)
    // EXTENSION switch: the refraction block of engine.rs:196-228 (commented out there).  With it a
    // reflective hit has two children, so the recursion is a tree; it is cut where it is REFRACTION_MAX_DEPTH
    // hits deep (the reference has no limit) and the cuts are counted in `refraction_depth_cuts`.
    bool refraction = false;
    static constexpr uint64_t REFRACTION_MAX_DEPTH = 16;
    mutable std::atomic<uint64_t> refraction_depth_cuts{0};
    mutable std::vector<uint64_t> fold_order;
    const std::vector<uint64_t> *order() const {
        if (fold_order.size() != objects.size()) {
            static const int rank[6] = {4, 0, 1, 2, 3, 5};  // by ObjType: planes X,Y,Z, generic, spheres, boxes
            fold_order.resize(objects.size());
            for (uint64_t i = 0; i < objects.size(); i++) fold_order[i] = i;
            std::stable_sort(fold_order.begin(), fold_order.end(),
                             [&](uint64_t a, uint64_t b) { return rank[objects[a]->type] < rank[objects[b]->type]; });
        }
        return &fold_order;
    }
    std::vector<std::shared_ptr<Texture>> textures;
    std::string error;
    Counters counters;  // merged totals

    // engine.rs:129-145
    Colour cast_sight_ray(const Vec3 &position, const Vec3 &direction, uint64_t depth) const {
        CNT_MAX(max_depth, depth);
        Ray ray{position, direction};
        ray.smooth_k = smooth_k;
        ray.order = order();
        uint64_t has_hit = ray.march(objects, NONE);
        if (has_hit == NONE) {
            CNT(sky_hits, 1);
            return sky_colour();
        }
        return shade_object(has_hit, ray.position, direction, depth);
    }

    // engine.rs:147-244
    Colour shade_object(uint64_t obj_index, const Vec3 &position, const Vec3 &direction, uint64_t depth) const {
        CNT(shade_calls, 1);
        const EngineObject &object = *objects[obj_index];
        Colour object_colour = object.colour(position);
        const Material &object_mat = object.material;
        Vec3 n = object.calculate_normal(position);

        Colour ambient;
        if (!object.has_lightmap)
            ambient = object_colour * object_mat.ambient;
        else
            ambient = object.sample_lightmap(position);

        if (ambient.mag_sqd() < powi2(object_mat.ambient)) ambient = ambient.normalized() * object_mat.ambient;
        ambient = ambient.element_mul(object_colour);

        Vec3 vector_to_light = light.position - position;
        float distance_to_light = vector_to_light.mag();
        vector_to_light = vector_to_light / distance_to_light;

        float diffuse, specular;
        phong_ds(n, vector_to_light, distance_to_light, light.intensity, object_mat, direction, diffuse, specular);

        Ray shadow{position, vector_to_light};
        shadow.smooth_k = smooth_k;
        shadow.order = order();
        float shade = shadow.smooth_shadow_march(objects, obj_index, distance_to_light, 16.0f);

        Colour final_colour = ambient + object_colour * (shade * (diffuse + specular));

        if (refraction && object_mat.reflectivity > 1e-3f) {
            if (depth >= REFRACTION_MAX_DEPTH) {
                refraction_depth_cuts++;
                return final_colour;
            }
            // engine.rs:199-226
            const float ior = 2.5f;
            Vec3 refraction_vector = direction.refract(n, 1.0f / ior);
            Ray enter_ray{position + refraction_vector * (3.0f * SMALL_DISTANCE), refraction_vector};
            enter_ray.internal_march(object);
            Vec3 exit_n = object.calculate_normal(enter_ray.position);
            Vec3 exit_direction = refraction_vector.refract(-exit_n, ior);
            if (exit_direction.mag_sqd() == 0.0f) exit_direction = refraction_vector.reflect(-exit_n);  // total internal reflection
            Ray exit_ray{enter_ray.position, exit_direction};
            exit_ray.smooth_k = smooth_k;
            exit_ray.order = order();
            uint64_t refr_hit = exit_ray.march(objects, obj_index);
            if (refr_hit != NONE) final_colour += shade_object(refr_hit, exit_ray.position, exit_ray.direction, depth + 1);
        }

        if (object_mat.reflectivity > 1e-3f) {
            float fresnel = powi5(rs_clamp(1.0f - n.dot(-direction), 0.0f, 1.0f));
            Vec3 reflection_vector = direction.reflect(n);
            CNT(reflection_rays, 1);
            Colour reflection_colour =
                cast_sight_ray(position + (reflection_vector * 3.0f * SMALL_DISTANCE), reflection_vector, depth + 1);
            final_colour += reflection_colour.element_mul(object_colour) *
                            rs_clamp(fresnel + object_mat.reflectivity, 0.0f, 1.0f);
        }
        return final_colour;
    }

    struct Pose {
        float sx, cx, sy, cy, zdepth, exposure;
    };

    // engine.rs:63-75 (stateful: the camera moves by rot(rel_move) each call)
    Pose apply_input(int32_t mouse_x, int32_t mouse_y, const Vec3 &rel_move, int32_t scroll) {
        Pose p;
        p.exposure = 1.0f;
        float theta_x = (float)(-mouse_x) / 600.0f;
        float theta_y = (float)mouse_y / 600.0f;
        p.sx = std::sin(theta_x);
        p.cx = std::cos(theta_x);
        p.sy = std::sin(theta_y);
        p.cy = std::cos(theta_y);
        camera_position += rot(rel_move, p.sx, p.cx, p.sy, p.cy);
        float fov_deg = 90.0f - ((float)scroll * 5.0f);
        const float PI = 3.14159265358979323846f;
        float rad = (fov_deg * 0.5f) * (PI / 180.0f);  // f32::to_radians
        p.zdepth = 1.0f / std::tan(rad);
        return p;
    }

    // engine.rs:77-86, one pixel of the direction pass
    static Vec3 pixel_direction(const Pose &p, int W, int H, int x, int y_inv) {
        int y = H - y_inv;
        float u = 2.0f * ((float)x - (0.5f * ((float)W))) / (float)H;
        float v = 2.0f * ((float)y - (0.5f * ((float)H))) / (float)H;
        return rot(Vec3(u, v, p.zdepth), p.sx, p.cx, p.sy, p.cy).normalized();
    }

    // engine.rs:38-127 over rows row0, row0+row_step, ... < row1; `buffer` holds one
    // W*4-byte row per rendered row (row_step > 1 is a sampling aid for timing).
    void render_rows(uint8_t *buffer, int W, int H, int row0, int row1, int row_step, const Pose &pose, int threads) {
        Counters total;
        (void)order();  // built here, before the threads start reading it
#ifdef _OPENMP
        if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
        {
            tl_counters = Counters();
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
            for (int y_inv = row0; y_inv < row1; y_inv += row_step) {
                for (int x = 0; x < W; x++) {
                    CNT(pixels, 1);
                    Vec3 dir = pixel_direction(pose, W, H, x, y_inv);
                    Colour colour_linear;
                    if (ssaa == 4) {
                        // engine.rs:95-108 (commented out there)
                        const Vec3 offsets[4] = {Vec3(0.5f / (float)W, 0.0f / (float)H, 0.0f), Vec3(-0.5f / (float)W, 0.0f / (float)H, 0.0f),
                                                 Vec3(0.0f / (float)W, 0.5f / (float)H, 0.0f), Vec3(0.0f / (float)W, -0.5f / (float)H, 0.0f)};
                        Colour c[4];
                        for (int i = 0; i < 4; i++) c[i] = cast_sight_ray(camera_position, (dir + offsets[i]).normalized(), 0);
                        colour_linear = (c[0] + c[1] + c[2] + c[3]) / 4.0f;
                    } else {
                        colour_linear = cast_sight_ray(camera_position, dir, 0);
                    }
                    Colour colour_srgb = aces_film(colour_linear * pose.exposure).sqrt();  // engine.rs:110
                    uint8_t *px = buffer + ((size_t)((y_inv - row0) / row_step) * W + x) * 4;
                    px[0] = to_pixel_range(colour_srgb.z());  // engine.rs:120-125
                    px[1] = to_pixel_range(colour_srgb.y());
                    px[2] = to_pixel_range(colour_srgb.x());
                    px[3] = 0;
                }
            }
#ifdef _OPENMP
#pragma omp critical
#endif
            total.add(tl_counters);
        }
        counters.add(total);
    }

    // Per-pixel trace of the primary ray only (test aid, not in the reference).
    void trace_primary(int W, int H, const Pose &pose, int32_t *hit, int32_t *steps, float *hit_pos) {
        for (int y_inv = 0; y_inv < H; y_inv++)
            for (int x = 0; x < W; x++) {
                Vec3 dir = pixel_direction(pose, W, H, x, y_inv);
                Ray ray{camera_position, dir};
                ray.smooth_k = smooth_k;
        ray.order = order();
                uint64_t h = ray.march(objects, NONE);
                size_t i = (size_t)y_inv * W + x;
                hit[i] = (h == NONE) ? -1 : (int32_t)h;
                steps[i] = (int32_t)ray.last_steps;
                if (hit_pos) {
                    hit_pos[3 * i] = ray.position.x();
                    hit_pos[3 * i + 1] = ray.position.y();
                    hit_pos[3 * i + 2] = ray.position.z();
                }
            }
    }

    // engine.rs:246-326.  stage_cb(stage, lightmaps, emissive): stage 0 is after
    // direct lighting, stage k>0 after bounce k.
    typedef void (*StageFn)(void *user, int stage, const std::vector<Lightmap> &lm, const std::vector<Lightmap> &em);

    bool compute_lightmaps(int bounces, StageFn stage_cb, void *user) {
        tl_counters = Counters();
        for (auto &o : objects) o->clear_lightmap();
        std::vector<uint64_t> obj_indexes;
        for (uint64_t i = 0; i < objects.size(); i++)
            if (objects[i]->has_lightmap) obj_indexes.push_back(i);
        const int M = map_size;
        const size_t n = obj_indexes.size();
        std::vector<Lightmap> point_cloud(n, Lightmap(M)), colour_cloud(n, Lightmap(M)), normal_cloud(n, Lightmap(M));

        // engine.rs:269-281
        for (size_t ci = 0; ci < n; ci++) {
            const EngineObject &obj = *objects[obj_indexes[ci]];
            for (int x = 0; x < M; x++)
                for (int y = 0; y < M; y++) {
                    bool ok = true;
                    Vec3 sample_pos = obj.get_sample_pos(x, y, ok);
                    if (!ok) {
                        error = "get_sample_pos is unimplemented for this object type (the reference aborts)";
                        return false;
                    }
                    colour_cloud[ci].at(x, y) = obj.colour(sample_pos);
                    normal_cloud[ci].at(x, y) = obj.calculate_normal(sample_pos);
                    point_cloud[ci].at(x, y) = sample_pos;
                }
        }

        // radiosity.rs:52-101
        std::vector<Lightmap> emissive_maps, new_lightmaps;
        for (size_t ci = 0; ci < n; ci++) {
            uint64_t obj_index = obj_indexes[ci];
            const EngineObject &object = *objects[obj_index];
            Lightmap obj_lightmap(M), emissive_map(M);
            float emissivity = object.material.emissive;
            for (int x = 0; x < M; x++)
                for (int y = 0; y < M; y++) {
                    Vec3 origin = point_cloud[ci].at(x, y);
                    Colour colour = colour_cloud[ci].at(x, y);
                    Vec3 vector_to_light = light.position - origin;
                    float distance_to_light = vector_to_light.mag();
                    vector_to_light = vector_to_light / distance_to_light;
                    Ray shadow_ray{origin, vector_to_light};
                    shadow_ray.smooth_k = smooth_k;
                    shadow_ray.order = order();
                    (void)shadow_ray.march(objects, obj_index);
                    if ((shadow_ray.position - origin).mag() < (distance_to_light + 3.0f * SMALL_DISTANCE)) continue;
                    Vec3 nrm = object.calculate_normal(origin);
                    float diffuse = rs_max(nrm.dot(vector_to_light), 0.0f) * light.intensity / powi2(distance_to_light);
                    emissive_map.at(x, y) = colour * (diffuse + emissivity);
                    obj_lightmap.at(x, y) = colour * emissivity;
                }
            emissive_maps.push_back(emissive_map);
            new_lightmaps.push_back(obj_lightmap);
        }
        for (size_t ci = 0; ci < n; ci++) objects[obj_indexes[ci]]->lightmap = new_lightmaps[ci];  // engine.rs:286-288
        if (stage_cb) stage_cb(user, 0, new_lightmaps, emissive_maps);

        // engine.rs:299-325
        std::vector<Lightmap> lightmaps, new_emissive_maps;
        for (int bounce = 0; bounce < bounces; bounce++) {
            lightmaps.clear();
            new_emissive_maps.clear();
            for (size_t ci = 0; ci < n; ci++) {
                uint64_t obj_eng_index = obj_indexes[ci];
                // radiosity.rs:159-196
                Lightmap obj_lightmap = objects[obj_eng_index]->lightmap;
                Lightmap new_emissive_map(M);
                for (int x = 0; x < M; x++)
                    for (int y = 0; y < M; y++) {
                        Colour incident = compute_patch_radiosity(point_cloud, normal_cloud, emissive_maps,
                                                                  point_cloud[ci].at(x, y), normal_cloud[ci].at(x, y),
                                                                  ci, obj_eng_index);
                        const float PI = 3.14159265358979323846f;
                        incident = incident / PI;
                        obj_lightmap.at(x, y) += incident;
                        new_emissive_map.at(x, y) = incident.element_mul(colour_cloud[ci].at(x, y));
                    }
                lightmaps.push_back(obj_lightmap);
                new_emissive_maps.push_back(new_emissive_map);
            }
            for (size_t ci = 0; ci < n; ci++) {
                objects[obj_indexes[ci]]->lightmap = lightmaps[ci];
                emissive_maps[ci] = new_emissive_maps[ci];
            }
            if (stage_cb) stage_cb(user, bounce + 1, lightmaps, emissive_maps);
        }
        counters.add(tl_counters);
        return true;
    }

    // radiosity.rs:103-157
    Colour compute_patch_radiosity(const std::vector<Lightmap> &point_cloud, const std::vector<Lightmap> &normal_cloud,
                                   const std::vector<Lightmap> &emissive_maps, const Vec3 &sample_pos,
                                   const Vec3 &sample_normal, size_t obj_cloud_index, uint64_t obj_eng_index) const {
        Colour incident(0.0f, 0.0f, 0.0f);
        const int M = map_size;
        for (size_t lci = 0; lci < point_cloud.size(); lci++) {
            if (obj_cloud_index == lci) continue;
            const Lightmap &lighting_lightmap = emissive_maps[lci];
            for (int a = 0; a < M; a++)
                for (int b = 0; b < M; b++) {
                    Vec3 light_source = point_cloud[lci].at(a, b);
                    Colour light_colour = lighting_lightmap.at(a, b);
                    Vec3 light_normal = normal_cloud[lci].at(a, b);

                    Vec3 vector_to_light = light_source - sample_pos;
                    float distance_to_light = vector_to_light.mag();
                    vector_to_light = vector_to_light / distance_to_light;

                    // `3.0 * SMALL_DISTANCE * -v` = (3.0*0.001) * (-v)
                    Ray shadow_ray{light_source + ((-vector_to_light) * (3.0f * SMALL_DISTANCE)), -vector_to_light};
                    shadow_ray.smooth_k = smooth_k;
                    shadow_ray.order = order();
                    uint64_t hit = shadow_ray.march(objects, NONE, true);
                    if (hit != NONE) {
                        if (hit != obj_eng_index) {
                            if ((shadow_ray.position - sample_pos).mag_sqd() > powi2(0.1f)) continue;
                        }
                    }
                    float attenuation = sample_normal.dot(vector_to_light) * -light_normal.dot(vector_to_light);
                    float diffuse = rs_max(attenuation, 0.0f) / powi2(distance_to_light);
                    incident += light_colour * rs_min(diffuse, 2.0f);
                }
        }
        return incident;
    }
};

// ---------------------------------------------------------------------------
// fxaa.txt:1-135.  The reference's FXAA is a commented-out text file written
// against an older Vec3{x,y,z}; nothing compiles or calls it.  Two readings:
//   in-place     - the text verbatim: later pixels read already-filtered ones;
//   out-of-place - same arithmetic, every tap read from the unfiltered source.
// ---------------------------------------------------------------------------

static void fxaa_pixel(const uint8_t *src, uint8_t *dst, int W, int x, int y_inv) {
    const float FXAA_SPAN_MAX = 8.0f, FXAA_REDUCE_MUL = 1.0f / 8.0f, FXAA_REDUCE_MIN = 1.0f / 128.0f;
    auto load = [&](size_t i, float *c) {
        c[0] = (float)src[i * 4 + 0] / 255.0f;
        c[1] = (float)src[i * 4 + 1] / 255.0f;
        c[2] = (float)src[i * 4 + 2] / 255.0f;
    };
    auto luma = [](const float *c) { return 0.299f * c[0] + 0.587f * c[1] + 0.114f * c[2]; };
    size_t i = (size_t)y_inv * W + x;
    size_t iNW = (size_t)(y_inv - 1) * W + (x - 1), iNE = (size_t)(y_inv - 1) * W + (x + 1);
    size_t iSW = (size_t)(y_inv + 1) * W + (x - 1), iSE = (size_t)(y_inv + 1) * W + (x + 1);
    float rgbNW[3], rgbNE[3], rgbSW[3], rgbSE[3], rgbM[3];
    load(iNW, rgbNW);
    load(iNE, rgbNE);
    load(iSW, rgbSW);
    load(iSE, rgbSE);
    load(i, rgbM);
    float lumaNW = luma(rgbNW), lumaNE = luma(rgbNE), lumaSW = luma(rgbSW), lumaSE = luma(rgbSE), lumaM = luma(rgbM);
    float lumaMin = rs_min(lumaM, rs_min(rs_min(lumaNW, lumaNE), rs_min(lumaSW, lumaSE)));
    float lumaMax = rs_max(lumaM, rs_max(rs_max(lumaNW, lumaNE), rs_max(lumaSW, lumaSE)));
    float dir[2];
    dir[0] = -((lumaNW + lumaNE) - (lumaSW + lumaSE));
    dir[1] = (lumaNW + lumaSW) - (lumaNE + lumaSE);
    float dirReduce = rs_max((lumaNW + lumaNE + lumaSW + lumaSE) * (0.25f * FXAA_REDUCE_MUL), FXAA_REDUCE_MIN);
    float rcpDirMin = 1.0f / (rs_min(std::fabs(dir[0]), std::fabs(dir[1])) + dirReduce);
    dir[0] = rs_min(FXAA_SPAN_MAX, rs_max(-FXAA_SPAN_MAX, dir[0] * rcpDirMin));
    dir[1] = rs_min(FXAA_SPAN_MAX, rs_max(-FXAA_SPAN_MAX, dir[1] * rcpDirMin));
    const float kA0 = 1.0f / 3.0f - 0.5f, kA1 = 2.0f / 2.0f - 0.5f, kB0 = 0.0f / 3.0f - 0.5f, kB1 = 3.0f / 3.0f - 0.5f;
    float A0x = dir[0] * kA0, A0y = dir[1] * kA0, A1x = dir[0] * kA1, A1y = dir[1] * kA1;
    float B0x = dir[0] * kB0, B0y = dir[1] * kB0, B1x = dir[0] * kB1, B1y = dir[1] * kB1;
    // negative offsets saturate to 0 in the `as usize` cast
    size_t iA0 = ((size_t)y_inv + as_usize(std::round(A0y))) * W + ((size_t)x + as_usize(std::round(A0x)));
    size_t iA1 = ((size_t)y_inv + as_usize(std::round(A1y))) * W + ((size_t)x + as_usize(std::round(A1x)));
    size_t iB0 = ((size_t)y_inv + as_usize(std::round(B0y))) * W + ((size_t)x + as_usize(std::round(B0x)));
    size_t iB1 = ((size_t)y_inv + as_usize(std::round(B1y))) * W + ((size_t)x + as_usize(std::round(B1x)));
    float a0[3], a1[3], b0[3], b1[3], rgbA[3], rgbB[3];
    load(iA0, a0);
    load(iA1, a1);
    load(iB0, b0);
    load(iB1, b1);
    for (int c = 0; c < 3; c++) {
        rgbA[c] = (1.0f / 2.0f) * (a0[c] + a1[c]);
        rgbB[c] = rgbA[c] * (1.0f / 2.0f) + (1.0f / 4.0f) * (b0[c] + b1[c]);
    }
    float lumaB = luma(rgbB);
    const float *cv = ((lumaB < lumaMin) || (lumaB > lumaMax)) ? rgbA : rgbB;
    dst[i * 4 + 0] = as_u8(255.0f * cv[0]);
    dst[i * 4 + 1] = as_u8(255.0f * cv[1]);
    dst[i * 4 + 2] = as_u8(255.0f * cv[2]);
}

static void fxaa(const uint8_t *src, uint8_t *dst, int W, int H, bool in_place) {
    if (!in_place) {
        std::memcpy(dst, src, (size_t)W * H * 4);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (int y_inv = 4; y_inv < H - 4; y_inv++)
            for (int x = 4; x < W - 4; x++) fxaa_pixel(src, dst, W, x, y_inv);
    } else {
        if (dst != src) std::memcpy(dst, src, (size_t)W * H * 4);
        for (int y_inv = 4; y_inv < H - 4; y_inv++)
            for (int x = 4; x < W - 4; x++) fxaa_pixel(dst, dst, W, x, y_inv);
    }
}

}  // namespace orc

// ===========================================================================
// C interface for ctypes (tests, smoke, cpu_baseline)
// ===========================================================================

using namespace orc;

extern "C" {

void *orc_engine_create(int map_size) {
    Engine *e = new Engine();
    e->map_size = map_size > 0 ? map_size : 8;
    return e;
}
void orc_engine_destroy(void *h) { delete static_cast<Engine *>(h); }
const char *orc_last_error(void *h) { return static_cast<Engine *>(h)->error.c_str(); }

int orc_texture_create_rgb8(void *h, const uint8_t *rgb, int w, int hgt, float uscale, float vscale) {
    Engine *e = static_cast<Engine *>(h);
    e->textures.push_back(std::make_shared<Texture>(Texture::from_rgb8(rgb, w, hgt, uscale, vscale)));
    return (int)e->textures.size() - 1;
}

static void set_common(EngineObject &o, const float *mat, const float *colour) {
    o.material = Material{mat[0], mat[1], mat[2], mat[3], mat[4], mat[5]};
    o.colour_const = Colour(colour[0], colour[1], colour[2]);
}

int orc_add_sphere(void *h, const float *pos, float radius, const float *mat, const float *colour) {
    Engine *e = static_cast<Engine *>(h);
    auto s = std::make_unique<Sphere>(e->map_size);
    s->position = Vec3(pos[0], pos[1], pos[2]);
    s->radius = radius;
    set_common(*s, mat, colour);
    e->objects.push_back(std::move(s));
    return (int)e->objects.size() - 1;
}
int orc_add_xplane(void *h, float x, float dir, const float *mat, const float *colour) {
    Engine *e = static_cast<Engine *>(h);
    auto s = std::make_unique<XPlane>(e->map_size);
    s->x = x;
    s->dir = dir;
    set_common(*s, mat, colour);
    e->objects.push_back(std::move(s));
    return (int)e->objects.size() - 1;
}
int orc_add_yplane(void *h, float y, float dir, const float *mat, const float *colour, int texture) {
    Engine *e = static_cast<Engine *>(h);
    if (texture < 0 || texture >= (int)e->textures.size()) {
        e->error = "YPlane needs a texture";
        return -1;
    }
    auto s = std::make_unique<YPlane>(e->map_size);
    s->y = y;
    s->dir = dir;
    s->texture = e->textures[texture];
    set_common(*s, mat, colour);
    e->objects.push_back(std::move(s));
    return (int)e->objects.size() - 1;
}
int orc_add_zplane(void *h, float z, float dir, const float *mat, const float *colour) {
    Engine *e = static_cast<Engine *>(h);
    auto s = std::make_unique<ZPlane>(e->map_size);
    s->z = z;
    s->dir = dir;
    set_common(*s, mat, colour);
    e->objects.push_back(std::move(s));
    return (int)e->objects.size() - 1;
}
int orc_add_plane(void *h, const float *normal, float distance, const float *mat, const float *colour) {
    Engine *e = static_cast<Engine *>(h);
    auto s = std::make_unique<Plane>(e->map_size);
    s->normal = Vec3(normal[0], normal[1], normal[2]);
    s->distance = distance;
    s->set_up_lightmap_axes();
    set_common(*s, mat, colour);
    e->objects.push_back(std::move(s));
    return (int)e->objects.size() - 1;
}
int orc_add_box(void *h, const float *pos, const float *half, const float *mat, const float *colour) {
    Engine *e = static_cast<Engine *>(h);
    auto s = std::make_unique<Box>(e->map_size);
    s->position = Vec3(pos[0], pos[1], pos[2]);
    s->half = Vec3(half[0], half[1], half[2]);
    set_common(*s, mat, colour);
    e->objects.push_back(std::move(s));
    return (int)e->objects.size() - 1;
}
void orc_set_smooth_union(void *h, float k) { static_cast<Engine *>(h)->smooth_k = k; }
void orc_set_ssaa(void *h, int samples) { static_cast<Engine *>(h)->ssaa = samples; }
void orc_set_refraction(void *h, int on) { static_cast<Engine *>(h)->refraction = on != 0; }
uint64_t orc_refraction_depth_cuts(void *h) { return static_cast<Engine *>(h)->refraction_depth_cuts.load(); }
// ray.rs:44-56 on a sphere: pos is moved to where the march leaves it
void orc_kat_internal_march(const float *centre, float radius, float *pos, const float *dir) {
    Sphere s(8);
    s.position = Vec3(centre[0], centre[1], centre[2]);
    s.radius = radius;
    Ray r{Vec3(pos[0], pos[1], pos[2]), Vec3(dir[0], dir[1], dir[2])};
    r.internal_march(s);
    for (int i = 0; i < 3; i++) pos[i] = r.position.l[i];
}
void orc_set_tetrahedral_normals(void *h, int on) {
    for (auto &o : static_cast<Engine *>(h)->objects) o->tetra_normals = on != 0;
}
float orc_kat_smooth_min(float a, float b, float k) { return smooth_min(a, b, k); }
void orc_set_light(void *h, const float *pos, float intensity) {
    Engine *e = static_cast<Engine *>(h);
    e->light.position = Vec3(pos[0], pos[1], pos[2]);
    e->light.intensity = intensity;
}
void orc_set_camera(void *h, const float *pos) {
    static_cast<Engine *>(h)->camera_position = Vec3(pos[0], pos[1], pos[2]);
}
void orc_get_camera(void *h, float *pos) {
    Engine *e = static_cast<Engine *>(h);
    for (int i = 0; i < 3; i++) pos[i] = e->camera_position.l[i];
}

struct StageDump {
    float *lightmaps;  // [stages][n_obj][M][M][3]
    float *emissive;
    size_t per_stage;
};
static void stage_dump(void *user, int stage, const std::vector<Lightmap> &lm, const std::vector<Lightmap> &em) {
    StageDump *d = static_cast<StageDump *>(user);
    size_t k = 0;
    for (size_t o = 0; o < lm.size(); o++)
        for (size_t i = 0; i < lm[o].s.size(); i++)
            for (int c = 0; c < 3; c++, k++) {
                if (d->lightmaps) d->lightmaps[stage * d->per_stage + k] = lm[o].s[i].l[c];
                if (d->emissive) d->emissive[stage * d->per_stage + k] = em[o].s[i].l[c];
            }
}

// lightmaps_out / emissive_out: optional [bounces+1][n_lightmapped][M][M][3]
int orc_compute_lightmaps(void *h, int bounces, float *lightmaps_out, float *emissive_out) {
    Engine *e = static_cast<Engine *>(h);
    size_t n = 0;
    for (auto &o : e->objects) n += o->has_lightmap ? 1 : 0;
    StageDump d{lightmaps_out, emissive_out, n * (size_t)e->map_size * e->map_size * 3};
    bool ok = e->compute_lightmaps(bounces, (lightmaps_out || emissive_out) ? stage_dump : nullptr, &d);
    return ok ? 0 : 1;
}

int orc_get_lightmap(void *h, int obj, float *out) {
    Engine *e = static_cast<Engine *>(h);
    if (obj < 0 || obj >= (int)e->objects.size()) return 1;
    const Lightmap &lm = e->objects[obj]->lightmap;
    for (size_t i = 0; i < lm.s.size(); i++)
        for (int c = 0; c < 3; c++) out[3 * i + c] = lm.s[i].l[c];
    return 0;
}
int orc_set_lightmap(void *h, int obj, const float *in) {
    Engine *e = static_cast<Engine *>(h);
    if (obj < 0 || obj >= (int)e->objects.size()) return 1;
    Lightmap &lm = e->objects[obj]->lightmap;
    for (size_t i = 0; i < lm.s.size(); i++) lm.s[i] = Colour(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
    return 0;
}

// Stateful render, as Engine::render: the camera moves by rot(rel_move).
int orc_render(void *h, uint8_t *bgra, int W, int H, int row0, int row1, int row_step, int mouse_x, int mouse_y,
               const float *rel_move, int scroll, int threads) {
    if (row_step < 1) row_step = 1;
    Engine *e = static_cast<Engine *>(h);
    Engine::Pose p = e->apply_input(mouse_x, mouse_y, Vec3(rel_move[0], rel_move[1], rel_move[2]), scroll);
    e->render_rows(bgra, W, H, row0, row1, row_step, p, threads);
    return 0;
}

// Host-side camera block only (engine.rs:63-75): out = {sx,cx,sy,cy,zdepth,cam[3]}
void orc_pose(void *h, int mouse_x, int mouse_y, const float *rel_move, int scroll, float *out) {
    Engine *e = static_cast<Engine *>(h);
    Engine::Pose p = e->apply_input(mouse_x, mouse_y, Vec3(rel_move[0], rel_move[1], rel_move[2]), scroll);
    out[0] = p.sx;
    out[1] = p.cx;
    out[2] = p.sy;
    out[3] = p.cy;
    out[4] = p.zdepth;
    for (int i = 0; i < 3; i++) out[5 + i] = e->camera_position.l[i];
}

int orc_trace_primary(void *h, int W, int H, int mouse_x, int mouse_y, int scroll, int32_t *hit, int32_t *steps,
                      float *hit_pos) {
    Engine *e = static_cast<Engine *>(h);
    Engine::Pose p = e->apply_input(mouse_x, mouse_y, Vec3(0, 0, 0), scroll);
    e->trace_primary(W, H, p, hit, steps, hit_pos);
    return 0;
}

void orc_fxaa(const uint8_t *src, uint8_t *dst, int W, int H, int in_place) { fxaa(src, dst, W, H, in_place != 0); }

int orc_counters_enabled() {
#ifdef ORC_COUNT
    return 1;
#else
    return 0;
#endif
}
int orc_counters_len() { return Counters::N; }
void orc_counters_get(void *h, uint64_t *out) {
    std::memcpy(out, &static_cast<Engine *>(h)->counters, sizeof(Counters));
}
void orc_counters_reset(void *h) { static_cast<Engine *>(h)->counters = Counters(); }

// ---- primitives exposed for known-answer tests ----------------------------

void orc_kat_rgb(float r, float g, float b, float *out) {
    Colour c = rgb(r, g, b);
    for (int i = 0; i < 3; i++) out[i] = c.l[i];
}
void orc_kat_aces(const float *in, float *out) {
    Colour c = aces_film(Colour(in[0], in[1], in[2]));
    for (int i = 0; i < 3; i++) out[i] = c.l[i];
}
int orc_kat_to_pixel_range(float v) { return to_pixel_range(v); }
// engine.rs:110 and :120-125 for n linear colours at exposure 1: B | G << 8 | R << 16
void orc_kat_pack_pixels(int n, const float *linear, uint32_t *out) {
    for (int i = 0; i < n; i++) {
        Colour s = aces_film(Colour(linear[3 * i], linear[3 * i + 1], linear[3 * i + 2]) * 1.0f).sqrt();
        out[i] = (uint32_t)to_pixel_range(s.z()) | ((uint32_t)to_pixel_range(s.y()) << 8) | ((uint32_t)to_pixel_range(s.x()) << 16);
    }
}
void orc_kat_texture_sample(void *h, int tex, float u, float v, float *out) {
    Colour c = static_cast<Engine *>(h)->textures[tex]->sample(u, v);
    for (int i = 0; i < 3; i++) out[i] = c.l[i];
}
float orc_kat_sdf(void *h, int obj, const float *p) {
    return static_cast<Engine *>(h)->objects[obj]->sdf(Vec3(p[0], p[1], p[2]));
}
void orc_kat_normal(void *h, int obj, const float *p, float *out) {
    Vec3 n = static_cast<Engine *>(h)->objects[obj]->calculate_normal(Vec3(p[0], p[1], p[2]));
    for (int i = 0; i < 3; i++) out[i] = n.l[i];
}
// returns hit index or -1; pos_io is updated to the final ray position
int orc_kat_march(void *h, float *pos_io, const float *dir, int ignore, int radiosity) {
    Ray r{Vec3(pos_io[0], pos_io[1], pos_io[2]), Vec3(dir[0], dir[1], dir[2])};
    r.smooth_k = static_cast<Engine *>(h)->smooth_k;
    r.order = static_cast<Engine *>(h)->order();
    uint64_t hit = r.march(static_cast<Engine *>(h)->objects, ignore < 0 ? NONE : (uint64_t)ignore, radiosity != 0);
    for (int i = 0; i < 3; i++) pos_io[i] = r.position.l[i];
    return hit == NONE ? -1 : (int)hit;
}
float orc_kat_shadow(void *h, const float *pos, const float *dir, int ignore, float light_dist, float k) {
    Ray r{Vec3(pos[0], pos[1], pos[2]), Vec3(dir[0], dir[1], dir[2])};
    r.smooth_k = static_cast<Engine *>(h)->smooth_k;
    r.order = static_cast<Engine *>(h)->order();
    return r.smooth_shadow_march(static_cast<Engine *>(h)->objects, (uint64_t)ignore, light_dist, k);
}
void orc_kat_phong(const float *n, const float *l, float dist, float intensity, const float *mat, const float *view,
                   float *out) {
    Material m{mat[0], mat[1], mat[2], mat[3], mat[4], mat[5]};
    phong_ds(Vec3(n[0], n[1], n[2]), Vec3(l[0], l[1], l[2]), dist, intensity, m, Vec3(view[0], view[1], view[2]),
             out[0], out[1]);
}
// points: 4 x {x, y, r, g, b}
void orc_kat_bilinear(float x, float y, const float *pts, float *out) {
    std::array<std::tuple<float, float, Colour>, 4> p;
    for (int i = 0; i < 4; i++)
        p[i] = std::make_tuple(pts[5 * i], pts[5 * i + 1], Colour(pts[5 * i + 2], pts[5 * i + 3], pts[5 * i + 4]));
    Colour c = bilinear_interpolation(x, y, p);
    for (int i = 0; i < 3; i++) out[i] = c.l[i];
}
void orc_kat_sample_lightmap(void *h, int obj, const float *p, float *out) {
    Colour c = static_cast<Engine *>(h)->objects[obj]->sample_lightmap(Vec3(p[0], p[1], p[2]));
    for (int i = 0; i < 3; i++) out[i] = c.l[i];
}
void orc_kat_sample_pos(void *h, int obj, int u, int v, float *out) {
    bool ok = true;
    Vec3 p = static_cast<Engine *>(h)->objects[obj]->get_sample_pos(u, v, ok);
    for (int i = 0; i < 3; i++) out[i] = p.l[i];
}
void orc_kat_pixel_direction(int W, int H, int x, int y_inv, const float *pose5, float *out) {
    Engine::Pose p{pose5[0], pose5[1], pose5[2], pose5[3], pose5[4], 1.0f};
    Vec3 d = Engine::pixel_direction(p, W, H, x, y_inv);
    for (int i = 0; i < 3; i++) out[i] = d.l[i];
}
void orc_kat_shade(void *h, int obj, const float *pos, const float *dir, float *out) {
    Colour c = static_cast<Engine *>(h)->shade_object((uint64_t)obj, Vec3(pos[0], pos[1], pos[2]),
                                                      Vec3(dir[0], dir[1], dir[2]), 0);
    for (int i = 0; i < 3; i++) out[i] = c.l[i];
}
void orc_kat_refract(const float *v, const float *n, float ior, float *out) {
    Vec3 r = Vec3(v[0], v[1], v[2]).refract(Vec3(n[0], n[1], n[2]), ior);
    for (int i = 0; i < 3; i++) out[i] = r.l[i];
}

}  // extern "C"
