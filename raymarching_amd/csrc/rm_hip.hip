// HIP kernels and the C ABI (include/rm.h) of the ray-marching hot path, gfx950.
//
// Host side: scene flattening, device buffers, the camera block of
// Engine::render (engine.rs:63-75) and launch geometry.  Device side: one
// kernel for the whole per-pixel path (engine.rs:77-126), the radiosity
// precompute (engine.rs:246-326, radiosity.rs) and FXAA (fxaa.txt).
//
// Citations are file:line in the reference tree (src/...).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rm.h"
#include "rm_device.hpp"
#include "rm_frame.hpp"
#include "rm_types.hpp"

// ===========================================================================
// Kernels
// ===========================================================================

// `count`: word 0 = slow tiles found (zero when the kernel starts), word 1 = the mean cost of a sample
// of the launch's tiles.  `count_next` = the count word of the stream's next plan, cleared here: a
// stream's plans alternate between two sets of words, and the set not in use was last read by the
// launch before this one, which has finished (same stream).  `previous` = the same words of the engine's plan before this one: a tile
// counts as slow only at `times_mean` (3) times that launch's mean, so that a scene in which every
// tile is slow is not rendered in quarters throughout.
__device__ __forceinline__ unsigned tile_cost_of(const RenderArgs &a, unsigned t) {
    const int by = a.blocks_y - 1 - (int)(t / (unsigned)a.blocks_x), bx = (int)(t % (unsigned)a.blocks_x);
    const uint4 c = reinterpret_cast<const uint4 *>(a.tile_cost)[frame_tile_of(a, bx, by)];
    return max(max(c.x, c.y), max(c.z, c.w));
}
__global__ __launch_bounds__(256) void split_plan_kernel(const RenderArgs a, unsigned *count, unsigned *list, unsigned char *flag,
                                                         unsigned threshold, const unsigned *previous, unsigned times_mean,
                                                         unsigned *count_next) {
    // (tiles are looked at last-handed-out first: if the list overflows, the slow tiles left off it are
    // those the usual order starts early anyway)
    const unsigned n = (unsigned)(a.blocks_x * a.blocks_y);
    const unsigned t = n - 1u - (blockIdx.x * 256 + threadIdx.x);  // (wraps to >= n beyond the launch)
    threshold = max(threshold, times_mean * previous[1]);
    bool slow = t < n && tile_cost_of(a, t) >= threshold;
    if (blockIdx.x == 0) {
        // the mean over 256 tiles spread evenly through the launch, for the next plan
        __shared__ unsigned sum16;
        if (threadIdx.x == 0) sum16 = 0;
        __syncthreads();
        const unsigned sample = (unsigned)(((unsigned long long)threadIdx.x * n) >> 8);
        atomicAdd(&sum16, min(tile_cost_of(a, sample), 1u << 20) >> 4);
        __syncthreads();
        if (threadIdx.x == 0) {
            count[1] = sum16 >> 4;  // (/ 256 tiles * 16)
            *count_next = 0;
        }
    }
    // one atomic per wave that has a slow tile
    const unsigned long long slow_lanes = __builtin_amdgcn_ballot_w64(slow);
    if (slow) {
        const unsigned lane = threadIdx.x & 63;
        const unsigned before = (unsigned)__builtin_popcountll(slow_lanes & ((1ULL << lane) - 1ULL));
        unsigned base = 0;
        if (before == 0) base = atomicAdd(count, (unsigned)__builtin_popcountll(slow_lanes));
        base = __builtin_amdgcn_readfirstlane(base);  // (the first slow lane is the first active one here)
        const unsigned at = base + before;
        if (at < a.plan_cap)
            list[at] = t;
        else
            slow = false;  // the list is full: the tile is rendered whole
    }
    if (t < n) flag[t] = slow ? 1 : 0;
}


// (The plain frame kernel of the small tiers is held to the 72 registers of seven waves: left alone
// the compiler takes 74-77 since the shading's loads went global and more of them fly at once, and a
// sixth of the wave slots would go for nothing.)
template <class G>
__global__ __launch_bounds__(BLOCK_THREADS, MinWavesFrame<G>::value) void render_kernel(const RenderArgs a, const G g) {
    render_tile(a, g);
}

// The same kernel compiled for 8 waves per SIMD, for launches of many times the machine's wave slots.
// A long launch of the small tiers runs 3-4 % faster with the eighth wave (3840x2160: 0.271 against
// 0.281 ms), although 4 vector and 11 scalar values then live in private memory; a short launch
// lasts as long as its slowest wave, which those spills lengthen (1920x1080: 0.164 against 0.159 ms).
template <class G>
__global__ __launch_bounds__(BLOCK_THREADS, MinWavesDense<G>::value) void render_dense_kernel(const RenderArgs a, const G g) {
    render_tile<G, false>(a, g);
}
// the same two kernels in the fast tier's arithmetic (rm_fast.hip)
void rm_fast_launch(int tier, const void *scene, bool dense, dim3 grid, hipStream_t stream, const RenderArgs &a);
constexpr unsigned kDenseLaunchTiles = 65536;  // between 1920x1080 (32400 tiles) and 3840x2160 (129600)
// ... and where a group keeps three or more (two) frames in flight: the machine then stays full across
// launches, and the 8-wave build pays for smaller ones (shares of the 3840x2160 frame, 8-wave build
// against the 7-wave one: a half 0.129/0.128 against 0.135/0.135 ms with two/three in flight, a quarter
// 0.077/0.065 against 0.072/0.067 ms)
constexpr unsigned kOverlapDenseTiles3 = 20000, kOverlapDenseTiles2 = 50000;

// ---- the frame in two passes (an experiment: RM_TWO_PASS=1) -------------------------------------
//
// Pass 1 takes every pixel to its first hit and shades it there (engine.rs:129-194); a pixel whose
// hit is reflective (engine.rs:231) leaves its reflection ray and what the unwinding needs - the local
// colour, the object colour and the weight of engine.rs:239-240 - in a queue instead of following the
// ray.  Pass 2 takes the queue in full waves: every lane follows one reflection ray to the end of its
// chain (cast_sight_ray, one level deeper) and finishes its pixel.  Same arithmetic in the same order
// per pixel, so the frame is the same byte for byte; what changes is which rays share a wave.
struct ReflectQueue {
    unsigned *count;     // entries appended so far
    float *rec;          // [14][capacity]: ray origin, ray direction, local colour, object colour, weight; pixel offset (bits)
    unsigned capacity;
};

template <class G>
__global__ __launch_bounds__(BLOCK_THREADS) void render_primary_kernel(const RenderArgs a, const ReflectQueue q, const G g) {
    int x, y_inv, out_row;
    tile_pixel(a, x, y_inv, out_row);
    if (x >= a.pose.W || y_inv >= a.row_end) return;
    SceneRef sv = scene_ref(a.sv);
    V3 dir = pixel_direction(a.pose, x, y_inv, a.screen_u, a.screen_v);
    V3 pos = v3(a.pose.cam[0], a.pose.cam[1], a.pose.cam[2]);
    int steps;
    bool capped;
    const int idx = march(acc_of(g), pos, dir, -1, sv.march_cap, steps, capped, a.camera_dist, a.camera_sph);
    if (capped) atomicAdd(&sv.stats[1], 1ULL);
    V3 colour = v3(sv.sky[0], sv.sky[1], sv.sky[2]);
    bool queued = false;
    const unsigned offset = (unsigned)((size_t)out_row * a.pitch + (size_t)x * 4);
    if (idx >= 0) {
        const ShadeResult r = shade_local(acc_of(g), sv, idx, pos, dir);
        colour = r.final_colour;
        queued = r.reflectivity > 1e-3f;
        const LaneMask want = __builtin_amdgcn_ballot_w64(queued);
        if (want != 0) {
            unsigned base = 0;
            if (__builtin_amdgcn_readfirstlane(threadIdx.x) == threadIdx.x) base = atomicAdd(q.count, (unsigned)__builtin_popcountll(want));
            base = __builtin_amdgcn_readfirstlane(base);
            if (queued) {
                const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
                const float fresnel = powi5(rs_clamp(1.0f - dot(r.n, -dir), 0.0f, 1.0f));
                const V3 rv = reflect(dir, r.n);
                const float kk = rs_clamp(fresnel + r.reflectivity, 0.0f, 1.0f);
                const V3 rp = pos + ((rv * 3.0f) * SMALL_DISTANCE);  // engine.rs:237
                const float vals[14] = {rp.x, rp.y, rp.z, rv.x, rv.y, rv.z, r.final_colour.x, r.final_colour.y, r.final_colour.z,
                                        r.object_colour.x, r.object_colour.y, r.object_colour.z, kk, __uint_as_float(offset)};
#pragma unroll
                for (int k = 0; k < 14; k++) q.rec[(size_t)k * q.capacity + slot] = vals[k];
            }
        }
    }
    if (!queued) *reinterpret_cast<unsigned *>(a.out + offset) = pack_pixel(colour, a.pose.exposure);
}

template <class G>
__global__ __launch_bounds__(BLOCK_THREADS) void render_reflect_kernel(const RenderArgs a, const ReflectQueue q, const G g) {
    SceneRef sv = scene_ref(a.sv);
    const unsigned count = *q.count;
    for (unsigned i = blockIdx.x * BLOCK_THREADS + threadIdx.x; i < count; i += gridDim.x * BLOCK_THREADS) {
        float v[14];
#pragma unroll
        for (int k = 0; k < 14; k++) v[k] = q.rec[(size_t)k * q.capacity + i];
        const V3 rc = cast_sight_ray(acc_of(g), sv, v3(v[0], v[1], v[2]), v3(v[3], v[4], v[5]), __builtin_nanf(""), __builtin_nanf(""), 1);
        const V3 colour = v3(v[6], v[7], v[8]) + emul(rc, v3(v[9], v[10], v[11])) * v[12];  // engine.rs:239-240
        *reinterpret_cast<unsigned *>(a.out + __float_as_uint(v[13])) = pack_pixel(colour, a.pose.exposure);
    }
}

// engine.rs:95-108 (disabled in the reference): four rays per pixel, half a pixel step left, right,
// up and down of the pixel's direction before normalising, averaged in that order.
template <class G>
__global__ __launch_bounds__(BLOCK_THREADS, MinWaves<G>::value) void render_ssaa_kernel(const RenderArgs a, const G g) {
    int x, y_inv, out_row;
    tile_pixel(a, x, y_inv, out_row);
    if (x >= a.pose.W || y_inv >= a.row_end) return;
    const V3 dir = pixel_direction(a.pose, x, y_inv, a.screen_u, a.screen_v);
    const V3 cam = v3(a.pose.cam[0], a.pose.cam[1], a.pose.cam[2]);
    const float ox = 0.5f / (float)a.pose.W, oy = 0.5f / (float)a.pose.H;
    V3 sum = v3(0.0f, 0.0f, 0.0f);
    for (int i = 0; i < 4; i++) {
        const V3 off = i == 0 ? v3(ox, 0.0f, 0.0f) : i == 1 ? v3(-ox, 0.0f, 0.0f) : i == 2 ? v3(0.0f, oy, 0.0f) : v3(0.0f, -oy, 0.0f);
        const V3 c = cast_sight_ray(acc_of(g), scene_ref(a.sv), cam, normalized(dir + off), a.camera_dist, a.camera_sph);
        sum = i == 0 ? c : sum + c;
    }
    unsigned px = pack_pixel(sum / 4.0f, a.pose.exposure);
    *reinterpret_cast<unsigned *>(a.out + (size_t)out_row * a.pitch + (size_t)x * 4) = px;
}

// The frame with the refraction block of engine.rs:196-228 switched on; `samples` = 4 adds the
// supersampling of engine.rs:95-108.  A kernel of its own: the tree walk needs more registers and
// private memory than the plain frame should pay for.
template <class G>
__global__ __launch_bounds__(BLOCK_THREADS, MinWaves<G>::value) void render_tree_kernel(const RenderArgs a, int samples, const G g) {
    int x, y_inv, out_row;
    tile_pixel(a, x, y_inv, out_row);
    if (x >= a.pose.W || y_inv >= a.row_end) return;
    const V3 dir = pixel_direction(a.pose, x, y_inv, a.screen_u, a.screen_v);
    const V3 cam = v3(a.pose.cam[0], a.pose.cam[1], a.pose.cam[2]);
    V3 colour;
    if (samples == 4) {
        const float ox = 0.5f / (float)a.pose.W, oy = 0.5f / (float)a.pose.H;
        V3 sum = v3(0.0f, 0.0f, 0.0f);
        for (int i = 0; i < 4; i++) {
            const V3 off = i == 0 ? v3(ox, 0.0f, 0.0f) : i == 1 ? v3(-ox, 0.0f, 0.0f) : i == 2 ? v3(0.0f, oy, 0.0f) : v3(0.0f, -oy, 0.0f);
            const V3 c = cast_sight_tree(acc_of(g), scene_ref(a.sv), cam, normalized(dir + off), -1, a.camera_dist, a.camera_sph);
            sum = i == 0 ? c : sum + c;
        }
        colour = sum / 4.0f;
    } else {
        colour = cast_sight_tree(acc_of(g), scene_ref(a.sv), cam, dir, -1, a.camera_dist, a.camera_sph);
    }
    *reinterpret_cast<unsigned *>(a.out + (size_t)out_row * a.pitch + (size_t)x * 4) = pack_pixel(colour, a.pose.exposure);
}

// Primary march only: hit object, iteration count, final position.
template <class G>
__global__ __launch_bounds__(BLOCK_THREADS) void trace_kernel(const RenderArgs a, int *hit, int *steps_out,
                                                              float *pos_out, const G g) {
    int x, y_inv, out_row;
    tile_pixel(a, x, y_inv, out_row);
    if (x >= a.pose.W || y_inv >= a.row_end) return;
    V3 dir = pixel_direction(a.pose, x, y_inv, a.screen_u, a.screen_v);
    V3 pos = v3(a.pose.cam[0], a.pose.cam[1], a.pose.cam[2]);
    int steps;
    bool capped;
    int idx = march(acc_of(g), pos, dir, -1, scene_ref(a.sv).march_cap, steps, capped, a.camera_dist, a.camera_sph);
    size_t i = (size_t)y_inv * a.pose.W + x;
    if (hit) hit[i] = idx;
    if (steps_out) steps_out[i] = steps;
    if (pos_out) {
        pos_out[3 * i] = pos.x;
        pos_out[3 * i + 1] = pos.y;
        pos_out[3 * i + 2] = pos.z;
    }
}

// Engine::shade_object at arbitrary points (engine.rs:147-244).
template <class G>
__global__ void probe_shade_kernel(const SceneView *svp, int n, const int *obj, const float *pos, const float *dir,
                                   float *out, const G g) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SceneRef sv = scene_ref(svp);
    V3 p = v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    V3 d = v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    int idx = obj[i];
    auto body = [&](auto sc) {
        if (sv.refraction) {
            const V3 c = cast_sight_tree(sc, sv, p, d, idx);
            out[3 * i] = c.x;
            out[3 * i + 1] = c.y;
            out[3 * i + 2] = c.z;
            return;
        }
        ShadeResult r = shade_local(sc, sv, idx, p, d);
        V3 c = r.final_colour;
        if (r.reflectivity > 1e-3f) {
            float fresnel = powi5(rs_clamp(1.0f - dot(r.n, -d), 0.0f, 1.0f));
            V3 rv = reflect(d, r.n);
            V3 rc = cast_sight_ray(sc, sv, p + ((rv * 3.0f) * SMALL_DISTANCE), rv);
            c = c + emul(rc, r.object_colour) * rs_clamp(fresnel + r.reflectivity, 0.0f, 1.0f);
        }
        out[3 * i] = c.x;
        out[3 * i + 1] = c.y;
        out[3 * i + 2] = c.z;
    };
    body(acc_of(g));
}

// Tone map and pack (engine.rs:109-125) for arbitrary linear colours: the path the frames take
// (estimate, exact operations only near a rounding boundary) and the exact operations alone.
__global__ void probe_pack_kernel(int n, const float *linear, float exposure, unsigned *packed, unsigned *packed_exact) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 c = v3(linear[3 * i], linear[3 * i + 1], linear[3 * i + 2]);
    packed[i] = pack_pixel(c, exposure);
    packed_exact[i] = pack_pixel_exact(c * exposure);
}

// ---- radiosity -------------------------------------------------------------

// Device memory the lightmap precompute works in besides the scene and the lightmaps themselves.
struct RadBuffers {
    float *col = nullptr, *nrm = nullptr;                      // [NP][3] clouds (engine.rs:274-281)
    float *emissive = nullptr, *emissive_next = nullptr;       // [NP][3]
    float *ft = nullptr;                                       // form factors, rad_ft_floats(NP) of them (blocks of 64 receivers, rad_ft_at)
};

struct RadArgs {
    const SceneView *sv;
    int n_patches;          // n_lm * M * M
    const int *patch_obj;   // [n_lm] engine index of each cloud object
    const float *pos;       // [NP][3] point cloud (engine.rs:272)
    float *col;             // [NP][3] colour cloud
    float *nrm;             // [NP][3] normal cloud
    float *emissive;        // [NP][3] current emissive maps
    float *emissive_next;   // [NP][3]
    float *lightmaps;       // [NP][3]
    float *ft;              // form factors, source j -> receiver i at rad_ft_at(i, j, NP); -1 = no contribution
};
// The form-factor matrix lies in blocks of 64 receivers: block b holds its receivers' values for
// source 0, then for source 1, ... (64 floats each), so the workgroup of a bounce, which folds every
// source into 64 receivers, reads one unbroken run of memory, and the wave of a form-factor launch
// (64 receivers, one source) writes one 256-byte piece of it.  ceil(NP / 64) * 64 * NP floats.
__host__ __device__ __forceinline__ size_t rad_ft_at(int i, int j, int NP) { return ((size_t)(i >> 6) * NP + j) * 64 + (i & 63); }
// (plus room for the deepest slab of a bounce to read past the last source of the last block: what
// such a read brings is discarded, but it has to stay inside the allocation)
constexpr size_t kRadFtPadFloats = 192 * 64;
__host__ __device__ __forceinline__ size_t rad_ft_floats(int NP) { return (size_t)((NP + 63) / 64) * 64 * NP + kRadFtPadFloats; }

// engine.rs:269-281: colour and normal of every sample point.
__global__ void rad_cloud_kernel(const RadArgs a) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_patches) return;
    const int mm = scene_ref(a.sv).map_size * scene_ref(a.sv).map_size;
    int obj = a.patch_obj[p / mm];
    V3 pos = v3(a.pos[3 * p], a.pos[3 * p + 1], a.pos[3 * p + 2]);
    const ShadeRec sr = scene_ref(a.sv).shade[obj];
    const Geom g = scene_ref(a.sv).geom[obj];
    V3 c = object_colour(scene_ref(a.sv), sr, g, pos);
    V3 n = calculate_normal(sr.type, g, pos, scene_ref(a.sv).tetra_normals != 0);
    a.col[3 * p] = c.x;
    a.col[3 * p + 1] = c.y;
    a.col[3 * p + 2] = c.z;
    a.nrm[3 * p] = n.x;
    a.nrm[3 * p + 1] = n.y;
    a.nrm[3 * p + 2] = n.z;
}

// radiosity.rs:52-101 for one patch
template <class G>
__device__ __forceinline__ void rad_direct_patch(const RadArgs &a, const G &grp, int p) {
    const int mm = scene_ref(a.sv).map_size * scene_ref(a.sv).map_size;
    int obj = a.patch_obj[p / mm];
    const ShadeRec sr = scene_ref(a.sv).shade[obj];
    const Geom g = scene_ref(a.sv).geom[obj];
    V3 origin = v3(a.pos[3 * p], a.pos[3 * p + 1], a.pos[3 * p + 2]);
    V3 colour = v3(a.col[3 * p], a.col[3 * p + 1], a.col[3 * p + 2]);
    V3 light = v3(scene_ref(a.sv).light[0], scene_ref(a.sv).light[1], scene_ref(a.sv).light[2]);
    V3 to_light = light - origin;
    float dist = mag(to_light);
    to_light = to_light / dist;
    V3 rp = origin;
    int steps;
    bool capped;
    march(acc_of(grp), rp, to_light, obj, 0, steps, capped);
    V3 em = v3(0.0f, 0.0f, 0.0f), lm = v3(0.0f, 0.0f, 0.0f);
    if (!(mag(rp - origin) < (dist + 3.0f * SMALL_DISTANCE))) {
        V3 n = calculate_normal(sr.type, g, origin, scene_ref(a.sv).tetra_normals != 0);
        float diffuse = rs_max(dot(n, to_light), 0.0f) * scene_ref(a.sv).intensity / (dist * dist);
        em = colour * (diffuse + sr.emissive);
        lm = colour * sr.emissive;
    }
    a.emissive[3 * p] = em.x;
    a.emissive[3 * p + 1] = em.y;
    a.emissive[3 * p + 2] = em.z;
    a.lightmaps[3 * p] = lm.x;
    a.lightmaps[3 * p + 1] = lm.y;
    a.lightmaps[3 * p + 2] = lm.z;
}
template <class G>
__global__ void rad_direct_kernel(const RadArgs a, const G grp) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_patches) return;
    rad_direct_patch(a, grp, p);
}

// radiosity.rs:119-152 for one (receiver i, source j) pair: everything of the
// inner loop that does not depend on the bounce.  x = receiver, y = source.
// `direct` = 1: the launch has one more row of workgroups, its first, and that row does the direct
// lighting of the receivers instead (rad_direct_patch: it reads the clouds like this kernel and writes
// the emissive map and the lightmaps, which this kernel does not touch) - a few waves, each one long
// march, that would otherwise be a launch of their own with the machine all but empty.
template <class G>
__device__ __forceinline__ float rad_form_pair(const RadArgs &a, const G &grp, int j, int recv_obj, V3 sample_pos, V3 sample_n) {
    V3 src = v3(a.pos[3 * j], a.pos[3 * j + 1], a.pos[3 * j + 2]);
    V3 src_n = v3(a.nrm[3 * j], a.nrm[3 * j + 1], a.nrm[3 * j + 2]);
    V3 to_light = src - sample_pos;
    float dist = mag(to_light);
    to_light = to_light / dist;
    // radiosity.rs:147-152: what the pair contributes per unit of emission if nothing is in the way
    const float att = dot(sample_n, to_light) * -dot(src_n, to_light);
    const float diffuse = rs_max(att, 0.0f) / (dist * dist);
    const float open = rs_min(diffuse, 2.0f);
    // A pair that faces away contributes emission * 0 if it is open and nothing if it is blocked:
    // +0 or -0 added to a sum that started at +0 leaves it as it is either way (emission values are
    // finite), so the occlusion march is run only where its outcome can show.  A wave whose 64
    // receivers all face away from the source skips it altogether.
    // (With emission values that are not finite the two cases differ - infinity * 0 is NaN where the
    // reference's blocked pair adds nothing - and rm_engine_set_exact_occlusion makes every pair march.)
    const bool matters = !(open == 0.0f) || scene_ref(a.sv).exact_occlusion != 0;
    float f = open;
    if (wave_any(matters)) {
        if (matters) {
            V3 back = -to_light;
            V3 rp = src + back * (3.0f * SMALL_DISTANCE);  // radiosity.rs:131
            int steps;
            bool capped;
            int hit = march(acc_of(grp), rp, back, -1, 0, steps, capped);
            bool blocked = (hit >= 0) && (hit != recv_obj) && (mag_sqd(rp - sample_pos) > 0.1f * 0.1f);
            if (blocked) f = -1.0f;
        }
    }
    return f;
}
template <class G>
__global__ void rad_form_kernel(const RadArgs a, int p0, int p1, int j0, int direct, const G grp) {
    int i = p0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p1) return;
    if (direct && blockIdx.y == 0) {
        rad_direct_patch(a, grp, i);
        return;
    }
    int j = j0 + (int)blockIdx.y - direct;
    const int mm = scene_ref(a.sv).map_size * scene_ref(a.sv).map_size;
    float f = -1.0f;
    if (i / mm != j / mm) {  // radiosity.rs:112: no self illumination
        int recv_obj = a.patch_obj[i / mm];
        V3 sample_pos = v3(a.pos[3 * i], a.pos[3 * i + 1], a.pos[3 * i + 2]);
        V3 sample_n = v3(a.nrm[3 * i], a.nrm[3 * i + 1], a.nrm[3 * i + 2]);
        f = rad_form_pair(a, grp, j, recv_obj, sample_pos, sample_n);
    }
    a.ft[rad_ft_at(i, j, a.n_patches)] = f;
}
// The same with `per` sources per workgroup, one after the other: what a wave sets up - the scene in
// scalar registers, its receivers' points and normals - then serves several marches.  For launches
// that take every wave slot of the machine many times over; a small map needs every wave it can get,
// one source each (the kernel above).
template <class G>
__global__ void rad_form_multi_kernel(const RadArgs a, int p0, int p1, int j0, int j1, int per, const G grp) {
    int i = p0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p1) return;
    const int jb = j0 + (int)blockIdx.y * per, je = min(jb + per, j1);
    const int mm = scene_ref(a.sv).map_size * scene_ref(a.sv).map_size;
    const int recv_obj = a.patch_obj[i / mm];
    const V3 sample_pos = v3(a.pos[3 * i], a.pos[3 * i + 1], a.pos[3 * i + 2]);
    const V3 sample_n = v3(a.nrm[3 * i], a.nrm[3 * i + 1], a.nrm[3 * i + 2]);
    for (int j = jb; j < je; j++) {
        float f = -1.0f;
        if (i / mm != j / mm) f = rad_form_pair(a, grp, j, recv_obj, sample_pos, sample_n);  // radiosity.rs:112
        a.ft[rad_ft_at(i, j, a.n_patches)] = f;
    }
}

// radiosity.rs:107-156 + 184-192 for 64 receivers per workgroup: the ordered sums over sources.
// The additions of one receiver and channel must run in source order (f32 addition does not
// associate), but fetching the form factors need not: all 256 threads stream a slab of FT
// (CHUNK sources x 64 receivers, receiver-contiguous rows) and the matching emissive
// values into LDS with many loads in flight, then three waves - one per colour channel - fold the
// slab in order, one receiver per lane, and the next slab follows.  A pair that contributes nothing
// (form factor -1) enters the slab as 0: its term emission * 0 is +0 (emission values are finite and
// not negative), which leaves a sum that started at +0 unchanged bit for bit.
// The slab depth CHUNK (sources per slab) sets how many bytes a workgroup keeps in flight: 64 sources
// are 16 KiB, enough once eight workgroups share a compute unit; with fewer workgroups (smaller
// lightmaps) deeper slabs keep the memory system as busy.  The order of the additions does not depend
// on it.
// EXACT (rm_engine_set_exact_occlusion): a pair that contributes nothing keeps its marker in the slab and
// its term is left out of the sum, as the reference's `continue` leaves it out (radiosity.rs:112,137):
// with an emission value that is not finite, emission * 0 would be NaN instead of nothing.
template <int CHUNK, bool EXACT = false>
__global__ __launch_bounds__(256) void rad_bounce_kernel(const RadArgs a, int p0, int p1) {
    // The slab lies receiver-major in LDS, a receiver's CHUNK sources side by side (rows of CHUNK + 4
    // words: 16-byte aligned, and sixteen lanes' 16-byte accesses fall on all 64 banks once), and the
    // emissive values channel-major: a fold step's operands then come four sources to a 16-byte read -
    // eight reads per sixteen sources where single words took thirty-two, and for the lone wave per
    // SIMD that folds a small map those reads are most of a bounce's time.
    constexpr int kStride = CHUNK + 4;
    __shared__ __attribute__((aligned(16))) float slab[64][kStride];
    __shared__ __attribute__((aligned(16))) float em_s[3][CHUNK];
    constexpr int kGroups = CHUNK / 16;               // groups of four rows of a slab each thread fetches
    constexpr int kPerThread = 4 * kGroups;
    constexpr int kEmPerThread = (CHUNK * 3 + 255) / 256;  // emissive values of a slab each thread fetches
    const int NP = a.n_patches;
    const int i0 = p0 + blockIdx.x * 64;
    // part: row-group phase when loading, channel when summing; the same in every lane of a wave, and
    // said so, so that the row numbers and row addresses below are worked out on the scalar unit
    const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = i0 + lane;
    const bool valid = i < p1;
    // The next slab travels from memory into registers while the current one is folded, so a fold
    // waits for memory only once, at the first slab (with few workgroups in flight - 8 for the 512
    // patches of the default scene - nothing else would hide that wait).
    float pre[kPerThread], em_pre[kEmPerThread];
    // Every load of a fetch is issued unconditionally, at an address clamped into the arrays: a load
    // under a condition becomes a branch with a wait behind it, and the fetches of a slab would go to
    // memory one after the other.  What a clamped load brought is discarded when the slab is stored.
    // Thread (part, lane) fetches rows 4g .. 4g+3 for g = part, part + 4, ...: each load is a row
    // segment of 64 receivers, and the four values of a group leave for LDS in one 16-byte store.
    // Rows past the last source are read where they would lie (the next block, or the padding behind
    // the last one): one address per group, the four rows at fixed distances from it.
    static_assert((size_t)CHUNK * 64 <= kRadFtPadFloats, "the padding behind the matrix covers a slab");
    const int ic = min(i, NP - 1);
    const float *const column = a.ft + rad_ft_at(ic, 0, NP);
    const auto fetch = [&](int j0) {
#pragma unroll
        for (int k = 0; k < kGroups; k++) {
            const float *const g = column + (size_t)(j0 + 4 * (part + 4 * k)) * 64;
#pragma unroll
            for (int q = 0; q < 4; q++) pre[4 * k + q] = g[q * 64];
        }
#pragma unroll
        for (int k = 0; k < kEmPerThread; k++) em_pre[k] = a.emissive[min(3 * j0 + (int)threadIdx.x + 256 * k, 3 * NP - 1)];
    };
    fetch(0);
    float acc = 0.0f;
    for (int j0 = 0; j0 < NP; j0 += CHUNK) {
        const int rows = min(CHUNK, NP - j0);
#pragma unroll
        for (int k = 0; k < kGroups; k++) {
            const int row = 4 * (part + 4 * k);
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // a pair that contributes nothing (-1), a row beyond the last source, a receiver beyond p1: emission * 0 = +0
                const bool counts = valid && row + q < rows && !(pre[4 * k + q] == -1.0f);
                v[q] = counts ? pre[4 * k + q] : (EXACT ? -1.0f : 0.0f);
            }
            *reinterpret_cast<float4 *>(&slab[lane][row]) = make_float4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int k = 0; k < kEmPerThread; k++) {
            const int t = (int)threadIdx.x + 256 * k;
            if (t < CHUNK * 3) em_s[t % 3][t / 3] = t < rows * 3 ? em_pre[k] : 0.0f;
        }
        __syncthreads();
        if (j0 + CHUNK < NP) fetch(j0 + CHUNK);
        if (part < 3 && valid) {
            // The additions form one dependent chain per receiver and channel; everything else of a
            // step - the reads from the slab and the product - does not depend on the sum so far.
            // Sixteen steps at a time: their operands are fetched and multiplied first, then added in
            // source order, so that a step costs one addition's latency and not a round trip to LDS.
            int jj = 0;
            if constexpr (EXACT) {
                for (; jj < rows; jj++) {
                    const float f = slab[lane][jj];
                    if (!(f == -1.0f)) acc = acc + em_s[part][jj] * f;
                }
            }
            for (; jj + 16 <= rows; jj += 16) {
                float term[16];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float4 e = *reinterpret_cast<const float4 *>(&em_s[part][jj + 4 * u]);
                    const float4 f = *reinterpret_cast<const float4 *>(&slab[lane][jj + 4 * u]);
                    term[4 * u] = e.x * f.x;
                    term[4 * u + 1] = e.y * f.y;
                    term[4 * u + 2] = e.z * f.z;
                    term[4 * u + 3] = e.w * f.w;
                }
#pragma unroll
                for (int u = 0; u < 16; u++) acc = acc + term[u];
            }
            for (; jj < rows; jj++) acc = acc + em_s[part][jj] * slab[lane][jj];
        }
        __syncthreads();
    }
    if (part < 3 && valid) {
        const float incident = acc / 3.14159265358979323846f;
        a.lightmaps[3 * i + part] = a.lightmaps[3 * i + part] + incident;
        a.emissive_next[3 * i + part] = incident * a.col[3 * i + part];
    }
}

// ---- FXAA (fxaa.txt:1-135, out of place) -------------------------------------

// b / 255.0f for a byte b, correctly rounded, in two operations instead of a full division:
// 1/255 = c + c_lo with c the nearest float, and fma(b, c, b * c_lo) rounds b * c (exact inside the
// fused operation) plus a correction that is right to 2^-24 of itself - b/255 is never that close to
// a rounding boundary.  Checked against the IEEE quotient for all 256 inputs (tests/test_host.py).
__device__ __forceinline__ float fx_unorm(float b) {
    const float c = 1.0f / 255.0f;
    const float c_lo = (float)(1.0 / 255.0 - (double)c);
    return __builtin_fmaf(b, c, b * c_lo);
}
__device__ __forceinline__ void fx_unpack(unsigned px, float c[3]) {
    c[0] = fx_unorm((float)(px & 255u));
    c[1] = fx_unorm((float)((px >> 8) & 255u));
    c[2] = fx_unorm((float)((px >> 16) & 255u));
}
__device__ __forceinline__ float fx_luma(const float c[3]) { return (0.299f * c[0] + 0.587f * c[1]) + 0.114f * c[2]; }
// `round() as usize` of a tap offset (fxaa.txt:90-97): half away from zero, negative becomes 0.  The
// offsets here are finite and at most 4 in size, so o - floor(o) is exact and the rounding is
// floor(o) plus one when that fraction reaches a half; whatever a negative offset gives is <= 0.
__device__ __forceinline__ int fx_off(float o) {
    const float fl = floorf(o);
    const int r = (int)fl + ((o - fl) >= 0.5f ? 1 : 0);
    return max(r, 0);
}
// `(255.0 * c) as u8` (fxaa.txt:131-133): truncation; the values are averages of bytes over 255, so
// they lie in [0, 255] and the clamp never binds
__device__ __forceinline__ unsigned fx_u8(float f) { return (unsigned)min(max((int)f, 0), 255); }

// The value of the lane below / above in the wave (all 64 lanes must be running).
__device__ __forceinline__ float fx_from_lane_below(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float fx_from_lane_above(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// fxaa.txt:66-133 for a pixel whose taps may leave it: the direction, the taps, the two averages and
// the choice between them.  `i` is the pixel's index, `centre` its four bytes.
__device__ __forceinline__ unsigned fx_filtered(const unsigned char *__restrict__ src, unsigned i, unsigned W, unsigned centre,
                                                float lNW, float lNE, float lSW, float lSE, float lM, float d0, float d1, float den) {
    const auto in = [src](unsigned pixel) { return *reinterpret_cast<const unsigned *>(src + (pixel << 2)); };
    const float lmin = rs_min(lM, rs_min(rs_min(lNW, lNE), rs_min(lSW, lSE)));
    const float lmax = rs_max(lM, rs_max(rs_max(lNW, lNE), rs_max(lSW, lSE)));
    const float rcp = 1.0f / den;
    d0 = rs_min(8.0f, rs_max(-8.0f, d0 * rcp));
    d1 = rs_min(8.0f, rs_max(-8.0f, d1 * rcp));
    // fxaa.txt:80-88: the second tap of each pair sits at the same factor, 2/2 - 1/2 = 3/3 - 1/2 = 1/2
    // (the text's "2.0 / 2.0" where FXAA has 2/3): one fetch serves both
    const float kA0 = 1.0f / 3.0f - 0.5f, kB0 = 0.0f / 3.0f - 0.5f, k1 = 0.5f;
    static_assert(2.0f / 2.0f - 0.5f == 0.5f && 3.0f / 3.0f - 0.5f == 0.5f, "shared tap");
    const unsigned iA0 = i + (unsigned)fx_off(d1 * kA0) * W + (unsigned)fx_off(d0 * kA0);
    const unsigned iB0 = i + (unsigned)fx_off(d1 * kB0) * W + (unsigned)fx_off(d0 * kB0);
    const unsigned i1 = i + (unsigned)fx_off(d1 * k1) * W + (unsigned)fx_off(d0 * k1);
    float a0[3], t1[3], b0[3], A[3], B[3];
    fx_unpack(in(iA0), a0);
    fx_unpack(in(i1), t1);
    fx_unpack(in(iB0), b0);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        A[c] = 0.5f * (a0[c] + t1[c]);
        B[c] = A[c] * 0.5f + 0.25f * (b0[c] + t1[c]);
    }
    const float lB = fx_luma(B);
    const bool useA = (lB < lmin) || (lB > lmax);
    const unsigned o0 = fx_u8(255.0f * (useA ? A[0] : B[0]));
    const unsigned o1 = fx_u8(255.0f * (useA ? A[1] : B[1]));
    const unsigned o2 = fx_u8(255.0f * (useA ? A[2] : B[2]));
    return o0 | (o1 << 8) | (o2 << 16) | (centre & 0xFF000000u);
}

// A wave filters FX_COLS columns by FX_ROWS rows.  Each lane owns one column: it fetches its pixels
// of the rows around the strip once, forms their lumas once, and takes the lumas of the columns left
// and right of it from the neighbouring lanes - a pixel's four diagonal neighbours cost
// (FX_ROWS + 2) / FX_ROWS fetches and lumas instead of four.  The outermost lanes only supply lumas.
// Byte offsets into the frame fit 32 bits (the host checks the frame size), so every fetch is the
// base pointer plus a 32-bit offset.
#ifndef RM_FX_ROWS
#define RM_FX_ROWS 2
#endif
constexpr int FX_ROWS = RM_FX_ROWS;
constexpr int FX_COLS = 62;
__global__ __launch_bounds__(256) void fxaa_kernel(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst,
                                                   int W, int H, int row_begin, int row_end) {
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * FX_COLS - 1 + lane;
    const int y0 = row_begin + (blockIdx.y * 4 + (threadIdx.x >> 6)) * FX_ROWS;
    if (y0 >= row_end) return;  // the whole wave
    const auto in = [src](unsigned pixel) { return *reinterpret_cast<const unsigned *>(src + (pixel << 2)); };
    const unsigned xc = (unsigned)min(max(x, 0), W - 1);
    unsigned px[FX_ROWS + 2];
    float lu[FX_ROWS + 2], left[FX_ROWS + 2], right[FX_ROWS + 2];
#pragma unroll
    for (int r = 0; r < FX_ROWS + 2; r++) {
        const unsigned yy = (unsigned)min(max(y0 - 1 + r, 0), H - 1);
        px[r] = in(yy * (unsigned)W + xc);
    }
#pragma unroll
    for (int r = 0; r < FX_ROWS + 2; r++) {
        float c[3];
        fx_unpack(px[r], c);
        lu[r] = fx_luma(c);
        left[r] = fx_from_lane_below(lu[r]);
        right[r] = fx_from_lane_above(lu[r]);
    }
    const bool writes = lane >= 1 && lane <= FX_COLS && x < W;
#pragma unroll
    for (int r = 0; r < FX_ROWS; r++) {
        const int y = y0 + r;
        if (y >= row_end) continue;  // the whole wave
        const unsigned i = (unsigned)y * (unsigned)W + xc;
        const unsigned centre = px[r + 1];
        const float lNW = left[r], lNE = right[r], lSW = left[r + 2], lSE = right[r + 2];
        const float d0 = -((lNW + lNE) - (lSW + lSE));
        const float d1 = (lNW + lSW) - (lNE + lSE);
        const float reduce = rs_max((((lNW + lNE) + lSW) + lSE) * (0.25f * (1.0f / 8.0f)), 1.0f / 128.0f);
        const float den = rs_min(fabsf(d0), fabsf(d1)) + reduce;
        // Away from edges the scaled direction stays below 1 in size, every tap offset rounds to 0 and
        // all four taps are the pixel itself: rgbA = 0.5 * (m + m) = m and rgbB = 0.5 * m + 0.25 * (m + m)
        // = m without a rounding, so whichever the luma test picks, a channel comes out as
        // 255 * (b / 255) truncated - and that is b again for every byte b (the product rounds to b
        // exactly; tests/test_host.py goes through all 256).  Such a pixel passes through unchanged, and
        // so does the border fxaa.txt:7-8 leaves out.  |d| <= 0.99 * den keeps d * (1 / den) below 1
        // through the roundings of the reciprocal, of the product and of 0.99 * den.
        const bool calm = rs_max(fabsf(d0), fabsf(d1)) <= 0.99f * den;
        const bool filtered = writes && !calm && x >= 4 && x < W - 4 && y >= 4 && y < H - 4;
        unsigned out = centre;
        if (rm::wave_any(filtered)) {
            if (filtered) out = fx_filtered(src, i, (unsigned)W, centre, lNW, lNE, lSW, lSE, lu[r + 1], d0, d1, den);
        }
        if (writes) *reinterpret_cast<unsigned *>(dst + (i << 2)) = out;
    }
}

// ---- frames between GPUs: 3 bytes per pixel on the wire ------------------------

// Four B,G,R,0 pixels (16 bytes) <-> four B,G,R triples (12 bytes).
__global__ __launch_bounds__(256) void pack_bgr_kernel(const uint4 *__restrict__ src, unsigned *__restrict__ dst, size_t groups) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= groups) return;
    const uint4 p = src[t];
    dst[3 * t] = (p.x & 0xFFFFFFu) | (p.y << 24);
    dst[3 * t + 1] = ((p.y >> 8) & 0xFFFFu) | (p.z << 16);
    dst[3 * t + 2] = ((p.z >> 16) & 0xFFu) | (p.w << 8);
}

// Output row y of the frame comes from packed row (part * per + s / world) * 8 + y % 8 with s = y / 8
// and part = (s % world - shift) mod world: strip s is number s / world of the strips s % world,
// s % world + world, ..., which the rank `shift` places before rank s % world rendered (shift = 0:
// rank s % world itself; a group rotates the assignment from frame to frame, see rm_group_frame).
__global__ __launch_bounds__(256) void unpack_bgr_strips_kernel(const unsigned *__restrict__ src, uint4 *__restrict__ dst, int W4,
                                                                int H, int world, int per, int shift) {
    const int g = blockIdx.x * 256 + threadIdx.x;  // group of four pixels in the row
    const int y = blockIdx.y;
    if (g >= W4) return;
    const int s = y / RM_STRIP_ROWS;
    const size_t srow = (size_t)(((s % world - shift + world) % world) * per + s / world) * RM_STRIP_ROWS + (y % RM_STRIP_ROWS);
    const unsigned *w = src + (srow * W4 + g) * 3;
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
    uint4 p;
    p.x = w0 & 0xFFFFFFu;
    p.y = (w0 >> 24) | ((w1 & 0xFFFFu) << 8);
    p.z = (w1 >> 16) | ((w2 & 0xFFu) << 16);
    p.w = w2 >> 8;
    dst[(size_t)y * W4 + g] = p;
}

// ---- the shader clock, measured ------------------------------------------------

// One wave reads the shader-clock cycle counter and the constant-rate counter (100 MHz), spins until
// `ticks` of the latter have passed and reads both again: cycles per tick is the clock the vector
// units ran at meanwhile.  The loop ends by itself: the constant-rate counter never stops.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *out, unsigned ticks) {
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long t1;
    do {
        __builtin_amdgcn_s_sleep(8);
        t1 = wall_clock64();
    } while (t1 - t0 < (unsigned long long)ticks);
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = t1 - t0;
    }
}

// ===========================================================================
// Host runtime
// ===========================================================================

// Static scratch for rm_engine_compute_lightmaps_static: four [NP][3] float arrays (each rounded up
// to 64 floats) and the form-factor matrix (rad_ft_floats(NP) floats) for NP <= RM_STATIC_MAX_PATCHES.  Every GPU the
// library runs on holds its own copy (4.10 MiB) at its own address until the process ends.  In host
// code the name below is only a handle for hipGetSymbolAddress: the address kernels and memsets are
// given is the one static_scratch_of() returns for the current device.
constexpr size_t kStaticCloudFloats = ((size_t)RM_STATIC_MAX_PATCHES * 3 + 63) & ~(size_t)63;
constexpr size_t kStaticScratchFloats = 4 * kStaticCloudFloats + (size_t)RM_STATIC_MAX_PATCHES * RM_STATIC_MAX_PATCHES + kRadFtPadFloats;
static __device__ __align__(256) float scratch[kStaticScratchFloats];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static float *g_scratch_ptr[kMaxDevices];  // per device, looked up once
// One precompute at a time works in a device's copy of the array: calls from several engines (or
// threads) on one GPU take turns instead of overwriting each other's clouds and form factors.
static std::mutex g_scratch_in_use[kMaxDevices];

// The address of `scratch` on `device`, which must be the current device.
static hipError_t static_scratch_of(int device, float **out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t err = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (err != hipSuccess) return err;
        g_scratch_ptr[device] = static_cast<float *>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

static thread_local std::string g_error;

static int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(RM_ERR_HIP, std::string(#expr) + " failed: " + hipGetErrorString(e__));         \
    } while (0)

struct HostObject {
    int type;
    float p[8];
    rm_material mat;
    float colour[3];
    int texture;
};

struct HostTexture {
    std::vector<float> texels;  // [h][w][4]
    int w, h;
    float uscale, vscale;
};

struct rm_scene {
    std::vector<HostObject> objects;
    std::vector<HostTexture> textures;
    float light[3] = {0.0f, 0.0f, 0.0f};
    float intensity = 1.0f;
    float camera[3] = {0.0f, 0.0f, 0.0f};
    float smooth_k = 0.0f;  // 0 = hard minimum (the reference's union)
};

constexpr int kStatSlots = 32;
constexpr int kBands = 8;  // row bands of a frame on its way to the host, at most
constexpr int kDefaultBands = 4;  // measured best at 3840x2160: more bands add launches and copies, fewer overlap less
constexpr size_t kBandedFramePixels = 4u << 20;  // smaller frames leave in one piece: their bands would each wait for a slow tile
constexpr int kGridMinPrimitives = 16;  // fewer bounded primitives than this: no spatial index
constexpr int kGridMaxDim = 48;         // cells per axis at most
constexpr int kGridMinList = 16;        // a cell lists at least this many primitives (its reach grows to admit them)
constexpr int kGridMaxList = 48;        // and at most this many (its reach shrinks)
constexpr int kGridShortList = 8;       // the short lists tried first: this many nearest primitives per cell

struct rm_engine {
    int device = 0;
    int W = 0, H = 0, M = 8;
    hipStream_t stream = nullptr;
    rm_scene scene;  // private copy
    float camera[3];
    Pose pose;
    int march_cap = 0;
    bool ssaa = false;  // four rays per pixel (engine.rs:95-108)
    bool tetra_normals = false;  // objects.rs:142-149
    bool refraction = false;     // engine.rs:196-228
    bool exact_occlusion = false;  // rm_engine_set_exact_occlusion
    bool shadow_proof = true;      // shadow rays that are proven open are not marched (shadow_stays_open)
    bool fast = false;             // rm_engine_set_precision: the frame kernel of the fast tier
    size_t grid_entries = 0;  // list entries of the spatial index (0 = none)

    // device-resident scene
    int n = 0, n_lm = 0, n_patches = 0;
    int tier = 2;  // -1: GroupedTierX, 0: GroupedTierA, 1: GroupedTierB, 2: GroupedLarge
    GroupedTierX tier_x;
    GroupedTierA tier_a;
    GroupedTierB tier_b;
    GroupedLarge large_scene;
    SceneView h_view;            // host copy of the device-resident SceneView
    SceneView *d_view = nullptr;
    bool view_dirty = true;
    std::vector<void *> d_group_arrays;
    std::vector<int> lm_slot;    // per object, -1 = none
    std::vector<int> patch_obj;  // per lightmap slot: engine index
    int *d_type = nullptr;
    Geom *d_geom = nullptr;
    ShadeRec *d_shade = nullptr;
    TextureRec *d_tex = nullptr;
    std::vector<float *> d_texels;
    float *d_lightmaps = nullptr;  // [n_patches][3]
    float *d_screen = nullptr;  // u of every column, then v of every row (engine.rs:79-82)
    // slow tiles in quarters (SplitPlan): the tiles' last lifetimes, and a ring of per-launch plans
    unsigned *d_tile_cost = nullptr;    // [frame tiles][4]
    unsigned *d_plan_words = nullptr;   // kPlanRing x (count padded to 64 words + list[kPlanCap])
    unsigned char *d_plan_flags = nullptr;  // kPlanRing x [frame tiles]
    static constexpr unsigned kPlans = 32;  // plans kept: one per (stream, set of tiles) that launches frames of this engine
    hipStream_t plan_stream[kPlans] = {};    // the stream each plan belongs to
    bool plan_taken[kPlans] = {};
    unsigned char plan_parity[kPlans] = {};  // which of a plan's two sets of count words its next launch uses
    unsigned long long plan_used[kPlans] = {};  // when the plan last served a launch (plan_clock)
    unsigned long long plan_clock = 0;
    // a plan serves the launches that follow it on its stream while they name the same tiles (see launch_render)
    struct PlanKey {
        unsigned grid_x = 0, per = 0, cap = 0;
        int row_begin = 0, row_end = 0, tile_row_stride = 0, blocks_x = 0, blocks_y = 0;
        bool operator==(const PlanKey &o) const {
            return grid_x == o.grid_x && per == o.per && cap == o.cap && row_begin == o.row_begin && row_end == o.row_end &&
                   tile_row_stride == o.tile_row_stride && blocks_x == o.blocks_x && blocks_y == o.blocks_y;
        }
    } plan_key[kPlans];
    unsigned plan_launches[kPlans] = {};       // launches the plan has served since the first
    unsigned *plan_count_word[kPlans] = {};    // the count word of the plan in force
    int plan_every = -1;                  // a fresh plan every so many launches (RM_PLAN_EVERY)
    int split_threshold = -1;           // ticks; 0 = off; -1 = not read yet (RM_SPLIT_US)
    bool split_on = true;               // rm_engine_set_tile_splitting
    unsigned dense_tiles = 0;           // launches of the small tiers from this many tiles on take the 8-wave build (0 = the default)
    unsigned split_times_mean = 3;      // ... and at least this many times the last launch's mean tile cost
    unsigned early_times_mean = 2;      // the same for the tiles a long launch hands out first (RM_EARLY_TIMES_MEAN)
    const unsigned *last_plan_count = nullptr;  // the count word of the most recent plan (device)
    unsigned last_plan_cap = 0;                 // and that plan's list length
    unsigned long long *d_stats = nullptr;  // kStatSlots words: RM_STAT_COUNT public ones, the rest for diagnostic builds
    unsigned char *d_frame = nullptr;  // W*H*4, for rm_engine_render
    unsigned char *h_stage = nullptr;  // pinned W*H*4 staging for callers that pass pageable memory

    // frames in flight for rm_engine_render_async: each slot renders and copies out on a stream of its
    // own, so that consecutive frames overlap (the long-running tiles of one beside the bulk of the next)
    unsigned char *d_async[RM_FRAMES_IN_FLIGHT] = {};
    hipStream_t slot_stream[RM_FRAMES_IN_FLIGHT] = {};
    hipEvent_t ev_main = nullptr;  // orders a slot after what the engine's own stream was given before
    hipEvent_t ev_copied[RM_FRAMES_IN_FLIGHT] = {};
    bool async_pending[RM_FRAMES_IN_FLIGHT] = {};
    // rm_engine_render: the frame leaves in row bands, copied out on `xfer_stream` while later bands render
    hipStream_t xfer_stream = nullptr;
    hipEvent_t ev_band_rendered[kBands] = {};
    hipEvent_t ev_band_copied[kBands] = {};
    int async_head = 0;  // slot the next frame goes to
    int async_tail = 0;  // oldest pending slot

    // radiosity work buffers
    int *d_patch_obj = nullptr;
    RadBuffers rad;  // allocated by rm_engine_radiosity_begin / _form_factors
    bool rad_begun = false;
    float *d_sample_pos = nullptr;  // [n_patches][3]: the point cloud of engine.rs:269-273, uploaded at creation
    bool sample_pos_ok = false;     // false: some lightmapped type has no sample positions
    hipStream_t clock_stream = nullptr;     // rm_engine_shader_clock: beside whatever the engine's streams run
    unsigned long long *d_clock = nullptr;  // its two counter differences
    float *scratch = nullptr;  // this device's copy of the library's static scratch array
    std::vector<hipStream_t> attached_streams;  // streams of rm_group frames in flight on this engine
    // the two-pass frame (an experiment, RM_TWO_PASS=1 when the engine is created)
    int two_pass = 0;            // 0 off; n > 0: pass 2 runs n waves (grid-stride over the queue)
    unsigned *d_queue_count = nullptr;
    float *d_queue = nullptr;
    size_t queue_capacity = 0;

    SceneView view() const {
        SceneView sv;
        sv.n = n;
        sv.type = d_type;
        sv.geom = d_geom;
        sv.shade = d_shade;
        sv.tex = d_tex;
        sv.lightmaps = d_lightmaps;
        sv.map_size = M;
        for (int i = 0; i < 3; i++) sv.light[i] = scene.light[i];
        sv.intensity = scene.intensity;
        // engine.rs:15 SKY_COLOUR = rgb![135, 206, 235]
        const float sky8[3] = {135.0f, 206.0f, 235.0f};
        for (int i = 0; i < 3; i++) sv.sky[i] = (sky8[i] / 255.0f) * (sky8[i] / 255.0f);
        sv.march_cap = march_cap;
        sv.tetra_normals = tetra_normals ? 1 : 0;
        sv.refraction = refraction ? 1 : 0;
        sv.exact_occlusion = exact_occlusion ? 1 : 0;
        sv.light_clear = std::numeric_limits<float>::infinity();
        for (const HostObject &o : scene.objects) {
            float at_light, rise;
            if (o.type == T_XPLANE || o.type == T_YPLANE || o.type == T_ZPLANE) {
                at_light = o.p[1] * (scene.light[o.type - T_XPLANE] - o.p[0]);  // objects.rs:275,295,315
                rise = std::fabs(o.p[1]);
            } else if (o.type == T_PLANE) {
                at_light = ((scene.light[0] * o.p[0] + scene.light[1] * o.p[1]) + scene.light[2] * o.p[2]) - o.p[3];
                rise = std::sqrt(o.p[0] * o.p[0] + o.p[1] * o.p[1] + o.p[2] * o.p[2]);
            } else {
                continue;
            }
            if (!(rise <= 1.001f) || !(at_light == at_light)) sv.light_clear = -std::numeric_limits<float>::infinity();
            sv.light_clear = std::min(sv.light_clear, at_light);
        }
        if (!shadow_proof) sv.light_clear = -std::numeric_limits<float>::infinity();
        sv.stats = d_stats;
        return sv;
    }
    RadArgs rad_args(const RadBuffers &b) const {
        RadArgs a;
        a.sv = d_view;
        a.n_patches = n_patches;
        a.patch_obj = d_patch_obj;
        a.pos = d_sample_pos;
        a.col = b.col;
        a.nrm = b.nrm;
        a.emissive = b.emissive;
        a.emissive_next = b.emissive_next;
        a.lightmaps = d_lightmaps;
        a.ft = b.ft;
        return a;
    }
};

// engine.rs:24-32 on the host (camera movement)
static void host_rot(const float v[3], float sx, float cx, float sy, float cy, float out[3]) {
    float x = v[0], y = v[1], z = v[2];
    float ny = y * cy - z * sy, nz = y * sy + z * cy;
    y = ny;
    z = nz;
    float nx = x * cx - z * sx;
    nz = x * sx + z * cx;
    out[0] = nx;
    out[1] = y;
    out[2] = nz;
}

// objects.rs:278-284, 298-304, 318-324, 345-352 (sin/cos stay on the host)
static bool host_sample_pos(const HostObject &o, int M, int u, int v, float out[3]) {
    const float MS = (float)M;
    const float a = ((float)u) - (MS / 2.0f) + 0.5f;
    const float b = ((float)v) - (MS / 2.0f) + 0.5f;
    switch (o.type) {
        case T_SPHERE: {
            const float PI = 3.14159265358979323846f;
            float theta = ((((float)u / MS) * 2.0f) - 1.0f) * PI;
            float phi = (((float)v / MS) - 0.5f) * PI;
            float nx = sinf(theta), ny = sinf(phi), nz = -cosf(theta);
            out[0] = o.p[0] + nx * o.p[3];
            out[1] = o.p[1] + ny * o.p[3];
            out[2] = o.p[2] + nz * o.p[3];
            return true;
        }
        case T_XPLANE: out[0] = o.p[0]; out[1] = a; out[2] = b; return true;
        case T_YPLANE: out[0] = a; out[1] = o.p[0]; out[2] = b; return true;
        case T_ZPLANE: out[0] = a; out[1] = b; out[2] = o.p[0]; return true;
        case T_PLANE: {
            // (origin + t * a) + bitangent * b, origin = normal * distance (see rm_scene_add_plane)
            const float *n = o.p, *t = o.p + 4, sgn = o.p[7];
            const float c[3] = {(n[1] * t[2] - n[2] * t[1]) * sgn, (n[2] * t[0] - n[0] * t[2]) * sgn, (n[0] * t[1] - n[1] * t[0]) * sgn};
            for (int k = 0; k < 3; k++) out[k] = (n[k] * o.p[3] + t[k] * a) + c[k] * b;
            return true;
        }
        default: return false;
    }
}

template <class T>
static hipError_t dev_alloc(T **p, size_t count) {
    return hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T));
}

static int engine_upload(rm_engine *e) {
    const rm_scene &s = e->scene;
    e->n = (int)s.objects.size();
    std::vector<int> type(e->n);
    std::vector<Geom> geom(e->n);
    std::vector<ShadeRec> shade(e->n);
    e->lm_slot.assign(e->n, -1);
    e->patch_obj.clear();
    // objects grouped by type for the march loops (original indices kept alongside)
    std::vector<int> gi_sph, gi_ax[3], gi_gp, gi_box;
    std::vector<float> g_sph, g_ax[3], g_gp, g_box;
    for (int i = 0; i < e->n; i++) {
        const HostObject &o = s.objects[i];
        type[i] = o.type;
        memcpy(geom[i].p, o.p, sizeof(float) * 8);
        ShadeRec r;
        memset(&r, 0, sizeof(r));
        r.ambient = o.mat.ambient;
        r.diffuse = o.mat.diffuse;
        r.specular = o.mat.specular;
        r.shininess = o.mat.shininess;
        r.reflectivity = o.mat.reflectivity;
        r.emissive = o.mat.emissive;
        r.cr = o.colour[0];
        r.cg = o.colour[1];
        r.cb = o.colour[2];
        r.type = o.type;
        r.texture = o.texture;
        r.textured = (o.type == T_YPLANE && !(o.p[0] > 0.0f)) ? 1 : 0;  // objects.rs:268
        if (o.type != T_BOX) {
            r.lm_slot = (int)e->patch_obj.size();
            e->lm_slot[i] = r.lm_slot;
            e->patch_obj.push_back(i);
        } else {
            r.lm_slot = -1;
        }
        shade[i] = r;
        switch (o.type) {
            case T_SPHERE:
                gi_sph.push_back(i);
                g_sph.insert(g_sph.end(), o.p, o.p + 4);
                break;
            case T_XPLANE:
            case T_YPLANE:
            case T_ZPLANE: {
                int a = o.type - T_XPLANE;
                gi_ax[a].push_back(i);
                g_ax[a].insert(g_ax[a].end(), o.p, o.p + 2);
                break;
            }
            case T_PLANE:
                gi_gp.push_back(i);
                g_gp.insert(g_gp.end(), o.p, o.p + 4);
                break;
            default:
                gi_box.push_back(i);
                g_box.insert(g_box.end(), o.p, o.p + 8);
                break;
        }
    }
    const int n_boxes = (int)gi_box.size();
    bool unit_dirs = true;  // every axis plane faces along +-1 (the march loop of the exact tier relies on it)
    for (int a = 0; a < 3; a++)
        for (size_t k = 0; k < gi_ax[a].size(); k++) unit_dirs = unit_dirs && fabsf(g_ax[a][2 * k + 1]) == 1.0f;
    auto fits = [&](int ns, int na, int ng) {
        return (int)gi_sph.size() <= ns && (int)gi_ax[0].size() <= na && (int)gi_ax[1].size() <= na &&
               (int)gi_ax[2].size() <= na && (int)gi_gp.size() <= ng;
    };
    float sph_extent = 0.0f;
    for (size_t k = 0; k < gi_sph.size(); k++) {
        const float *c = &g_sph[4 * k];
        sph_extent = fmaxf(sph_extent, fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fabsf(c[2])) + fabsf(c[3]));
    }
    auto fill = [&](auto &gs) {
        memset(&gs, 0, sizeof(gs));
        gs.sph_extent = sph_extent;
        gs.n_sph = (int)gi_sph.size();
        gs.n_x = (int)gi_ax[0].size();
        gs.n_y = (int)gi_ax[1].size();
        gs.n_z = (int)gi_ax[2].size();
        gs.n_gp = (int)gi_gp.size();
        for (size_t k = 0; k < gi_sph.size(); k++) {
            gs.idx_sph[k] = gi_sph[k];
            memcpy(gs.sph[k], &g_sph[4 * k], sizeof(float) * 4);
        }
        for (int a = 0; a < 3; a++)
            for (size_t k = 0; k < gi_ax[a].size(); k++) {
                gs.idx_ax[a][k] = gi_ax[a][k];
                memcpy(gs.ax[a][k], &g_ax[a][2 * k], sizeof(float) * 2);
            }
        for (size_t k = 0; k < gi_gp.size(); k++) {
            gs.idx_gp[k] = gi_gp[k];
            memcpy(gs.gp[k], &g_gp[4 * k], sizeof(float) * 4);
        }
    };
    memset(&e->tier_a, 0, sizeof(e->tier_a));
    memset(&e->tier_b, 0, sizeof(e->tier_b));
    memset(&e->tier_x, 0, sizeof(e->tier_x));
    if (n_boxes > 0 || s.smooth_k > 0.0f) {
        e->tier = 2;  // boxes and smooth unions live in the large layout only
    } else if (gi_sph.size() == 2 && gi_ax[0].size() == 2 && gi_ax[1].size() == 2 && gi_ax[2].size() == 2 && gi_gp.empty() &&
               unit_dirs) {
        e->tier = -1;
        fill(e->tier_x);
    } else if (fits(GroupedTierA::kSpheres, GroupedTierA::kAxis, GroupedTierA::kPlanes)) {
        e->tier = 0;
        fill(e->tier_a);
    } else if (fits(GroupedTierB::kSpheres, GroupedTierB::kAxis, GroupedTierB::kPlanes)) {
        e->tier = 1;
        fill(e->tier_b);
    } else {
        e->tier = 2;
    }
    {
        GroupedLarge &gl = e->large_scene;
        memset(&gl, 0, sizeof(gl));
        gl.n_sph = (int)gi_sph.size();
        gl.n_x = (int)gi_ax[0].size();
        gl.n_y = (int)gi_ax[1].size();
        gl.n_z = (int)gi_ax[2].size();
        gl.n_gp = (int)gi_gp.size();
        gl.n_box = n_boxes;
        gl.smooth_k = s.smooth_k;
        gl.sph_extent = sph_extent;
        auto up_i = [&](const std::vector<int> &v, const int **dst) -> hipError_t {
            int *d = nullptr;
            hipError_t err = dev_alloc(&d, v.size());
            if (err != hipSuccess) return err;
            e->d_group_arrays.push_back(d);
            *dst = d;
            return v.empty() ? hipSuccess : hipMemcpy(d, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice);
        };
        auto up_f = [&](const std::vector<float> &v, const float **dst) -> hipError_t {
            float *d = nullptr;
            hipError_t err = dev_alloc(&d, v.size());
            if (err != hipSuccess) return err;
            e->d_group_arrays.push_back(d);
            *dst = d;
            return v.empty() ? hipSuccess : hipMemcpy(d, v.data(), sizeof(float) * v.size(), hipMemcpyHostToDevice);
        };
        HIP_TRY(up_i(gi_sph, &gl.idx_sph));
        HIP_TRY(up_f(g_sph, &gl.sph));
        for (int a = 0; a < 3; a++) {
            HIP_TRY(up_i(gi_ax[a], &gl.idx_ax[a]));
            HIP_TRY(up_f(g_ax[a], &gl.ax[a]));
        }
        HIP_TRY(up_i(gi_gp, &gl.idx_gp));
        HIP_TRY(up_f(g_gp, &gl.gp));
        HIP_TRY(up_i(gi_box, &gl.idx_box));
        HIP_TRY(up_f(g_box, &gl.box));
        {
            // the planes once more as one array of records in the order of the fold
            std::vector<float> all;
            auto word = [](int v) {
                float f;
                memcpy(&f, &v, sizeof f);
                return f;
            };
            for (int a = 0; a < 3; a++)
                for (size_t k = 0; k < gi_ax[a].size(); k++) {
                    const float r[8] = {word(a), word(gi_ax[a][k]), g_ax[a][2 * k], g_ax[a][2 * k + 1], 0, 0, 0, 0};
                    all.insert(all.end(), r, r + 8);
                }
            for (size_t k = 0; k < gi_gp.size(); k++) {
                const float r[8] = {word(3),         word(gi_gp[k]),  g_gp[4 * k], g_gp[4 * k + 1],
                                    g_gp[4 * k + 2], g_gp[4 * k + 3], 0,           0};
                all.insert(all.end(), r, r + 8);
            }
            gl.n_planes_all = (int)(all.size() / 8);
            HIP_TRY(up_f(all, &gl.planes_all));
        }

        // Spatial index (see GridView) once the bounded primitives are many enough to pay for it.
        const int n_bounded = (int)gi_sph.size() + n_boxes;
        if (n_bounded >= kGridMinPrimitives) {
            // bounds of every primitive
            std::vector<float> lo(3 * n_bounded), hi(3 * n_bounded);
            float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int i = 0; i < n_bounded; i++) {
                const bool is_sphere = i < (int)gi_sph.size();
                const float *g = is_sphere ? &g_sph[4 * i] : &g_box[8 * (i - gi_sph.size())];
                for (int a = 0; a < 3; a++) {
                    const float ext = is_sphere ? fabsf(g[3]) : fabsf(g[3 + a]);
                    lo[3 * i + a] = g[a] - ext;
                    hi[3 * i + a] = g[a] + ext;
                    blo[a] = fminf(blo[a], lo[3 * i + a]);
                    bhi[a] = fmaxf(bhi[a], hi[3 * i + a]);
                }
            }
            bool finite = true;
            for (int a = 0; a < 3; a++) finite = finite && std::isfinite(blo[a]) && std::isfinite(bhi[a]);
            const double ex[3] = {(double)bhi[0] - blo[0], (double)bhi[1] - blo[1], (double)bhi[2] - blo[2]};
            const double longest = std::max(ex[0], std::max(ex[1], ex[2]));
            if (finite && longest > 0.0) {
                // Mean spacing of the primitives; cells a quarter of it, base reach equal to it plus the
                // blend width (the fastest of a sweep over both on the 250-primitive stress scenes: longer
                // lists cost more than the full walks that shorter ones cause; the reach was 0.85 of the
                // spacing until a wave could pass over a listed primitive with one compare).  A cell with fewer than
                // kGridMinList primitives within the base reach takes the reach that admits that many, so
                // that open space far from everything still gets the few nearest primitives and a large
                // reach instead of a full walk; kGridMaxList caps a list by lowering the cell's reach.
                const double vol = std::max(ex[0], 0.05 * longest) * std::max(ex[1], 0.05 * longest) * std::max(ex[2], 0.05 * longest);
                const double spacing = std::cbrt(vol / n_bounded);
                // (the environment overrides are tuning aids; values outside the ranges the lists and the
                // 16-bit counts of a cell record can hold are brought inside them)
                const auto env_mul = [](const char *name, double dflt) {
                    const char *v = getenv(name);
                    const double x = v ? atof(v) : dflt;
                    return (x >= 0.01 && x <= 100.0) ? x : dflt;  // also refuses NaN
                };
                const auto env_count = [&](const char *name, int dflt) {
                    const char *v = getenv(name);
                    return std::max(1, std::min({v ? atoi(v) : dflt, n_bounded, 65535}));
                };
                const double reach_mul = env_mul("RM_GRID_REACH", 1.0);
                const double cell_mul = env_mul("RM_GRID_CELL", 0.25);
                const int min_list = env_count("RM_GRID_MINLIST", kGridMinList);
                const int max_list = std::max(min_list, env_count("RM_GRID_MAXLIST", kGridMaxList));
                const int short_list = env_count("RM_GRID_SHORTLIST", kGridShortList);
                const float reach = (float)(reach_mul * spacing + 2.0 * s.smooth_k);
                // the grid spans the primitives and half their extent around them; beyond it every
                // primitive is at least `apron` away
                const float apron = fmaxf(reach, (float)(0.5 * longest));
                float cell = (float)(cell_mul * spacing);
                cell = fmaxf(cell, (float)((longest + 2.0 * apron) / kGridMaxDim));
                const float scale = fmaxf(fmaxf(fabsf(blo[0]), fabsf(bhi[0])), fmaxf(fmaxf(fabsf(blo[1]), fabsf(bhi[1])), fmaxf(fabsf(blo[2]), fabsf(bhi[2])))) + apron;
                const float margin = 1e-3f + 1e-5f * scale;  // rounding of the cell lookup and of the distances
                GridView gv;
                memset(&gv, 0, sizeof(gv));
                gv.ox = blo[0] - apron - margin;
                gv.oy = blo[1] - apron - margin;
                gv.oz = blo[2] - apron - margin;
                gv.inv_cell = 1.0f / cell;
                gv.reach = apron;
                gv.nx = std::max(1, (int)std::ceil((bhi[0] + apron + margin - gv.ox) / cell));
                gv.ny = std::max(1, (int)std::ceil((bhi[1] + apron + margin - gv.oy) / cell));
                gv.nz = std::max(1, (int)std::ceil((bhi[2] + apron + margin - gv.oz) / cell));
                gv.cells = gv.nx * gv.ny * gv.nz;
                // per level: base reach, fewest and most primitives a cell lists.  Level 0 holds the few
                // nearest primitives of every cell and nothing else; level 1 everything within `reach`.
                struct LevelPlan {
                    float base;
                    int min_list, max_list;
                } plans[kGridLevels] = {{0.0f, short_list, short_list}, {reach, min_list, max_list}};
                struct LevelData {
                    std::vector<float> cell_reach;
                } lv[kGridLevels];
                for (auto &l : lv) l.cell_reach.assign(gv.cells, 0.0f);
                const float origin[3] = {gv.ox, gv.oy, gv.oz};
                // phase 1, host threads over slabs of cells: every cell's reach and member list per level
                std::vector<std::vector<uint16_t>> members[kGridLevels];
                std::vector<std::vector<float>> member_lb[kGridLevels];  // each member's distance bound over the cell
                for (auto &m : members) m.resize(gv.cells);
                for (auto &m : member_lb) m.resize(gv.cells);
                auto slab = [&](int z0, int z1) {
                    std::vector<float> lb(n_bounded), sorted(n_bounded);
                    for (int cz = z0; cz < z1; cz++)
                        for (int cy = 0; cy < gv.ny; cy++)
                            for (int cx = 0; cx < gv.nx; cx++) {
                                const int c = (cz * gv.ny + cy) * gv.nx + cx;
                                const int ci[3] = {cx, cy, cz};
                                float clo[3], chi[3];
                                for (int a = 0; a < 3; a++) {
                                    // the device computes the cell as floor((p - origin) * inv_cell): widen by the margin
                                    clo[a] = origin[a] + (float)ci[a] * cell - margin;
                                    chi[a] = origin[a] + (float)(ci[a] + 1) * cell + margin;
                                }
                                // lower bound of every primitive's distance over the cell
                                for (int i = 0; i < n_bounded; i++) {
                                    const bool is_sphere = i < (int)gi_sph.size();
                                    double d2 = 0.0;
                                    if (is_sphere) {
                                        const float *g = &g_sph[4 * i];
                                        for (int a = 0; a < 3; a++) {
                                            const double t = std::max(0.0, std::max((double)clo[a] - g[a], (double)g[a] - chi[a]));
                                            d2 += t * t;
                                        }
                                        lb[i] = (float)(std::sqrt(d2) - fabs((double)g[3])) - margin;
                                    } else {
                                        for (int a = 0; a < 3; a++) {
                                            const double t = std::max(0.0, std::max((double)clo[a] - hi[3 * i + a], (double)lo[3 * i + a] - chi[a]));
                                            d2 += t * t;
                                        }
                                        lb[i] = (float)std::sqrt(d2) - margin;
                                    }
                                }
                                sorted = lb;
                                std::sort(sorted.begin(), sorted.end());
                                for (int L = 0; L < kGridLevels; L++) {
                                    // the cell's reach: every primitive with lb < r is listed, every other is >= r away
                                    float r = plans[L].base;
                                    const int want = std::min(plans[L].min_list, n_bounded);
                                    if (want > 0 && !(sorted[want - 1] < r))
                                        r = want < n_bounded ? std::nextafter(sorted[want - 1], INFINITY) : INFINITY;
                                    if (n_bounded > plans[L].max_list && sorted[plans[L].max_list] < r) r = sorted[plans[L].max_list];
                                    lv[L].cell_reach[c] = r;
                                    for (int i = 0; i < n_bounded; i++)
                                        if (lb[i] < r) {
                                            members[L][c].push_back((uint16_t)i);
                                            member_lb[L][c].push_back(lb[i]);
                                        }
                                }
                            }
                };
                {
                    const int n_threads = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, gv.nz}));
                    std::vector<std::thread> pool;
                    for (int t = 0; t < n_threads; t++)
                        pool.emplace_back(slab, (int)((long long)gv.nz * t / n_threads), (int)((long long)gv.nz * (t + 1) / n_threads));
                    for (auto &th : pool) th.join();
                }
                // phase 2: the pool (see GridView): both cell tables, then every cell's blocks
                std::vector<int> pool(4 * (size_t)gv.cells * kGridLevels, 0);
                const auto word = [](float f) {
                    int v;
                    memcpy(&v, &f, sizeof v);
                    return v;
                };
                e->grid_entries = 0;
                for (int L = 0; L < kGridLevels; L++) {
                    size_t entries = 0;
                    int longest_list = 0;
                    for (int c = 0; c < gv.cells; c++) {
                        std::vector<int> sph, box;
                        std::vector<float> sph_lb, box_lb;
                        for (size_t m = 0; m < members[L][c].size(); m++) {
                            const uint16_t i = members[L][c][m];
                            (i < gi_sph.size() ? sph : box).push_back(i);
                            (i < gi_sph.size() ? sph_lb : box_lb).push_back(member_lb[L][c][m]);
                        }
                        const size_t s_at = pool.size();
                        for (size_t g0 = 0; g0 < sph.size(); g0 += 4) {
                            int group[kGridSphereGroup];
                            for (int u = 0; u < 4; u++) {
                                const bool there = g0 + u < sph.size();
                                for (int w = 0; w < 4; w++) group[4 * u + w] = there ? word(g_sph[4 * sph[g0 + u] + w]) : 0;
                                group[16 + u] = there ? gi_sph[sph[g0 + u]] : -1;
                                group[20 + u] = word(there ? sph_lb[g0 + u] : INFINITY);
                            }
                            pool.insert(pool.end(), group, group + kGridSphereGroup);
                        }
                        const size_t b_at = pool.size();
                        for (size_t m = 0; m < box.size(); m++) {
                            const int b = box[m] - (int)gi_sph.size();
                            int rec[8];
                            for (int w = 0; w < 6; w++) rec[w] = word(g_box[8 * b + w]);
                            rec[6] = gi_box[b];
                            rec[7] = word(box_lb[m]);
                            pool.insert(pool.end(), rec, rec + 8);
                        }
                        int *rec = &pool[4 * ((size_t)L * gv.cells + c)];
                        rec[0] = (int)s_at;
                        rec[1] = (int)b_at;
                        rec[2] = (int)sph.size() | ((int)box.size() << 16);
                        rec[3] = word(lv[L].cell_reach[c]);
                        entries += sph.size() + box.size();
                        longest_list = std::max(longest_list, (int)(sph.size() + box.size()));
                    }
                    e->grid_entries += entries;
                    if (getenv("RM_VERBOSE"))
                        fprintf(stderr, "rm: spatial index level %d: %d x %d x %d cells of %.3f, base reach %.3f, apron %.3f, %zu entries (%.1f per cell, longest list %d) for %d primitives\n",
                                L, gv.nx, gv.ny, gv.nz, cell, plans[L].base, apron, entries, (double)entries / gv.cells, longest_list, n_bounded);
                }
                pool.resize(pool.size() + 24, 0);  // a wave-uniform walk fetches whole groups and pairs
                if (pool.size() > (size_t)INT32_MAX) return fail(RM_ERR_INVALID, "the spatial index of this scene would not fit 32-bit offsets");
                HIP_TRY(up_i(pool, &gv.pool));
                gl.grid = gv;
            }
        }
    }
    e->n_lm = (int)e->patch_obj.size();
    e->n_patches = e->n_lm * e->M * e->M;

    const int n_pad = e->n;
    HIP_TRY(dev_alloc(&e->d_type, n_pad));
    HIP_TRY(dev_alloc(&e->d_geom, n_pad));
    HIP_TRY(dev_alloc(&e->d_shade, e->n));
    HIP_TRY(hipMemcpy(e->d_type, type.data(), sizeof(int) * n_pad, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_geom, geom.data(), sizeof(Geom) * n_pad, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_shade, shade.data(), sizeof(ShadeRec) * e->n, hipMemcpyHostToDevice));

    std::vector<TextureRec> tex(s.textures.size());
    e->d_texels.assign(s.textures.size(), nullptr);
    for (size_t t = 0; t < s.textures.size(); t++) {
        const HostTexture &ht = s.textures[t];
        HIP_TRY(dev_alloc(&e->d_texels[t], ht.texels.size()));
        HIP_TRY(hipMemcpy(e->d_texels[t], ht.texels.data(), sizeof(float) * ht.texels.size(), hipMemcpyHostToDevice));
        const bool pow2 = (ht.w & (ht.w - 1)) == 0 && (ht.h & (ht.h - 1)) == 0;
        tex[t] = TextureRec{e->d_texels[t], ht.w, ht.h, ht.uscale, ht.vscale, pow2 ? 1.0f / (float)ht.w : 0.0f, pow2 ? 1.0f / (float)ht.h : 0.0f};
    }
    HIP_TRY(dev_alloc(&e->d_tex, tex.size()));
    if (!tex.empty()) HIP_TRY(hipMemcpy(e->d_tex, tex.data(), sizeof(TextureRec) * tex.size(), hipMemcpyHostToDevice));

    HIP_TRY(dev_alloc(&e->d_lightmaps, (size_t)e->n_patches * 3));
    HIP_TRY(hipMemset(e->d_lightmaps, 0, std::max<size_t>((size_t)e->n_patches * 3, 1) * sizeof(float)));
    HIP_TRY(dev_alloc(&e->d_view, 1));
    e->view_dirty = true;
    {
        // engine.rs:79-82: y = HEIGHT - y_inv; u = 2 (x - W/2) / H; v = 2 (y - H/2) / H, in f32 as written
        std::vector<float> screen((size_t)e->W + e->H);
        for (int x = 0; x < e->W; x++) screen[x] = 2.0f * ((float)x - (0.5f * (float)e->W)) / (float)e->H;
        for (int y_inv = 0; y_inv < e->H; y_inv++)
            screen[(size_t)e->W + y_inv] = 2.0f * ((float)(e->H - y_inv) - (0.5f * (float)e->H)) / (float)e->H;
        HIP_TRY(dev_alloc(&e->d_screen, screen.size()));
        HIP_TRY(hipMemcpy(e->d_screen, screen.data(), sizeof(float) * screen.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY(dev_alloc(&e->d_stats, kStatSlots));
    HIP_TRY(hipMemset(e->d_stats, 0, sizeof(unsigned long long) * kStatSlots));
    HIP_TRY(dev_alloc(&e->d_patch_obj, e->n_lm));
    if (e->n_lm)
        HIP_TRY(hipMemcpy(e->d_patch_obj, e->patch_obj.data(), sizeof(int) * e->n_lm, hipMemcpyHostToDevice));
    {
        // engine.rs:269-273: the world position of every sample point (sin/cos stay on the host)
        const int M = e->M;
        std::vector<float> pos((size_t)e->n_patches * 3);
        bool ok = true;
        for (int ci = 0; ci < e->n_lm && ok; ci++) {
            const HostObject &o = s.objects[e->patch_obj[ci]];
            for (int x = 0; x < M && ok; x++)
                for (int y = 0; y < M && ok; y++) ok = host_sample_pos(o, M, x, y, &pos[((size_t)(ci * M + x) * M + y) * 3]);
        }
        e->sample_pos_ok = ok;
        HIP_TRY(dev_alloc(&e->d_sample_pos, pos.size()));
        if (ok && !pos.empty()) HIP_TRY(hipMemcpy(e->d_sample_pos, pos.data(), sizeof(float) * pos.size(), hipMemcpyHostToDevice));
    }
    return RM_OK;
}

static void engine_free(rm_engine *e) {
    (void)hipSetDevice(e->device);
    (void)hipFree(e->d_type);
    (void)hipFree(e->d_geom);
    (void)hipFree(e->d_shade);
    (void)hipFree(e->d_tex);
    for (float *p : e->d_texels) (void)hipFree(p);
    (void)hipFree(e->d_lightmaps);
    (void)hipFree(e->d_stats);
    (void)hipFree(e->d_screen);
    (void)hipFree(e->d_tile_cost);
    (void)hipFree(e->d_plan_words);
    (void)hipFree(e->d_plan_flags);
    (void)hipFree(e->d_view);
    (void)hipFree(e->d_frame);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    for (int i = 0; i < RM_FRAMES_IN_FLIGHT; i++) {
        (void)hipFree(e->d_async[i]);
        if (e->ev_copied[i]) (void)hipEventDestroy(e->ev_copied[i]);
        if (e->slot_stream[i]) (void)hipStreamDestroy(e->slot_stream[i]);
    }
    if (e->ev_main) (void)hipEventDestroy(e->ev_main);
    for (int b = 0; b < kBands; b++) {
        if (e->ev_band_rendered[b]) (void)hipEventDestroy(e->ev_band_rendered[b]);
        if (e->ev_band_copied[b]) (void)hipEventDestroy(e->ev_band_copied[b]);
    }
    if (e->xfer_stream) (void)hipStreamDestroy(e->xfer_stream);
    if (e->clock_stream) (void)hipStreamDestroy(e->clock_stream);
    (void)hipFree(e->d_clock);
    (void)hipFree(e->d_patch_obj);
    (void)hipFree(e->d_queue);
    (void)hipFree(e->d_queue_count);
    (void)hipFree(e->d_sample_pos);
    (void)hipFree(e->rad.col);
    (void)hipFree(e->rad.nrm);
    (void)hipFree(e->rad.emissive);
    (void)hipFree(e->rad.emissive_next);
    (void)hipFree(e->rad.ft);
    for (void *p : e->d_group_arrays) (void)hipFree(p);
}

// Launches KERNEL<G>(args..., grouped scene) for the engine's scene tier.
// -DRM_EXP_LDS: a diagnostic build whose kernels of the small exact-fill tier are launched with
// RM_EXP_LDS (environment) bytes of unused LDS per workgroup, which caps the waves a compute unit
// holds (tools/occupancy_probe.py).
#ifdef RM_EXP_LDS
static size_t exp_lds_bytes() {
    static const size_t n = getenv("RM_EXP_LDS") ? (size_t)atoi(getenv("RM_EXP_LDS")) : 0;
    return n;
}
#else
static constexpr size_t exp_lds_bytes() { return 0; }
#endif
#define RM_LAUNCH(KERNEL, e, grid, block, ...)                                                                \
    do {                                                                                                      \
        if ((e)->tier == -1)                                                                                  \
            hipLaunchKernelGGL((KERNEL<GroupedTierX>), grid, block, exp_lds_bytes(), (e)->stream, __VA_ARGS__, (e)->tier_x); \
        else if ((e)->tier == 0)                                                                              \
            hipLaunchKernelGGL((KERNEL<GroupedTierA>), grid, block, 0, (e)->stream, __VA_ARGS__, (e)->tier_a); \
        else if ((e)->tier == 1)                                                                              \
            hipLaunchKernelGGL((KERNEL<GroupedTierB>), grid, block, 0, (e)->stream, __VA_ARGS__, (e)->tier_b); \
        else                                                                                                  \
            hipLaunchKernelGGL((KERNEL<GroupedLarge>), grid, block, 0, (e)->stream, __VA_ARGS__,              \
                               (e)->large_scene);                                                             \
    } while (0)

// Waits for the frames of rm_engine_render_async that are still rendering or on their way to the
// host (they stay pending for rm_engine_render_wait): what they read - the SceneView, the lightmaps -
// is about to change.
static int quiesce_frames_in_flight(rm_engine *e) {
    for (int i = 0; i < RM_FRAMES_IN_FLIGHT; i++)
        if (e->async_pending[i] && e->slot_stream[i]) HIP_TRY(hipStreamSynchronize(e->slot_stream[i]));
    for (hipStream_t s : e->attached_streams) HIP_TRY(hipStreamSynchronize(s));
    return RM_OK;
}

// Brings the device copy of the SceneView up to date (stream ordered).  There is one copy for all
// frames: a change of settings first lets the frames in flight finish with the old one.
static int sync_view(rm_engine *e) {
    if (!e->view_dirty) return RM_OK;
    if (int rc_ = quiesce_frames_in_flight(e)) return rc_;
    e->h_view = e->view();
    HIP_TRY(hipMemcpyAsync(e->d_view, &e->h_view, sizeof(SceneView), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // h_view may change before an async copy would read it
    e->view_dirty = false;
    return RM_OK;
}

// The first distance evaluation of every primary march, in the operations of the march loop
// (objects.rs:261,275,295,315,334; the box and the smooth minimum as rm_device.hpp defines them):
// see RenderArgs.  With a smooth union the distances are folded in the march loop's group order -
// X, Y, Z planes, generic planes, spheres, boxes, by index inside a group.
static float host_smooth_min(float a, float b, float k) {
    const float gap = fabsf(a - b);
    if (gap >= k) return fminf(a, b);
    float h = fmaxf(k - gap, 0.0f) / k;
    return fminf(a, b) - h * h * k * (1.0f / 4.0f);
}
static float host_box_sdf(const float *c, const float *h, const float *p) {
    const float qx = fabsf(p[0] - c[0]) - h[0], qy = fabsf(p[1] - c[1]) - h[1], qz = fabsf(p[2] - c[2]) - h[2];
    const float rx = fmaxf(qx, 0.0f), ry = fmaxf(qy, 0.0f), rz = fmaxf(qz, 0.0f);
    return sqrtf((rx * rx + ry * ry) + rz * rz) + fminf(fmaxf(qx, fmaxf(qy, qz)), 0.0f);
}
static void camera_distances(const rm_engine *e, float &nearest, float &nearest_sphere) {
    nearest = NAN;
    nearest_sphere = INFINITY;
    const float *p = e->pose.cam;
    const float k = e->scene.smooth_k;
    float d = INFINITY;
    static const int group_order[6] = {T_XPLANE, T_YPLANE, T_ZPLANE, T_PLANE, T_SPHERE, T_BOX};
    for (int group : group_order)
        for (const HostObject &o : e->scene.objects) {
            if (o.type != group) continue;
            float s;
            switch (o.type) {
                case T_SPHERE: {
                    const float x = p[0] - o.p[0], y = p[1] - o.p[1], z = p[2] - o.p[2];
                    s = sqrtf((x * x + y * y) + z * z) - o.p[3];
                    nearest_sphere = fminf(nearest_sphere, s);
                    break;
                }
                case T_XPLANE: s = o.p[1] * (p[0] - o.p[0]); break;
                case T_YPLANE: s = o.p[1] * (p[1] - o.p[0]); break;
                case T_ZPLANE: s = o.p[1] * (p[2] - o.p[0]); break;
                case T_PLANE: s = ((p[0] * o.p[0] + p[1] * o.p[1]) + p[2] * o.p[2]) - o.p[3]; break;
                default: s = host_box_sdf(o.p, o.p + 3, p); break;
            }
            if (!(s == s)) return;
            d = k > 0.0f ? host_smooth_min(d, s, k) : fminf(d, s);
        }
    if (d >= SMALL_DISTANCE && d < MAX_MARCH_DISTANCE) nearest = d;
}

static RenderArgs make_render_args(rm_engine *e, void *d_out, size_t pitch, int row_begin, int row_end, dim3 &grid) {
    RenderArgs a;
    a.sv = e->d_view;
    a.pose = e->pose;
    a.row_begin = row_begin;
    a.row_end = row_end;
    a.blocks_x = (e->W + TILE - 1) / TILE;
    a.blocks_y = (row_end - row_begin + TILE - 1) / TILE;
    a.tile_row_stride = 1;
    a.out = static_cast<unsigned char *>(d_out);
    a.pitch = pitch;
    camera_distances(e, a.camera_dist, a.camera_sph);
    a.screen_u = e->d_screen;
    a.screen_v = e->d_screen + e->W;
    a.plan_count = nullptr;
    a.plan_list = nullptr;
    a.plan_flag = nullptr;
    a.tile_cost = nullptr;
    a.plan_cap = 0;
    a.plan_per = 4;
    a.cost_tiles = 0;
    grid = dim3((unsigned)(a.blocks_x * a.blocks_y), 1, 1);
    return a;
}

// ===========================================================================
// C ABI
// ===========================================================================

extern "C" {

const char *rm_last_error(void) { return g_error.c_str(); }
int rm_version(void) { return 1; }

void *rm_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail(RM_ERR_HIP, "hipHostMalloc failed");
        return nullptr;
    }
    return p;
}
void rm_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int rm_device_count(int *count) {
    if (!count) return fail(RM_ERR_INVALID, "count is NULL");
    HIP_TRY(hipGetDeviceCount(count));
    return RM_OK;
}

rm_scene *rm_scene_create(void) { return new rm_scene(); }
void rm_scene_destroy(rm_scene *scene) { delete scene; }

int rm_texture_create_rgb8(rm_scene *scene, const uint8_t *rgb, int width, int height, float uscale, float vscale) {
    if (!scene || !rgb || width <= 0 || height <= 0) return -fail(RM_ERR_INVALID, "bad texture arguments");
    HostTexture t;
    t.w = width;
    t.h = height;
    t.uscale = uscale;
    t.vscale = vscale;
    t.texels.resize((size_t)width * height * 4);
    for (size_t i = 0; i < (size_t)width * height; i++) {
        for (int c = 0; c < 3; c++) {
            float f = (float)rgb[3 * i + c];
            t.texels[4 * i + c] = (f / 255.0f) * (f / 255.0f);  // colour.rs:7-15 via texture.rs:18-21
        }
        t.texels[4 * i + 3] = 0.0f;
    }
    scene->textures.push_back(std::move(t));
    return (int)scene->textures.size() - 1;
}

static int add_object(rm_scene *scene, int type, const float p[8], const rm_material *m, const float colour[3],
                      int texture) {
    if (!scene || !m || !colour) return -fail(RM_ERR_INVALID, "NULL argument");
    if (scene->objects.size() >= RM_MAX_OBJECTS) return -fail(RM_ERR_LIMIT, "too many objects");
    HostObject o;
    o.type = type;
    memcpy(o.p, p, sizeof(float) * 8);
    o.mat = *m;
    memcpy(o.colour, colour, sizeof(float) * 3);
    o.texture = texture;
    scene->objects.push_back(o);
    return (int)scene->objects.size() - 1;
}

int rm_scene_add_sphere(rm_scene *scene, const float position[3], float radius, const rm_material *material,
                        const float colour[3]) {
    if (!position) return -fail(RM_ERR_INVALID, "NULL position");
    float p[8] = {position[0], position[1], position[2], radius};
    return add_object(scene, T_SPHERE, p, material, colour, -1);
}
int rm_scene_add_xplane(rm_scene *scene, float x, float dir, const rm_material *material, const float colour[3]) {
    float p[8] = {x, dir};
    return add_object(scene, T_XPLANE, p, material, colour, -1);
}
int rm_scene_add_yplane(rm_scene *scene, float y, float dir, const rm_material *material, const float colour[3],
                        int texture) {
    if (!scene || texture < 0 || texture >= (int)scene->textures.size())
        return -fail(RM_ERR_INVALID, "YPlane::new takes a texture (objects.rs:236)");
    float p[8] = {y, dir};
    return add_object(scene, T_YPLANE, p, material, colour, texture);
}
int rm_scene_add_zplane(rm_scene *scene, float z, float dir, const rm_material *material, const float colour[3]) {
    float p[8] = {z, dir};
    return add_object(scene, T_ZPLANE, p, material, colour, -1);
}
int rm_scene_add_plane(rm_scene *scene, const float normal[3], float distance, const rm_material *material,
                       const float colour[3]) {
    if (!normal) return -fail(RM_ERR_INVALID, "NULL normal");
    float p[8] = {normal[0], normal[1], normal[2], distance};
    // The lightmap of a Plane (the reference gives it one, objects.rs:216, and no way to address it,
    // objects.rs:39,42) lies over two tangent directions, chosen so that an axis normal reproduces the
    // layout of the matching XPlane / YPlane / ZPlane: a = axis of the largest |normal component|,
    // e1 < e2 the other two; t = normalized(unit(e1) - normal * normal[e1]); b = normal x t, negated if
    // its e2 component is negative.  t and that sign travel with the geometry (p[4..7]).
    const float *n = normal;
    int a = 0;
    if (fabsf(n[1]) > fabsf(n[a])) a = 1;
    if (fabsf(n[2]) > fabsf(n[a])) a = 2;
    const int e1 = a == 0 ? 1 : 0, e2 = a == 2 ? 1 : 2;
    const float d1 = n[e1];
    const float raw[3] = {(e1 == 0 ? 1.0f : 0.0f) - n[0] * d1, (e1 == 1 ? 1.0f : 0.0f) - n[1] * d1, (e1 == 2 ? 1.0f : 0.0f) - n[2] * d1};
    const float inv = 1.0f / sqrtf((raw[0] * raw[0] + raw[1] * raw[1]) + raw[2] * raw[2]);  // vector.rs:27-30
    const float t[3] = {raw[0] * inv, raw[1] * inv, raw[2] * inv};
    const float c[3] = {n[1] * t[2] - n[2] * t[1], n[2] * t[0] - n[0] * t[2], n[0] * t[1] - n[1] * t[0]};
    p[4] = t[0];
    p[5] = t[1];
    p[6] = t[2];
    p[7] = c[e2] < 0.0f ? -1.0f : 1.0f;
    return add_object(scene, T_PLANE, p, material, colour, -1);
}
int rm_scene_add_box(rm_scene *scene, const float position[3], const float half_extent[3], const rm_material *material,
                     const float colour[3]) {
    if (!position || !half_extent) return -fail(RM_ERR_INVALID, "NULL argument");
    float p[8] = {position[0], position[1], position[2], half_extent[0], half_extent[1], half_extent[2]};
    return add_object(scene, T_BOX, p, material, colour, -1);
}
int rm_scene_set_smooth_union(rm_scene *scene, float k) {
    if (!scene || !(k >= 0.0f)) return fail(RM_ERR_INVALID, "smooth union width must be >= 0");
    scene->smooth_k = k;
    return RM_OK;
}
int rm_scene_set_light(rm_scene *scene, const float position[3], float intensity) {
    if (!scene || !position) return fail(RM_ERR_INVALID, "NULL argument");
    memcpy(scene->light, position, sizeof(float) * 3);
    scene->intensity = intensity;
    return RM_OK;
}
int rm_scene_set_camera(rm_scene *scene, const float position[3]) {
    if (!scene || !position) return fail(RM_ERR_INVALID, "NULL argument");
    memcpy(scene->camera, position, sizeof(float) * 3);
    return RM_OK;
}

rm_engine *rm_engine_create(const rm_scene *scene, int width, int height, int map_size, int device) {
    if (!scene || width <= 0 || height <= 0 || map_size <= 0 || map_size > 256) {
        fail(RM_ERR_INVALID, "bad engine arguments");
        return nullptr;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        fail(RM_ERR_HIP, "no HIP device is visible: the ray-marching kernels need an MI355X");
        return nullptr;
    }
    if (device < 0 || device >= count) {
        fail(RM_ERR_INVALID, "device index out of range");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        fail(RM_ERR_HIP, "hipSetDevice failed");
        return nullptr;
    }
    rm_engine *e = new rm_engine();
    e->device = device;
    e->W = width;
    e->H = height;
    e->M = map_size;
    e->scene = *scene;
    memcpy(e->camera, scene->camera, sizeof(float) * 3);
    if (engine_upload(e) != RM_OK) {
        engine_free(e);
        delete e;
        return nullptr;
    }
    if (hipError_t err = static_scratch_of(device, &e->scratch); err != hipSuccess) {
        fail(RM_ERR_HIP, std::string("hipGetSymbolAddress(scratch) failed: ") + hipGetErrorString(err));
        engine_free(e);
        delete e;
        return nullptr;
    }
    if (const char *v = getenv("RM_SHADOW_PROOF")) e->shadow_proof = atoi(v) != 0;  // (0: every wanted shadow ray is marched; for measurements and tests)
    if (const char *v = getenv("RM_TWO_PASS")) e->two_pass = std::max(0, std::min(1 << 20, atoi(v) == 1 ? 8192 : atoi(v)));  // experiment switch
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    rm_engine_input(e, 0, 0, zero, 0);
    return e;
}

void rm_engine_destroy(rm_engine *engine) {
    if (!engine) return;
    engine_free(engine);
    delete engine;
}

int rm_engine_set_stream(rm_engine *e, void *hip_stream) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    e->stream = static_cast<hipStream_t>(hip_stream);
    return RM_OK;
}

int rm_engine_input(rm_engine *e, int32_t mouse_x, int32_t mouse_y, const float rel_move[3], int32_t scroll) {
    if (!e || !rel_move) return fail(RM_ERR_INVALID, "NULL argument");
    Pose &p = e->pose;
    p.exposure = 1.0f;                               // engine.rs:63
    float theta_x = (float)(-mouse_x) / 600.0f;      // engine.rs:65
    float theta_y = (float)mouse_y / 600.0f;         // engine.rs:66
    p.sx = sinf(theta_x);
    p.cx = cosf(theta_x);
    p.sy = sinf(theta_y);
    p.cy = cosf(theta_y);
    float moved[3];
    host_rot(rel_move, p.sx, p.cx, p.sy, p.cy, moved);  // engine.rs:72
    for (int i = 0; i < 3; i++) e->camera[i] = e->camera[i] + moved[i];
    float fov_deg = 90.0f - ((float)scroll * 5.0f);  // engine.rs:74
    const float PI = 3.14159265358979323846f;
    float rad = (fov_deg * 0.5f) * (PI / 180.0f);
    p.zdepth = 1.0f / tanf(rad);  // engine.rs:75
    for (int i = 0; i < 3; i++) p.cam[i] = e->camera[i];
    p.W = e->W;
    p.H = e->H;
    return RM_OK;
}

int rm_engine_get_camera(rm_engine *e, float position[3]) {
    if (!e || !position) return fail(RM_ERR_INVALID, "NULL argument");
    memcpy(position, e->camera, sizeof(float) * 3);
    return RM_OK;
}
int rm_engine_set_camera(rm_engine *e, const float position[3]) {
    if (!e || !position) return fail(RM_ERR_INVALID, "NULL argument");
    memcpy(e->camera, position, sizeof(float) * 3);
    memcpy(e->pose.cam, position, sizeof(float) * 3);
    return RM_OK;
}

constexpr unsigned kPlanRing = rm_engine::kPlans;  // plans kept per engine
constexpr unsigned kPlanCap = 1024;    // tiles listed for quarters per launch at most
constexpr unsigned kPlanCapEarly = 65536;  // ... listed to go first in a long launch
constexpr unsigned kPlanStride = 64 + kPlanCapEarly;  // words per plan: the count in a line of its own, then the list

// The frame kernel (`dense`: its build for 8 waves per SIMD) in the engine's precision.
static void launch_frame_kernel(rm_engine *e, bool dense, dim3 grid, const RenderArgs &a) {
    if (e->fast) {
        const void *scene = e->tier == -1 ? (const void *)&e->tier_x : e->tier == 0 ? (const void *)&e->tier_a : e->tier == 1 ? (const void *)&e->tier_b : (const void *)&e->large_scene;
        rm_fast_launch(e->tier, scene, dense, grid, e->stream, a);
    } else if (dense) {
        RM_LAUNCH(render_dense_kernel, e, grid, dim3(BLOCK_THREADS), a);
    } else {
        RM_LAUNCH(render_kernel, e, grid, dim3(BLOCK_THREADS), a);
    }
}

static int launch_render(rm_engine *e, dim3 grid, const RenderArgs &a0) {
    const bool long_launch = grid.x >= kDenseLaunchTiles;
    if (e->dense_tiles == 0) {
        const char *v = getenv("RM_DENSE_TILES");  // (for measurements)
        e->dense_tiles = v && atoi(v) > 0 ? (unsigned)atoi(v) : kDenseLaunchTiles;
    }
    if (e->tier <= 1 && grid.x >= e->dense_tiles) {
        // the small tiers (scene in scalar registers) have the 8-wave build for this, and no tail worth a
        // plan: the 3840x2160 frame takes 0.284 ms with its slow tiles first against 0.283 ms without
        launch_frame_kernel(e, true, grid, a0);
        return RM_OK;
    }
    RenderArgs a = a0;
#if !defined(RM_DIAG)
    // Every other launch is planned from the tiles' last lifetimes (see RenderArgs): a short one renders
    // its slow tiles first and in quarters, a long one (a large scene) only hands them out first.
    if (e->split_threshold < 0) {
        const char *v = getenv("RM_SPLIT_US");  // threshold for a quarter in microseconds; 0 switches the plans off
        const double us = v ? atof(v) : kSplitThresholdTicks / 100.0;
        e->split_threshold = us > 0.0 ? (int)std::min(us * 100.0, 1e9) : 0;
        const char *m = getenv("RM_SPLIT_TIMES_MEAN");  // (0 = the absolute threshold alone)
        if (m) e->split_times_mean = (unsigned)std::min(std::max(atoi(m), 0), 1000);
        const char *m2 = getenv("RM_EARLY_TIMES_MEAN");
        if (m2) e->early_times_mean = (unsigned)std::min(std::max(atoi(m2), 0), 1000);
    }
    const size_t frame_tiles = (size_t)((e->W + TILE - 1) / TILE) * ((e->H + TILE - 1) / TILE + 1);
    // A long launch of a large scene keeps the machine full but ends with the slow tiles of the rows
    // handed out last (the 7680x4320 stress frame: 99 % of the waves have ended after 14.2 of 15.8 ms):
    // its plan lists the slow tiles to go first, whole - quarters would cost throughput there.
    // (A launch that names more tiles than the frame has - strips far below it - is not planned: the
    // plan's flags are one per frame tile.)
    if (e->split_on && e->split_threshold > 0 && grid.x <= frame_tiles) {
        if (!e->d_tile_cost) {
            // (cleared before this call returns: the other streams that launch frames of this engine keep
            // their plans in other parts of the same buffers, and nothing else would order their first plan
            // after a clearing enqueued on this launch's stream)
            HIP_TRY(dev_alloc(&e->d_tile_cost, frame_tiles * 4));
            HIP_TRY(hipMemset(e->d_tile_cost, 0, frame_tiles * 4 * sizeof(unsigned)));
            HIP_TRY(dev_alloc(&e->d_plan_words, (size_t)kPlanRing * kPlanStride));
            HIP_TRY(hipMemset(e->d_plan_words, 0, (size_t)kPlanRing * kPlanStride * sizeof(unsigned)));
            HIP_TRY(dev_alloc(&e->d_plan_flags, (size_t)kPlanRing * frame_tiles));
            HIP_TRY(hipDeviceSynchronize());
        }
#ifdef RM_DIAG_TIME
        a.plan_per = 1u;  // (the timeline build marks a tile's first pixels: whole tiles only)
#else
        a.plan_per = long_launch ? 1u : 4u;
#endif
        a.plan_cap = std::min<unsigned>(long_launch ? kPlanCapEarly : kPlanCap, std::max<unsigned>(16u, grid.x / 16u));
        rm_engine::PlanKey key;
        key.grid_x = grid.x;
        key.per = a.plan_per;
        key.cap = a.plan_cap;
        key.row_begin = a.row_begin;
        key.row_end = a.row_end;
        key.tile_row_stride = a.tile_row_stride;
        key.blocks_x = a.blocks_x;
        key.blocks_y = a.blocks_y;
        // A plan is written and read in the order of one stream and names the tiles of one launch shape,
        // so there is a plan per (stream, set of tiles): launches on one stream cannot overlap, and a
        // stream that alternates between several sets of tiles (a group whose ranks take turns with the
        // strips, rm_group_frame) finds each set's plan again.  When every plan is taken, a stream gives up
        // the one of its own it has not used for longest (the launch that read it is ahead of this one on
        // the same stream); a stream without any goes unplanned.
        unsigned slot = kPlanRing;
        for (unsigned i = 0; i < kPlanRing && slot == kPlanRing; i++)
            if (e->plan_taken[i] && e->plan_stream[i] == e->stream && key == e->plan_key[i]) slot = i;
        for (unsigned i = 0; i < kPlanRing && slot == kPlanRing; i++)
            if (!e->plan_taken[i]) {
                e->plan_taken[i] = true;
                e->plan_stream[i] = e->stream;
                e->plan_launches[i] = 0;
                slot = i;
            }
        if (slot == kPlanRing) {
            for (unsigned i = 0; i < kPlanRing; i++)
                if (e->plan_stream[i] == e->stream && (slot == kPlanRing || e->plan_used[i] < e->plan_used[slot])) slot = i;
            if (slot != kPlanRing) e->plan_launches[slot] = 0;
        }
        if (slot == kPlanRing) {
            launch_frame_kernel(e, false, grid, a);
            return RM_OK;
        }
        e->plan_used[slot] = ++e->plan_clock;
        unsigned *const words = e->d_plan_words + (size_t)slot * kPlanStride, *list = words + 64;
        unsigned char *flag = e->d_plan_flags + (size_t)slot * frame_tiles;
        a.tile_cost = e->d_tile_cost;
        a.cost_tiles = (unsigned)frame_tiles;
        // A plan stays in force for the launches of its stream and shape that follow it: list and flags
        // are left as they are, so every pixel is still rendered exactly once, and a plan that has aged
        // costs time at worst, like any wrong guess.  A fresh one is made for a plan's first four launches
        // (the first has no lifetimes to go by, the second no mean to hold them against: a scene full of
        // slow tiles overflows its list) and every eighth after that - the plan kernel is 5 us in front of
        // a launch that may take 100.
        if (e->plan_every < 0) {
            const char *v = getenv("RM_PLAN_EVERY");  // (1: a fresh plan for every launch)
            e->plan_every = v ? std::max(1, std::min(atoi(v), 1 << 20)) : 8;
        }
        const unsigned served = e->plan_launches[slot]++;
        // (a long launch always gets a fresh plan: 5 us are nothing to it, and a plan made from the lifetimes
        // of the frame before serves it better than an older one - the 7680x4320 stress frame takes 13.6 ms
        // with fresh plans and 15.0 ms with one kept for eight frames)
        const bool fresh = long_launch || served < 4 || served % (unsigned)e->plan_every == 0 || !(key == e->plan_key[slot]);
        if (fresh) {
            unsigned *count = words + 16 * e->plan_parity[slot], *count_next = words + 16 * (e->plan_parity[slot] ^ 1);
            e->plan_parity[slot] ^= 1;
            const unsigned *previous = e->last_plan_count ? e->last_plan_count : words + 32;  // (the first plan: words that stay zero)
            hipLaunchKernelGGL(split_plan_kernel, dim3((grid.x + 255) / 256), dim3(256), 0, e->stream, a, count, list, flag,
                               (unsigned)e->split_threshold, previous, long_launch ? e->early_times_mean : e->split_times_mean,
                               count_next);
            e->plan_key[slot] = key;
            e->plan_count_word[slot] = count;
        }
        a.plan_count = e->plan_count_word[slot];
        e->last_plan_count = a.plan_count;
        e->last_plan_cap = a.plan_cap;
        a.plan_list = list;
        a.plan_flag = flag;
        grid = dim3(grid.x + a.plan_per * a.plan_cap, 1, 1);
    }
#endif
    launch_frame_kernel(e, false, grid, a);
    return RM_OK;
}

int rm_engine_render_rows_device(rm_engine *e, void *d_bgra, size_t pitch_bytes, int row_begin, int row_end) {
    if (!e || !d_bgra) return fail(RM_ERR_INVALID, "NULL argument");
    if (row_begin < 0 || row_end > e->H || row_begin > row_end) return fail(RM_ERR_INVALID, "row range outside the frame");
    if (pitch_bytes < (size_t)e->W * 4 || (pitch_bytes & 3)) return fail(RM_ERR_INVALID, "pitch must be >= width*4 and a multiple of 4");
    if (row_begin == row_end) return RM_OK;
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    dim3 grid;
    RenderArgs a = make_render_args(e, d_bgra, pitch_bytes, row_begin, row_end, grid);
    if (e->refraction)
        RM_LAUNCH(render_tree_kernel, e, grid, dim3(BLOCK_THREADS), a, e->ssaa ? 4 : 1);
    else if (e->ssaa)
        RM_LAUNCH(render_ssaa_kernel, e, grid, dim3(BLOCK_THREADS), a);
    else if (e->two_pass > 0 && (size_t)(row_end - row_begin) * pitch_bytes < ((size_t)1 << 32)) {
        const size_t capacity = (size_t)a.blocks_x * a.blocks_y * BLOCK_THREADS;
        if (capacity > e->queue_capacity) {
            HIP_TRY(hipStreamSynchronize(e->stream));
            (void)hipFree(e->d_queue);
            e->d_queue = nullptr;
            HIP_TRY(dev_alloc(&e->d_queue, capacity * 14));
            e->queue_capacity = capacity;
        }
        if (!e->d_queue_count) HIP_TRY(dev_alloc(&e->d_queue_count, 1));
        HIP_TRY(hipMemsetAsync(e->d_queue_count, 0, sizeof(unsigned), e->stream));
        const ReflectQueue q = {e->d_queue_count, e->d_queue, (unsigned)e->queue_capacity};
        RM_LAUNCH(render_primary_kernel, e, grid, dim3(BLOCK_THREADS), a, q);
        const dim3 grid2((unsigned)std::min<size_t>((size_t)e->two_pass, (capacity + BLOCK_THREADS - 1) / BLOCK_THREADS));
        RM_LAUNCH(render_reflect_kernel, e, grid2, dim3(BLOCK_THREADS), a, q);
    } else
        if (int rc_ = launch_render(e, grid, a)) return rc_;
    HIP_TRY(hipGetLastError());
    return RM_OK;
}

int rm_engine_render_strips_device(rm_engine *e, void *d_bgra, size_t pitch_bytes, int first_strip, int strip_stride,
                                   int n_strips) {
    if (!e || !d_bgra) return fail(RM_ERR_INVALID, "NULL argument");
    if (first_strip < 0 || strip_stride < 1 || n_strips < 0) return fail(RM_ERR_INVALID, "bad strip selection");
    if (pitch_bytes < (size_t)e->W * 4 || (pitch_bytes & 3)) return fail(RM_ERR_INVALID, "pitch must be >= width*4 and a multiple of 4");
    if (n_strips == 0 || first_strip * RM_STRIP_ROWS >= e->H) return RM_OK;
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    dim3 grid;
    RenderArgs a = make_render_args(e, d_bgra, pitch_bytes, first_strip * RM_STRIP_ROWS, e->H, grid);
    a.blocks_y = n_strips;
    a.tile_row_stride = strip_stride;
    grid = dim3((unsigned)(a.blocks_x * a.blocks_y), 1, 1);
    if (e->refraction)
        RM_LAUNCH(render_tree_kernel, e, grid, dim3(BLOCK_THREADS), a, e->ssaa ? 4 : 1);
    else if (e->ssaa)
        RM_LAUNCH(render_ssaa_kernel, e, grid, dim3(BLOCK_THREADS), a);
    else
        if (int rc_ = launch_render(e, grid, a)) return rc_;
    HIP_TRY(hipGetLastError());
    return RM_OK;
}

int rm_engine_render(rm_engine *e, uint8_t *bgra, size_t pitch_bytes, int32_t mouse_x, int32_t mouse_y,
                     const float rel_move[3], int32_t scroll) {
    if (!e || !bgra) return fail(RM_ERR_INVALID, "NULL argument");
    if (pitch_bytes < (size_t)e->W * 4) return fail(RM_ERR_INVALID, "pitch must be >= width*4");
    int rc = rm_engine_input(e, mouse_x, mouse_y, rel_move, scroll);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    if (!e->d_frame) HIP_TRY(dev_alloc(&e->d_frame, (size_t)e->W * e->H * 4));
    // Pinned destinations (rm_host_alloc) take the copy directly at PCIe rate; pageable ones go
    // through a pinned staging frame and a host memcpy.  Either way a large frame leaves in row
    // bands, bottom band first (the expensive one in scenes with a floor): a band is copied out while
    // the next ones still render, and with a pageable destination the host copies band b out of the
    // staging frame while band b+1 crosses the bus.
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, bgra) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();  // an unregistered pointer reports an error: not a failure here
    if (!pinned && !e->h_stage)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_stage), (size_t)e->W * e->H * 4, hipHostMallocDefault));
    if (!e->xfer_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&e->xfer_stream, hipStreamNonBlocking));
        for (int b = 0; b < kBands; b++) {
            HIP_TRY(hipEventCreateWithFlags(&e->ev_band_rendered[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&e->ev_band_copied[b], hipEventDisableTiming));
        }
    }
    const size_t row_bytes = (size_t)e->W * 4;
    const int strips = (e->H + RM_STRIP_ROWS - 1) / RM_STRIP_ROWS;
    int r0s[kBands], r1s[kBands], n_bands = 0;
    int want = (size_t)e->W * e->H >= kBandedFramePixels ? kDefaultBands : 1;
    if (const char *v = getenv("RM_BANDS")) want = std::max(1, std::min(kBands, atoi(v)));  // tuning aid
    for (int b = want - 1; b >= 0; b--) {  // bottom band first
        const int r0 = (int)((long long)strips * b / want) * RM_STRIP_ROWS;
        const int r1 = std::min(e->H, (int)((long long)strips * (b + 1) / want) * RM_STRIP_ROWS);
        if (r1 <= r0) continue;
        r0s[n_bands] = r0;
        r1s[n_bands] = r1;
        n_bands++;
    }
    for (int b = 0; b < n_bands; b++) {
        const int r0 = r0s[b], r1 = r1s[b];
        rc = rm_engine_render_rows_device(e, e->d_frame + (size_t)r0 * row_bytes, row_bytes, r0, r1);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(e->ev_band_rendered[b], e->stream));
        HIP_TRY(hipStreamWaitEvent(e->xfer_stream, e->ev_band_rendered[b], 0));
        if (pinned && pitch_bytes == row_bytes)
            HIP_TRY(hipMemcpyAsync(bgra + (size_t)r0 * row_bytes, e->d_frame + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes,
                                   hipMemcpyDeviceToHost, e->xfer_stream));
        else if (pinned)
            HIP_TRY(hipMemcpy2DAsync(bgra + (size_t)r0 * pitch_bytes, pitch_bytes, e->d_frame + (size_t)r0 * row_bytes, row_bytes,
                                     row_bytes, r1 - r0, hipMemcpyDeviceToHost, e->xfer_stream));
        else
            HIP_TRY(hipMemcpyAsync(e->h_stage + (size_t)r0 * row_bytes, e->d_frame + (size_t)r0 * row_bytes,
                                   (size_t)(r1 - r0) * row_bytes, hipMemcpyDeviceToHost, e->xfer_stream));
        HIP_TRY(hipEventRecord(e->ev_band_copied[b], e->xfer_stream));
    }
    if (pinned) {
        HIP_TRY(hipStreamSynchronize(e->xfer_stream));
        return RM_OK;
    }
    for (int b = 0; b < n_bands; b++) {
        HIP_TRY(hipEventSynchronize(e->ev_band_copied[b]));
        const int r0 = r0s[b], r1 = r1s[b];
        if (pitch_bytes == row_bytes) {
            memcpy(bgra + (size_t)r0 * row_bytes, e->h_stage + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes);
        } else {
            for (int y = r0; y < r1; y++) memcpy(bgra + (size_t)y * pitch_bytes, e->h_stage + (size_t)y * row_bytes, row_bytes);
        }
    }
    return RM_OK;
}

int rm_engine_render_wait(rm_engine *e) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    const int slot = e->async_tail;
    if (!e->async_pending[slot]) return fail(RM_ERR_INVALID, "no frame is pending");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventSynchronize(e->ev_copied[slot]));
    e->async_pending[slot] = false;
    e->async_tail = (slot + 1) % RM_FRAMES_IN_FLIGHT;
    return RM_OK;
}

int rm_engine_render_async(rm_engine *e, uint8_t *bgra, size_t pitch_bytes, int32_t mouse_x, int32_t mouse_y,
                           const float rel_move[3], int32_t scroll) {
    if (!e || !bgra) return fail(RM_ERR_INVALID, "NULL argument");
    if (pitch_bytes < (size_t)e->W * 4) return fail(RM_ERR_INVALID, "pitch must be >= width*4");
    HIP_TRY(hipSetDevice(e->device));
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, bgra) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    if (!pinned) return fail(RM_ERR_INVALID, "rm_engine_render_async needs a page-locked destination (rm_host_alloc)");
    const int slot = e->async_head;
    if (e->async_pending[slot]) {  // every slot is in use: the oldest frame has to land first
        int rc = rm_engine_render_wait(e);
        if (rc) return rc;
    }
    if (!e->ev_main) HIP_TRY(hipEventCreateWithFlags(&e->ev_main, hipEventDisableTiming));
    if (!e->d_async[slot]) {
        HIP_TRY(dev_alloc(&e->d_async[slot], (size_t)e->W * e->H * 4));
        HIP_TRY(hipStreamCreateWithFlags(&e->slot_stream[slot], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_copied[slot], hipEventDisableTiming));
    }
    int rc = rm_engine_input(e, mouse_x, mouse_y, rel_move, scroll);
    if (rc) return rc;
    // whatever the engine's stream holds so far (lightmaps, an earlier synchronous frame) comes first
    HIP_TRY(hipEventRecord(e->ev_main, e->stream));
    HIP_TRY(hipStreamWaitEvent(e->slot_stream[slot], e->ev_main, 0));
    // the previous copy out of this slot has finished (its frame was waited for), so it can be overwritten
    hipStream_t own = e->stream;
    e->stream = e->slot_stream[slot];
    rc = rm_engine_render_rows_device(e, e->d_async[slot], (size_t)e->W * 4, 0, e->H);
    e->stream = own;
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(bgra, pitch_bytes, e->d_async[slot], (size_t)e->W * 4, (size_t)e->W * 4, e->H,
                             hipMemcpyDeviceToHost, e->slot_stream[slot]));
    HIP_TRY(hipEventRecord(e->ev_copied[slot], e->slot_stream[slot]));
    e->async_pending[slot] = true;
    e->async_head = (slot + 1) % RM_FRAMES_IN_FLIGHT;
    return RM_OK;
}

int rm_engine_fxaa_device(rm_engine *e, const void *d_src, void *d_dst, int row_begin, int row_end) {
    if (!e || !d_src || !d_dst) return fail(RM_ERR_INVALID, "NULL argument");
    if (d_src == d_dst) return fail(RM_ERR_INVALID, "FXAA runs out of place: source and destination must differ");
    if (row_begin < 0 || row_end > e->H || row_begin > row_end) return fail(RM_ERR_INVALID, "row range outside the frame");
    if (row_begin == row_end) return RM_OK;
    if ((size_t)e->W * e->H >= ((size_t)1 << 30)) return fail(RM_ERR_UNSUPPORTED, "FXAA addresses the frame with 32-bit byte offsets: frame too large");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    dim3 grid((e->W + FX_COLS - 1) / FX_COLS, (row_end - row_begin + 4 * FX_ROWS - 1) / (4 * FX_ROWS));
    hipLaunchKernelGGL(fxaa_kernel, grid, dim3(256), 0, e->stream, static_cast<const unsigned char *>(d_src),
                       static_cast<unsigned char *>(d_dst), e->W, e->H, row_begin, row_end);
    HIP_TRY(hipGetLastError());
    return RM_OK;
}

int rm_engine_pack_bgr_device(rm_engine *e, const void *d_bgra, void *d_bgr, size_t n_pixels) {
    if (!e || !d_bgra || !d_bgr) return fail(RM_ERR_INVALID, "NULL argument");
    if (n_pixels % 4) return fail(RM_ERR_INVALID, "pixels are packed four at a time: the count must be a multiple of 4");
    if (n_pixels == 0) return RM_OK;
    HIP_TRY(hipSetDevice(e->device));
    const size_t groups = n_pixels / 4;
    if ((groups + 255) / 256 > 0x7FFFFFFFu) return fail(RM_ERR_UNSUPPORTED, "too many pixels for one launch");
    hipLaunchKernelGGL(pack_bgr_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, e->stream,
                       static_cast<const uint4 *>(d_bgra), static_cast<unsigned *>(d_bgr), groups);
    HIP_TRY(hipGetLastError());
    return RM_OK;
}

static int unpack_bgr_strips(rm_engine *e, const void *d_bgr, void *d_bgra_frame, int world, int strips_per_rank, int shift);
int rm_engine_unpack_bgr_strips_device(rm_engine *e, const void *d_bgr, void *d_bgra_frame, int world, int strips_per_rank) {
    return unpack_bgr_strips(e, d_bgr, d_bgra_frame, world, strips_per_rank, 0);
}
}  // extern "C"
// (`shift`: part p of the packed rows holds the strips (p + shift) mod world, (p + shift) mod world + world, ...)
static int unpack_bgr_strips(rm_engine *e, const void *d_bgr, void *d_bgra_frame, int world, int strips_per_rank, int shift) {
    if (!e || !d_bgr || !d_bgra_frame) return fail(RM_ERR_INVALID, "NULL argument");
    if (e->W % 4) return fail(RM_ERR_UNSUPPORTED, "packed rows need a frame width that is a multiple of 4");
    const int strips = (e->H + RM_STRIP_ROWS - 1) / RM_STRIP_ROWS;
    if (world < 1 || strips_per_rank < 1 || (long long)world * strips_per_rank < strips)
        return fail(RM_ERR_INVALID, "the ranks' strips do not cover the frame");
    if (e->H > 65535) return fail(RM_ERR_UNSUPPORTED, "frame too high for one launch");
    HIP_TRY(hipSetDevice(e->device));
    const int W4 = e->W / 4;
    hipLaunchKernelGGL(unpack_bgr_strips_kernel, dim3((W4 + 255) / 256, e->H), dim3(256), 0, e->stream,
                       static_cast<const unsigned *>(d_bgr), static_cast<uint4 *>(d_bgra_frame), W4, e->H, world, strips_per_rank, shift);
    HIP_TRY(hipGetLastError());
    return RM_OK;
}
extern "C" {

int rm_engine_trace_primary(rm_engine *e, int32_t *hit, int32_t *steps, float *position) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    size_t npx = (size_t)e->W * e->H;
    int *d_hit = nullptr, *d_steps = nullptr;
    float *d_pos = nullptr;
    hipError_t err = dev_alloc(&d_hit, npx);  // (every path below reaches the frees)
    if (err == hipSuccess) err = dev_alloc(&d_steps, npx);
    if (err == hipSuccess) err = dev_alloc(&d_pos, npx * 3);
    if (err == hipSuccess) {
        dim3 grid;
        RenderArgs a = make_render_args(e, d_hit, (size_t)e->W * 4, 0, e->H, grid);
        RM_LAUNCH(trace_kernel, e, grid, dim3(BLOCK_THREADS), a, d_hit, d_steps, d_pos);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err == hipSuccess && hit) err = hipMemcpy(hit, d_hit, npx * sizeof(int), hipMemcpyDeviceToHost);
    if (err == hipSuccess && steps) err = hipMemcpy(steps, d_steps, npx * sizeof(int), hipMemcpyDeviceToHost);
    if (err == hipSuccess && position) err = hipMemcpy(position, d_pos, npx * 3 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_hit);
    (void)hipFree(d_steps);
    (void)hipFree(d_pos);
    if (err != hipSuccess) return fail(RM_ERR_HIP, std::string("trace_primary: ") + hipGetErrorString(err));
    return RM_OK;
}

int rm_engine_probe_shade(rm_engine *e, int n, const int32_t *object, const float *position, const float *direction,
                          float *colour) {
    if (!e || !object || !position || !direction || !colour || n < 0) return fail(RM_ERR_INVALID, "bad argument");
    if (n == 0) return RM_OK;
    for (int i = 0; i < n; i++)
        if (object[i] < 0 || object[i] >= e->n) return fail(RM_ERR_INVALID, "object index out of range");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    int *d_obj = nullptr;
    float *d_pos = nullptr, *d_dir = nullptr, *d_out = nullptr;
    hipError_t err = dev_alloc(&d_obj, n);  // (every path below reaches the frees)
    if (err == hipSuccess) err = dev_alloc(&d_pos, (size_t)n * 3);
    if (err == hipSuccess) err = dev_alloc(&d_dir, (size_t)n * 3);
    if (err == hipSuccess) err = dev_alloc(&d_out, (size_t)n * 3);
    if (err == hipSuccess) err = hipMemcpy(d_obj, object, sizeof(int) * n, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d_pos, position, sizeof(float) * 3 * n, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d_dir, direction, sizeof(float) * 3 * n, hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        dim3 grid((n + 63) / 64);
        RM_LAUNCH(probe_shade_kernel, e, grid, dim3(64), (const SceneView *)e->d_view, n, (const int *)d_obj,
                  (const float *)d_pos, (const float *)d_dir, d_out);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err == hipSuccess) err = hipMemcpy(colour, d_out, sizeof(float) * 3 * n, hipMemcpyDeviceToHost);
    (void)hipFree(d_obj);
    (void)hipFree(d_pos);
    (void)hipFree(d_dir);
    (void)hipFree(d_out);
    if (err != hipSuccess) return fail(RM_ERR_HIP, std::string("probe_shade: ") + hipGetErrorString(err));
    return RM_OK;
}

int rm_engine_probe_pack(rm_engine *e, int n, const float *linear, float exposure, uint32_t *packed, uint32_t *packed_exact) {
    if (!e || !linear || !packed || !packed_exact || n < 0) return fail(RM_ERR_INVALID, "bad argument");
    if (n == 0) return RM_OK;
    HIP_TRY(hipSetDevice(e->device));
    float *d_in = nullptr;
    unsigned *d_out = nullptr;
    hipError_t err = dev_alloc(&d_in, (size_t)n * 3);
    if (err == hipSuccess) err = dev_alloc(&d_out, (size_t)n * 2);
    if (err == hipSuccess) err = hipMemcpy(d_in, linear, sizeof(float) * 3 * n, hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        hipLaunchKernelGGL(probe_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, (const float *)d_in, exposure, d_out,
                           d_out + n);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err == hipSuccess) err = hipMemcpy(packed, d_out, sizeof(unsigned) * n, hipMemcpyDeviceToHost);
    if (err == hipSuccess) err = hipMemcpy(packed_exact, d_out + n, sizeof(unsigned) * n, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (err != hipSuccess) return fail(RM_ERR_HIP, std::string("probe_pack: ") + hipGetErrorString(err));
    return RM_OK;
}

// ---- radiosity -------------------------------------------------------------

}  // extern "C" (the steps below are shared by the allocating and the non-allocating entry points)

// engine.rs:259-288: clouds and direct lighting of every patch into `b`.
// (`direct` = false leaves the direct lighting to the form-factor launch that follows: rad_form_on
// with `direct` = true over every patch.)
static int rad_begin_on(rm_engine *e, const RadBuffers &b, bool direct = true) {
    const int NP = e->n_patches;
    if (NP == 0) return RM_OK;
    if (!e->sample_pos_ok) return fail(RM_ERR_UNSUPPORTED, "object type has no lightmap parameterisation");
    HIP_TRY(hipMemsetAsync(b.emissive_next, 0, sizeof(float) * 3 * NP, e->stream));
    RadArgs a = e->rad_args(b);
    dim3 grid((NP + 63) / 64);
    hipLaunchKernelGGL(rad_cloud_kernel, grid, dim3(64), 0, e->stream, a);
    if (direct) RM_LAUNCH(rad_direct_kernel, e, grid, dim3(64), a);
    HIP_TRY(hipGetLastError());
    return RM_OK;
}

static int rad_form_on(rm_engine *e, const RadBuffers &b, int p0, int p1, bool direct = false) {
    if (p0 == p1) return RM_OK;
    const int NP = e->n_patches;
    if (direct && (p0 != 0 || p1 != NP)) return fail(RM_ERR_INVALID, "direct lighting rides on a form-factor launch over every patch only");
    RadArgs a = e->rad_args(b);
    // sources go into the grid's y dimension, 65535 rows at most per launch (RM_RAD_CHUNK: test hook)
    int chunk = 65535;
    if (const char *c = getenv("RM_RAD_CHUNK")) chunk = std::max(2, std::min(65535, atoi(c)));
    // Sources per workgroup: one while the launch has few waves (its longest march bounds it), more
    // once every wave slot of the machine is taken many times over - as many as leave a quarter of a
    // million workgroups, sixteen at most (MAP_SIZE 32: 0.99 -> 0.89 ms with four, 64: 11.5 -> 10.3 ms
    // with sixteen; the matrix is the same bit for bit, each pair is computed as before).
    int per = 1;
    while (per < 16 && (long)((p1 - p0 + 63) / 64) * NP / (2 * per) >= 250000) per *= 2;
    if (const char *c = getenv("RM_RAD_PER")) per = std::max(1, std::min(64, atoi(c)));  // tuning aid
    if (per == 1) {
        for (int j0 = 0; j0 < NP;) {
            const int extra = direct && j0 == 0 ? 1 : 0;  // the first launch's first row
            const int rows = std::min(chunk - extra, NP - j0);
            dim3 grid((p1 - p0 + 63) / 64, rows + extra);
            RM_LAUNCH(rad_form_kernel, e, grid, dim3(64), a, p0, p1, j0, extra);
            HIP_TRY(hipGetLastError());
            j0 += rows;
        }
        return RM_OK;
    }
    if (direct) {  // (a launch of its own here: the machine is busy for milliseconds either way)
        RM_LAUNCH(rad_direct_kernel, e, dim3((NP + 63) / 64), dim3(64), a);
        HIP_TRY(hipGetLastError());
    }
    for (int j0 = 0; j0 < NP;) {
        const int rows = (int)std::min((long)chunk * per, (long)(NP - j0));
        dim3 grid((p1 - p0 + 63) / 64, (rows + per - 1) / per);
        RM_LAUNCH(rad_form_multi_kernel, e, grid, dim3(64), a, p0, p1, j0, j0 + rows, per);
        HIP_TRY(hipGetLastError());
        j0 += rows;
    }
    return RM_OK;
}

static int rad_bounce_on(rm_engine *e, const RadBuffers &b, int p0, int p1) {
    if (p0 == p1) return RM_OK;
    RadArgs a = e->rad_args(b);
    dim3 grid((p1 - p0 + 63) / 64);
    // Slabs of 128 sources (four workgroups per compute unit) - except where 192 (three per unit) still
    // hold every workgroup of the launch at once and each unit has at least one: more bytes in flight
    // per workgroup, and no second round to wait for (MAP_SIZE 64, 512 workgroups: 0.73 against 0.78 ms;
    // smaller launches are bound by the fold's latency and larger ones by rounds, both better at 128).
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device);
    int chunk = ((int)grid.x >= cus && (int)grid.x <= 3 * cus) ? 192 : 128;
    if (const char *c = getenv("RM_BOUNCE_CHUNK")) chunk = atoi(c);  // tuning aid
    if (e->exact_occlusion)
        hipLaunchKernelGGL((rad_bounce_kernel<128, true>), grid, dim3(256), 0, e->stream, a, p0, p1);
    else if (chunk == 64)
        hipLaunchKernelGGL(rad_bounce_kernel<64>, grid, dim3(256), 0, e->stream, a, p0, p1);
    else if (chunk == 192)
        hipLaunchKernelGGL(rad_bounce_kernel<192>, grid, dim3(256), 0, e->stream, a, p0, p1);
    else
        hipLaunchKernelGGL(rad_bounce_kernel<128>, grid, dim3(256), 0, e->stream, a, p0, p1);
    HIP_TRY(hipGetLastError());
    return RM_OK;
}

static int rad_commit_on(rm_engine *e, const RadBuffers &b) {
    if (e->n_patches)
        HIP_TRY(hipMemcpyAsync(b.emissive, b.emissive_next, sizeof(float) * 3 * e->n_patches, hipMemcpyDeviceToDevice, e->stream));
    return RM_OK;
}

extern "C" {

int rm_engine_radiosity_begin(rm_engine *e) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    if (int rc_ = quiesce_frames_in_flight(e)) return rc_;  // the precompute rewrites the lightmaps they sample
    const int NP = e->n_patches;
    if (!e->rad.col) {
        HIP_TRY(dev_alloc(&e->rad.col, (size_t)NP * 3));
        HIP_TRY(dev_alloc(&e->rad.nrm, (size_t)NP * 3));
        HIP_TRY(dev_alloc(&e->rad.emissive, (size_t)NP * 3));
        HIP_TRY(dev_alloc(&e->rad.emissive_next, (size_t)NP * 3));
    }
    e->rad_begun = true;
    return rad_begin_on(e, e->rad);
}

static int check_patch_range(rm_engine *e, int p0, int p1) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    if (!e->rad_begun) return fail(RM_ERR_INVALID, "call rm_engine_radiosity_begin first");
    if (p0 < 0 || p1 > e->n_patches || p0 > p1) return fail(RM_ERR_INVALID, "patch range outside [0, n_patches]");
    return RM_OK;
}

int rm_engine_radiosity_form_factors(rm_engine *e, int p0, int p1) {
    int rc = check_patch_range(e, p0, p1);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    const int NP = e->n_patches;
    if (!e->rad.ft) HIP_TRY(dev_alloc(&e->rad.ft, rad_ft_floats(NP)));
    return rad_form_on(e, e->rad, p0, p1);
}

int rm_engine_radiosity_bounce(rm_engine *e, int p0, int p1) {
    int rc = check_patch_range(e, p0, p1);
    if (rc) return rc;
    if (!e->rad.ft) return fail(RM_ERR_INVALID, "call rm_engine_radiosity_form_factors first");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    return rad_bounce_on(e, e->rad, p0, p1);
}

int rm_engine_radiosity_commit(rm_engine *e) {
    if (!e || !e->rad_begun) return fail(RM_ERR_INVALID, "radiosity not begun");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    return rad_commit_on(e, e->rad);
}

int rm_engine_radiosity_buffers(rm_engine *e, void **d_next_emissive, void **d_lightmaps, int *n_patches) {
    if (!e || !e->rad_begun) return fail(RM_ERR_INVALID, "radiosity not begun");
    if (d_next_emissive) *d_next_emissive = e->rad.emissive_next;
    if (d_lightmaps) *d_lightmaps = e->d_lightmaps;
    if (n_patches) *n_patches = e->n_patches;
    return RM_OK;
}

// The whole precompute on one GPU, shared by the two entry points below.  Against the stepwise calls
// (which patch sharding needs) it saves launches: the direct lighting rides on the form-factor launch,
// and a bounce over every patch rewrites the whole of emissive_next, so the two emissive maps trade
// places from bounce to bounce instead of being copied (rm_engine_radiosity_commit).  Returns with
// the latest emissive values in `b.emissive`.
static int rad_whole_on(rm_engine *e, RadBuffers &b, int bounces) {
    const int NP = e->n_patches;
    int rc = rad_begin_on(e, b, bounces == 0);
    if (rc) return rc;
    if (bounces > 0) {
        rc = rad_form_on(e, b, 0, NP, true);
        if (rc) return rc;
    }
    for (int k = 0; k < bounces; k++) {
        rc = rad_bounce_on(e, b, 0, NP);
        if (rc) return rc;
        std::swap(b.emissive, b.emissive_next);
    }
    return RM_OK;
}

int rm_engine_compute_lightmaps(rm_engine *e, int bounces) {
    if (!e || bounces < 0) return fail(RM_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    if (int rc_ = quiesce_frames_in_flight(e)) return rc_;  // the precompute rewrites the lightmaps they sample
    const int NP = e->n_patches;
    if (!e->rad.col) {
        HIP_TRY(dev_alloc(&e->rad.col, (size_t)NP * 3));
        HIP_TRY(dev_alloc(&e->rad.nrm, (size_t)NP * 3));
        HIP_TRY(dev_alloc(&e->rad.emissive, (size_t)NP * 3));
        HIP_TRY(dev_alloc(&e->rad.emissive_next, (size_t)NP * 3));
    }
    if (bounces > 0 && !e->rad.ft) HIP_TRY(dev_alloc(&e->rad.ft, rad_ft_floats(NP)));
    e->rad_begun = true;
    if (int rc = rad_whole_on(e, e->rad, bounces)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RM_OK;
}

// Engine::compute_lightmaps without a single allocation: the clouds, the emissive maps and the
// form-factor matrix live in the library's static scratch array (this device's copy of it) instead
// of buffers the engine allocates on first use.  The kernels are the same and take the memory as
// arguments, as they do for rm_engine_compute_lightmaps.
int rm_engine_compute_lightmaps_static(rm_engine *e, int bounces) {
    if (!e || bounces < 0) return fail(RM_ERR_INVALID, "bad argument");
    const int NP = e->n_patches;
    if (NP > RM_STATIC_MAX_PATCHES)
        return fail(RM_ERR_LIMIT, "rm_engine_compute_lightmaps_static: " + std::to_string(NP) + " patches exceed the static scratch (" +
                                      std::to_string(RM_STATIC_MAX_PATCHES) + " patches at most); use rm_engine_compute_lightmaps");
    if (!e->scratch) return fail(RM_ERR_HIP, "the static scratch array has no address on this device");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    if (int rc_ = quiesce_frames_in_flight(e)) return rc_;
    // (held until the stream has drained below: the call is synchronous anyway)
    std::lock_guard<std::mutex> in_use(g_scratch_in_use[e->device]);
    // the scratch array, cut up for NP patches: four [NP][3] arrays, then FT (rad_ft_floats(NP) floats) on a 256-byte boundary
    const size_t cloud = ((size_t)NP * 3 + 63) & ~(size_t)63;
    RadBuffers b;
    b.col = e->scratch;
    b.nrm = b.col + cloud;
    b.emissive = b.nrm + cloud;
    b.emissive_next = b.emissive + cloud;
    b.ft = b.emissive_next + cloud;
    // (writes every cloud entry and zeroes emissive_next; FT is written in full)
    if (int rc = rad_whole_on(e, b, bounces)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RM_OK;
}

int rm_engine_get_lightmap(rm_engine *e, int object, float *values) {
    if (!e || !values || object < 0 || object >= e->n) return fail(RM_ERR_INVALID, "bad argument");
    int slot = e->lm_slot[object];
    if (slot < 0) return fail(RM_ERR_UNSUPPORTED, "object has no lightmap");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    size_t count = (size_t)e->M * e->M * 3;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(values, e->d_lightmaps + slot * count, count * sizeof(float), hipMemcpyDeviceToHost));
    return RM_OK;
}

int rm_engine_set_lightmap(rm_engine *e, int object, const float *values) {
    if (!e || !values || object < 0 || object >= e->n) return fail(RM_ERR_INVALID, "bad argument");
    int slot = e->lm_slot[object];
    if (slot < 0) return fail(RM_ERR_UNSUPPORTED, "object has no lightmap");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    size_t count = (size_t)e->M * e->M * 3;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (int rc_ = quiesce_frames_in_flight(e)) return rc_;  // frames in flight still sample the old values
    HIP_TRY(hipMemcpy(e->d_lightmaps + slot * count, values, count * sizeof(float), hipMemcpyHostToDevice));
    return RM_OK;
}

// Lightmap file: "RMLM", u32 version (1), u32 map_size, u32 lightmap count, then per lightmap the u32
// object index and map_size * map_size * 3 f32 values, first index u (sample_map[u][v]); little endian.
int rm_engine_save_lightmaps(rm_engine *e, const char *path) {
    if (!e || !path) return fail(RM_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<float> all((size_t)e->n_patches * 3);
    if (!all.empty()) HIP_TRY(hipMemcpy(all.data(), e->d_lightmaps, all.size() * sizeof(float), hipMemcpyDeviceToHost));
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RM_ERR_INVALID, std::string("cannot write ") + path);
    const uint32_t head[4] = {0x4D4C4D52u /* "RMLM" */, 1u, (uint32_t)e->M, (uint32_t)e->n_lm};
    bool ok = fwrite(head, sizeof(head), 1, f) == 1;
    const size_t per = (size_t)e->M * e->M * 3;
    for (int slot = 0; slot < e->n_lm && ok; slot++) {
        const uint32_t obj = (uint32_t)e->patch_obj[slot];
        ok = fwrite(&obj, sizeof(obj), 1, f) == 1 && fwrite(all.data() + slot * per, sizeof(float), per, f) == per;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? RM_OK : fail(RM_ERR_INVALID, std::string("short write to ") + path);
}

int rm_engine_load_lightmaps(rm_engine *e, const char *path) {
    if (!e || !path) return fail(RM_ERR_INVALID, "NULL argument");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(RM_ERR_INVALID, std::string("cannot read ") + path);
    uint32_t head[4] = {0, 0, 0, 0};
    const size_t per = (size_t)e->M * e->M * 3;
    std::vector<float> all((size_t)e->n_patches * 3);
    std::string why;
    if (fread(head, sizeof(head), 1, f) != 1 || head[0] != 0x4D4C4D52u || head[1] != 1u)
        why = "not a lightmap file (version 1)";
    else if (head[2] != (uint32_t)e->M || head[3] != (uint32_t)e->n_lm)
        why = "the file holds " + std::to_string(head[3]) + " lightmaps of size " + std::to_string(head[2]) + ", the engine " +
              std::to_string(e->n_lm) + " of size " + std::to_string(e->M);
    for (int slot = 0; slot < e->n_lm && why.empty(); slot++) {
        uint32_t obj = 0;
        if (fread(&obj, sizeof(obj), 1, f) != 1 || fread(all.data() + slot * per, sizeof(float), per, f) != per)
            why = "the file ends early";
        else if (obj != (uint32_t)e->patch_obj[slot])
            why = "lightmap " + std::to_string(slot) + " belongs to object " + std::to_string(obj) + ", the engine expects object " +
                  std::to_string(e->patch_obj[slot]);
    }
    if (why.empty() && fgetc(f) != EOF) why = "trailing bytes";
    fclose(f);
    if (!why.empty()) return fail(RM_ERR_INVALID, std::string(path) + ": " + why);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (int rc_ = quiesce_frames_in_flight(e)) return rc_;  // frames in flight still sample the old values
    if (!all.empty()) HIP_TRY(hipMemcpy(e->d_lightmaps, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice));
    return RM_OK;
}

int rm_engine_set_ssaa(rm_engine *e, int samples) {
    if (!e || (samples != 1 && samples != 4)) return fail(RM_ERR_INVALID, "samples must be 1 or 4");
    e->ssaa = samples == 4;
    return RM_OK;
}

int rm_engine_set_tetrahedral_normals(rm_engine *e, int on) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    e->tetra_normals = on != 0;
    e->view_dirty = true;
    return RM_OK;
}

int rm_engine_set_refraction(rm_engine *e, int on) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    e->refraction = on != 0;
    e->view_dirty = true;
    return RM_OK;
}

int rm_engine_set_exact_occlusion(rm_engine *e, int on) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    e->exact_occlusion = on != 0;
    e->view_dirty = true;
    return RM_OK;
}

int rm_engine_set_precision(rm_engine *e, int precision) {
    if (!e || (precision != RM_PRECISION_STRICT && precision != RM_PRECISION_FAST)) return fail(RM_ERR_INVALID, "rm_engine_set_precision: unknown precision");
    e->fast = precision == RM_PRECISION_FAST;
    return RM_OK;
}

int rm_engine_set_tile_splitting(rm_engine *e, int on) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    e->split_on = on != 0;
    return RM_OK;
}

int rm_engine_split_tiles(rm_engine *e, uint32_t *count) {
    if (!e || !count) return fail(RM_ERR_INVALID, "NULL argument");
    *count = 0;
    if (!e->last_plan_count) return RM_OK;
    HIP_TRY(hipSetDevice(e->device));
    quiesce_frames_in_flight(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    unsigned n = 0;
    HIP_TRY(hipMemcpy(&n, e->last_plan_count, sizeof(n), hipMemcpyDeviceToHost));
    *count = std::min<unsigned>(n, e->last_plan_cap);  // (tiles found slow beyond the list's length were rendered whole)
    return RM_OK;
}

int rm_engine_set_march_cap(rm_engine *e, int max_steps) {
    if (!e || max_steps < 0) return fail(RM_ERR_INVALID, "bad argument");
    e->march_cap = max_steps;
    e->view_dirty = true;
    return RM_OK;
}

#ifdef RM_DIAG
// Diagnostic builds only: the public counters followed by the per-section cycle sums.
extern "C" int rm_diag_read(rm_engine *e, uint64_t out[kStatSlots]) {
    if (hipStreamSynchronize(e->stream) != hipSuccess) return -1;
    return hipMemcpy(out, e->d_stats, sizeof(uint64_t) * kStatSlots, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

int rm_engine_get_stats(rm_engine *e, uint64_t stats[RM_STAT_COUNT]) {
    if (!e || !stats) return fail(RM_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = rm_engine_synchronize(e)) return rc_;  // frames in flight count too
    HIP_TRY(hipMemcpy(stats, e->d_stats, sizeof(uint64_t) * RM_STAT_COUNT, hipMemcpyDeviceToHost));
    return RM_OK;
}
int rm_engine_reset_stats(rm_engine *e) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    HIP_TRY(hipMemsetAsync(e->d_stats, 0, sizeof(uint64_t) * kStatSlots, e->stream));
    return RM_OK;
}

int rm_engine_shader_clock(rm_engine *e, int microseconds, double *mhz) {
    if (!e || !mhz || microseconds < 1 || microseconds > 100000) return fail(RM_ERR_INVALID, "rm_engine_shader_clock: 1 to 100000 microseconds");
    HIP_TRY(hipSetDevice(e->device));
    if (!e->clock_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&e->clock_stream, hipStreamNonBlocking));
        HIP_TRY(dev_alloc(&e->d_clock, 2));
    }
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, e->clock_stream, e->d_clock, (unsigned)microseconds * 100u);
    HIP_TRY(hipGetLastError());
    unsigned long long v[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(v, e->d_clock, sizeof(v), hipMemcpyDeviceToHost, e->clock_stream));
    HIP_TRY(hipStreamSynchronize(e->clock_stream));
    *mhz = v[1] ? (double)v[0] / (double)v[1] * 100.0 : 0.0;  // cycles per 10 ns tick
    return RM_OK;
}

int rm_engine_synchronize(rm_engine *e) {
    if (!e) return fail(RM_ERR_INVALID, "NULL engine");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc_ = sync_view(e)) return rc_;
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < RM_FRAMES_IN_FLIGHT; i++)
        if (e->slot_stream[i]) HIP_TRY(hipStreamSynchronize(e->slot_stream[i]));
    if (e->xfer_stream) HIP_TRY(hipStreamSynchronize(e->xfer_stream));
    return RM_OK;
}

}  // extern "C"

#include "rm_group.hpp"
