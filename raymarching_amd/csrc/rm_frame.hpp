// The frame kernel's body (engine.rs:77-126 fused: direction, cast_sight_ray, tone map, pack) and its
// launch arguments, shared by the two translation units that compile it: rm_hip.hip (strict
// arithmetic, -ffp-contract=off) and rm_fast.hip (the fast tier, RM_FAST=1).
//
// Citations are file:line in the reference tree (src/...).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rm.h"
#include "rm_device.hpp"
#include "rm_types.hpp"

using namespace rm;

// A wave is an 8x8 pixel tile and a workgroup of its own: pixels never cooperate,
// so a finished wave frees its slot at once instead of waiting for siblings.
constexpr int TILE = 8;
static_assert(TILE == RM_STRIP_ROWS, "a strip is one row of wave tiles");
constexpr int BLOCK_THREADS = 64;

struct RenderArgs {
    const SceneView *sv;  // device memory: read where needed (scene_ref: by scalar loads)
    Pose pose;
    int row_begin, row_end;
    int blocks_x, blocks_y;
    int tile_row_stride;  // 1 = contiguous rows; n = every n-th 8-row strip, written compactly
    unsigned char *out;  // first byte of row `row_begin`
    size_t pitch;
    // Every primary ray starts at the camera, so the first distance evaluation of every primary march
    // is the same: the host makes it once per frame.  camera_dist = distance from the camera to the
    // nearest object (the smooth fold where the scene has one), camera_sph = to the nearest sphere
    // (+inf without spheres); camera_dist is NaN where that does not apply (a camera closer than
    // SMALL_DISTANCE to something, farther than MAX_MARCH_DISTANCE from everything, or not a number).
    float camera_dist, camera_sph;
    const float *screen_u, *screen_v;  // [W], [H]: engine.rs:79-82 per column and per row
    // Slow tiles in quarters (see SplitPlan below); all null/0 for a launch without a plan.
    const unsigned *plan_count;        // tiles on plan_list
    const unsigned *plan_list;         // launch-local numbers of the tiles rendered by four waves
    const unsigned char *plan_flag;    // [tiles of the launch]: 1 = on the list
    unsigned *tile_cost;               // [tiles of the frame][4]: how long each quarter last took, 10 ns ticks
    unsigned plan_cap;                 // the first plan_per * plan_cap workgroups render the listed tiles
    unsigned plan_per;                 // 4: a listed tile is rendered by four waves, a quarter each; 1: by one wave, only early
    unsigned cost_tiles;               // entries of tile_cost
};

// Slow tiles in quarters.  A launch that does not fill the machine many times over lasts as long as
// its slowest wave - in the default scene a tile in which two mirror spheres reflect each other:
// 7-8 marches one after the other, and twice the loop iterations of its busiest pixel, because the
// 64 pixels' chains drift apart.  Sixteen pixels drift apart less: the same tile as four waves of 16
// active lanes has its slowest wave done in 0.6 of the time (measured on a full machine).  Which
// tiles are slow is taken from the last time they were rendered: every wave leaves its lifetime in
// `tile_cost` (a whole-tile wave 5/8 of it in all four slots, the scale of a quarter), and before a
// launch `split_plan_kernel` lists the tiles whose slowest quarter reached the threshold.  The frame
// kernel's first 4 * cap workgroups take the listed tiles' quarters (or leave at once), so the slow
// tiles also start first; the workgroup of a listed tile in the usual order leaves at once.  The plan
// belongs to its launch (list and flags are written before it in stream order and read by nothing
// else), so every pixel is rendered exactly once whatever the history says; a wrong guess costs time
// only.  The pixels do not depend on which wave renders them.
constexpr unsigned kSplitThresholdTicks = 4500;  // 45 us for a quarter, 72 us for the whole tile (30-60 us measured alike)
__device__ __forceinline__ unsigned frame_tile_of(const RenderArgs &a, int bx, int by) {
    // (a launch of strips may name strips below the frame's last row: they share the last entry)
    return min((unsigned)((a.row_begin + by * a.tile_row_stride * TILE) / TILE) * (unsigned)a.blocks_x + (unsigned)bx, a.cost_tiles - 1u);
}
// Workgroup -> tile.  Tiles are handed out bottom row first: in scenes with a
// floor and things standing on it the lower rows cost most, and starting them
// first leaves the cheap tiles to fill the tail of the launch (measured: 10 %
// off the 4K frame of the default scene).  Any order renders the same pixels.
__device__ __forceinline__ void tile_pixel(const RenderArgs &a, int &x, int &y_inv, int &out_row) {
    const int lane = threadIdx.x & 63;
    const int by = a.blocks_y - 1 - blockIdx.x / a.blocks_x;
    const int bx = blockIdx.x % a.blocks_x;
    x = bx * TILE + (lane & 7);
    out_row = by * TILE + (lane >> 3);
    y_inv = a.row_begin + by * a.tile_row_stride * TILE + (lane >> 3);
}

// engine.rs:77-126 fused: direction, cast_sight_ray, tone map, pack.
// Waves per SIMD the compiler must make room for (second launch bound).  The large-scene kernel walks
// lists in memory and waits for them: it is bound by latency, not by instruction issue, and runs
// 20 % faster with 7 waves per SIMD and 21 of its values in private memory than with the 4 waves its
// 102 registers would allow (8 waves: the same time with twice the values spilled).  The small tiers
// keep their scene in scalar registers and are bound by instruction issue: the compiler's own choice
// (72 registers, 7 waves) is left alone.
template <class G>
struct MinWaves {
#ifdef RM_EXP_MINWAVES
    static constexpr int value = RM_EXP_MINWAVES;
#else
    static constexpr int value = 1;
#endif
};
#ifndef RM_LARGE_MINWAVES
#define RM_LARGE_MINWAVES 7
#endif
template <>
struct MinWaves<GroupedLarge> {
    static constexpr int value = RM_LARGE_MINWAVES;
};
// PLANS = false compiles the kernel without the plan and cost-table code (the 8-wave build, whose
// launches are never planned, has no registers to spare for it).
template <class G, bool PLANS = true>
__device__ __forceinline__ void render_tile(const RenderArgs &a, const G &g) {
    int x, y_inv, out_row;
#ifdef RM_DIAG_TIME
    const unsigned long long diag_t0 = wall_clock64();
    const unsigned long long diag_c0 = __builtin_readcyclecounter();
#endif
#if !defined(RM_DIAG)
    unsigned cost_at = 0, cost_t0 = 0;
    bool quarter = false;
    if (PLANS && a.tile_cost) {  // (the same in every wave of the launch)
        cost_t0 = (unsigned)wall_clock64();
        unsigned tile = blockIdx.x;
        if (a.plan_count) {
            const unsigned first_whole = a.plan_per * a.plan_cap;
            if (blockIdx.x < first_whole) {
                const unsigned listed = min(*a.plan_count, a.plan_cap);
                if (blockIdx.x >= a.plan_per * listed) return;
                quarter = a.plan_per == 4;
                tile = a.plan_list[quarter ? blockIdx.x >> 2 : blockIdx.x];
            } else {
                tile = blockIdx.x - first_whole;
                if (a.plan_flag[tile]) return;
            }
        }
        const int lane = threadIdx.x & 63;
        const int by = a.blocks_y - 1 - (int)(tile / (unsigned)a.blocks_x), bx = (int)(tile % (unsigned)a.blocks_x);
        int px, py;
        if (quarter) {
            const int q = blockIdx.x & 3;
            px = (q & 1) * 4 + (lane & 3);
            py = (q >> 1) * 4 + ((lane >> 2) & 3);
            cost_at = frame_tile_of(a, bx, by) * 4 + q;
        } else {
            px = lane & 7;
            py = lane >> 3;
            cost_at = frame_tile_of(a, bx, by) * 4;
        }
        x = (quarter && lane >= 16) ? 0x3fffffff : bx * TILE + px;
        out_row = by * TILE + py;
        y_inv = a.row_begin + by * a.tile_row_stride * TILE + py;
    } else
#endif
    tile_pixel(a, x, y_inv, out_row);
#ifdef RM_DIAG
    diag_mem()[threadIdx.x] = 0;
    if (threadIdx.x == 0) diag_mem()[64] = 0;
    if (threadIdx.x < 2 + DG_SECTIONS) diag_laps()[threadIdx.x] = 0;
    if (threadIdx.x < 8) diag_kinds()[threadIdx.x] = 0;
    if (threadIdx.x < 4) diag_lane_sums()[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) diag_laps()[0] = (unsigned)__builtin_readcyclecounter();
    const unsigned long long wave_t0 = wall_clock64();  // 100 MHz
#endif
    if (x >= a.pose.W || y_inv >= a.row_end) return;
    V3 dir = pixel_direction(a.pose, x, y_inv, a.screen_u, a.screen_v);
    V3 cam = v3(a.pose.cam[0], a.pose.cam[1], a.pose.cam[2]);
    diag_lap(DG_SETUP);
    V3 colour = cast_sight_ray(acc_of(g), scene_ref(a.sv), cam, dir, a.camera_dist, a.camera_sph);
    diag_lap(DG_REFLECT);
#ifdef RM_DIAG
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned mx = 0, sum = 0;
        for (int i = 0; i < 64; i++) {
            mx = max(mx, diag_mem()[i]);
            sum += diag_mem()[i];
        }
        atomicAdd(&a.sv->stats[4], (unsigned long long)diag_mem()[64]);  // loop iterations the wave ran
        atomicAdd(&a.sv->stats[5], (unsigned long long)mx);              // iterations of its busiest lane
        atomicAdd(&a.sv->stats[6], (unsigned long long)sum);             // lane iterations in use
    }
#endif
    unsigned px = pack_pixel(colour, a.pose.exposure);
#ifdef RM_DIAG_TIME
    // the first three pixels of each tile carry the wave's start, end (10 ns ticks, low 32 bits)
    // and the compute unit it ran on, instead of colours
    {
        const unsigned long long t1 = wall_clock64();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        hw = (hw & 0x0FFFFFFFu) | (xcc << 28);  // the die in place of the queue and state fields
        if (threadIdx.x == 0) px = (unsigned)diag_t0;
        if (threadIdx.x == 1) px = (unsigned)t1;
        if (threadIdx.x == 2) px = hw;
#ifndef RM_DIAG
        if (threadIdx.x == 3) px = (unsigned)(__builtin_readcyclecounter() - diag_c0);  // shader-clock cycles of the wave
#endif
#ifdef RM_DIAG
        // with the iteration counters also built in: loop iterations of the wave, of its busiest
        // lane, summed over lanes, and the number of marches
        __syncthreads();
        unsigned mx = 0, sum = 0;
        for (int i = 0; i < 64; i++) {
            mx = max(mx, diag_mem()[i]);
            sum += diag_mem()[i];
        }
        if (threadIdx.x == 3) px = diag_mem()[64];
        if (threadIdx.x == 4) px = mx;
        if (threadIdx.x == 5) px = sum;
        if (threadIdx.x == 6) px = diag_kinds()[5];
        if (threadIdx.x >= 8 && threadIdx.x < 8 + DG_SECTIONS) px = diag_laps()[2 + threadIdx.x - 8];  // cycles per section
        if (threadIdx.x >= 16 && threadIdx.x < 24) px = diag_kinds()[threadIdx.x - 16];
#endif
    }
#endif
    if constexpr (!PLANS) {
        // The pixel's place in the frame, worked out again from the lane and workgroup numbers (through a
        // copy of the lane number the compiler cannot trace back): kept from the top of the kernel, the
        // three values would each cost the 64-register build a spill - 1 KiB written per wave.
        int lane_again = (int)threadIdx.x;
        asm volatile("" : "+v"(lane_again));
        lane_again &= 63;
        const int by = a.blocks_y - 1 - (int)(blockIdx.x / (unsigned)a.blocks_x), bx = (int)(blockIdx.x % (unsigned)a.blocks_x);
        x = bx * TILE + (lane_again & 7);
        out_row = by * TILE + (lane_again >> 3);
    }
    *reinterpret_cast<unsigned *>(a.out + (size_t)out_row * a.pitch + (size_t)x * 4) = px;
#if !defined(RM_DIAG)
    if (PLANS && a.tile_cost && (threadIdx.x & 63) == 0) {
        const unsigned ticks = (unsigned)wall_clock64() - cost_t0;
        if (quarter) {
            a.tile_cost[cost_at] = ticks;
        } else {
            const unsigned as_quarter = (ticks * 5u) >> 3;
            reinterpret_cast<uint4 *>(a.tile_cost)[cost_at >> 2] = make_uint4(as_quarter, as_quarter, as_quarter, as_quarter);
        }
    }
#endif
#ifdef RM_DIAG
    diag_lap(DG_PACK);
    __syncthreads();
    if (threadIdx.x < DG_SECTIONS) atomicAdd(&a.sv->stats[RM_STAT_COUNT + threadIdx.x], (unsigned long long)diag_laps()[2 + threadIdx.x]);
    if (threadIdx.x < 8) atomicAdd(&a.sv->stats[RM_STAT_COUNT + 8 + threadIdx.x], (unsigned long long)diag_kinds()[threadIdx.x]);
    if (threadIdx.x == 0) {
        atomicAdd(&a.sv->stats[15], (unsigned long long)diag_lane_sums()[1]);  // lanes in use over march-loop iterations
        atomicAdd(&a.sv->stats[31], (unsigned long long)diag_lane_sums()[2]);  // ... over shadow-loop iterations
        const unsigned long long now = wall_clock64(), dur = now - wave_t0;
        atomicMax(&a.sv->stats[24], dur);           // longest wave, in 10 ns ticks
        atomicAdd(&a.sv->stats[25], dur);           // summed wave lifetimes
        atomicMax(&a.sv->stats[26], ~wave_t0);      // first start (complemented) and last end of the grid
        atomicMax(&a.sv->stats[27], now);
        if (dur * 10 > 20000) atomicAdd(&a.sv->stats[28], 1ULL);  // waves longer than 20 us
        if (dur * 10 > 50000) atomicAdd(&a.sv->stats[29], 1ULL);  // ... 50 us
        if (dur * 10 > 100000) atomicAdd(&a.sv->stats[30], 1ULL); // ... 100 us
    }
#endif
}

template <class G>
struct MinWavesFrame {
    static constexpr int value = MinWaves<G>::value > 7 ? MinWaves<G>::value : 7;
};
template <>
struct MinWavesFrame<GroupedLarge> {
    static constexpr int value = MinWaves<GroupedLarge>::value;
};
template <class G>
struct MinWavesDense {
    static constexpr int value = 8;
};
template <>
struct MinWavesDense<GroupedLarge> {
    static constexpr int value = MinWaves<GroupedLarge>::value;  // (never launched: see render_launch)
};
