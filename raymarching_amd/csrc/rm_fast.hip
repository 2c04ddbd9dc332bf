// The fast tier of the frame kernel (rm_engine_set_precision(engine, RM_PRECISION_FAST)): the same
// source as the strict kernel of rm_hip.hip, compiled in a translation unit of its own with RM_FAST = 1
// and -ffp-contract=fast-honor-pragmas.  What that relaxes and what it leaves strict is written at
// RM_FAST in rm_device.hpp.  The reference has one arithmetic (IEEE binary32 without contraction,
// vector.rs:22-32, ray.rs:28-39); this tier trades bit-identity with it for speed inside a stated
// tolerance: at most 2/255 per channel, at least 99.9 % of the pixels identical.
#define RM_FAST 1
#include "rm_frame.hpp"

template <class G>
__global__ __launch_bounds__(BLOCK_THREADS, MinWavesFrame<G>::value) void render_fast_kernel(const RenderArgs a, const G g) {
    render_tile(a, g);
}
// (for launches of many times the machine's wave slots: 8 waves per SIMD, no launch plans)
template <class G>
__global__ __launch_bounds__(BLOCK_THREADS, MinWavesDense<G>::value) void render_fast_dense_kernel(const RenderArgs a, const G g) {
    render_tile<G, false>(a, g);
}

// Called by launch_render in rm_hip.hip.  `scene` points at the engine's grouped scene of tier `tier`
// (-1: GroupedTierX, 0: GroupedTierA, 1: GroupedTierB, 2: GroupedLarge).
void rm_fast_launch(int tier, const void *scene, bool dense, dim3 grid, hipStream_t stream, const RenderArgs &a) {
    const dim3 block(BLOCK_THREADS);
    const auto go = [&](const auto &g) {
        typedef std::remove_cv_t<std::remove_reference_t<decltype(g)>> G;
        if (dense)
            hipLaunchKernelGGL((render_fast_dense_kernel<G>), grid, block, 0, stream, a, g);
        else
            hipLaunchKernelGGL((render_fast_kernel<G>), grid, block, 0, stream, a, g);
    };
    if (tier == -1)
        go(*static_cast<const GroupedTierX *>(scene));
    else if (tier == 0)
        go(*static_cast<const GroupedTierA *>(scene));
    else if (tier == 1)
        go(*static_cast<const GroupedTierB *>(scene));
    else
        go(*static_cast<const GroupedLarge *>(scene));
}
