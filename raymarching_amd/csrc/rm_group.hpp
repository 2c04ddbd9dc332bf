// Several GPUs behind the C ABI (include/rm.h, rm_group_* and rm_multi_*).  Part of rm_hip.hip's
// translation unit: it works on the engine's internals.
//
// The reference spreads a frame over CPU threads by rows (rayon, engine.rs:93) and is called once per
// frame from its main loop (main.rs:141-145).  Here a frame is spread over GPUs by 8-row strips:
//
//   rm_group   one rank per process (or thread) and GPU.  A frame is one C call per rank: this rank's
//              strips, three bytes per pixel on the wire, one all-gather on the frame's own stream
//              (RCCL over xGMI, looked up at run time so that hosts without it still load the
//              library; or an exchange function the host supplies), strips back into row order, FXAA
//              on the gathered frame.  Up to four frames are in flight on streams of their own.
//   rm_multi   one process, several GPUs, the frame wanted in host memory as Engine::render leaves
//              it: every GPU renders its strips and copies them over its own PCIe link straight into
//              the caller's frame - no exchange between GPUs at all.
//
// Citations are file:line in the reference tree (src/...).
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <functional>

// ---- kernels ---------------------------------------------------------------

// Strips as an all-gather of 4-byte pixels leaves them -> rows in frame order (the unpacked
// counterpart of unpack_bgr_strips_kernel, for frame widths that are not a multiple of 4).
__global__ __launch_bounds__(256) void reorder_strips_kernel(const unsigned *__restrict__ src, unsigned *__restrict__ dst, int W, int H,
                                                             int world, int per, int shift) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const int s = y / RM_STRIP_ROWS;
    const size_t srow = (size_t)(((s % world - shift + world) % world) * per + s / world) * RM_STRIP_ROWS + (y % RM_STRIP_ROWS);
    dst[(size_t)y * W + x] = src[srow * W + x];
}

// ---- RCCL, looked up at run time ---------------------------------------------

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    // optional (absent from old builds): a communicator per frame in flight, and the gather to one rank
    decltype(&ncclCommSplit) CommSplit = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

// The process's RCCL: the copy already loaded if there is one (a host such as PyTorch brings its
// own under the same name), the system's otherwise.
static const RcclApi *rccl_api(std::string &why) {
    static std::mutex mutex;
    static RcclApi api;
    static std::string error;
    std::lock_guard<std::mutex> lock(mutex);
    if (api.handle) return &api;
    if (!error.empty()) {
        why = error;
        return nullptr;
    }
    void *h = nullptr;
    // RM_RCCL_LIBRARY names the one file to take instead (a build of RCCL kept elsewhere; tests point it
    // at a file that does not exist to walk the failure path below)
    const char *named = getenv("RM_RCCL_LIBRARY");
    if (named && *named) {
        h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
    } else {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (h) break;
        }
        if (!h)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
    }
    if (!h) {
        const char *why_dl = dlerror();  // (one call: it hands the message out once and clears it)
        error = std::string("librccl.so could not be loaded: ") + (why_dl ? why_dl : "not found");
        why = error;
        return nullptr;
    }
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString) {
        error = "librccl.so lacks one of ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString";
        why = error;
        return nullptr;
    }
    a.CommSplit = reinterpret_cast<decltype(a.CommSplit)>(dlsym(h, "ncclCommSplit"));
    a.Gather = reinterpret_cast<decltype(a.Gather)>(dlsym(h, "ncclGather"));
    a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    api = a;
    return &api;
}

// ---- rm_group ----------------------------------------------------------------

constexpr int kGroupTimedFrames = 256;  // render-kernel timings kept (a ring)

struct GroupSlot {
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;            // the communicator this slot's frames travel through (RCCL groups)
    bool own_comm = false;                // split from the group's for this slot: destroyed with it
    unsigned char *d_strips = nullptr;    // this rank's strips, 4 bytes per pixel, compact
    unsigned char *d_wire = nullptr;      // the same as 3 bytes per pixel (packed exchange only)
    unsigned char *d_gathered = nullptr;  // every rank's strips as the all-gather leaves them
    unsigned char *d_gathered4 = nullptr;   // every rank's strips, 4 bytes per pixel, on the presenter of a frame (allocated on first use)
    unsigned char *d_unfiltered = nullptr;  // the gathered frame in row order ahead of FXAA (allocated on first use)
};

struct rm_group {
    rm_engine *e = nullptr;
    int rank = 0, world = 1, in_flight = 1;
    const RcclApi *api = nullptr;
    ncclComm_t comm = nullptr;
    int n_comms = 0;                   // communicators the frames use (0: custom exchange)
    rm_allgather_fn custom = nullptr;  // the host's exchange instead of RCCL
    void *custom_user = nullptr;
    rm_gather_fn custom_gather = nullptr;
    void *custom_gather_user = nullptr;
    int exchange = RM_EXCHANGE_ALL_GATHER;
    unsigned long long frames = 0;  // frames issued in the current exchange mode: frame f's presenter is rank f mod world
    unsigned long long issued = 0;  // frames issued since the group was made: which strips a rank renders rotates with it
    bool rotate = true;
    int strips_per_rank = 0;
    size_t band_pixels = 0;  // pixels of one rank's part (padded to whole strips)
    bool packed = false;     // 3 bytes per pixel on the wire
    int next_slot = 0;
    GroupSlot slot[RM_GROUP_MAX_FRAMES_IN_FLIGHT];
    // lightmap exchange: one rank's padded slice and the gathered whole, [per_patches * world][3] floats
    int per_patches = 0;
    float *d_slice = nullptr, *d_whole = nullptr;
    // what the group changed on its engine, to be put back when it goes
    bool retuned = false, saved_split_on = true;
    unsigned saved_dense_tiles = 0;
    // optional timing of the render kernel and of the exchange alone
    bool timing = false;
    int timed = 0;
    hipEvent_t ev0[kGroupTimedFrames] = {}, ev1[kGroupTimedFrames] = {};  // around the render launch
    hipEvent_t ex0[kGroupTimedFrames] = {}, ex1[kGroupTimedFrames] = {};  // around the collective
};

static int group_all_gather(rm_group *g, ncclComm_t comm, const void *d_send, void *d_recv, size_t count, bool floats, hipStream_t stream) {
    if (g->custom) {
        const int rc = g->custom(g->custom_user, d_send, d_recv, count * (floats ? sizeof(float) : 1), stream);
        if (rc) return fail(RM_ERR_HIP, "the host's all-gather function failed with " + std::to_string(rc));
        return RM_OK;
    }
    const ncclResult_t r = g->api->AllGather(d_send, d_recv, count, floats ? ncclFloat32 : ncclUint8, comm, stream);
    if (r != ncclSuccess) return fail(RM_ERR_HIP, std::string("ncclAllGather failed: ") + g->api->GetErrorString(r));
    return RM_OK;
}

// `count` bytes of every rank, rank 0's first, into `d_recv` on rank `root` (unused elsewhere).
static int group_gather_bytes(rm_group *g, ncclComm_t comm, const void *d_send, void *d_recv, size_t count, int root, hipStream_t stream) {
    if (g->custom) {
        int rc;
        if (g->custom_gather)
            rc = g->custom_gather(g->custom_gather_user, d_send, g->rank == root ? d_recv : nullptr, count, root, stream);
        else  // the host gave an all-gather only: every rank receives, the presenter alone goes on to use it
            rc = g->custom(g->custom_user, d_send, d_recv, count, stream);
        if (rc) return fail(RM_ERR_HIP, "the host's gather function failed with " + std::to_string(rc));
        return RM_OK;
    }
    const RcclApi *api = g->api;
    ncclResult_t r;
    if (api->Gather) {
        r = api->Gather(d_send, g->rank == root ? d_recv : nullptr, count, ncclUint8, root, comm, stream);
    } else if (api->Send && api->Recv && api->GroupStart && api->GroupEnd) {
        // the same as one group of point-to-point operations: every rank sends to the root, the root
        // receives every rank's part where it belongs
        r = api->GroupStart();
        if (r == ncclSuccess) r = api->Send(d_send, count, ncclUint8, root, comm, stream);
        if (g->rank == root)
            for (int p = 0; p < g->world && r == ncclSuccess; p++)
                r = api->Recv(static_cast<unsigned char *>(d_recv) + (size_t)p * count, count, ncclUint8, p, comm, stream);
        const ncclResult_t end = api->GroupEnd();
        if (r == ncclSuccess) r = end;
    } else {
        return fail(RM_ERR_UNSUPPORTED, "this librccl.so has neither ncclGather nor ncclSend/ncclRecv");
    }
    if (r != ncclSuccess) return fail(RM_ERR_HIP, std::string("the gather to rank ") + std::to_string(root) + " failed: " + api->GetErrorString(r));
    return RM_OK;
}

static void group_free(rm_group *g) {
    if (!g) return;
    rm_engine *e = g->e;
    (void)hipSetDevice(e->device);
    for (GroupSlot &s : g->slot) {
        if (s.stream) {
            (void)hipStreamSynchronize(s.stream);
            auto &att = e->attached_streams;
            att.erase(std::remove(att.begin(), att.end(), s.stream), att.end());
            // the launch plan the stream's frames had goes with it (a later stream may get the same handle)
            for (unsigned i = 0; i < rm_engine::kPlans; i++)
                if (e->plan_taken[i] && e->plan_stream[i] == s.stream) {
                    e->plan_taken[i] = false;
                    e->plan_stream[i] = nullptr;
                    e->plan_launches[i] = 0;
                    e->plan_key[i] = rm_engine::PlanKey();
                    if (e->last_plan_count == e->plan_count_word[i]) e->last_plan_count = nullptr;
                    e->plan_count_word[i] = nullptr;
                }
        }
        if (s.own_comm && s.comm) (void)g->api->CommDestroy(s.comm);
        (void)hipFree(s.d_strips);
        (void)hipFree(s.d_wire);
        (void)hipFree(s.d_gathered);
        (void)hipFree(s.d_gathered4);
        (void)hipFree(s.d_unfiltered);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    (void)hipFree(g->d_slice);
    (void)hipFree(g->d_whole);
    for (int i = 0; i < kGroupTimedFrames; i++) {
        if (g->ev0[i]) (void)hipEventDestroy(g->ev0[i]);
        if (g->ev1[i]) (void)hipEventDestroy(g->ev1[i]);
        if (g->ex0[i]) (void)hipEventDestroy(g->ex0[i]);
        if (g->ex1[i]) (void)hipEventDestroy(g->ex1[i]);
    }
    if (g->comm) (void)g->api->CommDestroy(g->comm);
    if (g->retuned) {
        e->split_on = g->saved_split_on;
        e->dense_tiles = g->saved_dense_tiles;
    }
    delete g;
}

// How the engine launches frames while the group lives (put back by group_free): applied once the
// group stands, so that a creation that fails leaves the engine as it was.
// With several frames in flight the machine is busy while a frame's slow tiles finish, and the
// quarter waves of rm_engine_set_tile_splitting cost throughput - unless a rank's share is so
// small that even three of them leave the machine waiting for their slow tiles (measured on
// shares of the 3840x2160 frame, two or three in flight: a half and a quarter - 64800 and
// 32400 tiles - are 10 % slower with the plans, an eighth - 16200 tiles - 6-23 % faster).
static void group_retune_engine(rm_group *g) {
    rm_engine *e = g->e;
    g->saved_split_on = e->split_on;
    g->saved_dense_tiles = e->dense_tiles;
    g->retuned = true;
    const long share_tiles = (long)g->strips_per_rank * ((e->W + TILE - 1) / TILE);
    if (g->in_flight > 1 && share_tiles > 20000) e->split_on = false;
    if (g->in_flight > 1 && !getenv("RM_DENSE_TILES")) e->dense_tiles = g->in_flight >= 3 ? kOverlapDenseTiles3 : kOverlapDenseTiles2;
}

static rm_group *group_build(rm_engine *e, int rank, int world, int frames_in_flight) {
    if (!e || world < 1 || rank < 0 || rank >= world || frames_in_flight < 1 || frames_in_flight > RM_GROUP_MAX_FRAMES_IN_FLIGHT) {
        fail(RM_ERR_INVALID, "rm_group: bad rank, world or frames in flight");
        return nullptr;
    }
    if (hipSetDevice(e->device) != hipSuccess) {
        fail(RM_ERR_HIP, "hipSetDevice failed");
        return nullptr;
    }
    rm_group *g = new rm_group();
    g->e = e;
    g->rank = rank;
    g->world = world;
    g->in_flight = frames_in_flight;
    const int strips = (e->H + RM_STRIP_ROWS - 1) / RM_STRIP_ROWS;
    g->strips_per_rank = (strips + world - 1) / world;
    if (const char *v = getenv("RM_GROUP_ROTATE")) g->rotate = atoi(v) != 0;  // (the same on every rank; 0 = rank r always renders strips r, r + world, ...)
    g->band_pixels = (size_t)g->strips_per_rank * RM_STRIP_ROWS * e->W;
    g->packed = e->W % 4 == 0 && e->H <= 65535;
    const size_t wire = g->packed ? 3 : 4;
    hipError_t err = hipSuccess;
    for (int i = 0; i < frames_in_flight && err == hipSuccess; i++) {
        GroupSlot &s = g->slot[i];
        err = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (err == hipSuccess) e->attached_streams.push_back(s.stream);  // settings and lightmaps wait for these before they change
        if (err == hipSuccess) err = dev_alloc(&s.d_strips, g->band_pixels * 4);
        if (err == hipSuccess) err = hipMemset(s.d_strips, 0, g->band_pixels * 4);
        if (err == hipSuccess && g->packed) err = dev_alloc(&s.d_wire, g->band_pixels * 3);
        if (err == hipSuccess) err = dev_alloc(&s.d_gathered, g->band_pixels * wire * world);
    }
    g->per_patches = (e->n_patches + world - 1) / world;
    if (err == hipSuccess) err = dev_alloc(&g->d_slice, (size_t)g->per_patches * 3);
    if (err == hipSuccess) err = dev_alloc(&g->d_whole, (size_t)g->per_patches * 3 * world);
    if (err != hipSuccess) {
        fail(RM_ERR_HIP, std::string("rm_group: ") + hipGetErrorString(err));
        group_free(g);
        return nullptr;
    }
    return g;
}

// This rank's patches [p0, p1) of the lightmap precompute.
static void group_patch_range(const rm_group *g, int &p0, int &p1) {
    p0 = std::min(g->rank * g->per_patches, g->e->n_patches);
    p1 = std::min(p0 + g->per_patches, g->e->n_patches);
}

// All-gathers the ranks' slices [p0, p1) of the [n_patches][3] array `d_src` into the whole of `d_dst`
// (the same array, or another of the same shape).
static int group_gather_patches(rm_group *g, const float *d_src, float *d_dst) {
    rm_engine *e = g->e;
    int p0, p1;
    group_patch_range(g, p0, p1);
    const size_t per = (size_t)g->per_patches * 3;
    if (per == 0) return RM_OK;
    // Equal slices that end up in another array travel as they lie: from this rank's part of the source
    // straight into the destination, no staging copies.
    if ((size_t)g->per_patches * g->world == (size_t)e->n_patches && d_src != d_dst)
        return group_all_gather(g, g->comm, d_src + (size_t)p0 * 3, d_dst, per, true, e->stream);
    // otherwise slices are equal parts on the wire; the tail of the last ranks' is padding
    HIP_TRY(hipMemsetAsync(g->d_slice, 0, per * sizeof(float), e->stream));
    if (p1 > p0)
        HIP_TRY(hipMemcpyAsync(g->d_slice, d_src + (size_t)p0 * 3, (size_t)(p1 - p0) * 3 * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    if (int rc = group_all_gather(g, g->comm, g->d_slice, g->d_whole, per, true, e->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, g->d_whole, (size_t)e->n_patches * 3 * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    return RM_OK;
}

extern "C" {

int rm_group_unique_id(uint8_t id[RM_GROUP_ID_BYTES]) {
    static_assert(RM_GROUP_ID_BYTES == sizeof(ncclUniqueId), "the id travels as RCCL's unique id");
    if (!id) return fail(RM_ERR_INVALID, "NULL id");
    std::string why;
    const RcclApi *api = rccl_api(why);
    if (!api) return fail(RM_ERR_UNSUPPORTED, why);
    ncclUniqueId uid;
    const ncclResult_t r = api->GetUniqueId(&uid);
    if (r != ncclSuccess) return fail(RM_ERR_HIP, std::string("ncclGetUniqueId failed: ") + api->GetErrorString(r));
    memcpy(id, &uid, sizeof(uid));
    return RM_OK;
}

rm_group *rm_group_create(rm_engine *engine, const uint8_t id[RM_GROUP_ID_BYTES], int rank, int world, int frames_in_flight) {
    if (!id) {
        fail(RM_ERR_INVALID, "NULL id");
        return nullptr;
    }
    std::string why;
    const RcclApi *api = rccl_api(why);
    if (!api) {
        fail(RM_ERR_UNSUPPORTED, why);
        return nullptr;
    }
    rm_group *g = group_build(engine, rank, world, frames_in_flight);
    if (!g) return nullptr;
    g->api = api;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    const ncclResult_t r = api->CommInitRank(&g->comm, world, uid, rank);  // collective: every rank of the group calls it
    if (r != ncclSuccess) {
        g->comm = nullptr;
        fail(RM_ERR_HIP, std::string("ncclCommInitRank failed: ") + api->GetErrorString(r));
        group_free(g);
        return nullptr;
    }
    // A communicator per frame in flight.  RCCL orders what is launched through one communicator, on
    // whatever stream: with one for all, frame f+1's exchange would start only after frame f's has
    // finished, although the two are on streams of their own - and the gathers of consecutive frames
    // to different presenters, which use different links, could never overlap.  Every rank issues its
    // frames in the same order, so the communicators are used in the same order everywhere.
    // (ncclCommSplit is collective over the parent, like the call above.  RM_GROUP_SHARED_COMM=1, the
    // same on every rank, keeps the one communicator.)
    g->n_comms = 1;
    g->slot[0].comm = g->comm;
    const char *shared = getenv("RM_GROUP_SHARED_COMM");
    const bool split = api->CommSplit && !(shared && atoi(shared) != 0);
    for (int i = 1; i < frames_in_flight; i++) {
        g->slot[i].comm = g->comm;
        if (!split) continue;
        ncclComm_t c = nullptr;
        const ncclResult_t rs = api->CommSplit(g->comm, 0, rank, &c, nullptr);
        if (rs != ncclSuccess || !c) {
            fail(RM_ERR_HIP, std::string("ncclCommSplit failed: ") + api->GetErrorString(rs));
            group_free(g);
            return nullptr;
        }
        g->slot[i].comm = c;
        g->slot[i].own_comm = true;
        g->n_comms++;
    }
    group_retune_engine(g);
    return g;
}

rm_group *rm_group_create_custom(rm_engine *engine, int rank, int world, int frames_in_flight, rm_allgather_fn all_gather, void *user) {
    if (!all_gather) {
        fail(RM_ERR_INVALID, "NULL all-gather function");
        return nullptr;
    }
    rm_group *g = group_build(engine, rank, world, frames_in_flight);
    if (!g) return nullptr;
    g->custom = all_gather;
    g->custom_user = user;
    group_retune_engine(g);
    return g;
}

void rm_group_destroy(rm_group *group) { group_free(group); }

int rm_group_info(const rm_group *g, int *rank, int *world, int *strips_per_rank, int *bytes_per_pixel_exchanged) {
    if (!g) return fail(RM_ERR_INVALID, "NULL group");
    if (rank) *rank = g->rank;
    if (world) *world = g->world;
    if (strips_per_rank) *strips_per_rank = g->strips_per_rank;
    if (bytes_per_pixel_exchanged) *bytes_per_pixel_exchanged = g->packed ? 3 : 4;
    return RM_OK;
}

int rm_group_set_exchange(rm_group *g, int mode) {
    if (!g || (mode != RM_EXCHANGE_ALL_GATHER && mode != RM_EXCHANGE_PRESENTER)) return fail(RM_ERR_INVALID, "rm_group_set_exchange: unknown mode");
    if (int rc = rm_group_wait(g)) return rc;
    g->exchange = mode;
    g->frames = 0;
    return RM_OK;
}

int rm_group_next_presenter(const rm_group *g, int *rank) {
    if (!g || !rank) return fail(RM_ERR_INVALID, "NULL argument");
    *rank = g->exchange == RM_EXCHANGE_PRESENTER ? (int)(g->frames % (unsigned long long)g->world) : -1;
    return RM_OK;
}

int rm_group_set_custom_gather(rm_group *g, rm_gather_fn gather, void *user) {
    if (!g) return fail(RM_ERR_INVALID, "NULL group");
    if (!g->custom) return fail(RM_ERR_INVALID, "rm_group_set_custom_gather: the group exchanges through RCCL");
    g->custom_gather = gather;
    g->custom_gather_user = user;
    return RM_OK;
}

int rm_group_communicators(const rm_group *g, int *count) {
    if (!g || !count) return fail(RM_ERR_INVALID, "NULL argument");
    *count = g->n_comms;
    return RM_OK;
}

int rm_group_frame(rm_group *g, void *d_bgra_frame, int fxaa, int32_t mouse_x, int32_t mouse_y, const float rel_move[3], int32_t scroll) {
    if (!g) return fail(RM_ERR_INVALID, "NULL argument");
    const int presenter = g->exchange == RM_EXCHANGE_PRESENTER ? (int)(g->frames % (unsigned long long)g->world) : -1;
    const bool lands_here = presenter < 0 || presenter == g->rank;
    if (lands_here && !d_bgra_frame) return fail(RM_ERR_INVALID, "NULL frame on a rank the frame lands on");
    rm_engine *e = g->e;
    int rc = rm_engine_input(e, mouse_x, mouse_y, rel_move, scroll);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    GroupSlot &s = g->slot[g->next_slot];
    if (fxaa && lands_here && !s.d_unfiltered) HIP_TRY(dev_alloc(&s.d_unfiltered, (size_t)e->W * e->H * 4));
    // (a host that supplied an all-gather only makes every rank receive)
    const bool receives = presenter == g->rank || (presenter >= 0 && g->custom && !g->custom_gather);
    if (receives && !s.d_gathered4) HIP_TRY(dev_alloc(&s.d_gathered4, g->band_pixels * 4 * g->world));
    g->next_slot = (g->next_slot + 1) % g->in_flight;
    g->frames++;
    // Which strips a rank renders rotates from frame to frame: in frame i rank r takes the strips
    // (r + i) mod world, (r + i) mod world + world, ...  A rank's pace with k frames in flight is bounded
    // below by the latency of its slowest tiles over k; the slowest tiles of a view (two mirrors that
    // show each other) lie in a few strips, and whoever holds those in every frame sets the pace of the
    // whole group.  Rotated, every rank holds them in one frame in `world`, beside two cheaper ones.
    const int shift = g->rotate ? (int)(g->issued % (unsigned long long)g->world) : 0;
    const int role = (g->rank + shift) % g->world;
    g->issued++;
    // Everything of this frame goes to the slot's stream, in order; the frame that used the slot's
    // buffers before it is ahead of it on the same stream.
    hipStream_t own = e->stream;
    e->stream = s.stream;
    const auto enqueue = [&]() -> int {
        const int t = g->timed % kGroupTimedFrames;
        if (g->timing) {
            if (!g->ev0[t]) {
                HIP_TRY(hipEventCreate(&g->ev0[t]));
                HIP_TRY(hipEventCreate(&g->ev1[t]));
                HIP_TRY(hipEventCreate(&g->ex0[t]));
                HIP_TRY(hipEventCreate(&g->ex1[t]));
            }
            HIP_TRY(hipEventRecord(g->ev0[t], s.stream));
        }
        if (int r = rm_engine_render_strips_device(e, s.d_strips, (size_t)e->W * 4, role, g->world, g->strips_per_rank)) return r;
        if (g->timing) HIP_TRY(hipEventRecord(g->ev1[t], s.stream));
        void *frame = fxaa ? s.d_unfiltered : static_cast<unsigned char *>(d_bgra_frame);
        const auto reorder = [&](const unsigned char *gathered) {
            hipLaunchKernelGGL(reorder_strips_kernel, dim3((e->W + 255) / 256, e->H), dim3(256), 0, s.stream,
                               reinterpret_cast<const unsigned *>(gathered), static_cast<unsigned *>(frame), e->W, e->H, g->world,
                               g->strips_per_rank, shift);
        };
        if (presenter >= 0) {
            // the strips travel as rendered, 4 bytes per pixel: a link carries one frame in `world`, so the
            // packing pass would cost every rank more time than the fourth byte costs the links
            if (g->timing) HIP_TRY(hipEventRecord(g->ex0[t], s.stream));
            if (int r = group_gather_bytes(g, s.comm, s.d_strips, s.d_gathered4, g->band_pixels * 4, presenter, s.stream)) return r;
            if (g->timing) HIP_TRY(hipEventRecord(g->ex1[t], s.stream));
            if (lands_here) {
                reorder(s.d_gathered4);
                HIP_TRY(hipGetLastError());
            }
        } else if (g->packed) {
            // byte 3 of every rendered pixel is 0 (engine.rs:116-126): strips cross the links as B,G,R triples
            if (int r = rm_engine_pack_bgr_device(e, s.d_strips, s.d_wire, g->band_pixels)) return r;
            if (g->timing) HIP_TRY(hipEventRecord(g->ex0[t], s.stream));
            if (int r = group_all_gather(g, s.comm, s.d_wire, s.d_gathered, g->band_pixels * 3, false, s.stream)) return r;
            if (g->timing) HIP_TRY(hipEventRecord(g->ex1[t], s.stream));
            if (int r = unpack_bgr_strips(e, s.d_gathered, frame, g->world, g->strips_per_rank, shift)) return r;
        } else {
            if (g->timing) HIP_TRY(hipEventRecord(g->ex0[t], s.stream));
            if (int r = group_all_gather(g, s.comm, s.d_strips, s.d_gathered, g->band_pixels * 4, false, s.stream)) return r;
            if (g->timing) HIP_TRY(hipEventRecord(g->ex1[t], s.stream));
            reorder(s.d_gathered);
            HIP_TRY(hipGetLastError());
        }
        if (g->timing) g->timed++;
        // FXAA reads up to four rows below and one above a pixel (fxaa.txt:7-8,80-97): it runs on the
        // gathered frame, so the strips need no halo and stay interleaved (even load)
        if (fxaa && lands_here)
            if (int r = rm_engine_fxaa_device(e, s.d_unfiltered, d_bgra_frame, 0, e->H)) return r;
        return RM_OK;
    };
    rc = enqueue();
    e->stream = own;
    return rc;
}

int rm_group_wait(rm_group *g) {
    if (!g) return fail(RM_ERR_INVALID, "NULL group");
    HIP_TRY(hipSetDevice(g->e->device));
    for (int i = 0; i < g->in_flight; i++) HIP_TRY(hipStreamSynchronize(g->slot[i].stream));
    return RM_OK;
}

int rm_group_set_timing(rm_group *g, int on) {
    if (!g) return fail(RM_ERR_INVALID, "NULL group");
    g->timing = on != 0;
    g->timed = 0;
    return RM_OK;
}

static int group_mean_ms(rm_group *g, hipEvent_t *from, hipEvent_t *to, double *mean_ms, int *frames) {
    if (!g || !mean_ms) return fail(RM_ERR_INVALID, "NULL argument");
    if (int rc = rm_group_wait(g)) return rc;
    const int n = std::min(g->timed, kGroupTimedFrames);
    double sum = 0.0;
    for (int i = 0; i < n; i++) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, from[i], to[i]));
        sum += ms;
    }
    *mean_ms = n ? sum / n : 0.0;
    if (frames) *frames = n;
    return RM_OK;
}
int rm_group_render_time(rm_group *g, double *mean_ms, int *frames) { return group_mean_ms(g, g ? g->ev0 : nullptr, g ? g->ev1 : nullptr, mean_ms, frames); }
int rm_group_exchange_time(rm_group *g, double *mean_ms, int *frames) { return group_mean_ms(g, g ? g->ex0 : nullptr, g ? g->ex1 : nullptr, mean_ms, frames); }

// Engine::compute_lightmaps (engine.rs:246-326) over the group: every rank builds the clouds and the
// direct lighting (cheap), computes the form factors of its own receiver patches once, and per bounce
// its slice of the next emissive map; one all-gather per bounce exchanges the slices (a bounce reads
// only the previous bounce's values, engine.rs:299-325), one more at the end the lightmaps.
int rm_group_compute_lightmaps(rm_group *g, int bounces) {
    if (!g || bounces < 0) return fail(RM_ERR_INVALID, "bad argument");
    rm_engine *e = g->e;
    if (int rc = rm_group_wait(g)) return rc;  // frames in flight sample the lightmaps
    int rc = rm_engine_radiosity_begin(e);
    if (rc) return rc;
    int p0, p1;
    group_patch_range(g, p0, p1);
    if (bounces > 0) {
        rc = rm_engine_radiosity_form_factors(e, p0, p1);
        if (rc) return rc;
    }
    for (int b = 0; b < bounces; b++) {
        rc = rm_engine_radiosity_bounce(e, p0, p1);
        if (rc) return rc;
        // the gathered slices are the next bounce's emissive map: they land in it directly
        // (rm_engine_radiosity_commit, the copy from the one map to the other, has nothing left to do)
        rc = group_gather_patches(g, e->rad.emissive_next, e->rad.emissive);
        if (rc) return rc;
    }
    if (bounces > 0) {
        rc = group_gather_patches(g, e->d_lightmaps, e->d_lightmaps);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RM_OK;
}

}  // extern "C"

// ---- rm_multi: one process, several GPUs, frames into host memory ---------------

struct rm_multi {
    int n = 0;
    int W = 0, H = 0;
    int strips_per_rank = 0;
    std::vector<rm_engine *> engines;
    std::vector<hipStream_t> streams;           // one per engine: its renders and copies
    std::vector<unsigned char *> d_strips;      // per engine: its strips, compact
    unsigned char *h_stage = nullptr;           // page-locked frame for pageable destinations
    float *h_patches = nullptr;                 // page-locked [n_patches][3]: the lightmap exchange goes through the host
    int per_patches = 0;
    // one host thread per GPU, so that the launches and copies of the GPUs are issued side by side
    std::vector<std::thread> threads;
    std::mutex mutex;
    std::condition_variable go, done;
    unsigned long long epoch = 0;
    int pending = 0;
    bool quit = false;
    std::function<int(int)> task;
    std::vector<int> rc;
    std::vector<std::string> error;
};

static void multi_worker(rm_multi *m, int i) {
    (void)hipSetDevice(m->engines[i]->device);
    unsigned long long seen = 0;
    for (;;) {
        std::function<int(int)> task;
        {
            std::unique_lock<std::mutex> lock(m->mutex);
            m->go.wait(lock, [&] { return m->quit || m->epoch != seen; });
            if (m->quit) return;
            seen = m->epoch;
            task = m->task;
        }
        const int rc = task(i);
        {
            std::lock_guard<std::mutex> lock(m->mutex);
            m->rc[i] = rc;
            if (rc) m->error[i] = g_error;  // the message is per thread: hand it to the caller's
            if (--m->pending == 0) m->done.notify_one();
        }
    }
}

// Runs task(i) on every GPU's thread and waits for all of them; the first failure is reported.
static int multi_run(rm_multi *m, std::function<int(int)> task) {
    std::unique_lock<std::mutex> lock(m->mutex);
    m->task = std::move(task);
    m->pending = m->n;
    m->epoch++;
    m->go.notify_all();
    m->done.wait(lock, [&] { return m->pending == 0; });
    for (int i = 0; i < m->n; i++)
        if (m->rc[i]) return fail(m->rc[i], "GPU " + std::to_string(m->engines[i]->device) + ": " + m->error[i]);
    return RM_OK;
}

static void multi_free(rm_multi *m) {
    if (!m) return;
    {
        std::lock_guard<std::mutex> lock(m->mutex);
        m->quit = true;
        m->go.notify_all();
    }
    for (std::thread &t : m->threads)
        if (t.joinable()) t.join();
    for (int i = 0; i < (int)m->engines.size(); i++) {
        if (!m->engines[i]) continue;
        (void)hipSetDevice(m->engines[i]->device);
        if (i < (int)m->streams.size() && m->streams[i]) {
            (void)hipStreamSynchronize(m->streams[i]);
            m->engines[i]->stream = nullptr;
            (void)hipStreamDestroy(m->streams[i]);
        }
        if (i < (int)m->d_strips.size()) (void)hipFree(m->d_strips[i]);
        rm_engine_destroy(m->engines[i]);
    }
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    if (m->h_patches) (void)hipHostFree(m->h_patches);
    delete m;
}

extern "C" {

rm_multi *rm_multi_create(const rm_scene *scene, int width, int height, int map_size, const int *devices, int n_devices) {
    if (!devices || n_devices < 1 || n_devices > 64) {
        fail(RM_ERR_INVALID, "rm_multi_create: 1 to 64 devices");
        return nullptr;
    }
    rm_multi *m = new rm_multi();
    m->n = n_devices;
    m->W = width;
    m->H = height;
    m->engines.assign(n_devices, nullptr);
    m->streams.assign(n_devices, nullptr);
    m->d_strips.assign(n_devices, nullptr);
    m->rc.assign(n_devices, 0);
    m->error.assign(n_devices, std::string());
    for (int i = 0; i < n_devices; i++) {
        m->engines[i] = rm_engine_create(scene, width, height, map_size, devices[i]);  // (sets the device)
        if (!m->engines[i]) {
            multi_free(m);
            return nullptr;
        }
        const int strips = (height + RM_STRIP_ROWS - 1) / RM_STRIP_ROWS;
        m->strips_per_rank = (strips + n_devices - 1) / n_devices;
        hipError_t err = hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking);
        if (err == hipSuccess) err = dev_alloc(&m->d_strips[i], (size_t)m->strips_per_rank * RM_STRIP_ROWS * width * 4);
        if (err != hipSuccess) {
            fail(RM_ERR_HIP, std::string("rm_multi_create: ") + hipGetErrorString(err));
            multi_free(m);
            return nullptr;
        }
        m->engines[i]->stream = m->streams[i];
    }
    const int NP = m->engines[0]->n_patches;
    m->per_patches = (NP + n_devices - 1) / n_devices;
    if (hipHostMalloc(reinterpret_cast<void **>(&m->h_patches), std::max<size_t>((size_t)NP * 3 * sizeof(float), 4), hipHostMallocPortable) != hipSuccess) {
        fail(RM_ERR_HIP, "rm_multi_create: hipHostMalloc failed");
        multi_free(m);
        return nullptr;
    }
    for (int i = 0; i < n_devices; i++) m->threads.emplace_back(multi_worker, m, i);
    return m;
}

void rm_multi_destroy(rm_multi *multi) { multi_free(multi); }

int rm_multi_device_count(const rm_multi *m) { return m ? m->n : 0; }

rm_engine *rm_multi_engine(rm_multi *m, int index) {
    if (!m || index < 0 || index >= m->n) {
        fail(RM_ERR_INVALID, "rm_multi_engine: index out of range");
        return nullptr;
    }
    return m->engines[index];
}

// Engine::compute_lightmaps (engine.rs:246-326), receiver patches split over the GPUs.  Jacobi
// bounces (engine.rs:299-325): a bounce reads only the previous one's emissive values, so the GPUs
// exchange their slices once per bounce - through page-locked host memory, which every GPU reaches
// over its own PCIe link (the slices are a few KiB to MiB; the form factors are the work).
int rm_multi_compute_lightmaps(rm_multi *m, int bounces) {
    if (!m || bounces < 0) return fail(RM_ERR_INVALID, "bad argument");
    const int NP = m->engines[0]->n_patches;
    const auto range = [m, NP](int i, int &p0, int &p1) {
        p0 = std::min(i * m->per_patches, NP);
        p1 = std::min(p0 + m->per_patches, NP);
    };
    // every GPU's slice of `which` into the host array, then (after all have arrived) the whole array back
    const auto slice_out = [m, range](int i, float *d_values) -> int {
        rm_engine *e = m->engines[i];
        int p0, p1;
        range(i, p0, p1);
        if (p1 > p0)
            HIP_TRY(hipMemcpyAsync(m->h_patches + (size_t)p0 * 3, d_values + (size_t)p0 * 3, (size_t)(p1 - p0) * 3 * sizeof(float),
                                   hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        return RM_OK;
    };
    const auto whole_in = [m, NP](int i, float *d_values) -> int {
        rm_engine *e = m->engines[i];
        if (NP) HIP_TRY(hipMemcpyAsync(d_values, m->h_patches, (size_t)NP * 3 * sizeof(float), hipMemcpyHostToDevice, e->stream));
        return RM_OK;
    };
    int rc = multi_run(m, [&](int i) -> int {
        rm_engine *e = m->engines[i];
        if (int r = rm_engine_radiosity_begin(e)) return r;
        int p0, p1;
        range(i, p0, p1);
        return bounces > 0 ? rm_engine_radiosity_form_factors(e, p0, p1) : RM_OK;
    });
    for (int b = 0; b < bounces && !rc; b++) {
        rc = multi_run(m, [&](int i) -> int {
            rm_engine *e = m->engines[i];
            int p0, p1;
            range(i, p0, p1);
            if (int r = rm_engine_radiosity_bounce(e, p0, p1)) return r;
            return slice_out(i, e->rad.emissive_next);
        });
        if (rc) break;
        rc = multi_run(m, [&](int i) -> int {
            rm_engine *e = m->engines[i];
            if (int r = whole_in(i, e->rad.emissive_next)) return r;
            if (int r = rm_engine_radiosity_commit(e)) return r;
            HIP_TRY(hipStreamSynchronize(e->stream));  // the host array is rewritten by the next bounce
            return RM_OK;
        });
    }
    if (!rc && bounces > 0) {
        rc = multi_run(m, [&](int i) -> int { return slice_out(i, m->engines[i]->d_lightmaps); });
        if (!rc)
            rc = multi_run(m, [&](int i) -> int {
                if (int r = whole_in(i, m->engines[i]->d_lightmaps)) return r;
                HIP_TRY(hipStreamSynchronize(m->engines[i]->stream));
                return RM_OK;
            });
    } else if (!rc) {
        rc = multi_run(m, [&](int i) -> int {
            HIP_TRY(hipStreamSynchronize(m->engines[i]->stream));
            return RM_OK;
        });
    }
    return rc;
}

// Engine::render (engine.rs:38-127) over the GPUs: GPU i of n renders the 8-row strips i, i+n, ...
// and copies them over its own PCIe link into the rows of `bgra` they belong to.
int rm_multi_render(rm_multi *m, uint8_t *bgra, size_t pitch_bytes, int32_t mouse_x, int32_t mouse_y, const float rel_move[3],
                    int32_t scroll) {
    if (!m || !bgra || !rel_move) return fail(RM_ERR_INVALID, "NULL argument");
    const size_t row_bytes = (size_t)m->W * 4;
    if (pitch_bytes < row_bytes) return fail(RM_ERR_INVALID, "pitch must be >= width*4");
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, bgra) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();  // an unregistered pointer reports an error: not a failure here
    if (!pinned && !m->h_stage)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&m->h_stage), row_bytes * m->H, hipHostMallocPortable));
    const float move[3] = {rel_move[0], rel_move[1], rel_move[2]};
    return multi_run(m, [=](int i) -> int {
        rm_engine *e = m->engines[i];
        // every engine takes the same input, so every camera follows the same path (engine.rs:72)
        if (int r = rm_engine_input(e, mouse_x, mouse_y, move, scroll)) return r;
        if (int r = rm_engine_render_strips_device(e, m->d_strips[i], row_bytes, i, m->n, m->strips_per_rank)) return r;
        const size_t strip_bytes = row_bytes * RM_STRIP_ROWS;
        uint8_t *dst = pinned ? bgra : m->h_stage;
        const size_t dpitch = pinned ? pitch_bytes : row_bytes;
        int whole = 0;  // this GPU's strips that lie in the frame with all their rows
        while (whole < m->strips_per_rank && ((i + whole * m->n) + 1) * RM_STRIP_ROWS <= m->H) whole++;
        if (whole > 0) {
            if (dpitch == row_bytes) {
                // a strip is one contiguous run of bytes on both sides: all strips in one strided copy
                HIP_TRY(hipMemcpy2DAsync(dst + (size_t)i * strip_bytes, strip_bytes * m->n, m->d_strips[i], strip_bytes, strip_bytes, whole,
                                         hipMemcpyDeviceToHost, e->stream));
            } else {
                for (int k = 0; k < whole; k++)
                    HIP_TRY(hipMemcpy2DAsync(dst + (size_t)(i + k * m->n) * RM_STRIP_ROWS * dpitch, dpitch, m->d_strips[i] + (size_t)k * strip_bytes,
                                             row_bytes, row_bytes, RM_STRIP_ROWS, hipMemcpyDeviceToHost, e->stream));
            }
        }
        const int last_row = (i + whole * m->n) * RM_STRIP_ROWS;  // a last strip cut by the frame's edge
        if (whole < m->strips_per_rank && last_row < m->H)
            HIP_TRY(hipMemcpy2DAsync(dst + (size_t)last_row * dpitch, dpitch, m->d_strips[i] + (size_t)whole * strip_bytes, row_bytes, row_bytes,
                                     m->H - last_row, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (!pinned)  // each GPU's thread moves its own strips out of the staging frame
            for (int k = 0; k < m->strips_per_rank; k++) {
                const int y0 = (i + k * m->n) * RM_STRIP_ROWS, y1 = std::min(y0 + RM_STRIP_ROWS, m->H);
                for (int y = y0; y < y1; y++) memcpy(bgra + (size_t)y * pitch_bytes, m->h_stage + (size_t)y * row_bytes, row_bytes);
            }
        return RM_OK;
    });
}

}  // extern "C"
