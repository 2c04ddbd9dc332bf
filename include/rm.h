/*
 * rm.h - C ABI of the MI355X ray-marching hot path.
 *
 * The reference renderer (joshua-jose/RayMarching) has no FFI: its seam is the
 * Rust-internal call from main.rs into `Engine`.  Every entry point below names
 * the reference interface it stands in for (file:line in the reference tree).
 * All functions return 0 on success and a non-zero rm_status otherwise (the
 * reference aborts instead); rm_last_error() describes the last failure on the
 * calling thread.  Nothing here aborts the process.
 *
 * Threading: one engine is driven by one host thread at a time (the reference
 * calls Engine from its main thread only, main.rs:141-145).
 *
 * Pixel format: 4 bytes per pixel, byte0 = blue, byte1 = green, byte2 = red,
 * byte3 = 0, rows top to bottom - what the reference writes into its SDL
 * ARGB8888 streaming texture (engine.rs:116-126, main.rs:56).
 */
#ifndef RM_H
#define RM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rm_scene rm_scene;
typedef struct rm_engine rm_engine;

/* material.rs:2-9 */
typedef struct rm_material {
    float ambient;
    float diffuse;
    float specular;
    float shininess;
    float reflectivity;
    float emissive;
} rm_material;

typedef enum rm_status {
    RM_OK = 0,
    RM_ERR_INVALID = 1,     /* bad argument */
    RM_ERR_HIP = 2,         /* a HIP runtime call failed */
    RM_ERR_UNSUPPORTED = 3, /* the reference itself aborts on this input */
    RM_ERR_LIMIT = 4        /* a documented size limit was exceeded */
} rm_status;

/* Indices into the array filled by rm_engine_get_stats(). */
enum {
    RM_STAT_DEPTH_CAP_HITS = 0, /* reflection chains cut at RM_MAX_REFLECTION_DEPTH */
    RM_STAT_STEP_CAP_HITS = 1,  /* marches cut by the optional iteration cap */
    RM_STAT_LIGHTMAP_OOB = 2,   /* lightmap fetches the reference would abort on */
    RM_STAT_TEXTURE_OOB = 3,
    RM_STAT_COUNT = 8
};

#define RM_MAX_REFLECTION_DEPTH 16
#define RM_MAX_OBJECTS 1024

const char *rm_last_error(void);
int rm_version(void);
/* Number of HIP devices visible to the process. */
int rm_device_count(int *count);

/* Page-locked host memory for frame buffers: rm_engine_render copies into such a
 * buffer directly at PCIe rate (the reference's destination is the SDL streaming
 * texture it locks per frame, main.rs:141-145; pageable memory works too, through
 * an internal staging copy). */
void *rm_host_alloc(size_t bytes);
void rm_host_free(void *ptr);

/* ---- scene construction: main.rs:62-71,154-193 ------------------------- */

/* `Vec<Box<dyn EngineObject>>` under construction (main.rs:154-193). */
rm_scene *rm_scene_create(void);
void rm_scene_destroy(rm_scene *scene);

/* Texture::new (texture.rs:13-29) after the file decode: `rgb_topdown` is the
 * decoded RGB8 image, rows top to bottom.  Returns the texture id (>= 0) or a
 * negative rm_status. */
int rm_texture_create_rgb8(rm_scene *scene, const uint8_t *rgb_topdown, int width, int height, float uscale,
                           float vscale);

/* Object constructors.  Objects keep insertion order; each returns the object
 * index (>= 0) or a negative rm_status.
 * Sphere{position,radius,material,colour,..}   objects.rs:158-165
 * XPlane::new / YPlane::new / ZPlane::new       objects.rs:224-257
 * Plane{normal,distance,..}                     objects.rs:167-175 */
int rm_scene_add_sphere(rm_scene *scene, const float position[3], float radius, const rm_material *material,
                        const float colour[3]);
int rm_scene_add_xplane(rm_scene *scene, float x, float dir, const rm_material *material, const float colour[3]);
int rm_scene_add_yplane(rm_scene *scene, float y, float dir, const rm_material *material, const float colour[3],
                        int texture);
int rm_scene_add_zplane(rm_scene *scene, float z, float dir, const rm_material *material, const float colour[3]);
/* The reference gives a generic Plane a lightmap (plane_funcs!, objects.rs:216) but no way to address
 * it: get_sample_pos and sample_uv_from_pos are unimplemented!() (objects.rs:39-42), so lighting or
 * shading one aborts.  Defined here (an extension, the same in the CPU oracle): the lightmap lies over
 * two tangent directions through the plane's point nearest the origin, origin = normal * distance, with
 * a = the axis of the largest |normal component|, e1 < e2 the other two axes,
 * t = normalized(unit(e1) - normal * normal[e1]), b = normal x t (negated if its e2 component is
 * negative): get_sample_pos(u, v) = origin + t (u - M/2 + 0.5) + b (v - M/2 + 0.5) and
 * sample_uv_from_pos(p) = ((p - origin).t + M/2 - 0.5, (p - origin).b + M/2 - 0.5).  A Plane with an
 * axis normal thereby gets exactly the lightmap of the matching XPlane / YPlane / ZPlane.  The normal
 * is taken as given (unit length expected, as the reference's sdf expects it). */
int rm_scene_add_plane(rm_scene *scene, const float normal[3], float distance, const rm_material *material,
                       const float colour[3]);

/* Extensions of the reference's object set (no live counterpart; the benchmark's "mixed
 * primitives with smooth unions" scene needs them):
 * - a box: the commented-out distance function of objects.rs:370-380 with centre and half
 *   extents as parameters; shaded like Plane above (no lightmap);
 * - smooth unions: with k > 0 the per-object distances of every march are folded with the
 *   polynomial smooth minimum  min(a,b) - h*h*k/4, h = max(k-|a-b|,0)/k.  The fold runs over the
 *   objects grouped by type - X, Y, Z planes, generic planes, spheres, boxes - and by index
 *   inside a group (the smooth minimum does not associate, so the order is part of the
 *   definition).  The hit object stays the one with the smallest own distance.  k = 0
 *   (default) is the reference's hard minimum (ray.rs:23-33). */
int rm_scene_add_box(rm_scene *scene, const float position[3], const float half_extent[3],
                     const rm_material *material, const float colour[3]);
int rm_scene_set_smooth_union(rm_scene *scene, float k);

/* PointLight{position,intensity} and Engine.camera_position (main.rs:64-71). */
int rm_scene_set_light(rm_scene *scene, const float position[3], float intensity);
int rm_scene_set_camera(rm_scene *scene, const float position[3]);

/* ---- engine ------------------------------------------------------------- */

/* Engine{objects,camera_position,light} (engine.rs:329-333) bound to one HIP
 * device.  width/height replace the compile-time WIDTH/HEIGHT (engine.rs:9-10)
 * and map_size replaces MAP_SIZE (radiosity.rs:12; pass 8 for the reference's).
 * The scene is copied: it may be destroyed afterwards. */
rm_engine *rm_engine_create(const rm_scene *scene, int width, int height, int map_size, int device);
void rm_engine_destroy(rm_engine *engine);

/* Work is enqueued on `hip_stream` (a hipStream_t; NULL = the null stream). */
int rm_engine_set_stream(rm_engine *engine, void *hip_stream);

/* Engine::compute_lightmaps (engine.rs:246-326) with the bounce count of
 * engine.rs:299 as a parameter (4 in the reference). */
int rm_engine_compute_lightmaps(rm_engine *engine, int bounces);

/* The same precompute in steps, for patch-sharded runs: receiver patches are
 * numbered p = (cloud_object * map_size + x) * map_size + y.
 *   begin        clouds (engine.rs:259-281) + direct lighting (radiosity.rs:52-101), all patches
 *   form_factors visibility x geometry of radiosity.rs:126-152 for receivers [p0,p1)
 *   bounce       radiosity.rs:159-196 for receivers [p0,p1): writes their slice of the
 *                next emissive map and of the lightmap
 *   commit       engine.rs:321-324: next emissive becomes current (after every rank's
 *                slice has been gathered into the buffers below)
 * rm_engine_radiosity_buffers exposes the device arrays a caller all-gathers:
 * each is [n_patches][3] float; n_patches is returned too. */
int rm_engine_radiosity_begin(rm_engine *engine);
int rm_engine_radiosity_form_factors(rm_engine *engine, int p0, int p1);
int rm_engine_radiosity_bounce(rm_engine *engine, int p0, int p1);
int rm_engine_radiosity_commit(rm_engine *engine);
int rm_engine_radiosity_buffers(rm_engine *engine, void **d_next_emissive, void **d_lightmaps, int *n_patches);

/* rm_engine_compute_lightmaps for callers that cannot allocate device memory (code under HIP graph
 * capture set-up, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same kernels, same result bit for bit; the clouds, the emissive maps and the
 * form-factor matrix live in a static device array of the library (4.10 MiB per GPU, held until the
 * process ends) instead of buffers allocated on first use, and the call allocates nothing.
 * Limit: map_size^2 * lightmapped objects <= RM_STATIC_MAX_PATCHES, else RM_ERR_LIMIT.  Every engine
 * on one GPU shares that GPU's array: calls from several engines or threads on one GPU take turns
 * (a lock per GPU, held for the length of the call; calls on different GPUs run side by side).  The
 * stepwise rm_engine_radiosity_* entry points keep using the engine's own buffers. */
#define RM_STATIC_MAX_PATCHES 1024
int rm_engine_compute_lightmaps_static(rm_engine *engine, int bounces);

/* EngineObject::get_lightmap / set_lightmap (objects.rs:34-35,216-217):
 * `values` is [map_size][map_size][3] float, first index u (as sample_map[u][v]). */
int rm_engine_get_lightmap(rm_engine *engine, int object, float *values);
int rm_engine_set_lightmap(rm_engine *engine, int object, const float *values);

/* The reference recomputes its lightmaps at every start (main.rs:73) and keeps them nowhere.  These
 * two write and read all lightmaps of an engine as one small file (6 KiB for the default scene:
 * magic, version, map_size, count, then object index and [map_size][map_size][3] floats per
 * lightmap), so that a host can skip the precompute.  Loading checks map_size, the lightmap count and
 * the object each lightmap belongs to, and changes nothing when the file does not fit. */
int rm_engine_save_lightmaps(rm_engine *engine, const char *path);
int rm_engine_load_lightmaps(rm_engine *engine, const char *path);

/* The camera block of Engine::render (engine.rs:63-75): accumulated mouse
 * position and scroll, and this frame's movement.  Stateful exactly as the
 * reference: camera_position += rot(rel_move) on every call. */
int rm_engine_input(rm_engine *engine, int32_t mouse_x, int32_t mouse_y, const float rel_move[3], int32_t scroll);
int rm_engine_get_camera(rm_engine *engine, float position[3]);
int rm_engine_set_camera(rm_engine *engine, const float position[3]);

/* Engine::render (engine.rs:38-127): input block + all three passes, result
 * copied to host memory.  `pitch_bytes` is the distance between rows (the
 * reference ignores SDL's pitch and assumes width*4, engine.rs:88-89). */
int rm_engine_render(rm_engine *engine, uint8_t *bgra, size_t pitch_bytes, int32_t mouse_x, int32_t mouse_y,
                     const float rel_move[3], int32_t scroll);

/* Engine::render for a frame loop (main.rs:82-151) without a stall per frame: the input
 * block is applied at once, the frame and its copy into `bgra` are enqueued, and the call
 * returns.  `bgra` must be page-locked (rm_host_alloc).  Up to RM_FRAMES_IN_FLIGHT frames
 * may be pending: the copy of frame f runs on a second stream while frame f+1 renders.
 * rm_engine_render_wait blocks until the oldest pending frame has landed in its buffer;
 * rm_engine_render_async waits by itself when every slot is taken. */
#define RM_FRAMES_IN_FLIGHT 2
int rm_engine_render_async(rm_engine *engine, uint8_t *bgra, size_t pitch_bytes, int32_t mouse_x, int32_t mouse_y,
                           const float rel_move[3], int32_t scroll);
int rm_engine_render_wait(rm_engine *engine);

/* Passes 1-3 of Engine::render (engine.rs:77-126) for image rows
 * [row_begin,row_end) at the pose of the last rm_engine_input, written to device
 * memory; `d_bgra` points at the first byte of row `row_begin`.  Asynchronous on
 * the engine's stream. */
int rm_engine_render_rows_device(rm_engine *engine, void *d_bgra, size_t pitch_bytes, int row_begin, int row_end);

/* The same passes for every `strip_stride`-th strip of RM_STRIP_ROWS image rows,
 * starting with strip `first_strip` (strip s covers rows [8s, 8s+8)), `n_strips` of
 * them, written compactly: strip k of the selection lands at rows [8k, 8k+8) of
 * `d_bgra`.  Strips beyond the frame are skipped.  This is how a frame is split
 * over GPUs with even load: rank r of n renders strips r, r+n, r+2n, ...  No
 * counterpart in the reference (one process, rayon rows, engine.rs:93). */
#define RM_STRIP_ROWS 8
int rm_engine_render_strips_device(rm_engine *engine, void *d_bgra, size_t pitch_bytes, int first_strip,
                                   int strip_stride, int n_strips);

/* Frames on their way between GPUs.  The fourth byte of every pixel the engine renders is 0
 * (engine.rs:116-126; FXAA passes it through), so rows that cross xGMI travel as 3 bytes per pixel:
 * rm_engine_pack_bgr_device turns `n_pixels` (a multiple of 4) B,G,R,0 pixels at `d_bgra` into
 * B,G,R triples at `d_bgr`; rm_engine_unpack_bgr_strips_device rebuilds the engine's whole frame of
 * B,G,R,0 rows at `d_bgra_frame` from the packed rows of `world` ranks as an all-gather leaves them
 * (rank r's part holds `strips_per_rank` strips of RM_STRIP_ROWS rows: frame strips r, r+world, ...;
 * world = 1 means the rows are already in frame order).  The frame width must be a multiple of 4.
 * Asynchronous on the engine's stream.  No counterpart in the reference (one process). */
int rm_engine_pack_bgr_device(rm_engine *engine, const void *d_bgra, void *d_bgr, size_t n_pixels);
int rm_engine_unpack_bgr_strips_device(rm_engine *engine, const void *d_bgr, void *d_bgra_frame, int world,
                                       int strips_per_rank);

/* FXAA (fxaa.txt:1-135, a commented-out block nothing in the reference calls),
 * defined here out of place: every tap is read from `d_src`.  Both buffers hold
 * whole frames of width*4-byte rows; rows [row_begin,row_end) of `d_dst` are
 * written, reading rows row-1 .. row+4 of `d_src`. */
int rm_engine_fxaa_device(rm_engine *engine, const void *d_src, void *d_dst, int row_begin, int row_end);

/* Primary rays only (Ray::march from the camera, ray.rs:15-42): per pixel the
 * hit object (-1 = none), the number of march iterations and the final ray
 * position.  Host arrays of width*height (x3 for positions); any may be NULL. */
int rm_engine_trace_primary(rm_engine *engine, int32_t *hit, int32_t *steps, float *position);

/* Engine::shade_object (engine.rs:147-244) for `n` arbitrary hits: object index,
 * position[3], view direction[3] in, linear colour[3] out (host arrays). */
int rm_engine_probe_shade(rm_engine *engine, int n, const int32_t *object, const float *position,
                          const float *direction, float *colour);

/* The last pass of Engine::render for `n` arbitrary linear colours (host arrays, [n][3] in, [n] out):
 * colour * exposure -> ACESFilm -> sqrt -> to_pixel_range -> B | G << 8 | R << 16 (engine.rs:109-125,
 * colour.rs:33-44).  `packed` comes from the code the frames use, which estimates each channel with
 * the hardware's one-ulp reciprocal and square root and falls back to the exact operations when an
 * estimate lies within 2e-4 of a rounding boundary; `packed_exact` from the exact operations alone.
 * The two must agree everywhere; a test aid with no counterpart in the reference. */
int rm_engine_probe_pack(rm_engine *engine, int n, const float *linear, float exposure, uint32_t *packed,
                         uint32_t *packed_exact);

/* Supersampling, engine.rs:95-108 (commented out in the reference): samples = 4 casts four rays per
 * pixel - the pixel's direction moved by (+-0.5/WIDTH, 0, 0) and (0, +-0.5/HEIGHT, 0) before it is
 * normalised - and averages their linear colours ((c0 + c1) + c2 + c3) / 4 ahead of tone mapping.
 * samples = 1 (the default) is the reference's live path.  Applies to every render entry point. */
int rm_engine_set_ssaa(rm_engine *engine, int samples);

/* Surface normals, objects.rs:135-149: on = 0 (default) takes the central differences of the live
 * code (step 0.0001); on = 1 the four-point tetrahedral form the reference carries commented out
 * (step 0.0015).  Applies to shading and to the normals of the lightmap precompute. */
int rm_engine_set_tetrahedral_normals(rm_engine *engine, int on);

/* The refraction block the reference carries commented out in shade_object, engine.rs:196-228
 * (with Ray::_internal_march, ray.rs:44-56, and Vec3::refract, vector.rs:40-48): on = 1 makes every
 * hit on an object with reflectivity > 0.001 also cast a ray refracted through the object (index
 * ratio 2.5; total internal reflection mirrors it) and add the shade of what that ray hits, before
 * the mirror ray is added.  A hit then has two children: the recursion is cut 16 hits deep, cuts are
 * counted in RM_STAT_DEPTH_CAP_HITS.  Default 0.  Applies to every render entry point and to
 * rm_engine_probe_shade. */
int rm_engine_set_refraction(rm_engine *engine, int on);

/* The arithmetic of the frame kernel (rm_engine_render*, rm_group_frame, rm_multi_render without
 * supersampling and refraction; everything else - lightmaps, FXAA, the probes - has one arithmetic).
 *   RM_PRECISION_STRICT (default)  every operation is the reference's IEEE binary32 operation in the
 *       reference's order, no fused multiply-adds (vector.rs:22-32, ray.rs:28-39): frames equal a strict
 *       CPU evaluation bit for bit except where powf/atan2f enter;
 *   RM_PRECISION_FAST  a pixel's chain of sight rays (direction, march, hit, surface normal, mirror
 *       direction) stays strict; what a hit's colour is made of - texture and lightmap coordinates, light
 *       terms, Phong, the soft shadow's own steps, the sums, the tone map - uses fused multiply-adds and
 *       the hardware's one-ulp root, reciprocal, exp2 and log2.  Frames stay within the tolerance this
 *       renderer states against the reference - at most 2/255 per channel, at least 99.9 % of the pixels
 *       identical - and are the same bytes from run to run; they are not bit-identical to the strict tier.
 * No counterpart in the reference, which has one arithmetic. */
#define RM_PRECISION_STRICT 0
#define RM_PRECISION_FAST 1
int rm_engine_set_precision(rm_engine *engine, int precision);

/* Occlusion of patch pairs in the lightmap precompute, radiosity.rs:126-152.  The reference marches an
 * occlusion ray for every pair and leaves a blocked pair out of the sum (`continue`, :137); an open
 * pair adds emission * min(max(attenuation, 0) / d^2, 2).  on = 0 (default): a pair whose attenuation
 * is <= 0 (the patches face away from each other) is not marched - blocked or open, it adds nothing or
 * emission * 0 - and a blocked pair enters the sum as emission * 0.  For emission values that are
 * finite that is the reference's result bit for bit; an emission value that is infinite or NaN (a
 * material with an infinite `emissive`, a light sitting in a patch) turns emission * 0 into NaN, so
 * NaN then also reaches patches whose pair with that source is blocked, where the reference adds
 * nothing.  on = 1: every pair is marched and a blocked pair is left out of the sum, as in the
 * reference, whatever the emission values are (the precompute takes about 1.4 times as long). */
int rm_engine_set_exact_occlusion(rm_engine *engine, int on);

/* Optional divergence cap on march iterations (0 = none; the loop of ray.rs:19
 * ends by itself after at most 50/0.001 steps).  Cut marches are counted. */
int rm_engine_set_march_cap(rm_engine *engine, int max_steps);

/* Slow tiles first (no counterpart in the reference; the pixels do not depend on it).  A
 * launch of fewer than 65536 tiles lasts as long as its slowest wave, so the frame kernel renders
 * the tiles that took longest the last time they were rendered first and as four waves of 4x4
 * pixels each (a quarter's pixels drift apart less between reflection levels: 0.6 of the time).
 * A tile counts as slow there at 45 us for a quarter and at least three times the mean tile of
 * the engine's previous planned launch (RM_SPLIT_US and RM_SPLIT_TIMES_MEAN in the environment set
 * the two; RM_SPLIT_US=0 switches every plan off).  A longer launch of a scene with a spatial index
 * only hands out first, whole, the tiles that took at least twice the mean (RM_EARLY_TIMES_MEAN).
 * A plan belongs to a stream and to the set of tiles its launches name (a row range, a selection of
 * strips); an engine keeps 32 of them, a stream that finds none free gives up the one of its own it
 * has not used for longest, and a stream that has none goes unplanned.  On by default.  A caller that keeps several launches of
 * one engine in flight, where the machine is busy anyway and the quarters' idle lanes cost
 * throughput, switches it off (rm_group_create does for more than one frame in flight, unless a
 * rank's share is below 20000 tiles, where the plans still pay).  rm_engine_split_tiles waits for
 * the engine's work and reports how many tiles the most recent planned launch had on its list. */
int rm_engine_set_tile_splitting(rm_engine *engine, int on);
int rm_engine_split_tiles(rm_engine *engine, uint32_t *count);

/* Event counters accumulated since creation or the last reset (RM_STAT_*). */
int rm_engine_get_stats(rm_engine *engine, uint64_t stats[RM_STAT_COUNT]);
int rm_engine_reset_stats(rm_engine *engine);

/* Measurement aid (no counterpart in the reference): the clock the GPU's vector units run at right now,
 * in MHz.  One wave on a stream of its own counts shader-clock cycles over `microseconds` of the
 * constant-rate counter - beside whatever the engine's streams are running, so a caller that has
 * frames enqueued gets the clock under that load (the nominal peak assumes 2400 MHz; a chip that is
 * busy runs below it).  Returns when the wave has finished. */
int rm_engine_shader_clock(rm_engine *engine, int microseconds, double *mhz);

/* Block until everything enqueued on the engine's stream has finished. */
int rm_engine_synchronize(rm_engine *engine);

/* ---- several GPUs ----------------------------------------------------------
 *
 * The reference spreads a frame over CPU threads by image rows (rayon, engine.rs:93) inside the one
 * Engine::render call its main loop makes per frame (main.rs:141-145); it has no second device to
 * talk to.  Here the rows of a frame are spread over GPUs as interleaved strips of RM_STRIP_ROWS rows
 * (GPU r of n renders strips r, r+n, ...: even load wherever the expensive pixels are), behind two
 * forms of the same per-frame call. */

/* Form 1 - one rank per GPU, each in a process (or thread) of its own, frames left in HBM on every
 * rank.  The strips travel as 3 bytes per pixel through one all-gather per frame: RCCL over xGMI
 * (librccl.so is looked up at run time; RM_ERR_UNSUPPORTED if there is none), or an all-gather the
 * host supplies (MPI, a test double).  Replaces the rayon fan-out of engine.rs:93,116. */
typedef struct rm_group rm_group;
#define RM_GROUP_ID_BYTES 128
#define RM_GROUP_MAX_FRAMES_IN_FLIGHT 4

/* An RCCL unique id (ncclGetUniqueId): one rank makes it, the host hands its 128 bytes to the others. */
int rm_group_unique_id(uint8_t id[RM_GROUP_ID_BYTES]);

/* Rank `rank` of `world` over `engine` (same scene, size and map_size on every rank); collective:
 * every rank calls it with the same id.  `frames_in_flight` (1..4) frames overlap, each on a stream
 * of its own.  The engine must outlive the group. */
rm_group *rm_group_create(rm_engine *engine, const uint8_t id[RM_GROUP_ID_BYTES], int rank, int world, int frames_in_flight);

/* The same with the host's own exchange: all_gather must leave, in `d_recv` on every rank, the
 * `bytes_per_rank` bytes at `d_send` of rank 0, 1, ... one after the other, ordered on `hip_stream`
 * (or complete on return); non-zero return = failure. */
typedef int (*rm_allgather_fn)(void *user, const void *d_send, void *d_recv, size_t bytes_per_rank, void *hip_stream);
rm_group *rm_group_create_custom(rm_engine *engine, int rank, int world, int frames_in_flight, rm_allgather_fn all_gather, void *user);
void rm_group_destroy(rm_group *group);
int rm_group_info(const rm_group *group, int *rank, int *world, int *strips_per_rank, int *bytes_per_pixel_exchanged);

/* Engine::compute_lightmaps (engine.rs:246-326) with the receiver patches split over the ranks: each
 * rank computes the form factors of its patches once and per bounce its slice of the next emissive
 * map; one all-gather per bounce (a bounce reads only the previous one's values, engine.rs:299-325)
 * and one of the lightmaps at the end.  Collective; returns when this rank holds all lightmaps. */
int rm_group_compute_lightmaps(rm_group *group, int bounces);

/* Engine::render (engine.rs:38-127) for the group, one call per rank and frame: the input block is
 * applied (every rank passes the same values and so keeps the same camera), this rank's strips are
 * rendered (in the group's i-th frame rank r takes the strips (r + i) mod world, (r + i) mod world +
 * world, ...: the assignment rotates, so that the few strips with a view's slowest tiles are not the
 * same rank's in every frame; RM_GROUP_ROTATE=0 in every rank's environment pins rank r to strips r,
 * r + world, ...), all-gathered and put in row order, and with fxaa != 0 the gathered frame is filtered
 * (fxaa.txt, as rm_engine_fxaa_device) - the whole frame, B,G,R,0 rows of width*4 bytes, lands in
 * `d_bgra_frame` (device memory) on every rank.  Asynchronous: the call only enqueues; frame f and
 * frame f + frames_in_flight share a stream and buffers, all others overlap.  Collective: every rank
 * issues the same frames in the same order. */
int rm_group_frame(rm_group *group, void *d_bgra_frame, int fxaa, int32_t mouse_x, int32_t mouse_y, const float rel_move[3],
                   int32_t scroll);
/* Blocks until every frame enqueued so far has landed in its buffer. */
int rm_group_wait(rm_group *group);

/* Where a frame of the group ends up.  The reference has exactly one consumer of a frame - the window
 * texture its main loop locks (main.rs:141-145) - so nothing requires every rank to hold every frame:
 *   RM_EXCHANGE_ALL_GATHER (default)  every rank receives every frame (3 bytes per pixel on the links);
 *   RM_EXCHANGE_PRESENTER             frame f of the group lands on rank f mod world alone (the
 *                                     "presenter" of that frame, rotating so that every rank's links and
 *                                     post-passes carry one frame in `world`): the other ranks only send
 *                                     their strips (4 bytes per pixel as rendered, no packing pass) and
 *                                     run neither the re-ordering nor FXAA for it.
 * Collective in the sense that every rank must select the same mode before the same frame; the frame
 * count restarts at 0.  rm_group_next_presenter tells where the next rm_group_frame will leave its
 * frame (-1: on every rank); on the other ranks rm_group_frame ignores `d_bgra_frame` (NULL allowed). */
#define RM_EXCHANGE_ALL_GATHER 0
#define RM_EXCHANGE_PRESENTER 1
int rm_group_set_exchange(rm_group *group, int mode);
int rm_group_next_presenter(const rm_group *group, int *rank);
/* The gather-to-one-rank counterpart of rm_allgather_fn for groups made by rm_group_create_custom:
 * on rank `root` the `bytes_per_rank` bytes at `d_send` of rank 0, 1, ... must land one after the
 * other in `d_recv` (NULL on the other ranks), ordered on `hip_stream` or complete on return.  Without
 * one, RM_EXCHANGE_PRESENTER runs the group's all-gather function and lets the other ranks drop what
 * they receive. */
typedef int (*rm_gather_fn)(void *user, const void *d_send, void *d_recv, size_t bytes_per_rank, int root, void *hip_stream);
int rm_group_set_custom_gather(rm_group *group, rm_gather_fn gather, void *user);
/* RCCL communicators the group's frames travel through: one per frame in flight (split from the
 * group's own, so that the exchanges of consecutive frames are not chained through one communicator's
 * launch order), 1 where RCCL cannot split or RM_GROUP_SHARED_COMM=1 is set, 0 for a custom exchange. */
int rm_group_communicators(const rm_group *group, int *count);

/* Measurement aid: with on != 0 every frame's render launch and its exchange (the collective alone,
 * without packing and re-ordering) are bracketed by events on the frame's stream; rm_group_render_time
 * and rm_group_exchange_time wait for the frames and return the mean over the (up to 256) last ones.
 * With several frames in flight such a span also covers whatever else the GPU ran meanwhile: time a
 * kernel alone with one frame in flight, or with rm_group_wait between frames. */
int rm_group_set_timing(rm_group *group, int on);
int rm_group_render_time(rm_group *group, double *mean_ms, int *frames);
int rm_group_exchange_time(rm_group *group, double *mean_ms, int *frames);

/* Form 2 - one process, `n_devices` GPUs, the frame wanted in host memory: what a host shaped like
 * main.rs holds (Engine{..} at main.rs:64-71, compute_lightmaps at :73, render into the locked
 * texture at :141-145).  One engine per listed device (a device may be listed more than once), one
 * host thread per engine inside the library; every GPU copies its strips over its own PCIe link
 * straight into the caller's frame, so no data moves between GPUs for a frame. */
typedef struct rm_multi rm_multi;
rm_multi *rm_multi_create(const rm_scene *scene, int width, int height, int map_size, const int *devices, int n_devices);
void rm_multi_destroy(rm_multi *multi);
int rm_multi_device_count(const rm_multi *multi);
/* The engine on the index-th listed device, e.g. for rm_engine_get_lightmap or the rm_engine_set_*
 * switches (set them on every engine alike, while no call on `multi` is running). */
rm_engine *rm_multi_engine(rm_multi *multi, int index);
/* Engine::compute_lightmaps, receiver patches split over the GPUs, slices exchanged through
 * page-locked host memory once per bounce. */
int rm_multi_compute_lightmaps(rm_multi *multi, int bounces);
/* Engine::render: as rm_engine_render (same arguments, same bytes in `bgra`), GPU i of n rendering
 * strips i, i+n, ... */
int rm_multi_render(rm_multi *multi, uint8_t *bgra, size_t pitch_bytes, int32_t mouse_x, int32_t mouse_y, const float rel_move[3],
                    int32_t scroll);

#ifdef __cplusplus
}
#endif
#endif /* RM_H */
