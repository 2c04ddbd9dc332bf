// Plain structs shared by the host runtime and the HIP kernels.
#pragma once
#include <stdint.h>

namespace rm {

enum ObjType : int { T_SPHERE = 0, T_XPLANE = 1, T_YPLANE = 2, T_ZPLANE = 3, T_PLANE = 4, T_BOX = 5 };

// engine.rs:12-14
constexpr float MAX_MARCH_DISTANCE = 50.0f;
constexpr float SMALL_DISTANCE = 0.001f;
constexpr int MAX_SHAD_IT = 64;
// objects.rs:13
constexpr float STEP_SIZE = 0.0001f;
// engine.rs:192
constexpr float SHADOW_K = 16.0f;

#ifndef RM_MAX_DEPTH
#define RM_MAX_DEPTH 16
#endif
constexpr int MAX_DEPTH = RM_MAX_DEPTH;  // explicit reflection stack (the reference recurses without bound)

// Geometry of one object, 32 bytes.
//   sphere      : cx, cy, cz, radius
//   axis plane  : offset, dir
//   plane       : nx, ny, nz, distance, tx, ty, tz (lightmap tangent), sign of the bitangent normal x tangent
//   box         : cx, cy, cz, hx, hy, hz   (centre, half extents)
struct Geom {
    float p[8];
};

// Everything shading needs about one object, 64 bytes, indexed per lane.
struct ShadeRec {
    float ambient, diffuse, specular, shininess, reflectivity, emissive;  // material.rs:2-9
    float cr, cg, cb;                                                    // constant colour
    int type;
    int texture;   // texture id or -1
    int textured;  // YPlane with !(y > 0): colour comes from the texture (objects.rs:267-274)
    int lm_slot;   // index of this object's lightmap, -1 = none (box)
    int pad[3];
};
static_assert(sizeof(ShadeRec) == 64, "ShadeRec layout");

struct TextureRec {
    const float *texels;  // [h][w][4] floats, rgb already squared as rgb! does (colour.rs:7-15)
    int w, h;
    float uscale, vscale;
    float inv_w, inv_h;  // 1/w and 1/h where w and h are both powers of two (the wrap of texture.rs:32-33 then needs no fmod), else 0
};

// The march loops see the scene grouped by primitive type, so that they run
// without a per-object type dispatch.  The minimum over objects does not depend
// on the order; the index of the closest object (lowest index on ties, as the
// strict `<` of ray.rs:29) is recovered from `index` once a march has ended.
//
// A scene within the bounds <NS, NA, NG> (spheres, planes per axis, generic
// planes) is passed by value in the kernel arguments: every field is then
// addressed by a compile-time constant and lives in scalar registers.  Two tiers
// are built; larger scenes use the same layout in global memory (GroupedLarge).
// EXACT: every slot is in use, so the loops run without count checks.
template <int NS, int NA, int NG, bool EXACT = false>
struct GroupedSmall {
    static constexpr int kSpheres = NS, kAxis = NA, kPlanes = NG;
    static constexpr bool kExact = EXACT;
    int n_sph, n_x, n_y, n_z, n_gp;
    float sph_extent;  // largest |centre coordinate| + |radius| over the spheres (0 without spheres)
    int idx_sph[NS > 0 ? NS : 1];
    int idx_ax[3][NA > 0 ? NA : 1];
    int idx_gp[NG > 0 ? NG : 1];
    float sph[NS > 0 ? NS : 1][4];     // cx, cy, cz, r
    float ax[3][NA > 0 ? NA : 1][2];   // offset, dir
    float gp[NG > 0 ? NG : 1][4];      // nx, ny, nz, distance
};
typedef GroupedSmall<2, 2, 0, true> GroupedTierX;  // exactly two spheres and two planes per axis, each plane facing along +-1
typedef GroupedSmall<2, 2, 0> GroupedTierA;
typedef GroupedSmall<4, 4, 2> GroupedTierB;

// Spatial index over the bounded primitives (spheres, boxes) of a large scene.  Every cell of a
// uniform grid lists the primitives that come within the cell's reach of any point of the cell,
// spheres and boxes separately, each list in ascending object index with the geometry stored in
// place.  A primitive missing from a cell's lists is at least that cell's reach away from every
// point of the cell; a point outside the grid is at least `reach` away from every primitive.
// There are two sets of lists over the same cells: a short one (the few nearest primitives of the
// cell), tried first, and a long one (everything within a wider reach).  cells == 0: no index.
//
// All of it lives in one pool of dwords, so that a march loop keeps one pointer in scalar registers:
//   * the cell table of level L at dword 4 * cells * L, one record of four dwords per cell: where the
//     cell's sphere block starts, where its box block starts (dword offsets into the pool),
//     sphere count | box count << 16, and the cell's reach (float bits);
//   * a sphere block: groups of four spheres, 24 dwords each: cx, cy, cz, r of the four, then their
//     four object indices (the last group is filled up with zeros and -1), then for each of the four a
//     lower bound of its distance over the whole cell;
//   * a box block: eight dwords per box: cx, cy, cz, hx, hy, hz, object index, the same lower bound.
// The lower bounds let a wave whose lanes share a cell pass over an entry with one compare: where no
// lane's running value (plus the blend width) reaches an entry's bound, the entry cannot change
// any lane's fold.
// Blocks start at multiples of four dwords; the pool ends with sixteen dwords of padding, since a
// wave-uniform walk fetches a whole group of spheres or a pair of boxes at once.
constexpr int kGridLevels = 2;
constexpr int kGridSphereGroup = 24;  // dwords per group of four spheres
struct GridView {
    int nx, ny, nz, cells;
    float ox, oy, oz;   // lower corner
    float inv_cell;     // 1 / edge length
    float reach;        // outside the grid
    const int *pool;
};

struct GroupedLarge {
    int n_sph, n_x, n_y, n_z, n_gp, n_box;
    const int *idx_sph;  // original object index of each entry
    const int *idx_ax[3];
    const int *idx_gp;
    const int *idx_box;
    const float *sph;    // [n_sph][4]
    const float *ax[3];  // [n][2]
    const float *gp;     // [n_gp][4]
    const float *box;    // [n_box][8]: cx, cy, cz, hx, hy, hz, -, -
    // every plane once more, in the order of the fold (X, Y, Z, generic; by object index inside a
    // group), one 32-byte record each: kind (0, 1, 2 = the axis, 3 = generic) and object index as
    // bit patterns, then offset, direction (an axis plane) or normal, distance (a generic one).  The
    // march loops walk this one array: one pointer and one count in scalar registers instead of eight
    // and four.
    int n_planes_all;
    const float *planes_all;
    // Extension: > 0 folds the distances with a smooth minimum instead of the hard one.  The fold
    // runs in the order of the groups (X, Y, Z planes, generic planes, spheres, boxes; by object
    // index within a group), which is part of the definition since the smooth minimum does not
    // associate.
    float smooth_k;
    float sph_extent;  // as in GroupedSmall
    GridView grid;
};

struct SceneView {
    int n;
    const int *type;         // [n]
    const Geom *geom;        // [n]
    const ShadeRec *shade;   // [n]
    const TextureRec *tex;   // [n_tex]
    const float *lightmaps;  // [n_lm][M][M][3]
    int map_size;
    float light[3];
    float intensity;
    float sky[3];
    int march_cap;  // 0 = none
    int tetra_normals;  // 1 = the four-point normal of objects.rs:142-149 (commented out there)
    int refraction;     // 1 = the refraction block of engine.rs:196-228 (commented out there)
    int exact_occlusion;  // 1 = the lightmap precompute marches every patch pair, as radiosity.rs:130-141 does (see rm_engine_set_exact_occlusion)
    // The smallest distance from the light to any plane of the scene, as the march loops compute plane
    // distances (+inf without planes): a shadow ray of length L ends at the light, so no plane is nearer
    // than L / k to its far end while L / k stays below this (shadow_stays_open).  Minus infinity where a
    // plane's distance function rises faster than 1 per unit of length (a direction or normal longer
    // than 1.001), which that proof's error bounds do not allow for.
    float light_clear;
    unsigned long long *stats;
};

// One frame's camera (engine.rs:63-86).
struct Pose {
    float sx, cx, sy, cy, zdepth, exposure;
    float cam[3];
    int W, H;
};

}  // namespace rm
