// Plain structs shared by the host runtime and the HIP kernels.
#pragma once
#include <stdint.h>

namespace rm {

enum ObjType : int { T_SPHERE = 0, T_XPLANE = 1, T_YPLANE = 2, T_ZPLANE = 3, T_PLANE = 4 };

// engine.rs:12-14
constexpr float MAX_MARCH_DISTANCE = 50.0f;
constexpr float SMALL_DISTANCE = 0.001f;
constexpr int MAX_SHAD_IT = 64;
// objects.rs:13
constexpr float STEP_SIZE = 0.0001f;
// engine.rs:192
constexpr float SHADOW_K = 16.0f;

constexpr int MAX_DEPTH = 16;    // explicit reflection stack (the reference recurses without bound)
constexpr int SMALL_SCENE = 16;  // scenes up to this size travel in the kernel arguments (SGPRs)

// Geometry of one object, 16 bytes.
//   sphere      : cx, cy, cz, radius
//   axis plane  : offset, dir, -, -
//   plane       : nx, ny, nz, distance
struct Geom {
    float p[4];
};

// Everything shading needs about one object, 64 bytes, indexed per lane.
struct ShadeRec {
    float ambient, diffuse, specular, shininess, reflectivity, emissive;  // material.rs:2-9
    float cr, cg, cb;                                                    // constant colour
    int type;
    int texture;   // texture id or -1
    int textured;  // YPlane with !(y > 0): colour comes from the texture (objects.rs:267-274)
    int lm_slot;   // index of this object's lightmap, -1 = none (generic Plane)
    int pad[3];
};
static_assert(sizeof(ShadeRec) == 64, "ShadeRec layout");

struct TextureRec {
    const float *texels;  // [h][w][4] floats, rgb already squared as rgb! does (colour.rs:7-15)
    int w, h;
    float uscale, vscale;
};

// A scene small enough to be passed by value: the march loops index it with
// compile-time constants, so every field lives in scalar registers.
struct SmallScene {
    int n;
    int type[SMALL_SCENE];
    Geom geom[SMALL_SCENE];
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
    unsigned long long *stats;
};

// One frame's camera (engine.rs:63-86).
struct Pose {
    float sx, cx, sy, cy, zdepth, exposure;
    float cam[3];
    int W, H;
};

}  // namespace rm
