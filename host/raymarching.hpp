// C++ host-side mirror of the reference's scene and Engine API over the C ABI
// (include/rm.h).  The reference's host is Rust; this header keeps its type and
// function names so that a port of main.rs reads line for line:
//
//   reference (Rust)                              here (C++)
//   Vec3::new(x,y,z)            vector.rs:12      rm_host::Vec3{x,y,z}
//   Material{..}/Material::basic material.rs:2-21 rm_host::Material / Material::basic()
//   Texture::new(path,us,vs)    texture.rs:13     rm_host::Texture::create(path,us,vs)
//   Sphere{..}                  objects.rs:158    rm_host::Sphere{..}
//   XPlane::new / YPlane::new / ZPlane::new       rm_host::XPlane{..} / YPlane{..} / ZPlane{..}
//   PointLight{..}              objects.rs:205    rm_host::PointLight{..}
//   Engine{objects,camera_position,light}         rm_host::Engine(objects, camera, light, w, h)
//   engine.compute_lightmaps()  engine.rs:246     engine.compute_lightmaps()
//   engine.render(buffer,..)    engine.rs:38      engine.render(buffer, pitch, mouse_x, mouse_y, rel_move, scroll)
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "../include/rm.h"

namespace rm_host {

struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
};
typedef Vec3 Colour;

// colour.rs:7-15
inline Colour rgb(int r, int g, int b) {
    auto sq = [](int c) { return ((float)c / 255.0f) * ((float)c / 255.0f); };
    return Colour{sq(r), sq(g), sq(b)};
}
// colour.rs:17-22
inline Colour WHITE() { return rgb(255, 255, 255); }
inline Colour SOFT_RED() { return rgb(214, 81, 81); }
inline Colour SOFT_GREEN() { return rgb(81, 214, 81); }
inline Colour SOFT_GRAY() { return rgb(214, 214, 214); }
inline Colour SOFT_YELLOW() { return rgb(230, 230, 127); }

// material.rs:2-21
struct Material {
    float ambient = 0.25f, diffuse = 1.0f, specular = 0.0f, shininess = 4.0f, reflectivity = 0.0f, emissive = 0.0f;
    static Material basic() { return Material{}; }
};

// texture.rs:4-29: decoded RGB8 rows, top-down.  Only uncompressed 24-bit BMP is
// read here (the reference's one asset); other decoders can fill `rgb` directly.
struct Texture {
    std::vector<uint8_t> rgb;
    int width = 0, height = 0;
    float uscale = 1.0f, vscale = 1.0f;

    static std::shared_ptr<Texture> create(const std::string &path, float uscale, float vscale) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        std::vector<uint8_t> d;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
        auto u32 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24); };
        auto u16 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8); };
        if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') throw std::runtime_error(path + ": not a BMP file");
        uint32_t offset = u32(10);
        int32_t w = (int32_t)u32(18), h = (int32_t)u32(22);
        if (u16(26) != 1 || u16(28) != 24 || u32(30) != 0) throw std::runtime_error(path + ": only uncompressed 24-bit BMP is supported");
        bool bottom_up = h > 0;
        if (h < 0) h = -h;
        size_t stride = ((size_t)w * 3 + 3) & ~(size_t)3;
        if (d.size() < offset + stride * h) throw std::runtime_error(path + ": truncated");
        auto t = std::make_shared<Texture>();
        t->width = w;
        t->height = h;
        t->uscale = uscale;
        t->vscale = vscale;
        t->rgb.resize((size_t)w * h * 3);
        for (int y = 0; y < h; y++) {
            const uint8_t *row = &d[offset + stride * (bottom_up ? (h - 1 - y) : y)];
            for (int x = 0; x < w; x++) {
                t->rgb[((size_t)y * w + x) * 3 + 0] = row[3 * x + 2];
                t->rgb[((size_t)y * w + x) * 3 + 1] = row[3 * x + 1];
                t->rgb[((size_t)y * w + x) * 3 + 2] = row[3 * x + 0];
            }
        }
        return t;
    }
};

// objects.rs:158-209
struct Sphere {
    Vec3 position;
    float radius;
    Material material;
    Colour colour;
};
struct Plane {
    Vec3 normal;
    float distance;
    Material material;
    Colour colour;
};
struct XPlane {
    float x, dir;
    Material material;
    Colour colour;
};
struct YPlane {
    float y, dir;
    Material material;
    Colour colour;
    std::shared_ptr<Texture> texture;
};
struct ZPlane {
    float z, dir;
    Material material;
    Colour colour;
};
struct PointLight {
    Vec3 position;
    float intensity;
};

// Box<dyn EngineObject>
typedef std::variant<Sphere, Plane, XPlane, YPlane, ZPlane> EngineObject;

inline void check(int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + rm_last_error());
}

class Engine {
   public:
    // engine.rs:329-333; width/height replace the reference's WIDTH/HEIGHT constants.
    Engine(const std::vector<EngineObject> &objects, Vec3 camera_position, PointLight light, int width, int height,
           int map_size = 8, int device = 0)
        : width_(width), height_(height) {
        rm_scene *scene = build_scene(objects, camera_position, light);
        handle_ = rm_engine_create(scene, width, height, map_size, device);
        rm_scene_destroy(scene);
        if (!handle_) throw std::runtime_error(std::string("rm_engine_create: ") + rm_last_error());
        engines_.push_back(handle_);
    }
    // The same engine over several GPUs of this process (rm_multi_*): GPU i of n renders every n-th
    // strip of 8 rows and copies it into the caller's frame; the lightmap patches are split likewise.
    Engine(const std::vector<EngineObject> &objects, Vec3 camera_position, PointLight light, int width, int height,
           int map_size, const std::vector<int> &devices)
        : width_(width), height_(height) {
        rm_scene *scene = build_scene(objects, camera_position, light);
        multi_ = rm_multi_create(scene, width, height, map_size, devices.data(), (int)devices.size());
        rm_scene_destroy(scene);
        if (!multi_) throw std::runtime_error(std::string("rm_multi_create: ") + rm_last_error());
        for (int i = 0; i < rm_multi_device_count(multi_); i++) engines_.push_back(rm_multi_engine(multi_, i));
        handle_ = engines_[0];
    }
    ~Engine() {
        if (multi_)
            rm_multi_destroy(multi_);
        else
            rm_engine_destroy(handle_);
    }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;

    // engine.rs:246
    void compute_lightmaps(int bounces = 4) {
        if (multi_)
            check(rm_multi_compute_lightmaps(multi_, bounces), "compute_lightmaps");
        else
            check(rm_engine_compute_lightmaps(handle_, bounces), "compute_lightmaps");
    }
    // the same without any device allocation (work buffers in the library's static array; at most RM_STATIC_MAX_PATCHES patches)
    void compute_lightmaps_static(int bounces = 4) { check(rm_engine_compute_lightmaps_static(handle_, bounces), "compute_lightmaps_static"); }

    // all lightmaps to / from one file, to skip the precompute at the next start (the reference recomputes, main.rs:73)
    void save_lightmaps(const std::string &path) { check(rm_engine_save_lightmaps(handle_, path.c_str()), "save_lightmaps"); }
    void load_lightmaps(const std::string &path) { for (rm_engine *e : engines_) check(rm_engine_load_lightmaps(e, path.c_str()), "load_lightmaps"); }

    // engine.rs:38-41 (the `directions` scratch argument of the reference is gone)
    void render(uint8_t *buffer, size_t pitch, int32_t mouse_x, int32_t mouse_y, Vec3 rel_move, int32_t scroll) {
        if (multi_)
            check(rm_multi_render(multi_, buffer, pitch, mouse_x, mouse_y, &rel_move.x, scroll), "render");
        else
            check(rm_engine_render(handle_, buffer, pitch, mouse_x, mouse_y, &rel_move.x, scroll), "render");
    }

    // engine.rs:95-108 (commented out there): four rays per pixel
    void set_ssaa(bool on) { for (rm_engine *e : engines_) check(rm_engine_set_ssaa(e, on ? 4 : 1), "set_ssaa"); }

    // objects.rs:142-149 (commented out there): four-point normals
    void set_tetrahedral_normals(bool on) { for (rm_engine *e : engines_) check(rm_engine_set_tetrahedral_normals(e, on ? 1 : 0), "set_tetrahedral_normals"); }

    // engine.rs:196-228 (commented out there): reflective objects also refract
    void set_refraction(bool on) { for (rm_engine *e : engines_) check(rm_engine_set_refraction(e, on ? 1 : 0), "set_refraction"); }
    // slow tiles first and in quarters (small launches; on by default, the pixels do not depend on it)
    void set_tile_splitting(bool on) { for (rm_engine *e : engines_) check(rm_engine_set_tile_splitting(e, on ? 1 : 0), "set_tile_splitting"); }

    Vec3 camera_position() const {
        Vec3 p;
        check(rm_engine_get_camera(handle_, &p.x), "get_camera");
        return p;
    }
    int width() const { return width_; }
    int height() const { return height_; }
    rm_engine *handle() { return handle_; }

   private:
    // construct_objects() + the Engine literal (main.rs:64-71,154-193) as an rm_scene
    static rm_scene *build_scene(const std::vector<EngineObject> &objects, Vec3 camera_position, PointLight light) {
        rm_scene *scene = rm_scene_create();
        std::vector<std::pair<const Texture *, int>> textures;
        auto mat = [](const Material &m) { return rm_material{m.ambient, m.diffuse, m.specular, m.shininess, m.reflectivity, m.emissive}; };
        for (const EngineObject &o : objects) {
            int rc = 0;
            if (auto s = std::get_if<Sphere>(&o)) {
                rm_material m = mat(s->material);
                rc = rm_scene_add_sphere(scene, &s->position.x, s->radius, &m, &s->colour.x);
            } else if (auto p = std::get_if<Plane>(&o)) {
                rm_material m = mat(p->material);
                rc = rm_scene_add_plane(scene, &p->normal.x, p->distance, &m, &p->colour.x);
            } else if (auto xp = std::get_if<XPlane>(&o)) {
                rm_material m = mat(xp->material);
                rc = rm_scene_add_xplane(scene, xp->x, xp->dir, &m, &xp->colour.x);
            } else if (auto yp = std::get_if<YPlane>(&o)) {
                rm_material m = mat(yp->material);
                int id = -1;
                for (auto &t : textures)
                    if (t.first == yp->texture.get()) id = t.second;
                if (id < 0 && yp->texture) {
                    id = rm_texture_create_rgb8(scene, yp->texture->rgb.data(), yp->texture->width, yp->texture->height,
                                                yp->texture->uscale, yp->texture->vscale);
                    textures.push_back({yp->texture.get(), id});
                }
                rc = rm_scene_add_yplane(scene, yp->y, yp->dir, &m, &yp->colour.x, id);
            } else if (auto zp = std::get_if<ZPlane>(&o)) {
                rm_material m = mat(zp->material);
                rc = rm_scene_add_zplane(scene, zp->z, zp->dir, &m, &zp->colour.x);
            }
            if (rc < 0) {
                std::string msg = rm_last_error();
                rm_scene_destroy(scene);
                throw std::runtime_error("scene: " + msg);
            }
        }
        rm_scene_set_light(scene, &light.position.x, light.intensity);
        rm_scene_set_camera(scene, &camera_position.x);
        return scene;
    }

    rm_engine *handle_ = nullptr;       // the engine, or the first of several
    rm_multi *multi_ = nullptr;         // several GPUs
    std::vector<rm_engine *> engines_;  // all of them, for the switches
    int width_, height_;
};

}  // namespace rm_host
