// Headless counterpart of the reference's main.rs: same scene, camera and light
// (main.rs:62-71,154-193), lightmaps once (main.rs:73), then a frame loop that
// prints "Elapsed" per frame (main.rs:140-149).  Instead of an SDL window the
// last frame is written as a 32-bit BMP.
//
//   rm_headless [--frames N] [--width W] [--height H] [--out frame.bmp] [--texture path] [--devices 0,1,2,...]
//
// With --devices the frame and the lightmap patches are spread over the listed GPUs (one process,
// one engine per GPU; a GPU may be listed more than once).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "raymarching.hpp"

using namespace rm_host;

// main.rs:154-193
static std::vector<EngineObject> construct_objects(const std::string &texture_path) {
    auto wood_tex = Texture::create(texture_path, 32.0f, 32.0f);
    const Material BASIC_MAT = Material::basic();
    return {
        Sphere{Vec3{-1.2f, -1.0f, 0.1f}, 1.0f, Material{0.05f, 0.03f, 0.2f, 16.0f, 1.0f, 0.0f}, WHITE()},
        Sphere{Vec3{1.0f, -1.0f, -0.7f}, 1.0f, Material{0.1f, 1.0f, 0.9f, 32.0f, 0.25f, 0.0f}, SOFT_YELLOW()},
        YPlane{-2.0f, 1.0f, BASIC_MAT, SOFT_GRAY(), wood_tex},
        YPlane{4.0f, -1.0f, BASIC_MAT, SOFT_GRAY(), wood_tex},
        XPlane{-3.0f, 1.0f, BASIC_MAT, SOFT_RED()},
        XPlane{3.0f, -1.0f, BASIC_MAT, SOFT_GREEN()},
        ZPlane{2.0f, -1.0f, BASIC_MAT, SOFT_GRAY()},
        ZPlane{-4.0f, 1.0f, BASIC_MAT, SOFT_GRAY()},
    };
}

static void write_bmp32(const std::string &path, const std::vector<uint8_t> &bgra, int w, int h) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return;
    uint32_t size = 54 + (uint32_t)bgra.size();
    uint8_t hdr[54] = {'B', 'M'};
    auto put32 = [&](int o, uint32_t v) { for (int i = 0; i < 4; i++) hdr[o + i] = (uint8_t)(v >> (8 * i)); };
    put32(2, size);
    put32(10, 54);
    put32(14, 40);
    put32(18, (uint32_t)w);
    put32(22, (uint32_t)(-h));  // top-down rows, as the frame buffer
    hdr[26] = 1;
    hdr[28] = 32;
    put32(34, (uint32_t)bgra.size());
    fwrite(hdr, 1, 54, f);
    fwrite(bgra.data(), 1, bgra.size(), f);
    fclose(f);
}

int main(int argc, char **argv) {
    int frames = 5, width = 800, height = 600;  // engine.rs:9-10
    std::string out = "frame.bmp", texture = "assets/textures/Floor128.bmp";
    std::vector<int> devices;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i];
        if (k == "--frames") frames = atoi(argv[i + 1]);
        else if (k == "--width") width = atoi(argv[i + 1]);
        else if (k == "--height") height = atoi(argv[i + 1]);
        else if (k == "--out") out = argv[i + 1];
        else if (k == "--texture") texture = argv[i + 1];
        else if (k == "--devices") {
            for (const char *p = argv[i + 1]; *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
        }
    }
    printf("Hello, world!\n");  // main.rs:33
    try {
        const Vec3 camera{0.0f, 0.5f, -3.5f};
        const PointLight light{Vec3{2.0f, -1.0f, 1.5f}, 3.5f};
        std::unique_ptr<Engine> owned(devices.empty() ? new Engine(construct_objects(texture), camera, light, width, height)
                                                      : new Engine(construct_objects(texture), camera, light, width, height, 8, devices));
        Engine &engine = *owned;
        engine.compute_lightmaps();
        std::vector<uint8_t> buffer((size_t)width * height * 4);
        int32_t mouse_x = 0, mouse_y = 0, scroll = 0;  // main.rs:78-80
        for (int frame = 0; frame < frames; frame++) {
            Vec3 rel_move;  // no keys pressed
            auto now = std::chrono::steady_clock::now();
            engine.render(buffer.data(), (size_t)width * 4, mouse_x, mouse_y, rel_move, scroll);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - now).count();
            printf("Elapsed: %.2fms\n", ms);
        }
        write_bmp32(out, buffer, width, height);
        printf("{\"width\": %d, \"height\": %d, \"frames\": %d, \"gpus\": %d, \"image\": \"%s\"}\n", width, height, frames,
               devices.empty() ? 1 : (int)devices.size(), out.c_str());
    } catch (const std::exception &e) {
        fprintf(stderr, "rm_headless: %s\n", e.what());
        return 1;
    }
    return 0;
}
