//! The scene and engine types of the reference (main.rs:64-71,154-193; objects.rs:158-257;
//! engine.rs:38-41,246) over the C ABI of librm_hip.so.  The frame path and the lightmap
//! precompute run on the GPU; window, input and presentation stay with the caller.
//!
//! Not compiled in this repository's image (no Rust toolchain): source for a host that has one.
use rm_sys as sys;
use std::ffi::CStr;

pub type Vec3 = [f32; 3];
pub type Colour = Vec3;
pub use sys::rm_material as Material;

/// material.rs:12-21
pub fn basic_material() -> Material {
    Material { ambient: 0.25, diffuse: 1.0, specular: 0.0, shininess: 4.0, reflectivity: 0.0, emissive: 0.0 }
}

/// colour.rs:7-15
pub fn rgb(r: u8, g: u8, b: u8) -> Colour {
    let sq = |c: u8| (c as f32 / 255.0) * (c as f32 / 255.0);
    [sq(r), sq(g), sq(b)]
}

/// texture.rs:4-29: decoded RGB8 rows, top row first (before the squaring of `rgb!`).
pub struct Texture {
    pub rgb8: Vec<u8>,
    pub width: i32,
    pub height: i32,
    pub uscale: f32,
    pub vscale: f32,
}

/// objects.rs:158-203 (+ the box extension)
pub enum Object<'t> {
    Sphere { position: Vec3, radius: f32, material: Material, colour: Colour },
    Plane { normal: Vec3, distance: f32, material: Material, colour: Colour },
    XPlane { x: f32, dir: f32, material: Material, colour: Colour },
    YPlane { y: f32, dir: f32, material: Material, colour: Colour, texture: &'t Texture },
    ZPlane { z: f32, dir: f32, material: Material, colour: Colour },
    Box { position: Vec3, half_extent: Vec3, material: Material, colour: Colour },
}

/// objects.rs:205-209
pub struct PointLight {
    pub position: Vec3,
    pub intensity: f32,
}

#[derive(Debug)]
pub struct Error(pub String);

fn last_error() -> Error {
    Error(unsafe { CStr::from_ptr(sys::rm_last_error()) }.to_string_lossy().into_owned())
}

fn check(rc: i32) -> Result<(), Error> {
    if rc == 0 {
        Ok(())
    } else {
        Err(last_error())
    }
}

/// `construct_objects()` and the `Engine { .. }` literal (main.rs:64-71,154-193) as an rm_scene; the
/// caller destroys it.
unsafe fn build_scene(objects: &[Object], camera_position: Vec3, light: &PointLight) -> Result<*mut sys::rm_scene, Error> {
    let scene = sys::rm_scene_create();
    if scene.is_null() {
        return Err(last_error());
    }
    let mut registered: Vec<(*const Texture, i32)> = Vec::new();
    let mut result = Ok(());
    for o in objects {
        let rc = match o {
            Object::Sphere { position, radius, material, colour } =>
                sys::rm_scene_add_sphere(scene, position.as_ptr(), *radius, material, colour.as_ptr()),
            Object::Plane { normal, distance, material, colour } =>
                sys::rm_scene_add_plane(scene, normal.as_ptr(), *distance, material, colour.as_ptr()),
            Object::XPlane { x, dir, material, colour } =>
                sys::rm_scene_add_xplane(scene, *x, *dir, material, colour.as_ptr()),
            Object::ZPlane { z, dir, material, colour } =>
                sys::rm_scene_add_zplane(scene, *z, *dir, material, colour.as_ptr()),
            Object::Box { position, half_extent, material, colour } =>
                sys::rm_scene_add_box(scene, position.as_ptr(), half_extent.as_ptr(), material, colour.as_ptr()),
            Object::YPlane { y, dir, material, colour, texture } => {
                let key = *texture as *const Texture;
                let id = match registered.iter().find(|(p, _)| *p == key) {
                    Some((_, id)) => *id,
                    None => {
                        let id = sys::rm_texture_create_rgb8(scene, texture.rgb8.as_ptr(), texture.width,
                                                             texture.height, texture.uscale, texture.vscale);
                        registered.push((key, id));
                        id
                    }
                };
                if id < 0 { id } else { sys::rm_scene_add_yplane(scene, *y, *dir, material, colour.as_ptr(), id) }
            }
        };
        if rc < 0 {
            result = Err(last_error());
            break;
        }
    }
    if result.is_ok() {
        result = check(sys::rm_scene_set_light(scene, light.position.as_ptr(), light.intensity))
            .and_then(|_| check(sys::rm_scene_set_camera(scene, camera_position.as_ptr())));
    }
    match result {
        Ok(()) => Ok(scene),
        Err(e) => {
            sys::rm_scene_destroy(scene);
            Err(e)
        }
    }
}

/// engine.rs:329-333; width and height are run-time values here.
pub struct Engine {
    handle: *mut sys::rm_engine,
    pub width: usize,
    pub height: usize,
}

impl Engine {
    pub fn new(objects: &[Object], camera_position: Vec3, light: &PointLight, width: usize, height: usize,
               map_size: usize, device: i32) -> Result<Engine, Error> {
        unsafe {
            let scene = build_scene(objects, camera_position, light)?;
            let handle = sys::rm_engine_create(scene, width as i32, height as i32, map_size as i32, device);
            sys::rm_scene_destroy(scene);
            if handle.is_null() {
                return Err(last_error());
            }
            Ok(Engine { handle, width, height })
        }
    }

    /// engine.rs:246 (the reference runs four bounces)
    pub fn compute_lightmaps(&mut self) -> Result<(), Error> {
        check(unsafe { sys::rm_engine_compute_lightmaps(self.handle, 4) })
    }

    /// The same without any device allocation (work buffers in the library's static array; at most
    /// `rm_sys::RM_STATIC_MAX_PATCHES` patches).
    pub fn compute_lightmaps_static(&mut self) -> Result<(), Error> {
        check(unsafe { sys::rm_engine_compute_lightmaps_static(self.handle, 4) })
    }

    /// All lightmaps to / from one file, to skip the precompute at the next start (the reference
    /// recomputes at every start, main.rs:73).
    pub fn save_lightmaps(&mut self, path: &std::ffi::CStr) -> Result<(), Error> {
        check(unsafe { sys::rm_engine_save_lightmaps(self.handle, path.as_ptr()) })
    }
    pub fn load_lightmaps(&mut self, path: &std::ffi::CStr) -> Result<(), Error> {
        check(unsafe { sys::rm_engine_load_lightmaps(self.handle, path.as_ptr()) })
    }

    /// engine.rs:38-41; `buffer` holds `height` rows of `pitch` bytes, B,G,R,0 per pixel.
    pub fn render(&mut self, buffer: &mut [u8], pitch: usize, mouse_x: i32, mouse_y: i32, rel_move: Vec3,
                  scroll: i32) -> Result<(), Error> {
        assert!(pitch >= self.width * 4 && buffer.len() >= pitch * self.height);
        check(unsafe {
            sys::rm_engine_render(self.handle, buffer.as_mut_ptr(), pitch, mouse_x, mouse_y, rel_move.as_ptr(), scroll)
        })
    }

    /// The parts of the reference that are commented out there: four rays per pixel
    /// (engine.rs:95-108), four-point normals (objects.rs:142-149), refraction (engine.rs:196-228).
    pub fn set_ssaa(&mut self, on: bool) -> Result<(), Error> {
        check(unsafe { sys::rm_engine_set_ssaa(self.handle, if on { 4 } else { 1 }) })
    }
    pub fn set_tetrahedral_normals(&mut self, on: bool) -> Result<(), Error> {
        check(unsafe { sys::rm_engine_set_tetrahedral_normals(self.handle, on as i32) })
    }
    pub fn set_refraction(&mut self, on: bool) -> Result<(), Error> {
        check(unsafe { sys::rm_engine_set_refraction(self.handle, on as i32) })
    }
    /// Slow tiles first and in quarters (small launches; on by default, the pixels do not depend on it).
    pub fn set_tile_splitting(&mut self, on: bool) -> Result<(), Error> {
        check(unsafe { sys::rm_engine_set_tile_splitting(self.handle, on as i32) })
    }

    pub fn camera_position(&self) -> Result<Vec3, Error> {
        let mut p = [0.0f32; 3];
        check(unsafe { sys::rm_engine_get_camera(self.handle, p.as_mut_ptr()) })?;
        Ok(p)
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { sys::rm_engine_destroy(self.handle) }
    }
}

/// The same engine over several GPUs of this process (rm_multi_*): GPU i of n renders every n-th strip
/// of 8 rows and copies it into the caller's frame over its own PCIe link; the lightmap patches are
/// split likewise.  What the reference's main loop needs is unchanged: `compute_lightmaps()` once
/// (main.rs:73), `render(buffer, ..)` per frame (main.rs:141-145).
pub struct MultiEngine {
    handle: *mut sys::rm_multi,
    pub width: usize,
    pub height: usize,
}

impl MultiEngine {
    pub fn new(objects: &[Object], camera_position: Vec3, light: &PointLight, width: usize, height: usize,
               map_size: usize, devices: &[i32]) -> Result<MultiEngine, Error> {
        unsafe {
            let scene = build_scene(objects, camera_position, light)?;
            let handle = sys::rm_multi_create(scene, width as i32, height as i32, map_size as i32, devices.as_ptr(), devices.len() as i32);
            sys::rm_scene_destroy(scene);
            if handle.is_null() {
                return Err(last_error());
            }
            Ok(MultiEngine { handle, width, height })
        }
    }

    pub fn compute_lightmaps(&mut self) -> Result<(), Error> {
        check(unsafe { sys::rm_multi_compute_lightmaps(self.handle, 4) })
    }

    pub fn render(&mut self, buffer: &mut [u8], pitch: usize, mouse_x: i32, mouse_y: i32, rel_move: Vec3,
                  scroll: i32) -> Result<(), Error> {
        assert!(pitch >= self.width * 4 && buffer.len() >= pitch * self.height);
        check(unsafe {
            sys::rm_multi_render(self.handle, buffer.as_mut_ptr(), pitch, mouse_x, mouse_y, rel_move.as_ptr(), scroll)
        })
    }
}

impl Drop for MultiEngine {
    fn drop(&mut self) {
        unsafe { sys::rm_multi_destroy(self.handle) }
    }
}
