//! `TerrainRenderer` over `libtopo_hip.so`: the same five methods as the reference's
//! `topo-renderer/src/render/terrain_renderer.rs` (new / update / add_terrain / unload_terrain / render), minus the wgpu
//! handles, plus the multi-GPU panorama.  A maintainer swaps this in under `RenderEngine`
//! (render_engine.rs:143,160-166,183-189,212-214,276-284) and `UiController::change_location` (ui_controller.rs:48).
//!
//! SOURCE ONLY: the image this repository is built in has no cargo / rustc, so this crate has never been compiled.  The
//! ABI underneath is exercised by tests/abi_harness.c (C99) and the ctypes binding; INTEGRATION.md walks through the swap.
use std::ffi::CStr;
use std::ptr;

use topo_hip_sys as sys;

#[derive(Debug)]
pub struct TopoError {
    pub code: i32,
    pub message: String,
}

fn check(ctx: *mut sys::topo_ctx, rc: i32) -> Result<(), TopoError> {
    if rc == sys::TOPO_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::topo_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(TopoError { code: rc, message })
}

/// `CoordinateTransform` (common/coordinate_transform.rs:16-20): the six f32 a tile's GeoTIFF tags reduce to.
#[derive(Copy, Clone, Debug)]
pub struct CoordinateTransform {
    pub raster_point: [f32; 2],
    pub model_point: [f32; 2],
    pub pixel_scale: [f32; 2],
}

/// `pad_256` (data/mod.rs:9-11): the row pitch of the depth read-back.
pub fn pad_256(size: u32) -> u32 {
    unsafe { sys::topo_pad_256(size) }
}

pub struct TerrainRenderer {
    ctx: *mut sys::topo_ctx,
    target_size: (u32, u32),
    /// RGBA8 (sRGB-encoded), tightly packed: what the reference's post pass leaves in the surface texture
    pub frame: Vec<u8>,
    /// Depth32Float rows `pad_256(4 * width)` bytes apart: what `RenderEngine::get_visible_labels` indexes
    /// (render_engine.rs:364-370)
    pub depth_read: Vec<u8>,
}

impl TerrainRenderer {
    /// was: `new(device, format, target_size)` -- terrain_renderer.rs:37
    pub fn new(hip_device: i32, target_size: (u32, u32)) -> Result<Self, TopoError> {
        let mut ctx = ptr::null_mut();
        check(ptr::null_mut(), unsafe {
            sys::topo_create(&mut ctx, hip_device, target_size.0, target_size.1, sys::TOPO_FORMAT_RGBA8_UNORM_SRGB)
        })?;
        Ok(Self { ctx, target_size, frame: Vec::new(), depth_read: Vec::new() })
    }

    /// was: `update(device, queue, target_size, &uniforms, &postprocessing_uniforms)` -- :151.
    /// `Uniforms` / `PostprocessingUniforms` (render/data.rs:33-41,74-80) are `#[repr(C)]` Pod with exactly the layout of
    /// `topo_uniforms` / `topo_post_uniforms`: pass `bytemuck::bytes_of(..)` reinterpreted.
    pub fn update(&mut self, target_size: (u32, u32), uniforms: &sys::topo_uniforms, post: &sys::topo_post_uniforms) -> Result<(), TopoError> {
        self.target_size = target_size;
        check(self.ctx, unsafe { sys::topo_update(self.ctx, target_size.0, target_size.1, uniforms, post) })
    }

    /// was: `add_terrain(device, queue, location, height_map_data, coordinate_transform, size, proxy)` -- :173.
    /// `location` = `GeoLocation::to_numerical()` (topo-common/src/lib.rs:127-129).  The `NormalsComputed` event of the
    /// reference is informational only (render_engine.rs:330-332): nothing is sent.
    pub fn add_terrain(&mut self, location: (i32, i32), height_map_data: &[u8], ct: CoordinateTransform, size: (u32, u32)) -> Result<(), TopoError> {
        assert_eq!(height_map_data.len(), 4 * size.0 as usize * size.1 as usize);
        check(self.ctx, unsafe {
            sys::topo_add_terrain(self.ctx, location.0, location.1, height_map_data.as_ptr() as *const f32, size.0, size.1,
                                  ct.raster_point.as_ptr(), ct.model_point.as_ptr(), ct.pixel_scale.as_ptr())
        })
    }

    /// `fetch_terrain`'s decode step folded in (background_runner.rs:113-136): the downloaded GeoTIFF bytes straight to a tile.
    pub fn add_terrain_geotiff(&mut self, location: (i32, i32), tiff_bytes: &[u8]) -> Result<(), TopoError> {
        check(self.ctx, unsafe { sys::topo_add_terrain_geotiff(self.ctx, location.0, location.1, tiff_bytes.as_ptr(), tiff_bytes.len()) })
    }

    /// was: `unload_terrain(&location)` -- :361
    pub fn unload_terrain(&mut self, location: (i32, i32)) {
        unsafe { sys::topo_unload_terrain(self.ctx, location.0, location.1) };
    }

    /// `UiController::change_location` (ui_controller.rs:23-59): unloads what left the 100 km range, returns what to request.
    pub fn change_location(&mut self, latitude: f32, longitude: f32) -> Result<Vec<(i32, i32)>, TopoError> {
        let mut out = vec![0i32; 2 * 1024];
        let (mut n_request, mut n_unloaded) = (0u32, 0u32);
        check(self.ctx, unsafe {
            sys::topo_change_location(self.ctx, latitude, longitude, 100_000.0, out.as_mut_ptr(), 1024, &mut n_request, &mut n_unloaded)
        })?;
        Ok(out.chunks(2).take(n_request.min(1024) as usize).map(|p| (p[0], p[1])).collect())
    }

    /// was: `render(target, encoder, viewport)` + the depth copy of `RenderEngine::render` -- :365, render_engine.rs:219.
    /// Fills `frame` and (if asked) `depth_read`; `DepthBufferReady` can be raised synchronously afterwards.
    pub fn render(&mut self, want_depth: bool) -> Result<(), TopoError> {
        let (w, h) = (self.target_size.0 as usize, self.target_size.1 as usize);
        let pitch = pad_256(4 * w as u32) as usize;
        self.frame.resize(4 * w * h, 0);
        self.depth_read.resize(pitch * h, 0);
        let depth = if want_depth { self.depth_read.as_mut_ptr() as *mut f32 } else { ptr::null_mut() };
        check(self.ctx, unsafe { sys::topo_render(self.ctx, self.frame.as_mut_ptr(), 4 * w, depth, pitch) })
    }

    /// `RenderEngine::get_visible_labels` (render_engine.rs:338-396) against the depth of the last `render(true)`.
    pub fn visible_peaks(&mut self, peaks_xyz: &[[f32; 3]]) -> Result<Vec<Option<(u32, u32)>>, TopoError> {
        let n = peaks_xyz.len();
        let (mut vis, mut xy) = (vec![0u8; n], vec![0u32; 2 * n]);
        check(self.ctx, unsafe { sys::topo_visible_peaks(self.ctx, n as u32, peaks_xyz.as_ptr() as *const f32, vis.as_mut_ptr(), xy.as_mut_ptr()) })?;
        Ok((0..n).map(|i| if vis[i] != 0 { Some((xy[2 * i], xy[2 * i + 1])) } else { None }).collect())
    }

    /// The 360-degree strip, this process's share of it (new).  `strip_dev` / `depth_dev`: device memory the
    /// host allocated (HIP), `[8][sector_h][sector_w][4]` bytes / `[8][sector_h][sector_w]` floats, the same on every rank.
    /// Asynchronous: complete after `synchronize()`.
    pub fn render_panorama(&mut self, comm: Option<&Comm>, eye: [f32; 3], yaw0: f32, pitch: f32, sector: (u32, u32), sun_deg: (f32, f32),
                           view_mode: i32, strip_dev: *mut u8, depth_dev: *mut f32) -> Result<(), TopoError> {
        check(self.ctx, unsafe {
            sys::topo_render_panorama(self.ctx, comm.map_or(ptr::null_mut(), |c| c.raw), eye.as_ptr(), yaw0, pitch, sector.0, sector.1,
                                      sun_deg.0, sun_deg.1, view_mode, strip_dev, depth_dev)
        })
    }

    pub fn synchronize(&mut self) -> Result<(), TopoError> {
        check(self.ctx, unsafe { sys::topo_synchronize(self.ctx) })
    }
}

impl Drop for TerrainRenderer {
    fn drop(&mut self) {
        unsafe { sys::topo_destroy(self.ctx) }
    }
}

/// One process per GPU: this process's place among them (RCCL behind the C ABI).
pub struct Comm {
    raw: *mut sys::topo_comm,
}

impl Comm {
    /// Rank 0 creates the id and hands the 128 bytes to the other ranks over the host's own channel.
    pub fn unique_id() -> Result<[u8; 128], TopoError> {
        let mut id = [0u8; 128];
        check(ptr::null_mut(), unsafe { sys::topo_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }

    /// Collective: returns when all `world` ranks have joined.
    pub fn init(hip_device: i32, id: &[u8; 128], rank: i32, world: i32) -> Result<Self, TopoError> {
        let mut raw = ptr::null_mut();
        check(ptr::null_mut(), unsafe { sys::topo_comm_init(&mut raw, hip_device, id.as_ptr(), rank, world) })?;
        Ok(Self { raw })
    }
}

impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { sys::topo_comm_destroy(self.raw) }
    }
}
