//! `HipRenderer`: the `vello::Renderer` surface (vello/src/lib.rs:432-515) over libvello_hip.so.
//!
//! Everything above the seam is vello's own host code and runs unchanged: `Scene` -> `Encoding` ->
//! `Resolver::resolve` (packed scene bytes, `Layout`, gradient `Ramps`, `Images`).  Where `Renderer::render_to_texture`
//! turns those into a `Recording` for `WgpuEngine::run_recording` (vello/src/render.rs:84-112,
//! vello/src/wgpu_engine.rs:380-777), `HipRenderer::render_to_buffer` hands them to the C ABI.
//!
//! SOURCE ONLY: no Rust toolchain exists in the build image; see Cargo.toml.
pub mod ffi;

use core::ffi::{c_int, c_void, CStr};
use ffi::*;
use vello::{AaConfig, AaSupport, RenderParams, Scene};
use vello_encoding::{Layout, Resolver};

#[derive(Debug)]
pub enum Error {
    /// vello::Error::NoCompatibleDevice (vello/src/lib.rs:262): no gfx950 device; there is no CPU fallback.
    NoCompatibleDevice,
    /// An AA mode that was not enabled in the `AaSupport` given to `new` (render.rs:566-598 panics upstream),
    /// or a packed scene whose streams contradict each other.
    Invalid(String),
    /// A HIP runtime error.
    Hip(String),
    /// A bump-allocated pool overflowed and auto-grow is off: the target is untouched (fine.wgsl:1070-1074); the
    /// counters say what the frame needs (`HipRenderer::grow_pools`).
    Capacity(vello_hip_bump),
    /// An engine-internal wait gave up (a look-back or grid barrier whose partner never arrived): the frame is discarded.
    Internal(String),
}

pub struct HipRenderer {
    ctx: *mut vello_hip_ctx,
    resolver: Resolver,
    packed: Vec<u8>,
    atlas_size: (u32, u32),
}
// Renderer: Send, !Sync (vello/src/lib.rs:351-352): a context is single-threaded, different contexts are independent
unsafe impl Send for HipRenderer {}

fn to_hip_layout(l: &Layout) -> vello_hip_layout {
    vello_hip_layout {
        n_draw_objects: l.n_draw_objects,
        n_paths: l.n_paths,
        n_clips: l.n_clips,
        bin_data_start: l.bin_data_start,
        path_tag_base: l.path_tag_base,
        path_data_base: l.path_data_base,
        draw_tag_base: l.draw_tag_base,
        draw_data_base: l.draw_data_base,
        transform_base: l.transform_base,
        style_base: l.style_base,
    }
}

impl HipRenderer {
    /// `Renderer::new` (vello/src/lib.rs:432-459).  Takes the one field of `RendererOptions` that means something here --
    /// `antialiasing_support` -- as an `AaSupport`: `RendererOptions` itself is `#[cfg(feature = "wgpu")]`
    /// (vello/src/lib.rs:371-373) and this crate depends on vello WITHOUT that feature (no wgpu in the build at all);
    /// `use_cpu`, `num_init_threads` and `pipeline_cache` configure wgpu's shader compilation.
    pub fn new(device: i32, aa: AaSupport) -> Result<Self, Error> {
        let mask = (aa.area as u32) * VELLO_HIP_AA_MASK_AREA
            | (aa.msaa8 as u32) * VELLO_HIP_AA_MASK_MSAA8
            | (aa.msaa16 as u32) * VELLO_HIP_AA_MASK_MSAA16;
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { vello_hip_create(device as c_int, mask, core::ptr::null(), &mut ctx) };
        if rc != VELLO_HIP_OK {
            return Err(Error::NoCompatibleDevice);
        }
        // the robust path upstream is a TODO (lib.rs:753-764); here a frame that overflows grows the pools and re-runs
        unsafe { vello_hip_set_auto_grow(ctx, 1) };
        Ok(Self { ctx, resolver: Resolver::new(), packed: Vec::new(), atlas_size: (0, 0) })
    }

    /// What a failed call left in the context.  A free function over the raw context pointer (not `&self`): it is called
    /// while `render_to_buffer` still holds the `Ramps` / `Images` that `Resolver::resolve` lends out of `self.resolver`
    /// (`resolve<'a>(&'a mut self, ..) -> (Layout, Ramps<'a>, Images<'a>)`, vello_encoding/src/resolve.rs:183-187).
    fn error(ctx: *mut vello_hip_ctx, rc: c_int, bump: vello_hip_bump) -> Error {
        let msg = unsafe { CStr::from_ptr(vello_hip_last_error(ctx)) }.to_string_lossy().into_owned();
        match rc {
            VELLO_HIP_E_NO_DEVICE => Error::NoCompatibleDevice,
            VELLO_HIP_E_INVALID => Error::Invalid(msg),
            VELLO_HIP_E_CAPACITY => Error::Capacity(bump),
            VELLO_HIP_E_INTERNAL => Error::Internal(msg),
            _ => Error::Hip(msg),
        }
    }

    /// `Renderer::render_to_texture` (vello/src/lib.rs:474-515) into a caller-owned linear RGBA8 buffer (host memory, or
    /// device memory of the context's GPU): un-premultiplied, rows of `stride` bytes, origin top-left.
    pub fn render_to_buffer(&mut self, scene: &Scene, target: *mut c_void, stride: usize, on_device: bool,
                            params: &RenderParams) -> Result<(), Error> {
        // identical to Render::render_encoding_coarse up to the uploads (vello/src/render.rs:135-232).
        // Borrows: `resolve` borrows `self.resolver` mutably for as long as `ramps` / `images` live, and `self.packed`
        // for the call only.  Everything below therefore touches `self` through DISJOINT fields (`self.ctx`, a `Copy`
        // raw pointer read once up front; `self.atlas_size`; `self.packed`) and never through a `&self` / `&mut self`
        // method, which would borrow all of `self` while the resolver is lent out (E0502).
        let ctx = self.ctx;
        let (layout, ramps, images) = self.resolver.resolve(scene.encoding(), &mut self.packed);
        // vello/src/render.rs:160-203: the persistent image atlas follows the Resolver's image cache
        if (images.width, images.height) != self.atlas_size {
            let rc = unsafe { vello_hip_resize_image_atlas(ctx, images.width, images.height) };
            if rc != VELLO_HIP_OK {
                return Err(Self::error(ctx, rc, vello_hip_bump::default()));
            }
            self.atlas_size = (images.width, images.height);
        }
        // `images.images: &[(ImageData, u32, u32)]` (vello_encoding/src/image_cache.rs:24): iterate by reference
        for (image, x, y) in images.images.iter() {
            let bytes: &[u8] = image.data.data();
            let rc = unsafe { vello_hip_write_image(ctx, *x, *y, image.width, image.height, bytes.as_ptr(), 0) };
            if rc != VELLO_HIP_OK {
                return Err(Self::error(ctx, rc, vello_hip_bump::default()));
            }
        }
        let p = vello_hip_render_params {
            width: params.width,
            height: params.height,
            base_color: params.base_color.premultiply().to_rgba8().to_u32(), // vello_encoding/src/config.rs:183
            aa: match params.antialiasing_method {
                AaConfig::Area => VELLO_HIP_AA_AREA,
                AaConfig::Msaa8 => VELLO_HIP_AA_MSAA8,
                AaConfig::Msaa16 => VELLO_HIP_AA_MSAA16,
            },
        };
        let hl = to_hip_layout(&layout);
        let mut bump = vello_hip_bump::default();
        // `Ramps { data: &[u32], width, height }` is `Copy` (ramp_cache.rs:16-21); an empty slice's pointer is dangling but
        // non-null and is never read (n_ramps = 0)
        let rc = unsafe {
            vello_hip_render(ctx, self.packed.as_ptr(), self.packed.len(), &hl, &p, ramps.data.as_ptr(), ramps.height,
                             target, stride, on_device as c_int, &mut bump)
        };
        if rc == VELLO_HIP_OK { Ok(()) } else { Err(Self::error(ctx, rc, bump)) }
    }

    /// The vello_tests entry point (`render_then_debug_sync`, vello_tests/src/lib.rs:76): a frame into a fresh Vec.
    pub fn render_to_vec(&mut self, scene: &Scene, params: &RenderParams) -> Result<Vec<u8>, Error> {
        let mut out = vec![0u8; params.width as usize * params.height as usize * 4];
        self.render_to_buffer(scene, out.as_mut_ptr().cast(), params.width as usize * 4, false, params)?;
        Ok(out)
    }

    /// vello_hip_grow_pools after `Error::Capacity` when auto-grow was switched off.
    pub fn grow_pools(&mut self, demand: &vello_hip_bump) -> bool {
        unsafe { vello_hip_grow_pools(self.ctx, demand, core::ptr::null_mut()) == VELLO_HIP_OK }
    }

    pub fn raw(&self) -> *mut vello_hip_ctx {
        self.ctx
    }
}

impl Drop for HipRenderer {
    fn drop(&mut self) {
        unsafe { vello_hip_destroy(self.ctx) }
    }
}
