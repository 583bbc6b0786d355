// ===== src/ray_tracing/rendering/camera.rs:271-328 — Camera::render with the pixel loops on the GPU =================
// Same signature and side effects: the camera is initialised (camera.rs:89-129 runs inside rtxh_camera_render on the
// public fields), the image is resolved with the REQUESTED samples_per_pixel (camera.rs:317) and saved as PNG
// (camera.rs:322-325).  The indicatif progress bar (camera.rs:277-285) has nothing to tick per pixel any more;
// rtxh_camera_render_progressive reports per batch of sample rows instead.
pub fn render(&mut self, world: &dyn Hittable, lights: Option<Arc<dyn Hittable>>) {
    let ctx = rtx::Context::new(0).expect("no usable MI355X: this renderer has no CPU path");
    let mut s = rtx::Session::new();
    let w = world.record(&mut s).expect("the world holds an object that cannot be recorded");
    let l = match &lights { Some(l) => l.record(&mut s).expect("the lights hold an object that cannot be recorded"), None => -1 };
    let mut desc: rtx::sys::rtx_camera_desc = unsafe { std::mem::zeroed() };
    desc.aspect_ratio = self.aspect_ratio;
    desc.image_width = self.image_width; desc.samples_per_pixel = self.samples_per_pixel; desc.max_depth = self.max_depth;
    desc.background = [self.background.x, self.background.y, self.background.z];
    desc.vfov = self.vfov;
    desc.lookfrom = [self.lookfrom.x, self.lookfrom.y, self.lookfrom.z];
    desc.lookat = [self.lookat.x, self.lookat.y, self.lookat.z];
    desc.vup = [self.vup.x, self.vup.y, self.vup.z];
    desc.defocus_angle = self.defocus_angle; desc.focus_dist = self.focus_dist;
    let file = std::ffi::CString::new(self.output_filename.clone()).unwrap();
    let seed: u64 = rand::random();                               // the reference's renders are unseeded (random.rs:9-13)
    let st = unsafe { rtx::sys::rtxh_camera_render(s.raw, ctx.raw, &mut desc, w, l, seed, file.as_ptr(),
                                                   std::ptr::null_mut(), std::ptr::null_mut()) };
    ctx.status_with_session(st, &s).expect("render failed");
    self.image_height = desc.image_height;                        // what initialize() would have left behind
    self.sqrt_spp = desc.sqrt_spp;
}
