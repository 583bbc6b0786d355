// ===== src/ray_tracing/geometry/hittable.rs — one provided method on the trait (hittable.rs:97-115) ================
pub trait Hittable: Send + Sync {
    /* hit, bounding_box, pdf_value, random: unchanged */
    /// Re-record this object through the backend's constructors; returns its handle in the session.
    fn record(&self, _s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        Err(rtx::Error::Unsupported("Hittable without a record implementation"))
    }
}

// ===== src/ray_tracing/materials/material.rs (material.rs:51-66) ====================================================
pub trait Material: Send + Sync {
    /* scatter, emitted, scattering_pdf: unchanged */
    fn record(&self, _s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        Err(rtx::Error::Unsupported("Material without a record implementation"))
    }
}
impl Material for NoMaterial {   // material.rs:69-82
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> { s.check(unsafe { rtx::sys::rtxh_mat_none(s.raw) }) }
}

// ===== src/ray_tracing/materials/texture/mod.rs (texture/mod.rs:10-12) ==============================================
pub trait Texture: Send + Sync {
    /* value: unchanged */
    fn record(&self, _s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        Err(rtx::Error::Unsupported("Texture without a record implementation"))
    }
}

fn key<T: ?Sized>(p: &std::sync::Arc<T>) -> usize { std::sync::Arc::as_ptr(p) as *const () as usize }

// ===== materials/texture/solid_color.rs, checker.rs, noise.rs, image.rs =============================================
impl Texture for SolidColor {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        s.check(unsafe { rtx::sys::rtxh_tex_solid(s.raw, self.albedo.x, self.albedo.y, self.albedo.z) })
    }
}
impl Texture for CheckerTexture {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let (e, o) = (self.even.record(s)?, self.odd.record(s)?);
        // the constructor takes `scale` and stores 1/scale, as checker.rs:16-22 does
        s.check(unsafe { rtx::sys::rtxh_tex_checker(s.raw, 1.0 / self.inv_scale, e, o) })
    }
}
impl Texture for NoiseTexture {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        // the very tables this object holds (procedural/noise.rs:6-11), so that the device evaluates the same field
        let rv: Vec<f64> = self.noise.ranvec.iter().flat_map(|v| [v.x, v.y, v.z]).collect();
        s.check(unsafe { rtx::sys::rtxh_tex_noise_tables(s.raw, self.scale, rv.as_ptr(), self.noise.perm_x.as_ptr(),
                                                        self.noise.perm_y.as_ptr(), self.noise.perm_z.as_ptr()) })
    }
}
impl Texture for ImageTexture {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        match &self.image {                                   // image.rs:6-10; None = the cyan fallback (image.rs:61-63)
            Some(img) => { let rgb = img.to_rgb8();
                           s.check(unsafe { rtx::sys::rtxh_tex_image(s.raw, self.width, self.height, rgb.as_raw().as_ptr()) }) }
            None => s.check(unsafe { rtx::sys::rtxh_tex_image(s.raw, 0, 0, std::ptr::null()) }),
        }
    }
}

// ===== materials/lambertian.rs, metal.rs, dielectric.rs, diffuse_light.rs, isotropic.rs =============================
impl Material for Lambertian {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let t = self.albedo.record(s)?; s.check(unsafe { rtx::sys::rtxh_mat_lambertian(s.raw, t) })
    }
}
impl Material for Metal {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        s.check(unsafe { rtx::sys::rtxh_mat_metal(s.raw, self.albedo.x, self.albedo.y, self.albedo.z, self.fuzz) })
    }
}
impl Material for Dielectric {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        s.check(unsafe { rtx::sys::rtxh_mat_dielectric(s.raw, self.refraction_index) })
    }
}
impl Material for DiffuseLight {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let t = self.emit.record(s)?; s.check(unsafe { rtx::sys::rtxh_mat_diffuse_light(s.raw, t) })
    }
}
impl Material for Isotropic {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let t = self.albedo.record(s)?; s.check(unsafe { rtx::sys::rtxh_mat_isotropic(s.raw, t) })
    }
}

// ===== geometry/sphere.rs (fields: sphere.rs:12-17; center is a Ray: origin + time * direction) ====================
impl Hittable for Sphere {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let m = s.once(key(&self.mat), |s| self.mat.record(s))?;
        let c0 = self.center.orig; let c1 = c0 + self.center.dir;
        let (a, b) = ([c0.x, c0.y, c0.z], [c1.x, c1.y, c1.z]);
        if self.center.dir == Vec3::zeros() { s.check(unsafe { rtx::sys::rtxh_sphere(s.raw, a.as_ptr(), self.radius, m) }) }
        else { s.check(unsafe { rtx::sys::rtxh_sphere_moving(s.raw, a.as_ptr(), b.as_ptr(), self.radius, m) }) }
    }
}
// ===== geometry/quad.rs (quad.rs:12-22; the constructor's derived fields are recomputed by rtxh_quad, quad.rs:27-50)
impl Hittable for Quad {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let m = s.once(key(&self.mat), |s| self.mat.record(s))?;
        let (q, u, v) = ([self.q.x, self.q.y, self.q.z], [self.u.x, self.u.y, self.u.z], [self.v.x, self.v.y, self.v.z]);
        s.check(unsafe { rtx::sys::rtxh_quad(s.raw, q.as_ptr(), u.as_ptr(), v.as_ptr(), m) })
    }
}
// ===== geometry/hittable_list.rs (box_new returns a list of six quads, quad.rs:125-174: recorded like any list) ====
impl Hittable for HittableList {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let l = s.check(unsafe { rtx::sys::rtxh_list_new(s.raw) })?;
        for o in &self.objects {
            let h = s.once(key(o), |s| o.record(s))?;
            s.check(unsafe { rtx::sys::rtxh_list_add(s.raw, l, h) })?;
        }
        Ok(l)
    }
}
// ===== acceleration/bvh.rs: `pub struct BvhNode { left, right, bbox, source: Option<Vec<Arc<dyn Hittable>>> }`;
//       BvhNode::new stores Some(list.objects.clone()), from_slice's inner nodes store None =========================
impl Hittable for BvhNode {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let src = self.source.as_ref().ok_or(rtx::Error::Unsupported("inner BvhNode recorded on its own"))?;
        let l = s.check(unsafe { rtx::sys::rtxh_list_new(s.raw) })?;
        for o in src {
            let h = s.once(key(o), |s| o.record(s))?;
            s.check(unsafe { rtx::sys::rtxh_list_add(s.raw, l, h) })?;
        }
        s.check(unsafe { rtx::sys::rtxh_bvh(s.raw, l) })      // repeats bvh.rs:26-71 on the same list: the same tree
    }
}
// ===== geometry/transforms/translate.rs, rotate_y.rs ================================================================
impl Hittable for Translate {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let o = s.once(key(&self.object), |s| self.object.record(s))?;
        let off = [self.offset.x, self.offset.y, self.offset.z];
        s.check(unsafe { rtx::sys::rtxh_translate(s.raw, o, off.as_ptr()) })
    }
}
impl Hittable for RotateY {   // keeps `angle: f64` (degrees) next to sin_theta / cos_theta: rtxh_rotate_y repeats rotate_y.rs:20-24
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let o = s.once(key(&self.object), |s| self.object.record(s))?;
        s.check(unsafe { rtx::sys::rtxh_rotate_y(s.raw, o, self.angle) })
    }
}
// ===== volumes/constant_medium.rs (keeps `density: f64` and `texture: TexturePtr` as given to new, 22-38) ==========
impl Hittable for ConstantMedium {
    fn record(&self, s: &mut rtx::Session) -> Result<i32, rtx::Error> {
        let b = s.once(key(&self.boundary), |s| self.boundary.record(s))?;
        let t = self.texture.record(s)?;
        s.check(unsafe { rtx::sys::rtxh_medium(s.raw, b, self.density, t) })
    }
}
