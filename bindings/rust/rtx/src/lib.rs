//! Session / Context / Error: what the `record` hooks of reference_patch/ and the patched Camera::render use.
use std::collections::HashMap;
use std::ffi::CStr;

pub use rtx_sys as sys;

#[derive(Debug)]
pub enum Error {
    /// A trait object that has no `record` implementation (the provided default).
    Unsupported(&'static str),
    /// A constructor returned a negative handle; the session's message.
    Scene(String),
    /// A device-layer call failed; status and the context's message.
    Device(i32, String),
}

/// A recording session of the host layer (rtxh_session).  Handles are small integers local to it.
pub struct Session {
    pub raw: *mut sys::rtxh_session,
    /// Objects already recorded, by the address of their Arc's payload: an object that appears twice in the graph (the
    /// glass ball that is also a medium's boundary, final_scene.rs:107-117) is recorded once.
    seen: HashMap<usize, i32>,
}

impl Session {
    pub fn new() -> Self {
        Session { raw: unsafe { sys::rtxh_session_new() }, seen: HashMap::new() }
    }
    /// Turns a constructor's return value into a Result.
    pub fn check(&self, handle: i32) -> Result<i32, Error> {
        if handle >= 0 {
            return Ok(handle);
        }
        let msg = unsafe { CStr::from_ptr(sys::rtxh_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(Error::Scene(msg))
    }
    /// `record` of a shared object: runs `f` the first time `key` (the payload's address) is seen.
    pub fn once<F: FnOnce(&mut Session) -> Result<i32, Error>>(&mut self, key: usize, f: F) -> Result<i32, Error> {
        if let Some(&h) = self.seen.get(&key) {
            return Ok(h);
        }
        let h = f(self)?;
        self.seen.insert(key, h);
        Ok(h)
    }
}

impl Drop for Session {
    fn drop(&mut self) {
        unsafe { sys::rtxh_session_free(self.raw) }
    }
}

/// One per process and GPU.  Creation fails when no MI355X is usable: the backend has no CPU path.
pub struct Context {
    pub raw: *mut sys::rtx_context,
}

impl Context {
    pub fn new(device: i32) -> Result<Self, Error> {
        let mut raw = std::ptr::null_mut();
        let ids = [device];
        let st = unsafe { sys::rtx_context_create(1, ids.as_ptr(), &mut raw) };
        if st != sys::RTX_OK {
            let msg = unsafe { CStr::from_ptr(sys::rtx_last_error(std::ptr::null())) }.to_string_lossy().into_owned();
            return Err(Error::Device(st, msg));
        }
        Ok(Context { raw })
    }
    pub fn status(&self, st: i32) -> Result<(), Error> {
        if st == sys::RTX_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(sys::rtx_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(Error::Device(st, msg))
    }
    /// Status of a call of the host layer that also used a session (rtxh_camera_render and its progressive form): the
    /// message of whichever layer failed - the context's, the context-less one of the camera checks, else the
    /// session's (a failure while flattening leaves its message there and the context's empty).
    pub fn status_with_session(&self, st: i32, session: &Session) -> Result<(), Error> {
        if st == sys::RTX_OK {
            return Ok(());
        }
        let text = |p: *const std::os::raw::c_char| unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned();
        let mut msg = text(unsafe { sys::rtx_last_error(self.raw) });
        if msg.is_empty() {
            msg = text(unsafe { sys::rtx_last_error(std::ptr::null()) });
        }
        if msg.is_empty() {
            msg = text(unsafe { sys::rtxh_last_error(session.raw) });
        }
        Err(Error::Device(st, msg))
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::rtx_context_destroy(self.raw) }
    }
}
