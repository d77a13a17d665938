// ---- src/image.rs ------------------------------------------------------------------------
// new method in `impl Image` (after `set`, image.rs:26-29).  `data` is row-major with row 0 =
// the TOP row -- `set(u, v, ..)` writes row `height - 1 - v` -- which is the layout librtx
// writes; `Vec3` is three contiguous f32.
    pub fn data_mut(&mut self) -> &mut [Vec3] {
        &mut self.data
    }
