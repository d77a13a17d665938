// ---- src/main.rs -------------------------------------------------------------------------
// `render` replaces main.rs:13-31; `main` (main.rs:33-42) stays as it is: parse_scene, render,
// color_correction, write.  `use rand::Rng;` and `use trace::trace_ray;` (main.rs:10-11) go.

/// GPUs to deal the frame out to: RTX_GPUS=N in the environment, default 1.
fn gpu_count() -> i32 {
    std::env::var("RTX_GPUS").ok().and_then(|s| s.parse().ok()).unwrap_or(1)
}

fn render(scene: &mut Scene) {
    let axis = scene.camera.axis; // Matrix3: column-major storage = columns right, up, forward (parser.rs:64)
    let mut cam_axis = [0.0f32; 9];
    cam_axis.copy_from_slice(axis.as_slice());
    let pod = rtx_sys::rtx_scene {
        width: scene.image.width as u32,
        height: scene.image.height as u32,
        ray_depth: scene.ray_depth as u32,
        n_samples: scene.n_samples as u32,
        background: [scene.background_color.x, scene.background_color.y, scene.background_color.z],
        cam_pos: [scene.camera.position.x, scene.camera.position.y, scene.camera.position.z],
        cam_axis,
        tg_fov_x: scene.camera.tg_fov_x,
        tg_fov_y: scene.camera.tg_fov_y,
        n_prims: scene.pod.len() as u32,
        prims: scene.pod.as_ptr(),
    };
    // One context for the process would do for a host that renders more than once; this driver
    // renders one image, so the context (device buffers, events, one stream per GPU) lives for
    // this call.  The seed is fixed: the reference's thread_rng() has no seed to carry over.
    let opts = rtx_sys::rtx_opts { seed: 1, n_devices: gpu_count(), ..Default::default() };
    let mut ctx = rtx_sys::Context::new(&opts).unwrap_or_else(|e| panic!("rtx_create: {e}"));
    let pixels = scene.image.data_mut();
    let rgb = unsafe { std::slice::from_raw_parts_mut(pixels.as_mut_ptr() as *mut f32, pixels.len() * 3) };
    // deals the frame out to the GPUs, adds their buffers on the devices, divides, copies back
    ctx.render(&pod, &scene.pod, rgb).unwrap_or_else(|e| panic!("rtx_render: {e}"));
}
