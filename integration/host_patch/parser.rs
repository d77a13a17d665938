// ---- src/parser.rs -----------------------------------------------------------------------
// `Box<dyn Geometry>` erases the primitive kind (parser.rs:20), so the objects are flattened to
// plain records while `FigureType` is still known.

// (1) one more field in `pub struct Scene` (parser.rs:12-24):
//         pub pod: Vec<rtx_sys::rtx_prim>,

// (2) new function, next to `impl SceneParser`:
fn flatten_objects(figure_types: &[FigureType], objects: &[Object<Box<dyn Geometry>>]) -> Vec<rtx_sys::rtx_prim> {
    figure_types
        .iter()
        .zip(objects.iter())
        .map(|(fig_type, obj)| {
            // rtx_kind: 0 plane (normal), 1 ellipsoid (radii), 2 box (half-sizes)
            let (kind, param) = match fig_type {
                FigureType::Plane(normal) => (rtx_sys::RTX_PLANE, normal),
                FigureType::Ellipsoid(radiuses) => (rtx_sys::RTX_ELLIPSOID, radiuses),
                FigureType::Parallelipiped(sizes) => (rtx_sys::RTX_BOX, sizes),
            };
            let (material, ior) = match obj.material {
                Material::Diffuse => (rtx_sys::RTX_DIFFUSE, 1.0),
                Material::Metallic => (rtx_sys::RTX_METALLIC, 1.0),
                Material::Dielectric { ior } => (rtx_sys::RTX_DIELECTRIC, ior),
            };
            let q = obj.geometry.rotation.as_ref().coords; // nalgebra stores (i, j, k, w)
            rtx_sys::rtx_prim {
                kind,
                param: [param.x, param.y, param.z],
                position: [obj.geometry.position.x, obj.geometry.position.y, obj.geometry.position.z],
                rotation: [q.x, q.y, q.z, q.w],
                color: [obj.color.x, obj.color.y, obj.color.z],
                emission: [obj.emission.x, obj.emission.y, obj.emission.z],
                material,
                ior,
            }
        })
        .collect()
}

// (3) in `SceneParser::create_scene` (parser.rs:50), before `let lights = izip!(self.figure_types.into_iter(), ...`
//     at parser.rs:69 consumes the figure types:
//         let pod = flatten_objects(&self.figure_types, &self.objects);
//     and in the `Scene { .. }` literal at parser.rs:91-100:
//         pod,
