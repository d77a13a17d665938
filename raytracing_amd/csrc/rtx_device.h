// rtx_device.h -- data shared between the host side of librtx and its kernels.
//
// Device scene layout (one hipMalloc block, all arrays 16-byte aligned):
//   geom[3*i+0] = rotation quaternion (x, y, z, w)          \  the 48 bytes a closest-hit
//   geom[3*i+1] = position xyz, kind | flags << 8 (int bits)   > scan reads per object
//   geom[3*i+2] = param xyz (normal | radii | half-sizes), 0 /
//   mat[3*i+0]  = color rgb, material (int bits) in w        \  the 48 bytes shading reads
//   mat[3*i+1]  = emission rgb, ior in w                      > for the object that was hit
//   mat[3*i+2]  = color / pi (IEEE division at upload),      /
//                 w = 1 / (8 (ab + bc + ac)) for a box (its surface pdf), else 0
//   lights[k]   = object index of light k (emissive ellipsoids and boxes, object order)
//   light_bounds[2k] = bounding sphere of light k: centre xyz, squared radius inflated by 2 %
//   light_bounds[2k+1].x = slack: the sphere counts as larger by slack * |centre - ray origin|^2
//   scan_order[] = the objects by category (kCat*), see below
// The scan data and the shading data are separate arrays so that the scan streams 48
// contiguous bytes per object and never touches colours.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtx {

enum : int { kPlane = 0, kEllipsoid = 1, kBox = 2 };
enum : int { kDiffuse = 0, kMetallic = 1, kDielectric = 2 };
enum : int { kGeomIdentity = 1, kGeomPow2 = 2, kGeomPlainBoxLight = 4 };  // Geom::flags

// The closest-hit scan visits the objects category by category, so that each category's loop holds
// one figure's code with the pose and parameter shortcuts decided at compile time instead of a
// dispatch per object.  scan_order[0 .. kScanCats) = end of each category in the list that
// follows (kScanHeader entries); scan_order[kScanHeader + k] = object index, increasing within a category; two entries of
// padding (0) at the end, which the scan's read-ahead may load.
enum : int {
    kCatPlaneX = 0, kCatPlaneY, kCatPlaneZ,   // identity pose, normal +-1 along that axis and +-0 along the others
    kCatPlaneIdentity, kCatPlane,
    kCatBoxIdentity, kCatBox,
    kCatEllipsoidIdentityPow2, kCatEllipsoidIdentity, kCatEllipsoidPow2, kCatEllipsoid,
    kScanCats
};
constexpr int kScanHeader = 12;   // kScanCats rounded up to whole 16-byte words

// event counters kept per lane when statistics are requested
enum : int {
    C_SAMPLES = 0, C_RAYS, C_HIT_PLANE, C_HIT_ELLIPSOID, C_HIT_BOX, C_COSINE,
    C_LSAMPLE_ELLIPSOID, C_LSAMPLE_BOX, C_PDF_EVALS, C_LPDF_0, C_LPDF_1, C_LPDF_2,
    C_DIFFUSE, C_METALLIC, C_DIELECTRIC, C_LTEST_ELLIPSOID, C_LTEST_BOX,
    C_PTEST_PLANE, C_PTEST_ELLIPSOID, C_PTEST_BOX, C_NODE_TESTS, C_COUNT
};

struct BvhView {
    const uint4 *nodes;    // kBvhNodeVec 16-byte words per node, see rtx_bvh.h
    float grid_base[3];    // node boxes are in grid coordinates: q = (x - grid_base) * grid_inv_step
    float grid_inv_step[3];
    float grid_step[3];    // world units per grid unit
    float far_slope;       // growth of the boxes per world unit the ray origin lies outside the grid
    const int32_t *planes; // objects that are tested for every ray
    uint32_t n_nodes, n_planes;
    uint32_t stack_entries;   // postponed nodes a traversal can hold: the tree's depth + 1
    uint32_t leaf_cost;         // megakernel: lanes at a leaf wait until they outnumber the descending ones this many times
    uint32_t exit_sixteenths;   // megakernel: the wave leaves its traversal loop when fewer than this many sixteenths of its
                                // active lanes still traverse (the others go on to their surface interaction)
};

// kernel parameters: passed by value, so every field sits in scalar registers
struct RenderParams {
    const float4 *geom;
    const float4 *mat;
    const int32_t *lights;
    const float4 *light_bounds;  // two float4 per light, see above
    const int32_t *scan_order;   // kScanHeader + n_prims + 2 entries, 16-byte aligned, see above
    uint32_t n_prims, n_lights;
    uint32_t width, height, ray_depth;
    uint32_t sample_first, sample_count;
    float bg[3];
    float cam_pos[3];
    float cam_axis[9]; // columns right, up, forward
    float tg_fov_x, tg_fov_y;
    uint32_t seed_lo, seed_hi;
    uint32_t rng_const;     // 1: every draw block = const_words
    uint32_t const_words[4];
    float *out_sum;         // width*height*3, row 0 = top
    uint32_t accumulate;    // 1: the sums continue from the values already in out_sum
    uint32_t stop_on_zero;  // 1: a path whose throughput is exactly (0, 0, 0) ends (tame scenes only)
    unsigned long long *stats; // C_COUNT counters or nullptr
    // Dispatch order of the 8x8 tiles (see launch_tile_order): wave number -> tile, or nullptr for
    // the identity; and where a recording launch leaves each tile's cost (nullptr: not recorded).
    const uint32_t *tile_order;
    uint32_t *tile_cost;
    // Work queue of the two-pixel scan kernel (nullptr: every wave runs the unit of its launch slot for
    // all samples).  The sample range is cut into n_chunks chunks -- halves of what is left, the last
    // two alike -- and the launch has a wave per (chunk, unit) item, each drawing its item from
    // queue[0], a ticket counter (whatever order the hardware starts the waves in, a lower ticket is held
    // by a wave that has started): all units' first chunks in tile_order, then all second ones, and so
    // on.  queue[1 + unit] = chunks
    // of that unit that are complete: a unit's chunk k + 1 continues the sums chunk k left, so it waits
    // for that count to reach k + 1.  The end of a launch is then made of short items, and the order
    // in which a pixel's samples are added does not change.  All zero before the launch.
    // Only the end of a launch needs short items, and every item costs a wave's start and end (measured: as
    // much as 2.8 samples of each of its pixels), so the first n_whole units of the order stay whole.
    uint32_t *queue;
    uint32_t n_chunks;
    uint32_t n_whole;
    // The launch's share of the frame.  Work units (8x8 tiles, 16x8 blocks of the two-pixel kernel, a wavefront
    // workgroup's tiles) are dealt out round-robin to the GPUs of a job: this launch runs units share_index,
    // share_index + share_count, ... and touches no other pixel.  A pixel's samples are all traced and added on
    // one GPU, in order, so the frame put together from the shares has the bits of a one-GPU frame.
    // n_units = how many units that is (set by the launchers); tile_order, when given, lists them first.
    uint32_t share_index, share_count;
    uint32_t n_units;
    uint32_t ellipsoids_pow2;      // 1: every ellipsoid's radii are powers of two (kGeomPow2)
    uint32_t paths_end_at_lights;  // 1: a small all-diffuse scene whose lights are black (set at upload; picks the two-pixel scan kernel)
    uint32_t pool_entries;  // megakernel: float4 entries of a wave's hand-over pool in LDS (set by the launcher)
    BvhView bvh;
};

struct PrimaryParams {
    int32_t *hit_id;
    float *t;
    float *normal;
    uint8_t *inside;
};

// units share `index` of `count` takes out of n_units dealt round-robin
__host__ __device__ inline uint32_t share_units(uint32_t n_units, uint32_t index, uint32_t count)
{
    return n_units > index ? (n_units - index + count - 1u) / count : 0u;
}

// launchers implemented in the .hip files
enum : int { kSrcLds = 0, kSrcScalar = 1, kSrcBvh = 2 };
// scenes with more than kFewLights lights run the kernels built with the candidate-mask light pdf
constexpr uint32_t kFewLights = 2;
// pool_limit caps the hand-over pool of a wave (entries; kPoolEntriesMax = whatever fits the LDS)
hipError_t launch_megakernel(const RenderParams &p, int src, bool stats, uint32_t pool_limit, hipStream_t stream);
// work units of a megakernel launch: what tile_order permutes and tile_cost is indexed by
uint32_t megakernel_units(const RenderParams &p, int src);
// whether the megakernel chosen for (p, src) is the two-pixel scan kernel
bool megakernel_two_pixels(const RenderParams &p, int src);
// The unit a pixel belongs to under the kernel chosen for (p, src, wavefront or not): what decides which share
// of a multi-GPU job owns the pixel.  8x8 tiles in row-major order, `tiles` consecutive ones per unit, or
// (pair) blocks of 16x8 pixels in row-major order.
struct UnitShape {
    uint32_t pair, tiles;
};
UnitShape megakernel_unit_shape(const RenderParams &p, int src);
inline uint32_t unit_of_pixel(UnitShape s, uint32_t width, uint32_t x, uint32_t y)
{
    if (s.pair) return (y >> 3) * ((width + 15u) >> 4) + (x >> 4);
    return ((y >> 3) * ((width + 7u) >> 3) + (x >> 3)) / s.tiles;
}
// order[0..n_tiles) = the tiles by decreasing cost[] (buckets of one unit, costs above
// kTileCostMax share the first bucket)
constexpr uint32_t kTileCostMax = 2047u;
constexpr uint32_t kPoolEntriesMax = 1023u;
hipError_t launch_zero_words(uint32_t *p, uint64_t n, hipStream_t stream);
hipError_t launch_tile_order(const uint32_t *cost, uint32_t *order, uint32_t n_tiles, hipStream_t stream);
hipError_t launch_primary(const RenderParams &p, const PrimaryParams &out, int src, hipStream_t stream);
hipError_t launch_divide(float *d, uint64_t n, float divisor, hipStream_t stream);
hipError_t launch_tonemap(const float *sum, uint64_t n, float divisor, uint8_t *bytes, hipStream_t stream);
size_t scene_lds_bytes(uint32_t n_prims, uint32_t n_lights);

}  // namespace rtx
