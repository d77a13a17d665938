// Host-only check of the BVH builder (raytracing_amd/csrc/rtx_bvh.cpp), compiled and run by
// tests/test_bvh_builder.py.  Reads objects (count, then rtx_prim records), builds the tree and prints, as JSON:
// node count, depth, the always-tested list's length, the summed half-area of the inner nodes relative to the
// root's (what a traversal's node visits follow), how often each object appears in a leaf, whether every child box
// stored in a parent contains the boxes stored in that child, and every leaf's box by object.
#include "rtx_bvh.h"
#include <cstdio>
#include <cstring>
#include <algorithm>
using namespace rtx;
static float asf(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
struct Box { float lo[3], hi[3]; };
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return 2;
    std::vector<rtx_prim> prims(n);
    if (n && fread(prims.data(), sizeof(rtx_prim), n, f) != n) return 2;
    fclose(f);
    BvhHost H;
    build_bvh(prims.data(), n, H);
    const size_t nn = H.nodes.size() / kBvhNodeVec;
    std::vector<Box> box(2 * nn);
    std::vector<int32_t> ref(2 * nn);
    for (size_t k = 0; k < nn; ++k) {
        const uint4 a = H.nodes[4 * k], b = H.nodes[4 * k + 1], c = H.nodes[4 * k + 2], e = H.nodes[4 * k + 3];
        box[2 * k] = Box{{asf(a.x), asf(a.z), asf(b.x)}, {asf(a.y), asf(a.w), asf(b.y)}};
        box[2 * k + 1] = Box{{asf(c.x), asf(c.z), asf(e.x)}, {asf(c.y), asf(c.w), asf(e.y)}};
        ref[2 * k] = (int32_t)b.z;
        ref[2 * k + 1] = (int32_t)b.w;
    }
    auto area = [&](const Box &x) {
        double e0 = ((double)x.hi[0] - x.lo[0]) * H.grid_step[0], e1 = ((double)x.hi[1] - x.lo[1]) * H.grid_step[1],
               e2 = ((double)x.hi[2] - x.lo[2]) * H.grid_step[2];
        return e0 * e1 + e1 * e2 + e2 * e0;
    };
    std::vector<uint32_t> seen(n, 0);
    double inner = 0.0, root_area = 1.0;
    long not_contained = 0, bad_ref = 0;
    if (nn) {
        Box r = box[0];
        if (ref[1] != kBvhEmpty)
            for (int a = 0; a < 3; ++a) { r.lo[a] = std::min(r.lo[a], box[1].lo[a]); r.hi[a] = std::max(r.hi[a], box[1].hi[a]); }
        root_area = area(r);
        inner = 1.0;
    }
    // depth by walking (the tree is at most kBvhMaxDepth deep if the builder kept its word)
    int depth = 0;
    std::vector<std::pair<int32_t, int>> todo;
    std::vector<uint32_t> visited(nn, 0);
    if (nn) todo.emplace_back(0, 1);
    while (!todo.empty()) {
        const auto [k, d] = todo.back();
        todo.pop_back();
        visited[(size_t)k]++;
        for (int c = 0; c < 2; ++c) {
            const int32_t r = ref[2 * (size_t)k + c];
            if (r == kBvhEmpty) continue;
            depth = std::max(depth, d);
            if (r < 0) {
                if ((uint32_t)~r >= n) bad_ref++;
                else seen[(size_t)~r]++;
                continue;
            }
            if ((size_t)r >= nn || r <= k) { bad_ref++; continue; }   // breadth-first order: children come later
            inner += area(box[2 * (size_t)k + c]) / root_area;
            const Box &p = box[2 * (size_t)k + c];
            for (int cc = 0; cc < 2; ++cc) {
                if (ref[2 * (size_t)r + cc] == kBvhEmpty) continue;
                const Box &q = box[2 * (size_t)r + cc];
                for (int a = 0; a < 3; ++a)
                    if (q.lo[a] < p.lo[a] || q.hi[a] > p.hi[a]) not_contained++;
            }
            todo.emplace_back(r, d + 1);
        }
    }
    long unvisited = 0, twice = 0;
    for (size_t k = 0; k < nn; ++k) { unvisited += visited[k] == 0; twice += visited[k] > 1; }
    printf("{\"nodes\": %zu, \"depth\": %d, \"builder_depth\": %d, \"always_tested\": %zu, \"inner_area\": %.6f, \"not_contained\": %ld, "
           "\"bad_refs\": %ld, \"unvisited_nodes\": %ld, \"nodes_reached_twice\": %ld, \"far_slope\": %.9g,\n \"always_tested_list\": [",
           nn, depth, H.depth, H.planes.size(), inner, not_contained, bad_ref, unvisited, twice, (double)H.far_slope);
    for (size_t i = 0; i < H.planes.size(); ++i) printf("%s%d", i ? ", " : "", H.planes[i]);
    printf("],\n \"leaf_count_by_object\": [");
    for (uint32_t i = 0; i < n; ++i) printf("%s%u", i ? ", " : "", seen[i]);
    printf("],\n \"leaf_boxes\": {");
    bool first = true;
    for (size_t k = 0; k < nn; ++k)
        for (int c = 0; c < 2; ++c) {
            const int32_t r = ref[2 * k + c];
            if (r == kBvhEmpty || r >= 0) continue;
            const Box &b = box[2 * k + c];
            printf("%s\"%d\": [%g, %g, %g, %g, %g, %g]", first ? "" : ", ", ~r, b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2]);
            first = false;
        }
    printf("}}\n");
    return 0;
}
