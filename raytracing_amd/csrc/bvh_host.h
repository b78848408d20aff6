// bvh_host.h -- host side of the tracer: BVH construction in the reference's node format and the
// flattening of that tree into the arrays the GPU kernels walk.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "rtgi_hip.h"

namespace rtgi {

struct BuildConfig {
	int layout = RTGI_LAYOUT_FLAT;
	bool esc = false;
	int split = RTGI_SPLIT_OM;
	int planes = 0;    // only meaningful for RTGI_SPLIT_SAH ("bvh sah N")
	int max_tris = 1;  // "bvh triangles multiple N"
};

// Strided read-only view of scene->vertices[].pos and scene->triangles[].{a,b,c}.
struct SceneView {
	const unsigned char *vertices = nullptr;
	size_t vertex_stride = 0;
	uint32_t n_vertices = 0;
	const unsigned char *triangles = nullptr;
	size_t tri_stride = 0;
	uint32_t n_triangles = 0;
	const float *pos(uint32_t v) const { return reinterpret_cast<const float *>(vertices + (size_t)v * vertex_stride); }
	const uint32_t *tri(uint32_t t) const { return reinterpret_cast<const uint32_t *>(triangles + (size_t)t * tri_stride); }
};

// The tree exactly as binary_bvh_tracer keeps it (rt/bbvh-base/bvh.h:59-68).
struct HostTree {
	std::vector<rtgi_node56> nodes;
	std::vector<uint32_t> order;  // leaf-order slot -> original triangle id
	uint32_t root = 0;
	uint32_t depth = 0;  // deepest leaf, root = 0
};

// GPU form. Inner nodes only, 64 bytes each as four float4, the two child boxes interleaved per coordinate:
//   [0] = l.min.x r.min.x l.min.y r.min.y    [1] = l.min.z r.min.z l.max.x r.max.x
//   [2] = l.max.y r.max.y l.max.z r.max.z    [3] = bits(link_l) bits(link_r) 0 0
// A link >= 0 is an inner-node index; a link < 0 is ~slot of the first triangle of a leaf.
// Nodes with an index below implicit_limit are numbered like a heap (root 1, children of i: 2i and 2i + 1, all of
// them inner nodes): their links are stored like all others, but a reader need not fetch them.
// Triangles, 48 bytes each as three float4, in leaf order:
//   [0] = A.xyz, bits(id reported in rtgi_hit.ref)
//   [1] = (A - B).xyz, bits(1 if last triangle of its leaf else 0)
//   [2] = (A - C).xyz, 0
struct FlatTree {
	std::vector<float> nodes;  // 16 floats per inner node
	std::vector<float> tris;   // 12 floats per slot
	std::vector<float> corners;  // 12 floats (A, B, C as float4) per id that a hit can report
	int32_t root_ref = 0;
	uint32_t n_inner = 0, n_slots = 0;  // n_inner: record slots in `nodes` (a few stay empty)
	uint32_t implicit_limit = 0;
	uint32_t heap_levels = 0;  // levels 0 .. heap_levels - 1 are heap-numbered (0 or >= 2); implicit_limit = 2^(heap_levels - 1)
	uint32_t depth = 0;
};

bool check_scene(const SceneView &scene, std::string &err);
bool build_tree(const BuildConfig &cfg, const SceneView &scene, HostTree &out, std::string &err);
// Checks a tree handed in from outside and computes its depth.
bool validate_tree(HostTree &tree, uint32_t n_triangles, std::string &err);
void flatten_tree(const HostTree &tree, const SceneView &scene, int layout, FlatTree &out);
std::string tree_statistics(const HostTree &tree);
bool export_tree(const HostTree &tree, uint32_t max_depth, const std::string &filename, std::string &err);

}  // namespace rtgi
