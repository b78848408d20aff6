// bvh_host.cpp -- BVH construction (object median / spatial median / binned SAH, optional early split
// clipping) producing the same tree as rt/bbvh-base/bvh2.cpp of the reference, and flattening for the GPU.
//
// The tree must be the reference's tree bit for bit: which boxes a ray enters decides which triangle wins
// a near-tie, so every comparison and every rounding step below keeps the reference's form (noted per
// function as file:line of "Assignment 3/rtgi-2021-a03-ref/"). Build with -ffp-contract=off.
#include "bvh_host.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <exception>
#include <sstream>
#include <system_error>
#include <thread>

namespace rtgi {
namespace {

struct F3 {
	float v[3];
	float operator[](int i) const { return v[i]; }
	float &operator[](int i) { return v[i]; }
};

// aabb of libgi/intersect.h:5-16; grow() keeps glm's operand order: min = (min < p) ? min : p,
// max = (p < max) ? max : p.
struct Bounds {
	F3 lo{{FLT_MAX, FLT_MAX, FLT_MAX}}, hi{{-FLT_MAX, -FLT_MAX, -FLT_MAX}};
	void add(const F3 &p) {
		for (int k = 0; k < 3; ++k) {
			lo[k] = (lo[k] < p[k]) ? lo[k] : p[k];
			hi[k] = (p[k] < hi[k]) ? hi[k] : p[k];
		}
	}
	void add(const Bounds &b) {
		for (int k = 0; k < 3; ++k) {
			lo[k] = (lo[k] < b.lo[k]) ? lo[k] : b.lo[k];
			hi[k] = (b.hi[k] < hi[k]) ? hi[k] : b.hi[k];
		}
	}
	float surface() const {  // bvh2.cpp:315-318
		float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
		return 2 * (ex * ey + ex * ez + ey * ez);
	}
	// axis choice of bvh2.cpp:212-222: x wins ties over y, y over z
	int longest_axis(F3 &ext) const {
		for (int k = 0; k < 3; ++k) ext[k] = hi[k] - lo[k];
		float m = (ext[1] < ext[2]) ? ext[2] : ext[1];
		float largest = (ext[0] < m) ? m : ext[0];
		return largest == ext[0] ? 0 : (largest == ext[1] ? 1 : 2);
	}
};

F3 load_pos(const SceneView &s, uint32_t v) {
	const float *p = s.pos(v);
	return F3{{p[0], p[1], p[2]}};
}

void store_box(float *dst6, const Bounds &b) {
	for (int k = 0; k < 3; ++k) {
		dst6[k] = b.lo[k];
		dst6[3 + k] = b.hi[k];
	}
}

class Builder {
public:
	Builder(const BuildConfig &cfg, const SceneView &scene, HostTree &out) : cfg(cfg), scene(scene), out(out) {}

	bool run(std::string &err) {
		const uint32_t n = scene.n_triangles;
		boxes.resize(n);
		owner.resize(n);
		slots.resize(n);
		for (uint32_t i = 0; i < n; ++i) {  // bvh2.cpp:35-43
			const uint32_t *t = scene.tri(i);
			boxes[i].add(load_pos(scene, t[0]));
			boxes[i].add(load_pos(scene, t[1]));
			boxes[i].add(load_pos(scene, t[2]));
			owner[i] = i;
			slots[i] = i;
		}
		if (cfg.esc && !clip_large_triangles(err)) return false;
		centre.resize(boxes.size());
		for (size_t i = 0; i < boxes.size(); ++i)  // prim::center(), bvh.h:52
			for (int k = 0; k < 3; ++k) centre[i][k] = (boxes[i].lo[k] + boxes[i].hi[k]) * 0.5f;

		// The two halves of a range are independent (each call sorts and partitions only its own part of
		// `slots`), so the first levels of the recursion hand their left half to another thread. Every thread
		// numbers its nodes from 0 in an arena of its own; joining appends the arenas in the reference's
		// pre-order (parent, left subtree, right subtree), which gives the node ids a single thread would.
		unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
		if (const char *e = getenv("RTGI_HIP_BUILD_THREADS")) threads = (unsigned)std::max(1, std::min(64, atoi(e)));
		int fork_levels = 0;
		while ((1u << fork_levels) < threads) ++fork_levels;
		Arena arena;
		arena.nodes.reserve(2 * boxes.size());
		out.root = split_range(0, (uint32_t)boxes.size(), 0, arena, fork_levels);
		out.nodes = std::move(arena.nodes);
		out.depth = arena.deepest;

		// commit_shuffled_triangles, bvh2.cpp:172-190: slot -> original triangle id
		out.order.resize(slots.size());
		for (size_t s = 0; s < slots.size(); ++s) out.order[s] = owner[slots[s]];
		return true;
	}

private:
	const BuildConfig &cfg;
	const SceneView &scene;
	HostTree &out;
	std::vector<Bounds> boxes;    // one per primitive box (more than triangles after clipping)
	std::vector<uint32_t> owner;  // primitive box -> triangle
	std::vector<F3> centre;
	std::vector<uint32_t> slots;  // the 'index' array the reference sorts

	struct Arena {
		std::vector<rtgi_node56> nodes;
		uint32_t deepest = 0;
	};
	static constexpr uint32_t FORK_MIN = 1u << 14;  // smaller ranges are not worth a thread

	static uint32_t new_node(Arena &a) {
		a.nodes.emplace_back();
		rtgi_node56 &n = a.nodes.back();
		store_box(n.box_l, Bounds());
		store_box(n.box_r, Bounds());
		n.link_l = n.link_r = 0;
		return (uint32_t)a.nodes.size() - 1;
	}
	static uint32_t leaf(uint32_t start, uint32_t end, uint32_t depth, Arena &a) {
		uint32_t id = new_node(a);
		a.nodes[id].link_l = -(int32_t)start;
		a.nodes[id].link_r = -(int32_t)(end - start);
		if (depth > a.deepest) a.deepest = depth;
		return id;
	}
	// Appends a subtree built in its own arena (root = its node 0); returns the root's id in `a`.
	static uint32_t append(Arena &a, const Arena &sub) {
		const int32_t offset = (int32_t)a.nodes.size();
		for (rtgi_node56 n : sub.nodes) {
			if (n.link_r >= 0) n.link_l += offset, n.link_r += offset;  // inner node: links are node ids
			a.nodes.push_back(n);
		}
		if (sub.deepest > a.deepest) a.deepest = sub.deepest;
		return (uint32_t)offset;
	}
	Bounds range_bounds(uint32_t start, uint32_t end) const {
		Bounds b;
		for (uint32_t i = start; i < end; ++i) b.add(boxes[slots[i]]);
		return b;
	}
	void sort_by_centre(uint32_t start, uint32_t end, int axis) {
		std::sort(slots.data() + start, slots.data() + end,
		          [this, axis](uint32_t a, uint32_t b) { return centre[a][axis] < centre[b][axis]; });
	}
	// Two-pointer partition of bvh2.cpp:265-277 / :338-353. The slot on which the pointers meet is
	// left unexamined on the right-hand side, exactly as there.
	uint32_t partition_at(uint32_t start, uint32_t end, int axis, float plane) {
		uint32_t mid = start, l = start, r = end - 1;
		while (l < r) {
			while (centre[slots[l]][axis] <= plane && l < r) {
				++l;
				++mid;
			}
			while (centre[slots[r]][axis] > plane && l < r) --r;
			if (centre[slots[l]][axis] > centre[slots[r]][axis] && l < r) std::swap(slots[l], slots[r]);
		}
		return mid;
	}

	uint32_t split_range(uint32_t start, uint32_t end, uint32_t depth, Arena &arena, int fork_levels) {
		if (end - start <= (uint32_t)cfg.max_tris) return leaf(start, end, depth, arena);
		Bounds all = range_bounds(start, end);
		F3 ext;
		int axis = all.longest_axis(ext);
		uint32_t mid = start;
		Bounds left, right;
		bool have_child_boxes = false;

		if (cfg.split == RTGI_SPLIT_OM) {  // bvh2.cpp:192-235
			sort_by_centre(start, end, axis);
			mid = start + (end - start) / 2;
		} else if (cfg.split == RTGI_SPLIT_SM) {  // bvh2.cpp:237-298
			float plane = all.lo[axis] + (all.hi[axis] - all.lo[axis]) * 0.5f;
			mid = partition_at(start, end, axis, plane);
			if (mid == start || mid == end - 1) {
				sort_by_centre(start, end, axis);
				mid = start + (end - start) / 2;
			}
		} else {  // bvh2.cpp:300-413
			float cost_l = FLT_MAX, cost_r = FLT_MAX;
			bool median_fallback_left = true;
			int n_planes = cfg.planes;
			if ((int)(end - start) < n_planes) n_planes = (int)(end - start);
			const float whole = all.surface();
			for (int i = 0; i < n_planes; ++i) {
				float plane = all.lo[axis] + (i + 1) * (ext[axis] / (n_planes + 1));
				uint32_t m = partition_at(start, end, axis, plane);
				if (m == start || m == end - 1) {
					if (!median_fallback_left) continue;
					sort_by_centre(start, end, axis);
					m = start + (end - start) / 2;
					median_fallback_left = false;
				}
				Bounds bl = range_bounds(start, m), br = range_bounds(m, end);
				float cl = (bl.surface() / whole) * (m - start);
				float cr = (br.surface() / whole) * (end - m);
				if (cl + cr < cost_l + cost_r) {
					left = bl;
					right = br;
					cost_l = cl;
					cost_r = cr;
					mid = m;
				}
			}
			have_child_boxes = true;
			if (mid <= start || mid >= end) {
				// No plane was accepted: the range's box has no surface (all primitives in one point or on one
				// axis-parallel line), so every cost is NaN or inf and never below the best so far. The reference
				// would recurse into an empty range here and stop at its assert (bvh2.cpp:302); split at the
				// object median instead, which always makes progress.
				sort_by_centre(start, end, axis);
				mid = start + (end - start) / 2;
				have_child_boxes = false;
			}
			if (cfg.max_tris > 1)  // K_T = K_I = 1, bvh2.cpp:13-14 and :395-403
				if ((float)(end - start) < (1 + 1 * (cost_l + cost_r)) && (end - start) <= (uint32_t)cfg.max_tris)
					return leaf(start, end, depth, arena);
		}

		uint32_t id = new_node(arena);
		uint32_t l, r;
		if (fork_levels > 0 && end - start >= FORK_MIN) {
			Arena sub_l, sub_r;
			std::exception_ptr failure;
			auto left_half = [&]() {
				try {
					split_range(start, mid, depth + 1, sub_l, fork_levels - 1);
				} catch (...) {
					failure = std::current_exception();
				}
			};
			std::thread other;
			try {
				other = std::thread(left_half);
			} catch (const std::system_error &) {  // no thread to be had: this one does both halves
			}
			try {
				split_range(mid, end, depth + 1, sub_r, fork_levels - 1);
			} catch (...) {
				if (other.joinable()) other.join();
				throw;
			}
			if (other.joinable())
				other.join();
			else
				left_half();
			if (failure) std::rethrow_exception(failure);
			l = append(arena, sub_l);
			r = append(arena, sub_r);
		} else {
			l = split_range(start, mid, depth + 1, arena, 0);
			r = split_range(mid, end, depth + 1, arena, 0);
		}
		if (!have_child_boxes) {
			left = range_bounds(start, mid);
			right = range_bounds(mid, end);
		}
		rtgi_node56 &n = arena.nodes[id];
		n.link_l = (int32_t)l;
		n.link_r = (int32_t)r;
		store_box(n.box_l, left);
		store_box(n.box_r, right);
		return id;
	}

	// split(), bvh2.cpp:65-126: cut a polygon at the middle of the longest axis of its box until every
	// piece's box surface is at most `limit`; collects the piece boxes.
	bool cut_polygon(const std::vector<F3> &poly, float limit, std::vector<Bounds> &pieces, int level) {
		Bounds b;
		for (const F3 &p : poly) b.add(p);
		if (b.surface() <= limit) {
			pieces.push_back(b);
			return true;
		}
		if (level > 64 || poly.empty()) return false;
		F3 ext;
		int axis = b.longest_axis(ext);
		const float mid = (b.hi[axis] + b.lo[axis]) * 0.5f;
		const int n = (int)poly.size();
		bool prev_left = poly[0][axis] < mid;
		int first = -1, second = -1;
		for (int i = 0; i < n; ++i) {
			bool is_left = poly[i][axis] < mid;
			if (is_left != prev_left) {
				if (first == -1)
					first = i - 1;
				else
					second = i - 1;
			}
			prev_left = is_left;
		}
		if (second == -1) second = n - 1;
		if (first == -1) return false;  // bvh2.cpp:95 asserts
		auto crossing = [&](int a) {
			int nb = a + 1 < n ? a + 1 : 0;
			float w = (mid - poly[a][axis]) / (poly[nb][axis] - poly[a][axis]);
			F3 p;
			for (int k = 0; k < 3; ++k) p[k] = poly[a][k] + (poly[nb][k] - poly[a][k]) * w;
			return p;
		};
		const F3 c1 = crossing(first), c2 = crossing(second);
		std::vector<F3> one(poly.begin(), poly.begin() + first + 1), two;
		one.push_back(c1);
		two.push_back(c1);
		two.insert(two.end(), poly.begin() + first + 1, poly.begin() + second + 1);
		two.push_back(c2);
		one.push_back(c2);
		one.insert(one.end(), poly.begin() + second + 1, poly.end());
		auto same = [](const F3 &p, const F3 &q) { return p[0] == q[0] && p[1] == q[1] && p[2] == q[2]; };
		one.erase(std::unique(one.begin(), one.end(), same), one.end());
		two.erase(std::unique(two.begin(), two.end(), same), two.end());
		return cut_polygon(one, limit, pieces, level + 1) && cut_polygon(two, limit, pieces, level + 1);
	}

	// early_split_clipping, bvh2.cpp:128-170: triangles whose box is larger than the 90th percentile
	// are represented by several smaller boxes.
	bool clip_large_triangles(std::string &err) {
		std::vector<float> area(boxes.size());
		for (size_t i = 0; i < boxes.size(); ++i) area[i] = boxes[i].surface();
		std::sort(area.begin(), area.end());
		const float limit = area[9 * area.size() / 10];
		const uint32_t n = (uint32_t)boxes.size();
		for (uint32_t i = 0; i < n; ++i) {
			const uint32_t *t = scene.tri(i);
			std::vector<F3> poly{load_pos(scene, t[0]), load_pos(scene, t[1]), load_pos(scene, t[2])};
			std::vector<Bounds> pieces;
			if (!cut_polygon(poly, limit, pieces, 0)) {
				err = "early split clipping: triangle " + std::to_string(i) + " cannot be split (degenerate)";
				return false;
			}
			boxes[i] = pieces[0];
			for (size_t j = 1; j < pieces.size(); ++j) {
				boxes.push_back(pieces[j]);
				owner.push_back(i);
				slots.push_back((uint32_t)boxes.size() - 1);
			}
		}
		return true;
	}
};

}  // namespace

bool check_scene(const SceneView &s, std::string &err) {
	if (!s.vertices || !s.triangles || s.n_vertices == 0 || s.n_triangles == 0) {
		err = "there is no scene data to work with";
		return false;
	}
	if (s.vertex_stride < 12 || s.tri_stride < 12 || s.vertex_stride % 4 || s.tri_stride % 4) {
		err = "vertex/triangle stride must be a multiple of 4 and at least 12 bytes";
		return false;
	}
	if (s.n_triangles > 0x3fffffffu) {
		err = "too many triangles";
		return false;
	}
	for (uint32_t v = 0; v < s.n_vertices; ++v) {
		const float *p = s.pos(v);
		if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) {
			err = "vertex " + std::to_string(v) + " has a non-finite position";
			return false;
		}
	}
	for (uint32_t t = 0; t < s.n_triangles; ++t) {
		const uint32_t *i = s.tri(t);
		if (i[0] >= s.n_vertices || i[1] >= s.n_vertices || i[2] >= s.n_vertices) {
			err = "triangle " + std::to_string(t) + " references a vertex out of range";
			return false;
		}
	}
	return true;
}

bool build_tree(const BuildConfig &cfg, const SceneView &scene, HostTree &out, std::string &err) {
	if (!check_scene(scene, err)) return false;
	if (cfg.layout == RTGI_LAYOUT_FLAT && cfg.esc) {
		err = "flat layout with early split clipping is technically problematic";
		return false;
	}
	if (cfg.split == RTGI_SPLIT_SAH && cfg.planes <= 0) {
		err = "\"bvh sah\" needs a positive number of planes";
		return false;
	}
	if (cfg.max_tris < 1) {
		err = "\"bvh triangles multiple\" needs a positive count";
		return false;
	}
	Builder b(cfg, scene, out);
	return b.run(err);
}

bool validate_tree(HostTree &t, uint32_t n_triangles, std::string &err) {
	const uint32_t n = (uint32_t)t.nodes.size();
	if (n == 0 || t.root >= n) {
		err = "tree is empty or the root is out of range";
		return false;
	}
	for (uint32_t id : t.order)
		if (id >= n_triangles) {
			err = "triangle index entry out of range";
			return false;
		}
	std::vector<uint8_t> seen(n, 0);
	std::vector<std::pair<uint32_t, uint32_t>> todo{{t.root, 0u}};
	t.depth = 0;
	while (!todo.empty()) {
		auto [id, d] = todo.back();
		todo.pop_back();
		if (seen[id]) {
			err = "node " + std::to_string(id) + " is reachable twice (not a tree)";
			return false;
		}
		seen[id] = 1;
		const rtgi_node56 &nd = t.nodes[id];
		if (nd.link_r >= 0) {
			if (nd.link_l < 0 || (uint32_t)nd.link_l >= n || (uint32_t)nd.link_r >= n) {
				err = "node " + std::to_string(id) + " links out of range";
				return false;
			}
			todo.push_back({(uint32_t)nd.link_l, d + 1});
			todo.push_back({(uint32_t)nd.link_r, d + 1});
		} else {
			int64_t off = -(int64_t)nd.link_l, cnt = -(int64_t)nd.link_r;
			if (off < 0 || cnt < 1 || off + cnt > (int64_t)t.order.size()) {
				err = "leaf " + std::to_string(id) + " covers slots outside the triangle index";
				return false;
			}
			if (d > t.depth) t.depth = d;
		}
	}
	return true;
}

namespace {
// fn(begin, end) over [0, n) in contiguous parts, one per thread (as many as the builder uses); serial for small n
// or when no thread can be had.
template <class F>
void parallel_ranges(uint32_t n, F fn) {
	unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
	if (const char *e = getenv("RTGI_HIP_BUILD_THREADS")) threads = (unsigned)std::max(1, std::min(64, atoi(e)));
	if (n < (1u << 16) || threads < 2) {
		fn(0u, n);
		return;
	}
	const uint32_t part = (n + threads - 1) / threads;
	std::vector<std::thread> pool;
	uint32_t begin = 0;
	for (; begin + part < n; begin += part) {
		try {
			pool.emplace_back(fn, begin, begin + part);
		} catch (const std::system_error &) {
			break;  // the calling thread does the rest
		}
	}
	fn(begin, n);
	for (std::thread &th : pool) th.join();
}
}  // namespace

void flatten_tree(const HostTree &t, const SceneView &scene, int layout, FlatTree &out) {
	const uint32_t n = (uint32_t)t.nodes.size();
	// Inner nodes are renumbered level by level (breadth first): the upper levels, which every ray visits,
	// become one contiguous block (the kernel keeps the first ones in LDS), and two inner siblings share one
	// aligned 128-byte line, so that fetching the nearer child brings in the one that waits on the stack.
	// RTGI_HIP_NODE_ORDER=dfs keeps the reference's pre-order, =bfs numbers without the alignment gaps.
	std::vector<int32_t> inner_id(n, -1);
	uint32_t n_inner = 0;
	const char *order_env = getenv("RTGI_HIP_NODE_ORDER");
	if (order_env && std::string(order_env) == "dfs") {
		for (uint32_t i = 0; i < n; ++i)
			if (t.nodes[i].link_r >= 0) inner_id[i] = (int32_t)n_inner++;
	} else if (!(order_env && std::string(order_env) == "bfs")) {
		std::vector<uint32_t> queue;
		queue.reserve(n);
		if (t.nodes[t.root].link_r >= 0) {
			queue.push_back(t.root);
			inner_id[t.root] = (int32_t)n_inner++;
		}
		for (size_t head = 0; head < queue.size(); ++head) {
			const rtgi_node56 &nd = t.nodes[queue[head]];
			const bool li = t.nodes[nd.link_l].link_r >= 0, ri = t.nodes[nd.link_r].link_r >= 0;
			if (li && ri && (n_inner & 1u)) ++n_inner;
			if (li) inner_id[nd.link_l] = (int32_t)n_inner++, queue.push_back((uint32_t)nd.link_l);
			if (ri) inner_id[nd.link_r] = (int32_t)n_inner++, queue.push_back((uint32_t)nd.link_r);
		}
	} else {
		std::vector<uint32_t> queue;
		queue.reserve(n);
		if (t.nodes[t.root].link_r >= 0) queue.push_back(t.root);
		for (size_t head = 0; head < queue.size(); ++head) {
			const rtgi_node56 &nd = t.nodes[queue[head]];
			inner_id[queue[head]] = (int32_t)n_inner++;
			if (t.nodes[nd.link_l].link_r >= 0) queue.push_back((uint32_t)nd.link_l);
			if (t.nodes[nd.link_r].link_r >= 0) queue.push_back((uint32_t)nd.link_r);
		}
	}
	auto ref_of = [&](uint32_t node) -> int32_t {
		const rtgi_node56 &nd = t.nodes[node];
		if (nd.link_r >= 0) return inner_id[node];
		return ~(int32_t)(-nd.link_l);
	};
	out.n_inner = n_inner;
	out.n_slots = (uint32_t)t.order.size();
	out.depth = t.depth;
	out.root_ref = ref_of(t.root);
	// (the three record arrays are filled by several threads: every record depends on its own node or slot only)
	out.nodes.assign((size_t)n_inner * 16, 0.0f);
	parallel_ranges(n, [&](uint32_t begin, uint32_t end) {
		for (uint32_t i = begin; i < end; ++i) {
			const rtgi_node56 &nd = t.nodes[i];
			if (nd.link_r < 0) continue;
			float *g = out.nodes.data() + (size_t)inner_id[i] * 16;
			int32_t rl = ref_of((uint32_t)nd.link_l), rr = ref_of((uint32_t)nd.link_r);
			// child boxes interleaved (l, r) per coordinate: one packed instruction serves both children
			const float *L = nd.box_l, *R = nd.box_r;  // min.xyz, max.xyz
			const float packed[12] = {L[0], R[0], L[1], R[1], L[2], R[2], L[3], R[3], L[4], R[4], L[5], R[5]};
			memcpy(g, packed, sizeof(packed));
			memcpy(g + 12, &rl, 4);
			memcpy(g + 13, &rr, 4);
		}
	});
	out.tris.assign((size_t)out.n_slots * 12, 0.0f);
	parallel_ranges(out.n_slots, [&](uint32_t begin, uint32_t end) {
		for (uint32_t s = begin; s < end; ++s) {
			const uint32_t *v = scene.tri(t.order[s]);
			const float *A = scene.pos(v[0]), *B = scene.pos(v[1]), *C = scene.pos(v[2]);
			float *g = out.tris.data() + (size_t)s * 12;
			// hit ids: position in the permuted list (flat, bvh.h:96-100) or original id (indexed, :102-106)
			uint32_t id = (layout == RTGI_LAYOUT_FLAT) ? s : t.order[s];
			for (int k = 0; k < 3; ++k) {
				g[k] = A[k];
				g[4 + k] = A[k] - B[k];  // a, b, c of libgi/intersect.h:26-28
				g[8 + k] = A[k] - C[k];  // d, e, f of :31-33
			}
			memcpy(g + 3, &id, 4);
		}
	});
	// corner positions by reported id, for the stages that turn hits into new rays
	const uint32_t n_ids = (layout == RTGI_LAYOUT_FLAT) ? out.n_slots : scene.n_triangles;
	out.corners.assign((size_t)n_ids * 12, 0.0f);
	parallel_ranges(n_ids, [&](uint32_t begin, uint32_t end) {
		for (uint32_t id = begin; id < end; ++id) {
			const uint32_t *v = scene.tri(layout == RTGI_LAYOUT_FLAT ? t.order[id] : id);
			for (int c = 0; c < 3; ++c) memcpy(out.corners.data() + (size_t)id * 12 + 4 * c, scene.pos(v[c]), 12);
		}
	});
	for (uint32_t i = 0; i < n; ++i) {
		const rtgi_node56 &nd = t.nodes[i];
		if (nd.link_r >= 0) continue;
		uint32_t last = (uint32_t)(-nd.link_l) + (uint32_t)(-nd.link_r) - 1;
		uint32_t one = 1;
		memcpy(out.tris.data() + (size_t)last * 12 + 7, &one, 4);
	}
}

// print_node_stats, bvh2.cpp:599-630 (same figures, including its min/max update order and its
// median-of-even-count rule)
std::string tree_statistics(const HostTree &t) {
	std::vector<int> counts;
	uint32_t total = 0;
	int mx = 0, mn = INT32_MAX;
	for (const rtgi_node56 &nd : t.nodes)
		if (nd.link_r < 0) {
			int c = -nd.link_r;
			counts.push_back(c);
			if (c < mn)
				mn = c;
			else if (c > mx)
				mx = c;
			total += c;
		}
	std::sort(counts.begin(), counts.end());
	int median = 0;
	size_t n = counts.size();
	if (n % 2 == 1)
		median = counts[n / 2];
	else if (n)
		median = (int)(0.5 * (counts[n / 2] + counts[std::min(n / 2 + 1, n - 1)]));
	std::ostringstream o;
	o << "number of leaf nodes: " << n << "\n"
	  << "minimum triangles per node: " << mn << "\n"
	  << "maximum triangles per node: " << mx << "\n"
	  << "average triangles per node: " << (n ? total / n : 0) << "\n"
	  << "median of triangles per node: " << median << "\n";
	return o.str();
}

// export_bvh, bvh2.cpp:557-597: child boxes of the first max_depth levels as OBJ quads
bool export_tree(const HostTree &t, uint32_t max_depth, const std::string &filename, std::string &err) {
	std::ofstream f(filename, std::ios::trunc);
	if (!f) {
		err = "cannot write " + filename;
		return false;
	}
	uint32_t box_no = 0;
	auto emit = [&](const float *b, uint32_t depth) {
		const float *lo = b, *hi = b + 3;
		uint32_t v0 = box_no++ * 8 + 1;
		f << "v " << lo[0] << " " << lo[1] << " " << lo[2] << "\n" << "v " << hi[0] << " " << lo[1] << " " << lo[2] << "\n"
		  << "v " << hi[0] << " " << hi[1] << " " << lo[2] << "\n" << "v " << lo[0] << " " << hi[1] << " " << lo[2] << "\n"
		  << "v " << lo[0] << " " << lo[1] << " " << hi[2] << "\n" << "v " << hi[0] << " " << lo[1] << " " << hi[2] << "\n"
		  << "v " << hi[0] << " " << hi[1] << " " << hi[2] << "\n" << "v " << lo[0] << " " << hi[1] << " " << hi[2] << "\n";
		f << "g level" << depth + 1 << "\n";
		static const int q[6][4] = {{0, 1, 2, 3}, {1, 5, 6, 2}, {0, 1, 5, 4}, {3, 2, 6, 7}, {4, 7, 6, 5}, {4, 0, 3, 7}};
		for (auto &face : q) f << "f " << v0 + face[0] << " " << v0 + face[1] << " " << v0 + face[2] << " " << v0 + face[3] << "\n";
	};
	struct Item {
		uint32_t node, depth;
	};
	std::vector<Item> todo{{t.root, 0}};
	while (!todo.empty()) {
		Item it = todo.back();
		todo.pop_back();
		const rtgi_node56 &nd = t.nodes[it.node];
		if (nd.link_r < 0 || it.depth >= max_depth) continue;
		emit(nd.box_l, it.depth);
		emit(nd.box_r, it.depth);
		todo.push_back({(uint32_t)nd.link_r, it.depth + 1});  // left subtree is written first
		todo.push_back({(uint32_t)nd.link_l, it.depth + 1});
	}
	return true;
}

}  // namespace rtgi
