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
		unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
		if (const char *e = getenv("RTGI_HIP_BUILD_THREADS")) threads = (unsigned)std::max(1, std::min(64, atoi(e)));
		const int prep_parts = n >= FORK_MIN ? (int)std::min(threads, 16u) : 1;
		boxes.resize(n);
		owner.resize(n);
		slots.resize(n);
		on_threads(prep_parts, n, [this](uint32_t lo, uint32_t hi) {
			for (uint32_t i = lo; i < hi; ++i) {  // bvh2.cpp:35-43
				const uint32_t *t = scene.tri(i);
				boxes[i].add(load_pos(scene, t[0]));
				boxes[i].add(load_pos(scene, t[1]));
				boxes[i].add(load_pos(scene, t[2]));
				owner[i] = i;
				slots[i] = i;
			}
		});
		if (cfg.esc && !clip_large_triangles(err)) return false;
		// One record per primitive box, kept in the order of the reference's `index` array: sorting and partitioning
		// move the records themselves, so that a range of the recursion is a contiguous piece of memory.
		prims.resize(boxes.size());
		on_threads(prep_parts, (uint32_t)boxes.size(), [this](uint32_t lo, uint32_t hi) {
			for (uint32_t i = lo; i < hi; ++i) {
				Prim &p = prims[i];
				p.box = boxes[slots[i]];
				p.owner = owner[slots[i]];
				for (int k = 0; k < 3; ++k) p.centre[k] = (p.box.lo[k] + p.box.hi[k]) * 0.5f;  // prim::center(), bvh.h:52
			}
		});
		std::vector<Bounds>().swap(boxes);
		std::vector<uint32_t>().swap(owner);
		std::vector<uint32_t>().swap(slots);

		// The two halves of a range are independent (each call sorts and partitions only its own part of
		// `prims`), so the first levels of the recursion hand their left half to another thread. Every thread
		// numbers its nodes from 0 in an arena of its own; the arenas are put together at the end in the reference's
		// pre-order (parent, left subtree, right subtree), which gives the node ids a single thread would.
		int fork_levels = 0;
		while ((1u << fork_levels) < threads) ++fork_levels;
		Arena arena;
		if (fork_levels == 0) arena.nodes.reserve(2 * prims.size());
		out.root = split_range(0, (uint32_t)prims.size(), 0, arena, fork_levels);
		assemble(arena, out.nodes, out.depth);
		std::vector<Keyed>().swap(keyed_buffer());
		std::vector<Prim>().swap(moved_buffer());

		// commit_shuffled_triangles, bvh2.cpp:172-190: slot -> original triangle id
		out.order.resize(prims.size());
		on_threads(prep_parts, (uint32_t)prims.size(), [this](uint32_t lo, uint32_t hi) {
			for (uint32_t s = lo; s < hi; ++s) out.order[s] = prims[s].owner;
		});
		return true;
	}

private:
	const BuildConfig &cfg;
	const SceneView &scene;
	HostTree &out;
	// while the primitive boxes are made (early split clipping adds some):
	std::vector<Bounds> boxes;    // one per primitive box (more than triangles after clipping)
	std::vector<uint32_t> owner;  // primitive box -> triangle
	std::vector<uint32_t> slots;  // the reference's 'index' array before the first split
	// while the tree is built: position i holds what the reference reaches through index[i]
	struct Prim {
		Bounds box;
		F3 centre;
		uint32_t owner;
	};
	std::vector<Prim> prims;

	// Nodes in pre-order, numbered from 0. A call of the recursion that handed its halves to other threads holds its
	// one node here and the two subtrees as arenas of their own, which follow that node in the numbering.
	struct Arena {
		std::vector<rtgi_node56> nodes;
		std::vector<Arena> then;  // subtrees numbered after `nodes`, in this order
		uint32_t deepest = 0;
		uint32_t count = 0;       // nodes.size() + the counts of `then`, kept by attach()
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
	static uint32_t size_of(const Arena &a) { return a.then.empty() ? (uint32_t)a.nodes.size() : a.count; }
	// Numbers a subtree built in its own arena (root = its node 0) after everything `a` holds so far; returns the
	// root's id in `a`. Nothing is copied before assemble().
	static uint32_t attach(Arena &a, Arena &&sub) {
		if (a.then.empty()) a.count = (uint32_t)a.nodes.size();
		const uint32_t offset = a.count;
		a.count += size_of(sub);
		if (sub.deepest > a.deepest) a.deepest = sub.deepest;
		a.then.push_back(std::move(sub));
		return offset;
	}
	// All arenas into one array, each copied once (by a thread of its own) to where its numbering starts; the links of
	// inner nodes are ids inside their arena and move with it.
	static void assemble(Arena &root, std::vector<rtgi_node56> &nodes, uint32_t &deepest) {
		deepest = root.deepest;
		if (root.then.empty()) {
			nodes = std::move(root.nodes);
			return;
		}
		struct Piece {
			const Arena *arena;
			uint32_t at;
		};
		std::vector<Piece> pieces;
		auto collect = [&pieces](auto &&self, const Arena &a, uint32_t at) -> void {
			pieces.push_back({&a, at});
			uint32_t next = at + (uint32_t)a.nodes.size();
			for (const Arena &sub : a.then) {
				self(self, sub, next);
				next += size_of(sub);
			}
		};
		collect(collect, root, 0);
		nodes.resize(size_of(root));
		rtgi_node56 *out = nodes.data();
		auto place = [out](const Piece &p) {
			for (size_t i = 0; i < p.arena->nodes.size(); ++i) {
				rtgi_node56 n = p.arena->nodes[i];
				if (n.link_r >= 0) n.link_l += (int32_t)p.at, n.link_r += (int32_t)p.at;  // inner node: links are node ids
				out[p.at + i] = n;
			}
		};
		std::vector<std::thread> threads;
		threads.reserve(64);
		for (const Piece &p : pieces) {
			bool started = false;
			if (p.arena->nodes.size() >= FORK_MIN && threads.size() < 64) try {
					threads.emplace_back([&place, p] { place(p); });
					started = true;
				} catch (const std::system_error &) {
				}
			if (!started) place(p);
		}
		for (std::thread &t : threads) t.join();
	}
	Bounds range_bounds(uint32_t start, uint32_t end) const {
		Bounds b;
		for (uint32_t i = start; i < end; ++i) b.add(prims[i].box);
		return b;
	}
	// std::sort of the range by the centre along `axis` (bvh2.cpp:214-222). The order std::sort leaves equal centres
	// in is part of the reference's tree (a million random centres hold tens of thousands of equal pairs), so the
	// result has to be that of libstdc++'s std::sort on this range with this comparison. std::sort's moves depend on
	// the outcomes of its comparisons only, so sorting small (centre, position) records instead of the reference's
	// indices leaves everything in the same places, ties included; the primitive records are then put into that
	// order in one pass. A large range near the root, where few threads have work yet, goes through sort_as_std,
	// which forks where std::sort recurses.
	struct Keyed {
		float key;
		uint32_t slot;
	};
	static bool by_key(const Keyed &a, const Keyed &b) { return a.key < b.key; }
	static constexpr uint32_t PARALLEL_SORT_MIN = 1u << 15;

	// working memory of sort_by_centre, one set per thread (a thread's set goes with the thread; run() gives back
	// the calling thread's)
	static std::vector<Keyed> &keyed_buffer() {
		static thread_local std::vector<Keyed> v;
		return v;
	}
	static std::vector<Prim> &moved_buffer() {
		static thread_local std::vector<Prim> v;
		return v;
	}
	void sort_by_centre(uint32_t start, uint32_t end, int axis, int spare_levels = 0) {
		const uint32_t n = end - start;
		if (n < 32) {
			std::sort(prims.data() + start, prims.data() + end, [axis](const Prim &a, const Prim &b) { return a.centre[axis] < b.centre[axis]; });
			return;
		}
		std::vector<Keyed> &keyed = keyed_buffer();
		std::vector<Prim> &moved = moved_buffer();
		keyed.resize(n);
		moved.resize(n);
		for (uint32_t i = 0; i < n; ++i) keyed[i] = Keyed{prims[start + i].centre[axis], i};
		if (spare_levels > 0 && n >= PARALLEL_SORT_MIN) {
			sort_as_std(keyed.data(), keyed.data() + n, spare_levels);
			const Keyed *k = keyed.data();
			Prim *to = moved.data();
			const Prim *from = prims.data() + start;
			on_threads(1 << std::min(spare_levels, 4), n, [=](uint32_t lo, uint32_t hi) {
				for (uint32_t i = lo; i < hi; ++i) to[i] = from[k[i].slot];
			});
		} else {
			std::sort(keyed.begin(), keyed.end(), by_key);
			for (uint32_t i = 0; i < n; ++i) moved[i] = prims[start + keyed[i].slot];
		}
		std::memcpy(prims.data() + start, moved.data(), (size_t)n * sizeof(Prim));
	}

	// fn(lo, hi) over [0, n) cut into `parts` pieces, one thread each (the last piece on this thread; a piece whose
	// thread cannot be started runs here too).
	template <class F>
	static void on_threads(int parts, uint32_t n, F fn) {
		std::vector<std::thread> threads;
		threads.reserve((size_t)std::max(parts, 1));  // (no reallocation, and so no exception, while threads are running)
		for (int i = 0; i < parts; ++i) {
			const uint32_t lo = (uint32_t)((uint64_t)n * (uint32_t)i / (uint32_t)parts), hi = (uint32_t)((uint64_t)n * (uint32_t)(i + 1) / (uint32_t)parts);
			bool started = false;
			if (i + 1 < parts) try {
					threads.emplace_back([=] { fn(lo, hi); });
					started = true;
				} catch (const std::system_error &) {
				}
			if (!started) fn(lo, hi);
		}
		for (std::thread &t : threads) t.join();
	}

	// libstdc++'s std::sort (introsort: median-of-three quicksort down to pieces of 16, heapsort below a depth limit,
	// one insertion sort over everything at the end), restated so that the two sides of a partition can be sorted by
	// different threads. Each step compares and moves what std::sort would, so equal keys end up where std::sort
	// leaves them; tests compare the resulting trees with those built through std::sort itself.
	static constexpr long SORT_PIECE = 16;
	static void insert_unguarded(Keyed *last) {  // a smaller-or-equal element is known to lie to the left
		const Keyed v = *last;
		Keyed *next = last - 1;
		while (by_key(v, *next)) {
			*last = *next;
			last = next;
			--next;
		}
		*last = v;
	}
	static void insertion_sort(Keyed *first, Keyed *last) {
		if (first == last) return;
		for (Keyed *i = first + 1; i != last; ++i) {
			if (by_key(*i, *first)) {
				const Keyed v = *i;
				std::move_backward(first, i, i + 1);
				*first = v;
			} else
				insert_unguarded(i);
		}
	}
	static Keyed *partition_around_median(Keyed *first, Keyed *last) {
		Keyed *a = first + 1, *b = first + (last - first) / 2, *c = last - 1;
		// the median of a, b, c goes to `first`
		if (by_key(*a, *b)) {
			if (by_key(*b, *c)) std::iter_swap(first, b);
			else if (by_key(*a, *c)) std::iter_swap(first, c);
			else std::iter_swap(first, a);
		} else if (by_key(*a, *c)) std::iter_swap(first, a);
		else if (by_key(*b, *c)) std::iter_swap(first, c);
		else std::iter_swap(first, b);
		Keyed *lo = first + 1, *hi = last;
		for (;;) {
			while (by_key(*lo, *first)) ++lo;
			--hi;
			while (by_key(*first, *hi)) --hi;
			if (!(lo < hi)) return lo;
			std::iter_swap(lo, hi);
			++lo;
		}
	}
	// Insertion sort of [first, last) that never looks left of `first`. For a part of the array whose left neighbours
	// are all smaller or equal it moves exactly what the unguarded loop at the end of std::sort moves: that loop
	// stops at the part's first element or before, since it passes strictly greater elements only.
	static void insertion_sort_from(Keyed *first, Keyed *last) {
		for (Keyed *i = first; i != last; ++i) {
			const Keyed v = *i;
			Keyed *j = i;
			while (j != first && by_key(v, *(j - 1))) {
				*j = *(j - 1);
				--j;
			}
			*j = v;
		}
	}
	// Returns where the part still waiting for its insertion pass ends: parts handed to other threads lie to the
	// right of it and are finished, insertion pass included, when this returns.
	static Keyed *quicksort_pieces(Keyed *first, Keyed *last, long depth_limit, int spare_levels) {
		std::vector<std::thread> forked;
		try {
			forked.reserve(8);  // (at most four forks per call; no reallocation while threads are running)
		} catch (const std::bad_alloc &) {
			spare_levels = 0;  // nothing below can throw then: this may be running on a thread of its own
		}
		Keyed *mine_end = last;
		while (last - first > SORT_PIECE) {
			if (depth_limit == 0) {
				std::partial_sort(first, last, last, by_key);  // heapsort of the piece
				break;
			}
			--depth_limit;
			Keyed *cut = partition_around_median(first, last);
			// std::sort recurses into [cut, last) and goes on with [first, cut). Everything in [cut, last) is greater or
			// equal to everything left of it, so another thread can take it to the end (as long as the parts given
			// away form one stretch at the right end).
			bool started = false;
			if (spare_levels > 0 && last == mine_end && last - cut >= (long)(PARALLEL_SORT_MIN / 8)) {
				--spare_levels;
				const int levels = spare_levels;
				try {
					forked.emplace_back([=] {
						Keyed *todo = quicksort_pieces(cut, last, depth_limit, levels);
						insertion_sort_from(cut, todo);
					});
					started = true;
					mine_end = cut;
				} catch (const std::system_error &) {
				}
			}
			if (!started) (void)quicksort_pieces(cut, last, depth_limit, 0);
			last = cut;
		}
		for (std::thread &t : forked) t.join();
		return mine_end;
	}
	static void sort_as_std(Keyed *first, Keyed *last, int spare_levels) {
		if (first == last) return;
		long lg = 0;
		for (unsigned long m = (unsigned long)(last - first); m > 1; m >>= 1) ++lg;
		Keyed *todo = quicksort_pieces(first, last, 2 * lg, std::min(spare_levels, 4));
		// std::sort's last step over what no other thread has finished
		if (todo - first > SORT_PIECE) {
			insertion_sort(first, first + SORT_PIECE);
			for (Keyed *i = first + SORT_PIECE; i != todo; ++i) insert_unguarded(i);
		} else
			insertion_sort(first, todo);
	}

	// Two-pointer partition of bvh2.cpp:265-277 / :338-353. The slot on which the pointers meet is
	// left unexamined on the right-hand side, exactly as there.
	uint32_t partition_at(uint32_t start, uint32_t end, int axis, float plane) {
		uint32_t mid = start, l = start, r = end - 1;
		while (l < r) {
			while (prims[l].centre[axis] <= plane && l < r) {
				++l;
				++mid;
			}
			while (prims[r].centre[axis] > plane && l < r) --r;
			if (prims[l].centre[axis] > prims[r].centre[axis] && l < r) std::swap(prims[l], prims[r]);
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
			sort_by_centre(start, end, axis, fork_levels);
			mid = start + (end - start) / 2;
		} else if (cfg.split == RTGI_SPLIT_SM) {  // bvh2.cpp:237-298
			float plane = all.lo[axis] + (all.hi[axis] - all.lo[axis]) * 0.5f;
			mid = partition_at(start, end, axis, plane);
			if (mid == start || mid == end - 1) {
				sort_by_centre(start, end, axis, fork_levels);
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
					sort_by_centre(start, end, axis, fork_levels);
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
				sort_by_centre(start, end, axis, fork_levels);
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
			l = attach(arena, std::move(sub_l));
			r = attach(arena, std::move(sub_r));
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
	// n_inner counts record slots (the numbering leaves a few empty), not nodes.
	// RTGI_HIP_NODE_ORDER=dfs keeps the reference's pre-order, =bfs numbers without the alignment gaps.
	std::vector<int32_t> inner_id(n, -1);
	uint32_t n_inner = 0;
	out.implicit_limit = 0;
	out.heap_levels = 0;
	const char *order_env = getenv("RTGI_HIP_NODE_ORDER");
	if (order_env && std::string(order_env) == "dfs") {
		for (uint32_t i = 0; i < n; ++i)
			if (t.nodes[i].link_r >= 0) inner_id[i] = (int32_t)n_inner++;
	} else if (!(order_env && std::string(order_env) == "bfs")) {
		std::vector<uint32_t> queue;
		queue.reserve(n);
		size_t head = 0;
		if (t.nodes[t.root].link_r >= 0) {
			// The levels at the top of the tree in which every node is an inner node are numbered like a heap, the root
			// as 1: the children of node i are 2i and 2i + 1, which the kernel computes instead of fetching the links.
			// (Median splits give such levels down to the last one or two; any other tree has a few.) Slot 0 stays empty.
			queue.push_back(t.root);
			inner_id[t.root] = 1;
			n_inner = 2;
			size_t level_size = 1;
			uint32_t levels = 1;  // levels numbered so far
			for (;;) {
				bool all_inner = levels < 24;  // (ids stay far below the 2^25 nodes a buffer descriptor can address)
				for (size_t k = head; k < head + level_size && all_inner; ++k) {
					const rtgi_node56 &nd = t.nodes[queue[k]];
					all_inner = t.nodes[nd.link_l].link_r >= 0 && t.nodes[nd.link_r].link_r >= 0;
				}
				if (!all_inner) break;
				for (size_t k = head; k < head + level_size; ++k) {
					const rtgi_node56 &nd = t.nodes[queue[k]];
					const int32_t id = inner_id[queue[k]];
					inner_id[nd.link_l] = 2 * id, inner_id[nd.link_r] = 2 * id + 1;
					queue.push_back((uint32_t)nd.link_l), queue.push_back((uint32_t)nd.link_r);
				}
				head += level_size;
				level_size *= 2;
				++levels;
				n_inner = 1u << levels;
			}
			// nodes of levels 0 .. levels - 2 have their children among the heap-numbered ones
			out.implicit_limit = levels >= 2 ? 1u << (levels - 1) : 0u;
			out.heap_levels = levels >= 2 ? levels : 0u;
		}
		// the rest level by level; two inner siblings share one aligned 128-byte line
		for (; head < queue.size(); ++head) {
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
	parallel_ranges(n, [&](uint32_t begin, uint32_t end) {  // (leaves cover disjoint slots)
		for (uint32_t i = begin; i < end; ++i) {
			const rtgi_node56 &nd = t.nodes[i];
			if (nd.link_r >= 0) continue;
			uint32_t last = (uint32_t)(-nd.link_l) + (uint32_t)(-nd.link_r) - 1;
			uint32_t one = 1;
			memcpy(out.tris.data() + (size_t)last * 12 + 7, &one, 4);
		}
	});
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
