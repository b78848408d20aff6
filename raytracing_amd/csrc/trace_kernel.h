// trace_kernel.h -- the gfx950 traversal kernel (device code only; included by rtgi_hip.hip).
//
// closest_hit / any_hit of rt/bbvh-base/bvh2.cpp:415-501, ray by ray in the reference's order: nearer child
// first (the right one on equal distances), a child is entered only if its slab test passes and, for
// closest_hit, its entry distance is below the closest t so far; triangles by Cramer's rule in the operation
// order of libgi/intersect.h:43-57; slab test of libgi/intersect.h:204-236.
//
// One ray per lane, wave64, one persistent workgroup per CU. While rays are in supply the kernel is bound by the rate of
// random 64-byte node fetches below L1, at the end of a launch by the latency of its longest rays; the
// structure keeps lanes busy and instructions per useful lane-step few:
//  * a wave refills finished lanes from a global ray counter instead of waiting for its slowest ray;
//  * lanes that reach a leaf wait until a handful have gathered, then the (long) triangle code runs for all
//    of them at once; per ray the order of node visits and triangle tests is unchanged;
//  * node records interleave the two child boxes (l.x r.x l.y r.y ...) so that one packed instruction
//    subtracts / multiplies for both children;
//  * loads go through buffer descriptors: a 32-bit per-lane offset instead of 64-bit address arithmetic,
//    and an out-of-range index reads zeros instead of faulting;
//  * the pending-node stack keeps its top in a register and the rest in LDS as [level][lane] (any mix of
//    depths is conflict-free); trees deeper than MAX_STACK_LDS use the DEEP variant, which continues the stack
//    in a global area sized from the tree depth, so the stack cannot overflow;
//  * the first nodes of the breadth-first numbering, which every ray visits, are read from a copy in LDS (as
//    many as fit beside the stacks);
//  * the levels at the top of the tree that consist of inner nodes only are numbered like a heap, so the children of
//    node i are 2i and 2i + 1 and a visit fetches the twelve box planes (three 16-byte loads) but no links; median
//    splits give such levels down to the last one or two;
//  * large batches are traced through a permutation (octant, length class, Morton code of the origin), each XCD
//    starting on the rays of one direction octant, the long rays first.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <type_traits>

namespace rtgi_dev {

// One workgroup of 16 waves per CU (four per SIMD), so that the CU's 160 KiB of LDS hold ONE copy of the top of the
// tree beside the stacks: the launch divides the LDS between `stack_levels` pending-node slots per lane and as many
// of the first nodes of the breadth-first numbering as fit in the rest (2^20 triangles, depth 20: 80 KiB of stacks,
// 1280 nodes = the upper ten levels; a small scene's whole tree).
#ifndef RTGI_BLOCK
#define RTGI_BLOCK 1024
#endif
constexpr int BLOCK = RTGI_BLOCK;  // threads per workgroup
#ifndef RTGI_WAVES_PER_SIMD
#define RTGI_WAVES_PER_SIMD (RTGI_BLOCK / 256)
#endif
constexpr int WAVES_PER_SIMD = RTGI_WAVES_PER_SIMD;  // resident waves per SIMD the register budget is kept within (all workgroups of a CU together)
constexpr int MAX_STACK_LDS = 24;  // at most this many pending nodes per lane in LDS (beyond the one in a register)
constexpr int DEEP_STACK_LDS = 20; // trees deeper than MAX_STACK_LDS: this many in LDS, the rest in the global spill area
constexpr int COUNTER_STRIDE = 32;
constexpr int MAX_BATCHES = 32;
constexpr int MAX_CHUNKS = 6;
constexpr int32_t IDLE = INT32_MIN;  // lane has no ray; leaf references are ~slot with slot < 2^30

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct TraceArgs {
	const void *nodes;      // 64 B per inner node, see flatten_tree() in bvh_host.cpp
	const void *tris;       // 48 B per leaf-order slot
	const float4 *rays;     // 32 B per ray: o.xyz d.x | d.yz t_min t_max
	const uint32_t *order;  // optional: the i-th ray to trace is rays[order[i]] (spatially sorted batches)
	const uint32_t *seg_bounds;  // optional: n_segments + 1 positions cutting the (sorted) batch; else equal parts
	float4 *hits;           // closest_hit output
	uint8_t *occluded;      // any_hit output
	uint32_t *counts;       // optional: 2 per ray
	unsigned long long *totals;  // optional, counting kernels only: six sums over the launch, see the end of the kernel
	int32_t *spill;         // DEEP only: [level - stack_levels][global lane]
	uint32_t *ray_counter;  // n_segments counters, COUNTER_STRIDE words apart (one cache line each)
	uint32_t n_rays;
	uint32_t n_segments;    // the ray range is cut into this many contiguous segments (1 or 8)
	uint32_t node_bytes, tri_bytes;
	uint32_t spill_stride;  // = gridDim.x * BLOCK
	int32_t root_ref;
	int32_t refill_min;     // refill a wave once this many lanes are idle
	int32_t leaf_min;       // leave the inner-node loop once this many lanes have left it
	int32_t drain_leaf_min; // the same once the ray supply is used up
	// MODE 2 (several batches in one launch): ray i of the launch is ray i - batch_begin[b] of batch b, the last b
	// with batch_begin[b] <= i; rays / hits / occluded above are unused
	uint32_t n_batches;
	uint32_t batch_begin[MAX_BATCHES];
	const float4 *batch_rays[MAX_BATCHES];
	void *batch_out[MAX_BATCHES];       // float4 per ray (closest_hit) or one byte per ray (any_hit)
	uint32_t batch_any;                 // bit b: batch b is an any_hit batch
	// Rays that are still on their way when the launch begins (host batches: the upload of the later parts runs beside
	// the traversal of the earlier ones). n_chunks > 1: rays [chunk_end[c - 1], chunk_end[c]) form chunk c, with a
	// permutation and octant bounds of its own (or none); chunk c may be touched once *chunks_ready > c. A wave that has
	// waited wait_ticks (100 MHz) for a chunk sets *gave_up and ends; the caller then traces what is missing another way.
	uint32_t n_chunks;
	uint32_t chunk_end[MAX_CHUNKS];
	const uint32_t *chunk_order[MAX_CHUNKS];   // indexed by position in the whole batch, like `order`
	const uint32_t *chunk_bounds[MAX_CHUNKS];  // n_segments + 1 positions relative to the chunk's first
	const uint32_t *chunks_ready;
	uint32_t *gave_up;
	uint32_t wait_ticks;
	int32_t top_nodes;      // visits to inner nodes 0 .. top_nodes-1 are served from a copy in LDS
	int32_t stack_levels;   // pending-node slots per lane in LDS; dynamic LDS: see lds_bytes_for()
	int32_t implicit_limit; // inner nodes below this index are numbered like a heap: children 2i and 2i + 1, no link fetch
	int32_t boxes_ordered;  // every child box of the tree has min <= max on every axis (checked at upload): waves whose rays
	                        // share a direction octant take the near and far planes by position instead of by min / max
#ifdef RTGI_DIAG
	unsigned long long *diag;  // diagnostic build only: 8 counters summed over all waves
#endif
};

// Single-instruction min / max for values that cannot be NaN. (fminf/fmaxf would first canonicalise every
// operand with an extra v_max_f32 x, x because the compiler cannot see that.)
__device__ __forceinline__ float vmin(float a, float b) {
	float r;
	asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
	float r;
	asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
	float r;
	asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
	float r;
	asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}

__device__ __forceinline__ float next_up(float x) {
	// smallest float greater than x, for finite or -inf x; +inf stays
	uint32_t b = __float_as_uint(x);
	if (x == 0.0f) return __uint_as_float(1u);
	if (x != x || b == 0x7f800000u) return x;
	return __uint_as_float((b & 0x80000000u) ? b - 1u : b + 1u);
}

// Dynamic LDS of a workgroup: the stacks, then a copy of the first top_nodes node records without their links
// (48 bytes each, node after node: lanes on different nodes then start in 16 different bank groups, as conflict-free
// as four separate arrays, and one address serves the three reads), then the links (8 bytes) of those copied nodes
// that are not heap-numbered.
__host__ __device__ inline size_t lds_bytes_for(uint32_t stack_levels, uint32_t top_nodes, uint32_t implicit_limit) {
	return (size_t)stack_levels * BLOCK * sizeof(int32_t) + (size_t)top_nodes * 48 + (top_nodes > implicit_limit ? (size_t)(top_nodes - implicit_limit) * 8 : 0);
}

// MODE 0: closest_hit, 1: any_hit, 2: a launch over several batches, each of one kind or the other.
template <int MODE, bool COUNT, bool DEEP>
__global__ __launch_bounds__(BLOCK, WAVES_PER_SIMD) void trace_kernel(TraceArgs a) {
	constexpr bool MIXED = MODE == 2;
	extern __shared__ __align__(16) unsigned char s_mem[];
	const int32_t stack_levels = a.stack_levels;
	int32_t *const s_stack = reinterpret_cast<int32_t *>(s_mem);  // [level][thread]: bank = lane, conflict-free for any mix of depths
	u32x4 *const s_top = reinterpret_cast<u32x4 *>(s_mem + (size_t)stack_levels * BLOCK * sizeof(int32_t));  // [node][3]
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const uint32_t gl = blockIdx.x * BLOCK + tid;

	const uint32_t n_rays = a.n_rays;
	const __amdgpu_buffer_rsrc_t node_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.nodes), 0, (int)a.node_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t tri_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.tris), 0, (int)a.tri_bytes, 0x00020000);

	// Every ray walks the top of the tree; a copy in LDS takes those visits off the vector cache, whose access
	// rate is what the kernel runs against. (Reads beyond the node array return zeros.)
	const int32_t top_nodes = a.top_nodes;
	const int32_t implicit_limit = a.implicit_limit;
	u32x2 *const s_links = reinterpret_cast<u32x2 *>(s_top + 3 * top_nodes);  // [node - implicit_limit], nodes of the copy only
	const uint32_t top_offset = (uint32_t)stack_levels * BLOCK * sizeof(int32_t);
	if (top_nodes) {
		for (int i = tid; i < 3 * top_nodes; i += BLOCK) {
			const int node = i / 3, quad = i - 3 * node;
			s_top[i] = __builtin_amdgcn_raw_buffer_load_b128(node_rsrc, (node << 6) + (quad << 4), 0, 0);
		}
		for (int i = implicit_limit + tid; i < top_nodes; i += BLOCK)
			s_links[i - implicit_limit] = __builtin_amdgcn_raw_buffer_load_b64(node_rsrc, (i << 6) + 48, 0, 0);
		__syncthreads();
	}

	float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, t_min = 0, t_max = 0;
	float lim = 0;       // plain rays: entry distances below this pass "t1 <= t_max (&& t1 < closest t)"
	float lim0 = 0;      // next_up(t_max)
	float best_t = FLT_MAX, best_b = 0.0f, best_g = 0.0f;
	uint32_t best_ref = 0;
	uint32_t ray_id = 0;
	bool any_lane = MODE == 1;  // this lane's ray only asks whether anything is hit
	const float4 *my_rays = a.rays;
	void *my_out = nullptr;     // MIXED: where this lane's result goes
	uint32_t n_inner = 0, n_tri = 0, n_top = 0;
	unsigned long long sum_rays = 0, sum_inner = 0, sum_tri = 0, sum_top = 0, sum_spill = 0, sum_links = 0;  // COUNT: over all rays of this lane
	int32_t cur = IDLE;
	int sp = 0;          // pending entries in memory (the next node to visit is `cur`, in a register)
	bool plain = true;   // no NaN can arise in this ray's slab tests
	bool more = true;    // wave-uniform: some segment may still have rays
	// Each XCD has its own L2. Waves of one XCD draw rays from the same contiguous part of the batch, so that
	// an L2 sees neighbouring rays (callers hand rays over in image or sorted order); a wave whose segment is
	// used up helps with the next one. Placement affects speed only.
	const uint32_t home_seg = (__builtin_amdgcn_s_getreg(6164) & 15u) % a.n_segments;  // HW_REG_XCC_ID, bits 3:0
	uint32_t seg = home_seg;
	uint32_t segs_done = 0;
	uint32_t chunk = 0;              // wave-uniform: the chunk this wave draws from (0 if the batch is all there)
	unsigned long long wait_since = 0;  // wave-uniform: when this wave first found the next chunk missing (0: not waiting)

	auto push = [&](int32_t v) {
		if (!DEEP || sp < stack_levels)
			s_stack[sp * BLOCK + tid] = v;
		else {
			a.spill[(size_t)(sp - stack_levels) * a.spill_stride + gl] = v;
			if (COUNT) ++sum_spill;
		}
		++sp;
	};
	auto pop = [&]() -> int32_t {
		--sp;
		if (!DEEP || sp < stack_levels) return s_stack[sp * BLOCK + tid];
		return a.spill[(size_t)(sp - stack_levels) * a.spill_stride + gl];
	};
	auto finish = [&]() {
		if (MIXED) {
			if (any_lane)
				static_cast<uint8_t *>(my_out)[ray_id] = best_ref ? 1 : 0;
			else
				static_cast<float4 *>(my_out)[ray_id] = make_float4(best_t, best_b, best_g, __uint_as_float(best_ref));
		} else if (MODE == 1)
			a.occluded[ray_id] = best_ref ? 1 : 0;
		else
			a.hits[ray_id] = make_float4(best_t, best_b, best_g, __uint_as_float(best_ref));
		if (COUNT) {
			a.counts[2 * (size_t)ray_id] = n_inner;
			a.counts[2 * (size_t)ray_id + 1] = n_tri;
			sum_rays += 1, sum_inner += n_inner, sum_tri += n_tri, sum_top += n_top;
		}
		cur = IDLE;
	};
	auto advance = [&]() {  // next pending node, or the ray is done
		if (sp == 0)
			finish();
		else
			cur = pop();
	};

#ifdef RTGI_DIAG
	const unsigned long long dg_t0 = __builtin_amdgcn_s_memrealtime();
	unsigned long long dg_t_dry = 0;
	unsigned long long dg_it_more = 0, dg_ln_more = 0, dg_it_drain = 0, dg_ln_drain = 0, dg_it_leaf = 0, dg_ln_leaf = 0, dg_refills = 0, dg_outer = 0;
#endif
	for (;;) {
#ifdef RTGI_DIAG
		++dg_outer;
#endif
		// ---- hand new rays to idle lanes --------------------------------------------------------------
		const unsigned long long idle = __ballot(cur == IDLE);
		if (more && __popcll(idle) >= a.refill_min) {
#ifdef RTGI_DIAG
			++dg_refills;
#endif
			const uint32_t want = (uint32_t)__popcll(idle);
			const int leader = (int)__ffsll((long long)idle) - 1;
			uint32_t first = 0, avail = 0;
			const bool chunked = a.n_chunks > 1;
			const uint32_t *order = a.order;
			for (;;) {  // draw from this wave's segment; when it is used up, move on to the next one
				uint32_t part_lo = 0, part_len = n_rays;  // the chunk's range of positions
				const uint32_t *bounds = a.seg_bounds;
				if (chunked) {
					if (wait_since) {  // the next chunk was not there last time: is it now?
						// (asked without acquire semantics: an acquire empties this XCD's L2 of what the other waves, and
						// the sort that everybody is waiting for, are working with; the fence follows once, when the answer is yes)
						if (__hip_atomic_load(a.chunks_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > chunk) {
							__atomic_thread_fence(__ATOMIC_ACQUIRE);
							wait_since = 0;
						} else {
							if (__builtin_amdgcn_s_memrealtime() - wait_since > a.wait_ticks) {
								if (lane == 0) __hip_atomic_store(a.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
								more = false;
							}
							break;
						}
					}
					part_lo = chunk ? a.chunk_end[chunk - 1] : 0u;
					part_len = a.chunk_end[chunk] - part_lo;
					bounds = a.chunk_bounds[chunk];
					order = a.chunk_order[chunk];
				}
				// (bounds of a chunk were written while this kernel ran: they are read past the caches the kernel keeps)
				const uint32_t seg_lo = !bounds ? (uint32_t)((uint64_t)part_len * seg / a.n_segments)
				                        : chunked ? __hip_atomic_load(bounds + seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : bounds[seg];
				const uint32_t seg_hi = !bounds ? (uint32_t)((uint64_t)part_len * (seg + 1) / a.n_segments)
				                        : chunked ? __hip_atomic_load(bounds + seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : bounds[seg + 1];
				const uint32_t seg_len = seg_hi - seg_lo;
				uint32_t base = 0;
				if (lane == leader) base = atomicAdd(a.ray_counter + (chunk * a.n_segments + seg) * COUNTER_STRIDE, want);
				base = __shfl(base, leader);
				if (base < seg_len) {
					first = part_lo + seg_lo + base;
					avail = min(want, seg_len - base);
					break;
				}
				seg = seg + 1 == a.n_segments ? 0 : seg + 1;
				if (++segs_done < a.n_segments) continue;
				if (chunked && chunk + 1 < a.n_chunks) {  // on to the next chunk, once it is there
					++chunk;
					segs_done = 0;
					seg = home_seg;
					if (__hip_atomic_load(a.chunks_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > chunk) {
						__atomic_thread_fence(__ATOMIC_ACQUIRE);
						continue;
					}
					wait_since = __builtin_amdgcn_s_memrealtime() | 1ull;
					break;
				}
				more = false;
#ifdef RTGI_DIAG
				dg_t_dry = __builtin_amdgcn_s_memrealtime();
#endif
				break;
			}
			const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
			// (an index from the order array is trusted only as far as the batch goes)
			const uint32_t mine = (rank < avail) ? (order ? order[first + rank] : first + rank) : 0xffffffffu;
			if (cur == IDLE && mine < n_rays) {
				uint32_t local = mine;
				if (MIXED) {  // which batch, and which ray of it
					for (uint32_t b = 0; b < a.n_batches; ++b)
						if (mine >= a.batch_begin[b]) {
							local = mine - a.batch_begin[b];
							my_rays = a.batch_rays[b];
							my_out = a.batch_out[b];
							any_lane = (a.batch_any >> b) & 1u;
						}
				}
				const float4 q0 = my_rays[2 * (size_t)local], q1 = my_rays[2 * (size_t)local + 1];
				ox = q0.x, oy = q0.y, oz = q0.z;
				dx = q0.w, dy = q1.x, dz = q1.y;
				t_min = q1.z, t_max = q1.w;
				ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;  // ray::ray, libgi/rt.h:24
				auto usable = [](float v) { return __builtin_isfinite(v) && v != 0.0f; };
				plain = usable(ix) && usable(iy) && usable(iz) && __builtin_isfinite(ox) && __builtin_isfinite(oy) &&
				        __builtin_isfinite(oz) && t_min == t_min && t_max == t_max && t_max != __builtin_inff();
				lim0 = next_up(t_max);
				ray_id = local;
				best_t = FLT_MAX, best_b = 0.0f, best_g = 0.0f, best_ref = 0;
				n_inner = n_tri = n_top = 0;
				cur = a.root_ref;
				sp = 0;
				lim = any_lane ? lim0 : __builtin_fminf(best_t, lim0);
			}
		}
		const unsigned long long busy = __ballot(cur != IDLE);
		if (busy == 0) {
			if (!more) break;
			if (wait_since)  // nothing to do until the next chunk arrives: a few microseconds out of everybody's way
				for (int nap = 0; nap < 8; ++nap) __builtin_amdgcn_s_sleep(127);
			continue;
		}
		const bool wave_plain = __ballot(cur != IDLE && !plain) == 0;
		// While rays are in supply, lanes that are through with a phase wait for company (each phase is a long piece of
		// code, run for however many lanes need it); once the supply is dry the launch ends with its slowest ray, so
		// they wait for fewer. A phase runs once at least if it has a lane, and goes on until `leave` of the lanes that
		// were busy when it began stand outside it.
		const int n_busy = (int)__popcll(busy);
		const int leave = __builtin_amdgcn_readfirstlane(more ? a.leaf_min : a.drain_leaf_min);

		// one step of one lane: the two child boxes of its node against its ray; which ones to enter, and which one first
		auto boxes = [&](auto plain_tag, auto octant_tag, const u32x4 q0, const u32x4 q1, const u32x4 q2, bool &hl, bool &hr, bool &left_first) {
			constexpr bool PLAIN = decltype(plain_tag)::value;
			constexpr int OCT = decltype(octant_tag)::value;  // 0..7: signs of the direction of every ray of the wave; -1: mixed
			if (COUNT) ++n_inner;
			auto pair = [](uint32_t u, uint32_t v) { return f32x2{__uint_as_float(u), __uint_as_float(v)}; };
			// .x = left child, .y = right child; (box - o) * id as in libgi/intersect.h:210-221
			const f32x2 ax = (pair(q0.x, q0.y) - f32x2{ox, ox}) * f32x2{ix, ix};
			const f32x2 ay = (pair(q0.z, q0.w) - f32x2{oy, oy}) * f32x2{iy, iy};
			const f32x2 az = (pair(q1.x, q1.y) - f32x2{oz, oz}) * f32x2{iz, iz};
			const f32x2 bx = (pair(q1.z, q1.w) - f32x2{ox, ox}) * f32x2{ix, ix};
			const f32x2 by = (pair(q2.x, q2.y) - f32x2{oy, oy}) * f32x2{iy, iy};
			const f32x2 bz = (pair(q2.z, q2.w) - f32x2{oz, oz}) * f32x2{iz, iz};
			float dl, dr;
			if (PLAIN && OCT >= 0) {
				// All rays of the wave look the same way along every axis. For a box with min <= max and a positive inverse
				// direction, (min - o) * id <= (max - o) * id holds after rounding too (subtraction and multiplication by a
				// positive number are monotonic), so the smaller product of intersect.h:212-217 is the min plane's, and the max
				// plane's for a negative inverse direction: the twelve min / max of the general case are not needed. (Where the
				// two products are equal they may differ in the sign of a zero, which no comparison below sees.)
				const f32x2 nx = (OCT & 1) ? bx : ax, fx = (OCT & 1) ? ax : bx;
				const f32x2 ny = (OCT & 2) ? by : ay, fy = (OCT & 2) ? ay : by;
				const f32x2 nz = (OCT & 4) ? bz : az, fz = (OCT & 4) ? az : bz;
				dl = vmax3(nx.x, ny.x, nz.x);
				dr = vmax3(nx.y, ny.y, nz.y);
				const float fl = vmin3(fx.x, fy.x, fz.x);
				const float fr = vmin3(fx.y, fy.y, fz.y);
				hl = (vmax(dl, t_min) <= fl) && (dl < lim);
				hr = (vmax(dr, t_min) <= fr) && (dr < lim);
			} else if (PLAIN) {
				// no NaN possible: the selects of intersect.h:212-228 are exact min / max, and
				// !(t1 > t2) && !(t2 < t_min) && !(t1 > t_max) [&& t1 < closest t] folds into two compares
				dl = vmax3(vmin(ax.x, bx.x), vmin(ay.x, by.x), vmin(az.x, bz.x));
				dr = vmax3(vmin(ax.y, bx.y), vmin(ay.y, by.y), vmin(az.y, bz.y));
				const float fl = vmin3(vmax(ax.x, bx.x), vmax(ay.x, by.x), vmax(az.x, bz.x));
				const float fr = vmin3(vmax(ax.y, bx.y), vmax(ay.y, by.y), vmax(az.y, bz.y));
				hl = (vmax(dl, t_min) <= fl) && (dl < lim);
				hr = (vmax(dr, t_min) <= fr) && (dr < lim);
			} else {
				auto slab = [&](float a_x, float b_x, float a_y, float b_y, float a_z, float b_z, float &entry) {
					float nx = (a_x < b_x) ? a_x : b_x, fx = (b_x < a_x) ? a_x : b_x;
					float ny = (a_y < b_y) ? a_y : b_y, fy = (b_y < a_y) ? a_y : b_y;
					float nz = (a_z < b_z) ? a_z : b_z, fz = (b_z < a_z) ? a_z : b_z;
					float t1 = (nx < ny) ? ny : nx;
					t1 = (nz < t1) ? t1 : nz;
					float t2 = (fx < fy) ? fx : fy;
					t2 = (fz < t2) ? fz : t2;
					entry = t1;
					bool hit = !(t1 > t2) && !(t2 < t_min) && !(t1 > t_max);
					if (!any_lane) hit = hit && t1 < best_t;  // bvh2.cpp:432-433; not in any_hit (:476-477)
					return hit;
				};
				hl = slab(ax.x, bx.x, ay.x, by.x, az.x, bz.x, dl);
				hr = slab(ax.y, bx.y, ay.y, by.y, az.y, bz.y, dr);
			}
			// bvh2.cpp:434-446: both hit -> nearer first, the right child on a tie, the other one pending
			// (& and | on purpose: lane masks combined by scalar instructions instead of branches)
			left_first = hl & ((dl < dr) | !hr);
		};
		// the twelve box planes of inner node `cur` (and its links, where they are not implied), from the copy in LDS or
		// from memory. (All loads from memory go out before the first LDS read, which has to wait for them: the compiler
		// cannot know that the two write different lanes of the same registers.)
		auto fetch = [&](bool with_links, u32x4 &q0, u32x4 &q1, u32x4 &q2, u32x2 &links) {
			if (cur >= top_nodes) {
				const int voff = cur << 6;
				q0 = __builtin_amdgcn_raw_buffer_load_b128(node_rsrc, voff, 0, 0);
				q1 = __builtin_amdgcn_raw_buffer_load_b128(node_rsrc, voff + 16, 0, 0);
				q2 = __builtin_amdgcn_raw_buffer_load_b128(node_rsrc, voff + 32, 0, 0);
				if (with_links) {
					links = __builtin_amdgcn_raw_buffer_load_b64(node_rsrc, voff + 48, 0, 0);
					if (COUNT) ++sum_links;
				}
			} else {
				const u32x4 *rec = reinterpret_cast<const u32x4 *>(s_mem + top_offset + __umul24((uint32_t)cur, 48u));
				q0 = rec[0], q1 = rec[1], q2 = rec[2];
				if (with_links) links = s_links[cur - implicit_limit];
				if (COUNT) ++n_top;
			}
		};

		// ---- inner nodes: every lane standing on an inner node takes one step per iteration; the loop ends once `leave`
		// lanes have left it (reached a leaf or finished their ray), or nobody is left -----------------------------------
		auto inner_phase = [&](auto plain_tag, auto octant_tag) {
			const int keep_going = max(0, n_busy - leave);
			// (the count of lanes on inner nodes is wave-uniform and carried from the end of one iteration to the next)
			for (unsigned long long on_inner = __ballot(cur >= 0); (int)__popcll(on_inner) > keep_going; on_inner = __ballot(cur >= 0)) {
				const bool at_inner = cur >= 0;
#ifdef RTGI_DIAG
				if (more) ++dg_it_more, dg_ln_more += __popcll(__ballot(at_inner));
				else ++dg_it_drain, dg_ln_drain += __popcll(__ballot(at_inner));
#endif
				if (at_inner) {
					u32x4 q0, q1, q2;
					u32x2 links = u32x2{(uint32_t)cur << 1, ((uint32_t)cur << 1) | 1u};  // heap-numbered nodes: no link fetch
					fetch(cur >= implicit_limit, q0, q1, q2, links);
					bool hl, hr, left_first;
					boxes(plain_tag, octant_tag, q0, q1, q2, hl, hr, left_first);
					const int32_t first = (int32_t)(left_first ? links.x : links.y);
					const int32_t second = (int32_t)(left_first ? links.y : links.x);
					if (hl && hr) push(second);
					if (hl || hr)
						cur = first;
					else
						advance();
				}
			}
		};
		// Sorted batches hand a wave rays of one direction octant (and so do neighbouring camera rays): is that so now?
		int octant = -1;
		if (wave_plain && a.boxes_ordered) {
			const unsigned long long neg_x = __ballot(ix < 0.0f) & busy, neg_y = __ballot(iy < 0.0f) & busy, neg_z = __ballot(iz < 0.0f) & busy;
			if ((neg_x == 0 || neg_x == busy) && (neg_y == 0 || neg_y == busy) && (neg_z == 0 || neg_z == busy))
				octant = (neg_x ? 1 : 0) | (neg_y ? 2 : 0) | (neg_z ? 4 : 0);
		}
		if (!wave_plain)
			inner_phase(std::false_type{}, std::integral_constant<int, -1>{});
		else
			switch (octant) {
			case 0: inner_phase(std::true_type{}, std::integral_constant<int, 0>{}); break;
			case 1: inner_phase(std::true_type{}, std::integral_constant<int, 1>{}); break;
			case 2: inner_phase(std::true_type{}, std::integral_constant<int, 2>{}); break;
			case 3: inner_phase(std::true_type{}, std::integral_constant<int, 3>{}); break;
			case 4: inner_phase(std::true_type{}, std::integral_constant<int, 4>{}); break;
			case 5: inner_phase(std::true_type{}, std::integral_constant<int, 5>{}); break;
			case 6: inner_phase(std::true_type{}, std::integral_constant<int, 6>{}); break;
			case 7: inner_phase(std::true_type{}, std::integral_constant<int, 7>{}); break;
			default: inner_phase(std::true_type{}, std::integral_constant<int, -1>{}); break;
			}

		// ---- leaves: all triangles of the leaf, in slot order ---------------------------------------------
		if (cur < 0 && cur != IDLE) {
			int voff = (int)(~cur) * 48;
			bool done = false;
			for (;;) {
#ifdef RTGI_DIAG
				++dg_it_leaf, dg_ln_leaf += __popcll(__ballot(true));
#endif
				const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(tri_rsrc, voff, 0, 0);
				const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(tri_rsrc, voff + 16, 0, 0);
				const u32x4 t2 = __builtin_amdgcn_raw_buffer_load_b128(tri_rsrc, voff + 32, 0, 0);
				if (COUNT) ++n_tri;
				// libgi/intersect.h:26-57, letters as there; a..f were subtracted on the host (same rounding)
				const float ea = __uint_as_float(t1.x), eb = __uint_as_float(t1.y), ec = __uint_as_float(t1.z);
				const float ed = __uint_as_float(t2.x), ee = __uint_as_float(t2.y), ef = __uint_as_float(t2.z);
				const float g = dx, h = dy, i = dz;
				const float j = __uint_as_float(t0.x) - ox, k = __uint_as_float(t0.y) - oy, l = __uint_as_float(t0.z) - oz;
				float c1 = ee * i - h * ef;
				float c2 = g * ef - ed * i;
				float c3 = ed * h - ee * g;
				const float M = ea * c1 + eb * c2 + ec * c3;
				float beta = j * c1 + k * c2 + l * c3;
				c1 = ea * k - j * eb;
				c2 = j * ec - ea * l;
				c3 = eb * l - k * ec;
				float gamma = i * c1 + h * c2 + g * c3;
				float tt = -(ef * c1 + ee * c2 + ed * c3);
				beta /= M;
				gamma /= M;
				tt /= M;
				if (tt > t_min && tt < t_max && beta > 0 && gamma > 0 && beta + gamma <= 1) {
					if (any_lane) {
						best_ref = 1;
						done = true;
					} else if (tt < best_t) {
						best_t = tt, best_b = beta, best_g = gamma;
						best_ref = t0.w;
						lim = __builtin_fminf(best_t, lim0);
					}
				}
				if (done || t1.w) break;
				voff += 48;
			}
			if (done)
				finish();
			else
				advance();
		}
	}
	if (COUNT && a.totals) {  // rays, inner nodes popped, triangles tested, inner nodes read from the copy in LDS, pushes beyond the
		// LDS stack, links fetched from memory
		unsigned long long v[6] = {sum_rays, sum_inner, sum_tri, sum_top, sum_spill, sum_links};
		for (int k = 0; k < 6; ++k) {
			for (int step = 32; step > 0; step >>= 1) v[k] += __shfl_xor(v[k], step);
			if (lane == 0) atomicAdd(a.totals + k, v[k]);
		}
	}
#ifdef RTGI_DIAG
	if (lane == 0 && a.diag) {
		atomicAdd(a.diag + 0, dg_it_more), atomicAdd(a.diag + 1, dg_ln_more), atomicAdd(a.diag + 2, dg_it_drain);
		atomicAdd(a.diag + 3, dg_ln_drain), atomicAdd(a.diag + 4, dg_it_leaf), atomicAdd(a.diag + 5, dg_ln_leaf);
		atomicAdd(a.diag + 6, dg_refills), atomicAdd(a.diag + 7, dg_outer);
		const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
		atomicMin(a.diag + 8, dg_t0), atomicMax(a.diag + 9, t_end);
		if (dg_t_dry) atomicMin(a.diag + 10, dg_t_dry), atomicMax(a.diag + 11, dg_t_dry);
		atomicAdd(a.diag + 12, t_end - dg_t0);
	}
#endif

}

// ---- spatial ordering of a ray batch ---------------------------------------------------------------------
// Rays that start close together and head the same way walk the same nodes, which the caches reward; the
// order of tracing does not change any ray's result. A batch can therefore be traced through a permutation
// sorted by a 24-bit key: direction octant (3 bits) | length class (5 bits, longest first) | the upper 16 bits of
// the 18-bit Morton code of the origin inside the scene's bounds.
constexpr int KEY_BITS = 24;
#ifndef RTGI_CLASS_BITS
#define RTGI_CLASS_BITS 5
#endif
constexpr int CLASS_BITS = RTGI_CLASS_BITS;  // 3..8 measured: 32 classes are the optimum (-1.3 % against 8)
struct KeyArgs {
	const float4 *rays;
	uint32_t *keys;
	uint32_t *index;
	float lo[3], hi[3], inv_extent[3];
	float inv_diagonal;
	uint32_t n;
	uint32_t first;  // position of this batch's first ray in the launch: keys and indices go to [first, first + n)
	uint32_t index_base;  // added to the indices written (a chunk of a batch, sorted on its own: its first ray's number)
};
__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 6 bits -> every third bit
	v = (v | (v << 8)) & 0x0000F00Fu;
	v = (v | (v << 4)) & 0x000C30C3u;
	v = (v | (v << 2)) & 0x00249249u;
	return v;
}
// How far a ray can travel inside the scene's bounds, clipped to its own [t_min, t_max]: an upper bound on
// its traversal work that costs one slab test. <= 0 or NaN for rays that miss the scene.
__device__ __forceinline__ float ray_span(const KeyArgs &a, const float4 q0, const float4 q1) {
	const float dir[3] = {q0.w, q1.x, q1.y}, org[3] = {q0.x, q0.y, q0.z};
	float t_in = q1.z, t_out = q1.w;
	for (int k = 0; k < 3; ++k) {
		const float inv = 1.0f / dir[k];
		const float ta = (a.lo[k] - org[k]) * inv, tb = (a.hi[k] - org[k]) * inv;
		t_in = __builtin_fmaxf(t_in, __builtin_fminf(ta, tb));
		t_out = __builtin_fminf(t_out, __builtin_fmaxf(ta, tb));
	}
	return t_out - t_in;
}

__device__ __forceinline__ uint32_t ray_key(const KeyArgs &a, const float4 q0, const float4 q1) {
	auto cell = [](float x, float lo, float inv) {
		const float f = (x - lo) * inv * 64.0f;
		return (uint32_t)__builtin_fminf(__builtin_fmaxf(f == f ? f : 0.0f, 0.0f), 63.0f);
	};
	const uint32_t morton = spread3(cell(q0.x, a.lo[0], a.inv_extent[0])) | (spread3(cell(q0.y, a.lo[1], a.inv_extent[1])) << 1) |
	                        (spread3(cell(q0.z, a.lo[2], a.inv_extent[2])) << 2);
	const uint32_t octant = (q0.w < 0.0f ? 1u : 0u) | (q1.x < 0.0f ? 2u : 0u) | (q1.y < 0.0f ? 4u : 0u);
	// Length class, longest first, in equal steps of the scene's diagonal. A launch ends when its last ray does,
	// so rays that can be long are started first and the end of the launch is left to rays that must be short.
	constexpr float top = (float)((1u << CLASS_BITS) - 1u);
	const float rel = ray_span(a, q0, q1) * a.inv_diagonal * (top + 1.0f);
	const uint32_t cls = (uint32_t)top - (uint32_t)__builtin_fminf(__builtin_fmaxf(rel == rel ? rel : 0.0f, 0.0f), top);
	return (octant << 21) | (cls << (21 - CLASS_BITS)) | (morton >> (CLASS_BITS - 3));
}

__global__ __launch_bounds__(256) void ray_keys_kernel(KeyArgs a) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= a.n) return;
	a.keys[a.first + i] = ray_key(a, a.rays[2 * (size_t)i], a.rays[2 * (size_t)i + 1]);
	a.index[a.first + i] = a.index_base + a.first + i;
}

// ---- ordering a chunk of a host batch beside a running traversal -------------------------------------------------
// A chunk that arrives while the launch runs is wanted soon, and every kernel started beside the launch waits tens of
// microseconds for its turn whatever its size; the radix sort's chain of eight is most of a chunk's delay. Such a chunk is
// ordered by the upper eight bits of the key only - direction octant and length class, 256 bins, in which rays keep the
// caller's order roughly (tile by tile) - in two kernels: bins and their counts; places by one LDS counter per bin and
// tile plus one global counter per bin. Inside windows of the size of a chunk this order traces 3 % slower than the full
// key's; the last workgroup to finish announces the chunk.
constexpr int COARSE_BINS = 256;
constexpr int COARSE_BLOCK = 256;  // (small workgroups: they have to fit into what the traversal launch leaves of every CU)
constexpr int COARSE_TILE = 16 * COARSE_BLOCK;  // rays per workgroup
__global__ __launch_bounds__(COARSE_BLOCK) void ray_bins_kernel(KeyArgs a, uint8_t *bins, uint32_t *hist) {
	__shared__ uint32_t s_hist[COARSE_BINS];
	s_hist[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t tile = blockIdx.x * COARSE_TILE;
	for (uint32_t j = 0; j < COARSE_TILE; j += COARSE_BLOCK) {
		const uint32_t i = tile + j + threadIdx.x;
		if (i < a.n) {
			const uint32_t bin = ray_key(a, a.rays[2 * (size_t)i], a.rays[2 * (size_t)i + 1]) >> (KEY_BITS - 8);
			bins[i] = (uint8_t)bin;
			atomicAdd(&s_hist[bin], 1u);
		}
	}
	__syncthreads();
	if (s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}

// order[0, n): the rays of the chunk (numbered from index_base) bin after bin; bounds[0..8]: where the octants begin.
// cursor[256] and *done start at zero; the last workgroup through stores `ready_value` to *ready (if ready is given).
__global__ __launch_bounds__(COARSE_BLOCK) void coarse_scatter_kernel(const uint8_t *bins, uint32_t n, uint32_t index_base, const uint32_t *hist,
                                                                      uint32_t *cursor, uint32_t *order, uint32_t *bounds, uint32_t *done,
                                                                      uint32_t *ready, uint32_t ready_value) {
	static_assert(COARSE_BLOCK == COARSE_BINS, "one thread per bin");
	__shared__ uint32_t s_scan[COARSE_BINS], s_count[COARSE_BINS], s_next[COARSE_BINS];
	const uint32_t t = threadIdx.x;
	s_scan[t] = hist[t], s_count[t] = 0;
	__syncthreads();
	for (uint32_t step = 1; step < COARSE_BINS; step <<= 1) {  // inclusive prefix sums of the 256 counts
		const uint32_t v = t >= step ? s_scan[t - step] : 0u;
		__syncthreads();
		s_scan[t] += v;
		__syncthreads();
	}
	const uint32_t tile = blockIdx.x * COARSE_TILE;
	for (uint32_t j = 0; j < COARSE_TILE; j += COARSE_BLOCK)
		if (tile + j + t < n) atomicAdd(&s_count[bins[tile + j + t]], 1u);
	__syncthreads();
	// this tile's rays of a bin go behind those of the tiles that asked before it
	s_next[t] = (t ? s_scan[t - 1] : 0u) + (s_count[t] ? atomicAdd(&cursor[t], s_count[t]) : 0u);
	__syncthreads();
	for (uint32_t j = 0; j < COARSE_TILE; j += COARSE_BLOCK) {
		const uint32_t i = tile + j + t;
		if (i < n) order[atomicAdd(&s_next[bins[i]], 1u)] = index_base + i;
	}
	if (blockIdx.x == 0 && t <= 8) bounds[t] = t == 0 ? 0u : t == 8 ? n : s_scan[t * 32 - 1];
	// everything this workgroup wrote is visible to the whole GPU before it counts itself as done
	__threadfence();
	__syncthreads();
	if (t == 0 && atomicAdd(done, 1u) == gridDim.x - 1 && ready) {
		__threadfence();
		__hip_atomic_store(ready, ready_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// Announces that a chunk of a batch (its rays and its permutation) is in place: runs on the stream that put it there,
// after that work; the traversal kernel, already running, reads the word with acquire semantics.
__global__ void chunk_ready_kernel(uint32_t *chunks_ready, uint32_t value) {
	if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(chunks_ready, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// Waits up to `ticks` (100 MHz) for *word to become non-zero and stores whether it did: launched on one stream while a
// chunk_ready_kernel on another raises the word, it tells whether kernels of the two streams can run side by side at all
// (streams that share a hardware queue run one after the other).
__global__ void await_word_kernel(const uint32_t *word, uint32_t *seen, uint32_t ticks) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const unsigned long long since = __builtin_amdgcn_s_memrealtime();
	uint32_t v = 0;
	while ((v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && __builtin_amdgcn_s_memrealtime() - since < ticks)
		__builtin_amdgcn_s_sleep(32);
	*seen = v != 0;
}

// Positions in the sorted key array where the direction octant changes: the 8 parts of the batch that the 8
// XCDs start on (bounds[8] = n). One wave per boundary; every step cuts the candidate range into 64 parts.
// It also sets the launch's ray counters (the COUNTER_STRIDE * 8 words before `bounds`) to zero, if asked to: one small
// operation less on the stream before every sorted launch.
__global__ __launch_bounds__(512) void segment_bounds_kernel(const uint32_t *sorted_keys, uint32_t n, uint32_t *bounds, uint32_t *counters) {
	if (counters && threadIdx.x < 8 * COUNTER_STRIDE) counters[threadIdx.x] = 0;
	const uint32_t s = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint32_t want = s << 21;
	uint32_t lo = 0, hi = n;  // the first position whose key is >= want lies in [lo, hi]
	while (hi - lo > 0) {
		const uint32_t step = (hi - lo + 63u) / 64u;
		const uint32_t probe = lo + lane * step;  // lanes probe lo, lo + step, ...
		const bool below = probe < hi && sorted_keys[probe] < want;
		const unsigned long long m = __ballot(below);  // a prefix of the lanes, because the keys are sorted
		const uint32_t k = (uint32_t)__popcll(m);
		if (k == 0) {
			hi = lo;
		} else {
			const uint32_t new_lo = lo + (k - 1u) * step + 1u;
			hi = k < 64u && lo + k * step < hi ? lo + k * step : hi;
			lo = new_lo;
		}
	}
	if (lane == 0) bounds[s] = lo;
	if (threadIdx.x == 0) bounds[8] = n;
}

}  // namespace rtgi_dev
