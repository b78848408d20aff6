// rtgi_hip.hip -- gfx950 traversal kernels and the C ABI of include/rtgi_hip.h.
//
// closest_hit / any_hit follow rt/bbvh-base/bvh2.cpp:415-501 of the reference ray by ray: same node
// visiting order (near child first, right child first on equal distances), same pruning
// (entry distance < closest t), same triangle arithmetic (libgi/intersect.h:19-69, Cramer's rule in the
// reference's operation order) and same slab test (libgi/intersect.h:204-236). Built with
// -ffp-contract=off so that no multiply-add is fused; divisions are IEEE-correct (hipcc default).
//
// Execution model: one ray per lane, wave64. A launch is a fixed-size grid of persistent workgroups;
// each wave draws rays from a global counter and refills lanes whose ray has finished, so that lanes do
// not idle behind the longest ray of a 64-ray packet. The per-ray stack keeps its top in a register, the
// next STACK_LDS entries in LDS ([level][lane]: conflict-free for any mix of depths) and anything deeper
// in a global spill area sized from the tree depth at build time, so the stack cannot overflow.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdio>
#include <cstring>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "bvh_host.h"
#include "rtgi_hip.h"

namespace {

constexpr int BLOCK = 256;       // 4 waves per workgroup
constexpr int STACK_LDS = 16;    // stack entries per lane held in LDS
constexpr int REFILL_MIN = 16;   // refill a wave once this many lanes are idle
constexpr int LEAF_MIN = 12;     // leave the inner-node loop once this many lanes wait at a leaf

struct TraceArgs {
	const float4 *nodes;
	const float4 *tris;
	const rtgi_ray *rays;
	rtgi_hit *hits;        // closest_hit output
	uint8_t *occluded;     // any_hit output
	uint32_t *counts;      // optional: 2 per ray
	int32_t *spill;        // [level][global lane]
	uint32_t *ray_counter;
	uint32_t n_rays;
	uint32_t spill_stride;  // = gridDim.x * BLOCK
	int32_t root_ref;
};

struct LaneRay {
	float ox, oy, oz, dx, dy, dz, ix, iy, iz, t_min, t_max;
};

// Slab test of libgi/intersect.h:204-236, literal form: every select keeps the reference's operand order,
// which decides the outcome when a product is NaN (axis-parallel ray with its origin on a box plane).
__device__ __forceinline__ bool slab_exact(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                           const LaneRay &r, float &entry) {
	float ax = (lox - r.ox) * r.ix, bx = (hix - r.ox) * r.ix;
	float nx = (ax < bx) ? ax : bx, fx = (bx < ax) ? ax : bx;
	float ay = (loy - r.oy) * r.iy, by = (hiy - r.oy) * r.iy;
	float ny = (ay < by) ? ay : by, fy = (by < ay) ? ay : by;
	float az = (loz - r.oz) * r.iz, bz = (hiz - r.oz) * r.iz;
	float nz = (az < bz) ? az : bz, fz = (bz < az) ? az : bz;
	float t1 = (nx < ny) ? ny : nx;
	t1 = (nz < t1) ? t1 : nz;
	float t2 = (fx < fy) ? fx : fy;
	t2 = (fz < t2) ? fz : t2;
	entry = t1;
	return !(t1 > t2) && !(t2 < r.t_min) && !(t1 > r.t_max);
}

// The same test for rays whose origin and reciprocal direction are finite and non-zero: then no product
// can be NaN, each select is an exact min or max, and only the sign of a zero may differ, which no later
// comparison can see.
__device__ __forceinline__ bool slab_fast(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                          const LaneRay &r, float &entry) {
	float ax = (lox - r.ox) * r.ix, bx = (hix - r.ox) * r.ix;
	float ay = (loy - r.oy) * r.iy, by = (hiy - r.oy) * r.iy;
	float az = (loz - r.oz) * r.iz, bz = (hiz - r.oz) * r.iz;
	float t1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
	float t2 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
	entry = t1;
	return !(t1 > t2) && !(t2 < r.t_min) && !(t1 > r.t_max);
}

__device__ __forceinline__ bool ray_is_plain(const LaneRay &r) {
	auto ok = [](float v) { return __builtin_isfinite(v) && v != 0.0f; };
	return ok(r.ix) && ok(r.iy) && ok(r.iz) && __builtin_isfinite(r.ox) && __builtin_isfinite(r.oy) &&
	       __builtin_isfinite(r.oz);
}

template <bool ANY, bool COUNT>
__global__ __launch_bounds__(BLOCK) void trace_kernel(TraceArgs a) {
	__shared__ int32_t s_stack[STACK_LDS][BLOCK];
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const uint32_t gl = blockIdx.x * BLOCK + tid;

	LaneRay r = {};
	float best_t = FLT_MAX, best_b = 0.0f, best_g = 0.0f;
	uint32_t best_ref = 0;
	uint32_t ray_id = 0;
	uint32_t n_inner = 0, n_tri = 0;
	int32_t cur = 0;
	int sp = 0;          // entries below the register-held top
	bool alive = false;  // this lane carries an unfinished ray
	bool plain = true;
	bool more = true;    // wave-uniform: the ray counter may still have rays

	auto push = [&](int32_t v) {
		if (sp < STACK_LDS)
			s_stack[sp][tid] = v;
		else
			a.spill[(size_t)(sp - STACK_LDS) * a.spill_stride + gl] = v;
		++sp;
	};
	auto pop = [&]() -> int32_t {
		--sp;
		return (sp < STACK_LDS) ? s_stack[sp][tid] : a.spill[(size_t)(sp - STACK_LDS) * a.spill_stride + gl];
	};
	auto finish = [&]() {
		if (ANY)
			a.occluded[ray_id] = best_ref ? 1 : 0;
		else
			reinterpret_cast<float4 *>(a.hits)[ray_id] = make_float4(best_t, best_b, best_g, __uint_as_float(best_ref));
		if (COUNT) {
			a.counts[2 * (size_t)ray_id] = n_inner;
			a.counts[2 * (size_t)ray_id + 1] = n_tri;
		}
		alive = false;
	};
	// take the next entry off the stack, or end the ray
	auto advance = [&]() {
		if (sp == 0)
			finish();
		else
			cur = pop();
	};

	for (;;) {
		// ---- hand new rays to idle lanes ------------------------------------------------------
		unsigned long long idle = __ballot(!alive);
		if (more && (__popcll(idle) >= REFILL_MIN)) {
			uint32_t want = (uint32_t)__popcll(idle);
			uint32_t base = 0;
			if (lane == (int)__ffsll((long long)idle) - 1) base = atomicAdd(a.ray_counter, want);
			base = __shfl(base, (int)__ffsll((long long)idle) - 1);
			if (base + want >= a.n_rays) more = false;
			uint32_t mine = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
			if (!alive && mine < a.n_rays) {
				const float4 *rp = reinterpret_cast<const float4 *>(a.rays + mine);
				float4 q0 = rp[0], q1 = rp[1];
				r.ox = q0.x, r.oy = q0.y, r.oz = q0.z;
				r.dx = q0.w, r.dy = q1.x, r.dz = q1.y;
				r.t_min = q1.z, r.t_max = q1.w;
				r.ix = 1.0f / r.dx, r.iy = 1.0f / r.dy, r.iz = 1.0f / r.dz;  // ray::ray, libgi/rt.h:24
				plain = ray_is_plain(r);
				ray_id = mine;
				best_t = FLT_MAX, best_b = 0.0f, best_g = 0.0f, best_ref = 0;
				n_inner = n_tri = 0;
				cur = a.root_ref;
				sp = 0;
				alive = true;
			}
		}
		if (__ballot(alive) == 0) {
			if (!more) break;
			continue;
		}
		const bool wave_plain = __ballot(alive && !plain) == 0;

		// ---- inner nodes: every lane standing on an inner node takes one step per iteration ----
		for (;;) {
			const bool at_inner = alive && cur >= 0;
			if (__ballot(at_inner) == 0) break;
			if (__popcll(__ballot(alive && cur < 0)) >= LEAF_MIN) break;
			if (more && __popcll(__ballot(!alive)) >= REFILL_MIN) break;
			if (at_inner) {
				const float4 *np = a.nodes + (size_t)cur * 4;
				const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
				const int32_t link_l = __float_as_int(n0.w), link_r = __float_as_int(n1.w);
				if (COUNT) ++n_inner;
				float dl, dr;
				bool hl, hr;
				if (wave_plain) {
					hl = slab_fast(n0.x, n0.y, n0.z, n1.x, n1.y, n1.z, r, dl);
					hr = slab_fast(n2.x, n2.y, n2.z, n3.x, n3.y, n3.z, r, dr);
				} else {
					hl = slab_exact(n0.x, n0.y, n0.z, n1.x, n1.y, n1.z, r, dl);
					hr = slab_exact(n2.x, n2.y, n2.z, n3.x, n3.y, n3.z, r, dr);
				}
				if (!ANY) {  // bvh2.cpp:432-433; any_hit has no such pruning (:476-477)
					hl = hl && dl < best_t;
					hr = hr && dr < best_t;
				}
				if (hl && hr) {  // bvh2.cpp:434-442: the nearer child is visited first, the right one on a tie
					const bool left_first = dl < dr;
					push(left_first ? link_r : link_l);
					cur = left_first ? link_l : link_r;
				} else if (hl)
					cur = link_l;
				else if (hr)
					cur = link_r;
				else
					advance();
			}
		}

		// ---- leaves: test every triangle of the leaf, in slot order -----------------------------
		if (alive && cur < 0) {
			uint32_t slot = (uint32_t)~cur;
			bool done = false;
			for (;;) {
				const float4 *tp = a.tris + (size_t)slot * 3;
				const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
				if (COUNT) ++n_tri;
				// libgi/intersect.h:26-57, letters as there
				const float ea = t1.x, eb = t1.y, ec = t1.z;
				const float ed = t2.x, ee = t2.y, ef = t2.z;
				const float g = r.dx, h = r.dy, i = r.dz;
				const float j = t0.x - r.ox, k = t0.y - r.oy, l = t0.z - r.oz;
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
				if (tt > r.t_min && tt < r.t_max && beta > 0 && gamma > 0 && beta + gamma <= 1) {
					if (ANY) {
						best_ref = 1;
						done = true;
					} else if (tt < best_t) {
						best_t = tt, best_b = beta, best_g = gamma;
						best_ref = __float_as_uint(t0.w);
					}
				}
				if (done || __float_as_uint(t1.w)) break;
				++slot;
			}
			if (done)
				finish();
			else
				advance();
		}
	}
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
thread_local std::string g_create_error;

}  // namespace

struct rtgi_hip_tracer {
	int device = 0;  // -1: host-only handle (builds and inspects trees, cannot trace)
	rtgi::BuildConfig cfg;
	rtgi::HostTree tree;
	bool built = false;
	std::string error;

	float4 *d_nodes = nullptr;
	float4 *d_tris = nullptr;
	int32_t *d_spill = nullptr;
	uint32_t *d_counter = nullptr;
	int32_t root_ref = 0;
	uint32_t spill_levels = 0;
	uint32_t grid_blocks = 0;

	// staging for the host-buffer entry points
	void *d_stage_rays = nullptr;
	void *d_stage_out = nullptr;
	size_t stage_rays_cap = 0, stage_out_cap = 0;
};

namespace {

int fail(rtgi_hip_tracer *h, int code, const std::string &msg) {
	if (h) h->error = msg;
	return code;
}
#define HIP_TRY(h, expr)                                                                           \
	do {                                                                                           \
		hipError_t e_ = (expr);                                                                    \
		if (e_ != hipSuccess) return fail(h, RTGI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

void free_device_tree(rtgi_hip_tracer *h) {
	if (h->d_nodes) (void)hipFree(h->d_nodes);
	if (h->d_tris) (void)hipFree(h->d_tris);
	if (h->d_spill) (void)hipFree(h->d_spill);
	h->d_nodes = h->d_tris = nullptr;
	h->d_spill = nullptr;
	h->built = false;
}

int upload(rtgi_hip_tracer *h, const rtgi::SceneView &scene) {
	rtgi::FlatTree flat;
	rtgi::flatten_tree(h->tree, scene, h->cfg.layout, flat);
	if (h->device < 0) return RTGI_OK;
	HIP_TRY(h, hipSetDevice(h->device));
	free_device_tree(h);
	hipDeviceProp_t prop;
	HIP_TRY(h, hipGetDeviceProperties(&prop, h->device));
	h->grid_blocks = (uint32_t)prop.multiProcessorCount * 8u;  // 8 workgroups of 4 waves = 32 waves per CU
	// a near-first depth-first walk never has more pending entries than one per level
	// of the tree's depth (the entry in flight sits in a register, its pending siblings in memory)
	h->spill_levels = flat.depth > (uint32_t)STACK_LDS ? flat.depth - STACK_LDS : 0;
	size_t node_bytes = std::max<size_t>(flat.nodes.size() * sizeof(float), 64);
	size_t tri_bytes = std::max<size_t>(flat.tris.size() * sizeof(float), 48);
	HIP_TRY(h, hipMalloc((void **)&h->d_nodes, node_bytes));
	HIP_TRY(h, hipMalloc((void **)&h->d_tris, tri_bytes));
	if (!flat.nodes.empty()) HIP_TRY(h, hipMemcpy(h->d_nodes, flat.nodes.data(), flat.nodes.size() * sizeof(float), hipMemcpyHostToDevice));
	HIP_TRY(h, hipMemcpy(h->d_tris, flat.tris.data(), flat.tris.size() * sizeof(float), hipMemcpyHostToDevice));
	if (h->spill_levels)
		HIP_TRY(h, hipMalloc((void **)&h->d_spill, (size_t)h->spill_levels * h->grid_blocks * BLOCK * sizeof(int32_t)));
	if (!h->d_counter) HIP_TRY(h, hipMalloc((void **)&h->d_counter, 256));
	h->root_ref = flat.root_ref;
	h->built = true;
	return RTGI_OK;
}

template <bool ANY>
int launch(rtgi_hip_tracer *h, const rtgi_ray *d_rays, uint32_t n, void *d_out, uint32_t *d_counts, hipStream_t stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: traversal needs a GPU");
	if (!h->built) return fail(h, RTGI_ERR_NOT_BUILT, "no acceleration structure: call build first");
	if (n == 0) return RTGI_OK;
	if (!d_rays || !d_out) return fail(h, RTGI_ERR_INVALID, "null ray or output buffer");
	if ((reinterpret_cast<uintptr_t>(d_rays) & 15u) != 0) return fail(h, RTGI_ERR_INVALID, "ray buffer must be 16-byte aligned");
	if (!ANY && (reinterpret_cast<uintptr_t>(d_out) & 15u) != 0) return fail(h, RTGI_ERR_INVALID, "hit buffer must be 16-byte aligned");
	HIP_TRY(h, hipSetDevice(h->device));
	TraceArgs a;
	a.nodes = h->d_nodes;
	a.tris = h->d_tris;
	a.rays = d_rays;
	a.hits = ANY ? nullptr : (rtgi_hit *)d_out;
	a.occluded = ANY ? (uint8_t *)d_out : nullptr;
	a.counts = d_counts;
	a.spill = h->d_spill;
	a.ray_counter = h->d_counter;
	a.n_rays = n;
	a.root_ref = h->root_ref;
	uint32_t blocks = (n + BLOCK - 1) / BLOCK;
	if (blocks > h->grid_blocks) blocks = h->grid_blocks;
	a.spill_stride = blocks * BLOCK;
	HIP_TRY(h, hipMemsetAsync(h->d_counter, 0, sizeof(uint32_t), stream));
	if (d_counts)
		hipLaunchKernelGGL((trace_kernel<ANY, true>), dim3(blocks), dim3(BLOCK), 0, stream, a);
	else
		hipLaunchKernelGGL((trace_kernel<ANY, false>), dim3(blocks), dim3(BLOCK), 0, stream, a);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}

int ensure_stage(rtgi_hip_tracer *h, size_t ray_bytes, size_t out_bytes) {
	if (ray_bytes > h->stage_rays_cap) {
		if (h->d_stage_rays) (void)hipFree(h->d_stage_rays);
		h->d_stage_rays = nullptr;
		h->stage_rays_cap = 0;
		HIP_TRY(h, hipMalloc(&h->d_stage_rays, ray_bytes));
		h->stage_rays_cap = ray_bytes;
	}
	if (out_bytes > h->stage_out_cap) {
		if (h->d_stage_out) (void)hipFree(h->d_stage_out);
		h->d_stage_out = nullptr;
		h->stage_out_cap = 0;
		HIP_TRY(h, hipMalloc(&h->d_stage_out, out_bytes));
		h->stage_out_cap = out_bytes;
	}
	return RTGI_OK;
}

template <bool ANY>
int trace_host(rtgi_hip_tracer *h, const rtgi_ray *rays, uint32_t n, void *out) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: traversal needs a GPU");
	if (!h->built) return fail(h, RTGI_ERR_NOT_BUILT, "no acceleration structure: call build first");
	if (n == 0) return RTGI_OK;
	if (!rays || !out) return fail(h, RTGI_ERR_INVALID, "null ray or output buffer");
	HIP_TRY(h, hipSetDevice(h->device));
	const size_t out_bytes = (size_t)n * (ANY ? 1 : sizeof(rtgi_hit));
	int rc = ensure_stage(h, (size_t)n * sizeof(rtgi_ray), out_bytes);
	if (rc) return rc;
	HIP_TRY(h, hipMemcpy(h->d_stage_rays, rays, (size_t)n * sizeof(rtgi_ray), hipMemcpyHostToDevice));
	rc = launch<ANY>(h, (const rtgi_ray *)h->d_stage_rays, n, h->d_stage_out, nullptr, nullptr);
	if (rc) return rc;
	HIP_TRY(h, hipMemcpy(out, h->d_stage_out, out_bytes, hipMemcpyDeviceToHost));
	return RTGI_OK;
}

}  // namespace

extern "C" {

int rtgi_hip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

rtgi_hip_tracer *rtgi_hip_create(int device, int layout, int esc) {
	if ((layout != RTGI_LAYOUT_FLAT && layout != RTGI_LAYOUT_INDEXED)) {
		g_create_error = "there is no such bbvh variant";
		return nullptr;
	}
	if (layout == RTGI_LAYOUT_FLAT && esc) {
		g_create_error = "this combination is technically problematic";
		return nullptr;
	}
	int n = device == -1 ? 0 : rtgi_hip_device_count();
	if (device != -1 && (device < 0 || device >= n)) {
		g_create_error = "no such GPU: device " + std::to_string(device) + " of " + std::to_string(n);
		return nullptr;
	}
	rtgi_hip_tracer *h = new (std::nothrow) rtgi_hip_tracer;
	if (!h) {
		g_create_error = "out of memory";
		return nullptr;
	}
	h->device = device;
	h->cfg.layout = layout;
	h->cfg.esc = esc != 0;
	return h;
}

void rtgi_hip_destroy(rtgi_hip_tracer *h) {
	if (!h) return;
	if (h->device >= 0) (void)hipSetDevice(h->device);
	free_device_tree(h);
	if (h->d_counter) (void)hipFree(h->d_counter);
	if (h->d_stage_rays) (void)hipFree(h->d_stage_rays);
	if (h->d_stage_out) (void)hipFree(h->d_stage_out);
	delete h;
}

int rtgi_hip_interprete(rtgi_hip_tracer *h, const char *command, const char *args) {
	if (!h || !command) return RTGI_ERR_INVALID;
	if (std::string(command) != "bvh") return 0;
	std::istringstream in(args ? args : "");
	std::string value;
	in >> value;
	auto complete = [&]() { return !(in.bad() || in.fail()) && (in >> std::ws).eof(); };
	if (value == "om") {
		h->cfg.split = RTGI_SPLIT_OM;
		return 1;
	}
	if (value == "sm") {
		h->cfg.split = RTGI_SPLIT_SM;
		return 1;
	}
	if (value == "sah") {
		int planes = 0;
		in >> planes;
		if (!complete() || planes <= 0)
			return fail(h, RTGI_ERR_SYNTAX, "Syntax error, \"bvh sah\" requires exactly one positive integral value");
		h->cfg.split = RTGI_SPLIT_SAH;
		h->cfg.planes = planes;
		return 1;
	}
	if (value == "triangles") {
		in >> value;
		if (value == "multiple") {
			int count = 0;
			in >> count;
			if (!complete() || count <= 0)
				return fail(h, RTGI_ERR_SYNTAX, "Syntax error, \"triangles multiple\" requires exactly one positive integral value");
			h->cfg.max_tris = count;
			return 1;
		}
		if (value == "single") {
			h->cfg.max_tris = 1;
			return 1;
		}
		return fail(h, RTGI_ERR_SYNTAX, "Syntax error, \"bvh triangles\" requires a mode (single or multiple)");
	}
	if (value == "statistics") {
		if (h->tree.nodes.empty()) return fail(h, RTGI_ERR_NOT_BUILT, "no acceleration structure: call build first");
		fputs(rtgi::tree_statistics(h->tree).c_str(), stdout);
		fflush(stdout);
		return 1;
	}
	if (value == "export") {
		int depth = 0;
		std::string file;
		in >> depth >> file;
		if (!complete() || depth < 0 || file.empty())
			return fail(h, RTGI_ERR_SYNTAX, "Syntax error, \"export\" requires exactly one positive integral value and a filename.obj");
		if (h->tree.nodes.empty()) return fail(h, RTGI_ERR_NOT_BUILT, "no acceleration structure: call build first");
		std::string err;
		if (!rtgi::export_tree(h->tree, (uint32_t)depth, file, err)) return fail(h, RTGI_ERR_INVALID, err);
		printf("bvh exported to %s\n", file.c_str());
		fflush(stdout);
		return 1;
	}
	return fail(h, RTGI_ERR_SYNTAX, "Unknown bvh subcommand " + value);
}

int rtgi_hip_build(rtgi_hip_tracer *h, const void *vertices, size_t vertex_stride, uint32_t n_vertices,
                   const void *triangles, size_t tri_stride, uint32_t n_triangles) {
	if (!h) return RTGI_ERR_INVALID;
	rtgi::SceneView scene;
	scene.vertices = (const unsigned char *)vertices;
	scene.vertex_stride = vertex_stride;
	scene.n_vertices = n_vertices;
	scene.triangles = (const unsigned char *)triangles;
	scene.tri_stride = tri_stride;
	scene.n_triangles = n_triangles;
	std::string err;
	rtgi::HostTree tree;
	try {
		if (!rtgi::build_tree(h->cfg, scene, tree, err)) return fail(h, RTGI_ERR_INVALID, err);
		h->tree = std::move(tree);
		return upload(h, scene);
	} catch (const std::bad_alloc &) {
		return fail(h, RTGI_ERR_NOMEM, "out of host memory while building the BVH");
	}
}

int rtgi_hip_upload_bvh(rtgi_hip_tracer *h, const rtgi_node56 *nodes, uint32_t n_nodes, uint32_t root,
                        const uint32_t *index, uint32_t n_index, const void *vertices, size_t vertex_stride,
                        uint32_t n_vertices, const void *triangles, size_t tri_stride, uint32_t n_triangles) {
	if (!h) return RTGI_ERR_INVALID;
	if (!nodes || n_nodes == 0) return fail(h, RTGI_ERR_INVALID, "no nodes");
	rtgi::SceneView scene;
	scene.vertices = (const unsigned char *)vertices;
	scene.vertex_stride = vertex_stride;
	scene.n_vertices = n_vertices;
	scene.triangles = (const unsigned char *)triangles;
	scene.tri_stride = tri_stride;
	scene.n_triangles = n_triangles;
	std::string err;
	try {
		if (!rtgi::check_scene(scene, err)) return fail(h, RTGI_ERR_INVALID, err);
		rtgi::HostTree tree;
		tree.nodes.assign(nodes, nodes + n_nodes);
		tree.root = root;
		if (index)
			tree.order.assign(index, index + n_index);
		else {
			tree.order.resize(n_triangles);
			for (uint32_t i = 0; i < n_triangles; ++i) tree.order[i] = i;
		}
		if (!rtgi::validate_tree(tree, n_triangles, err)) return fail(h, RTGI_ERR_INVALID, err);
		h->tree = std::move(tree);
		return upload(h, scene);
	} catch (const std::bad_alloc &) {
		return fail(h, RTGI_ERR_NOMEM, "out of host memory while taking over the BVH");
	}
}

uint32_t rtgi_hip_num_nodes(const rtgi_hip_tracer *h) { return h ? (uint32_t)h->tree.nodes.size() : 0; }
uint32_t rtgi_hip_num_index(const rtgi_hip_tracer *h) { return h ? (uint32_t)h->tree.order.size() : 0; }
uint32_t rtgi_hip_root(const rtgi_hip_tracer *h) { return h ? h->tree.root : 0; }
uint32_t rtgi_hip_depth(const rtgi_hip_tracer *h) { return h ? h->tree.depth : 0; }
int rtgi_hip_get_nodes(const rtgi_hip_tracer *h, rtgi_node56 *out) {
	if (!h || !out) return RTGI_ERR_INVALID;
	memcpy(out, h->tree.nodes.data(), h->tree.nodes.size() * sizeof(rtgi_node56));
	return RTGI_OK;
}
int rtgi_hip_get_triangle_order(const rtgi_hip_tracer *h, uint32_t *out) {
	if (!h || !out) return RTGI_ERR_INVALID;
	memcpy(out, h->tree.order.data(), h->tree.order.size() * sizeof(uint32_t));
	return RTGI_OK;
}

int rtgi_hip_closest_hit(rtgi_hip_tracer *h, const rtgi_ray *rays, uint32_t n, rtgi_hit *hits) {
	return trace_host<false>(h, rays, n, hits);
}
int rtgi_hip_any_hit(rtgi_hip_tracer *h, const rtgi_ray *rays, uint32_t n, uint8_t *occluded) {
	return trace_host<true>(h, rays, n, occluded);
}
int rtgi_hip_closest_hit_device(rtgi_hip_tracer *h, const rtgi_ray *d_rays, uint32_t n, rtgi_hit *d_hits,
                                uint32_t *d_counts, void *stream) {
	return launch<false>(h, d_rays, n, d_hits, d_counts, (hipStream_t)stream);
}
int rtgi_hip_any_hit_device(rtgi_hip_tracer *h, const rtgi_ray *d_rays, uint32_t n, uint8_t *d_occluded,
                            uint32_t *d_counts, void *stream) {
	return launch<true>(h, d_rays, n, d_occluded, d_counts, (hipStream_t)stream);
}

const char *rtgi_hip_last_error(const rtgi_hip_tracer *h) { return h ? h->error.c_str() : g_create_error.c_str(); }

}  // extern "C"
