// rtgi_hip.hip -- gfx950 traversal kernels and the C ABI of include/rtgi_hip.h.
//
// The traversal kernel itself is in trace_kernel.h; this file holds the handle, the uploads and the launches.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "bvh_host.h"
#include "rtgi_hip.h"
#include "trace_kernel.h"
#include "wavefront_kernels.h"

using rtgi_dev::BLOCK;
using rtgi_dev::TraceArgs;
using rtgi_dev::trace_kernel;

namespace {

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
thread_local std::string g_create_error;
constexpr size_t PROBE_RAYS = 1u << 15;
constexpr size_t SMALL_BATCH = 256;  // host batches up to this size skip the staging copies

// Words at the start of the counter block: 8 ray counters, COUNTER_STRIDE words apart (a cache line each), then
// the 9 segment bounds of a sorted batch.
constexpr size_t COUNTER_WORDS = 8 * rtgi_dev::COUNTER_STRIDE + 16;

// Fixed scratch memory of rtgi_hip_closest_hit_device_noalloc: one array per GPU, part of the code object, so
// that entry point never has to allocate. Layout: counters | four sort arrays (keys, keys, indices, indices) of
// NOALLOC_MAX_RAYS words each | the radix sort's own scratch | stack spill area for trees deeper than the LDS stack.
constexpr uint32_t NOALLOC_MAX_RAYS = 1u << 21;
constexpr size_t NOALLOC_COUNTER_BYTES = 4096;
constexpr size_t NOALLOC_SORT_BYTES = 4 * (size_t)NOALLOC_MAX_RAYS * sizeof(uint32_t);
constexpr size_t NOALLOC_TEMP_BYTES = 4u << 20;
constexpr size_t NOALLOC_SPILL_BYTES = 8u << 20;
constexpr size_t NOALLOC_BYTES = NOALLOC_COUNTER_BYTES + NOALLOC_SORT_BYTES + NOALLOC_TEMP_BYTES + NOALLOC_SPILL_BYTES;
static_assert(COUNTER_WORDS * sizeof(uint32_t) <= NOALLOC_COUNTER_BYTES, "counter block too small");
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

// The array's address on each GPU (every GPU holds its own copy), looked up once per device. Host code must use
// these pointers, never the array's name.
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_guard;
unsigned char *g_scratch_ptr[MAX_DEVICES] = {nullptr};

// Device memory a traversal launch needs beyond rays and results.
struct Scratch {
	uint32_t *counter = nullptr;  // COUNTER_WORDS words
	uint32_t *sort = nullptr;     // four arrays of sort_cap words: keys, keys, indices, indices
	size_t sort_cap = 0;
	void *sort_temp = nullptr;
	size_t sort_temp_bytes = 0;
	int32_t *spill = nullptr;
	size_t spill_words = 0;
};

}  // namespace

// A few threads that copy memory in parallel: one thread moves ~10 GB/s, the staging copies of a host batch in
// ordinary (pageable) memory need several times that to keep up with the PCIe link.
class CopyPool {
public:
	~CopyPool() {
		{
			std::lock_guard<std::mutex> lock(guard);
			quit = true;
		}
		wake.notify_all();
		for (std::thread &t : workers) t.join();
	}
	// dst[0, bytes) = src[0, bytes), split over the workers and the calling thread; returns when it is done
	void copy(void *dst, const void *src, size_t bytes) {
		if (bytes < (1u << 20) || !started()) {  // (started() creates the workers on first use)
			memcpy(dst, src, bytes);
			return;
		}
		const size_t parts = workers.size() + 1;
		const size_t piece = ((bytes + parts - 1) / parts + 4095) & ~size_t(4095);
		{
			std::lock_guard<std::mutex> lock(guard);
			job_dst = static_cast<char *>(dst), job_src = static_cast<const char *>(src), job_bytes = bytes, job_piece = piece;
			pending = workers.size();
			++generation;
		}
		wake.notify_all();
		memcpy(dst, src, std::min(piece, bytes));  // piece 0 is the caller's
		std::unique_lock<std::mutex> lock(guard);
		done.wait(lock, [&] { return pending == 0; });
	}

private:
	bool started() {
		if (tried) return !workers.empty();
		tried = true;
		try {
			for (unsigned i = 0; i < 3; ++i) workers.emplace_back([this, i] { run(i + 1); });
		} catch (const std::system_error &) {  // no threads to be had: the caller copies alone
		}
		return !workers.empty();
	}
	void run(size_t part) {
		uint64_t seen = 0;
		for (;;) {
			std::unique_lock<std::mutex> lock(guard);
			wake.wait(lock, [&] { return quit || generation != seen; });
			if (quit) return;
			seen = generation;
			char *dst = job_dst;
			const char *src = job_src;
			const size_t bytes = job_bytes, lo = std::min(bytes, part * job_piece), hi = std::min(bytes, lo + job_piece);
			lock.unlock();
			if (hi > lo) memcpy(dst + lo, src + lo, hi - lo);
			lock.lock();
			if (--pending == 0) done.notify_all();
		}
	}
	std::vector<std::thread> workers;
	std::mutex guard;
	std::condition_variable wake, done;
	char *job_dst = nullptr;
	const char *job_src = nullptr;
	size_t job_bytes = 0, job_piece = 0, pending = 0;
	uint64_t generation = 0;
	bool quit = false, tried = false;
};

// Host batches go through the GPU in chunks on three streams: while chunk k is traced, chunk k + 1 comes up and
// the results of chunk k - 1 go down. Memory that is not page-locked is staged through page-locked slots.
struct HostPipe {
	static constexpr int SLOTS = 3;
	hipStream_t up = nullptr, run = nullptr, down = nullptr, aux = nullptr;  // aux: sorts that run beside a traversal launch
	hipEvent_t came_up[SLOTS] = {}, traced[SLOTS] = {}, went_down[SLOTS] = {};
	hipEvent_t chunk_up[rtgi_dev::MAX_CHUNKS] = {}, begun = nullptr, first_ready = nullptr, aux_done = nullptr;
	// RTGI_HIP_STREAM_TIMES: when each chunk of a streamed batch was up and sorted, and when its launch ended, on stderr
	hipEvent_t clock_begin = nullptr, clock_up[rtgi_dev::MAX_CHUNKS] = {}, clock_ready[rtgi_dev::MAX_CHUNKS] = {}, clock_end = nullptr;
	void *slot_rays[SLOTS] = {}, *slot_out[SLOTS] = {};
	size_t slot_rays_cap = 0;  // rays per slot
	CopyPool pool;
};

struct rtgi_hip_tracer {
	int device = 0;  // -1: host-only handle (builds and inspects trees, cannot trace)
	rtgi::BuildConfig cfg;
	rtgi::HostTree tree;
	bool built = false;
	std::string error;

	void *d_nodes = nullptr;
	void *d_tris = nullptr;
	float4 *d_corners = nullptr;  // triangle corners by reported id
	float4 *d_normals = nullptr;  // shading normal and material index by reported id (set_shading)
	float4 *d_albedo = nullptr;
	float4 *d_matparams = nullptr;  // per material: ior, roughness, brdf kind (set_shading: 1.3, 0.1, lambert; set_materials)
	// albedo textures (set_textures): texture coordinates by reported id, texture of each material, all texels
	float4 *d_tex_coords = nullptr, *d_texels = nullptr;
	uint4 *d_tex_of_material = nullptr;
	uint32_t n_materials = 0, n_scene_triangles = 0, n_scene_vertices = 0;
	// Launches that use the handle's scratch (ray counters, sort buffers, spill area, block counts) are ordered on
	// the GPU even when the caller moves from one stream to another: see scratch_begin / scratch_end.
	hipEvent_t scratch_free = nullptr;
	hipStream_t scratch_stream = nullptr;
	bool scratch_used = false;
	uint32_t *d_block_counts = nullptr;
	size_t block_counts_cap = 0;

	// per-launch kernel timing (profile_begin / profile_end) and the work sums of counting launches
	unsigned long long *d_totals = nullptr;
	bool profiling = false;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
	size_t events_used = 0;

	// spatial reordering of ray batches ("hip reorder auto|on|off")
	int reorder = 2;                  // 0 off, 1 always, 2 by the scene (see probe_pops)
	uint32_t reorder_min = 1u << 16;  // smaller batches are traced as they come
	float scene_lo[3] = {0, 0, 0}, scene_hi[3] = {1, 1, 1}, scene_inv_extent[3] = {1, 1, 1};
	uint32_t *d_sort = nullptr;  // keys in, keys out, index in, index out
	void *d_sort_temp = nullptr;
	size_t sort_cap = 0, sort_temp_bytes = 0;
	// "auto": sorting pays where incoherent rays are expensive. probe_pops is the mean number of inner nodes a
	// closest_hit ray pops when it starts on a random triangle in a random direction, measured at upload.
	float probe_pops = 0.0f;
	float probe_tris = 0.0f;  // ... and the mean number of triangles it tests; a triangle test counts as two node visits
	float reorder_min_pops = 120.0f;
	std::vector<rtgi_ray> probe_rays;
	uint32_t node_bytes = 0, tri_bytes = 0;
	int refill_min = 24, leaf_min = 12, drain_leaf_min = 4;
	bool refill_min_given = false;  // RTGI_HIP_REFILL_MIN is set: the scene's probe does not choose it
	bool leaf_min_given = false, drain_leaf_min_given = false;  // RTGI_HIP_LEAF_MIN / _DRAIN_LEAF_MIN are set: upload does not choose them by the tree's leaves
	bool top_in_lds = true;
	uint32_t blocks_per_cu = 1;
	// division of a workgroup's LDS (set at upload from the tree's depth and size, see trace_kernel.h)
	uint32_t stack_levels = 1, top_nodes = 0, lds_bytes = 0, implicit_limit = 0;
	bool boxes_ordered = false;  // every child box has min <= max on every axis: octant-pure waves skip the min / max of the slab test
	int force_stack_levels = 0, max_top_nodes = 1 << 30;
	uint32_t n_segments = 8;
	int32_t *d_spill = nullptr;
	uint32_t *d_counter = nullptr;
	int32_t root_ref = 0;
	uint32_t spill_levels = 0;
	uint32_t grid_blocks = 0;

	// staging for the host-buffer entry points
	void *d_stage_rays = nullptr;
	void *d_stage_out = nullptr;
	size_t stage_rays_cap = 0, stage_out_cap = 0;
	// small host batches (the per-ray interface): pinned host memory that the kernel reads and writes directly
	void *pinned = nullptr, *d_pinned = nullptr;
	HostPipe *pipe = nullptr;
	// page-locked host batches in ONE launch that takes up the later chunks as their upload and sort complete
	bool host_streamed = true;
	bool stream_coarse = true;         // chunks of such a batch are ordered by octant and length class only (two kernels instead of the sort's eight)
	uint32_t stream_min = 1u << 19;    // batches of at least this many rays
	uint32_t stream_lds_bytes = 96u << 10;  // the launch leaves LDS free for the sort kernels that run beside it
	uint32_t stream_wait_ticks = 2000000;  // how long a wave waits for a chunk before the launch gives up (100 MHz: 20 ms)
	uint64_t streamed_calls = 0, streamed_gave_up = 0;
	// rays this handle last wrote as camera rays (rtgi_hip_camera_rays*_device): neighbours in the buffer are neighbours in
	// the image and share their origin, which "reorder auto" does not sort
	const void *camera_rays_at = nullptr;
	uint32_t camera_rays_n = 0;
	uint64_t launches = 0, launches_sorted = 0;
	bool stream_probed = false;  // whether the two streams of the one-launch path have been seen to run side by side
	uint32_t *d_stream_ctl = nullptr;  // ray counters of the chunks | octant bounds of the chunks | chunks_ready | gave_up
	// where the chunks end, as shares of the batch: the first one short (the launch begins when it is sorted), the
	// following ones as large as the traversal of their predecessors takes the next upload and sort to complete
	float stream_parts[8] = {0.12f, 0.40f, 1.0f, 0, 0, 0, 0, 0};
	uint32_t host_chunk = 768u << 10;  // rays per chunk of a large host batch (every chunk's launch ends in its own tail)
	uint32_t host_lead = 256u << 10, host_lead_min = 192u << 10;  // the short first chunk: at most / batches of twice this have one
	bool host_direct = true;           // closest_hit on host batches: the kernel writes the hit records to host memory itself
};

namespace {

// The radix sort of (24-bit key, ray number) pairs: rocPRIM's Onesweep. For batches of up to a few million pairs the
// library's own choice of tile (and, below 2^20 pairs, of a merge sort) leaves most of the GPU idle; workgroups of 1024 x 4
// pairs and a histogram kernel of 512 x 16 sort the benchmark batch (1.45 M pairs) in 87 instead of 108 microseconds and
// half a million pairs in 58 instead of 120; from about 3 M pairs on the library's choice is the faster one
// (scripts/ubench/sort_configs.hip). The size query (temp == nullptr) and the sort itself choose alike.
// beside_launch: the sort runs while a traversal launch holds half of every CU's wave slots and registers; the 1024-thread
// workgroups would never be scheduled there, the library's own configuration is.
hipError_t sort_pairs(void *temp, size_t &temp_bytes, hipcub::DoubleBuffer<uint32_t> &keys, hipcub::DoubleBuffer<uint32_t> &index, uint32_t n,
                      hipStream_t stream, bool beside_launch = false) {
	if (n >= 3000000u || beside_launch) return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, index, (int)n, 0, rtgi_dev::KEY_BITS, stream);
	using small_batches = rocprim::radix_sort_config<
	    rocprim::default_config, rocprim::default_config,
	    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 16>, rocprim::kernel_config<1024, 4>, 8, rocprim::block_radix_rank_algorithm::match>, 1024>;
	rocprim::double_buffer<uint32_t> k(keys.Current(), keys.Alternate()), v(index.Current(), index.Alternate());
	const hipError_t e = rocprim::radix_sort_pairs<small_batches>(temp, temp_bytes, k, v, n, 0u, (unsigned)rtgi_dev::KEY_BITS, stream);
	if (temp && e == hipSuccess) {  // (which array of each pair holds the result)
		if (k.current() != keys.Current()) keys.selector ^= 1;
		if (v.current() != index.Current()) index.selector ^= 1;
	}
	return e;
}

int fail(rtgi_hip_tracer *h, int code, const std::string &msg) {
	if (h) h->error = msg;
	return code;
}
#define HIP_TRY(h, expr)                                                                           \
	do {                                                                                           \
		hipError_t e_ = (expr);                                                                    \
		if (e_ != hipSuccess) return fail(h, RTGI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

// Makes the handle's GPU current for the duration of a call and gives the caller's choice back afterwards (the
// host application, or PyTorch in this repository's tools, may be working on another device).
struct DeviceScope {
	int previous = -1;
	hipError_t status;
	explicit DeviceScope(int device) {
		if (device < 0) {  // host-only handle: nothing to select
			status = hipSuccess;
			return;
		}
		// (an error some earlier call on this thread chose to ignore is not this call's: launches below are checked with
		// hipGetLastError, which would hand it to them)
		(void)hipGetLastError();
		status = hipGetDevice(&previous);
		if (status != hipSuccess) previous = -1;
		status = previous == device ? hipSuccess : hipSetDevice(device);
	}
	~DeviceScope() {
		int now = -1;
		if (previous >= 0 && hipGetDevice(&now) == hipSuccess && now != previous) (void)hipSetDevice(previous);
	}
	DeviceScope(const DeviceScope &) = delete;
	DeviceScope &operator=(const DeviceScope &) = delete;
};

void free_textures(rtgi_hip_tracer *h) {
	if (h->d_tex_coords) (void)hipFree(h->d_tex_coords);
	if (h->d_texels) (void)hipFree(h->d_texels);
	if (h->d_tex_of_material) (void)hipFree(h->d_tex_of_material);
	h->d_tex_coords = h->d_texels = nullptr;
	h->d_tex_of_material = nullptr;
}

void free_device_tree(rtgi_hip_tracer *h) {
	if (h->d_nodes) (void)hipFree(h->d_nodes);
	if (h->d_tris) (void)hipFree(h->d_tris);
	if (h->d_spill) (void)hipFree(h->d_spill);
	if (h->d_corners) (void)hipFree(h->d_corners);
	if (h->d_normals) (void)hipFree(h->d_normals);
	if (h->d_albedo) (void)hipFree(h->d_albedo);
	if (h->d_matparams) (void)hipFree(h->d_matparams);
	free_textures(h);
	h->d_corners = h->d_normals = h->d_albedo = h->d_matparams = nullptr;
	h->n_materials = 0;
	h->d_nodes = h->d_tris = nullptr;
	h->d_spill = nullptr;
	h->built = false;
}

// Tuning knobs for experiments; the defaults are the measured optimum on MI355X.
void read_tunables(rtgi_hip_tracer *h) {
	auto number = [](const char *name, int lo, int hi, int fallback) {
		const char *e = getenv(name);
		return e ? std::max(lo, std::min(hi, atoi(e))) : fallback;
	};
	h->blocks_per_cu = (uint32_t)number("RTGI_HIP_BLOCKS_PER_CU", 1, 16, (int)h->blocks_per_cu);
	h->refill_min = number("RTGI_HIP_REFILL_MIN", 1, 64, h->refill_min);
	h->leaf_min = number("RTGI_HIP_LEAF_MIN", 1, 64, h->leaf_min);
	h->refill_min_given = getenv("RTGI_HIP_REFILL_MIN") != nullptr;
	h->leaf_min_given = getenv("RTGI_HIP_LEAF_MIN") != nullptr;
	h->drain_leaf_min_given = getenv("RTGI_HIP_DRAIN_LEAF_MIN") != nullptr;
	h->drain_leaf_min = number("RTGI_HIP_DRAIN_LEAF_MIN", 1, 64, h->drain_leaf_min);
	h->n_segments = number("RTGI_HIP_SEGMENTS", 1, 8, 8) == 1 ? 1u : 8u;
	h->top_in_lds = number("RTGI_HIP_TOP_IN_LDS", 0, 1, 1) != 0;
	h->force_stack_levels = number("RTGI_HIP_STACK_LEVELS", 0, 64, 0);
	h->max_top_nodes = number("RTGI_HIP_TOP_NODES", 0, 1 << 30, h->max_top_nodes);
	h->reorder = number("RTGI_HIP_REORDER", 0, 2, h->reorder);
	h->host_chunk = (uint32_t)number("RTGI_HIP_HOST_CHUNK", 1 << 16, 1 << 30, (int)h->host_chunk);
	h->host_lead = (uint32_t)number("RTGI_HIP_HOST_LEAD", 0, 1 << 30, (int)h->host_lead);
	h->host_direct = number("RTGI_HIP_HOST_DIRECT", 0, 1, 1) != 0;
	h->host_streamed = number("RTGI_HIP_HOST_STREAMED", 0, 1, 1) != 0;
	h->stream_coarse = number("RTGI_HIP_STREAM_COARSE", 0, 1, h->stream_coarse ? 1 : 0) != 0;
	h->stream_min = (uint32_t)number("RTGI_HIP_STREAM_MIN", 1 << 16, 1 << 30, (int)h->stream_min);
	h->stream_lds_bytes = (uint32_t)number("RTGI_HIP_STREAM_LDS_KIB", 16, 160, (int)(h->stream_lds_bytes >> 10)) << 10;
	h->stream_wait_ticks = (uint32_t)number("RTGI_HIP_STREAM_WAIT_TICKS", 1, 1 << 30, (int)h->stream_wait_ticks);
	if (const char *e = getenv("RTGI_HIP_STREAM_PARTS")) {  // e.g. "0.2,0.5,1"
		int c = 0;
		for (const char *q = e; *q && c < 8; ++c) {
			h->stream_parts[c] = (float)atof(q);
			while (*q && *q != ',') ++q;
			if (*q == ',') ++q;
		}
		for (; c < 8; ++c) h->stream_parts[c] = 0.0f;
	}
}

// The GPU form of h->tree (already flattened: `flat`) goes to the handle's GPU, together with what the launches derive
// from it (division of the LDS, spill area, probe rays).
int upload_flat(rtgi_hip_tracer *h, const rtgi::FlatTree &flat, const rtgi::SceneView &scene) {
	if (h->device < 0) return RTGI_OK;
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	free_device_tree(h);
	hipDeviceProp_t prop;
	HIP_TRY(h, hipGetDeviceProperties(&prop, h->device));
	// persistent grid: one workgroup of 16 waves per CU (four per SIMD), which owns the CU's LDS
	h->grid_blocks = (uint32_t)prop.multiProcessorCount * h->blocks_per_cu;
	// A near-first depth-first walk never has more pending entries than one per level of the tree's depth (the
	// entry in flight sits in a register, its pending siblings in memory): the LDS holds that many slots per lane,
	// and the first nodes of the breadth-first numbering in what is left. Deeper trees keep DEEP_STACK_LDS slots
	// in LDS and the rest of the stack in a global spill area.
	const uint32_t lds_cu = prop.maxSharedMemoryPerMultiProcessor >= 65536 ? (uint32_t)prop.maxSharedMemoryPerMultiProcessor : 65536u;
	// (what one workgroup may ask for: the opt-in limit where the runtime reports one, else the 64 KiB every GPU grants)
	const uint32_t lds_block = prop.sharedMemPerBlockOptin >= 65536 ? (uint32_t)prop.sharedMemPerBlockOptin : 65536u;
	const uint32_t lds_budget = std::min<uint32_t>({lds_cu / h->blocks_per_cu, lds_block, 160u << 10});
	const bool implicit_links = !getenv("RTGI_HIP_NO_IMPLICIT_LINKS") && flat.heap_levels >= 2;
	const uint32_t stack_need = std::max<uint32_t>(flat.depth, 1);
	h->stack_levels = stack_need <= (uint32_t)rtgi_dev::MAX_STACK_LDS ? stack_need : (uint32_t)rtgi_dev::DEEP_STACK_LDS;
	if (h->force_stack_levels) h->stack_levels = (uint32_t)h->force_stack_levels;
	while (h->stack_levels > 1 && (size_t)h->stack_levels * BLOCK * sizeof(int32_t) > lds_budget) --h->stack_levels;
	h->spill_levels = stack_need > h->stack_levels ? stack_need - h->stack_levels : 0;
	const size_t stack_bytes = (size_t)h->stack_levels * BLOCK * sizeof(int32_t);
	if (stack_bytes > lds_budget) return fail(h, RTGI_ERR_DEVICE, "this GPU's LDS does not hold one stack slot per lane");
	// the copy of the top of the tree: 48 bytes per node, 8 more for each copied node whose links must be read
	h->implicit_limit = implicit_links ? flat.implicit_limit : 0u;
	{
		const size_t left = lds_budget - stack_bytes;
		size_t fit = left / 48;
		if (fit > h->implicit_limit) fit = h->implicit_limit + (left - (size_t)h->implicit_limit * 48) / 56;
		h->top_nodes = h->top_in_lds ? (uint32_t)std::min<size_t>({fit, (size_t)flat.n_inner, (size_t)h->max_top_nodes}) : 0;
	}
	h->lds_bytes = (uint32_t)rtgi_dev::lds_bytes_for(h->stack_levels, h->top_nodes, h->implicit_limit);
	// buffer descriptors address the arrays with 32-bit byte offsets
	if (flat.n_inner >= (1u << 25) || flat.n_slots >= 44000000u)
		return fail(h, RTGI_ERR_INVALID, "scene too large: at most 2^25 inner nodes and 44M leaf slots");
	size_t node_bytes = std::max<size_t>(flat.nodes.size() * sizeof(float), 64);
	size_t tri_bytes = std::max<size_t>(flat.tris.size() * sizeof(float), 48);
	HIP_TRY(h, hipMalloc(&h->d_nodes, node_bytes));
	HIP_TRY(h, hipMalloc(&h->d_tris, tri_bytes));
	h->node_bytes = (uint32_t)node_bytes;
	h->tri_bytes = (uint32_t)tri_bytes;
	if (!flat.nodes.empty()) HIP_TRY(h, hipMemcpy(h->d_nodes, flat.nodes.data(), flat.nodes.size() * sizeof(float), hipMemcpyHostToDevice));
	HIP_TRY(h, hipMemcpy(h->d_tris, flat.tris.data(), flat.tris.size() * sizeof(float), hipMemcpyHostToDevice));
	HIP_TRY(h, hipMalloc((void **)&h->d_corners, std::max<size_t>(flat.corners.size() * sizeof(float), 48)));
	HIP_TRY(h, hipMemcpy(h->d_corners, flat.corners.data(), flat.corners.size() * sizeof(float), hipMemcpyHostToDevice));
	if (h->spill_levels)
		HIP_TRY(h, hipMalloc((void **)&h->d_spill, (size_t)h->spill_levels * h->grid_blocks * BLOCK * sizeof(int32_t)));
	if (!h->d_counter) HIP_TRY(h, hipMalloc((void **)&h->d_counter, COUNTER_WORDS * sizeof(uint32_t)));
	// bounds of the whole scene (union of the root's child boxes, or of the single leaf's triangles)
	float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
	for (uint32_t v = 0; v < scene.n_vertices; ++v)
		for (int k = 0; k < 3; ++k) {
			lo[k] = std::min(lo[k], scene.pos(v)[k]);
			hi[k] = std::max(hi[k], scene.pos(v)[k]);
		}
	for (int k = 0; k < 3; ++k) {
		h->scene_lo[k] = lo[k];
		h->scene_hi[k] = hi[k];
		h->scene_inv_extent[k] = hi[k] > lo[k] ? 1.0f / (hi[k] - lo[k]) : 0.0f;
	}
	h->root_ref = flat.root_ref;
	// How many lanes wait at leaves before the triangle code runs for them: a leaf's triangles are tested one after the other,
	// each a fetch and a hundred instructions for the lanes that wait, so trees with larger leaves gather more lanes first.
	// Measured on the 2^20-triangle soup (`bvh triangles multiple` 1 / 2 / 4 / 8: 4.6 / 21 / 53 / 128 triangle tests per ray):
	// best at 12 / 20-24 / 24-32 / 32 waiting lanes, +5 / +12 / +29 % over 12 for the last three.
	if (!h->leaf_min_given) {
		const double per_leaf = (double)flat.n_slots / (double)(flat.n_inner + 1u);
		h->leaf_min = (int)std::min(32.0, std::max(12.0, 12.0 + 10.0 * std::log2(std::max(per_leaf, 1.0))));
	}
	// (once the supply is dry a third of that: 4 for one triangle per leaf; four-triangle leaves 8-12: another 3-5 %)
	if (!h->drain_leaf_min_given) h->drain_leaf_min = std::max(4, h->leaf_min / 3);
	// node record: l.min.x r.min.x l.min.y r.min.y l.min.z r.min.z | the six max planes in the same order | links
	h->boxes_ordered = !getenv("RTGI_HIP_NO_OCTANT_WAVES");
	for (size_t i = 0; h->boxes_ordered && i + 16 <= flat.nodes.size(); i += 16)
		for (int k = 0; k < 6; ++k)
			if (!(flat.nodes[i + k] <= flat.nodes[i + 6 + k])) h->boxes_ordered = false;
	h->n_scene_triangles = scene.n_triangles;
	h->n_scene_vertices = scene.n_vertices;
	h->built = true;
	// probe rays for the reorder policy: from random points of random triangles into random directions
	h->probe_rays.clear();
	h->probe_pops = 0.0f;
	h->probe_tris = 0.0f;
	if (!h->refill_min_given) h->refill_min = 40;  // (scenes too small for probe rays: short rays; larger ones: see probe_scene)
	if (flat.n_slots >= 4096) {
		uint64_t state = 0x9E3779B97F4A7C15ull;
		auto uniform = [&]() {  // 24 random bits -> [0, 1)
			state = state * 6364136223846793005ull + 1442695040888963407ull;
			return (float)((state >> 40) & 0xFFFFFFu) * (1.0f / 16777216.0f);
		};
		h->probe_rays.resize(PROBE_RAYS);
		for (rtgi_ray &r : h->probe_rays) {
			const size_t slot = std::min<size_t>((size_t)(uniform() * (float)flat.n_slots), flat.n_slots - 1);
			const float *t = &flat.tris[12 * slot];  // A | A - B | A - C
			float u = uniform(), v = uniform();
			if (u + v > 1.0f) u = 1.0f - u, v = 1.0f - v;
			float d[3], len2;
			do {  // uniform direction: a point in the unit ball, normalised
				len2 = 0.0f;
				for (int k = 0; k < 3; ++k) d[k] = 2.0f * uniform() - 1.0f, len2 += d[k] * d[k];
			} while (len2 > 1.0f || len2 < 1e-4f);
			const float inv = 1.0f / std::sqrt(len2);
			for (int k = 0; k < 3; ++k) {
				r.o[k] = t[k] - u * t[4 + k] - v * t[8 + k];
				r.d[k] = d[k] * inv;
			}
			r.t_min = 1e-4f;
			r.t_max = FLT_MAX;
		}
	}
	return RTGI_OK;
}

int upload(rtgi_hip_tracer *h, const rtgi::SceneView &scene) {
	rtgi::FlatTree flat;
	const std::chrono::steady_clock::time_point flatten_begin = std::chrono::steady_clock::now();
	rtgi::flatten_tree(h->tree, scene, h->cfg.layout, flat);
	if (getenv("RTGI_HIP_BUILD_TIMES"))
		fprintf(stderr, "rtgi_hip_build: flatten %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - flatten_begin).count());
	return upload_flat(h, flat, scene);
}

// RTGI_HIP_DEBUG_SYNC=1: wait for the stream after every stage and name the stage on stderr, so that a
// failing launch can be told from its neighbours.
hipError_t debug_sync(rtgi_hip_tracer *h, const char *stage, hipStream_t stream) {
	static const bool on = getenv("RTGI_HIP_DEBUG_SYNC") != nullptr;
	hipError_t e = hipGetLastError();
	if (e == hipSuccess && on) {
		e = hipStreamSynchronize(stream);
		fprintf(stderr, "[rtgi_hip] %s: %s\n", stage, hipGetErrorString(e));
		fflush(stderr);
	}
	(void)h;
	return e;
}

// The handle has one set of scratch buffers. Work that uses them on a stream other than the previous one's first
// waits (on the GPU, not on the host) for the previous one to finish.
int scratch_begin(rtgi_hip_tracer *h, hipStream_t stream) {
	if (!h->scratch_free) HIP_TRY(h, hipEventCreateWithFlags(&h->scratch_free, hipEventDisableTiming));
	if (h->scratch_used && stream != h->scratch_stream) HIP_TRY(h, hipStreamWaitEvent(stream, h->scratch_free, 0));
	return RTGI_OK;
}
int scratch_end(rtgi_hip_tracer *h, hipStream_t stream) {
	HIP_TRY(h, hipEventRecord(h->scratch_free, stream));
	h->scratch_stream = stream;
	h->scratch_used = true;
	return RTGI_OK;
}

struct BatchView {
	const rtgi_ray *rays;
	uint32_t n;
	bool any;
	void *out;
};

// The handle's own scratch memory, grown on demand for a launch of n rays (sort = the launch will be sorted).
int handle_scratch(rtgi_hip_tracer *h, uint32_t n, bool sort, Scratch &sc) {
	if (!h->d_counter) HIP_TRY(h, hipMalloc((void **)&h->d_counter, COUNTER_WORDS * sizeof(uint32_t)));
	if (sort && n > h->sort_cap) {
		if (h->d_sort) (void)hipFree(h->d_sort);
		h->d_sort = nullptr;
		h->sort_cap = 0;
		const size_t cap = (size_t)n + n / 8 + 1024;
		HIP_TRY(h, hipMalloc((void **)&h->d_sort, 4 * cap * sizeof(uint32_t)));
		h->sort_cap = cap;
	}
	if (sort) {
		// the sort's scratch size depends on the item count (the library picks its algorithm by size): ask for this n
		size_t temp = 0;
		hipcub::DoubleBuffer<uint32_t> keys(h->d_sort, h->d_sort + h->sort_cap), index(h->d_sort + 2 * h->sort_cap, h->d_sort + 3 * h->sort_cap);
		HIP_TRY(h, sort_pairs(nullptr, temp, keys, index, (uint32_t)(n), nullptr));
		if (temp > h->sort_temp_bytes) {
			if (h->d_sort_temp) (void)hipFree(h->d_sort_temp);
			h->d_sort_temp = nullptr;
			h->sort_temp_bytes = 0;
			HIP_TRY(h, hipMalloc(&h->d_sort_temp, temp + temp / 4 + 256));
			h->sort_temp_bytes = temp + temp / 4 + 256;
		}
	}
	sc.counter = h->d_counter;
	sc.sort = h->d_sort;
	sc.sort_cap = h->sort_cap;
	sc.sort_temp = h->d_sort_temp;
	sc.sort_temp_bytes = h->sort_temp_bytes;
	sc.spill = h->d_spill;
	sc.spill_words = (size_t)h->spill_levels * h->grid_blocks * BLOCK;
	return RTGI_OK;
}

// Sorts the rays of a launch (one or several batches, numbered through) by the key of ray_keys_kernel, inside the
// scratch memory given; *order receives the array that holds the permutation, the octant boundaries go behind the
// ray counters. Allocates nothing; the caller has checked that n rays and the sort's own scratch fit.
int sort_rays(rtgi_hip_tracer *h, const BatchView *batches, uint32_t n_batches, uint32_t n, const Scratch &sc, hipStream_t stream,
              const uint32_t **order) {
	const size_t cap = sc.sort_cap;
	hipcub::DoubleBuffer<uint32_t> keys(sc.sort, sc.sort + cap), index(sc.sort + 2 * cap, sc.sort + 3 * cap);
	rtgi_dev::KeyArgs k;
	k.keys = keys.Current();
	k.index = index.Current();
	float diag2 = 0.0f;
	for (int c = 0; c < 3; ++c) {
		k.lo[c] = h->scene_lo[c], k.hi[c] = h->scene_hi[c], k.inv_extent[c] = h->scene_inv_extent[c];
		diag2 += (h->scene_hi[c] - h->scene_lo[c]) * (h->scene_hi[c] - h->scene_lo[c]);
	}
	k.inv_diagonal = diag2 > 0.0f ? 1.0f / std::sqrt(diag2) : 0.0f;
	k.first = 0;
	k.index_base = 0;
	for (uint32_t b = 0; b < n_batches; ++b) {
		k.rays = (const float4 *)batches[b].rays;
		k.n = batches[b].n;
		hipLaunchKernelGGL(rtgi_dev::ray_keys_kernel, dim3((k.n + 255) / 256), dim3(256), 0, stream, k);
		HIP_TRY(h, debug_sync(h, "ray_keys_kernel", stream));
		k.first += k.n;
	}
	// Only the KEY_BITS low bits hold the key. (Sorting a bit range that does not start at bit 0 returned wrong
	// permutations for batches of up to a few 100 k items with rocPRIM of ROCm 7.2.) The two array pairs are the
	// sort's double buffers: which of each pair holds the result is known on the host when the call returns.
	size_t temp_size = sc.sort_temp_bytes;
	HIP_TRY(h, sort_pairs(sc.sort_temp, temp_size, keys, index, (uint32_t)(n), stream));
	HIP_TRY(h, debug_sync(h, "radix sort", stream));
	hipLaunchKernelGGL(rtgi_dev::segment_bounds_kernel, dim3(1), dim3(512), 0, stream, (const uint32_t *)keys.Current(), n,
	                   sc.counter + 8 * rtgi_dev::COUNTER_STRIDE, sc.counter);  // (also zeroes the ray counters)
	HIP_TRY(h, debug_sync(h, "segment_bounds_kernel", stream));
	*order = index.Current();
	return RTGI_OK;
}

// A launch with `lds` bytes of dynamic LDS. More than 64 KiB needs the kernel's limit raised first, per kernel
// instantiation and GPU, to what this launch asks for (upload() keeps that within the GPU's per-workgroup limit).
template <int MODE, bool COUNT, bool DEEP>
hipError_t launch_kernel(int device, uint32_t blocks, uint32_t lds, hipStream_t stream, const TraceArgs &a) {
	static std::mutex guard;
	static uint32_t raised[MAX_DEVICES] = {};
	if (lds > (64u << 10) && device >= 0 && device < MAX_DEVICES) {
		std::lock_guard<std::mutex> lock(guard);
		if (lds > raised[device]) {
			hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&trace_kernel<MODE, COUNT, DEEP>),
			                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			if (e != hipSuccess) return e;
			raised[device] = lds;
		}
	}
	hipLaunchKernelGGL((trace_kernel<MODE, COUNT, DEEP>), dim3(blocks), dim3(BLOCK), lds, stream, a);
	return hipGetLastError();
}
template <int MODE>
hipError_t launch_variant(int device, bool count, bool deep, uint32_t blocks, uint32_t lds, hipStream_t stream, const TraceArgs &a) {
	if (count && deep) return launch_kernel<MODE, true, true>(device, blocks, lds, stream, a);
	if (count) return launch_kernel<MODE, true, false>(device, blocks, lds, stream, a);
	if (deep) return launch_kernel<MODE, false, true>(device, blocks, lds, stream, a);
	return launch_kernel<MODE, false, false>(device, blocks, lds, stream, a);
}

// Records the "scratch is free again" event when a launch sequence ends, on every path out of it.
struct ScratchTurn {
	rtgi_hip_tracer *h;
	hipStream_t stream;
	bool open = false;
	int begin() {
		int rc = scratch_begin(h, stream);
		open = rc == RTGI_OK;
		return rc;
	}
	int end() {
		open = false;
		return scratch_end(h, stream);
	}
	~ScratchTurn() {
		if (open) (void)scratch_end(h, stream);
	}
};

// One traversal launch over one batch or over several (each closest_hit or any_hit). d_counts: single batch only.
// fixed == nullptr: the handle's own scratch memory is used and grown as needed. Otherwise the launch lives in
// *fixed, allocates nothing, creates nothing, and adapts to what fits (the sort is left out, the grid shrinks
// until the spill area holds its stacks).
int launch_batches(rtgi_hip_tracer *h, const BatchView *given, uint32_t n_given, uint32_t *d_counts, hipStream_t stream,
                   const Scratch *fixed = nullptr) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: traversal needs a GPU");
	if (!h->built) return fail(h, RTGI_ERR_NOT_BUILT, "no acceleration structure: call build first");
	if (n_given > (uint32_t)rtgi_dev::MAX_BATCHES) return fail(h, RTGI_ERR_INVALID, "at most 32 batches per launch");
	BatchView batches[rtgi_dev::MAX_BATCHES];
	uint32_t n_batches = 0;
	uint64_t total = 0;
	for (uint32_t b = 0; b < n_given; ++b) {
		const BatchView &v = given[b];
		if (v.n == 0) continue;
		if (!v.rays || !v.out) return fail(h, RTGI_ERR_INVALID, "null ray or output buffer");
		if ((reinterpret_cast<uintptr_t>(v.rays) & 15u) != 0) return fail(h, RTGI_ERR_INVALID, "ray buffer must be 16-byte aligned");
		if (!v.any && (reinterpret_cast<uintptr_t>(v.out) & 15u) != 0) return fail(h, RTGI_ERR_INVALID, "hit buffer must be 16-byte aligned");
		batches[n_batches++] = v;
		total += v.n;
	}
	if (n_batches == 0) return RTGI_OK;
	if (total > 0xfffffff0ull) return fail(h, RTGI_ERR_INVALID, "too many rays in one launch");
	if (n_batches > 1 && d_counts) return fail(h, RTGI_ERR_INVALID, "per-ray counts are available for single batches only");
	const uint32_t n = (uint32_t)total;
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	ScratchTurn turn{h, stream};
	if (!fixed)  // (launches in fixed scratch are ordered by their callers)
		if (int rc = turn.begin()) return rc;
	// (auto: not for a batch that this handle itself filled with camera rays, which are in image order)
	const bool camera_batch = n_batches == 1 && batches[0].rays == h->camera_rays_at && n == h->camera_rays_n;
	bool sort = (h->reorder == 1 || (h->reorder == 2 && h->probe_pops + 2.0f * h->probe_tris > h->reorder_min_pops && !camera_batch)) && n >= h->reorder_min &&
	            n < (1u << 31);  // (the sort counts in int)
	Scratch sc;
	if (fixed) {
		sc = *fixed;
		if (sort) {  // only if the rays and the sort's own scratch fit
			size_t temp = 0;
			hipcub::DoubleBuffer<uint32_t> keys(sc.sort, sc.sort + sc.sort_cap), index(sc.sort + 2 * sc.sort_cap, sc.sort + 3 * sc.sort_cap);
			sort = n <= sc.sort_cap &&
			       sort_pairs(nullptr, temp, keys, index, (uint32_t)(n), stream) == hipSuccess &&
			       temp <= sc.sort_temp_bytes;
		}
	} else if (int rc = handle_scratch(h, n, sort, sc))
		return rc;
	TraceArgs a;
	memset(&a, 0, sizeof(a));
	a.nodes = h->d_nodes;
	a.tris = h->d_tris;
	const int mode = n_batches > 1 ? 2 : (batches[0].any ? 1 : 0);
	if (mode == 2) {
		a.n_batches = n_batches;
		uint32_t begin = 0;
		for (uint32_t b = 0; b < n_batches; ++b) {
			a.batch_begin[b] = begin;
			a.batch_rays[b] = (const float4 *)batches[b].rays;
			a.batch_out[b] = batches[b].out;
			if (batches[b].any) a.batch_any |= 1u << b;
			begin += batches[b].n;
		}
	} else {
		a.rays = (const float4 *)batches[0].rays;
		a.hits = mode == 0 ? (float4 *)batches[0].out : nullptr;
		a.occluded = mode == 1 ? (uint8_t *)batches[0].out : nullptr;
	}
	a.counts = d_counts;
	a.totals = (!fixed && h->profiling && d_counts) ? h->d_totals : nullptr;
	a.spill = sc.spill;
	a.ray_counter = sc.counter;
	a.n_rays = n;
	a.n_segments = h->n_segments;
	a.node_bytes = h->node_bytes;
	a.tri_bytes = h->tri_bytes;
	a.root_ref = h->root_ref;
#ifdef RTGI_DIAG
	static unsigned long long *d_diag = nullptr;
	auto diag_reset = [&]() -> hipError_t {
		unsigned long long init[16] = {0};
		init[8] = init[10] = ~0ull;
		return hipMemcpy(d_diag, init, sizeof(init), hipMemcpyHostToDevice);
	};
	if (!d_diag) {
		HIP_TRY(h, hipMalloc((void **)&d_diag, 128));
		HIP_TRY(h, diag_reset());
	}
	a.diag = d_diag;
	if (getenv("RTGI_DIAG_PRINT")) {
		unsigned long long v[16];
		HIP_TRY(h, hipDeviceSynchronize());
		HIP_TRY(h, hipMemcpy(v, d_diag, 128, hipMemcpyDeviceToHost));
		HIP_TRY(h, diag_reset());
		if (v[7])
			fprintf(stderr, "diag(prev launch): inner iters fed %llu (%.1f%% lanes) | dry %llu (%.1f%% lanes) | leaf iters %llu (%.1f%% lanes) | refills %llu outer %llu | wall %.3f ms, supply dry at %.3f..%.3f ms, mean wave life %.3f ms\n",
			        v[0], v[0] ? 100.0 * v[1] / (64.0 * v[0]) : 0.0, v[2], v[2] ? 100.0 * v[3] / (64.0 * v[2]) : 0.0, v[4], v[4] ? 100.0 * v[5] / (64.0 * v[4]) : 0.0, v[6], v[7],
			        (v[9] - v[8]) * 1e-5, (v[10] - v[8]) * 1e-5, (v[11] - v[8]) * 1e-5, v[12] * 1e-5 / (double)(h->grid_blocks * (BLOCK / 64)));
	}
#endif
	a.refill_min = h->refill_min;
	a.leaf_min = h->leaf_min;
	a.drain_leaf_min = h->drain_leaf_min;
	a.top_nodes = (int32_t)h->top_nodes;
	a.implicit_limit = (int32_t)h->implicit_limit;
	a.boxes_ordered = h->boxes_ordered ? 1 : 0;
	a.stack_levels = (int32_t)h->stack_levels;
	// (a small batch still spreads over the CUs: a workgroup per 256 rays, its other waves find the supply dry at once)
	uint32_t blocks = (n + 255) / 256;
	if (blocks > h->grid_blocks) blocks = h->grid_blocks;
	const bool deep = h->spill_levels > 0;  // the tree is deeper than the LDS part of the stack
	if (deep) {  // every lane of the grid owns spill_levels words of the spill area
		const size_t most = sc.spill_words / ((size_t)h->spill_levels * BLOCK);
		if (most == 0) return fail(h, RTGI_ERR_INVALID, "tree too deep for the fixed scratch memory");
		if (blocks > most) blocks = (uint32_t)most;
	}
	a.spill_stride = blocks * BLOCK;
	a.order = nullptr;
	a.seg_bounds = nullptr;
	++h->launches;
	h->launches_sorted += sort;
	if (sort) {
		int rc = sort_rays(h, batches, n_batches, n, sc, stream, &a.order);
		if (rc) return rc;
		a.seg_bounds = sc.counter + 8 * rtgi_dev::COUNTER_STRIDE;  // one part per direction octant
		a.n_segments = 8;
	}
	if (!sort) HIP_TRY(h, hipMemsetAsync(sc.counter, 0, 8 * rtgi_dev::COUNTER_STRIDE * sizeof(uint32_t), stream));
	hipEvent_t ev_start = nullptr, ev_stop = nullptr;
	if (!fixed && h->profiling && h->events_used < 4096) {
		if (h->events_used == h->events.size()) {
			hipEvent_t e0, e1;
			HIP_TRY(h, hipEventCreate(&e0));
			HIP_TRY(h, hipEventCreate(&e1));
			h->events.push_back({e0, e1});
		}
		ev_start = h->events[h->events_used].first;
		ev_stop = h->events[h->events_used].second;
		++h->events_used;
		HIP_TRY(h, hipEventRecord(ev_start, stream));
	}
	hipError_t launched;
	if (mode == 2)
		launched = launch_variant<2>(h->device, false, deep, blocks, h->lds_bytes, stream, a);
	else if (mode == 1)
		launched = launch_variant<1>(h->device, d_counts != nullptr, deep, blocks, h->lds_bytes, stream, a);
	else
		launched = launch_variant<0>(h->device, d_counts != nullptr, deep, blocks, h->lds_bytes, stream, a);
	HIP_TRY(h, launched);
	if (ev_stop) HIP_TRY(h, hipEventRecord(ev_stop, stream));
	if (!fixed)
		if (int rc = turn.end()) return rc;
	HIP_TRY(h, debug_sync(h, mode == 2 ? "trace_kernel batches" : mode == 1 ? "trace_kernel any_hit" : "trace_kernel closest_hit", stream));
	return RTGI_OK;
}

template <bool ANY>
int launch(rtgi_hip_tracer *h, const rtgi_ray *d_rays, uint32_t n, void *d_out, uint32_t *d_counts, hipStream_t stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (n != 0 && (!d_rays || !d_out)) return fail(h, RTGI_ERR_INVALID, "null ray or output buffer");
	const BatchView one{d_rays, n, ANY, d_out};
	return launch_batches(h, &one, 1, d_counts, stream);
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

// Streams and events of the host-batch pipeline, created on first use.
int ensure_pipe(rtgi_hip_tracer *h) {
	if (h->pipe) return RTGI_OK;
	HostPipe *p = new (std::nothrow) HostPipe;
	if (!p) return fail(h, RTGI_ERR_NOMEM, "out of host memory");
	h->pipe = p;  // (destroy releases whatever part of it exists)
	HIP_TRY(h, hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking));
	HIP_TRY(h, hipStreamCreateWithFlags(&p->run, hipStreamNonBlocking));
	HIP_TRY(h, hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking));
	HIP_TRY(h, hipStreamCreateWithFlags(&p->aux, hipStreamNonBlocking));
	for (int i = 0; i < rtgi_dev::MAX_CHUNKS; ++i) HIP_TRY(h, hipEventCreateWithFlags(&p->chunk_up[i], hipEventDisableTiming));
	HIP_TRY(h, hipEventCreateWithFlags(&p->first_ready, hipEventDisableTiming));
	HIP_TRY(h, hipEventCreateWithFlags(&p->aux_done, hipEventDisableTiming));
	if (getenv("RTGI_HIP_STREAM_TIMES")) {
		HIP_TRY(h, hipEventCreate(&p->clock_begin));
		HIP_TRY(h, hipEventCreate(&p->clock_end));
		for (int i = 0; i < rtgi_dev::MAX_CHUNKS; ++i) {
			HIP_TRY(h, hipEventCreate(&p->clock_up[i]));
			HIP_TRY(h, hipEventCreate(&p->clock_ready[i]));
		}
	}
	HIP_TRY(h, hipEventCreateWithFlags(&p->begun, hipEventDisableTiming));
	for (int i = 0; i < HostPipe::SLOTS; ++i) {
		HIP_TRY(h, hipEventCreateWithFlags(&p->came_up[i], hipEventDisableTiming));
		HIP_TRY(h, hipEventCreateWithFlags(&p->traced[i], hipEventDisableTiming));
		HIP_TRY(h, hipEventCreateWithFlags(&p->went_down[i], hipEventDisableTiming));
	}
	return RTGI_OK;
}

// Page-locked staging slots for `chunk` rays and their results each.
int ensure_slots(rtgi_hip_tracer *h, uint32_t chunk) {
	HostPipe &p = *h->pipe;
	if (chunk <= p.slot_rays_cap) return RTGI_OK;
	for (int i = 0; i < HostPipe::SLOTS; ++i) {
		if (p.slot_rays[i]) (void)hipHostFree(p.slot_rays[i]);
		if (p.slot_out[i]) (void)hipHostFree(p.slot_out[i]);
		p.slot_rays[i] = p.slot_out[i] = nullptr;
	}
	p.slot_rays_cap = 0;
	for (int i = 0; i < HostPipe::SLOTS; ++i) {
		HIP_TRY(h, hipHostMalloc(&p.slot_rays[i], (size_t)chunk * sizeof(rtgi_ray), hipHostMallocDefault));
		HIP_TRY(h, hipHostMalloc(&p.slot_out[i], (size_t)chunk * sizeof(rtgi_hit), hipHostMallocDefault));
	}
	p.slot_rays_cap = chunk;
	return RTGI_OK;
}

// Whether the copy engines can reach this host memory directly (rtgi_hip_host_alloc / rtgi_hip_host_register).
bool is_page_locked(const void *ptr) {
	hipPointerAttribute_t attr;
	if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
		(void)hipGetLastError();  // ordinary memory is unknown to the runtime: not an error of this call
		return false;
	}
	return attr.type == hipMemoryTypeHost;
}

// The one-launch path needs kernels of p.aux to run WHILE a kernel of p.run runs. Streams are mapped onto few hardware
// queues, and two streams on one queue take turns: a launch there would wait for its chunks until it gives up. Tried
// once per handle, with two one-thread kernels: one on p.run waits (1 ms at most) for a word that one on p.aux raises. If it
// never sees the word, p.aux is replaced by a new stream (which lands on another queue) and the trial repeated, a few times.
int probe_side_by_side(rtgi_hip_tracer *h, uint32_t *d_words, bool *ok) {
	using namespace rtgi_dev;
	HostPipe &p = *h->pipe;
	*ok = false;
	for (int attempt = 0; attempt < 6 && !*ok; ++attempt) {
		if (attempt > 0) {
			HIP_TRY(h, hipStreamSynchronize(p.aux));
			HIP_TRY(h, hipStreamDestroy(p.aux));
			p.aux = nullptr;
			HIP_TRY(h, hipStreamCreateWithFlags(&p.aux, hipStreamNonBlocking));
		}
		HIP_TRY(h, hipMemsetAsync(d_words, 0, 2 * sizeof(uint32_t), p.run));
		HIP_TRY(h, hipEventRecord(p.begun, p.run));
		hipLaunchKernelGGL(await_word_kernel, dim3(1), dim3(64), 0, p.run, d_words, d_words + 1, 100000u);
		HIP_TRY(h, hipStreamWaitEvent(p.aux, p.begun, 0));
		// (RTGI_HIP_STREAM_PROBE_FAIL, for the tests: both kernels on one stream, as if the two streams shared a queue)
		hipLaunchKernelGGL(chunk_ready_kernel, dim3(1), dim3(64), 0, getenv("RTGI_HIP_STREAM_PROBE_FAIL") ? p.run : p.aux, d_words, 1u);
		HIP_TRY(h, hipGetLastError());
		uint32_t seen = 0;
		HIP_TRY(h, hipMemcpyAsync(&seen, d_words + 1, sizeof(seen), hipMemcpyDeviceToHost, p.run));
		HIP_TRY(h, hipStreamSynchronize(p.run));
		HIP_TRY(h, hipStreamSynchronize(p.aux));
		*ok = seen != 0;
	}
	return RTGI_OK;
}

// A page-locked host batch in ONE traversal launch. The batch goes up in a few chunks of growing size; each chunk is
// keyed and sorted on its own as soon as it has arrived (on a stream of its own, beside the traversal, which leaves
// LDS and wave slots free for those kernels), and the launch, begun when the first chunk is ready, takes up the later
// ones as a word in GPU memory announces them: one tail for the whole batch instead of one per chunk, and the upload
// hidden behind the traversal. If the launch ever waits in vain for a chunk (the other kernels found no place to run)
// it gives up and says so; *fell_back then tells the caller to trace the batch the ordinary way.
// closest_hit stores its records straight into the (page-locked) destination; any_hit's bytes come down in one copy.
template <bool ANY>
int trace_host_streamed(rtgi_hip_tracer *h, const rtgi_ray *rays, uint32_t n, void *out, bool sort, bool *fell_back) {
	using namespace rtgi_dev;
	*fell_back = true;
	HostPipe &p = *h->pipe;
	constexpr uint32_t CTL_COUNTERS = MAX_CHUNKS * 8 * COUNTER_STRIDE, CTL_BOUNDS = CTL_COUNTERS, CTL_READY = CTL_BOUNDS + MAX_CHUNKS * 16,
	                   CTL_GAVE_UP = CTL_READY + 1, CTL_COARSE = CTL_READY + 16, COARSE_WORDS = 2 * COARSE_BINS + 16,  // per chunk: counts | cursors | done
	                   CTL_WORDS = CTL_COARSE + MAX_CHUNKS * COARSE_WORDS;
	if (!h->d_stream_ctl) HIP_TRY(h, hipMalloc((void **)&h->d_stream_ctl, CTL_WORDS * sizeof(uint32_t)));
	if (!h->stream_probed) {
		bool side_by_side = false;
		if (int rc = probe_side_by_side(h, h->d_stream_ctl, &side_by_side)) return rc;
		h->stream_probed = true;
		if (!side_by_side) {
			h->host_streamed = false;  // (the ordinary way, from now on)
			return RTGI_OK;
		}
	}
	void *d_out_host = nullptr;  // the destination as the GPU sees it
	if (!ANY && (hipHostGetDevicePointer(&d_out_host, out, 0) != hipSuccess || !d_out_host)) {
		(void)hipGetLastError();
		return RTGI_OK;  // (not mapped into the GPU's address space: the ordinary way)
	}
	// chunks: a short first one to get the launch going, the rest in growing parts (the upload outruns the traversal)
	uint32_t n_chunks = 0;
	uint32_t ends[MAX_CHUNKS];
	for (int c = 0; c < MAX_CHUNKS && h->stream_parts[c] > 0.0f; ++c) {
		const uint32_t end = h->stream_parts[c] >= 1.0f ? n : (uint32_t)((double)n * h->stream_parts[c]) & ~63u;
		if (end > (n_chunks ? ends[n_chunks - 1] : 0u)) ends[n_chunks++] = end;
		if (end == n) break;
	}
	if (n_chunks == 0 || ends[n_chunks - 1] != n) {
		if (n_chunks == MAX_CHUNKS) --n_chunks;
		ends[n_chunks++] = n;
	}
	if (n_chunks == 1) return RTGI_OK;  // (nothing to take up later: the ordinary way, *fell_back is set)
	Scratch sc;
	if (int rc = handle_scratch(h, n, sort, sc)) return rc;
	const size_t cap = sc.sort_cap;
	const bool coarse = sort && h->stream_coarse;
	if (sort && !coarse) {  // the sort's own scratch must hold for every chunk's size (the library picks its algorithm by size)
		size_t need = 0;
		for (uint32_t c = 0; c < n_chunks; ++c) {
			size_t temp = 0;
			hipcub::DoubleBuffer<uint32_t> keys(sc.sort, sc.sort + cap), index(sc.sort + 2 * cap, sc.sort + 3 * cap);
			HIP_TRY(h, sort_pairs(nullptr, temp, keys, index, (uint32_t)((ends[c] - (c ? ends[c - 1] : 0))), nullptr, true));
			need = std::max(need, temp);
		}
		if (need > h->sort_temp_bytes) {
			HIP_TRY(h, hipDeviceSynchronize());
			if (h->d_sort_temp) (void)hipFree(h->d_sort_temp);
			h->d_sort_temp = nullptr;
			h->sort_temp_bytes = 0;
			HIP_TRY(h, hipMalloc(&h->d_sort_temp, need + need / 4 + 256));
			h->sort_temp_bytes = need + need / 4 + 256;
			sc.sort_temp = h->d_sort_temp;
			sc.sort_temp_bytes = h->sort_temp_bytes;
		}
	}
	ScratchTurn turn{h, p.run};
	if (int rc = turn.begin()) return rc;
	HIP_TRY(h, hipMemsetAsync(h->d_stream_ctl, 0, CTL_WORDS * sizeof(uint32_t), p.run));
	HIP_TRY(h, hipEventRecord(p.begun, p.run));  // the scratch memory is this call's from here on
	const bool clocked = p.clock_begin != nullptr;
	if (clocked) (void)hipEventRecord(p.clock_begin, p.run);
	HIP_TRY(h, hipStreamWaitEvent(p.aux, p.begun, 0));
	TraceArgs a;
	memset(&a, 0, sizeof(a));
	rtgi_ray *d_rays = (rtgi_ray *)h->d_stage_rays;
	KeyArgs k;
	float diag2 = 0.0f;
	for (int c = 0; c < 3; ++c) {
		k.lo[c] = h->scene_lo[c], k.hi[c] = h->scene_hi[c], k.inv_extent[c] = h->scene_inv_extent[c];
		diag2 += (h->scene_hi[c] - h->scene_lo[c]) * (h->scene_hi[c] - h->scene_lo[c]);
	}
	k.inv_diagonal = diag2 > 0.0f ? 1.0f / std::sqrt(diag2) : 0.0f;
	for (uint32_t c = 0; c < n_chunks; ++c) {
		const uint32_t begin = c ? ends[c - 1] : 0, count = ends[c] - begin;
		HIP_TRY(h, hipMemcpyAsync(d_rays + begin, rays + begin, (size_t)count * sizeof(rtgi_ray), hipMemcpyHostToDevice, p.up));
		HIP_TRY(h, hipEventRecord(p.chunk_up[c], p.up));
		HIP_TRY(h, hipStreamWaitEvent(p.aux, p.chunk_up[c], 0));
		a.chunk_end[c] = ends[c];
		if (coarse) {
			// two kernels: bin of every ray and the bins' counts; then the rays bin after bin. The second one's last workgroup
			// announces the chunk.
			k.rays = (const float4 *)(d_rays + begin);
			k.keys = nullptr, k.index = nullptr;
			k.n = count;
			k.first = 0;
			k.index_base = begin;
			uint8_t *bins = reinterpret_cast<uint8_t *>(sc.sort) + begin;
			uint32_t *order = sc.sort + 2 * cap;
			uint32_t *ctl = h->d_stream_ctl + CTL_COARSE + c * COARSE_WORDS, *bounds = h->d_stream_ctl + CTL_BOUNDS + c * 16;
			const uint32_t tiles = (count + COARSE_TILE - 1) / COARSE_TILE;
			hipLaunchKernelGGL(ray_bins_kernel, dim3(tiles), dim3(COARSE_BLOCK), 0, p.aux, k, bins, ctl);
			hipLaunchKernelGGL(coarse_scatter_kernel, dim3(tiles), dim3(COARSE_BLOCK), 0, p.aux, (const uint8_t *)bins, count, begin, (const uint32_t *)ctl,
			                   ctl + COARSE_BINS, order + begin, bounds, ctl + 2 * COARSE_BINS, h->d_stream_ctl + CTL_READY, c + 1);
			a.chunk_order[c] = order;  // (indexed by the position in the whole batch)
			a.chunk_bounds[c] = bounds;
		} else if (sort) {
			// this chunk's part of the four sort arrays; which of each pair holds the result is known when the call returns
			hipcub::DoubleBuffer<uint32_t> keys(sc.sort + begin, sc.sort + cap + begin), index(sc.sort + 2 * cap + begin, sc.sort + 3 * cap + begin);
			k.rays = (const float4 *)(d_rays + begin);
			k.keys = keys.Current();
			k.index = index.Current();
			k.n = count;
			k.first = 0;
			// (keys and indices at [0, count) of the chunk's arrays, the indices counted from the batch's first ray)
			k.index_base = begin;
			hipLaunchKernelGGL(ray_keys_kernel, dim3((count + 255) / 256), dim3(256), 0, p.aux, k);
			size_t temp_size = sc.sort_temp_bytes;
			HIP_TRY(h, sort_pairs(sc.sort_temp, temp_size, keys, index, (uint32_t)(count), p.aux, true));
			uint32_t *bounds = h->d_stream_ctl + CTL_BOUNDS + c * 16;
			hipLaunchKernelGGL(segment_bounds_kernel, dim3(1), dim3(512), 0, p.aux, (const uint32_t *)keys.Current(), count, bounds, (uint32_t *)nullptr);
			a.chunk_order[c] = index.Current() - begin;  // (indexed by the position in the whole batch)
			a.chunk_bounds[c] = bounds;
		}
		if (!coarse) hipLaunchKernelGGL(chunk_ready_kernel, dim3(1), dim3(64), 0, p.aux, h->d_stream_ctl + CTL_READY, c + 1);
		HIP_TRY(h, hipGetLastError());
		if (clocked) (void)hipEventRecord(p.clock_ready[c], p.aux), (void)hipEventRecord(p.clock_up[c], p.up);
		if (c == 0) HIP_TRY(h, hipEventRecord(p.first_ready, p.aux));
	}
	HIP_TRY(h, hipEventRecord(p.aux_done, p.aux));
	// the launch: begun once the first chunk is ready, with less than a CU's LDS so that the sorts find room beside it
	a.nodes = h->d_nodes;
	a.tris = h->d_tris;
	a.rays = (const float4 *)d_rays;
	a.hits = ANY ? nullptr : (float4 *)d_out_host;
	a.occluded = ANY ? (uint8_t *)h->d_stage_out : nullptr;
	a.spill = sc.spill;
	a.ray_counter = h->d_stream_ctl;
	a.n_rays = n;
	a.n_segments = 8;
	a.node_bytes = h->node_bytes;
	a.tri_bytes = h->tri_bytes;
	a.root_ref = h->root_ref;
	a.refill_min = h->refill_min;
	a.leaf_min = h->leaf_min;
	a.drain_leaf_min = h->drain_leaf_min;
	a.implicit_limit = (int32_t)h->implicit_limit;
	a.boxes_ordered = h->boxes_ordered ? 1 : 0;
	a.stack_levels = (int32_t)h->stack_levels;
	a.n_chunks = n_chunks;
	a.chunks_ready = h->d_stream_ctl + CTL_READY;
	a.gave_up = h->d_stream_ctl + CTL_GAVE_UP;
	a.wait_ticks = h->stream_wait_ticks;
	uint32_t top = h->top_nodes;
	while (top > 0 && lds_bytes_for(h->stack_levels, top, h->implicit_limit) > h->stream_lds_bytes) top -= std::min<uint32_t>(top, 16);
	a.top_nodes = (int32_t)top;
	const uint32_t lds = (uint32_t)lds_bytes_for(h->stack_levels, top, h->implicit_limit);
	uint32_t blocks = h->grid_blocks;
	const bool deep = h->spill_levels > 0;
	if (deep) {
		const size_t most = sc.spill_words / ((size_t)h->spill_levels * BLOCK);
		if (most == 0) return fail(h, RTGI_ERR_INVALID, "tree too deep for the spill area");
		if (blocks > most) blocks = (uint32_t)most;
	}
	a.spill_stride = blocks * BLOCK;
	HIP_TRY(h, hipStreamWaitEvent(p.run, p.first_ready, 0));
	HIP_TRY(h, (launch_variant<ANY ? 1 : 0>(h->device, false, deep, blocks, lds, p.run, a)));
	if (clocked) (void)hipEventRecord(p.clock_end, p.run);
	HIP_TRY(h, hipStreamWaitEvent(p.run, p.aux_done, 0));  // (the scratch is free again when both streams are through)
	uint32_t gave_up = 0;
	if (ANY) HIP_TRY(h, hipMemcpyAsync(out, h->d_stage_out, n, hipMemcpyDeviceToHost, p.run));
	HIP_TRY(h, hipMemcpyAsync(&gave_up, h->d_stream_ctl + CTL_GAVE_UP, sizeof(gave_up), hipMemcpyDeviceToHost, p.run));
	if (int rc = turn.end()) return rc;
	HIP_TRY(h, hipStreamSynchronize(p.run));
	*fell_back = gave_up != 0;
	++h->streamed_calls;
	h->streamed_gave_up += gave_up != 0;
	// Two launches that waited in vain: on this GPU, in this process, the kernels beside the launch do not get to run (e.g.
	// their stream shares a hardware queue with the launch's): no more one-launch batches on this handle.
	if (h->streamed_gave_up >= 2) h->host_streamed = false;
	if (clocked) {
		fprintf(stderr, "streamed:");
		for (uint32_t c = 0; c < n_chunks; ++c) {
			float up = 0, ready = 0;
			(void)hipEventElapsedTime(&up, p.clock_begin, p.clock_up[c]), (void)hipEventElapsedTime(&ready, p.clock_begin, p.clock_ready[c]);
			fprintf(stderr, " chunk %u (to ray %u) up %.3f ready %.3f |", c, ends[c], up, ready);
		}
		float done = 0;
		(void)hipEventElapsedTime(&done, p.clock_begin, p.clock_end);
		fprintf(stderr, " kernel done %.3f ms%s\n", done, gave_up ? " GAVE UP" : "");
	}
	if (gave_up && getenv("RTGI_HIP_DEBUG_SYNC")) fprintf(stderr, "[rtgi_hip] streamed host batch: the launch waited in vain for a chunk, tracing the batch in separate launches\n");
	return RTGI_OK;
}

template <bool ANY>
int trace_host(rtgi_hip_tracer *h, const rtgi_ray *rays, uint32_t n, void *out) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: traversal needs a GPU");
	if (!h->built) return fail(h, RTGI_ERR_NOT_BUILT, "no acceleration structure: call build first");
	if (n == 0) return RTGI_OK;
	if (!rays || !out) return fail(h, RTGI_ERR_INVALID, "null ray or output buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	const size_t out_bytes = (size_t)n * (ANY ? 1 : sizeof(rtgi_hit));
	if (n <= SMALL_BATCH) {
		// a handful of rays: two staged copies would cost more than the traversal; the kernel works on mapped host memory
		if (!h->pinned) {
			HIP_TRY(h, hipHostMalloc(&h->pinned, SMALL_BATCH * (sizeof(rtgi_ray) + sizeof(rtgi_hit)), hipHostMallocMapped));
			HIP_TRY(h, hipHostGetDevicePointer(&h->d_pinned, h->pinned, 0));
		}
		memcpy(h->pinned, rays, (size_t)n * sizeof(rtgi_ray));
		char *out_host = (char *)h->pinned + SMALL_BATCH * sizeof(rtgi_ray), *out_dev = (char *)h->d_pinned + SMALL_BATCH * sizeof(rtgi_ray);
		int rc = launch<ANY>(h, (const rtgi_ray *)h->d_pinned, n, out_dev, nullptr, nullptr);
		if (rc) return rc;
		HIP_TRY(h, hipStreamSynchronize(nullptr));
		memcpy(out, out_host, out_bytes);
		return RTGI_OK;
	}
	int rc = ensure_stage(h, (size_t)n * sizeof(rtgi_ray), out_bytes);
	if (rc) return rc;
	rc = ensure_pipe(h);
	if (rc) return rc;
	HostPipe &p = *h->pipe;
	const bool rays_locked = is_page_locked(rays), out_locked = is_page_locked(out);
	// (the kernel stores hit records as 16-byte words: a destination that is not aligned to them gets its results by a copy)
	const bool out_aligned = ANY || (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
	if (h->host_streamed && rays_locked && out_locked && out_aligned && n >= h->stream_min && n < (1u << 31)) {
		const bool sort = (h->reorder == 1 || (h->reorder == 2 && h->probe_pops + 2.0f * h->probe_tris > h->reorder_min_pops)) && n >= 4 * h->reorder_min;
		HIP_TRY(h, hipStreamSynchronize(nullptr));  // (work that earlier calls left on the default stream comes first)
		bool fell_back = true;
		rc = trace_host_streamed<ANY>(h, rays, n, out, sort, &fell_back);
		if (rc != RTGI_OK) {  // nothing of this call stays in flight
			(void)hipStreamSynchronize(h->pipe->up), (void)hipStreamSynchronize(h->pipe->aux), (void)hipStreamSynchronize(h->pipe->run);
			return rc;
		}
		if (!fell_back) return RTGI_OK;
		(void)hipStreamSynchronize(h->pipe->up), (void)hipStreamSynchronize(h->pipe->aux);
	}
	// Chunks: every chunk's launch ends in a tail of its own, so there are few. From page-locked memory a short first
	// one (a good quarter of the batch, at most host_lead rays) gets the GPU going early and the rest follows in
	// parts of up to 5/3 host_chunk rays, each travelling up while its predecessor is traced; from ordinary memory,
	// where the staging copies of the host threads set the pace, equal parts of at most host_chunk rays.
	std::vector<size_t> bounds = {0};
	{
		const size_t limit = rays_locked ? (size_t)h->host_chunk * 5 / 3 : h->host_chunk;
		size_t lead = rays_locked && n >= 2 * (size_t)h->host_lead_min ? std::min<size_t>((size_t)(n * 0.28), h->host_lead) & ~(size_t)63 : 0;
		if (lead) bounds.push_back(lead);
		const size_t rest = n - lead, parts = (rest + limit - 1) / limit;
		const size_t part = ((rest + parts - 1) / parts + 63u) & ~(size_t)63;
		for (size_t at = lead + part; at < n; at += part) bounds.push_back(at);
		bounds.push_back(n);
	}
	const uint32_t n_chunks = (uint32_t)bounds.size() - 1;
	uint32_t chunk = 0;  // the largest
	for (uint32_t k = 0; k < n_chunks; ++k) chunk = std::max<uint32_t>(chunk, (uint32_t)(bounds[k + 1] - bounds[k]));
	const size_t out_item = ANY ? 1 : sizeof(rtgi_hit);
	if (!rays_locked || !out_locked) {
		rc = ensure_slots(h, chunk);
		if (rc) return rc;
	}
	auto first_of = [&](uint32_t k) { return bounds[k]; };
	auto count_of = [&](uint32_t k) { return (uint32_t)(bounds[k + 1] - bounds[k]); };
	// results of chunk k, once they have come down into its slot, go to the caller's buffer
	auto hand_over = [&](uint32_t k) -> int {
		HIP_TRY(h, hipEventSynchronize(p.went_down[k % HostPipe::SLOTS]));
		if (!out_locked) p.pool.copy((char *)out + first_of(k) * out_item, p.slot_out[k % HostPipe::SLOTS], count_of(k) * out_item);
		return RTGI_OK;
	};
	// work that earlier calls left on the default stream (the usual place for this handle's launches) comes first
	HIP_TRY(h, hipStreamSynchronize(nullptr));
	// Whatever way this call ends, nothing of it is left in flight: copies out of `rays`, kernels that store into
	// `out` and the slots' events are all reused or released by the caller right after the return.
	struct Drain {
		HostPipe &p;
		~Drain() {
			(void)hipStreamSynchronize(p.up);
			(void)hipStreamSynchronize(p.run);
			(void)hipStreamSynchronize(p.down);
		}
	} drain{p};
	for (uint32_t k = 0; k < n_chunks; ++k) {
		const int slot = (int)(k % HostPipe::SLOTS);
		const size_t first = first_of(k);
		const uint32_t count = count_of(k);
		// the slot's previous user was chunk k - SLOTS: its rays have gone up and its results have been handed over
		if (k >= (uint32_t)HostPipe::SLOTS) {
			if (!rays_locked) HIP_TRY(h, hipEventSynchronize(p.came_up[slot]));
			if ((rc = hand_over(k - HostPipe::SLOTS)) != RTGI_OK) return rc;
		}
		const rtgi_ray *src = rays + first;
		if (!rays_locked) {
			p.pool.copy(p.slot_rays[slot], src, (size_t)count * sizeof(rtgi_ray));
			src = (const rtgi_ray *)p.slot_rays[slot];
		}
		rtgi_ray *d_rays = (rtgi_ray *)h->d_stage_rays + first;
		char *d_out = (char *)h->d_stage_out + first * out_item;
		// Hit records need no copy down: the kernel stores them (16 bytes each, once per ray) straight into the
		// page-locked destination, which the stores reach over PCIe while the traversal goes on. (any_hit's single
		// bytes would make as many PCIe writes for a sixteenth of the data: they come down as one copy.)
		bool direct = !ANY && h->host_direct && (out_aligned || !out_locked);
		if (direct) {
			void *host = out_locked ? (void *)((char *)out + first * out_item) : p.slot_out[slot], *dev = nullptr;
			if (hipHostGetDevicePointer(&dev, host, 0) == hipSuccess && dev)
				d_out = (char *)dev;
			else {
				(void)hipGetLastError();
				direct = false;
			}
		}
		HIP_TRY(h, hipMemcpyAsync(d_rays, src, (size_t)count * sizeof(rtgi_ray), hipMemcpyHostToDevice, p.up));
		HIP_TRY(h, hipEventRecord(p.came_up[slot], p.up));
		HIP_TRY(h, hipStreamWaitEvent(p.run, p.came_up[slot], 0));
		rc = launch<ANY>(h, d_rays, count, d_out, nullptr, p.run);
		if (rc) return rc;
		HIP_TRY(h, hipEventRecord(p.traced[slot], p.run));
		HIP_TRY(h, hipStreamWaitEvent(p.down, p.traced[slot], 0));
		void *dst = out_locked ? (void *)((char *)out + first * out_item) : p.slot_out[slot];
		if (!direct) HIP_TRY(h, hipMemcpyAsync(dst, d_out, (size_t)count * out_item, hipMemcpyDeviceToHost, p.down));
		HIP_TRY(h, hipEventRecord(p.went_down[slot], p.down));
	}
	for (uint32_t k = n_chunks > (uint32_t)HostPipe::SLOTS ? n_chunks - HostPipe::SLOTS : 0; k < n_chunks; ++k)
		if ((rc = hand_over(k)) != RTGI_OK) return rc;
	return RTGI_OK;
}

// Traces the probe rays of upload() in the order they come and stores the mean number of inner nodes popped.
int probe_scene(rtgi_hip_tracer *h) {
	if (h->device < 0 || h->probe_rays.empty()) return RTGI_OK;
	DeviceScope device_scope(h->device);  // the staging buffers below must live on the handle's GPU
	HIP_TRY(h, device_scope.status);
	const uint32_t n = (uint32_t)h->probe_rays.size();
	int rc = ensure_stage(h, (size_t)n * sizeof(rtgi_ray), (size_t)n * sizeof(rtgi_hit));
	if (rc) return rc;
	uint32_t *d_counts = nullptr;
	HIP_TRY(h, hipMalloc((void **)&d_counts, (size_t)n * 2 * sizeof(uint32_t)));
	std::vector<uint32_t> counts((size_t)n * 2);
	hipError_t e = hipMemcpy(h->d_stage_rays, h->probe_rays.data(), (size_t)n * sizeof(rtgi_ray), hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		const int reorder = h->reorder;
		h->reorder = 0;
		rc = launch<false>(h, (const rtgi_ray *)h->d_stage_rays, n, h->d_stage_out, d_counts, nullptr);
		h->reorder = reorder;
		if (rc == RTGI_OK) e = hipMemcpy(counts.data(), d_counts, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
	}
	(void)hipFree(d_counts);
	h->probe_rays.clear();
	h->probe_rays.shrink_to_fit();
	if (rc) return rc;
	HIP_TRY(h, e);
	double pops = 0.0, tris = 0.0;
	for (uint32_t i = 0; i < n; ++i) pops += counts[2 * (size_t)i], tris += counts[2 * (size_t)i + 1];
	h->probe_pops = (float)(pops / n);
	h->probe_tris = (float)(tris / n);
	// Idle lanes before a wave draws new rays: a refill is an atomic, a fetch of the order and a fetch of the rays, one after the
	// other. Where rays are short (few nodes each) refills come often and larger ones pay: 40 idle lanes against 24 make camera
	// and shadow rays of the terrain scene 8 / 13 % faster, of the sponza-class scene 3 / 0 %, of the stadium scene 5 / 5 %, bounce
	// rays the same; the soup's long rays (144 pops per probe ray) keep 24 (16 ... 32 measured: 24-28 best).
	if (!h->refill_min_given) h->refill_min = h->probe_pops < 100.0f ? 40 : 24;
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
	read_tunables(h);
	if (device >= 0 && device < MAX_DEVICES) {
		// set-up for rtgi_hip_closest_hit_device_noalloc: where this GPU keeps its copy of the fixed scratch array
		std::lock_guard<std::mutex> lock(g_scratch_guard);
		if (!g_scratch_ptr[device]) {
			DeviceScope device_scope(device);
			// Every kernel of the library is looked up once here, by one thread: the runtime loads a kernel when it is
			// first used, and first launches of one kernel from several host threads at the same moment (the frame
			// level drives each GPU from a thread of its own) have been seen to fail with "invalid argument".
			if (device_scope.status == hipSuccess) {
				using namespace rtgi_dev;
				const void *kernels[] = {
				    (const void *)&trace_kernel<0, false, false>, (const void *)&trace_kernel<0, false, true>, (const void *)&trace_kernel<0, true, false>,
				    (const void *)&trace_kernel<0, true, true>, (const void *)&trace_kernel<1, false, false>, (const void *)&trace_kernel<1, false, true>,
				    (const void *)&trace_kernel<1, true, false>, (const void *)&trace_kernel<1, true, true>, (const void *)&trace_kernel<2, false, false>,
				    (const void *)&trace_kernel<2, false, true>, (const void *)&ray_keys_kernel, (const void *)&segment_bounds_kernel,
				    (const void *)&chunk_ready_kernel, (const void *)&await_word_kernel, (const void *)&ray_bins_kernel, (const void *)&coarse_scatter_kernel,
				    (const void *)&camera_rays_kernel, (const void *)&shadow_rays_kernel, (const void *)&pixel_offsets_kernel,
				    (const void *)&bounce_count_kernel, (const void *)&scan_kernel, (const void *)&bounce_write_kernel,
				    (const void *)&shade_primary_kernel, (const void *)&shade_local_kernel, (const void *)&shade_local_brdf_kernel,
				    (const void *)&framebuffer_add_kernel, (const void *)&path_begin_kernel, (const void *)&path_add_kernel,
				    (const void *)&path_extend_kernel, (const void *)&framebuffer_scatter_kernel, (const void *)&framebuffer_sums_kernel,
				    (const void *)&framebuffer_accumulate_kernel, (const void *)&framebuffer_rgb8_kernel};
				for (const void *k : kernels) {
					hipFuncAttributes attr;
					if (hipFuncGetAttributes(&attr, k) != hipSuccess) (void)hipGetLastError();  // (a launch will report it)
				}
			}
			void *ptr = nullptr;
			hipError_t e = device_scope.status;
			if (e == hipSuccess) e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
			if (e != hipSuccess || !ptr) {
				g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
				(void)hipGetLastError();
				delete h;
				return nullptr;
			}
			g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
		}
	}
	return h;
}

void rtgi_hip_destroy(rtgi_hip_tracer *h) {
	if (!h) return;
	DeviceScope device_scope(h->device);
	free_device_tree(h);
	if (h->d_counter) (void)hipFree(h->d_counter);
	if (h->d_totals) (void)hipFree(h->d_totals);
	if (h->d_stage_rays) (void)hipFree(h->d_stage_rays);
	if (h->d_stage_out) (void)hipFree(h->d_stage_out);
	if (h->d_block_counts) (void)hipFree(h->d_block_counts);
	if (h->d_stream_ctl) (void)hipFree(h->d_stream_ctl);
	if (h->scratch_free) (void)hipEventDestroy(h->scratch_free);
	if (h->pinned) (void)hipHostFree(h->pinned);
	if (HostPipe *p = h->pipe) {
		for (int i = 0; i < HostPipe::SLOTS; ++i) {
			if (p->slot_rays[i]) (void)hipHostFree(p->slot_rays[i]);
			if (p->slot_out[i]) (void)hipHostFree(p->slot_out[i]);
			if (p->came_up[i]) (void)hipEventDestroy(p->came_up[i]);
			if (p->traced[i]) (void)hipEventDestroy(p->traced[i]);
			if (p->went_down[i]) (void)hipEventDestroy(p->went_down[i]);
		}
		for (int i = 0; i < rtgi_dev::MAX_CHUNKS; ++i)
			if (p->chunk_up[i]) (void)hipEventDestroy(p->chunk_up[i]);
		if (p->begun) (void)hipEventDestroy(p->begun);
		for (hipEvent_t e : {p->first_ready, p->aux_done, p->clock_begin, p->clock_end})
			if (e) (void)hipEventDestroy(e);
		for (int i = 0; i < rtgi_dev::MAX_CHUNKS; ++i) {
			if (p->clock_up[i]) (void)hipEventDestroy(p->clock_up[i]);
			if (p->clock_ready[i]) (void)hipEventDestroy(p->clock_ready[i]);
		}
		if (p->aux) (void)hipStreamDestroy(p->aux);
		if (p->up) (void)hipStreamDestroy(p->up);
		if (p->run) (void)hipStreamDestroy(p->run);
		if (p->down) (void)hipStreamDestroy(p->down);
		delete p;
	}
	if (h->d_sort) (void)hipFree(h->d_sort);
	if (h->d_sort_temp) (void)hipFree(h->d_sort_temp);
	for (auto &e : h->events) {
		(void)hipEventDestroy(e.first);
		(void)hipEventDestroy(e.second);
	}
	delete h;
}

int rtgi_hip_interprete(rtgi_hip_tracer *h, const char *command, const char *args) {
	if (!h || !command) return RTGI_ERR_INVALID;
	std::istringstream in(args ? args : "");
	std::string value;
	in >> value;
	if (std::string(command) == "hip") {  // backend-specific options, through the same hook as "bvh ..."
		if (value == "reorder") {
			std::string onoff;
			in >> onoff;
			if (onoff != "on" && onoff != "off" && onoff != "auto")
				return fail(h, RTGI_ERR_SYNTAX, "Syntax error, \"hip reorder\" requires auto, on or off");
			h->reorder = onoff == "off" ? 0 : onoff == "on" ? 1 : 2;
			return 1;
		}
		return fail(h, RTGI_ERR_SYNTAX, "Unknown hip subcommand " + value);
	}
	if (std::string(command) != "bvh") return 0;
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
		using clock = std::chrono::steady_clock;
		auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		const clock::time_point t0 = clock::now();
		if (!rtgi::build_tree(h->cfg, scene, tree, err)) return fail(h, RTGI_ERR_INVALID, err);
		h->tree = std::move(tree);
		const clock::time_point t1 = clock::now();
		int rc_upload = upload(h, scene);
		const clock::time_point t2 = clock::now();
		const int rc = rc_upload ? rc_upload : probe_scene(h);
		if (getenv("RTGI_HIP_BUILD_TIMES"))  // where a commit's time goes
			fprintf(stderr, "rtgi_hip_build: tree %.1f ms, flatten and upload %.1f ms, probe rays %.1f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, clock::now()));
		return rc;
	} catch (const std::bad_alloc &) {
		return fail(h, RTGI_ERR_NOMEM, "out of host memory while building the BVH");
	} catch (const std::exception &e) {
		return fail(h, RTGI_ERR_INVALID, std::string("building the BVH failed: ") + e.what());
	} catch (...) {
		return fail(h, RTGI_ERR_INVALID, "building the BVH failed");
	}
}

int rtgi_hip_build_shared(rtgi_hip_tracer *const *handles, uint32_t n_handles, const void *vertices, size_t vertex_stride, uint32_t n_vertices,
                          const void *triangles, size_t tri_stride, uint32_t n_triangles) {
	if (!handles || n_handles == 0 || !handles[0]) return RTGI_ERR_INVALID;
	rtgi_hip_tracer *lead = handles[0];
	for (uint32_t i = 1; i < n_handles; ++i) {
		rtgi_hip_tracer *o = handles[i];
		if (!o) return fail(lead, RTGI_ERR_INVALID, "build_shared: null handle");
		if (o->cfg.layout != lead->cfg.layout || o->cfg.esc != lead->cfg.esc || o->cfg.split != lead->cfg.split ||
		    o->cfg.planes != lead->cfg.planes || o->cfg.max_tris != lead->cfg.max_tris)
			return fail(lead, RTGI_ERR_INVALID, "build_shared: the handles were not given the same layout and bvh settings");
	}
	rtgi::SceneView scene;
	scene.vertices = (const unsigned char *)vertices;
	scene.vertex_stride = vertex_stride;
	scene.n_vertices = n_vertices;
	scene.triangles = (const unsigned char *)triangles;
	scene.tri_stride = tri_stride;
	scene.n_triangles = n_triangles;
	try {
		std::string err;
		rtgi::HostTree tree;
		if (!rtgi::build_tree(lead->cfg, scene, tree, err)) return fail(lead, RTGI_ERR_INVALID, err);
		rtgi::FlatTree flat;
		rtgi::flatten_tree(tree, scene, lead->cfg.layout, flat);
		// every handle keeps the host tree (read-back, `bvh statistics`, shading tables) and uploads the one flat form to
		// its own GPU, each on a thread of its own: the uploads of different GPUs do not wait for each other
		std::vector<int> rc(n_handles, RTGI_OK);
		auto serve = [&](uint32_t i) {
			rtgi_hip_tracer *h = handles[i];
			try {
				if (i + 1 == n_handles)
					h->tree = std::move(tree);  // (the last one to be served takes the original; see the order below)
				else
					h->tree = tree;
				rc[i] = upload_flat(h, flat, scene);
				if (rc[i] == RTGI_OK) rc[i] = probe_scene(h);
			} catch (const std::bad_alloc &) {
				rc[i] = fail(h, RTGI_ERR_NOMEM, "out of host memory while taking over the BVH");
			}
		};
		std::vector<std::thread> threads;
		for (uint32_t i = 0; i + 1 < n_handles; ++i) {
			try {
				threads.emplace_back(serve, i);
			} catch (const std::system_error &) {
				serve(i);  // no thread to be had: one after the other
			}
		}
		for (std::thread &t : threads) t.join();
		serve(n_handles - 1);  // after the copies have been taken
		for (uint32_t i = 0; i < n_handles; ++i)
			if (rc[i] < 0) return i == 0 ? rc[i] : fail(lead, rc[i], "GPU " + std::to_string(handles[i]->device) + ": " + handles[i]->error);
		return RTGI_OK;
	} catch (const std::bad_alloc &) {
		return fail(lead, RTGI_ERR_NOMEM, "out of host memory while building the BVH");
	} catch (const std::exception &e) {
		return fail(lead, RTGI_ERR_INVALID, std::string("building the BVH failed: ") + e.what());
	} catch (...) {
		return fail(lead, RTGI_ERR_INVALID, "building the BVH failed");
	}
}

int rtgi_hip_device_of(const rtgi_hip_tracer *h) { return h ? h->device : -1; }

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
		int rc_upload = upload(h, scene);
		return rc_upload ? rc_upload : probe_scene(h);
	} catch (const std::bad_alloc &) {
		return fail(h, RTGI_ERR_NOMEM, "out of host memory while taking over the BVH");
	} catch (const std::exception &e) {
		return fail(h, RTGI_ERR_INVALID, std::string("taking over the BVH failed: ") + e.what());
	} catch (...) {
		return fail(h, RTGI_ERR_INVALID, "taking over the BVH failed");
	}
}

uint32_t rtgi_hip_num_nodes(const rtgi_hip_tracer *h) { return h ? (uint32_t)h->tree.nodes.size() : 0; }
uint32_t rtgi_hip_num_index(const rtgi_hip_tracer *h) { return h ? (uint32_t)h->tree.order.size() : 0; }
uint32_t rtgi_hip_root(const rtgi_hip_tracer *h) { return h ? h->tree.root : 0; }
uint32_t rtgi_hip_depth(const rtgi_hip_tracer *h) { return h ? h->tree.depth : 0; }
float rtgi_hip_incoherent_ray_cost(const rtgi_hip_tracer *h) { return h ? h->probe_pops : 0.0f; }
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

int rtgi_hip_closest_hit_device_noalloc(rtgi_hip_tracer *h, const rtgi_ray *d_rays, uint32_t n, rtgi_hit *d_hits, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: traversal needs a GPU");
	if (n > NOALLOC_MAX_RAYS)
		return fail(h, RTGI_ERR_INVALID, "closest_hit_device_noalloc takes at most " + std::to_string(NOALLOC_MAX_RAYS) + " rays per call");
	if (n != 0 && (!d_rays || !d_hits)) return fail(h, RTGI_ERR_INVALID, "null ray or output buffer");
	unsigned char *base = h->device < MAX_DEVICES ? g_scratch_ptr[h->device] : nullptr;  // set by rtgi_hip_create
	if (!base) return fail(h, RTGI_ERR_DEVICE, "no fixed scratch memory on this GPU");
	Scratch sc;
	sc.counter = reinterpret_cast<uint32_t *>(base);
	sc.sort = reinterpret_cast<uint32_t *>(base + NOALLOC_COUNTER_BYTES);
	sc.sort_cap = NOALLOC_MAX_RAYS;
	sc.sort_temp = base + NOALLOC_COUNTER_BYTES + NOALLOC_SORT_BYTES;
	sc.sort_temp_bytes = NOALLOC_TEMP_BYTES;
	sc.spill = reinterpret_cast<int32_t *>(base + NOALLOC_COUNTER_BYTES + NOALLOC_SORT_BYTES + NOALLOC_TEMP_BYTES);
	sc.spill_words = NOALLOC_SPILL_BYTES / sizeof(int32_t);
	const BatchView one{d_rays, n, false, d_hits};
	return launch_batches(h, &one, 1, nullptr, (hipStream_t)stream, &sc);
}

void *rtgi_hip_host_alloc(size_t bytes) {
	void *p = nullptr;
	if (bytes == 0) return nullptr;
	const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
	if (e != hipSuccess) {
		(void)hipGetLastError();  // the caller falls back to ordinary memory: not an error of a later call
		g_create_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
		return nullptr;
	}
	return p;
}

void rtgi_hip_host_free(void *p) {
	if (p) (void)hipHostFree(p);
}

int rtgi_hip_host_register(void *p, size_t bytes) {
	if (!p || bytes == 0) return RTGI_ERR_INVALID;
	const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
	if (e == hipSuccess) return RTGI_OK;
	(void)hipGetLastError();  // optional service: its failure must not surface in a later call
	g_create_error = std::string("hipHostRegister: ") + hipGetErrorString(e);
	return RTGI_ERR_DEVICE;
}

int rtgi_hip_host_unregister(void *p) {
	if (!p) return RTGI_ERR_INVALID;
	const hipError_t e = hipHostUnregister(p);
	if (e == hipSuccess) return RTGI_OK;
	(void)hipGetLastError();
	g_create_error = std::string("hipHostUnregister: ") + hipGetErrorString(e);
	return RTGI_ERR_DEVICE;
}

int rtgi_hip_trace_batches_device(rtgi_hip_tracer *h, const rtgi_batch *batches, uint32_t n_batches, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (n_batches == 0) return RTGI_OK;
	if (!batches) return fail(h, RTGI_ERR_INVALID, "null batch list");
	if (n_batches > (uint32_t)rtgi_dev::MAX_BATCHES) return fail(h, RTGI_ERR_INVALID, "at most 32 batches per launch");
	BatchView views[rtgi_dev::MAX_BATCHES];
	for (uint32_t b = 0; b < n_batches; ++b) {
		if (batches[b].kind != RTGI_BATCH_CLOSEST_HIT && batches[b].kind != RTGI_BATCH_ANY_HIT)
			return fail(h, RTGI_ERR_INVALID, "batch kind must be RTGI_BATCH_CLOSEST_HIT or RTGI_BATCH_ANY_HIT");
		views[b] = BatchView{batches[b].d_rays, batches[b].n, batches[b].kind == RTGI_BATCH_ANY_HIT, batches[b].d_out};
	}
	return launch_batches(h, views, n_batches, nullptr, (hipStream_t)stream);
}

namespace {
int camera_rays(rtgi_hip_tracer *h, const rtgi_camera *cam, const uint32_t *d_pixels, uint32_t n_pixels, const float *d_offsets,
                rtgi_ray *d_rays, void *stream);
}
int rtgi_hip_camera_rays_device(rtgi_hip_tracer *h, const rtgi_camera *cam, const float *d_offsets, rtgi_ray *d_rays,
                                void *stream) {
	return camera_rays(h, cam, nullptr, 0, d_offsets, d_rays, stream);
}
int rtgi_hip_camera_rays_pixels_device(rtgi_hip_tracer *h, const rtgi_camera *cam, const uint32_t *d_pixels, uint32_t n,
                                       const float *d_offsets, rtgi_ray *d_rays, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (n == 0) return RTGI_OK;
	if (!d_pixels) return fail(h, RTGI_ERR_INVALID, "camera_rays_pixels: null pixel list");
	return camera_rays(h, cam, d_pixels, n, d_offsets, d_rays, stream);
}
namespace {
int camera_rays(rtgi_hip_tracer *h, const rtgi_camera *cam, const uint32_t *d_pixels, uint32_t n_pixels, const float *d_offsets,
                rtgi_ray *d_rays, void *stream) {
	if (!h || !cam) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (cam->w <= 0 || cam->h <= 0 || !d_rays || (reinterpret_cast<uintptr_t>(d_rays) & 15u))
		return fail(h, RTGI_ERR_INVALID, "camera_rays: bad resolution or ray buffer (must be 16-byte aligned)");
	if ((uint64_t)cam->w * (uint64_t)cam->h > 0x7fffffffull) return fail(h, RTGI_ERR_INVALID, "camera_rays: resolution too large");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	rtgi_dev::CamArgs a;
	const float *d = cam->dir, *u = cam->up;
	// cross(dir, up) and cross(U, dir) with glm's component formulas (libgi/camera.cpp:13-14)
	const float U[3] = {d[1] * u[2] - u[1] * d[2], d[2] * u[0] - u[2] * d[0], d[0] * u[1] - u[0] * d[1]};
	const float V[3] = {U[1] * d[2] - d[1] * U[2], U[2] * d[0] - d[2] * U[0], U[0] * d[1] - d[0] * U[1]};
	for (int k = 0; k < 3; ++k) a.dir[k] = d[k], a.pos[k] = cam->pos[k], a.U[k] = U[k], a.V[k] = V[k];
	const float aspect = float(cam->w) / cam->h;  // camera::update_frustum, libgi/camera.h:18-24
	a.near_h = tanf(float(M_PI) * cam->fovy * 0.5f / 180.0f);
	a.near_w = aspect * a.near_h;
	a.w = cam->w, a.h = cam->h;
	a.offsets = (const float2 *)d_offsets;
	a.rays = (float4 *)d_rays;
	const uint32_t n = d_pixels ? n_pixels : (uint32_t)cam->w * (uint32_t)cam->h;
	a.pixels = d_pixels;  // (the caller vouches for pixels[i] < w * h: they only enter arithmetic, no address)
	a.n = n;
	hipLaunchKernelGGL(rtgi_dev::camera_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
	HIP_TRY(h, hipGetLastError());
	// (camera rays written right behind the last ones - the next sample of the same pixels - make one longer batch in image order)
	if (h->camera_rays_at && d_rays == static_cast<const rtgi_ray *>(h->camera_rays_at) + h->camera_rays_n && h->camera_rays_n <= 0x7fffffffu - n)
		h->camera_rays_n += n;
	else {
		h->camera_rays_at = d_rays;
		h->camera_rays_n = n;
	}
	return RTGI_OK;
}
}  // namespace

namespace {
int pixel_offsets(rtgi_hip_tracer *h, const uint32_t *d_pixels, uint32_t n, uint32_t sample, uint32_t seed, float *d_offsets, void *stream);
}
int rtgi_hip_pixel_offsets_device(rtgi_hip_tracer *h, uint32_t n, uint32_t sample, uint32_t seed, float *d_offsets, void *stream) {
	return pixel_offsets(h, nullptr, n, sample, seed, d_offsets, stream);
}
int rtgi_hip_pixel_offsets_pixels_device(rtgi_hip_tracer *h, const uint32_t *d_pixels, uint32_t n, uint32_t sample, uint32_t seed,
                                         float *d_offsets, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (n != 0 && !d_pixels) return fail(h, RTGI_ERR_INVALID, "pixel_offsets_pixels: null pixel list");
	return pixel_offsets(h, d_pixels, n, sample, seed, d_offsets, stream);
}
namespace {
int pixel_offsets(rtgi_hip_tracer *h, const uint32_t *d_pixels, uint32_t n, uint32_t sample, uint32_t seed, float *d_offsets, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (n == 0) return RTGI_OK;
	if (!d_offsets || (reinterpret_cast<uintptr_t>(d_offsets) & 7u)) return fail(h, RTGI_ERR_INVALID, "pixel_offsets: null or misaligned buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	hipLaunchKernelGGL(rtgi_dev::pixel_offsets_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float2 *)d_offsets,
	                   d_pixels, n, sample, seed);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}
}  // namespace

int rtgi_hip_path_begin_device(rtgi_hip_tracer *h, uint32_t n, float *d_throughput, uint32_t *d_slot, float *d_radiance, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (n == 0) return RTGI_OK;
	if (!d_throughput || !d_slot || !d_radiance || ((reinterpret_cast<uintptr_t>(d_throughput) | reinterpret_cast<uintptr_t>(d_radiance)) & 15u))
		return fail(h, RTGI_ERR_INVALID, "path_begin: null or misaligned buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	hipLaunchKernelGGL(rtgi_dev::path_begin_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float4 *)d_throughput, d_slot,
	                   (float4 *)d_radiance, n);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}

int rtgi_hip_path_add_device(rtgi_hip_tracer *h, const float *d_direct, const float *d_throughput, const uint32_t *d_slot, uint32_t m,
                             float *d_radiance, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (m == 0) return RTGI_OK;
	if (!d_direct || !d_throughput || !d_slot || !d_radiance ||
	    ((reinterpret_cast<uintptr_t>(d_direct) | reinterpret_cast<uintptr_t>(d_throughput) | reinterpret_cast<uintptr_t>(d_radiance)) & 15u))
		return fail(h, RTGI_ERR_INVALID, "path_add: null or misaligned buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	hipLaunchKernelGGL(rtgi_dev::path_add_kernel, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_direct,
	                   (const float4 *)d_throughput, d_slot, m, (float4 *)d_radiance);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}

int rtgi_hip_path_extend_device(rtgi_hip_tracer *h, const float *d_albedo, const uint32_t *d_parent, uint32_t m,
                                const float *d_throughput_in, const uint32_t *d_slot_in, float *d_throughput_out, uint32_t *d_slot_out,
                                void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	const uint32_t m_max = m;
	if (m_max == 0) return RTGI_OK;
	if (!d_albedo || !d_parent || !d_throughput_in || !d_slot_in || !d_throughput_out || !d_slot_out ||
	    ((reinterpret_cast<uintptr_t>(d_albedo) | reinterpret_cast<uintptr_t>(d_throughput_in) | reinterpret_cast<uintptr_t>(d_throughput_out)) & 15u))
		return fail(h, RTGI_ERR_INVALID, "path_extend: null or misaligned buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	hipLaunchKernelGGL(rtgi_dev::path_extend_kernel, dim3((m_max + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_albedo,
	                   d_parent, m_max, (const float4 *)d_throughput_in, d_slot_in, (float4 *)d_throughput_out, d_slot_out);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}

int rtgi_hip_framebuffer_scatter_device(rtgi_hip_tracer *h, const float *d_part, const uint32_t *d_pixels, uint32_t n, float *d_full,
                                        void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (n == 0) return RTGI_OK;
	if (!d_part || !d_pixels || !d_full || ((reinterpret_cast<uintptr_t>(d_part) | reinterpret_cast<uintptr_t>(d_full)) & 15u))
		return fail(h, RTGI_ERR_INVALID, "framebuffer_scatter: null or misaligned buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	hipLaunchKernelGGL(rtgi_dev::framebuffer_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
	                   (const float4 *)d_part, d_pixels, n, (float4 *)d_full);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}

int rtgi_hip_framebuffer_sums_device(rtgi_hip_tracer *h, float *d_framebuffer, uint32_t n, int to_sums, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (n == 0) return RTGI_OK;
	if (!d_framebuffer || (reinterpret_cast<uintptr_t>(d_framebuffer) & 15u)) return fail(h, RTGI_ERR_INVALID, "framebuffer_sums: null or misaligned buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	hipLaunchKernelGGL(rtgi_dev::framebuffer_sums_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float4 *)d_framebuffer, n,
	                   to_sums);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}

int rtgi_hip_framebuffer_accumulate_device(rtgi_hip_tracer *h, float *d_dst, const float *d_src, uint32_t n, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (n == 0) return RTGI_OK;
	if (!d_dst || !d_src || ((reinterpret_cast<uintptr_t>(d_dst) | reinterpret_cast<uintptr_t>(d_src)) & 15u))
		return fail(h, RTGI_ERR_INVALID, "framebuffer_accumulate: null or misaligned buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	hipLaunchKernelGGL(rtgi_dev::framebuffer_accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float4 *)d_dst,
	                   (const float4 *)d_src, n);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}

int rtgi_hip_shadow_rays_device(rtgi_hip_tracer *h, const rtgi_hit *d_hits, uint32_t n, const float light_pos[3],
                                rtgi_ray *d_shadow, void *stream) {
	if (!h || !light_pos) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (!h->built) return fail(h, RTGI_ERR_NOT_BUILT, "no acceleration structure: call build first");
	if (n == 0) return RTGI_OK;
	if (!d_hits || !d_shadow || ((reinterpret_cast<uintptr_t>(d_hits) | reinterpret_cast<uintptr_t>(d_shadow)) & 15u))
		return fail(h, RTGI_ERR_INVALID, "shadow_rays: null or misaligned buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	rtgi_dev::ShadowArgs a;
	a.hits = (const float4 *)d_hits;
	a.corners = h->d_corners;
	a.out = (float4 *)d_shadow;
	for (int k = 0; k < 3; ++k) a.light[k] = light_pos[k];
	a.n = n;
	hipLaunchKernelGGL(rtgi_dev::shadow_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
	HIP_TRY(h, hipGetLastError());
	if (d_shadow == h->camera_rays_at) h->camera_rays_at = nullptr;  // (that buffer holds other rays now)
	return RTGI_OK;
}

int rtgi_hip_set_shading(rtgi_hip_tracer *h, const void *vertex_normals, size_t normal_stride, uint32_t n_vertices,
                         const void *triangles, size_t tri_stride, uint32_t n_triangles, const float *albedo_rgb,
                         uint32_t n_materials) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (!h->built) return fail(h, RTGI_ERR_NOT_BUILT, "no acceleration structure: call build first");
	if (!vertex_normals || !triangles || !albedo_rgb || n_materials == 0 || normal_stride < 12 || tri_stride < 16)
		return fail(h, RTGI_ERR_INVALID, "set_shading: null array, no materials, or strides below 12 / 16 bytes");
	if (n_triangles != h->n_scene_triangles || n_vertices != h->n_scene_vertices)
		return fail(h, RTGI_ERR_INVALID, "set_shading: not the scene that was built");
	const bool flat = h->cfg.layout == RTGI_LAYOUT_FLAT;
	const uint32_t n_ids = flat ? (uint32_t)h->tree.order.size() : n_triangles;
	std::vector<float> normals((size_t)n_ids * 4);
	for (uint32_t id = 0; id < n_ids; ++id) {
		const uint32_t src = flat ? h->tree.order[id] : id;  // hits of the flat layout name slots of the permuted list
		uint32_t t[4];
		memcpy(t, (const unsigned char *)triangles + (size_t)src * tri_stride, 16);
		if (t[0] >= n_vertices || t[1] >= n_vertices || t[2] >= n_vertices)
			return fail(h, RTGI_ERR_INVALID, "set_shading: vertex id out of range");
		if (t[3] >= n_materials) return fail(h, RTGI_ERR_INVALID, "set_shading: material id out of range");
		float na[3], nb[3], nc[3], sum[3];
		memcpy(na, (const unsigned char *)vertex_normals + (size_t)t[0] * normal_stride, 12);
		memcpy(nb, (const unsigned char *)vertex_normals + (size_t)t[1] * normal_stride, 12);
		memcpy(nc, (const unsigned char *)vertex_normals + (size_t)t[2] * normal_stride, 12);
		for (int k = 0; k < 3; ++k) sum[k] = na[k] + nb[k] + nc[k];
		// glm::normalize: v * inversesqrt(dot(v, v))
		const float px = sum[0] * sum[0], py = sum[1] * sum[1], pz = sum[2] * sum[2];
		const float inv = 1.0f / std::sqrt(px + py + pz);
		float *o = &normals[(size_t)id * 4];
		for (int k = 0; k < 3; ++k) o[k] = sum[k] * inv;
		memcpy(o + 3, &t[3], 4);
	}
	std::vector<float> albedo((size_t)n_materials * 4, 0.0f);
	for (uint32_t m = 0; m < n_materials; ++m)
		for (int k = 0; k < 3; ++k) albedo[(size_t)m * 4 + k] = albedo_rgb[(size_t)m * 3 + k];
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	if (h->d_normals) (void)hipFree(h->d_normals);
	if (h->d_albedo) (void)hipFree(h->d_albedo);
	if (h->d_matparams) (void)hipFree(h->d_matparams);
	free_textures(h);
	h->d_normals = h->d_albedo = h->d_matparams = nullptr;
	h->n_materials = 0;
	// (material defaults of libgi/material.h:58-59: ior 1.3, roughness 0.1; brdf lambert, libgi/scene.cpp:100-102)
	std::vector<float> params((size_t)n_materials * 4, 0.0f);
	for (uint32_t m = 0; m < n_materials; ++m) params[(size_t)m * 4] = 1.3f, params[(size_t)m * 4 + 1] = 0.1f;
	HIP_TRY(h, hipMalloc((void **)&h->d_normals, std::max<size_t>(normals.size() * sizeof(float), 16)));
	HIP_TRY(h, hipMalloc((void **)&h->d_albedo, albedo.size() * sizeof(float)));
	HIP_TRY(h, hipMalloc((void **)&h->d_matparams, params.size() * sizeof(float)));
	if (!normals.empty()) HIP_TRY(h, hipMemcpy(h->d_normals, normals.data(), normals.size() * sizeof(float), hipMemcpyHostToDevice));
	HIP_TRY(h, hipMemcpy(h->d_albedo, albedo.data(), albedo.size() * sizeof(float), hipMemcpyHostToDevice));
	HIP_TRY(h, hipMemcpy(h->d_matparams, params.data(), params.size() * sizeof(float), hipMemcpyHostToDevice));
	h->n_materials = n_materials;
	return RTGI_OK;
}

int rtgi_hip_set_materials(rtgi_hip_tracer *h, const rtgi_material *materials, uint32_t n_materials) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (!h->n_materials) return fail(h, RTGI_ERR_NOT_BUILT, "no shading data: call set_shading after build");
	if (!materials || n_materials != h->n_materials) return fail(h, RTGI_ERR_INVALID, "set_materials: as many materials as set_shading was given");
	std::vector<float> albedo((size_t)n_materials * 4, 0.0f), params((size_t)n_materials * 4, 0.0f);
	for (uint32_t m = 0; m < n_materials; ++m) {
		if (materials[m].brdf > RTGI_BRDF_LAYERED_GTR2) return fail(h, RTGI_ERR_INVALID, "set_materials: no such brdf");
		for (int k = 0; k < 3; ++k) albedo[(size_t)m * 4 + k] = materials[m].albedo[k];
		params[(size_t)m * 4] = materials[m].ior;
		params[(size_t)m * 4 + 1] = materials[m].roughness;
		memcpy(&params[(size_t)m * 4 + 2], &materials[m].brdf, 4);
	}
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	// (stream-ordered behind whatever still shades with the old values on the null stream's peers: wait for the GPU)
	HIP_TRY(h, hipDeviceSynchronize());
	HIP_TRY(h, hipMemcpy(h->d_albedo, albedo.data(), albedo.size() * sizeof(float), hipMemcpyHostToDevice));
	HIP_TRY(h, hipMemcpy(h->d_matparams, params.data(), params.size() * sizeof(float), hipMemcpyHostToDevice));
	return RTGI_OK;
}

int rtgi_hip_set_textures(rtgi_hip_tracer *h, const void *vertex_tc, size_t tc_stride, uint32_t n_vertices, const void *triangles,
                          size_t tri_stride, uint32_t n_triangles, const rtgi_texture *textures, uint32_t n_textures,
                          const int32_t *material_texture, uint32_t n_materials) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (!h->n_materials) return fail(h, RTGI_ERR_NOT_BUILT, "no shading data: call set_shading after build");
	if (n_materials != h->n_materials || n_triangles != h->n_scene_triangles || n_vertices != h->n_scene_vertices)
		return fail(h, RTGI_ERR_INVALID, "set_textures: not the scene and material table that set_shading was given");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	HIP_TRY(h, hipDeviceSynchronize());  // (shading launches may still read the old tables)
	free_textures(h);
	if (n_textures == 0) return RTGI_OK;  // `material texture drop` for all
	if (!vertex_tc || !triangles || !textures || !material_texture || tc_stride < 8 || tri_stride < 12)
		return fail(h, RTGI_ERR_INVALID, "set_textures: null array or strides below 8 / 12 bytes");
	std::vector<uint32_t> first(n_textures);
	size_t total = 0;
	for (uint32_t t = 0; t < n_textures; ++t) {
		if (!textures[t].rgb || textures[t].w == 0 || textures[t].h == 0 || textures[t].w > 32768 || textures[t].h > 32768)
			return fail(h, RTGI_ERR_INVALID, "set_textures: empty or oversized texture");
		first[t] = (uint32_t)total;
		total += (size_t)textures[t].w * textures[t].h;
		if (total > 0x7fffffffu) return fail(h, RTGI_ERR_INVALID, "set_textures: more than 2^31 texels");
	}
	std::vector<uint32_t> of_material((size_t)n_materials * 4, 0u);
	for (uint32_t m = 0; m < n_materials; ++m) {
		const int32_t t = material_texture[m];
		if (t < 0) continue;
		if ((uint32_t)t >= n_textures) return fail(h, RTGI_ERR_INVALID, "set_textures: texture index out of range");
		of_material[4 * (size_t)m] = first[(size_t)t], of_material[4 * (size_t)m + 1] = textures[t].w, of_material[4 * (size_t)m + 2] = textures[t].h;
		of_material[4 * (size_t)m + 3] = 1u;
	}
	const bool flat = h->cfg.layout == RTGI_LAYOUT_FLAT;
	const uint32_t n_ids = flat ? (uint32_t)h->tree.order.size() : n_triangles;
	std::vector<float> coords((size_t)n_ids * 8, 0.0f);
	for (uint32_t id = 0; id < n_ids; ++id) {
		const uint32_t src = flat ? h->tree.order[id] : id;  // hits of the flat layout name slots of the permuted list
		uint32_t v[3];
		memcpy(v, (const unsigned char *)triangles + (size_t)src * tri_stride, 12);
		for (int c = 0; c < 3; ++c) {
			if (v[c] >= n_vertices) return fail(h, RTGI_ERR_INVALID, "set_textures: vertex id out of range");
			memcpy(&coords[(size_t)id * 8 + 2 * (size_t)c], (const unsigned char *)vertex_tc + (size_t)v[c] * tc_stride, 8);
		}
	}
	std::vector<float> texels(total * 4, 0.0f);
	for (uint32_t t = 0; t < n_textures; ++t)
		for (size_t i = 0, n = (size_t)textures[t].w * textures[t].h; i < n; ++i)
			memcpy(&texels[(first[t] + i) * 4], textures[t].rgb + 3 * i, 12);
	HIP_TRY(h, hipMalloc((void **)&h->d_tex_coords, std::max<size_t>(coords.size() * sizeof(float), 32)));
	HIP_TRY(h, hipMalloc((void **)&h->d_texels, texels.size() * sizeof(float)));
	HIP_TRY(h, hipMalloc((void **)&h->d_tex_of_material, of_material.size() * sizeof(uint32_t)));
	if (!coords.empty()) HIP_TRY(h, hipMemcpy(h->d_tex_coords, coords.data(), coords.size() * sizeof(float), hipMemcpyHostToDevice));
	HIP_TRY(h, hipMemcpy(h->d_texels, texels.data(), texels.size() * sizeof(float), hipMemcpyHostToDevice));
	HIP_TRY(h, hipMemcpy(h->d_tex_of_material, of_material.data(), of_material.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	return RTGI_OK;
}

namespace {
int shade(rtgi_hip_tracer *h, bool local, const rtgi_ray *d_rays, const rtgi_hit *d_hits, const uint8_t *d_occluded, uint32_t n, const float *light_pos,
          const float *light_col, float *d_radiance, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (!h->built) return fail(h, RTGI_ERR_NOT_BUILT, "no acceleration structure: call build first");
	if (!h->n_materials) return fail(h, RTGI_ERR_NOT_BUILT, "no shading data: call set_shading after build");
	if (n == 0) return RTGI_OK;
	if (!d_hits || !d_radiance || ((reinterpret_cast<uintptr_t>(d_hits) | reinterpret_cast<uintptr_t>(d_radiance)) & 15u))
		return fail(h, RTGI_ERR_INVALID, "shade: null or misaligned buffer");
	if (local && (!d_occluded || !light_pos || !light_col)) return fail(h, RTGI_ERR_INVALID, "shade_local: null argument");
	if (d_rays && (reinterpret_cast<uintptr_t>(d_rays) & 15u)) return fail(h, RTGI_ERR_INVALID, "shade: misaligned ray buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	rtgi_dev::ShadeArgs a;
	a.hits = (const float4 *)d_hits;
	a.occluded = d_occluded;
	a.corners = h->d_corners;
	a.normals = h->d_normals;
	a.albedo = h->d_albedo;
	a.out = (float4 *)d_radiance;
	const float pi = (float)3.14159265358979323846;  // libgi/rt.h:14
	for (int k = 0; k < 3; ++k) {
		a.light[k] = local ? light_pos[k] : 0.0f;
		a.power[k] = local ? 4 * pi * light_col[k] : 0.0f;  // pointlight::power, libgi/scene.cpp:216-218
	}
	a.n = n;
	a.n_materials = h->n_materials;
	a.rays = (const float4 *)d_rays;
	a.params = h->d_matparams;
	a.tex_coords = h->d_tex_coords;
	a.tex_of_material = h->d_tex_of_material;
	a.texels = h->d_texels;
	if (local && d_rays)
		hipLaunchKernelGGL(rtgi_dev::shade_local_brdf_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
	else if (local)
		hipLaunchKernelGGL(rtgi_dev::shade_local_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
	else
		hipLaunchKernelGGL(rtgi_dev::shade_primary_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}
}  // namespace

int rtgi_hip_shade_primary_device(rtgi_hip_tracer *h, const rtgi_hit *d_hits, uint32_t n, float *d_radiance, void *stream) {
	return shade(h, false, nullptr, d_hits, nullptr, n, nullptr, nullptr, d_radiance, stream);
}

int rtgi_hip_shade_local_device(rtgi_hip_tracer *h, const rtgi_hit *d_hits, const uint8_t *d_occluded, uint32_t n,
                                const float light_pos[3], const float light_col[3], float *d_radiance, void *stream) {
	return shade(h, true, nullptr, d_hits, d_occluded, n, light_pos, light_col, d_radiance, stream);
}

int rtgi_hip_shade_local_brdf_device(rtgi_hip_tracer *h, const rtgi_ray *d_rays, const rtgi_hit *d_hits, const uint8_t *d_occluded,
                                     uint32_t n, const float light_pos[3], const float light_col[3], float *d_radiance, void *stream) {
	if (h && n && !d_rays) return fail(h, RTGI_ERR_INVALID, "shade_local_brdf: null ray buffer");
	return shade(h, true, d_rays, d_hits, d_occluded, n, light_pos, light_col, d_radiance, stream);
}

int rtgi_hip_framebuffer_add_device(rtgi_hip_tracer *h, float *d_framebuffer, const float *d_radiance, uint32_t n, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (n == 0) return RTGI_OK;
	if (!d_framebuffer || !d_radiance ||
	    ((reinterpret_cast<uintptr_t>(d_framebuffer) | reinterpret_cast<uintptr_t>(d_radiance)) & 15u))
		return fail(h, RTGI_ERR_INVALID, "framebuffer_add: null or misaligned buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	hipLaunchKernelGGL(rtgi_dev::framebuffer_add_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
	                   (float4 *)d_framebuffer, (const float4 *)d_radiance, n);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}

int rtgi_hip_framebuffer_rgb8_device(rtgi_hip_tracer *h, const float *d_framebuffer, uint32_t w, uint32_t height,
                                     uint8_t *d_rgb, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (w == 0 || height == 0) return RTGI_OK;
	if ((uint64_t)w * height > 0x7fffffffull) return fail(h, RTGI_ERR_INVALID, "framebuffer_rgb8: image too large");
	if (!d_framebuffer || !d_rgb || (reinterpret_cast<uintptr_t>(d_framebuffer) & 15u))
		return fail(h, RTGI_ERR_INVALID, "framebuffer_rgb8: null or misaligned buffer");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	hipLaunchKernelGGL(rtgi_dev::framebuffer_rgb8_kernel, dim3((w * height + 255) / 256), dim3(256), 0, (hipStream_t)stream,
	                   (const float4 *)d_framebuffer, w, height, d_rgb);
	HIP_TRY(h, hipGetLastError());
	return RTGI_OK;
}

int rtgi_hip_bounce_rays_device(rtgi_hip_tracer *h, const rtgi_ray *d_rays, const rtgi_hit *d_hits, uint32_t n,
                                uint32_t seed, uint32_t bounce, rtgi_ray *d_out, uint32_t *d_parent, uint32_t *d_count,
                                void *stream) {
	return rtgi_hip_bounce_rays_keyed_device(h, d_rays, d_hits, n, nullptr, nullptr, seed, bounce, d_out, d_parent, d_count, stream);
}

int rtgi_hip_bounce_rays_keyed_device(rtgi_hip_tracer *h, const rtgi_ray *d_rays, const rtgi_hit *d_hits, uint32_t n,
                                      const uint32_t *d_keys, const uint32_t *d_key_table, uint32_t seed, uint32_t bounce,
                                      rtgi_ray *d_out, uint32_t *d_parent, uint32_t *d_count, void *stream) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (!h->built) return fail(h, RTGI_ERR_NOT_BUILT, "no acceleration structure: call build first");
	if (!d_count) return fail(h, RTGI_ERR_INVALID, "bounce_rays: d_count is required");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	hipStream_t st = (hipStream_t)stream;
	if (n == 0) {
		HIP_TRY(h, hipMemsetAsync(d_count, 0, sizeof(uint32_t), st));
		return RTGI_OK;
	}
	if (!d_rays || !d_hits || !d_out ||
	    ((reinterpret_cast<uintptr_t>(d_rays) | reinterpret_cast<uintptr_t>(d_hits) | reinterpret_cast<uintptr_t>(d_out)) & 15u))
		return fail(h, RTGI_ERR_INVALID, "bounce_rays: null or misaligned buffer");
	const uint32_t n_blocks = (n + 1023) / 1024;  // one count per 1024-ray block
	if (n_blocks > h->block_counts_cap) {
		if (h->d_block_counts) (void)hipFree(h->d_block_counts);
		h->d_block_counts = nullptr;
		h->block_counts_cap = 0;
		HIP_TRY(h, hipMalloc((void **)&h->d_block_counts, (size_t)n_blocks * sizeof(uint32_t)));
		h->block_counts_cap = n_blocks;
	}
	if (d_out == h->camera_rays_at) h->camera_rays_at = nullptr;  // (that buffer holds other rays now)
	rtgi_dev::BounceArgs a;
	a.rays = (const float4 *)d_rays;
	a.hits = (const float4 *)d_hits;
	a.corners = h->d_corners;
	a.block_counts = h->d_block_counts;
	a.out = (float4 *)d_out;
	a.parent = d_parent;
	a.keys = d_keys, a.key_table = d_key_table;
	a.n = n, a.seed = seed, a.bounce = bounce;
	if (int rc = scratch_begin(h, st)) return rc;
	hipLaunchKernelGGL(rtgi_dev::bounce_count_kernel, dim3(n_blocks), dim3(1024), 0, st, a);
	hipLaunchKernelGGL(rtgi_dev::scan_kernel, dim3(1), dim3(1024), 0, st, h->d_block_counts, n_blocks, d_count);
	hipLaunchKernelGGL(rtgi_dev::bounce_write_kernel, dim3(n_blocks), dim3(1024), 0, st, a);
	HIP_TRY(h, hipGetLastError());
	return scratch_end(h, st);
}

int rtgi_hip_profile_begin(rtgi_hip_tracer *h) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	if (!h->d_totals) HIP_TRY(h, hipMalloc((void **)&h->d_totals, 8 * sizeof(unsigned long long)));
	HIP_TRY(h, hipMemset(h->d_totals, 0, 8 * sizeof(unsigned long long)));
	h->profiling = true;
	h->events_used = 0;
	return RTGI_OK;
}

int rtgi_hip_profile_work(rtgi_hip_tracer *h, uint64_t out[4]) {
	if (!h || !out) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (!h->d_totals) return fail(h, RTGI_ERR_INVALID, "profile_work: call profile_begin first");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	HIP_TRY(h, hipDeviceSynchronize());
	unsigned long long v[4];
	HIP_TRY(h, hipMemcpy(v, h->d_totals, sizeof(v), hipMemcpyDeviceToHost));
	for (int k = 0; k < 4; ++k) out[k] = v[k];
	return RTGI_OK;
}

int rtgi_hip_launch_stats(const rtgi_hip_tracer *h, uint64_t *launches, uint64_t *sorted) {
	if (!h) return RTGI_ERR_INVALID;
	if (launches) *launches = h->launches;
	if (sorted) *sorted = h->launches_sorted;
	return RTGI_OK;
}

int rtgi_hip_host_path_stats(const rtgi_hip_tracer *h, uint64_t *one_launch, uint64_t *gave_up) {
	if (!h) return RTGI_ERR_INVALID;
	if (one_launch) *one_launch = h->streamed_calls;
	if (gave_up) *gave_up = h->streamed_gave_up;
	return RTGI_OK;
}

int rtgi_hip_profile_details(rtgi_hip_tracer *h, uint64_t *spill_pushes, uint64_t *link_fetches, uint32_t *lds_levels, uint32_t *spill_levels) {
	if (!h) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	if (lds_levels) *lds_levels = h->stack_levels;
	if (spill_levels) *spill_levels = h->spill_levels;
	if (spill_pushes) *spill_pushes = 0;
	if (link_fetches) *link_fetches = 0;
	if ((!spill_pushes && !link_fetches) || !h->d_totals) return RTGI_OK;  // (no profile_begin yet: nothing has been counted)
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	HIP_TRY(h, hipDeviceSynchronize());
	unsigned long long v[2] = {0, 0};
	HIP_TRY(h, hipMemcpy(v, h->d_totals + 4, sizeof(v), hipMemcpyDeviceToHost));
	if (spill_pushes) *spill_pushes = v[0];
	if (link_fetches) *link_fetches = v[1];
	return RTGI_OK;
}

int rtgi_hip_profile_end(rtgi_hip_tracer *h, double *kernel_ms_total, uint32_t *launches) {
	if (!h || !kernel_ms_total || !launches) return RTGI_ERR_INVALID;
	if (h->device < 0) return fail(h, RTGI_ERR_DEVICE, "host-only handle: needs a GPU");
	DeviceScope device_scope(h->device);
	HIP_TRY(h, device_scope.status);
	double total = 0.0;
	for (size_t i = 0; i < h->events_used; ++i) {
		HIP_TRY(h, hipEventSynchronize(h->events[i].second));
		float ms = 0.0f;
		HIP_TRY(h, hipEventElapsedTime(&ms, h->events[i].first, h->events[i].second));
		total += ms;
	}
	*kernel_ms_total = total;
	*launches = (uint32_t)h->events_used;
	h->profiling = false;
	h->events_used = 0;
	return RTGI_OK;
}

const char *rtgi_hip_last_error(const rtgi_hip_tracer *h) { return h ? h->error.c_str() : g_create_error.c_str(); }

}  // extern "C"
