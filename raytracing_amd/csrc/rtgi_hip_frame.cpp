// rtgi_hip_frame.cpp -- whole frames of the wavefront algorithms on one or several GPUs (C ABI of
// include/rtgi_hip_frame.h). Host code only: every stage is a call into librtgi_hip.so, the exchange between GPUs is
// one RCCL all-reduce of the framebuffer. One host thread and one stream per GPU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rtgi_hip_frame.h"

namespace {

thread_local std::string g_create_error;

double now_ms() {
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

constexpr uint32_t MAX_GROUP = 31;           // samples of a part's pixels traced together (a path's launches: one batch of all their shadow
                                             // rays and one batch of bounce rays per sample, at most 32 batches)
constexpr uint32_t GROUP_RAYS = 8u << 20;    // ... as many as make about this many camera rays (2 Mi: 1133, 4 Mi: 1221, 8 Mi: 1260,
                                             // 16 Mi: 1273 Mrays/s over a path-traced 1080p frame)

// One GPU's share of the renderer: its tracer handle, its stream and its buffers (all sized for `cap` pixels).
struct Part {
	int device = 0;
	rtgi_hip_tracer *h = nullptr;
	hipStream_t stream = nullptr;
	ncclComm_t comm = nullptr;
	std::string error;
	int rc = RTGI_OK;

	// this frame's pixels (tile sharding: this GPU's tiles; else all) and samples
	std::vector<uint32_t> pixels;
	bool all_pixels = true;
	uint32_t n = 0;  // pixels of this part
	uint64_t rays = 0;
	double t_traced = 0.0;

	size_t cap = 0, cap_full = 0;
	uint32_t *d_pixels = nullptr, *d_slot[2] = {nullptr, nullptr}, *d_parent = nullptr, *d_count = nullptr;
	float *d_offsets = nullptr, *d_direct = nullptr, *d_albedo = nullptr, *d_throughput[2] = {nullptr, nullptr};
	float *d_radiance = nullptr, *d_fb = nullptr, *d_full = nullptr;
	rtgi_ray *d_rays[2] = {nullptr, nullptr}, *d_shadow = nullptr;
	rtgi_hit *d_hits = nullptr, *d_hits2 = nullptr;
	uint8_t *d_occluded = nullptr, *d_rgb = nullptr;
	int32_t *d_status = nullptr;  // one process per GPU: the word the ranks agree on before the framebuffer exchange
};

}  // namespace

struct rtgi_hip_renderer {
	std::vector<Part> parts;
	bool distinct = true;  // every part on a GPU of its own: RCCL combines the framebuffers
	bool comms_ready = false;
	// one process per GPU: this renderer is rank world_rank of world_size processes (0: all parts live in this process)
	int world_rank = 0, world_size = 0;
	std::string error;
};

namespace {

int fail(rtgi_hip_renderer *r, int code, const std::string &msg) {
	if (r) r->error = msg;
	return code;
}
int fail(Part &p, int code, const std::string &msg) {
	p.error = msg;
	p.rc = code;
	return code;
}

#define PART_HIP(p, expr)                                                                                   \
	do {                                                                                                    \
		hipError_t e_ = (expr);                                                                             \
		if (e_ != hipSuccess) return fail(p, RTGI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)
#define PART_RT(p, expr)                                                                     \
	do {                                                                                     \
		int rc_ = (expr);                                                                    \
		if (rc_ < 0) return fail(p, rc_, std::string(#expr) + ": " + rtgi_hip_last_error((p).h)); \
	} while (0)

void free_buffers(Part &p) {
	if (hipSetDevice(p.device) != hipSuccess) return;
	void *all[] = {p.d_pixels, p.d_slot[0], p.d_slot[1], p.d_parent, p.d_count, p.d_offsets, p.d_direct, p.d_albedo, p.d_throughput[0],
	               p.d_throughput[1], p.d_radiance, p.d_fb, p.d_full, p.d_rays[0], p.d_rays[1], p.d_shadow, p.d_hits, p.d_hits2, p.d_occluded, p.d_rgb};
	for (void *q : all)
		if (q) (void)hipFree(q);
	p.d_pixels = p.d_slot[0] = p.d_slot[1] = p.d_parent = p.d_count = nullptr;
	p.d_offsets = p.d_direct = p.d_albedo = p.d_throughput[0] = p.d_throughput[1] = p.d_radiance = p.d_fb = p.d_full = nullptr;
	p.d_rays[0] = p.d_rays[1] = p.d_shadow = nullptr;
	p.d_hits = p.d_hits2 = nullptr;
	p.d_occluded = p.d_rgb = nullptr;
	p.cap = p.cap_full = 0;
}

// Buffers for n pixels of this part and a frame of n_full pixels; kept between frames, grown on demand.
int ensure_buffers(Part &p, size_t n, size_t n_full) {
	if (n <= p.cap && n_full <= p.cap_full) return RTGI_OK;
	free_buffers(p);
	PART_HIP(p, hipSetDevice(p.device));
	const size_t c = std::max<size_t>(n, 1), f = std::max<size_t>(n_full, 1);
	PART_HIP(p, hipMalloc((void **)&p.d_pixels, c * 4));
	PART_HIP(p, hipMalloc((void **)&p.d_slot[0], c * 4));
	PART_HIP(p, hipMalloc((void **)&p.d_slot[1], c * 4));
	PART_HIP(p, hipMalloc((void **)&p.d_parent, c * 4));
	PART_HIP(p, hipMalloc((void **)&p.d_count, 256));
	PART_HIP(p, hipMalloc((void **)&p.d_offsets, c * 8));
	PART_HIP(p, hipMalloc((void **)&p.d_direct, c * 16));
	PART_HIP(p, hipMalloc((void **)&p.d_albedo, c * 16));
	PART_HIP(p, hipMalloc((void **)&p.d_throughput[0], c * 16));
	PART_HIP(p, hipMalloc((void **)&p.d_throughput[1], c * 16));
	PART_HIP(p, hipMalloc((void **)&p.d_radiance, c * 16));
	PART_HIP(p, hipMalloc((void **)&p.d_fb, c * 16));
	PART_HIP(p, hipMalloc((void **)&p.d_full, f * 16));
	PART_HIP(p, hipMalloc((void **)&p.d_rays[0], c * sizeof(rtgi_ray)));
	PART_HIP(p, hipMalloc((void **)&p.d_rays[1], c * sizeof(rtgi_ray)));
	PART_HIP(p, hipMalloc((void **)&p.d_shadow, c * sizeof(rtgi_ray)));
	PART_HIP(p, hipMalloc((void **)&p.d_hits, c * sizeof(rtgi_hit)));
	PART_HIP(p, hipMalloc((void **)&p.d_hits2, c * sizeof(rtgi_hit)));
	PART_HIP(p, hipMalloc((void **)&p.d_occluded, c));
	PART_HIP(p, hipMalloc((void **)&p.d_rgb, f * 3));
	p.cap = c;
	p.cap_full = f;
	return RTGI_OK;
}

// The pixels of every `world`-th tile starting with tile `rank`, tiles numbered row by row, pixels of a tile row by
// row: neighbouring rays of a batch are neighbours in the image.
void tile_pixels(int w, int h, uint32_t tile, int world, int rank, std::vector<uint32_t> &out) {
	out.clear();
	int t = 0;
	for (int y0 = 0; y0 < h; y0 += (int)tile)
		for (int x0 = 0; x0 < w; x0 += (int)tile, ++t) {
			if (t % world != rank) continue;
			for (int y = y0; y < std::min(h, y0 + (int)tile); ++y)
				for (int x = x0; x < std::min(w, x0 + (int)tile); ++x) out.push_back((uint32_t)y * (uint32_t)w + (uint32_t)x);
		}
}

// The samples of a pixel that part `rank` of `world` takes: first, first + step, ... (all of them unless the frame is
// shared by samples).
void sample_share(uint32_t sharding, int world, int rank, uint32_t &first, uint32_t &step) {
	const bool by_samples = world > 1 && sharding == RTGI_SHARD_SAMPLES;
	first = by_samples ? (uint32_t)rank : 0u;
	step = by_samples ? (uint32_t)world : 1u;
}

// This part's share of the frame, up to its framebuffer in place in p.d_full (zero where it has no samples).
int render_part(Part &p, const rtgi_frame &f, int rank, int world) {
	PART_HIP(p, hipSetDevice(p.device));
	// fault injection for the tests of the error paths: RTGI_HIP_FAIL_PART=i makes part (or rank) i fail here
	if (const char *e = getenv("RTGI_HIP_FAIL_PART"))
		if (atoi(e) == rank) return fail(p, RTGI_ERR_DEVICE, "failure injected by RTGI_HIP_FAIL_PART");
	const uint32_t n_full = (uint32_t)f.cam.w * (uint32_t)f.cam.h;
	const bool by_tiles = world > 1 && f.sharding == RTGI_SHARD_TILES;
	p.all_pixels = !by_tiles;
	if (by_tiles)
		tile_pixels(f.cam.w, f.cam.h, f.tile ? f.tile : 64u, world, rank, p.pixels);
	p.n = by_tiles ? (uint32_t)p.pixels.size() : n_full;
	p.rays = 0;
	// Samples are traced in groups: a launch over fewer rays than the GPU holds at once (256 CUs x 1024 lanes) is all
	// tail, and a GPU that has an eighth of a frame's tiles gets that few per sample. The buffers hold `group` samples of
	// this part's pixels one behind the other, the per-ray stages run per sample or over the whole group, and the
	// traversal launches take the group's rays together. The order in which samples enter the framebuffer is unchanged,
	// and so is every value in it.
	uint32_t first, step;
	sample_share(f.sharding, world, rank, first, step);
	std::vector<uint32_t> samples;
	for (uint32_t sample = first; sample < f.sppx; sample += step) samples.push_back(sample);
	uint32_t group = 1;
	{
		const char *e = getenv("RTGI_HIP_FRAME_GROUP");  // most samples per group (1: every sample on its own)
		const uint32_t most = e ? (uint32_t)std::max(1, std::min(atoi(e), (int)MAX_GROUP)) : MAX_GROUP;
		const char *e2 = getenv("RTGI_HIP_FRAME_GROUP_RAYS");  // camera rays per group to aim for
		const uint32_t aim = e2 ? (uint32_t)std::max(1, std::min(atoi(e2), 1 << 28)) : GROUP_RAYS;
		if (p.n > 0) group = std::max<uint32_t>(1, std::min<uint32_t>({most, aim / p.n, (uint32_t)std::max<size_t>(samples.size(), 1)}));
		// (groups of equal size rather than full ones and a small rest)
		const uint32_t n_groups = ((uint32_t)samples.size() + group - 1) / group;
		if (n_groups > 0) group = ((uint32_t)samples.size() + n_groups - 1) / n_groups;
	}
	if (int rc = ensure_buffers(p, (size_t)p.n * group, n_full)) return rc;
	hipStream_t st = p.stream;
	const uint32_t n = p.n;
	const uint32_t *d_pixels = nullptr;
	if (by_tiles) {
		PART_HIP(p, hipMemcpyAsync(p.d_pixels, p.pixels.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
		d_pixels = p.d_pixels;
	}
	float *fb = by_tiles ? p.d_fb : p.d_full;  // framebuffer::clear
	PART_HIP(p, hipMemsetAsync(fb, 0, (size_t)std::max<uint32_t>(n, 1) * 16, st));
	for (size_t g0 = 0; g0 < samples.size() && n > 0; g0 += group) {
		const uint32_t S = (uint32_t)std::min<size_t>(group, samples.size() - g0);
		const uint32_t all = n * S;  // (n * group <= GROUP_RAYS or group == 1: no overflow)
		for (uint32_t k = 0; k < S; ++k) {  // sample k of the group lives at [k * n, (k + 1) * n) of every buffer
			const uint32_t sample = samples[g0 + k];
			const float *offsets = nullptr;
			if (f.jitter) {
				float *mine = p.d_offsets + (size_t)k * n * 2;
				if (d_pixels)
					PART_RT(p, rtgi_hip_pixel_offsets_pixels_device(p.h, d_pixels, n, sample, f.seed, mine, st));
				else
					PART_RT(p, rtgi_hip_pixel_offsets_device(p.h, n, sample, f.seed, mine, st));
				offsets = mine;
			}
			if (d_pixels)
				PART_RT(p, rtgi_hip_camera_rays_pixels_device(p.h, &f.cam, d_pixels, n, offsets, p.d_rays[0] + (size_t)k * n, st));
			else
				PART_RT(p, rtgi_hip_camera_rays_device(p.h, &f.cam, offsets, p.d_rays[0] + (size_t)k * n, st));
		}
		PART_RT(p, rtgi_hip_closest_hit_device(p.h, p.d_rays[0], all, p.d_hits, nullptr, st));
		p.rays += all;
		if (f.algo == RTGI_ALGO_PRIMARY) {
			PART_RT(p, rtgi_hip_shade_primary_device(p.h, p.d_hits, all, p.d_radiance, st));
		} else if (f.algo == RTGI_ALGO_LOCAL) {
			PART_RT(p, rtgi_hip_shadow_rays_device(p.h, p.d_hits, all, f.light_pos, p.d_shadow, st));
			PART_RT(p, rtgi_hip_any_hit_device(p.h, p.d_shadow, all, p.d_occluded, nullptr, st));
			p.rays += all;
			PART_RT(p, rtgi_hip_shade_local_brdf_device(p.h, p.d_rays[0], p.d_hits, p.d_occluded, all, f.light_pos, f.light_col, p.d_radiance, st));
		} else {
			// a path per pixel sample: light every vertex, then follow one diffuse bounce, `bounces` times. The shadow
			// rays of a vertex and the rays of the next segment do not depend on each other: they are traced in ONE
			// launch (rtgi_hip_trace_batches_device), which ends in one tail instead of two - for all samples of the group.
			rtgi_hit *hits[2] = {p.d_hits, p.d_hits2};
			uint32_t m[MAX_GROUP], next[MAX_GROUP], at_shadow[MAX_GROUP];  // at_shadow: where a sample's shadow rays begin in the group's
			for (uint32_t k = 0; k < S; ++k) {
				const size_t at = (size_t)k * n;
				PART_RT(p, rtgi_hip_path_begin_device(p.h, n, p.d_throughput[0] + at * 4, p.d_slot[0] + at, p.d_radiance + at * 4, st));
				m[k] = n;
			}
			int cur = 0, hcur = 0;
			for (uint32_t depth = 0;; ++depth) {
				uint32_t n_shadow = 0;  // the shadow rays of the group's samples lie one behind the other: one any_hit batch
				for (uint32_t k = 0; k < S; ++k) {
					const size_t at = (size_t)k * n;
					next[k] = 0;
					at_shadow[k] = n_shadow;
					if (m[k] == 0) continue;
					PART_RT(p, rtgi_hip_shadow_rays_device(p.h, hits[hcur] + at, m[k], f.light_pos, p.d_shadow + at_shadow[k], st));
					n_shadow += m[k];
					if (depth < f.bounces) {
						PART_RT(p, rtgi_hip_shade_primary_device(p.h, hits[hcur] + at, m[k], p.d_albedo + at * 4, st));
						// (seed: one stream of numbers per sample; keys: the pixel, so that a path does not depend on the sharding)
						PART_RT(p, rtgi_hip_bounce_rays_keyed_device(p.h, p.d_rays[cur] + at, hits[hcur] + at, m[k], p.d_slot[cur] + at, d_pixels,
						                                             f.seed * 0x9E3779B1u + samples[g0 + k], depth + 1, p.d_rays[1 - cur] + at,
						                                             p.d_parent + at, p.d_count + k, st));
					}
				}
				if (depth < f.bounces) {
					// the numbers of rays that go on are needed on the host: the next launches are sized by them
					PART_HIP(p, hipMemcpyAsync(next, p.d_count, S * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
					PART_HIP(p, hipStreamSynchronize(st));
					for (uint32_t k = 0; k < S; ++k)
						if (m[k] == 0) next[k] = 0;
				}
				rtgi_batch both[1 + MAX_GROUP];
				uint32_t n_both = 0, more = 0;
				if (n_shadow) both[n_both++] = {p.d_shadow, p.d_occluded, n_shadow, RTGI_BATCH_ANY_HIT};
				for (uint32_t k = 0; k < S; ++k) {
					const size_t at = (size_t)k * n;
					if (next[k]) both[n_both++] = {p.d_rays[1 - cur] + at, hits[1 - hcur] + at, next[k], RTGI_BATCH_CLOSEST_HIT};
					p.rays += (uint64_t)m[k] + next[k];
					more += next[k] != 0;
				}
				if (n_both) PART_RT(p, rtgi_hip_trace_batches_device(p.h, both, n_both, st));
				for (uint32_t k = 0; k < S; ++k) {
					const size_t at = (size_t)k * n;
					if (m[k] == 0) continue;
					PART_RT(p, rtgi_hip_shade_local_brdf_device(p.h, p.d_rays[cur] + at, hits[hcur] + at, p.d_occluded + at_shadow[k], m[k], f.light_pos, f.light_col,
					                                            p.d_direct + at * 4, st));
					PART_RT(p, rtgi_hip_path_add_device(p.h, p.d_direct + at * 4, p.d_throughput[cur] + at * 4, p.d_slot[cur] + at, m[k], p.d_radiance + at * 4, st));
					if (next[k])
						PART_RT(p, rtgi_hip_path_extend_device(p.h, p.d_albedo + at * 4, p.d_parent + at, next[k], p.d_throughput[cur] + at * 4, p.d_slot[cur] + at,
						                                       p.d_throughput[1 - cur] + at * 4, p.d_slot[1 - cur] + at, st));
					m[k] = next[k];
				}
				if (more == 0) break;
				cur = 1 - cur;
				hcur = 1 - hcur;
			}
		}
		for (uint32_t k = 0; k < S; ++k)  // in the order of the samples: the framebuffer keeps a running mean
			PART_RT(p, rtgi_hip_framebuffer_add_device(p.h, fb, p.d_radiance + (size_t)k * n * 4, n, st));
	}
	if (by_tiles) {
		PART_HIP(p, hipMemsetAsync(p.d_full, 0, (size_t)n_full * 16, st));
		PART_RT(p, rtgi_hip_framebuffer_scatter_device(p.h, p.d_fb, d_pixels, n, p.d_full, st));
	}
	if (world > 1 && f.sharding == RTGI_SHARD_SAMPLES) PART_RT(p, rtgi_hip_framebuffer_sums_device(p.h, p.d_full, n_full, 1, st));
	PART_HIP(p, hipStreamSynchronize(st));
	p.t_traced = now_ms();
	return RTGI_OK;
}

int check_frame(rtgi_hip_renderer *r, const rtgi_frame *f) {
	if (!f) return fail(r, RTGI_ERR_INVALID, "render: null frame");
	if (f->cam.w <= 0 || f->cam.h <= 0 || (uint64_t)f->cam.w * (uint64_t)f->cam.h > 0x7fffffffull)
		return fail(r, RTGI_ERR_INVALID, "render: bad resolution");
	if (f->sppx == 0) return fail(r, RTGI_ERR_INVALID, "render: sppx must be positive");
	if (f->algo > RTGI_ALGO_PATH) return fail(r, RTGI_ERR_INVALID, "render: no such algorithm");
	if (f->sharding > RTGI_SHARD_SAMPLES) return fail(r, RTGI_ERR_INVALID, "render: no such sharding");
	if (f->bounces > 64) return fail(r, RTGI_ERR_INVALID, "render: at most 64 bounces");
	return RTGI_OK;
}

// Runs fn(part index) for every part, on a thread of its own when there are several; a part's failure is kept in it.
// fn must not enter a collective: a part that fails early would leave its peers waiting inside one. (The framebuffer
// exchange is issued for all parts by one thread, see exchange_in_process.)
template <class F>
int for_each_part(rtgi_hip_renderer *r, F fn) {
	const int n = (int)r->parts.size();
	for (Part &p : r->parts) p.rc = RTGI_OK, p.error.clear();
	// (nothing may leave a thread or this library as an exception: a part that runs out of memory fails like any other)
	auto guarded = [r, &fn](int i) {
		try {
			fn(i);
		} catch (const std::exception &e) {
			fail(r->parts[(size_t)i], RTGI_ERR_NOMEM, std::string("exception: ") + e.what());
		} catch (...) {
			fail(r->parts[(size_t)i], RTGI_ERR_NOMEM, "exception");
		}
	};
	std::vector<std::thread> threads;
	for (int i = 1; i < n; ++i) {
		try {
			threads.emplace_back([&guarded, i] { guarded(i); });
		} catch (const std::exception &) {
			guarded(i);  // no thread to be had: this part takes its turn on the calling thread
		}
	}
	guarded(0);
	for (std::thread &t : threads) t.join();
	for (int i = 0; i < n; ++i)
		if (r->parts[i].rc < 0) return fail(r, r->parts[i].rc, "GPU " + std::to_string(r->parts[i].device) + " (part " + std::to_string(i) + "): " + r->parts[i].error);
	return RTGI_OK;
}

// After a collective went wrong nothing can be said about the communicators' state: they are given up (ncclCommAbort
// also ends collectives that are still waiting for ranks that never came) and the next frame starts new ones.
void abort_comms(rtgi_hip_renderer *r) {
	for (Part &p : r->parts)
		if (p.comm) {
			(void)ncclCommAbort(p.comm);
			p.comm = nullptr;
		}
	r->comms_ready = false;
}

// All GPUs in this process: the all-reduce of every part is issued by ONE thread inside an RCCL group, after everything
// that can be checked has been checked, so either all parts enter the collective or none does.
int exchange_in_process(rtgi_hip_renderer *r, uint32_t n_full) {
	for (const Part &p : r->parts)
		if (!p.comm || !p.d_full || !p.stream) return fail(r, RTGI_ERR_DEVICE, "framebuffer exchange: a part has no communicator, buffer or stream");
	ncclResult_t first_bad = ncclGroupStart();
	if (first_bad != ncclSuccess) return fail(r, RTGI_ERR_DEVICE, std::string("ncclGroupStart: ") + ncclGetErrorString(first_bad));
	for (Part &p : r->parts) {
		first_bad = ncclAllReduce(p.d_full, p.d_full, (size_t)n_full * 4, ncclFloat, ncclSum, p.comm, p.stream);
		if (first_bad != ncclSuccess) break;
	}
	const ncclResult_t ended = ncclGroupEnd();
	if (first_bad != ncclSuccess || ended != ncclSuccess) {
		const std::string what = std::string("ncclAllReduce: ") + ncclGetErrorString(first_bad != ncclSuccess ? first_bad : ended);
		abort_comms(r);
		return fail(r, RTGI_ERR_DEVICE, what);
	}
	// every part has entered: all of them are waited for, whatever the others report
	std::string bad;
	for (Part &p : r->parts) {
		hipError_t e = hipSetDevice(p.device);
		if (e == hipSuccess) e = hipStreamSynchronize(p.stream);
		if (e != hipSuccess && bad.empty()) bad = "GPU " + std::to_string(p.device) + " after ncclAllReduce: " + hipGetErrorString(e);
	}
	if (!bad.empty()) {
		abort_comms(r);
		return fail(r, RTGI_ERR_DEVICE, bad);
	}
	return RTGI_OK;
}

// One process per GPU: render() is a collective call, and a rank must not stay away from a collective its peers
// enter. Every rank therefore first takes part in a one-word all-reduce (max) of "my part failed", whatever happened
// to its own part; only if nobody failed do all of them go on to the framebuffer. local_rc: this rank's own result.
int exchange_by_rank(rtgi_hip_renderer *r, uint32_t n_full, int local_rc) {
	Part &p = r->parts[0];
	if (!p.comm || !p.d_status) return fail(r, RTGI_ERR_DEVICE, "the communicator was given up after an earlier failure: create the renderer again on all ranks");
	const std::string local_error = r->error;
	int32_t status = local_rc < 0 ? 1 : 0;
	auto lost = [&](const std::string &what) {
		abort_comms(r);
		return fail(r, RTGI_ERR_DEVICE, what + (local_rc < 0 ? " (after: " + local_error + ")" : std::string()));
	};
	hipError_t e = hipSetDevice(p.device);
	if (e == hipSuccess) e = hipMemcpyAsync(p.d_status, &status, sizeof(status), hipMemcpyHostToDevice, p.stream);
	if (e != hipSuccess) return lost(std::string("status exchange: ") + hipGetErrorString(e));
	ncclResult_t ne = ncclAllReduce(p.d_status, p.d_status, 1, ncclInt32, ncclMax, p.comm, p.stream);
	if (ne != ncclSuccess) return lost(std::string("status exchange, ncclAllReduce: ") + ncclGetErrorString(ne));
	e = hipMemcpyAsync(&status, p.d_status, sizeof(status), hipMemcpyDeviceToHost, p.stream);
	if (e == hipSuccess) e = hipStreamSynchronize(p.stream);
	if (e != hipSuccess) return lost(std::string("status exchange: ") + hipGetErrorString(e));
	if (local_rc < 0) return local_rc;  // (r->error still has this rank's own text)
	if (status != 0) return fail(r, RTGI_ERR_DEVICE, "another rank failed to render its part of the frame");
	if (!p.d_full) return lost("framebuffer exchange: no buffer");
	ne = ncclAllReduce(p.d_full, p.d_full, (size_t)n_full * 4, ncclFloat, ncclSum, p.comm, p.stream);
	if (ne != ncclSuccess) return lost(std::string("ncclAllReduce: ") + ncclGetErrorString(ne));
	e = hipStreamSynchronize(p.stream);
	if (e != hipSuccess) return lost(std::string("after ncclAllReduce: ") + hipGetErrorString(e));
	return RTGI_OK;
}

}  // namespace

extern "C" {

rtgi_hip_renderer *rtgi_hip_renderer_create(const int *devices, int n, int layout, int esc) {
	if (!devices || n < 1 || n > 64) {
		g_create_error = "renderer_create: between 1 and 64 devices";
		return nullptr;
	}
	rtgi_hip_renderer *r = new (std::nothrow) rtgi_hip_renderer;
	if (!r) {
		g_create_error = "out of memory";
		return nullptr;
	}
	try {
		r->parts.resize((size_t)n);
	} catch (const std::exception &) {
		delete r;
		g_create_error = "out of memory";
		return nullptr;
	}
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < i; ++j)
			if (devices[i] == devices[j]) r->distinct = false;
	// (the caller's current GPU is the caller's business: it is put back on every way out)
	struct KeepDevice {
		int was = -1;
		KeepDevice() {
			if (hipGetDevice(&was) != hipSuccess) was = -1, (void)hipGetLastError();
		}
		~KeepDevice() {
			if (was >= 0) (void)hipSetDevice(was);
		}
	} keep_device;
	for (int i = 0; i < n; ++i) {
		Part &p = r->parts[(size_t)i];
		p.device = devices[i];
		p.h = rtgi_hip_create(devices[i], layout, esc);
		if (!p.h) {
			g_create_error = rtgi_hip_last_error(nullptr);
			rtgi_hip_renderer_destroy(r);
			return nullptr;
		}
		hipError_t e = hipSetDevice(p.device);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking);
		if (e != hipSuccess) {
			g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
			rtgi_hip_renderer_destroy(r);
			return nullptr;
		}
	}
	return r;
}

int rtgi_hip_renderer_unique_id(void *out, size_t bytes) {
	if (!out || bytes < sizeof(ncclUniqueId)) return RTGI_ERR_INVALID;
	ncclUniqueId id;
	const ncclResult_t e = ncclGetUniqueId(&id);
	if (e != ncclSuccess) {
		g_create_error = std::string("ncclGetUniqueId: ") + ncclGetErrorString(e);
		return RTGI_ERR_DEVICE;
	}
	std::memset(out, 0, bytes);
	std::memcpy(out, &id, sizeof(id));
	return RTGI_OK;
}

rtgi_hip_renderer *rtgi_hip_renderer_create_rank(int device, int layout, int esc, int rank, int world, const void *unique_id) {
	if (world < 1 || rank < 0 || rank >= world || !unique_id) {
		g_create_error = "renderer_create_rank: rank must be in [0, world) and the id of rtgi_hip_renderer_unique_id given";
		return nullptr;
	}
	rtgi_hip_renderer *r = rtgi_hip_renderer_create(&device, 1, layout, esc);
	if (!r) return nullptr;
	r->world_rank = rank;
	r->world_size = world;
	ncclUniqueId id;
	std::memcpy(&id, unique_id, sizeof(id));
	Part &p = r->parts[0];
	hipError_t he = hipSetDevice(p.device);
	const ncclResult_t e = he == hipSuccess ? ncclCommInitRank(&p.comm, world, id, rank) : ncclUnhandledCudaError;
	if (e != ncclSuccess) {
		g_create_error = std::string("ncclCommInitRank: ") + ncclGetErrorString(e);
		p.comm = nullptr;
		rtgi_hip_renderer_destroy(r);
		return nullptr;
	}
	he = hipMalloc((void **)&p.d_status, 256);
	if (he != hipSuccess) {
		g_create_error = std::string("hipMalloc: ") + hipGetErrorString(he);
		rtgi_hip_renderer_destroy(r);
		return nullptr;
	}
	r->comms_ready = true;
	return r;
}

void rtgi_hip_renderer_destroy(rtgi_hip_renderer *r) {
	if (!r) return;
	for (Part &p : r->parts) {
		free_buffers(p);
		if (p.d_status) (void)hipFree(p.d_status);
		if (p.comm) (void)ncclCommDestroy(p.comm);
		if (p.stream && hipSetDevice(p.device) == hipSuccess) (void)hipStreamDestroy(p.stream);
		if (p.h) rtgi_hip_destroy(p.h);
	}
	delete r;
}

int rtgi_hip_renderer_device_count(const rtgi_hip_renderer *r) { return r ? (int)r->parts.size() : 0; }

rtgi_hip_tracer *rtgi_hip_renderer_handle(rtgi_hip_renderer *r, int i) {
	return (r && i >= 0 && i < (int)r->parts.size()) ? r->parts[(size_t)i].h : nullptr;
}

int rtgi_hip_renderer_interprete(rtgi_hip_renderer *r, const char *command, const char *args) {
	if (!r) return RTGI_ERR_INVALID;
	int result = 0;
	for (size_t i = 0; i < r->parts.size(); ++i) {
		// ("bvh statistics" and "bvh export" print or write once)
		const std::string a = args ? args : "";
		if (i > 0 && command && std::string(command) == "bvh" && (a.find("statistics") != std::string::npos || a.find("export") != std::string::npos)) continue;
		const int rc = rtgi_hip_interprete(r->parts[i].h, command, args);
		if (rc < 0) return fail(r, rc, rtgi_hip_last_error(r->parts[i].h));
		if (i == 0) result = rc;
	}
	return result;
}

int rtgi_hip_renderer_build(rtgi_hip_renderer *r, const void *vertices, size_t vertex_stride, uint32_t n_vertices, const void *triangles,
                            size_t tri_stride, uint32_t n_triangles, const float *albedo_rgb, uint32_t n_materials) {
	if (!r) return RTGI_ERR_INVALID;
	if (!vertices || !triangles) return fail(r, RTGI_ERR_INVALID, "there is no scene data to work with");
	if (vertex_stride < 24 || tri_stride < 16) return fail(r, RTGI_ERR_INVALID, "renderer_build: vertex records need position and normal, triangle records a material id");
	// one tree for all GPUs: built and flattened once, uploaded to each (rtgi_hip_build_shared); then every part
	// derives its shading tables
	std::vector<rtgi_hip_tracer *> handles;
	for (Part &p : r->parts) handles.push_back(p.h);
	const int brc = rtgi_hip_build_shared(handles.data(), (uint32_t)handles.size(), vertices, vertex_stride, n_vertices, triangles, tri_stride, n_triangles);
	if (brc < 0) return fail(r, brc, rtgi_hip_last_error(handles[0]));
	if (!albedo_rgb || !n_materials) return RTGI_OK;
	return for_each_part(r, [&](int i) {
		Part &p = r->parts[(size_t)i];
		const int rc = rtgi_hip_set_shading(p.h, (const char *)vertices + 12, vertex_stride, n_vertices, triangles, tri_stride, n_triangles, albedo_rgb,
		                                    n_materials);
		if (rc < 0) fail(p, rc, rtgi_hip_last_error(p.h));
	});
}

int rtgi_hip_renderer_set_materials(rtgi_hip_renderer *r, const rtgi_material *materials, uint32_t n_materials) {
	if (!r) return RTGI_ERR_INVALID;
	for (Part &p : r->parts) {
		const int rc = rtgi_hip_set_materials(p.h, materials, n_materials);
		if (rc < 0) return fail(r, rc, rtgi_hip_last_error(p.h));
	}
	return RTGI_OK;
}

int rtgi_hip_renderer_set_textures(rtgi_hip_renderer *r, const void *vertex_tc, size_t tc_stride, uint32_t n_vertices, const void *triangles,
                                   size_t tri_stride, uint32_t n_triangles, const rtgi_texture *textures, uint32_t n_textures,
                                   const int32_t *material_texture, uint32_t n_materials) {
	if (!r) return RTGI_ERR_INVALID;
	for (Part &p : r->parts) {
		const int rc = rtgi_hip_set_textures(p.h, vertex_tc, tc_stride, n_vertices, triangles, tri_stride, n_triangles, textures, n_textures,
		                                     material_texture, n_materials);
		if (rc < 0) return fail(r, rc, rtgi_hip_last_error(p.h));
	}
	return RTGI_OK;
}

namespace {
int render_frame(rtgi_hip_renderer *r, const rtgi_frame *frame, float *framebuffer, uint8_t *rgb8, rtgi_frame_stats *stats);
}
int rtgi_hip_renderer_render(rtgi_hip_renderer *r, const rtgi_frame *frame, float *framebuffer, uint8_t *rgb8, rtgi_frame_stats *stats) {
	if (!r) return RTGI_ERR_INVALID;
	try {
		return render_frame(r, frame, framebuffer, rgb8, stats);
	} catch (const std::exception &e) {
		return fail(r, RTGI_ERR_NOMEM, std::string("render: ") + e.what());
	} catch (...) {
		return fail(r, RTGI_ERR_NOMEM, "render: exception");
	}
}
namespace {
int render_frame(rtgi_hip_renderer *r, const rtgi_frame *frame, float *framebuffer, uint8_t *rgb8, rtgi_frame_stats *stats) {
	if (int rc = check_frame(r, frame)) return rc;
	const rtgi_frame f = *frame;
	const bool by_rank = r->world_size > 0;
	const int world = by_rank ? r->world_size : (int)r->parts.size();
	const uint32_t n_full = (uint32_t)f.cam.w * (uint32_t)f.cam.h;
	const double t0 = now_ms();
	// communicators for the framebuffer exchange, once: one rank per GPU, all in this process
	if (!by_rank && world > 1 && r->distinct && !r->comms_ready) {
		std::vector<ncclComm_t> comms((size_t)world);
		std::vector<int> devs((size_t)world);
		for (int i = 0; i < world; ++i) devs[(size_t)i] = r->parts[(size_t)i].device;
		const ncclResult_t e = ncclCommInitAll(comms.data(), world, devs.data());
		if (e != ncclSuccess) return fail(r, RTGI_ERR_DEVICE, std::string("ncclCommInitAll: ") + ncclGetErrorString(e));
		for (int i = 0; i < world; ++i) r->parts[(size_t)i].comm = comms[(size_t)i];
		r->comms_ready = true;
	}
	int rc = for_each_part(r, [&](int i) { render_part(r->parts[(size_t)i], f, by_rank ? r->world_rank : i, world); });
	// All parts of this process have come back. In-process: if any failed, none enters the exchange. By rank: the
	// other processes cannot know yet, see exchange_by_rank.
	if (rc && !by_rank) return rc;
	double t_traced = t0;
	for (const Part &p : r->parts) t_traced = std::max(t_traced, p.t_traced);
	// ---- the one exchange of a frame: framebuffers as sums and counts (tiles: every pixel has one owner, the
	// others hold zeros, and x + 0 = x keeps the owner's value to the bit) ----
	bool used_rccl = false;
	if (world > 1 || by_rank) {
		if (by_rank) {
			if ((rc = exchange_by_rank(r, n_full, rc)) != RTGI_OK) return rc;
			used_rccl = true;
		} else if (r->distinct) {
			if ((rc = exchange_in_process(r, n_full)) != RTGI_OK) return rc;
			used_rccl = true;
		} else {
			// handles that share a GPU (a rehearsal of the multi-GPU path): the same sum by a kernel on that GPU
			Part &p0 = r->parts[0];
			for (int i = 1; i < world; ++i) {
				if (r->parts[(size_t)i].device != p0.device) return fail(r, RTGI_ERR_INVALID, "a GPU may be named twice only if all parts share one GPU");
				const int arc = rtgi_hip_framebuffer_accumulate_device(p0.h, p0.d_full, r->parts[(size_t)i].d_full, n_full, p0.stream);
				if (arc < 0) return fail(r, arc, rtgi_hip_last_error(p0.h));
			}
		}
		if (world > 1 && f.sharding == RTGI_SHARD_SAMPLES) {
			const int src = rtgi_hip_framebuffer_sums_device(r->parts[0].h, r->parts[0].d_full, n_full, 0, r->parts[0].stream);
			if (src < 0) return fail(r, src, rtgi_hip_last_error(r->parts[0].h));
		}
	}
	Part &p0 = r->parts[0];
	if (hipSetDevice(p0.device) != hipSuccess) return fail(r, RTGI_ERR_DEVICE, "hipSetDevice");
	hipError_t e = hipStreamSynchronize(p0.stream);
	const double t_reduced = now_ms();
	if (e == hipSuccess && framebuffer) e = hipMemcpyAsync(framebuffer, p0.d_full, (size_t)n_full * 16, hipMemcpyDeviceToHost, p0.stream);
	if (e == hipSuccess && rgb8) {
		const int qrc = rtgi_hip_framebuffer_rgb8_device(p0.h, p0.d_full, (uint32_t)f.cam.w, (uint32_t)f.cam.h, p0.d_rgb, p0.stream);
		if (qrc < 0) return fail(r, qrc, rtgi_hip_last_error(p0.h));
		e = hipMemcpyAsync(rgb8, p0.d_rgb, (size_t)n_full * 3, hipMemcpyDeviceToHost, p0.stream);
	}
	if (e == hipSuccess) e = hipStreamSynchronize(p0.stream);
	if (e != hipSuccess) return fail(r, RTGI_ERR_DEVICE, std::string("reading the frame back: ") + hipGetErrorString(e));
	if (stats) {
		stats->ms_total = now_ms() - t0;
		stats->ms_trace = t_traced - t0;
		stats->ms_reduce = (world > 1 || by_rank) ? t_reduced - t_traced : 0.0;
		stats->rays = 0;
		for (const Part &p : r->parts) stats->rays += p.rays;
		stats->n_devices = world;
		stats->used_rccl = used_rccl ? 1 : 0;
	}
	return RTGI_OK;
}
}  // namespace

int rtgi_hip_renderer_rt_bench(rtgi_hip_renderer *r, const rtgi_frame *frame, uint64_t *n_rays, double *ms) {
	if (!r) return RTGI_ERR_INVALID;
	if (int rc = check_frame(r, frame)) return rc;
	Part &p = r->parts[0];
	const size_t n = (size_t)frame->cam.w * (size_t)frame->cam.h;
	// everything below lives on part 0's GPU: buffers, launches and the waits that bracket the timing
	int was = -1;
	if (hipGetDevice(&was) != hipSuccess) was = -1;
	hipError_t e = hipSetDevice(p.device);
	rtgi_ray *d_rays = nullptr;
	rtgi_hit *d_hits = nullptr;
	if (e == hipSuccess) e = hipMalloc((void **)&d_rays, n * sizeof(rtgi_ray));
	if (e == hipSuccess) e = hipMalloc((void **)&d_hits, n * sizeof(rtgi_hit));
	int rc = RTGI_OK;
	double elapsed = 0.0;
	if (e == hipSuccess) rc = rtgi_hip_camera_rays_device(p.h, &frame->cam, nullptr, d_rays, p.stream);
	if (e == hipSuccess && rc == RTGI_OK) e = hipStreamSynchronize(p.stream);
	if (e == hipSuccess && rc == RTGI_OK) {
		const double t0 = now_ms();
		rc = rtgi_hip_closest_hit_device(p.h, d_rays, (uint32_t)n, d_hits, nullptr, p.stream);
		if (rc == RTGI_OK) e = hipStreamSynchronize(p.stream);
		elapsed = now_ms() - t0;
	}
	if (d_rays) (void)hipFree(d_rays);
	if (d_hits) (void)hipFree(d_hits);
	if (was >= 0) (void)hipSetDevice(was);
	if (rc < 0) return fail(r, rc, rtgi_hip_last_error(p.h));
	if (e != hipSuccess) return fail(r, RTGI_ERR_DEVICE, std::string("rt_bench: ") + hipGetErrorString(e));
	if (n_rays) *n_rays = n;
	if (ms) *ms = elapsed;
	return RTGI_OK;
}

int rtgi_hip_frame_share(uint32_t sharding, int w, int h, uint32_t tile, uint32_t sppx, int world, int rank, uint32_t *out, uint32_t *n) {
	if (!n || world < 1 || rank < 0 || rank >= world || sharding > RTGI_SHARD_SAMPLES) return RTGI_ERR_INVALID;
	if (sharding == RTGI_SHARD_SAMPLES) {
		uint32_t count = 0, first, step;
		sample_share(sharding, world, rank, first, step);
		for (uint32_t sample = first; sample < sppx; sample += step, ++count)
			if (out) out[count] = sample;
		*n = count;
		return RTGI_OK;
	}
	if (w < 1 || h < 1) return RTGI_ERR_INVALID;
	std::vector<uint32_t> pixels;
	tile_pixels(w, h, tile ? tile : 64u, world, rank, pixels);
	*n = (uint32_t)pixels.size();
	if (out) std::copy(pixels.begin(), pixels.end(), out);
	return RTGI_OK;
}

const char *rtgi_hip_renderer_last_error(const rtgi_hip_renderer *r) { return r ? r->error.c_str() : g_create_error.c_str(); }

}  // extern "C"
