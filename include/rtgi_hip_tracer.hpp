// rtgi_hip_tracer.hpp -- C++ host adapter: the GPU tracer behind the renderer's own tracer interfaces.
//
// Two ways to use this header:
//  * inside the rtgi tree: define RTGI_HIP_USE_LIBGI before including it. It then includes libgi/rt.h,
//    libgi/scene.h, libgi/wavefront-rt.h and derives from the renderer's individual_ray_tracer and
//    wf::batch_ray_tracer, so that `scene.use(new hip_bvh_batch_tracer(...))` works unchanged;
//  * stand-alone (this repository's tests and tools): it declares mirror types with the same names, member
//    order and sizes as the renderer's (citations are file:line of "Assignment 3/rtgi-2021-a03-ref/").
//
// Everything below the interface is one C handle from rtgi_hip.h; there is no CPU traversal in here.
#pragma once

#include <cfloat>
#include <cstdint>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "rtgi_hip.h"

#ifdef RTGI_HIP_USE_LIBGI
#include "libgi/rt.h"
#include "libgi/scene.h"
#include "libgi/wavefront-rt.h"
namespace rtgi_hip {
using ::individual_ray_tracer;
using ::ray;
using ::ray_tracer;
using ::scene;
using ::triangle;
using ::triangle_intersection;
using ::vertex;
namespace wf = ::wf;
}  // namespace rtgi_hip
#else
namespace rtgi_hip {

struct vec2 {
	float x = 0, y = 0;
};
struct vec3 {
	float x = 0, y = 0, z = 0;
	vec3() {}
	vec3(float x, float y, float z) : x(x), y(y), z(z) {}
};

struct ray {  // libgi/rt.h:20-27
	static constexpr float eps = 1e-4f;
	vec3 o, d, id;
	float t_min = eps, t_max = FLT_MAX;
	ray(const vec3 &o, const vec3 &d) : o(o), d(d), id(1.0f / d.x, 1.0f / d.y, 1.0f / d.z) {}
	ray() {}
	void length_exclusive(float len) { t_max = len - eps; }
};

struct vertex {  // libgi/rt.h:29-33
	vec3 pos;
	vec3 norm;
	vec2 tc;
};

struct triangle {  // libgi/rt.h:35-38
	uint32_t a, b, c;
	uint32_t material_id;
};

struct triangle_intersection {  // libgi/rt.h:40-62
	typedef unsigned int uint;
	float t, beta, gamma;
	uint ref;
	triangle_intersection() : t(FLT_MAX), ref(0) {}
	triangle_intersection(uint t) : t(FLT_MAX), ref(t) {}
	bool valid() const { return t != FLT_MAX; }
	void reset() {
		t = FLT_MAX;
		ref = 0;
	}
};

struct scene {  // the part of libgi/scene.h:74-113 a tracer touches
	std::vector<vertex> vertices;
	std::vector<triangle> triangles;
};

class ray_tracer {  // libgi/rt.h:91-98
protected:
	rtgi_hip::scene *scene = nullptr;

public:
	virtual void build(rtgi_hip::scene *) = 0;
	virtual bool interprete(const std::string &, std::istringstream &) { return false; }
	virtual ~ray_tracer() {}
};

class individual_ray_tracer : public ray_tracer {  // libgi/rt.h:100-107
public:
	virtual triangle_intersection closest_hit(const ray &) = 0;
	virtual bool any_hit(const ray &) = 0;
};

namespace wf {
struct batch_ray_tracer : public ray_tracer {  // libgi/wavefront-rt.h:10-16
	ray *rays = nullptr;
	triangle_intersection *intersections = nullptr;
	virtual void compute_closest_hit() = 0;
	virtual void compute_any_hit() = 0;
};
}  // namespace wf

}  // namespace rtgi_hip
#endif

namespace rtgi_hip {

static_assert(sizeof(ray) == 44, "ray must keep the renderer's layout");
static_assert(sizeof(vertex) == 32 && sizeof(triangle) == 16, "scene records must keep the renderer's layout");
static_assert(sizeof(triangle_intersection) == sizeof(rtgi_hit), "hit records are copied verbatim");

// Owner of one C handle; shared by the two adapter classes below.
class hip_bvh_core {
public:
	// `raytracer hip [indexed] [esc] [device N]`: same tags as `raytracer bbvh` (driver/interaction.cpp:182-200)
	hip_bvh_core(bool indexed, bool esc, int device = 0) : indexed(indexed) {
		h = rtgi_hip_create(device, indexed ? RTGI_LAYOUT_INDEXED : RTGI_LAYOUT_FLAT, esc ? 1 : 0);
		if (!h) throw std::runtime_error(std::string("rtgi_hip_create: ") + rtgi_hip_last_error(nullptr));
	}
	~hip_bvh_core() { rtgi_hip_destroy(h); }
	hip_bvh_core(const hip_bvh_core &) = delete;
	hip_bvh_core &operator=(const hip_bvh_core &) = delete;

	// build(scene*), rt/bbvh-base/bvh2.cpp:27-63. For the flat layout the scene's triangle list is replaced by
	// the permuted one, exactly like commit_shuffled_triangles (bvh2.cpp:172-180) does.
	template <class Scene>
	void build(Scene *s) {
		check(rtgi_hip_build(h, &s->vertices[0].pos, sizeof(s->vertices[0]), (uint32_t)s->vertices.size(), &s->triangles[0].a,
		                     sizeof(s->triangles[0]), (uint32_t)s->triangles.size()));
		permute(s);
	}

	// Hands the GPU shading stages (rtgi_hip_shade_*_device) their part of the scene: vertex normals, material
	// ids and one albedo per material (3 floats each, material::albedo of libgi/material.h:54). Call between
	// rtgi_hip_build and the permutation of scene->triangles, i.e. use build(s, albedo, n) below.
	template <class Scene>
	void build(Scene *s, const float *albedo_rgb, uint32_t n_materials) {
		check(rtgi_hip_build(h, &s->vertices[0].pos, sizeof(s->vertices[0]), (uint32_t)s->vertices.size(), &s->triangles[0].a,
		                     sizeof(s->triangles[0]), (uint32_t)s->triangles.size()));
		check(rtgi_hip_set_shading(h, &s->vertices[0].norm, sizeof(s->vertices[0]), (uint32_t)s->vertices.size(), &s->triangles[0].a,
		                           sizeof(s->triangles[0]), (uint32_t)s->triangles.size(), albedo_rgb, n_materials));
		permute(s);
	}

	// interprete(command, in), bvh2.cpp:503-555. Errors are printed like the renderer's error() macro does
	// (bvh2.cpp:15) and the command counts as consumed.
	bool interprete(const std::string &command, std::istringstream &in) {
		std::string rest;
		std::getline(in, rest);
		int rc = rtgi_hip_interprete(h, command.c_str(), rest.c_str());
		if (rc < 0) {
			fprintf(stderr, "command  (%s): %s\n", command.c_str(), rtgi_hip_last_error(h));
			return true;
		}
		return rc == 1;
	}

	void closest_hit(const ray *rays, size_t n, triangle_intersection *out) {
		pack(rays, n);
		check(rtgi_hip_closest_hit(h, staged.data(), (uint32_t)n, reinterpret_cast<rtgi_hit *>(out)));
	}
	// compute_any_hit stores the bool through triangle_intersection(uint): t = FLT_MAX, ref = 0 or 1
	// (libgi/wavefront-rt.h:79, libgi/rt.h:46)
	void any_hit(const ray *rays, size_t n, triangle_intersection *out) {
		pack(rays, n);
		occluded.resize(n);
		check(rtgi_hip_any_hit(h, staged.data(), (uint32_t)n, occluded.data()));
#pragma omp parallel for schedule(static) if (n > 65536)
		for (size_t i = 0; i < n; ++i) out[i] = triangle_intersection((unsigned int)occluded[i]);
	}
	rtgi_hip_tracer *handle() { return h; }

private:
	template <class Scene>
	void permute(Scene *s) {
		if (indexed) return;
		std::vector<uint32_t> order(rtgi_hip_num_index(h));
		check(rtgi_hip_get_triangle_order(h, order.data()));
		auto shuffled = s->triangles;
		for (size_t i = 0; i < order.size(); ++i) shuffled[i] = s->triangles[order[i]];
		s->triangles = std::move(shuffled);
	}
	void check(int rc) {
		if (rc < 0) throw std::runtime_error(std::string("rtgi_hip: ") + rtgi_hip_last_error(h));
	}
	void pack(const ray *rays, size_t n) {  // drop the derived member id: 44-byte rays -> 32-byte rays
		staged.resize(n);  // page-locked memory: the GPU's copy engine reads it directly
#pragma omp parallel for schedule(static) if (n > 65536)
		for (size_t i = 0; i < n; ++i) {
			const ray &r = rays[i];
			staged[i] = rtgi_ray{{r.o.x, r.o.y, r.o.z}, {r.d.x, r.d.y, r.d.z}, r.t_min, r.t_max};
		}
	}
	rtgi_hip_tracer *h = nullptr;
	bool indexed;
	// std::vector-like buffer in page-locked memory (falls back to ordinary memory if that cannot be had)
	template <class T>
	struct pinned_buffer {
		T *p = nullptr;
		size_t cap = 0, n = 0;
		bool locked = false;
		~pinned_buffer() { release(); }
		void release() {
			if (p && locked) rtgi_hip_host_free(p);
			else free(p);
			p = nullptr, cap = 0;
		}
		void resize(size_t count) {
			if (count > cap) {
				release();
				cap = count + count / 8 + 64;
				p = static_cast<T *>(rtgi_hip_host_alloc(cap * sizeof(T)));
				locked = p != nullptr;
				if (!p) p = static_cast<T *>(malloc(cap * sizeof(T)));
				if (!p) throw std::bad_alloc();
			}
			n = count;
		}
		T *data() { return p; }
		T &operator[](size_t i) { return p[i]; }
	};
	pinned_buffer<rtgi_ray> staged;
	pinned_buffer<uint8_t> occluded;
};

// The batch interface is the one the GPU is built for: one launch per compute_*() call over n_rays rays.
// The renderer sizes its ray buffers to the resolution (libgi/wavefront-rt.h:24-50); set_batch() tells the
// tracer which buffers and how many rays the next compute_*() calls cover.
class hip_bvh_batch_tracer : public wf::batch_ray_tracer {
public:
	hip_bvh_batch_tracer(bool indexed = false, bool esc = false, int device = 0) : core(indexed, esc, device) {}
	// The intersection buffer is page-locked while it is the current one (it lives until the next `resolution`
	// command, libgi/wavefront-rt.h:36-43), so that hits arrive in it without a staging copy.
	void set_batch(ray *r, triangle_intersection *is, size_t n) {
		if (is != intersections || n != n_rays) {
			unlock_results();
			locked_results = is && n && rtgi_hip_host_register(is, n * sizeof(triangle_intersection)) == RTGI_OK;
		}
		rays = r;
		intersections = is;
		n_rays = n;
	}
	~hip_bvh_batch_tracer() { unlock_results(); }
	void build(rtgi_hip::scene *s) override {
		this->scene = s;
		core.build(s);
	}
	bool interprete(const std::string &command, std::istringstream &in) override { return core.interprete(command, in); }
	void compute_closest_hit() override { core.closest_hit(rays, n_rays, intersections); }
	void compute_any_hit() override { core.any_hit(rays, n_rays, intersections); }
	hip_bvh_core core;

private:
	void unlock_results() {
		if (locked_results) rtgi_hip_host_unregister(intersections);
		locked_results = false;
	}
	size_t n_rays = 0;
	bool locked_results = false;
};

// Per-ray interface for the renderer's recursive algorithms (gi/primary-hit.cpp:21,36,50). The renderer calls
// the tracer from all of its OpenMP threads at once (libgi/framebuffer.h:41-46); a GPU round trip per ray would
// serialise them, so calls that arrive while a batch is under way are collected and go out together as the next
// batch: with T host threads a round trip carries up to T rays. Correct for any number of threads (one thread
// simply gets one-ray batches), and slow by construction next to the batch tracer above -- algorithms that
// care about speed use that one.
class hip_bvh_individual_tracer : public individual_ray_tracer {
public:
	hip_bvh_individual_tracer(bool indexed = false, bool esc = false, int device = 0) : core(indexed, esc, device) {}
	void build(rtgi_hip::scene *s) override {
		this->scene = s;
		core.build(s);
	}
	bool interprete(const std::string &command, std::istringstream &in) override { return core.interprete(command, in); }
	triangle_intersection closest_hit(const ray &r) override {
		request q{&r, false, {}, false};
		submit(q);
		return q.result;
	}
	bool any_hit(const ray &r) override {
		request q{&r, true, {}, false};
		submit(q);
		return q.result.ref != 0;
	}
	hip_bvh_core core;

private:
	struct request {
		const ray *r;
		bool any;
		triangle_intersection result;
		bool done;
	};
	std::mutex guard;
	std::condition_variable served;
	std::vector<request *> waiting;
	bool batch_under_way = false;

	// The first caller that finds no batch under way serves everything that is waiting (its own request
	// included), then hands over; the others sleep until their request has been served.
	void submit(request &q) {
		std::unique_lock<std::mutex> lock(guard);
		waiting.push_back(&q);
		while (!q.done) {
			if (batch_under_way) {
				served.wait(lock);
				continue;
			}
			batch_under_way = true;
			std::vector<request *> batch;
			batch.swap(waiting);
			lock.unlock();
			std::exception_ptr failure;
			try {
				serve(batch);
			} catch (...) {
				failure = std::current_exception();
			}
			lock.lock();
			for (request *p : batch) p->done = true;
			batch_under_way = false;
			served.notify_all();
			if (failure) std::rethrow_exception(failure);
		}
	}
	void serve(const std::vector<request *> &batch) {
		std::vector<ray> rays;
		std::vector<triangle_intersection> out;
		for (int any = 0; any < 2; ++any) {
			rays.clear();
			for (const request *p : batch)
				if (p->any == (any != 0)) rays.push_back(*p->r);
			if (rays.empty()) continue;
			out.assign(rays.size(), triangle_intersection());
			if (any)
				core.any_hit(rays.data(), rays.size(), out.data());
			else
				core.closest_hit(rays.data(), rays.size(), out.data());
			size_t k = 0;
			for (request *p : batch)
				if (p->any == (any != 0)) p->result = out[k++];
		}
	}
};

}  // namespace rtgi_hip
