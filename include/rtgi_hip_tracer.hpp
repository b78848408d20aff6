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
#include "rtgi_hip_frame.h"

#ifdef RTGI_HIP_USE_LIBGI
#include "libgi/rt.h"
#include "libgi/material.h"
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
namespace cpu {
struct raydata {  // libgi/wavefront-rt.h:24-50 (there the renderer's `resolution` command re-sizes it through
	int w, h;     // rc->call_at_resolution_change; here the owner calls resize)
	ray *rays = nullptr;
	triangle_intersection *intersections = nullptr;
	raydata(int w, int h) : w(0), h(0) { resize(w, h); }
	raydata(const raydata &) = delete;
	raydata &operator=(const raydata &) = delete;
	~raydata() {
		delete[] rays;
		delete[] intersections;
	}
	void resize(int new_w, int new_h) {
		delete[] rays;
		delete[] intersections;
		rays = nullptr, intersections = nullptr;
		w = new_w, h = new_h;
		if (w > 0 && h > 0) {
			rays = new ray[(size_t)w * h];
			intersections = new triangle_intersection[(size_t)w * h];
		}
	}
};
struct batch_rt : public batch_ray_tracer {  // libgi/wavefront-rt.h:52-56
	raydata rd;
	batch_rt(int w = 0, int h = 0) : rd(w, h) {}
};
}  // namespace cpu
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
		require_scene(s);
		check(rtgi_hip_build(h, &s->vertices[0].pos, sizeof(s->vertices[0]), (uint32_t)s->vertices.size(), &s->triangles[0].a,
		                     sizeof(s->triangles[0]), (uint32_t)s->triangles.size()));
		permute(s);
	}

	// Hands the GPU shading stages (rtgi_hip_shade_*_device) their part of the scene: vertex normals, material
	// ids and one albedo per material (3 floats each, material::albedo of libgi/material.h:54). Call between
	// rtgi_hip_build and the permutation of scene->triangles, i.e. use build(s, albedo, n) below.
	template <class Scene>
	void build(Scene *s, const float *albedo_rgb, uint32_t n_materials) {
		require_scene(s);
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
		if (n > 0xffffffffull) throw std::length_error("rtgi_hip: more than 2^32 - 1 rays in one batch");
		pack(rays, n);
		check(rtgi_hip_closest_hit(h, staged.data(), (uint32_t)n, reinterpret_cast<rtgi_hit *>(out)));
	}
	// compute_any_hit stores the bool through triangle_intersection(uint): t = FLT_MAX, ref = 0 or 1
	// (libgi/wavefront-rt.h:79, libgi/rt.h:46)
	void any_hit(const ray *rays, size_t n, triangle_intersection *out) {
		if (n > 0xffffffffull) throw std::length_error("rtgi_hip: more than 2^32 - 1 rays in one batch");
		pack(rays, n);
		occluded.resize(n);
		check(rtgi_hip_any_hit(h, staged.data(), (uint32_t)n, occluded.data()));
#pragma omp parallel for schedule(static) if (n > 65536)
		for (size_t i = 0; i < n; ++i) out[i] = triangle_intersection((unsigned int)occluded[i]);
	}
	rtgi_hip_tracer *handle() { return h; }

private:
	template <class Scene>
	static void require_scene(Scene *s) {  // (the arrays' first elements are addressed below)
		if (!s || s->vertices.empty() || s->triangles.empty()) throw std::runtime_error("rtgi_hip: there is no scene data to work with");
	}
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

// The batch interface is the one the GPU is built for: one launch per compute_*() call. Like the renderer's
// batch_rt_adapter (libgi/wavefront-rt.h:58-88) it derives from wf::cpu::batch_rt, whose ray and intersection
// buffers follow the resolution by themselves (rc->call_at_resolution_change, libgi/wavefront-rt.h:36-43), and
// traces rd.w * rd.h rays from rd.rays into rd.intersections: an unmodified wavefront algorithm works as it is.
// A caller with buffers of its own points the base class's `rays` / `intersections` at them with set_batch().
// compute_*() without any rays to trace throws instead of silently doing nothing.
class hip_bvh_batch_tracer : public wf::cpu::batch_rt {
public:
	hip_bvh_batch_tracer(bool indexed = false, bool esc = false, int device = 0) : core(indexed, esc, device) {}
	void set_batch(ray *r, triangle_intersection *is, size_t n) {
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
	void compute_closest_hit() override {
		batch b = current();
		core.closest_hit(b.rays, b.n, b.out);
	}
	void compute_any_hit() override {
		batch b = current();
		core.any_hit(b.rays, b.n, b.out);
	}
	hip_bvh_core core;

private:
	struct batch {
		ray *rays;
		triangle_intersection *out;
		size_t n;
	};
	// the caller's buffers if set_batch named some, else the renderer-sized ones of the base class
	batch current() {
		batch b = rays ? batch{rays, intersections, n_rays} : batch{rd.rays, rd.intersections, rd.w > 0 && rd.h > 0 ? (size_t)rd.w * (size_t)rd.h : 0};
		if (!b.rays || !b.out || b.n == 0)
			throw std::logic_error("hip_bvh_batch_tracer: no ray batch to trace (resolution 0x0 and no set_batch)");
		// The intersection buffer is page-locked while it is the current one, so that hits arrive in it without a
		// staging copy.
		if (b.out != locked_ptr || b.n != locked_n) {
			unlock_results();
			if (rtgi_hip_host_register(b.out, b.n * sizeof(triangle_intersection)) == RTGI_OK) locked_ptr = b.out, locked_n = b.n;
		}
		return b;
	}
	void unlock_results() {
		if (locked_ptr) rtgi_hip_host_unregister(locked_ptr);
		locked_ptr = nullptr, locked_n = 0;
	}
	size_t n_rays = 0;
	triangle_intersection *locked_ptr = nullptr;
	size_t locked_n = 0;
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
		request q{&r, false, {}, false, nullptr};
		submit(q);
		return q.result;
	}
	bool any_hit(const ray &r) override {
		request q{&r, true, {}, false, nullptr};
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
		std::exception_ptr failure;  // the batch this request travelled in failed: every caller in it gets the exception
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
			for (request *p : batch) p->done = true, p->failure = failure;
			batch_under_way = false;
			served.notify_all();
		}
		if (q.failure) std::rethrow_exception(q.failure);
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

// The tracer for whole frames: `raytracer hip [indexed] [esc] [devices N]` as one object that holds a handle per
// GPU (include/rtgi_hip_frame.h) and, besides the BVH, the part of the scene the shading stages read. It has no
// per-ray or per-batch entry: the wavefront algorithms of rtgi_hip_algorithms.hpp hand it a whole frame
// (camera, light, sppx), and rays, hits and the framebuffer stay on the GPUs until the frame is done.
struct hip_material {  // struct material, libgi/material.h:54-61, as far as the GPU stages read it
	float albedo[3];
	float ior = 1.3f, roughness = 0.1f;
	uint32_t brdf = RTGI_BRDF_LAMBERT;
};
class hip_bvh_frame_tracer : public ray_tracer {
public:
	hip_bvh_frame_tracer(bool indexed = false, bool esc = false, std::vector<int> devices = {0}) : indexed(indexed) {
		r = rtgi_hip_renderer_create(devices.data(), (int)devices.size(), indexed ? RTGI_LAYOUT_INDEXED : RTGI_LAYOUT_FLAT, esc ? 1 : 0);
		if (!r) throw std::runtime_error(std::string("rtgi_hip_renderer_create: ") + rtgi_hip_renderer_last_error(nullptr));
	}
	~hip_bvh_frame_tracer() { rtgi_hip_renderer_destroy(r); }
	hip_bvh_frame_tracer(const hip_bvh_frame_tracer &) = delete;
	hip_bvh_frame_tracer &operator=(const hip_bvh_frame_tracer &) = delete;

	// ray_tracer::build. Materials are read by materials_of(scene) (inside the renderer: scene->materials; the
	// stand-alone mirror scene has none, so set_materials() is called by hand); lambertian grey until then.
	void build(rtgi_hip::scene *s) override {
		if (!s || s->vertices.empty() || s->triangles.empty()) throw std::runtime_error("rtgi_hip: there is no scene data to work with");
		this->scene = s;
		std::vector<hip_material> m = materials_of(s);
		std::vector<float> albedo;
		for (const hip_material &x : m) albedo.insert(albedo.end(), x.albedo, x.albedo + 3);
		uint32_t top = 0;
		for (const auto &t : s->triangles) top = t.material_id > top ? t.material_id : top;
		while (albedo.size() < 3 * ((size_t)top + 1)) albedo.push_back(0.5f);
		check(rtgi_hip_renderer_build(r, &s->vertices[0].pos, sizeof(s->vertices[0]), (uint32_t)s->vertices.size(), &s->triangles[0].a,
		                              sizeof(s->triangles[0]), (uint32_t)s->triangles.size(), albedo.data(), (uint32_t)(albedo.size() / 3)));
		n_materials = (uint32_t)(albedo.size() / 3);
		if (!m.empty()) set_materials(m);
		built_triangles.clear();
		if (!indexed) built_triangles.assign(s->triangles.begin(), s->triangles.end());  // the list as built: set_textures wants that one
		push_textures(s);
		if (!indexed) {  // commit_shuffled_triangles, rt/bbvh-base/bvh2.cpp:172-180
			rtgi_hip_tracer *h = rtgi_hip_renderer_handle(r, 0);
			std::vector<uint32_t> order(rtgi_hip_num_index(h));
			if (rtgi_hip_get_triangle_order(h, order.data()) < 0) throw std::runtime_error(rtgi_hip_last_error(h));
			auto shuffled = s->triangles;
			for (size_t i = 0; i < order.size(); ++i) shuffled[i] = s->triangles[order[i]];
			s->triangles = std::move(shuffled);
		}
	}
	// `material ...` commands after `commit` (driver/interaction.cpp:250-327): the table may be replaced at any time
	void set_materials(const std::vector<hip_material> &m) {
		if (m.size() != n_materials) throw std::invalid_argument("hip_bvh_frame_tracer: as many materials as the committed scene has");
		std::vector<rtgi_material> c(m.size());
		for (size_t i = 0; i < m.size(); ++i) c[i] = rtgi_material{{m[i].albedo[0], m[i].albedo[1], m[i].albedo[2]}, m[i].ior, m[i].roughness, m[i].brdf};
		check(rtgi_hip_renderer_set_materials(r, c.data(), (uint32_t)c.size()));
	}
	// the scene's material table as it is now (inside the renderer; a no-op for the stand-alone mirror scene)
	void refresh_materials() {
		std::vector<hip_material> m = materials_of(this->scene);
		if (!m.empty()) set_materials(m);
		push_textures(this->scene);
	}
	bool interprete(const std::string &command, std::istringstream &in) override {
		std::string rest;
		std::getline(in, rest);
		int rc = rtgi_hip_renderer_interprete(r, command.c_str(), rest.c_str());
		if (rc < 0) {
			fprintf(stderr, "command  (%s): %s\n", command.c_str(), rtgi_hip_renderer_last_error(r));
			return true;
		}
		return rc == 1;
	}
	// One frame: framebuffer.clear, sppx samples per pixel, framebuffer::add per sample (run(algo) of
	// driver/main.cpp:38-48 with a wavefront algorithm). framebuffer: w*h x (mean rgb, sample count), the layout of
	// rc->framebuffer.color (libgi/framebuffer.h:13-50: vec4 per pixel, row y at y*w).
	rtgi_frame_stats render(const rtgi_frame &f, float *framebuffer) {
		rtgi_frame_stats st;
		check(rtgi_hip_renderer_render(r, &f, framebuffer, nullptr, &st));
		return st;
	}
	rtgi_hip_renderer *renderer() { return r; }

private:
	void check(int rc) {
		if (rc < 0) throw std::runtime_error(std::string("rtgi_hip: ") + rtgi_hip_renderer_last_error(r));
	}
#ifdef RTGI_HIP_USE_LIBGI
	// brdf objects carry no name: the kind is told by the dynamic type (libgi/material.h:21-51)
	static uint32_t brdf_kind(::brdf *f) {
		if (auto *l = dynamic_cast<layered_brdf *>(f)) return dynamic_cast<gtr2_reflection *>(l->coat) ? RTGI_BRDF_LAYERED_GTR2 : RTGI_BRDF_LAYERED_PHONG;
		if (dynamic_cast<phong_specular_reflection *>(f)) return RTGI_BRDF_PHONG;
		if (dynamic_cast<gtr2_reflection *>(f)) return RTGI_BRDF_GTR2;
		return RTGI_BRDF_LAMBERT;
	}
	static std::vector<hip_material> materials_of(rtgi_hip::scene *s) {
		std::vector<hip_material> m(s->materials.size());
		for (size_t i = 0; i < m.size(); ++i) {
			const ::material &x = s->materials[i];
			m[i].albedo[0] = x.albedo.x, m[i].albedo[1] = x.albedo.y, m[i].albedo[2] = x.albedo.z;
			m[i].ior = x.ior, m[i].roughness = x.roughness, m[i].brdf = brdf_kind(x.brdf);
		}
		return m;
	}
	// material::albedo_tex (libgi/material.h:57): the textures in use and which material uses which
	void push_textures(rtgi_hip::scene *s) {
		std::vector<rtgi_texture> tex;
		std::vector<const ::texture *> seen;
		std::vector<int32_t> of_material(s->materials.size(), -1);
		for (size_t i = 0; i < s->materials.size(); ++i) {
			const ::texture *t = s->materials[i].albedo_tex;
			if (!t) continue;
			size_t k = 0;
			while (k < seen.size() && seen[k] != t) ++k;
			if (k == seen.size()) {
				seen.push_back(t);
				tex.push_back(rtgi_texture{&t->texel[0].x, t->w, t->h});
			}
			of_material[i] = (int32_t)k;
		}
		if (tex.empty() && !textures_pushed) return;
		const triangle *tris = built_triangles.empty() ? &s->triangles[0] : built_triangles.data();  // (flat: before the permutation)
		check(rtgi_hip_renderer_set_textures(r, &s->vertices[0].tc, sizeof(s->vertices[0]), (uint32_t)s->vertices.size(), &tris[0].a,
		                                     sizeof(triangle), (uint32_t)s->triangles.size(), tex.data(), (uint32_t)tex.size(),
		                                     of_material.data(), (uint32_t)of_material.size()));
		textures_pushed = !tex.empty();
	}
#else
	static std::vector<hip_material> materials_of(rtgi_hip::scene *) { return {}; }
	void push_textures(rtgi_hip::scene *) {}
#endif
	bool textures_pushed = false;
	std::vector<triangle> built_triangles;
	rtgi_hip_renderer *r = nullptr;
	bool indexed;
	uint32_t n_materials = 0;
};

}  // namespace rtgi_hip
