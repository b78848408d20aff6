// Exercises include/rtgi_hip_tracer.hpp the way the renderer would: through ray_tracer* / dynamic_cast
// (libgi/scene.cpp:207-212), script-style "bvh ..." commands, build(scene*), compute_closest_hit() /
// compute_any_hit() and the per-ray closest_hit() / any_hit(). Results are checked against the CPU oracle.
//   adapter_test host   -> no GPU needed: build + interprete + triangle permutation semantics
//   adapter_test gpu    -> full run on device 0
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <omp.h>
#include <random>

#include "rtgi_hip_tracer.hpp"

extern "C" {  // CPU oracle (test infrastructure), oracle/rtgi_oracle.cpp
void *orc_bbvh_build(const float *pos, uint32_t nv, const uint32_t *tri, uint32_t nt, int layout, int esc, int split, int planes, int max_tris);
void orc_bbvh_destroy(void *b);
uint32_t orc_bbvh_num_nodes(void *b);
void orc_bbvh_copy_nodes(void *b, void *out56);
void orc_bbvh_copy_triangles(void *b, uint32_t *tri);
void orc_bbvh_closest_hit(void *b, const float *rays8, uint32_t n, void *hits, uint32_t *counts, uint32_t *max_stack, int threads);
void orc_bbvh_any_hit(void *b, const float *rays8, uint32_t n, uint8_t *out, uint32_t *counts, int threads);
}

using namespace rtgi_hip;

static int failures = 0;
#define EXPECT(cond)                                                  \
	do {                                                              \
		if (!(cond)) {                                                \
			printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
			++failures;                                               \
		}                                                             \
	} while (0)

static void make_soup(rtgi_hip::scene &s, int n, unsigned seed) {
	std::mt19937 rng(seed);
	std::uniform_real_distribution<float> centre(-10.f, 10.f), offset(-0.3f, 0.3f);
	for (int i = 0; i < n; ++i) {
		vec3 c(centre(rng), centre(rng), centre(rng));
		for (int k = 0; k < 3; ++k) {
			vertex v;
			v.pos = vec3(c.x + offset(rng), c.y + offset(rng), c.z + offset(rng));
			v.norm = vec3(0, 1, 0);
			s.vertices.push_back(v);
		}
		s.triangles.push_back(triangle{(uint32_t)(3 * i), (uint32_t)(3 * i + 1), (uint32_t)(3 * i + 2), 0u});
	}
}

int main(int argc, char **argv) {
	const bool gpu = argc > 1 && std::string(argv[1]) == "gpu";
	const int device = gpu ? 0 : -1;
	const int n_tris = 3000, n_rays = 20000;

	rtgi_hip::scene sc;
	make_soup(sc, n_tris, 7);
	std::vector<float> pos(sc.vertices.size() * 3);
	std::vector<uint32_t> tri(sc.triangles.size() * 3);
	for (size_t i = 0; i < sc.vertices.size(); ++i) {
		pos[3 * i] = sc.vertices[i].pos.x, pos[3 * i + 1] = sc.vertices[i].pos.y, pos[3 * i + 2] = sc.vertices[i].pos.z;
	}
	for (size_t i = 0; i < sc.triangles.size(); ++i) {
		tri[3 * i] = sc.triangles[i].a, tri[3 * i + 1] = sc.triangles[i].b, tri[3 * i + 2] = sc.triangles[i].c;
	}

	// "raytracer hip" + "bvh sah 8" + "commit", through the base-class interface
	std::unique_ptr<ray_tracer> rt(new hip_bvh_batch_tracer(/*indexed*/ false, /*esc*/ false, device));
	auto *batch = dynamic_cast<wf::batch_ray_tracer *>(rt.get());
	EXPECT(batch != nullptr);
	EXPECT(dynamic_cast<individual_ray_tracer *>(rt.get()) == nullptr);
	std::istringstream cmd("sah 8"), bad("sah"), other("select pod");
	EXPECT(rt->interprete("bvh", cmd));
	EXPECT(rt->interprete("bvh", bad));          // malformed: reported on stderr, still consumed
	EXPECT(!rt->interprete("material", other));  // not a tracer command
	rt->build(&sc);

	// the oracle with the same settings; the flat layout must have permuted scene.triangles identically
	void *o = orc_bbvh_build(pos.data(), (uint32_t)sc.vertices.size(), tri.data(), (uint32_t)sc.triangles.size(), 0, 0, 2, 8, 1);
	EXPECT(o != nullptr);
	std::vector<uint32_t> otri(tri.size());
	orc_bbvh_copy_triangles(o, otri.data());
	bool same_perm = true;
	for (size_t i = 0; i < sc.triangles.size(); ++i)
		same_perm = same_perm && sc.triangles[i].a == otri[3 * i] && sc.triangles[i].b == otri[3 * i + 1] && sc.triangles[i].c == otri[3 * i + 2];
	EXPECT(same_perm);
	auto *hb = static_cast<hip_bvh_batch_tracer *>(rt.get());
	std::vector<rtgi_node56> nodes(rtgi_hip_num_nodes(hb->core.handle())), onodes(orc_bbvh_num_nodes(o));
	EXPECT(nodes.size() == onodes.size());
	rtgi_hip_get_nodes(hb->core.handle(), nodes.data());
	orc_bbvh_copy_nodes(o, onodes.data());
	EXPECT(nodes.size() == onodes.size() && memcmp(nodes.data(), onodes.data(), nodes.size() * sizeof(rtgi_node56)) == 0);

	if (gpu) {
		std::mt19937 rng(11);
		std::uniform_real_distribution<float> u(-10.f, 10.f), dir(-1.f, 1.f);
		std::vector<ray> rays;
		std::vector<float> rays8;
		for (int i = 0; i < n_rays; ++i) {
			vec3 d(dir(rng), dir(rng), dir(rng));
			float len = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
			ray r(vec3(u(rng), u(rng), u(rng)), vec3(d.x / len, d.y / len, d.z / len));
			if (i % 3 == 0) r.length_exclusive(4.0f);
			rays.push_back(r);
			const float f[8] = {r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, r.t_min, r.t_max};
			rays8.insert(rays8.end(), f, f + 8);
		}
		std::vector<triangle_intersection> got(n_rays), want(n_rays);
		std::vector<uint8_t> want_any(n_rays);
		orc_bbvh_closest_hit(o, rays8.data(), n_rays, want.data(), nullptr, nullptr, 0);
		orc_bbvh_any_hit(o, rays8.data(), n_rays, want_any.data(), nullptr, 0);
		hb->set_batch(rays.data(), got.data(), rays.size());
		batch->compute_closest_hit();
		int hits = 0;
		bool same = true;
		for (int i = 0; i < n_rays; ++i) {
			hits += got[i].valid();
			same = same && memcmp(&got[i].t, &want[i].t, 4) == 0 && got[i].ref == want[i].ref;
			if (want[i].valid()) same = same && memcmp(&got[i].beta, &want[i].beta, 8) == 0;
		}
		EXPECT(same);
		EXPECT(hits > n_rays / 20);
		batch->compute_any_hit();
		bool same_any = true;
		for (int i = 0; i < n_rays; ++i) same_any = same_any && !got[i].valid() && got[i].ref == (unsigned)want_any[i];
		EXPECT(same_any);

		// per-ray interface on a tracer of its own (indexed layout: ids are original ids, scene untouched)
		rtgi_hip::scene sc2;
		make_soup(sc2, n_tris, 7);
		hip_bvh_individual_tracer single(/*indexed*/ true);
		individual_ray_tracer *irt = &single;
		irt->build(&sc2);
		EXPECT(sc2.triangles[5].a == 15);
		void *o2 = orc_bbvh_build(pos.data(), (uint32_t)sc2.vertices.size(), tri.data(), (uint32_t)sc2.triangles.size(), 1, 0, 0, 0, 1);
		std::vector<triangle_intersection> w2(64);
		std::vector<uint8_t> a2(64);
		orc_bbvh_closest_hit(o2, rays8.data(), 64, w2.data(), nullptr, nullptr, 1);
		orc_bbvh_any_hit(o2, rays8.data(), 64, a2.data(), nullptr, 1);
		bool same_single = true;
		for (int i = 0; i < 64; ++i) {
			triangle_intersection is = irt->closest_hit(rays[i]);
			same_single = same_single && memcmp(&is.t, &w2[i].t, 4) == 0 && is.ref == w2[i].ref;
			same_single = same_single && irt->any_hit(rays[i]) == (a2[i] != 0);
		}
		EXPECT(same_single);
		// the way the renderer calls it: from all OpenMP threads at once (libgi/framebuffer.h:41-46); concurrent calls
		// travel to the GPU together
		{
			const int n_mt = 4000 < n_rays ? 4000 : n_rays;
			std::vector<triangle_intersection> w3(n_mt);
			std::vector<uint8_t> a3(n_mt);
			orc_bbvh_closest_hit(o2, rays8.data(), n_mt, w3.data(), nullptr, nullptr, 0);
			orc_bbvh_any_hit(o2, rays8.data(), n_mt, a3.data(), nullptr, 0);
			int wrong = 0;
			const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 16) num_threads(8) reduction(+ : wrong)
			for (int i = 0; i < n_mt; ++i) {
				triangle_intersection is = irt->closest_hit(rays[i]);
				if (memcmp(&is.t, &w3[i].t, 4) != 0 || is.ref != w3[i].ref) ++wrong;
				if ((i & 3) == 0 && irt->any_hit(rays[i]) != (a3[i] != 0)) ++wrong;
			}
			const double dt = omp_get_wtime() - t0;
			EXPECT(wrong == 0);
			printf("per-ray interface from 8 threads: %.0f calls/s\n", 1.25 * n_mt / dt);
		}
		orc_bbvh_destroy(o2);
		// build together with the shading data (vertex normals, material ids, one albedo per material): same tree,
		// same permutation of the scene's triangle list; an out-of-range material id is refused
		rtgi_hip::scene sc3;
		make_soup(sc3, n_tris, 7);
		for (auto &v : sc3.vertices) v.norm = rtgi_hip::vec3(0, 0, 1);
		for (size_t i = 0; i < sc3.triangles.size(); ++i) sc3.triangles[i].material_id = (uint32_t)(i & 1);
		const float albedo[6] = {0.8f, 0.2f, 0.1f, 0.1f, 0.3f, 0.9f};
		hip_bvh_batch_tracer shaded(false, false);
		std::istringstream sah8(" sah 8");
		shaded.core.interprete("bvh", sah8);
		shaded.core.build(&sc3, albedo, 2);
		bool same_order = true;
		for (size_t i = 0; i < sc3.triangles.size(); ++i) same_order = same_order && sc3.triangles[i].a == sc.triangles[i].a;
		EXPECT(same_order);
		bool refused = false;
		try {
			rtgi_hip::scene sc4;
			make_soup(sc4, n_tris, 7);
			for (auto &v : sc4.vertices) v.norm = rtgi_hip::vec3(0, 0, 1);
			sc4.triangles[3].material_id = 2;
			hip_bvh_batch_tracer other(true, false);
			other.core.build(&sc4, albedo, 2);
		} catch (const std::runtime_error &) {
			refused = true;
		}
		EXPECT(refused);
		// a frame-sized batch through the renderer's buffers (host memory): rate of compute_closest_hit as a whole
		{
			const size_t big = 1u << 20;
			std::vector<ray> many(big);
			for (size_t i = 0; i < big; ++i) many[i] = rays[i % rays.size()];
			std::vector<triangle_intersection> res(big);
			hb->set_batch(many.data(), res.data(), big);
			if (rtgi_hip_host_register(many.data(), 4096) == RTGI_OK) rtgi_hip_host_unregister(many.data());
			else printf("(page-locking caller memory is not available here: %s)\n", rtgi_hip_last_error(nullptr));
			batch->compute_closest_hit();
			const double t0 = omp_get_wtime();
			for (int rep = 0; rep < 3; ++rep) batch->compute_closest_hit();
			const double dt = (omp_get_wtime() - t0) / 3;
			bool same_big = true;
			for (size_t i = 0; i < big; i += 997) same_big = same_big && memcmp(&res[i].t, &want[i % rays.size()].t, 4) == 0 && res[i].ref == want[i % rays.size()].ref;
			EXPECT(same_big);
			printf("batch interface on host buffers: %.0f Mrays/s (2^20 rays per call, packing and copies included)\n", big / dt / 1e6);
			hb->set_batch(rays.data(), got.data(), rays.size());
		}
		// an unmodified wavefront algorithm fills the tracer's own resolution-sized buffers (wf::cpu::batch_rt::rd,
		// libgi/wavefront-rt.h:52-56) and calls compute_*() without further ado
		{
			rtgi_hip::scene sc5;
			make_soup(sc5, n_tris, 7);
			hip_bvh_batch_tracer own(/*indexed*/ true);
			own.build(&sc5);
			bool threw = false;
			try {
				own.compute_closest_hit();  // resolution 0x0: nothing to trace is an error, not a silent no-op
			} catch (const std::logic_error &) {
				threw = true;
			}
			EXPECT(threw);
			own.rd.resize(100, 64);  // `resolution 100 64`
			for (int i = 0; i < 6400; ++i) own.rd.rays[i] = rays[i];
			wf::batch_ray_tracer *as_batch = &own;
			as_batch->compute_closest_hit();
			std::vector<triangle_intersection> w5(6400);
			void *o5 = orc_bbvh_build(pos.data(), (uint32_t)sc5.vertices.size(), tri.data(), (uint32_t)sc5.triangles.size(), 1, 0, 0, 0, 1);
			orc_bbvh_closest_hit(o5, rays8.data(), 6400, w5.data(), nullptr, nullptr, 0);
			orc_bbvh_destroy(o5);
			bool same_own = true;
			for (int i = 0; i < 6400; ++i) same_own = same_own && memcmp(&own.rd.intersections[i].t, &w5[i].t, 4) == 0 && own.rd.intersections[i].ref == w5[i].ref;
			EXPECT(same_own);
		}
		// the frame tracer (`raytracer hip devices ...`): one GPU named twice renders the frame of one handle
		{
			rtgi_hip::scene s6, s7;
			make_soup(s6, n_tris, 7);
			make_soup(s7, n_tris, 7);
			for (size_t i = 0; i < s6.triangles.size(); ++i) s6.triangles[i].material_id = s7.triangles[i].material_id = (uint32_t)(i % 3);
			hip_bvh_frame_tracer one(false, false, {0}), two(false, false, {0, 0});
			one.build(&s6);
			two.build(&s7);
				std::vector<hip_material> mats(3);
			for (int m = 0; m < 3; ++m) mats[m].albedo[0] = 0.2f + 0.3f * m, mats[m].albedo[1] = 0.5f, mats[m].albedo[2] = 0.9f - 0.3f * m, mats[m].brdf = (uint32_t)m;
			mats[1].roughness = 0.3f;
			one.set_materials(mats);
			two.set_materials(mats);
			rtgi_frame f{};
			f.cam = rtgi_camera{{0, 0, -1}, {0, 0, 14}, {0, 1, 0}, 65.0f, 160, 100};
			f.sppx = 3, f.algo = RTGI_ALGO_LOCAL, f.jitter = 1, f.seed = 5, f.sharding = RTGI_SHARD_TILES, f.tile = 16;
			f.light_pos[0] = 0, f.light_pos[1] = 12, f.light_pos[2] = 12;
			f.light_col[0] = f.light_col[1] = f.light_col[2] = 30.0f;
			std::vector<float> fb1(160 * 100 * 4), fb2(fb1.size());
			const rtgi_frame_stats st1 = one.render(f, fb1.data()), st2 = two.render(f, fb2.data());
			EXPECT(st1.rays == st2.rays && st2.n_devices == 2);
			EXPECT(memcmp(fb1.data(), fb2.data(), fb1.size() * 4) == 0);
			double sum = 0;
			bool counts = true;
			for (size_t i = 0; i < fb1.size(); i += 4) sum += fb1[i] + fb1[i + 1] + fb1[i + 2], counts = counts && fb1[i + 3] == 3.0f;
			EXPECT(counts && sum > 0);
			bool refused = false;
			try {
				one.set_materials(std::vector<hip_material>(2));
			} catch (const std::invalid_argument &) {
				refused = true;
			}
			EXPECT(refused);
		}
		printf("gpu: %d rays, %d hits\n", n_rays, hits);
	}
	orc_bbvh_destroy(o);
	printf(failures ? "adapter_test: %d FAILURES\n" : "adapter_test: OK\n", failures);
	return failures ? 1 : 0;
}
