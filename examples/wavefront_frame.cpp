// wavefront_frame.cpp -- a whole frame of a wavefront algorithm in the renderer's own language, on the C ABI alone:
// camera rays -> closest_hit -> shadow rays -> any_hit -> local illumination -> framebuffer -> 8-bit image.
// Everything between the scene arrays and the final image lives in GPU memory (hipMalloc); the loop is the one of
// "Assignment 2/rtgi-2021-a02/gi/primary-hit.cpp:61-74" with the per-pixel work of
// "Assignment 3/rtgi-2021-a03-ref/gi/primary-hit.cpp:32-57".
//
//   wavefront_frame <width> <height> <sppx> <out.ppm>      (scene: a closed box with two blocks and a point light)
//
// Build: make -C examples   (hipcc as host compiler for the HIP runtime headers; links librtgi_hip.so)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rtgi_hip.h"
#include "rtgi_hip_tracer.hpp"  // hip_bvh_frame_tracer: the same frame through the renderer-side class

struct vertex {  // libgi/rt.h:29-33
	float pos[3], norm[3], tc[2];
};
struct triangle {  // libgi/rt.h:35-38
	uint32_t a, b, c, material_id;
};

#define HIP_OK(x)                                                                        \
	do {                                                                                 \
		hipError_t e_ = (x);                                                             \
		if (e_ != hipSuccess) {                                                          \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
			return 1;                                                                    \
		}                                                                                \
	} while (0)
#define RT_OK(x)                                                                         \
	do {                                                                                 \
		if ((x) < 0) {                                                                   \
			fprintf(stderr, "%s: %s\n", #x, rtgi_hip_last_error(rt));                    \
			return 1;                                                                    \
		}                                                                                \
	} while (0)

// an axis-aligned quad as two triangles with its own four vertices; normal n
static void quad(std::vector<vertex> &v, std::vector<triangle> &t, const float o[3], const float du[3], const float dv[3],
                 const float n[3], uint32_t material) {
	const uint32_t base = (uint32_t)v.size();
	for (int k = 0; k < 4; ++k) {
		vertex x{};
		const float s = (k == 1 || k == 2) ? 1.0f : 0.0f, r = (k >= 2) ? 1.0f : 0.0f;
		for (int c = 0; c < 3; ++c) x.pos[c] = o[c] + s * du[c] + r * dv[c], x.norm[c] = n[c];
		v.push_back(x);
	}
	t.push_back(triangle{base, base + 1, base + 2, material});
	t.push_back(triangle{base, base + 2, base + 3, material});
}
static void box(std::vector<vertex> &v, std::vector<triangle> &t, const float lo[3], const float hi[3], bool inward, uint32_t material) {
	for (int axis = 0; axis < 3; ++axis)
		for (int side = 0; side < 2; ++side) {
			const int u = (axis + 1) % 3, w = (axis + 2) % 3;
			float o[3] = {lo[0], lo[1], lo[2]}, du[3] = {0, 0, 0}, dv[3] = {0, 0, 0}, n[3] = {0, 0, 0};
			o[axis] = side ? hi[axis] : lo[axis];
			du[u] = hi[u] - lo[u];
			dv[w] = hi[w] - lo[w];
			n[axis] = (side ? 1.0f : -1.0f) * (inward ? -1.0f : 1.0f);
			quad(v, t, o, du, dv, n, material + (uint32_t)axis * (inward ? 1u : 0u));
		}
}

int main(int argc, char **argv) {
	const int w = argc > 1 ? atoi(argv[1]) : 640, h = argc > 2 ? atoi(argv[2]) : 360, sppx = argc > 3 ? atoi(argv[3]) : 4;
	const char *out = argc > 4 ? argv[4] : "frame.ppm";
	if (w <= 0 || h <= 0 || sppx <= 0) return 2;
	std::vector<vertex> vertices;
	std::vector<triangle> triangles;
	const float room_lo[3] = {-1, -1, -1}, room_hi[3] = {1, 1, 1.5f};
	const float b1_lo[3] = {-0.7f, -1, -0.6f}, b1_hi[3] = {-0.1f, 0.2f, 0.0f}, b2_lo[3] = {0.15f, -1, -0.1f}, b2_hi[3] = {0.7f, -0.4f, 0.5f};
	box(vertices, triangles, room_lo, room_hi, true, 0);   // materials 0, 1, 2: walls by axis
	box(vertices, triangles, b1_lo, b1_hi, false, 3);
	box(vertices, triangles, b2_lo, b2_hi, false, 4);
	const float albedo[5 * 3] = {0.7f, 0.15f, 0.12f, 0.75f, 0.75f, 0.75f, 0.15f, 0.2f, 0.7f, 0.8f, 0.7f, 0.3f, 0.3f, 0.8f, 0.4f};

	rtgi_hip_tracer *rt = rtgi_hip_create(0, RTGI_LAYOUT_FLAT, 0);  // raytracer hip
	if (!rt) {
		fprintf(stderr, "rtgi_hip_create: %s\n", rtgi_hip_last_error(nullptr));
		return 1;
	}
	RT_OK(rtgi_hip_interprete(rt, "bvh", "sah 8"));
	RT_OK(rtgi_hip_build(rt, vertices[0].pos, sizeof(vertex), (uint32_t)vertices.size(), &triangles[0].a, sizeof(triangle),
	                     (uint32_t)triangles.size()));  // commit
	RT_OK(rtgi_hip_set_shading(rt, vertices[0].norm, sizeof(vertex), (uint32_t)vertices.size(), &triangles[0].a, sizeof(triangle),
	                           (uint32_t)triangles.size(), albedo, 5));

	const uint32_t n = (uint32_t)w * (uint32_t)h;
	float *d_offsets, *d_radiance, *d_fb;
	rtgi_ray *d_rays, *d_shadow;
	rtgi_hit *d_hits;
	uint8_t *d_occluded, *d_rgb;
	HIP_OK(hipMalloc((void **)&d_offsets, (size_t)n * 8));
	HIP_OK(hipMalloc((void **)&d_rays, (size_t)n * sizeof(rtgi_ray)));
	HIP_OK(hipMalloc((void **)&d_shadow, (size_t)n * sizeof(rtgi_ray)));
	HIP_OK(hipMalloc((void **)&d_hits, (size_t)n * sizeof(rtgi_hit)));
	HIP_OK(hipMalloc((void **)&d_occluded, n));
	HIP_OK(hipMalloc((void **)&d_radiance, (size_t)n * 16));
	HIP_OK(hipMalloc((void **)&d_fb, (size_t)n * 16));
	HIP_OK(hipMalloc((void **)&d_rgb, (size_t)n * 3));
	HIP_OK(hipMemset(d_fb, 0, (size_t)n * 16));  // framebuffer::clear

	const rtgi_camera cam = {{0, 0, -1}, {0, 0, 1.45f}, {0, 1, 0}, 65.0f, w, h};
	const float light_pos[3] = {0.0f, 0.9f, 0.4f}, light_col[3] = {0.25f, 0.25f, 0.25f};
	hipStream_t stream;
	HIP_OK(hipStreamCreate(&stream));
	hipEvent_t t0, t1;
	HIP_OK(hipEventCreate(&t0));
	HIP_OK(hipEventCreate(&t1));
	HIP_OK(hipEventRecord(t0, stream));
	for (int sample = 0; sample < sppx; ++sample) {  // nothing in this loop waits for the GPU
		RT_OK(rtgi_hip_pixel_offsets_device(rt, n, (uint32_t)sample, 0, d_offsets, stream));
		RT_OK(rtgi_hip_camera_rays_device(rt, &cam, d_offsets, d_rays, stream));
		RT_OK(rtgi_hip_closest_hit_device(rt, d_rays, n, d_hits, nullptr, stream));
		RT_OK(rtgi_hip_shadow_rays_device(rt, d_hits, n, light_pos, d_shadow, stream));
		RT_OK(rtgi_hip_any_hit_device(rt, d_shadow, n, d_occluded, nullptr, stream));
		RT_OK(rtgi_hip_shade_local_device(rt, d_hits, d_occluded, n, light_pos, light_col, d_radiance, stream));
		RT_OK(rtgi_hip_framebuffer_add_device(rt, d_fb, d_radiance, n, stream));
	}
	RT_OK(rtgi_hip_framebuffer_rgb8_device(rt, d_fb, (uint32_t)w, (uint32_t)h, d_rgb, stream));
	HIP_OK(hipEventRecord(t1, stream));
	std::vector<uint8_t> rgb((size_t)n * 3);
	HIP_OK(hipMemcpyAsync(rgb.data(), d_rgb, rgb.size(), hipMemcpyDeviceToHost, stream));
	HIP_OK(hipStreamSynchronize(stream));
	float ms = 0;
	HIP_OK(hipEventElapsedTime(&ms, t0, t1));

	FILE *f = fopen(out, "wb");
	if (!f) return 1;
	fprintf(f, "P6\n%d %d\n255\n", w, h);
	fwrite(rgb.data(), 1, rgb.size(), f);
	fclose(f);
	uint64_t sum = 0, lit = 0;
	for (size_t i = 0; i < rgb.size(); ++i) sum = sum * 1099511628211ull + rgb[i], lit += rgb[i] > 0;
	printf("%dx%d, %d sppx, %zu triangles: %.2f ms on the GPU, %.0f Mrays/s; lit bytes %.1f%%; checksum %016llx; wrote %s\n", w, h, sppx,
	       triangles.size(), ms, 2.0 * n * sppx / ms / 1e3, 100.0 * lit / rgb.size(), (unsigned long long)sum, out);
	// The same frame through hip_bvh_frame_tracer, the object `raytracer hip` puts into scene.rt and the wavefront
	// algorithms of rtgi_hip_algorithms.hpp render with: its framebuffer must be the stage-by-stage one, bit for bit.
	std::vector<float> fb_stages((size_t)n * 4), fb_class((size_t)n * 4);
	HIP_OK(hipMemcpy(fb_stages.data(), d_fb, fb_stages.size() * 4, hipMemcpyDeviceToHost));
	int rc = 0;
	try {
		rtgi_hip::scene sc;
		sc.vertices.resize(vertices.size());
		sc.triangles.resize(triangles.size());
		static_assert(sizeof(rtgi_hip::vertex) == sizeof(vertex) && sizeof(rtgi_hip::triangle) == sizeof(triangle), "same records");
		memcpy((void *)sc.vertices.data(), vertices.data(), vertices.size() * sizeof(vertex));
		memcpy((void *)sc.triangles.data(), triangles.data(), triangles.size() * sizeof(triangle));
		rtgi_hip::hip_bvh_frame_tracer frame_rt(false, false, {0});
		rtgi_hip::ray_tracer *as_base = &frame_rt;
		std::istringstream sah("sah 8");
		as_base->interprete("bvh", sah);
		as_base->build(&sc);
		std::vector<rtgi_hip::hip_material> mats(5);
		for (int m = 0; m < 5; ++m)
			for (int k = 0; k < 3; ++k) mats[m].albedo[k] = albedo[3 * m + k];
		frame_rt.set_materials(mats);
		rtgi_frame f{};
		f.cam = cam;
		f.sppx = (uint32_t)sppx, f.algo = RTGI_ALGO_LOCAL, f.jitter = 1, f.seed = 0;
		for (int k = 0; k < 3; ++k) f.light_pos[k] = light_pos[k], f.light_col[k] = light_col[k];
		const rtgi_frame_stats st = frame_rt.render(f, fb_class.data());
		const bool same = memcmp(fb_stages.data(), fb_class.data(), fb_stages.size() * 4) == 0;
		printf("hip_bvh_frame_tracer: %llu rays in %.2f ms, framebuffer %s the stage-by-stage one\n", (unsigned long long)st.rays, st.ms_total,
		       same ? "identical to" : "DIFFERS from");
		if (!same) rc = 1;
	} catch (const std::exception &e) {
		fprintf(stderr, "hip_bvh_frame_tracer: %s\n", e.what());
		rc = 1;
	}
	rtgi_hip_destroy(rt);
	return rc;
}
