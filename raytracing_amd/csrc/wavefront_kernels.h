// wavefront_kernels.h -- the ray populations around the tracer, produced on the GPU so that rays and hits
// never leave it: camera rays, shadow rays, and diffuse bounce rays with order-preserving compaction.
// (Device code only; included by rtgi_hip.hip.) File:line citations: "Assignment 3/rtgi-2021-a03-ref/".
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "rtgi_hip.h"

namespace rtgi_dev {

constexpr float RAY_EPS = 1e-4f;  // ray::eps, libgi/rt.h:21

struct CamArgs {
	float dir[3], pos[3], U[3], V[3];  // U = cross(dir, up), V = cross(U, dir): libgi/camera.cpp:13-14
	float near_w, near_h;              // libgi/camera.h:22-23
	int w, h;
	const float2 *offsets;  // optional sub-pixel jitter per pixel, as the caller's rng would give (camera.cpp:11)
	float4 *rays;
	const uint32_t *pixels;  // optional: ray i (and offset i) belongs to pixel pixels[i] = y * w + x, n of them
	uint32_t n;
};

// cam_ray, libgi/camera.cpp:11-22, for every pixel (batch_cam_ray_setup_cpu, libgi/wavefront-rt.h:90-102)
__global__ __launch_bounds__(256) void camera_rays_kernel(CamArgs a) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= a.n) return;
	const uint32_t pixel = a.pixels ? a.pixels[i] : i;
	const int x = (int)(pixel % (uint32_t)a.w), y = (int)(pixel / (uint32_t)a.w);
	float ox = 0.0f, oy = 0.0f;
	if (a.offsets) {
		const float2 o = a.offsets[i];
		ox = o.x, oy = o.y;
	}
	float u = ((float)x + 0.5f + ox) / (float)a.w * 2.0f - 1.0f;
	float v = ((float)y + 0.5f + oy) / (float)a.h * 2.0f - 1.0f;
	u = a.near_w * u;
	v = a.near_h * v;
	// dir + U*u + V*v, then glm::normalize = v * inversesqrt(dot(v, v))
	const float dx = a.dir[0] + a.U[0] * u + a.V[0] * v;
	const float dy = a.dir[1] + a.U[1] * u + a.V[1] * v;
	const float dz = a.dir[2] + a.U[2] * u + a.V[2] * v;
	const float px = dx * dx, py = dy * dy, pz = dz * dz;
	const float inv = 1.0f / sqrtf(px + py + pz);
	a.rays[2 * (size_t)i] = make_float4(a.pos[0], a.pos[1], a.pos[2], dx * inv);
	a.rays[2 * (size_t)i + 1] = make_float4(dy * inv, dz * inv, RAY_EPS, FLT_MAX);
}

// Triangle corner positions by the id a hit reports: 3 x float4 per triangle (A, B, C).
__device__ __forceinline__ void hit_point(const float4 *corners, const float4 hit, float &x, float &y, float &z, float4 &A,
                                          float4 &B, float4 &C) {
	const uint32_t id = __float_as_uint(hit.w);
	A = corners[3 * (size_t)id], B = corners[3 * (size_t)id + 1], C = corners[3 * (size_t)id + 2];
	const float beta = hit.y, gamma = hit.z;
	const float alpha = 1.0f - beta - gamma;  // diff_geom::x, libgi/rt.cpp:9
	x = alpha * A.x + beta * B.x + gamma * C.x;
	y = alpha * A.y + beta * B.y + gamma * C.y;
	z = alpha * A.z + beta * B.z + gamma * C.z;
}

struct ShadowArgs {
	const float4 *hits;
	const float4 *corners;
	float4 *out;
	float light[3];
	uint32_t n;
};

// Shadow rays of gi/primary-hit.cpp:43-50, one slot per input ray: from the hit point towards the light,
// ending eps before it (length_exclusive). Rays that hit nothing get an empty interval (t_max < t_min), which
// the tracer answers with "not occluded" after one node.
__global__ __launch_bounds__(256) void shadow_rays_kernel(ShadowArgs a) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= a.n) return;
	const float4 hit = a.hits[i];
	if (hit.x == FLT_MAX) {
		a.out[2 * (size_t)i] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
		a.out[2 * (size_t)i + 1] = make_float4(0.0f, 0.0f, RAY_EPS, -1.0f);
		return;
	}
	float x, y, z;
	float4 A, B, C;
	hit_point(a.corners, hit, x, y, z, A, B, C);
	const float lx = a.light[0] - x, ly = a.light[1] - y, lz = a.light[2] - z;
	const float px = lx * lx, py = ly * ly, pz = lz * lz;
	const float d = sqrtf(px + py + pz);
	const float inv = 1.0f / d;
	a.out[2 * (size_t)i] = make_float4(x, y, z, lx * inv);
	a.out[2 * (size_t)i + 1] = make_float4(ly * inv, lz * inv, RAY_EPS, d - RAY_EPS);
}

// Counter-based random numbers: a 32-bit finaliser keyed by (ray index, stream, seed); 24 bits per float.
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
	x ^= x >> 16;
	x *= 0x7FEB352Du;
	x ^= x >> 15;
	x *= 0x846CA68Bu;
	x ^= x >> 16;
	return x;
}
__device__ __forceinline__ float uniform01(uint32_t index, uint32_t stream, uint32_t seed) {
	uint32_t key = hash_u32(index * 0x9E3779B1u + stream * 0x85EBCA77u + seed * 0xC2B2AE3Du);
	key = hash_u32(key ^ 0x27D4EB2Fu);
	return (float)(key >> 8) * (1.0f / 16777216.0f);
}

// Sub-pixel offsets for one sample of every pixel, uniform in [-0.5, 0.5): what the renderer draws from its
// rng per pixel (gi/primary-hit.cpp:20), here keyed by (pixel, sample, seed) so that a frame is reproducible.
__global__ __launch_bounds__(256) void pixel_offsets_kernel(float2 *offsets, const uint32_t *pixels, uint32_t n, uint32_t sample,
                                                            uint32_t seed) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t pixel = pixels ? pixels[i] : i;  // the key is the pixel, wherever in the batch its ray sits
	offsets[i] = make_float2(uniform01(pixel, 2u * sample, seed) - 0.5f, uniform01(pixel, 2u * sample + 1u, seed) - 0.5f);
}

struct BounceArgs {
	const float4 *rays;
	const float4 *hits;
	const float4 *corners;
	uint32_t *block_counts;  // per 1024 rays: pass 1 out, pass 2 in (after the exclusive scan: offsets)
	float4 *out;
	uint32_t *parent;  // optional: index of the ray each bounce ray continues
	// what the random numbers of ray i are keyed by: i itself, or keys[i], or key_table[keys[i]] (a pixel id that
	// stays the same however the frame's rays are batched)
	const uint32_t *keys, *key_table;
	uint32_t n, seed, bounce;
};

// Pass 1: how many rays of each 1024-ray block hit something (ballot + popcount per wave, 16 waves per block).
__global__ __launch_bounds__(1024) void bounce_count_kernel(BounceArgs a) {
	__shared__ uint32_t s_wave[16];
	const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
	const bool valid = i < a.n && a.hits[i].x != FLT_MAX;
	const unsigned long long m = __ballot(valid);
	if ((threadIdx.x & 63u) == 0) s_wave[threadIdx.x >> 6] = (uint32_t)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t sum = 0;
		for (int k = 0; k < 16; ++k) sum += s_wave[k];
		a.block_counts[blockIdx.x] = sum;
	}
}

// Exclusive scan of the per-block counts by one workgroup; total[0] receives the sum.
__global__ __launch_bounds__(1024) void scan_kernel(uint32_t *counts, uint32_t n, uint32_t *total) {
	__shared__ uint32_t partial[1024];
	const uint32_t per = (n + 1023u) / 1024u;
	const uint32_t begin = threadIdx.x * per, end = min(begin + per, n);
	uint32_t sum = 0;
	for (uint32_t i = begin; i < end; ++i) sum += counts[i];
	partial[threadIdx.x] = sum;
	__syncthreads();
	for (uint32_t step = 1; step < 1024u; step <<= 1) {
		uint32_t v = threadIdx.x >= step ? partial[threadIdx.x - step] : 0u;
		__syncthreads();
		partial[threadIdx.x] += v;
		__syncthreads();
	}
	uint32_t run = partial[threadIdx.x] - sum;
	for (uint32_t i = begin; i < end; ++i) {
		const uint32_t c = counts[i];
		counts[i] = run;
		run += c;
	}
	if (threadIdx.x == 1023u) total[0] = partial[1023];
}

// Pass 2: one diffuse bounce for every ray that hit something, written densely in the order of the input
// rays: from the hit point into a cosine-weighted random direction about the geometric normal that faces the
// incoming ray. (The renderer has no path tracer; this is this backend's definition of the bounce population,
// mirrored on the host by raytracing_amd/wavefront.py.)
__global__ __launch_bounds__(1024) void bounce_write_kernel(BounceArgs a) {
	__shared__ uint32_t s_wave[16];
	const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
	const float4 hit = i < a.n ? a.hits[i] : make_float4(FLT_MAX, 0, 0, 0);
	const bool valid = i < a.n && hit.x != FLT_MAX;
	const unsigned long long m = __ballot(valid);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
	__syncthreads();
	if (!valid) return;
	uint32_t before = a.block_counts[blockIdx.x];
	for (uint32_t k = 0; k < wave; ++k) before += s_wave[k];
	const uint32_t slot = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
	float x, y, z;
	float4 A, B, C;
	hit_point(a.corners, hit, x, y, z, A, B, C);
	const float e1x = B.x - A.x, e1y = B.y - A.y, e1z = B.z - A.z;
	const float e2x = C.x - A.x, e2y = C.y - A.y, e2z = C.z - A.z;
	float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
	const float nl = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-30f);
	nx /= nl, ny /= nl, nz /= nl;
	const float4 r0 = a.rays[2 * (size_t)i], r1 = a.rays[2 * (size_t)i + 1];
	if (nx * r0.w + ny * r1.x + nz * r1.y > 0.0f) nx = -nx, ny = -ny, nz = -nz;
	uint32_t key = a.keys ? a.keys[i] : i;
	if (a.key_table) key = a.key_table[key];
	const float u1 = uniform01(key, 2u * a.bounce, a.seed), u2 = uniform01(key, 2u * a.bounce + 1u, a.seed);
	const float r = sqrtf(u1), phi = 6.2831853071795864769f * u2;
	float sn, cs;
	sincosf(phi, &sn, &cs);
	const float lx = r * cs, ly = r * sn, lz = sqrtf(fmaxf(0.0f, 1.0f - u1));
	// orthonormal basis around n (Duff et al., branchless)
	const float sign = nz >= 0.0f ? 1.0f : -1.0f;
	const float aa = -1.0f / (sign + nz);
	const float bb = nx * ny * aa;
	const float t1x = 1.0f + sign * nx * nx * aa, t1y = sign * bb, t1z = -sign * nx;
	const float t2x = bb, t2y = sign + ny * ny * aa, t2z = -ny;
	float dx = t1x * lx + t2x * ly + nx * lz;
	float dy = t1y * lx + t2y * ly + ny * lz;
	float dz = t1z * lx + t2z * ly + nz * lz;
	const float dl = sqrtf(dx * dx + dy * dy + dz * dz);
	dx /= dl, dy /= dl, dz /= dl;
	a.out[2 * (size_t)slot] = make_float4(x, y, z, dx);
	a.out[2 * (size_t)slot + 1] = make_float4(dy, dz, RAY_EPS, FLT_MAX);
	if (a.parent) a.parent[slot] = i;
}

// ---- shading and accumulation: what the renderer's algorithms do with a hit ---------------------------------

struct ShadeArgs {
	const float4 *hits;
	const uint8_t *occluded;  // local illumination: any_hit result of the shadow rays
	const float4 *corners;    // 3 x float4 per reported triangle id
	const float4 *normals;    // per reported triangle id: shading normal xyz, material index (bits) in w
	const float4 *albedo;     // per material
	float4 *out;              // radiance rgb, w = 0
	float light[3], power[3];
	uint32_t n, n_materials;
};

__device__ __forceinline__ float4 material_albedo(const ShadeArgs &a, float4 normal) {
	const uint32_t m = __float_as_uint(normal.w);
	return m < a.n_materials ? a.albedo[m] : make_float4(0, 0, 0, 0);
}

// primary_hit_display::sample_pixel, gi/primary-hit.cpp:16-30: the albedo of what the ray hit, else black
// (diff_geom::albedo without a texture, libgi/rt.cpp:30-34)
__global__ __launch_bounds__(256) void shade_primary_kernel(ShadeArgs a) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= a.n) return;
	const float4 hit = a.hits[i];
	float4 c = make_float4(0, 0, 0, 0);
	if (hit.x != FLT_MAX) {
		c = material_albedo(a, a.normals[__float_as_uint(hit.w)]);
		c.w = 0.0f;
	}
	a.out[i] = c;
}

// local_illumination::sample_pixel, gi/primary-hit.cpp:32-57, for a point light and the lambertian BRDF
// (libgi/material.cpp:17-20): power * f / d^2 where the shadow ray found nothing, with power = 4 pi col
// (libgi/scene.cpp:216-218) and f = albedo / pi on the side of the shading normal that faces the light.
__global__ __launch_bounds__(256) void shade_local_kernel(ShadeArgs a) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= a.n) return;
	const float4 hit = a.hits[i];
	float4 c = make_float4(0, 0, 0, 0);
	if (hit.x != FLT_MAX && !a.occluded[i]) {
		float x, y, z;
		float4 A, B, C;
		hit_point(a.corners, hit, x, y, z, A, B, C);
		const float4 ns = a.normals[__float_as_uint(hit.w)];
		const float lx = a.light[0] - x, ly = a.light[1] - y, lz = a.light[2] - z;
		const float px = lx * lx, py = ly * ly, pz = lz * lz;
		const float d = sqrtf(px + py + pz);
		const float inv = 1.0f / d;
		const float wx = lx * inv, wy = ly * inv, wz = lz * inv;
		const float qx = ns.x * wx, qy = ns.y * wy, qz = ns.z * wz;
		if (qx + qy + qz > 0.0f) {  // same_hemisphere(w_i, ns), libgi/util.h:115-117
			const float one_over_pi = (float)(1.0 / 3.14159265358979323846);  // libgi/rt.h:15
			const float4 alb = material_albedo(a, ns);
			const float dd = d * d;
			c.x = a.power[0] * (one_over_pi * alb.x) / dd;
			c.y = a.power[1] * (one_over_pi * alb.y) / dd;
			c.z = a.power[2] * (one_over_pi * alb.z) / dd;
		}
	}
	a.out[i] = c;
}

// framebuffer::add, libgi/framebuffer.cpp:10-18, with one sample per pixel: rgb is the running mean, w the
// number of samples so far.
__global__ __launch_bounds__(256) void framebuffer_add_kernel(float4 *fb, const float4 *radiance, uint32_t n) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const float4 c = fb[i], r = radiance[i];
	const float new_count = c.w + 1.0f;
	fb[i] = make_float4((c.x * c.w + r.x) / new_count, (c.y * c.w + r.y) / new_count, (c.z * c.w + r.z) / new_count, new_count);
}

// Paths of more than one segment (no counterpart in the reference, whose algorithms stop at the first hit): a ray
// of segment k carries the product of the albedos its path has met (cosine-weighted sampling of a lambertian
// surface has weight albedo) and the slot of the pixel sample it contributes to.
// begin: throughput 1, slot i, radiance 0 for the n camera rays of a sample.
__global__ __launch_bounds__(256) void path_begin_kernel(float4 *throughput, uint32_t *slot, float4 *radiance, uint32_t n) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	throughput[i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
	slot[i] = i;
	radiance[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
// add: what the light sends to the path's vertex, weighted: radiance[slot[j]] += throughput[j] * direct[j]. Every
// slot occurs at most once among the m rays of a segment, so plain stores do.
__global__ __launch_bounds__(256) void path_add_kernel(const float4 *direct, const float4 *throughput, const uint32_t *slot, uint32_t m,
                                                       float4 *radiance) {
	const uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j >= m) return;
	const float4 d = direct[j], t = throughput[j];
	const uint32_t s = slot[j];
	const float4 r = radiance[s];
	radiance[s] = make_float4(r.x + t.x * d.x, r.y + t.y * d.y, r.z + t.z * d.z, 0.0f);
}
// extend: the rays of the next segment (ray j continues ray parent[j]) inherit slot and throughput * albedo.
__global__ __launch_bounds__(256) void path_extend_kernel(const float4 *albedo, const uint32_t *parent, uint32_t m, const float4 *throughput_in,
                                                          const uint32_t *slot_in, float4 *throughput_out, uint32_t *slot_out) {
	const uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j >= m) return;
	const uint32_t p = parent[j];
	const float4 t = throughput_in[p], a = albedo[p];
	throughput_out[j] = make_float4(t.x * a.x, t.y * a.y, t.z * a.z, 0.0f);
	slot_out[j] = slot_in[p];
}

// The framebuffer of a frame that several GPUs rendered parts of. Each GPU accumulates its pixels (or its samples)
// like framebuffer::add does; the parts are then combined as sums and counts, never as means (a running mean with
// its count in .w, libgi/framebuffer.cpp:10-18, is only meaningful together with that count).
// full[pixels[i]] = part[i]: a GPU's densely stored pixels into their places in the frame.
__global__ __launch_bounds__(256) void framebuffer_scatter_kernel(const float4 *part, const uint32_t *pixels, uint32_t n, float4 *full) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) full[pixels[i]] = part[i];
}
// (mean rgb, count) -> (sum rgb, count) and back; a pixel without samples stays (0, 0, 0, 0)
__global__ __launch_bounds__(256) void framebuffer_sums_kernel(float4 *fb, uint32_t n, int to_sums) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	float4 c = fb[i];
	if (to_sums)
		c.x *= c.w, c.y *= c.w, c.z *= c.w;
	else if (c.w > 0.0f)
		c.x /= c.w, c.y /= c.w, c.z /= c.w;
	fb[i] = c;
}
__global__ __launch_bounds__(256) void framebuffer_accumulate_kernel(float4 *dst, const float4 *src, uint32_t n) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const float4 a = dst[i], b = src[i];
	dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// framebuffer::png, libgi/framebuffer.cpp:21-30: clamp to [0,1], gamma 1/2.2, scale to 255, truncate to a
// byte, first image row = last framebuffer row.
__global__ __launch_bounds__(256) void framebuffer_rgb8_kernel(const float4 *fb, uint32_t w, uint32_t h, uint8_t *rgb) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= w * h) return;
	const uint32_t x = i % w, y = i / w;
	const float4 c = fb[i];
	auto tone = [](float v) {
		v = fminf(fmaxf(v, 0.0f), 1.0f);
		return (uint8_t)(powf(v, 1.0f / 2.2f) * 255.0f);
	};
	uint8_t *o = rgb + 3 * ((size_t)(h - 1u - y) * w + x);
	o[0] = tone(c.x), o[1] = tone(c.y), o[2] = tone(c.z);
}

}  // namespace rtgi_dev
