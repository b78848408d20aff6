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
	const float4 *rays;       // shade_local_brdf: the rays whose hits are shaded (w_o = -d)
	const float4 *params;     // shade_local_brdf: per material ior, roughness, brdf kind (bits)
	// albedo textures (all optional): per reported triangle id two float4 (a.tc b.tc | c.tc 0 0); per material
	// (first texel, w, h, 1) as integers or all zero; the texels of all textures, rgb in a float4 each
	const float4 *tex_coords;
	const uint4 *tex_of_material;
	const float4 *texels;
};

// diff_geom::albedo, libgi/rt.cpp:27-31: the material's colour, or its texture at the hit's texture coordinate
// (diff_geom::tc, libgi/rt.cpp:11, sampled like texture::sample, libgi/scene.h:24-32: wrap, nearest texel).
__device__ __forceinline__ float4 material_albedo(const ShadeArgs &a, float4 normal, float4 hit) {
	const uint32_t m = __float_as_uint(normal.w);
	if (m >= a.n_materials) return make_float4(0, 0, 0, 0);
	if (a.tex_of_material) {
		const uint4 t = a.tex_of_material[m];
		if (t.w) {
			const uint32_t id = __float_as_uint(hit.w);
			const float4 ab = a.tex_coords[2 * (size_t)id], c = a.tex_coords[2 * (size_t)id + 1];
			const float beta = hit.y, gamma = hit.z;
			const float alpha = 1.0f - beta - gamma;
			float u = alpha * ab.x + beta * ab.z + gamma * c.x;
			float v = alpha * ab.y + beta * ab.w + gamma * c.y;
			u = u - floorf(u);
			v = v - floorf(v);
			int x = (int)(u * (float)t.y + 0.5f);
			int y = (int)(v * (float)t.z + 0.5f);
			if (x == (int)t.y) x = 0;
			if (y == (int)t.z) y = 0;
			return a.texels[(size_t)t.x + (size_t)y * t.y + (size_t)x];
		}
	}
	return a.albedo[m];
}

// primary_hit_display::sample_pixel, gi/primary-hit.cpp:16-30: the albedo of what the ray hit, else black
// (diff_geom::albedo without a texture, libgi/rt.cpp:30-34)
__global__ __launch_bounds__(256) void shade_primary_kernel(ShadeArgs a) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= a.n) return;
	const float4 hit = a.hits[i];
	float4 c = make_float4(0, 0, 0, 0);
	if (hit.x != FLT_MAX) {
		c = material_albedo(a, a.normals[__float_as_uint(hit.w)], hit);
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
			const float4 alb = material_albedo(a, ns, hit);
			const float dd = d * d;
			c.x = a.power[0] * (one_over_pi * alb.x) / dd;
			c.y = a.power[1] * (one_over_pi * alb.y) / dd;
			c.z = a.power[2] * (one_over_pi * alb.z) / dd;
		}
	}
	a.out[i] = c;
}

// ---- the renderer's other BRDFs (libgi/material.cpp:7-75), evaluated like the reference does: same operations in
// the same order, so that everything but powf is bit-identical to a host evaluation ----
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) { return ax * bx + ay * by + az * bz; }
__device__ __forceinline__ float cdot3(float ax, float ay, float az, float bx, float by, float bz) {  // libgi/util.h:12-15
	const float x = ax * bx + ay * by + az * bz;
	return x < 0.0f ? 0.0f : x;
}
// fresnel_dielectric, libgi/util.h:44-60
__device__ __forceinline__ float fresnel_dielectric(float cos_wi, float ior_medium, float ior_material) {
	const float ei = cos_wi < 0.0f ? ior_material : ior_medium;
	const float et = cos_wi < 0.0f ? ior_medium : ior_material;
	cos_wi = fminf(fmaxf(fabsf(cos_wi), 0.0f), 1.0f);
	const float sin_t = (ei / et) * sqrtf(1.0f - cos_wi * cos_wi);
	if (sin_t >= 1.0f) return 1.0f;
	const float rev_sin2 = 1.0f - sin_t * sin_t;
	const float cos_t = sqrtf(rev_sin2 > 0.0f ? rev_sin2 : 0.0f);
	const float Rparl = ((et * cos_wi) - (ei * cos_t)) / ((et * cos_wi) + (ei * cos_t));
	const float Rperp = ((ei * cos_wi) - (et * cos_t)) / ((ei * cos_wi) + (et * cos_t));
	return (Rparl * Rparl + Rperp * Rperp) / 2;
}
__device__ __forceinline__ float tan2_theta(float c) {  // libgi/util.h:73-87
	const float res = 1.0f - c * c;
	return (res < 0.0f ? 0.0f : res) / (c * c);
}
__device__ __forceinline__ float ggx_d(float NdotH, float roughness) {  // libgi/material.cpp:40-46
	if (NdotH <= 0) return 0.f;
	const float tan2 = tan2_theta(NdotH);
	if (!isfinite(tan2)) return 0.f;
	const float a2 = roughness * roughness;
	const float pi = (float)3.14159265358979323846;
	const float n2 = NdotH * NdotH, s = a2 + tan2;
	return a2 / (pi * (n2 * n2) * (s * s));
}
__device__ __forceinline__ float ggx_g1(float NdotV, float roughness) {  // libgi/material.cpp:48-53
	if (NdotV <= 0) return 0.f;
	const float tan2 = tan2_theta(NdotV);
	if (!isfinite(tan2)) return 0.f;
	return 2.f / (1.f + sqrtf(1.f + (roughness * roughness) * tan2));
}

enum { BRDF_LAMBERT = 0, BRDF_PHONG = 1, BRDF_LAYERED_PHONG = 2, BRDF_GTR2 = 3, BRDF_LAYERED_GTR2 = 4 };

// lambertian_reflection::f, libgi/material.cpp:17-20
__device__ __forceinline__ void brdf_lambert(float nx, float ny, float nz, float wix, float wiy, float wiz, float4 alb, float &r, float &g,
                                             float &b) {
	r = g = b = 0.0f;
	if (!(dot3(nx, ny, nz, wix, wiy, wiz) > 0.0f)) return;  // same_hemisphere(w_i, ns): dot(w_i, ns), products commute
	const float one_over_pi = (float)(1.0 / 3.14159265358979323846);
	r = one_over_pi * alb.x, g = one_over_pi * alb.y, b = one_over_pi * alb.z;
}
// phong_specular_reflection::f, libgi/material.cpp:25-32 (ng = ns, libgi/rt.cpp:13)
__device__ __forceinline__ void brdf_phong(float nx, float ny, float nz, float wox, float woy, float woz, float wix, float wiy, float wiz,
                                           float4 alb, float roughness, bool coat, float &r, float &g, float &b) {
	r = g = b = 0.0f;
	if (!(dot3(wix, wiy, wiz, nx, ny, nz) > 0.0f)) return;
	const float exponent = 2 / (roughness * roughness) - 2;  // exponent_from_roughness, libgi/material.h:15-17
	const float d = dot3(wix, wiy, wiz, nx, ny, nz);
	const float rx = 2.0f * nx * d - wix, ry = 2.0f * ny * d - wiy, rz = 2.0f * nz * d - wiz;
	const float cos_theta = cdot3(wox, woy, woz, rx, ry, rz);
	const float one_over_2pi = (float)(1.0 / (2 * 3.14159265358979323846));
	const float norm_f = (exponent + 2.0f) * one_over_2pi;
	const float p = powf(cos_theta, exponent), c = cdot3(wix, wiy, wiz, nx, ny, nz);
	r = (coat ? 1.0f : alb.x) * p * norm_f * c;
	g = (coat ? 1.0f : alb.y) * p * norm_f * c;
	b = (coat ? 1.0f : alb.z) * p * norm_f * c;
}
// gtr2_reflection::f, libgi/material.cpp:58-73
__device__ __forceinline__ void brdf_gtr2(float nx, float ny, float nz, float wox, float woy, float woz, float wix, float wiy, float wiz,
                                          float4 alb, float roughness, float ior, bool coat, float &r, float &g, float &b) {
	r = g = b = 0.0f;
	if (!(dot3(nx, ny, nz, wix, wiy, wiz) > 0.0f)) return;
	const float NdotV = cdot3(nx, ny, nz, wox, woy, woz);
	const float NdotL = cdot3(nx, ny, nz, wix, wiy, wiz);
	if (NdotV == 0.f) return;
	const float hx = wox + wix, hy = woy + wiy, hz = woz + wiz;
	const float inv = 1.0f / sqrtf(dot3(hx, hy, hz, hx, hy, hz));
	const float Hx = hx * inv, Hy = hy * inv, Hz = hz * inv;
	const float NdotH = cdot3(nx, ny, nz, Hx, Hy, Hz);
	const float HdotL = cdot3(Hx, Hy, Hz, wix, wiy, wiz);
	const float F = fresnel_dielectric(HdotL, 1.f, ior);
	const float D = ggx_d(NdotH, roughness);
	const float G = ggx_g1(NdotV, roughness) * ggx_g1(NdotL, roughness);
	const float microfacet = (F * D * G) / (4 * fabsf(NdotV) * fabsf(NdotL));
	r = coat ? microfacet : alb.x * microfacet;
	g = coat ? microfacet : alb.y * microfacet;
	b = coat ? microfacet : alb.z * microfacet;
}

// local_illumination::sample_pixel, gi/primary-hit.cpp:32-57, with the BRDF of the hit's material (lambert, phong,
// gtr2, or one of the two under a lambertian base weighted by the Fresnel term: layered_brdf::f, libgi/material.cpp:7-12).
__global__ __launch_bounds__(256) void shade_local_brdf_kernel(ShadeArgs a) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= a.n) return;
	const float4 hit = a.hits[i];
	float4 c = make_float4(0, 0, 0, 0);
	if (hit.x != FLT_MAX && !a.occluded[i]) {
		float x, y, z;
		float4 A, B, C;
		hit_point(a.corners, hit, x, y, z, A, B, C);
		const float4 ns = a.normals[__float_as_uint(hit.w)];
		const uint32_t m = __float_as_uint(ns.w);
		const float4 alb = material_albedo(a, ns, hit);
		const float4 par = m < a.n_materials ? a.params[m] : make_float4(1.3f, 0.1f, 0.0f, 0.0f);
		const uint32_t kind = __float_as_uint(par.z);
		const float lx = a.light[0] - x, ly = a.light[1] - y, lz = a.light[2] - z;
		const float d = sqrtf(dot3(lx, ly, lz, lx, ly, lz));
		const float inv = 1.0f / d;
		const float wix = lx * inv, wiy = ly * inv, wiz = lz * inv;
		const float4 r0 = a.rays[2 * (size_t)i], r1 = a.rays[2 * (size_t)i + 1];
		const float wox = -r0.w, woy = -r1.x, woz = -r1.y;
		float fr, fg, fb;
		if (kind == BRDF_LAMBERT)
			brdf_lambert(ns.x, ns.y, ns.z, wix, wiy, wiz, alb, fr, fg, fb);
		else if (kind == BRDF_PHONG)
			brdf_phong(ns.x, ns.y, ns.z, wox, woy, woz, wix, wiy, wiz, alb, par.y, false, fr, fg, fb);
		else if (kind == BRDF_GTR2)
			brdf_gtr2(ns.x, ns.y, ns.z, wox, woy, woz, wix, wiy, wiz, alb, par.y, par.x, false, fr, fg, fb);
		else {
			float dr, dg, db, sr, sg, sb;
			const float dn = dot3(ns.x, ns.y, ns.z, wox, woy, woz);
			const float F = fresnel_dielectric(dn < 0.0f ? -dn : dn, 1.0f, par.x);  // absdot(ns, w_o)
			brdf_lambert(ns.x, ns.y, ns.z, wix, wiy, wiz, alb, dr, dg, db);
			if (kind == BRDF_LAYERED_PHONG)
				brdf_phong(ns.x, ns.y, ns.z, wox, woy, woz, wix, wiy, wiz, alb, par.y, true, sr, sg, sb);
			else
				brdf_gtr2(ns.x, ns.y, ns.z, wox, woy, woz, wix, wiy, wiz, alb, par.y, par.x, true, sr, sg, sb);
			fr = (1.0f - F) * dr + F * sr, fg = (1.0f - F) * dg + F * sg, fb = (1.0f - F) * db + F * sb;
		}
		const float dd = d * d;
		c.x = a.power[0] * fr / dd, c.y = a.power[1] * fg / dd, c.z = a.power[2] * fb / dd;
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
