/* rtgi_hip.h -- C ABI of the MI355X ray-traversal backend for the rtgi renderer.
 *
 * One handle = one tracer object bound to one GPU.  It plays the role of a
 * binary_bvh_tracer<layout, esc> (rt/bbvh-base/bvh.h:29-107) behind the ray_tracer /
 * individual_ray_tracer / wf::batch_ray_tracer interfaces (libgi/rt.h:91-107,
 * libgi/wavefront-rt.h:10-16): build() on the host, closest_hit / any_hit on the GPU.
 *
 * Conventions: every function returns RTGI_OK (0) or a negative RTGI_ERR_* code, never throws and
 * never aborts; rtgi_hip_last_error() gives the text.  Calls on one handle must be serialised by the
 * calling threads; different handles may be used from different host threads.  On the GPU, launches of
 * one handle share its scratch buffers: the library orders them itself, also across streams (a launch
 * on another stream than the previous one waits for it on the GPU), so they never overlap.  Pointers named d_* are device pointers on the handle's GPU, all others are
 * host pointers.  File:line citations are relative to the reference tree
 * "Assignment 3/rtgi-2021-a03-ref/".
 */
#ifndef RTGI_HIP_H
#define RTGI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtgi_hip_tracer rtgi_hip_tracer;

/* struct ray, libgi/rt.h:20-27, without the derived member id = 1/d (the kernels recompute it with the
 * same IEEE division the constructor uses). 32 bytes. */
typedef struct rtgi_ray {
	float o[3];
	float d[3];
	float t_min; /* reference default 1e-4f */
	float t_max; /* reference default FLT_MAX; length_exclusive(len) stores len - 1e-4f */
} rtgi_ray;

/* struct triangle_intersection, libgi/rt.h:40-62. t == FLT_MAX means "no hit" (then ref, beta, gamma
 * are 0). 16 bytes. */
typedef struct rtgi_hit {
	float t, beta, gamma;
	uint32_t ref;
} rtgi_hit;

/* binary_bvh_tracer::node, rt/bbvh-base/bvh.h:39-47: box = min.xyz, max.xyz; a leaf has link_r < 0,
 * then -link_l is the first slot in the triangle index and -link_r the triangle count. 56 bytes. */
typedef struct rtgi_node56 {
	float box_l[6];
	float box_r[6];
	int32_t link_l, link_r;
} rtgi_node56;

enum {
	RTGI_OK = 0,
	RTGI_ERR_INVALID = -1,   /* bad argument, malformed scene or tree */
	RTGI_ERR_DEVICE = -2,    /* a HIP call failed */
	RTGI_ERR_NOT_BUILT = -3, /* traversal before build()/upload */
	RTGI_ERR_NOMEM = -4,
	RTGI_ERR_SYNTAX = -5     /* interprete(): command recognised, arguments malformed */
};
enum { RTGI_LAYOUT_FLAT = 0, RTGI_LAYOUT_INDEXED = 1 }; /* bvh.h:29 bbvh_triangle_layout */
enum { RTGI_SPLIT_OM = 0, RTGI_SPLIT_SM = 1, RTGI_SPLIT_SAH = 2 }; /* bvh.h:62 binary_split_type */

/* Number of usable GPUs (0 if none; never fails). */
int rtgi_hip_device_count(void);

/* "raytracer bbvh [indexed] [esc]", driver/interaction.cpp:177-203: creates the tracer the registry
 * would create with `new binary_bvh_tracer<layout, esc>`. flat + esc is refused like the registry does
 * (interaction.cpp:195-196). device = -1 gives a host-only handle: it builds, takes over and reads back
 * trees but every traversal call fails with RTGI_ERR_DEVICE. Returns NULL on failure. */
rtgi_hip_tracer *rtgi_hip_create(int device, int layout, int esc);

/* ~ray_tracer(), libgi/rt.h:97. Frees all host and device memory of the handle. */
void rtgi_hip_destroy(rtgi_hip_tracer *h);

/* ray_tracer::interprete(command, in), bvh2.cpp:503-555: `args` is the rest of the command line.
 * Handles "bvh om | sm | sah N | triangles single | triangles multiple N | statistics | export D file" and
 * the backend's own "hip reorder auto | on | off" (trace batches of 65536 rays and more in a spatially sorted
 * order: where the scene makes incoherent rays expensive, always, never; default auto; results do not
 * depend on it).
 * Returns 1 if the command was consumed, 0 if it is not for this tracer, RTGI_ERR_SYNTAX if it was
 * recognised but malformed (the reference prints an error and still reports "consumed"). */
int rtgi_hip_interprete(rtgi_hip_tracer *h, const char *command, const char *args);

/* ray_tracer::build(scene*), bvh2.cpp:27-63: builds the tree from scene->vertices / scene->triangles
 * (libgi/scene.h:80-81) with the split scheme selected through interprete(), then flattens it for the
 * GPU and uploads it. `vertices` points at the first position (3 floats) of records vertex_stride bytes
 * apart (sizeof(vertex) = 32, rt.h:29-33); `triangles` at the first of three uint32 vertex ids of
 * records tri_stride bytes apart (sizeof(triangle) = 16, rt.h:35-38). Non-finite positions and
 * out-of-range vertex ids are refused. The scene arrays are not modified: for the flat layout, apply
 * rtgi_hip_get_triangle_order() to scene->triangles afterwards (commit_shuffled_triangles,
 * bvh2.cpp:172-180). */
int rtgi_hip_build(rtgi_hip_tracer *h, const void *vertices, size_t vertex_stride, uint32_t n_vertices,
                   const void *triangles, size_t tri_stride, uint32_t n_triangles);

/* build() for a tracer that spans several GPUs (`raytracer hip devices N`: one handle per GPU, every GPU with the
 * whole BVH; no counterpart in the reference, which has one tracer in one process): the tree is built ONCE on the
 * host from the settings of handles[0] (all handles must have been created and configured alike), flattened once,
 * and uploaded to every handle's GPU, the uploads running side by side. Each handle afterwards is in the state
 * rtgi_hip_build leaves it in. Errors are reported on handles[0]. */
int rtgi_hip_build_shared(rtgi_hip_tracer *const *handles, uint32_t n_handles, const void *vertices, size_t vertex_stride,
                          uint32_t n_vertices, const void *triangles, size_t tri_stride, uint32_t n_triangles);

/* The GPU a handle is bound to (the `device` given to rtgi_hip_create; -1 for a host-only handle or NULL). */
int rtgi_hip_device_of(const rtgi_hip_tracer *h);

/* Takes a tree built elsewhere (e.g. by the reference's own builder) verbatim: `nodes`/`root` as in
 * binary_bvh_tracer::nodes/root (bvh.h:59-68); `index` = the leaf-order triangle list, i.e.
 * binary_bvh_tracer::index for the indexed layout, NULL (identity over the already permuted
 * scene->triangles) for the flat layout. The tree is checked (links in range, every node reached once,
 * leaf ranges inside the index) before anything is uploaded. */
int rtgi_hip_upload_bvh(rtgi_hip_tracer *h, const rtgi_node56 *nodes, uint32_t n_nodes, uint32_t root,
                        const uint32_t *index, uint32_t n_index, const void *vertices, size_t vertex_stride,
                        uint32_t n_vertices, const void *triangles, size_t tri_stride, uint32_t n_triangles);

/* Read-back of the host tree in the reference's own format (bvh.h:59-68), e.g. for `bvh export`. */
uint32_t rtgi_hip_num_nodes(const rtgi_hip_tracer *h);
uint32_t rtgi_hip_num_index(const rtgi_hip_tracer *h); /* leaf-order slots (> n_triangles with esc) */
uint32_t rtgi_hip_root(const rtgi_hip_tracer *h);
uint32_t rtgi_hip_depth(const rtgi_hip_tracer *h);     /* deepest leaf, root = 0 */
/* What "hip reorder auto" decides by (no counterpart in the reference): the mean number of inner nodes a
 * closest_hit ray pops when it starts on a random triangle in a random direction, measured on the GPU at
 * build / upload with 32768 seeded rays. Batches are sorted before tracing where this plus twice the triangles such a
 * ray tests exceeds 120; where it is below 100 waves draw new rays in larger portions (short rays, frequent refills).
 * 0 for host-only handles and scenes of fewer than 4096 leaf slots. */
float rtgi_hip_incoherent_ray_cost(const rtgi_hip_tracer *h);
int rtgi_hip_get_nodes(const rtgi_hip_tracer *h, rtgi_node56 *out);
/* Leaf-order list of ORIGINAL triangle ids (index[] of the indexed layout; for the flat layout the
 * permutation new_tris[i] = old_tris[out[i]] of bvh2.cpp:176-179). */
int rtgi_hip_get_triangle_order(const rtgi_hip_tracer *h, uint32_t *out);

/* wf::batch_ray_tracer::compute_closest_hit(), libgi/wavefront-rt.h:67-73, over n rays in host memory:
 * hits[i] = closest_hit(rays[i]) (bvh2.cpp:415-463). ref is the id the reference would return: an
 * index into the permuted triangle list for the flat layout, an original id for the indexed one.
 * Synchronous. Batches of more than 256 rays go through the GPU in chunks on three streams (rays of chunk k + 1 on
 * their way up while chunk k is traced; the kernel stores hit records straight into page-locked host memory).
 * Buffers in ordinary memory are staged through page-locked slots by a few copying threads; buffers from
 * rtgi_hip_host_alloc / rtgi_hip_host_register are used in place, and batches of 2^19 rays and more in such buffers
 * are traced by ONE launch that begins when the first eighth of the batch has arrived and takes up the rest as it
 * comes (one tail instead of one per chunk; rtgi_hip_host_path_stats counts these launches). */
int rtgi_hip_closest_hit(rtgi_hip_tracer *h, const rtgi_ray *rays, uint32_t n, rtgi_hit *hits);

/* compute_any_hit(), wavefront-rt.h:74-80 / bvh2.cpp:465-501: occluded[i] = any_hit(rays[i]) as 0/1. */
int rtgi_hip_any_hit(rtgi_hip_tracer *h, const rtgi_ray *rays, uint32_t n, uint8_t *occluded);

/* What "hip reorder auto" did (no counterpart in the reference): *launches = traversal launches of the device entry points on
 * this handle so far (rtgi_hip_closest_hit_device, rtgi_hip_any_hit_device, rtgi_hip_trace_batches_device, and the chunks of
 * host batches traced in separate launches), *sorted = those whose rays were traced in sorted order. With "auto" a scene
 * whose incoherent rays are expensive (rtgi_hip_incoherent_ray_cost) has its batches sorted, except a batch that
 * rtgi_hip_camera_rays_device / rtgi_hip_camera_rays_pixels_device of this handle has just written (same buffer, same
 * number of rays): camera rays lie in image order and share their origin, and sorting them costs more than it gains
 * (2^20-triangle soup, 1080p: 1181 against 1118 Mrays/s). Either pointer may be NULL. */
int rtgi_hip_launch_stats(const rtgi_hip_tracer *h, uint64_t *launches, uint64_t *sorted);

/* How the page-locked batches of the two calls above went (no counterpart in the reference): *one_launch = batches traced
 * by one launch beside their own upload; *gave_up = those among them whose launch waited in vain for a part of the
 * batch (its upload or sort found no room on the GPU beside the launch, e.g. while other processes use it) and which
 * were then traced again chunk by chunk: the results are the same, the time (20 ms) is lost. After the second such batch the
 * handle stops using the one-launch path. Either pointer may be NULL. */
int rtgi_hip_host_path_stats(const rtgi_hip_tracer *h, uint64_t *one_launch, uint64_t *gave_up);

/* Page-locked host memory for the host-buffer entry points above (no counterpart in the reference). Rays and
 * results in ordinary (pageable) memory are copied through a staging area by the HIP runtime; page-locked memory
 * is read and written by the GPU's copy engines directly, which about doubles the rate of the copies.
 * rtgi_hip_host_alloc returns page-locked memory (NULL on failure), to be released with rtgi_hip_host_free;
 * rtgi_hip_host_register page-locks memory the caller already owns (e.g. the renderer's intersection buffer,
 * libgi/wavefront-rt.h:33-34) until rtgi_hip_host_unregister. Using them is optional. */
void *rtgi_hip_host_alloc(size_t bytes);
void rtgi_hip_host_free(void *p);
int rtgi_hip_host_register(void *p, size_t bytes);
int rtgi_hip_host_unregister(void *p);

/* The same two operations on rays that already live on the GPU (the wavefront pipeline's form):
 * asynchronous on `stream` (a hipStream_t, NULL = default stream). d_counts, if not NULL, receives two
 * uint32 per ray: inner nodes popped and triangles tested (what a COUNT_HITS build, bvh.h:10, counts,
 * minus the leaf-node pops). */
int rtgi_hip_closest_hit_device(rtgi_hip_tracer *h, const rtgi_ray *d_rays, uint32_t n, rtgi_hit *d_hits,
                                uint32_t *d_counts, void *stream);
int rtgi_hip_any_hit_device(rtgi_hip_tracer *h, const rtgi_ray *d_rays, uint32_t n, uint8_t *d_occluded,
                            uint32_t *d_counts, void *stream);

/* rtgi_hip_closest_hit_device for callers that cannot allocate GPU memory at the time of the call (code under
 * HIP graph capture, tight serving loops, harnesses that forbid it). It has no counterpart in the reference. Same
 * arguments (without per-ray counts) and bit-identical results, but it calls no allocating routine: its working
 * memory (ray counters, the four arrays and the scratch of the ray sort, the stack spill area of deep trees) is a
 * fixed 44 MiB array inside the library's code object, one copy per GPU, held until the process ends. At most
 * 2^21 = 2097152 rays per call, more is refused with RTGI_ERR_INVALID. Where something does not fit, the call adapts
 * instead of failing: a sort whose scratch would not fit is left out, and for trees deeper than the 24 stack levels that can be
 * kept in LDS the launch uses as many workgroups as the 8 MiB spill area gives stacks to. All calls on one GPU,
 * from any handle, share that GPU's array: calls that overlap in time on one GPU race; order them on one stream
 * or with events. */
int rtgi_hip_closest_hit_device_noalloc(rtgi_hip_tracer *h, const rtgi_ray *d_rays, uint32_t n, rtgi_hit *d_hits,
                                        void *stream);

/* Several batches in one launch (no counterpart in the reference, whose batch adapter takes one buffer at a
 * time, libgi/wavefront-rt.h:67-80): up to 32 batches of rays on the GPU, each either a closest_hit batch (d_out:
 * rtgi_hit per ray) or an any_hit batch (d_out: one byte per ray). Results are those of the single calls; what is
 * gained is that the batches share one ray supply, so the lanes that the end of one batch leaves idle work on the
 * others (e.g. the shadow rays and the bounce rays of one sample). Empty batches are allowed. */
enum { RTGI_BATCH_CLOSEST_HIT = 0, RTGI_BATCH_ANY_HIT = 1 };
typedef struct rtgi_batch {
	const rtgi_ray *d_rays;
	void *d_out;
	uint32_t n;
	uint32_t kind;
} rtgi_batch;
int rtgi_hip_trace_batches_device(rtgi_hip_tracer *h, const rtgi_batch *batches, uint32_t n_batches, void *stream);

/* ---- ray populations around the tracer, produced on the GPU (the wavefront pipeline's other stages) ---- */

/* struct camera, libgi/camera.h:7-25 (near_w / near_h are derived from fovy, w, h as update_frustum does). */
typedef struct rtgi_camera {
	float dir[3], pos[3], up[3];
	float fovy;
	int32_t w, h;
} rtgi_camera;

/* batch_cam_ray_setup_cpu::run(), libgi/wavefront-rt.h:90-102: d_rays[y*w+x] = cam_ray(cam, x, y, offset)
 * (libgi/camera.cpp:11-22). d_offsets: w*h pairs of sub-pixel offsets (what the caller's rng would supply,
 * uniform in [-0.5, 0.5)), or NULL for pixel centres. Needs no build(). */
int rtgi_hip_camera_rays_device(rtgi_hip_tracer *h, const rtgi_camera *cam, const float *d_offsets, rtgi_ray *d_rays,
                                void *stream);

/* The sub-pixel offsets of one sample for n pixels, uniform in [-0.5, 0.5) like the two rng draws per pixel of
 * gi/primary-hit.cpp:20, from a counter-based generator keyed by (pixel, sample, seed): input for
 * rtgi_hip_camera_rays_device. d_offsets: n pairs of floats. Needs no build(). */
int rtgi_hip_pixel_offsets_device(rtgi_hip_tracer *h, uint32_t n, uint32_t sample, uint32_t seed, float *d_offsets, void *stream);

/* The same two for a part of the frame (one GPU's tiles of a frame that several GPUs share): ray i and offset i
 * belong to pixel d_pixels[i] = y * w + x of the camera's image, n of them, each below w * h. A pixel's offsets and
 * its ray are those of the whole-frame calls, whichever part of the frame it is traced in. */
int rtgi_hip_pixel_offsets_pixels_device(rtgi_hip_tracer *h, const uint32_t *d_pixels, uint32_t n, uint32_t sample, uint32_t seed,
                                         float *d_offsets, void *stream);
int rtgi_hip_camera_rays_pixels_device(rtgi_hip_tracer *h, const rtgi_camera *cam, const uint32_t *d_pixels, uint32_t n,
                                       const float *d_offsets, rtgi_ray *d_rays, void *stream);

/* Shadow rays of local_illumination::sample_pixel, gi/primary-hit.cpp:43-50, one slot per hit record: origin
 * diff_geom::x (libgi/rt.cpp:9), direction towards light_pos, t_max = distance - 1e-4 (ray::length_exclusive).
 * Slots whose hit record is a miss get an empty ray (t_max < t_min), which any_hit reports as not occluded. */
int rtgi_hip_shadow_rays_device(rtgi_hip_tracer *h, const rtgi_hit *d_hits, uint32_t n, const float light_pos[3],
                                rtgi_ray *d_shadow, void *stream);

/* ---- shading and accumulation on the GPU: the callers after the tracer ---------------------------------- */

/* The part of the scene the shading stages read, given once after build(): per-vertex normals (3 floats,
 * `normal_stride` bytes apart: &scene.vertices[0].norm, 32), the triangles as passed to build() with the
 * material index as fourth uint32 (libgi/rt.h:35-38, stride 16), and one albedo per material
 * (material::albedo, libgi/material.h:54; textures: rtgi_hip_set_textures). Stores, per triangle id as hits
 * report it, the shading normal normalize(a.norm + b.norm + c.norm) of diff_geom (libgi/rt.cpp:10). */
int rtgi_hip_set_shading(rtgi_hip_tracer *h, const void *vertex_normals, size_t normal_stride, uint32_t n_vertices,
                         const void *triangles, size_t tri_stride, uint32_t n_triangles, const float *albedo_rgb,
                         uint32_t n_materials);

/* primary_hit_display::sample_pixel, gi/primary-hit.cpp:16-30, for a batch: d_radiance[i] = albedo of the
 * triangle d_hits[i] names, black for a miss. d_radiance: n x float4 (rgb, 0). */
int rtgi_hip_shade_primary_device(rtgi_hip_tracer *h, const rtgi_hit *d_hits, uint32_t n, float *d_radiance, void *stream);

/* local_illumination::sample_pixel, gi/primary-hit.cpp:32-57, for a batch, a point light (pointlight,
 * libgi/scene.h:58-63: power = 4 pi col) and the lambertian BRDF (libgi/material.cpp:17-20): d_occluded is
 * any_hit's answer for the rays of rtgi_hip_shadow_rays_device on the same hits. */
int rtgi_hip_shade_local_device(rtgi_hip_tracer *h, const rtgi_hit *d_hits, const uint8_t *d_occluded, uint32_t n,
                                const float light_pos[3], const float light_col[3], float *d_radiance, void *stream);

/* struct material, libgi/material.h:54-61, as far as local illumination reads it, and the brdf names of new_brdf
 * (libgi/material.cpp:77-102: "lambert", "phong", "layered-phong", "gtr2", "layered-gtr2"). Albedo textures are
 * given by rtgi_hip_set_textures. */
enum { RTGI_BRDF_LAMBERT = 0, RTGI_BRDF_PHONG = 1, RTGI_BRDF_LAYERED_PHONG = 2, RTGI_BRDF_GTR2 = 3, RTGI_BRDF_LAYERED_GTR2 = 4 };
typedef struct rtgi_material {
	float albedo[3];
	float ior;       /* default 1.3 */
	float roughness; /* default 0.1 */
	uint32_t brdf;   /* RTGI_BRDF_* */
} rtgi_material;

/* `material albedo | ior | roughness | brdf ...` and `default-brdf` (driver/interaction.cpp:250-341): replaces the
 * material table given to rtgi_hip_set_shading (which sets ior 1.3, roughness 0.1, lambert), same number of
 * materials. Waits for the GPU; no rebuild needed. */
int rtgi_hip_set_materials(rtgi_hip_tracer *h, const rtgi_material *materials, uint32_t n_materials);

/* Albedo textures: material::albedo_tex (libgi/material.h:57) as diff_geom::albedo reads it (libgi/rt.cpp:27-31): a
 * material with a texture takes its albedo from the texel at the hit's interpolated texture coordinate (diff_geom::tc,
 * libgi/rt.cpp:11; texture::sample, libgi/scene.h:24-32: coordinates wrap, nearest texel), in every shading stage.
 * texel[y * w + x] as in struct texture (libgi/scene.h:16-32): linear rgb, rows as the loader left them (load_image3f
 * flips the image and raises it to the power 2.2, libgi/scene.cpp:45-68; that is the caller's business here).
 * vertex_tc: first vertex::tc (2 floats, tc_stride bytes apart: &scene.vertices[0].tc, 32); triangles as given to
 * rtgi_hip_build; material_texture: per material an index into `textures`, or -1 (`material texture drop`). Call
 * after rtgi_hip_set_shading (which forgets the textures); n_textures = 0 drops them all. The data is copied. */
typedef struct rtgi_texture {
	const float *rgb;
	uint32_t w, h;
} rtgi_texture;
int rtgi_hip_set_textures(rtgi_hip_tracer *h, const void *vertex_tc, size_t tc_stride, uint32_t n_vertices, const void *triangles,
                          size_t tri_stride, uint32_t n_triangles, const rtgi_texture *textures, uint32_t n_textures,
                          const int32_t *material_texture, uint32_t n_materials);

/* rtgi_hip_shade_local_device with the BRDF of each hit's material (brdf::f, libgi/material.cpp:7-75): needs the
 * rays the hits belong to, for w_o = -ray.d (gi/primary-hit.cpp:45). For lambertian materials the result is
 * bit-identical to rtgi_hip_shade_local_device. */
int rtgi_hip_shade_local_brdf_device(rtgi_hip_tracer *h, const rtgi_ray *d_rays, const rtgi_hit *d_hits, const uint8_t *d_occluded,
                                     uint32_t n, const float light_pos[3], const float light_col[3], float *d_radiance, void *stream);

/* framebuffer::add, libgi/framebuffer.cpp:10-18, one sample for each of n pixels: d_framebuffer is n x float4
 * (running mean rgb, sample count in w), zero-filled by the caller like framebuffer::clear. */
int rtgi_hip_framebuffer_add_device(rtgi_hip_tracer *h, float *d_framebuffer, const float *d_radiance, uint32_t n, void *stream);

/* Combining the framebuffers of several GPUs (no counterpart in the reference, which renders in one process). A
 * framebuffer entry is a running mean with its sample count in w (libgi/framebuffer.cpp:10-18), so parts are
 * combined as sums and counts. scatter: d_full[d_pixels[i]] = d_part[i], a GPU's densely stored pixels into their
 * places in a frame-sized buffer (zero elsewhere). sums: in place, (mean, count) -> (mean * count, count) if to_sums,
 * else back (pixels without samples stay zero). accumulate: d_dst[i] += d_src[i] on all four components. */
int rtgi_hip_framebuffer_scatter_device(rtgi_hip_tracer *h, const float *d_part, const uint32_t *d_pixels, uint32_t n, float *d_full,
                                        void *stream);
int rtgi_hip_framebuffer_sums_device(rtgi_hip_tracer *h, float *d_framebuffer, uint32_t n, int to_sums, void *stream);
int rtgi_hip_framebuffer_accumulate_device(rtgi_hip_tracer *h, float *d_dst, const float *d_src, uint32_t n, void *stream);

/* framebuffer::png, libgi/framebuffer.cpp:21-30, without the file: 8-bit rgb, clamped, gamma 1/2.2, rows
 * flipped. d_rgb: 3 * w * h bytes. */
int rtgi_hip_framebuffer_rgb8_device(rtgi_hip_tracer *h, const float *d_framebuffer, uint32_t w, uint32_t height,
                                     uint8_t *d_rgb, void *stream);

/* One diffuse bounce (no counterpart in the reference, which has no path tracer): for every ray that hit
 * something, a ray from the hit point into a cosine-weighted direction about the geometric normal facing the
 * incoming ray, random numbers keyed by (ray index, bounce, seed). The new rays are written densely to d_out in
 * the order of their parent rays (wave ballot/popcount compaction + one scan); d_parent (optional) receives
 * each one's parent index, *d_count their number. d_out needs room for n rays. */
int rtgi_hip_bounce_rays_device(rtgi_hip_tracer *h, const rtgi_ray *d_rays, const rtgi_hit *d_hits, uint32_t n,
                                uint32_t seed, uint32_t bounce, rtgi_ray *d_out, uint32_t *d_parent, uint32_t *d_count,
                                void *stream);

/* Paths of several segments (no counterpart in the reference, whose algorithms stop at the first hit; used by the
 * frame loop of rtgi_hip_frame.h for "algo path"). A ray of a later segment carries its path's throughput (the
 * product of the albedos met so far: cosine-weighted sampling of a lambertian surface weighs by the albedo) and
 * the slot of the pixel sample it belongs to. All buffers: float4 / uint32 per ray.
 * begin: for the n camera rays of a sample, throughput = 1, slot = i, radiance = 0.
 * add: radiance[slot[j]] += throughput[j] * direct[j] for the m rays of a segment (direct = what
 *   rtgi_hip_shade_local_device returned for them).
 * extend: for the m rays that rtgi_hip_bounce_rays_device made (ray j continues ray d_parent[j]):
 *   throughput_out[j] = throughput_in[parent] * albedo[parent], slot_out[j] = slot_in[parent]
 *   (albedo = what rtgi_hip_shade_primary_device returned for the parent rays' hits). */
int rtgi_hip_path_begin_device(rtgi_hip_tracer *h, uint32_t n, float *d_throughput, uint32_t *d_slot, float *d_radiance, void *stream);
int rtgi_hip_path_add_device(rtgi_hip_tracer *h, const float *d_direct, const float *d_throughput, const uint32_t *d_slot, uint32_t m,
                             float *d_radiance, void *stream);
int rtgi_hip_path_extend_device(rtgi_hip_tracer *h, const float *d_albedo, const uint32_t *d_parent, uint32_t m,
                                const float *d_throughput_in, const uint32_t *d_slot_in, float *d_throughput_out, uint32_t *d_slot_out,
                                void *stream);

/* rtgi_hip_bounce_rays_device with the random numbers of ray i keyed by d_keys[i] (or, with a table, by
 * d_key_table[d_keys[i]]) instead of by i: keyed by its pixel, a path draws the same numbers however the frame's
 * rays are split into batches or over GPUs. Either pointer may be NULL. */
int rtgi_hip_bounce_rays_keyed_device(rtgi_hip_tracer *h, const rtgi_ray *d_rays, const rtgi_hit *d_hits, uint32_t n,
                                      const uint32_t *d_keys, const uint32_t *d_key_table, uint32_t seed, uint32_t bounce,
                                      rtgi_ray *d_out, uint32_t *d_parent, uint32_t *d_count, void *stream);

/* Timing of the traversal kernel alone (what the reference's time_this_block(closest_hit), bvh2.cpp:417,
 * measures per call, here per launch): after profile_begin every closest_hit / any_hit launch is bracketed by
 * HIP events on its stream; profile_end waits for them and returns the summed kernel time and the number of
 * launches (at most 4096 are kept). Sorting and copies are not included. */
int rtgi_hip_profile_begin(rtgi_hip_tracer *h);
int rtgi_hip_profile_end(rtgi_hip_tracer *h, double *kernel_ms_total, uint32_t *launches);
/* Work done by the counting launches (closest_hit / any_hit with d_counts) since the last profile_begin, summed
 * over their rays: out[0] rays, out[1] inner nodes popped, out[2] triangles tested (what a COUNT_HITS build of the
 * reference counts per ray, bvh.h:10, bvh2.cpp:422-461), out[3] the inner-node visits among out[1] that were served
 * from the kernel's copy of the top of the tree in LDS instead of the vector cache. Waits for the GPU. */
int rtgi_hip_profile_work(rtgi_hip_tracer *h, uint64_t out[4]);

/* More about the counting launches since the last profile_begin, and about the handle's stack (any pointer may be NULL;
 * waits for the GPU if a count is asked for):
 *   *lds_levels / *spill_levels: the traversal keeps as many pending nodes per ray as the tree is deep (the reference has
 *     25 slots per ray and does not check them, bvh2.cpp:419,469): the first *lds_levels in LDS and, for deeper trees,
 *     *spill_levels more in a global area;
 *   *spill_pushes: pushes that went to the global area (0 for a tree that fits in LDS);
 *   *link_fetches: inner-node visits that had to fetch the node's two links from memory. The levels at the top of the tree
 *     that hold inner nodes only are numbered like a heap, so most visits compute the children's indices instead. */
int rtgi_hip_profile_details(rtgi_hip_tracer *h, uint64_t *spill_pushes, uint64_t *link_fetches, uint32_t *lds_levels,
                             uint32_t *spill_levels);

/* Text of the last error on this handle (h may be NULL: last error of a failed create). */
const char *rtgi_hip_last_error(const rtgi_hip_tracer *h);

#ifdef __cplusplus
}
#endif
#endif /* RTGI_HIP_H */
