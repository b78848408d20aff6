/* rtgi_hip_frame.h -- C ABI of the frame level of the MI355X backend: whole frames of the renderer's wavefront
 * algorithms on one GPU or on several (librtgi_hip_frame.so; needs librtgi_hip.so and, for several GPUs, RCCL).
 *
 * What it stands for in the renderer ("Assignment 3/rtgi-2021-a03-ref/" unless a path starts with "Assignment"):
 *   - `raytracer hip [indexed] [esc] [devices N]`: the tracer registry entry (driver/interaction.cpp:177-203) for a
 *     tracer that holds one rtgi_hip.h handle per GPU, every GPU with the whole BVH;
 *   - `commit` -> build (driver/interaction.cpp:204-212), `bvh ...` / `hip ...` -> interprete (:399-400);
 *   - `run` -> render: run(algo) of driver/main.cpp:38-48 for the batch algorithms, i.e. framebuffer.clear, sppx
 *     samples of every pixel, framebuffer::add per sample, and the 8-bit image of framebuffer::png, with the loop
 *     of wavefront_algorithm ("Assignment 2/rtgi-2021-a02/gi/primary-hit.cpp:61-74": set up camera rays, trace the
 *     batch, shade the batch) kept on the GPU from the first ray to the last pixel.
 *
 * Several GPUs: rays are independent (rt/bbvh-base/bvh2.cpp:415-501 reads only the tree and the scene), so a frame
 * is split by 64x64-pixel tiles dealt round-robin, or by sample index, one host thread and one stream per GPU, no
 * traffic between GPUs while tracing. The only exchange is the framebuffer at the end of the frame: one RCCL
 * all-reduce over xGMI of sums and counts (framebuffer::add keeps a running mean with its count in .w,
 * libgi/framebuffer.cpp:10-18; means cannot be added). With tiles a pixel has samples on one GPU only, and the
 * combined frame is bit-identical to the frame one GPU renders. A GPU traces several samples of its pixels per launch
 * (as many as make about 8 Mi camera rays, at most 31: a launch over the few pixels of an eighth of a frame would be
 * over before the GPU is busy); samples still enter the framebuffer in their order, the frame does not change.
 *
 * Conventions as in rtgi_hip.h: 0 or a negative RTGI_ERR_* code, never throws, never aborts; one renderer is used
 * from one host thread at a time.
 */
#ifndef RTGI_HIP_FRAME_H
#define RTGI_HIP_FRAME_H

#include "rtgi_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtgi_hip_renderer rtgi_hip_renderer;

/* `algo primary`: primary_hit_display, gi/primary-hit.cpp:16-30 (albedo of the closest hit).
 * `algo local`: local_illumination, gi/primary-hit.cpp:32-57 (point light, shadow ray, lambertian BRDF).
 * `algo path`: no counterpart in the reference, which has no path tracer: local illumination at every vertex of a
 * diffuse path of up to `bounces` further segments, directions cosine-weighted about the geometric normal
 * (rtgi_hip_bounce_rays_keyed_device, keyed by pixel), weight = the albedos met. */
enum { RTGI_ALGO_PRIMARY = 0, RTGI_ALGO_LOCAL = 1, RTGI_ALGO_PATH = 2 };
enum { RTGI_SHARD_TILES = 0, RTGI_SHARD_SAMPLES = 1 };

typedef struct rtgi_frame {
	rtgi_camera cam;     /* at / look / up / resolution */
	uint32_t sppx;       /* `sppx N`, rc->sppx (libgi/context.h:20) */
	uint32_t algo;       /* RTGI_ALGO_* */
	uint32_t bounces;    /* RTGI_ALGO_PATH: segments after the camera ray */
	float light_pos[3];  /* `pointlight pos x y z col r g b` (driver/interaction.cpp:241-254) */
	float light_col[3];
	uint32_t seed;       /* of the counter-based generator for pixel offsets and bounce directions */
	uint32_t jitter;     /* 0: pixel centres (the reference's rt_bench rays, driver/main.cpp:55-58), else random offsets */
	uint32_t sharding;   /* RTGI_SHARD_*; ignored on one GPU */
	uint32_t tile;       /* tile edge in pixels for RTGI_SHARD_TILES, 0 = 64 */
} rtgi_frame;

typedef struct rtgi_frame_stats {
	double ms_total;     /* wall time of render() */
	double ms_trace;     /* until the slowest GPU had finished its part of the frame */
	double ms_reduce;    /* combining the framebuffers (0 on one GPU) */
	uint64_t rays;       /* closest_hit + any_hit rays traced, all GPUs */
	int32_t n_devices;
	int32_t used_rccl;   /* 1: the framebuffers were combined by ncclAllReduce; 0: one GPU, or handles sharing a GPU */
} rtgi_frame_stats;

/* `raytracer hip [indexed] [esc] devices N`: one handle per entry of `devices` (n >= 1). The same GPU may be
 * named more than once (to rehearse the multi-GPU path on one GPU): the framebuffers are then added up by a kernel
 * on that GPU instead of RCCL, which refuses two ranks on one GPU. Returns NULL on failure
 * (rtgi_hip_renderer_last_error(NULL) has the text). */
rtgi_hip_renderer *rtgi_hip_renderer_create(const int *devices, int n, int layout, int esc);
/* The same tracer with one process per GPU (the launch of `torch.distributed.run`, MPI and the like): every process
 * creates the renderer of its own GPU as rank `rank` of `world`. The ranks find each other through a 128-byte id
 * that one process gets from rtgi_hip_renderer_unique_id and hands to the others by whatever means the launcher has
 * (a broadcast, a file); create_rank returns when all `world` processes have called it (ncclCommInitRank). render()
 * is then a collective call: every rank calls it with the same frame, renders its tiles (or samples), and receives
 * the whole frame through one ncclAllReduce; stats->rays counts this rank's rays only. A rank whose part fails still
 * takes part in a one-word status exchange first, so that all ranks return an error together instead of the others
 * waiting in the all-reduce; if the exchange itself fails the communicator is given up (ncclCommAbort) and the
 * renderer must be created again. */
int rtgi_hip_renderer_unique_id(void *out, size_t bytes);
rtgi_hip_renderer *rtgi_hip_renderer_create_rank(int device, int layout, int esc, int rank, int world, const void *unique_id);
void rtgi_hip_renderer_destroy(rtgi_hip_renderer *r);
int rtgi_hip_renderer_device_count(const rtgi_hip_renderer *r);
/* The i-th GPU's tracer handle, e.g. for the read-back calls of rtgi_hip.h (owned by the renderer). */
rtgi_hip_tracer *rtgi_hip_renderer_handle(rtgi_hip_renderer *r, int i);

/* ray_tracer::interprete on every handle (same return convention as rtgi_hip_interprete). */
int rtgi_hip_renderer_interprete(rtgi_hip_renderer *r, const char *command, const char *args);

/* ray_tracer::build + the shading data, arguments as rtgi_hip_build and rtgi_hip_set_shading: vertices = first
 * struct vertex (position at offset 0, normal at offset 12; sizeof(vertex) = 32, libgi/rt.h:29-33), triangles =
 * first struct triangle (a, b, c, material_id). The tree is built once on the host and uploaded to every GPU
 * (rtgi_hip_build_shared). */
int rtgi_hip_renderer_build(rtgi_hip_renderer *r, const void *vertices, size_t vertex_stride, uint32_t n_vertices,
                            const void *triangles, size_t tri_stride, uint32_t n_triangles, const float *albedo_rgb,
                            uint32_t n_materials);

/* `material ...` / `default-brdf` after `commit` (driver/interaction.cpp:250-341): rtgi_hip_set_materials on every
 * GPU. The frames' shading is brdf::f of each hit's material (libgi/material.cpp:7-75). */
int rtgi_hip_renderer_set_materials(rtgi_hip_renderer *r, const rtgi_material *materials, uint32_t n_materials);

/* `material texture <file> | drop` and the textures `load` finds (libgi/scene.cpp:128-134, driver/interaction.cpp:290-302):
 * rtgi_hip_set_textures on every GPU, same arguments (the scene arrays are those given to rtgi_hip_renderer_build). */
int rtgi_hip_renderer_set_textures(rtgi_hip_renderer *r, const void *vertex_tc, size_t tc_stride, uint32_t n_vertices, const void *triangles,
                                   size_t tri_stride, uint32_t n_triangles, const rtgi_texture *textures, uint32_t n_textures,
                                   const int32_t *material_texture, uint32_t n_materials);

/* `run`. framebuffer: w*h float4 (mean rgb, sample count) like rc->framebuffer.color, or NULL; rgb8: 3*w*h bytes like
 * framebuffer::png (top row first), or NULL; stats: or NULL. */
int rtgi_hip_renderer_render(rtgi_hip_renderer *r, const rtgi_frame *frame, float *framebuffer, uint8_t *rgb8,
                             rtgi_frame_stats *stats);

/* `rt_bench`, driver/main.cpp:50-65: one closest_hit per pixel on camera rays through the pixel centres of `frame`,
 * timed (the trace call alone, rays already on the GPU), on the first GPU of the renderer. *n_rays = w * h, *ms = wall
 * time of the call; either may be NULL. The caller's current GPU is left as it was. */
int rtgi_hip_renderer_rt_bench(rtgi_hip_renderer *r, const rtgi_frame *frame, uint64_t *n_rays, double *ms);

/* Which share of a frame part `rank` of `world` takes (host code only, no GPU needed; what rtgi_hip_renderer_render deals
 * by): with RTGI_SHARD_TILES the pixels y * w + x of every world-th tile of `tile` x `tile` pixels (0 = 64) starting with
 * tile `rank`, tiles numbered row by row, pixels of a tile row by row; with RTGI_SHARD_SAMPLES the sample indices
 * rank, rank + world, ... below sppx. *n receives their number; out, if not NULL, must have room for them
 * (call once with NULL to learn *n). */
int rtgi_hip_frame_share(uint32_t sharding, int w, int h, uint32_t tile, uint32_t sppx, int world, int rank, uint32_t *out, uint32_t *n);

const char *rtgi_hip_renderer_last_error(const rtgi_hip_renderer *r);

#ifdef __cplusplus
}
#endif
#endif /* RTGI_HIP_FRAME_H */
