// rtgi_hip_algorithms.hpp -- wavefront algorithms for the renderer that run whole frames on the GPUs.
//
// For use inside the rtgi tree only (it needs libgi/algorithm.h, libgi/context.h and the global render context `rc`);
// citations are file:line of "Assignment 3/rtgi-2021-a03-ref/" unless they start with "Assignment". The renderer's
// own algorithms trace one ray per virtual call (gi/primary-hit.cpp:16-57, recursive_algorithm); these are their
// wavefront_algorithm counterparts (libgi/algorithm.h:72-76, modelled on primary_hit_display_wf of
// "Assignment 2/rtgi-2021-a02/gi/primary-hit.cpp:61-74": set up camera rays, trace the batch, shade the batch), with
// every stage on the GPU: they need `raytracer hip ...` (a hip_bvh_frame_tracer) as scene.rt and write
// rc->framebuffer.color exactly as framebuffer::add would have (running mean in rgb, sample count in w).
//
//   algo primary-wf   -> new rtgi_hip::primary_hit_display_wf     (primary_hit_display, gi/primary-hit.cpp:16-30)
//   algo local-wf     -> new rtgi_hip::local_illumination_wf      (local_illumination,  gi/primary-hit.cpp:32-57)
//   algo path-wf      -> new rtgi_hip::diffuse_path_wf            (no counterpart: the reference has no path tracer)
#pragma once

#ifndef RTGI_HIP_USE_LIBGI
#error "rtgi_hip_algorithms.hpp is for builds inside the rtgi tree: define RTGI_HIP_USE_LIBGI"
#endif

#include "rtgi_hip_tracer.hpp"

#include "libgi/algorithm.h"
#include "libgi/context.h"
#include "libgi/global-context.h"

#include <random>

namespace rtgi_hip {

class frame_algorithm : public ::wavefront_algorithm {
protected:
	uint32_t algo;
	uint32_t bounces = 3, sharding = RTGI_SHARD_TILES, tile = 0;
	explicit frame_algorithm(uint32_t algo) : algo(algo) {}

public:
	// `bounces N`, `sharding tiles|samples`, `tile N` (forwarded here by the REPL, driver/interaction.cpp:399)
	bool interprete(const std::string &command, std::istringstream &in) override {
		if (command == "bounces") {
			in >> bounces;
			return true;
		}
		if (command == "sharding") {
			std::string how;
			in >> how;
			sharding = how == "samples" ? RTGI_SHARD_SAMPLES : RTGI_SHARD_TILES;
			return true;
		}
		if (command == "tile") {
			in >> tile;
			return true;
		}
		return false;
	}
	// (wavefront_algorithm::prepare_frame asserts a batch tracer, libgi/algorithm.cpp:77-79; this one needs the frame tracer)
	void prepare_frame() override {
		if (!dynamic_cast<hip_bvh_frame_tracer *>(rc->scene.rt)) throw std::runtime_error("this algorithm needs `raytracer hip` (hip_bvh_frame_tracer)");
	}
	void compute_samples() override {
		auto *rt = dynamic_cast<hip_bvh_frame_tracer *>(rc->scene.rt);
		if (!rt) throw std::runtime_error("this algorithm needs `raytracer hip` (hip_bvh_frame_tracer)");
		const ::camera &cam = rc->scene.camera;
		rtgi_frame f{};
		f.cam = rtgi_camera{{cam.dir.x, cam.dir.y, cam.dir.z}, {cam.pos.x, cam.pos.y, cam.pos.z}, {cam.up.x, cam.up.y, cam.up.z}, cam.fovy, cam.w, cam.h};
		f.sppx = rc->sppx;
		f.algo = algo;
		f.bounces = bounces;
		if (algo != RTGI_ALGO_PRIMARY) {
			auto *pl = rc->scene.lights.empty() ? nullptr : dynamic_cast<pointlight *>(rc->scene.lights[0]);  // gi/primary-hit.cpp:41-42
			if (!pl) throw std::runtime_error("this algorithm needs a point light");
			f.light_pos[0] = pl->pos.x, f.light_pos[1] = pl->pos.y, f.light_pos[2] = pl->pos.z;
			f.light_col[0] = pl->col.x, f.light_col[1] = pl->col.y, f.light_col[2] = pl->col.z;
		}
		f.seed = std::random_device{}();  // a new frame, new samples (the reference's rng is seeded per run, libgi/random.h)
		f.jitter = 1;
		f.sharding = sharding;
		f.tile = tile;
		// `material ...` commands may have come after `commit`: the table is small, hand it over every frame
		rt->refresh_materials();
		if (rc->framebuffer.color.w != (unsigned)cam.w || rc->framebuffer.color.h != (unsigned)cam.h)
			throw std::runtime_error("framebuffer and camera disagree about the resolution");
		rt->render(f, &rc->framebuffer.color.data[0].x);  // buffer<vec4>: row y at y*w (libgi/framebuffer.h:13-50)
	}
};

struct primary_hit_display_wf : public frame_algorithm {
	primary_hit_display_wf() : frame_algorithm(RTGI_ALGO_PRIMARY) {}
};
struct local_illumination_wf : public frame_algorithm {
	local_illumination_wf() : frame_algorithm(RTGI_ALGO_LOCAL) {}
};
struct diffuse_path_wf : public frame_algorithm {
	diffuse_path_wf() : frame_algorithm(RTGI_ALGO_PATH) {}
};

}  // namespace rtgi_hip
