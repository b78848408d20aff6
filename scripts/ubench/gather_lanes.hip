// Micro-benchmark: does a divergent 16-byte gather instruction cost by the wave-instruction or by the active lane?
// Every wave runs the same dependent chain of random 64-byte record fetches (4 loads each, L2-resident 1 MiB hot
// region with probability 0.77, else anywhere in a 128 MiB table), but only the first `active` lanes of each wave
// take part. Prints time per wave-iteration and per active lane.
// Usage: gather_lanes [waves_per_simd=3]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int QUADS>
__global__ __launch_bounds__(256) void gather(const void *table, unsigned bytes, unsigned n_records, unsigned n_hot, unsigned p_hot_1024,
                                              int iters, int active, unsigned *out) {
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(table), 0, (int)bytes, 0x00020000);
	unsigned gl = blockIdx.x * 256 + threadIdx.x;
	unsigned idx = hash(gl) % n_hot;
	unsigned acc = 0;
	if ((int)(threadIdx.x & 63) < active)
		for (int it = 0; it < iters; ++it) {
			int off = (int)(idx << 6);
			u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0), b = {0, 0, 0, 0}, c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
			if (QUADS > 1) b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
			if (QUADS > 2) c = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 0);
			if (QUADS > 3) d = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 48, 0, 0);
			acc += a.x ^ b.y ^ c.z ^ d.w;
			const unsigned h = hash(idx + a.x + it), h2 = hash(h ^ 0x9e3779b9u);
			idx = ((h & 1023u) < p_hot_1024) ? h2 % n_hot : h2 % n_records;
		}
	out[gl] = acc;
}

int main(int argc, char **argv) {
	const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 3;
	const unsigned max_bytes = 128u << 20;
	void *table;
	unsigned *out;
	const int blocks = 256 * waves_per_simd, iters = 1000;
	CHECK(hipMalloc(&table, max_bytes));
	CHECK(hipMalloc(&out, blocks * 256 * 4));
	std::vector<unsigned> h(max_bytes / 4);
	for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
	CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	printf("waves per SIMD %d, 77%% of the fetches to a 1 MiB hot region\n", waves_per_simd);
	for (int quads = 4; quads >= 2; quads -= 2)
		for (int active : {64, 48, 32, 16, 8, 1}) {
			float best = 1e30f;
			for (int rep = 0; rep < 3; ++rep) {
				CHECK(hipEventRecord(e0));
				if (quads == 4) hipLaunchKernelGGL((gather<4>), dim3(blocks), dim3(256), 0, 0, table, max_bytes, max_bytes >> 6, (1u << 20) >> 6, 788u, iters, active, out);
				else hipLaunchKernelGGL((gather<2>), dim3(blocks), dim3(256), 0, 0, table, max_bytes, max_bytes >> 6, (1u << 20) >> 6, 788u, iters, active, out);
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			printf("%d loads per record, %2d active lanes: %.3f ms, %.0f ns per wave-iteration, %.1f G records/s\n", quads, active, best,
			       best * 1e6 / iters, (double)blocks * 4 * active * iters / best / 1e6);
		}
	return 0;
}
