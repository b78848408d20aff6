// Micro-benchmark: random dependent gathers of 16/32/64-byte records where a fraction p of the fetches goes to a
// small region that stays in every XCD's L2 and the rest to the whole 128 MiB table (Infinity Cache). Answers:
//  * do the L2-hit and the L2-miss share of a mixed gather add (one pipe) or overlap (two resources)?
//    compare the measured time with  n (p / R_hit + (1 - p) / R_miss)  and with  n max(p / R_hit, (1 - p) / R_miss);
//  * what does a record of half / a quarter of the bytes buy at each hit ratio?
// Usage: gather_mix [waves_per_simd=3] [quick]
// quick: only the 64-byte / 4-load form at hit shares 0.77 and 0, one line each for bench.py to read:
//   LANE_LOADS_PER_S <16-byte lane loads per second at 77 % L2 hits>   MISS_RECORDS_PER_S <64-byte records per second, no L2 hits>
//   LANE_LOADS_PER_S_3 <the first, with three loads per record instead of four>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// QUADS 16-byte loads per record; records are 64 bytes apart in every case (same lines touched, fewer bytes asked for)
// STRIDE_SHIFT: log2 of the distance between records (6: 64-byte records, 5: records packed at 32 bytes)
template <int QUADS, int STRIDE_SHIFT>
__global__ __launch_bounds__(256) void gather(const void *table, unsigned bytes, unsigned n_records, unsigned n_hot, unsigned p_hot_1024,
                                              int iters, unsigned *out) {
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(table), 0, (int)bytes, 0x00020000);
	unsigned gl = blockIdx.x * 256 + threadIdx.x;
	unsigned idx = hash(gl) % n_hot;
	unsigned acc = 0;
	for (int it = 0; it < iters; ++it) {
		int off = (int)(idx << STRIDE_SHIFT);
		u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0), b = {0, 0, 0, 0}, c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
		if (QUADS > 1) b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
		if (QUADS > 2) c = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 0);
		if (QUADS > 3) d = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 48, 0, 0);
		acc += a.x ^ b.y ^ c.z ^ d.w;
		const unsigned h = hash(idx + a.x + it);  // dependent: the next record is known only after the load
		const unsigned h2 = hash(h ^ 0x9e3779b9u);
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
	const unsigned p_list[] = {0, 256, 512, 788, 922, 1024};  // share of fetches that go to the hot region, in 1/1024
	const unsigned hot_bytes = 1u << 20;                        // 1 MiB: stays in each XCD's 4 MiB L2
	if (argc > 2 && std::string(argv[2]) == "quick") {
		// [0] 4 loads per record, 77 % hits; [1] 4 loads, no hits; [2] 3 loads per record (the first 48 bytes), 77 % hits
		double rate[3];
		const unsigned p_quick[3] = {788, 0, 788};
		for (int k = 0; k < 3; ++k) {
			float best = 1e30f;
			for (int rep = 0; rep < 3; ++rep) {
				CHECK(hipEventRecord(e0));
				if (k < 2) hipLaunchKernelGGL((gather<4, 6>), dim3(blocks), dim3(256), 0, 0, table, max_bytes, max_bytes >> 6, hot_bytes >> 6, p_quick[k], iters, out);
				else hipLaunchKernelGGL((gather<3, 6>), dim3(blocks), dim3(256), 0, 0, table, max_bytes, max_bytes >> 6, hot_bytes >> 6, p_quick[k], iters, out);
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			rate[k] = (double)blocks * 256 * iters / (best * 1e-3);
		}
		printf("LANE_LOADS_PER_S %.4g\nMISS_RECORDS_PER_S %.4g\nLANE_LOADS_PER_S_3 %.4g\n", 4 * rate[0], rate[1], 3 * rate[2]);
		return 0;
	}
	printf("waves per SIMD %d; hot region %u KiB, table %u MiB\n", waves_per_simd, hot_bytes >> 10, max_bytes >> 20);
	for (int variant = 0; variant < 4; ++variant) {
		// 0: 64-byte records, 4 loads; 1: 64-byte stride, 2 loads; 2: 32-byte records packed, 2 loads; 3: 64-byte stride, 1 load
		const int shift = variant == 2 ? 5 : 6;
		const unsigned n_records = max_bytes >> shift, n_hot = hot_bytes >> shift;
		double r_hit = 0, r_miss = 0;
		double meas[6];
		for (int pi = 0; pi < 6; ++pi) {
			float best = 1e30f;
			for (int rep = 0; rep < 3; ++rep) {
				CHECK(hipEventRecord(e0));
				if (variant == 0) hipLaunchKernelGGL((gather<4, 6>), dim3(blocks), dim3(256), 0, 0, table, max_bytes, n_records, n_hot, p_list[pi], iters, out);
				if (variant == 1) hipLaunchKernelGGL((gather<2, 6>), dim3(blocks), dim3(256), 0, 0, table, max_bytes, n_records, n_hot, p_list[pi], iters, out);
				if (variant == 2) hipLaunchKernelGGL((gather<2, 5>), dim3(blocks), dim3(256), 0, 0, table, max_bytes, n_records, n_hot, p_list[pi], iters, out);
				if (variant == 3) hipLaunchKernelGGL((gather<1, 6>), dim3(blocks), dim3(256), 0, 0, table, max_bytes, n_records, n_hot, p_list[pi], iters, out);
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			meas[pi] = best;
		}
		const double records = (double)blocks * 256 * iters;
		r_miss = records / meas[0];
		r_hit = records / meas[5];
		static const char *names[4] = {"64-B records, 4 loads", "64-B stride, 2 loads (first 32 B)", "32-B records packed, 2 loads", "64-B stride, 1 load (first 16 B)"};
		printf("%s: all-miss %.1f Grec/s, all-hit %.1f Grec/s\n", names[variant], r_miss / 1e6, r_hit / 1e6);
		for (int pi = 0; pi < 6; ++pi) {
			const double p = p_list[pi] / 1024.0;
			const double t_add = records * (p / r_hit + (1 - p) / r_miss), t_hit = records * p / r_hit, t_miss = records * (1 - p) / r_miss;
			printf("  p_hot %.3f: %.3f ms = %.1f Grec/s | additive model %.3f ms, overlap model %.3f ms\n", p, meas[pi], records / meas[pi] / 1e6,
			       t_add, t_hit > t_miss ? t_hit : t_miss);
		}
	}
	return 0;
}
