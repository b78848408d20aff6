// Micro-benchmark: how fast can a wave gather random 64-byte records from a 128 MiB table?
//  mode 0: every lane reads its own record with four 16-byte loads (4 x 64 different lines per wave-instruction)
//  mode 1: the 4 lanes of a quad read one record together, 16 bytes each (one 64-byte segment per quad), 4 rounds
// Both modes move the same bytes per wave. Prints ns per record.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void gather(const void *table, unsigned bytes, unsigned n_records, int iters, unsigned *out) {
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(table), 0, (int)bytes, 0x00020000);
	unsigned gl = blockIdx.x * 256 + threadIdx.x;
	unsigned idx = hash(gl) % n_records;
	unsigned acc = 0;
	for (int it = 0; it < iters; ++it) {
		if (MODE == 0) {
			int off = (int)(idx << 6);
			u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
			u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
			u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 0);
			u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 48, 0, 0);
			acc += a.x ^ b.y ^ c.z ^ d.w;
			idx = hash(idx + a.x + it) % n_records;  // dependent: the next record is known only after the load
		} else {
			u32x4 r[4];
			for (int q = 0; q < 4; ++q) {
				unsigned other = __shfl(idx, (threadIdx.x & ~3u) + q, 64);
				r[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((other << 6) + ((threadIdx.x & 3u) << 4)), 0, 0);
			}
			unsigned mine = r[0].x ^ r[1].y ^ r[2].z ^ r[3].w;
			acc += mine;
			idx = hash(idx + r[threadIdx.x & 3].x + it) % n_records;
		}
	}
	out[gl] = acc;
}

int main() {
	const unsigned max_records = 1u << 21;  // 128 MiB of 64-byte records
	void *table;
	unsigned *out;
	const int blocks = 256 * 4, iters = 2000;
	CHECK(hipMalloc(&table, max_records * 64u));
	CHECK(hipMalloc(&out, blocks * 256 * 4));
	std::vector<unsigned> h(max_records * 16u);
	for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
	CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	// table sizes: fits the 32 KiB L1 of a CU / fits a 4 MiB L2 / fits the 256 MiB Infinity Cache only
	const unsigned sizes[3] = {256u, 1u << 14, 1u << 21};
	for (unsigned n_records : sizes)
		for (int mode = 0; mode < 2; ++mode) {
			float best = 1e30f;
			for (int rep = 0; rep < 3; ++rep) {
				CHECK(hipEventRecord(e0));
				if (mode == 0)
					hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, table, n_records * 64u, n_records, iters, out);
				else
					hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, table, n_records * 64u, n_records, iters, out);
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			double records = (double)blocks * 256 * iters;
			printf("table %8u KiB mode %d: %.3f ms, %.1f Grecords/s, %.2f TB/s, %.2f lane-loads/ns/CU\n", n_records / 16, mode, best,
			       records / best / 1e6, records * 64 / best / 1e9, records * 4 / best / 1e6 / 256);
		}
	return 0;
}
