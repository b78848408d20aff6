// Micro-benchmark: what does a one-step-ahead cache prefetch buy a chain of dependent random record fetches?
// Every active lane walks a chain of 64-byte records (4 loads each; 77 % of them in an L2-resident 1 MiB region,
// the rest anywhere in a 128 MiB table). The address of a record depends on the data of the one before (the table
// holds zeros, the loaded words are added to the index, so the hardware sees a true dependency while the program
// knows the chain in advance). Mode 1 asks, together with record k, for one word of the line of record k+1 and
// never waits for it: the demand fetch of k+1 then finds its line on the way or in the vector cache.
// Usage: gather_prefetch [waves_per_simd=4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// n_records and n_hot are powers of two: the choice of the next record must be cheap, it sits on the chain
__device__ inline unsigned pick(unsigned seed, unsigned n_records, unsigned n_hot, unsigned p_hot_1024) {
	const unsigned h = hash(seed), h2 = (h * 0x9e3779b1u) >> 8;
	return h2 & (((h & 1023u) < p_hot_1024 ? n_hot : n_records) - 1u);
}

template <int MODE>
__global__ __launch_bounds__(256) void chain(const void *table, unsigned bytes, unsigned n_records, unsigned n_hot, unsigned p_hot_1024,
                                             int iters, int active, unsigned *out) {
	const unsigned gl = blockIdx.x * 256 + threadIdx.x;
	unsigned acc = 0, z = 0, ahead = 0;
	if ((int)(threadIdx.x & 63) < active) {
		unsigned idx = pick(gl * 4099u, n_records, n_hot, p_hot_1024);
		// The loads and the waits are written out: the compiler's scheduler would otherwise move the look-ahead load
		// behind the wait for the record. Loads return in order, so "at most one outstanding" after the five loads
		// of a step means the record is here and only the look-ahead word may still be under way.
		typedef unsigned int rsrc_words __attribute__((ext_vector_type(4)));
		const unsigned long long ta = (unsigned long long)table;
		const rsrc_words rw = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ta), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ta >> 32) & 0xffff),
		                       (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
		for (int it = 0; it < iters; ++it) {
			const unsigned next = pick(gl * 4099u + it + 1, n_records, n_hot, p_hot_1024);
			const unsigned off = (idx + z) << 6, off_next = next << 6;
			unsigned a, b, c, d;
			if (MODE == 1) {
				asm volatile("buffer_load_dword %0, %5, %7, 0 offen\n\tbuffer_load_dword %1, %5, %7, 0 offen offset:20\n\t"
				             "buffer_load_dword %2, %5, %7, 0 offen offset:40\n\tbuffer_load_dword %3, %5, %7, 0 offen offset:60\n\t"
				             "buffer_load_dword %4, %6, %7, 0 offen\n\ts_waitcnt vmcnt(1)"
				             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "+v"(ahead) : "v"(off), "v"(off_next), "s"(rw) : "memory");
			} else {
				asm volatile("buffer_load_dword %0, %4, %5, 0 offen\n\tbuffer_load_dword %1, %4, %5, 0 offen offset:20\n\t"
				             "buffer_load_dword %2, %4, %5, 0 offen offset:40\n\tbuffer_load_dword %3, %4, %5, 0 offen offset:60\n\t"
				             "s_waitcnt vmcnt(0)"
				             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(off), "s"(rw) : "memory");
			}
			z = a + b + c + d;  // zero, known only when the record is here
			acc += z;
			idx = next;
		}
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(ahead)::"memory");
	}
	out[gl] = acc + ahead;
}

int main(int argc, char **argv) {
	const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
	const unsigned max_bytes = 128u << 20;
	void *table;
	unsigned *out;
	const int blocks = 256 * waves_per_simd, iters = 1000;
	CHECK(hipMalloc(&table, max_bytes));
	CHECK(hipMalloc(&out, blocks * 256 * 4));
	CHECK(hipMemset(table, 0, max_bytes));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	printf("waves per SIMD %d\n", waves_per_simd);
	// hot region of 1 MiB (L2) at three shares, then a hot "region" of one record (always in the vector cache)
	for (int cfg = 0; cfg < 4; ++cfg)
		for (int active : {64, 32, 16, 8, 1})
			for (int mode = 0; mode < 2; ++mode) {
				const unsigned p_hot = cfg == 0 ? 788u : cfg == 1 ? 0u : 1024u, n_hot = cfg == 3 ? 1u : (1u << 20) >> 6;
				float best = 1e30f;
				for (int rep = 0; rep < 3; ++rep) {
					CHECK(hipEventRecord(e0));
					if (mode == 0) hipLaunchKernelGGL((chain<0>), dim3(blocks), dim3(256), 0, 0, table, max_bytes, max_bytes >> 6, n_hot, p_hot, iters, active, out);
					else hipLaunchKernelGGL((chain<1>), dim3(blocks), dim3(256), 0, 0, table, max_bytes, max_bytes >> 6, n_hot, p_hot, iters, active, out);
					CHECK(hipEventRecord(e1));
					CHECK(hipEventSynchronize(e1));
					float ms;
					CHECK(hipEventElapsedTime(&ms, e0, e1));
					if (ms < best) best = ms;
				}
				printf("%s, hot share %.2f, %2d active lanes, %s: %.0f ns per step\n", cfg == 3 ? "one hot record" : "1 MiB hot", p_hot / 1024.0, active, mode ? "prefetch one ahead" : "no prefetch      ", best * 1e6 / iters);
			}
	return 0;
}
