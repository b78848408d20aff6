// rocPRIM's radix sort of (24-bit key, index) pairs at the tracer's batch size under different Onesweep configurations:
// workgroup size, items per thread and bits per pass. Prints microseconds per sort and checks the result.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class Config>
void run(const char *name, unsigned n, int end_bit) {
	const size_t cap = (size_t)n + 1024;
	unsigned *d;
	CHECK(hipMalloc((void **)&d, 4 * cap * 4));
	std::vector<unsigned> keys(n), idx(n);
	unsigned x = 12345;
	for (unsigned i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; keys[i] = x >> 8; idx[i] = i; }
	CHECK(hipMemcpy(d, keys.data(), n * 4, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d + 2 * cap, idx.data(), n * 4, hipMemcpyHostToDevice));
	size_t temp = 0;
	CHECK((rocprim::radix_sort_pairs<Config>(nullptr, temp, d, d + cap, d + 2 * cap, d + 3 * cap, n, 0, end_bit, 0)));
	void *t;
	CHECK(hipMalloc(&t, temp));
	CHECK((rocprim::radix_sort_pairs<Config>(t, temp, d, d + cap, d + 2 * cap, d + 3 * cap, n, 0, end_bit, 0)));
	CHECK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	CHECK(hipEventRecord(e0, 0));
	for (int rep = 0; rep < 50; ++rep)
		CHECK((rocprim::radix_sort_pairs<Config>(t, temp, d, d + cap, d + 2 * cap, d + 3 * cap, n, 0, end_bit, 0)));
	CHECK(hipEventRecord(e1, 0));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<unsigned> ko(n), vo(n);
	CHECK(hipMemcpy(ko.data(), d + cap, n * 4, hipMemcpyDeviceToHost));
	CHECK(hipMemcpy(vo.data(), d + 3 * cap, n * 4, hipMemcpyDeviceToHost));
	std::vector<char> seen(n, 0);
	size_t bad = 0;
	for (unsigned i = 0; i < n; ++i) {
		if (vo[i] >= n || seen[vo[i]] || keys[vo[i]] != ko[i]) ++bad; else seen[vo[i]] = 1;
		if (i && ko[i - 1] > ko[i]) ++bad;
	}
	printf("%-44s n %8u: %7.1f us per sort, temp %zu, bad %zu\n", name, n, ms * 20.0f, temp, bad);
	CHECK(hipFree(t));
	CHECK(hipFree(d));
}

template <unsigned BS, unsigned IPT, unsigned RB, unsigned HB = 256, unsigned HI = 12>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<HB, HI>, rocprim::kernel_config<BS, IPT>, RB,
                                                                           rocprim::block_radix_rank_algorithm::match>, 1024>;

int main() {
	for (unsigned n : {1451349u, 500000u, 200000u, 4000000u, 12000000u}) {
		run<rocprim::default_config>("default", n, 24);
		run<cfg<1024, 4, 8>>("1024 x 4", n, 24);
		run<cfg<1024, 3, 8>>("1024 x 3", n, 24);
		run<cfg<1024, 2, 8>>("1024 x 2", n, 24);
		run<cfg<1024, 5, 8>>("1024 x 5", n, 24);
		run<cfg<1024, 8, 8>>("1024 x 8", n, 24);
		run<cfg<1024, 4, 8, 1024, 4>>("1024 x 4, histogram 1024 x 4", n, 24);
		run<cfg<1024, 4, 8, 512, 16>>("1024 x 4, histogram 512 x 16", n, 24);
		run<cfg<1024, 4, 8, 256, 4>>("1024 x 4, histogram 256 x 4", n, 24);
	}
	return 0;
}
