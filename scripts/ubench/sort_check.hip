// Checks hipcub::DeviceRadixSort::SortPairs on the sizes / bit ranges the tracer uses: is values_out a
// permutation, are keys_out ordered on the sorted bits?
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
	const unsigned sizes[] = {70000, 100003, 300000, 1451349, 2073600};
	const int ranges[][2] = {{0, 32}, {0, 24}, {0, 16}, {8, 32}};
	for (unsigned n : sizes)
		for (auto &range : ranges) {
			const int begin = range[0], end = range[1];
			const size_t cap = (size_t)n + n / 8 + 1024;
			unsigned *d;
			CHECK(hipMalloc((void **)&d, 4 * cap * 4));
			CHECK(hipMemset(d, 0xff, 4 * cap * 4));
			std::vector<unsigned> keys(n), idx(n);
			unsigned x = 12345;
			for (unsigned i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; keys[i] = x; idx[i] = i; }
			CHECK(hipMemcpy(d, keys.data(), n * 4, hipMemcpyHostToDevice));
			CHECK(hipMemcpy(d + 2 * cap, idx.data(), n * 4, hipMemcpyHostToDevice));
			size_t temp = 0;
			CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp, d, d + cap, d + 2 * cap, d + 3 * cap, (int)n, begin, end, 0));
			void *t;
			CHECK(hipMalloc(&t, temp));
			CHECK(hipcub::DeviceRadixSort::SortPairs(t, temp, d, d + cap, d + 2 * cap, d + 3 * cap, (int)n, begin, end, 0));
			CHECK(hipDeviceSynchronize());
			hipEvent_t e0, e1;
			CHECK(hipEventCreate(&e0));
			CHECK(hipEventCreate(&e1));
			CHECK(hipEventRecord(e0, 0));
			for (int rep = 0; rep < 20; ++rep)
				CHECK(hipcub::DeviceRadixSort::SortPairs(t, temp, d, d + cap, d + 2 * cap, d + 3 * cap, (int)n, begin, end, 0));
			CHECK(hipEventRecord(e1, 0));
			CHECK(hipEventSynchronize(e1));
			float ms = 0;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			std::vector<unsigned> ko(n), vo(n);
			CHECK(hipMemcpy(ko.data(), d + cap, n * 4, hipMemcpyDeviceToHost));
			CHECK(hipMemcpy(vo.data(), d + 3 * cap, n * 4, hipMemcpyDeviceToHost));
			std::vector<char> seen(n, 0);
			size_t bad_perm = 0, bad_order = 0, bad_pair = 0;
			for (unsigned i = 0; i < n; ++i) {
				if (vo[i] >= n || seen[vo[i]]) ++bad_perm; else { seen[vo[i]] = 1; if (keys[vo[i]] != ko[i]) ++bad_pair; }
				const unsigned mask = end == 32 ? 0xffffffffu : ((1u << end) - 1u);
				if (i && ((ko[i - 1] & mask) >> begin) > ((ko[i] & mask) >> begin)) ++bad_order;
			}
			printf("n %8u bits [%d,%d) temp %zu: bad_perm %zu bad_order %zu bad_pair %zu, %.1f us per sort\n", n, begin, end, temp, bad_perm, bad_order, bad_pair, ms * 50.0f);
			CHECK(hipFree(t));
			CHECK(hipFree(d));
		}
	return 0;
}
