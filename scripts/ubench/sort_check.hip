// Checks hipcub::DeviceRadixSort::SortPairs on the sizes / bit ranges the tracer uses: is values_out a
// permutation, are keys_out ordered on the sorted bits?
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
	const unsigned sizes[] = {70000, 100003, 300000, 1451349};
	const int begins[] = {0, 8};
	for (unsigned n : sizes)
		for (int begin : begins) {
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
			CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp, d, d + cap, d + 2 * cap, d + 3 * cap, (int)n, begin, 32, 0));
			void *t;
			CHECK(hipMalloc(&t, temp));
			CHECK(hipcub::DeviceRadixSort::SortPairs(t, temp, d, d + cap, d + 2 * cap, d + 3 * cap, (int)n, begin, 32, 0));
			CHECK(hipDeviceSynchronize());
			std::vector<unsigned> ko(n), vo(n);
			CHECK(hipMemcpy(ko.data(), d + cap, n * 4, hipMemcpyDeviceToHost));
			CHECK(hipMemcpy(vo.data(), d + 3 * cap, n * 4, hipMemcpyDeviceToHost));
			std::vector<char> seen(n, 0);
			size_t bad_perm = 0, bad_order = 0, bad_pair = 0;
			for (unsigned i = 0; i < n; ++i) {
				if (vo[i] >= n || seen[vo[i]]) ++bad_perm; else { seen[vo[i]] = 1; if (keys[vo[i]] != ko[i]) ++bad_pair; }
				if (i && (ko[i - 1] >> begin) > (ko[i] >> begin)) ++bad_order;
			}
			printf("n %8u begin %d temp %zu: bad_perm %zu bad_order %zu bad_pair %zu\n", n, begin, temp, bad_perm, bad_order, bad_pair);
			CHECK(hipFree(t));
			CHECK(hipFree(d));
		}
	return 0;
}
