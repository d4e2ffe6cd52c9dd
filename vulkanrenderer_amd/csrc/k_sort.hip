// k_sort.hip — orders every tile's bin by primitive index for the passes
// whose result depends on submission order (the overshading counter,
// main/main.cpp:4122-4123).  A segmented radix sort over the bin list, one
// segment per tile.
#include <cstring>
#include <cstdlib>

#include "vkr_kernels.h"

#include <rocprim/rocprim.hpp>

size_t vkr_sort_bins_temp_bytes(uint32_t maxEntries, uint32_t nTiles)
{
	size_t bytes = 0;
	rocprim::double_buffer<uint32_t> k(nullptr, nullptr), v(nullptr, nullptr);
	const uint32_t* off = nullptr;
	hipError_t e = rocprim::segmented_radix_sort_pairs(nullptr, bytes, k, v, maxEntries, nTiles, off, off + 1, 0, 32, 0);
	if (e != hipSuccess) return 0;
	return bytes;
}

// On return *keysIn/valsIn or *keysOut/valsOut holds the result; the function
// returns 0 if the result is in the "In" buffers, 1 if in the "Out" buffers,
// negative on error.
int vkr_sort_bins(hipStream_t s, void* temp, size_t tempBytes, uint32_t* keysIn, uint32_t* keysOut, uint32_t* valsIn,
                  uint32_t* valsOut, uint32_t maxEntries, uint32_t nTiles, const uint32_t* tileOffset)
{
	rocprim::double_buffer<uint32_t> k(keysIn, keysOut), v(valsIn, valsOut);
	hipError_t e = rocprim::segmented_radix_sort_pairs(temp, tempBytes, k, v, maxEntries, nTiles, tileOffset, tileOffset + 1,
	                                                   0, 32, s);
	if (e != hipSuccess) return -1;
	return v.current() == valsIn ? 0 : 1;
}
