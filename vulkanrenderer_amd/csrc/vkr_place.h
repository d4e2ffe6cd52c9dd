// vkr_place.h - which part of the screen a workgroup takes, given its dispatch index.
//
// Workgroups are dealt to the eight XCDs round-robin by dispatch index, and every XCD has an L2 of its own: screen
// neighbours that share texel lines, triangle records or bin entries would each pull them into a different L2.
// The kernels that walk the screen therefore take their place from this map instead of from the dispatch index:
// runs of CH consecutive places go to the workgroups of one XCD (those with equal n mod 8), and the runs are dealt to
// the XCDs in turn.  Measured on k_resolve_gbuf at 3840x2160 (L2 misses per launch, time): dispatch order 337-379 MB,
// 204 us; blocks of 4x4 workgroups 256 MB, 8x8 235 MB, 16x16 218 MB, all within 2 % of 201 us; one contiguous band
// of the screen per XCD 203 MB but 224 us, and the rasterisers 70 % slower - a band hands all of the foliage, or all
// of the near field, to one XCD.  Hence short runs: most of the sharing, the balance of the round-robin.
// Bijective for every grid: the places behind the last whole group of 8 * CH keep their dispatch index.  A pure
// placement choice: which XCD runs a workgroup never affects the result.
// Plain C++ (no HIP types) so that the host can check the maps (tests/test_abi.py builds tools/place_check.cpp).
#pragma once
#include <stdint.h>
#ifndef VKR_PLACE_FN
#define VKR_PLACE_FN __host__ __device__ __forceinline__
#endif
#ifndef VKR_XCD_SWIZZLE
#define VKR_XCD_SWIZZLE 1
#endif

template <uint32_t CH>
VKR_PLACE_FN uint32_t vkr_xcd_place(uint32_t n, uint32_t nwg)
{
#if VKR_XCD_SWIZZLE
	const uint32_t whole = nwg - nwg % (8u * CH);
	if (n >= whole) return n;
	const uint32_t xcd = n & 7u, k = n >> 3; // k-th workgroup of its XCD
	return ((k / CH) * 8u + xcd) * CH + k % CH;
#else
	return n;
#endif
}

// The same for a two-dimensional grid of gx x gy workgroups: a run is a block of CX x CY neighbouring workgroups
// (sharing goes both ways on the screen); the cells right of and below the last whole block follow in row-major order.
template <uint32_t CX, uint32_t CY>
VKR_PLACE_FN void vkr_xcd_place_2d(uint32_t ix, uint32_t iy, uint32_t gx, uint32_t gy, uint32_t& bx, uint32_t& by)
{
	const uint32_t p = vkr_xcd_place<CX * CY>(iy * gx + ix, gx * gy);
	const uint32_t ncx = gx / CX, GX = ncx * CX, RY = (gy / CY) * CY;
	if (p < GX * RY) {
		const uint32_t c = p / (CX * CY), j = p % (CX * CY);
		bx = (c % ncx) * CX + j % CX; by = (c / ncx) * CY + j / CX;
		return;
	}
	uint32_t q = p - GX * RY;
	const uint32_t rx = gx - GX;
	if (q < RY * rx) { by = q / rx; bx = GX + q % rx; return; }
	q -= RY * rx;
	by = RY + q / gx; bx = q % gx;
}
