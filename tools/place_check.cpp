// Host check of the workgroup placement maps of vulkanrenderer_amd/csrc/vkr_place.h: every map must be a bijection of
// the grid onto itself, for every grid size (a workgroup placed twice leaves pixels undrawn).  Prints "bijective".
#define VKR_PLACE_FN inline
#include "../vulkanrenderer_amd/csrc/vkr_place.h"
#include <cstdio>
#include <vector>

template <uint32_t CX, uint32_t CY> static int check2d()
{
	for (uint32_t gx = 1; gx <= 90; gx++)
		for (uint32_t gy = 1; gy <= 50; gy++) {
			std::vector<char> seen(gx * gy, 0);
			for (uint32_t y = 0; y < gy; y++)
				for (uint32_t x = 0; x < gx; x++) {
					uint32_t bx = ~0u, by = ~0u;
					vkr_xcd_place_2d<CX, CY>(x, y, gx, gy, bx, by);
					if (bx >= gx || by >= gy || seen[by * gx + bx]++) { printf("2d %ux%u blocks: grid %ux%u is not mapped one to one\n", CX, CY, gx, gy); return 1; }
				}
		}
	return 0;
}

template <uint32_t CH> static int check1d()
{
	for (uint32_t n = 1; n <= 3000; n++) {
		std::vector<char> seen(n, 0);
		for (uint32_t i = 0; i < n; i++) {
			const uint32_t p = vkr_xcd_place<CH>(i, n);
			if (p >= n || seen[p]++) { printf("1d runs of %u: %u workgroups are not mapped one to one\n", CH, n); return 1; }
		}
	}
	return 0;
}

int main()
{
	// the block shapes the kernels use (vkr_screen_place: 8x8 and 4x4; k_raster_alpha: runs of 4) and a few others
	const int bad = check2d<8, 8>() | check2d<4, 4>() | check2d<2, 2>() | check2d<3, 5>() | check2d<1, 1>() | check1d<4>() | check1d<1>() | check1d<7>();
	// workgroups of one XCD (equal dispatch index mod 8) take whole blocks: the first 64 places of XCD 0 on a 240 x 135 grid
	uint32_t bx, by, x0 = ~0u, y0 = ~0u, x1 = 0, y1 = 0;
	for (uint32_t k = 0; k < 64; k++) {
		const uint32_t n = k * 8;
		vkr_xcd_place_2d<8, 8>(n % 240, n / 240, 240, 135, bx, by);
		x0 = bx < x0 ? bx : x0; y0 = by < y0 ? by : y0; x1 = bx > x1 ? bx : x1; y1 = by > y1 ? by : y1;
	}
	const int spread = !(x1 - x0 == 7 && y1 - y0 == 7);
	if (spread) printf("the first 64 workgroups of an XCD do not form an 8x8 block\n");
	if (!bad && !spread) printf("bijective\n");
	return bad | spread;
}
