// k_raster.hip — tile rasterisers.
//
// k_raster<MODE>: one workgroup owns one 32x32 screen tile whose visibility
// buffer lives in LDS.  The tile's bin (indices of the sub-triangles that
// touch it) is walked 256 entries at a time, dealt round-robin to the four
// wavefronts: small sub-triangles are scan-converted one per lane, large ones
// by a whole wavefront (one lane per pixel of an 8x8 block).  Depth test +
// write is a single 64-bit LDS atomicMin on
// (depthBits << 32) | (0xFFFFFFFE - prim): smaller depth wins and, at equal
// depth, the later primitive wins, which is VK_COMPARE_OP_LESS_OR_EQUAL
// applied in submission order (main/main.cpp:2321-2327).  The finished tile
// is written to HBM once, coalesced.
// k_raster_alpha: the alpha-tested sub-triangles, pixel-owner (see below).
// k_overshade: the order-dependent overshading counter, pixel-owner.
//
// Replaces the fixed-function rasteriser + depth/stencil unit for the draws of
// main/main.cpp:3873-3934 (G-buffer pass), :3968-3982 (shadow pass),
// :4014-4075 (forward pass), :4111-4126 (overdraw counter) and :4164-4190
// (debug views).  Compiled with -ffp-contract=off.
#include "vkr_kernels.h"

#define ALPHA_DEEP 96   // alpha-tested + many-tile bin entries above which a tile's quadrants are split over ALPHA_SPLIT workgroups
#define ALPHA_SPLIT 8
#define ALPHA_BUSY 4096u   // quadrants with alpha-tested entries above which deep tiles are not split
#define ALPHA_HELPERS 3584u // helper workgroups launched behind the per-quadrant ones: room for 128 deep tiles
#define BIG_AREA 48 // bbox pixels inside the tile above which a wave takes the sub-triangle

// Edge functions are integer-valued and, with the guard band of
// vkr_guard_band(), stay below 2^53 in magnitude: they are evaluated exactly
// in double (two DP instructions per edge instead of a 64-bit integer
// multiply sequence), and stepping adds exact constants.
struct EdgeSetup {
	double e0, e1, e2;    // biased edge values at the start pixel
	double sx0, sx1, sx2; // per +1 pixel in x
	double sy0, sy1, sy2; // per +1 pixel in y
};

// Edge i runs from vertex i to vertex i+1; E = dx*(PY - Y) - dy*(PX - X) >= 0
// inside (orientation positive); a pixel exactly on an edge is covered only
// if that edge is a top edge (dy == 0, dx > 0) or a left edge (dy < 0).
__device__ __forceinline__ void edge_setup(const SubTri& s, int px, int py, EdgeSetup& E)
{
	const int PX = px * 256 + 128, PY = py * 256 + 128;
	const int dx0 = s.X1 - s.X0, dy0 = s.Y1 - s.Y0;
	const int dx1 = s.X2 - s.X1, dy1 = s.Y2 - s.Y1;
	const int dx2 = s.X0 - s.X2, dy2 = s.Y0 - s.Y2;
	const bool tl0 = (dy0 == 0 && dx0 > 0) || (dy0 < 0);
	const bool tl1 = (dy1 == 0 && dx1 > 0) || (dy1 < 0);
	const bool tl2 = (dy2 == 0 && dx2 > 0) || (dy2 < 0);
	// each product is an exact integer below 2^52; the fused subtraction is exact too
	// (the bias is added, not subtracted: x + 0.0 is +0.0 where x - 0.0 would keep a -0.0, and edges_inside reads sign bits)
	E.e0 = fma((double)dx0, (double)(PY - s.Y0), -((double)dy0 * (double)(PX - s.X0))) + (tl0 ? 0.0 : -1.0);
	E.e1 = fma((double)dx1, (double)(PY - s.Y1), -((double)dy1 * (double)(PX - s.X1))) + (tl1 ? 0.0 : -1.0);
	E.e2 = fma((double)dx2, (double)(PY - s.Y2), -((double)dy2 * (double)(PX - s.X2))) + (tl2 ? 0.0 : -1.0);
	E.sx0 = -(double)dy0 * 256.0; E.sx1 = -(double)dy1 * 256.0; E.sx2 = -(double)dy2 * 256.0;
	E.sy0 = (double)dx0 * 256.0; E.sy1 = (double)dx1 * 256.0; E.sy2 = (double)dx2 * 256.0;
}

// All three edge values >= 0.  They are integers held in doubles and never -0.0 (edge_setup adds +0.0 to the start
// values, and neither a step added to such a value nor an exact cancellation gives -0.0 in round-to-nearest), so
// the test is the sign bit of the three high words: one OR and one integer compare instead of the five double-precision
// instructions (three of them NaN canonicalisations) of an fmin chain.
__device__ __forceinline__ bool edges_inside(double e0, double e1, double e2)
{
	return (__double2hiint(e0) | __double2hiint(e1) | __double2hiint(e2)) >= 0;
}

__device__ __forceinline__ void tri_bbox(const SubTri& s, int& px0, int& py0, int& px1, int& py1)
{
	int32_t xmin = min(s.X0, min(s.X1, s.X2)), xmax = max(s.X0, max(s.X1, s.X2));
	int32_t ymin = min(s.Y0, min(s.Y1, s.Y2)), ymax = max(s.Y0, max(s.Y1, s.Y2));
	px0 = (xmin - 128 + 255) >> 8; px1 = (xmax - 128) >> 8;
	py0 = (ymin - 128 + 255) >> 8; py1 = (ymax - 128) >> 8;
}

template <int MODE>
__device__ __forceinline__ void fragment(unsigned long long* tb64, uint32_t* tb32, int li, const SubTri& s, int px, int py)
{
	double sx = (double)px + 0.5, sy = (double)py + 0.5;
	double z64 = fma(s.Az, sx, fma(s.Bz, sy, s.Cz));
	float z = (float)z64;
	if (!(z >= 0.0f && z <= 1.0f)) return; // clip volume 0 <= z <= w, depth clamp disabled (main/main.cpp:2300)
	if (MODE == RASTER_VIS) {
		unsigned long long key = ((unsigned long long)f2u(z) << 32) | (unsigned long long)(0xFFFFFFFEu - s.prim);
		atomicMin(&tb64[li], key);
	} else if (MODE == RASTER_SHADOW) {
		float zb = (float)(z64 + s.bias);
		if (zb > 1.0f) zb = 1.0f;
		if (zb < 0.0f) zb = 0.0f;
		atomicMin(&tb32[li], f2u(zb));
	} else {
		atomicAdd(&tb32[li], 1u);
	}
}

// Broadcasts lane `src`'s sub-triangle and its tile-clipped bounding box to the wave.  `src` is wave-uniform, so
// the values travel through v_readlane into scalar registers (no LDS permute).
__device__ __forceinline__ int rl_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double rl_d(double v, int src)
{
	const unsigned long long b = (unsigned long long)__double_as_longlong(v);
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src);
	return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ void bcast_subtri(const SubTri& s, int bx0, int by0, int bx1, int by1, int src, SubTri& w, int& wx0,
                                             int& wy0, int& wx1, int& wy1)
{
	w.X0 = rl_i(s.X0, src); w.Y0 = rl_i(s.Y0, src);
	w.X1 = rl_i(s.X1, src); w.Y1 = rl_i(s.Y1, src);
	w.X2 = rl_i(s.X2, src); w.Y2 = rl_i(s.Y2, src);
	w.prim = (uint32_t)rl_i((int)s.prim, src); w.flags = (uint32_t)rl_i((int)s.flags, src);
	w.Az = rl_d(s.Az, src); w.Bz = rl_d(s.Bz, src); w.Cz = rl_d(s.Cz, src);
	w.bias = rl_d(s.bias, src);
	wx0 = rl_i(bx0, src); wy0 = rl_i(by0, src);
	wx1 = rl_i(bx1, src); wy1 = rl_i(by1, src);
}

// Every sub-triangle of the tile's bin that needs no alpha test.  In
// RASTER_VIS mode alpha-tested sub-triangles are left to k_raster_alpha and
// such tiles are flagged in alphaTiles.
#ifdef VKR_WG_TIMES
// measurement build only: per-workgroup phase clocks (100 MHz) and bin sizes of the last k_raster launch of each mode
__device__ unsigned long long g_wgTimes[3][8 * 32768];
extern "C" __attribute__((visibility("default"))) int vkr_debug_wg_times(int mode, void* dst, size_t bytes)
{
	return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wgTimes), bytes, (size_t)mode * sizeof(g_wgTimes[0]), hipMemcpyDeviceToHost);
}
#define WG_T(k) if (threadIdx.x == 0) wgT[k] = wall_clock64()
#else
#define WG_T(k)
#endif

#ifdef VKR_RASTER_STATS
// measurement build only (make EXTRA=-DVKR_RASTER_STATS): what the opaque camera pass walks.  [0] sub-triangles taken one
// per lane group, [1] pixels of their boxes, [2] sub-triangles taken by a whole wavefront, [3] pixels of their boxes,
// [4] 8x8 block steps spent on them, [5] steps a row-major walk of the same boxes would take
__device__ unsigned long long g_rstats[8];
extern "C" __attribute__((visibility("default"))) int vkr_debug_raster_stats(void* dst, int clear)
{
	if (clear) { static unsigned long long z[8]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rstats), z, sizeof z); }
	return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rstats), sizeof(g_rstats), 0, hipMemcpyDeviceToHost);
}
#define RS_ADD(k, v) atomicAdd(&g_rstats[k], (unsigned long long)(v))
#else
#define RS_ADD(k, v)
#endif

template <int MODE>
__global__ __launch_bounds__(256) void k_raster(RasterArgs a)
{
#ifdef VKR_WG_TIMES
	unsigned long long wgT[6];
	WG_T(0);
#endif
	__shared__ unsigned long long tb64[MODE == RASTER_VIS ? VKR_TILE_PIX : 1];
	__shared__ uint32_t tb32[MODE == RASTER_VIS ? 1 : VKR_TILE_PIX];
	__shared__ int sAnyAlpha;
	// longest bins first (the order comes from the tile scan): the light's depth pass, whose bins are very uneven, ends
	// a fifth sooner (101 -> 82 us for a 2048^2 map), the camera's pass by 1-2 %.  (Runs of neighbouring tiles per XCD
	// bought no time here and cost L2 misses, 27 -> 31 MB.)
	const uint32_t tile = a.tileOrder ? a.tileOrder[blockIdx.x] : blockIdx.x;
	const uint32_t tx = tile % a.view.tilesX, ty = tile / a.view.tilesX;
	const int ox = (int)(a.view.x0 + tx * VKR_TILE), oy = vkr_view_gy(a.view, ty * VKR_TILE);
	const int ex = min(ox + VKR_TILE, (int)a.view.x1) - 1, ey = oy + (int)min((uint32_t)VKR_TILE, a.view.rows - ty * VKR_TILE) - 1; // inclusive
	const uint32_t tid = threadIdx.x, lane = tid & 63;

	for (uint32_t i = tid; i < VKR_TILE_PIX; i += 256) {
		if (MODE == RASTER_VIS) tb64[i] = 0x3F800000FFFFFFFFull;
		else if (MODE == RASTER_SHADOW) tb32[i] = 0x3F800000u;
		else tb32[i] = 0u;
	}
	if (tid == 0) sAnyAlpha = 0;
	__syncthreads();
	WG_T(1);

	uint32_t beg = a.tileOffset[tile], end = a.tileOffset[tile + 1];
	if (end > a.binCap) end = a.binCap;
	if (beg > end) beg = end;
	// the bin: nO records without alpha test, nA alpha-tested ones (left to k_raster_alpha), then the records
	// that touch many tiles, of either kind
	const uint32_t nO = min(a.tileCount[tile], end - beg), nA = min(a.tileCountA[tile], end - beg - nO);
	const uint32_t n = (end - beg) - nA;
	if (MODE == RASTER_VIS && tid == 0 && nA) sAnyAlpha = 1;

	// 1, 2 or 4 lanes share one bin entry, so that a short bin (tens of entries per tile is the rule) still
	// fills the workgroup: a small sub-triangle covers a handful of pixels per row, and with one entry per
	// lane nine lanes in ten would idle.  The lanes of a group take every group-th column of the bounding box.
	const uint32_t logG = n <= 64u ? 2u : n <= 128u ? 1u : 0u, G = 1u << logG;
	const uint32_t sub = lane & (G - 1u);
#ifdef VKR_WG_TIMES
	if (n == 0xffffffffu) return; // never: makes the header loads complete before the clock is read
#endif
	WG_T(2);
	for (uint32_t base = 0; base < n; base += 256u >> logG) {
		// entries are dealt to the four wavefronts round-robin, so a short bin still spreads its
		// large (wave-cooperative) sub-triangles over all of them
		const uint32_t i = base + (lane >> logG) * 4u + (tid >> 6);
		bool has = i < n;
		SubTri s;
		int bx0 = 0, by0 = 0, bx1 = -1, by1 = -1;
		if (has) {
			s = a.recs[a.binList[beg + (i < nO ? i : i + nA)]];
			if (MODE == RASTER_VIS && (s.flags & VKR_ST_ALPHA)) { sAnyAlpha = 1; has = false; }
		}
		if (has) {
			tri_bbox(s, bx0, by0, bx1, by1);
			bx0 = max(bx0, ox); by0 = max(by0, oy); bx1 = min(bx1, ex); by1 = min(by1, ey);
		} else {
			s.X0 = s.Y0 = s.X1 = s.Y1 = s.X2 = s.Y2 = 0; s.prim = 0; s.flags = 0; s.Az = s.Bz = s.Cz = s.bias = 0.0;
		}
		const bool nonEmpty = has && bx0 <= bx1 && by0 <= by1;
		const int area = nonEmpty ? (bx1 - bx0 + 1) * (by1 - by0 + 1) : 0;
		const bool big = nonEmpty && area > BIG_AREA;

		if (MODE == RASTER_VIS && sub == 0 && nonEmpty && !big) { RS_ADD(0, 1); RS_ADD(1, area); }
		if (nonEmpty && !big) {
			EdgeSetup E;
			edge_setup(s, bx0 + (int)sub, by0, E);
			const double gd = (double)G;
			const double gx0 = E.sx0 * gd, gx1 = E.sx1 * gd, gx2 = E.sx2 * gd;
			for (int py = by0; py <= by1; py++) {
				double e0 = E.e0, e1 = E.e1, e2 = E.e2;
				for (int px = bx0 + (int)sub; px <= bx1; px += (int)G) {
					if (edges_inside(e0, e1, e2)) fragment<MODE>(tb64, tb32, (py - oy) * VKR_TILE + (px - ox), s, px, py);
					e0 += gx0; e1 += gx1; e2 += gx2;
				}
				E.e0 += E.sy0; E.e1 += E.sy1; E.e2 += E.sy2;
			}
		}

		// wave-cooperative pass over the big sub-triangles of this wave
		unsigned long long mask = __ballot(big && sub == 0);
		while (mask) {
			const int src = __ffsll((long long)mask) - 1;
			mask &= mask - 1;
			SubTri w;
			int wx0, wy0, wx1, wy1;
			bcast_subtri(s, bx0, by0, bx1, by1, src, w, wx0, wy0, wx1, wy1);
			const int lx = (int)(lane & 7), ly = (int)(lane >> 3);
			if (MODE == RASTER_VIS && lane == 0) {
				RS_ADD(2, 1); RS_ADD(3, (wx1 - wx0 + 1) * (wy1 - wy0 + 1)); RS_ADD(4, (((wx1 - wx0) >> 3) + 1) * (((wy1 - wy0) >> 3) + 1));
				RS_ADD(5, ((wx1 - wx0 + 1) * (wy1 - wy0 + 1) + 63) / 64);
			}
			// edge values at this lane's pixel of the first 8x8 block, then stepped block by block
			EdgeSetup E;
			edge_setup(w, wx0 + lx, wy0 + ly, E);
			const double bx0s = E.sx0 * 8.0, bx1s = E.sx1 * 8.0, bx2s = E.sx2 * 8.0;
			const double by0s = E.sy0 * 8.0, by1s = E.sy1 * 8.0, by2s = E.sy2 * 8.0;
			double r0 = E.e0, r1 = E.e1, r2 = E.e2;
			for (int yy = wy0; yy <= wy1; yy += 8) {
				double e0 = r0, e1 = r1, e2 = r2;
				for (int xx = wx0; xx <= wx1; xx += 8) {
					const int px = xx + lx, py = yy + ly;
					if ((px <= wx1) && (py <= wy1) && edges_inside(e0, e1, e2))
						fragment<MODE>(tb64, tb32, (py - oy) * VKR_TILE + (px - ox), w, px, py);
					e0 += bx0s; e1 += bx1s; e2 += bx2s;
				}
				r0 += by0s; r1 += by1s; r2 += by2s;
			}
		}
	}
	WG_T(3);
	__syncthreads();
	WG_T(4);

	const uint32_t rw = a.view.x1 - a.view.x0;
	for (uint32_t i = tid; i < VKR_TILE_PIX; i += 256) {
		const int px = ox + (int)(i % VKR_TILE), py = oy + (int)(i / VKR_TILE);
		if (px > ex || py > ey) continue;
		const size_t p = (size_t)(ty * VKR_TILE + (uint32_t)(py - oy)) * rw + (size_t)(px - (int)a.view.x0);
		if (MODE == RASTER_VIS) a.vis[p] = tb64[i];
		else if (MODE == RASTER_SHADOW) a.depthOut[p] = u2f(tb32[i]);
		else a.countOut[p] = (uint8_t)min(tb32[i], 255u); // INCREMENT_AND_CLAMP, main/main.cpp:2783
	}
	if (MODE == RASTER_VIS && tid == 0) {
		// 0 = no alpha-tested entries, 1 = some, 2 + k = many: k-th entry of the list of deep tiles, whose
		// quadrants get helper workgroups in k_raster_alpha
		uint32_t flag = (uint32_t)sAnyAlpha;
		if (sAnyAlpha) atomicAdd(a.alphaTileCount, 1u);
		if (sAnyAlpha && (end - beg) - nO > ALPHA_DEEP) {
			const uint32_t k = atomicAdd(a.deepTileCount, 1u);
			a.deepTiles[k] = tile; // the list is as long as the tile array
			flag = 2u + k;
		}
		a.alphaTiles[tile] = flag;
	}
#ifdef VKR_WG_TIMES
	WG_T(5);
	if (tid == 0 && tile < 32768u) {
		for (int k = 0; k < 6; k++) g_wgTimes[MODE][8 * tile + k] = wgT[k];
		g_wgTimes[MODE][8 * tile + 6] = nO; g_wgTimes[MODE][8 * tile + 7] = n - nO;
	}
#endif
}

// ---------------------------------------------------------------------------
// Second visibility pass: the alpha-tested sub-triangles (discard on
// alpha < 0.5, gBufWrite.frag:22-23 / default.frag:91-92) of the tiles that
// hold any.  One 256-thread workgroup per 16x16 quadrant of such a tile (four
// times the workgroups of a per-tile split: the tiles that hold foliage are
// few and deep), pixel-owner: each wavefront owns one 8x8 block and every lane one pixel, whose
// (depth | primitive) key it keeps in registers, seeded from the opaque pass.
// The tile's alpha-tested bin entries are staged through LDS 128 at a time
// (edge functions, depth plane, interpolation planes, texture description,
// computed once per entry), and every wavefront walks the staged list:
//   - entries whose bounding box misses the block, or whose nearest possible
//     depth over the block lies behind the block's farthest stored depth, cost a
//     few scalar-uniform instructions;
//   - a lane runs the alpha test only if its fragment is covered, inside the
//     depth range and not behind what it already holds.
// No atomics: each pixel has one owner, and the result does not depend on the
// order of the entries (the key minimum is order-free; a skipped fragment
// could not have won).
// ---------------------------------------------------------------------------
#define ALPHA_CHUNK 256 // entries staged per round by k_overshade
#define ALPHA_STAGE 128 // and by k_raster_alpha (coverage + interpolation data: 27 KB, three workgroups per CU)

// Both staged records are 16-byte aligned so that the walk fetches them with 16-byte LDS reads (a ds_read2_b64 keeps the
// LDS array busy twice as long per byte).
struct __attribute__((aligned(16))) AlphaCover {          // coverage + depth of one staged entry
	double e0, e1, e2;       // biased edge values at the tile's first pixel centre
	double sx0, sx1, sx2;    // per pixel in x
	double sy0, sy1, sy2;    // per pixel in y
	double Az, Bz, Cz;
	uint32_t prim;
	uint32_t bbox;           // tile-local inclusive x0 | y0 << 8 | x1 << 16 | y1 << 24
	uint32_t _pad[2];
};

struct __attribute__((aligned(16))) AlphaAttr {           // what the alpha test of a fragment needs
	double a[3], b[3], c[3];
	float2 t0, t1, t2;
	uint32_t off0, w0, h0;
	int levels;
};

// Largest value over the 64 lanes, the same in every lane.  Data-parallel-primitive moves inside the vector ALU - a
// shifted running maximum inside each row of 16 lanes, then the two row broadcasts - instead of six round trips
// through the LDS crossbar (__shfl_xor): the pixel-owner kernels take this maximum once per batch of entries.
// Lanes a move does not reach take 0, the identity of an unsigned maximum.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
	int x = (int)v;
	x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false)); // row_shr:1
	x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false)); // row_shr:2
	x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false)); // row_shr:4
	x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false)); // row_shr:8: lane 15 of a row holds the row's
	x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1 and 3
	x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false)); // row_bcast:31 into rows 2 and 3: lane 63 holds all
	return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// Straight-line footprint of the alpha fetch for a texture whose sizes are powers of two: both levels are always
// addressed (the second with weight 0 when one level suffices, which leaves the value unchanged), so two
// fragments' texel loads can be in flight together.  Same arithmetic as vkr_trilinear_alpha_wh.
struct AlphaFetch {
	const uint8_t* b0; const uint8_t* b1;
	LevelTaps t0, t1;
	float fr;
};

__device__ __forceinline__ void alpha_fetch_setup(const uint8_t* __restrict__ alphaTexels, const AlphaAttr& t, int px, int py, AlphaFetch& f)
{
	TriPlanes pl;
#pragma unroll
	for (int k = 0; k < 3; k++) { pl.a[k] = t.a[k]; pl.b[k] = t.b[k]; pl.c[k] = t.c[k]; }
	pl.Az = pl.Bz = pl.Cz = 0.0;
	float b[3], uv[2], d[4];
	vkr_bary_at(pl, px, py, b);
	vkr_uv_derivs(pl, px, py, b, t.t0, t.t1, t.t2, uv, d);
	int lv;
	vkr_lod_split(vkr_rho2_wh((float)t.w0, (float)t.h0, d), t.levels - 1, lv, f.fr);
	const int lv1 = min(lv + 1, t.levels - 1);
	const int w0 = max((int)(t.w0 >> lv), 1), h0 = max((int)(t.h0 >> lv), 1);
	const int w1 = max((int)(t.w0 >> lv1), 1), h1 = max((int)(t.h0 >> lv1), 1);
	vkr_level_taps<true>(w0, h0, uv[0], uv[1], f.t0);
	vkr_level_taps<true>(w1, h1, uv[0], uv[1], f.t1);
	const uint32_t o0 = vkr_level_offset(t.off0, t.w0, t.h0, lv);
	f.b0 = alphaTexels + o0;
	f.b1 = alphaTexels + ((lv1 == lv) ? o0 : o0 + (uint32_t)w0 * (uint32_t)h0);
}

struct AlphaRaw { uint32_t a00, a10, a01, a11, c00, c10, c01, c11; };

__device__ __forceinline__ void alpha_fetch_issue(const AlphaFetch& f, AlphaRaw& r)
{
	r.a00 = f.b0[f.t0.i00]; r.a10 = f.b0[f.t0.i10]; r.a01 = f.b0[f.t0.i01]; r.a11 = f.b0[f.t0.i11];
	r.c00 = f.b1[f.t1.i00]; r.c10 = f.b1[f.t1.i10]; r.c01 = f.b1[f.t1.i01]; r.c11 = f.b1[f.t1.i11];
}

__device__ __forceinline__ float alpha_fetch_finish(const AlphaFetch& f, const AlphaRaw& r)
{
	const float a = vkr_lerp(vkr_lerp(vkr_unorm8(r.a00), vkr_unorm8(r.a10), f.t0.fx), vkr_lerp(vkr_unorm8(r.a01), vkr_unorm8(r.a11), f.t0.fx), f.t0.fy);
	const float b = vkr_lerp(vkr_lerp(vkr_unorm8(r.c00), vkr_unorm8(r.c10), f.t1.fx), vkr_lerp(vkr_unorm8(r.c01), vkr_unorm8(r.c11), f.t1.fx), f.t1.fy);
	return (f.fr == 0.0f) ? a : vkr_lerp(a, b, f.fr);
}

#define ALPHA_PARTS 1 // wavefronts that share one 8x8 block, each taking every fourth staged entry

#ifdef VKR_WG_TIMES
__device__ unsigned long long g_alphaTimes[8 * 65536];
extern "C" __attribute__((visibility("default"))) int vkr_debug_alpha_times(void* dst, size_t bytes)
{
	return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_alphaTimes), bytes, 0, hipMemcpyDeviceToHost);
}
#define AT_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atAcc[k] += t_ - atLast; atLast = t_; } } while (0)
#else
#define AT_MARK(k)
#endif

__global__ __launch_bounds__(256 * ALPHA_PARTS) void k_raster_alpha(RasterArgs a)
{
#ifdef VKR_WG_TIMES
	unsigned long long atAcc[5] = { 0, 0, 0, 0, 0 }, atLast = wall_clock64();
	const unsigned long long atStart = atLast;
#endif
	// Workgroups [0, 4 * tiles): one per quadrant.  Behind them ALPHA_HELPERS workgroups: (ALPHA_SPLIT - 1) per
	// quadrant of the first deep tiles.  The entries of a deep tile's bin section are dealt round-robin to the
	// ALPHA_SPLIT workgroups of a quadrant, each resolves its share from the same starting keys, and the shares
	// meet in an atomicMin on the visibility buffer (the minimum does not depend on who found it): a tile
	// with hundreds of overlapping alpha-tested triangles is a chain of dependent alpha tests per 8x8 block,
	// and at 1920x1080 one such chain was the whole kernel.
	// Splitting repeats the per-workgroup set-up and weakens the depth rejection (a share does not see what the
	// other shares found), so it only pays while the GPU has workgroup slots to spare: with more than
	// ALPHA_BUSY quadrants to do, the kernel is bound by throughput and the tiles are left whole.
	const uint32_t nQuads = a.view.tilesX * a.view.tilesY * 4u;
	const bool spare = a.alphaTileCount[0] * 4u <= ALPHA_BUSY;
	uint32_t tile, quad, sub = 0, split = 1;
	if (blockIdx.x < nQuads) {
		// the four quadrants of a tile read the same bin section and records: one XCD for all of them
		const uint32_t q = vkr_xcd_place<4u>(blockIdx.x, nQuads);
		tile = q / 4; quad = q & 3u;
		const uint32_t flag = a.alphaTiles[tile];
		if (!flag) return;
		if (spare && flag >= 2u && (flag - 2u) * 4u * (ALPHA_SPLIT - 1u) + 4u * (ALPHA_SPLIT - 1u) <= ALPHA_HELPERS) split = ALPHA_SPLIT;
	} else {
		const uint32_t e = blockIdx.x - nQuads;
		const uint32_t k = e / (4u * (ALPHA_SPLIT - 1u));
		if (!spare || k >= min(a.deepTileCount[0], nQuads / 4u)) return;
		tile = a.deepTiles[k];
		quad = (e / (ALPHA_SPLIT - 1u)) & 3u;
		sub = 1u + e % (ALPHA_SPLIT - 1u);
		split = ALPHA_SPLIT;
	}
	const int qx = (int)(quad & 1) * 16, qy = (int)((quad >> 1) & 1) * 16; // the quadrant, tile-local
	__shared__ AlphaCover sCov[ALPHA_STAGE];
	__shared__ AlphaAttr sAttr[ALPHA_STAGE];
	__shared__ uint32_t sCount;
	__shared__ uint32_t sZKey[ALPHA_STAGE];  // depth of the entry's plane at the tile centre
	__shared__ uint16_t sOrder[ALPHA_STAGE]; // staged entries, nearest first
	__shared__ uint16_t sListJ[4][ALPHA_STAGE];   // per block of the quadrant: the staged entries whose bounding box touches it,
	__shared__ float sListZ[4][ALPHA_STAGE];      // nearest first, with the nearest depth their plane can take over the block
	__shared__ uint32_t sListN[4], sCnt0[4];
	__shared__ unsigned long long sKey[ALPHA_PARTS][4][64]; // per-chunk exchange of the keys of the wavefronts that share a block
	const uint32_t tx = tile % a.view.tilesX, ty = tile / a.view.tilesX;
	const int ox = (int)(a.view.x0 + tx * VKR_TILE), oy = vkr_view_gy(a.view, ty * VKR_TILE);
	const int ex = min(ox + VKR_TILE, (int)a.view.x1) - 1, ey = oy + (int)min((uint32_t)VKR_TILE, a.view.rows - ty * VKR_TILE) - 1;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	// wavefront -> (block of the quadrant, part): the ALPHA_PARTS wavefronts of a block each walk every
	// ALPHA_PARTS-th staged entry on their own copy of the block's keys and take the minimum after each chunk;
	// the evaluation of one fragment is a long dependent chain, so the deep tiles are bound by how many
	// wavefronts work on them
	const uint32_t blk = wid & 3u, part = wid >> 2;
	const int bxl = qx + (int)(blk & 1) * 8, byl = qy + (int)(blk >> 1) * 8; // the wave's block, tile-local
	const int lx = bxl + (int)(lane & 7), ly = byl + (int)(lane >> 3);       // the lane's pixel, tile-local
	const int px = ox + lx, py = oy + ly;
	const bool inRect = px <= ex && py <= ey;
	const uint32_t rw = a.view.x1 - a.view.x0;
	const size_t p = (size_t)(ty * VKR_TILE + (uint32_t)ly) * rw + (size_t)(px - (int)a.view.x0);
	if (ox + qx > ex || oy + qy > ey) return; // quadrant outside the rectangle (whole workgroup)
	unsigned long long key = inRect ? a.vis[p] : 0ull;
	uint32_t wmax = wave_max_u32(inRect ? (uint32_t)(key >> 32) : 0u); // farthest depth held in the block
	const double dlx = (double)lx, dly = (double)ly;

	uint32_t beg = a.tileOffset[tile], end = a.tileOffset[tile + 1];
	if (end > a.binCap) end = a.binCap;
	if (beg > end) beg = end;
	// the alpha-tested section of the bin, then the records that touch many tiles (of either kind)
	const uint32_t nO = min(a.tileCount[tile], end - beg);
	beg += nO;
	const uint32_t nAll = end - beg;
	const uint32_t n = (nAll + split - 1u - sub) / split; // this workgroup's share: entries sub, sub + split, ...

	AT_MARK(0); // header, seed keys
	for (uint32_t base = 0; base < n; base += ALPHA_STAGE) {
		if (tid == 0) sCount = 0;
		__syncthreads();
		if (tid < ALPHA_STAGE && base + tid < n) {
			const uint32_t ri = a.binList[beg + (base + tid) * split + sub];
			const SubTri s = a.recs[ri];
			if (s.flags & VKR_ST_ALPHA) {
				int bx0, by0, bx1, by1;
				tri_bbox(s, bx0, by0, bx1, by1);
				bx0 = max(bx0, ox + qx); by0 = max(by0, oy + qy); bx1 = min(bx1, min(ex, ox + qx + 15)); by1 = min(by1, min(ey, oy + qy + 15));
				if (bx0 <= bx1 && by0 <= by1) {
					// alpha-tested passes submit in order A, so prim indexes the per-triangle tables directly
					const TriPlanes pl = a.planes[s.prim];
					const AlphaAux ax = a.sc.alphaAux[s.prim];
					{
						const uint32_t slot = atomicAdd(&sCount, 1u);
						AlphaCover& c = sCov[slot];
						EdgeSetup E;
						edge_setup(s, ox, oy, E);
						c.e0 = E.e0; c.e1 = E.e1; c.e2 = E.e2;
						c.sx0 = E.sx0; c.sx1 = E.sx1; c.sx2 = E.sx2;
						c.sy0 = E.sy0; c.sy1 = E.sy1; c.sy2 = E.sy2;
						c.Az = s.Az; c.Bz = s.Bz; c.Cz = s.Cz;
						c.prim = s.prim;
						c.bbox = (uint32_t)(bx0 - ox) | ((uint32_t)(by0 - oy) << 8) | ((uint32_t)(bx1 - ox) << 16) | ((uint32_t)(by1 - oy) << 24);
						sZKey[slot] = f2u(fminf(fmaxf(vkr_depth_at(s.Az, s.Bz, s.Cz, ox + VKR_TILE / 2, oy + VKR_TILE / 2), 0.0f), 1.0f));
						AlphaAttr& t = sAttr[slot];
#pragma unroll
						for (int k = 0; k < 3; k++) { t.a[k] = pl.a[k]; t.b[k] = pl.b[k]; t.c[k] = pl.c[k]; }
						t.t0 = ax.t0; t.t1 = ax.t1; t.t2 = ax.t2;
						t.off0 = ax.off0; t.w0 = ax.wh0 & 0xffffu; t.h0 = ax.wh0 >> 16; t.levels = ax.levels;
					}
				}
			}
		}
		__syncthreads();
		AT_MARK(1); // staging
		const uint32_t m = sCount;
		// Near entries first, so that what lies behind them fails the depth test before its alpha test: each
		// staged entry finds its rank by counting the entries in front of it.  The order changes the work, not
		// the result.
		if (tid < m) {
			const uint32_t mine = sZKey[tid];
			uint32_t rank = 0;
			for (uint32_t k = 0; k < m; k++) {
				const uint32_t other = sZKey[k];
				rank += (other < mine || (other == mine && k < tid)) ? 1u : 0u;
			}
			sOrder[rank] = (uint16_t)tid;
		}
		__syncthreads();
		// Per block, the list of the entries that can touch it, in near-to-far order (ranks 0..63 are handled by
		// wavefront 0, 64..127 by wavefront 1; positions by ballot + popcount), so that the walk below reads a
		// compact list instead of testing every staged entry.
		{
			const bool valid = tid < m;
			uint32_t j = 0, bb = 0x0000ffffu;
			double Az = 0.0, Bz = 0.0, Cz = 0.0;
			if (valid) { j = sOrder[tid]; bb = sCov[j].bbox; Az = sCov[j].Az; Bz = sCov[j].Bz; Cz = sCov[j].Cz; }
			const int x0 = (int)(bb & 255u), y0 = (int)((bb >> 8) & 255u), x1 = (int)((bb >> 16) & 255u), y1 = (int)(bb >> 24);
			// the bound must never exceed the nearest depth a fragment of the block can take: plane at the block's centre
			// minus the largest change over 3.5 pixels either way, in double, then rounded towards minus infinity (a
			// fragment's own depth is the plane's double value rounded to nearest, which rounding down never exceeds)
			const double slack = 3.5 * (fabs(Az) + fabs(Bz));
			uint32_t pos[4];
			bool ov[4];
			float zn[4];
#pragma unroll
			for (int b = 0; b < 4; b++) {
				const int bx = qx + (b & 1) * 8, by = qy + (b >> 1) * 8;
				ov[b] = valid && !(x1 < bx || x0 > bx + 7 || y1 < by || y0 > by + 7);
				// nearest depth the plane can take over the block's pixel centres
				zn[b] = __double2float_rd(fma(Az, (double)(ox + bx) + 4.0, fma(Bz, (double)(oy + by) + 4.0, Cz)) - slack);
				const unsigned long long mask = __ballot(ov[b]);
				pos[b] = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
				if (tid < 128 && lane == 0) { if (wid == 0) sCnt0[b] = (uint32_t)__popcll(mask); else sListN[b] = (uint32_t)__popcll(mask); }
			}
			__syncthreads();
			if (tid < 128) {
#pragma unroll
				for (int b = 0; b < 4; b++)
					if (ov[b]) {
						const uint32_t at = pos[b] + (wid == 1 ? sCnt0[b] : 0u);
						sListJ[b][at] = (uint16_t)j;
						sListZ[b][at] = zn[b];
					}
			}
			__syncthreads();
		}
		AT_MARK(2); // ranking + per-block lists
		// Entries are taken two at a time: the second one's coverage and depth test run against the key as it
		// is before the first one's alpha test (a fragment tested needlessly cannot change the minimum), so both
		// fragments' texel loads are in flight together.
		const uint32_t nb = sCnt0[blk] + sListN[blk];
		uint32_t jj = part;
		while (jj < nb) {
			uint32_t cj[2] = { 0, 0 };
			bool cc[2] = { false, false };
			float cz[2] = { 0.0f, 0.0f };
			int nc = 0;
			while (jj < nb && nc < 2) {
				const uint32_t j = sListJ[blk][jj];
				const float znear = sListZ[blk][jj];
				jj += ALPHA_PARTS;
				if (znear > u2f(wmax)) continue;
				const AlphaCover& c = sCov[j];
				const uint32_t bb = c.bbox;
				const int x0 = (int)(bb & 255u), y0 = (int)((bb >> 8) & 255u), x1 = (int)((bb >> 16) & 255u), y1 = (int)(bb >> 24);
				const double Az = c.Az, Bz = c.Bz, Cz = c.Cz;
				// exact: integers below 2^53
				const double e0 = fma(c.sy0, dly, fma(c.sx0, dlx, c.e0));
				const double e1 = fma(c.sy1, dly, fma(c.sx1, dlx, c.e1));
				const double e2 = fma(c.sy2, dly, fma(c.sx2, dlx, c.e2));
				bool cand = inRect && lx >= x0 && lx <= x1 && ly >= y0 && ly <= y1 && edges_inside(e0, e1, e2);
				float z = 0.0f;
				if (cand) {
					z = vkr_depth_at(Az, Bz, Cz, px, py);
					cand = (z >= 0.0f && z <= 1.0f) && !(f2u(z) > (uint32_t)(key >> 32));
				}
				if (__ballot(cand) == 0ull) continue;
				if (nc == 0) { cj[0] = j; cc[0] = cand; cz[0] = z; } else { cj[1] = j; cc[1] = cand; cz[1] = z; }
				nc++;
			}
			if (nc == 0) break;
			const AlphaAttr& tA = sAttr[cj[0]];
			const AlphaAttr& tB = sAttr[cj[1]];
			const bool pow2 = ((tA.w0 & (tA.w0 - 1)) | (tA.h0 & (tA.h0 - 1)) | (tB.w0 & (tB.w0 - 1)) | (tB.h0 & (tB.h0 - 1))) == 0;
			float alphaA, alphaB = 0.0f;
			if (nc == 2 && pow2) {
				AlphaFetch fA, fB;
				AlphaRaw rA, rB;
				alpha_fetch_setup(a.sc.alphaTexels, tA, px, py, fA);
				alpha_fetch_setup(a.sc.alphaTexels, tB, px, py, fB);
				alpha_fetch_issue(fA, rA);
				alpha_fetch_issue(fB, rB);
				alphaA = alpha_fetch_finish(fA, rA);
				alphaB = alpha_fetch_finish(fB, rB);
			} else {
				TriPlanes pl;
				float b[3], uv[2], d[4];
#pragma unroll
				for (int k = 0; k < 3; k++) { pl.a[k] = tA.a[k]; pl.b[k] = tA.b[k]; pl.c[k] = tA.c[k]; }
				pl.Az = pl.Bz = pl.Cz = 0.0;
				vkr_bary_at(pl, px, py, b);
				vkr_uv_derivs(pl, px, py, b, tA.t0, tA.t1, tA.t2, uv, d);
				alphaA = vkr_trilinear_alpha_wh(a.sc.alphaTexels, tA.off0, tA.w0, tA.h0, tA.levels, uv[0], uv[1], d);
				if (nc == 2) {
#pragma unroll
					for (int k = 0; k < 3; k++) { pl.a[k] = tB.a[k]; pl.b[k] = tB.b[k]; pl.c[k] = tB.c[k]; }
					vkr_bary_at(pl, px, py, b);
					vkr_uv_derivs(pl, px, py, b, tB.t0, tB.t1, tB.t2, uv, d);
					alphaB = vkr_trilinear_alpha_wh(a.sc.alphaTexels, tB.off0, tB.w0, tB.h0, tB.levels, uv[0], uv[1], d);
				}
			}
			bool upd = false;
			if (cc[0] && !(alphaA < 0.5f)) {
				const unsigned long long k2 = ((unsigned long long)f2u(cz[0]) << 32) | (unsigned long long)(0xFFFFFFFEu - sCov[cj[0]].prim);
				if (k2 < key) { key = k2; upd = true; }
			}
			if (nc == 2 && cc[1] && !(alphaB < 0.5f)) {
				const unsigned long long k2 = ((unsigned long long)f2u(cz[1]) << 32) | (unsigned long long)(0xFFFFFFFEu - sCov[cj[1]].prim);
				if (k2 < key) { key = k2; upd = true; }
			}
			if (__ballot(upd)) wmax = wave_max_u32(inRect ? (uint32_t)(key >> 32) : 0u);
		}
		// the block's wavefronts agree on the nearest key found so far
		sKey[part][blk][lane] = key;
		__syncthreads(); // also: the stage is rewritten by the next chunk
		AT_MARK(3); // the walk (until the slowest wavefront of the workgroup is through)
		{
			unsigned long long k = key;
#pragma unroll
			for (int q = 0; q < ALPHA_PARTS; q++) { const unsigned long long o = sKey[q][blk][lane]; k = (o < k) ? o : k; }
			if (__ballot(k != key)) { key = k; wmax = wave_max_u32(inRect ? (uint32_t)(key >> 32) : 0u); }
		}
		__syncthreads();
	}
	if (inRect && part == 0) {
		if (split > 1u) atomicMin(&a.vis[p], key);
		else a.vis[p] = key;
	}
#ifdef VKR_WG_TIMES
	AT_MARK(4);
	if (tid == 0 && blockIdx.x < 65536u) {
		for (int k = 0; k < 5; k++) g_alphaTimes[8 * blockIdx.x + k] = atAcc[k];
		g_alphaTimes[8 * blockIdx.x + 5] = atStart; g_alphaTimes[8 * blockIdx.x + 6] = nAll; g_alphaTimes[8 * blockIdx.x + 7] = 1;
	}
#endif
}

// ---------------------------------------------------------------------------
// Overshading counter (debug view 6): stencil increments only for fragments
// that pass LESS_OR_EQUAL against the depth written so far, in submission
// order (main/main.cpp:2781-2797, 4122-4123).  Order matters, so the kernel is
// pixel-owner like k_raster_alpha: one 1024-thread workgroup per tile, a
// wavefront per 8x8 block, a lane per pixel holding its depth and counter in
// registers.  The workgroup first sorts the tile's bin by primitive - a bitonic
// network over (primitive << 32 | record) words in LDS; bins longer than the
// LDS array are sorted in place in their global segment, the long-distance
// steps in global memory and the short ones chunk by chunk through LDS - and
// then walks it, staged through LDS 256 entries at a time with the edge
// functions set up once per entry.
// ---------------------------------------------------------------------------
#define OS_SORT_CAP 4096u // bin entries sorted entirely in LDS (32 KB)
#define OS_RADIX_MIN 512u  // bins longer than this take the radix sort (a flat ~9 us), shorter ones the bitonic network (45 steps at most)

// Compare-exchange steps of the normalised bitonic network (every comparator puts the smaller word at the lower
// index, so positions >= n behave as +infinity padding that never moves).  One step pairs i with i ^ mask.
__device__ __forceinline__ void os_lds_step(unsigned long long* s, uint32_t m, uint32_t mask, uint32_t tid)
{
	for (uint32_t i = tid; i < m; i += 1024u) {
		const uint32_t l = i ^ mask;
		if (l > i && l < m) {
			const unsigned long long x = s[i], y = s[l];
			if (y < x) { s[i] = y; s[l] = x; }
		}
	}
	__syncthreads();
}

// sorts s[0, m) (m a power of two <= OS_SORT_CAP) ascending; steps with block size <= kFrom / 2 are taken as done
__device__ __forceinline__ void os_lds_sort(unsigned long long* s, uint32_t m, uint32_t tid)
{
	for (uint32_t k = 2u; k <= m; k <<= 1) {
		os_lds_step(s, m, k - 1u, tid);
		for (uint32_t j = k >> 2; j >= 1u; j >>= 1) os_lds_step(s, m, j, tid);
	}
}

// Least-significant-digit radix sort of s[0, n) (n <= OS_SORT_CAP) by the primitive in the high word, 8 bits a pass, as
// many passes as the largest primitive of the bin has bytes.  The (primitive | record) words stay where they are; the
// passes permute 16-bit positions (permA <-> permB), and the words are moved once at the end.  A pass: every wavefront
// ranks its 256 consecutive positions, 64 at a time (the lanes with equal digits find each other through eight
// ballots; their rank is a popcount, the running count per (wavefront, digit) lives in hist), the 16 x 256 counts are
// scanned digit-major, and the positions are scattered.  14 barriers for three passes against the 45-78 steps of
// the bitonic network on 512-4096 entries.  scratch: OS_RADIX_SCRATCH bytes the caller does not need meanwhile.
#define OS_RADIX_SCRATCH (2u * OS_SORT_CAP * 2u + 16u * 256u * 2u + 128u)
__device__ __noinline__ void os_radix_sort(unsigned long long* s, uint32_t n, unsigned char* scratch, uint32_t tid)
{
	uint16_t* permA = (uint16_t*)scratch;
	uint16_t* permB = permA + OS_SORT_CAP;
	uint16_t* hist = permB + OS_SORT_CAP;          // [wavefront][digit]
	uint32_t* wtot = (uint32_t*)(hist + 16 * 256); // per-wavefront totals of the scan, [16] = largest primitive
	const uint32_t lane = tid & 63u, w = tid >> 6;
	const unsigned long long ltMask = (1ull << lane) - 1ull;
	// how many digits the keys have
	if (tid == 0) wtot[16] = 0u;
	__syncthreads();
	{
		uint32_t mx = 0;
		for (uint32_t i = tid; i < n; i += 1024u) mx = max(mx, (uint32_t)(s[i] >> 32));
		mx = wave_max_u32(mx);
		if (lane == 0) atomicMax(&wtot[16], mx);
	}
	__syncthreads();
	const uint32_t top = wtot[16];
	const uint32_t passes = top < (1u << 8) ? 1u : top < (1u << 16) ? 2u : top < (1u << 24) ? 3u : 4u;
	uint16_t* src = permA;
	uint16_t* dst = permB;
	const uint32_t nW = (n + 255u) >> 8; // wavefronts that hold positions; the others only keep the barriers
	for (uint32_t pass = 0; pass < passes; pass++) {
		const uint32_t shift = 8u * pass;
		uint32_t idx[4], dig[4], loc[4];
		bool val[4];
		if (w < nW) {
			for (uint32_t k = lane; k < 256u; k += 64u) hist[w * 256u + k] = 0;
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const uint32_t i = w * 256u + (uint32_t)r * 64u + lane;
				val[r] = i < n;
				idx[r] = 0u; dig[r] = 0u; loc[r] = 0u;
				if (w * 256u + (uint32_t)r * 64u >= n) continue; // wave-uniform: nothing left for this wavefront
				idx[r] = val[r] ? (pass ? (uint32_t)src[i] : i) : 0u;
				dig[r] = val[r] ? (((uint32_t)(s[idx[r]] >> 32)) >> shift) & 255u : 0u;
				unsigned long long peers = __ballot(val[r]);
#pragma unroll
				for (int b = 0; b < 8; b++) {
					const unsigned long long bal = __ballot((dig[r] >> b) & 1u);
					peers &= ((dig[r] >> b) & 1u) ? bal : ~bal;
				}
				const uint32_t rank = (uint32_t)__popcll(peers & ltMask);
				uint32_t prev = 0;
				if (val[r]) prev = hist[w * 256u + dig[r]];
				__builtin_amdgcn_wave_barrier(); // every peer has read the count before their first lane raises it
				if (val[r] && rank == 0) hist[w * 256u + dig[r]] = (uint16_t)(prev + (uint32_t)__popcll(peers));
				__builtin_amdgcn_wave_barrier();
				loc[r] = prev + rank;
			}
		} else {
#pragma unroll
			for (int r = 0; r < 4; r++) { val[r] = false; idx[r] = dig[r] = loc[r] = 0u; }
		}
		__syncthreads();
		// exclusive scan of the counts in (digit, wavefront) order: thread t owns digit t / 4, wavefronts 4 (t % 4) .. + 3
		{
			const uint32_t d = tid >> 2, w0 = (tid & 3u) * 4u;
			uint32_t c[4], sum = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) { c[k] = (w0 + k < nW) ? (uint32_t)hist[(w0 + k) * 256u + d] : 0u; sum += c[k]; }
			const uint32_t inc = vkr_wave_scan_incl_u32(sum);
			if (lane == 63) wtot[w] = inc;
			__syncthreads();
			uint32_t base = 0;
#pragma unroll
			for (uint32_t k = 0; k < 16u; k++) base += (k < w) ? wtot[k] : 0u;
			uint32_t run = base + inc - sum;
#pragma unroll
			for (int k = 0; k < 4; k++) { if (w0 + k < nW) hist[(w0 + k) * 256u + d] = (uint16_t)run; run += c[k]; }
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < 4; r++)
			if (val[r]) dst[hist[w * 256u + dig[r]] + loc[r]] = (uint16_t)idx[r];
		__syncthreads();
		uint16_t* t = src; src = dst; dst = t;
	}
	// move the words into their places
	unsigned long long v[4];
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint32_t i = tid + (uint32_t)r * 1024u;
		v[r] = (i < n) ? s[src[i]] : 0ull;
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint32_t i = tid + (uint32_t)r * 1024u;
		if (i < n) s[i] = v[r];
	}
	__syncthreads();
}

__device__ __forceinline__ void os_global_step(uint32_t* keys, uint32_t* vals, uint32_t n, uint32_t m, uint32_t mask, uint32_t tid)
{
	for (uint32_t i = tid; i < m; i += 1024u) {
		const uint32_t l = i ^ mask;
		if (l > i && l < n) {
			const unsigned long long x = ((unsigned long long)keys[i] << 32) | vals[i], y = ((unsigned long long)keys[l] << 32) | vals[l];
			if (y < x) { keys[i] = (uint32_t)(y >> 32); vals[i] = (uint32_t)y; keys[l] = (uint32_t)(x >> 32); vals[l] = (uint32_t)x; }
		}
	}
	__syncthreads(); // one workgroup owns the segment: its own stores are visible to it behind the barrier
}

// In-place sort of a bin longer than the LDS array: (keys[i], vals[i]) ascending by (key, val).
__device__ __noinline__ void os_sort_long(unsigned long long* s, uint32_t* keys, uint32_t* vals, uint32_t n, uint32_t tid)
{
	uint32_t m = OS_SORT_CAP;
	while (m < n) m <<= 1;
	const uint32_t chunks = (n + OS_SORT_CAP - 1u) / OS_SORT_CAP;
	for (uint32_t k = OS_SORT_CAP; k <= m; k <<= 1) {
		// block size k: the first round sorts every chunk outright; later rounds take the steps that pair positions
		// at least OS_SORT_CAP apart in global memory, then finish every chunk's remaining steps in LDS
		if (k > OS_SORT_CAP) {
			os_global_step(keys, vals, n, m, k - 1u, tid);
			for (uint32_t j = k >> 2; j >= OS_SORT_CAP; j >>= 1) os_global_step(keys, vals, n, m, j, tid);
		}
		for (uint32_t c = 0; c < chunks; c++) {
			const uint32_t base = c * OS_SORT_CAP;
			for (uint32_t i = tid; i < OS_SORT_CAP; i += 1024u)
				s[i] = (base + i < n) ? (((unsigned long long)keys[base + i] << 32) | vals[base + i]) : ~0ull;
			__syncthreads();
			if (k == OS_SORT_CAP) os_lds_sort(s, OS_SORT_CAP, tid);
			else for (uint32_t j = OS_SORT_CAP >> 1; j >= 1u; j >>= 1) os_lds_step(s, OS_SORT_CAP, j, tid);
			for (uint32_t i = tid; i < OS_SORT_CAP; i += 1024u)
				if (base + i < n) { keys[base + i] = (uint32_t)(s[i] >> 32); vals[base + i] = (uint32_t)s[i]; }
			__syncthreads();
		}
	}
}

// coverage + depth of one staged entry of k_overshade: edge values at the tile's first pixel centre, then the
// per-pixel steps in x and in y (three each)
// (16-byte aligned, so that the walk fetches it with 16-byte LDS reads: a ds_read2_b64 takes the LDS array twice as long
// per byte, and with 16 wavefronts walking the array is what the deep tiles wait for)
struct __attribute__((aligned(16))) OsCover {
	double e[9];
	double Az, Bz, Cz;
	uint32_t bbox;           // tile-local inclusive x0 | y0 << 8 | x1 << 16 | y1 << 24
	uint32_t _pad[3];
};

#ifdef VKR_WG_TIMES
// measurement build only: per tile of the last k_overshade launch: start, sorted, done (100 MHz clock), bin entries
__device__ unsigned long long g_osTimes[8 * 32768];
extern "C" __attribute__((visibility("default"))) int vkr_debug_overshade_times(void* dst, size_t bytes)
{
	return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_osTimes), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

// 8 wavefronts per SIMD = two of these 16-wavefront workgroups per CU (LDS: 2 x 58 KB): one register over 64 would
// leave every CU with a single workgroup, i.e. half the tiles in flight.
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_overshade(OvershadeArgs a)
{
#ifdef VKR_WG_TIMES
	const unsigned long long osT0 = wall_clock64();
#endif
	__shared__ OsCover sCov[ALPHA_CHUNK];
	__shared__ unsigned long long sSort[OS_SORT_CAP];
#ifdef VKR_WG_TIMES
	__shared__ uint32_t sStat[4]; // entries walked (summed over the wavefronts), the most by one wavefront, box hits culled by depth
	if (threadIdx.x < 4) sStat[threadIdx.x] = 0;
	uint32_t myHits = 0, myCulled = 0;
#endif
	// deepest bins first (the order comes from the tile scan): the kernel's time is the time of its longest tiles,
	// which in raster order start somewhere in the middle of the launch
	const uint32_t tile = a.tileOrder ? a.tileOrder[blockIdx.x] : blockIdx.x;
	const uint32_t tx = tile % a.view.tilesX, ty = tile / a.view.tilesX;
	const int ox = (int)(a.view.x0 + tx * VKR_TILE), oy = vkr_view_gy(a.view, ty * VKR_TILE);
	const int ex = min(ox + VKR_TILE, (int)a.view.x1) - 1, ey = oy + (int)min((uint32_t)VKR_TILE, a.view.rows - ty * VKR_TILE) - 1;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const int bxl = (int)(wid & 3) * 8, byl = (int)(wid >> 2) * 8;
	const int lx = bxl + (int)(lane & 7), ly = byl + (int)(lane >> 3);
	const int px = ox + lx, py = oy + ly;
	const bool inRect = px <= ex && py <= ey;
	const double dlx = (double)lx, dly = (double)ly;
	float depth = 1.0f;
	uint32_t cnt = 0;

	uint32_t beg = a.tileOffset[tile], end = a.tileOffset[tile + 1];
	if (end > a.binCap) end = a.binCap;
	if (beg > end) beg = end;
	const uint32_t n = end - beg;
	// submission order: the bin sorted by primitive (two fan triangles of one clipped primitive never cover the same
	// pixel, so their mutual order is immaterial; the record index makes the order total all the same)
	const bool inLds = n <= OS_SORT_CAP;
	static_assert(sizeof(OsCover) * ALPHA_CHUNK >= OS_RADIX_SCRATCH, "the staging array doubles as the radix sort's scratch");
	if (inLds && n > OS_RADIX_MIN) {
		for (uint32_t i = tid; i < n; i += 1024u)
			sSort[i] = ((unsigned long long)a.binKeys[beg + i] << 32) | a.binList[beg + i];
		__syncthreads();
		os_radix_sort(sSort, n, (unsigned char*)sCov, tid); // the staging array is idle until the walk
	} else if (inLds) {
		uint32_t m = 1u;
		while (m < n) m <<= 1;
		for (uint32_t i = tid; i < m; i += 1024u)
			sSort[i] = (i < n) ? (((unsigned long long)a.binKeys[beg + i] << 32) | a.binList[beg + i]) : ~0ull;
		__syncthreads();
		os_lds_sort(sSort, m, tid);
	} else {
		os_sort_long(sSort, a.binKeys + beg, a.binList + beg, n, tid);
	}
#ifdef VKR_WG_TIMES
	__syncthreads();
	const unsigned long long osT1 = wall_clock64();
#endif
#ifdef VKR_WG_TIMES
	unsigned long long osStage = 0;
#endif
	for (uint32_t base = 0; base < n; base += ALPHA_CHUNK) {
		__syncthreads();
#ifdef VKR_WG_TIMES
		const unsigned long long osS0 = wall_clock64();
#endif
		if (tid < ALPHA_CHUNK && base + tid < n) {
			// (requesting the next chunk's record while this one is walked costs 16 registers the kernel does not have: spills, +8 %)
			const SubTri s = a.recs[inLds ? (uint32_t)sSort[base + tid] : a.binList[beg + base + tid]];
			OsCover& c = sCov[tid];
			int bx0, by0, bx1, by1;
			tri_bbox(s, bx0, by0, bx1, by1);
			bx0 = max(bx0, ox); by0 = max(by0, oy); bx1 = min(bx1, ex); by1 = min(by1, ey);
			if (bx0 <= bx1 && by0 <= by1) {
				EdgeSetup E;
				edge_setup(s, ox, oy, E);
				c.e[0] = E.e0; c.e[1] = E.e1; c.e[2] = E.e2;
				c.e[3] = E.sx0; c.e[4] = E.sx1; c.e[5] = E.sx2;
				c.e[6] = E.sy0; c.e[7] = E.sy1; c.e[8] = E.sy2;
				c.Az = s.Az; c.Bz = s.Bz; c.Cz = s.Cz;
				c.bbox = (uint32_t)(bx0 - ox) | ((uint32_t)(by0 - oy) << 8) | ((uint32_t)(bx1 - ox) << 16) | ((uint32_t)(by1 - oy) << 24);
			} else {
				c.bbox = 0x0000ffffu; // x0 = y0 = 255 > x1 = y1 = 0: misses every block
			}
		}
		__syncthreads();
#ifdef VKR_WG_TIMES
		osStage += wall_clock64() - osS0;
#endif
		const uint32_t m = min((uint32_t)ALPHA_CHUNK, n - base);
		for (uint32_t j0 = 0; j0 < m; j0 += 64u) {
			// 64 entries at a time: every lane tests one entry against the wavefront's block - its bounding box, and
			// the nearest depth its plane can take over the block's pixel centres against the farthest depth the block
			// holds (depths only come nearer while the batch is walked, so the bound taken here stays valid): an entry
			// that lies behind everything drawn so far passes LESS_OR_EQUAL nowhere and changes nothing.  The entries
			// left are then walked in order (the bits of the ballot, lowest first).
			const float wfar = u2f(wave_max_u32(inRect ? f2u(depth) : 0u));
			bool hit = false;
			if (j0 + lane < m) {
				const OsCover& c = sCov[j0 + lane];
				const uint32_t bb = c.bbox;
				const int x0 = (int)(bb & 255u), y0 = (int)((bb >> 8) & 255u), x1 = (int)((bb >> 16) & 255u), y1 = (int)(bb >> 24);
				hit = !(x1 < bxl || x0 > bxl + 7 || y1 < byl || y0 > byl + 7);
				if (hit) {
					// plane at the block's centre minus its largest change over 3.5 pixels either way, in double, rounded
					// down: never above the depth of a fragment of the block (the plane's double value rounded to nearest)
					const double Az = c.Az, Bz = c.Bz;
					const float znear = __double2float_rd(fma(Az, (double)(ox + bxl) + 4.0, fma(Bz, (double)(oy + byl) + 4.0, c.Cz)) - 3.5 * (fabs(Az) + fabs(Bz)));
#ifdef VKR_WG_TIMES
					if (znear > wfar) myCulled++;
#endif
					hit = !(znear > wfar);
				}
			}
			unsigned long long mask = __ballot(hit);
#ifdef VKR_WG_TIMES
			myHits += (uint32_t)__popcll(mask);
#endif
			while (mask) {
				const uint32_t j = j0 + (uint32_t)(__ffsll((long long)mask) - 1);
				mask &= mask - 1;
				const OsCover& c = sCov[j];
				// exact (integers below 2^53); a pixel centre inside all three edges lies inside the entry's bounding box, so
				// there is no box test per lane.  (Entries with short edges evaluated in 32-bit integers instead - 24-bit
				// multiplies, a sign test on the OR - is exact too, but the second code path costs registers the kernel does
				// not have under its cap of 64: spills, +7 %.)
				const double e0 = fma(c.e[6], dly, fma(c.e[3], dlx, c.e[0]));
				const double e1 = fma(c.e[7], dly, fma(c.e[4], dlx, c.e[1]));
				const double e2 = fma(c.e[8], dly, fma(c.e[5], dlx, c.e[2]));
				const bool in = edges_inside(e0, e1, e2);
				if (!(inRect && in)) continue;
				const float z = vkr_depth_at(c.Az, c.Bz, c.Cz, px, py);
				if (!(z >= 0.0f && z <= 1.0f)) continue;
				if (!(z <= depth)) continue;
				depth = z;
				if (cnt < 255u) cnt++;
			}
		}
	}
	if (inRect) {
		const uint32_t rw = a.view.x1 - a.view.x0;
		const size_t p = (size_t)(ty * VKR_TILE + (uint32_t)ly) * rw + (size_t)(px - (int)a.view.x0);
		a.countOut[p] = (uint8_t)cnt;
		if (a.depthOut) a.depthOut[p] = depth;
	}
#ifdef VKR_WG_TIMES
	if (lane == 0) { atomicAdd(&sStat[0], myHits); atomicMax(&sStat[1], myHits); }
	atomicAdd(&sStat[2], myCulled);
	__syncthreads();
	if (tid == 0 && tile < 32768u) {
		g_osTimes[8 * tile] = osT0; g_osTimes[8 * tile + 1] = osT1; g_osTimes[8 * tile + 2] = wall_clock64(); g_osTimes[8 * tile + 3] = n;
		g_osTimes[8 * tile + 4] = sStat[0]; g_osTimes[8 * tile + 5] = sStat[1]; g_osTimes[8 * tile + 6] = sStat[2]; g_osTimes[8 * tile + 7] = osStage;
	}
#endif
}

void vkr_launch_raster(hipStream_t s, int mode, const RasterArgs& a)
{
	const uint32_t nTiles = a.view.tilesX * a.view.tilesY;
	if (!nTiles) return;
	if (mode == RASTER_VIS) {
		hipLaunchKernelGGL(k_raster<RASTER_VIS>, dim3(nTiles), dim3(256), 0, s, a);
		if (a.alphaPass) hipLaunchKernelGGL(k_raster_alpha, dim3(nTiles * 4 + ALPHA_HELPERS), dim3(256 * ALPHA_PARTS), 0, s, a);
	}
	else if (mode == RASTER_SHADOW) hipLaunchKernelGGL(k_raster<RASTER_SHADOW>, dim3(nTiles), dim3(256), 0, s, a);
	else hipLaunchKernelGGL(k_raster<RASTER_COUNT>, dim3(nTiles), dim3(256), 0, s, a);
}

void vkr_launch_overshade(hipStream_t s, const OvershadeArgs& a)
{
	const uint32_t nTiles = a.view.tilesX * a.view.tilesY;
	if (!nTiles) return;
	hipLaunchKernelGGL(k_overshade, dim3(nTiles), dim3(1024), 0, s, a);
}
