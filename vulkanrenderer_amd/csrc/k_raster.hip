// k_raster.hip — tile rasteriser.
//
// One workgroup owns one 32x32 screen tile whose visibility buffer lives in
// LDS.  The tile's bin (indices of the sub-triangles that touch it) is walked
// 256 entries at a time: small sub-triangles are scan-converted one per lane,
// large or alpha-tested ones by a whole wavefront (one lane per pixel of an
// 8x8 block).  Depth test + write is a single 64-bit LDS atomicMin on
// (depthBits << 32) | (0xFFFFFFFE - prim): smaller depth wins and, at equal
// depth, the later primitive wins, which is VK_COMPARE_OP_LESS_OR_EQUAL
// applied in submission order (main/main.cpp:2321-2327).  The finished tile
// is written to HBM once, coalesced.
//
// Replaces the fixed-function rasteriser + depth/stencil unit for the draws of
// main/main.cpp:3873-3934 (G-buffer pass), :3968-3982 (shadow pass),
// :4014-4075 (forward pass), :4111-4126 (overdraw counter) and :4164-4190
// (debug views).  Compiled with -ffp-contract=off.
#include "vkr_kernels.h"

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
	E.e0 = fma((double)dx0, (double)(PY - s.Y0), -((double)dy0 * (double)(PX - s.X0))) - (tl0 ? 0.0 : 1.0);
	E.e1 = fma((double)dx1, (double)(PY - s.Y1), -((double)dy1 * (double)(PX - s.X1))) - (tl1 ? 0.0 : 1.0);
	E.e2 = fma((double)dx2, (double)(PY - s.Y2), -((double)dy2 * (double)(PX - s.X2))) - (tl2 ? 0.0 : 1.0);
	E.sx0 = -(double)dy0 * 256.0; E.sx1 = -(double)dy1 * 256.0; E.sx2 = -(double)dy2 * 256.0;
	E.sy0 = (double)dx0 * 256.0; E.sy1 = (double)dx1 * 256.0; E.sy2 = (double)dx2 * 256.0;
}

__device__ __forceinline__ bool edges_inside(double e0, double e1, double e2)
{
	return fmin(e0, fmin(e1, e2)) >= 0.0;
}

__device__ __forceinline__ void tri_bbox(const SubTri& s, int& px0, int& py0, int& px1, int& py1)
{
	int32_t xmin = min(s.X0, min(s.X1, s.X2)), xmax = max(s.X0, max(s.X1, s.X2));
	int32_t ymin = min(s.Y0, min(s.Y1, s.Y2)), ymax = max(s.Y0, max(s.Y1, s.Y2));
	px0 = (xmin - 128 + 255) >> 8; px1 = (xmax - 128) >> 8;
	py0 = (ymin - 128 + 255) >> 8; py1 = (ymax - 128) >> 8;
}

// Alpha test of gBufWrite.frag:22-23 / default.frag:91-92 for one fragment.
__device__ __forceinline__ bool alpha_pass(const RasterArgs& a, const TriPlanes& pl, float2 t0, float2 t1, float2 t2,
                                           const TexInfo& tex, int px, int py)
{
	float b[3], uv[2], d[4];
	vkr_bary_at(pl, px, py, b);
	vkr_uv_derivs(pl, px, py, b, t0, t1, t2, uv, d);
	return !(vkr_trilinear_alpha(a.sc, tex, uv[0], uv[1], d) < 0.5f);
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

// Broadcasts lane `src`'s sub-triangle and its tile-clipped bounding box to the wave.
__device__ __forceinline__ void bcast_subtri(const SubTri& s, int bx0, int by0, int bx1, int by1, int src, SubTri& w, int& wx0,
                                             int& wy0, int& wx1, int& wy1)
{
	w.X0 = __shfl(s.X0, src, 64); w.Y0 = __shfl(s.Y0, src, 64);
	w.X1 = __shfl(s.X1, src, 64); w.Y1 = __shfl(s.Y1, src, 64);
	w.X2 = __shfl(s.X2, src, 64); w.Y2 = __shfl(s.Y2, src, 64);
	w.prim = __shfl(s.prim, src, 64); w.flags = __shfl(s.flags, src, 64);
	w.Az = __shfl(s.Az, src, 64); w.Bz = __shfl(s.Bz, src, 64); w.Cz = __shfl(s.Cz, src, 64);
	w.bias = __shfl(s.bias, src, 64);
	wx0 = __shfl(bx0, src, 64); wy0 = __shfl(by0, src, 64);
	wx1 = __shfl(bx1, src, 64); wy1 = __shfl(by1, src, 64);
}

// Every sub-triangle of the tile's bin that needs no alpha test.  In
// RASTER_VIS mode alpha-tested sub-triangles are left to k_raster_alpha and
// such tiles are appended to alphaTiles.
template <int MODE>
__global__ __launch_bounds__(256) void k_raster(RasterArgs a)
{
	__shared__ unsigned long long tb64[MODE == RASTER_VIS ? VKR_TILE_PIX : 1];
	__shared__ uint32_t tb32[MODE == RASTER_VIS ? 1 : VKR_TILE_PIX];
	__shared__ int sAnyAlpha;
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % a.view.tilesX, ty = tile / a.view.tilesX;
	const int ox = (int)(a.view.x0 + tx * VKR_TILE), oy = (int)(a.view.y0 + ty * VKR_TILE);
	const int ex = min(ox + VKR_TILE, (int)a.view.x1) - 1, ey = min(oy + VKR_TILE, (int)a.view.y1) - 1; // inclusive
	const uint32_t tid = threadIdx.x, lane = tid & 63;

	for (uint32_t i = tid; i < VKR_TILE_PIX; i += 256) {
		if (MODE == RASTER_VIS) tb64[i] = 0x3F800000FFFFFFFFull;
		else if (MODE == RASTER_SHADOW) tb32[i] = 0x3F800000u;
		else tb32[i] = 0u;
	}
	if (tid == 0) sAnyAlpha = 0;
	__syncthreads();

	uint32_t beg = a.tileOffset[tile], end = a.tileOffset[tile + 1];
	if (end > a.binCap) end = a.binCap;
	if (beg > end) beg = end;
	const uint32_t n = end - beg;

	for (uint32_t base = 0; base < n; base += 256) {
		const uint32_t i = base + tid;
		bool has = i < n;
		SubTri s;
		int bx0 = 0, by0 = 0, bx1 = -1, by1 = -1;
		if (has) {
			s = a.recs[a.binList[beg + i]];
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

		if (nonEmpty && !big) {
			EdgeSetup E;
			edge_setup(s, bx0, by0, E);
			for (int py = by0; py <= by1; py++) {
				double e0 = E.e0, e1 = E.e1, e2 = E.e2;
				for (int px = bx0; px <= bx1; px++) {
					if (edges_inside(e0, e1, e2)) fragment<MODE>(tb64, tb32, (py - oy) * VKR_TILE + (px - ox), s, px, py);
					e0 += E.sx0; e1 += E.sx1; e2 += E.sx2;
				}
				E.e0 += E.sy0; E.e1 += E.sy1; E.e2 += E.sy2;
			}
		}

		// wave-cooperative pass over the big sub-triangles of this wave
		unsigned long long mask = __ballot(big);
		while (mask) {
			const int src = __ffsll((long long)mask) - 1;
			mask &= mask - 1;
			SubTri w;
			int wx0, wy0, wx1, wy1;
			bcast_subtri(s, bx0, by0, bx1, by1, src, w, wx0, wy0, wx1, wy1);
			const int lx = (int)(lane & 7), ly = (int)(lane >> 3);
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
	__syncthreads();

	const uint32_t rw = a.view.x1 - a.view.x0;
	for (uint32_t i = tid; i < VKR_TILE_PIX; i += 256) {
		const int px = ox + (int)(i % VKR_TILE), py = oy + (int)(i / VKR_TILE);
		if (px > ex || py > ey) continue;
		const size_t p = (size_t)(py - (int)a.view.y0) * rw + (size_t)(px - (int)a.view.x0);
		if (MODE == RASTER_VIS) a.vis[p] = tb64[i];
		else if (MODE == RASTER_SHADOW) a.depthOut[p] = u2f(tb32[i]);
		else a.countOut[p] = (uint8_t)min(tb32[i], 255u); // INCREMENT_AND_CLAMP, main/main.cpp:2783
	}
	if (MODE == RASTER_VIS && tid == 0 && sAnyAlpha) a.alphaTiles[atomicAdd(a.alphaTileCount, 1u)] = tile;
}

// ---------------------------------------------------------------------------
// Second visibility pass: the alpha-tested sub-triangles (discard on
// alpha < 0.5, gBufWrite.frag:22-23 / default.frag:91-92) of the tiles that
// hold any.  The tile comes back from HBM into LDS with every opaque fragment
// in it, and a hierarchical depth test keeps most of the work away from the
// texture:
//   - per 8x8 block the farthest stored depth is kept in LDS; a sub-triangle
//     whose nearest possible depth over a block lies behind it skips the block;
//   - inside a block, fragments behind the stored depth are dropped;
//   - the survivors are packed into a per-wave queue and alpha-tested 64 at a
//     time with every lane busy.
// Interpolation planes are computed once per sub-triangle, lane-parallel, and
// staged in LDS; each wave walks its sub-triangles near-to-far.
// ---------------------------------------------------------------------------
struct AlphaStage {
	double a[3], b[3], c[3];
	float2 t0, t1, t2;
	uint32_t tex;
	uint32_t ok;
};

// farthest depth stored in the 8x8 block the wave's lanes map onto
__device__ __forceinline__ uint32_t block_max_depth(const unsigned long long* tb64, int bxl, int byl, uint32_t lane)
{
	uint32_t d = (uint32_t)(tb64[(byl * 8 + (int)(lane >> 3)) * VKR_TILE + bxl * 8 + (int)(lane & 7)] >> 32);
#pragma unroll
	for (int k = 32; k >= 1; k >>= 1) d = max(d, (uint32_t)__shfl_xor((int)d, k, 64));
	return d;
}

__global__ __launch_bounds__(256) void k_raster_alpha(RasterArgs a)
{
	// work item = (tile with alpha-tested entries, slice of its bin): a dense tile is spread over
	// VKR_ALPHA_SLICES workgroups, each with its own LDS copy of the tile, merged by atomicMin
	const uint32_t nAlpha = min(a.alphaTileCount[0], a.view.tilesX * a.view.tilesY);
	if (blockIdx.x / VKR_ALPHA_SLICES >= nAlpha) return;
	const uint32_t tile = a.alphaTiles[blockIdx.x / VKR_ALPHA_SLICES], slice = blockIdx.x % VKR_ALPHA_SLICES;
	__shared__ unsigned long long tb64[VKR_TILE_PIX];
	__shared__ AlphaStage stage[256];
	__shared__ uint16_t sQueue[4 * 128];
	__shared__ uint32_t sBlockMax[16]; // depth bits; stale values are only ever too far (conservative)
	uint32_t* tb32 = nullptr; // unused, keeps fragment<>'s signature
	const uint32_t tx = tile % a.view.tilesX, ty = tile / a.view.tilesX;
	const int ox = (int)(a.view.x0 + tx * VKR_TILE), oy = (int)(a.view.y0 + ty * VKR_TILE);
	const int ex = min(ox + VKR_TILE, (int)a.view.x1) - 1, ey = min(oy + VKR_TILE, (int)a.view.y1) - 1;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wbase = tid & ~63u;
	uint16_t* wq = &sQueue[wid * 128];
	const uint32_t rw = a.view.x1 - a.view.x0;

	for (uint32_t i = tid; i < VKR_TILE_PIX; i += 256) {
		const int px = ox + (int)(i % VKR_TILE), py = oy + (int)(i / VKR_TILE);
		unsigned long long v = 0ull; // outside the rectangle: nothing can be nearer
		if (px <= ex && py <= ey) v = a.vis[(size_t)(py - (int)a.view.y0) * rw + (size_t)(px - (int)a.view.x0)];
		tb64[i] = v;
	}
	__syncthreads();
	for (int b = (int)wid; b < 16; b += 4) {
		const uint32_t m = block_max_depth(tb64, b & 3, b >> 2, lane);
		if (lane == 0) sBlockMax[b] = m;
	}

	uint32_t beg = a.tileOffset[tile], end = a.tileOffset[tile + 1];
	if (end > a.binCap) end = a.binCap;
	if (beg > end) beg = end;
	const uint32_t n = end - beg;
	if (slice * 256 >= n) return;

	for (uint32_t base = slice * 256; base < n; base += VKR_ALPHA_SLICES * 256) {
		const uint32_t i = base + tid;
		bool has = i < n;
		SubTri s;
		int bx0 = 0, by0 = 0, bx1 = -1, by1 = -1;
		if (has) {
			const uint32_t ri = a.binList[beg + i];
			has = (a.recs[ri].flags & VKR_ST_ALPHA) != 0;
			if (has) s = a.recs[ri];
		}
		if (has) {
			tri_bbox(s, bx0, by0, bx1, by1);
			bx0 = max(bx0, ox); by0 = max(by0, oy); bx1 = min(bx1, ex); by1 = min(by1, ey);
		} else {
			s.X0 = s.Y0 = s.X1 = s.Y1 = s.X2 = s.Y2 = 0; s.prim = 0; s.flags = 0; s.Az = s.Bz = s.Cz = s.bias = 0.0;
		}
		const bool nonEmpty = has && bx0 <= bx1 && by0 <= by1;
		if (nonEmpty) {
			// alpha-tested passes submit in order A, so prim indexes sc.tri directly
			const uint32_t i0 = a.sc.tri[3 * (size_t)s.prim + 0], i1 = a.sc.tri[3 * (size_t)s.prim + 1],
			               i2 = a.sc.tri[3 * (size_t)s.prim + 2];
			TriPlanes pl;
			AlphaStage& st = stage[tid];
			st.ok = vkr_tri_planes<false>(a.clip[i0], a.clip[i1], a.clip[i2], a.view.W, a.view.H, pl) ? 1u : 0u;
#pragma unroll
			for (int k = 0; k < 3; k++) { st.a[k] = pl.a[k]; st.b[k] = pl.b[k]; st.c[k] = pl.c[k]; }
			st.t0 = make_float2(a.sc.uv[2 * (size_t)i0], a.sc.uv[2 * (size_t)i0 + 1]);
			st.t1 = make_float2(a.sc.uv[2 * (size_t)i1], a.sc.uv[2 * (size_t)i1 + 1]);
			st.t2 = make_float2(a.sc.uv[2 * (size_t)i2], a.sc.uv[2 * (size_t)i2 + 1]);
			st.tex = a.sc.mat[a.sc.triMat[s.prim]].tex[3];
		}
		// near entries first: 8 rounds over the depth (at the tile centre) range of the wave's entries
		const float zc = nonEmpty ? fminf(fmaxf(vkr_depth_at(s.Az, s.Bz, s.Cz, ox + VKR_TILE / 2, oy + VKR_TILE / 2), 0.0f), 1.0f) : 2.0f;
		float zlo = nonEmpty ? zc : 2.0f, zhi = nonEmpty ? zc : -1.0f;
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) {
			zlo = fminf(zlo, __shfl_xor(zlo, d, 64));
			zhi = fmaxf(zhi, __shfl_xor(zhi, d, 64));
		}
		const float zscale = (zhi > zlo) ? 7.999f / (zhi - zlo) : 0.0f;
		const int bucket = nonEmpty ? (int)((zc - zlo) * zscale) : -1;
		__syncthreads();

		for (int round = 0; round < 8; round++) {
			unsigned long long mask = __ballot(bucket == round);
			while (mask) {
				const int src = __ffsll((long long)mask) - 1;
				mask &= mask - 1;
				SubTri w;
				int wx0, wy0, wx1, wy1;
				bcast_subtri(s, bx0, by0, bx1, by1, src, w, wx0, wy0, wx1, wy1);
				const AlphaStage& st = stage[wbase + (uint32_t)src];
				if (!st.ok) continue;
				TriPlanes pl;
#pragma unroll
				for (int k = 0; k < 3; k++) { pl.a[k] = st.a[k]; pl.b[k] = st.b[k]; pl.c[k] = st.c[k]; }
				pl.Az = pl.Bz = pl.Cz = 0.0;
				const float2 t0 = st.t0, t1 = st.t1, t2 = st.t2;
				const TexInfo& tex = a.sc.tex[st.tex];
				const int lx = (int)(lane & 7), ly = (int)(lane >> 3);
				// walk the tile's own 8x8 blocks that the clipped bounding box touches
				const int cbx0 = (wx0 - ox) >> 3, cbx1 = (wx1 - ox) >> 3, cby0 = (wy0 - oy) >> 3, cby1 = (wy1 - oy) >> 3;
				EdgeSetup E;
				edge_setup(w, ox + cbx0 * 8 + lx, oy + cby0 * 8 + ly, E);
				const double bx0s = E.sx0 * 8.0, bx1s = E.sx1 * 8.0, bx2s = E.sx2 * 8.0;
				const double by0s = E.sy0 * 8.0, by1s = E.sy1 * 8.0, by2s = E.sy2 * 8.0;
				// nearest depth the plane can take over a block, measured from the block's first pixel centre
				const float zslack = (float)(3.5 * (fabs(w.Az) + fabs(w.Bz))) + 2.0e-6f;
				double r0 = E.e0, r1 = E.e1, r2 = E.e2;
				uint32_t qn = 0;   // candidates waiting in this wave's queue
				uint32_t wrote = 0; // bit b set: block b received a fragment from this sub-triangle
				for (int cby = cby0; cby <= cby1; cby++) {
					double e0 = r0, e1 = r1, e2 = r2;
					for (int cbx = cbx0; cbx <= cbx1; cbx++) {
						const double f0 = e0, f1 = e1, f2 = e2;
						e0 += bx0s; e1 += bx1s; e2 += bx2s;
						const int b = cby * 4 + cbx;
						// block-level depth rejection (uniform across the wave)
						const float zmid = (float)fma(w.Az, (double)(ox + cbx * 8) + 4.0, fma(w.Bz, (double)(oy + cby * 8) + 4.0, w.Cz));
						const float znear = zmid - zslack;
						if (znear > u2f(sBlockMax[b])) continue;
						const int px = ox + cbx * 8 + lx, py = oy + cby * 8 + ly;
						const bool in = px >= wx0 && px <= wx1 && py >= wy0 && py <= wy1 && edges_inside(f0, f1, f2);
						const int li = (py - oy) * VKR_TILE + (px - ox);
						bool cand = false;
						if (in) {
							const float z = vkr_depth_at(w.Az, w.Bz, w.Cz, px, py);
							cand = (z >= 0.0f && z <= 1.0f) && !(f2u(z) > (uint32_t)(tb64[li] >> 32));
						}
						const unsigned long long cm = __ballot(cand);
						if (cm == 0) continue;
						wrote |= 1u << b;
						if (cand) wq[qn + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull))] = (uint16_t)li;
						qn += (uint32_t)__popcll(cm);
						if (qn >= 64) {
							const int qi = (int)wq[lane];
							const int qx = ox + (qi % VKR_TILE), qy = oy + (qi / VKR_TILE);
							if (alpha_pass(a, pl, t0, t1, t2, tex, qx, qy)) fragment<RASTER_VIS>(tb64, tb32, qi, w, qx, qy);
							const uint16_t carry = wq[64 + lane];
							qn -= 64;
							if (lane < qn) wq[lane] = carry;
						}
					}
					r0 += by0s; r1 += by1s; r2 += by2s;
				}
				if (lane < qn) {
					const int qi = (int)wq[lane];
					const int qx = ox + (qi % VKR_TILE), qy = oy + (qi / VKR_TILE);
					if (alpha_pass(a, pl, t0, t1, t2, tex, qx, qy)) fragment<RASTER_VIS>(tb64, tb32, qi, w, qx, qy);
				}
				// refresh the farthest depth of the blocks that may have changed
				while (wrote) {
					const int b = __ffs((int)wrote) - 1;
					wrote &= wrote - 1;
					const uint32_t m = block_max_depth(tb64, b & 3, b >> 2, lane);
					if (lane == 0) sBlockMax[b] = m;
				}
			}
		}
		__syncthreads(); // the stage is rewritten by the next chunk
	}
	__syncthreads();
	for (uint32_t i = tid; i < VKR_TILE_PIX; i += 256) {
		const int px = ox + (int)(i % VKR_TILE), py = oy + (int)(i / VKR_TILE);
		if (px > ex || py > ey) continue;
		atomicMin(&a.vis[(size_t)(py - (int)a.view.y0) * rw + (size_t)(px - (int)a.view.x0)], tb64[i]);
	}
}

// ---------------------------------------------------------------------------
// Overshading counter (debug view 6): stencil increments only for fragments
// that pass LESS_OR_EQUAL against the depth written so far, in submission
// order (main/main.cpp:2781-2797, 4122-4123).  Order matters, so each thread
// owns four pixels and walks the tile's bin, sorted by primitive, staged
// through LDS 256 sub-triangles at a time.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_overshade(OvershadeArgs a)
{
	__shared__ SubTri q[256];
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % a.view.tilesX, ty = tile / a.view.tilesX;
	const int ox = (int)(a.view.x0 + tx * VKR_TILE), oy = (int)(a.view.y0 + ty * VKR_TILE);
	const int ex = min(ox + VKR_TILE, (int)a.view.x1) - 1, ey = min(oy + VKR_TILE, (int)a.view.y1) - 1;
	const uint32_t tid = threadIdx.x;
	// thread -> pixels (x, y + 8*k), k = 0..3
	const int lx = (int)(tid & 31), ly = (int)(tid >> 5);
	float depth[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
	uint32_t cnt[4] = { 0, 0, 0, 0 };

	uint32_t beg = a.tileOffset[tile], end = a.tileOffset[tile + 1];
	if (end > a.binCap) end = a.binCap;
	if (beg > end) beg = end;
	const uint32_t n = end - beg;
	for (uint32_t base = 0; base < n; base += 256) {
		__syncthreads();
		if (base + tid < n) q[tid] = a.recs[a.binList[beg + base + tid]];
		__syncthreads();
		const uint32_t m = min(256u, n - base);
		for (uint32_t j = 0; j < m; j++) {
			const SubTri& s = q[j];
			int bx0, by0, bx1, by1;
			tri_bbox(s, bx0, by0, bx1, by1);
			const int px = ox + lx;
			if (px < bx0 || px > bx1) continue;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int py = oy + ly + 8 * k;
				if (py < by0 || py > by1 || px > ex || py > ey) continue;
				EdgeSetup E;
				edge_setup(s, px, py, E);
				if (!edges_inside(E.e0, E.e1, E.e2)) continue;
				float z = vkr_depth_at(s.Az, s.Bz, s.Cz, px, py);
				if (!(z >= 0.0f && z <= 1.0f)) continue;
				if (!(z <= depth[k])) continue;
				depth[k] = z;
				if (cnt[k] < 255u) cnt[k]++;
			}
		}
	}
	const uint32_t rw = a.view.x1 - a.view.x0;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int px = ox + lx, py = oy + ly + 8 * k;
		if (px > ex || py > ey) continue;
		const size_t p = (size_t)(py - (int)a.view.y0) * rw + (size_t)(px - (int)a.view.x0);
		a.countOut[p] = (uint8_t)cnt[k];
		if (a.depthOut) a.depthOut[p] = depth[k];
	}
}

void vkr_launch_raster(hipStream_t s, int mode, const RasterArgs& a)
{
	const uint32_t nTiles = a.view.tilesX * a.view.tilesY;
	if (!nTiles) return;
	if (mode == RASTER_VIS) {
		hipLaunchKernelGGL(k_raster<RASTER_VIS>, dim3(nTiles), dim3(256), 0, s, a);
		if (a.alphaPass) hipLaunchKernelGGL(k_raster_alpha, dim3(nTiles * VKR_ALPHA_SLICES), dim3(256), 0, s, a);
	}
	else if (mode == RASTER_SHADOW) hipLaunchKernelGGL(k_raster<RASTER_SHADOW>, dim3(nTiles), dim3(256), 0, s, a);
	else hipLaunchKernelGGL(k_raster<RASTER_COUNT>, dim3(nTiles), dim3(256), 0, s, a);
}

void vkr_launch_overshade(hipStream_t s, const OvershadeArgs& a)
{
	const uint32_t nTiles = a.view.tilesX * a.view.tilesY;
	if (!nTiles) return;
	hipLaunchKernelGGL(k_overshade, dim3(nTiles), dim3(256), 0, s, a);
}
