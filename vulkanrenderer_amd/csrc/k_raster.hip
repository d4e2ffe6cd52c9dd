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

struct EdgeSetup {
	int64_t e0, e1, e2;    // biased edge values at the start pixel
	int64_t sx0, sx1, sx2; // per +1 pixel in x
	int64_t sy0, sy1, sy2; // per +1 pixel in y
};

// Edge i runs from vertex i to vertex i+1; E = dx*(PY - Y) - dy*(PX - X) >= 0
// inside (orientation positive); a pixel exactly on an edge is covered only
// if that edge is a top edge (dy == 0, dx > 0) or a left edge (dy < 0).
__device__ __forceinline__ void edge_setup(const SubTri& s, int px, int py, EdgeSetup& E)
{
	int64_t PX = (int64_t)px * 256 + 128, PY = (int64_t)py * 256 + 128;
	int64_t dx0 = (int64_t)s.X1 - s.X0, dy0 = (int64_t)s.Y1 - s.Y0;
	int64_t dx1 = (int64_t)s.X2 - s.X1, dy1 = (int64_t)s.Y2 - s.Y1;
	int64_t dx2 = (int64_t)s.X0 - s.X2, dy2 = (int64_t)s.Y0 - s.Y2;
	bool tl0 = (dy0 == 0 && dx0 > 0) || (dy0 < 0);
	bool tl1 = (dy1 == 0 && dx1 > 0) || (dy1 < 0);
	bool tl2 = (dy2 == 0 && dx2 > 0) || (dy2 < 0);
	E.e0 = dx0 * (PY - s.Y0) - dy0 * (PX - s.X0) - (tl0 ? 0 : 1);
	E.e1 = dx1 * (PY - s.Y1) - dy1 * (PX - s.X1) - (tl1 ? 0 : 1);
	E.e2 = dx2 * (PY - s.Y2) - dy2 * (PX - s.X2) - (tl2 ? 0 : 1);
	E.sx0 = -dy0 * 256; E.sx1 = -dy1 * 256; E.sx2 = -dy2 * 256;
	E.sy0 = dx0 * 256; E.sy1 = dx1 * 256; E.sy2 = dx2 * 256;
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
	float b[3], uv[2], d[4], al[4];
	vkr_bary_at(pl, px, py, b);
	vkr_uv_derivs(pl, px, py, b, t0, t1, t2, uv, d);
	vkr_trilinear(a.sc, tex, uv[0], uv[1], d, al);
	return !(al[3] < 0.5f);
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

template <int MODE>
__global__ __launch_bounds__(256) void k_raster(RasterArgs a)
{
	__shared__ unsigned long long tb64[MODE == RASTER_VIS ? VKR_TILE_PIX : 1];
	__shared__ uint32_t tb32[MODE == RASTER_VIS ? 1 : VKR_TILE_PIX];
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
	__shared__ int sAnyAlpha;
	if (tid == 0) sAnyAlpha = 0;
	__syncthreads();

	uint32_t beg = a.tileOffset[tile], end = a.tileOffset[tile + 1];
	if (end > a.binCap) end = a.binCap;
	if (beg > end) beg = end;
	const uint32_t n = end - beg;

	// Visibility passes run twice over the bin: first every sub-triangle that
	// needs no alpha test, then the alpha-tested ones, whose (expensive)
	// texture lookup is skipped for fragments already behind the tile's depth.
	const int nPass = (MODE == RASTER_VIS) ? 2 : 1;
	for (int pass = 0; pass < nPass; pass++) {
	if (pass == 1) {
		__syncthreads();
		if (!sAnyAlpha) break;
	}
	for (uint32_t base = 0; base < n; base += 256) {
		const uint32_t i = base + tid;
		bool has = i < n;
		SubTri s;
		int bx0 = 0, by0 = 0, bx1 = -1, by1 = -1;
		if (has) {
			const uint32_t ri = a.binList[beg + i];
			if (MODE == RASTER_VIS) {
				const bool isAlpha = (a.recs[ri].flags & VKR_ST_ALPHA) != 0;
				if (pass == 0 && isAlpha) sAnyAlpha = 1;
				if (isAlpha != (pass == 1)) has = false;
			}
			if (has) {
				s = a.recs[ri];
				tri_bbox(s, bx0, by0, bx1, by1);
				bx0 = max(bx0, ox); by0 = max(by0, oy); bx1 = min(bx1, ex); by1 = min(by1, ey);
			}
		}
		if (!has) {
			s.X0 = s.Y0 = s.X1 = s.Y1 = s.X2 = s.Y2 = 0; s.prim = 0; s.flags = 0; s.Az = s.Bz = s.Cz = s.bias = 0.0;
		}
		const bool nonEmpty = has && bx0 <= bx1 && by0 <= by1;
		const int area = nonEmpty ? (bx1 - bx0 + 1) * (by1 - by0 + 1) : 0;
		const bool alpha = (MODE == RASTER_VIS) && (s.flags & VKR_ST_ALPHA);
		const bool big = nonEmpty && (area > BIG_AREA || alpha);

		if (nonEmpty && !big) {
			EdgeSetup E;
			edge_setup(s, bx0, by0, E);
			for (int py = by0; py <= by1; py++) {
				int64_t e0 = E.e0, e1 = E.e1, e2 = E.e2;
				for (int px = bx0; px <= bx1; px++) {
					if ((e0 | e1 | e2) >= 0) fragment<MODE>(tb64, tb32, (py - oy) * VKR_TILE + (px - ox), s, px, py);
					e0 += E.sx0; e1 += E.sx1; e2 += E.sx2;
				}
				E.e0 += E.sy0; E.e1 += E.sy1; E.e2 += E.sy2;
			}
		}

		// wave-cooperative pass over the big / alpha-tested sub-triangles of this wave
		unsigned long long mask = __ballot(big);
		while (mask) {
			const int src = __ffsll((long long)mask) - 1;
			mask &= mask - 1;
			SubTri w;
			w.X0 = __shfl(s.X0, src, 64); w.Y0 = __shfl(s.Y0, src, 64);
			w.X1 = __shfl(s.X1, src, 64); w.Y1 = __shfl(s.Y1, src, 64);
			w.X2 = __shfl(s.X2, src, 64); w.Y2 = __shfl(s.Y2, src, 64);
			w.prim = __shfl(s.prim, src, 64); w.flags = __shfl(s.flags, src, 64);
			w.Az = __shfl(s.Az, src, 64); w.Bz = __shfl(s.Bz, src, 64); w.Cz = __shfl(s.Cz, src, 64);
			w.bias = __shfl(s.bias, src, 64);
			const int wx0 = __shfl(bx0, src, 64), wy0 = __shfl(by0, src, 64);
			const int wx1 = __shfl(bx1, src, 64), wy1 = __shfl(by1, src, 64);
			const bool walpha = (MODE == RASTER_VIS) && (w.flags & VKR_ST_ALPHA);
			TriPlanes pl;
			float2 t0 = make_float2(0, 0), t1 = t0, t2 = t0;
			uint32_t alphaTex = 0;
			bool plOk = true;
			if (walpha) {
				// alpha-tested passes submit in order A, so prim indexes sc.tri directly
				const uint32_t i0 = a.sc.tri[3 * (size_t)w.prim + 0], i1 = a.sc.tri[3 * (size_t)w.prim + 1],
				               i2 = a.sc.tri[3 * (size_t)w.prim + 2];
				plOk = vkr_tri_planes(a.clip[i0], a.clip[i1], a.clip[i2], a.view.W, a.view.H, pl);
				t0 = make_float2(a.sc.uv[2 * (size_t)i0], a.sc.uv[2 * (size_t)i0 + 1]);
				t1 = make_float2(a.sc.uv[2 * (size_t)i1], a.sc.uv[2 * (size_t)i1 + 1]);
				t2 = make_float2(a.sc.uv[2 * (size_t)i2], a.sc.uv[2 * (size_t)i2 + 1]);
				alphaTex = a.sc.mat[a.sc.triMat[w.prim]].tex[3];
			}
			const int lx = (int)(lane & 7), ly = (int)(lane >> 3);
			for (int yy = wy0; yy <= wy1; yy += 8)
				for (int xx = wx0; xx <= wx1; xx += 8) {
					const int px = xx + lx, py = yy + ly;
					if (px > wx1 || py > wy1) continue;
					EdgeSetup E;
					edge_setup(w, px, py, E);
					if ((E.e0 | E.e1 | E.e2) < 0) continue;
					const int li = (py - oy) * VKR_TILE + (px - ox);
					if (walpha) {
						if (!plOk) continue;
						// a fragment behind the depth already in the tile cannot win the atomicMin
						const float z = vkr_depth_at(w.Az, w.Bz, w.Cz, px, py);
						if (!(z >= 0.0f && z <= 1.0f)) continue;
						if (f2u(z) > (uint32_t)(tb64[li] >> 32)) continue;
						if (!alpha_pass(a, pl, t0, t1, t2, a.sc.tex[alphaTex], px, py)) continue;
					}
					fragment<MODE>(tb64, tb32, li, w, px, py);
				}
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
				if ((E.e0 | E.e1 | E.e2) < 0) continue;
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
	if (mode == RASTER_VIS) hipLaunchKernelGGL(k_raster<RASTER_VIS>, dim3(nTiles), dim3(256), 0, s, a);
	else if (mode == RASTER_SHADOW) hipLaunchKernelGGL(k_raster<RASTER_SHADOW>, dim3(nTiles), dim3(256), 0, s, a);
	else hipLaunchKernelGGL(k_raster<RASTER_COUNT>, dim3(nTiles), dim3(256), 0, s, a);
}

void vkr_launch_overshade(hipStream_t s, const OvershadeArgs& a)
{
	const uint32_t nTiles = a.view.tilesX * a.view.tilesY;
	if (!nTiles) return;
	hipLaunchKernelGGL(k_overshade, dim3(nTiles), dim3(256), 0, s, a);
}
