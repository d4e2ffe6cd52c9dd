// k_geometry.hip — vertex transform (inside triangle setup), triangle setup (clip,
// cull, snap, depth plane), and screen-tile binning (count -> scan -> fill).
//
// Replaces, for every draw of a pass, the vertex shader's
// gl_Position = projCam * vec4(iPosition,1) (default.vert:85, gBufWrite.vert:70,
// debug.vert:17, overVisWrite.vert:12, shadowOffscreen.vert:10) and the
// fixed-function primitive assembly / clipping / culling / viewport stages
// configured at main/main.cpp:2274-2305 (and :3105-3115 for the shadow pass).
// Compiled with -ffp-contract=off: the arithmetic order defines coverage.
#include "vkr_kernels.h"

// ---------------------------------------------------------------------------
struct ClipV { double x, y, z, w; };

__device__ __forceinline__ double plane_dist(const ClipV& v, int plane, double VKR_GUARD)
{
	switch (plane) {
	case 0: return v.z;
	case 1: return v.x + VKR_GUARD * v.w;
	case 2: return VKR_GUARD * v.w - v.x;
	case 3: return v.y + VKR_GUARD * v.w;
	default: return VKR_GUARD * v.w - v.y;
	}
}

// Sutherland-Hodgman against one plane; new vertices are always interpolated
// from the inside vertex towards the outside one.
__device__ __noinline__ int clip_poly(ClipV* v, int n, int plane, double guard)
{
	double d[10];
	bool anyOut = false;
	for (int i = 0; i < n; i++) {
		d[i] = plane_dist(v[i], plane, guard);
		if (!(d[i] >= 0.0)) anyOut = true;
	}
	if (!anyOut) return n;
	ClipV o[10];
	int m = 0;
	for (int i = 0; i < n; i++) {
		int j = (i + 1 == n) ? 0 : i + 1;
		bool ini = d[i] >= 0.0, inj = d[j] >= 0.0;
		if (ini) o[m++] = v[i];
		if (ini != inj) {
			const ClipV A = ini ? v[i] : v[j];
			const ClipV B = ini ? v[j] : v[i];
			double dA = ini ? d[i] : d[j], dB = ini ? d[j] : d[i];
			double t = dA / (dA - dB);
			ClipV p;
			p.x = A.x + t * (B.x - A.x);
			p.y = A.y + t * (B.y - A.y);
			p.z = A.z + t * (B.z - A.z);
			p.w = A.w + t * (B.w - A.w);
			o[m++] = p;
		}
	}
	for (int i = 0; i < m; i++) v[i] = o[i];
	return m;
}

__device__ __forceinline__ void snap_vertex(double x, double y, double w, double hW, double hH, int32_t& X, int32_t& Y)
{
	double iw = 1.0 / w;
	double sx = (x * iw + 1.0) * hW;
	double sy = (y * iw + 1.0) * hH;
	X = (int32_t)(long long)floor(sx * (double)VKR_SUBPIX + 0.5);
	Y = (int32_t)(long long)floor(sy * (double)VKR_SUBPIX + 0.5);
}

// Finishes one fan triangle: facing, cull, orientation, bounding box, bias.
// Returns false if nothing is to be drawn; otherwise fills the record and its
// tile range.
__device__ __forceinline__ bool prepare_subtri(const SetupArgs& a, int32_t X0, int32_t Y0, int32_t X1, int32_t Y1, int32_t X2,
                                               int32_t Y2, const TriPlanes& pl, uint32_t prim, uint32_t flags, int cull,
                                               SubTri& r, uint2& tiles)
{
	int64_t ax = (int64_t)X1 - X0, ay = (int64_t)Y1 - Y0, bx = (int64_t)X2 - X0, by = (int64_t)Y2 - Y0;
	int64_t area2 = ax * by - bx * ay;
	if (area2 == 0) return false;
	bool front = area2 < 0; // Vulkan: a = -area2/2 > 0, frontFace = COUNTER_CLOCKWISE
	if (cull == 1 && !front) return false;
	if (area2 < 0) { int32_t t = X1; X1 = X2; X2 = t; t = Y1; Y1 = Y2; Y2 = t; }
	int32_t xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
	int32_t ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
	// pixel centres at p*256+128
	int32_t px0 = (xmin - 128 + 255) >> 8, px1 = (xmax - 128) >> 8;
	int32_t py0 = (ymin - 128 + 255) >> 8, py1 = (ymax - 128) >> 8;
	px0 = max(px0, (int32_t)a.view.x0); py0 = max(py0, (int32_t)a.view.y0);
	px1 = min(px1, (int32_t)a.view.x1 - 1); py1 = min(py1, (int32_t)a.view.y1 - 1);
	if (px0 > px1 || py0 > py1) return false;

	r.X0 = X0; r.Y0 = Y0; r.X1 = X1; r.Y1 = Y1; r.X2 = X2; r.Y2 = Y2;
	r.prim = prim;
	r.flags = flags | (front ? VKR_ST_FRONT : 0u);
	r.Az = pl.Az; r.Bz = pl.Bz; r.Cz = pl.Cz;
	r.bias = 0.0;
	if (a.shadow) {
		// o = m * slopeFactor + r * constantFactor (main/main.cpp:3112-3114)
		double m = fabs(pl.Az) > fabs(pl.Bz) ? fabs(pl.Az) : fabs(pl.Bz);
		double zmax = 0.0;
		double zv = fma(pl.Az, (double)X0 / 256.0, fma(pl.Bz, (double)Y0 / 256.0, pl.Cz)); if (zv > zmax) zmax = zv;
		zv = fma(pl.Az, (double)X1 / 256.0, fma(pl.Bz, (double)Y1 / 256.0, pl.Cz)); if (zv > zmax) zmax = zv;
		zv = fma(pl.Az, (double)X2 / 256.0, fma(pl.Bz, (double)Y2 / 256.0, pl.Cz)); if (zv > zmax) zmax = zv;
		if (zmax > 1.0) zmax = 1.0;
		float zmf = (float)zmax;
		int e = (int)((f2u(zmf) >> 23) & 0xffu) - 127;
		if (e < -126) e = -126;
		double rr = __longlong_as_double((long long)(e - 23 + 1023) << 52);
		r.bias = m * 10.0 + rr * 7.0;
	}
	uint32_t tx0 = (uint32_t)(px0 - (int32_t)a.view.x0) / VKR_TILE, tx1 = (uint32_t)(px1 - (int32_t)a.view.x0) / VKR_TILE;
	uint32_t ly0, ly1; // the rows of this context that the bounding box reaches (a stripe partition may own none of them)
	if (!vkr_view_local_rows(a.view, py0, py1, ly0, ly1)) return false;
	uint32_t ty0 = ly0 / VKR_TILE, ty1 = ly1 / VKR_TILE;
	tiles = make_uint2(tx0 | (ty0 << 16), tx1 | (ty1 << 16));
	return true;
}

// Tile counting with the position kept: a record that touches at most 8 tiles
// remembers, per tile, the value the counter had (its index inside the tile's
// bin), so the fill pass needs no second round of atomics for it.  Records that
// touch more tiles are counted separately and take their positions, behind the
// others, from a cursor in the fill pass.
__device__ __forceinline__ void count_small(const SetupArgs& a, uint32_t slot, uint2 tiles, bool alpha)
{
	uint32_t* counter = alpha ? a.tileCountA : a.tileCount;
	const uint32_t tx0 = tiles.x & 0xffffu, ty0 = tiles.x >> 16, tx1 = tiles.y & 0xffffu, ty1 = tiles.y >> 16;
	uint32_t k = 0;
	for (uint32_t ty = ty0; ty <= ty1; ty++)
		for (uint32_t tx = tx0; tx <= tx1; tx++, k++) a.recLocal[(size_t)slot * 8 + k] = atomicAdd(&counter[ty * a.view.tilesX + tx], 1u);
}

__device__ __forceinline__ bool commit_subtri(const SetupArgs& a, uint32_t slot, const SubTri& r, uint2 tiles)
{
	if (slot >= a.recCap) { atomicOr(a.overflow, 1u); return false; }
	a.recs[slot] = r;
	a.recTiles[slot] = make_uint2(tiles.x, tiles.y | ((r.flags & VKR_ST_ALPHA) ? 0x80000000u : 0u));
	const uint32_t tx0 = tiles.x & 0xffffu, ty0 = tiles.x >> 16, tx1 = tiles.y & 0xffffu, ty1 = tiles.y >> 16;
	if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) <= 8) { count_small(a, slot, tiles, (r.flags & VKR_ST_ALPHA) != 0); return true; }
	a.bigQueue[atomicAdd(a.bigCount, 1u)] = slot; // counted by k_count_big
	return true;
}

// fan triangles of a clipped polygon: slots after the one-per-triangle region
__device__ __forceinline__ void emit_subtri(const SetupArgs& a, int32_t X0, int32_t Y0, int32_t X1, int32_t Y1, int32_t X2,
                                            int32_t Y2, const TriPlanes& pl, uint32_t prim, uint32_t flags, int cull)
{
	SubTri r; uint2 tiles;
	if (!prepare_subtri(a, X0, Y0, X1, Y1, X2, Y2, pl, prim, flags, cull, r, tiles)) return;
	commit_subtri(a, a.count + atomicAdd(a.recCount, 1u), r, tiles);
}

__device__ __noinline__ void setup_clipped(const SetupArgs& a, float4 c0, float4 c1, float4 c2, const TriPlanes& pl,
                                           uint32_t prim, uint32_t flags, int cull, double hW, double hH)
{
	ClipV v[10];
	v[0].x = c0.x; v[0].y = c0.y; v[0].z = c0.z; v[0].w = c0.w;
	v[1].x = c1.x; v[1].y = c1.y; v[1].z = c1.z; v[1].w = c1.w;
	v[2].x = c2.x; v[2].y = c2.y; v[2].z = c2.z; v[2].w = c2.w;
	int n = 3;
	for (int p = 0; p < 5 && n >= 3; p++) n = clip_poly(v, n, p, a.guard);
	if (n < 3) return;
	int32_t X[10], Y[10];
	for (int i = 0; i < n; i++) snap_vertex(v[i].x, v[i].y, v[i].w, hW, hH, X[i], Y[i]);
	for (int i = 1; i + 1 < n; i++) emit_subtri(a, X[0], Y[0], X[i], Y[i], X[i + 1], Y[i + 1], pl, prim, flags, cull);
}

#define SETUP_TABLE 512u // slots of k_setup's per-workgroup tile table

// What both setup kernels do first for triangle t: load, reject, planes, pass state.
struct TriFront {
	float4 c0, c1, c2;
	TriPlanes pl;
	uint32_t prim, flags;
	int cull;
	bool needClip;
};

__device__ __forceinline__ bool tri_front(const SetupArgs& a, uint32_t t, TriFront& f)
{
	const uint32_t i0 = a.sc.tri[3 * (size_t)t + 0], i1 = a.sc.tri[3 * (size_t)t + 1], i2 = a.sc.tri[3 * (size_t)t + 2];
	// the vertex stage: every triangle transforms its own three vertices (the same arithmetic per vertex whoever does it)
	const float4 c0 = vkr_xform(a.M, a.sc.pos[3 * (size_t)i0], a.sc.pos[3 * (size_t)i0 + 1], a.sc.pos[3 * (size_t)i0 + 2]);
	const float4 c1 = vkr_xform(a.M, a.sc.pos[3 * (size_t)i1], a.sc.pos[3 * (size_t)i1 + 1], a.sc.pos[3 * (size_t)i1 + 2]);
	const float4 c2 = vkr_xform(a.M, a.sc.pos[3 * (size_t)i2], a.sc.pos[3 * (size_t)i2 + 1], a.sc.pos[3 * (size_t)i2 + 2]);
	f.c0 = c0; f.c1 = c1; f.c2 = c2;
	bool ok = true;
	// NaN anywhere -> drop
	if (!(c0.x == c0.x) || !(c0.y == c0.y) || !(c0.z == c0.z) || !(c0.w == c0.w)) ok = false;
	if (!(c1.x == c1.x) || !(c1.y == c1.y) || !(c1.z == c1.z) || !(c1.w == c1.w)) ok = false;
	if (!(c2.x == c2.x) || !(c2.y == c2.y) || !(c2.z == c2.z) || !(c2.w == c2.w)) ok = false;
	// trivial reject against -w<=x,y<=w, 0<=z<=w
	if ((c0.x < -c0.w) & (c1.x < -c1.w) & (c2.x < -c2.w)) ok = false;
	if ((c0.x > c0.w) & (c1.x > c1.w) & (c2.x > c2.w)) ok = false;
	if ((c0.y < -c0.w) & (c1.y < -c1.w) & (c2.y < -c2.w)) ok = false;
	if ((c0.y > c0.w) & (c1.y > c1.w) & (c2.y > c2.w)) ok = false;
	if ((c0.z < 0.0f) & (c1.z < 0.0f) & (c2.z < 0.0f)) ok = false;
	if ((c0.z > c0.w) & (c1.z > c1.w) & (c2.z > c2.w)) ok = false;
	if (!ok) return false;
	if (!vkr_tri_planes(c0, c1, c2, a.view.W, a.view.H, f.pl)) return false;
	const bool isAlphaMesh = t >= a.sc.nTriOpaque;
	f.cull = isAlphaMesh ? a.cullAlpha : a.cullOpaque;
	f.prim = a.useOrderB ? a.sc.orderB[t] : t;
	f.flags = 0;
	if (a.alphaTest && a.sc.mat[a.sc.triMat[t]].alphaTest) f.flags |= VKR_ST_ALPHA;
	// does any vertex violate the near or a guard-band plane?
	bool needClip = false;
	const double G = a.guard;
	const float4 c[3] = { c0, c1, c2 };
#pragma unroll
	for (int i = 0; i < 3; i++) {
		double x = c[i].x, y = c[i].y, z = c[i].z, w = c[i].w;
		if (!(z >= 0.0)) needClip = true;
		if (!(x + G * w >= 0.0)) needClip = true;
		if (!(G * w - x >= 0.0)) needClip = true;
		if (!(y + G * w >= 0.0)) needClip = true;
		if (!(G * w - y >= 0.0)) needClip = true;
	}
	f.needClip = needClip;
	return true;
}

// One thread per triangle.  Unclipped triangles (the common case) write the
// record slot of their own index; triangles that cross the
// near plane or the guard band are queued for k_setup_clip, which keeps the
// polygon clipper's local arrays (and their scratch memory) out of this kernel.
#ifdef VKR_WG_TIMES
// measurement build: per-workgroup phase clocks (100 MHz) of the last k_setup launch
__device__ unsigned long long g_setupTimes[8 * 4096];
extern "C" __attribute__((visibility("default"))) int vkr_debug_setup_times(void* dst, size_t bytes)
{
	return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_setupTimes), bytes, 0, hipMemcpyDeviceToHost);
}
#define SETUP_CLOCK(k) do { if (threadIdx.x == 0) sClk[k] = wall_clock64(); } while (0)
#else
#define SETUP_CLOCK(k) do { } while (0)
#endif

// At most 80 scalar registers: a wavefront is given its count rounded up to 16 plus 16 more out of 800 per SIMD, so the 106 the
// compiler would take leave room for six wavefronts per SIMD where the vector registers allow eight (the overflow is
// kept in lanes of a vector register).
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_setup(SetupArgs a)
{
#ifdef VKR_WG_TIMES
	__shared__ unsigned long long sClk[6]; // in LDS, so that the measurement build keeps the registers of the real one
	SETUP_CLOCK(0);
#endif
	__shared__ uint32_t sTag[SETUP_TABLE], sCnt[SETUP_TABLE], sBase[SETUP_TABLE];
	__shared__ uint4 sStage[4][192]; // per wavefront: 32 triangles' planes or records on their way out
	// the next pass's counters (the other set of the pair; nothing in flight uses it) are zeroed here, so that no pass
	// needs a fill launch of its own
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.zeroWords; i += gridDim.x * blockDim.x) a.zeroBase[i] = 0u;
	for (uint32_t h = threadIdx.x; h < SETUP_TABLE; h += blockDim.x) { sTag[h] = 0xffffffffu; sCnt[h] = 0u; }
	__syncthreads();
	const uint32_t t = a.first + blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63;
	bool emit = false, queue = false;
	SubTri rec; uint2 tiles = make_uint2(0, 0);
	TriPlanes rec_planes;
	if (t < a.first + a.count) {
		TriFront f;
		if (tri_front(a, t, f)) {
			rec_planes = f.pl;
			if (f.needClip) {
				queue = true;
			} else {
				const double hW = (double)a.view.W * 0.5, hH = (double)a.view.H * 0.5;
				int32_t X0, Y0, X1, Y1, X2, Y2;
				snap_vertex(f.c0.x, f.c0.y, f.c0.w, hW, hH, X0, Y0);
				snap_vertex(f.c1.x, f.c1.y, f.c1.w, hW, hH, X1, Y1);
				snap_vertex(f.c2.x, f.c2.y, f.c2.w, hW, hH, X2, Y2);
				emit = prepare_subtri(a, X0, Y0, X1, Y1, X2, Y2, f.pl, f.prim, f.flags, f.cull, rec, tiles);
			}
		}
	}
	SETUP_CLOCK(1);
	const unsigned long long qmask = __ballot(queue);
	if (qmask) {
		const int leader = __ffsll((long long)qmask) - 1;
		uint32_t base = 0;
		if ((int)lane == leader) base = atomicAdd(a.clipCount, (uint32_t)__popcll(qmask));
		base = __shfl(base, leader, 64);
		if (queue) a.clipQueue[base + (uint32_t)__popcll(qmask & ((1ull << lane) - 1ull))] = t;
	}
	// Record slot = triangle index inside the pass (no allocation: a counter
	// every wave has to bump serialises in L2); slots of triangles that draw
	// nothing carry an empty tile range.
	const uint32_t slot = t - a.first;
	if (t < a.first + a.count) {
		// bit 31 of the second word: alpha-tested (bin fill then needs nothing else from the record)
		// the interpolation planes: once per drawn triangle instead of once per shaded pixel
		if (emit) a.recTiles[slot] = make_uint2(tiles.x, tiles.y | ((rec.flags & VKR_ST_ALPHA) ? 0x80000000u : 0u));
		else a.recTiles[slot] = make_uint2(0xffffffffu, 0u);
	}
	// The planes (96 B) and the record (64 B) of a wavefront's 64 triangles are neighbours in memory.  Stored lane by
	// lane, every store instruction would scatter 64 separate 16-byte pieces over 4-6 KB (ten instructions, 640
	// requests at the L2 per wavefront); staged through LDS half a wavefront at a time they leave as whole kilobytes
	// (the same ten instructions, one request per line).  Pieces of triangles that draw nothing are not written.
	{
		const unsigned long long emask = __ballot(emit);
		if (emask) {
			uint4* stage = sStage[threadIdx.x >> 6];
			const uint32_t l32 = lane & 31u;
			uint4* dstP = (uint4*)(a.planes + (t - lane));
			uint4* dstR = (uint4*)(a.recs + (slot - lane));
			static_assert(sizeof(TriPlanes) == 96 && sizeof(SubTri) == 64, "six and four 16-byte pieces");
			const double pd[12] = { rec_planes.a[0], rec_planes.a[1], rec_planes.a[2], rec_planes.b[0], rec_planes.b[1], rec_planes.b[2],
			                        rec_planes.c[0], rec_planes.c[1], rec_planes.c[2], rec_planes.Az, rec_planes.Bz, rec_planes.Cz };
#pragma unroll
			for (uint32_t half = 0; half < 2; half++) {
				if (!((emask >> (32u * half)) & 0xffffffffull)) continue; // wave-uniform
				if (emit && (lane >> 5) == half) {
#pragma unroll
					for (int j = 0; j < 6; j++)
						stage[l32 * 6u + j] = make_uint4((uint32_t)__double2loint(pd[2 * j]), (uint32_t)__double2hiint(pd[2 * j]),
						                                 (uint32_t)__double2loint(pd[2 * j + 1]), (uint32_t)__double2hiint(pd[2 * j + 1]));
				}
				__builtin_amdgcn_wave_barrier();
#pragma unroll
				for (uint32_t i = 0; i < 3; i++) {
					const uint32_t c = i * 64u + lane, owner = 32u * half + c / 6u;
					if ((emask >> owner) & 1ull) dstP[192u * half + c] = stage[c];
				}
				__builtin_amdgcn_wave_barrier();
				if (emit && (lane >> 5) == half) {
					stage[l32 * 4u + 0] = make_uint4((uint32_t)rec.X0, (uint32_t)rec.Y0, (uint32_t)rec.X1, (uint32_t)rec.Y1);
					stage[l32 * 4u + 1] = make_uint4((uint32_t)rec.X2, (uint32_t)rec.Y2, rec.prim, rec.flags);
					stage[l32 * 4u + 2] = make_uint4((uint32_t)__double2loint(rec.Az), (uint32_t)__double2hiint(rec.Az),
					                                 (uint32_t)__double2loint(rec.Bz), (uint32_t)__double2hiint(rec.Bz));
					stage[l32 * 4u + 3] = make_uint4((uint32_t)__double2loint(rec.Cz), (uint32_t)__double2hiint(rec.Cz),
					                                 (uint32_t)__double2loint(rec.bias), (uint32_t)__double2hiint(rec.bias));
				}
				__builtin_amdgcn_wave_barrier();
#pragma unroll
				for (uint32_t i = 0; i < 2; i++) {
					const uint32_t c = i * 64u + lane, owner = 32u * half + c / 4u;
					if ((emask >> owner) & 1ull) dstR[128u * half + c] = stage[c];
				}
				__builtin_amdgcn_wave_barrier();
			}
		}
	}
	SETUP_CLOCK(2);
	const uint32_t tx0 = tiles.x & 0xffffu, ty0 = tiles.x >> 16, tx1 = tiles.y & 0xffffu, ty1 = tiles.y >> 16;
	const uint32_t nt = emit ? (tx1 - tx0 + 1) * (ty1 - ty0 + 1) : 0u;
	// Records touching at most 8 tiles: the workgroup first counts per tile in an LDS table (its 256 consecutive
	// triangles are neighbours on screen, so they share few tiles), then bumps each tile's global counter once for
	// the whole group, and every record adds the group's base to its index inside the group.  An entry whose table
	// slot is taken by another tile goes to the global counter directly.  (~7x fewer global atomics, which run at
	// ~6 G/s.)
	const bool small = nt >= 1 && nt <= 8;
	const uint32_t abit = (emit && (rec.flags & VKR_ST_ALPHA)) ? 0x80000000u : 0u;
	// the record's index in each of its tiles (row-major over its tile rectangle), kept in registers until the
	// group's bases are known: bit 31 = index inside the group, to be completed
	uint32_t vloc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	if (small) {
		uint32_t tx = tx0, ty = ty0;
#pragma unroll
		for (uint32_t k = 0; k < 8; k++) {
			if (k < nt) {
				const uint32_t tile = ty * a.view.tilesX + tx, key = tile | abit;
				const uint32_t h = (tile ^ (tile >> 7) ^ (abit >> 23)) & (SETUP_TABLE - 1u);
				const uint32_t prev = atomicCAS(&sTag[h], 0xffffffffu, key);
				if (prev == 0xffffffffu || prev == key) vloc[k] = atomicAdd(&sCnt[h], 1u) | 0x80000000u;
				else vloc[k] = atomicAdd(&(abit ? a.tileCountA : a.tileCount)[tile], 1u);
				if (++tx > tx1) { tx = tx0; ty++; }
			}
		}
	}
	__syncthreads();
	SETUP_CLOCK(3);
	for (uint32_t h = threadIdx.x; h < SETUP_TABLE; h += blockDim.x) {
		const uint32_t key = sTag[h];
		if (key != 0xffffffffu) sBase[h] = atomicAdd(&((key & 0x80000000u) ? a.tileCountA : a.tileCount)[key & 0x7fffffffu], sCnt[h]);
	}
	__syncthreads();
	SETUP_CLOCK(4);
	if (small) {
		uint32_t tx = tx0, ty = ty0;
#pragma unroll
		for (uint32_t k = 0; k < 8; k++) {
			if (k < nt) {
				if (vloc[k] & 0x80000000u) {
					const uint32_t tile = ty * a.view.tilesX + tx;
					const uint32_t h = (tile ^ (tile >> 7) ^ (abit >> 23)) & (SETUP_TABLE - 1u);
					vloc[k] = sBase[h] + (vloc[k] & 0x7fffffffu);
				}
				if (++tx > tx1) { tx = tx0; ty++; }
			}
		}
		uint4* out = (uint4*)(a.recLocal + (size_t)slot * 8);
		out[0] = make_uint4(vloc[0], vloc[1], vloc[2], vloc[3]);
		if (nt > 4) out[1] = make_uint4(vloc[4], vloc[5], vloc[6], vloc[7]);
	}
	// records that touch many tiles: queued for k_count_big / the first workgroups of k_bin_fill (one wavefront per record; they come in
	// runs - the near-field triangles of a mesh - and would otherwise pile up in a few wavefronts)
	unsigned long long bmask = __ballot(nt > 8);
	if (bmask) {
		const int leader = __ffsll((long long)bmask) - 1;
		uint32_t base = 0;
		if ((int)lane == leader) base = atomicAdd(a.bigCount, (uint32_t)__popcll(bmask));
		base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
		if (nt > 8) a.bigQueue[base + (uint32_t)__popcll(bmask & ((1ull << lane) - 1ull))] = slot;
	}
#ifdef VKR_WG_TIMES
	{
		const uint32_t ne = (uint32_t)__popcll(__ballot(emit));
		__shared__ uint32_t sEmit;
		if (threadIdx.x == 0) sEmit = 0;
		__syncthreads();
		if (lane == 0) atomicAdd(&sEmit, ne);
		__syncthreads();
		if (threadIdx.x == 0 && blockIdx.x < 4096u) {
			sClk[5] = wall_clock64();
			for (int k = 0; k < 6; k++) g_setupTimes[8 * blockIdx.x + k] = sClk[k];
			g_setupTimes[8 * blockIdx.x + 6] = sEmit;
			uint32_t used = 0;
			for (uint32_t h = 0; h < SETUP_TABLE; h++) used += sTag[h] != 0xffffffffu;
			g_setupTimes[8 * blockIdx.x + 7] = used;
		}
	}
#endif
}

// The queued triangles: polygon clip, fan, one record per fan triangle.
__global__ __launch_bounds__(64) void k_setup_clip(SetupArgs a)
{
	const uint32_t n = *a.clipCount;
	for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
		const uint32_t t = a.clipQueue[q];
		TriFront f;
		if (!tri_front(a, t, f)) continue;
		a.planes[t] = f.pl;
		const double hW = (double)a.view.W * 0.5, hH = (double)a.view.H * 0.5;
		setup_clipped(a, f.c0, f.c1, f.c2, f.pl, f.prim, f.flags, f.cull, hW, hH);
	}
}

// Counts the queued many-tile records into tileCountBig, one wavefront per record.
__global__ __launch_bounds__(256) void k_count_big(SetupArgs a)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6;
	const uint32_t n = min(*a.bigCount, a.recCap);
	for (uint32_t q = wave; q < n; q += nWaves) {
		const uint2 tr = a.recTiles[a.bigQueue[q]];
		const uint32_t wx0 = tr.x & 0xffffu, wy0 = tr.x >> 16, wx1 = tr.y & 0xffffu, wy1 = (tr.y >> 16) & 0x7fffu;
		const uint32_t w = wx1 - wx0 + 1, cnt = w * (wy1 - wy0 + 1);
		for (uint32_t k = lane; k < cnt; k += 64) atomicAdd(&a.tileCountBig[(wy0 + k / w) * a.view.tilesX + (wx0 + k % w)], 1u);
	}
}

// ---------------------------------------------------------------------------
// exclusive scan of the per-tile counts (one workgroup), cursor reset
// ---------------------------------------------------------------------------
// bin-length class of a tile for the dispatch order: 0 = the longest bins
__device__ __forceinline__ uint32_t tile_class(uint32_t n) { return n >= 1024u ? 0u : n >= 256u ? 1u : n >= 64u ? 2u : 3u; }

template <bool ORDER> // ORDER: also list the tiles by descending bin length (tileOrder); the plain scan pays nothing for it
__global__ __launch_bounds__(1024) void k_scan_tiles(const uint32_t* __restrict__ count, const uint32_t* __restrict__ countA,
                                                      const uint32_t* __restrict__ countBig,
                                                      uint32_t* __restrict__ offset, uint32_t* __restrict__ cursor, uint32_t nTiles,
                                                      uint32_t cap, uint32_t* overflow, uint32_t* statusOut, uint32_t* __restrict__ tileOrder)
{
	// 8192 tiles per round: the 24 loads of a thread are issued together, the eight 1024-wide scans run from
	// registers and meet in one barrier (a round per 1024 tiles is three barriers and a load round trip each)
	__shared__ uint32_t wsum[8][16];
	__shared__ uint32_t carry;
	// tileOrder: tiles per (round, wavefront, class), turned into each wavefront's first position inside its class
	#define SCAN_MAX_ROUNDS 8u // 65536 tiles; beyond that the order stays the raster order
	__shared__ uint32_t wcls[SCAN_MAX_ROUNDS][16][4];
	uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const bool ordered = ORDER && tileOrder && nTiles <= SCAN_MAX_ROUNDS * 8192u;
	if (tid == 0) carry = 0;
	__syncthreads();
	uint32_t keep[8]; // the first round's lengths: most frames have no second round, and the order pass then reads nothing again
	for (uint32_t base = 0; base < nTiles; base += 8192) {
		uint32_t v[8], s[8];
		uint32_t cc[4] = { 0, 0, 0, 0 };
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const uint32_t i = base + j * 1024 + tid;
			v[j] = (i < nTiles) ? count[i] + countA[i] + countBig[i] : 0u;
			if (ORDER && base == 0) keep[j] = v[j];
		}
		if (ORDER && ordered) {
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const bool in = base + j * 1024 + tid < nTiles;
				const uint32_t c = tile_class(v[j]);
#pragma unroll
				for (uint32_t k = 0; k < 4; k++) cc[k] += (uint32_t)__popcll(__ballot(in && c == k));
			}
			if (lane < 4) wcls[base >> 13][wid][lane] = lane == 0 ? cc[0] : lane == 1 ? cc[1] : lane == 2 ? cc[2] : cc[3];
		}
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const uint32_t t = vkr_wave_scan_incl_u32(v[j]);
			s[j] = t;
			if (lane == 63) wsum[j][wid] = t;
		}
		__syncthreads();
		uint32_t running = carry;
#pragma unroll
		for (int j = 0; j < 8; j++) {
			uint32_t wpre = 0, tot = 0;
#pragma unroll
			for (uint32_t w = 0; w < 16; w++) {
				const uint32_t x = wsum[j][w];
				if (w < wid) wpre += x;
				tot += x;
			}
			const uint32_t i = base + j * 1024 + tid;
			if (i < nTiles) { offset[i] = running + wpre + s[j] - v[j]; cursor[i] = 0; }
			running += tot;
		}
		__syncthreads();
		if (tid == 0) carry = running;
		__syncthreads();
	}
	if (ORDER && tileOrder && !ordered)
		for (uint32_t i = tid; i < nTiles; i += 1024u) tileOrder[i] = i;
	if (ORDER && ordered) {
		// A counting sort of the tiles by class, longest bins first: a kernel that gives one workgroup to every tile and
		// whose time is set by its deepest tiles (k_overshade, the light's depth pass) starts those first.  Positions
		// come from ballots and popcounts; inside a class the tiles keep their raster order.  The order schedules work
		// and decides no result.
		const uint32_t rounds = (nTiles + 8191u) >> 13;
		__syncthreads();
		if (wid == 0) {
			// exclusive prefix over (class, round, wavefront), 64 at a time, by the first wavefront
			const uint32_t Q = 4u * rounds * 16u;
			uint32_t run = 0;
			for (uint32_t q0 = 0; q0 < Q; q0 += 64u) {
				const uint32_t q = q0 + lane;
				const uint32_t k = q / (rounds * 16u), r = (q / 16u) % rounds, w = q & 15u;
				const uint32_t c = wcls[r][w][k];
				const uint32_t t = vkr_wave_scan_incl_u32(c);
				wcls[r][w][k] = run + t - c;
				run += (uint32_t)__shfl((int)t, 63, 64);
			}
		}
		__syncthreads();
		const unsigned long long lt = (1ull << lane) - 1ull;
		for (uint32_t base = 0; base < nTiles; base += 8192) {
			uint32_t at[4];
#pragma unroll
			for (uint32_t k = 0; k < 4; k++) at[k] = wcls[base >> 13][wid][k];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const uint32_t i = base + j * 1024 + tid;
				const bool in = i < nTiles;
				const uint32_t n = base == 0 ? keep[j] : (in ? count[i] + countA[i] + countBig[i] : 0u);
				const uint32_t c = tile_class(n);
#pragma unroll
				for (uint32_t k = 0; k < 4; k++) {
					const unsigned long long b = __ballot(in && c == k);
					if (in && c == k) tileOrder[at[k] + (uint32_t)__popcll(b & lt)] = i;
					at[k] += (uint32_t)__popcll(b);
				}
			}
		}
	}
	if (tid == 0) {
		offset[nTiles] = carry;
		if (carry > cap) atomicOr(overflow, 2u);
		// status of the pass (appended records, overflow bits, bin entries needed), written where the host reads it
		// after the stream is synchronised: pinned host memory, no copy commands
		statusOut[0] = overflow[-1];
		statusOut[1] = overflow[0] | (carry > cap ? 2u : 0u);
		statusOut[2] = carry;
	}
}

// One thread per record slot; a record that touches more than 8 tiles is
// spread over the whole wavefront.
#define BIG_FILL_BLOCKS 256u // workgroups of k_bin_fill that take the many-tile records

__device__ __forceinline__ void bin_fill_big(const BinFillArgs& a);

// One launch fills both kinds of entries: the first BIG_FILL_BLOCKS workgroups take the records that touch more than
// 8 tiles (cursor atomics, behind the others' entries), the rest one thread per record slot.
__global__ __launch_bounds__(256) void k_bin_fill(BinFillArgs a)
{
	if (blockIdx.x < BIG_FILL_BLOCKS) { bin_fill_big(a); return; }
	const uint32_t r = (blockIdx.x - BIG_FILL_BLOCKS) * blockDim.x + threadIdx.x;
	const uint32_t n = min(a.baseCount + *a.recCount, a.recCap);
	bool has = r < n;
	uint32_t tx0 = 0, ty0 = 0, tx1 = 0, ty1 = 0, key = 0;
	bool alpha = false;
	uint2 tr = make_uint2(0xffffffffu, 0u);
	if (has) tr = a.recTiles[r];
	has = tr.x != 0xffffffffu; // slot of a triangle that draws nothing
	if (has) {
		tx0 = tr.x & 0xffffu; ty0 = tr.x >> 16; tx1 = tr.y & 0xffffu; ty1 = (tr.y >> 16) & 0x7fffu;
		if (a.keys) key = a.recs[r].prim;
		alpha = (tr.y & 0x80000000u) != 0;
	}
	const uint32_t nt = has ? (tx1 - tx0 + 1) * (ty1 - ty0 + 1) : 0;
	const bool big = nt > 8;
	if (has && !big) {
		// positions were fixed by the counting pass
		uint32_t k = 0;
		for (uint32_t ty = ty0; ty <= ty1; ty++)
			for (uint32_t tx = tx0; tx <= tx1; tx++, k++) {
				const uint32_t tile = ty * a.tilesX + tx;
				// a tile's bin: records without alpha test, then alpha-tested ones, then those touching many tiles
				const uint32_t idx = a.tileOffset[tile] + (alpha ? a.tileCount[tile] : 0u) + a.recLocal[(size_t)r * 8 + k];
				if (idx < a.binCap) {
					a.binList[idx] = r;
					if (a.keys) a.keys[idx] = key;
				}
			}
	}
}

// The records that touch more than 8 tiles, one wavefront per record (lanes over its tiles); their entries go behind
// the others' in each tile's bin, positions from a cursor.
__device__ __forceinline__ void bin_fill_big(const BinFillArgs& a)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (BIG_FILL_BLOCKS * blockDim.x) >> 6;
	const uint32_t n = min(*a.bigCount, a.recCap);
	for (uint32_t q = wave; q < n; q += nWaves) {
		const uint32_t r = a.bigQueue[q];
		const uint2 tr = a.recTiles[r];
		const uint32_t wx0 = tr.x & 0xffffu, wy0 = tr.x >> 16, wx1 = tr.y & 0xffffu, wy1 = (tr.y >> 16) & 0x7fffu;
		const uint32_t key = a.keys ? a.recs[r].prim : 0u;
		const uint32_t w = wx1 - wx0 + 1, cnt = w * (wy1 - wy0 + 1);
		for (uint32_t k = lane; k < cnt; k += 64) {
			const uint32_t tile = (wy0 + k / w) * a.tilesX + (wx0 + k % w);
			const uint32_t idx = a.tileOffset[tile] + a.tileCount[tile] + a.tileCountA[tile] + atomicAdd(&a.cursor[tile], 1u);
			if (idx < a.binCap) {
				a.binList[idx] = r;
				if (a.keys) a.keys[idx] = key;
			}
		}
	}
}

// ---------------------------------------------------------------------------
void vkr_launch_setup(hipStream_t s, const SetupArgs& a)
{
	// at least one workgroup: it also zeroes the next pass's counters
	hipLaunchKernelGGL(k_setup, dim3(a.count ? (a.count + 255) / 256 : 1), dim3(256), 0, s, a);
	if (!a.count) return;
	// few triangles need the clipper; a fixed grid strides over however many were queued
	hipLaunchKernelGGL(k_setup_clip, dim3(512), dim3(64), 0, s, a);
	hipLaunchKernelGGL(k_count_big, dim3(256), dim3(256), 0, s, a);
}

void vkr_launch_scan_tiles(hipStream_t s, const uint32_t* count, const uint32_t* countA, const uint32_t* countBig, uint32_t* offset, uint32_t* cursor,
                           uint32_t nTiles, uint32_t cap, uint32_t* overflow, uint32_t* statusOut, uint32_t* tileOrder)
{
	if (tileOrder) hipLaunchKernelGGL(k_scan_tiles<true>, dim3(1), dim3(1024), 0, s, count, countA, countBig, offset, cursor, nTiles, cap, overflow, statusOut, tileOrder);
	else hipLaunchKernelGGL(k_scan_tiles<false>, dim3(1), dim3(1024), 0, s, count, countA, countBig, offset, cursor, nTiles, cap, overflow, statusOut, tileOrder);
}

void vkr_launch_bin_fill(hipStream_t s, const BinFillArgs& a, uint32_t maxRecs)
{
	if (!maxRecs) return;
	hipLaunchKernelGGL(k_bin_fill, dim3(BIG_FILL_BLOCKS + (maxRecs + 255) / 256), dim3(256), 0, s, a);
}
