// k_resolve.hip — per-pixel shading off the visibility buffer.
//
// Each thread takes the (depth | primitive) key of one pixel, fetches the
// triangle's interpolation planes (computed by triangle setup) and runs the
// fragment program of the pass exactly once per visible pixel:
//   k_resolve_gbuf     gBufWrite.frag:16-27        (deferred geometry subpass)
//   k_resolve_forward  default.frag:82-104         (forward pass incl. shadow lookup)
//   k_resolve_debug    debug.frag:19-62            (mip / depth / derivative views)
// The alpha-test discard of the first two already happened in the rasteriser.
// Texture LOD uses fine derivatives inside the pixel's 2x2 quad, evaluated on
// the triangle's own planes (helper-invocation semantics).
// Compiled with -ffp-contract=off (the mip-level buffer is an integer output).
#include "vkr_kernels.h"
#include "vkr_light.h"

struct Frag {
	uint32_t prim;   // submission-order index
	uint32_t tA;     // index into sc.tri (order A)
	const TriShade* ts; // the triangle's static shading record
	TriPlanes pl;
	float b[3];
	float uv[2], d[4];
	uint32_t mat;
};

// loads the visible fragment of pixel (px,py); returns false if none
template <bool NEEDZ>
__device__ __forceinline__ bool load_frag(const ResolveArgs& a, int px, int py, uint32_t tA, Frag& f)
{
	f.tA = tA;
	f.ts = a.sc.triShade + tA;
	f.pl = a.planes[tA]; // every visible triangle went through k_setup, which stored its planes
	vkr_bary_at(f.pl, px, py, f.b);
	vkr_uv_derivs(f.pl, px, py, f.b, f.ts->t0, f.ts->t1, f.ts->t2, f.uv, f.d);
	f.mat = f.ts->mat;
	return true;
}

// colour-only (nothing integer depends on it): hardware reciprocal square root
__device__ __forceinline__ void normalize3(float v[3])
{
	float inv = __builtin_amdgcn_rsqf(vkr_dot3(v[0], v[1], v[2], v[0], v[1], v[2]));
	v[0] *= inv; v[1] *= inv; v[2] *= inv;
}


// normal = v2fTBN * normalize(texture(uNormalMap).rgb*2-1), TBN interpolated
// per vertex (default.vert:57-81 decoded once per vertex at scene creation)
__device__ __forceinline__ void shading_normal(const ResolveArgs& a, const Frag& f, const float nm[4], float n[3])
{
	float m[9];
#pragma unroll
	for (int i = 0; i < 9; i++) m[i] = vkr_interp(f.b, f.ts->tbn[0][i], f.ts->tbn[1][i], f.ts->tbn[2][i]);
	float t[3] = { nm[0] * 2.0f - 1.0f, nm[1] * 2.0f - 1.0f, nm[2] * 2.0f - 1.0f };
	normalize3(t);
#pragma unroll
	for (int r = 0; r < 3; r++) n[r] = fmaf(m[6 + r], t[2], fmaf(m[3 + r], t[1], m[0 + r] * t[0]));
}

// The four material fetches of gBufWrite.frag / default.frag: normal map rgb,
// base colour rgb, metalness r, roughness r.  The footprint (level, weights,
// texel indices) is rebuilt only when a texture's size differs from the
// previous one's; materials with repacked chains take two fetches.
struct MatSample { float nm[4], alb[4], metal, rough; };

__device__ __forceinline__ void sample_material(const ResolveArgs& a, const Frag& f, const float* dec, MatSample& o)
{
	TexTaps tp;
	float t[4];
	if (f.ts->levels) {
		// the four textures have one size: (normal.rgb, metalness.r) and (baseColor.rgb, roughness.r) per texel;
		// the chains are described by the triangle's record, nothing else is read before the texels
		const uint32_t wh0 = f.ts->wh0;
		vkr_make_taps_wh(wh0 & 0xffffu, wh0 >> 16, f.ts->levels, f.uv[0], f.uv[1], f.d, tp);
		vkr_fetch_chain<15>(a.sc, f.ts->nmOff0, tp, nullptr, o.nm);
		o.metal = o.nm[3];
		vkr_fetch_chain<15>(a.sc, f.ts->arOff0, tp, f.ts->srgbAR ? dec : nullptr, o.alb);
		o.rough = o.alb[3];
		return;
	}
	const MatInfo& mi = a.sc.mat[f.mat];
	const TexInfo& tn = a.sc.tex[mi.tex[4]];
	vkr_make_taps(tn, f.uv[0], f.uv[1], f.d, tp);
	vkr_fetch<7>(a.sc, tn, tp, dec, o.nm);
	const TexInfo& ta = a.sc.tex[mi.tex[0]];
	if (!vkr_same_dims(ta, tp)) vkr_make_taps(ta, f.uv[0], f.uv[1], f.d, tp);
	vkr_fetch<7>(a.sc, ta, tp, dec, o.alb);
	const TexInfo& tm = a.sc.tex[mi.tex[1]];
	if (!vkr_same_dims(tm, tp)) vkr_make_taps(tm, f.uv[0], f.uv[1], f.d, tp);
	vkr_fetch<1>(a.sc, tm, tp, dec, t);
	o.metal = t[0];
	const TexInfo& tr = a.sc.tex[mi.tex[2]];
	if (!vkr_same_dims(tr, tp)) vkr_make_taps(tr, f.uv[0], f.uv[1], f.d, tp);
	vkr_fetch<1>(a.sc, tr, tp, dec, t);
	o.rough = t[0];
}

__device__ __forceinline__ void decode_key(unsigned long long key, float& depth, uint32_t& prim)
{
	depth = u2f((uint32_t)(key >> 32));
	prim = 0xFFFFFFFEu - (uint32_t)key; // 0xFFFFFFFF when nothing was drawn
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resolve_gbuf(ResolveArgs a)
{
	__shared__ float sDec[256];
	const uint32_t rw = a.view.x1 - a.view.x0, rh = a.view.rows;
	uint32_t bx, by;
	vkr_screen_place(bx, by);
	const uint32_t x = bx * 16 + (threadIdx.x & 15), y = by * 16 + (threadIdx.x >> 4);
	const bool inside = x < rw && y < rh;
	const size_t p = (size_t)y * rw + x;
	// the visibility key is requested before the decode table's round trip and the barrier behind it
	const unsigned long long key = inside ? a.vis[p] : 0ull;
	sDec[threadIdx.x] = a.sc.srgbDecode[threadIdx.x];
	__syncthreads();
	if (!inside) return;
	const int px = (int)(a.view.x0 + x), py = vkr_view_gy(a.view, y);
	float depth; uint32_t prim;
	decode_key(key, depth, prim);
	a.depth[p] = depth;
	if (a.triId) a.triId[p] = prim;
	// attachment clears 0.1 (main/main.cpp:3843-3851): half(0.1) = 0x2E66
	uint2 gn = make_uint2(0x2E662E66u, 0x2E662E66u), ga = gn;
	Frag f;
	if (prim != VKR_NO_PRIM && load_frag<false>(a, px, py, prim, f)) {
		MatSample ms;
		float N[3];
		sample_material(a, f, sDec, ms);
		shading_normal(a, f, ms.nm, N);
		gn.x = vkr_f2h_hw(N[0]) | (vkr_f2h_hw(N[1]) << 16);
		gn.y = vkr_f2h_hw(N[2]) | (vkr_f2h_hw(ms.metal) << 16);
		ga.x = vkr_f2h_hw(ms.alb[0]) | (vkr_f2h_hw(ms.alb[1]) << 16);
		ga.y = vkr_f2h_hw(ms.alb[2]) | (vkr_f2h_hw(ms.rough) << 16);
	}
	a.gN[p] = gn;
	a.gA[p] = ga;
}

// ---------------------------------------------------------------------------
// BRDF of default.frag:33-80 (forward flavour: nDotH floored at 0.0001)
__device__ __forceinline__ void brdf_forward(const float L[3], const float V[3], const float N[3], const float albedo[3],
                                             float metal, float roughTex, float out[3])
{
	const float PI = 3.14159265359f;
	float Hh[3] = { V[0] + L[0], V[1] + L[1], V[2] + L[2] };
	normalize3(Hh);
	float rough = roughTex * roughTex;
	float ndh = vkr_dot3(N[0], N[1], N[2], Hh[0], Hh[1], Hh[2]);
	float ndv = vkr_dot3(N[0], N[1], N[2], V[0], V[1], V[2]);
	float ndl = vkr_dot3(N[0], N[1], N[2], L[0], L[1], L[2]);
	float vdh = vkr_dot3(V[0], V[1], V[2], Hh[0], Hh[1], Hh[2]);
	float nh = fmaxf(ndh, 0.0001f);
	float nh2 = nh * nh, nh4 = nh2 * nh2, r2 = rough * rough;
	// colour only: hardware reciprocal and exp2
	float ndf = __builtin_amdgcn_exp2f((nh2 - 1.0f) * 1.4426950408889634f * __builtin_amdgcn_rcpf(r2 * nh2)) * __builtin_amdgcn_rcpf(0.001f + PI * r2 * nh4);
	float x = 1.0f - vdh;
	float x2 = x * x, x5 = x2 * x2 * x;
	const float rvdh = __builtin_amdgcn_rcpf(vdh);
	float tl = 2.0f * (fmaxf(0.0f, ndh) * fmaxf(0.0f, ndv) * rvdh);
	float tr = 2.0f * (fmaxf(0.0f, ndh) * fmaxf(0.0f, ndl) * rvdh);
	float G = fminf(1.0f, fminf(tl, tr));
	float rdenom = __builtin_amdgcn_rcpf(0.001f + 4.0f * fmaxf(ndv, 0.0f) * fmaxf(ndl, 0.0f));
#pragma unroll
	for (int i = 0; i < 3; i++) {
		float f0 = (1.0f - metal) * 0.04f + metal * albedo[i];
		float F = f0 + (1.0f - f0) * x5;
		float diffuse = (albedo[i] * (1.0f / PI)) * (1.0f - F) * (1.0f - metal);
		out[i] = diffuse + (ndf * F * G) * rdenom;
	}
}

// texture(shadowMap, coord) with the compare sampler of utils/vkutil.cpp:198-217
__device__ __forceinline__ float shadow_lookup(const float* sm, int S, float u, float v, float ref)
{
	float x = u * (float)S - 0.5f, y = v * (float)S - 0.5f;
	if (!(fabsf(x) < 1.0e9f)) x = 0.0f;
	if (!(fabsf(y) < 1.0e9f)) y = 0.0f;
	float x0f = floorf(x), y0f = floorf(y);
	float fx = x - x0f, fy = y - y0f;
	int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
	x0 = min(max(x0, 0), S - 1); x1 = min(max(x1, 0), S - 1);
	y0 = min(max(y0, 0), S - 1); y1 = min(max(y1, 0), S - 1);
	float t00 = (ref <= sm[(size_t)y0 * S + x0]) ? 1.0f : 0.0f;
	float t10 = (ref <= sm[(size_t)y0 * S + x1]) ? 1.0f : 0.0f;
	float t01 = (ref <= sm[(size_t)y1 * S + x0]) ? 1.0f : 0.0f;
	float t11 = (ref <= sm[(size_t)y1 * S + x1]) ? 1.0f : 0.0f;
	return vkr_lerp(vkr_lerp(t00, t10, fx), vkr_lerp(t01, t11, fx), fy);
}

__global__ __launch_bounds__(256) void k_resolve_forward(ResolveArgs a)
{
	__shared__ float sDec[256];
	__shared__ float sT[257];
	const uint32_t rw = a.view.x1 - a.view.x0, rh = a.view.rows;
	uint32_t bx, by;
	vkr_screen_place(bx, by);
	const uint32_t x = bx * 16 + (threadIdx.x & 15), y = by * 16 + (threadIdx.x >> 4);
	const bool inside = x < rw && y < rh;
	const size_t p = (size_t)y * rw + x;
	const unsigned long long key = inside ? a.vis[p] : 0ull; // requested before the tables' round trip
	sDec[threadIdx.x] = a.sc.srgbDecode[threadIdx.x];
	vkr_load_srgb_table(sT, a.sc.srgbThresh);
	__syncthreads();
	if (!inside) return;
	const int px = (int)(a.view.x0 + x), py = vkr_view_gy(a.view, y);
	float depth; uint32_t prim;
	decode_key(key, depth, prim);
	if (a.depth) a.depth[p] = depth;
	if (a.triId) a.triId[p] = prim;
	// clear colour (0.1,0.1,0.1,1), main/main.cpp:3988-3991
	uint32_t colour = vkr_pack_colour_lds(sT, 0.1f, 0.1f, 0.1f, 1.0f);
	Frag f;
	if (prim != VKR_NO_PRIM && load_frag<false>(a, px, py, prim, f)) {
		MatSample ms;
		float N[3], P[3];
		sample_material(a, f, sDec, ms);
		const float* alb = ms.alb;
		shading_normal(a, f, ms.nm, N);
#pragma unroll
		for (int i = 0; i < 3; i++)
			P[i] = vkr_interp(f.b, f.ts->pos[0][i], f.ts->pos[1][i], f.ts->pos[2][i]);
		float Lv[3] = { a.lightPos[0] - P[0], a.lightPos[1] - P[1], a.lightPos[2] - P[2] };
		float Vv[3] = { a.camPos[0] - P[0], a.camPos[1] - P[1], a.camPos[2] - P[2] };
		float dist2 = vkr_dot3(Lv[0], Lv[1], Lv[2], Lv[0], Lv[1], Lv[2]);
		float L[3] = { Lv[0], Lv[1], Lv[2] };
		normalize3(L);
		normalize3(Vv);
		float br[3];
		brdf_forward(L, Vv, N, alb, ms.metal, ms.rough, br);
		float ndl = fmaxf(vkr_dot3(N[0], N[1], N[2], L[0], L[1], L[2]), 0.0001f);
		float att = 1.0f / dist2;
		float4 ls = vkr_xform(a.lightMat, P[0], P[1], P[2]);
		float sh = shadow_lookup(a.shadow, (int)a.shadowSize, ls.x / ls.w, ls.y / ls.w, ls.z / ls.w);
		sh = fmaxf(sh, 0.1f);
		float c[3];
#pragma unroll
		for (int i = 0; i < 3; i++) {
			float ambient = 0.03f * alb[i];
			float lit = (br[i] * 100.0f) * a.lightColour[i] * ndl;
			c[i] = ambient + (lit * sh) * att;
		}
		colour = vkr_pack_colour_lds(sT, c[0], c[1], c[2], 1.0f);
	}
	a.colour[p] = colour;
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ float linearise_depth(float d)
{
	return (2.0f * 0.1f * 100.0f) / (100.0f + 0.1f - (2.0f * d - 1.0f) * (100.0f - 0.1f));
}

__global__ __launch_bounds__(256) void k_resolve_debug(ResolveArgs a)
{
	__shared__ float sT[257];
	vkr_load_srgb_table(sT, a.sc.srgbThresh);
	__syncthreads();
	const uint32_t rw = a.view.x1 - a.view.x0, rh = a.view.rows;
	uint32_t bx, by;
	vkr_screen_place(bx, by);
	const uint32_t x = bx * 16 + (threadIdx.x & 15), y = by * 16 + (threadIdx.x >> 4);
	if (x >= rw || y >= rh) return;
	const size_t p = (size_t)y * rw + x;
	const int px = (int)(a.view.x0 + x), py = vkr_view_gy(a.view, y);
	float depth; uint32_t prim;
	decode_key(a.vis[p], depth, prim);
	if (a.depth) a.depth[p] = depth;
	if (a.triId) a.triId[p] = prim;
	uint32_t colour = vkr_pack_colour_lds(sT, 0.1f, 0.1f, 0.1f, 1.0f); // main/main.cpp:4139-4143
	uint8_t mip = 0xff;
	if (prim != VKR_NO_PRIM) {
		// passes in file order carry the file-order rank; map back to the order-A triangle
		const uint32_t tA = a.useOrderB ? a.sc.orderB[a.sc.nTri + prim] : prim;
		Frag f;
		const int mode = a.debugMode;
		if (load_frag<true>(a, px, py, tA, f)) {
			float c[4] = { 0, 0, 0, 1.0f };
			if (mode == 3) {
				float v = linearise_depth(depth) / 100.0f;
				c[0] = c[1] = c[2] = v;
			} else if (mode == 4) {
				float zc = linearise_depth(depth);
				float zx = linearise_depth(vkr_depth_at(f.pl.Az, f.pl.Bz, f.pl.Cz, px ^ 1, py));
				float zy = linearise_depth(vkr_depth_at(f.pl.Az, f.pl.Bz, f.pl.Cz, px, py ^ 1));
				float dx = (px & 1) ? (zc - zx) : (zx - zc);
				float dy = (py & 1) ? (zc - zy) : (zy - zc);
				c[0] = dx * 5.0f; c[1] = dy * 5.0f; c[2] = 0.0f;
			} else {
				const TexInfo& t = a.sc.tex[a.sc.mat[f.mat].tex[0]];
				int lv; float fr;
				vkr_lod_split(vkr_rho2(t, f.d), t.levels - 1, lv, fr);
				mip = (uint8_t)lv;
				if (mode == 2) {
					const float col[7][3] = { { 17.0f, 119.0f, 51.0f }, { 136.0f, 34.0f, 85.0f }, { 0.0f, 114.0f, 178.0f },
						{ 204.0f, 121.0f, 167.0f }, { 0.0f, 158.0f, 115.0f }, { 213.0f, 94.0f, 0.0f }, { 240.0f, 228.0f, 66.0f } };
					int fl = lv, ce = (fr > 0.0f) ? lv + 1 : lv;
					float fc = fr;
					if (fr >= 1.0f) { fl = lv + 1; ce = lv + 1; fc = 0.0f; }
#pragma unroll
					for (int i = 0; i < 3; i++) {
						float ca = col[fl % 7][i] / 255.0f, cb = col[ce % 7][i] / 255.0f;
						c[i] = ca * (1.0f - fc) + cb * fc;
					}
				} else {
					vkr_trilinear(a.sc, t, f.uv[0], f.uv[1], f.d, c);
				}
			}
			colour = vkr_pack_colour_lds(sT, c[0], c[1], c[2], c[3]);
		}
	}
	a.colour[p] = colour;
	if (a.mip) a.mip[p] = mip;
}

// ---------------------------------------------------------------------------
// G-buffer resolve and deferred lighting in one kernel (an option of the deferred path): the pixel's G-buffer
// texels go from the resolve to the lighting in registers, rounded to half precision exactly as the RGBA16F
// attachments round them, so the image is the one the two-kernel path produces.  The resolve keeps the
// texture-address units busy and the lighting the vector ALU; in one kernel the wavefronts of a CU are in
// different phases and the two overlap.  The attachments are written only if the caller asked for them.
// A wavefront shades an 8x8 block (what the lighting's N.L skip wants).
template <bool SAMECOL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_resolve_light(ResolveArgs a, LightArgs l)
{
	__shared__ float sDec[256];
	__shared__ float sT[257];
	const uint32_t rw = a.view.x1 - a.view.x0, rh = a.view.rows;
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	uint32_t bx, by;
	vkr_screen_place(bx, by);
	const uint32_t x = bx * 16u + (wv & 1u) * 8u + (lane & 7u), y = by * 16u + (wv >> 1) * 8u + (lane >> 3);
	const bool inside = x < rw && y < rh;
	const size_t p = (size_t)y * rw + x;
	const unsigned long long key = inside ? a.vis[p] : 0ull; // requested before the tables' round trip
	sDec[threadIdx.x] = a.sc.srgbDecode[threadIdx.x];
	vkr_load_srgb_table(sT, l.srgbThresh);
	__syncthreads();
	if (!inside) return;
	const int px = (int)(a.view.x0 + x), py = vkr_view_gy(a.view, y);
	float depth; uint32_t prim;
	decode_key(key, depth, prim);
	if (a.depth) a.depth[p] = depth;
	if (a.triId) a.triId[p] = prim;
	uint2 gn = make_uint2(0x2E662E66u, 0x2E662E66u), ga = gn; // attachment clears 0.1
	Frag f;
	if (prim != VKR_NO_PRIM && load_frag<false>(a, px, py, prim, f)) {
		MatSample ms;
		float N[3];
		sample_material(a, f, sDec, ms);
		shading_normal(a, f, ms.nm, N);
		gn.x = vkr_f2h_hw(N[0]) | (vkr_f2h_hw(N[1]) << 16);
		gn.y = vkr_f2h_hw(N[2]) | (vkr_f2h_hw(ms.metal) << 16);
		ga.x = vkr_f2h_hw(ms.alb[0]) | (vkr_f2h_hw(ms.alb[1]) << 16);
		ga.y = vkr_f2h_hw(ms.alb[2]) | (vkr_f2h_hw(ms.rough) << 16);
	}
	if (a.gN) a.gN[p] = gn;
	if (a.gA) a.gA[p] = ga;
	// deferredShading.frag discards where depth == 1: the clear colour stays
	a.colour[p] = (depth == 1.0f) ? l.clearColour : vkr_light_pixel<SAMECOL>(l, sT, depth, gn, ga, x, y);
}

static dim3 grid16(const ViewDev& v) { return dim3((v.x1 - v.x0 + 15) / 16, (v.rows + 15) / 16); }

void vkr_launch_resolve_gbuf(hipStream_t s, const ResolveArgs& a)
{
	hipLaunchKernelGGL(k_resolve_gbuf, grid16(a.view), dim3(256), 0, s, a);
}
void vkr_launch_resolve_forward(hipStream_t s, const ResolveArgs& a)
{
	hipLaunchKernelGGL(k_resolve_forward, grid16(a.view), dim3(256), 0, s, a);
}
void vkr_launch_resolve_debug(hipStream_t s, const ResolveArgs& a)
{
	hipLaunchKernelGGL(k_resolve_debug, grid16(a.view), dim3(256), 0, s, a);
}
void vkr_launch_resolve_light(hipStream_t s, const ResolveArgs& a, const LightArgs& l)
{
	if (vkr_lights_share_colour(l)) hipLaunchKernelGGL((k_resolve_light<true>), grid16(a.view), dim3(256), 0, s, a, l);
	else hipLaunchKernelGGL((k_resolve_light<false>), grid16(a.view), dim3(256), 0, s, a, l);
}
