// vkr_device.h — device-side data layout and the exact-arithmetic helpers the
// kernels share.  Everything that feeds an integer output (coverage, depth
// bits, mip level, alpha test) is written with explicit fma()/fmaf() and is
// compiled with -ffp-contract=off, so the operation order below IS the
// definition of the result.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VKR_TILE 32            // screen tile edge in pixels
#define VKR_TILE_PIX (VKR_TILE * VKR_TILE)
#define VKR_SUBPIX 256         // 8 sub-pixel bits
#define VKR_MAX_LEVELS 16
#define VKR_MAX_SUBTRI 6       // a triangle clipped by near + 4 guard planes: <= 8 vertices
#define VKR_NUM_LIGHTS_K 21     // deferredShading.frag:5
#define VKR_NO_PRIM 0xffffffffu

// Guard band in NDC half-widths: small enough that the snapped coordinates
// keep every edge function below 2^53, so edges are exact in double.
static inline double vkr_guard_band(uint32_t W, uint32_t H) { return ((W > H ? W : H) <= 4096u) ? 32.0 : 8.0; }

struct Mat4 { float m[16]; };  // column-major, glm::mat4

// One texture: all mip levels in the scene's texel pool (RGBA8, one u32 per texel).
struct alignas(16) TexInfo {
	// header: what a fetch needs before it knows the level (one 16-byte load)
	uint32_t wh0;      // level-0 width | height << 16
	int32_t levels;
	int32_t srgb;
	uint32_t alphaOff; // level-0 offset of this texture's chain in the alpha byte pool, ~0u if it has none
	uint32_t offset[VKR_MAX_LEVELS]; // texel offset of each level in the pool
	uint16_t w[VKR_MAX_LEVELS], h[VKR_MAX_LEVELS];
};

// Material -> the five bindings of the material descriptor set
// (main/main.cpp:770-846): 0 baseColor, 1 metalness, 2 roughness,
// 3 alphaMask (baseColor when absent), 4 normalMap.
// packNM / packAR: when the four shaded textures of the material have one
// size, two repacked chains built at scene creation hold (normal.rgb,
// metalness.r) and (baseColor.rgb, roughness.r) per texel, so shading takes
// two trilinear fetches instead of four; ~0u when the material has none.
struct MatInfo {
	uint32_t tex[5];
	uint32_t alphaTest; // 1 if the alpha texture holds any texel with alpha < 128
	uint32_t packNM, packAR;
};

// Texture coordinates of a triangle's vertices and the description of the texture its alpha test samples
// (binding 3 of its material), gathered once at scene creation: the alpha pass reads one record instead of
// walking triangle -> vertices and triangle -> material -> texture.
struct alignas(16) AlphaAux {
	float2 t0, t1, t2;
	uint32_t off0, wh0; // chain offset in the alpha byte pool; level-0 width | height << 16
	int32_t levels;
	uint32_t _pad[3];
};

// Screen-space sub-triangle produced by triangle setup (64 bytes).
struct SubTri {
	int32_t X0, Y0, X1, Y1, X2, Y2; // snapped to 1/256 px, orientation positive
	uint32_t prim;                  // submission-order primitive index
	uint32_t flags;                 // bit0: alpha-tested, bit1: front-facing
	double Az, Bz, Cz;              // z_ndc(sx,sy) = Az*sx + Bz*sy + Cz
	double bias;                    // depth bias (shadow pass), 0 otherwise
};
#define VKR_ST_ALPHA 1u
#define VKR_ST_FRONT 2u

// Per triangle (submission order A), built once per scene: everything the shading kernels need that does not change
// from frame to frame, so that a pixel's chain of dependent loads is visibility key -> (interpolation planes, this
// record) -> texels, instead of key -> indices -> vertices / material -> texture description -> level offset -> texels.
struct alignas(16) TriShade {
	float2 t0, t1, t2;        // texture coordinates of the three vertices
	uint32_t nmOff0, arOff0;  // first texel of the repacked chains (normal.rgb, metalness.r) and (baseColor.rgb, roughness.r)
	uint32_t wh0;             // their size, width | height << 16
	int32_t levels;           // 0: the material has no repacked chains (the general path shades)
	uint32_t srgbAR;          // the base-colour bytes are sRGB
	uint32_t mat;
	float tbn[3][9];          // decoded TBN matrix of the three vertices (default.vert:57-81), column-major
	float pos[3][3];          // world positions of the three vertices (the forward shader interpolates them)
};
static_assert(sizeof(TriShade) == 192, "TriShade is twelve 16-byte loads");

struct SceneDev {
	const float* pos;        // 3 per vertex
	const float* uv;         // 2 per vertex
	const uint32_t* tcomp;   // packed TBN quaternion per vertex
	const float4* tbn;       // decoded TBN matrix per vertex: 3 float4 (9 entries, column-major, + 3 unused)
	const uint32_t* tri;     // 3 global vertex indices per triangle, submission order A
	const uint32_t* triMat;  // material per triangle (order A)
	const uint32_t* orderB;  // file-order rank of triangle t (order A index -> order B index)
	const TexInfo* tex;
	const MatInfo* mat;
	const uint32_t* texels;  // texel pool
	const uint8_t* alphaTexels; // alpha bytes of the textures an alpha test samples, same level layout per chain
	const struct AlphaAux* alphaAux; // per triangle (order A): what an alpha test needs besides the interpolation planes
	const TriShade* triShade;        // per triangle (order A)
	const float* srgbDecode; // 256
	const float* srgbThresh; // 255
	uint32_t nTri, nTriOpaque, nVert, nTex, nMat;
};

// Frame geometry shared by the raster / resolve kernels.
struct ViewDev {
	uint32_t W, H;           // full extent of the pass (viewport)
	uint32_t x0, y0, x1, y1; // rectangle rendered by this context
	uint32_t tilesX, tilesY;
	// Rows rendered and how they map to the frame.  stripeN == 1: the rectangle's rows, local row ly = frame row y0 + ly.
	// stripeN > 1 (sort-first by interleaved stripes; x0 = y0 = 0, x1 = W, y1 = H): of the frame cut into stripes of
	// stripeH rows (a multiple of the tile height) this context renders stripes stripeI, stripeI + stripeN, ... and
	// stores them back to back, so a tile row never straddles two stripes.
	uint32_t rows;
	uint32_t stripeH, stripeN, stripeI;
};

// frame row of local row ly
#include "vkr_place.h"
// (blockIdx.x, blockIdx.y) of the place this workgroup of a screen-walking kernel takes (vkr_place.h): blocks of 8 x 8
// workgroups per XCD on large grids, 4 x 4 on small ones, where 8 x 8 would leave an XCD only a handful of blocks.
__device__ __forceinline__ void vkr_screen_place(uint32_t& bx, uint32_t& by)
{
	if (gridDim.x * gridDim.y >= 16384u) vkr_xcd_place_2d<8u, 8u>(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, bx, by);
	else vkr_xcd_place_2d<4u, 4u>(blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, bx, by);
}

__host__ __device__ __forceinline__ int vkr_view_gy(const ViewDev& v, uint32_t ly)
{
	if (v.stripeN <= 1u) return (int)(v.y0 + ly);
	const uint32_t k = ly / v.stripeH;
	return (int)((k * v.stripeN + v.stripeI) * v.stripeH + (ly - k * v.stripeH));
}

// local rows [ly0, ly1] (inclusive) that hold the frame rows [gy0, gy1] (inclusive, inside [y0, y1)); false if none do
__host__ __device__ __forceinline__ bool vkr_view_local_rows(const ViewDev& v, int gy0, int gy1, uint32_t& ly0, uint32_t& ly1)
{
	if (gy0 > gy1) return false;
	if (v.stripeN <= 1u) { ly0 = (uint32_t)gy0 - v.y0; ly1 = (uint32_t)gy1 - v.y0; return true; }
	const uint32_t H = v.stripeH, N = v.stripeN, I = v.stripeI;
	// first owned row at or below gy0 ... (rows grow downwards)
	uint32_t s0 = (uint32_t)gy0 / H, r0 = s0 % N;
	long long a;
	if (r0 == I) a = (long long)(s0 / N) * H + ((uint32_t)gy0 - s0 * H);
	else { const uint32_t s = s0 + ((I + N - r0) % N); a = (long long)(s / N) * H; if ((long long)s * H > gy1) return false; }
	// last owned row at or above gy1
	uint32_t s1 = (uint32_t)gy1 / H, r1 = s1 % N;
	long long b;
	if (r1 == I) b = (long long)(s1 / N) * H + ((uint32_t)gy1 - s1 * H);
	else { const uint32_t back = (r1 + N - I) % N; if (back > s1) return false; const uint32_t s = s1 - back; b = (long long)(s / N) * H + (H - 1); }
	if (a > b) return false;
	ly0 = (uint32_t)a; ly1 = (uint32_t)b;
	return true;
}

// ---------------------------------------------------------------------------
// wave-wide primitives on data-parallel-primitive moves (inside the vector ALU; a __shfl is a round trip through the
// LDS crossbar).  Lanes a move does not reach take 0, the identity of both operations below.
// ---------------------------------------------------------------------------
// inclusive prefix sum over the 64 lanes: shifted sums inside each row of 16 lanes, then the two row broadcasts
__device__ __forceinline__ uint32_t vkr_wave_scan_incl_u32(uint32_t v)
{
	int x = (int)v;
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
	return (uint32_t)x;
}

// ---------------------------------------------------------------------------
// bit helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// float -> half, round to nearest even
__device__ __forceinline__ uint16_t vkr_f2h(float f)
{
	uint32_t x = f2u(f);
	uint32_t sign = (x >> 16) & 0x8000u;
	uint32_t ax = x & 0x7fffffffu;
	if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
	if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
	if (ax < 0x33000001u) return (uint16_t)sign;
	if (ax < 0x38800000u) {
		uint32_t e = ax >> 23;
		uint32_t m = (ax & 0x7fffffu) | 0x800000u;
		uint32_t shift = 126u - e;
		uint32_t h = m >> shift;
		uint32_t rem = m & ((1u << shift) - 1u);
		uint32_t half = 1u << (shift - 1u);
		if (rem > half || (rem == half && (h & 1u))) h++;
		return (uint16_t)(sign | h);
	}
	uint32_t h = ((ax - 0x38000000u) >> 13);
	uint32_t rem = ax & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
	return (uint16_t)(sign | h);
}

__device__ __forceinline__ float vkr_h2f(uint16_t h)
{
	uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
	uint32_t e = (h >> 10) & 0x1fu;
	uint32_t m = h & 0x3ffu;
	if (e == 0) {
		if (m == 0) return u2f(sign);
		float v = (float)m * 5.9604644775390625e-8f;
		return sign ? -v : v;
	}
	if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
	return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

// float -> half with the conversion instruction (round to nearest even, half
// denormals kept): the same value as vkr_f2h for every input; NaNs are mapped
// to the one quiet NaN vkr_f2h produces.
__device__ __forceinline__ uint32_t vkr_f2h_hw(float f)
{
	uint32_t h = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f);
	if (f != f) h = ((f2u(f) >> 16) & 0x8000u) | 0x7e00u;
	return h;
}

// (float)k / 255.0f for k in [0,255] and (float)k / 1023.0f for k in [0,1023]
// without the division sequence: with 1/n split into a float and the float
// nearest to the remainder, fma(k, hi, k * lo) rounds the quotient once, from
// a value within 2^-48 of it - the correctly rounded quotient for every k of
// the range (checked against exact rational arithmetic for all 256 / 1024
// inputs).
__device__ __forceinline__ float vkr_unorm8(uint32_t k)
{
	const float HI = 0x1.010102p-8f;   // float(1/255)
	const float LO = -0x1.fdfdfep-33f; // float(1/255 - HI)
	const float kf = (float)k;
	return fmaf(kf, HI, kf * LO);
}
__device__ __forceinline__ float vkr_unorm10(uint32_t k)
{
	const float HI = 0x1.004010p-10f;  // float(1/1023)
	const float LO = 0x1.004010p-40f;  // float(1/1023 - HI)
	const float kf = (float)k;
	return fmaf(kf, HI, kf * LO);
}

__device__ __forceinline__ float vkr_clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// linear [0,1] -> sRGB code: number of thresholds <= x
__device__ __forceinline__ uint32_t vkr_srgb_encode(const float* thresh, float x)
{
	if (!(x > 0.0f)) return 0;
	int lo = 0, hi = 255;
#pragma unroll
	for (int it = 0; it < 8; it++) {
		int mid = (lo + hi) >> 1;
		bool ge = (lo < hi) && (x >= thresh[mid]);
		bool lt = (lo < hi) && !(x >= thresh[mid]);
		if (ge) lo = mid + 1;
		if (lt) hi = mid;
	}
	return (uint32_t)lo;
}

__device__ __forceinline__ uint32_t vkr_unorm_encode(float x)
{
	if (!(x > 0.0f)) return 0;
	if (x >= 1.0f) return 255;
	return (uint32_t)(int)floorf(x * 255.0f + 0.5f);
}

__device__ __forceinline__ uint32_t vkr_pack_colour(const float* thresh, float r, float g, float b, float a)
{
	return vkr_srgb_encode(thresh, vkr_clamp01(r)) | (vkr_srgb_encode(thresh, vkr_clamp01(g)) << 8) |
	       (vkr_srgb_encode(thresh, vkr_clamp01(b)) << 16) | (vkr_unorm_encode(vkr_clamp01(a)) << 24);
}

// ---- sRGB encode through an LDS copy of the threshold table -------------------------------------------
// T holds 257 floats: T[0] = -inf, T[k+1] = threshold k, T[256] = +inf.  A hardware log2/exp2 estimate of
// round(255*OETF(x)) is corrected by at most one step against the exact thresholds, so the result equals
// the table search of vkr_srgb_encode for every x.
__device__ __forceinline__ void vkr_load_srgb_table(float* T, const float* __restrict__ thresh)
{
	for (uint32_t i = threadIdx.x; i < 257u; i += blockDim.x)
		T[i] = (i == 0) ? -__builtin_inff() : (i == 256) ? __builtin_inff() : thresh[i - 1];
}

__device__ __forceinline__ uint32_t vkr_srgb_encode_lds(const float* T, float x)
{
	x = fminf(fmaxf(x, 0.0f), 1.0f); // NaN -> 0, as vkr_srgb_encode
	const float e = (x <= 0.0031308f) ? 12.92f * x
	                                  : 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.4f)) - 0.055f;
	int code = (int)(e * 255.0f + 0.5f);
	code = min(max(code, 0), 255);
	if (x < T[code]) code--;             // code is right iff T[code] <= x < T[code+1]
	else if (x >= T[code + 1]) code++;
	return (uint32_t)code;
}

__device__ __forceinline__ uint32_t vkr_pack_colour_lds(const float* T, float r, float g, float b, float a)
{
	return vkr_srgb_encode_lds(T, r) | (vkr_srgb_encode_lds(T, g) << 8) | (vkr_srgb_encode_lds(T, b) << 16) |
	       (vkr_unorm_encode(vkr_clamp01(a)) << 24);
}

__device__ __forceinline__ float vkr_lerp(float a, float b, float t) { return fmaf(t, b - a, a); }
__device__ __forceinline__ float vkr_dot3(float ax, float ay, float az, float bx, float by, float bz)
{
	return fmaf(az, bz, fmaf(ay, by, ax * bx));
}

// gl_Position = M * vec4(p,1)
__device__ __forceinline__ float4 vkr_xform(const Mat4& M, float x, float y, float z)
{
	float r[4];
#pragma unroll
	for (int i = 0; i < 4; i++) {
		float t = M.m[0 * 4 + i] * x;
		t = fmaf(M.m[1 * 4 + i], y, t);
		t = fmaf(M.m[2 * 4 + i], z, t);
		r[i] = t + M.m[3 * 4 + i];
	}
	return make_float4(r[0], r[1], r[2], r[3]);
}

// ---------------------------------------------------------------------------
// attribute planes of a triangle from its clip-space vertices (homogeneous
// rasterisation): lambda_i(sx,sy) = a_i*sx + b_i*sy + c_i, sx,sy pixel coords.
// ---------------------------------------------------------------------------
struct TriPlanes {
	double a[3], b[3], c[3];
	double Az, Bz, Cz;
};

// returns false for a degenerate (zero-determinant) triangle; NEEDZ = also the depth plane
template <bool NEEDZ = true>
__device__ __forceinline__ bool vkr_tri_planes(const float4 c0, const float4 c1, const float4 c2, uint32_t W, uint32_t H,
                                               TriPlanes& pl)
{
	double hW = (double)W * 0.5, hH = (double)H * 0.5;
	double hx0 = ((double)c0.x + (double)c0.w) * hW, hy0 = ((double)c0.y + (double)c0.w) * hH, hw0 = (double)c0.w;
	double hx1 = ((double)c1.x + (double)c1.w) * hW, hy1 = ((double)c1.y + (double)c1.w) * hH, hw1 = (double)c1.w;
	double hx2 = ((double)c2.x + (double)c2.w) * hW, hy2 = ((double)c2.y + (double)c2.w) * hH, hw2 = (double)c2.w;
	// i=0: j=1,k=2
	pl.a[0] = hy1 * hw2 - hy2 * hw1; pl.b[0] = hx2 * hw1 - hx1 * hw2; pl.c[0] = hx1 * hy2 - hx2 * hy1;
	// i=1: j=2,k=0
	pl.a[1] = hy2 * hw0 - hy0 * hw2; pl.b[1] = hx0 * hw2 - hx2 * hw0; pl.c[1] = hx2 * hy0 - hx0 * hy2;
	// i=2: j=0,k=1
	pl.a[2] = hy0 * hw1 - hy1 * hw0; pl.b[2] = hx1 * hw0 - hx0 * hw1; pl.c[2] = hx0 * hy1 - hx1 * hy0;
	double det = (hx0 * pl.a[0] + hy0 * pl.b[0]) + hw0 * pl.c[0];
	if (det == 0.0 || !(det == det)) return false;
	if (NEEDZ) {
		double z0 = c0.z, z1 = c1.z, z2 = c2.z;
		double idet = 1.0 / det;
		pl.Az = ((z0 * pl.a[0] + z1 * pl.a[1]) + z2 * pl.a[2]) * idet;
		pl.Bz = ((z0 * pl.b[0] + z1 * pl.b[1]) + z2 * pl.b[2]) * idet;
		pl.Cz = ((z0 * pl.c[0] + z1 * pl.c[1]) + z2 * pl.c[2]) * idet;
	} else {
		pl.Az = pl.Bz = pl.Cz = 0.0;
	}
	return true;
}

__device__ __forceinline__ float vkr_depth_at(double Az, double Bz, double Cz, int px, int py)
{
	double sx = (double)px + 0.5, sy = (double)py + 0.5;
	return (float)fma(Az, sx, fma(Bz, sy, Cz));
}

__device__ __forceinline__ void vkr_bary_at(const TriPlanes& pl, int px, int py, float b[3])
{
	double sx = (double)px + 0.5, sy = (double)py + 0.5;
	double l0 = fma(pl.a[0], sx, fma(pl.b[0], sy, pl.c[0]));
	double l1 = fma(pl.a[1], sx, fma(pl.b[1], sy, pl.c[1]));
	double l2 = fma(pl.a[2], sx, fma(pl.b[2], sy, pl.c[2]));
	double is = 1.0 / ((l0 + l1) + l2);
	b[0] = (float)(l0 * is); b[1] = (float)(l1 * is); b[2] = (float)(l2 * is);
}

__device__ __forceinline__ float vkr_interp(const float b[3], float f0, float f1, float f2)
{
	return fmaf(b[2], f2, fmaf(b[1], f1, b[0] * f0));
}

// uv at the pixel and fine derivatives inside its 2x2 quad: d = (du/dx, dv/dx, du/dy, dv/dy)
__device__ __forceinline__ void vkr_uv_derivs(const TriPlanes& pl, int px, int py, const float b[3], float2 t0, float2 t1,
                                              float2 t2, float uv[2], float d[4])
{
	float bx[3], by[3];
	vkr_bary_at(pl, px ^ 1, py, bx);
	vkr_bary_at(pl, px, py ^ 1, by);
	uv[0] = vkr_interp(b, t0.x, t1.x, t2.x);
	uv[1] = vkr_interp(b, t0.y, t1.y, t2.y);
	float ux = vkr_interp(bx, t0.x, t1.x, t2.x), vx = vkr_interp(bx, t0.y, t1.y, t2.y);
	float uy = vkr_interp(by, t0.x, t1.x, t2.x), vy = vkr_interp(by, t0.y, t1.y, t2.y);
	if (px & 1) { d[0] = uv[0] - ux; d[1] = uv[1] - vx; } else { d[0] = ux - uv[0]; d[1] = vx - uv[1]; }
	if (py & 1) { d[2] = uv[0] - uy; d[3] = uv[1] - vy; } else { d[2] = uy - uv[0]; d[3] = vy - uv[1]; }
}

// ---------------------------------------------------------------------------
// texture sampling (sampler of utils/vkutil.cpp:179-196)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float vkr_log2_mantissa(float m)
{
	float t = m - 1.0f;
	float s = t / (2.0f + t);
	float s2 = s * s;
	float p = 0.09090909090909091f;
	p = fmaf(p, s2, 0.1111111111111111f);
	p = fmaf(p, s2, 0.14285714285714285f);
	p = fmaf(p, s2, 0.2f);
	p = fmaf(p, s2, 0.3333333333333333f);
	p = fmaf(p, s2, 1.0f);
	float r = (s * p) * 2.8853900817779268f;
	if (r > 1.0f) r = 1.0f;
	return r;
}

__device__ __forceinline__ void vkr_lod_split(float rho2, int q, int& d, float& f)
{
	d = 0; f = 0.0f;
	if (!(rho2 > 1.0f)) return;
	uint32_t bits = f2u(rho2);
	int e = (int)((bits >> 23) & 0xffu) - 127;
	float m = u2f((bits & 0x7fffffu) | 0x3f800000u);
	float p = vkr_log2_mantissa(m);
	float fr = fmaf(p, 0.5f, (e & 1) ? 0.5f : 0.0f);
	int lv = e >> 1;
	if (lv >= q) { d = q; f = 0.0f; return; }
	d = lv; f = fr;
}

__device__ __forceinline__ float vkr_rho2_wh(float tw, float th, const float d[4])
{
	float mx = d[0] * tw, my = d[1] * th;
	float nx = d[2] * tw, ny = d[3] * th;
	float rx = fmaf(my, my, mx * mx);
	float ry = fmaf(ny, ny, nx * nx);
	return (rx > ry) ? rx : ry;
}
__device__ __forceinline__ float vkr_rho2(const TexInfo& t, const float d[4]) { return vkr_rho2_wh((float)(t.wh0 & 0xffffu), (float)(t.wh0 >> 16), d); }

__device__ __forceinline__ int vkr_wrap(int i, int n)
{
	if ((n & (n - 1)) == 0) return i & (n - 1); // power of two: two's complement AND is the non-negative remainder
	int r = i % n;
	return r < 0 ? r + n : r;
}

// Bilinear footprint of one mip level: texel indices inside the level and the
// two weights.  REPEAT addressing; x1 = x0 + 1 wrapped.
struct LevelTaps {
	uint32_t i00, i10, i01, i11;
	float fx, fy;
};

template <bool POW2 = false>
__device__ __forceinline__ void vkr_level_taps(int w, int h, float u, float v, LevelTaps& o)
{
	float x = u * (float)w - 0.5f;
	float y = v * (float)h - 0.5f;
	if (!(fabsf(x) < 1.0e9f)) x = 0.0f;
	if (!(fabsf(y) < 1.0e9f)) y = 0.0f;
	float x0f = floorf(x), y0f = floorf(y);
	o.fx = x - x0f; o.fy = y - y0f;
	// POW2: the caller knows both sizes are powers of two (no branch, no division)
	int x0 = POW2 ? ((int)x0f & (w - 1)) : vkr_wrap((int)x0f, w), y0 = POW2 ? ((int)y0f & (h - 1)) : vkr_wrap((int)y0f, h);
	int x1 = x0 + 1, y1 = y0 + 1;
	if (x1 == w) x1 = 0;
	if (y1 == h) y1 = 0;
	o.i00 = (uint32_t)(y0 * w + x0); o.i10 = (uint32_t)(y0 * w + x1);
	o.i01 = (uint32_t)(y1 * w + x0); o.i11 = (uint32_t)(y1 * w + x1);
}

// Everything about a trilinear fetch that depends only on the texture's
// dimensions: level, fraction and the footprint on each of the two levels.
// Textures of one material usually share their size, so one TexTaps serves
// several fetches (vkr_same_dims).
struct TexTaps {
	uint32_t w0, h0;
	int levels;
	int lv;      // lower level
	float fr;    // weight of level lv+1; 0 = single level
	LevelTaps t0, t1;
};

__device__ __forceinline__ bool vkr_same_dims(const TexInfo& t, const TexTaps& tp)
{
	return t.wh0 == (tp.w0 | (tp.h0 << 16)) && t.levels == tp.levels;
}

__device__ __forceinline__ void vkr_make_taps_wh(uint32_t w0, uint32_t h0, int levels, float u, float v, const float d[4], TexTaps& tp)
{
	tp.w0 = w0; tp.h0 = h0; tp.levels = levels;
	vkr_lod_split(vkr_rho2_wh((float)w0, (float)h0, d), levels - 1, tp.lv, tp.fr);
	// level sizes halve, rounding down, not below 1 (utils/vkimage.cpp:165-197)
	int w = (int)(tp.w0 >> tp.lv), h = (int)(tp.h0 >> tp.lv);
	if (w < 1) w = 1;
	if (h < 1) h = 1;
	vkr_level_taps(w, h, u, v, tp.t0);
	if (tp.fr != 0.0f) {
		int w1 = (int)(tp.w0 >> (tp.lv + 1)), h1 = (int)(tp.h0 >> (tp.lv + 1));
		if (w1 < 1) w1 = 1;
		if (h1 < 1) h1 = 1;
		vkr_level_taps(w1, h1, u, v, tp.t1);
	}
}
__device__ __forceinline__ void vkr_make_taps(const TexInfo& t, float u, float v, const float d[4], TexTaps& tp)
{
	vkr_make_taps_wh(t.wh0 & 0xffffu, t.wh0 >> 16, t.levels, u, v, d, tp);
}

// texel offset of level lv of a full mip chain that starts at off0 (levels are stored back to back)
__device__ __forceinline__ uint32_t vkr_level_offset(uint32_t off0, uint32_t w0, uint32_t h0, int lv)
{
	uint32_t off = off0;
	for (int k = 0; k < lv; k++) {
		uint32_t w = w0 >> k, h = h0 >> k;
		off += (w ? w : 1u) * (h ? h : 1u);
	}
	return off;
}

// channel c (0..2 colour, 3 alpha) of a packed texel; dec = 256-entry sRGB decode table or nullptr for UNORM
__device__ __forceinline__ float vkr_channel(uint32_t p, int c, const float* dec)
{
	const uint32_t k = (p >> (8 * c)) & 0xffu;
	return (dec && c < 3) ? dec[k] : vkr_unorm8(k);
}

// bilinear fetch of the channels in MASK on one level
template <int MASK>
__device__ __forceinline__ void vkr_bilinear_taps(const uint32_t* __restrict__ base, const LevelTaps& lt, const float* dec, float out[4])
{
	const uint32_t p00 = base[lt.i00], p10 = base[lt.i10], p01 = base[lt.i01], p11 = base[lt.i11];
#pragma unroll
	for (int c = 0; c < 4; c++)
		if (MASK & (1 << c))
			out[c] = vkr_lerp(vkr_lerp(vkr_channel(p00, c, dec), vkr_channel(p10, c, dec), lt.fx),
			                  vkr_lerp(vkr_channel(p01, c, dec), vkr_channel(p11, c, dec), lt.fx), lt.fy);
}

// texture(sampler, uv) for the channels in MASK with a prepared footprint.
// srgbDecode: the 256-entry table (global or LDS) used when the texture is sRGB.
template <int MASK>
__device__ __forceinline__ void vkr_fetch(const SceneDev& sc, const TexInfo& t, const TexTaps& tp, const float* srgbDecode, float out[4])
{
	const float* dec = t.srgb ? srgbDecode : nullptr;
	const uint32_t* b0 = sc.texels + t.offset[tp.lv];
	if (tp.fr == 0.0f) { vkr_bilinear_taps<MASK>(b0, tp.t0, dec, out); return; }
	const uint32_t* b1 = sc.texels + t.offset[tp.lv + 1];
	float a[4], b[4];
	vkr_bilinear_taps<MASK>(b0, tp.t0, dec, a);
	vkr_bilinear_taps<MASK>(b1, tp.t1, dec, b);
#pragma unroll
	for (int c = 0; c < 4; c++)
		if (MASK & (1 << c)) out[c] = vkr_lerp(a[c], b[c], tp.fr);
}

// vkr_fetch for a chain described by (first texel, size): the level offsets are recomputed instead of read from the
// texture table, which takes a memory round trip out of the pixel's dependent chain.  Same texels, same arithmetic.
template <int MASK>
__device__ __forceinline__ void vkr_fetch_chain(const SceneDev& sc, uint32_t off0, const TexTaps& tp, const float* dec, float out[4])
{
	const uint32_t o0 = vkr_level_offset(off0, tp.w0, tp.h0, tp.lv);
	const uint32_t* b0 = sc.texels + o0;
	if (tp.fr == 0.0f) { vkr_bilinear_taps<MASK>(b0, tp.t0, dec, out); return; }
	uint32_t w = tp.w0 >> tp.lv, h = tp.h0 >> tp.lv;
	const uint32_t* b1 = sc.texels + o0 + (w ? w : 1u) * (h ? h : 1u);
	float a[4], b[4];
	vkr_bilinear_taps<MASK>(b0, tp.t0, dec, a);
	vkr_bilinear_taps<MASK>(b1, tp.t1, dec, b);
#pragma unroll
	for (int c = 0; c < 4; c++)
		if (MASK & (1 << c)) out[c] = vkr_lerp(a[c], b[c], tp.fr);
}

// .a of texture(sampler, uv) read from the alpha byte pool, for a texture described by (off0, w0, h0,
// levels): the alpha test needs nothing else; bytes instead of RGBA texels keep a foliage texture's whole
// chain inside L2, and with the level offsets recomputed the texel loads are the only memory round trip.
// Same arithmetic as the .a component of vkr_fetch.
__device__ __forceinline__ float vkr_bilinear_alpha8(const uint8_t* __restrict__ base, const LevelTaps& lt)
{
	const float a00 = vkr_unorm8(base[lt.i00]), a10 = vkr_unorm8(base[lt.i10]);
	const float a01 = vkr_unorm8(base[lt.i01]), a11 = vkr_unorm8(base[lt.i11]);
	return vkr_lerp(vkr_lerp(a00, a10, lt.fx), vkr_lerp(a01, a11, lt.fx), lt.fy);
}

__device__ __forceinline__ float vkr_trilinear_alpha_wh(const uint8_t* __restrict__ alphaTexels, uint32_t off0, uint32_t w0, uint32_t h0,
                                                        int levels, float u, float v, const float d[4])
{
	TexTaps tp;
	vkr_make_taps_wh(w0, h0, levels, u, v, d, tp);
	const uint32_t o0 = vkr_level_offset(off0, w0, h0, tp.lv);
	const float a = vkr_bilinear_alpha8(alphaTexels + o0, tp.t0);
	if (tp.fr == 0.0f) return a;
	uint32_t w = w0 >> tp.lv, h = h0 >> tp.lv;
	const uint32_t o1 = o0 + (w ? w : 1u) * (h ? h : 1u);
	const float b = vkr_bilinear_alpha8(alphaTexels + o1, tp.t1);
	return vkr_lerp(a, b, tp.fr);
}

// one-off trilinear fetch (all four channels)
__device__ __forceinline__ void vkr_trilinear(const SceneDev& sc, const TexInfo& t, float u, float v, const float d[4],
                                              float out[4])
{
	TexTaps tp;
	vkr_make_taps(t, u, v, d, tp);
	vkr_fetch<15>(sc, t, tp, sc.srgbDecode, out);
}

// TBN decode (default.vert:57-81): GLSL mat3 column-major m[c*3+r]
__device__ __forceinline__ void vkr_decode_tbn(uint32_t packed, float m[9])
{
	const float SQRT2 = 1.41421354f, INV_SQRT2 = 0.707106769f;
	float r = vkr_unorm10((packed >> 20) & 0x3ffu);
	float g = vkr_unorm10((packed >> 10) & 0x3ffu);
	float b = vkr_unorm10(packed & 0x3ffu);
	int maxIndex = (int)(packed >> 30);
	float s0 = r * SQRT2 - INV_SQRT2, s1 = g * SQRT2 - INV_SQRT2, s2 = b * SQRT2 - INV_SQRT2;
	float dd = fmaf(s2, s2, fmaf(s1, s1, s0 * s0));
	float maxc = sqrtf(1.0f - dd);
	float qx, qy, qz, qw;
	if (maxIndex == 0) { qx = maxc; qy = s0; qz = s1; qw = s2; }
	else if (maxIndex == 1) { qx = s0; qy = maxc; qz = s1; qw = s2; }
	else if (maxIndex == 2) { qx = s0; qy = s1; qz = maxc; qw = s2; }
	else { qx = s0; qy = s1; qz = s2; qw = maxc; }
	float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz;
	float qxz = qx * qz, qxy = qx * qy, qyz = qy * qz;
	float qwx = qw * qx, qwy = qw * qy, qwz = qw * qz;
	m[0] = 1.0f - 2.0f * (qyy + qzz); m[1] = 2.0f * (qxy + qwz);        m[2] = 2.0f * (qxz - qwy);
	m[3] = 2.0f * (qxy - qwz);        m[4] = 1.0f - 2.0f * (qxx + qzz); m[5] = 2.0f * (qyz + qwx);
	m[6] = 2.0f * (qxz + qwy);        m[7] = 2.0f * (qyz - qwx);        m[8] = 1.0f - 2.0f * (qxx + qyy);
}
