// k_light.hip — full-screen passes: deferred brazier lighting, the overdraw /
// overshading colour ramp and the mosaic post-process.
// These produce colour only (parity within 1/255), so this file is compiled
// with floating-point contraction enabled and uses the hardware's fast
// reciprocal / rsqrt / exp2.
#include "vkr_light.h"

// ---------------------------------------------------------------------------
// deferredShading.frag:92-119.  One thread per pixel; reads the two RGBA16F
// G-buffer attachments and D32 depth (20 B), writes RGBA8 (4 B).
//
// The kernel is bound by vector-ALU issue, not by memory, so the shader's
// arithmetic is reorganised to the fewest instructions per (pixel, light):
//  * With H = (V+L)/|V+L| and unit V, L:  |V+L|^2 = 2(1+V.L),
//    N.H = (N.V+N.L)/|V+L|,  V.H = (1+V.L)/|V+L| = 1/(2 invh) with
//    invh = rsq(2(1+V.L)); H itself is never formed and 1/(V.H) = 2 invh.
//  * D's exponent (nh^2-1)/(r^2 nh^2) = (1 - 1/nh^2)/r^2: one rcp, one fma.
//  * D's denominator, the BRDF denominator and the 1/nh^2 of D's exponent share one reciprocal.
//  * The Fresnel/diffuse/specular mix is linear in three per-light scalars:
//      brdf_c = kd_c (1-f0_c) (1-x^5) + f0_c spec + (1-f0_c) spec x^5,  x = 1 - V.H,
//    so when every light has the same colour (the reference's 21 braziers,
//    main/main.cpp:1350-1352) the per-channel work leaves the loop entirely:
//    the loop accumulates S1 = sum t(1-x^5), S2 = sum t spec, S3 = sum t spec x^5
//    (t = N.L/d^2) and the channels are combined once at the end.
//  * max(.,0) of quantities bounded by 1 uses the free clamp modifier.
//  * A wavefront shades an 8x8 block and skips a light whose N.L is <= 0 at all of its pixels (every term
//    but the ambient one carries max(N.L, 0)).
// Per light: 45 plain VALU instructions and 4 transcendentals (2 rsq, 1 rcp, 1 exp2).  Measured on MI355X
// (tools/microbench/): a plain f32 instruction costs ~2.5 SIMD cycles per wavefront, one with an SGPR operand,
// a v_min or a v_cmp ~4, a transcendental ~7.5; the kernel keeps the vector ALU busy ~100 % of the time.
// The per-pixel body lives in vkr_light.h (the fused resolve + lighting kernel shares it).
// ---------------------------------------------------------------------------
template <bool SAMECOL>
__global__ __launch_bounds__(256) void k_deferred_light(LightArgs a)
{
	__shared__ float sT[257];
	const uint32_t rw = a.view.x1 - a.view.x0, rh = a.view.rows;
	// a workgroup shades 32x8 pixels, each wavefront an 8x8 block: the pixels of a wavefront mostly lie on
	// one surface, which is what lets whole wavefronts skip the lights behind it (below), and a workgroup's
	// rows are whole 128-byte lines of the depth buffer (16 pixels wide, half of each line was fetched twice)
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	const uint32_t x = blockIdx.x * 32u + wv * 8u + (lane & 7u), y = blockIdx.y * 8u + (lane >> 3);
	const bool inside = x < rw && y < rh;
	const uint32_t p = y * rw + x;
	// the pixel's inputs are requested before the encode table's round trip and the barrier behind it
	float depth = 1.0f;
	uint2 gn = make_uint2(0u, 0u), ga = gn;
	if (inside) { depth = a.depth[p]; gn = a.gN[p]; ga = a.gA[p]; }
	vkr_load_srgb_table(sT, a.srgbThresh);
	__syncthreads();
	if (!inside) return;
	if (depth == 1.0f) { a.colour[p] = a.clearColour; return; } // discard: the clear colour stays
	a.colour[p] = vkr_light_pixel<SAMECOL>(a, sT, depth, gn, ga, x, y);
}

// overVisRead.frag:7-13: oColor = (0,0.3,0,0.1) + vec4(vec3(stencil/25), 0).  The colour depends on the
// 8-bit counter only: each workgroup builds the 256-entry table once, pixels are lookups.
__global__ __launch_bounds__(256) void k_overvis_colour(const uint8_t* __restrict__ stencil, uint32_t* __restrict__ colour,
                                                         const float* __restrict__ thresh, uint32_t n)
{
	__shared__ uint32_t lut[256];
	{
		const float s = (float)threadIdx.x / 25.0f;
		lut[threadIdx.x] = vkr_pack_colour(thresh, 0.0f + s, 0.3f + s, 0.0f + s, 0.1f);
	}
	__syncthreads();
	// four pixels per thread
	const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
	if (p + 3 < n && ((size_t)stencil & 3u) == 0 && ((size_t)colour & 15u) == 0) {
		const uint32_t s4 = *(const uint32_t*)(stencil + p);
		*(uint4*)(colour + p) = make_uint4(lut[s4 & 255u], lut[(s4 >> 8) & 255u], lut[(s4 >> 16) & 255u], lut[s4 >> 24]);
	} else {
		for (uint32_t k = p; k < n && k < p + 4; k++) colour[k] = lut[stencil[k]];
	}
}

// postProcess.frag:9-25 sampling the offscreen R8G8B8A8_SRGB image
// (main/main.cpp:1607, 3216) with the default sampler.  src covers the
// rectangle [srcX0, srcX0+srcW) x [srcY0, srcY0+srcH) of the frame.
__global__ __launch_bounds__(256) void k_mosaic(const uint32_t* __restrict__ src, uint32_t srcX0, uint32_t srcY0, uint32_t srcW,
                                                 uint32_t srcH, uint32_t* __restrict__ dst, ViewDev view,
                                                 const float* __restrict__ decode, const float* __restrict__ thresh)
{
	__shared__ float sT[257];
	__shared__ float sDec[256];
	vkr_load_srgb_table(sT, thresh);
	sDec[threadIdx.x] = decode[threadIdx.x];
	__syncthreads();
	const uint32_t rw = view.x1 - view.x0, rh = view.rows;
	// 16 x 64 pixels per workgroup, four rows per thread: the two tables are loaded once for all of them
	const uint32_t x = blockIdx.x * 16 + (threadIdx.x & 15);
	if (x >= rw) return;
	for (uint32_t r = 0; r < 4; r++) {
	const uint32_t y = blockIdx.y * 64 + r * 16 + (threadIdx.x >> 4);
	if (y >= rh) return;
	const int W = (int)view.W, H = (int)view.H;
	const float sx = (float)W, sy = (float)H;
	const float kw = 5.0f / sx, kh = 3.0f / sy;
	const float ox = 0.5f / sx, oy = 0.5f / sy;
	const float u = ((float)(view.x0 + x) + 0.5f) / sx, v = ((float)(view.y0 + y) + 0.5f) / sy;
	float su = kw * floorf(u / kw), sv = kh * floorf(v / kh);
	su = fminf(fmaxf(su, 0.0f + ox), 1.0f - ox);
	sv = fminf(fmaxf(sv, 0.0f + oy), 1.0f - oy);
	const float fxp = su * sx - 0.5f, fyp = sv * sy - 0.5f;
	const float x0f = floorf(fxp), y0f = floorf(fyp);
	const float fx = fxp - x0f, fy = fyp - y0f;
	// REPEAT addressing; the clamped coordinate keeps the texel index inside [-1, size], so the wrap is one
	// conditional add or subtract instead of a remainder by a size that is no power of two
	int x0 = (int)x0f, x1 = x0 + 1, y0 = (int)y0f, y1 = y0 + 1;
	x0 += (x0 < 0) ? W : 0; x1 -= (x1 >= W) ? W : 0;
	y0 += (y0 < 0) ? H : 0; y1 -= (y1 >= H) ? H : 0;
	// the source rectangle holds every texel a block of this rectangle can touch
	x0 = min(max(x0 - (int)srcX0, 0), (int)srcW - 1); x1 = min(max(x1 - (int)srcX0, 0), (int)srcW - 1);
	y0 = min(max(y0 - (int)srcY0, 0), (int)srcH - 1); y1 = min(max(y1 - (int)srcY0, 0), (int)srcH - 1);
	const uint32_t t00 = src[(size_t)y0 * srcW + x0], t10 = src[(size_t)y0 * srcW + x1];
	const uint32_t t01 = src[(size_t)y1 * srcW + x0], t11 = src[(size_t)y1 * srcW + x1];
	float c[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const uint32_t sh = 8u * k;
		float a00, a10, a01, a11;
		if (k < 3) {
			a00 = sDec[(t00 >> sh) & 0xffu]; a10 = sDec[(t10 >> sh) & 0xffu];
			a01 = sDec[(t01 >> sh) & 0xffu]; a11 = sDec[(t11 >> sh) & 0xffu];
		} else {
			a00 = vkr_unorm8(t00 >> 24); a10 = vkr_unorm8(t10 >> 24);
			a01 = vkr_unorm8(t01 >> 24); a11 = vkr_unorm8(t11 >> 24);
		}
		c[k] = vkr_lerp(vkr_lerp(a00, a10, fx), vkr_lerp(a01, a11, fx), fy);
	}
	dst[(size_t)y * rw + x] = vkr_pack_colour_lds(sT, c[0], c[1], c[2], c[3]);
	}
}

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* dst, uint32_t value, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] = value;
}

void vkr_launch_deferred_light(hipStream_t s, const LightArgs& a)
{
	const uint32_t n = (a.view.x1 - a.view.x0) * a.view.rows;
	if (!n) return;
	// every light the same colour (the reference's braziers): the per-channel work leaves the light loop
	bool same = true;
	for (int i = 1; i < VKR_NUM_LIGHTS_K; i++)
		for (int c = 0; c < 3; c++) same = same && a.lightCol[i][c] == a.lightCol[0][c];
	const dim3 grid((a.view.x1 - a.view.x0 + 31) / 32, (a.view.rows + 7) / 8);
	if (same) hipLaunchKernelGGL((k_deferred_light<true>), grid, dim3(256), 0, s, a);
	else hipLaunchKernelGGL((k_deferred_light<false>), grid, dim3(256), 0, s, a);
}

void vkr_launch_overvis_colour(hipStream_t s, const uint8_t* stencil, uint32_t* colour, const float* srgbThresh, uint32_t n)
{
	if (!n) return;
	hipLaunchKernelGGL(k_overvis_colour, dim3((n + 1023) / 1024), dim3(256), 0, s, stencil, colour, srgbThresh, n);
}

void vkr_launch_mosaic(hipStream_t s, const uint32_t* src, uint32_t srcX0, uint32_t srcY0, uint32_t srcW, uint32_t srcH,
                       uint32_t* dst, ViewDev view, const float* srgbDecode, const float* srgbThresh)
{
	dim3 g((view.x1 - view.x0 + 15) / 16, (view.rows + 63) / 64);
	hipLaunchKernelGGL(k_mosaic, g, dim3(256), 0, s, src, srcX0, srcY0, srcW, srcH, dst, view, srgbDecode, srgbThresh);
}

void vkr_launch_fill_u32(hipStream_t s, uint32_t* dst, uint32_t value, size_t n)
{
	if (!n) return;
	hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, value, n);
}
