// vkr_light.h — the per-pixel body of the deferred lighting pass (deferredShading.frag:92-119), shared by
// k_deferred_light and by the fused resolve + lighting kernel.  Colour only (parity within 1/255): floating-point
// contraction is switched on for this function whatever the including file is compiled with.
#pragma once
#include "vkr_kernels.h"

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float sat(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

// n / d from the hardware reciprocal r ~ 1/d and one residual correction: the rounded quotient (but for rare ties)
__device__ __forceinline__ float vkr_div_fix(float n, float d, float r)
{
	const float q = n * r;
	return __builtin_fmaf(__builtin_fmaf(-q, d, n), r, q);
}
__device__ __forceinline__ float vkr_div_rn(float n, float d) { return vkr_div_fix(n, d, fast_rcp(d)); }

__device__ __forceinline__ float h2f_lo(uint32_t v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu)); }
__device__ __forceinline__ float h2f_hi(uint32_t v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16)); }

// One pixel: G-buffer texels (RGBA16F pairs), depth and pixel position in the rendered rectangle -> packed RGBA8.
// sT = the sRGB threshold table in LDS (vkr_load_srgb_table).
template <bool SAMECOL>
__device__ __forceinline__ uint32_t vkr_light_pixel(const LightArgs& a, const float* sT, const float depth, const uint2 gn, const uint2 ga,
                                                    const uint32_t x, const uint32_t y)
{
#pragma clang fp contract(fast)
	const float px = (float)(a.view.x0 + x), py = (float)vkr_view_gy(a.view, y);
	// posFromDepth: clip = (uv*2-1, depth, 1); world = (inverse(camera)*inverse(projection)) * clip.
	// The position is the one ill-conditioned input of the pass: w = m23*depth + m33 cancels to ~1/100 of its terms at
	// far depths, and next to a brazier N.L is a difference of positions.  Every operation up to the position is
	// therefore written out in one fixed order (product rounded, then fused adds, as the CPU restatement does) instead
	// of being left to the compiler's contraction, and the quotients are corrected to the rounded result.
	float vs[4], Px, Py, Pz;
	{
#pragma clang fp contract(off)
		const float cx = vkr_div_rn(px + 0.5f, (float)a.view.W) * 2.0f - 1.0f;
		const float cy = vkr_div_rn(py + 0.5f, (float)a.view.H) * 2.0f - 1.0f;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			float t = a.invPC.m[0 * 4 + i] * cx;
			t = __builtin_fmaf(a.invPC.m[1 * 4 + i], cy, t);
			t = __builtin_fmaf(a.invPC.m[2 * 4 + i], depth, t);
			vs[i] = t + a.invPC.m[3 * 4 + i];
		}
		const float iw = fast_rcp(vs[3]);
		Px = vkr_div_fix(vs[0], vs[3], iw); Py = vkr_div_fix(vs[1], vs[3], iw); Pz = vkr_div_fix(vs[2], vs[3], iw);
	}
	float Vx = a.camPos[0] - Px, Vy = a.camPos[1] - Py, Vz = a.camPos[2] - Pz;
	{
		float inv = fast_rsq(Vx * Vx + Vy * Vy + Vz * Vz);
		Vx *= inv; Vy *= inv; Vz *= inv;
	}
	float Nx = h2f_lo(gn.x), Ny = h2f_hi(gn.x), Nz = h2f_lo(gn.y);
	{
		float inv = fast_rsq(Nx * Nx + Ny * Ny + Nz * Nz);
		Nx *= inv; Ny *= inv; Nz *= inv;
	}
	const float metal = h2f_hi(gn.y);
	const float alb[3] = { h2f_lo(ga.x), h2f_hi(ga.x), h2f_lo(ga.y) };
	const float roughTex = h2f_hi(ga.y);
	const float rough = roughTex * roughTex;
	const float r2 = rough * rough;
	const float PI = 3.14159265359f;
	const float LOG2E = 1.4426950408889634f;
	// light-independent terms
	const float ndv = Nx * Vx + Ny * Vy + Nz * Vz;
	const float ndvp = fmaxf(ndv, 0.0f);
	const float pir2x4 = 4.0f * PI * r2;
	const float cexp = LOG2E * fast_rcp(fmaxf(r2, 1.0e-30f)); // exponent of D is cexp * (1 - 1/nh^2)
	const float V2x = Vx + Vx, V2y = Vy + Vy, V2z = Vz + Vz;
	const float om = 1.0f - metal;
	float f0[3], omf0[3], kdo[3];
#pragma unroll
	for (int c = 0; c < 3; c++) {
		f0[c] = om * 0.04f + metal * alb[c];
		omf0[c] = 1.0f - f0[c];
		kdo[c] = (alb[c] * (1.0f / PI)) * om * omf0[c];
	}
	float S0 = 0.0f;                          // sum 1/d^2 (ambient)
	float S1[3] = { 0, 0, 0 }, S2[3] = { 0, 0, 0 }, S3[3] = { 0, 0, 0 }; // SAMECOL: only [0] is used
#ifndef VKR_LIGHT_UNROLL
#define VKR_LIGHT_UNROLL 3 // measured: 3 lights per trip beats 1, 7 and 21 (the whole table in scalar registers ahead of the loop)
#endif
#pragma unroll VKR_LIGHT_UNROLL
	for (int i = 0; i < VKR_NUM_LIGHTS_K; i++) {
		const float Lx = a.lightPos[i][0] - Px, Ly = a.lightPos[i][1] - Py, Lz = a.lightPos[i][2] - Pz;
		const float d2 = Lx * Lx + Ly * Ly + Lz * Lz;
		const float invd = fast_rsq(d2);
		const float invd2 = invd * invd;
		const float nl = Nx * Lx + Ny * Ly + Nz * Lz;
		S0 += invd2;
		// every term below carries the factor max(N.L, 0): a light behind every surface of the wavefront adds nothing
		if (__builtin_amdgcn_ballot_w64(nl > 0.0f) == 0) continue;
		const float ndl = nl * invd;
		const float ndlp = sat(nl * invd);                       // max(N.L, 0); N.L <= 1
		const float opv2 = (V2x * Lx + V2y * Ly + V2z * Lz) * invd + 2.0f; // 2 (1 + V.L) = |V+L|^2
		const float invh = fast_rsq(opv2);
		const float xx = 1.0f - 0.5f * (opv2 * invh);           // 1 - V.H
		const float ndh = sat((ndv + ndl) * invh);               // max(N.H, 0)
		const float nh2 = fmaf(ndh, ndh, 1.0e-30f), nh4 = nh2 * nh2; // the offset keeps the shared reciprocal finite
		// D's denominator times the BRDF denominator, and 1/nh^2 for D's exponent, from one reciprocal
		const float den = (0.004f + pir2x4 * nh4) * (0.00025f + ndvp * ndlp);
		const float rr = fast_rcp(den * nh2);
		const float ex = fast_exp2(cexp - cexp * (rr * den));
		const float rden = rr * nh2;                             // 1 / (4 (0.001 + pi r^2 nh^4) (0.00025 + n.v n.l))
		const float x2 = xx * xx, x5 = x2 * x2 * xx;
		// G = min(1, 2 n.h min(n.v, n.l) / v.h) with 1/(2 v.h) = invh
		const float G = sat(4.0f * ((ndh * invh) * fminf(ndvp, ndlp)));
		const float spec = ex * G * rden;                        // D G / (0.001 + 4 n.v n.l)
		const float t = ndlp * invd2;
		const float ts = t * spec;
		const float ty = t - t * x5;                             // t (1 - x^5)
		const float tsx = ts * x5;
		if (SAMECOL) {
			S1[0] += ty; S2[0] += ts; S3[0] += tsx;
		} else {
#pragma unroll
			for (int c = 0; c < 3; c++) {
				S1[c] += a.lightCol[i][c] * ty; S2[c] += a.lightCol[i][c] * ts; S3[c] += a.lightCol[i][c] * tsx;
			}
		}
	}
	// colour_c = 10 col_c (kd_c(1-f0_c) S1 + f0_c S2 + (1-f0_c) S3) + 0.03 albedo_c S0: the ambient term is
	// added per light and attenuated with it (deferredShading.frag:108,115)
	float acc[3];
#pragma unroll
	for (int c = 0; c < 3; c++) {
		const float lit = SAMECOL ? a.lightCol[0][c] * (kdo[c] * S1[0] + f0[c] * S2[0] + omf0[c] * S3[0])
		                          : (kdo[c] * S1[c] + f0[c] * S2[c] + omf0[c] * S3[c]);
		acc[c] = 10.0f * lit + 0.03f * alb[c] * S0;
	}
	return vkr_pack_colour_lds(sT, acc[0], acc[1], acc[2], S0);
}

// every light the same colour (the reference's braziers): the per-channel work leaves the light loop
inline bool vkr_lights_share_colour(const LightArgs& a)
{
	bool same = true;
	for (int i = 1; i < VKR_NUM_LIGHTS_K; i++)
		for (int c = 0; c < 3; c++) same = same && a.lightCol[i][c] == a.lightCol[0][c];
	return same;
}
