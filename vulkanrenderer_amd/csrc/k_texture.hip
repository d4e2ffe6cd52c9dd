// k_texture.hip — mip-chain generation on the GPU.
//
// Replaces the vkCmdBlitImage(..., VK_FILTER_LINEAR) loop of
// load_image_texture2d (utils/vkimage.cpp:163-216): level n is a linear blit
// of level n-1 — destination texel centres mapped into source space, bilinear
// with clamp-to-edge, filtered in linear space for sRGB formats and
// re-encoded.  Compiled with -ffp-contract=off: texel values feed the alpha
// test and the integer outputs.
#include "vkr_kernels.h"

__device__ __forceinline__ void decode_texel(uint32_t p, int srgb, const float* decode, float out[4])
{
	if (srgb) {
		out[0] = decode[p & 0xffu]; out[1] = decode[(p >> 8) & 0xffu]; out[2] = decode[(p >> 16) & 0xffu];
	} else {
		out[0] = (float)(p & 0xffu) / 255.0f; out[1] = (float)((p >> 8) & 0xffu) / 255.0f;
		out[2] = (float)((p >> 16) & 0xffu) / 255.0f;
	}
	out[3] = (float)(p >> 24) / 255.0f;
}

// One launch builds the same level of many textures: blockIdx.z picks the job.  The grid covers the largest
// destination of the batch; blocks beyond a smaller one leave at once.
__global__ __launch_bounds__(256) void k_mip_level(uint32_t* texels, const MipJob* __restrict__ jobs, const float* __restrict__ decode,
                                                    const float* __restrict__ thresh)
{
	const MipJob jb = jobs[blockIdx.z];
	const uint32_t sw = jb.sw, sh = jb.sh, dw = jb.dw, dh = jb.dh;
	const int srgb = jb.srgb;
	const uint32_t i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
	if (i >= dw || j >= dh) return;
	const float scx = (float)sw / (float)dw, scy = (float)sh / (float)dh;
	const float sv = ((float)j + 0.5f) * scy - 0.5f;
	const float su = ((float)i + 0.5f) * scx - 0.5f;
	const float y0f = floorf(sv), x0f = floorf(su);
	const float fy = sv - y0f, fx = su - x0f;
	int y0 = (int)y0f, y1 = y0 + 1, x0 = (int)x0f, x1 = x0 + 1;
	y0 = min(max(y0, 0), (int)sh - 1); y1 = min(max(y1, 0), (int)sh - 1);
	x0 = min(max(x0, 0), (int)sw - 1); x1 = min(max(x1, 0), (int)sw - 1);
	const uint32_t* src = texels + jb.srcOff;
	float t00[4], t10[4], t01[4], t11[4];
	decode_texel(src[(size_t)y0 * sw + x0], srgb, decode, t00);
	decode_texel(src[(size_t)y0 * sw + x1], srgb, decode, t10);
	decode_texel(src[(size_t)y1 * sw + x0], srgb, decode, t01);
	decode_texel(src[(size_t)y1 * sw + x1], srgb, decode, t11);
	uint32_t out = 0;
#pragma unroll
	for (int c = 0; c < 4; c++) {
		float top = vkr_lerp(t00[c], t10[c], fx);
		float bot = vkr_lerp(t01[c], t11[c], fx);
		float v = vkr_lerp(top, bot, fy);
		uint32_t code = (srgb && c < 3) ? vkr_srgb_encode(thresh, v) : vkr_unorm_encode(v);
		out |= code << (8 * c);
	}
	texels[jb.dstOff + (size_t)j * dw + i] = out;
}

void vkr_launch_mip_levels(hipStream_t s, uint32_t* texels, const MipJob* jobs, uint32_t count, uint32_t maxW, uint32_t maxH,
                           const float* srgbDecode, const float* srgbThresh)
{
	// grid.z is limited to 65535 blocks
	for (uint32_t first = 0; first < count; first += 65535u) {
		const uint32_t n = count - first < 65535u ? count - first : 65535u;
		hipLaunchKernelGGL(k_mip_level, dim3((maxW + 15) / 16, (maxH + 15) / 16, n), dim3(256), 0, s, texels, jobs + first, srgbDecode, srgbThresh);
	}
}

// ---------------------------------------------------------------------------
// TBN matrices of default.vert:57-81, decoded once per vertex at scene
// creation instead of three times per shaded pixel: 3 float4 per vertex
// holding the 9 column-major entries.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tbn_table(const uint32_t* __restrict__ tcomp, float4* __restrict__ out, uint32_t n)
{
	const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= n) return;
	float m[9];
	vkr_decode_tbn(tcomp[v], m);
	out[3 * (size_t)v + 0] = make_float4(m[0], m[1], m[2], m[3]);
	out[3 * (size_t)v + 1] = make_float4(m[4], m[5], m[6], m[7]);
	out[3 * (size_t)v + 2] = make_float4(m[8], 0.0f, 0.0f, 0.0f);
}

void vkr_launch_tbn_table(hipStream_t s, const uint32_t* tcomp, float4* out, uint32_t nVert)
{
	if (!nVert) return;
	hipLaunchKernelGGL(k_tbn_table, dim3((nVert + 255) / 256), dim3(256), 0, s, tcomp, out, nVert);
}

// ---------------------------------------------------------------------------
// Alpha bytes of one texture's whole mip chain (levels are stored back to
// back), for the alpha test of the rasteriser.
// ---------------------------------------------------------------------------
// Both kernels below walk a list of spans (blockIdx.y picks the span, blockIdx.x strides over its texels), so that one
// launch serves every texture of the scene.
__global__ __launch_bounds__(256) void k_extract_alpha(const uint32_t* __restrict__ texels, uint8_t* __restrict__ out,
                                                        const SpanJob* __restrict__ jobs)
{
	const SpanJob jb = jobs[blockIdx.y];
	const uint32_t* src = texels + jb.a;
	uint8_t* dst = out + jb.out;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < jb.n; i += gridDim.x * blockDim.x) dst[i] = (uint8_t)(src[i] >> 24);
}

static uint32_t span_blocks(uint32_t maxN) { return (maxN + 255) / 256 < 4096u ? (maxN + 255) / 256 : 4096u; }

void vkr_launch_extract_alpha(hipStream_t s, const uint32_t* texels, uint8_t* out, const SpanJob* jobs, uint32_t count, uint32_t maxN)
{
	if (!maxN) return;
	for (uint32_t first = 0; first < count; first += 65535u) // grid.y is limited to 65535 blocks
		hipLaunchKernelGGL(k_extract_alpha, dim3(span_blocks(maxN), count - first < 65535u ? count - first : 65535u), dim3(256), 0, s,
		                   texels, out, jobs + first);
}

// ---------------------------------------------------------------------------
// Repacked chain: the colour bytes of one chain and the first byte of another
// (same size, levels stored back to back) in one texel.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_chain(uint32_t* texels, const SpanJob* __restrict__ jobs)
{
	const SpanJob jb = jobs[blockIdx.y];
	const uint32_t* rgb = texels + jb.rgb;
	const uint32_t* a = texels + jb.a;
	uint32_t* out = texels + jb.out;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < jb.n; i += gridDim.x * blockDim.x)
		out[i] = (rgb[i] & 0x00ffffffu) | (a[i] << 24);
}

void vkr_launch_pack_chains(hipStream_t s, uint32_t* texels, const SpanJob* jobs, uint32_t count, uint32_t maxN)
{
	if (!maxN) return;
	for (uint32_t first = 0; first < count; first += 65535u)
		hipLaunchKernelGGL(k_pack_chain, dim3(span_blocks(maxN), count - first < 65535u ? count - first : 65535u), dim3(256), 0, s,
		                   texels, jobs + first);
}

// ---------------------------------------------------------------------------
// TriShade table: one record per triangle (submission order A).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tri_shade(SceneDev sc, TriShade* __restrict__ out)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= sc.nTri) return;
	const uint32_t idx[3] = { sc.tri[3 * (size_t)t + 0], sc.tri[3 * (size_t)t + 1], sc.tri[3 * (size_t)t + 2] };
	TriShade r;
	r.t0 = make_float2(sc.uv[2 * (size_t)idx[0]], sc.uv[2 * (size_t)idx[0] + 1]);
	r.t1 = make_float2(sc.uv[2 * (size_t)idx[1]], sc.uv[2 * (size_t)idx[1] + 1]);
	r.t2 = make_float2(sc.uv[2 * (size_t)idx[2]], sc.uv[2 * (size_t)idx[2] + 1]);
	r.mat = sc.triMat[t];
	const MatInfo& mi = sc.mat[r.mat];
	r.nmOff0 = r.arOff0 = 0u; r.wh0 = 0u; r.levels = 0; r.srgbAR = 0u;
	if (mi.packNM != 0xffffffffu) {
		const TexInfo& tn = sc.tex[mi.packNM];
		const TexInfo& ta = sc.tex[mi.packAR];
		r.nmOff0 = tn.offset[0]; r.arOff0 = ta.offset[0];
		r.wh0 = tn.wh0; r.levels = tn.levels; r.srgbAR = (uint32_t)ta.srgb;
	}
	for (int v = 0; v < 3; v++) {
		const float4 c0 = sc.tbn[3 * (size_t)idx[v] + 0], c1 = sc.tbn[3 * (size_t)idx[v] + 1], c2 = sc.tbn[3 * (size_t)idx[v] + 2];
		r.tbn[v][0] = c0.x; r.tbn[v][1] = c0.y; r.tbn[v][2] = c0.z; r.tbn[v][3] = c0.w;
		r.tbn[v][4] = c1.x; r.tbn[v][5] = c1.y; r.tbn[v][6] = c1.z; r.tbn[v][7] = c1.w;
		r.tbn[v][8] = c2.x;
		for (int k = 0; k < 3; k++) r.pos[v][k] = sc.pos[3 * (size_t)idx[v] + k];
	}
	out[t] = r;
}

void vkr_launch_tri_shade(hipStream_t s, const SceneDev& sc, TriShade* out)
{
	if (!sc.nTri) return;
	hipLaunchKernelGGL(k_tri_shade, dim3((sc.nTri + 255) / 256), dim3(256), 0, s, sc, out);
}

// ---------------------------------------------------------------------------
// AlphaAux table: one record per triangle (submission order A).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_alpha_aux(SceneDev sc, AlphaAux* __restrict__ out)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= sc.nTri) return;
	const uint32_t i0 = sc.tri[3 * (size_t)t + 0], i1 = sc.tri[3 * (size_t)t + 1], i2 = sc.tri[3 * (size_t)t + 2];
	const TexInfo& tex = sc.tex[sc.mat[sc.triMat[t]].tex[3]];
	AlphaAux x;
	x.t0 = make_float2(sc.uv[2 * (size_t)i0], sc.uv[2 * (size_t)i0 + 1]);
	x.t1 = make_float2(sc.uv[2 * (size_t)i1], sc.uv[2 * (size_t)i1 + 1]);
	x.t2 = make_float2(sc.uv[2 * (size_t)i2], sc.uv[2 * (size_t)i2 + 1]);
	x.off0 = tex.alphaOff; x.wh0 = tex.wh0; x.levels = tex.levels;
	x._pad[0] = x._pad[1] = x._pad[2] = 0u;
	out[t] = x;
}

void vkr_launch_alpha_aux(hipStream_t s, const SceneDev& sc, AlphaAux* out)
{
	if (!sc.nTri) return;
	hipLaunchKernelGGL(k_alpha_aux, dim3((sc.nTri + 255) / 256), dim3(256), 0, s, sc, out);
}
