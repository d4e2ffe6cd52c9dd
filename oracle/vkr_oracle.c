/*
 * vkr_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar, single-threaded CPU restatement of what the reference renders in
 * one frame: its GLSL programs (main/shaders/*.vert, *.frag) plus the Vulkan
 * fixed-function behaviour its pipelines rely on (main/main.cpp:1539-3452)
 * and the mode dispatch / draw order of record_commands
 * (main/main.cpp:3709-4324).  It draws the way the reference does: triangle by
 * triangle in submission order, depth-testing and shading every fragment.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file.  The HIP renderer never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" for everything the Vulkan driver decides
 * (rasterisation rules, LOD, filtering, sRGB, half conversion, depth bias):
 * the reference holds no golden images and neither a Vulkan ICD nor a GLSL
 * compiler exists in this environment.  Those parts follow the Vulkan 1.3
 * specification text.  The parts pinned by the reference's own compilable
 * sources (glm matrix arithmetic, the baked-model file format, the TBN
 * quaternion packing) are checked against fixtures generated from those
 * sources (tests/golden/, generator oracle/ref_fixtures.cpp).
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -shared -fPIC (see oracle/Makefile).
 * Floating-point contraction must stay off: the integer outputs (coverage,
 * depth bits, mip level, counters) depend on the exact operation order.
 */
#include "../include/vkr_renderer.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORA_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* bit helpers                                                              */
/* ------------------------------------------------------------------------ */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* float -> IEEE half, round to nearest even (RGBA16F attachment writes,
 * main/main.cpp:1852,1862). */
static uint16_t float_to_half(float f)
{
	uint32_t x = f2u(f);
	uint32_t sign = (x >> 16) & 0x8000u;
	uint32_t ax = x & 0x7fffffffu;
	if (ax >= 0x7f800000u) /* inf / nan */
		return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
	if (ax >= 0x477ff000u) /* rounds to >= 65520 -> inf */
		return (uint16_t)(sign | 0x7c00u);
	if (ax < 0x33000001u) /* < 2^-25 (or == 2^-25: tie to even -> 0) */
		return (uint16_t)sign;
	if (ax < 0x38800000u) { /* subnormal half */
		uint32_t e = ax >> 23;                        /* biased exponent, 102..112 */
		uint32_t m = (ax & 0x7fffffu) | 0x800000u;    /* 24-bit significand */
		uint32_t shift = 126u - e;                    /* 14..24 */
		uint32_t h = m >> shift;
		uint32_t rem = m & ((1u << shift) - 1u);
		uint32_t half = 1u << (shift - 1u);
		if (rem > half || (rem == half && (h & 1u))) h++;
		return (uint16_t)(sign | h);
	}
	{
		uint32_t h = ((ax - 0x38000000u) >> 13);
		uint32_t rem = ax & 0x1fffu;
		if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
		return (uint16_t)(sign | h);
	}
}

static float half_to_float(uint16_t h)
{
	uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
	uint32_t e = (h >> 10) & 0x1fu;
	uint32_t m = h & 0x3ffu;
	if (e == 0) {
		if (m == 0) return u2f(sign);
		/* subnormal: m * 2^-24 */
		float v = (float)m * 5.9604644775390625e-8f;
		return (sign ? -v : v);
	}
	if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
	return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

/* ------------------------------------------------------------------------ */
/* sRGB tables (Vulkan spec "sRGB EOTF"; encode as threshold search so that  */
/* no pow() is evaluated per pixel and CPU/GPU agree bit for bit)            */
/* ------------------------------------------------------------------------ */
static float g_srgb_decode[256];
static float g_srgb_thresh[255]; /* linear value at which code k becomes k+1 */
static int g_tables_ready = 0;

static double srgb_eotf(double c)
{
	return (c <= 0.04045) ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4);
}

static void init_tables(void)
{
	if (g_tables_ready) return;
	for (int i = 0; i < 256; i++) g_srgb_decode[i] = (float)srgb_eotf((double)i / 255.0);
	for (int i = 0; i < 255; i++) g_srgb_thresh[i] = (float)srgb_eotf(((double)i + 0.5) / 255.0);
	g_tables_ready = 1;
}

/* linear [0,1] -> sRGB code = number of thresholds <= x */
static uint8_t srgb_encode(float x)
{
	int lo = 0, hi = 255; /* answer in [lo,hi] */
	if (!(x > 0.0f)) return 0;
	while (lo < hi) {
		int mid = (lo + hi) >> 1; /* test threshold mid: code >= mid+1 ? */
		if (x >= g_srgb_thresh[mid]) lo = mid + 1; else hi = mid;
	}
	return (uint8_t)lo;
}

static uint8_t unorm_encode(float x)
{
	if (!(x > 0.0f)) return 0;
	if (x >= 1.0f) return 255;
	return (uint8_t)(int)floorf(x * 255.0f + 0.5f);
}

/* ------------------------------------------------------------------------ */
/* scene                                                                     */
/* ------------------------------------------------------------------------ */
#define ORA_MAX_LEVELS 16

typedef struct {
	uint32_t w[ORA_MAX_LEVELS], h[ORA_MAX_LEVELS];
	uint8_t* data[ORA_MAX_LEVELS];
	int levels;
	int srgb;
} OraTexture;

typedef struct {
	uint32_t material, V, I;
	float* pos;       /* V*3 */
	float* uv;        /* V*2 */
	uint32_t* tcomp;  /* V */
	uint32_t* idx;    /* I */
	int alpha;        /* hasAlphaMask, main/main.cpp:1080-1081 */
} OraMesh;

typedef struct { uint32_t mesh, tri; } OraPrim;

typedef struct ora_scene {
	uint32_t nTex, nMat, nMesh;
	OraTexture* tex;
	vkr_material_desc* mat;
	OraMesh* mesh;
	/* submission orders */
	uint64_t nPrim, nPrimOpaque;
	OraPrim* orderA; /* opaque meshes in file order, then alpha-masked ones (main/main.cpp:3873-3934) */
	OraPrim* orderB; /* all meshes in file order (main/main.cpp:4111-4126, 4164-4190) */
} ora_scene;

/* levels = 32 - clz(w|h)  (compute_mip_level_count, utils/vkimage.cpp:289-294) */
static int mip_level_count(uint32_t w, uint32_t h)
{
	uint32_t bits = w | h;
	int n = 0;
	while (bits) { n++; bits >>= 1; }
	return n;
}

static inline float lerpf(float a, float b, float t) { return fmaf(t, b - a, a); }

static inline void texel_decode(const OraTexture* t, int level, int x, int y, float out[4])
{
	const uint8_t* p = t->data[level] + ((size_t)y * t->w[level] + (size_t)x) * 4;
	if (t->srgb) {
		out[0] = g_srgb_decode[p[0]];
		out[1] = g_srgb_decode[p[1]];
		out[2] = g_srgb_decode[p[2]];
	} else {
		out[0] = (float)p[0] / 255.0f;
		out[1] = (float)p[1] / 255.0f;
		out[2] = (float)p[2] / 255.0f;
	}
	out[3] = (float)p[3] / 255.0f;
}

/* One vkCmdBlitImage(..., VK_FILTER_LINEAR) from level n-1 to level n
 * (utils/vkimage.cpp:165-197): destination texel centre mapped to source
 * space, bilinear with clamp-to-edge, filtered in linear space, re-encoded. */
static void build_mip_level(OraTexture* t, int level)
{
	uint32_t sw = t->w[level - 1], sh = t->h[level - 1];
	uint32_t dw = t->w[level], dh = t->h[level];
	float scx = (float)sw / (float)dw, scy = (float)sh / (float)dh;
	for (uint32_t j = 0; j < dh; j++) {
		float sv = ((float)j + 0.5f) * scy - 0.5f;
		float y0f = floorf(sv);
		float fy = sv - y0f;
		int y0 = (int)y0f, y1 = y0 + 1;
		if (y0 < 0) y0 = 0; if (y0 > (int)sh - 1) y0 = (int)sh - 1;
		if (y1 < 0) y1 = 0; if (y1 > (int)sh - 1) y1 = (int)sh - 1;
		for (uint32_t i = 0; i < dw; i++) {
			float su = ((float)i + 0.5f) * scx - 0.5f;
			float x0f = floorf(su);
			float fx = su - x0f;
			int x0 = (int)x0f, x1 = x0 + 1;
			if (x0 < 0) x0 = 0; if (x0 > (int)sw - 1) x0 = (int)sw - 1;
			if (x1 < 0) x1 = 0; if (x1 > (int)sw - 1) x1 = (int)sw - 1;
			float t00[4], t10[4], t01[4], t11[4];
			texel_decode(t, level - 1, x0, y0, t00);
			texel_decode(t, level - 1, x1, y0, t10);
			texel_decode(t, level - 1, x0, y1, t01);
			texel_decode(t, level - 1, x1, y1, t11);
			uint8_t* d = t->data[level] + ((size_t)j * dw + i) * 4;
			for (int c = 0; c < 4; c++) {
				float top = lerpf(t00[c], t10[c], fx);
				float bot = lerpf(t01[c], t11[c], fx);
				float v = lerpf(top, bot, fy);
				d[c] = (t->srgb && c < 3) ? srgb_encode(v) : unorm_encode(v);
			}
		}
	}
}

ORA_API void ora_scene_destroy(ora_scene* s)
{
	if (!s) return;
	for (uint32_t i = 0; i < s->nTex; i++)
		for (int l = 0; l < s->tex[i].levels; l++) free(s->tex[i].data[l]);
	for (uint32_t i = 0; i < s->nMesh; i++) {
		free(s->mesh[i].pos); free(s->mesh[i].uv); free(s->mesh[i].tcomp); free(s->mesh[i].idx);
	}
	free(s->tex); free(s->mat); free(s->mesh); free(s->orderA); free(s->orderB);
	free(s);
}

static void* dupmem(const void* p, size_t n)
{
	void* r = malloc(n ? n : 1);
	if (r && n) memcpy(r, p, n);
	return r;
}

ORA_API ora_scene* ora_scene_create(const vkr_scene_desc* d)
{
	init_tables();
	ora_scene* s = (ora_scene*)calloc(1, sizeof(ora_scene));
	s->nTex = d->textureCount; s->nMat = d->materialCount; s->nMesh = d->meshCount;
	s->tex = (OraTexture*)calloc(s->nTex ? s->nTex : 1, sizeof(OraTexture));
	s->mat = (vkr_material_desc*)dupmem(d->materials, sizeof(vkr_material_desc) * s->nMat);
	s->mesh = (OraMesh*)calloc(s->nMesh ? s->nMesh : 1, sizeof(OraMesh));
	for (uint32_t i = 0; i < s->nTex; i++) {
		const vkr_texture_desc* td = &d->textures[i];
		OraTexture* t = &s->tex[i];
		t->srgb = td->space == 1;
		t->levels = mip_level_count(td->width, td->height);
		uint32_t w = td->width, h = td->height;
		for (int l = 0; l < t->levels; l++) {
			t->w[l] = w; t->h[l] = h;
			t->data[l] = (uint8_t*)malloc((size_t)w * h * 4);
			if (l == 0) memcpy(t->data[0], td->rgba8, (size_t)w * h * 4);
			else build_mip_level(t, l);
			w >>= 1; if (w == 0) w = 1;
			h >>= 1; if (h == 0) h = 1;
		}
	}
	uint64_t nPrim = 0, nOpaque = 0;
	for (uint32_t i = 0; i < s->nMesh; i++) {
		const vkr_mesh_desc* md = &d->meshes[i];
		OraMesh* m = &s->mesh[i];
		m->material = md->materialId; m->V = md->vertexCount; m->I = md->indexCount;
		m->pos = (float*)dupmem(md->positions, sizeof(float) * 3 * m->V);
		m->uv = (float*)dupmem(md->texcoords, sizeof(float) * 2 * m->V);
		m->tcomp = (uint32_t*)dupmem(md->tangentsComp, sizeof(uint32_t) * m->V);
		m->idx = (uint32_t*)dupmem(md->indices, sizeof(uint32_t) * m->I);
		m->alpha = s->mat[m->material].alphaMaskTextureId != VKR_NO_TEXTURE;
		nPrim += m->I / 3;
		if (!m->alpha) nOpaque += m->I / 3;
	}
	s->nPrim = nPrim; s->nPrimOpaque = nOpaque;
	s->orderA = (OraPrim*)malloc(sizeof(OraPrim) * (nPrim ? nPrim : 1));
	s->orderB = (OraPrim*)malloc(sizeof(OraPrim) * (nPrim ? nPrim : 1));
	uint64_t a = 0, b = 0;
	for (int pass = 0; pass < 2; pass++)
		for (uint32_t i = 0; i < s->nMesh; i++) {
			if (s->mesh[i].alpha != pass) continue;
			for (uint32_t t = 0; t < s->mesh[i].I / 3; t++) { s->orderA[a].mesh = i; s->orderA[a].tri = t; a++; }
		}
	for (uint32_t i = 0; i < s->nMesh; i++)
		for (uint32_t t = 0; t < s->mesh[i].I / 3; t++) { s->orderB[b].mesh = i; s->orderB[b].tri = t; b++; }
	return s;
}

ORA_API int ora_texture_level(const ora_scene* s, uint32_t tex, uint32_t level, uint8_t* out, uint32_t* w, uint32_t* h)
{
	if (tex >= s->nTex || (int)level >= s->tex[tex].levels) return 1;
	const OraTexture* t = &s->tex[tex];
	*w = t->w[level]; *h = t->h[level];
	if (out) memcpy(out, t->data[level], (size_t)t->w[level] * t->h[level] * 4);
	return 0;
}

/* ------------------------------------------------------------------------ */
/* texture sampling: sampler of utils/vkutil.cpp:179-196 (LINEAR/LINEAR/     */
/* mip LINEAR, REPEAT, lod [0,inf), bias 0) following Vulkan 1.3 "Texel       */
/* filtering".                                                               */
/* ------------------------------------------------------------------------ */

/* log2 of m in [1,2) from the atanh series, only +,*,/,fma: identical on any
 * IEEE machine. */
static float log2_mantissa(float m)
{
	float t = m - 1.0f;
	float s = t / (2.0f + t);
	float s2 = s * s;
	float p = 0.09090909090909091f;          /* 1/11 */
	p = fmaf(p, s2, 0.1111111111111111f);    /* 1/9 */
	p = fmaf(p, s2, 0.14285714285714285f);   /* 1/7 */
	p = fmaf(p, s2, 0.2f);
	p = fmaf(p, s2, 0.3333333333333333f);
	p = fmaf(p, s2, 1.0f);
	float r = (s * p) * 2.8853900817779268f; /* 2/ln 2 */
	if (r > 1.0f) r = 1.0f;
	return r;
}

/* Splits lambda = 0.5*log2(rho^2), clamped to [0,q], into integer level d and
 * fraction f in [0,1]. */
static void lod_split(float rho2, int q, int* d, float* f)
{
	*d = 0; *f = 0.0f;
	if (!(rho2 > 1.0f)) return;
	uint32_t bits = f2u(rho2);
	int e = (int)((bits >> 23) & 0xffu) - 127; /* >= 0 */
	float m = u2f((bits & 0x7fffffu) | 0x3f800000u);
	float p = log2_mantissa(m);
	float fr = fmaf(p, 0.5f, (e & 1) ? 0.5f : 0.0f);
	int lv = e >> 1;
	if (lv >= q) { *d = q; *f = 0.0f; return; }
	*d = lv; *f = fr;
}

static inline int wrap_repeat(int i, int n)
{
	int r = i % n;
	return r < 0 ? r + n : r;
}

static void sample_bilinear(const OraTexture* t, int level, float u, float v, float out[4])
{
	int w = (int)t->w[level], h = (int)t->h[level];
	float x = u * (float)w - 0.5f;
	float y = v * (float)h - 0.5f;
	if (!(fabsf(x) < 1.0e9f)) x = 0.0f;
	if (!(fabsf(y) < 1.0e9f)) y = 0.0f;
	float x0f = floorf(x), y0f = floorf(y);
	float fx = x - x0f, fy = y - y0f;
	int x0 = wrap_repeat((int)x0f, w), x1 = wrap_repeat((int)x0f + 1, w);
	int y0 = wrap_repeat((int)y0f, h), y1 = wrap_repeat((int)y0f + 1, h);
	float t00[4], t10[4], t01[4], t11[4];
	texel_decode(t, level, x0, y0, t00);
	texel_decode(t, level, x1, y0, t10);
	texel_decode(t, level, x0, y1, t01);
	texel_decode(t, level, x1, y1, t11);
	for (int c = 0; c < 4; c++)
		out[c] = lerpf(lerpf(t00[c], t10[c], fx), lerpf(t01[c], t11[c], fx), fy);
}

/* derivatives: d[0]=du/dx d[1]=dv/dx d[2]=du/dy d[3]=dv/dy */
static float rho2_for(const OraTexture* t, const float d[4])
{
	float tw = (float)t->w[0], th = (float)t->h[0];
	float mx = d[0] * tw, my = d[1] * th;
	float nx = d[2] * tw, ny = d[3] * th;
	float rx = fmaf(my, my, mx * mx);
	float ry = fmaf(ny, ny, nx * nx);
	return (rx > ry) ? rx : ry; /* NaN in rx picks ry; NaN in ry picks ry -> lod_split yields 0 */
}

static void sample_trilinear(const OraTexture* t, float u, float v, const float d[4], float out[4])
{
	int lv; float fr;
	lod_split(rho2_for(t, d), t->levels - 1, &lv, &fr);
	if (fr == 0.0f) { sample_bilinear(t, lv, u, v, out); return; }
	float a[4], b[4];
	sample_bilinear(t, lv, u, v, a);
	sample_bilinear(t, lv + 1, u, v, b);
	for (int c = 0; c < 4; c++) out[c] = lerpf(a[c], b[c], fr);
}

/* ------------------------------------------------------------------------ */
/* glm restatements for the uniform blocks                                   */
/* ------------------------------------------------------------------------ */
/* m[c*4+r], column-major like glm::mat4 */

static void mat4_mul(const float a[16], const float b[16], float out[16])
{
	/* glm non-SIMD mul4x4: tmp = A0*b.x; tmp += A1*b.y; tmp += A2*b.z; tmp += A3*b.w
	 * (third_party/glm/include/glm/detail/type_mat4x4.inl:681-700) */
	float r[16];
	for (int c = 0; c < 4; c++)
		for (int i = 0; i < 4; i++) {
			float t = a[0 * 4 + i] * b[c * 4 + 0];
			t = t + a[1 * 4 + i] * b[c * 4 + 1];
			t = t + a[2 * 4 + i] * b[c * 4 + 2];
			t = t + a[3 * 4 + i] * b[c * 4 + 3];
			r[c * 4 + i] = t;
		}
	memcpy(out, r, sizeof r);
}

/* glm::inverse for mat4 (third_party/glm/include/glm/detail/func_matrix.inl:388-446):
 * 2x2 sub-determinants, cofactor columns with alternating signs, divided by
 * the determinant taken along the first column. */
static void mat4_inverse(const float m[16], float out[16])
{
#define M(c, r) m[(c) * 4 + (r)]
	float c00 = M(2,2) * M(3,3) - M(3,2) * M(2,3);
	float c02 = M(1,2) * M(3,3) - M(3,2) * M(1,3);
	float c03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
	float c04 = M(2,1) * M(3,3) - M(3,1) * M(2,3);
	float c06 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
	float c07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
	float c08 = M(2,1) * M(3,2) - M(3,1) * M(2,2);
	float c10 = M(1,1) * M(3,2) - M(3,1) * M(1,2);
	float c11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
	float c12 = M(2,0) * M(3,3) - M(3,0) * M(2,3);
	float c14 = M(1,0) * M(3,3) - M(3,0) * M(1,3);
	float c15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
	float c16 = M(2,0) * M(3,2) - M(3,0) * M(2,2);
	float c18 = M(1,0) * M(3,2) - M(3,0) * M(1,2);
	float c19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
	float c20 = M(2,0) * M(3,1) - M(3,0) * M(2,1);
	float c22 = M(1,0) * M(3,1) - M(3,0) * M(1,1);
	float c23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
	float f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 }, f2[4] = { c08, c08, c10, c11 };
	float f3[4] = { c12, c12, c14, c15 }, f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
	float v0[4] = { M(1,0), M(0,0), M(0,0), M(0,0) }, v1[4] = { M(1,1), M(0,1), M(0,1), M(0,1) };
	float v2[4] = { M(1,2), M(0,2), M(0,2), M(0,2) }, v3[4] = { M(1,3), M(0,3), M(0,3), M(0,3) };
	static const float sa[4] = { 1, -1, 1, -1 }, sb[4] = { -1, 1, -1, 1 };
	float inv[16];
	for (int i = 0; i < 4; i++) {
		inv[0 * 4 + i] = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i]) * sa[i];
		inv[1 * 4 + i] = ((v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i]) * sb[i];
		inv[2 * 4 + i] = ((v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i]) * sa[i];
		inv[3 * 4 + i] = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb[i];
	}
	float d0 = M(0,0) * inv[0 * 4 + 0], d1 = M(0,1) * inv[1 * 4 + 0];
	float d2 = M(0,2) * inv[2 * 4 + 0], d3 = M(0,3) * inv[3 * 4 + 0];
	float det = (d0 + d1) + (d2 + d3);
	float ood = 1.0f / det;
	for (int i = 0; i < 16; i++) out[i] = inv[i] * ood;
#undef M
}

/* glm::perspectiveRH_ZO (third_party/glm/include/glm/ext/matrix_clip_space.inl:233-246) */
static void perspective_rh_zo(float fovy, float aspect, float zn, float zf, float out[16])
{
	float th = tanf(fovy / 2.0f);
	memset(out, 0, sizeof(float) * 16);
	out[0 * 4 + 0] = 1.0f / (aspect * th);
	out[1 * 4 + 1] = 1.0f / th;
	out[2 * 4 + 2] = zf / (zn - zf);
	out[2 * 4 + 3] = -1.0f;
	out[3 * 4 + 2] = -(zf * zn) / (zf - zn);
}

static float glm_dot3(const float a[3], const float b[3])
{
	float t0 = a[0] * b[0], t1 = a[1] * b[1], t2 = a[2] * b[2];
	return t0 + t1 + t2;
}
static void glm_normalize3(const float a[3], float out[3])
{
	float inv = 1.0f / sqrtf(glm_dot3(a, a));
	out[0] = a[0] * inv; out[1] = a[1] * inv; out[2] = a[2] * inv;
}
static void glm_cross(const float x[3], const float y[3], float out[3])
{
	out[0] = x[1] * y[2] - y[1] * x[2];
	out[1] = x[2] * y[0] - y[2] * x[0];
	out[2] = x[0] * y[1] - y[0] * x[1];
}

/* glm::lookAtRH (third_party/glm/include/glm/ext/matrix_transform.inl:153-173) */
static void look_at_rh(const float eye[3], const float center[3], const float up[3], float out[16])
{
	float d[3] = { center[0] - eye[0], center[1] - eye[1], center[2] - eye[2] };
	float f[3], s[3], u[3], t[3];
	glm_normalize3(d, f);
	glm_cross(f, up, t);
	glm_normalize3(t, s);
	glm_cross(s, f, u);
	memset(out, 0, sizeof(float) * 16);
	out[15] = 1.0f;
	out[0 * 4 + 0] = s[0]; out[1 * 4 + 0] = s[1]; out[2 * 4 + 0] = s[2];
	out[0 * 4 + 1] = u[0]; out[1 * 4 + 1] = u[1]; out[2 * 4 + 1] = u[2];
	out[0 * 4 + 2] = -f[0]; out[1 * 4 + 2] = -f[1]; out[2 * 4 + 2] = -f[2];
	out[3 * 4 + 0] = -glm_dot3(s, eye);
	out[3 * 4 + 1] = -glm_dot3(u, eye);
	out[3 * 4 + 2] = glm_dot3(f, eye);
}

/* update_scene_uniforms, main/main.cpp:3681-3693; kCameraFov = 60 degrees
 * converted by labutils::deg_to_rad (utils/angle.hpp:11-14), near 0.1, far 100
 * (main/main.cpp:75-78). */
ORA_API void ora_update_scene_uniforms(vkr_scene_uniform* u, uint32_t w, uint32_t h, const float cam2world[16])
{
	const float pi = 3.14159265358979323846f;
	float fov = 60.0f * pi / 180.0f;
	float aspect = (float)w / (float)h;
	perspective_rh_zo(fov, aspect, 0.1f, 100.0f, u->projection);
	u->projection[1 * 4 + 1] *= -1.0f;
	mat4_inverse(cam2world, u->camera);
	mat4_mul(u->projection, u->camera, u->projCam);
	u->camPos[0] = cam2world[12]; u->camPos[1] = cam2world[13]; u->camPos[2] = cam2world[14]; u->camPos[3] = 1.0f;
}

/* update_depth_mvp_uniforms, main/main.cpp:3699-3707.  lut::Radians(90.0f)
 * constructs an angle of 90 RADIANS (utils/angle.hpp:65), so the light
 * frustum's fovy is 90 rad; reproduced literally. */
ORA_API void ora_update_depth_mvp_uniforms(vkr_depth_mvp* d, const float lightPos[4])
{
	float proj[16], view[16];
	perspective_rh_zo(90.0f, 1.0f, 0.1f, 100.0f, proj);
	proj[1 * 4 + 1] *= -1.0f;
	float eye[3] = { lightPos[0], lightPos[1], lightPos[2] };
	float center[3] = { 0.0f, 0.0f, -48.0f }, up[3] = { 0.0f, 1.0f, 0.0f };
	look_at_rh(eye, center, up, view);
	mat4_mul(proj, view, d->depthMVP);
}

/* main/main.cpp:1313-1352 */
ORA_API void ora_default_lights(vkr_light_uniform* fwd, vkr_multiple_lights* ml)
{
	static const float P[21][3] = {
		{ 8.0377f, -1.1000f, -15.8845f }, { 5.6632f, -1.1000f, -12.2500f }, { 2.1255f, -1.1000f, -10.1000f },
		{ -2.0867f, -1.1000f, -10.2500f }, { -5.7367f, -1.1000f, -12.2000f }, { -7.8244f, -1.1000f, -15.8500f },
		{ 3.0632f, -1.1000f, -25.8500f }, { -3.0632f, -1.1000f, -25.8500f }, { -7.4867f, -1.1000f, -36.0500f },
		{ -2.4367f, -3.1000f, -46.5500f }, { 2.3632f, -3.1000f, -46.5500f }, { -0.0514f, -3.9000f, -49.6000f },
		{ -7.2367f, -3.1000f, -61.2500f }, { 7.1088f, -3.1000f, -61.2500f }, { -2.0867f, -3.1000f, -66.5500f },
		{ 2.0867f, -3.1000f, -66.5500f }, { -2.0867f, -3.1000f, -68.9000f }, { 2.0867f, -3.1000f, -68.9000f },
		{ 0.0000f, -4.1000f, -76.0000f }, { -1.6867f, -3.1000f, -88.8500f }, { 1.6867f, -3.1000f, -88.8500f } };
	if (fwd) {
		fwd->lightPos[0] = -0.2972f; fwd->lightPos[1] = 7.3100f; fwd->lightPos[2] = -11.9532f; fwd->lightPos[3] = 0.0f;
		fwd->lightColour[0] = fwd->lightColour[1] = fwd->lightColour[2] = fwd->lightColour[3] = 1.0f;
	}
	if (ml)
		for (int i = 0; i < 21; i++) {
			ml->light[i].lightPos[0] = P[i][0]; ml->light[i].lightPos[1] = P[i][1];
			ml->light[i].lightPos[2] = P[i][2]; ml->light[i].lightPos[3] = 0.0f;
			ml->light[i].lightColour[0] = 0.9882f; ml->light[i].lightColour[1] = 0.4549f;
			ml->light[i].lightColour[2] = 0.0196f; ml->light[i].lightColour[3] = 1.0f;
		}
}

ORA_API void ora_make_uniforms(vkr_uniforms* u, uint32_t w, uint32_t h, const float cam2world[16], int32_t debug)
{
	memset(u, 0, sizeof *u);
	ora_update_scene_uniforms(&u->sceneUniforms, w, h, cam2world);
	u->debugUniforms.debug = debug;
	ora_default_lights(&u->lightUniforms, &u->multipleLightsUniform);
	ora_update_depth_mvp_uniforms(&u->depthMVPUniform, u->lightUniforms.lightPos);
}

/* Column c of an identity matrix scaled by k, accumulated the way glm's
 * vec4 operators do it: m[0]*k0 + m[1]*k1 + m[2]*k2 (+ m[3]). */
static void glm_identity_combo(const float k[3], int addLast, float out[4])
{
	for (int r = 0; r < 4; r++) {
		float e0 = (r == 0) ? 1.0f : 0.0f, e1 = (r == 1) ? 1.0f : 0.0f, e2 = (r == 2) ? 1.0f : 0.0f, e3 = (r == 3) ? 1.0f : 0.0f;
		float p0 = e0 * k[0], p1 = e1 * k[1], p2 = e2 * k[2];
		float acc = p0 + p1;
		acc = acc + p2;
		if (addLast) acc = acc + e3;
		out[r] = acc;
	}
}

/* glm::rotate(angle, v) = rotate(mat4(1), angle, v)
 * (third_party/glm/include/glm/gtx/transform.inl:12-15,
 *  third_party/glm/include/glm/ext/matrix_transform.inl:18-46) */
static void glm_rotate(float angle, const float v[3], float out[16])
{
	float c = cosf(angle), s = sinf(angle);
	float axis[3], temp[3], rc[3];
	glm_normalize3(v, axis);
	float omc = 1.0f - c;
	for (int i = 0; i < 3; i++) temp[i] = omc * axis[i];
	rc[0] = c + temp[0] * axis[0];
	rc[1] = temp[0] * axis[1] + s * axis[2];
	rc[2] = temp[0] * axis[2] - s * axis[1];
	glm_identity_combo(rc, 0, out + 0);
	rc[0] = temp[1] * axis[0] - s * axis[2];
	rc[1] = c + temp[1] * axis[1];
	rc[2] = temp[1] * axis[2] + s * axis[0];
	glm_identity_combo(rc, 0, out + 4);
	rc[0] = temp[2] * axis[0] + s * axis[1];
	rc[1] = temp[2] * axis[1] - s * axis[0];
	rc[2] = c + temp[2] * axis[2];
	glm_identity_combo(rc, 0, out + 8);
	out[12] = 0.0f; out[13] = 0.0f; out[14] = 0.0f; out[15] = 1.0f;
}

/* glm::translate(v) = translate(mat4(1), v)
 * (gtx/transform.inl:6-9, ext/matrix_transform.inl:10-15) */
static void glm_translate(float x, float y, float z, float out[16])
{
	float v[3] = { x, y, z };
	memset(out, 0, sizeof(float) * 16);
	out[0] = out[5] = out[10] = 1.0f;
	glm_identity_combo(v, 1, out + 12);
}

/* update_user_state, main/main.cpp:3643-3679; constants main/main.cpp:76-84 */
ORA_API void ora_update_user_state(vkr_user_state* st, float elapsed)
{
	float t[16];
	if (st->inputMap[VKR_INPUT_MOUSING]) {
		if (st->wasMousing) {
			float sens = 0.01f;
			float dx = sens * (st->mouseX - st->previousX);
			float dy = sens * (st->mouseY - st->previousY);
			float ax[3] = { 1.0f, 0.0f, 0.0f }, ay[3] = { 0.0f, 1.0f, 0.0f };
			glm_rotate(-dy, ax, t);
			mat4_mul(st->camera2world, t, st->camera2world);
			glm_rotate(-dx, ay, t);
			mat4_mul(st->camera2world, t, st->camera2world);
		}
		st->previousX = st->mouseX;
		st->previousY = st->mouseY;
		st->wasMousing = 1;
	} else {
		st->wasMousing = 0;
	}
	float move = elapsed * 1.7f;
	move = move * (st->inputMap[VKR_INPUT_FAST] ? 7.0f : 1.0f);
	move = move * (st->inputMap[VKR_INPUT_SLOW] ? 0.05f : 1.0f);
	static const struct { int key; float x, y, z; } dirs[6] = {
		{ VKR_INPUT_FORWARD, 0.f, 0.f, -1.f }, { VKR_INPUT_BACKWARD, 0.f, 0.f, 1.f },
		{ VKR_INPUT_STRAFE_LEFT, -1.f, 0.f, 0.f }, { VKR_INPUT_STRAFE_RIGHT, 1.f, 0.f, 0.f },
		{ VKR_INPUT_LEVITATE, 0.f, 1.f, 0.f }, { VKR_INPUT_SINK, 0.f, -1.f, 0.f } };
	for (int i = 0; i < 6; i++)
		if (st->inputMap[dirs[i].key]) {
			/* the reference passes +-move and literal 0.f components */
			float x = dirs[i].x == 0.f ? 0.f : (dirs[i].x < 0.f ? -move : move);
			float y = dirs[i].y == 0.f ? 0.f : (dirs[i].y < 0.f ? -move : move);
			float z = dirs[i].z == 0.f ? 0.f : (dirs[i].z < 0.f ? -move : move);
			glm_translate(x, y, z, t);
			mat4_mul(st->camera2world, t, st->camera2world);
		}
}

/* ------------------------------------------------------------------------ */
/* TBN quaternion packing                                                    */
/* ------------------------------------------------------------------------ */

/* Decoder of default.vert:57-81 / gBufWrite.vert:44-68 with the
 * A2R10G10B10_UNORM_PACK32 attribute fetch (main/main.cpp:2262-2265). m is the
 * GLSL mat3 in column-major order: m[c*3+r]. */
ORA_API void ora_decode_tbn(uint32_t packed, float m[9])
{
	const float SQRT2 = 1.41421354f, INV_SQRT2 = 0.707106769f;
	float rgb[3] = { (float)((packed >> 20) & 0x3ffu) / 1023.0f, (float)((packed >> 10) & 0x3ffu) / 1023.0f,
	                 (float)(packed & 0x3ffu) / 1023.0f };
	int maxIndex = (int)(packed >> 30);
	float sm[3];
	for (int i = 0; i < 3; i++) sm[i] = rgb[i] * SQRT2 - INV_SQRT2;
	float dd = fmaf(sm[2], sm[2], fmaf(sm[1], sm[1], sm[0] * sm[0]));
	float maxc = sqrtf(1.0f - dd);
	float q[4];
	int k = 0;
	for (int i = 0; i < 4; i++) q[i] = (i == maxIndex) ? maxc : sm[k++];
	float qxx = q[0] * q[0], qyy = q[1] * q[1], qzz = q[2] * q[2];
	float qxz = q[0] * q[2], qxy = q[0] * q[1], qyz = q[1] * q[2];
	float qwx = q[3] * q[0], qwy = q[3] * q[1], qwz = q[3] * q[2];
	m[0] = 1.0f - 2.0f * (qyy + qzz); m[1] = 2.0f * (qxy + qwz);        m[2] = 2.0f * (qxz - qwy);
	m[3] = 2.0f * (qxy - qwz);        m[4] = 1.0f - 2.0f * (qxx + qzz); m[5] = 2.0f * (qyz + qwx);
	m[6] = 2.0f * (qxz + qwy);        m[7] = 2.0f * (qyz - qwx);        m[8] = 1.0f - 2.0f * (qxx + qyy);
}

/* Encoder of main-bake/index_mesh.cpp:178-220: T (already multiplied by the
 * tangent's handedness) and N -> B = N x T -> mat3(T,B,N) -> glm::quat_cast ->
 * normalise -> drop the largest component -> three 10-bit fields, truncated. */
ORA_API uint32_t ora_pack_tbn(const float Tin[3], const float Nin[3])
{
	float T[3], N[3], B[3], c[3];
	glm_normalize3(Tin, T);
	glm_normalize3(Nin, N);
	glm_cross(N, T, c);
	glm_normalize3(c, B);
	/* glm::quat_cast(mat3), third_party/glm/include/glm/gtc/quaternion.inl: m[col][row] */
	float m00 = T[0], m01 = T[1], m02 = T[2];
	float m10 = B[0], m11 = B[1], m12 = B[2];
	float m20 = N[0], m21 = N[1], m22 = N[2];
	float fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
	int big = 0; float fb = fw;
	if (fx > fb) { fb = fx; big = 1; }
	if (fy > fb) { fb = fy; big = 2; }
	if (fz > fb) { fb = fz; big = 3; }
	float bv = sqrtf(fb + 1.0f) * 0.5f;
	float mult = 0.25f / bv;
	float q[4]; /* x y z w */
	switch (big) {
	case 0: q[3] = bv; q[0] = (m12 - m21) * mult; q[1] = (m20 - m02) * mult; q[2] = (m01 - m10) * mult; break;
	case 1: q[3] = (m12 - m21) * mult; q[0] = bv; q[1] = (m01 + m10) * mult; q[2] = (m20 + m02) * mult; break;
	case 2: q[3] = (m20 - m02) * mult; q[0] = (m01 + m10) * mult; q[1] = bv; q[2] = (m12 + m21) * mult; break;
	default: q[3] = (m01 - m10) * mult; q[0] = (m20 + m02) * mult; q[1] = (m12 + m21) * mult; q[2] = bv; break;
	}
	/* glm::normalize(quat): length = sqrt(dot), dot = (w*w + x*x) + (y*y + z*z); q * (1/len) */
	{
		float len = sqrtf((q[3] * q[3] + q[0] * q[0]) + (q[1] * q[1] + q[2] * q[2]));
		if (len <= 0.0f) { q[0] = q[1] = q[2] = 0.0f; q[3] = 1.0f; }
		else { float ol = 1.0f / len; for (int i = 0; i < 4; i++) q[i] *= ol; }
	}
	int maxIndex = 0; float mv = fabsf(q[0]);
	for (int i = 1; i < 4; i++) if (fabsf(q[i]) > mv) { mv = fabsf(q[i]); maxIndex = i; }
	if (q[maxIndex] < 0.0f) for (int i = 0; i < 4; i++) q[i] = -q[i];
	uint32_t small[3]; int k = 0;
	for (int i = 0; i < 4; i++) {
		if (i == maxIndex) continue;
		/* (float + double) / double * float, evaluated in double like the C++ expression */
		double v = ((double)q[i] + (1.0 / sqrt(2.0))) / sqrt(2.0) * (double)1023.0f;
		small[k++] = (uint32_t)v;
	}
	return ((uint32_t)maxIndex << 30) | (small[0] << 20) | (small[1] << 10) | small[2];
}

/* ------------------------------------------------------------------------ */
/* geometry: transform, clip, snap, planes                                   */
/* ------------------------------------------------------------------------ */

/* guard band in NDC half-widths: small enough that the snapped coordinates keep
 * every edge function below 2^53 (exact in double as well as in int64) */
static double guard_band(uint32_t W, uint32_t H) { return ((W > H ? W : H) <= 4096u) ? 32.0 : 8.0; }
#define SUBPIX 256

/* gl_Position = M * vec4(p, 1) */
static inline void xform(const float M[16], const float p[3], float out[4])
{
	for (int i = 0; i < 4; i++) {
		float t = M[0 * 4 + i] * p[0];
		t = fmaf(M[1 * 4 + i], p[1], t);
		t = fmaf(M[2 * 4 + i], p[2], t);
		out[i] = t + M[3 * 4 + i];
	}
}

typedef struct {
	double a[3], b[3], c[3]; /* rows of adj([hx;hy;hw]): lambda_i = a_i*sx + b_i*sy + c_i */
	double Az, Bz, Cz;       /* z_ndc(sx,sy) */
} TriPlanes;

typedef struct {
	int64_t X[3], Y[3]; /* snapped vertices, orientation made positive */
	int front;          /* Vulkan facing with frontFace = COUNTER_CLOCKWISE */
} SubTri;

typedef struct { double x, y, z, w; } ClipV;

static int clip_poly(ClipV* v, int n, int plane, double GUARD)
{
	/* plane: 0 z>=0, 1 x>=-Gw, 2 x<=Gw, 3 y>=-Gw, 4 y<=Gw */
	double d[10];
	int anyOut = 0;
	for (int i = 0; i < n; i++) {
		switch (plane) {
		case 0: d[i] = v[i].z; break;
		case 1: d[i] = v[i].x + GUARD * v[i].w; break;
		case 2: d[i] = GUARD * v[i].w - v[i].x; break;
		case 3: d[i] = v[i].y + GUARD * v[i].w; break;
		default: d[i] = GUARD * v[i].w - v[i].y; break;
		}
		if (!(d[i] >= 0.0)) anyOut = 1;
	}
	if (!anyOut) return n;
	ClipV o[10]; int m = 0;
	for (int i = 0; i < n; i++) {
		int j = (i + 1 == n) ? 0 : i + 1;
		int ini = d[i] >= 0.0, inj = d[j] >= 0.0;
		if (ini) o[m++] = v[i];
		if (ini != inj) {
			/* always interpolate from the inside vertex to the outside one so
			 * that two triangles sharing the edge get the same point */
			const ClipV* A = ini ? &v[i] : &v[j];
			const ClipV* B = ini ? &v[j] : &v[i];
			double dA = ini ? d[i] : d[j], dB = ini ? d[j] : d[i];
			double t = dA / (dA - dB);
			ClipV p;
			p.x = A->x + t * (B->x - A->x);
			p.y = A->y + t * (B->y - A->y);
			p.z = A->z + t * (B->z - A->z);
			p.w = A->w + t * (B->w - A->w);
			o[m++] = p;
		}
	}
	memcpy(v, o, sizeof(ClipV) * (size_t)m);
	return m;
}

/* cull: 0 none, 1 back.  Returns number of sub-triangles (0..6). */
static int tri_setup(const float* c0, const float* c1, const float* c2, uint32_t W, uint32_t H, int cull,
                     TriPlanes* pl, SubTri out[8])
{
	const float* c[3] = { c0, c1, c2 };
	/* trivial reject against the view volume -w<=x,y<=w, 0<=z<=w */
	int ox0 = 1, ox1 = 1, oy0 = 1, oy1 = 1, oz0 = 1, oz1 = 1;
	for (int i = 0; i < 3; i++) {
		float x = c[i][0], y = c[i][1], z = c[i][2], w = c[i][3];
		if (!(x == x) || !(y == y) || !(z == z) || !(w == w)) return 0;
		ox0 &= (x < -w); ox1 &= (x > w); oy0 &= (y < -w); oy1 &= (y > w); oz0 &= (z < 0.0f); oz1 &= (z > w);
	}
	if (ox0 | ox1 | oy0 | oy1 | oz0 | oz1) return 0;

	double hW = (double)W * 0.5, hH = (double)H * 0.5;
	double hx[3], hy[3], hw[3];
	for (int i = 0; i < 3; i++) {
		hx[i] = ((double)c[i][0] + (double)c[i][3]) * hW;
		hy[i] = ((double)c[i][1] + (double)c[i][3]) * hH;
		hw[i] = (double)c[i][3];
	}
	for (int i = 0; i < 3; i++) {
		int j = (i + 1) % 3, k = (i + 2) % 3;
		pl->a[i] = hy[j] * hw[k] - hy[k] * hw[j];
		pl->b[i] = hx[k] * hw[j] - hx[j] * hw[k];
		pl->c[i] = hx[j] * hy[k] - hx[k] * hy[j];
	}
	double det = (hx[0] * pl->a[0] + hy[0] * pl->b[0]) + hw[0] * pl->c[0];
	if (det == 0.0 || !(det == det)) return 0;
	double z0 = c[0][2], z1 = c[1][2], z2 = c[2][2];
	double idet = 1.0 / det;
	pl->Az = ((z0 * pl->a[0] + z1 * pl->a[1]) + z2 * pl->a[2]) * idet;
	pl->Bz = ((z0 * pl->b[0] + z1 * pl->b[1]) + z2 * pl->b[2]) * idet;
	pl->Cz = ((z0 * pl->c[0] + z1 * pl->c[1]) + z2 * pl->c[2]) * idet;

	ClipV v[10];
	int n = 3;
	for (int i = 0; i < 3; i++) { v[i].x = c[i][0]; v[i].y = c[i][1]; v[i].z = c[i][2]; v[i].w = c[i][3]; }
	const double G = guard_band(W, H);
	for (int p = 0; p < 5 && n >= 3; p++) n = clip_poly(v, n, p, G);
	if (n < 3) return 0;
	int64_t X[10], Y[10];
	for (int i = 0; i < n; i++) {
		double iw = 1.0 / v[i].w;
		double sx = (v[i].x * iw + 1.0) * hW;
		double sy = (v[i].y * iw + 1.0) * hH;
		X[i] = (int64_t)floor(sx * (double)SUBPIX + 0.5);
		Y[i] = (int64_t)floor(sy * (double)SUBPIX + 0.5);
	}
	int m = 0;
	for (int i = 1; i + 1 < n; i++) {
		int64_t ax = X[i] - X[0], ay = Y[i] - Y[0], bx = X[i + 1] - X[0], by = Y[i + 1] - Y[0];
		int64_t area2 = ax * by - bx * ay;
		if (area2 == 0) continue;
		int front = area2 < 0; /* a = -area2/2 > 0 */
		if (cull == 1 && !front) continue;
		SubTri* s = &out[m++];
		s->front = front;
		s->X[0] = X[0]; s->Y[0] = Y[0];
		if (area2 > 0) { s->X[1] = X[i]; s->Y[1] = Y[i]; s->X[2] = X[i + 1]; s->Y[2] = Y[i + 1]; }
		else { s->X[1] = X[i + 1]; s->Y[1] = Y[i + 1]; s->X[2] = X[i]; s->Y[2] = Y[i]; }
	}
	return m;
}

static inline float depth_at(const TriPlanes* pl, int px, int py)
{
	double sx = (double)px + 0.5, sy = (double)py + 0.5;
	return (float)fma(pl->Az, sx, fma(pl->Bz, sy, pl->Cz));
}

static inline void bary_at(const TriPlanes* pl, int px, int py, float b[3])
{
	double sx = (double)px + 0.5, sy = (double)py + 0.5;
	double l0 = fma(pl->a[0], sx, fma(pl->b[0], sy, pl->c[0]));
	double l1 = fma(pl->a[1], sx, fma(pl->b[1], sy, pl->c[1]));
	double l2 = fma(pl->a[2], sx, fma(pl->b[2], sy, pl->c[2]));
	double is = 1.0 / ((l0 + l1) + l2);
	b[0] = (float)(l0 * is); b[1] = (float)(l1 * is); b[2] = (float)(l2 * is);
}

static inline float interp(const float b[3], float f0, float f1, float f2)
{
	return fmaf(b[2], f2, fmaf(b[1], f1, b[0] * f0));
}

/* ------------------------------------------------------------------------ */
/* frame state                                                               */
/* ------------------------------------------------------------------------ */
enum { PIPE_GBUF, PIPE_FORWARD, PIPE_DEBUG, PIPE_OVERVIS, PIPE_SHADOW };

typedef struct {
	const ora_scene* sc;
	const vkr_frame_params* fp;
	uint32_t W, H;
	uint32_t sy0, sy1;  /* scissor rows [sy0, sy1) of the colour passes (ora_render_rows); the shadow pass ignores it */
	int pipe;
	/* attachments */
	float* depth;       /* W*H */
	uint32_t* tri;      /* W*H */
	uint8_t* rgba;      /* W*H*4 colour attachment (sRGB8) */
	uint8_t* mip;       /* W*H */
	uint8_t* stencil;   /* W*H */
	uint16_t* gN; uint16_t* gA; /* W*H*4 halfs */
	float* shadow; uint32_t S;  /* S*S */
	float lightMat[16];         /* biasMat * depthMVP */
	/* set by ora_render_bands, which draws one frame as many row ranges: what every range would otherwise compute
	 * for itself (clip-space vertices per mesh for the camera, the finished light depth map); NULL otherwise */
	float* const* sharedClip;
	float* const* sharedClipLight; /* the same under the light's matrix */
	int sharedShadow;
	uint32_t ly0, ly1;  /* rows [ly0, ly1) of the light depth map a PIPE_SHADOW draw may write (0, S: all) */
	/* per draw */
	const OraMesh* mesh;
	const float* clip; /* V*4 for current mesh */
	uint32_t i0, i1, i2;
	uint32_t prim;
	TriPlanes pl;
} Frame;

static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

static void write_colour(uint8_t* dst, const float c[4])
{
	dst[0] = srgb_encode(clamp01(c[0]));
	dst[1] = srgb_encode(clamp01(c[1]));
	dst[2] = srgb_encode(clamp01(c[2]));
	dst[3] = unorm_encode(clamp01(c[3]));
}

static inline float dot3f(const float a[3], const float b[3]) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
static inline void normalize3f(float v[3])
{
	float inv = 1.0f / sqrtf(dot3f(v, v));
	v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

/* uv at the pixel and its fine derivatives inside the 2x2 quad (helper
 * pixels evaluate the triangle's own planes). */
static void uv_and_derivs(const Frame* f, int px, int py, const float b[3], float uv[2], float d[4])
{
	const float* t0 = f->mesh->uv + 2 * (size_t)f->i0;
	const float* t1 = f->mesh->uv + 2 * (size_t)f->i1;
	const float* t2 = f->mesh->uv + 2 * (size_t)f->i2;
	float bx[3], by[3];
	bary_at(&f->pl, px ^ 1, py, bx);
	bary_at(&f->pl, px, py ^ 1, by);
	uv[0] = interp(b, t0[0], t1[0], t2[0]);
	uv[1] = interp(b, t0[1], t1[1], t2[1]);
	float ux = interp(bx, t0[0], t1[0], t2[0]), vx = interp(bx, t0[1], t1[1], t2[1]);
	float uy = interp(by, t0[0], t1[0], t2[0]), vy = interp(by, t0[1], t1[1], t2[1]);
	if (px & 1) { d[0] = uv[0] - ux; d[1] = uv[1] - vx; } else { d[0] = ux - uv[0]; d[1] = vx - uv[1]; }
	if (py & 1) { d[2] = uv[0] - uy; d[3] = uv[1] - vy; } else { d[2] = uy - uv[0]; d[3] = vy - uv[1]; }
}

static void world_pos(const Frame* f, const float b[3], float p[3])
{
	const float* p0 = f->mesh->pos + 3 * (size_t)f->i0;
	const float* p1 = f->mesh->pos + 3 * (size_t)f->i1;
	const float* p2 = f->mesh->pos + 3 * (size_t)f->i2;
	for (int i = 0; i < 3; i++) p[i] = interp(b, p0[i], p1[i], p2[i]);
}

/* normal = v2fTBN * normalize(texture(uNormalMap).rgb*2-1)  (default.frag:83,
 * gBufWrite.frag:17); the mat3 is a per-vertex output interpolated like any
 * varying (default.vert:81). */
static void shading_normal(const Frame* f, const float b[3], const float nm[4], float n[3])
{
	float m0[9], m1[9], m2[9], m[9];
	ora_decode_tbn(f->mesh->tcomp[f->i0], m0);
	ora_decode_tbn(f->mesh->tcomp[f->i1], m1);
	ora_decode_tbn(f->mesh->tcomp[f->i2], m2);
	for (int i = 0; i < 9; i++) m[i] = interp(b, m0[i], m1[i], m2[i]);
	float t[3] = { nm[0] * 2.0f - 1.0f, nm[1] * 2.0f - 1.0f, nm[2] * 2.0f - 1.0f };
	normalize3f(t);
	for (int r = 0; r < 3; r++) n[r] = fmaf(m[6 + r], t[2], fmaf(m[3 + r], t[1], m[0 + r] * t[0]));
}

static const OraTexture* mat_tex(const Frame* f, int which)
{
	const vkr_material_desc* m = &f->sc->mat[f->mesh->material];
	uint32_t id;
	switch (which) {
	case 0: id = m->baseColorTextureId; break;
	case 1: id = m->metalnessTextureId; break;  /* binding 1, main/main.cpp:770-846 */
	case 2: id = m->roughnessTextureId; break;  /* binding 2 */
	case 3: id = (m->alphaMaskTextureId == VKR_NO_TEXTURE) ? m->baseColorTextureId : m->alphaMaskTextureId; break; /* main/main.cpp:812-817 */
	default: id = m->normalMapTextureId; break;
	}
	return &f->sc->tex[id];
}

/* BRDF pieces shared by default.frag:33-80 and deferredShading.frag:43-90;
 * nhFloor = 0.0001 (forward) or 0.0 (deferred). */
static void brdf_eval(const float L[3], const float V[3], const float N[3], const float albedo[3], float metal,
                      float roughTex, float nhFloor, float out[3])
{
	const float PI = 3.14159265359f;
	float Hh[3] = { V[0] + L[0], V[1] + L[1], V[2] + L[2] };
	normalize3f(Hh);
	float rough = roughTex * roughTex;
	float ndh = dot3f(N, Hh), ndv = dot3f(N, V), ndl = dot3f(N, L), vdh = dot3f(V, Hh);
	/* DistributionFunction */
	float nh = fmaxf(ndh, nhFloor);
	float nh2 = nh * nh, nh4 = nh2 * nh2, r2 = rough * rough;
	float ndf = expf((nh2 - 1.0f) / (r2 * nh2)) / (0.001f + PI * r2 * nh4);
	/* Fresnel: dot(halfway, view) */
	float x = 1.0f - vdh;
	float x2 = x * x, x5 = x2 * x2 * x;
	float F[3];
	for (int i = 0; i < 3; i++) {
		float f0 = (1.0f - metal) * 0.04f + metal * albedo[i];
		F[i] = f0 + (1.0f - f0) * x5;
	}
	/* GeometryFunction */
	float tl = 2.0f * (fmaxf(0.0f, ndh) * fmaxf(0.0f, ndv) / vdh);
	float tr = 2.0f * (fmaxf(0.0f, ndh) * fmaxf(0.0f, ndl) / vdh);
	float G = fminf(1.0f, fminf(tl, tr));
	float denom = 0.001f + 4.0f * fmaxf(ndv, 0.0f) * fmaxf(ndl, 0.0f);
	for (int i = 0; i < 3; i++) {
		float diffuse = (albedo[i] / PI) * (1.0f - F[i]) * (1.0f - metal);
		out[i] = diffuse + (ndf * F[i] * G) / denom;
	}
}

/* texture(shadowMap, coord): compare sampler of utils/vkutil.cpp:198-217
 * (LINEAR, CLAMP_TO_EDGE, LESS_OR_EQUAL): compare each of the four taps, then
 * blend the 0/1 results. */
static float shadow_lookup(const Frame* f, float u, float v, float ref)
{
	int S = (int)f->S;
	float x = u * (float)S - 0.5f, y = v * (float)S - 0.5f;
	if (!(fabsf(x) < 1.0e9f)) x = 0.0f;
	if (!(fabsf(y) < 1.0e9f)) y = 0.0f;
	float x0f = floorf(x), y0f = floorf(y);
	float fx = x - x0f, fy = y - y0f;
	int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
	if (x0 < 0) x0 = 0; if (x0 > S - 1) x0 = S - 1; if (x1 < 0) x1 = 0; if (x1 > S - 1) x1 = S - 1;
	if (y0 < 0) y0 = 0; if (y0 > S - 1) y0 = S - 1; if (y1 < 0) y1 = 0; if (y1 > S - 1) y1 = S - 1;
	float t00 = (ref <= f->shadow[(size_t)y0 * S + x0]) ? 1.0f : 0.0f;
	float t10 = (ref <= f->shadow[(size_t)y0 * S + x1]) ? 1.0f : 0.0f;
	float t01 = (ref <= f->shadow[(size_t)y1 * S + x0]) ? 1.0f : 0.0f;
	float t11 = (ref <= f->shadow[(size_t)y1 * S + x1]) ? 1.0f : 0.0f;
	return lerpf(lerpf(t00, t10, fx), lerpf(t01, t11, fx), fy);
}

/* default.frag:82-104.  Returns 0 if the fragment is discarded. */
static int shade_forward(const Frame* f, int px, int py, float out[4])
{
	float b[3], uv[2], d[4];
	bary_at(&f->pl, px, py, b);
	uv_and_derivs(f, px, py, b, uv, d);
	float al[4];
	sample_trilinear(mat_tex(f, 3), uv[0], uv[1], d, al);
	if (al[3] < 0.5f) return 0;
	float nm[4], alb[4], me[4], ro[4], N[3], P[3];
	sample_trilinear(mat_tex(f, 4), uv[0], uv[1], d, nm);
	sample_trilinear(mat_tex(f, 0), uv[0], uv[1], d, alb);
	sample_trilinear(mat_tex(f, 1), uv[0], uv[1], d, me);
	sample_trilinear(mat_tex(f, 2), uv[0], uv[1], d, ro);
	shading_normal(f, b, nm, N);
	world_pos(f, b, P);
	const vkr_uniforms* u = &f->fp->uniforms;
	float Lv[3] = { u->lightUniforms.lightPos[0] - P[0], u->lightUniforms.lightPos[1] - P[1], u->lightUniforms.lightPos[2] - P[2] };
	float Vv[3] = { u->sceneUniforms.camPos[0] - P[0], u->sceneUniforms.camPos[1] - P[1], u->sceneUniforms.camPos[2] - P[2] };
	float dist2 = dot3f(Lv, Lv);
	float L[3] = { Lv[0], Lv[1], Lv[2] };
	normalize3f(L);
	normalize3f(Vv);
	float br[3];
	brdf_eval(L, Vv, N, alb, me[0], ro[0], 0.0001f, br);
	float ndl = fmaxf(dot3f(N, L), 0.0001f);
	float att = 1.0f / dist2; /* 1 / pow(length(.), 2) */
	/* v2fLightSpacePosition = (biasMat * depthMVP) * vec4(pos,1), default.vert:83 */
	float ls[4];
	xform(f->lightMat, P, ls);
	float sh = shadow_lookup(f, ls[0] / ls[3], ls[1] / ls[3], ls[2] / ls[3]);
	sh = fmaxf(sh, 0.1f);
	for (int i = 0; i < 3; i++) {
		float ambient = 0.03f * alb[i];
		float lit = (br[i] * 100.0f) * u->lightUniforms.lightColour[i] * ndl;
		out[i] = ambient + (lit * sh) * att;
	}
	out[3] = 1.0f;
	return 1;
}

/* gBufWrite.frag:16-27.  Returns 0 if discarded. */
static int shade_gbuf(const Frame* f, int px, int py, uint16_t gN[4], uint16_t gA[4])
{
	float b[3], uv[2], d[4];
	bary_at(&f->pl, px, py, b);
	uv_and_derivs(f, px, py, b, uv, d);
	float al[4];
	sample_trilinear(mat_tex(f, 3), uv[0], uv[1], d, al);
	if (al[3] < 0.5f) return 0;
	float nm[4], alb[4], me[4], ro[4], N[3];
	sample_trilinear(mat_tex(f, 4), uv[0], uv[1], d, nm);
	sample_trilinear(mat_tex(f, 0), uv[0], uv[1], d, alb);
	sample_trilinear(mat_tex(f, 1), uv[0], uv[1], d, me);
	sample_trilinear(mat_tex(f, 2), uv[0], uv[1], d, ro);
	shading_normal(f, b, nm, N);
	gN[0] = float_to_half(N[0]); gN[1] = float_to_half(N[1]); gN[2] = float_to_half(N[2]); gN[3] = float_to_half(me[0]);
	gA[0] = float_to_half(alb[0]); gA[1] = float_to_half(alb[1]); gA[2] = float_to_half(alb[2]); gA[3] = float_to_half(ro[0]);
	return 1;
}

/* debug.frag:15-17 */
static inline float linearise_depth(float d)
{
	return (2.0f * 0.1f * 100.0f) / (100.0f + 0.1f - (2.0f * d - 1.0f) * (100.0f - 0.1f));
}

/* debug.frag:19-62 */
static void shade_debug(const Frame* f, int px, int py, float z, float out[4], uint8_t* mipOut)
{
	int mode = f->fp->uniforms.debugUniforms.debug;
	*mipOut = 0xff;
	if (mode == 3) {
		float v = linearise_depth(z) / 100.0f;
		out[0] = out[1] = out[2] = v; out[3] = 1.0f;
		return;
	}
	if (mode == 4) {
		float zc = linearise_depth(z);
		float zx = linearise_depth(depth_at(&f->pl, px ^ 1, py));
		float zy = linearise_depth(depth_at(&f->pl, px, py ^ 1));
		float dx = (px & 1) ? (zc - zx) : (zx - zc);
		float dy = (py & 1) ? (zc - zy) : (zy - zc);
		out[0] = dx * 5.0f; out[1] = dy * 5.0f; out[2] = 0.0f; out[3] = 1.0f;
		return;
	}
	float b[3], uv[2], d[4];
	bary_at(&f->pl, px, py, b);
	uv_and_derivs(f, px, py, b, uv, d);
	const OraTexture* t = mat_tex(f, 0);
	int lv; float fr;
	lod_split(rho2_for(t, d), t->levels - 1, &lv, &fr);
	*mipOut = (uint8_t)lv;
	if (mode == 2) {
		static const float col[7][3] = { { 17.0f, 119.0f, 51.0f }, { 136.0f, 34.0f, 85.0f }, { 0.0f, 114.0f, 178.0f },
			{ 204.0f, 121.0f, 167.0f }, { 0.0f, 158.0f, 115.0f }, { 213.0f, 94.0f, 0.0f }, { 240.0f, 228.0f, 66.0f } };
		/* lambda = lv + fr; floor = lv, ceil = lv + (fr>0), fract = fr (fr==1 only by rounding) */
		int fl = lv, ce = (fr > 0.0f) ? lv + 1 : lv;
		float fc = fr;
		if (fr >= 1.0f) { fl = lv + 1; ce = lv + 1; fc = 0.0f; }
		for (int i = 0; i < 3; i++) {
			float a = col[fl % 7][i] / 255.0f, c = col[ce % 7][i] / 255.0f;
			out[i] = a * (1.0f - fc) + c * fc; /* mix() */
		}
		out[3] = 1.0f;
		return;
	}
	sample_trilinear(t, uv[0], uv[1], d, out);
}

/* ------------------------------------------------------------------------ */
/* rasterisation of one primitive                                            */
/* ------------------------------------------------------------------------ */
/* A row range's cheap test in front of the triangle set-up: 1 when the triangle cannot touch rows [r0, r1).
 * With every w > 0 the clipped polygon lies inside the projected triangle, so the projected vertices bound its
 * rows; one pixel of slack covers the snapping.  Anything else is left to the set-up. */
static int rows_missed(const float* c0, const float* c1, const float* c2, uint32_t H, uint32_t r0, uint32_t r1)
{
	if (!(c0[3] > 0.0f && c1[3] > 0.0f && c2[3] > 0.0f)) return 0;
	const double h = (double)H * 0.5;
	const double y0 = ((double)c0[1] / (double)c0[3] + 1.0) * h, y1 = ((double)c1[1] / (double)c1[3] + 1.0) * h,
	             y2 = ((double)c2[1] / (double)c2[3] + 1.0) * h;
	if (!(y0 == y0) || !(y1 == y1) || !(y2 == y2)) return 0;
	const double lo = fmin(y0, fmin(y1, y2)), hi = fmax(y0, fmax(y1, y2));
	return hi < (double)r0 - 1.0 || lo > (double)r1 + 1.0;
}

static void raster_prim(Frame* f, int cull, uint32_t W, uint32_t H)
{
	SubTri st[8];
	const float* c0 = f->clip + 4 * (size_t)f->i0;
	const float* c1 = f->clip + 4 * (size_t)f->i1;
	const float* c2 = f->clip + 4 * (size_t)f->i2;
	const uint32_t r0 = (f->pipe == PIPE_SHADOW) ? f->ly0 : f->sy0, r1 = (f->pipe == PIPE_SHADOW) ? f->ly1 : f->sy1;
	if ((r0 > 0 || r1 < H) && rows_missed(c0, c1, c2, H, r0, r1)) return;
	int n = tri_setup(c0, c1, c2, W, H, cull, &f->pl, st);
	for (int k = 0; k < n; k++) {
		const SubTri* s = &st[k];
		int64_t xmin = s->X[0], xmax = s->X[0], ymin = s->Y[0], ymax = s->Y[0];
		for (int i = 1; i < 3; i++) {
			if (s->X[i] < xmin) xmin = s->X[i]; if (s->X[i] > xmax) xmax = s->X[i];
			if (s->Y[i] < ymin) ymin = s->Y[i]; if (s->Y[i] > ymax) ymax = s->Y[i];
		}
		/* pixel centres at (p*256+128): p >= (min-128)/256, p <= (max-128)/256 */
		int64_t px0 = (xmin - 128 + 255) >> 8, px1 = (xmax - 128) >> 8;
		int64_t py0 = (ymin - 128 + 255) >> 8, py1 = (ymax - 128) >> 8;
		if (px0 < 0) px0 = 0; if (py0 < 0) py0 = 0;
		if (px1 > (int64_t)W - 1) px1 = (int64_t)W - 1;
		if (py1 > (int64_t)H - 1) py1 = (int64_t)H - 1;
		if (py0 < (int64_t)r0) py0 = (int64_t)r0;
		if (py1 > (int64_t)r1 - 1) py1 = (int64_t)r1 - 1;
		if (px0 > px1 || py0 > py1) continue;
		/* edge i: from vertex i to vertex (i+1)%3; E = dx*(py-y) - dy*(px-x) */
		int64_t dx[3], dy[3]; int tl[3];
		for (int i = 0; i < 3; i++) {
			int j = (i + 1) % 3;
			dx[i] = s->X[j] - s->X[i]; dy[i] = s->Y[j] - s->Y[i];
			tl[i] = (dy[i] == 0 && dx[i] > 0) || (dy[i] < 0); /* top edge or left edge */
		}
		/* shadow bias (main/main.cpp:3112-3114): o = m*slope + r*constant */
		double bias = 0.0;
		if (f->pipe == PIPE_SHADOW) {
			double m = fabs(f->pl.Az) > fabs(f->pl.Bz) ? fabs(f->pl.Az) : fabs(f->pl.Bz);
			double zmax = 0.0;
			for (int i = 0; i < 3; i++) {
				double zv = fma(f->pl.Az, (double)s->X[i] / 256.0, fma(f->pl.Bz, (double)s->Y[i] / 256.0, f->pl.Cz));
				if (zv > zmax) zmax = zv;
			}
			if (zmax > 1.0) zmax = 1.0;
			float zmf = (float)zmax;
			int e = (int)((f2u(zmf) >> 23) & 0xffu) - 127;
			if (e < -126) e = -126;
			double r = ldexp(1.0, e - 23);
			bias = m * 10.0 + r * 7.0;
		}
		for (int64_t py = py0; py <= py1; py++) {
			int64_t PY = py * 256 + 128;
			for (int64_t px = px0; px <= px1; px++) {
				int64_t PX = px * 256 + 128;
				int inside = 1;
				for (int i = 0; i < 3; i++) {
					int64_t E = dx[i] * (PY - s->Y[i]) - dy[i] * (PX - s->X[i]);
					if (E < 0 || (E == 0 && !tl[i])) { inside = 0; break; }
				}
				if (!inside) continue;
				float z = depth_at(&f->pl, (int)px, (int)py);
				if (!(z >= 0.0f && z <= 1.0f)) continue; /* clip volume 0 <= z <= w, no depth clamp */
				size_t p = (size_t)py * W + (size_t)px;
				switch (f->pipe) {
				case PIPE_SHADOW: {
					double sx = (double)px + 0.5, sy = (double)py + 0.5;
					float zb = (float)(fma(f->pl.Az, sx, fma(f->pl.Bz, sy, f->pl.Cz)) + bias);
					if (zb > 1.0f) zb = 1.0f;
					if (zb < 0.0f) zb = 0.0f;
					if (zb <= f->shadow[p]) f->shadow[p] = zb;
					break;
				}
				case PIPE_OVERVIS:
					/* stencil INCREMENT_AND_CLAMP on pass; depth test only in mode 6
					 * (main/main.cpp:2781-2797, 4120-4123); no depth write without a depth test */
					if (f->fp->debugVisualisation == 6) {
						if (!(z <= f->depth[p])) break;
						f->depth[p] = z;
					}
					if (f->stencil[p] < 255) f->stencil[p]++;
					break;
				case PIPE_DEBUG: {
					if (!(z <= f->depth[p])) break;
					float col[4]; uint8_t mp;
					shade_debug(f, (int)px, (int)py, z, col, &mp);
					f->depth[p] = z; f->tri[p] = f->prim; f->mip[p] = mp;
					write_colour(f->rgba + p * 4, col);
					break;
				}
				case PIPE_GBUF: {
					if (!(z <= f->depth[p])) break;
					uint16_t gn[4], ga[4];
					if (!shade_gbuf(f, (int)px, (int)py, gn, ga)) break;
					f->depth[p] = z; f->tri[p] = f->prim;
					memcpy(f->gN + p * 4, gn, 8); memcpy(f->gA + p * 4, ga, 8);
					break;
				}
				default: { /* PIPE_FORWARD */
					if (!(z <= f->depth[p])) break;
					float col[4];
					if (!shade_forward(f, (int)px, (int)py, col)) break;
					f->depth[p] = z; f->tri[p] = f->prim;
					write_colour(f->rgba + p * 4, col);
					break;
				}
				}
			}
		}
	}
}

/* Draws prims [first,last) of an order list with the given cull mode. */
static void draw_range(Frame* f, const OraPrim* order, uint64_t first, uint64_t last, const float M[16], int cullOpaque,
                       int cullAlpha, uint32_t W, uint32_t H)
{
	const ora_scene* sc = f->sc;
	float* const* shared = (f->pipe == PIPE_SHADOW) ? f->sharedClipLight : f->sharedClip;
	/* transform every mesh once */
	float** clip = (float**)calloc(sc->nMesh ? sc->nMesh : 1, sizeof(float*));
	for (uint64_t i = first; i < last; i++) {
		uint32_t mi = order[i].mesh;
		const OraMesh* m = &sc->mesh[mi];
		if (shared) {
			f->mesh = m; f->clip = shared[mi];
		} else if (!clip[mi]) {
			clip[mi] = (float*)malloc(sizeof(float) * 4 * (m->V ? m->V : 1));
			for (uint32_t v = 0; v < m->V; v++) xform(M, m->pos + 3 * (size_t)v, clip[mi] + 4 * (size_t)v);
		}
		if (!shared) { f->mesh = m; f->clip = clip[mi]; }
		f->i0 = m->idx[3 * (size_t)order[i].tri + 0];
		f->i1 = m->idx[3 * (size_t)order[i].tri + 1];
		f->i2 = m->idx[3 * (size_t)order[i].tri + 2];
		if (f->i0 >= m->V || f->i1 >= m->V || f->i2 >= m->V) continue;
		f->prim = (uint32_t)i;
		raster_prim(f, m->alpha ? cullAlpha : cullOpaque, W, H);
	}
	for (uint32_t i = 0; i < sc->nMesh; i++) free(clip[i]);
	free(clip);
}

/* deferredShading.frag:92-119 over the whole frame */
static void deferred_light_pass(Frame* f)
{
	const vkr_uniforms* u = &f->fp->uniforms;
	float invCam[16], invProj[16], invPC[16];
	mat4_inverse(u->sceneUniforms.camera, invCam);
	mat4_inverse(u->sceneUniforms.projection, invProj);
	mat4_mul(invCam, invProj, invPC);
	for (uint32_t py = f->sy0; py < f->sy1; py++)
		for (uint32_t px = 0; px < f->W; px++) {
			size_t p = (size_t)py * f->W + px;
			float depth = f->depth[p];
			if (depth == 1.0f) continue; /* discard: clear colour stays */
			float tu = ((float)px + 0.5f) / (float)f->W, tv = ((float)py + 0.5f) / (float)f->H;
			float cx = tu * 2.0f - 1.0f, cy = tv * 2.0f - 1.0f;
			float vs[4];
			for (int i = 0; i < 4; i++) {
				float t = invPC[0 * 4 + i] * cx;
				t = fmaf(invPC[1 * 4 + i], cy, t);
				t = fmaf(invPC[2 * 4 + i], depth, t);
				vs[i] = t + invPC[3 * 4 + i];
			}
			float pos[3] = { vs[0] / vs[3], vs[1] / vs[3], vs[2] / vs[3] };
			float V[3] = { u->sceneUniforms.camPos[0] - pos[0], u->sceneUniforms.camPos[1] - pos[1], u->sceneUniforms.camPos[2] - pos[2] };
			normalize3f(V);
			float N[3] = { half_to_float(f->gN[p * 4 + 0]), half_to_float(f->gN[p * 4 + 1]), half_to_float(f->gN[p * 4 + 2]) };
			normalize3f(N);
			float metal = half_to_float(f->gN[p * 4 + 3]);
			float alb[3] = { half_to_float(f->gA[p * 4 + 0]), half_to_float(f->gA[p * 4 + 1]), half_to_float(f->gA[p * 4 + 2]) };
			float rough = half_to_float(f->gA[p * 4 + 3]);
			float col[4] = { 0, 0, 0, 0 };
			for (int i = 0; i < VKR_NUM_LIGHTS; i++) {
				const vkr_light_uniform* li = &u->multipleLightsUniform.light[i];
				float Lv[3] = { li->lightPos[0] - pos[0], li->lightPos[1] - pos[1], li->lightPos[2] - pos[2] };
				float dist2 = dot3f(Lv, Lv);
				float L[3] = { Lv[0], Lv[1], Lv[2] };
				normalize3f(L);
				float br[3];
				brdf_eval(L, V, N, alb, metal, rough, 0.0f, br);
				float ndl = fmaxf(dot3f(N, L), 0.0f);
				for (int c = 0; c < 3; c++) {
					float ambient = 0.03f * alb[c];
					col[c] += (ambient + (br[c] * 10.0f) * li->lightColour[c] * ndl) / dist2;
				}
				col[3] += 1.0f / dist2;
			}
			write_colour(f->rgba + p * 4, col);
		}
}

/* postProcess.frag:9-25 with the default sampler on the offscreen
 * R8G8B8A8_SRGB image (main/main.cpp:1607, 3216); rows [r0, r1) of the result. */
static void mosaic_pass(const uint8_t* src, uint8_t* dst, uint32_t W, uint32_t H, uint32_t r0, uint32_t r1)
{
	float sx = (float)W, sy = (float)H;
	float kw = 5.0f / sx, kh = 3.0f / sy;
	float ox = 0.5f / sx, oy = 0.5f / sy;
	for (uint32_t py = r0; py < r1; py++)
		for (uint32_t px = 0; px < W; px++) {
			float u = ((float)px + 0.5f) / sx, v = ((float)py + 0.5f) / sy;
			float su = kw * floorf(u / kw), sv = kh * floorf(v / kh);
			su = fminf(fmaxf(su, 0.0f + ox), 1.0f - ox);
			sv = fminf(fmaxf(sv, 0.0f + oy), 1.0f - oy);
			float x = su * sx - 0.5f, y = sv * sy - 0.5f;
			float x0f = floorf(x), y0f = floorf(y);
			float fx = x - x0f, fy = y - y0f;
			int x0 = wrap_repeat((int)x0f, (int)W), x1 = wrap_repeat((int)x0f + 1, (int)W);
			int y0 = wrap_repeat((int)y0f, (int)H), y1 = wrap_repeat((int)y0f + 1, (int)H);
			const uint8_t* t00 = src + ((size_t)y0 * W + x0) * 4; const uint8_t* t10 = src + ((size_t)y0 * W + x1) * 4;
			const uint8_t* t01 = src + ((size_t)y1 * W + x0) * 4; const uint8_t* t11 = src + ((size_t)y1 * W + x1) * 4;
			float c[4];
			for (int k = 0; k < 4; k++) {
				float a00, a10, a01, a11;
				if (k < 3) { a00 = g_srgb_decode[t00[k]]; a10 = g_srgb_decode[t10[k]]; a01 = g_srgb_decode[t01[k]]; a11 = g_srgb_decode[t11[k]]; }
				else { a00 = (float)t00[k] / 255.0f; a10 = (float)t10[k] / 255.0f; a01 = (float)t01[k] / 255.0f; a11 = (float)t11[k] / 255.0f; }
				c[k] = lerpf(lerpf(a00, a10, fx), lerpf(a01, a11, fx), fy);
			}
			write_colour(dst + ((size_t)py * W + px) * 4, c);
		}
}

/* rows [sy0, sy1) of the colour attachment */
static void fill_colour_rows(Frame* f, float r, float g, float b, float a)
{
	float c[4] = { r, g, b, a };
	uint8_t px[4];
	write_colour(px, c);
	for (size_t i = (size_t)f->sy0 * f->W; i < (size_t)f->sy1 * f->W; i++) memcpy(f->rgba + i * 4, px, 4);
}

/* The light's depth pass (main/main.cpp:3950-3984): rows [ly0, ly1) of the map, opaque meshes only. */
static void shadow_pass(Frame* f)
{
	const ora_scene* sc = f->sc;
	for (size_t i = (size_t)f->ly0 * f->S; i < (size_t)f->ly1 * f->S; i++) f->shadow[i] = 1.0f;
	f->pipe = PIPE_SHADOW;
	draw_range(f, sc->orderA, 0, sc->nPrimOpaque, f->fp->uniforms.depthMVPUniform.depthMVP, 0, 0, f->S, f->S);
}

/* Shadow pass (main/main.cpp:3950-3984) followed by the regular pass
 * (main/main.cpp:3987-4077). */
static void forward_passes(Frame* f)
{
	const ora_scene* sc = f->sc;
	const vkr_uniforms* u = &f->fp->uniforms;
	if (!f->sharedShadow) shadow_pass(f);
	/* biasMat * depthMVP, default.vert:46-50,83 */
	static const float biasMat[16] = { 0.5f, 0, 0, 0, 0, 0.5f, 0, 0, 0, 0, 1.0f, 0, 0.5f, 0.5f, 0, 1.0f };
	mat4_mul(biasMat, u->depthMVPUniform.depthMVP, f->lightMat);
	fill_colour_rows(f, 0.1f, 0.1f, 0.1f, 1.0f);
	f->pipe = PIPE_FORWARD;
	draw_range(f, sc->orderA, 0, sc->nPrim, u->sceneUniforms.projCam, 1, 0, f->W, f->H);
}

/* What ora_render_bands computes once per frame for all of its row ranges. */
typedef struct {
	float** clip;      /* per mesh: clip-space vertices under the camera's matrix */
	float** clipLight; /* the same under the light's matrix (when the mode has a light pass) */
	float* shadow;     /* the finished light depth map, or NULL when the mode has none */
} BandShared;

/* Rows [y0, y1) of one frame.  The attachments are allocated at frame size and indexed by frame position, but
 * only the rows this call draws are cleared and touched (untouched pages of a large allocation cost nothing). */
static int render_rows(const ora_scene* sc, const vkr_frame_params* fp, const vkr_frame_out* out, uint32_t y0, uint32_t y1,
                       const BandShared* sh)
{
	init_tables();
	Frame f;
	memset(&f, 0, sizeof f);
	f.sc = sc; f.fp = fp; f.W = fp->width; f.H = fp->height;
	if (y1 > f.H) y1 = f.H;
	if (y0 >= y1) return 1;
	f.sy0 = y0; f.sy1 = y1;
	/* the mosaic pass reads the top row of each 3-row block of the image below it */
	if (fp->mosaicEffect) { const uint32_t top = y0 - (y0 % 3u); f.sy0 = top > 4u ? top - 4u : 0u; }
	f.S = fp->shadow_size ? fp->shadow_size : 2048;
	f.ly0 = 0; f.ly1 = f.S;
	if (sh) { f.sharedClip = sh->clip; f.sharedShadow = sh->shadow != NULL; f.shadow = sh->shadow; }
	size_t n = (size_t)f.W * f.H;
	if (n == 0) return 1;
	const size_t r0 = (size_t)f.sy0 * f.W, r1 = (size_t)f.sy1 * f.W;
	f.depth = (float*)malloc(n * 4); f.tri = (uint32_t*)malloc(n * 4); f.rgba = (uint8_t*)malloc(n * 4);
	f.mip = (uint8_t*)malloc(n); f.stencil = (uint8_t*)malloc(n);
	f.gN = (uint16_t*)malloc(n * 8); f.gA = (uint16_t*)malloc(n * 8);
	for (size_t i = r0; i < r1; i++) { f.depth[i] = 1.0f; f.tri[i] = VKR_NO_TRIANGLE; f.mip[i] = 0xff; f.stencil[i] = 0; }
	const vkr_uniforms* u = &fp->uniforms;
	int dbg = fp->debugVisualisation;
	int needShadow = 0;

	{
		uint16_t h01 = float_to_half(0.1f);
		for (size_t i = r0 * 4; i < r1 * 4; i++) { f.gN[i] = h01; f.gA[i] = h01; } /* clears, main/main.cpp:3843-3851 */
	}
	if (fp->deferredShading && !fp->mosaicEffect) {
		fill_colour_rows(&f, 0.1f, 0.1f, 0.1f, 0.1f); /* main/main.cpp:3838-3841 */
		f.pipe = PIPE_GBUF;
		draw_range(&f, sc->orderA, 0, sc->nPrim, u->sceneUniforms.projCam, 1, 0, f.W, f.H);
		deferred_light_pass(&f);
		if ((fp->flags & VKR_FRAME_SHADOW_PASS) && out->shadow_map) {
			/* the "deferred + shadow map" composition (not a reference combination: its deferred path records no
			 * shadow pass, main/main.cpp:3836-3946): the light's depth pass of main/main.cpp:3950-3984 is drawn as
			 * well and handed out; the lighting does not sample it */
			needShadow = 1;
			if (!f.sharedShadow) {
				f.shadow = (float*)malloc((size_t)f.S * f.S * 4);
				shadow_pass(&f);
			}
		}
	} else if (!fp->mosaicEffect && dbg == 1) {
		needShadow = 1;
		if (!f.sharedShadow) f.shadow = (float*)malloc((size_t)f.S * f.S * 4);
		forward_passes(&f);
	} else if (!fp->mosaicEffect && (dbg == 5 || dbg == 6)) {
		f.pipe = PIPE_OVERVIS;
		draw_range(&f, sc->orderB, 0, sc->nPrim, u->sceneUniforms.projCam, 1, 1, f.W, f.H);
		/* overVisRead.frag:7-13 */
		for (size_t p = r0; p < r1; p++) {
			float s = (float)f.stencil[p] / 25.0f;
			float c[4] = { 0.0f + s, 0.3f + s, 0.0f + s, 0.1f + 0.0f };
			write_colour(f.rgba + p * 4, c);
		}
	} else if (!fp->mosaicEffect) {
		fill_colour_rows(&f, 0.1f, 0.1f, 0.1f, 1.0f); /* main/main.cpp:4139-4143 */
		f.pipe = PIPE_DEBUG;
		draw_range(&f, sc->orderB, 0, sc->nPrim, u->sceneUniforms.projCam, 1, 1, f.W, f.H);
	} else {
		/* Mosaic: forward pass into the offscreen sRGB image, then the 5x3
		 * block post-process (main/main.cpp:4195-4320).  The reference records
		 * no shadow pass and binds no shadow descriptors here, which leaves the
		 * shadow term undefined; this restatement renders the shadow map as the
		 * regular path does. */
		needShadow = 1;
		if (!f.sharedShadow) f.shadow = (float*)malloc((size_t)f.S * f.S * 4);
		forward_passes(&f);
		uint8_t* dst = (uint8_t*)malloc(n * 4);
		mosaic_pass(f.rgba, dst, f.W, f.H, y0, y1);
		free(f.rgba);
		f.rgba = dst;
	}

	{
		const size_t o = (size_t)y0 * f.W, m = (size_t)(y1 - y0) * f.W;
		if (out->rgba8) memcpy((uint8_t*)out->rgba8 + o * 4, f.rgba + o * 4, m * 4);
		if (out->tri_id) memcpy(out->tri_id + o, f.tri + o, m * 4);
		if (out->depth_bits) memcpy(out->depth_bits + o, f.depth + o, m * 4);
		if (out->mip_level) memcpy(out->mip_level + o, f.mip + o, m);
		if (out->stencil) memcpy(out->stencil + o, f.stencil + o, m);
		if (out->gbuf_normal) memcpy(out->gbuf_normal + o * 4, f.gN + o * 4, m * 8);
		if (out->gbuf_albedo) memcpy(out->gbuf_albedo + o * 4, f.gA + o * 4, m * 8);
	}
	if (!f.sharedShadow) {
		if (out->shadow_map && needShadow) memcpy(out->shadow_map, f.shadow, (size_t)f.S * f.S * 4);
		free(f.shadow);
	}
	free(f.depth); free(f.tri); free(f.rgba); free(f.mip); free(f.stencil); free(f.gN); free(f.gA);
	return 0;
}

/* Rows [y0, y1) of one frame (rect fields of the parameters ignored); all outputs host memory, full-frame
 * sized, only those rows written (the shadow map whole).  Calls for disjoint row ranges of the same frame
 * may run concurrently on the same output buffers as long as no shadow map is asked for (every call would
 * write the whole map): the scene is read-only and every call owns its attachments. */
ORA_API int ora_render_rows(const ora_scene* sc, const vkr_frame_params* fp, const vkr_frame_out* out, uint32_t y0, uint32_t y1)
{
	return render_rows(sc, fp, out, y0, y1, NULL);
}

/* One whole frame as bands of band_rows rows, the way a CPU renderer with several cores would draw it: the vertices
 * are transformed once, the light's depth map is drawn once (as bands of its own rows) and then only read, and every
 * band clears, draws and shades its own rows only.  Compiled with -fopenmp the three steps are shared by an OpenMP
 * team, dynamically scheduled (the cost of a band depends on what it shows); without OpenMP the pragmas are ignored
 * and everything runs in turn.  Same frame either way, and the same as ora_render's. */
ORA_API int ora_render_bands(const ora_scene* sc, const vkr_frame_params* fp, const vkr_frame_out* out, uint32_t band_rows)
{
	if (!band_rows) band_rows = 64;
	init_tables();
	const uint32_t S = fp->shadow_size ? fp->shadow_size : 2048;
	const int dbg = fp->debugVisualisation;
	const int deferred = fp->deferredShading && !fp->mosaicEffect;
	const int shadowed = deferred ? ((fp->flags & VKR_FRAME_SHADOW_PASS) && out->shadow_map) : (fp->mosaicEffect || dbg == 1);
	BandShared sh;
	sh.clip = (float**)calloc(sc->nMesh ? sc->nMesh : 1, sizeof(float*));
	sh.shadow = shadowed ? (float*)malloc((size_t)S * S * 4) : NULL;
	sh.clipLight = (float**)calloc(sc->nMesh ? sc->nMesh : 1, sizeof(float*));
	for (uint32_t mi = 0; mi < sc->nMesh; mi++) {
		const size_t nv = sc->mesh[mi].V ? sc->mesh[mi].V : 1;
		sh.clip[mi] = (float*)malloc(sizeof(float) * 4 * nv);
		if (shadowed && !sc->mesh[mi].alpha) sh.clipLight[mi] = (float*)malloc(sizeof(float) * 4 * nv);
	}
	/* the vertex stage, once per frame and pass */
#pragma omp parallel for schedule(dynamic, 1)
	for (int mi = 0; mi < (int)sc->nMesh; mi++) {
		const OraMesh* m = &sc->mesh[mi];
		for (uint32_t v = 0; v < m->V; v++) xform(fp->uniforms.sceneUniforms.projCam, m->pos + 3 * (size_t)v, sh.clip[mi] + 4 * (size_t)v);
		if (sh.clipLight[mi])
			for (uint32_t v = 0; v < m->V; v++) xform(fp->uniforms.depthMVPUniform.depthMVP, m->pos + 3 * (size_t)v, sh.clipLight[mi] + 4 * (size_t)v);
	}
	if (shadowed) {
		const int nl = (int)((S + 63u) / 64u);
#pragma omp parallel for schedule(dynamic, 1)
		for (int b = 0; b < nl; b++) {
			Frame f;
			memset(&f, 0, sizeof f);
			f.sc = sc; f.fp = fp; f.W = fp->width; f.H = fp->height; f.S = S; f.shadow = sh.shadow;
			f.sharedClipLight = sh.clipLight;
			f.ly0 = (uint32_t)b * 64u; f.ly1 = f.ly0 + 64u < S ? f.ly0 + 64u : S;
			shadow_pass(&f);
		}
	}
	const int nb = (int)((fp->height + band_rows - 1) / band_rows);
	int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : bad)
	for (int b = 0; b < nb; b++) {
		const uint32_t y0 = (uint32_t)b * band_rows, y1 = y0 + band_rows;
		bad |= render_rows(sc, fp, out, y0, y1 < fp->height ? y1 : fp->height, &sh);
	}
	if (sh.shadow && out->shadow_map) memcpy(out->shadow_map, sh.shadow, (size_t)S * S * 4);
	for (uint32_t mi = 0; mi < sc->nMesh; mi++) { free(sh.clip[mi]); free(sh.clipLight[mi]); }
	free(sh.clip); free(sh.clipLight); free(sh.shadow);
	return bad;
}

/* threads ora_render_bands will use */
ORA_API int ora_band_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

ORA_API void ora_set_band_threads(int n)
{
#ifdef _OPENMP
	if (n > 0) omp_set_num_threads(n);
#else
	(void)n;
#endif
}

/* One whole frame. */
ORA_API int ora_render(const ora_scene* sc, const vkr_frame_params* fp, const vkr_frame_out* out)
{
	return ora_render_rows(sc, fp, out, 0, fp->height);
}

/* ------------------------------------------------------------------------ */
/* single-function entry points for known-answer tests                       */
/* ------------------------------------------------------------------------ */
ORA_API uint16_t ora_float_to_half(float f) { return float_to_half(f); }
ORA_API float ora_half_to_float(uint16_t h) { return half_to_float(h); }
ORA_API uint8_t ora_srgb_encode(float x) { init_tables(); return srgb_encode(clamp01(x)); }
ORA_API float ora_srgb_decode(uint8_t v) { init_tables(); return g_srgb_decode[v]; }
ORA_API float ora_linearise_depth(float d) { return linearise_depth(d); }
ORA_API void ora_lod_split(float rho2, int q, int* d, float* f) { lod_split(rho2, q, d, f); }
ORA_API void ora_brdf(const float L[3], const float V[3], const float N[3], const float albedo[3], float metal,
                      float roughTex, float nhFloor, float out[3])
{
	brdf_eval(L, V, N, albedo, metal, roughTex, nhFloor, out);
}
ORA_API void ora_mat4_inverse(const float m[16], float out[16]) { mat4_inverse(m, out); }
ORA_API void ora_mat4_mul(const float a[16], const float b[16], float out[16]) { mat4_mul(a, b, out); }
ORA_API void ora_xform(const float M[16], const float p[3], float out[4]) { xform(M, p, out); }
/* coverage of one clip-space triangle over a WxH frame: writes 1 per covered
 * pixel into mask (caller-zeroed), returns the number of sub-triangles. */
ORA_API int ora_coverage(const float c0[4], const float c1[4], const float c2[4], uint32_t W, uint32_t H, int cull,
                         uint8_t* mask, float* depth)
{
	TriPlanes pl; SubTri st[8];
	int n = tri_setup(c0, c1, c2, W, H, cull, &pl, st);
	for (int k = 0; k < n; k++) {
		const SubTri* s = &st[k];
		for (uint32_t py = 0; py < H; py++)
			for (uint32_t px = 0; px < W; px++) {
				int64_t PX = (int64_t)px * 256 + 128, PY = (int64_t)py * 256 + 128;
				int inside = 1;
				for (int i = 0; i < 3; i++) {
					int j = (i + 1) % 3;
					int64_t dx = s->X[j] - s->X[i], dy = s->Y[j] - s->Y[i];
					int tl = (dy == 0 && dx > 0) || (dy < 0);
					int64_t E = dx * (PY - s->Y[i]) - dy * (PX - s->X[i]);
					if (E < 0 || (E == 0 && !tl)) { inside = 0; break; }
				}
				if (inside) {
					mask[(size_t)py * W + px]++;
					if (depth) depth[(size_t)py * W + px] = depth_at(&pl, (int)px, (int)py);
				}
			}
	}
	return n;
}
