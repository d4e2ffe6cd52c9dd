// vkr_api.cpp — host side of the renderer behind the C ABI of
// include/vkr_renderer.h: context and scene lifetime, the per-frame mode
// dispatch that stands in for record_commands() (main/main.cpp:3709-4324),
// the uniform helpers, stage timing.
#include "../../include/vkr_renderer.h"
#include "vkr_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <mutex>
#include <vector>

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int set_error(int code, const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_CHECK(x)                                                                                        \
	do {                                                                                                    \
		hipError_t e_ = (x);                                                                                \
		if (e_ != hipSuccess) return set_error(VKR_E_DEVICE, "%s failed: %s", #x, hipGetErrorString(e_));   \
	} while (0)

extern "C" const char* vkr_last_error(void) { return g_err; }

// for the other translation units of the library
int vkr_set_error_ext(int code, const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

// No exception may leave the library: the containers used on the host side can throw (std::bad_alloc,
// std::length_error); every entry point that touches them runs its body under this guard.
template <class F> static int guarded(const char* fn, F&& body) noexcept
{
	try { return body(); }
	catch (const std::bad_alloc&) { return set_error(VKR_E_NOMEM, "%s: out of host memory", fn); }
	catch (const std::exception& e) { return set_error(VKR_E_INVALID, "%s: %s", fn, e.what()); }
	catch (...) { return set_error(VKR_E_INVALID, "%s: unknown exception", fn); }
}

// ---------------------------------------------------------------------------
// stages
// ---------------------------------------------------------------------------
enum Stage {
	ST_SETUP, ST_BIN_SCAN, ST_BIN_FILL, ST_RASTER_VIS, ST_RASTER_SHADOW, ST_RASTER_COUNT,
	ST_OVERSHADE, ST_RESOLVE_GBUF, ST_RESOLVE_FORWARD, ST_RESOLVE_DEBUG, ST_DEFERRED_LIGHT, ST_OVERVIS_COLOUR, ST_MOSAIC,
	ST_COPY_OUT, ST_RESOLVE_LIGHT, ST_COUNT
};
static const char* kStageNames[ST_COUNT] = {
	"k_setup", "k_scan_tiles", "k_bin_fill", "k_raster_vis", "k_raster_shadow", "k_raster_count",
	"k_overshade", "k_resolve_gbuf", "k_resolve_forward", "k_resolve_debug", "k_deferred_light", "k_overvis_colour",
	"k_mosaic", "copy_out", "k_resolve_light"
};

// The allocation-free entry point (vkr_render_static) draws every device buffer of a frame from one fixed array that
// lives in the library's device image: file scope, a stated maximum size, one copy per GPU, held until the process
// ends.  In host code the array's name only stands in for the symbol; its device address is looked up once per device
// (static_scratch_base) and everything - kernel arguments, memsets, copies - works on that address.
static __device__ __align__(256) unsigned char scratch[VKR_STATIC_SCRATCH_BYTES];

struct Arena {
	char* base = nullptr;
	size_t size = 0, used = 0;
};

struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
	Arena* arena = nullptr; // set: the buffer is a slice of the static scratch array, handed out anew for every frame
	int ensure(size_t bytes)
	{
		if (p && bytes <= cap) return VKR_OK;
		if (arena) {
			const size_t at = (arena->used + 255u) & ~(size_t)255u;
			if (at + bytes > arena->size)
				return set_error(VKR_E_LIMIT, "vkr_render_static: the frame needs more than the %zu bytes of static scratch (%zu in use, %zu more asked for); "
				                              "render it with vkr_render", arena->size, arena->used, bytes);
			p = arena->base + at; cap = bytes; arena->used = at + bytes;
			return VKR_OK;
		}
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		size_t want = bytes + bytes / 8 + 256;
		hipError_t e = hipMalloc(&p, want);
		if (e != hipSuccess) { p = nullptr; return set_error(VKR_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
		cap = want;
		return VKR_OK;
	}
	void release() { if (p && !arena) (void)hipFree(p); p = nullptr; cap = 0; }
	template <class T> T* as() const { return (T*)p; }
};

struct Timed { int stage; hipEvent_t a, b; };

struct PendingFrame {
	const vkr_scene* scene;
	vkr_frame_params params;
	vkr_frame_out out;
	uint32_t statusSlot; // first of the frame's status slots
	uint32_t nPasses;
};

struct PassStatus { uint32_t recCount, overflow, binTotal, _pad; };

struct vkr_context {
	int device = 0;
	hipStream_t stream = nullptr;
	float* dDecode = nullptr; // 256
	float* dThresh = nullptr; // 255
	float hDecode[256], hThresh[255];
	// geometry scratch
	DevBuf clip, planes, recs, recTiles, recLocal, clipQueue, bigQueue, zeroSets, tileOffset, cursor, tileFlags, binList, binKeys, tileOrder;
	// counters + per-tile counts of a pass, twice: a pass uses one set while its k_setup zeroes the other for the next pass
	// layout of a set: 16 counters ([0] recCount, [1] overflow, [2] alpha tile count, [3] clip queue length,
	// [4] big-record queue length, [5] deep alpha tile count), then tileCount, tileCountA, tileCountBig (zeroCapTiles each)
	uint32_t zeroCapTiles = 0; int zset = 0;
	uint32_t* zs(int which) const { return (uint32_t*)zeroSets.p + (size_t)which * (16 + 3 * (size_t)zeroCapTiles); }
	uint32_t* counters() const { return zs(zset); }
	uint32_t* tileCount() const { return zs(zset) + 16; }
	uint32_t* tileCountA() const { return zs(zset) + 16 + zeroCapTiles; }
	uint32_t* tileCountBig() const { return zs(zset) + 16 + 2 * (size_t)zeroCapTiles; }
	uint32_t recCap = 0, binCap = 0;
	bool tileOrdered = false; // the last geometry pass left its tiles listed by descending bin length in tileOrder
	double capScale = 1.0;
	uint64_t binFloor = 0; // bin entries a pass has been seen to need (from an overflowed attempt)
	// frame buffers
	DevBuf vis, depth, triId, gN, gA, colour, colour2, stencil, mip, shadow;
	// status
	PassStatus* hStatus = nullptr; // pinned
	uint32_t statusCap = 0, statusUsed = 0;
	std::vector<PendingFrame> pending;
	// profiling
	int profiling = 0; // 0 off, 1 every stage, 2 the lighting stage only
	std::vector<Timed> timed;
	std::vector<hipEvent_t> eventPool;
	double stageMs[ST_COUNT] = { 0 };
	uint64_t stageLaunches[ST_COUNT] = { 0 };
	// frames in flight: lane 0 is this context, lanes 1.. are contexts of their own (stream, scratch, frame buffers)
	// on the same device that only this one hands frames to
	std::vector<vkr_context*> lanes;
	vkr_context* parent = nullptr;
	uint32_t nextLane = 0, lastLane = 0; // lane the next frame goes to / the last one went to
	// forward path: the light's depth pass runs on a helper context (stream + geometry scratch + the shadow map of its
	// own) beside the camera's geometry and raster passes; the two streams fork and join through these events
	vkr_context* shadowHelper = nullptr;
	hipEvent_t evFork = nullptr, evJoin = nullptr;
	// allocation-free frames (vkr_render_static): a context of its own kind that borrows this one's stream and tables,
	// takes every device buffer from the static scratch array and never grows anything
	vkr_context* staticCtx = nullptr;
	bool isStatic = false;
	Arena arena;
};

// every device buffer a context owns
#define VKR_CTX_NBUFS 24
static void ctx_bufs(vkr_context* c, DevBuf* out[VKR_CTX_NBUFS])
{
	DevBuf* all[VKR_CTX_NBUFS] = { &c->clip, &c->planes, &c->recs, &c->recTiles, &c->recLocal, &c->clipQueue, &c->bigQueue, &c->zeroSets,
		&c->tileOffset, &c->cursor, &c->tileFlags, &c->binList, &c->binKeys, &c->vis, &c->depth, &c->triId, &c->gN, &c->gA, &c->colour,
		&c->colour2, &c->stencil, &c->mip, &c->shadow, &c->tileOrder };
	for (int i = 0; i < VKR_CTX_NBUFS; i++) out[i] = all[i];
}

struct HostTex { uint32_t w[VKR_MAX_LEVELS], h[VKR_MAX_LEVELS], off[VKR_MAX_LEVELS]; int levels; int srgb; };

struct vkr_scene {
	vkr_context* ctx = nullptr;
	DevBuf pos, uv, tcomp, tbn, tri, triMat, orderB, tex, mat, texels, alphaTexels, alphaAux, triShade;
	SceneDev dev;
	std::vector<HostTex> htex;
	uint64_t nTri = 0, nVert = 0, texelCount = 0;
	bool anyAlphaTest = false; // some material's alpha texture can fail the 0.5 test
};

// ---------------------------------------------------------------------------
// sRGB tables: decode = EOTF(i/255); thresh[k] = EOTF((k+0.5)/255), both
// evaluated in double and rounded to float once.
// ---------------------------------------------------------------------------
static double srgb_eotf(double c) { return (c <= 0.04045) ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4); }

static uint32_t host_srgb_encode(const float* thresh, float x)
{
	if (!(x > 0.0f)) return 0;
	int lo = 0, hi = 255;
	while (lo < hi) {
		int mid = (lo + hi) >> 1;
		if (x >= thresh[mid]) lo = mid + 1; else hi = mid;
	}
	return (uint32_t)lo;
}

// ---------------------------------------------------------------------------
// glm-compatible host math (same operation order as glm 1.0.2 so that the
// uniform blocks equal the reference's bit for bit)
// ---------------------------------------------------------------------------
static void mat4_mul(const float a[16], const float b[16], float out[16])
{
	float r[16];
	for (int c = 0; c < 4; c++)
		for (int i = 0; i < 4; i++) {
			float t = a[0 * 4 + i] * b[c * 4 + 0];
			t += a[1 * 4 + i] * b[c * 4 + 1];
			t += a[2 * 4 + i] * b[c * 4 + 2];
			t += a[3 * 4 + i] * b[c * 4 + 3];
			r[c * 4 + i] = t;
		}
	memcpy(out, r, sizeof r);
}

static void mat4_inverse(const float m[16], float out[16])
{
	auto M = [&](int c, int r) { return m[c * 4 + r]; };
	// 2x2 minors of rows 1..3, named by the column pair they skip
	float s00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), s02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3), s03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
	float s04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3), s06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3), s07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
	float s08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), s10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2), s11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
	float s12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3), s14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3), s15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
	float s16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), s18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2), s19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
	float s20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1), s22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1), s23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
	const float F0[4] = { s00, s00, s02, s03 }, F1[4] = { s04, s04, s06, s07 }, F2[4] = { s08, s08, s10, s11 };
	const float F3[4] = { s12, s12, s14, s15 }, F4[4] = { s16, s16, s18, s19 }, F5[4] = { s20, s20, s22, s23 };
	const float V0[4] = { M(1, 0), M(0, 0), M(0, 0), M(0, 0) }, V1[4] = { M(1, 1), M(0, 1), M(0, 1), M(0, 1) };
	const float V2[4] = { M(1, 2), M(0, 2), M(0, 2), M(0, 2) }, V3[4] = { M(1, 3), M(0, 3), M(0, 3), M(0, 3) };
	float inv[16];
	for (int i = 0; i < 4; i++) {
		const float sgnA = (i & 1) ? -1.0f : 1.0f, sgnB = -sgnA;
		inv[0 * 4 + i] = ((V1[i] * F0[i] - V2[i] * F1[i]) + V3[i] * F2[i]) * sgnA;
		inv[1 * 4 + i] = ((V0[i] * F0[i] - V2[i] * F3[i]) + V3[i] * F4[i]) * sgnB;
		inv[2 * 4 + i] = ((V0[i] * F1[i] - V1[i] * F3[i]) + V3[i] * F5[i]) * sgnA;
		inv[3 * 4 + i] = ((V0[i] * F2[i] - V1[i] * F4[i]) + V2[i] * F5[i]) * sgnB;
	}
	const float det = (M(0, 0) * inv[0] + M(0, 1) * inv[4]) + (M(0, 2) * inv[8] + M(0, 3) * inv[12]);
	const float ood = 1.0f / det;
	for (int i = 0; i < 16; i++) out[i] = inv[i] * ood;
}

static void perspective_rh_zo(float fovy, float aspect, float zn, float zf, float out[16])
{
	const float th = tanf(fovy / 2.0f);
	memset(out, 0, sizeof(float) * 16);
	out[0] = 1.0f / (aspect * th);
	out[5] = 1.0f / th;
	out[10] = zf / (zn - zf);
	out[11] = -1.0f;
	out[14] = -(zf * zn) / (zf - zn);
}

static float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void norm3(const float a[3], float o[3])
{
	const float inv = 1.0f / sqrtf(dot3(a, a));
	o[0] = a[0] * inv; o[1] = a[1] * inv; o[2] = a[2] * inv;
}
static void cross3(const float x[3], const float y[3], float o[3])
{
	o[0] = x[1] * y[2] - y[1] * x[2];
	o[1] = x[2] * y[0] - y[2] * x[0];
	o[2] = x[0] * y[1] - y[0] * x[1];
}

static void look_at_rh(const float eye[3], const float center[3], const float up[3], float out[16])
{
	float d[3] = { center[0] - eye[0], center[1] - eye[1], center[2] - eye[2] }, f[3], s[3], u[3], t[3];
	norm3(d, f);
	cross3(f, up, t);
	norm3(t, s);
	cross3(s, f, u);
	memset(out, 0, sizeof(float) * 16);
	out[15] = 1.0f;
	out[0] = s[0]; out[4] = s[1]; out[8] = s[2];
	out[1] = u[0]; out[5] = u[1]; out[9] = u[2];
	out[2] = -f[0]; out[6] = -f[1]; out[10] = -f[2];
	out[12] = -dot3(s, eye); out[13] = -dot3(u, eye); out[14] = dot3(f, eye);
}

extern "C" void vkr_update_scene_uniforms(vkr_scene_uniform* u, uint32_t width, uint32_t height, const float camera2world[16])
{
	// main/main.cpp:3681-3693; kCameraFov 60 deg -> rad as utils/angle.hpp:11-14, near/far main/main.cpp:75-76
	const float pi = 3.14159265358979323846f;
	const float fov = 60.0f * pi / 180.0f;
	const float aspect = (float)width / (float)height;
	perspective_rh_zo(fov, aspect, 0.1f, 100.0f, u->projection);
	u->projection[5] *= -1.0f;
	mat4_inverse(camera2world, u->camera);
	mat4_mul(u->projection, u->camera, u->projCam);
	u->camPos[0] = camera2world[12]; u->camPos[1] = camera2world[13]; u->camPos[2] = camera2world[14]; u->camPos[3] = 1.0f;
}

extern "C" void vkr_update_depth_mvp_uniforms(vkr_depth_mvp* d, const float lightPos[4])
{
	// main/main.cpp:3699-3707.  lut::Radians(90.0f) is 90 radians (utils/angle.hpp:65), kept as is.
	float proj[16], view[16];
	perspective_rh_zo(90.0f, 1.0f, 0.1f, 100.0f, proj);
	proj[5] *= -1.0f;
	const float eye[3] = { lightPos[0], lightPos[1], lightPos[2] }, center[3] = { 0.0f, 0.0f, -48.0f }, up[3] = { 0.0f, 1.0f, 0.0f };
	look_at_rh(eye, center, up, view);
	mat4_mul(proj, view, d->depthMVP);
}

extern "C" void vkr_default_lights(vkr_light_uniform* fwd, vkr_multiple_lights* ml)
{
	// main/main.cpp:1313-1352
	static const float P[VKR_NUM_LIGHTS][3] = {
		{ 8.0377f, -1.1000f, -15.8845f }, { 5.6632f, -1.1000f, -12.2500f }, { 2.1255f, -1.1000f, -10.1000f },
		{ -2.0867f, -1.1000f, -10.2500f }, { -5.7367f, -1.1000f, -12.2000f }, { -7.8244f, -1.1000f, -15.8500f },
		{ 3.0632f, -1.1000f, -25.8500f }, { -3.0632f, -1.1000f, -25.8500f }, { -7.4867f, -1.1000f, -36.0500f },
		{ -2.4367f, -3.1000f, -46.5500f }, { 2.3632f, -3.1000f, -46.5500f }, { -0.0514f, -3.9000f, -49.6000f },
		{ -7.2367f, -3.1000f, -61.2500f }, { 7.1088f, -3.1000f, -61.2500f }, { -2.0867f, -3.1000f, -66.5500f },
		{ 2.0867f, -3.1000f, -66.5500f }, { -2.0867f, -3.1000f, -68.9000f }, { 2.0867f, -3.1000f, -68.9000f },
		{ 0.0000f, -4.1000f, -76.0000f }, { -1.6867f, -3.1000f, -88.8500f }, { 1.6867f, -3.1000f, -88.8500f } };
	if (fwd) {
		fwd->lightPos[0] = -0.2972f; fwd->lightPos[1] = 7.3100f; fwd->lightPos[2] = -11.9532f; fwd->lightPos[3] = 0.0f;
		for (int i = 0; i < 4; i++) fwd->lightColour[i] = 1.0f;
	}
	if (ml)
		for (int i = 0; i < VKR_NUM_LIGHTS; i++) {
			for (int k = 0; k < 3; k++) ml->light[i].lightPos[k] = P[i][k];
			ml->light[i].lightPos[3] = 0.0f;
			ml->light[i].lightColour[0] = 0.9882f; ml->light[i].lightColour[1] = 0.4549f;
			ml->light[i].lightColour[2] = 0.0196f; ml->light[i].lightColour[3] = 1.0f;
		}
}

extern "C" void vkr_make_uniforms(vkr_uniforms* u, uint32_t width, uint32_t height, const float camera2world[16], int32_t debug)
{
	memset(u, 0, sizeof *u);
	vkr_update_scene_uniforms(&u->sceneUniforms, width, height, camera2world);
	u->debugUniforms.debug = debug;
	vkr_default_lights(&u->lightUniforms, &u->multipleLightsUniform);
	vkr_update_depth_mvp_uniforms(&u->depthMVPUniform, u->lightUniforms.lightPos);
}

// cam = cam * glm::rotate(angle, axis) for a unit axis (gtx/transform.inl:12-15 over
// ext/matrix_transform.inl:18-46): the rotation is first applied to an identity
// matrix, zeros and ones included, so that signed zeros come out as glm's do.
static void post_rotate(float cam[16], float angle, const float axis[3])
{
	const float c = cosf(angle), s = sinf(angle);
	const float temp[3] = { (1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2] };
	float R[3][3];
	R[0][0] = c + temp[0] * axis[0];
	R[0][1] = temp[0] * axis[1] + s * axis[2];
	R[0][2] = temp[0] * axis[2] - s * axis[1];
	R[1][0] = temp[1] * axis[0] - s * axis[2];
	R[1][1] = c + temp[1] * axis[1];
	R[1][2] = temp[1] * axis[2] + s * axis[0];
	R[2][0] = temp[2] * axis[0] + s * axis[1];
	R[2][1] = temp[2] * axis[1] - s * axis[0];
	R[2][2] = c + temp[2] * axis[2];
	static const float I[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
	float rot[16];
	for (int col = 0; col < 3; col++)
		for (int i = 0; i < 4; i++)
			rot[col * 4 + i] = (I[0 + i] * R[col][0] + I[4 + i] * R[col][1]) + I[8 + i] * R[col][2];
	for (int i = 0; i < 4; i++) rot[12 + i] = I[12 + i];
	mat4_mul(cam, rot, cam);
}

// cam = cam * glm::translate(v) (gtx/transform.inl:6-9 over ext/matrix_transform.inl:10-15)
static void post_translate(float cam[16], float x, float y, float z)
{
	static const float I[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
	float tr[16];
	memcpy(tr, I, sizeof tr);
	for (int i = 0; i < 4; i++) tr[12 + i] = ((I[0 + i] * x + I[4 + i] * y) + I[8 + i] * z) + I[12 + i];
	mat4_mul(cam, tr, cam);
}

extern "C" void vkr_user_state_init(vkr_user_state* st)
{
	memset(st, 0, sizeof *st);
	st->camera2world[0] = st->camera2world[5] = st->camera2world[10] = st->camera2world[15] = 1.0f;
	st->debugVisualisation = 1;
}

// glfw_callback_key_press, main/main.cpp:1422-1514 (key and action values of GLFW/glfw3.h)
extern "C" int vkr_user_state_key(vkr_user_state* st, int key, int action)
{
	if (!st) return 0;
	const bool isReleased = (action == 0);
	if (isReleased) {
		switch (key) {
		case '1': case '2': case '3': case '4': st->debugVisualisation = key - '0'; break;
		case '5': st->mosaicEffect = !st->mosaicEffect; break;
		case '6': st->debugVisualisation = 5; break;
		case '7': st->debugVisualisation = 6; break;
		case '8': st->deferredShading = !st->deferredShading; break;
		case '9': st->shadows = !st->shadows; break;
		default: break;
		}
	}
	switch (key) {
	case 'W': st->inputMap[VKR_INPUT_FORWARD] = !isReleased; break;
	case 'S': st->inputMap[VKR_INPUT_BACKWARD] = !isReleased; break;
	case 'A': st->inputMap[VKR_INPUT_STRAFE_LEFT] = !isReleased; break;
	case 'D': st->inputMap[VKR_INPUT_STRAFE_RIGHT] = !isReleased; break;
	case 'E': st->inputMap[VKR_INPUT_LEVITATE] = !isReleased; break;
	case 'Q': st->inputMap[VKR_INPUT_SINK] = !isReleased; break;
	case 340: case 344: st->inputMap[VKR_INPUT_FAST] = !isReleased; break; // left / right shift
	case 341: case 345: st->inputMap[VKR_INPUT_SLOW] = !isReleased; break; // left / right control
	default: break;
	}
	return (key == 256 && action == 1) ? 1 : 0; // escape pressed
}

// glfw_callback_button, main/main.cpp:1516-1529
extern "C" void vkr_user_state_button(vkr_user_state* st, int button, int action)
{
	if (st && button == 1 && action == 1) st->inputMap[VKR_INPUT_MOUSING] = !st->inputMap[VKR_INPUT_MOUSING];
}

// glfw_callback_motion, main/main.cpp:1531-1537
extern "C" void vkr_user_state_motion(vkr_user_state* st, double x, double y)
{
	if (!st) return;
	st->mouseX = (float)x;
	st->mouseY = (float)y;
}

extern "C" void vkr_frame_params_set_modes(vkr_frame_params* fp, const vkr_user_state* st)
{
	if (!fp || !st) return;
	fp->debugVisualisation = st->debugVisualisation;
	fp->mosaicEffect = st->mosaicEffect;
	fp->deferredShading = st->deferredShading;
}

extern "C" void vkr_update_user_state(vkr_user_state* st, float elapsed)
{
	// main/main.cpp:3643-3679 with the constants of main/main.cpp:76-84
	const float kBaseSpeed = 1.7f, kFastMult = 7.0f, kSlowMult = 0.05f, kMouseSensitivity = 0.01f;
	float* cam = st->camera2world;
	if (st->inputMap[VKR_INPUT_MOUSING]) {
		if (st->wasMousing) {
			const float dx = kMouseSensitivity * (st->mouseX - st->previousX);
			const float dy = kMouseSensitivity * (st->mouseY - st->previousY);
			const float xAxis[3] = { 1.0f, 0.0f, 0.0f }, yAxis[3] = { 0.0f, 1.0f, 0.0f };
			post_rotate(cam, -dy, xAxis);
			post_rotate(cam, -dx, yAxis);
		}
		st->previousX = st->mouseX;
		st->previousY = st->mouseY;
		st->wasMousing = 1;
	} else
		st->wasMousing = 0;

	const float move = elapsed * kBaseSpeed * (st->inputMap[VKR_INPUT_FAST] ? kFastMult : 1.0f) *
	                   (st->inputMap[VKR_INPUT_SLOW] ? kSlowMult : 1.0f);
	if (st->inputMap[VKR_INPUT_FORWARD]) post_translate(cam, 0.0f, 0.0f, -move);
	if (st->inputMap[VKR_INPUT_BACKWARD]) post_translate(cam, 0.0f, 0.0f, +move);
	if (st->inputMap[VKR_INPUT_STRAFE_LEFT]) post_translate(cam, -move, 0.0f, 0.0f);
	if (st->inputMap[VKR_INPUT_STRAFE_RIGHT]) post_translate(cam, +move, 0.0f, 0.0f);
	if (st->inputMap[VKR_INPUT_LEVITATE]) post_translate(cam, 0.0f, +move, 0.0f);
	if (st->inputMap[VKR_INPUT_SINK]) post_translate(cam, 0.0f, -move, 0.0f);
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
// Device address of the static scratch array on `device` (which must be current), looked up once per device.
#define VKR_MAX_DEVICES 64
static std::mutex g_scratchMutex;                    // guards the two tables below
static void* g_scratchBase[VKR_MAX_DEVICES];
// The array is one per GPU and every context of that GPU cuts its allocation-free frames from it: the context whose
// frames are pending owns it until its fence (vkr_wait) or its destruction; a second context that submits meanwhile
// is refused instead of racing.
static vkr_context* g_scratchOwner[VKR_MAX_DEVICES];
static int static_scratch_acquire(vkr_context* t)
{
	std::lock_guard<std::mutex> lock(g_scratchMutex);
	vkr_context*& owner = g_scratchOwner[t->device];
	if (owner && owner != t)
		return set_error(VKR_E_LIMIT, "vkr_render_static: another context has allocation-free frames pending on GPU %d, and the GPU's "
		                              "static scratch array holds one context's frames at a time; call vkr_wait on that context first", t->device);
	owner = t;
	return VKR_OK;
}
static void static_scratch_release(vkr_context* t)
{
	std::lock_guard<std::mutex> lock(g_scratchMutex);
	if (t->device >= 0 && t->device < VKR_MAX_DEVICES && g_scratchOwner[t->device] == t) g_scratchOwner[t->device] = nullptr;
}
static int static_scratch_base(int device, void** base)
{
	if (device < 0 || device >= VKR_MAX_DEVICES) return set_error(VKR_E_LIMIT, "device %d: the static scratch table holds %d devices", device, VKR_MAX_DEVICES);
	std::lock_guard<std::mutex> lock(g_scratchMutex);
	if (!g_scratchBase[device]) {
		void* ptr = nullptr;
		hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
		if (e != hipSuccess || !ptr) return set_error(VKR_E_DEVICE, "hipGetSymbolAddress(scratch) on device %d failed: %s", device, hipGetErrorString(e));
		g_scratchBase[device] = ptr;
	}
	*base = g_scratchBase[device];
	return VKR_OK;
}

// kind: 0 = a context of the caller's (vkr_context_create), 1 = a lane or helper of `owner` (stream, scratch and tables
// of its own), 2 = the allocation-free context of `owner` (borrows its stream and tables, device buffers from the
// static scratch array)
static int context_create_impl(int device, vkr_context** out, int kind, vkr_context* owner)
{
	*out = nullptr;
	int count = 0;
	HIP_CHECK(hipGetDeviceCount(&count));
	if (device < 0 || device >= count) return set_error(VKR_E_INVALID, "vkr_context_create: device %d of %d", device, count);
	HIP_CHECK(hipSetDevice(device));
	vkr_context* c = new (std::nothrow) vkr_context();
	if (!c) return set_error(VKR_E_NOMEM, "out of host memory");
	c->device = device;
	c->parent = owner;
	for (int i = 0; i < 256; i++) c->hDecode[i] = (float)srgb_eotf((double)i / 255.0);
	for (int i = 0; i < 255; i++) c->hThresh[i] = (float)srgb_eotf(((double)i + 0.5) / 255.0);
	int rc = VKR_OK;
	hipError_t e = hipSuccess;
	if (kind == 2) {
		c->isStatic = true;
		c->stream = owner->stream; c->dDecode = owner->dDecode; c->dThresh = owner->dThresh;
		void* base = nullptr;
		rc = static_scratch_base(device, &base);
		if (!rc) {
			c->arena.base = (char*)base; c->arena.size = VKR_STATIC_SCRATCH_BYTES; c->arena.used = 0;
			DevBuf* b[VKR_CTX_NBUFS];
			ctx_bufs(c, b);
			for (DevBuf* d : b) d->arena = &c->arena;
			// everything a frame needs on the host side exists before the first frame: status slots, the pending list
			c->pending.reserve(VKR_STATIC_MAX_PENDING);
			e = hipHostMalloc((void**)&c->hStatus, sizeof(PassStatus) * 2 * VKR_STATIC_MAX_PENDING, hipHostMallocDefault);
			if (e == hipSuccess) c->statusCap = 2 * VKR_STATIC_MAX_PENDING;
		}
	} else {
		e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
		if (e == hipSuccess) e = hipMalloc((void**)&c->dDecode, 256 * sizeof(float));
		if (e == hipSuccess) e = hipMalloc((void**)&c->dThresh, 256 * sizeof(float));
		if (e == hipSuccess) e = hipMemcpy(c->dDecode, c->hDecode, 256 * sizeof(float), hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMemcpy(c->dThresh, c->hThresh, 255 * sizeof(float), hipMemcpyHostToDevice);
	}
	if (e != hipSuccess) rc = set_error(VKR_E_DEVICE, "vkr_context_create: %s", hipGetErrorString(e));
	if (!rc && kind == 0) rc = context_create_impl(device, &c->staticCtx, 2, c);
	if (rc) {
		const std::string keep = g_err;
		vkr_context_destroy(c);
		snprintf(g_err, sizeof g_err, "%s", keep.c_str());
		return rc;
	}
	*out = c;
	return VKR_OK;
}

extern "C" int vkr_context_create(int device, vkr_context** out)
{
	if (!out) return set_error(VKR_E_INVALID, "vkr_context_create: out is NULL");
	return guarded("vkr_context_create", [&] { return context_create_impl(device, out, 0, nullptr); });
}

extern "C" void vkr_context_destroy(vkr_context* c)
{
	if (!c) return;
	for (vkr_context* l : c->lanes) vkr_context_destroy(l);
	c->lanes.clear();
	if (c->shadowHelper) { c->shadowHelper->hStatus = nullptr; vkr_context_destroy(c->shadowHelper); c->shadowHelper = nullptr; }
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->staticCtx) { vkr_context_destroy(c->staticCtx); c->staticCtx = nullptr; }
	if (c->evFork) (void)hipEventDestroy(c->evFork);
	if (c->evJoin) (void)hipEventDestroy(c->evJoin);
	DevBuf* bufs[VKR_CTX_NBUFS];
	ctx_bufs(c, bufs);
	for (DevBuf* b : bufs) b->release();
	for (auto& t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
	for (auto e : c->eventPool) (void)hipEventDestroy(e);
	if (c->hStatus) (void)hipHostFree(c->hStatus);
	if (c->isStatic) static_scratch_release(c); // the stream was drained above
	if (!c->isStatic) { // the allocation-free context borrows its owner's tables and stream
		if (c->dDecode) (void)hipFree(c->dDecode);
		if (c->dThresh) (void)hipFree(c->dThresh);
		if (c->stream) (void)hipStreamDestroy(c->stream);
	}
	delete c;
}

extern "C" void* vkr_context_stream(vkr_context* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int vkr_context_device(vkr_context* c) { return c ? c->device : -1; }
extern "C" int vkr_context_frame_lane(vkr_context* c) { return (c && c->lastLane <= c->lanes.size()) ? (int)c->lastLane : 0; }
extern "C" void* vkr_context_frame_stream(vkr_context* c)
{
	if (!c) return nullptr;
	const uint32_t k = c->lastLane;
	return (void*)((k && k <= c->lanes.size()) ? c->lanes[k - 1]->stream : c->stream);
}

extern "C" int vkr_device_malloc(vkr_context* c, size_t bytes, void** out)
{
	if (!c || !out) return set_error(VKR_E_INVALID, "vkr_device_malloc: NULL argument");
	*out = nullptr;
	HIP_CHECK(hipSetDevice(c->device));
	hipError_t e = hipMalloc(out, bytes ? bytes : 1);
	if (e != hipSuccess) { *out = nullptr; return set_error(VKR_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
	return VKR_OK;
}

extern "C" void vkr_device_free(vkr_context* c, void* ptr)
{
	if (!c || !ptr) return;
	(void)hipSetDevice(c->device);
	(void)hipFree(ptr);
}

extern "C" int vkr_device_copy(vkr_context* c, void* dst, const void* src, size_t bytes)
{
	if (!c || (bytes && (!dst || !src))) return set_error(VKR_E_INVALID, "vkr_device_copy: NULL argument");
	HIP_CHECK(hipSetDevice(c->device));
	if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->stream));
	HIP_CHECK(hipStreamSynchronize(c->stream));
	return VKR_OK;
}

// ---------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------
static hipEvent_t get_event(vkr_context* c)
{
	if (!c->eventPool.empty()) { hipEvent_t e = c->eventPool.back(); c->eventPool.pop_back(); return e; }
	hipEvent_t e = nullptr;
	if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	return e;
}

struct StageTimer {
	vkr_context* c; Timed t; bool on;
	StageTimer(vkr_context* ctx, int stage) : c(ctx), on(ctx->profiling == 1 || (ctx->profiling == 2 && (stage == ST_DEFERRED_LIGHT || stage == ST_RESOLVE_LIGHT)))
	{
		if (ctx->isStatic) on = false; // allocation-free frames create no events
		if (!on) return;
		t.stage = stage; t.a = get_event(c); t.b = get_event(c);
		if (!t.a || !t.b) {
			// no event to be had: this stage goes untimed rather than recording a null event
			if (t.a) c->eventPool.push_back(t.a);
			if (t.b) c->eventPool.push_back(t.b);
			on = false;
			return;
		}
		(void)hipEventRecord(t.a, c->stream);
	}
	~StageTimer()
	{
		if (!on) return;
		(void)hipEventRecord(t.b, c->stream);
		c->timed.push_back(t);
	}
};

static void drain_timers(vkr_context* c)
{
	for (auto& t : c->timed) {
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { c->stageMs[t.stage] += ms; c->stageLaunches[t.stage]++; }
		c->eventPool.push_back(t.a); c->eventPool.push_back(t.b);
	}
	c->timed.clear();
}

extern "C" int vkr_profile_enable(vkr_context* c, int on)
{
	if (!c) return set_error(VKR_E_INVALID, "NULL context");
	c->profiling = on;
	if (c->shadowHelper) c->shadowHelper->profiling = on;
	for (vkr_context* l : c->lanes) { l->profiling = on; if (l->shadowHelper) l->shadowHelper->profiling = on; }
	return VKR_OK;
}
static int profile_reset_impl(vkr_context* c)
{
	if (!c) return set_error(VKR_E_INVALID, "NULL context");
	HIP_CHECK(hipSetDevice(c->device));
	HIP_CHECK(hipStreamSynchronize(c->stream));
	drain_timers(c);
	for (int i = 0; i < ST_COUNT; i++) { c->stageMs[i] = 0; c->stageLaunches[i] = 0; }
	if (c->shadowHelper) {
		int rc = profile_reset_impl(c->shadowHelper);
		if (rc) return rc;
	}
	for (vkr_context* l : c->lanes) {
		int rc = profile_reset_impl(l);
		if (rc) return rc;
	}
	return VKR_OK;
}
extern "C" int vkr_stage_count(void) { return ST_COUNT; }
extern "C" const char* vkr_stage_name(int s) { return (s >= 0 && s < ST_COUNT) ? kStageNames[s] : ""; }
static int profile_get_impl(vkr_context* c, int stage, double* ms, uint64_t* launches)
{
	if (!c || stage < 0 || stage >= ST_COUNT) return set_error(VKR_E_INVALID, "bad stage");
	HIP_CHECK(hipSetDevice(c->device));
	HIP_CHECK(hipStreamSynchronize(c->stream));
	drain_timers(c);
	double total = c->stageMs[stage];
	uint64_t n = c->stageLaunches[stage];
	if (c->shadowHelper) {
		double lm = 0.0; uint64_t ln = 0;
		int rc = profile_get_impl(c->shadowHelper, stage, &lm, &ln);
		if (rc) return rc;
		total += lm; n += ln;
	}
	for (vkr_context* l : c->lanes) {
		double lm = 0.0; uint64_t ln = 0;
		int rc = profile_get_impl(l, stage, &lm, &ln);
		if (rc) return rc;
		total += lm; n += ln;
	}
	if (ms) *ms = total;
	if (launches) *launches = n;
	return VKR_OK;
}

// ---------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------
static int mip_level_count(uint32_t w, uint32_t h)
{
	uint32_t bits = w | h; // compute_mip_level_count, utils/vkimage.cpp:289-294
	int n = 0;
	while (bits) { n++; bits >>= 1; }
	return n;
}

extern "C" void vkr_scene_destroy(vkr_scene* s)
{
	if (!s) return;
	if (s->ctx) {
		(void)hipSetDevice(s->ctx->device);
		(void)hipStreamSynchronize(s->ctx->stream);
		for (vkr_context* l : s->ctx->lanes) (void)hipStreamSynchronize(l->stream);
	}
	DevBuf* bufs[] = { &s->pos, &s->uv, &s->tcomp, &s->tbn, &s->tri, &s->triMat, &s->orderB, &s->tex, &s->mat, &s->texels, &s->alphaTexels, &s->alphaAux, &s->triShade };
	for (DevBuf* b : bufs) b->release();
	delete s;
}

static int scene_create_impl(vkr_context* c, const vkr_scene_desc* d, vkr_scene** out)
{
	if (!c || !d || !out) return set_error(VKR_E_INVALID, "vkr_scene_create: NULL argument");
	*out = nullptr;
	HIP_CHECK(hipSetDevice(c->device));
	// ---- validate
	if ((d->textureCount && !d->textures) || (d->materialCount && !d->materials) || (d->meshCount && !d->meshes))
		return set_error(VKR_E_INVALID, "vkr_scene_create: a count is non-zero but its array is NULL");
	for (uint32_t i = 0; i < d->textureCount; i++) {
		const vkr_texture_desc& t = d->textures[i];
		if (!t.rgba8 || t.width == 0 || t.height == 0 || t.width > 32768 || t.height > 32768)
			return set_error(VKR_E_INVALID, "texture %u: bad size %ux%u or NULL data", i, t.width, t.height);
	}
	for (uint32_t i = 0; i < d->materialCount; i++) {
		const vkr_material_desc& m = d->materials[i];
		// main/baked_model.cpp:126-129 asserts these; the normal map has no fallback (main/main.cpp:770-846)
		if (m.baseColorTextureId >= d->textureCount || m.roughnessTextureId >= d->textureCount ||
		    m.metalnessTextureId >= d->textureCount || m.normalMapTextureId >= d->textureCount)
			return set_error(VKR_E_INVALID, "material %u names a texture outside [0,%u)", i, d->textureCount);
		if (m.alphaMaskTextureId != VKR_NO_TEXTURE && m.alphaMaskTextureId >= d->textureCount)
			return set_error(VKR_E_INVALID, "material %u: alpha mask texture %u out of range", i, m.alphaMaskTextureId);
	}
	uint64_t nVert = 0, nTri = 0, nTriOpaque = 0;
	for (uint32_t i = 0; i < d->meshCount; i++) {
		const vkr_mesh_desc& m = d->meshes[i];
		if (m.materialId >= d->materialCount) return set_error(VKR_E_INVALID, "mesh %u: material %u out of range", i, m.materialId);
		if (m.indexCount % 3) return set_error(VKR_E_INVALID, "mesh %u: index count %u is not a multiple of 3", i, m.indexCount);
		if (m.vertexCount && (!m.positions || !m.texcoords || !m.tangentsComp)) return set_error(VKR_E_INVALID, "mesh %u: NULL vertex array", i);
		if (m.indexCount && !m.indices) return set_error(VKR_E_INVALID, "mesh %u: NULL index array", i);
		for (uint32_t k = 0; k < m.indexCount; k++)
			if (m.indices[k] >= m.vertexCount) return set_error(VKR_E_INVALID, "mesh %u: index %u = %u >= vertexCount %u", i, k, m.indices[k], m.vertexCount);
		nVert += m.vertexCount;
		nTri += m.indexCount / 3;
		if (d->materials[m.materialId].alphaMaskTextureId == VKR_NO_TEXTURE) nTriOpaque += m.indexCount / 3;
	}
	if (nVert >= 0xfffffff0ull || nTri >= 0x1fffffffull) return set_error(VKR_E_LIMIT, "scene too large (%llu vertices, %llu triangles)", (unsigned long long)nVert, (unsigned long long)nTri);

	vkr_scene* s = new (std::nothrow) vkr_scene();
	if (!s) return set_error(VKR_E_NOMEM, "out of host memory");
	s->ctx = c; s->nTri = nTri; s->nVert = nVert;
	int rc = VKR_OK;
	// released on every exit that does not hand the scene to the caller (error returns and exceptions alike)
	struct Owner { vkr_scene* s; ~Owner() { if (s) vkr_scene_destroy(s); } } owner{ s };
	auto fail = [&](int code) { return code; };

	// ---- geometry: vertices concatenated in file order; triangles in
	// submission order A = opaque meshes first, then alpha-masked meshes
	// (main/main.cpp:1080-1081, 3873-3934)
	std::vector<float> pos(3 * nVert), uv(2 * nVert);
	std::vector<uint32_t> tcomp(nVert), tri(3 * nTri), triMat(nTri), orderB(2 * nTri);
	std::vector<uint64_t> vbase(d->meshCount), tbaseB(d->meshCount);
	{
		uint64_t v = 0, t = 0;
		for (uint32_t i = 0; i < d->meshCount; i++) {
			const vkr_mesh_desc& m = d->meshes[i];
			vbase[i] = v; tbaseB[i] = t;
			if (m.vertexCount) {
				memcpy(&pos[3 * v], m.positions, sizeof(float) * 3 * m.vertexCount);
				memcpy(&uv[2 * v], m.texcoords, sizeof(float) * 2 * m.vertexCount);
				memcpy(&tcomp[v], m.tangentsComp, sizeof(uint32_t) * m.vertexCount);
			}
			v += m.vertexCount; t += m.indexCount / 3;
		}
		uint64_t a = 0;
		for (int pass = 0; pass < 2; pass++)
			for (uint32_t i = 0; i < d->meshCount; i++) {
				const vkr_mesh_desc& m = d->meshes[i];
				const int isAlpha = d->materials[m.materialId].alphaMaskTextureId != VKR_NO_TEXTURE;
				if (isAlpha != pass) continue;
				for (uint32_t k = 0; k < m.indexCount / 3; k++, a++) {
					tri[3 * a + 0] = (uint32_t)(vbase[i] + m.indices[3 * k + 0]);
					tri[3 * a + 1] = (uint32_t)(vbase[i] + m.indices[3 * k + 1]);
					tri[3 * a + 2] = (uint32_t)(vbase[i] + m.indices[3 * k + 2]);
					triMat[a] = m.materialId;
					const uint32_t b = (uint32_t)(tbaseB[i] + k);
					orderB[a] = b;                 // A -> file-order rank
					orderB[nTri + b] = (uint32_t)a; // file-order rank -> A
				}
			}
	}
	auto upload = [&](DevBuf& b, const void* src, size_t bytes) -> int {
		int r = b.ensure(bytes ? bytes : 4);
		if (r) return r;
		if (bytes) {
			hipError_t e = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
			if (e != hipSuccess) return set_error(VKR_E_DEVICE, "scene upload: %s", hipGetErrorString(e));
		}
		return VKR_OK;
	};
	if ((rc = upload(s->pos, pos.data(), pos.size() * 4))) return fail(rc);
	if ((rc = upload(s->uv, uv.data(), uv.size() * 4))) return fail(rc);
	if ((rc = upload(s->tcomp, tcomp.data(), tcomp.size() * 4))) return fail(rc);
	if ((rc = s->tbn.ensure(nVert ? nVert * 3 * sizeof(float4) : 16))) return fail(rc);
	vkr_launch_tbn_table(c->stream, s->tcomp.as<uint32_t>(), s->tbn.as<float4>(), (uint32_t)nVert);
	if ((rc = upload(s->tri, tri.data(), tri.size() * 4))) return fail(rc);
	if ((rc = upload(s->triMat, triMat.data(), triMat.size() * 4))) return fail(rc);
	if ((rc = upload(s->orderB, orderB.data(), orderB.size() * 4))) return fail(rc);

	// ---- textures: one texel pool, level 0 uploaded, levels 1.. built on the GPU
	std::vector<TexInfo> tinfo(d->textureCount ? d->textureCount : 1);
	s->htex.resize(d->textureCount);
	uint64_t texels = 0;
	auto add_chain = [&](size_t i, uint32_t w, uint32_t hh, int srgb) {
		HostTex& h = s->htex[i];
		h.levels = mip_level_count(w, hh);
		h.srgb = srgb;
		memset(&tinfo[i], 0, sizeof(TexInfo));
		tinfo[i].wh0 = w | (hh << 16);
		for (int l = 0; l < h.levels; l++) {
			h.w[l] = w; h.h[l] = hh; h.off[l] = (uint32_t)texels;
			tinfo[i].offset[l] = (uint32_t)texels; tinfo[i].w[l] = (uint16_t)w; tinfo[i].h[l] = (uint16_t)hh;
			texels += (uint64_t)w * hh;
			w >>= 1; if (!w) w = 1;
			hh >>= 1; if (!hh) hh = 1;
		}
		tinfo[i].levels = h.levels; tinfo[i].srgb = h.srgb;
		tinfo[i].alphaOff = 0xffffffffu;
	};
	auto chain_texels = [&](size_t i) {
		const HostTex& h = s->htex[i];
		return (uint32_t)(h.off[h.levels - 1] + h.w[h.levels - 1] * h.h[h.levels - 1] - h.off[0]);
	};
	for (uint32_t i = 0; i < d->textureCount; i++) {
		add_chain(i, d->textures[i].width, d->textures[i].height, d->textures[i].space == 1);
		if (texels >= 0xffffffffull) return fail(set_error(VKR_E_LIMIT, "texel pool exceeds 2^32 texels"));
	}
	s->texelCount = texels;
	// Repacked chains: a material whose normal map, base colour, metalness and roughness textures have one
	// size gets (normal.rgb, metalness.r) and (baseColor.rgb, roughness.r) stored per texel, every level copied
	// byte for byte from the source chains, so that shading fetches two footprints instead of four.  The fourth
	// byte is read as a linear value, so metalness / roughness textures in an sRGB format keep the four fetches.
	struct Pack { uint32_t rgb, a, id; };
	std::vector<Pack> packs;
	std::vector<std::pair<uint32_t, uint32_t>> matPack(d->materialCount, { 0xffffffffu, 0xffffffffu });
	for (uint32_t i = 0; i < d->materialCount; i++) {
		const vkr_material_desc& m = d->materials[i];
		const uint32_t ids[4] = { m.normalMapTextureId, m.metalnessTextureId, m.baseColorTextureId, m.roughnessTextureId };
		bool ok = true;
		for (uint32_t id : ids) ok = ok && id < d->textureCount;
		for (int k = 1; ok && k < 4; k++)
			ok = d->textures[ids[k]].width == d->textures[ids[0]].width && d->textures[ids[k]].height == d->textures[ids[0]].height;
		ok = ok && d->textures[ids[1]].space != 1 && d->textures[ids[3]].space != 1;
		if (!ok) continue;
		auto find_or_add = [&](uint32_t rgb, uint32_t a) {
			for (const Pack& p : packs) if (p.rgb == rgb && p.a == a) return p.id;
			const uint32_t id = (uint32_t)s->htex.size();
			s->htex.emplace_back();
			tinfo.emplace_back();
			add_chain(id, d->textures[rgb].width, d->textures[rgb].height, d->textures[rgb].space == 1);
			packs.push_back({ rgb, a, id });
			return id;
		};
		matPack[i].first = find_or_add(ids[0], ids[1]);
		matPack[i].second = find_or_add(ids[2], ids[3]);
		if (texels >= 0xffffffffull) return fail(set_error(VKR_E_LIMIT, "texel pool exceeds 2^32 texels"));
	}
	if ((rc = s->texels.ensure(texels ? texels * 4 : 4))) return fail(rc);
	for (uint32_t i = 0; i < d->textureCount; i++) {
		const vkr_texture_desc& t = d->textures[i];
		hipError_t e = hipMemcpyAsync(s->texels.as<uint32_t>() + s->htex[i].off[0], t.rgba8, (size_t)t.width * t.height * 4,
		                              hipMemcpyHostToDevice, c->stream);
		if (e != hipSuccess) return fail(set_error(VKR_E_DEVICE, "texture upload: %s", hipGetErrorString(e)));
	}
	// Levels 1.. of every chain and the repacked chains, batched: level l of all textures is one launch (it reads
	// level l-1, finished by the launch before it on the stream), all repacked chains are one more.
	struct Ingest { DevBuf mip, span; ~Ingest() { mip.release(); span.release(); } } ingest;
	{
		int maxLevels = 0;
		for (uint32_t i = 0; i < d->textureCount; i++) maxLevels = s->htex[i].levels > maxLevels ? s->htex[i].levels : maxLevels;
		std::vector<MipJob> jobs;
		struct Batch { uint32_t first, count, maxW, maxH; };
		std::vector<Batch> batches;
		for (int l = 1; l < maxLevels; l++) {
			Batch b{ (uint32_t)jobs.size(), 0, 0, 0 };
			for (uint32_t i = 0; i < d->textureCount; i++) {
				const HostTex& h = s->htex[i];
				if (l >= h.levels) continue;
				jobs.push_back({ h.off[l - 1], h.off[l], (uint16_t)h.w[l - 1], (uint16_t)h.h[l - 1], (uint16_t)h.w[l], (uint16_t)h.h[l], h.srgb });
				b.maxW = h.w[l] > b.maxW ? h.w[l] : b.maxW;
				b.maxH = h.h[l] > b.maxH ? h.h[l] : b.maxH;
				b.count++;
			}
			if (b.count) batches.push_back(b);
		}
		if (!jobs.empty()) {
			if ((rc = upload(ingest.mip, jobs.data(), jobs.size() * sizeof(MipJob)))) return fail(rc);
			for (const Batch& b : batches)
				vkr_launch_mip_levels(c->stream, s->texels.as<uint32_t>(), ingest.mip.as<MipJob>() + b.first, b.count, b.maxW, b.maxH, c->dDecode, c->dThresh);
		}
	}
	std::vector<SpanJob> spans;
	uint32_t packMax = 0;
	for (const Pack& p : packs) {
		spans.push_back({ s->htex[p.rgb].off[0], s->htex[p.a].off[0], s->htex[p.id].off[0], chain_texels(p.id) });
		packMax = spans.back().n > packMax ? spans.back().n : packMax;
	}
	const size_t nPackSpans = spans.size();

	// ---- materials: binding table + whether the alpha test can ever discard
	std::vector<int> lowAlpha(d->textureCount, -1);
	std::vector<MatInfo> minfo(d->materialCount ? d->materialCount : 1);
	for (uint32_t i = 0; i < d->materialCount; i++) {
		const vkr_material_desc& m = d->materials[i];
		MatInfo& mi = minfo[i];
		memset(&mi, 0, sizeof mi);
		mi.tex[0] = m.baseColorTextureId;
		mi.tex[1] = m.metalnessTextureId; // binding 1
		mi.tex[2] = m.roughnessTextureId; // binding 2
		mi.tex[3] = (m.alphaMaskTextureId == VKR_NO_TEXTURE) ? m.baseColorTextureId : m.alphaMaskTextureId; // main/main.cpp:812-817
		mi.tex[4] = m.normalMapTextureId;
		const uint32_t at = mi.tex[3];
		if (lowAlpha[at] < 0) {
			// filtered alpha is a convex combination of texel alphas (mips included), so it can
			// fall below 0.5 only if some level-0 texel is below 128/255
			const vkr_texture_desc& t = d->textures[at];
			int low = 0;
			const size_t n = (size_t)t.width * t.height;
			for (size_t k = 0; k < n; k++) if (t.rgba8[4 * k + 3] < 128) { low = 1; break; }
			lowAlpha[at] = low;
		}
		mi.alphaTest = (uint32_t)lowAlpha[at];
		mi.packNM = matPack[i].first; mi.packAR = matPack[i].second;
		if (mi.alphaTest) s->anyAlphaTest = true;
	}
	if ((rc = upload(s->mat, minfo.data(), minfo.size() * sizeof(MatInfo)))) return fail(rc);
	// ---- alpha byte pool: the chains of the textures an alpha test can sample
	{
		uint64_t bytes = 0;
		for (uint32_t i = 0; i < d->materialCount; i++) {
			if (!minfo[i].alphaTest) continue;
			TexInfo& ti = tinfo[minfo[i].tex[3]];
			if (ti.alphaOff != 0xffffffffu) continue;
			ti.alphaOff = (uint32_t)bytes;
			bytes += chain_texels(minfo[i].tex[3]);
		}
		if ((rc = s->alphaTexels.ensure(bytes ? bytes : 4))) return fail(rc);
		uint32_t alphaMax = 0;
		for (uint32_t t = 0; t < d->textureCount; t++) {
			if (tinfo[t].alphaOff == 0xffffffffu) continue;
			spans.push_back({ 0u, s->htex[t].off[0], tinfo[t].alphaOff, chain_texels(t) });
			alphaMax = spans.back().n > alphaMax ? spans.back().n : alphaMax;
		}
		if (!spans.empty()) {
			if ((rc = upload(ingest.span, spans.data(), spans.size() * sizeof(SpanJob)))) return fail(rc);
			vkr_launch_pack_chains(c->stream, s->texels.as<uint32_t>(), ingest.span.as<SpanJob>(), (uint32_t)nPackSpans, packMax);
			vkr_launch_extract_alpha(c->stream, s->texels.as<uint32_t>(), s->alphaTexels.as<uint8_t>(), ingest.span.as<SpanJob>() + nPackSpans,
			                         (uint32_t)(spans.size() - nPackSpans), alphaMax);
		}
	}
	if ((rc = upload(s->tex, tinfo.data(), tinfo.size() * sizeof(TexInfo)))) return fail(rc);
	if ((rc = s->alphaAux.ensure(nTri ? nTri * sizeof(AlphaAux) : sizeof(AlphaAux)))) return fail(rc);
	SceneDev& dv = s->dev;
	dv.pos = s->pos.as<float>(); dv.uv = s->uv.as<float>(); dv.tcomp = s->tcomp.as<uint32_t>(); dv.tbn = s->tbn.as<float4>();
	dv.tri = s->tri.as<uint32_t>(); dv.triMat = s->triMat.as<uint32_t>(); dv.orderB = s->orderB.as<uint32_t>();
	dv.tex = s->tex.as<TexInfo>(); dv.mat = s->mat.as<MatInfo>(); dv.texels = s->texels.as<uint32_t>(); dv.alphaTexels = s->alphaTexels.as<uint8_t>();
	dv.srgbDecode = c->dDecode; dv.srgbThresh = c->dThresh;
	dv.nTri = (uint32_t)nTri; dv.nTriOpaque = (uint32_t)nTriOpaque; dv.nVert = (uint32_t)nVert;
	dv.nTex = (uint32_t)s->htex.size(); dv.nMat = d->materialCount;
	dv.alphaAux = s->alphaAux.as<AlphaAux>();
	vkr_launch_alpha_aux(c->stream, dv, s->alphaAux.as<AlphaAux>());
	if ((rc = s->triShade.ensure(nTri ? nTri * sizeof(TriShade) : sizeof(TriShade)))) return fail(rc);
	dv.triShade = s->triShade.as<TriShade>();
	vkr_launch_tri_shade(c->stream, dv, s->triShade.as<TriShade>());
	{
		hipError_t e = hipStreamSynchronize(c->stream);
		if (e != hipSuccess) return fail(set_error(VKR_E_DEVICE, "mip generation: %s", hipGetErrorString(e)));
	}
	owner.s = nullptr;
	*out = s;
	return VKR_OK;
}

extern "C" int vkr_scene_stats(const vkr_scene* s, uint64_t* triangles, uint64_t* vertices, uint64_t* texture_bytes)
{
	if (!s) return set_error(VKR_E_INVALID, "NULL scene");
	if (triangles) *triangles = s->nTri;
	if (vertices) *vertices = s->nVert;
	if (texture_bytes) *texture_bytes = s->texelCount * 4;
	return VKR_OK;
}

extern "C" int vkr_scene_read_texture_level(const vkr_scene* s, uint32_t tex, uint32_t level, uint8_t* out, uint32_t* w, uint32_t* h)
{
	if (!s || tex >= s->htex.size() || (int)level >= s->htex[tex].levels) return set_error(VKR_E_INVALID, "no such texture level");
	const HostTex& t = s->htex[tex];
	if (w) *w = t.w[level];
	if (h) *h = t.h[level];
	if (out) {
		HIP_CHECK(hipSetDevice(s->ctx->device));
		HIP_CHECK(hipMemcpy(out, s->texels.as<uint32_t>() + t.off[level], (size_t)t.w[level] * t.h[level] * 4, hipMemcpyDeviceToHost));
	}
	return VKR_OK;
}

// ---------------------------------------------------------------------------
// frame
// ---------------------------------------------------------------------------
static ViewDev make_view(uint32_t W, uint32_t H, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
	ViewDev v;
	v.W = W; v.H = H; v.x0 = x0; v.y0 = y0; v.x1 = x1; v.y1 = y1;
	v.rows = y1 - y0; v.stripeH = 0; v.stripeN = 1; v.stripeI = 0;
	v.tilesX = (x1 - x0 + VKR_TILE - 1) / VKR_TILE;
	v.tilesY = (y1 - y0 + VKR_TILE - 1) / VKR_TILE;
	return v;
}

// sort-first by interleaved stripes: stripes index, index + count, ... of the whole frame, stored back to back
static ViewDev make_view_stripes(uint32_t W, uint32_t H, uint32_t stripeH, uint32_t count, uint32_t index)
{
	ViewDev v = make_view(W, H, 0, 0, W, H);
	uint32_t rows = 0;
	(void)vkr_sortfirst_rows(H, count, index, stripeH, nullptr, &rows);
	v.rows = rows; v.stripeH = stripeH; v.stripeN = count; v.stripeI = index;
	v.tilesY = (rows + VKR_TILE - 1) / VKR_TILE;
	return v;
}

static int ensure_status(vkr_context* c, uint32_t need)
{
	if (c->statusUsed + need <= c->statusCap) return VKR_OK;
	if (c->isStatic) return set_error(VKR_E_LIMIT, "vkr_render_static: %d frames are pending, call vkr_wait first", VKR_STATIC_MAX_PENDING);
	// only grows while nothing is pending
	if (!c->pending.empty()) return set_error(VKR_E_LIMIT, "too many frames in flight without vkr_wait (%zu)", c->pending.size());
	uint32_t cap = c->statusCap ? c->statusCap * 2 : 4096;
	while (cap < need) cap *= 2;
	if (c->hStatus) (void)hipHostFree(c->hStatus);
	c->hStatus = nullptr; c->statusCap = 0; c->statusUsed = 0;
	HIP_CHECK(hipHostMalloc((void**)&c->hStatus, sizeof(PassStatus) * cap, hipHostMallocDefault));
	c->statusCap = cap;
	return VKR_OK;
}

struct GeomPass {
	Mat4 M; ViewDev view; uint32_t first, count; int cullOpaque, cullAlpha, useOrderB, alphaTest, shadow; bool keys;
};

// transform -> setup -> bin for one pass; leaves recs / tileOffset / binList ready
static int run_geometry(vkr_context* c, const vkr_scene* s, const GeomPass& g, uint32_t statusSlot)
{
	const uint32_t nTiles = g.view.tilesX * g.view.tilesY;
	if (nTiles == 0 || nTiles > (1u << 24)) return set_error(VKR_E_LIMIT, "tile count %u out of range", nTiles);
	// one slot per triangle + room for the fan triangles of clipped ones
	const uint32_t recCap = (uint32_t)std::min<double>(4.0e9, (double)g.count + ((double)g.count * 0.25 + 4096.0) * c->capScale);
	const uint32_t binCap = (uint32_t)std::min<double>(4.0e9, std::max<double>(((double)recCap * 3.0 + nTiles) * c->capScale, (double)c->binFloor));
	int rc;
	if ((rc = c->planes.ensure((size_t)(s->dev.nTri ? s->dev.nTri : 1) * sizeof(TriPlanes)))) return rc;
	if ((rc = c->recs.ensure((size_t)recCap * sizeof(SubTri)))) return rc;
	if ((rc = c->recTiles.ensure((size_t)recCap * sizeof(uint2)))) return rc;
	if ((rc = c->recLocal.ensure((size_t)recCap * 8 * 4))) return rc;
	if ((rc = c->bigQueue.ensure((size_t)recCap * 4))) return rc;
	if ((rc = c->clipQueue.ensure((size_t)(g.count ? g.count : 1) * 4))) return rc;
	if ((rc = c->tileOffset.ensure((size_t)(nTiles + 1) * 4))) return rc;
	if ((rc = c->cursor.ensure((size_t)nTiles * 4))) return rc;
	if ((rc = c->tileFlags.ensure((size_t)nTiles * 8))) return rc; // per-tile flags, then the list of deep tiles
	if ((rc = c->binList.ensure((size_t)binCap * 4))) return rc;
	if (g.keys && (rc = c->binKeys.ensure((size_t)binCap * 4))) return rc;
	if ((rc = c->tileOrder.ensure((size_t)nTiles * 4))) return rc;
	// every buffer of the pass is in place; only now is the counter-set state touched, so an allocation failure
	// above leaves the ping-pong pair as the last complete pass left it
	if (nTiles > c->zeroCapTiles || !c->zeroSets.p) {
		// both sets start out zero; from here on every pass's k_setup zeroes the set of the pass after it
		const uint32_t cap = nTiles + nTiles / 4 + 64;
		c->zeroCapTiles = 0;
		if ((rc = c->zeroSets.ensure(2 * (16 + 3 * (size_t)cap) * 4))) return rc;
		HIP_CHECK(hipMemsetAsync(c->zeroSets.p, 0, 2 * (16 + 3 * (size_t)cap) * 4, c->stream));
		c->zeroCapTiles = cap; c->zset = 0;
	} else {
		c->zset ^= 1;
	}
	c->recCap = recCap; c->binCap = binCap;
	uint32_t* counters = c->counters();
	{
		StageTimer t(c, ST_SETUP);
		SetupArgs a;
		a.sc = s->dev; a.view = g.view; a.M = g.M;
		a.zeroBase = c->zs(c->zset ^ 1); a.zeroWords = 16 + 3 * c->zeroCapTiles;
		a.first = g.first; a.count = g.count; a.cullOpaque = g.cullOpaque; a.cullAlpha = g.cullAlpha;
		a.useOrderB = g.useOrderB; a.alphaTest = g.alphaTest; a.shadow = g.shadow;
		a.guard = vkr_guard_band(g.view.W, g.view.H);
		a.recs = c->recs.as<SubTri>(); a.recTiles = c->recTiles.as<uint2>(); a.recCap = recCap;
		a.recCount = counters; a.tileCount = c->tileCount(); a.overflow = counters + 1;
		a.tileCountBig = c->tileCountBig(); a.tileCountA = c->tileCountA(); a.recLocal = c->recLocal.as<uint32_t>();
		a.clipQueue = c->clipQueue.as<uint32_t>(); a.clipCount = counters + 3; a.bigQueue = c->bigQueue.as<uint32_t>(); a.bigCount = counters + 4; a.planes = c->planes.as<TriPlanes>();
		vkr_launch_setup(c->stream, a);
	}
	{
		StageTimer t(c, ST_BIN_SCAN);
		// for the passes whose bins are very uneven (the light's depth pass, the overshading pass) the scan also lists
		// the tiles by descending bin length, and their per-tile kernels start the longest tiles first
		c->tileOrdered = g.shadow || g.keys;
		vkr_launch_scan_tiles(c->stream, c->tileCount(), c->tileCountA(), c->tileCountBig(), c->tileOffset.as<uint32_t>(), c->cursor.as<uint32_t>(), nTiles,
		                      binCap, counters + 1, &c->hStatus[statusSlot].recCount, c->tileOrdered ? c->tileOrder.as<uint32_t>() : nullptr);
	}
	{
		StageTimer t(c, ST_BIN_FILL);
		BinFillArgs b;
		b.recs = c->recs.as<SubTri>(); b.recTiles = c->recTiles.as<uint2>(); b.recCount = counters; b.baseCount = g.count; b.recCap = recCap;
		b.tileOffset = c->tileOffset.as<uint32_t>(); b.tileCount = c->tileCount(); b.tileCountA = c->tileCountA(); b.recLocal = c->recLocal.as<uint32_t>(); b.bigQueue = c->bigQueue.as<uint32_t>(); b.bigCount = counters + 4; b.cursor = c->cursor.as<uint32_t>(); b.binList = c->binList.as<uint32_t>();
		b.keys = g.keys ? c->binKeys.as<uint32_t>() : nullptr; b.binCap = binCap; b.tilesX = g.view.tilesX;
		vkr_launch_bin_fill(c->stream, b, recCap);
	}
	// status for the overflow check at vkr_wait
	// (the pass's status for the overflow check at vkr_wait is written by k_scan_tiles into pinned host memory)
	HIP_CHECK(hipGetLastError());
	return VKR_OK;
}

static void fill_raster_args(vkr_context* c, const vkr_scene* s, const ViewDev& view, RasterArgs& r)
{
	r.sc = s->dev; r.view = view; r.clip = c->clip.as<float4>(); r.planes = c->planes.as<TriPlanes>(); r.recs = c->recs.as<SubTri>();
	r.tileOffset = c->tileOffset.as<uint32_t>(); r.tileCount = c->tileCount(); r.tileCountA = c->tileCountA(); r.binList = c->binList.as<uint32_t>(); r.binCap = c->binCap;
	r.vis = nullptr; r.depthOut = nullptr; r.countOut = nullptr;
	r.alphaTiles = c->tileFlags.as<uint32_t>(); r.alphaPass = 0; r.tileOrder = c->tileOrdered ? c->tileOrder.as<uint32_t>() : nullptr;
	r.deepTiles = c->tileFlags.as<uint32_t>() + (size_t)view.tilesX * view.tilesY; r.deepTileCount = c->counters() + 5; r.alphaTileCount = c->counters() + 2;
}

static int copy_out(vkr_context* c, void* dst, const void* src, size_t bytes)
{
	if (!dst || !bytes) return VKR_OK;
	HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->stream));
	return VKR_OK;
}

static void cam_mat(const float m[16], Mat4& M) { memcpy(M.m, m, sizeof(float) * 16); }

// shadow pass (main/main.cpp:3950-3984) + regular pass (main/main.cpp:3987-4077)
// over `view`; colour lands in `colourDst` (view-rectangle sized)
// Depth-only pass from the light (shadowOffscreen.vert:9-11; main/main.cpp:3086-3137, 3950-3984): opaque meshes,
// no culling, slope-scaled bias; leaves the map in c->shadow.
static int run_shadow(vkr_context* c, const vkr_scene* s, const vkr_frame_params& fp, uint32_t statusSlot)
{
	const uint32_t S = fp.shadow_size ? fp.shadow_size : 2048;
	int rc;
	if ((rc = c->shadow.ensure((size_t)S * S * 4))) return rc;
	GeomPass g;
	cam_mat(fp.uniforms.depthMVPUniform.depthMVP, g.M);
	g.view = make_view(S, S, 0, 0, S, S);
	g.first = 0; g.count = s->dev.nTriOpaque; // opaque meshes only
	g.cullOpaque = 0; g.cullAlpha = 0; g.useOrderB = 0; g.alphaTest = 0; g.shadow = 1; g.keys = false;
	if ((rc = run_geometry(c, s, g, statusSlot))) return rc;
	StageTimer t(c, ST_RASTER_SHADOW);
	RasterArgs r;
	fill_raster_args(c, s, g.view, r);
	r.depthOut = c->shadow.as<float>();
	vkr_launch_raster(c->stream, RASTER_SHADOW, r);
	return VKR_OK;
}

// The light's pass does not depend on the camera's passes of the same frame: it runs on the helper context's stream,
// forked after everything queued on this context so far (the previous frame's shading read the helper's map); the
// caller joins (join_shadow) in front of whatever reads the map.
static int run_shadow_beside(vkr_context* c, const vkr_scene* s, const vkr_frame_params& fp, uint32_t statusSlot)
{
	int rc;
	// allocation-free frames have no helper context to create: the light's pass runs on the frame's own stream, with
	// the frame's own scratch, in front of the camera's passes
	if (c->isStatic) return run_shadow(c, s, fp, statusSlot);
	vkr_context* h = c->shadowHelper;
	if (!h) {
		if ((rc = context_create_impl(c->device, &h, 1, c))) return rc;
		c->shadowHelper = h;
		HIP_CHECK(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
		HIP_CHECK(hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming));
	}
	h->profiling = c->profiling; h->capScale = c->capScale; h->binFloor = c->binFloor;
	HIP_CHECK(hipEventRecord(c->evFork, c->stream));
	HIP_CHECK(hipStreamWaitEvent(h->stream, c->evFork, 0));
	h->hStatus = c->hStatus; // the pass status lands in this context's slots, where vkr_wait looks for it
	rc = run_shadow(h, s, fp, statusSlot);
	h->hStatus = nullptr;
	if (rc) return rc;
	HIP_CHECK(hipEventRecord(c->evJoin, h->stream));
	return VKR_OK;
}

// the context that holds the shadow map rendered by run_shadow_beside
static vkr_context* shadow_owner(vkr_context* c) { return c->isStatic ? c : c->shadowHelper; }

static int join_shadow(vkr_context* c)
{
	if (c->isStatic) return VKR_OK; // same stream: already ordered
	HIP_CHECK(hipStreamWaitEvent(c->stream, c->evJoin, 0));
	return VKR_OK;
}

static int run_forward(vkr_context* c, const vkr_scene* s, const vkr_frame_params& fp, const ViewDev& view, uint32_t* colourDst,
                       uint32_t statusSlot, const vkr_frame_out* out, bool writeSide)
{
	const vkr_uniforms& u = fp.uniforms;
	const uint32_t S = fp.shadow_size ? fp.shadow_size : 2048;
	int rc;
	if ((rc = run_shadow_beside(c, s, fp, statusSlot))) return rc;
	vkr_context* h = shadow_owner(c);
	const size_t n = (size_t)(view.x1 - view.x0) * view.rows;
	if ((rc = c->vis.ensure(n * 8))) return rc;
	if ((rc = c->depth.ensure(n * 4))) return rc;
	if ((rc = c->triId.ensure(n * 4))) return rc;
	{
		GeomPass g;
		cam_mat(u.sceneUniforms.projCam, g.M);
		g.view = view; g.first = 0; g.count = s->dev.nTri;
		g.cullOpaque = 1; g.cullAlpha = 0; g.useOrderB = 0; g.alphaTest = 1; g.shadow = 0; g.keys = false;
		if ((rc = run_geometry(c, s, g, statusSlot + 1))) return rc;
		{
			StageTimer t(c, ST_RASTER_VIS);
			RasterArgs r;
			fill_raster_args(c, s, view, r);
			r.vis = c->vis.as<unsigned long long>();
			r.alphaPass = s->anyAlphaTest ? 1 : 0;
			vkr_launch_raster(c->stream, RASTER_VIS, r);
		}
		if ((rc = join_shadow(c))) return rc;
		StageTimer t(c, ST_RESOLVE_FORWARD);
		ResolveArgs a;
		memset(&a, 0, sizeof a);
		a.sc = s->dev; a.view = view; a.clip = c->clip.as<float4>(); a.planes = c->planes.as<TriPlanes>(); a.vis = c->vis.as<unsigned long long>();
		a.depth = c->depth.as<float>(); a.triId = c->triId.as<uint32_t>(); a.colour = colourDst;
		a.shadow = h->shadow.as<float>(); a.shadowSize = S;
		static const float biasMat[16] = { 0.5f, 0, 0, 0, 0, 0.5f, 0, 0, 0, 0, 1.0f, 0, 0.5f, 0.5f, 0, 1.0f }; // default.vert:46-50
		mat4_mul(biasMat, u.depthMVPUniform.depthMVP, a.lightMat.m);
		memcpy(a.lightPos, u.lightUniforms.lightPos, 16); memcpy(a.lightColour, u.lightUniforms.lightColour, 16);
		memcpy(a.camPos, u.sceneUniforms.camPos, 16);
		vkr_launch_resolve_forward(c->stream, a);
	}
	if (writeSide) {
		StageTimer t(c, ST_COPY_OUT);
		if ((rc = copy_out(c, out->tri_id, c->triId.p, n * 4))) return rc;
		if ((rc = copy_out(c, out->depth_bits, c->depth.p, n * 4))) return rc;
		if ((rc = copy_out(c, out->shadow_map, h->shadow.p, (size_t)S * S * 4))) return rc;
	}
	return VKR_OK;
}

static int submit_frame(vkr_context* c, const vkr_scene* s, const vkr_frame_params& fp, const vkr_frame_out& out, uint32_t statusSlot,
                        uint32_t* nPasses)
{
	const vkr_uniforms& u = fp.uniforms;
	uint32_t x0 = fp.rect_x0, y0 = fp.rect_y0, x1 = fp.rect_x1, y1 = fp.rect_y1;
	if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = fp.width; y1 = fp.height; }
	if (fp.width == 0 || fp.height == 0 || fp.width > 16384 || fp.height > 16384)
		return set_error(VKR_E_INVALID, "extent %ux%u out of range", fp.width, fp.height);
	if (x0 >= x1 || y0 >= y1 || x1 > fp.width || y1 > fp.height)
		return set_error(VKR_E_INVALID, "rectangle [%u,%u)x[%u,%u) outside the %ux%u frame", x0, x1, y0, y1, fp.width, fp.height);
	const uint32_t S = fp.shadow_size ? fp.shadow_size : 2048;
	if (S > 16384) return set_error(VKR_E_INVALID, "shadow_size %u out of range", S);
	ViewDev view = make_view(fp.width, fp.height, x0, y0, x1, y1);
	if (fp.stripes) {
		const uint32_t sh = fp.stripes & 0xffffu, sn = (fp.stripes >> 16) & 0xffu, si = fp.stripes >> 24;
		if (sh == 0 || sh % VKR_TILE || sn == 0 || si >= sn)
			return set_error(VKR_E_INVALID, "stripes: height %u (a multiple of %d), index %u of %u", sh, VKR_TILE, si, sn);
		if (fp.rect_x0 || fp.rect_y0 || fp.rect_x1 || fp.rect_y1 || fp.mosaicEffect)
			return set_error(VKR_E_INVALID, "stripes cannot be combined with a rectangle or with the mosaic pass");
		if (sn > 1) view = make_view_stripes(fp.width, fp.height, sh, sn, si);
	}
	const size_t n = (size_t)(x1 - x0) * view.rows;
	if (view.rows == 0) { *nPasses = 0; return VKR_OK; } // a rank whose stripes lie beyond the frame's last row
	int rc;
	// The presented image is written straight into the caller's buffer when that is memory of this GPU
	// (a 33 MB device-to-device copy per 4K frame otherwise); host buffers get a copy from the context's own.
	uint32_t* colourDst = nullptr;
	if (out.rgba8 && ((uintptr_t)out.rgba8 & 3u) == 0) {
		hipPointerAttribute_t attr;
		if (hipPointerGetAttributes(&attr, out.rgba8) == hipSuccess && attr.type == hipMemoryTypeDevice && attr.device == c->device)
			colourDst = (uint32_t*)out.rgba8;
		else
			(void)hipGetLastError(); // an ordinary host pointer is not an error
	}
	const bool colourDirect = colourDst != nullptr;
	if (!colourDirect) {
		if ((rc = c->colour.ensure(n * 4))) return rc;
		colourDst = c->colour.as<uint32_t>();
	}
	void* const rgbaCopy = colourDirect ? nullptr : out.rgba8; // copy_out skips a NULL destination
	const int dbg = fp.debugVisualisation;
	*nPasses = 0;

	if (fp.deferredShading && !fp.mosaicEffect) {
		// ---- deferred: G-buffer subpass + lighting subpass (main/main.cpp:3836-3946)
		if ((rc = c->vis.ensure(n * 8)) || (rc = c->depth.ensure(n * 4)) || (rc = c->triId.ensure(n * 4)) ||
		    (rc = c->gN.ensure(n * 8)) || (rc = c->gA.ensure(n * 8))) return rc;
		GeomPass g;
		cam_mat(u.sceneUniforms.projCam, g.M);
		g.view = view; g.first = 0; g.count = s->dev.nTri;
		g.cullOpaque = 1; g.cullAlpha = 0; g.useOrderB = 0; g.alphaTest = 1; g.shadow = 0; g.keys = false;
		*nPasses = 1;
		if (fp.flags & VKR_FRAME_SHADOW_PASS) {
			// not a reference combination (its deferred path has no shadow pass): the map is rendered, handed out
			// if asked for, and not sampled
			if ((rc = run_shadow_beside(c, s, fp, statusSlot + 1))) return rc;
			*nPasses = 2;
		}
		if ((rc = run_geometry(c, s, g, statusSlot))) return rc;
		{
			StageTimer t(c, ST_RASTER_VIS);
			RasterArgs r;
			fill_raster_args(c, s, view, r);
			r.vis = c->vis.as<unsigned long long>();
			r.alphaPass = s->anyAlphaTest ? 1 : 0;
			vkr_launch_raster(c->stream, RASTER_VIS, r);
		}
		ResolveArgs a;
		memset(&a, 0, sizeof a);
		a.sc = s->dev; a.view = view; a.clip = c->clip.as<float4>(); a.planes = c->planes.as<TriPlanes>(); a.vis = c->vis.as<unsigned long long>();
		a.gN = c->gN.as<uint2>(); a.gA = c->gA.as<uint2>(); a.depth = c->depth.as<float>(); a.triId = c->triId.as<uint32_t>();
		LightArgs l;
		memset(&l, 0, sizeof l);
		l.view = view; l.gN = c->gN.as<uint2>(); l.gA = c->gA.as<uint2>(); l.depth = c->depth.as<float>();
		l.colour = colourDst; l.srgbThresh = c->dThresh;
		float invCam[16], invProj[16];
		mat4_inverse(u.sceneUniforms.camera, invCam);
		mat4_inverse(u.sceneUniforms.projection, invProj);
		mat4_mul(invCam, invProj, l.invPC.m); // deferredShading.frag:36
		memcpy(l.camPos, u.sceneUniforms.camPos, 16);
		for (int i = 0; i < VKR_NUM_LIGHTS; i++) {
			memcpy(l.lightPos[i], u.multipleLightsUniform.light[i].lightPos, 16);
			memcpy(l.lightCol[i], u.multipleLightsUniform.light[i].lightColour, 16);
		}
		// clear (0.1,0.1,0.1,0.1), main/main.cpp:3838-3841
		const uint32_t c01 = host_srgb_encode(c->hThresh, 0.1f);
		l.clearColour = c01 | (c01 << 8) | (c01 << 16) | ((uint32_t)floorf(0.1f * 255.0f + 0.5f) << 24);
		if (fp.flags & VKR_FRAME_FUSED_LIGHTING) {
			// one kernel; the attachments and the integer buffers are written only when the caller wants them
			StageTimer t(c, ST_RESOLVE_LIGHT);
			if (!out.gbuf_normal) a.gN = nullptr;
			if (!out.gbuf_albedo) a.gA = nullptr;
			if (!out.depth_bits) a.depth = nullptr;
			if (!out.tri_id) a.triId = nullptr;
			a.colour = colourDst;
			vkr_launch_resolve_light(c->stream, a, l);
		} else {
			{
				StageTimer t(c, ST_RESOLVE_GBUF);
				if (!out.tri_id) a.triId = nullptr; // the lighting pass reads depth and the attachments, not the ids
				vkr_launch_resolve_gbuf(c->stream, a);
			}
			StageTimer t(c, ST_DEFERRED_LIGHT);
			vkr_launch_deferred_light(c->stream, l);
		}
		if ((fp.flags & VKR_FRAME_SHADOW_PASS) && (rc = join_shadow(c))) return rc;
		StageTimer t(c, ST_COPY_OUT);
		if ((rc = copy_out(c, rgbaCopy, c->colour.p, n * 4)) || (rc = copy_out(c, out.tri_id, c->triId.p, n * 4)) ||
		    (rc = copy_out(c, out.depth_bits, c->depth.p, n * 4)) || (rc = copy_out(c, out.gbuf_normal, c->gN.p, n * 8)) ||
		    (rc = copy_out(c, out.gbuf_albedo, c->gA.p, n * 8))) return rc;
		if ((fp.flags & VKR_FRAME_SHADOW_PASS) && (rc = copy_out(c, out.shadow_map, shadow_owner(c)->shadow.p, (size_t)S * S * 4))) return rc;
	} else if (!fp.mosaicEffect && dbg == 1) {
		// ---- forward + shadow map (main/main.cpp:3948-4079)
		if ((rc = run_forward(c, s, fp, view, colourDst, statusSlot, &out, true))) return rc;
		*nPasses = 2;
		StageTimer t(c, ST_COPY_OUT);
		if ((rc = copy_out(c, rgbaCopy, c->colour.p, n * 4))) return rc;
	} else if (!fp.mosaicEffect && (dbg == 5 || dbg == 6)) {
		// ---- overdraw / overshading counters (main/main.cpp:4081-4136): every mesh in file order, cull BACK
		if ((rc = c->stencil.ensure(n)) || (rc = c->depth.ensure(n * 4))) return rc;
		GeomPass g;
		cam_mat(u.sceneUniforms.projCam, g.M);
		g.view = view; g.first = 0; g.count = s->dev.nTri;
		g.cullOpaque = 1; g.cullAlpha = 1; g.useOrderB = 1; g.alphaTest = 0; g.shadow = 0; g.keys = (dbg == 6);
		if ((rc = run_geometry(c, s, g, statusSlot))) return rc;
		*nPasses = 1;
		if (dbg == 5) {
			StageTimer t(c, ST_RASTER_COUNT);
			RasterArgs r;
			fill_raster_args(c, s, view, r);
			r.countOut = c->stencil.as<uint8_t>();
			vkr_launch_raster(c->stream, RASTER_COUNT, r);
			// depth test disabled => no depth writes: the attachment keeps its clear value
			vkr_launch_fill_u32(c->stream, c->depth.as<uint32_t>(), 0x3F800000u, n);
		} else {
			// submission order matters: the kernel sorts every tile's bin by primitive before walking it
			StageTimer t(c, ST_OVERSHADE);
			OvershadeArgs o;
			o.view = view; o.recs = c->recs.as<SubTri>(); o.tileOffset = c->tileOffset.as<uint32_t>();
			o.binList = c->binList.as<uint32_t>(); o.binKeys = c->binKeys.as<uint32_t>();
			o.binCap = c->binCap; o.countOut = c->stencil.as<uint8_t>(); o.depthOut = c->depth.as<float>();
			o.tileOrder = c->tileOrder.as<uint32_t>();
			vkr_launch_overshade(c->stream, o);
		}
		{
			StageTimer t(c, ST_OVERVIS_COLOUR);
			vkr_launch_overvis_colour(c->stream, c->stencil.as<uint8_t>(), colourDst, c->dThresh, (uint32_t)n);
		}
		StageTimer t(c, ST_COPY_OUT);
		if ((rc = copy_out(c, rgbaCopy, c->colour.p, n * 4)) || (rc = copy_out(c, out.stencil, c->stencil.p, n)) ||
		    (rc = copy_out(c, out.depth_bits, c->depth.p, n * 4))) return rc;
	} else if (!fp.mosaicEffect) {
		// ---- debug views 2/3/4 (main/main.cpp:4138-4193): every mesh in file order, cull BACK, no discard
		if ((rc = c->vis.ensure(n * 8)) || (rc = c->depth.ensure(n * 4)) || (rc = c->triId.ensure(n * 4)) || (rc = c->mip.ensure(n))) return rc;
		GeomPass g;
		cam_mat(u.sceneUniforms.projCam, g.M);
		g.view = view; g.first = 0; g.count = s->dev.nTri;
		g.cullOpaque = 1; g.cullAlpha = 1; g.useOrderB = 1; g.alphaTest = 0; g.shadow = 0; g.keys = false;
		if ((rc = run_geometry(c, s, g, statusSlot))) return rc;
		*nPasses = 1;
		{
			StageTimer t(c, ST_RASTER_VIS);
			RasterArgs r;
			fill_raster_args(c, s, view, r);
			r.vis = c->vis.as<unsigned long long>();
			vkr_launch_raster(c->stream, RASTER_VIS, r);
		}
		{
			StageTimer t(c, ST_RESOLVE_DEBUG);
			ResolveArgs a;
			memset(&a, 0, sizeof a);
			a.sc = s->dev; a.view = view; a.clip = c->clip.as<float4>(); a.planes = c->planes.as<TriPlanes>(); a.vis = c->vis.as<unsigned long long>();
			a.depth = c->depth.as<float>(); a.triId = c->triId.as<uint32_t>(); a.mip = c->mip.as<uint8_t>();
			a.colour = colourDst; a.debugMode = u.debugUniforms.debug; a.useOrderB = 1;
			vkr_launch_resolve_debug(c->stream, a);
		}
		StageTimer t(c, ST_COPY_OUT);
		if ((rc = copy_out(c, rgbaCopy, c->colour.p, n * 4)) || (rc = copy_out(c, out.tri_id, c->triId.p, n * 4)) ||
		    (rc = copy_out(c, out.depth_bits, c->depth.p, n * 4)) || (rc = copy_out(c, out.mip_level, c->mip.p, n))) return rc;
	} else {
		// ---- mosaic (main/main.cpp:4195-4320): forward pass into the offscreen sRGB image, then the
		// 5x3 block post-process.  The reference records no shadow pass here and leaves the shadow
		// descriptors unbound (undefined shadow term); the shadow map is rendered as in the regular path.
		// A block of the rectangle can read up to 5 columns / 3 rows outside it.
		const uint32_t hx0 = x0 >= 6 ? x0 - 6 : 0, hy0 = y0 >= 4 ? y0 - 4 : 0;
		const uint32_t hx1 = std::min(fp.width, x1 + 1), hy1 = std::min(fp.height, y1 + 1);
		const ViewDev hview = make_view(fp.width, fp.height, hx0, hy0, hx1, hy1);
		const size_t hn = (size_t)(hx1 - hx0) * (hy1 - hy0);
		if ((rc = c->colour2.ensure(hn * 4))) return rc;
		const bool full = (hx0 == x0 && hy0 == y0 && hx1 == x1 && hy1 == y1);
		if ((rc = run_forward(c, s, fp, hview, c->colour2.as<uint32_t>(), statusSlot, &out, full))) return rc;
		*nPasses = 2;
		{
			StageTimer t(c, ST_MOSAIC);
			vkr_launch_mosaic(c->stream, c->colour2.as<uint32_t>(), hx0, hy0, hx1 - hx0, hy1 - hy0, colourDst, view,
			                  c->dDecode, c->dThresh);
		}
		StageTimer t(c, ST_COPY_OUT);
		if ((rc = copy_out(c, rgbaCopy, c->colour.p, n * 4))) return rc;
	}
	HIP_CHECK(hipGetLastError());
	return VKR_OK;
}

// A submit that failed half-way has kernels queued that still write this frame's status slots, may have forked work
// onto the helper stream that nothing joins, and leaves the ping-pong counter sets in an unknown phase: drain both
// streams and make the next pass start from freshly zeroed sets.
static void recover_failed_submit(vkr_context* c)
{
	char keep[sizeof g_err]; // the caller reports the submit's error, not the clean-up's
	memcpy(keep, g_err, sizeof keep);
	(void)hipStreamSynchronize(c->stream);
	if (c->shadowHelper) {
		(void)hipStreamSynchronize(c->shadowHelper->stream);
		c->shadowHelper->zeroCapTiles = 0;
	}
	c->zeroCapTiles = 0; // run_geometry re-zeroes both sets
	(void)hipGetLastError();
	memcpy(g_err, keep, sizeof keep);
}

static int render_impl(vkr_context* c, const vkr_scene* s, const vkr_frame_params* fp, const vkr_frame_out* out)
{
	if (!c || !s || !fp || !out) return set_error(VKR_E_INVALID, "vkr_render: NULL argument");
	if (s->ctx != c && s->ctx != c->parent) return set_error(VKR_E_INVALID, "vkr_render: scene belongs to another context");
	if (!c->lanes.empty()) {
		// frames go round the lanes in submission order; the scene is read-only while rendering
		const uint32_t k = c->nextLane;
		c->nextLane = (k + 1) % (uint32_t)(c->lanes.size() + 1);
		c->lastLane = k;
		if (k) return render_impl(c->lanes[k - 1], s, fp, out);
	}
	HIP_CHECK(hipSetDevice(c->device));
	int rc;
	if ((rc = ensure_status(c, 2))) return rc;
	PendingFrame pf;
	pf.scene = s; pf.params = *fp; pf.out = *out; pf.statusSlot = c->statusUsed; pf.nPasses = 0;
	memset(&c->hStatus[pf.statusSlot], 0, sizeof(PassStatus) * 2);
	if ((rc = submit_frame(c, s, pf.params, pf.out, pf.statusSlot, &pf.nPasses))) { recover_failed_submit(c); return rc; }
	c->statusUsed += 2;
	c->pending.push_back(pf);
	return VKR_OK;
}

// vkr_render_static: the frame goes to the context's allocation-free twin.  Every device buffer is cut anew from the
// static scratch array (the previous frame's slices are dead by then: same stream), the counter sets are zeroed by a
// memset on that address, nothing is created, grown or freed.
static int render_static_impl(vkr_context* c, const vkr_scene* s, const vkr_frame_params* fp, const vkr_frame_out* out)
{
	if (!c || !s || !fp || !out) return set_error(VKR_E_INVALID, "vkr_render_static: NULL argument");
	if (c->parent || !c->staticCtx) return set_error(VKR_E_INVALID, "vkr_render_static: not a context created by vkr_context_create");
	if (s->ctx != c) return set_error(VKR_E_INVALID, "vkr_render_static: scene belongs to another context");
	vkr_context* t = c->staticCtx;
	HIP_CHECK(hipSetDevice(c->device));
	if (t->pending.size() >= VKR_STATIC_MAX_PENDING)
		return set_error(VKR_E_LIMIT, "vkr_render_static: %d frames are pending, call vkr_wait first", VKR_STATIC_MAX_PENDING);
	int rc;
	if ((rc = static_scratch_acquire(t))) return rc;
	DevBuf* b[VKR_CTX_NBUFS];
	ctx_bufs(t, b);
	for (DevBuf* d : b) { d->p = nullptr; d->cap = 0; }
	t->arena.used = 0;
	t->capScale = c->capScale; t->binFloor = c->binFloor; // what vkr_render has learnt about this scene's bins
	t->zeroCapTiles = 0; // the counter sets are re-cut and zeroed for this frame (the array keeps the last frame's contents)
	if ((rc = ensure_status(t, 2))) { if (t->pending.empty()) static_scratch_release(t); return rc; }
	PendingFrame pf;
	pf.scene = s; pf.params = *fp; pf.out = *out; pf.statusSlot = t->statusUsed; pf.nPasses = 0;
	memset(&t->hStatus[pf.statusSlot], 0, sizeof(PassStatus) * 2);
	if ((rc = submit_frame(t, s, pf.params, pf.out, pf.statusSlot, &pf.nPasses))) {
		recover_failed_submit(t);
		if (t->pending.empty()) static_scratch_release(t);
		return rc;
	}
	t->statusUsed += 2;
	t->pending.push_back(pf); // capacity reserved at creation
	return VKR_OK;
}

// the fence for allocation-free frames: nothing can be grown, so a pass that ran out of room is an error
static int wait_static(vkr_context* t)
{
	HIP_CHECK(hipSetDevice(t->device));
	HIP_CHECK(hipStreamSynchronize(t->stream));
	int rc = VKR_OK;
	for (auto& pf : t->pending)
		for (uint32_t k = 0; k < pf.nPasses; k++)
			if (t->hStatus[pf.statusSlot + k].overflow)
				rc = set_error(VKR_E_LIMIT, "vkr_render_static: a pass needs %u bin entries, more than its share of the static scratch holds; "
				                            "the frame is incomplete, render it with vkr_render", t->hStatus[pf.statusSlot + k].binTotal);
	t->pending.clear();
	t->statusUsed = 0;
	static_scratch_release(t); // nothing of this context is in flight any more
	return rc;
}

static int wait_impl(vkr_context* c)
{
	if (!c) return set_error(VKR_E_INVALID, "vkr_wait: NULL context");
	if (c->isStatic) return wait_static(c);
	HIP_CHECK(hipSetDevice(c->device));
	HIP_CHECK(hipStreamSynchronize(c->stream));
	// stage timers are folded into the totals at every fence, so their events return to the pool however long the
	// host goes without reading the profile
	drain_timers(c);
	if (c->shadowHelper) { HIP_CHECK(hipStreamSynchronize(c->shadowHelper->stream)); drain_timers(c->shadowHelper); }
	// a pass that ran out of record or bin space produced an incomplete image:
	// grow the scratch buffers and render those frames again
	for (int attempt = 0; attempt < 6; attempt++) {
		std::vector<PendingFrame> redo;
		for (auto& pf : c->pending)
			for (uint32_t k = 0; k < pf.nPasses; k++)
				if (c->hStatus[pf.statusSlot + k].overflow) {
					// the tile scan totals every entry whether or not it fits: size the bins from it
					for (uint32_t j = 0; j < pf.nPasses; j++) {
						const uint64_t need = c->hStatus[pf.statusSlot + j].binTotal;
						c->binFloor = std::max<uint64_t>(c->binFloor, need + need / 8 + 1024);
					}
					redo.push_back(pf);
					break;
				}
		c->pending.clear();
		c->statusUsed = 0;
		if (redo.empty()) return VKR_OK;
		c->capScale *= 2.0;
		for (auto& pf : redo) {
			memset(&c->hStatus[c->statusUsed], 0, sizeof(PassStatus) * 2);
			pf.statusSlot = c->statusUsed;
			int rc = submit_frame(c, pf.scene, pf.params, pf.out, pf.statusSlot, &pf.nPasses);
			if (rc) { recover_failed_submit(c); c->pending.clear(); c->statusUsed = 0; return rc; }
			c->statusUsed += 2;
			c->pending.push_back(pf);
		}
		HIP_CHECK(hipStreamSynchronize(c->stream));
	}
	c->pending.clear();
	c->statusUsed = 0;
	return set_error(VKR_E_LIMIT, "tile bins still overflow after growing the scratch buffers");
}

// ---------------------------------------------------------------------------
// guarded entry points
// ---------------------------------------------------------------------------
extern "C" int vkr_scene_create(vkr_context* c, const vkr_scene_desc* d, vkr_scene** out)
{
	return guarded("vkr_scene_create", [&] { return scene_create_impl(c, d, out); });
}
extern "C" int vkr_render(vkr_context* c, const vkr_scene* s, const vkr_frame_params* fp, const vkr_frame_out* out)
{
	return guarded("vkr_render", [&] { return render_impl(c, s, fp, out); });
}
extern "C" int vkr_render_static(vkr_context* c, const vkr_scene* s, const vkr_frame_params* fp, const vkr_frame_out* out)
{
	// no guard: nothing on this path allocates on the host either (the pending list's capacity was reserved when the
	// context was created), so there is no exception to catch
	return render_static_impl(c, s, fp, out);
}
extern "C" int vkr_wait(vkr_context* c)
{
	return guarded("vkr_wait", [&] {
		int rc = wait_impl(c);
		if (c) {
			for (vkr_context* l : c->lanes) {
				const int r = wait_impl(l);
				if (r && !rc) rc = r;
			}
			if (c->staticCtx) {
				const int r = wait_impl(c->staticCtx);
				if (r && !rc) rc = r;
			}
		}
		return rc;
	});
}

extern "C" int vkr_wait_next_lane(vkr_context* c)
{
	return guarded("vkr_wait_next_lane", [&] {
		if (!c) return set_error(VKR_E_INVALID, "vkr_wait_next_lane: NULL context");
		return wait_impl(c->nextLane ? c->lanes[c->nextLane - 1] : c);
	});
}

static int set_lanes_impl(vkr_context* c, uint32_t n)
{
	if (!c || c->parent) return set_error(VKR_E_INVALID, "vkr_context_set_frames_in_flight: not a context created by vkr_context_create");
	if (n < 1 || n > VKR_MAX_FRAMES_IN_FLIGHT) return set_error(VKR_E_LIMIT, "vkr_context_set_frames_in_flight: %u lanes (1..%d)", n, VKR_MAX_FRAMES_IN_FLIGHT);
	bool busy = !c->pending.empty();
	for (vkr_context* l : c->lanes) busy = busy || !l->pending.empty();
	if (busy) return set_error(VKR_E_INVALID, "vkr_context_set_frames_in_flight: frames are pending, call vkr_wait first");
	while (c->lanes.size() + 1 > n) { vkr_context_destroy(c->lanes.back()); c->lanes.pop_back(); }
	c->lanes.reserve(n);
	while (c->lanes.size() + 1 < n) {
		vkr_context* l = nullptr;
		int rc = context_create_impl(c->device, &l, 1, c);
		if (rc) return rc;
		l->profiling = c->profiling;
		c->lanes.push_back(l);
	}
	c->nextLane = 0; c->lastLane = 0;
	return VKR_OK;
}

extern "C" int vkr_context_set_frames_in_flight(vkr_context* c, uint32_t n)
{
	return guarded("vkr_context_set_frames_in_flight", [&] { return set_lanes_impl(c, n); });
}
extern "C" int vkr_profile_reset(vkr_context* c)
{
	return guarded("vkr_profile_reset", [&] { return profile_reset_impl(c); });
}
extern "C" int vkr_profile_get(vkr_context* c, int stage, double* ms, uint64_t* launches)
{
	return guarded("vkr_profile_get", [&] { return profile_get_impl(c, stage, ms, launches); });
}
