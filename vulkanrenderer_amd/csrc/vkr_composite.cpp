// vkr_composite.cpp — sort-first compositing across the GPUs of a node behind the C ABI: every rank renders its rows of
// the frame (a band, or every world-th stripe), and the RGBA8 rows meet in one grouped RCCL send / receive on the
// context's own stream, straight into their place in the destination frame.
//
// The reference renders on one queue of one device (utils/vulkan_window.cpp:223-255) and presents from it
// (main/main.cpp:1391-1405); this is what stands in for "present" when eight GPUs share a frame.  xGMI is point to
// point, so a gather to one rank is world-1 independent transfers, one per link into the root, and the all-to-all
// variant (every rank ends up with the frame) uses every link once in each direction.
//
// RCCL is loaded at run time (dlopen), so a host without it can still render on one GPU; a process that already has
// an RCCL loaded (PyTorch ships one) shares that copy.
#include "../../include/vkr_renderer.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

int vkr_set_error_ext(int code, const char* fmt, ...);

namespace {

struct Rccl {
	void* lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

std::mutex g_rcclLock;

int load_rccl()
{
	std::lock_guard<std::mutex> guard(g_rcclLock); // contexts of several host threads may create their compositors at once
	if (g_rccl.lib) return VKR_OK;
	// The RCCL that belongs to the HIP runtime this process runs on: a process can hold two ROCm installations (PyTorch
	// ships a HIP runtime and an RCCL of its own; whichever runtime was loaded first serves everybody), and a
	// communicator must be given streams of the runtime its library was built against.  So look beside the loaded
	// libamdhip64 first, then fall back to the search path.
	void* h = nullptr;
	Dl_info info;
	if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
		std::string dir(info.dli_fname);
		const size_t slash = dir.rfind('/');
		if (slash != std::string::npos) {
			dir.resize(slash);
			for (const char* n : { "/librccl.so.1", "/librccl.so" }) {
				h = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL);
				if (h) break;
			}
		}
	}
	const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	for (const char* n : names) {
		if (h) break;
		h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
	}
	if (!h) return vkr_set_error_ext(VKR_E_DEVICE, "RCCL not found (librccl.so.1): %s", dlerror());
	Rccl r;
	r.lib = h;
	bool ok = true;
	auto sym = [&](const char* name) { void* p = dlsym(h, name); if (!p) ok = false; return p; };
	r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
	r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
	r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
	r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
	r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
	r.Send = (decltype(r.Send))sym("ncclSend");
	r.Recv = (decltype(r.Recv))sym("ncclRecv");
	r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
	if (!ok) { dlclose(h); return vkr_set_error_ext(VKR_E_DEVICE, "RCCL library lacks an expected entry point"); }
	g_rccl = r;
	return VKR_OK;
}

#define RCCL_CHECK(x)                                                                                                  \
	do {                                                                                                               \
		ncclResult_t r_ = (x);                                                                                         \
		if (r_ != ncclSuccess) return vkr_set_error_ext(VKR_E_DEVICE, "%s failed: %s", #x, g_rccl.GetErrorString(r_)); \
	} while (0)

} // namespace

struct vkr_compositor {
	vkr_context* ctx = nullptr;
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1, device = 0;
};

// ---- the partition: which rows of the frame a rank renders ------------------------------------------------------
// stripe_height == 0: `world` horizontal bands of equal, even height (the last ones may be shorter or empty).
// stripe_height  > 0: stripes of that many rows dealt round-robin, rank r takes stripes r, r + world, ...; its rows
// are stored back to back (what vkr_frame_params::stripes renders).
extern "C" int vkr_sortfirst_rows(uint32_t height, uint32_t world, uint32_t rank, uint32_t stripe_height, uint32_t* first_row,
                                  uint32_t* row_count)
{
	if (!world || rank >= world) return vkr_set_error_ext(VKR_E_INVALID, "vkr_sortfirst_rows: rank %u of %u", rank, world);
	uint32_t first = 0, count = 0;
	if (stripe_height == 0) {
		uint32_t band = (height + world - 1) / world;
		band += band & 1u;
		const uint64_t y0 = (uint64_t)rank * band, y1 = (uint64_t)(rank + 1) * band;
		first = (uint32_t)(y0 < height ? y0 : height);
		count = (uint32_t)(y1 < height ? y1 : height) - first;
	} else {
		first = rank * stripe_height < height ? rank * stripe_height : height;
		for (uint64_t y = (uint64_t)rank * stripe_height; y < height; y += (uint64_t)world * stripe_height)
			count += (uint32_t)((y + stripe_height <= height) ? stripe_height : height - y);
	}
	if (first_row) *first_row = first;
	if (row_count) *row_count = count;
	return VKR_OK;
}

extern "C" int vkr_composite_unique_id(uint8_t id[VKR_COMPOSITE_ID_BYTES])
{
	if (!id) return vkr_set_error_ext(VKR_E_INVALID, "vkr_composite_unique_id: NULL");
	static_assert(sizeof(ncclUniqueId) <= VKR_COMPOSITE_ID_BYTES, "the id blob holds an ncclUniqueId");
	int rc = load_rccl();
	if (rc) return rc;
	ncclUniqueId u;
	RCCL_CHECK(g_rccl.GetUniqueId(&u));
	memset(id, 0, VKR_COMPOSITE_ID_BYTES);
	memcpy(id, &u, sizeof u);
	return VKR_OK;
}

extern "C" int vkr_compositor_create(vkr_context* ctx, const uint8_t id[VKR_COMPOSITE_ID_BYTES], int rank, int world, vkr_compositor** out)
{
	if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world)
		return vkr_set_error_ext(VKR_E_INVALID, "vkr_compositor_create: bad argument (rank %d of %d)", rank, world);
	*out = nullptr;
	int rc = load_rccl();
	if (rc) return rc;
	vkr_compositor* c = new (std::nothrow) vkr_compositor();
	if (!c) return vkr_set_error_ext(VKR_E_NOMEM, "out of host memory");
	c->ctx = ctx; c->rank = rank; c->world = world; c->device = vkr_context_device(ctx);
	if (hipSetDevice(c->device) != hipSuccess) { delete c; return vkr_set_error_ext(VKR_E_DEVICE, "hipSetDevice(%d) failed", c->device); }
	ncclUniqueId u;
	memcpy(&u, id, sizeof u);
	ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
	if (r != ncclSuccess) { delete c; return vkr_set_error_ext(VKR_E_DEVICE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r)); }
	*out = c;
	return VKR_OK;
}

extern "C" void vkr_compositor_destroy(vkr_compositor* c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize(); // exchanges may be queued on any lane's stream
	if (c->comm) (void)g_rccl.CommDestroy(c->comm);
	delete c;
}

// One grouped exchange on the stream of the frame submitted last, behind that frame and in front of whatever that
// lane is handed next (so its next frame cannot overwrite `local` under the send).  Each contiguous run of rows is one
// transfer: a band is one run; a stripe partition has one run per stripe.  A rank's own rows travel through RCCL as
// well (send to self) unless they already sit in place.
extern "C" int vkr_composite_gather(vkr_compositor* c, uint32_t width, uint32_t height, uint32_t stripe_height, const void* local,
                                    void* frame, int root)
{
	return vkr_composite_gather_rows(c, (size_t)width * 4, height, stripe_height, local, frame, root);
}

extern "C" int vkr_composite_gather_rows(vkr_compositor* c, size_t row_bytes, uint32_t height, uint32_t stripe_height, const void* local,
                                         void* frame, int root)
{
	if (!c || !row_bytes || !height) return vkr_set_error_ext(VKR_E_INVALID, "vkr_composite_gather: bad argument");
	if (root >= c->world) return vkr_set_error_ext(VKR_E_INVALID, "vkr_composite_gather: root %d of %d", root, c->world);
	const bool receiver = root < 0 || root == c->rank;
	if (receiver && !frame) return vkr_set_error_ext(VKR_E_INVALID, "vkr_composite_gather: this rank receives the frame but `frame` is NULL");
	const size_t rowBytes = row_bytes;
	const uint32_t W = (uint32_t)c->world;
	const uint32_t runH = stripe_height ? stripe_height : 0;
	{
		// a rank without rows (more ranks than bands of two rows) has nothing to send and needs no `local`
		uint32_t myFirst = 0, myCount = 0;
		vkr_sortfirst_rows(height, W, (uint32_t)c->rank, stripe_height, &myFirst, &myCount);
		if (!local && myCount) return vkr_set_error_ext(VKR_E_INVALID, "vkr_composite_gather: `local` is NULL but this rank owns %u rows", myCount);
		// `local` may be the rank's band in its place inside `frame` (then nothing is copied); any other overlap would
		// make a send and a receive of one group touch the same bytes
		if (local && frame && myCount) {
			const char* l0 = (const char*)local; const char* l1 = l0 + (size_t)myCount * rowBytes;
			const char* f0 = (const char*)frame; const char* f1 = f0 + (size_t)height * rowBytes;
			const bool inPlace = !runH && l0 == f0 + (size_t)myFirst * rowBytes;
			if (!inPlace && l0 < f1 && f0 < l1)
				return vkr_set_error_ext(VKR_E_INVALID, "vkr_composite_gather: `local` overlaps `frame` without being this rank's band in its place");
		}
	}
	if (hipSetDevice(c->device) != hipSuccess) return vkr_set_error_ext(VKR_E_DEVICE, "hipSetDevice(%d) failed", c->device);
	// the stream of the frame submitted last: with frames in flight that is the lane's own stream, so the exchange is
	// ordered behind that frame only and the next frame (on another lane) renders meanwhile
	hipStream_t stream = (hipStream_t)vkr_context_frame_stream(c->ctx);
	RCCL_CHECK(g_rccl.GroupStart());
	ncclResult_t bad = ncclSuccess;
	for (uint32_t r = 0; r < W && bad == ncclSuccess; r++) {
		// the runs of rank r: (first row in the frame, rows, first row inside r's local buffer)
		uint32_t first = 0, count = 0;
		vkr_sortfirst_rows(height, W, r, stripe_height, &first, &count);
		uint32_t localRow = 0;
		for (uint64_t y = first; count && y < height; y += runH ? (uint64_t)W * runH : height) {
			const uint32_t rows = runH ? (uint32_t)((y + runH <= height) ? runH : height - y) : count;
			char* dst = frame ? (char*)frame + (size_t)y * rowBytes : nullptr;
			const char* src = (const char*)local + (size_t)localRow * rowBytes;
			const size_t bytes = (size_t)rows * rowBytes;
			if ((int)r == c->rank) {
				// my rows: to the root (or to everybody)
				for (uint32_t peer = 0; peer < W && bad == ncclSuccess; peer++) {
					if (root >= 0 && (int)peer != root) continue;
					if ((int)peer == c->rank && dst == src) continue; // already in place
					bad = g_rccl.Send(src, bytes, ncclUint8, (int)peer, c->comm, stream);
				}
			}
			if (receiver && bad == ncclSuccess && !((int)r == c->rank && dst == src))
				bad = g_rccl.Recv(dst, bytes, ncclUint8, (int)r, c->comm, stream);
			localRow += rows;
			if (!runH) break;
		}
	}
	ncclResult_t endr = g_rccl.GroupEnd();
	if (bad != ncclSuccess) return vkr_set_error_ext(VKR_E_DEVICE, "RCCL send/receive failed: %s", g_rccl.GetErrorString(bad));
	if (endr != ncclSuccess) return vkr_set_error_ext(VKR_E_DEVICE, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(endr));
	return VKR_OK;
}
