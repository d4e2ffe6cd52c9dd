/* vkr_bench.c — headless host of libvkr_hip.so in plain C99: what main() of the
 * reference does per frame (main/main.cpp:1103-1406: update state, update
 * uniforms, record + submit, present) with the window, swapchain and input
 * handling taken away.  It loads a baked model, flies the camera along
 *   pose i = translate(0, -1, -90 i/511) * rotateY(0.35 sin(2 pi i / 128)),
 * renders `frames` frames with `lanes` frames in flight and prints one line
 * with the frame rate; --png writes the last frame.
 *
 *   vkr_bench scene.comp5892mesh [--width W] [--height H] [--frames N]
 *             [--first I] [--mode deferred|forward|mosaic|debug2..6]
 *             [--fused] [--lanes L] [--device D] [--png out.png]
 *             [--ranks N --rank R --uid-file F [--session S] [--stripe-height S]]
 *
 * With --ranks the program is one of N processes (one per GPU, started by whatever the host has: a shell loop,
 * mpirun, a job scheduler) that share every frame sort-first: each renders its rows - a band, or with
 * --stripe-height every N-th stripe - into device memory, and the rows meet on rank 0 through the library's RCCL
 * compositor.  Rank 0 creates the communicator's id and leaves it in file F, the others pick it up there.  The file
 * also names the session it belongs to (--session S; by default the id of the process that started the ranks), and a
 * reader takes no id of another session: rank 0 removes whatever an earlier run left behind before it writes, and
 * removes its own file once every rank has joined, but a reader may look before either has happened.
 *
 * Build: make -C tools   (gcc; no HIP headers needed, the ABI is plain C)
 */
#define _POSIX_C_SOURCE 200809L
#include "vkr_renderer.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* column-major camera2world of pose i, as glm builds it */
static void fly_pose(int i, float m[16])
{
	const double a = 0.35 * sin(2.0 * 3.14159265358979323846 * (double)i / 128.0);
	const double c = cos(a), s = sin(a);
	memset(m, 0, 16 * sizeof(float));
	m[0] = (float)c; m[2] = (float)-s;
	m[5] = 1.0f;
	m[8] = (float)s; m[10] = (float)c;
	m[13] = -1.0f; m[14] = (float)(-90.0 * (double)i / 511.0); m[15] = 1.0f;
}

/* the communicator id travels through a file - the id, then the session's name - written under another name and
 * renamed, so a reader never sees half of it */
#define SESSION_BYTES 64
static int publish_id(const char* path, const uint8_t* id, const char* session)
{
	char tmp[4096];
	char tag[SESSION_BYTES];
	memset(tag, 0, sizeof tag);
	snprintf(tag, sizeof tag, "%s", session);
	snprintf(tmp, sizeof tmp, "%s.tmp", path);
	(void)remove(path); /* an earlier run's file */
	FILE* f = fopen(tmp, "wb");
	if (!f) return 1;
	const int ok = fwrite(id, 1, VKR_COMPOSITE_ID_BYTES, f) == VKR_COMPOSITE_ID_BYTES && fwrite(tag, 1, sizeof tag, f) == sizeof tag;
	return (fclose(f) == 0 && ok && rename(tmp, path) == 0) ? 0 : 1;
}

static int fetch_id(const char* path, uint8_t* id, const char* session)
{
	char want[SESSION_BYTES], tag[SESSION_BYTES];
	memset(want, 0, sizeof want);
	snprintf(want, sizeof want, "%s", session);
	for (int tries = 0; tries < 600; tries++) { /* up to a minute */
		FILE* f = fopen(path, "rb");
		if (f) {
			const int ok = fread(id, 1, VKR_COMPOSITE_ID_BYTES, f) == VKR_COMPOSITE_ID_BYTES && fread(tag, 1, sizeof tag, f) == sizeof tag;
			fclose(f);
			if (ok && !memcmp(tag, want, sizeof tag)) return 0; /* anything else is another run's file: keep waiting for ours */
		}
		{ struct timespec nap = { 0, 100000000L }; nanosleep(&nap, NULL); }
	}
	return 1;
}

static int fail(const char* what)
{
	fprintf(stderr, "vkr_bench: %s: %s\n", what, vkr_last_error());
	return 1;
}

int main(int argc, char** argv)
{
	const char* scenePath = NULL; const char* png = NULL; const char* mode = "deferred";
	unsigned width = 1920, height = 1080, lanes = 1;
	int frames = 64, first = 0, device = 0, fused = 0, ranks = 0, rank = 0;
	unsigned stripeH = 0;
	const char* uidFile = NULL;
	const char* session = NULL;
	for (int i = 1; i < argc; i++) {
		const char* a = argv[i];
		const char* v = (i + 1 < argc) ? argv[i + 1] : NULL;
		if (!strcmp(a, "--fused")) fused = 1;
		else if (a[0] == '-' && a[1] == '-' && !v) { fprintf(stderr, "vkr_bench: %s needs a value\n", a); return 2; }
		else if (!strcmp(a, "--width")) { width = (unsigned)atoi(v); i++; }
		else if (!strcmp(a, "--height")) { height = (unsigned)atoi(v); i++; }
		else if (!strcmp(a, "--frames")) { frames = atoi(v); i++; }
		else if (!strcmp(a, "--first")) { first = atoi(v); i++; }
		else if (!strcmp(a, "--lanes")) { lanes = (unsigned)atoi(v); i++; }
		else if (!strcmp(a, "--device")) { device = atoi(v); i++; }
		else if (!strcmp(a, "--mode")) { mode = v; i++; }
		else if (!strcmp(a, "--png")) { png = v; i++; }
		else if (!strcmp(a, "--ranks")) { ranks = atoi(v); i++; }
		else if (!strcmp(a, "--rank")) { rank = atoi(v); i++; }
		else if (!strcmp(a, "--uid-file")) { uidFile = v; i++; }
		else if (!strcmp(a, "--session")) { session = v; i++; }
		else if (!strcmp(a, "--stripe-height")) { stripeH = (unsigned)atoi(v); i++; }
		else if (a[0] == '-') { fprintf(stderr, "vkr_bench: unknown option %s\n", a); return 2; }
		else scenePath = a;
	}
	if (!scenePath || frames < 1 || lanes < 1 || lanes > VKR_MAX_FRAMES_IN_FLIGHT || (ranks && (ranks < 1 || rank < 0 || rank >= ranks || !uidFile))) {
		fprintf(stderr, "usage: vkr_bench scene.comp5892mesh [--width W] [--height H] [--frames N] [--first I] "
		                "[--mode deferred|forward|mosaic|debug2..6] [--fused] [--lanes 1..%d] [--device D] [--png out.png] "
		                "[--ranks N --rank R --uid-file F [--session S] [--stripe-height S]]\n",
		        VKR_MAX_FRAMES_IN_FLIGHT);
		return 2;
	}
	char sessionDefault[SESSION_BYTES];
	if (!session) { /* the ranks of one launch share their parent */
		snprintf(sessionDefault, sizeof sessionDefault, "parent-%ld", (long)getppid());
		session = sessionDefault;
	}
	/* UserState's mode fields (main/main.cpp:117-119) */
	int debug = 1, mosaic = 0, deferred = 0;
	if (!strcmp(mode, "deferred")) deferred = 1;
	else if (!strcmp(mode, "mosaic")) mosaic = 1;
	else if (!strncmp(mode, "debug", 5) && mode[5] >= '2' && mode[5] <= '6' && !mode[6]) debug = mode[5] - '0';
	else if (strcmp(mode, "forward")) { fprintf(stderr, "vkr_bench: unknown mode %s\n", mode); return 2; }

	vkr_context* ctx = NULL;
	vkr_scene* scene = NULL;
	if (vkr_context_create(device, &ctx)) return fail("vkr_context_create");
	if (vkr_context_set_frames_in_flight(ctx, lanes)) return fail("vkr_context_set_frames_in_flight");
	const double t0 = now_s();
	if (vkr_scene_load(ctx, scenePath, &scene)) return fail("vkr_scene_load");
	uint64_t tris = 0, verts = 0, texBytes = 0;
	vkr_scene_stats(scene, &tris, &verts, &texBytes);
	const double tLoad = now_s() - t0;

	/* one host image per lane: frames in flight may finish out of order */
	const size_t imageBytes = (size_t)width * height * 4;
	uint8_t* images[VKR_MAX_FRAMES_IN_FLIGHT] = { 0 };
	for (unsigned l = 0; l < lanes; l++) {
		images[l] = (uint8_t*)malloc(imageBytes);
		if (!images[l]) { fprintf(stderr, "vkr_bench: out of memory\n"); return 1; }
	}

	vkr_frame_params fp;
	memset(&fp, 0, sizeof fp);
	fp.width = width; fp.height = height;
	fp.debugVisualisation = debug; fp.mosaicEffect = mosaic; fp.deferredShading = deferred;
	fp.flags = (fused && deferred) ? VKR_FRAME_FUSED_LIGHTING : VKR_FRAME_DEFAULT;

	/* sort-first: this rank's rows go into `local` (device memory), the whole frame arrives in `frame` on rank 0 */
	vkr_compositor* comp = NULL;
	void* local[VKR_MAX_FRAMES_IN_FLIGHT] = { 0 }; void* frame[VKR_MAX_FRAMES_IN_FLIGHT] = { 0 }; /* one set per frame in flight */
	uint32_t firstRow = 0, myRows = height;
	if (ranks) {
		uint8_t id[VKR_COMPOSITE_ID_BYTES];
		if (rank == 0) {
			if (vkr_composite_unique_id(id)) return fail("vkr_composite_unique_id");
			if (publish_id(uidFile, id, session)) { fprintf(stderr, "vkr_bench: cannot write %s\n", uidFile); return 1; }
		} else if (fetch_id(uidFile, id, session)) { fprintf(stderr, "vkr_bench: no communicator id of session %s in %s\n", session, uidFile); return 1; }
		if (vkr_compositor_create(ctx, id, rank, ranks, &comp)) return fail("vkr_compositor_create");
		if (rank == 0) (void)remove(uidFile); /* every rank has joined, so every rank has read it */
		if (vkr_sortfirst_rows(height, (uint32_t)ranks, (uint32_t)rank, stripeH, &firstRow, &myRows)) return fail("vkr_sortfirst_rows");
		for (unsigned l = 0; l < lanes; l++) {
			if (vkr_device_malloc(ctx, (size_t)(myRows ? myRows : 1) * width * 4, &local[l])) return fail("vkr_device_malloc");
			if (rank == 0 && vkr_device_malloc(ctx, imageBytes, &frame[l])) return fail("vkr_device_malloc");
		}
		if (stripeH) fp.stripes = VKR_STRIPES(stripeH, ranks, rank);
		else { fp.rect_x0 = 0; fp.rect_y0 = firstRow; fp.rect_x1 = width; fp.rect_y1 = firstRow + myRows; }
	}

	const int warm = frames < 4 ? frames : 4;
	double tStart = 0.0;
	unsigned submitted = 0; /* frame n goes to lane n % lanes, and into that lane's image */
	for (int k = -warm; k < frames; k++) {
		if (k == 0) {
			if (vkr_wait(ctx)) return fail("vkr_wait");
			tStart = now_s();
		}
		const int i = (k < 0 ? k + warm : k);
		float cam[16];
		fly_pose((first + i) % 512, cam);
		vkr_make_uniforms(&fp.uniforms, width, height, cam, debug); /* update_scene_uniforms & co., main/main.cpp:1313-1365 */
		vkr_frame_out fo;
		memset(&fo, 0, sizeof fo);
		const unsigned lane = submitted % lanes;
		fo.rgba8 = ranks ? (uint8_t*)local[lane] : images[lane];
		/* this lane's image is about to be reused: wait for the frame that was rendered into it, as the
		 * reference waits on the fence of the swapchain image it is about to record into (main/main.cpp:1268) */
		if (vkr_wait_next_lane(ctx)) return fail("vkr_wait_next_lane");
		if (myRows && vkr_render(ctx, scene, &fp, &fo)) return fail("vkr_render");
		/* queued on the lane the frame went to: the rows leave while the next frame renders on the other lane */
		if (ranks && vkr_composite_gather(comp, width, height, stripeH, myRows ? local[lane] : NULL, frame[lane], 0)) return fail("vkr_composite_gather");
		submitted++;
	}
	if (vkr_wait(ctx)) return fail("vkr_wait");
	const double dt = now_s() - tStart;

	printf("{\"scene\": \"%s\", \"triangles\": %llu, \"vertices\": %llu, \"texture_bytes\": %llu, \"load_s\": %.2f, "
	       "\"mode\": \"%s\", \"fused\": %d, \"width\": %u, \"height\": %u, \"frames\": %d, \"frames_in_flight\": %u, "
	       "\"ms_per_frame\": %.4f, \"frames_per_s\": %.1f, \"readback\": \"%s\", \"ranks\": %d, \"rank\": %d, \"rows\": %u}\n",
	       scenePath, (unsigned long long)tris, (unsigned long long)verts, (unsigned long long)texBytes, tLoad, mode,
	       fused, width, height, frames, lanes, dt / frames * 1e3, frames / dt, ranks ? "device rows, gathered on rank 0" : "host images",
	       ranks ? ranks : 1, rank, (unsigned)myRows);
	int rc = 0;
	if (ranks && rank == 0 && vkr_device_copy(ctx, images[0], frame[(submitted - 1) % lanes], imageBytes)) rc = fail("vkr_device_copy");
	if (png && (!ranks || rank == 0) && vkr_write_png_file(png, images[ranks ? 0 : (submitted - 1) % lanes], width, height, 0)) rc = fail("vkr_write_png_file");
	if (comp) vkr_compositor_destroy(comp);
	for (unsigned l = 0; l < lanes; l++) { vkr_device_free(ctx, local[l]); vkr_device_free(ctx, frame[l]); }
	for (unsigned l = 0; l < lanes; l++) free(images[l]);
	vkr_scene_destroy(scene);
	vkr_context_destroy(ctx);
	return rc;
}
