/*
 * vkr_renderer.h — C ABI of the MI355X-native Sun Temple renderer.
 *
 * One call, vkr_render(), replaces the body of the reference's
 * record_commands() + submit_commands() + present
 * (main/main.cpp:3709-4342, called from the frame loop at main/main.cpp:1367-1400):
 * it takes the same per-frame information (the five uniform blocks, the
 * extent, the three UserState mode fields) and produces the image the
 * reference would present, as tightly packed sRGB-encoded RGBA8, plus the
 * integer side buffers used for parity checks.
 *
 * Conventions
 *  - Every function returns 0 on success and a non-zero VKR_E_* code on
 *    failure; vkr_last_error() then returns a thread-local message.  No C++
 *    exception crosses this boundary (the reference throws labutils::Error,
 *    utils/error.hpp:16-26, and catches it once in main, main/main.cpp:1413-1418).
 *  - Matrices are column-major float[16], exactly glm::mat4 in memory.
 *  - A context belongs to one GPU and is not thread-safe; different contexts
 *    are independent.
 *  - Pointers in vkr_frame_out may be host or device memory (unified
 *    addressing tells them apart); NULL means "not wanted".
 */
#ifndef VKR_RENDERER_H
#define VKR_RENDERER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define VKR_API __attribute__((visibility("default")))
#else
#define VKR_API
#endif

#define VKR_OK 0
#define VKR_E_INVALID 1   /* bad argument */
#define VKR_E_DEVICE 2    /* HIP runtime error */
#define VKR_E_IO 3        /* file could not be read / parsed */
#define VKR_E_NOMEM 4     /* host or device allocation failed */
#define VKR_E_LIMIT 5     /* a stated size limit was exceeded */

#define VKR_NO_TEXTURE 0xffffffffu /* main/baked_model.hpp:92-94 */
#define VKR_NUM_LIGHTS 21          /* main/main.cpp:222, deferredShading.frag:5 */
#define VKR_NO_TRIANGLE 0xffffffffu

typedef struct vkr_context vkr_context; /* one per GPU */
typedef struct vkr_scene vkr_scene;     /* device-resident copy of a BakedModel */

/* ---- scene hand-off: mirrors BakedModel (main/baked_model.hpp:80-115) ---- */

/* One decoded texture: what stbi_load(path,...,4) returns after the vertical
 * flip the reference asks for (utils/vkimage.cpp:62-65), i.e. level 0 as it
 * is copied into the VkImage.  space: 0 = unorm, 1 = srgb
 * (ETextureSpace, main/baked_model.hpp:74-78). */
typedef struct vkr_texture_desc {
	const uint8_t* rgba8; /* width*height*4 bytes, row 0 first */
	uint32_t width, height;
	uint8_t space;
	uint8_t channels; /* BakedTextureInfo::channels, informational */
	uint8_t _pad[6];
} vkr_texture_desc;

/* BakedMaterialInfo (main/baked_model.hpp:87-95), same field order. */
typedef struct vkr_material_desc {
	uint32_t baseColorTextureId;
	uint32_t roughnessTextureId;
	uint32_t metalnessTextureId;
	uint32_t alphaMaskTextureId; /* VKR_NO_TEXTURE if none */
	uint32_t normalMapTextureId;
	uint32_t emissiveTextureId; /* never sampled, as in the reference */
} vkr_material_desc;

/* BakedMeshData (main/baked_model.hpp:97-108).  normals and tangents may be
 * NULL: no fragment shader of the reference reads them. */
typedef struct vkr_mesh_desc {
	uint32_t materialId;
	uint32_t vertexCount;
	uint32_t indexCount;
	uint32_t _pad;
	const float* positions;        /* vertexCount * 3 */
	const float* texcoords;        /* vertexCount * 2 */
	const float* normals;          /* vertexCount * 3 or NULL */
	const float* tangents;         /* vertexCount * 4 or NULL */
	const uint32_t* tangentsComp;  /* vertexCount, A2R10G10B10 packed TBN quaternion */
	const uint32_t* indices;       /* indexCount, triangle list */
} vkr_mesh_desc;

typedef struct vkr_scene_desc {
	uint32_t textureCount, materialCount, meshCount, _pad;
	const vkr_texture_desc* textures;
	const vkr_material_desc* materials;
	const vkr_mesh_desc* meshes;
} vkr_scene_desc;

/* ---- per-frame inputs: mirrors struct Uniforms (main/main.cpp:203-240) ---- */

typedef struct vkr_scene_uniform { /* glsl::SceneUniform, main/main.cpp:205-210 */
	float camera[16];
	float projection[16];
	float projCam[16];
	float camPos[4];
} vkr_scene_uniform;

typedef struct vkr_light_uniform { /* glsl::LightUniform, main/main.cpp:212-215 */
	float lightPos[4];
	float lightColour[4];
} vkr_light_uniform;

typedef struct vkr_debug_uniform { /* glsl::DebugUniform, main/main.cpp:217-219 */
	int32_t debug;
} vkr_debug_uniform;

typedef struct vkr_multiple_lights { /* glsl::MultipleLights, main/main.cpp:221-223 */
	vkr_light_uniform light[VKR_NUM_LIGHTS];
} vkr_multiple_lights;

typedef struct vkr_depth_mvp { /* glsl::DepthMVP, main/main.cpp:225-227 */
	float depthMVP[16];
} vkr_depth_mvp;

typedef struct vkr_uniforms { /* struct Uniforms, main/main.cpp:234-240 */
	vkr_scene_uniform sceneUniforms;
	vkr_light_uniform lightUniforms;
	vkr_debug_uniform debugUniforms;
	vkr_multiple_lights multipleLightsUniform;
	vkr_depth_mvp depthMVPUniform;
} vkr_uniforms;

/* flags for vkr_frame_params::flags */
#define VKR_FRAME_DEFAULT 0u
/* deferred mode only: also render the light's depth map (shadow_size^2) and hand it out through
 * vkr_frame_out::shadow_map.  The reference's deferred path has no shadow pass (main/main.cpp:3836-3946); this is
 * the "deferred + shadow map" stress composition, the map is not sampled by the lighting. */
#define VKR_FRAME_SHADOW_PASS 1u
/* deferred mode only: G-buffer resolve and lighting run as one kernel; the G-buffer stays in registers
 * (rounded to half precision as the RGBA16F attachments round it) and is written out only if
 * vkr_frame_out asks for it.  Same image, fewer microseconds; the two subpasses of the reference
 * (main/main.cpp:3836-3946) are no longer separately visible to a profiler. */
#define VKR_FRAME_FUSED_LIGHTING 2u

typedef struct vkr_frame_params {
	uint32_t width, height; /* VkExtent2D aExtent, main/main.cpp:3714 */
	uint32_t shadow_size;   /* 0 = 2048 (main/main.cpp:3958) */
	/* UserState mode fields (main/main.cpp:117-119); precedence as in
	 * record_commands: deferred && !mosaic, then !mosaic && debug==1, then
	 * debug 5/6, then other debug values, then mosaic (main/main.cpp:3836-4195). */
	int32_t debugVisualisation;
	int32_t mosaicEffect;
	int32_t deferredShading;
	/* sort-first sub-rectangle [x0,x1) x [y0,y1) in pixels, even-aligned;
	 * all four zero = the whole frame.  Output buffers are rectangle-sized. */
	uint32_t rect_x0, rect_y0, rect_x1, rect_y1;
	uint32_t flags;
	/* sort-first by interleaved stripes instead of one rectangle: 0 = off, else VKR_STRIPES(height, count, index): of
	 * the frame cut into stripes of `height` rows (a multiple of 32), this call renders stripes index, index + count,
	 * index + 2 count, ... and stores their rows back to back; output buffers hold vkr_sortfirst_rows() rows of the
	 * full width.  Foliage and other expensive regions are clustered on screen, so interleaved stripes balance the
	 * ranks better than bands.  Not combined with rect_* or the mosaic pass. */
	uint32_t stripes;
	vkr_uniforms uniforms;
} vkr_frame_params;
#define VKR_STRIPES(height, count, index) ((uint32_t)(height) | ((uint32_t)(count) << 16) | ((uint32_t)(index) << 24))

/* Output buffers, each (rect width * rect height) elements, row 0 = top,
 * tightly packed, except shadow_map which is shadow_size^2. */
typedef struct vkr_frame_out {
	void* rgba8;           /* the presented image: R8G8B8A8, RGB sRGB-encoded (utils/vulkan_window.cpp:349-360) */
	uint32_t* tri_id;      /* submission-order primitive index of the visible fragment, VKR_NO_TRIANGLE if none */
	uint32_t* depth_bits;  /* IEEE-754 bits of the D32 depth attachment */
	uint8_t* mip_level;    /* floor(textureQueryLod().x) of the base-colour texture (debug view 2), 0xff if no fragment */
	uint8_t* stencil;      /* S8 counter of the overdraw (5) / overshading (6) views */
	uint16_t* gbuf_normal; /* RGBA16F attachment 1 of the deferred pass (normal.xyz, metalness) */
	uint16_t* gbuf_albedo; /* RGBA16F attachment 2 of the deferred pass (albedo.rgb, roughness) */
	float* shadow_map;     /* D32 shadow map of the forward path */
} vkr_frame_out;

/* ---- functions ---- */

VKR_API const char* vkr_last_error(void);

/* Creates a context on HIP device `device` with its own stream. */
VKR_API int vkr_context_create(int device, vkr_context** out);
VKR_API void vkr_context_destroy(vkr_context* ctx);
/* Frames in flight.  The reference keeps as many frames in flight as the swapchain has images, each with its own
 * fence and semaphores (main/main.cpp:466-476, 1263-1296).  With n > 1 the context owns n lanes - HIP stream, scratch
 * and frame buffers each - and vkr_render hands consecutive frames to consecutive lanes, so the passes of one frame
 * that leave compute units idle (binning, the tails of the rasteriser) overlap the shading of another.  Frames may
 * complete out of order: n consecutive frames need n distinct sets of output buffers, as swapchain images are
 * distinct.  Default 1.  Fails while frames are pending. */
#define VKR_MAX_FRAMES_IN_FLIGHT 8
VKR_API int vkr_context_set_frames_in_flight(vkr_context* ctx, uint32_t n);
/* The per-image fence of the reference's loop (vkWaitForFences(frameDone[frameIndex]), main/main.cpp:1268): blocks
 * until the lane the next vkr_render will use has finished the frames it holds, i.e. until the output buffers
 * handed over n frames ago are complete and may be reused.  With one lane it is vkr_wait. */
VKR_API int vkr_wait_next_lane(vkr_context* ctx);

/* Replaces load_baked_model() + the texture / mesh upload loops
 * (main/baked_model.hpp:117, main/main.cpp:752-760, 870-1095): copies the
 * arrays to the GPU, builds the mip chains (utils/vkimage.cpp:163-216). */
VKR_API int vkr_scene_create(vkr_context* ctx, const vkr_scene_desc* desc, vkr_scene** out);
/* Same, reading a ".comp5892mesh" file (variant "21-tan",
 * main/baked_model.cpp:70-174) and the PNG textures it names. */
VKR_API int vkr_scene_load(vkr_context* ctx, const char* path, vkr_scene** out);
/* Waits for the frames that use the scene, then releases it.  Destroy a context's scenes before the context. */
VKR_API void vkr_scene_destroy(vkr_scene* scene);
/* Host-only: decodes one PNG the way the reference's texture loader does,
 * stbi_load(path, &w, &h, &c, 4) after stbi_set_flip_vertically_on_load(1)
 * (utils/vkimage.cpp:62-65): 8-bit RGBA, bottom row first.  *rgba8 is
 * allocated by the library; release it with vkr_free(). */
VKR_API int vkr_decode_png_file(const char* path, uint8_t** rgba8, uint32_t* width, uint32_t* height);
VKR_API void vkr_free(void* p);
/* Host-only: writes a frame (tightly packed R8G8B8A8, as vkr_render delivers it) as a PNG file.  The reference only
 * presents to the swapchain (vkQueuePresentKHR, main/main.cpp:1391-1405); this is the way out for a headless host.
 * flip: 0 = row 0 is the top row (vkr_render's order), 1 = row 0 is the bottom row (vkr_decode_png_file's order). */
VKR_API int vkr_write_png_file(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height, int flip);
/* number of triangles / vertices held by the scene */
VKR_API int vkr_scene_stats(const vkr_scene* scene, uint64_t* triangles, uint64_t* vertices, uint64_t* texture_bytes);
/* Reads mip level `level` of texture `tex` back (RGBA8); for parity tests of
 * the mip chain against vkCmdBlitImage semantics (utils/vkimage.cpp:165-216). */
VKR_API int vkr_scene_read_texture_level(const vkr_scene* scene, uint32_t tex, uint32_t level,
                                 uint8_t* rgba8_out, uint32_t* width, uint32_t* height);

/* Replaces record_commands + submit_commands for one frame
 * (main/main.cpp:1367-1389).  Asynchronous on the context's stream; params
 * are snapshotted on entry like vkCmdUpdateBuffer does (main/main.cpp:3741).
 * Output pointers must stay valid until vkr_wait(). */
VKR_API int vkr_render(vkr_context* ctx, const vkr_scene* scene, const vkr_frame_params* params,
               const vkr_frame_out* out);
/* vkr_render for callers that cannot allocate.  It has no counterpart in the reference; it is for hosts that may
 * not call hipMalloc while they render: code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation.  Same parameters, same kernels and the same output as vkr_render, but every device buffer of the frame
 * (triangle records, tile bins, visibility buffer, G-buffer, shadow map, ...) is a slice of one fixed array of
 * VKR_STATIC_SCRATCH_BYTES that the library's device image holds on each GPU it runs on until the process ends, and
 * nothing is created, grown or freed: no hipMalloc / hipFree / hipHostMalloc, no events, no helper stream (a forward
 * frame's light pass runs on the frame's own stream), no stage timing.
 *  - A frame that does not fit (about 200 bytes per triangle of the scene + 36 bytes per pixel + 4 bytes per
 *    shadow-map texel; the SunTemple-proxy scene fits up to 1920x1080) is rejected with VKR_E_LIMIT; a pass whose
 *    tile bins outgrow their slice is reported with VKR_E_LIMIT by vkr_wait and the frame is incomplete (vkr_render
 *    would have grown the bins and rendered it again).
 *  - At most VKR_STATIC_MAX_PENDING such frames between two vkr_wait calls.
 *  - Frames run on the context's stream in submission order, among its vkr_render frames.  All contexts of a
 *    process on one GPU share that GPU's array, one context at a time: the context whose allocation-free frames are
 *    pending owns the array until its vkr_wait (or its destruction), and a second context that submits meanwhile is
 *    refused with VKR_E_LIMIT rather than left to race.  Calls on one context still must not overlap each other (a
 *    context is not thread-safe).
 *  - Output pointers in device memory (or pinned host memory) keep the whole call free of staging allocations
 *    inside the HIP runtime. */
#define VKR_STATIC_SCRATCH_BYTES (256u << 20)
#define VKR_STATIC_MAX_PENDING 64
VKR_API int vkr_render_static(vkr_context* ctx, const vkr_scene* scene, const vkr_frame_params* params,
                      const vkr_frame_out* out);
/* The fence: blocks until every frame submitted on the context is complete
 * (vkWaitForFences, main/main.cpp:1268). */
VKR_API int vkr_wait(vkr_context* ctx);

/* Host-side uniform helpers, same arithmetic as the reference's
 * update_scene_uniforms (main/main.cpp:3681-3693),
 * update_depth_mvp_uniforms (main/main.cpp:3699-3707) and the constant light
 * tables of the frame loop (main/main.cpp:1313-1352). */
VKR_API void vkr_update_scene_uniforms(vkr_scene_uniform* out, uint32_t width, uint32_t height,
                               const float camera2world[16]);
VKR_API void vkr_update_depth_mvp_uniforms(vkr_depth_mvp* out, const float lightPos[4]);
VKR_API void vkr_default_lights(vkr_light_uniform* forwardLight, vkr_multiple_lights* braziers);
/* Fills every block of `u` for a camera pose and debug mode, as the frame
 * loop does at main/main.cpp:1307-1359. */
VKR_API void vkr_make_uniforms(vkr_uniforms* u, uint32_t width, uint32_t height,
                       const float camera2world[16], int32_t debugVisualisation);

/* The camera part of the reference's UserState (main/main.cpp:111-125) and
 * its input slots in EInputState order (main/main.cpp:98-109), for hosts that
 * replay a recorded input trace as a stream of poses instead of keeping glm. */
enum {
	VKR_INPUT_FORWARD = 0, VKR_INPUT_BACKWARD, VKR_INPUT_STRAFE_LEFT, VKR_INPUT_STRAFE_RIGHT,
	VKR_INPUT_LEVITATE, VKR_INPUT_SINK, VKR_INPUT_FAST, VKR_INPUT_SLOW, VKR_INPUT_MOUSING,
	VKR_INPUT_COUNT
};
typedef struct vkr_user_state {
	uint8_t inputMap[VKR_INPUT_COUNT]; /* non-zero = held */
	uint8_t wasMousing;
	uint8_t _pad[2];
	float mouseX, mouseY;
	float previousX, previousY;
	float camera2world[16];            /* column-major, identity at start */
	/* the mode fields the key callback sets (main/main.cpp:117-120); `shadows` is toggled but never read by the
	 * reference's draw code */
	int32_t debugVisualisation;        /* 1 at start */
	int32_t mosaicEffect, deferredShading, shadows;
} vkr_user_state;
/* Zeroes the inputs, sets camera2world to identity and debugVisualisation to 1 (main/main.cpp:111-125). */
VKR_API void vkr_user_state_init(vkr_user_state* state);
/* The three GLFW callbacks that feed UserState, for hosts that replay a recorded event stream.  Key, button and
 * action values are GLFW's (GLFW_KEY_W = 87, GLFW_KEY_1 = 49, GLFW_KEY_LEFT_SHIFT = 340, ...; GLFW_RELEASE = 0,
 * GLFW_PRESS = 1, GLFW_REPEAT = 2; GLFW_MOUSE_BUTTON_RIGHT = 1).
 *   key:    glfw_callback_key_press (main/main.cpp:1422-1514): on release 1-4 pick the view, 5 toggles mosaic, 6 / 7
 *           pick overdraw / overshading, 8 toggles deferred shading, 9 toggles shadows; W S A D E Q, shift, control
 *           are held while not released.  Returns 1 if the key was escape pressed (the reference closes the window).
 *   button: glfw_callback_button (main/main.cpp:1516-1529): a right-button press toggles mouse look.
 *   motion: glfw_callback_motion (main/main.cpp:1531-1537). */
VKR_API int vkr_user_state_key(vkr_user_state* state, int key, int action);
VKR_API void vkr_user_state_button(vkr_user_state* state, int button, int action);
VKR_API void vkr_user_state_motion(vkr_user_state* state, double x, double y);
/* Copies the mode fields into the frame parameters, as main() hands them to record_commands (main/main.cpp:1367-1389). */
VKR_API void vkr_frame_params_set_modes(vkr_frame_params* params, const vkr_user_state* state);
/* One step of update_user_state (main/main.cpp:3643-3679): mouse look while
 * `mousing` is held on two consecutive steps (0.01 rad per pixel, pitch then
 * yaw, post-multiplied), then translation by elapsedSeconds * 1.7 (x7 fast,
 * x0.05 slow) along the camera's own axes, in the reference's key order. */
VKR_API void vkr_update_user_state(vkr_user_state* state, float elapsedSeconds);

/* ---- sort-first compositing across the GPUs of a node ---- */

/* The reference presents from the one device it renders on (main/main.cpp:1391-1405, utils/vulkan_window.cpp:223-255);
 * when several GPUs share a frame (one process and one context per GPU, scene replicated) each renders its rows and
 * the rows meet in one grouped RCCL send / receive over xGMI.
 * Rows of rank `rank` out of `world`: stripe_height 0 = horizontal bands of equal, even height (first_row, row_count);
 * stripe_height > 0 = stripes dealt round-robin as VKR_STRIPES renders them (first_row = the first stripe's row,
 * row_count = all of the rank's rows, stored back to back). */
VKR_API int vkr_sortfirst_rows(uint32_t height, uint32_t world, uint32_t rank, uint32_t stripe_height, uint32_t* first_row,
                               uint32_t* row_count);
typedef struct vkr_compositor vkr_compositor;
#define VKR_COMPOSITE_ID_BYTES 128
/* One rank creates the id (an ncclUniqueId), the host hands it to the others by whatever channel it has (MPI, a file,
 * torch.distributed), every rank then creates its compositor: a collective call, like ncclCommInitRank. */
VKR_API int vkr_composite_unique_id(uint8_t id[VKR_COMPOSITE_ID_BYTES]);
VKR_API int vkr_compositor_create(vkr_context* ctx, const uint8_t id[VKR_COMPOSITE_ID_BYTES], int rank, int world,
                                  vkr_compositor** out);
VKR_API void vkr_compositor_destroy(vkr_compositor* c);
/* Collective, asynchronous on the stream of the frame submitted last (vkr_context_frame_stream: behind that frame, in
 * front of the next frame of the same lane; no host wait, no allocation): `local` = this rank's rows of the
 * width x height RGBA8 frame as vkr_render delivered them (device memory; NULL only for a rank that owns no rows),
 * `frame` = the whole frame on the receiving rank(s) (device memory; may be NULL elsewhere).
 * root >= 0: gather to that rank; root < 0: every rank receives the frame.  A band rendered straight into its place
 * (local == frame + first_row * width * 4) is not copied; any other overlap of `local` and `frame` is refused
 * (VKR_E_INVALID).  With more than one frame in flight (vkr_context_set_frames_in_flight) give every lane a `local`
 * (and, where the frames are read, a `frame`) of its own, as a swapchain has one image per frame in flight: the
 * exchange of one frame then runs under the rendering of the next.  Every rank must submit the same sequence of
 * frames and exchanges with the same number of lanes. */
VKR_API int vkr_composite_gather(vkr_compositor* c, uint32_t width, uint32_t height, uint32_t stripe_height, const void* local,
                                 void* frame, int root);
/* The same exchange for rows of any size: the side buffers of a frame (tri_id and depth_bits are 4 bytes per pixel,
 * mip_level and stencil 1, the G-buffer attachments 8) travel like the image. */
VKR_API int vkr_composite_gather_rows(vkr_compositor* c, size_t row_bytes, uint32_t height, uint32_t stripe_height, const void* local,
                                      void* frame, int root);

/* ---- measurement ---- */

/* Stage timing with HIP events on the context's stream.  Stage names are
 * returned by vkr_stage_name(); times are accumulated per stage since the
 * last vkr_profile_reset().  on: 0 = off, 1 = every stage, 2 = only the
 * deferred lighting kernel (two event records per frame). */
VKR_API int vkr_profile_enable(vkr_context* ctx, int on);
VKR_API int vkr_profile_reset(vkr_context* ctx);
VKR_API int vkr_stage_count(void);
VKR_API const char* vkr_stage_name(int stage);
/* Synchronises, then returns accumulated milliseconds and launch count. */
VKR_API int vkr_profile_get(vkr_context* ctx, int stage, double* total_ms, uint64_t* launches);
/* The HIP stream (hipStream_t) the context launches on, and its device. */
VKR_API void* vkr_context_stream(vkr_context* ctx);
VKR_API int vkr_context_device(vkr_context* ctx);
/* With frames in flight every lane has a stream of its own: the stream the frame submitted last runs on (the
 * context's stream with one lane), and that lane's index (0 .. lanes - 1).  Work queued on this stream is ordered
 * behind that frame and in front of the lane's next one.  The reference's counterpart is the per-swapchain-image
 * command buffer and fence (main/main.cpp:466-476, 1268). */
VKR_API void* vkr_context_frame_stream(vkr_context* ctx);
VKR_API int vkr_context_frame_lane(vkr_context* ctx);

/* Device memory for a host that has no HIP of its own (a plain C program): buffers on the context's GPU to render
 * into (vkr_frame_out accepts device pointers) and to exchange with vkr_composite_gather, and a copy in either
 * direction that is ordered behind everything queued on the context's stream and complete on return.
 * Replaces: lut::create_buffer + the staging copies of the reference (utils/vkbuffer.cpp:49-69, utils/vkimage.cpp:76-143). */
VKR_API int vkr_device_malloc(vkr_context* ctx, size_t bytes, void** out);
VKR_API void vkr_device_free(vkr_context* ctx, void* ptr);
VKR_API int vkr_device_copy(vkr_context* ctx, void* dst, const void* src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* VKR_RENDERER_H */
