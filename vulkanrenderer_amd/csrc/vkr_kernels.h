// vkr_kernels.h — kernel argument blocks and host-side launch functions.
#pragma once
#include "vkr_device.h"

struct SetupArgs {
	SceneDev sc;
	ViewDev view;
	Mat4 M;                    // object -> clip space of the pass (projCam, or the light's depthMVP)
	uint32_t* zeroBase;        // the counter set of the NEXT pass: k_setup zeroes it while this pass uses the other one
	uint32_t zeroWords;
	uint32_t first, count;     // triangle range, submission order A
	int cullOpaque, cullAlpha; // 0 none, 1 back; alpha-masked meshes start at sc.nTriOpaque
	int useOrderB;             // prim = file-order rank (passes that draw every mesh in file order)
	int alphaTest;             // the pass's fragment shader discards on alpha < 0.5
	int shadow;                // depth bias of the shadow pipeline
	double guard;              // guard band, vkr_guard_band(view.W, view.H)
	SubTri* recs;
	uint2* recTiles;
	uint32_t recCap;
	uint32_t* recCount;   // records of clipped triangles, stored from slot `count` on
	uint32_t* tileCount;    // entries of records touching <= 8 tiles, no alpha test
	uint32_t* tileCountA;   // the same, alpha-tested records
	uint32_t* tileCountBig; // entries of records touching more
	uint32_t* recLocal;     // 8 per record slot: index inside each touched tile's bin (records touching <= 8 tiles)
	uint32_t* overflow;
	TriPlanes* planes;    // per triangle (order A index): interpolation + depth planes, for the resolve and alpha passes
	uint32_t* bigQueue;   // record slots that touch more than 8 tiles (capacity: recCap)
	uint32_t* bigCount;   // zeroed by the caller
	uint32_t* clipQueue;  // triangles that need the polygon clipper (capacity: count)
	uint32_t* clipCount;  // zeroed by the caller
};

struct BinFillArgs {
	const SubTri* recs;
	const uint2* recTiles;
	const uint32_t* recCount; // records appended after the one-per-triangle slots
	uint32_t baseCount;       // triangles of the pass = one-per-triangle slots
	uint32_t recCap;
	const uint32_t* tileOffset;
	const uint32_t* tileCount; // entries per tile fixed by the counting pass (no alpha test / alpha-tested)
	const uint32_t* tileCountA;
	const uint32_t* recLocal;
	const uint32_t* bigQueue; // record slots that touch more than 8 tiles
	const uint32_t* bigCount;
	uint32_t* cursor;
	uint32_t* binList;
	uint32_t* keys; // optional: prim per entry (ordered passes)
	uint32_t binCap;
	uint32_t tilesX;
};

enum { RASTER_VIS = 0, RASTER_SHADOW = 1, RASTER_COUNT = 2 };

struct RasterArgs {
	SceneDev sc;
	ViewDev view;
	const float4* clip;
	const TriPlanes* planes;
	const SubTri* recs;
	const uint32_t* tileOffset;
	const uint32_t* tileCount;  // a tile's bin holds tileCount entries without alpha test, then tileCountA alpha-tested
	const uint32_t* tileCountA; // ones, then the records that touch more than 8 tiles (either kind)
	const uint32_t* binList;
	uint32_t binCap;
	unsigned long long* vis; // RASTER_VIS: rect-sized visibility buffer
	float* depthOut;         // RASTER_SHADOW: rect-sized depth
	uint8_t* countOut;       // RASTER_COUNT: rect-sized saturating counter
	uint32_t* alphaTiles;    // RASTER_VIS: per tile, 1 if its bin holds alpha-tested sub-triangles (written for every tile)
	uint32_t* alphaTileCount; // RASTER_VIS: number of tiles flagged in alphaTiles (zeroed with the pass counters)
	uint32_t* deepTiles;     // RASTER_VIS: the tiles with many alpha-tested entries (capacity: tiles)
	uint32_t* deepTileCount; // its length (zeroed with the pass counters)
	int alphaPass;           // RASTER_VIS: run the alpha-tested second pass
	const uint32_t* tileOrder; // tiles by descending bin length (k_scan_tiles), or NULL for raster order
};

struct OvershadeArgs {
	ViewDev view;
	const SubTri* recs;
	const uint32_t* tileOffset;
	uint32_t* binList;       // the tiles' bins and, per entry, its primitive: the kernel orders a tile's entries by
	uint32_t* binKeys;       // primitive itself (bins too long for its LDS array are sorted in place)
	uint32_t binCap;
	uint8_t* countOut;
	float* depthOut;
	const uint32_t* tileOrder; // tiles by descending bin length (k_scan_tiles), or NULL for raster order
};

struct ResolveArgs {
	SceneDev sc;
	ViewDev view;
	const float4* clip;
	const TriPlanes* planes; // written by k_setup of the same pass
	const unsigned long long* vis;
	// deferred
	uint2* gN; uint2* gA;
	// shared
	float* depth; uint32_t* triId; uint8_t* mip; uint32_t* colour;
	// forward
	const float* shadow; uint32_t shadowSize; Mat4 lightMat;
	float lightPos[4], lightColour[4], camPos[4];
	int debugMode; int useOrderB;
};

struct LightArgs {
	ViewDev view;
	const uint2* gN; const uint2* gA; const float* depth;
	uint32_t* colour;
	const float* srgbThresh;
	Mat4 invPC;          // inverse(camera) * inverse(projection)
	float camPos[4];
	float lightPos[VKR_NUM_LIGHTS_K][4];
	float lightCol[VKR_NUM_LIGHTS_K][4];
	uint32_t clearColour; // packed RGBA8 of the render pass clear value
};

void vkr_launch_setup(hipStream_t s, const SetupArgs& a);
void vkr_launch_scan_tiles(hipStream_t s, const uint32_t* count, const uint32_t* countA, const uint32_t* countBig, uint32_t* offset, uint32_t* cursor,
                           uint32_t nTiles, uint32_t cap, uint32_t* overflow, uint32_t* statusOut, uint32_t* tileOrder = nullptr);
void vkr_launch_bin_fill(hipStream_t s, const BinFillArgs& a, uint32_t maxRecs);
void vkr_launch_raster(hipStream_t s, int mode, const RasterArgs& a);
void vkr_launch_overshade(hipStream_t s, const OvershadeArgs& a);
void vkr_launch_resolve_gbuf(hipStream_t s, const ResolveArgs& a);
void vkr_launch_resolve_forward(hipStream_t s, const ResolveArgs& a);
void vkr_launch_resolve_debug(hipStream_t s, const ResolveArgs& a);
void vkr_launch_deferred_light(hipStream_t s, const LightArgs& a);
void vkr_launch_resolve_light(hipStream_t s, const ResolveArgs& a, const LightArgs& l); // gN/gA/depth/triId of `a` optional
void vkr_launch_overvis_colour(hipStream_t s, const uint8_t* stencil, uint32_t* colour, const float* srgbThresh, uint32_t n);
void vkr_launch_mosaic(hipStream_t s, const uint32_t* src, uint32_t srcX0, uint32_t srcY0, uint32_t srcW, uint32_t srcH,
                       uint32_t* dst, ViewDev view, const float* srgbDecode, const float* srgbThresh);
// scene ingest, batched: one launch per mip level over every texture that has it, one launch for all spans
struct MipJob { uint32_t srcOff, dstOff; uint16_t sw, sh, dw, dh; int32_t srgb; };
struct SpanJob { uint32_t rgb, a, out, n; }; // texel offsets of the sources (`rgb` unused by extract_alpha), destination offset, count
void vkr_launch_mip_levels(hipStream_t s, uint32_t* texels, const MipJob* jobs, uint32_t count, uint32_t maxW, uint32_t maxH,
                           const float* srgbDecode, const float* srgbThresh);
void vkr_launch_extract_alpha(hipStream_t s, const uint32_t* texels, uint8_t* out, const SpanJob* jobs, uint32_t count, uint32_t maxN);
void vkr_launch_alpha_aux(hipStream_t s, const SceneDev& sc, AlphaAux* out);
void vkr_launch_tri_shade(hipStream_t s, const SceneDev& sc, TriShade* out);
void vkr_launch_pack_chains(hipStream_t s, uint32_t* texels, const SpanJob* jobs, uint32_t count, uint32_t maxN);
void vkr_launch_tbn_table(hipStream_t s, const uint32_t* tcomp, float4* out, uint32_t nVert);
void vkr_launch_fill_u32(hipStream_t s, uint32_t* dst, uint32_t value, size_t n);
