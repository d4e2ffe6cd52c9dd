/* vkr_bake.h — C ABI of libvkr_bake.so: the asset bake step that produces the
 * ".comp5892mesh" ("21-tan") files vkr_scene_load() reads.
 *
 * Host-only (no GPU): it stands where the reference's main-bake tool stands
 *   main-bake/main.cpp:142-231        process_model_        -> vkr_bake_obj
 *   main-bake/load_model_obj.cpp      OBJ/MTL -> triangle soup per (shape, material)
 *   main-bake/index_mesh.cpp:76-228   make_indexed_mesh     -> vkr_index_mesh
 *   third_party/tgen/src/tgen.cpp     tangent generation (inside vkr_index_mesh)
 * and writes the file layout of main-bake/main.cpp:276-389.
 *
 * The input stage follows the reference tool's: the OBJ may be a Zstandard
 * stream (main-bake/zstdistream.cpp; decoded by the library's own decoder) or
 * plain text; quads are split along their shorter diagonal and larger polygons by
 * ear clipping on the plane of largest projected area, as rapidobj::Triangulate
 * does (load_model_obj.cpp:33); the meshes of a shape that uses several materials
 * come out in the iteration order of a std::unordered_set<std::size_t>
 * (load_model_obj.cpp:94-120), which is the reference's order when both are built
 * against the same C++ standard library.
 */
#ifndef VKR_BAKE_H
#define VKR_BAKE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(VKR_BAKE_BUILD)
#define VKR_BAKE_API __attribute__((visibility("default")))
#else
#define VKR_BAKE_API
#endif

/* Result of vkr_index_mesh; every array is owned by the library until
 * vkr_indexed_mesh_free(). */
typedef struct vkr_indexed_mesh {
	uint32_t vertexCount, indexCount;
	float* positions;       /* 3 per vertex */
	float* normals;         /* 3 per vertex */
	float* texcoords;       /* 2 per vertex */
	float* tangents;        /* 4 per vertex: tangent + handedness */
	uint32_t* tangentsComp; /* packed TBN quaternion, 2:10:10:10 */
	uint32_t* indices;
} vkr_indexed_mesh;

/* make_indexed_mesh (main-bake/index_mesh.cpp:76-228): welds the soup's
 * vertices whose position, normal and texture coordinate all lie within
 * `tolerance` of an earlier vertex (component-wise), keeps triangle order,
 * generates tangents (tgen) and packs the TBN frame.  soupVertexCount must be a
 * multiple of 3.  Returns 0, or non-zero with vkr_bake_last_error() set. */
VKR_BAKE_API int vkr_index_mesh(const float* positions, const float* normals, const float* texcoords,
                                uint32_t soupVertexCount, float tolerance, vkr_indexed_mesh* out);
VKR_BAKE_API void vkr_indexed_mesh_free(vkr_indexed_mesh* mesh);

/* process_model_ (main-bake/main.cpp:142-231) for a text OBJ with its MTL:
 * loads, substitutes the fallback textures (main.cpp:236-251), indexes every
 * mesh with tolerance 1e-5 (main.cpp:75-78), collects the unique textures
 * (main.cpp:426-457), writes `out_path` and - if copy_textures - copies the
 * texture files into "<out stem>-tex/" next to it (existing files are kept). */
VKR_BAKE_API int vkr_bake_obj(const char* obj_path, const char* out_path, int copy_textures);

/* Decodes a buffer of Zstandard frames (RFC 8878; what main-bake/zstdistream.cpp:35-117 feeds to rapidobj) with the
 * library's own decoder.  *out is allocated by the library; release it with vkr_bake_free(). */
VKR_BAKE_API int vkr_zstd_decode(const uint8_t* src, size_t len, uint8_t** out, size_t* out_len);
VKR_BAKE_API void vkr_bake_free(void* p);

VKR_BAKE_API const char* vkr_bake_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* VKR_BAKE_H */
