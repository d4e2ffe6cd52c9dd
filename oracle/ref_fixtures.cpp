// ref_fixtures.cpp — TEST INFRASTRUCTURE.  Generates golden vectors by calling
// the parts of the reference that compile without Vulkan: glm (vendored under
// third_party/glm), labutils' angle helpers (utils/angle.hpp), the bake tool's
// make_indexed_mesh (main-bake/index_mesh.cpp + third_party/tgen) and the
// baked-model loader (main/baked_model.cpp).  It is compiled against the
// sources where they lie under /root/reference (see oracle/Makefile, target
// _ref) and its outputs are committed under tests/golden/.  Nothing here is
// copied from the reference; the reference cannot travel to the GPU box.
//
// usage: ref_fixtures matrices            -> JSON on stdout
//        ref_fixtures tangents            -> JSON on stdout
//        ref_fixtures loadmodel <file>    -> JSON on stdout
//        ref_fixtures stb <png>...        -> JSON on stdout (stbi_load(...,4) flipped, as utils/vkimage.cpp:62-65)
//        ref_fixtures userstate           -> JSON on stdout (camera2world after each step of an input trace)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtx/transform.hpp>

#include <stb_image.h>
#include <fstream>
#include <filesystem>

#include "../utils/angle.hpp"
#include "../main-bake/index_mesh.hpp"
#include "../main/baked_model.hpp"

namespace lut = labutils;
using namespace labutils::literals;

static void print_mat(const char* name, glm::mat4 const& m, bool last = false)
{
	std::printf("    \"%s\": [", name);
	for (int c = 0; c < 4; c++)
		for (int r = 0; r < 4; r++) {
			std::uint32_t bits;
			float v = m[c][r];
			std::memcpy(&bits, &v, 4);
			std::printf("%u%s", bits, (c == 3 && r == 3) ? "" : ", ");
		}
	std::printf("]%s\n", last ? "" : ",");
}

static int do_matrices()
{
	// The same glm calls, with the same arguments, as update_scene_uniforms
	// (main/main.cpp:3681-3693) and update_depth_mvp_uniforms (main/main.cpp:3699-3707).
	constexpr auto kCameraFov = 60.0_degf;
	struct Pose { const char* name; glm::mat4 c2w; };
	std::vector<Pose> poses;
	poses.push_back({ "identity", glm::identity<glm::mat4>() });
	poses.push_back({ "p1", glm::translate(glm::vec3(0.f, -1.f, -30.f)) });
	poses.push_back({ "p2", glm::translate(glm::vec3(0.f, -2.f, -70.f)) * glm::rotate(3.14159265358979f, glm::vec3(0.f, 1.f, 0.f)) });
	poses.push_back({ "p3", glm::translate(glm::vec3(1.5f, 0.25f, -12.f)) * glm::rotate(0.35f, glm::vec3(0.f, 1.f, 0.f)) * glm::rotate(-0.2f, glm::vec3(1.f, 0.f, 0.f)) });
	const unsigned sizes[][2] = { { 1280, 720 }, { 1920, 1080 }, { 3840, 2160 }, { 64, 36 } };
	std::printf("{\n  \"scene\": [\n");
	bool first = true;
	for (auto const& sz : sizes)
		for (auto const& p : poses) {
			const float aspect = sz[0] / float(sz[1]);
			glm::mat4 projection = glm::perspectiveRH_ZO(lut::Radians(kCameraFov).value(), aspect, 0.1f, 100.f);
			projection[1][1] *= -1.0f;
			glm::mat4 camera = glm::inverse(p.c2w);
			glm::mat4 projCam = projection * camera;
			std::printf("%s  {\n    \"pose\": \"%s\", \"width\": %u, \"height\": %u,\n", first ? "" : ",\n", p.name, sz[0], sz[1]);
			print_mat("camera2world", p.c2w);
			print_mat("camera", camera);
			print_mat("projection", projection);
			print_mat("projCam", projCam);
			print_mat("invCamera", glm::inverse(camera));
			print_mat("invProjection", glm::inverse(projection), true);
			std::printf("  }");
			first = false;
		}
	std::printf("\n  ],\n");
	{
		glm::vec4 lightPos(-0.2972f, 7.3100f, -11.9532f, 0.0f);
		glm::mat4 depthProjection = glm::perspectiveRH_ZO(lut::Radians(90.0f).value(), 1.0f, 0.1f, 100.f);
		depthProjection[1][1] *= -1.0f;
		glm::mat4 depthView = glm::lookAt(glm::vec3(lightPos), glm::vec3(0.0f, 0.0f, -48.0f), glm::vec3(0.0f, 1.0f, 0.0f));
		std::printf("  \"depth\": {\n");
		print_mat("depthProjection", depthProjection);
		print_mat("depthView", depthView);
		print_mat("depthMVP", depthProjection * depthView, true);
		std::printf("  }\n}\n");
	}
	return 0;
}

static int do_tangents()
{
	// A small deterministic triangle soup through make_indexed_mesh
	// (main-bake/index_mesh.cpp:76-228): a bumpy 6x5 grid with varied normals
	// and a UV mapping that is rotated and mirrored on one half.
	TriangleSoup soup;
	auto P = [](int i, int j) {
		float x = 0.37f * i, z = 0.29f * j;
		float y = 0.15f * float((i * 7 + j * 3) % 5) + 0.05f * i * j;
		return glm::vec3(x, y, z);
	};
	auto N = [](int i, int j) {
		glm::vec3 n(0.3f * float((i * 5 + j) % 3 - 1), 1.0f, 0.25f * float((j * 2 + i) % 4 - 2));
		return glm::normalize(n);
	};
	auto T = [](int i, int j) {
		float u = 0.21f * i + 0.05f * j, v = 0.17f * j - 0.03f * i;
		if (i >= 3) u = 2.0f - u; // mirrored half
		return glm::vec2(u, v);
	};
	for (int j = 0; j < 4; j++)
		for (int i = 0; i < 5; i++) {
			int q[6][2] = { { i, j }, { i + 1, j }, { i + 1, j + 1 }, { i, j }, { i + 1, j + 1 }, { i, j + 1 } };
			for (auto& c : q) {
				soup.vert.push_back(P(c[0], c[1]));
				soup.norm.push_back(N(c[0], c[1]));
				soup.text.push_back(T(c[0], c[1]));
			}
		}
	IndexedMesh m = make_indexed_mesh(soup);
	std::printf("{\n  \"vertexCount\": %zu, \"indexCount\": %zu,\n  \"vertices\": [\n", m.vert.size(), m.indices.size());
	for (std::size_t i = 0; i < m.vert.size(); i++) {
		auto bits = [](float f) { std::uint32_t b; std::memcpy(&b, &f, 4); return b; };
		std::printf("    {\"normal\": [%u, %u, %u], \"tangent\": [%u, %u, %u, %u], \"packed\": %u}%s\n",
			bits(m.norm[i].x), bits(m.norm[i].y), bits(m.norm[i].z),
			bits(m.tangent[i].x), bits(m.tangent[i].y), bits(m.tangent[i].z), bits(m.tangent[i].w),
			m.tangentComp[i], i + 1 == m.vert.size() ? "" : ",");
	}
	std::printf("  ]\n}\n");
	return 0;
}

static std::uint64_t fnv1a(const void* p, std::size_t n, std::uint64_t h = 1469598103934665603ull)
{
	auto b = static_cast<const unsigned char*>(p);
	for (std::size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
	return h;
}

static int do_loadmodel(const char* path)
{
	BakedModel m = load_baked_model(path);
	std::printf("{\n  \"textures\": [\n");
	for (std::size_t i = 0; i < m.textures.size(); i++)
		std::printf("    {\"path\": \"%s\", \"space\": %u, \"channels\": %u}%s\n", m.textures[i].path.c_str(),
			unsigned(m.textures[i].space), unsigned(m.textures[i].channels), i + 1 == m.textures.size() ? "" : ",");
	std::printf("  ],\n  \"materials\": [\n");
	for (std::size_t i = 0; i < m.materials.size(); i++) {
		auto const& k = m.materials[i];
		std::printf("    [%u, %u, %u, %u, %u, %u]%s\n", k.baseColorTextureId, k.roughnessTextureId, k.metalnessTextureId,
			k.alphaMaskTextureId, k.normalMapTextureId, k.emissiveTextureId, i + 1 == m.materials.size() ? "" : ",");
	}
	std::printf("  ],\n  \"meshes\": [\n");
	for (std::size_t i = 0; i < m.meshes.size(); i++) {
		auto const& d = m.meshes[i];
		std::printf("    {\"materialId\": %u, \"V\": %zu, \"I\": %zu, \"positions\": \"%016llx\", \"normals\": \"%016llx\", "
		            "\"texcoords\": \"%016llx\", \"tangents\": \"%016llx\", \"tangentsComp\": \"%016llx\", \"indices\": \"%016llx\"}%s\n",
			d.materialId, d.positions.size(), d.indices.size(),
			(unsigned long long)fnv1a(d.positions.data(), d.positions.size() * sizeof(glm::vec3)),
			(unsigned long long)fnv1a(d.normals.data(), d.normals.size() * sizeof(glm::vec3)),
			(unsigned long long)fnv1a(d.texcoords.data(), d.texcoords.size() * sizeof(glm::vec2)),
			(unsigned long long)fnv1a(d.tangents.data(), d.tangents.size() * sizeof(glm::vec4)),
			(unsigned long long)fnv1a(d.tangentsComp.data(), d.tangentsComp.size() * sizeof(std::uint32_t)),
			(unsigned long long)fnv1a(d.indices.data(), d.indices.size() * sizeof(std::uint32_t)),
			i + 1 == m.meshes.size() ? "" : ",");
	}
	std::printf("  ]\n}\n");
	return 0;
}

static int do_stb(int n, char** files)
{
	stbi_set_flip_vertically_on_load(1);
	std::printf("{\n");
	for (int i = 0; i < n; i++) {
		int w = 0, h = 0, c = 0;
		stbi_uc* data = stbi_load(files[i], &w, &h, &c, 4);
		const char* base = std::strrchr(files[i], '/');
		base = base ? base + 1 : files[i];
		if (!data) { std::printf("  \"%s\": null%s\n", base, i + 1 == n ? "" : ","); continue; }
		std::printf("  \"%s\": {\"width\": %d, \"height\": %d, \"file_channels\": %d, \"fnv1a\": \"%016llx\", \"first\": [", base, w, h, c,
			(unsigned long long)fnv1a(data, (std::size_t)w * h * 4));
		int m = w * h * 4 < 32 ? w * h * 4 : 32;
		for (int k = 0; k < m; k++) std::printf("%u%s", data[k], k + 1 == m ? "" : ", ");
		std::printf("]}%s\n", i + 1 == n ? "" : ",");
		stbi_image_free(data);
	}
	std::printf("}\n");
	return 0;
}

// The input trace the user-state fixture is made of: per step the held keys
// (bit i = EInputState i), the cursor position and the frame time.
struct TraceStep { unsigned keys; float mouseX, mouseY, dt; };
static const TraceStep kTrace[] = {
	{ 0x001, 0.f, 0.f, 0.016f },               // forward
	{ 0x041, 0.f, 0.f, 0.0166667f },           // forward, fast
	{ 0x100, 400.f, 300.f, 0.0166667f },       // mousing starts: no rotation yet
	{ 0x100, 437.f, 282.f, 0.0166667f },       // look
	{ 0x108, 431.5f, 310.25f, 0.02f },         // look + strafe right
	{ 0x000, 900.f, 10.f, 0.01f },             // released: cursor moves freely
	{ 0x100, 120.f, 620.f, 0.01f },            // mousing again: first step only latches
	{ 0x185, 119.f, 640.f, 0.033f },           // look + forward + strafe left + slow
	{ 0x0c2, 0.f, 0.f, 0.25f },                // backward, fast and slow together
	{ 0x030, 0.f, 0.f, 0.0166667f },           // levitate and sink cancel (up to rounding)
	{ 0x010, 0.f, 0.f, 1.5f },                 // levitate, long frame
	{ 0x120, -250.f, 75.f, 0.0166667f },       // sink + mousing latch
	{ 0x13f, 1650.f, -830.f, 0.0166667f },     // every movement key, big mouse delta
	{ 0x101, 1650.f, -830.f, 0.0f },           // zero frame time, zero mouse delta
};

static int do_userstate()
{
	// The statements of update_user_state (main/main.cpp:3643-3679) on glm's own
	// types, with the constants of main/main.cpp:76-84.
	constexpr float kCameraBaseSpeed = 1.7f, kCameraFastMult = 7.0f, kCameraSlowMult = 0.05f, kCameraMouseSensitivity = 0.01f;
	glm::mat4 cam = glm::identity<glm::mat4>();
	float previousX = 0.f, previousY = 0.f;
	bool wasMousing = false;
	std::printf("{\n  \"steps\": [\n");
	const std::size_t n = sizeof(kTrace) / sizeof(kTrace[0]);
	for (std::size_t i = 0; i < n; i++) {
		TraceStep const& t = kTrace[i];
		auto held = [&](int k) { return ((t.keys >> k) & 1u) != 0; };
		if (held(8)) {
			if (wasMousing) {
				const auto sens = kCameraMouseSensitivity;
				const auto dx = sens * (t.mouseX - previousX);
				const auto dy = sens * (t.mouseY - previousY);
				cam = cam * glm::rotate(-dy, glm::vec3(1.0f, 0.0f, 0.0f));
				cam = cam * glm::rotate(-dx, glm::vec3(0.0f, 1.0f, 0.0f));
			}
			previousX = t.mouseX;
			previousY = t.mouseY;
			wasMousing = true;
		} else {
			wasMousing = false;
		}
		const auto move = t.dt * kCameraBaseSpeed * (held(6) ? kCameraFastMult : 1.0f) * (held(7) ? kCameraSlowMult : 1.0f);
		if (held(0)) cam = cam * glm::translate(glm::vec3(0.f, 0.f, -move));
		if (held(1)) cam = cam * glm::translate(glm::vec3(0.f, 0.f, +move));
		if (held(2)) cam = cam * glm::translate(glm::vec3(-move, 0.f, 0.f));
		if (held(3)) cam = cam * glm::translate(glm::vec3(+move, 0.f, 0.f));
		if (held(4)) cam = cam * glm::translate(glm::vec3(0.f, +move, 0.f));
		if (held(5)) cam = cam * glm::translate(glm::vec3(0.f, -move, 0.f));
		std::uint32_t mx, my, dt;
		std::memcpy(&mx, &t.mouseX, 4); std::memcpy(&my, &t.mouseY, 4); std::memcpy(&dt, &t.dt, 4);
		std::printf("  {\n    \"keys\": %u, \"mouseX\": %u, \"mouseY\": %u, \"dt\": %u, \"wasMousing\": %d,\n", t.keys, mx, my, dt, wasMousing ? 1 : 0);
		print_mat("camera2world", cam, true);
		std::printf("  }%s\n", i + 1 < n ? "," : "");
	}
	std::printf("  ]\n}\n");
	return 0;
}

int main(int argc, char** argv)
try {
	if (argc >= 2 && 0 == std::strcmp(argv[1], "matrices")) return do_matrices();
	if (argc >= 2 && 0 == std::strcmp(argv[1], "tangents")) return do_tangents();
	if (argc >= 2 && 0 == std::strcmp(argv[1], "userstate")) return do_userstate();
	if (argc >= 3 && 0 == std::strcmp(argv[1], "loadmodel")) return do_loadmodel(argv[2]);
	if (argc >= 3 && 0 == std::strcmp(argv[1], "stb")) return do_stb(argc - 2, argv + 2);
	std::fprintf(stderr, "usage: ref_fixtures matrices|tangents|userstate|loadmodel <file>|stb <png>...\n");
	return 2;
} catch (std::exception const& e) {
	std::fprintf(stderr, "error: %s\n", e.what());
	return 1;
}
