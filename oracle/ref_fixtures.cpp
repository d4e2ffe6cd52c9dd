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
//        ref_fixtures stbsha <png>...     -> the same for large files: size, channel count and SHA-256 of the decoded RGBA8
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

// SHA-256 (FIPS 180-4) of a byte buffer, as 64 hex digits
static void sha256_hex(const unsigned char* data, std::size_t len, char out[65])
{
	static const std::uint32_t K[64] = {
		0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
		0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
		0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
		0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
		0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
		0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2 };
	std::uint32_t h[8] = { 0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19 };
	auto rotr = [](std::uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
	auto block = [&](const unsigned char* p) {
		std::uint32_t w[64];
		for (int i = 0; i < 16; i++) w[i] = (std::uint32_t(p[4 * i]) << 24) | (std::uint32_t(p[4 * i + 1]) << 16) | (std::uint32_t(p[4 * i + 2]) << 8) | p[4 * i + 3];
		for (int i = 16; i < 64; i++) {
			const std::uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
			w[i] = w[i - 16] + s0 + w[i - 7] + s1;
		}
		std::uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
		for (int i = 0; i < 64; i++) {
			const std::uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
			const std::uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
			hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
		}
		h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
	};
	std::size_t i = 0;
	for (; i + 64 <= len; i += 64) block(data + i);
	unsigned char tail[128] = { 0 };
	const std::size_t rest = len - i;
	std::memcpy(tail, data + i, rest);
	tail[rest] = 0x80;
	const std::size_t padded = rest + 9 <= 64 ? 64 : 128;
	const unsigned long long bits = (unsigned long long)len * 8;
	for (int k = 0; k < 8; k++) tail[padded - 1 - k] = (unsigned char)(bits >> (8 * k));
	block(tail);
	if (padded == 128) block(tail + 64);
	for (int k = 0; k < 8; k++) std::snprintf(out + 8 * k, 9, "%08x", h[k]);
}

static int do_stbsha(int n, char** files)
{
	stbi_set_flip_vertically_on_load(1);
	std::printf("{\n");
	for (int i = 0; i < n; i++) {
		int w = 0, h = 0, c = 0;
		stbi_uc* data = stbi_load(files[i], &w, &h, &c, 4);
		const char* base = std::strrchr(files[i], '/');
		base = base ? base + 1 : files[i];
		if (!data) { std::printf("  \"%s\": null%s\n", base, i + 1 == n ? "" : ","); continue; }
		char hex[65];
		sha256_hex(data, (std::size_t)w * h * 4, hex);
		std::printf("  \"%s\": {\"width\": %d, \"height\": %d, \"file_channels\": %d, \"sha256\": \"%s\"}%s\n", base, w, h, c, hex, i + 1 == n ? "" : ",");
		stbi_image_free(data);
	}
	std::printf("}\n");
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
	if (argc >= 3 && 0 == std::strcmp(argv[1], "stbsha")) return do_stbsha(argc - 2, argv + 2);
	std::fprintf(stderr, "usage: ref_fixtures matrices|tangents|userstate|loadmodel <file>|stb <png>...|stbsha <png>...\n");
	return 2;
} catch (std::exception const& e) {
	std::fprintf(stderr, "error: %s\n", e.what());
	return 1;
}
