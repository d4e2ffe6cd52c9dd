// host-only robustness check: mutated .comp5892mesh files through vkr_scene_load's parser under ASan / UBSan
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
struct vkr_context; struct vkr_scene; struct vkr_scene_desc;
extern "C" int vkr_scene_load(vkr_context*, const char*, vkr_scene**);
int vkr_set_error_ext(int code, const char* fmt, ...) { (void)fmt; return code; }
static long created = 0;
extern "C" int vkr_scene_create(vkr_context*, const vkr_scene_desc*, vkr_scene**) { created++; return -1; }
static std::vector<uint8_t> slurp(const char* p) { std::vector<uint8_t> v; FILE* f = fopen(p, "rb"); if (!f) return v; int c; while ((c = fgetc(f)) != EOF) v.push_back((uint8_t)c); fclose(f); return v; }
// usage: fuzz_scene_file <good .comp5892mesh> <path of the mutated copy, beside the textures the file names>
int main(int argc, char** argv)
{
	if (argc < 3) return 2;
	std::mt19937 rng(4242);
	std::vector<uint8_t> good = slurp(argv[1]);
	long n = 0;
	for (int it = 0; it < 2000; it++) {
		std::vector<uint8_t> m = good;
		const int nmut = 1 + (int)(rng() % 3);
		for (int k = 0; k < nmut; k++) {
			const size_t at = (rng() % 3) ? rng() % 600 % m.size() : rng() % m.size();   // mostly the header region
			switch (rng() % 4) {
			case 0: m[at] ^= (uint8_t)(1u << (rng() % 8)); break;
			case 1: m[at] = (uint8_t)rng(); break;
			case 2: m.resize(at + 1); break;
			default: { uint32_t v = (rng() % 2) ? 0xffffffffu : (uint32_t)rng(); if (at + 4 <= m.size()) memcpy(&m[at], &v, 4); break; }
			}
		}
		FILE* f = fopen(argv[2], "wb"); fwrite(m.data(), 1, m.size(), f); fclose(f);
		vkr_scene* s = nullptr;
		vkr_scene_load((vkr_context*)0x1000, argv[2], &s);
		n++;
	}
	printf("%ld files parsed, %ld reached scene creation\n", n, created);
	return 0;
}
