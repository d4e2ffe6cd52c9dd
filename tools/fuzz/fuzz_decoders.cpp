// host-only robustness check: mutated Zstandard frames and PNG files through the decoders under ASan / UBSan
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "vkr_zstd.h"
extern "C" int vkr_decode_png_file(const char* path, uint8_t** rgba8, uint32_t* w, uint32_t* h);
extern "C" void vkr_free(void*);
static std::vector<uint8_t> slurp(const char* p) { std::vector<uint8_t> v; FILE* f = fopen(p, "rb"); if (!f) return v; int c; while ((c = fgetc(f)) != EOF) v.push_back((uint8_t)c); fclose(f); return v; }
// usage: fuzz_decoders <scratch dir> <file>...
int main(int argc, char** argv)
{
	std::mt19937 rng(12345);
	long ok = 0, bad = 0;
	if (argc < 3) return 2;
	const std::string tmp = std::string(argv[1]) + "/m.png";
	for (int a = 2; a < argc; a++) {
		std::vector<uint8_t> good = slurp(argv[a]);
		if (good.empty()) continue;
		const bool png = strstr(argv[a], ".png") != nullptr;
		const int iters = png ? 600 : 150;
		for (int it = 0; it < iters; it++) {
			std::vector<uint8_t> m = good;
			const int nmut = 1 + (int)(rng() % 4);
			for (int k = 0; k < nmut; k++) {
				const size_t at = rng() % m.size();
				switch (rng() % 4) {
				case 0: m[at] ^= (uint8_t)(1u << (rng() % 8)); break;
				case 1: m[at] = (uint8_t)rng(); break;
				case 2: m.resize(at + 1); break;
				default: m.insert(m.begin() + at, (uint8_t)rng()); break;
				}
			}
			if (png) {
				FILE* f = fopen(tmp.c_str(), "wb"); fwrite(m.data(), 1, m.size(), f); fclose(f);
				uint8_t* px = nullptr; uint32_t w = 0, h = 0;
				if (vkr_decode_png_file(tmp.c_str(), &px, &w, &h) == 0) { ok++; volatile uint8_t s = 0; for (size_t i = 0; i < (size_t)w * h * 4; i += 97) s ^= px[i]; vkr_free(px); } else bad++;
			} else {
				std::vector<uint8_t> out; std::string err;
				if (vkr_zstd_decompress(m.data(), m.size(), out, err)) ok++; else bad++;
			}
		}
	}
	printf("decoded %ld, rejected %ld\n", ok, bad);
	return 0;
}
// stand-ins for the two symbols of the HIP half that the loader file refers to (never reached by this driver)
int vkr_set_error_ext(int code, const char* fmt, ...) { (void)fmt; return code; }
struct vkr_context; struct vkr_scene; struct vkr_scene_desc;
extern "C" int vkr_scene_create(vkr_context*, const vkr_scene_desc*, vkr_scene**) { return -1; }
