// host-only robustness check: mutated OBJ / MTL text through the bake step under ASan / UBSan
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "vkr_bake.h"
static std::vector<uint8_t> slurp(const char* p) { std::vector<uint8_t> v; FILE* f = fopen(p, "rb"); if (!f) return v; int c; while ((c = fgetc(f)) != EOF) v.push_back((uint8_t)c); fclose(f); return v; }
static void spit(const char* p, const std::vector<uint8_t>& v) { FILE* f = fopen(p, "wb"); fwrite(v.data(), 1, v.size(), f); fclose(f); }
// usage: fuzz_bake <scratch dir> <model.obj> <model.mtl>
int main(int argc, char** argv)
{
	std::mt19937 rng(777);
	if (argc < 4) return 2;
	const std::string dir = argv[1];
	const std::string fo = dir + "/model.obj", fm = dir + "/model.mtl", out = dir + "/out.comp5892mesh";
	std::vector<uint8_t> obj = slurp(argv[2]), mtl = slurp(argv[3]);
	long ok = 0, bad = 0;
	const char* tokens[] = { "f ", "v ", "vt ", "vn ", "usemtl ", "g ", "o ", "mtllib ", "-1", "0", "99999999", "1e39", "nan", "//", "/", "\n", "  ", "#", "1/2/3 ", "-3/-2/-1 " };
	for (int it = 0; it < 1500; it++) {
		std::vector<uint8_t> o = obj, m = mtl;
		std::vector<uint8_t>& t = (rng() % 4) ? o : m;
		const int nmut = 1 + (int)(rng() % 5);
		for (int k = 0; k < nmut; k++) {
			const size_t at = rng() % t.size();
			switch (rng() % 5) {
			case 0: t[at] = (uint8_t)(32 + rng() % 95); break;
			case 1: t.erase(t.begin() + at, t.begin() + std::min(t.size(), at + 1 + rng() % 40)); break;
			case 2: { const char* s = tokens[rng() % (sizeof tokens / sizeof *tokens)]; t.insert(t.begin() + at, s, s + strlen(s)); break; }
			case 3: t.resize(at + 1); break;
			default: t[at] = (uint8_t)rng(); break;
			}
			if (t.empty()) t.push_back('\n');
		}
		spit(fo.c_str(), o); spit(fm.c_str(), m);
		if (vkr_bake_obj(fo.c_str(), out.c_str(), 0) == 0) ok++; else bad++;
	}
	printf("baked %ld, rejected %ld (last error: %s)\n", ok, bad, vkr_bake_last_error());
	return 0;
}
