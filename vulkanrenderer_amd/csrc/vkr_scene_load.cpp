// vkr_scene_load.cpp — reads a baked model file and the PNG textures it names,
// then hands the arrays to vkr_scene_create.
//
// File format: load_baked_model_ (main/baked_model.cpp:70-174), variant
// "21-tan".  Texture decode: what the reference gets from
// stbi_load(path, ..., 4) with stbi_set_flip_vertically_on_load(1)
// (utils/vkimage.cpp:62-65): 8-bit RGBA, grey replicated to RGB, missing alpha
// = 255, 16-bit samples reduced to their high byte, rows bottom-up.
#include "../../include/vkr_renderer.h"

#include <sched.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

int vkr_set_error_ext(int code, const char* fmt, ...); // vkr_api.cpp

namespace {

struct Reader {
	const uint8_t* p; size_t n, off = 0;
	bool take(void* dst, size_t k)
	{
		if (off + k > n) return false;
		memcpy(dst, p + off, k);
		off += k;
		return true;
	}
	bool u32(uint32_t& v) { return take(&v, 4); } // little-endian host
};

bool read_file(const std::string& path, std::vector<uint8_t>& out)
{
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	if (sz < 0) { fclose(f); return false; }
	out.resize((size_t)sz);
	size_t got = sz ? fread(out.data(), 1, (size_t)sz, f) : 0;
	fclose(f);
	return got == (size_t)sz;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
	int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	if (pa <= pb && pa <= pc) return a;
	if (pb <= pc) return b;
	return c;
}

// PNG (every colour type and bit depth, interlaced or not) -> RGBA8, flipped vertically, as stbi_load(path, ..., 4)
// delivers it after stbi_set_flip_vertically_on_load(1).  Returns an empty string on success.
std::string decode_png(const std::vector<uint8_t>& file, std::vector<uint8_t>& rgba, uint32_t& W, uint32_t& H)
{
	static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
	if (file.size() < 8 || memcmp(file.data(), sig, 8)) return "not a PNG file";
	size_t off = 8;
	uint32_t depth = 0, ctype = 0, interlace = 0;
	bool haveHdr = false;
	std::vector<uint8_t> idat, plte, trns;
	while (off + 12 <= file.size()) {
		uint32_t len = be32(&file[off]);
		const uint8_t* type = &file[off + 4];
		if (off + 12 + (size_t)len > file.size()) return "truncated chunk";
		const uint8_t* data = &file[off + 8];
		if (!memcmp(type, "IHDR", 4)) {
			if (len < 13) return "bad IHDR";
			W = be32(data); H = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
			haveHdr = true;
		} else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
		else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
		else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
		else if (!memcmp(type, "IEND", 4)) break;
		off += 12 + (size_t)len;
	}
	if (!haveHdr || W == 0 || H == 0) return "missing IHDR";
	if (interlace > 1) return "bad interlace method";
	int channels;
	switch (ctype) {
	case 0: channels = 1; break;
	case 2: channels = 3; break;
	case 3: channels = 1; break;
	case 4: channels = 2; break;
	case 6: channels = 4; break;
	default: return "bad colour type";
	}
	if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) return "bad bit depth";
	// depths below 8 exist for grey and palette images only, 16 not for palettes (PNG specification, table 11.1)
	if ((depth < 8 && ctype != 0 && ctype != 3) || (depth == 16 && ctype == 3)) return "bad bit depth for the colour type";
	if (W > (1u << 24) || H > (1u << 24)) return "image too large"; // stb_image's own limit (STBI_MAX_DIMENSIONS)
	const size_t bpp = (size_t)channels * depth;           // bits per pixel
	const size_t fbpp = (bpp + 7) / 8;                     // filter distance in bytes
	// The image as the scanlines are stored: in one piece, or as the seven reduced images of Adam7 interlacing (pass p
	// holds the pixels (x0 + i dx, y0 + j dy); empty passes are not stored), each filtered on its own.
	struct Pass { uint32_t x0, y0, dx, dy, w, h; size_t stride; };
	static const uint32_t kAdam7[7][4] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
	Pass passes[7];
	int nPasses = 0;
	size_t rawSize = 0;
	for (int p = 0; p < (interlace ? 7 : 1); p++) {
		Pass q;
		if (interlace) { q.x0 = kAdam7[p][0]; q.y0 = kAdam7[p][1]; q.dx = kAdam7[p][2]; q.dy = kAdam7[p][3]; }
		else { q.x0 = q.y0 = 0; q.dx = q.dy = 1; }
		q.w = (W - q.x0 + q.dx - 1) / q.dx; q.h = (H - q.y0 + q.dy - 1) / q.dy;
		if (W <= q.x0 || H <= q.y0 || !q.w || !q.h) continue;
		q.stride = ((size_t)q.w * bpp + 7) / 8;
		rawSize += (q.stride + 1) * (size_t)q.h;
		passes[nPasses++] = q;
	}
	// deflate expands at most about 1032 times: a header that promises more than the data can hold is refused before
	// anything of that size is allocated
	if (rawSize / 1100 > idat.size() + 16) return "image data too short for the image size";
	std::vector<uint8_t> raw(rawSize);
	uLongf rawLen = (uLongf)raw.size();
	if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return "zlib inflate failed";
	rgba.resize((size_t)W * H * 4);
	const uint32_t maxv = (1u << depth) - 1u;
	auto to8 = [&](uint32_t v) -> uint8_t {
		if (depth == 16) return (uint8_t)(v >> 8);
		if (depth == 8) return (uint8_t)v;
		return (uint8_t)(v * (255u / maxv));
	};
	std::vector<uint8_t> img;
	size_t rawAt = 0;
	for (int p = 0; p < nPasses; p++) {
		const Pass& q = passes[p];
		const size_t stride = q.stride;
		img.assign(stride * (size_t)q.h, 0);
		for (uint32_t y = 0; y < q.h; y++) {
			const uint8_t ft = raw[rawAt + (stride + 1) * y];
			const uint8_t* src = &raw[rawAt + (stride + 1) * y + 1];
			uint8_t* cur = &img[stride * y];
			const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
			if (ft > 4) return "bad filter type";
			for (size_t i = 0; i < stride; i++) {
				int a = i >= fbpp ? cur[i - fbpp] : 0, b = up ? up[i] : 0, c = (up && i >= fbpp) ? up[i - fbpp] : 0;
				int v = src[i];
				switch (ft) {
				case 0: break;
				case 1: v += a; break;
				case 2: v += b; break;
				case 3: v += (a + b) >> 1; break;
				default: v += paeth(a, b, c); break;
				}
				cur[i] = (uint8_t)v;
			}
		}
		rawAt += (stride + 1) * (size_t)q.h;
		for (uint32_t y = 0; y < q.h; y++) {
			const uint8_t* row = &img[stride * y];
			const uint32_t oy = q.y0 + y * q.dy;
			uint8_t* dst = &rgba[(size_t)(H - 1 - oy) * W * 4]; // vertical flip
			for (uint32_t x = 0; x < q.w; x++) {
				uint32_t s[4] = { 0, 0, 0, 0 }; // raw samples
				for (int ch = 0; ch < channels; ch++) {
					const size_t bit = ((size_t)x * channels + ch) * depth;
					if (depth == 8) s[ch] = row[bit >> 3];
					else if (depth == 16) s[ch] = ((uint32_t)row[bit >> 3] << 8) | row[(bit >> 3) + 1];
					else s[ch] = (row[bit >> 3] >> (8 - depth - (bit & 7))) & maxv;
				}
				uint8_t r, g, b, a = 255;
				if (ctype == 3) {
					const uint32_t idx = s[0];
					if ((size_t)idx * 3 + 2 < plte.size()) { r = plte[idx * 3]; g = plte[idx * 3 + 1]; b = plte[idx * 3 + 2]; }
					else { r = g = b = 0; }
					if (idx < trns.size()) a = trns[idx];
				} else if (ctype == 0 || ctype == 4) {
					r = g = b = to8(s[0]);
					if (ctype == 4) a = to8(s[1]);
					else if (trns.size() >= 2 && s[0] == (((uint32_t)trns[0] << 8) | trns[1])) a = 0;
				} else {
					r = to8(s[0]); g = to8(s[1]); b = to8(s[2]);
					if (ctype == 6) a = to8(s[3]);
					else if (trns.size() >= 6 && s[0] == (((uint32_t)trns[0] << 8) | trns[1]) &&
					         s[1] == (((uint32_t)trns[2] << 8) | trns[3]) && s[2] == (((uint32_t)trns[4] << 8) | trns[5])) a = 0;
				}
				const size_t ox = q.x0 + (size_t)x * q.dx;
				dst[4 * ox + 0] = r; dst[4 * ox + 1] = g; dst[4 * ox + 2] = b; dst[4 * ox + 3] = a;
			}
		}
	}
	return "";
}

} // namespace

static int decode_png_file_impl(const char* path, uint8_t** rgba8, uint32_t* width, uint32_t* height)
{
	if (!path || !rgba8 || !width || !height) return vkr_set_error_ext(VKR_E_INVALID, "vkr_decode_png_file: NULL argument");
	*rgba8 = nullptr;
	std::vector<uint8_t> png, out;
	if (!read_file(path, png)) return vkr_set_error_ext(VKR_E_IO, "%s: Unable to load texture base image (cannot open)", path);
	std::string err = decode_png(png, out, *width, *height);
	if (!err.empty()) return vkr_set_error_ext(VKR_E_IO, "%s: Unable to load texture base image (%s)", path, err.c_str());
	*rgba8 = (uint8_t*)malloc(out.size() ? out.size() : 1);
	if (!*rgba8) return vkr_set_error_ext(VKR_E_NOMEM, "out of host memory");
	memcpy(*rgba8, out.data(), out.size());
	return VKR_OK;
}

extern "C" void vkr_free(void* p) { free(p); }

// A frame as an 8-bit RGBA PNG (filter 0 on every row, zlib level 6).  Rows are written top first, the order
// vkr_render delivers them in; flip = 1 writes them bottom first (the order vkr_decode_png_file returns).
static int write_png_file_impl(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height, int flip)
{
	if (!path || !rgba8 || !width || !height) return vkr_set_error_ext(VKR_E_INVALID, "vkr_write_png_file: NULL argument or empty image");
	const size_t stride = (size_t)width * 4;
	std::vector<uint8_t> raw((stride + 1) * height);
	for (uint32_t y = 0; y < height; y++) {
		const uint32_t sy = flip ? height - 1 - y : y;
		raw[(stride + 1) * y] = 0; // filter type None
		memcpy(&raw[(stride + 1) * y + 1], rgba8 + stride * sy, stride);
	}
	uLongf zn = compressBound((uLong)raw.size());
	std::vector<uint8_t> z(zn);
	if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) return vkr_set_error_ext(VKR_E_NOMEM, "%s: deflate failed", path);
	FILE* f = fopen(path, "wb");
	if (!f) return vkr_set_error_ext(VKR_E_IO, "%s: cannot open for writing", path);
	auto be32 = [](uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
	bool ok = true;
	auto chunk = [&](const char* type, const uint8_t* data, uint32_t n) {
		uint8_t hdr[8];
		be32(hdr, n); memcpy(hdr + 4, type, 4);
		uint32_t crc = (uint32_t)crc32(0L, hdr + 4, 4);
		if (n) crc = (uint32_t)crc32(crc, data, n);
		uint8_t tail[4];
		be32(tail, crc);
		ok = ok && fwrite(hdr, 1, 8, f) == 8 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tail, 1, 4, f) == 4;
	};
	static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
	ok = fwrite(sig, 1, 8, f) == 8;
	uint8_t ihdr[13];
	be32(ihdr, width); be32(ihdr + 4, height);
	ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0; // 8-bit RGBA, deflate, adaptive filtering, no interlace
	chunk("IHDR", ihdr, 13);
	chunk("IDAT", z.data(), (uint32_t)zn);
	chunk("IEND", nullptr, 0);
	ok = (fclose(f) == 0) && ok;
	if (!ok) return vkr_set_error_ext(VKR_E_IO, "%s: write failed", path);
	return VKR_OK;
}

static int scene_load_impl(vkr_context* ctx, const char* path, vkr_scene** out)
{
	if (!ctx || !path || !out) return vkr_set_error_ext(VKR_E_INVALID, "vkr_scene_load: NULL argument");
	*out = nullptr;
	std::vector<uint8_t> file;
	if (!read_file(path, file)) return vkr_set_error_ext(VKR_E_IO, "load_baked_model(): unable to open '%s' for reading", path);
	Reader rd{ file.data(), file.size() };
	const char* slash = strrchr(path, '/');
	const std::string prefix = slash ? std::string(path, slash + 1) : std::string();

	static const char kMagic[16] = "\0\0COMP5892Mmesh"; // main/baked_model.cpp:12
	static const char kVariant[16] = "21-tan";           // main/baked_model.cpp:13
	char magic[16], variant[16];
	if (!rd.take(magic, 16) || memcmp(magic, kMagic, 16)) return vkr_set_error_ext(VKR_E_IO, "%s: invalid file signature!", path);
	if (!rd.take(variant, 16) || memcmp(variant, kVariant, 16)) {
		variant[15] = 0;
		return vkr_set_error_ext(VKR_E_IO, "%s: file variant is '%s', expected '%s'", path, variant, kVariant);
	}
	auto truncated = [&]() { return vkr_set_error_ext(VKR_E_IO, "%s: unexpected end of file at byte %zu", path, rd.off); };

	uint32_t nTex = 0;
	if (!rd.u32(nTex)) return truncated();
	// a count the rest of the file cannot hold is a truncated (or hostile) file, not a reason to allocate for it
	if ((uint64_t)nTex * 6 > rd.n - rd.off) return truncated();
	struct Tex { std::string path; uint8_t space, channels; std::vector<uint8_t> rgba; uint32_t w = 0, h = 0; };
	std::vector<Tex> tex(nTex);
	for (auto& t : tex) {
		uint32_t len = 0;
		if (!rd.u32(len)) return truncated();
		if (len >= 32 * 1024) return vkr_set_error_ext(VKR_E_IO, "read_string_(): unexpectedly long string (%u bytes)", len); // kMaxString
		std::string rel(len, '\0');
		if (len && !rd.take(&rel[0], len)) return truncated();
		// the writer stores the length including the terminating NUL; drop trailing NULs for the OS path
		while (!rel.empty() && rel.back() == '\0') rel.pop_back();
		t.path = prefix + rel;
		if (!rd.take(&t.space, 1) || !rd.take(&t.channels, 1)) return truncated();
	}
	uint32_t nMat = 0;
	if (!rd.u32(nMat)) return truncated();
	if ((uint64_t)nMat * sizeof(vkr_material_desc) > rd.n - rd.off) return truncated();
	std::vector<vkr_material_desc> mats(nMat);
	for (auto& m : mats)
		if (!rd.take(&m, sizeof m)) return truncated();
	uint32_t nMesh = 0;
	if (!rd.u32(nMesh)) return truncated();
	if ((uint64_t)nMesh * 12 > rd.n - rd.off) return truncated();
	struct Mesh { uint32_t mat, V, I; size_t pos, nrm, uv, tan, tc, idx; };
	std::vector<Mesh> meshes(nMesh);
	for (auto& m : meshes) {
		if (!rd.u32(m.mat) || !rd.u32(m.V) || !rd.u32(m.I)) return truncated();
		auto skip = [&](size_t bytes, size_t& at) { at = rd.off; if (rd.off + bytes > rd.n) return false; rd.off += bytes; return true; };
		if (!skip((size_t)m.V * 12, m.pos) || !skip((size_t)m.V * 12, m.nrm) || !skip((size_t)m.V * 8, m.uv) ||
		    !skip((size_t)m.V * 16, m.tan) || !skip((size_t)m.V * 4, m.tc) || !skip((size_t)m.I * 4, m.idx)) return truncated();
	}
	if (rd.off != rd.n) fprintf(stderr, "Note: '%s' contains trailing bytes\n", path); // main/baked_model.cpp:170-171

	// Every image is read and decoded on its own; a small pool of host threads takes them in file order.  The error
	// reported is that of the first texture in file order that failed, as a sequential loop would report it.
	{
		std::vector<std::string> errs(nTex);
		std::atomic<uint32_t> next{ 0 };
		std::atomic<bool> failed{ false };
		auto work = [&]() {
			for (uint32_t i; !failed.load(std::memory_order_relaxed) && (i = next.fetch_add(1)) < nTex;) {
				Tex& t = tex[i];
				try {
					std::vector<uint8_t> png;
					if (!read_file(t.path, png)) errs[i] = "cannot open";
					else errs[i] = decode_png(png, t.rgba, t.w, t.h);
				} catch (const std::exception& e) {
					errs[i] = e.what();
				}
				if (!errs[i].empty()) failed.store(true, std::memory_order_relaxed);
			}
		};
		cpu_set_t cpus;
		unsigned avail = sched_getaffinity(0, sizeof cpus, &cpus) == 0 ? (unsigned)CPU_COUNT(&cpus) : 1u;
		const unsigned nThreads = std::max(1u, std::min({ avail, 16u, (unsigned)nTex }));
		std::vector<std::thread> pool;
		try {
			for (unsigned k = 1; k < nThreads; k++) pool.emplace_back(work);
		} catch (const std::system_error&) {
			// no more threads to be had: the ones running and this one do the work
		}
		work();
		for (auto& th : pool) th.join();
		// a worker stops taking images once one has failed, so images after the failure may be missing: that is fine,
		// the load fails; the ones before it were all taken (indices are handed out in order) and have finished
		for (uint32_t i = 0; i < nTex; i++)
			if (!errs[i].empty())
				return vkr_set_error_ext(VKR_E_IO, "%s: Unable to load texture base image (%s)", tex[i].path.c_str(), errs[i].c_str());
	}
	// the arrays inside the file buffer may be unaligned: copy them out
	std::vector<std::vector<uint8_t>> keep;
	auto aligned = [&](size_t at, size_t bytes) -> const void* {
		keep.emplace_back(file.begin() + (long)at, file.begin() + (long)(at + bytes));
		return keep.back().data();
	};
	std::vector<vkr_texture_desc> td(nTex);
	for (uint32_t i = 0; i < nTex; i++) {
		memset(&td[i], 0, sizeof td[i]);
		td[i].rgba8 = tex[i].rgba.data(); td[i].width = tex[i].w; td[i].height = tex[i].h;
		td[i].space = tex[i].space; td[i].channels = tex[i].channels;
	}
	std::vector<vkr_mesh_desc> md(nMesh);
	keep.reserve((size_t)nMesh * 6);
	for (uint32_t i = 0; i < nMesh; i++) {
		const Mesh& m = meshes[i];
		memset(&md[i], 0, sizeof md[i]);
		md[i].materialId = m.mat; md[i].vertexCount = m.V; md[i].indexCount = m.I;
		md[i].positions = (const float*)aligned(m.pos, (size_t)m.V * 12);
		md[i].normals = (const float*)aligned(m.nrm, (size_t)m.V * 12);
		md[i].texcoords = (const float*)aligned(m.uv, (size_t)m.V * 8);
		md[i].tangents = (const float*)aligned(m.tan, (size_t)m.V * 16);
		md[i].tangentsComp = (const uint32_t*)aligned(m.tc, (size_t)m.V * 4);
		md[i].indices = (const uint32_t*)aligned(m.idx, (size_t)m.I * 4);
	}
	vkr_scene_desc d;
	memset(&d, 0, sizeof d);
	d.textureCount = nTex; d.materialCount = nMat; d.meshCount = nMesh;
	d.textures = td.data(); d.materials = mats.data(); d.meshes = md.data();
	return vkr_scene_create(ctx, &d, out);
}

// No exception may leave the library (std::bad_alloc / std::length_error from the containers above).
template <class F> static int guarded(const char* fn, F&& body) noexcept
{
	try { return body(); }
	catch (const std::bad_alloc&) { return vkr_set_error_ext(VKR_E_NOMEM, "%s: out of host memory", fn); }
	catch (const std::exception& e) { return vkr_set_error_ext(VKR_E_INVALID, "%s: %s", fn, e.what()); }
	catch (...) { return vkr_set_error_ext(VKR_E_INVALID, "%s: unknown exception", fn); }
}
extern "C" int vkr_decode_png_file(const char* path, uint8_t** rgba8, uint32_t* width, uint32_t* height)
{
	return guarded("vkr_decode_png_file", [&] { return decode_png_file_impl(path, rgba8, width, height); });
}
extern "C" int vkr_write_png_file(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height, int flip)
{
	return guarded("vkr_write_png_file", [&] { return write_png_file_impl(path, rgba8, width, height, flip); });
}
extern "C" int vkr_scene_load(vkr_context* ctx, const char* path, vkr_scene** out)
{
	return guarded("vkr_scene_load", [&] { return scene_load_impl(ctx, path, out); });
}
