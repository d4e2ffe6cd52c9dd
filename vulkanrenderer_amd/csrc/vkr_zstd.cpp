// vkr_zstd.cpp — a Zstandard frame decoder written from the format specification (RFC 8878), for the bake step: the
// reference's model source is a zstd-compressed Wavefront OBJ that it reads through libzstd's streaming decoder
// (main-bake/zstdistream.cpp:35-117, called from main-bake/load_model_obj.cpp:21-22).  Decoder only, whole buffer in,
// whole buffer out; every block type (raw, RLE, compressed), every literals type (raw, RLE, Huffman with direct or
// FSE-compressed weights, treeless), every sequence table mode (predefined, RLE, FSE-compressed, repeat), repeat
// offsets, skippable frames, concatenated frames and the optional content checksum (XXH64).  Dictionaries are not
// supported (the reference uses none).
#include "vkr_zstd.h"

#include <cstring>

namespace {

struct Err { const char* msg; };
[[noreturn]] void bad(const char* msg) { throw Err{ msg }; }

inline int highest_bit(uint64_t v) // index of the highest set bit, -1 for 0
{
	int n = -1;
	while (v) { n++; v >>= 1; }
	return n;
}

// ---- forward bit reader (FSE table descriptions): little-endian bit order ----
struct FwdBits {
	const uint8_t* p; size_t len; size_t bit = 0;
	uint32_t read(int n)
	{
		uint32_t v = 0;
		for (int i = 0; i < n; i++) {
			const size_t byte = (bit + i) >> 3;
			if (byte >= len) bad("table description runs past its section");
			v |= (uint32_t)((p[byte] >> ((bit + i) & 7)) & 1u) << i;
		}
		bit += n;
		return v;
	}
	void rewind(int n) { bit -= n; }
	size_t bytes_used() const { return (bit + 7) >> 3; }
};

// ---- backward bit reader (Huffman and FSE streams): the last byte carries a final-bit flag ----
struct BackBits {
	const uint8_t* p; size_t len; int64_t off; // off = number of unread bits; reads beyond the start deliver zeros
	BackBits(const uint8_t* src, size_t n) : p(src), len(n)
	{
		if (!len || !src[len - 1]) bad("bitstream without its end mark");
		off = (int64_t)len * 8 - (8 - highest_bit(src[len - 1]));
	}
	uint64_t read(int n)
	{
		off -= n;
		if (n == 0) return 0;
		if (off >= 0 && n <= 56) { // the common case: the bits lie inside the stream, at most 8 bytes cover them
			const size_t byte = (size_t)off >> 3;
			const size_t avail = len - byte;
			uint64_t w = 0;
			std::memcpy(&w, p + byte, avail >= 8 ? 8 : avail);
			return (w >> (off & 7)) & (((uint64_t)1 << n) - 1);
		}
		uint64_t v = 0;
		for (int i = 0; i < n; i++) {
			const int64_t b = off + i;
			if (b >= 0) v |= (uint64_t)((p[b >> 3] >> (b & 7)) & 1u) << i;
		}
		return v;
	}
};

// ---- FSE ----
struct FseTable {
	int accuracy = 0;
	std::vector<uint8_t> symbol, nbits;
	std::vector<uint16_t> base;
	bool valid = false;
};

void fse_build(FseTable& t, const int16_t* freq, int nsym, int accuracy)
{
	const int size = 1 << accuracy;
	t.accuracy = accuracy;
	t.symbol.assign(size, 0); t.nbits.assign(size, 0); t.base.assign(size, 0);
	std::vector<uint16_t> next(nsym);
	int high = size;
	for (int s = 0; s < nsym; s++)
		if (freq[s] == -1) { t.symbol[--high] = (uint8_t)s; next[s] = 1; }
	const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
	int pos = 0;
	for (int s = 0; s < nsym; s++) {
		if (freq[s] <= 0) continue;
		next[s] = (uint16_t)freq[s];
		for (int i = 0; i < freq[s]; i++) {
			t.symbol[pos] = (uint8_t)s;
			do { pos = (pos + step) & mask; } while (pos >= high);
		}
	}
	if (pos != 0) bad("FSE distribution does not fill its table");
	for (int i = 0; i < size; i++) {
		const int s = t.symbol[i];
		const int n = next[s]++;
		t.nbits[i] = (uint8_t)(accuracy - highest_bit((uint64_t)n));
		t.base[i] = (uint16_t)((n << t.nbits[i]) - size);
	}
	t.valid = true;
}

// reads a table description; returns the bytes it occupies
size_t fse_read(FseTable& t, const uint8_t* src, size_t len, int maxAccuracy, int maxSymbols)
{
	FwdBits in{ src, len };
	const int accuracy = 5 + (int)in.read(4);
	if (accuracy > maxAccuracy) bad("FSE accuracy log too large");
	int remaining = 1 << accuracy;
	int16_t freq[256];
	int symb = 0;
	while (remaining > 0 && symb < maxSymbols) {
		const int bits = highest_bit((uint64_t)remaining + 1) + 1;
		int val = (int)in.read(bits);
		const int lower = (1 << (bits - 1)) - 1;
		const int threshold = (1 << bits) - 1 - (remaining + 1);
		if ((val & lower) < threshold) { in.rewind(1); val &= lower; }
		else if (val > lower) val -= threshold;
		const int proba = val - 1;
		remaining -= proba < 0 ? -proba : proba;
		freq[symb++] = (int16_t)proba;
		if (proba == 0) {
			int repeat = (int)in.read(2);
			for (;;) {
				for (int i = 0; i < repeat && symb < maxSymbols; i++) freq[symb++] = 0;
				if (repeat == 3) repeat = (int)in.read(2); else break;
			}
		}
	}
	if (remaining != 0 || symb > maxSymbols) bad("FSE distribution is inconsistent");
	fse_build(t, freq, symb, accuracy);
	return in.bytes_used();
}

void fse_rle(FseTable& t, uint8_t sym)
{
	t.accuracy = 0; t.symbol.assign(1, sym); t.nbits.assign(1, 0); t.base.assign(1, 0); t.valid = true;
}

// ---- Huffman ----
struct HufTable {
	int maxBits = 0;
	std::vector<uint8_t> symbol, nbits;
	bool valid = false;
};

void huf_build(HufTable& h, const uint8_t* weights, int n)
{
	// the last symbol's weight completes the sum of 2^(w-1) to a power of two
	uint32_t sum = 0;
	for (int i = 0; i < n; i++) {
		if (weights[i] > 11) bad("Huffman weight too large");
		if (weights[i]) sum += 1u << (weights[i] - 1);
	}
	if (!sum) bad("Huffman tree without symbols");
	const int maxBits = highest_bit(sum) + 1;
	const uint32_t left = (1u << maxBits) - sum;
	if (left & (left - 1)) bad("Huffman weights do not sum to a power of two");
	if (maxBits > 11 || n + 1 > 256) bad("Huffman tree too deep");
	uint8_t bits[256];
	int count = n + 1;
	for (int i = 0; i < n; i++) bits[i] = weights[i] ? (uint8_t)(maxBits + 1 - weights[i]) : 0;
	bits[n] = (uint8_t)(maxBits + 1 - (highest_bit(left) + 1));
	h.maxBits = maxBits;
	const int size = 1 << maxBits;
	h.symbol.assign(size, 0); h.nbits.assign(size, 0);
	uint32_t rankCount[13] = { 0 }, rankIdx[13] = { 0 };
	for (int i = 0; i < count; i++) rankCount[bits[i]]++;
	rankIdx[maxBits] = 0;
	for (int i = maxBits; i >= 1; i--) {
		rankIdx[i - 1] = rankIdx[i] + rankCount[i] * (1u << (maxBits - i));
		for (uint32_t k = rankIdx[i]; k < rankIdx[i - 1]; k++) h.nbits[k] = (uint8_t)i;
	}
	if (rankIdx[0] != (uint32_t)size) bad("Huffman code is not complete");
	for (int i = 0; i < count; i++) {
		if (!bits[i]) continue;
		const uint32_t len = 1u << (maxBits - bits[i]);
		for (uint32_t k = 0; k < len; k++) h.symbol[rankIdx[bits[i]] + k] = (uint8_t)i;
		rankIdx[bits[i]] += len;
	}
	h.valid = true;
}

// reads a tree description; returns the bytes it occupies
size_t huf_read(HufTable& h, const uint8_t* src, size_t len)
{
	if (!len) bad("missing Huffman tree");
	const int header = src[0];
	uint8_t weights[256];
	int n = 0;
	size_t used;
	if (header >= 128) {
		n = header - 127;
		used = 1 + (size_t)(n + 1) / 2;
		if (used > len) bad("Huffman weights run past their section");
		for (int i = 0; i < n; i++) weights[i] = (i & 1) ? (src[1 + i / 2] & 15) : (src[1 + i / 2] >> 4);
	} else {
		used = 1 + (size_t)header;
		if (used > len || !header) bad("Huffman weights run past their section");
		FseTable t;
		const size_t th = fse_read(t, src + 1, header, 6, 12); // weights 0..11 with an accuracy log of at most 6 (RFC 8878 4.2.1.1)
		if (th >= (size_t)header) bad("Huffman weights: empty stream");
		BackBits in(src + 1 + th, header - th);
		uint32_t s1 = (uint32_t)in.read(t.accuracy), s2 = (uint32_t)in.read(t.accuracy);
		for (;;) {
			if (n >= 254) bad("too many Huffman weights");
			weights[n++] = t.symbol[s1];
			s1 = t.base[s1] + (uint32_t)in.read(t.nbits[s1]);
			if (in.off < 0) { weights[n++] = t.symbol[s2]; break; }
			if (n >= 254) bad("too many Huffman weights");
			weights[n++] = t.symbol[s2];
			s2 = t.base[s2] + (uint32_t)in.read(t.nbits[s2]);
			if (in.off < 0) { weights[n++] = t.symbol[s1]; break; }
		}
	}
	huf_build(h, weights, n);
	return used;
}

void huf_stream(const HufTable& h, const uint8_t* src, size_t len, uint8_t* dst, size_t count)
{
	BackBits in(src, len);
	const uint32_t mask = (1u << h.maxBits) - 1u;
	uint32_t state = (uint32_t)in.read(h.maxBits);
	for (size_t i = 0; i < count; i++) {
		dst[i] = h.symbol[state];
		const int nb = h.nbits[state];
		state = ((state << nb) & mask) | (uint32_t)in.read(nb);
	}
	if (in.off != -(int64_t)h.maxBits) bad("Huffman stream does not end where its symbols do");
}

// ---- sequences ----
const int16_t kLLDefault[36] = { 4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1 };
const int16_t kMLDefault[53] = { 1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
	1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1 };
const int16_t kOFDefault[29] = { 1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1 };
const uint32_t kLLBase[36] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048,
	4096, 8192, 16384, 32768, 65536 };
const uint8_t kLLBits[36] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };
const uint32_t kMLBase[53] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
	35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539 };
const uint8_t kMLBits[53] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5,
	7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };

struct FrameState {
	HufTable huf;
	FseTable ll, of, ml;
	uint64_t rep[3] = { 1, 4, 8 };
};

size_t table_for_mode(int mode, FseTable& t, const uint8_t* src, size_t len, const int16_t* def, int defN, int defAcc, int maxAcc, int maxSym)
{
	switch (mode) {
	case 0: fse_build(t, def, defN, defAcc); return 0;
	case 1: if (!len) bad("missing RLE symbol"); if (src[0] >= maxSym) bad("RLE symbol out of range"); fse_rle(t, src[0]); return 1;
	case 2: return fse_read(t, src, len, maxAcc, maxSym);
	default: if (!t.valid) bad("repeat mode without an earlier table"); return 0;
	}
}

void decode_block(FrameState& st, const uint8_t* src, size_t len, std::vector<uint8_t>& out, size_t frameStart, uint64_t windowSize, std::vector<uint8_t>& lit)
{
	if (len < 1) bad("empty compressed block");
	// ---- literals section
	const int ltype = src[0] & 3, sf = (src[0] >> 2) & 3;
	size_t regen = 0, comp = 0, hdr = 0;
	int streams = 1;
	if (ltype < 2) {
		if (sf == 0 || sf == 2) { hdr = 1; regen = src[0] >> 3; }
		else if (sf == 1) { if (len < 2) bad("truncated literals header"); hdr = 2; regen = (src[0] | (src[1] << 8)) >> 4; }
		else { if (len < 3) bad("truncated literals header"); hdr = 3; regen = ((uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16)) >> 4; }
	} else {
		const size_t need = sf < 2 ? 3 : sf == 2 ? 4 : 5;
		if (len < need) bad("truncated literals header");
		uint64_t v = 0;
		for (size_t i = 0; i < need; i++) v |= (uint64_t)src[i] << (8 * i);
		hdr = need;
		if (sf < 2) { regen = (v >> 4) & 0x3ff; comp = (v >> 14) & 0x3ff; streams = sf == 0 ? 1 : 4; }
		else if (sf == 2) { regen = (v >> 4) & 0x3fff; comp = (v >> 18) & 0x3fff; streams = 4; }
		else { regen = (v >> 4) & 0x3ffff; comp = (v >> 22) & 0x3ffff; streams = 4; }
	}
	if (regen > (128u << 10)) bad("literals larger than a block");
	lit.resize(regen);
	size_t pos = hdr;
	if (ltype == 0) {
		if (pos + regen > len) bad("raw literals run past the block");
		if (regen) std::memcpy(lit.data(), src + pos, regen);
		pos += regen;
	} else if (ltype == 1) {
		if (pos + 1 > len) bad("RLE literals run past the block");
		if (regen) std::memset(lit.data(), src[pos], regen);
		pos += 1;
	} else {
		if (pos + comp > len) bad("compressed literals run past the block");
		const uint8_t* p = src + pos;
		size_t left = comp;
		if (ltype == 2) {
			const size_t used = huf_read(st.huf, p, left);
			p += used; left -= used;
		} else if (!st.huf.valid) bad("treeless literals without an earlier tree");
		if (streams == 1) {
			huf_stream(st.huf, p, left, lit.data(), regen);
		} else {
			if (left < 6) bad("missing jump table");
			const size_t s1 = p[0] | (p[1] << 8), s2 = p[2] | (p[3] << 8), s3 = p[4] | (p[5] << 8);
			if (6 + s1 + s2 + s3 > left) bad("jump table runs past the literals");
			const size_t s4 = left - 6 - s1 - s2 - s3;
			const size_t q = (regen + 3) / 4;
			if (3 * q > regen) bad("literal streams do not add up");
			const uint8_t* b = p + 6;
			huf_stream(st.huf, b, s1, lit.data(), q);
			huf_stream(st.huf, b + s1, s2, lit.data() + q, q);
			huf_stream(st.huf, b + s1 + s2, s3, lit.data() + 2 * q, q);
			huf_stream(st.huf, b + s1 + s2 + s3, s4, lit.data() + 3 * q, regen - 3 * q);
		}
		pos += comp;
	}
	// ---- sequences section
	if (pos >= len) bad("missing sequences section");
	size_t nseq = src[pos++];
	if (nseq >= 128) {
		if (nseq == 255) { if (pos + 2 > len) bad("truncated sequence count"); nseq = (size_t)src[pos] + ((size_t)src[pos + 1] << 8) + 0x7f00; pos += 2; }
		else { if (pos + 1 > len) bad("truncated sequence count"); nseq = ((nseq - 128) << 8) + src[pos]; pos += 1; }
	}
	size_t litPos = 0;
	const size_t blockStart = out.size();
	const size_t kBlockMax = 128u << 10; // Block_Maximum_Size: no block regenerates more
	if (nseq) {
		if (pos >= len) bad("missing compression modes");
		const int modes = src[pos++];
		if (modes & 3) bad("reserved bits set in the compression modes");
		pos += table_for_mode((modes >> 6) & 3, st.ll, src + pos, len - pos, kLLDefault, 36, 6, 9, 36);
		pos += table_for_mode((modes >> 4) & 3, st.of, src + pos, len - pos, kOFDefault, 29, 5, 8, 32);
		pos += table_for_mode((modes >> 2) & 3, st.ml, src + pos, len - pos, kMLDefault, 53, 6, 9, 53);
		if (pos >= len) bad("missing sequence bitstream");
		BackBits in(src + pos, len - pos);
		uint32_t sLL = (uint32_t)in.read(st.ll.accuracy), sOF = (uint32_t)in.read(st.of.accuracy), sML = (uint32_t)in.read(st.ml.accuracy);
		for (size_t i = 0; i < nseq; i++) {
			const int ofCode = st.of.symbol[sOF], mlCode = st.ml.symbol[sML], llCode = st.ll.symbol[sLL];
			if (ofCode > 31 || mlCode > 52 || llCode > 35) bad("sequence code out of range");
			const uint64_t ofValue = ((uint64_t)1 << ofCode) + in.read(ofCode);
			const uint64_t mlen = kMLBase[mlCode] + in.read(kMLBits[mlCode]);
			const uint64_t llen = kLLBase[llCode] + in.read(kLLBits[llCode]);
			if (i + 1 < nseq) {
				sLL = st.ll.base[sLL] + (uint32_t)in.read(st.ll.nbits[sLL]);
				sML = st.ml.base[sML] + (uint32_t)in.read(st.ml.nbits[sML]);
				sOF = st.of.base[sOF] + (uint32_t)in.read(st.of.nbits[sOF]);
			}
			if (in.off < 0) bad("sequence bitstream exhausted");
			uint64_t offset;
			if (ofValue > 3) {
				offset = ofValue - 3;
				st.rep[2] = st.rep[1]; st.rep[1] = st.rep[0]; st.rep[0] = offset;
			} else {
				uint32_t idx = (uint32_t)ofValue - 1;
				if (llen == 0) idx++;
				if (idx == 0) offset = st.rep[0];
				else {
					offset = idx < 3 ? st.rep[idx] : st.rep[0] - 1;
					if (idx > 1) st.rep[2] = st.rep[1];
					st.rep[1] = st.rep[0]; st.rep[0] = offset;
				}
			}
			if (litPos + llen > lit.size()) bad("sequence asks for more literals than the block holds");
			if ((out.size() - blockStart) + llen + mlen > kBlockMax) bad("block regenerates more than the block maximum");
			out.insert(out.end(), lit.begin() + litPos, lit.begin() + litPos + llen);
			litPos += llen;
			const size_t have = out.size() - frameStart;
			if (offset == 0 || offset > have || offset > windowSize) bad("match offset outside the window");
			const size_t from = out.size() - (size_t)offset;
			out.resize(out.size() + (size_t)mlen);
			uint8_t* d = out.data();
			const size_t to = out.size() - (size_t)mlen;
			if (offset >= mlen) std::memcpy(d + to, d + from, (size_t)mlen);
			else for (size_t k = 0; k < mlen; k++) d[to + k] = d[from + k];
		}
		if (in.off != 0) bad("sequence bitstream not fully consumed");
	} else if (pos != len) bad("bytes after an empty sequences section");
	out.insert(out.end(), lit.begin() + litPos, lit.end());
}

// ---- XXH64 (content checksum) ----
const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull, P4 = 9650029242287828579ull,
               P5 = 2870177450012600261ull;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline uint64_t xround(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
inline uint64_t xmerge(uint64_t acc, uint64_t v) { return (acc ^ xround(0, v)) * P1 + P4; }

uint64_t xxh64(const uint8_t* p, size_t len)
{
	const uint8_t* const end = p + len;
	uint64_t h;
	if (len >= 32) {
		uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
		do {
			v1 = xround(v1, rd64(p)); v2 = xround(v2, rd64(p + 8)); v3 = xround(v3, rd64(p + 16)); v4 = xround(v4, rd64(p + 24));
			p += 32;
		} while (p + 32 <= end);
		h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
		h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
	} else h = P5;
	h += (uint64_t)len;
	while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
	if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
	while (p < end) { h ^= (uint64_t)(*p++) * P5; h = rotl(h, 11) * P1; }
	h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
	return h;
}

size_t decode_frame(const uint8_t* src, size_t len, std::vector<uint8_t>& out)
{
	size_t pos = 4;
	if (len < 5) bad("truncated frame header");
	const int fhd = src[pos++];
	if (fhd & 0x08) bad("reserved bit set in the frame header");
	const int fcsFlag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, dictFlag = fhd & 3;
	uint64_t windowSize = 0;
	if (!single) {
		if (pos >= len) bad("truncated frame header");
		const int wd = src[pos++];
		const int exponent = wd >> 3, mantissa = wd & 7;
		const uint64_t base = (uint64_t)1 << (10 + exponent);
		windowSize = base + (base / 8) * mantissa;
	}
	const int dictBytes = dictFlag == 0 ? 0 : dictFlag == 1 ? 1 : dictFlag == 2 ? 2 : 4;
	if (pos + dictBytes > len) bad("truncated frame header");
	uint32_t dictId = 0;
	for (int i = 0; i < dictBytes; i++) dictId |= (uint32_t)src[pos + i] << (8 * i);
	pos += dictBytes;
	if (dictId) bad("the frame needs a dictionary");
	const int fcsBytes = fcsFlag == 0 ? (single ? 1 : 0) : fcsFlag == 1 ? 2 : fcsFlag == 2 ? 4 : 8;
	if (pos + fcsBytes > len) bad("truncated frame header");
	uint64_t contentSize = 0;
	for (int i = 0; i < fcsBytes; i++) contentSize |= (uint64_t)src[pos + i] << (8 * i);
	if (fcsBytes == 2) contentSize += 256;
	pos += fcsBytes;
	if (single) windowSize = contentSize;
	if (windowSize > ((uint64_t)1 << 31)) bad("window larger than 2 GiB");
	const size_t frameStart = out.size();
	// a hint only, and a header can claim anything: never more than 256 MiB ahead of the data
	if (fcsBytes) out.reserve(out.size() + (size_t)(contentSize < ((uint64_t)256 << 20) ? contentSize : ((uint64_t)256 << 20)));
	FrameState st;
	std::vector<uint8_t> lit;
	// Block_Maximum_Size: the smaller of the window and 128 KiB (RFC 8878 3.1.1.2.4)
	const size_t blockMax = (size_t)(windowSize < (128u << 10) ? windowSize : (128u << 10));
	for (;;) {
		if (pos + 3 > len) bad("truncated block header");
		const uint32_t bh = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
		pos += 3;
		const int last = bh & 1, type = (bh >> 1) & 3;
		const size_t size = bh >> 3;
		if (type == 3) bad("reserved block type");
		if (type != 2 && size > blockMax) bad("block larger than the frame's block maximum");
		if (type == 0) {
			if (pos + size > len) bad("raw block runs past the input");
			out.insert(out.end(), src + pos, src + pos + size);
			pos += size;
		} else if (type == 1) {
			if (pos + 1 > len) bad("RLE block runs past the input");
			out.insert(out.end(), size, src[pos]);
			pos += 1;
		} else {
			if (pos + size > len) bad("compressed block runs past the input");
			if (size > blockMax) bad("compressed block larger than the frame's block maximum");
			const size_t before = out.size();
			decode_block(st, src + pos, size, out, frameStart, windowSize, lit);
			if (out.size() - before > blockMax) bad("block regenerates more than the block maximum");
			pos += size;
		}
		if (last) break;
	}
	if (fcsBytes && out.size() - frameStart != contentSize) bad("frame content size does not match its header");
	if (checksum) {
		if (pos + 4 > len) bad("missing content checksum");
		const uint32_t want = rd32(src + pos);
		pos += 4;
		if ((uint32_t)xxh64(out.data() + frameStart, out.size() - frameStart) != want) bad("content checksum mismatch");
	}
	return pos;
}

} // namespace

bool vkr_zstd_is_frame(const uint8_t* src, size_t len)
{
	if (len < 4) return false;
	const uint32_t magic = rd32(src);
	return magic == 0xFD2FB528u || (magic & 0xFFFFFFF0u) == 0x184D2A50u;
}

bool vkr_zstd_decompress(const uint8_t* src, size_t len, std::vector<uint8_t>& out, std::string& error)
{
	try {
		size_t pos = 0;
		bool any = false;
		while (pos < len) {
			if (len - pos < 4) bad("trailing bytes after the last frame");
			const uint32_t magic = rd32(src + pos);
			if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) { // skippable frame
				if (len - pos < 8) bad("truncated skippable frame");
				const uint64_t n = rd32(src + pos + 4);
				if (8 + n > len - pos) bad("skippable frame runs past the input");
				pos += 8 + (size_t)n;
				continue;
			}
			if (magic != 0xFD2FB528u) bad("not a Zstandard frame");
			pos += decode_frame(src + pos, len - pos, out);
			any = true;
		}
		if (!any) bad("no Zstandard frame in the input");
		return true;
	} catch (const Err& e) {
		error = std::string("zstd: ") + e.msg;
		return false;
	}
}
