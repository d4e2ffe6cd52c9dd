// vkr_bake.cpp — libvkr_bake.so: OBJ/MTL -> welded, tangent-carrying meshes ->
// ".comp5892mesh" ("21-tan").  Host code only; compiled with -ffp-contract=off
// and without FMA so that the double arithmetic of the tangent generator and
// the float arithmetic of the TBN packer round as the reference tool's do.
//
//   load_obj            main-bake/load_model_obj.cpp:16-172 (rapidobj replaced by a text parser)
//   index_mesh          main-bake/index_mesh.cpp:76-228, :390-471
//   tangent space       third_party/tgen/src/tgen.cpp:98-258
//   pack_tbn            main-bake/index_mesh.cpp:178-220
//   write_model         main-bake/main.cpp:276-389
//   unique textures     main-bake/main.cpp:426-476, fallbacks :236-251
#include "../../include/vkr_bake.h"
#include "vkr_zstd.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <charconv>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <iterator>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return 1;
}

// ---------------------------------------------------------------------------
// welding: main-bake/index_mesh.cpp:390-471.  For every soup vertex i not yet
// merged, every other unmerged vertex whose position, normal and texture
// coordinate lie within the tolerance of i's (component-wise, compared with
// i only, not transitively) collapses onto i; output vertices are numbered in
// the order their first soup vertex appears.  The reference finds the
// candidates through a hashed grid of cell size ~2*tolerance and its 27
// neighbours; any search that returns every vertex within the tolerance gives
// the same result, whatever order it returns them in.
// ---------------------------------------------------------------------------
struct Soup {
	const float* pos; const float* nrm; const float* uv;
	size_t n;
};

inline bool mergable(const Soup& s, size_t i, size_t j, float tol)
{
	for (int k = 0; k < 3; k++)
		if (std::fabs(s.pos[3 * i + k] - s.pos[3 * j + k]) > tol) return false;
	if (s.nrm)
		for (int k = 0; k < 3; k++)
			if (std::fabs(s.nrm[3 * i + k] - s.nrm[3 * j + k]) > tol) return false;
	for (int k = 0; k < 2; k++)
		if (std::fabs(s.uv[2 * i + k] - s.uv[2 * j + k]) > tol) return false;
	return true;
}

struct CellKey {
	int64_t x, y, z;
	bool operator==(const CellKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct CellHash {
	size_t operator()(const CellKey& k) const
	{
		uint64_t h = (uint64_t)k.x * 0x9E3779B97F4A7C15ull;
		h ^= (uint64_t)k.y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
		h ^= (uint64_t)k.z * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
		return (size_t)h;
	}
};

void weld(const Soup& s, float tol, std::vector<uint32_t>& indices, std::vector<size_t>& firstOf)
{
	indices.clear(); firstOf.clear();
	indices.reserve(s.n); firstOf.reserve(s.n);
	if (!s.n) return;
	double mn[3] = { s.pos[0], s.pos[1], s.pos[2] };
	for (size_t i = 0; i < s.n; i++)
		for (int k = 0; k < 3; k++) mn[k] = std::min(mn[k], (double)s.pos[3 * i + k]);
	// cells at least two tolerances wide: vertices within one tolerance sit in the same or an adjacent cell
	const double cell = tol > 0.0f ? 2.0 * (double)tol : 0.0;
	auto key_of = [&](size_t i) {
		CellKey k;
		if (cell > 0.0) {
			k.x = (int64_t)std::floor(((double)s.pos[3 * i + 0] - mn[0]) / cell);
			k.y = (int64_t)std::floor(((double)s.pos[3 * i + 1] - mn[1]) / cell);
			k.z = (int64_t)std::floor(((double)s.pos[3 * i + 2] - mn[2]) / cell);
		} else { // zero tolerance: only equal positions can merge (-0 == +0, so key by value, not by bits)
			auto bits = [](float f) { f += 0.0f; uint32_t u; std::memcpy(&u, &f, 4); return (int64_t)u; };
			k.x = bits(s.pos[3 * i + 0]); k.y = bits(s.pos[3 * i + 1]); k.z = bits(s.pos[3 * i + 2]);
		}
		return k;
	};
	std::unordered_map<CellKey, std::vector<uint32_t>, CellHash> grid;
	grid.reserve(s.n);
	for (size_t i = 0; i < s.n; i++) grid[key_of(i)].push_back((uint32_t)i);

	const size_t NONE = ~size_t(0);
	std::vector<size_t> collapse(s.n, NONE);
	size_t next = 0;
	const int reach = cell > 0.0 ? 1 : 0;
	for (size_t i = 0; i < s.n; i++) {
		if (collapse[i] != NONE) { indices.push_back((uint32_t)collapse[i]); continue; }
		const size_t to = next++;
		collapse[i] = to;
		firstOf.push_back(i);
		indices.push_back((uint32_t)to);
		const CellKey c = key_of(i);
		for (int dz = -reach; dz <= reach; dz++)
			for (int dy = -reach; dy <= reach; dy++)
				for (int dx = -reach; dx <= reach; dx++) {
					auto it = grid.find(CellKey{ c.x + dx, c.y + dy, c.z + dz });
					if (it == grid.end()) continue;
					for (uint32_t j : it->second) {
						if (j == i || collapse[j] != NONE) continue;
						if (mergable(s, i, j, tol)) collapse[j] = to;
					}
				}
	}
}

// ---------------------------------------------------------------------------
// tangent space, in double as tgen does (RealT = double, tgen.h:23)
// ---------------------------------------------------------------------------
inline void normalize3(double* v)
{
	const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
	const double inv = 1.0 / len;
	v[0] = v[0] * inv; v[1] = v[1] * inv; v[2] = v[2] * inv;
}
inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void cross3(const double* a, const double* b, double* r)
{
	r[0] = a[1] * b[2] - a[2] * b[1];
	r[1] = a[2] * b[0] - a[0] * b[2];
	r[2] = a[0] * b[1] - a[1] * b[0];
}

// positions/uvs/normals per vertex, indices per corner -> tangent4 per vertex
void tangent_space(const std::vector<uint32_t>& idx, const std::vector<double>& P, const std::vector<double>& UV,
                   const std::vector<double>& N, std::vector<double>& tan4)
{
	const size_t corners = idx.size(), verts = N.size() / 3;
	// computeCornerTSpace, tgen.cpp:98-169
	std::vector<double> cT(corners * 3), cB(corners * 3);
	for (size_t i = 0; i + 2 < corners; i += 3) {
		double e3[3][3], e2[3][2];
		for (size_t j = 0; j < 3; j++) {
			const size_t v0 = idx[i + j], v1 = idx[i + (j + 1) % 3];
			for (int k = 0; k < 3; k++) e3[j][k] = P[v1 * 3 + k] - P[v0 * 3 + k];
			for (int k = 0; k < 2; k++) e2[j][k] = UV[v1 * 2 + k] - UV[v0 * 2 + k];
		}
		for (size_t j = 0; j < 3; j++) {
			const size_t prev = (j + 2) % 3;
			const double* dP0 = e3[j]; const double* dP1n = e3[prev];
			const double* dU0 = e2[j]; const double* dU1n = e2[prev];
			const double denom = (dU0[0] * -dU1n[1] - dU0[1] * -dU1n[0]);
			const double r = std::fabs(denom) > 1e-10 ? 1.0 / denom : 0.0;
			const double s0 = -dU1n[1] * r, s1 = -dU0[1] * r, s2 = -dU0[0] * r, s3 = -dU1n[0] * r;
			for (int k = 0; k < 3; k++) {
				cT[(i + j) * 3 + k] = dP0[k] * s0 - dP1n[k] * s1;
				cB[(i + j) * 3 + k] = dP1n[k] * s2 - dP0[k] * s3;
			}
		}
	}
	// computeVertexTSpace, tgen.cpp:173-207 (the caller passes the index count as the vertex count; only the
	// first `verts` entries are read afterwards)
	std::vector<double> vT(verts * 3, 0.0), vB(verts * 3, 0.0);
	for (size_t i = 0; i < corners; i++) {
		const size_t v = idx[i];
		for (int k = 0; k < 3; k++) { vT[v * 3 + k] = cT[i * 3 + k] + vT[v * 3 + k]; vB[v * 3 + k] = cB[i * 3 + k] + vB[v * 3 + k]; }
	}
	for (size_t v = 0; v < verts; v++) { normalize3(&vT[v * 3]); normalize3(&vB[v * 3]); }
	// orthogonalizeTSpace, tgen.cpp:211-233; computeTangent4D, tgen.cpp:237-258
	tan4.resize(verts * 4);
	for (size_t v = 0; v < verts; v++) {
		const double* n = &N[v * 3];
		double* t = &vT[v * 3]; double* b = &vB[v * 3];
		const double d = dot3(n, t);
		for (int k = 0; k < 3; k++) t[k] = t[k] - n[k] * d;
		normalize3(t);
		cross3(n, t, b);
		double c[3];
		cross3(n, t, c);
		const double sign = dot3(c, b) > 0.0 ? 1.0 : -1.0;
		tan4[v * 4 + 0] = t[0]; tan4[v * 4 + 1] = t[1]; tan4[v * 4 + 2] = t[2]; tan4[v * 4 + 3] = sign;
	}
}

// ---------------------------------------------------------------------------
// TBN frame -> quaternion -> 2:10:10:10 (index_mesh.cpp:178-220), with glm's
// operation order: normalize = v * (1/sqrt(dot)), dot = (x*x + y*y) + z*z
// ---------------------------------------------------------------------------
inline void glm_normalize3(const float* a, float* o)
{
	const float t0 = a[0] * a[0], t1 = a[1] * a[1], t2 = a[2] * a[2];
	const float inv = 1.0f / std::sqrt(t0 + t1 + t2);
	o[0] = a[0] * inv; o[1] = a[1] * inv; o[2] = a[2] * inv;
}

uint32_t pack_tbn(const float tangent4[4], const float normal[3])
{
	const float Tin[3] = { tangent4[0] * tangent4[3], tangent4[1] * tangent4[3], tangent4[2] * tangent4[3] };
	float T[3], N[3], B[3], c[3];
	glm_normalize3(Tin, T);
	glm_normalize3(normal, N);
	c[0] = N[1] * T[2] - T[1] * N[2];
	c[1] = N[2] * T[0] - T[2] * N[0];
	c[2] = N[0] * T[1] - T[0] * N[1];
	glm_normalize3(c, B);
	// glm::quat_cast(mat3(T, B, N)), m[column][row]
	const float m00 = T[0], m01 = T[1], m02 = T[2], m10 = B[0], m11 = B[1], m12 = B[2], m20 = N[0], m21 = N[1], m22 = N[2];
	const float fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
	int big = 0; float fb = fw;
	if (fx > fb) { fb = fx; big = 1; }
	if (fy > fb) { fb = fy; big = 2; }
	if (fz > fb) { fb = fz; big = 3; }
	const float bv = std::sqrt(fb + 1.0f) * 0.5f, mult = 0.25f / bv;
	float q[4]; // x y z w
	switch (big) {
	case 0: q[3] = bv; q[0] = (m12 - m21) * mult; q[1] = (m20 - m02) * mult; q[2] = (m01 - m10) * mult; break;
	case 1: q[3] = (m12 - m21) * mult; q[0] = bv; q[1] = (m01 + m10) * mult; q[2] = (m20 + m02) * mult; break;
	case 2: q[3] = (m20 - m02) * mult; q[0] = (m01 + m10) * mult; q[1] = bv; q[2] = (m12 + m21) * mult; break;
	default: q[3] = (m01 - m10) * mult; q[0] = (m20 + m02) * mult; q[1] = (m12 + m21) * mult; q[2] = bv; break;
	}
	const float len = std::sqrt((q[3] * q[3] + q[0] * q[0]) + (q[1] * q[1] + q[2] * q[2]));
	if (len <= 0.0f) { q[0] = q[1] = q[2] = 0.0f; q[3] = 1.0f; }
	else { const float ol = 1.0f / len; for (float& v : q) v *= ol; }
	int maxIndex = 0; float mv = std::fabs(q[0]);
	for (int i = 1; i < 4; i++) if (std::fabs(q[i]) > mv) { mv = std::fabs(q[i]); maxIndex = i; }
	if (q[maxIndex] < 0.0f) for (float& v : q) v = -v;
	uint32_t small[3]; int k = 0;
	for (int i = 0; i < 4; i++) {
		if (i == maxIndex) continue;
		// (float + double) / double * float: evaluated in double, truncated
		const double v = ((double)q[i] + (1.0 / std::sqrt(2.0))) / std::sqrt(2.0) * (double)1023.0f;
		small[k++] = (v == v) ? (uint32_t)(int64_t)v : 0u;
	}
	return ((uint32_t)maxIndex << 30) | (small[0] << 20) | (small[1] << 10) | small[2];
}

struct Indexed {
	std::vector<float> pos, nrm, uv, tan;
	std::vector<uint32_t> tcomp, idx;
};

void index_mesh(const Soup& s, float tol, Indexed& m)
{
	std::vector<size_t> firstOf;
	weld(s, tol, m.idx, firstOf);
	const size_t V = firstOf.size();
	m.pos.resize(V * 3); m.nrm.resize(V * 3); m.uv.resize(V * 2);
	for (size_t v = 0; v < V; v++) {
		const size_t f = firstOf[v];
		for (int k = 0; k < 3; k++) { m.pos[v * 3 + k] = s.pos[f * 3 + k]; m.nrm[v * 3 + k] = s.nrm[f * 3 + k]; }
		for (int k = 0; k < 2; k++) m.uv[v * 2 + k] = s.uv[f * 2 + k];
	}
	std::vector<double> P(m.pos.begin(), m.pos.end()), UV(m.uv.begin(), m.uv.end()), N(m.nrm.begin(), m.nrm.end()), t4;
	tangent_space(m.idx, P, UV, N, t4);
	m.tan.resize(V * 4);
	for (size_t i = 0; i < V * 4; i++) m.tan[i] = (float)t4[i];
	m.tcomp.resize(V);
	for (size_t v = 0; v < V; v++) m.tcomp[v] = pack_tbn(&m.tan[v * 4], &m.nrm[v * 3]);
}

// ---------------------------------------------------------------------------
// OBJ / MTL
// ---------------------------------------------------------------------------
struct Material {
	std::string name;
	std::string baseColor, roughness, metalness, alphaMask, normalMap, emissive;
};
struct MeshInfo { std::string name; size_t material, first, count; };
struct Model {
	std::vector<Material> materials;
	std::vector<MeshInfo> meshes;
	std::vector<float> pos, nrm, uv; // triangle soup
};

inline void trim(std::string_view& s)
{
	while (!s.empty() && (s.front() == ' ' || s.front() == '\t' || s.front() == '\r')) s.remove_prefix(1);
	while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r' || s.back() == '\n')) s.remove_suffix(1);
}
inline std::string_view next_token(std::string_view& s)
{
	trim(s);
	size_t n = 0;
	while (n < s.size() && s[n] != ' ' && s[n] != '\t') n++;
	std::string_view t = s.substr(0, n);
	s.remove_prefix(n);
	return t;
}
inline bool parse_float(std::string_view t, float& out)
{
	if (!t.empty() && t.front() == '+') t.remove_prefix(1);
	auto r = std::from_chars(t.data(), t.data() + t.size(), out);
	return r.ec == std::errc() && r.ptr == t.data() + t.size();
}
inline bool parse_int(std::string_view t, long& out)
{
	auto r = std::from_chars(t.data(), t.data() + t.size(), out);
	return r.ec == std::errc() && r.ptr == t.data() + t.size();
}

// the file name of a texture statement: options ("-bm 1.0", "-o u v w", ...) come first
std::string texture_name(std::string_view rest)
{
	static const struct { const char* opt; int args; } kOpts[] = {
		{ "-blendu", 1 }, { "-blendv", 1 }, { "-boost", 1 }, { "-mm", 2 }, { "-o", 3 }, { "-s", 3 }, { "-t", 3 },
		{ "-texres", 1 }, { "-clamp", 1 }, { "-bm", 1 }, { "-imfchan", 1 }, { "-type", 1 } };
	for (;;) {
		trim(rest);
		bool took = false;
		for (auto& o : kOpts) {
			const size_t n = std::strlen(o.opt);
			if (rest.size() > n && rest.substr(0, n) == o.opt && (rest[n] == ' ' || rest[n] == '\t')) {
				rest.remove_prefix(n);
				for (int a = 0; a < o.args; a++) next_token(rest);
				took = true;
				break;
			}
		}
		if (!took) break;
	}
	trim(rest);
	return std::string(rest);
}

int load_mtl(const std::string& path, const std::string& prefix, std::vector<Material>& out)
{
	std::ifstream in(path);
	if (!in) return fail("cannot open material library '%s'", path.c_str());
	std::string line;
	while (std::getline(in, line)) {
		std::string_view s(line);
		const std::string_view key = next_token(s);
		if (key.empty() || key.front() == '#') continue;
		if (key == "newmtl") { trim(s); out.push_back({}); out.back().name = std::string(s); continue; }
		if (out.empty()) continue;
		Material& m = out.back();
		// load_model_obj.cpp:61-76: paths are prefixed with the OBJ's directory
		auto set = [&](std::string& dst) { const std::string n = texture_name(s); if (!n.empty()) dst = prefix + n; };
		if (key == "map_Kd") set(m.baseColor);
		else if (key == "map_Pr") set(m.roughness);
		else if (key == "map_Pm") set(m.metalness);
		else if (key == "map_d") set(m.alphaMask);
		else if (key == "norm") set(m.normalMap);
		else if (key == "map_Ke") set(m.emissive);
	}
	return 0;
}

struct Corner { long v, vt, vn; };

// ---------------------------------------------------------------------------
// Polygons with more than four corners.  rapidobj projects the face onto the
// coordinate plane in which it has the largest area and runs mapbox earcut on
// the 2-D ring (third_party/rapidobj/include/rapidobj/rapidobj.hpp:7430-7493);
// the triangles, and their order, decide the triangle soup and with it the
// vertex numbering of the baked mesh.  This is ear clipping by the same rules,
// on an index-linked ring: the ring is wound clockwise (earcut's convention for
// outer rings), a vertex is an ear if its corner is convex and no reflex-or-flat
// vertex lies inside its triangle, after a cut the walk skips the next vertex,
// and a ring without ears is first filtered (duplicate and collinear points),
// then cured of local self-intersections, then split along a valid diagonal.
// ---------------------------------------------------------------------------
class EarClipper {
public:
	// ring: 2-D points (floats widened to double, as earcut does); returns triples of ring indices (prev, ear, next)
	std::vector<uint32_t> run(const std::vector<std::array<float, 2>>& ring)
	{
		n_.clear(); out_.clear();
		const size_t len = ring.size();
		if (!len) return out_;
		double sum = 0;
		for (size_t i = 0, j = len - 1; i < len; j = i++)
			sum += ((double)ring[j][0] - (double)ring[i][0]) * ((double)ring[i][1] + (double)ring[j][1]);
		int last = -1;
		if (sum > 0) for (size_t i = 0; i < len; i++) last = insert((uint32_t)i, ring[i][0], ring[i][1], last);
		else for (size_t i = len; i-- > 0;) last = insert((uint32_t)i, ring[i][0], ring[i][1], last);
		if (last >= 0 && same(last, n_[last].next)) { unlink(last); last = n_[last].next; }
		if (last < 0 || n_[last].prev == n_[last].next) return out_;
		clip(last, 0);
		return out_;
	}

private:
	struct Node { uint32_t i; double x, y; int prev, next; };
	std::vector<Node> n_;
	std::vector<uint32_t> out_;

	int insert(uint32_t i, double x, double y, int last)
	{
		const int p = (int)n_.size();
		n_.push_back({ i, x, y, p, p });
		if (last >= 0) {
			n_[p].next = n_[last].next; n_[p].prev = last;
			n_[n_[last].next].prev = p; n_[last].next = p;
		}
		return p;
	}
	void unlink(int p) { n_[n_[p].next].prev = n_[p].prev; n_[n_[p].prev].next = n_[p].next; }
	bool same(int a, int b) const { return n_[a].x == n_[b].x && n_[a].y == n_[b].y; }
	double area(int p, int q, int r) const { return (n_[q].y - n_[p].y) * (n_[r].x - n_[q].x) - (n_[q].x - n_[p].x) * (n_[r].y - n_[q].y); }
	static bool in_triangle(double ax, double ay, double bx, double by, double cx, double cy, double px, double py)
	{
		return (cx - px) * (ay - py) >= (ax - px) * (cy - py) && (ax - px) * (by - py) >= (bx - px) * (ay - py) &&
		       (bx - px) * (cy - py) >= (cx - px) * (by - py);
	}
	int filter(int start, int end = -1)
	{
		if (end < 0) end = start;
		int p = start;
		bool again;
		do {
			again = false;
			if (same(p, n_[p].next) || area(n_[p].prev, p, n_[p].next) == 0) {
				unlink(p);
				p = end = n_[p].prev;
				if (p == n_[p].next) break;
				again = true;
			} else p = n_[p].next;
		} while (again || p != end);
		return end;
	}
	bool is_ear(int ear) const
	{
		const int a = n_[ear].prev, b = ear, c = n_[ear].next;
		if (area(a, b, c) >= 0) return false;
		for (int p = n_[c].next; p != a; p = n_[p].next)
			if (in_triangle(n_[a].x, n_[a].y, n_[b].x, n_[b].y, n_[c].x, n_[c].y, n_[p].x, n_[p].y) && area(n_[p].prev, p, n_[p].next) >= 0) return false;
		return true;
	}
	static int sgn(double v) { return (0.0 < v) - (v < 0.0); }
	bool on_segment(int p, int q, int r) const
	{
		return n_[q].x <= std::max(n_[p].x, n_[r].x) && n_[q].x >= std::min(n_[p].x, n_[r].x) && n_[q].y <= std::max(n_[p].y, n_[r].y) &&
		       n_[q].y >= std::min(n_[p].y, n_[r].y);
	}
	bool crosses(int p1, int q1, int p2, int q2) const
	{
		const int o1 = sgn(area(p1, q1, p2)), o2 = sgn(area(p1, q1, q2)), o3 = sgn(area(p2, q2, p1)), o4 = sgn(area(p2, q2, q1));
		if (o1 != o2 && o3 != o4) return true;
		if (o1 == 0 && on_segment(p1, p2, q1)) return true;
		if (o2 == 0 && on_segment(p1, q2, q1)) return true;
		if (o3 == 0 && on_segment(p2, p1, q2)) return true;
		if (o4 == 0 && on_segment(p2, q1, q2)) return true;
		return false;
	}
	bool locally_inside(int a, int b) const
	{
		return area(n_[a].prev, a, n_[a].next) < 0 ? area(a, b, n_[a].next) >= 0 && area(a, n_[a].prev, b) >= 0
		                                          : area(a, b, n_[a].prev) < 0 || area(a, n_[a].next, b) < 0;
	}
	bool crosses_ring(int a, int b) const
	{
		int p = a;
		do {
			const int q = n_[p].next;
			if (n_[p].i != n_[a].i && n_[q].i != n_[a].i && n_[p].i != n_[b].i && n_[q].i != n_[b].i && crosses(p, q, a, b)) return true;
			p = q;
		} while (p != a);
		return false;
	}
	bool middle_inside(int a, int b) const
	{
		bool inside = false;
		const double px = (n_[a].x + n_[b].x) / 2, py = (n_[a].y + n_[b].y) / 2;
		int p = a;
		do {
			const int q = n_[p].next;
			if (((n_[p].y > py) != (n_[q].y > py)) && n_[q].y != n_[p].y && (px < (n_[q].x - n_[p].x) * (py - n_[p].y) / (n_[q].y - n_[p].y) + n_[p].x))
				inside = !inside;
			p = q;
		} while (p != a);
		return inside;
	}
	bool valid_diagonal(int a, int b) const
	{
		return n_[n_[a].next].i != n_[b].i && n_[n_[a].prev].i != n_[b].i && !crosses_ring(a, b) &&
		       ((locally_inside(a, b) && locally_inside(b, a) && middle_inside(a, b) &&
		         (area(n_[a].prev, a, n_[b].prev) != 0.0 || area(a, n_[b].prev, b) != 0.0)) ||
		        (same(a, b) && area(n_[a].prev, a, n_[a].next) > 0 && area(n_[b].prev, b, n_[b].next) > 0));
	}
	int cure(int start)
	{
		int p = start;
		do {
			const int a = n_[p].prev, b = n_[n_[p].next].next;
			if (!same(a, b) && crosses(a, p, n_[p].next, b) && locally_inside(a, b) && locally_inside(b, a)) {
				out_.push_back(n_[a].i); out_.push_back(n_[p].i); out_.push_back(n_[b].i);
				unlink(p); unlink(n_[p].next);
				p = start = b;
			}
			p = n_[p].next;
		} while (p != start);
		return filter(p);
	}
	void split(int start)
	{
		int a = start;
		do {
			for (int b = n_[n_[a].next].next; b != n_[a].prev; b = n_[b].next) {
				if (n_[a].i != n_[b].i && valid_diagonal(a, b)) {
					// two rings: a .. b and a copy of b .. a copy of a
					const int a2 = (int)n_.size(); n_.push_back({ n_[a].i, n_[a].x, n_[a].y, 0, 0 });
					const int b2 = (int)n_.size(); n_.push_back({ n_[b].i, n_[b].x, n_[b].y, 0, 0 });
					const int an = n_[a].next, bp = n_[b].prev;
					n_[a].next = b; n_[b].prev = a;
					n_[a2].next = an; n_[an].prev = a2;
					n_[b2].next = a2; n_[a2].prev = b2;
					n_[bp].next = b2; n_[b2].prev = bp;
					const int ra = filter(a, n_[a].next), rc = filter(b2, n_[b2].next);
					clip(ra, 0);
					clip(rc, 0);
					return;
				}
			}
			a = n_[a].next;
		} while (a != start);
	}
	void clip(int ear, int pass)
	{
		if (ear < 0) return;
		int stop = ear;
		while (n_[ear].prev != n_[ear].next) {
			const int prev = n_[ear].prev, next = n_[ear].next;
			if (is_ear(ear)) {
				out_.push_back(n_[prev].i); out_.push_back(n_[ear].i); out_.push_back(n_[next].i);
				unlink(ear);
				ear = n_[next].next; // skipping a vertex gives fewer slivers
				stop = n_[next].next;
				continue;
			}
			ear = next;
			if (ear == stop) {
				if (pass == 0) clip(filter(ear), 1);
				else if (pass == 1) clip(cure(filter(ear)), 2);
				else split(ear);
				break;
			}
		}
	}
};

inline float polygon_area(const float* x, const float* y, size_t size)
{
	float area = 0.0f;
	for (size_t i = 1; i != size; ++i) {
		const float avgHeight = (y[i - 1] + y[i]) / 2;
		const float width = x[i] - x[i - 1];
		area += width * avgHeight;
	}
	const float avgHeight = (y[0] + y[size - 1]) / 2;
	const float width = x[0] - x[size - 1];
	area += width * avgHeight;
	return std::fabs(area);
}

int load_obj(const char* path, Model& model)
{
	// main-bake/zstdistream.cpp: the model source is a Zstandard stream; plain text is accepted as well
	std::vector<uint8_t> raw;
	{
		std::ifstream file(path, std::ios::binary);
		if (!file) return fail("cannot open '%s'", path);
		raw.assign(std::istreambuf_iterator<char>(file), std::istreambuf_iterator<char>());
	}
	if (vkr_zstd_is_frame(raw.data(), raw.size())) {
		std::vector<uint8_t> text;
		std::string err;
		if (!vkr_zstd_decompress(raw.data(), raw.size(), text, err)) return fail("%s: %s", path, err.c_str());
		raw.swap(text);
	}
	const char* slash = std::strrchr(path, '/');
	const std::string prefix = slash ? std::string(path, slash + 1) : std::string();

	std::vector<float> P, T, N;
	struct Face { Corner c[3]; long mat; };
	struct Shape { std::string name; std::vector<Face> faces; };
	std::vector<Shape> shapes(1);
	std::unordered_map<std::string, size_t> matIndex;
	long curMat = -1;
	bool haveLib = false;
	size_t lineNo = 0;
	for (size_t at = 0; at < raw.size();) {
		size_t eol = at;
		while (eol < raw.size() && raw[eol] != '\n') eol++;
		std::string_view s((const char*)raw.data() + at, eol - at);
		at = eol + 1;
		lineNo++;
		const std::string_view key = next_token(s);
		if (key.empty() || key.front() == '#') continue;
		if (key == "v" || key == "vn" || key == "vt") {
			const int want = key == "vt" ? 2 : 3;
			float f[3] = { 0, 0, 0 };
			for (int k = 0; k < want; k++)
				if (!parse_float(next_token(s), f[k])) return fail("%s:%zu: bad number", path, lineNo);
			std::vector<float>& dst = key == "v" ? P : key == "vn" ? N : T;
			dst.insert(dst.end(), f, f + want);
		} else if (key == "g" || key == "o") {
			trim(s);
			shapes.push_back({});
			shapes.back().name = std::string(s);
		} else if (key == "mtllib") {
			trim(s);
			if (!haveLib) {
				if (load_mtl(prefix + std::string(s), prefix, model.materials)) return 1;
				for (size_t i = 0; i < model.materials.size(); i++) matIndex.emplace(model.materials[i].name, i);
				haveLib = true;
			}
		} else if (key == "usemtl") {
			trim(s);
			auto it = matIndex.find(std::string(s));
			if (it == matIndex.end()) return fail("%s:%zu: material '%.*s' not found", path, lineNo, (int)s.size(), s.data());
			curMat = (long)it->second;
		} else if (key == "f") {
			std::vector<Corner> cs;
			for (;;) {
				std::string_view t = next_token(s);
				if (t.empty()) break;
				Corner c{ -1, -1, -1 };
				long* dst[3] = { &c.v, &c.vt, &c.vn };
				const long counts[3] = { (long)P.size() / 3, (long)T.size() / 2, (long)N.size() / 3 };
				for (int k = 0; k < 3; k++) { // v, v/vt, v//vn, v/vt/vn; negative = relative to the elements read so far
					size_t n = 0;
					while (n < t.size() && t[n] != '/') n++;
					if (n) {
						long v;
						if (!parse_int(t.substr(0, n), v) || v == 0) return fail("%s:%zu: bad face index", path, lineNo);
						*dst[k] = v > 0 ? v - 1 : counts[k] + v;
						if (*dst[k] < 0 || *dst[k] >= counts[k]) return fail("%s:%zu: face index out of range", path, lineNo);
					}
					if (n >= t.size()) break;
					t.remove_prefix(n + 1);
				}
				if (c.v < 0) return fail("%s:%zu: face corner without a position", path, lineNo);
				// the reference reads texcoords[idx*2] and normals[idx*3] unconditionally (load_model_obj.cpp:148-157)
				if (c.vt < 0 || c.vn < 0) return fail("%s:%zu: face corner without texture coordinate or normal", path, lineNo);
				cs.push_back(c);
			}
			if (cs.size() < 3) return fail("%s:%zu: face with fewer than three corners", path, lineNo);
			if (curMat < 0) return fail("%s:%zu: face without a material", path, lineNo);
			auto emit = [&](const Corner& a, const Corner& b, const Corner& c) { shapes.back().faces.push_back({ { a, b, c }, curMat }); };
			if (cs.size() == 3) emit(cs[0], cs[1], cs[2]);
			else if (cs.size() == 4) {
				// rapidobj: split along the shorter diagonal (rapidobj.hpp:7365-7426)
				auto d2 = [&](const Corner& a, const Corner& b) {
					const float ex = P[a.v * 3 + 0] - P[b.v * 3 + 0], ey = P[a.v * 3 + 1] - P[b.v * 3 + 1], ez = P[a.v * 3 + 2] - P[b.v * 3 + 2];
					return ex * ex + ey * ey + ez * ez;
				};
				const bool d02less = d2(cs[0], cs[2]) < d2(cs[1], cs[3]);
				emit(cs[0], cs[1], d02less ? cs[2] : cs[3]);
				emit(d02less ? cs[0] : cs[1], cs[2], cs[3]);
			} else {
				// rapidobj.hpp:7430-7493: the coordinate plane of the largest projected area, earcut, first two corners of
				// every triangle swapped
				if (cs.size() > 255) return fail("%s:%zu: face with more than 255 corners", path, lineNo);
				const size_t nv = cs.size();
				std::vector<float> xs(nv), ys(nv), zs(nv);
				for (size_t k = 0; k < nv; k++) { xs[k] = P[cs[k].v * 3 + 0]; ys[k] = P[cs[k].v * 3 + 1]; zs[k] = P[cs[k].v * 3 + 2]; }
				const float areaX = polygon_area(ys.data(), zs.data(), nv), areaY = polygon_area(xs.data(), zs.data(), nv),
				            areaZ = polygon_area(xs.data(), ys.data(), nv);
				if (FLT_MIN > std::max({ areaX, areaY, areaZ })) return fail("%s:%zu: degenerate polygon", path, lineNo);
				const int plane = areaX > areaY ? (areaX > areaZ ? 0 : 2) : (areaY > areaZ ? 1 : 2);
				std::vector<std::array<float, 2>> ring(nv);
				for (size_t k = 0; k < nv; k++)
					ring[k] = plane == 0 ? std::array<float, 2>{ ys[k], zs[k] } : plane == 1 ? std::array<float, 2>{ xs[k], zs[k] } : std::array<float, 2>{ xs[k], ys[k] };
				EarClipper clipper;
				const std::vector<uint32_t> tri = clipper.run(ring);
				if (tri.empty() || tri.size() % 3) return fail("%s:%zu: polygon could not be triangulated", path, lineNo);
				for (size_t k = 0; k < tri.size(); k += 3) emit(cs[tri[k + 1]], cs[tri[k]], cs[tri[k + 2]]);
			}
		}
	}
	// load_model_obj.cpp:95-169: per shape, one mesh per material used in it
	for (const Shape& sh : shapes) {
		// The reference collects a shape's materials in a std::unordered_set<std::size_t> - one emplace per index of
		// every triangle - and emits the shape's meshes in the set's iteration order (load_model_obj.cpp:94-120).  The
		// same container with the same sequence of insertions gives the same order under the same standard library.
		std::unordered_set<std::size_t> active;
		for (const auto& f : sh.faces)
			for (int k = 0; k < 3; k++) active.emplace((std::size_t)f.mat);
		for (const std::size_t matId : active) {
			const long mat = (long)matId;
			MeshInfo mi;
			mi.name = active.size() == 1 ? sh.name : sh.name + "::" + model.materials[mat].name;
			mi.material = (size_t)mat;
			mi.first = model.pos.size() / 3;
			for (const auto& f : sh.faces) {
				if (f.mat != mat) continue;
				for (const Corner& c : f.c) {
					model.pos.insert(model.pos.end(), &P[c.v * 3], &P[c.v * 3] + 3);
					model.uv.insert(model.uv.end(), &T[c.vt * 2], &T[c.vt * 2] + 2);
					model.nrm.insert(model.nrm.end(), &N[c.vn * 3], &N[c.vn * 3] + 3);
				}
			}
			mi.count = model.pos.size() / 3 - mi.first;
			model.meshes.push_back(std::move(mi));
		}
	}
	return 0;
}

// ---------------------------------------------------------------------------
// textures and the file
// ---------------------------------------------------------------------------
struct TexEntry { std::string srcPath, newPath; uint8_t space, channels; };

void write_u32(FILE* f, uint32_t v) { std::fwrite(&v, 4, 1, f); }
void write_string(FILE* f, const std::string& s)
{
	write_u32(f, (uint32_t)s.size() + 1);
	std::fwrite(s.c_str(), 1, s.size() + 1, f);
}

} // namespace

extern "C" const char* vkr_bake_last_error(void) { return g_err; }

static int index_mesh_entry(const float* positions, const float* normals, const float* texcoords, uint32_t n, float tolerance,
                              vkr_indexed_mesh* out)
{
	if (!positions || !normals || !texcoords || !out) return fail("vkr_index_mesh: NULL argument");
	if (n % 3) return fail("vkr_index_mesh: %u soup vertices is not a whole number of triangles", n);
	std::memset(out, 0, sizeof *out);
	Indexed m;
	index_mesh(Soup{ positions, normals, texcoords, n }, tolerance, m);
	auto dup = [](const auto& v) {
		using T = typename std::decay_t<decltype(v)>::value_type;
		T* p = (T*)std::malloc(std::max<size_t>(1, v.size()) * sizeof(T));
		if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
		return p;
	};
	out->vertexCount = (uint32_t)m.tcomp.size(); out->indexCount = (uint32_t)m.idx.size();
	out->positions = dup(m.pos); out->normals = dup(m.nrm); out->texcoords = dup(m.uv); out->tangents = dup(m.tan);
	out->tangentsComp = dup(m.tcomp); out->indices = dup(m.idx);
	if (!out->positions || !out->normals || !out->texcoords || !out->tangents || !out->tangentsComp || !out->indices) {
		vkr_indexed_mesh_free(out);
		return fail("vkr_index_mesh: out of memory");
	}
	return 0;
}

extern "C" void vkr_indexed_mesh_free(vkr_indexed_mesh* m)
{
	if (!m) return;
	std::free(m->positions); std::free(m->normals); std::free(m->texcoords); std::free(m->tangents);
	std::free(m->tangentsComp); std::free(m->indices);
	std::memset(m, 0, sizeof *m);
}

static int bake_obj_entry(const char* obj_path, const char* out_path, int copy_textures)
{
	namespace fs = std::filesystem;
	if (!obj_path || !out_path) return fail("vkr_bake_obj: NULL path");
	Model model;
	if (load_obj(obj_path, model)) return 1;
	// normalize_, main-bake/main.cpp:236-251
	for (Material& m : model.materials) {
		if (m.baseColor.empty()) m.baseColor = "assets-src/main/rgba1111.png";
		if (m.roughness.empty()) m.roughness = "assets-src/main/r1.png";
		if (m.metalness.empty()) m.metalness = "assets-src/main/r1.png";
		if (m.emissive.empty()) m.emissive = "assets-src/main/rgb000.png";
	}
	// index_meshes_, main-bake/main.cpp:394-421 (tolerance 1e-5, main.cpp:75-78)
	std::vector<Indexed> indexed(model.meshes.size());
	for (size_t i = 0; i < model.meshes.size(); i++) {
		const MeshInfo& mi = model.meshes[i];
		index_mesh(Soup{ &model.pos[mi.first * 3], &model.nrm[mi.first * 3], &model.uv[mi.first * 2], mi.count }, 1e-5f, indexed[i]);
	}
	// find_unique_textures_ + new_paths_, main-bake/main.cpp:426-476: ids in order of first use
	const fs::path outname(out_path);
	const fs::path rootdir = outname.parent_path();
	const fs::path texdir = outname.stem().string() + "-tex";
	std::vector<TexEntry> textures;
	std::unordered_map<std::string, uint32_t> texId;
	auto add = [&](const std::string& p, uint8_t space, uint8_t channels) {
		if (p.empty() || texId.count(p)) return;
		texId.emplace(p, (uint32_t)textures.size());
		textures.push_back({ p, (texdir / fs::path(p).filename()).string(), space, channels });
	};
	for (const Material& m : model.materials) {
		add(m.baseColor, 1, 4); add(m.roughness, 0, 1); add(m.metalness, 0, 1);
		add(m.alphaMask, 1, 4); add(m.normalMap, 0, 3); add(m.emissive, 1, 4);
	}
	std::error_code ec;
	if (!rootdir.empty()) fs::create_directories(rootdir, ec);
	fs::path mainpath = rootdir / outname.stem();
	mainpath.replace_extension("comp5892mesh");
	struct File { FILE* f; ~File() { if (f) std::fclose(f); } } file{ std::fopen(mainpath.string().c_str(), "wb") };
	FILE* f = file.f;
	if (!f) return fail("unable to open '%s' for writing", mainpath.string().c_str());
	// write_model_data_, main-bake/main.cpp:276-389
	static const char kMagic[16] = "\0\0COMP5892Mmesh";
	char variant[16] = "21-tan";
	std::fwrite(kMagic, 1, 16, f);
	std::fwrite(variant, 1, 16, f);
	write_u32(f, (uint32_t)textures.size());
	for (const TexEntry& t : textures) {
		write_string(f, t.newPath);
		std::fwrite(&t.space, 1, 1, f);
		std::fwrite(&t.channels, 1, 1, f);
	}
	write_u32(f, (uint32_t)model.materials.size());
	for (const Material& m : model.materials)
		for (const std::string* p : { &m.baseColor, &m.roughness, &m.metalness, &m.alphaMask, &m.normalMap, &m.emissive })
			write_u32(f, p->empty() ? 0xffffffffu : texId.at(*p));
	write_u32(f, (uint32_t)model.meshes.size());
	for (size_t i = 0; i < model.meshes.size(); i++) {
		const Indexed& m = indexed[i];
		write_u32(f, (uint32_t)model.meshes[i].material);
		write_u32(f, (uint32_t)m.tcomp.size());
		write_u32(f, (uint32_t)m.idx.size());
		std::fwrite(m.pos.data(), 4, m.pos.size(), f);
		std::fwrite(m.nrm.data(), 4, m.nrm.size(), f);
		std::fwrite(m.uv.data(), 4, m.uv.size(), f);
		std::fwrite(m.tan.data(), 4, m.tan.size(), f);
		std::fwrite(m.tcomp.data(), 4, m.tcomp.size(), f);
		std::fwrite(m.idx.data(), 4, m.idx.size(), f);
	}
	const bool bad = std::ferror(f) != 0;
	file.f = nullptr;
	if (std::fclose(f) != 0 || bad) return fail("writing '%s' failed", mainpath.string().c_str());
	if (copy_textures) {
		fs::create_directories(rootdir / texdir, ec);
		for (const TexEntry& t : textures) fs::copy_file(t.srcPath, rootdir / t.newPath, fs::copy_options::none, ec); // existing files are kept
	}
	return 0;
}

// No exception may leave the library (std::bad_alloc, std::filesystem errors, ...).
template <class F> static int guarded(const char* fn, F&& body) noexcept
{
	try { return body(); }
	catch (const std::bad_alloc&) { return fail("%s: out of host memory", fn); }
	catch (const std::exception& e) { return fail("%s: %s", fn, e.what()); }
	catch (...) { return fail("%s: unknown exception", fn); }
}

extern "C" int vkr_index_mesh(const float* positions, const float* normals, const float* texcoords, uint32_t n, float tolerance,
                              vkr_indexed_mesh* out)
{
	return guarded("vkr_index_mesh", [&] { return index_mesh_entry(positions, normals, texcoords, n, tolerance, out); });
}

extern "C" int vkr_zstd_decode(const uint8_t* src, size_t len, uint8_t** out, size_t* out_len)
{
	return guarded("vkr_zstd_decode", [&] {
		if (!src || !out || !out_len) return fail("vkr_zstd_decode: NULL argument");
		*out = nullptr; *out_len = 0;
		std::vector<uint8_t> data;
		std::string err;
		if (!vkr_zstd_decompress(src, len, data, err)) return fail("%s", err.c_str());
		uint8_t* p = (uint8_t*)std::malloc(data.size() ? data.size() : 1);
		if (!p) return fail("vkr_zstd_decode: out of memory");
		if (!data.empty()) std::memcpy(p, data.data(), data.size());
		*out = p; *out_len = data.size();
		return 0;
	});
}

extern "C" void vkr_bake_free(void* p) { std::free(p); }

extern "C" int vkr_bake_obj(const char* obj_path, const char* out_path, int copy_textures)
{
	return guarded("vkr_bake_obj", [&] { return bake_obj_entry(obj_path, out_path, copy_textures); });
}
