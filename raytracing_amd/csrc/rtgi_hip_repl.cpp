// rtgi_hip_repl -- the renderer's command surface on top of the MI355X backend: reads the commands of the reference's
// driver scripts (driver/interaction.cpp:57-406 of "Assignment 3/rtgi-2021-a03-ref/"; scripts/a1-tri, a2-tris,
// a3-brdf-test, a3-sibenik) from a script file and/or stdin, keeps camera, scene, materials, light and settings, and
// runs `commit` / `run` / `rt_bench` through the frame level (include/rtgi_hip_frame.h): rays, hits and the
// framebuffer stay on the GPUs, the image is written as a PNG.
//
//   rtgi_hip_repl [-s script | -l script] [-o out.png] [--dry-run] [--strict]
//     -s file    read commands from file instead of stdin            (driver/cmdline.cpp:60)
//     -l file    read commands from file, then from stdin            (driver/cmdline.cpp:61)
//     -o file    image file, overrides the script's `outfile`        (driver/cmdline.cpp:62, header comment :4)
//     --dry-run  parse and check every command, touch no GPU (no tracer is created, `commit`/`run` only check order)
//     --strict   exit status 1 if any command was refused (the reference prints the error and goes on; so does this)
//
// `raytracer hip [indexed] [esc] [devices N | devices a,b,...]` selects this backend; the reference's own names
// (`seq`, `naive-bvh`, `bbvh [indexed] [esc]`) are accepted and create the same GPU tracer with the layout asked for,
// so that the reference's scripts run unchanged. Host-side only: OBJ/MTL reading, PNG textures, the material table and
// the PNG file (the reference uses assimp, MagickWand and png++, none of which this program needs; zlib inflates).
#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include <zlib.h>

#include "rtgi_hip_frame.h"

namespace {

struct vec3 {
	float x = 0, y = 0, z = 0;
};
std::istream &operator>>(std::istream &in, vec3 &v) { return in >> v.x >> v.y >> v.z; }
std::ostream &operator<<(std::ostream &out, const vec3 &v) { return out << "vec3(" << v.x << ", " << v.y << ", " << v.z << ")"; }

// struct vertex / triangle / material of libgi/rt.h:29-38 and libgi/material.h:54-61
struct Vertex {
	float pos[3], norm[3], tc[2];
};
struct Triangle {
	uint32_t a, b, c, material_id;
};
struct Material {
	std::string name;
	vec3 albedo, emissive;
	float ior = 1.3f, roughness = 0.1f;
	uint32_t brdf = RTGI_BRDF_LAMBERT;
	int texture = -1;          // index into Scene::textures, -1: none (`material texture drop`)
	std::string texture_path;  // as named by the material library or the `material texture` command
};
// struct texture, libgi/scene.h:16-32
struct Texture {
	std::string path;
	unsigned w = 0, h = 0;
	std::vector<float> texel;  // rgb, texel[3 * (y * w + x)]
};
struct Object {
	std::string name;
	unsigned start, end, material_id;
};

const char *const BRDF_NAMES[] = {"lambert", "phong", "layered-phong", "gtr2", "layered-gtr2"};  // new_brdf, libgi/material.cpp:77-102
int brdf_by_name(const std::string &n) {
	for (int k = 0; k < 5; ++k)
		if (n == BRDF_NAMES[k]) return k;
	return -1;
}

struct Scene {
	std::vector<Texture> textures;
	std::vector<Vertex> vertices;
	std::vector<Triangle> triangles;
	std::vector<Material> materials;
	std::vector<Object> objects;
	uint32_t default_brdf = RTGI_BRDF_LAMBERT;
	bool default_brdf_set = false;
	bool have_light = false;
	vec3 light_pos, light_col;
	vec3 up{0, 1, 0};
	// camera(vec3(0,0,-1), vec3(0,0,0), vec3(0,1,0), 65, 1280, 720), libgi/scene.h:94
	vec3 cam_dir{0, 0, -1}, cam_pos{0, 0, 0}, cam_up{0, 1, 0};
	float fovy = 65.0f;
	int w = 1280, h = 720;
};

float luma(const vec3 &c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }  // libgi/color.h

std::string dir_of(const std::string &path) {
	const size_t k = path.find_last_of('/');
	return k == std::string::npos ? std::string() : path.substr(0, k + 1);
}

// ---- PNG reading (what the reference does through MagickWand): 8- or 16-bit grey, rgb, palette, with or without
// alpha, not interlaced; zlib does the inflating ----
bool read_png(const std::string &path, unsigned &w, unsigned &h, std::vector<float> &rgb01, std::string &err) {
	std::ifstream in(path, std::ios::binary);
	if (!in) return err = "cannot open " + path, false;
	std::vector<unsigned char> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
	static const unsigned char magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
	if (file.size() < 8 || memcmp(file.data(), magic, 8) != 0) return err = path + " is not a PNG file (the only image format read here)", false;
	auto be32 = [&](size_t at) { return ((uint32_t)file[at] << 24) | ((uint32_t)file[at + 1] << 16) | ((uint32_t)file[at + 2] << 8) | file[at + 3]; };
	unsigned depth = 0, type = 0, interlace = 0;
	std::vector<unsigned char> idat, palette;
	for (size_t at = 8; at + 12 <= file.size();) {
		const uint32_t len = be32(at);
		if (at + 12 + (size_t)len > file.size()) return err = path + ": truncated chunk", false;
		const std::string kind((const char *)&file[at + 4], 4);
		const unsigned char *data = &file[at + 8];
		if (kind == "IHDR" && len >= 13) {
			w = be32(at + 8), h = be32(at + 12);
			depth = data[8], type = data[9], interlace = data[12];
		} else if (kind == "PLTE")
			palette.assign(data, data + len);
		else if (kind == "IDAT")
			idat.insert(idat.end(), data, data + len);
		else if (kind == "IEND")
			break;
		at += 12 + (size_t)len;
	}
	const unsigned channels = type == 0 ? 1 : type == 2 ? 3 : type == 3 ? 1 : type == 4 ? 2 : type == 6 ? 4 : 0;
	if (!w || !h || w > 32768 || h > 32768 || (uint64_t)w * h > (1ull << 26) || !channels || interlace || (depth != 8 && depth != 16) || (type == 3 && depth != 8))
		return err = path + ": unsupported kind of PNG (8 or 16 bits per channel, not interlaced)", false;
	const size_t bpp = channels * depth / 8, stride = (size_t)w * bpp;
	std::vector<unsigned char> raw((stride + 1) * h);
	uLongf got = (uLongf)raw.size();
	if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) return err = path + ": damaged image data", false;
	std::vector<unsigned char> prev(stride, 0), row(stride);
	rgb01.assign((size_t)w * h * 3, 0.0f);
	for (unsigned y = 0; y < h; ++y) {
		const unsigned char *src = &raw[(stride + 1) * y];
		const unsigned filter = src[0];
		for (size_t i = 0; i < stride; ++i) {
			const int a = i >= bpp ? row[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
			int v = src[1 + i];
			if (filter == 1) v += a;
			else if (filter == 2) v += b;
			else if (filter == 3) v += (a + b) / 2;
			else if (filter == 4) {
				const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
				v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
			} else if (filter != 0)
				return err = path + ": damaged image data", false;
			row[i] = (unsigned char)v;
		}
		for (unsigned x = 0; x < w; ++x) {
			float *o = &rgb01[((size_t)y * w + x) * 3];
			auto sample = [&](unsigned ch) { return depth == 8 ? row[x * bpp + ch] / 255.0 : (row[x * bpp + 2 * ch] * 256 + row[x * bpp + 2 * ch + 1]) / 65535.0; };
			if (type == 3) {
				const size_t k = (size_t)row[x] * 3;
				for (int ch = 0; ch < 3; ++ch) o[ch] = k + 2 < palette.size() ? (float)(palette[k + ch] / 255.0) : 0.0f;
			} else if (channels <= 2)
				o[0] = o[1] = o[2] = (float)sample(0);
			else
				for (unsigned ch = 0; ch < 3; ++ch) o[ch] = (float)sample(ch);
		}
		prev = row;
	}
	return true;
}

// load_image3f, libgi/scene.cpp:45-68: rows flipped (MagickFlipImage), values to the power 2.2. Returns the index of
// the texture in `textures` (a file already loaded is shared), or -1 with `err` set.
int load_image3f(std::vector<Texture> &textures, const std::string &path, std::string &err) {
	for (size_t i = 0; i < textures.size(); ++i)
		if (textures[i].path == path) return (int)i;
	Texture t;
	std::vector<float> img;
	if (!read_png(path, t.w, t.h, img, err)) return -1;
	t.path = path;
	t.texel.resize(img.size());
	for (unsigned y = 0; y < t.h; ++y)
		for (size_t i = 0; i < (size_t)t.w * 3; ++i) t.texel[(size_t)y * t.w * 3 + i] = std::pow(img[(size_t)(t.h - 1 - y) * t.w * 3 + i], 2.2f);
	textures.push_back(std::move(t));
	return (int)textures.size() - 1;
}

// The material library of an OBJ file: what scene::add reads through assimp (libgi/scene.cpp:104-139): Kd or, if it
// is black, Ks as albedo, to the power 2.2; Ke; Ns as roughness (roughness_from_exponent, libgi/material.h:12-14);
// Ni as ior with 1.0 replaced by 1.3.
void load_mtl(const std::string &path, const std::string &prefix, uint32_t default_brdf, std::vector<Material> &out, std::map<std::string, uint32_t> &ids,
              std::vector<Texture> &textures) {
	std::ifstream in(path);
	if (!in) {
		std::cerr << "WARN: cannot open material library " << path << std::endl;
		return;
	}
	struct Raw {
		std::string name, tex;
		vec3 kd, ks, ke;
		bool has_ns = false, has_ni = false;
		float ns = 0, ni = 0;
	};
	std::vector<Raw> raws;
	std::string line;
	while (std::getline(in, line)) {
		std::istringstream ls(line);
		std::string key;
		ls >> key;
		if (key.size() == 2 && (key[0] == 'k' || key[0] == 'n')) key[0] = (char)std::toupper(key[0]);  // assimp reads "kd" like "Kd"
		if (key == "newmtl") {
			raws.emplace_back();
			ls >> raws.back().name;
		} else if (raws.empty())
			continue;
		else if (key == "Kd")
			ls >> raws.back().kd;
		else if (key == "Ks")
			ls >> raws.back().ks;
		else if (key == "Ke")
			ls >> raws.back().ke;
		else if (key == "Ns")
			raws.back().has_ns = bool(ls >> raws.back().ns);
		else if (key == "Ni")
			raws.back().has_ni = bool(ls >> raws.back().ni);
		else if (key == "map_Kd")
			ls >> raws.back().tex;
	}
	for (const Raw &r : raws) {
		Material m;
		m.name = prefix.empty() ? r.name : prefix + "/" + r.name;
		if (r.has_ns) m.roughness = std::sqrt(2.f / (r.ns + 2.f));
		if (r.has_ni) m.ior = r.ni;
		if (m.ior == 1.0f) m.ior = 1.3f;
		const vec3 base = luma(r.kd) > 1e-4f ? r.kd : r.ks;
		m.albedo = {std::pow(base.x, 2.2f), std::pow(base.y, 2.2f), std::pow(base.z, 2.2f)};
		m.emissive = r.ke;
		m.brdf = default_brdf;
		if (!r.tex.empty()) {  // libgi/scene.cpp:128-134 (there a texture that cannot be read ends the program)
			m.texture_path = dir_of(path) + r.tex;
			std::string err;
			m.texture = load_image3f(textures, m.texture_path, err);
			if (m.texture < 0) std::cerr << "note: texture of material " << m.name << " not loaded (" << err << "): its albedo colour is used" << std::endl;
		}
		ids[r.name] = (uint32_t)out.size();
		out.push_back(m);
	}
}

// scene::add (libgi/scene.cpp:88-175) for Wavefront OBJ files: polygons are cut into triangle fans, every triangle
// corner becomes a vertex of its own (assimp without vertex joining), corners without a normal get the face normal
// (aiProcess_GenNormals), texture coordinates default to (0, 0). Returns an error text or "".
std::string load_obj(Scene &scene, const std::string &path, const std::string &name) {
	std::ifstream in(path);
	if (!in) return "ERROR: Failed to load file: " + path + "!";
	std::vector<vec3> v, vn;
	std::vector<std::pair<float, float>> vt;
	std::map<std::string, uint32_t> mtl_ids;
	const uint32_t material_offset = (uint32_t)scene.materials.size();
	int32_t current = -1, fallback = -1;
	std::string line, object_name = path;
	auto open_object = [&](uint32_t material_id) {
		if (scene.objects.empty() || scene.objects.back().material_id != material_id || scene.objects.back().name != object_name)
			scene.objects.push_back({object_name, (unsigned)scene.triangles.size(), (unsigned)scene.triangles.size(), material_id});
	};
	size_t line_no = 0;
	while (std::getline(in, line)) {
		++line_no;
		std::istringstream ls(line);
		std::string key;
		ls >> key;
		if (key == "v") {
			vec3 p;
			if (!(ls >> p)) return path + ":" + std::to_string(line_no) + ": malformed vertex";
			v.push_back(p);
		} else if (key == "vn") {
			vec3 n;
			ls >> n;
			vn.push_back(n);
		} else if (key == "vt") {
			float s = 0, t = 0;
			ls >> s >> t;
			vt.push_back({s, t});
		} else if (key == "mtllib") {
			std::string file;
			ls >> file;
			load_mtl(dir_of(path) + file, name, scene.default_brdf, scene.materials, mtl_ids, scene.textures);
		} else if (key == "usemtl") {
			std::string m;
			ls >> m;
			auto it = mtl_ids.find(m);
			if (it == mtl_ids.end()) {
				std::cerr << "WARN: " << path << " uses the undefined material " << m << std::endl;
				current = -1;
			} else
				current = (int32_t)it->second;
		} else if (key == "o" || key == "g") {
			ls >> object_name;
		} else if (key == "f") {
			struct Corner {
				int v, t, n;
			};
			std::vector<Corner> corners;
			std::string tok;
			while (ls >> tok) {
				Corner c{0, 0, 0};
				const size_t s1 = tok.find('/');
				c.v = std::atoi(tok.substr(0, s1).c_str());
				if (s1 != std::string::npos) {
					const size_t s2 = tok.find('/', s1 + 1);
					const std::string t = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
					if (!t.empty()) c.t = std::atoi(t.c_str());
					if (s2 != std::string::npos && s2 + 1 < tok.size()) c.n = std::atoi(tok.substr(s2 + 1).c_str());
				}
				auto resolve = [](int i, size_t n) { return i > 0 ? i - 1 : (i < 0 ? (int)n + i : -1); };
				c.v = resolve(c.v, v.size()), c.t = resolve(c.t, vt.size()), c.n = resolve(c.n, vn.size());
				if (c.v < 0 || c.v >= (int)v.size()) return path + ":" + std::to_string(line_no) + ": vertex index out of range";
				if (c.t >= (int)vt.size()) c.t = -1;
				if (c.n >= (int)vn.size()) c.n = -1;
				corners.push_back(c);
			}
			if (corners.size() < 3) continue;
			uint32_t material_id;
			if (current >= 0)
				material_id = (uint32_t)current;
			else {
				if (fallback < 0) {  // assimp's "DefaultMaterial" for faces without one (grey, libgi/scene.cpp:115-125)
					Material m;
					m.name = name.empty() ? "DefaultMaterial" : name + "/DefaultMaterial";
					const float g = std::pow(0.6f, 2.2f);
					m.albedo = {g, g, g};
					m.brdf = scene.default_brdf;
					fallback = (int32_t)scene.materials.size();
					scene.materials.push_back(m);
				}
				material_id = (uint32_t)fallback;
			}
			open_object(material_id);
			for (size_t k = 1; k + 1 < corners.size(); ++k) {
				const Corner tri[3] = {corners[0], corners[k], corners[k + 1]};
				const vec3 &A = v[(size_t)tri[0].v], &B = v[(size_t)tri[1].v], &C = v[(size_t)tri[2].v];
				const float e1[3] = {B.x - A.x, B.y - A.y, B.z - A.z}, e2[3] = {C.x - A.x, C.y - A.y, C.z - A.z};
				float fn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
				const float len = std::sqrt(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]);
				if (len > 0) fn[0] /= len, fn[1] /= len, fn[2] /= len;
				const uint32_t base = (uint32_t)scene.vertices.size();
				for (const Corner &c : tri) {
					Vertex vx;
					const vec3 &p = v[(size_t)c.v];
					vx.pos[0] = p.x, vx.pos[1] = p.y, vx.pos[2] = p.z;
					if (c.n >= 0)
						vx.norm[0] = vn[(size_t)c.n].x, vx.norm[1] = vn[(size_t)c.n].y, vx.norm[2] = vn[(size_t)c.n].z;
					else
						vx.norm[0] = fn[0], vx.norm[1] = fn[1], vx.norm[2] = fn[2];
					vx.tc[0] = c.t >= 0 ? vt[(size_t)c.t].first : 0.f, vx.tc[1] = c.t >= 0 ? vt[(size_t)c.t].second : 0.f;
					scene.vertices.push_back(vx);
				}
				scene.triangles.push_back({base, base + 1, base + 2, material_id});
				scene.objects.back().end = (unsigned)scene.triangles.size();
			}
		}
	}
	(void)material_offset;
	return "";
}

// ---- PNG (8-bit rgb, stored deflate blocks: no compression library needed) ----
uint32_t crc32_of(const unsigned char *p, size_t n, uint32_t crc = 0) {
	static uint32_t table[256];
	static bool ready = false;
	if (!ready) {
		for (uint32_t i = 0; i < 256; ++i) {
			uint32_t c = i;
			for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
			table[i] = c;
		}
		ready = true;
	}
	crc = ~crc;
	for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
	return ~crc;
}
void put32(std::vector<unsigned char> &b, uint32_t v) {
	for (int s = 24; s >= 0; s -= 8) b.push_back((unsigned char)(v >> s));
}
void chunk(std::vector<unsigned char> &file, const char *type, const std::vector<unsigned char> &data) {
	put32(file, (uint32_t)data.size());
	const size_t start = file.size();
	file.insert(file.end(), type, type + 4);
	file.insert(file.end(), data.begin(), data.end());
	put32(file, crc32_of(&file[start], file.size() - start));
}
bool write_png(const std::string &path, const uint8_t *rgb, int w, int h) {
	std::vector<unsigned char> raw;
	raw.reserve((size_t)h * (3 * (size_t)w + 1));
	for (int y = 0; y < h; ++y) {
		raw.push_back(0);  // filter: none
		raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
	}
	std::vector<unsigned char> z = {0x78, 0x01};
	uint32_t a = 1, b = 0;
	for (unsigned char c : raw) a = (a + c) % 65521u, b = (b + a) % 65521u;
	for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
		const size_t len = std::min<size_t>(65535, raw.size() - off);
		z.push_back(off + len >= raw.size() ? 1 : 0);
		z.push_back((unsigned char)(len & 0xff)), z.push_back((unsigned char)(len >> 8));
		z.push_back((unsigned char)(~len & 0xff)), z.push_back((unsigned char)((~len >> 8) & 0xff));
		z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + len));
		if (raw.empty()) break;
	}
	put32(z, (b << 16) | a);
	std::vector<unsigned char> file = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'}, ihdr;
	put32(ihdr, (uint32_t)w), put32(ihdr, (uint32_t)h);
	const unsigned char rest[5] = {8, 2, 0, 0, 0};
	ihdr.insert(ihdr.end(), rest, rest + 5);
	chunk(file, "IHDR", ihdr);
	chunk(file, "IDAT", z);
	chunk(file, "IEND", {});
	std::ofstream out(path, std::ios::binary);
	out.write((const char *)file.data(), (std::streamsize)file.size());
	return bool(out);
}

// ---- the interpreter ----
struct Options {
	std::string script, outfile = "out.png";
	bool interact = true, outfile_from_cmdline = false, dry_run = false, strict = false;
};

struct Line;

struct Repl {
	Options opt;
	Scene scene;
	rtgi_hip_renderer *renderer = nullptr;
	std::string tracer_desc;
	int algo = -1;  // RTGI_ALGO_*
	unsigned sppx = 1, bounces = 3, seed = 0, jitter = 1, sharding = RTGI_SHARD_TILES, tile = 0;
	// what `run` checks (repl_update_checks, driver/interaction.h)
	int cmdid = 0, scene_touched_at = 0, tracer_touched_at = 0, accel_touched_at = 0;
	int errors = 0;
	bool textures_pushed = false;
	std::map<std::string, std::pair<double, unsigned>> timers;  // `stats print`

	~Repl() {
		if (renderer) rtgi_hip_renderer_destroy(renderer);
	}

	void push_materials() {
		if (!renderer || accel_touched_at == 0 || scene.materials.empty()) return;
		std::vector<rtgi_material> m(scene.materials.size());
		for (size_t i = 0; i < m.size(); ++i) {
			const Material &s = scene.materials[i];
			m[i] = {{s.albedo.x, s.albedo.y, s.albedo.z}, s.ior, s.roughness, s.brdf};
		}
		if (rtgi_hip_renderer_set_materials(renderer, m.data(), (uint32_t)m.size()) < 0)
			std::cerr << "materials: " << rtgi_hip_renderer_last_error(renderer) << std::endl;
		// the textures in use, and which material uses which
		std::vector<rtgi_texture> tex;
		std::vector<int32_t> of_material(m.size(), -1), slot(scene.textures.size(), -1);
		for (size_t i = 0; i < m.size(); ++i) {
			const int t = scene.materials[i].texture;
			if (t < 0) continue;
			if (slot[(size_t)t] < 0) {
				slot[(size_t)t] = (int32_t)tex.size();
				tex.push_back({scene.textures[(size_t)t].texel.data(), scene.textures[(size_t)t].w, scene.textures[(size_t)t].h});
			}
			of_material[i] = slot[(size_t)t];
		}
		if (!tex.empty() || textures_pushed) {
			if (rtgi_hip_renderer_set_textures(renderer, scene.vertices[0].tc, sizeof(Vertex), (uint32_t)scene.vertices.size(), scene.triangles.data(),
			                                   sizeof(Triangle), (uint32_t)scene.triangles.size(), tex.data(), (uint32_t)tex.size(), of_material.data(),
			                                   (uint32_t)of_material.size()) < 0)
				std::cerr << "textures: " << rtgi_hip_renderer_last_error(renderer) << std::endl;
			textures_pushed = !tex.empty();
		}
	}

	std::string commit() {
		std::vector<float> albedo(scene.materials.size() * 3);
		for (size_t i = 0; i < scene.materials.size(); ++i)
			albedo[3 * i] = scene.materials[i].albedo.x, albedo[3 * i + 1] = scene.materials[i].albedo.y, albedo[3 * i + 2] = scene.materials[i].albedo.z;
		const auto t0 = std::chrono::steady_clock::now();
		const int rc = rtgi_hip_renderer_build(renderer, scene.vertices.data(), sizeof(Vertex), (uint32_t)scene.vertices.size(), scene.triangles.data(),
		                                       sizeof(Triangle), (uint32_t)scene.triangles.size(), albedo.data(), (uint32_t)scene.materials.size());
		note_time("build", t0);
		if (rc < 0) return rtgi_hip_renderer_last_error(renderer);
		return "";
	}

	void print_timers() const {
		for (const auto &t : timers) std::cout << t.first << ": " << t.second.first << " ms in " << t.second.second << " call(s)" << std::endl;
	}

	void note_time(const std::string &what, std::chrono::steady_clock::time_point t0) {
		auto &t = timers[what];
		t.first += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		t.second += 1;
	}

	rtgi_frame frame() const {
		rtgi_frame f;
		std::memset(&f, 0, sizeof f);
		const Scene &s = scene;
		f.cam = {{s.cam_dir.x, s.cam_dir.y, s.cam_dir.z}, {s.cam_pos.x, s.cam_pos.y, s.cam_pos.z}, {s.cam_up.x, s.cam_up.y, s.cam_up.z}, s.fovy, s.w, s.h};
		f.sppx = sppx, f.algo = (uint32_t)algo, f.bounces = bounces;
		f.light_pos[0] = s.light_pos.x, f.light_pos[1] = s.light_pos.y, f.light_pos[2] = s.light_pos.z;
		f.light_col[0] = s.light_col.x, f.light_col[1] = s.light_col.y, f.light_col[2] = s.light_col.z;
		f.seed = seed, f.jitter = jitter, f.sharding = sharding, f.tile = tile;
		return f;
	}

	// run(algo), driver/main.cpp:38-48: clear, sppx samples of every pixel, PNG to cmdline.outfile
	std::string run() {
		push_materials();
		const rtgi_frame f = frame();
		std::vector<uint8_t> rgb((size_t)scene.w * scene.h * 3);
		rtgi_frame_stats st;
		const auto t0 = std::chrono::steady_clock::now();
		const int rc = rtgi_hip_renderer_render(renderer, &f, nullptr, rgb.data(), &st);
		note_time("run", t0);
		if (rc < 0) return rtgi_hip_renderer_last_error(renderer);
		std::cout << "rendered " << scene.w << "x" << scene.h << " at " << sppx << " sppx: " << st.rays << " rays in " << st.ms_total << " ms ("
		          << (double)st.rays / (st.ms_total * 1e3) << " Mrays/s) on " << st.n_devices << " GPU" << (st.n_devices > 1 ? "s" : "") << std::endl;
		if (!write_png(opt.outfile, rgb.data(), scene.w, scene.h)) return "cannot write " + opt.outfile;
		return "";
	}

	// rt_bench, driver/main.cpp:50-65: closest_hit on one frame of camera rays through pixel centres, timed
	std::string rt_bench() {
		const rtgi_frame f = frame();
		uint64_t n = 0;
		double ms = 0.0;
		const auto t0 = std::chrono::steady_clock::now();
		if (rtgi_hip_renderer_rt_bench(renderer, &f, &n, &ms) < 0) return rtgi_hip_renderer_last_error(renderer);
		note_time("rt_bench", t0);
		std::cout << "rt_bench: " << n << " rays in " << ms << " ms (" << (double)n / (ms * 1e3) << " Mrays/s)" << std::endl;
		return "";
	}

	// ---- the command interpreter: one table of commands, one small function per command ----
	// State that outlives a line.
	bool camera_up_given = false, scene_up_given = false, leave = false;
	std::string selected_material;  // by name: `load` may move the material table
	std::vector<std::string> journal;  // every line read so far (`history`)
	int line_no = 0;

	void interpret(const std::string &text);
	void read_commands(std::istream &input, bool prompt);

	Material *material_named(const std::string &name) {
		for (Material &m : scene.materials)
			if (m.name == name) return &m;
		return nullptr;
	}
	void require_gpu_state(bool for_bench) const;

	// handlers (the grammar is that of driver/interaction.cpp:81-404; see the table below for who serves what)
	void cmd_camera_vector(const Line &l);
	void cmd_scene_up(const Line &l);
	void cmd_load(const Line &l);
	void cmd_resolution(const Line &l);
	void cmd_algo(const Line &l);
	void cmd_bounces(const Line &l);
	void cmd_outfile(const Line &l);
	void cmd_raytracer(const Line &l);
	void cmd_commit(const Line &l);
	void cmd_sppx(const Line &l);
	void cmd_run(const Line &l);
	void cmd_mesh(const Line &l);
	void cmd_material(const Line &l);
	void cmd_default_brdf(const Line &l);
	void cmd_pointlight(const Line &l);
	void cmd_omp(const Line &l);
	void cmd_stats(const Line &l);
	void cmd_echo(const Line &l);
	void cmd_history(const Line &l);
	void cmd_leave(const Line &l);
	void cmd_backend(const Line &l);
};

// A command that cannot be carried out: reported on stderr with the line it came from, the script goes on.
struct Refused {
	std::string why;
};
[[noreturn]] void refuse(const std::string &why) { throw Refused{why}; }

bool whole_float(const std::string &s, float &v) {
	if (s.empty()) return false;
	char *stop = nullptr;
	v = std::strtof(s.c_str(), &stop);
	return stop && *stop == '\0';
}
bool whole_int(const std::string &s, long &v) {
	if (s.empty()) return false;
	char *stop = nullptr;
	v = std::strtol(s.c_str(), &stop, 10);
	return stop && *stop == '\0';
}

// One input line cut into blank-separated words; word[0] is the command.
struct Line {
	std::string text;
	std::vector<std::string> word;
	std::vector<size_t> begin;  // where each word starts in text
	explicit Line(const std::string &t) : text(t) {
		size_t at = 0;
		while (at < text.size()) {
			while (at < text.size() && std::isspace((unsigned char)text[at])) ++at;
			if (at == text.size()) break;
			const size_t from = at;
			while (at < text.size() && !std::isspace((unsigned char)text[at])) ++at;
			word.push_back(text.substr(from, at - from));
			begin.push_back(from);
		}
	}
	size_t args() const { return word.empty() ? 0 : word.size() - 1; }
	const std::string &arg(size_t i) const {
		static const std::string none;
		return i + 1 < word.size() ? word[i + 1] : none;
	}
	// everything after the first n arguments, blanks inside kept
	std::string tail(size_t n) const { return n + 1 < word.size() ? text.substr(begin[n + 1]) : std::string(); }
	float real(size_t i, const char *what) const {
		float v;
		if (!whole_float(arg(i), v)) refuse(std::string(what) + ": '" + arg(i) + "' is not a number");
		return v;
	}
	long integer(size_t i, const char *what) const {
		long v;
		if (!whole_int(arg(i), v)) refuse(std::string(what) + ": '" + arg(i) + "' is not a whole number");
		return v;
	}
	vec3 triple(size_t first, const char *what) const {
		vec3 v;
		v.x = real(first, what), v.y = real(first + 1, what), v.z = real(first + 2, what);
		return v;
	}
};

struct CommandEntry {
	const char *name;
	int least, most;  // number of arguments accepted (most < 0: any)
	const char *usage;
	void (Repl::*serve)(const Line &);
};
// The command surface. Names, argument order and meaning follow the renderer's scripts (driver/interaction.cpp:81-404);
// `bounces`, `hip ...` and `algo path` are this backend's additions.
const CommandEntry COMMANDS[] = {
    {"at", 0, 3, "at [x y z]", &Repl::cmd_camera_vector},
    {"look", 0, 3, "look [x y z]", &Repl::cmd_camera_vector},
    {"camup", 0, 3, "camup [x y z]", &Repl::cmd_camera_vector},
    {"up", 0, 3, "up [x y z]", &Repl::cmd_scene_up},
    {"load", 1, 2, "load file.obj [name]", &Repl::cmd_load},
    {"resolution", 2, 2, "resolution width height", &Repl::cmd_resolution},
    {"algo", 1, 1, "algo primary|local|path", &Repl::cmd_algo},
    {"bounces", 1, 1, "bounces N", &Repl::cmd_bounces},
    {"outfile", 1, 1, "outfile file.png", &Repl::cmd_outfile},
    {"raytracer", 1, -1, "raytracer seq|naive-bvh|bbvh [indexed] [esc]|hip [indexed] [esc] [devices N|a,b,..]", &Repl::cmd_raytracer},
    {"commit", 0, 0, "commit", &Repl::cmd_commit},
    {"sppx", 1, 1, "sppx N", &Repl::cmd_sppx},
    {"run", 0, 0, "run", &Repl::cmd_run},
    {"rt_bench", 0, 0, "rt_bench", &Repl::cmd_run},
    {"mesh", 1, 1, "mesh list", &Repl::cmd_mesh},
    {"material", 1, -1, "material list|select name|albedo r g b|emissive r g b|roughness v|ior v|texture file.png|texture drop|brdf name|show", &Repl::cmd_material},
    {"default-brdf", 1, 1, "default-brdf lambert|phong|layered-phong|gtr2|layered-gtr2", &Repl::cmd_default_brdf},
    {"pointlight", 8, 9, "pointlight [replace] pos x y z col r g b", &Repl::cmd_pointlight},
    {"omp", 1, 1, "omp on|off", &Repl::cmd_omp},
    {"stats", 1, 1, "stats print|clear", &Repl::cmd_stats},
    {"echo", 0, -1, "echo text", &Repl::cmd_echo},
    {"history", 0, 0, "history", &Repl::cmd_history},
    {"quit", 0, 0, "quit", &Repl::cmd_leave},
    {"exit", 0, 0, "exit", &Repl::cmd_leave},
    {"bvh", 1, -1, "bvh om|sm|sah N|triangles single|triangles multiple N|statistics|export depth file.obj", &Repl::cmd_backend},
    {"hip", 1, -1, "hip reorder auto|on|off|sharding tiles|samples|tile N|seed N|jitter on|off", &Repl::cmd_backend},
};

void Repl::cmd_camera_vector(const Line &l) {
	vec3 &v = l.word[0] == "at" ? scene.cam_pos : l.word[0] == "look" ? scene.cam_dir : scene.cam_up;
	if (l.args() == 0) {  // a query
		std::cout << v << std::endl;
		return;
	}
	if (l.args() != 3) refuse("three numbers expected (x y z), or none to print the current value");
	v = l.triple(0, l.word[0].c_str());
	if (l.word[0] == "camup") camera_up_given = true;
}

// The scene's up axis is fixed once; until `camup` says otherwise the camera stands upright in it.
void Repl::cmd_scene_up(const Line &l) {
	if (l.args() == 0 && scene_up_given) {
		std::cout << scene.up << std::endl;
		return;
	}
	if (scene_up_given) refuse("the scene's up vector is already set (camup turns the camera)");
	if (l.args() != 3) refuse("three numbers expected (x y z)");
	scene.up = l.triple(0, "up");
	scene_up_given = true;
	if (!camera_up_given) scene.cam_up = scene.up, camera_up_given = true;
}

void Repl::cmd_load(const Line &l) {
	const std::string problem = load_obj(scene, l.arg(0), l.arg(1));
	if (!problem.empty()) refuse(problem);
	scene_touched_at = line_no;
}

void Repl::cmd_resolution(const Line &l) {
	const long w = l.integer(0, "width"), h = l.integer(1, "height");
	if (w <= 0 || h <= 0 || w > 65536 || h > 65536) refuse("width and height must lie in 1..65536");
	scene.w = (int)w, scene.h = (int)h;
}

void Repl::cmd_algo(const Line &l) {
	static const struct {
		const char *name;
		int id;
	} known[] = {{"primary", RTGI_ALGO_PRIMARY}, {"local", RTGI_ALGO_LOCAL}, {"path", RTGI_ALGO_PATH}};
	for (const auto &k : known)
		if (l.arg(0) == k.name) {
			algo = k.id;
			return;
		}
	refuse("no algorithm named '" + l.arg(0) + "' (primary, local, path)");
}

void Repl::cmd_bounces(const Line &l) {
	const long n = l.integer(0, "bounces");
	if (n < 0 || n > 64) refuse("bounces must lie in 0..64");
	bounces = (unsigned)n;
}

void Repl::cmd_outfile(const Line &l) {
	if (!opt.outfile_from_cmdline) opt.outfile = l.arg(0);  // -o on the command line wins over the script
}

// "devices 3" -> 0,1,2; "devices 0,2,5" -> as listed
std::vector<int> device_list(const std::string &spec) {
	std::vector<int> out;
	long v;
	if (spec.find(',') == std::string::npos) {
		if (!whole_int(spec, v) || v < 1 || v > 64) refuse("devices: a count in 1..64 or a comma-separated list expected, got '" + spec + "'");
		for (long d = 0; d < v; ++d) out.push_back((int)d);
		return out;
	}
	std::istringstream parts(spec);
	for (std::string p; std::getline(parts, p, ',');) {
		if (!whole_int(p, v) || v < 0) refuse("devices: '" + p + "' is not a GPU number");
		out.push_back((int)v);
	}
	if (out.empty() || out.size() > 64) refuse("devices: between 1 and 64 GPUs");
	return out;
}

void Repl::cmd_raytracer(const Line &l) {
	const std::string &kind = l.arg(0);
	const bool own = kind == "hip", bbvh = kind == "bbvh";
	bool indexed = false, esc = false;
	std::vector<int> devices = {0};
	if (kind == "seq" || kind == "naive-bvh") {
		if (l.args() > 1) refuse("'" + kind + "' takes no options");
		std::cout << "note: '" << kind << "' is served by the hip tracer (flat layout)" << std::endl;
	} else if (own || bbvh) {
		for (size_t k = 1; k < l.args(); ++k) {
			const std::string &tag = l.arg(k);
			if (tag == "indexed")
				indexed = true;
			else if (tag == "esc")
				esc = true;
			else if (own && tag == "devices" && k + 1 < l.args())
				devices = device_list(l.arg(++k));
			else if (own)
				refuse("option '" + tag + "' not understood");
			// (further words after `bbvh` are passed over, as the renderer's registry does)
		}
		if (esc && !indexed) refuse("early split clipping needs the indexed layout (raytracer " + kind + " indexed esc)");
	} else
		refuse("no ray tracer named '" + kind + "' (seq, naive-bvh, bbvh, hip)");
	if (!opt.dry_run) {
		rtgi_hip_renderer *fresh = rtgi_hip_renderer_create(devices.data(), (int)devices.size(), indexed ? RTGI_LAYOUT_INDEXED : RTGI_LAYOUT_FLAT, esc ? 1 : 0);
		if (!fresh) refuse(rtgi_hip_renderer_last_error(nullptr));
		if (renderer) rtgi_hip_renderer_destroy(renderer);
		renderer = fresh;
	}
	tracer_desc = kind;
	tracer_touched_at = line_no;
}

void Repl::cmd_commit(const Line &) {
	if (scene.vertices.empty()) refuse("nothing loaded yet: there is no geometry to build a BVH over");
	if (tracer_touched_at == 0) refuse("no ray tracer chosen yet (raytracer ...) to hand the scene to");
	if (!opt.dry_run) {
		const std::string problem = commit();
		if (!problem.empty()) refuse(problem);
	}
	accel_touched_at = line_no;
}

void Repl::cmd_sppx(const Line &l) {
	const long n = l.integer(0, "sppx");
	if (n < 1 || n > (1 << 24)) refuse("sppx must be at least 1");
	sppx = (unsigned)n;
}

// What must have happened, and in which order, before rays can be traced (the rules of driver/interaction.cpp:220-225):
// a scene, a tracer, a commit later than both.
void Repl::require_gpu_state(bool for_bench) const {
	std::string lacking;
	auto need = [&](bool have, const char *what) {
		if (!have) lacking += (lacking.empty() ? "" : ", ") + std::string(what);
	};
	need(scene_touched_at != 0, "a scene (load)");
	need(tracer_touched_at != 0, "a ray tracer (raytracer)");
	need(accel_touched_at != 0, "a built BVH (commit)");
	if (!for_bench) need(algo >= 0, "an algorithm (algo)");
	if (!lacking.empty()) refuse("not ready, still missing: " + lacking);
	if (accel_touched_at < tracer_touched_at) refuse("the ray tracer was chosen after the last commit: its BVH is stale, commit again");
	if (accel_touched_at < scene_touched_at) refuse("the scene changed after the last commit: the BVH is stale, commit again");
	if (!for_bench && algo != RTGI_ALGO_PRIMARY && !scene.have_light) refuse("this algorithm needs a light (pointlight pos x y z col r g b)");
}

void Repl::cmd_run(const Line &l) {
	const bool bench = l.word[0] == "rt_bench";
	require_gpu_state(bench);
	if (opt.dry_run) return;
	const std::string problem = bench ? rt_bench() : run();
	if (!problem.empty()) refuse(problem);
}

void Repl::cmd_mesh(const Line &l) {
	if (l.arg(0) != "list") refuse("only 'mesh list' is available");
	for (const Object &o : scene.objects) std::cout << o.name << std::endl;
}

void Repl::cmd_material(const Line &l) {
	const std::string &what = l.arg(0);
	const size_t given = l.args() - 1;  // arguments after the sub-command
	auto exactly = [&](size_t n, const char *form) {
		if (given != n) refuse(std::string("expected: material ") + form);
	};
	if (what == "list") {
		exactly(0, "list");
		for (const Material &m : scene.materials) std::cout << m.name << std::endl;
		return;
	}
	if (what == "select") {
		exactly(1, "select name");
		if (!material_named(l.arg(1))) refuse("no material named '" + l.arg(1) + "' (material list shows them)");
		selected_material = l.arg(1);
		return;
	}
	Material *m = selected_material.empty() ? nullptr : material_named(selected_material);
	if (!m) refuse("select a material first (material select name)");
	if (what == "albedo" || what == "emissive") {
		exactly(3, "albedo|emissive r g b");
		(what == "albedo" ? m->albedo : m->emissive) = l.triple(1, what.c_str());
	} else if (what == "roughness" || what == "ior") {
		exactly(1, "roughness|ior value");
		(what == "ior" ? m->ior : m->roughness) = l.real(1, what.c_str());
	} else if (what == "texture") {
		exactly(1, "texture file.png | texture drop");
		if (l.arg(1) == "drop")
			m->texture = -1;  // (the image stays loaded: another material may show it)
		else {
			std::string problem;
			const int t = load_image3f(scene.textures, l.arg(1), problem);
			if (t < 0) refuse(problem);
			m->texture = t, m->texture_path = l.arg(1);
		}
	} else if (what == "brdf") {
		exactly(1, "brdf name|default");
		const int k = l.arg(1) == "default" ? (int)scene.default_brdf : brdf_by_name(l.arg(1));
		if (k < 0) refuse("no BRDF named '" + l.arg(1) + "'");
		m->brdf = (uint32_t)k;
	} else if (what == "show") {
		exactly(0, "show");
		std::cout << "albedo     " << m->albedo << "\nalbedo-tex " << (m->texture < 0 ? std::string("none") : scene.textures[(size_t)m->texture].path)
		          << "\nemissive   " << m->emissive << "\nroughness  " << m->roughness << "\nior        " << m->ior << "\nbrdf       " << BRDF_NAMES[m->brdf]
		          << std::endl;
	} else
		refuse("'" + what + "' is not a material sub-command");
}

void Repl::cmd_default_brdf(const Line &l) {
	const int k = brdf_by_name(l.arg(0));
	if (k < 0) refuse("no BRDF named '" + l.arg(0) + "'");
	if (scene.default_brdf_set && !scene.materials.empty())
		std::cout << "note: materials loaded so far keep the BRDF they were given" << std::endl;
	scene.default_brdf = (uint32_t)k;  // what materials loaded from now on start with (libgi/scene.cpp:136)
	scene.default_brdf_set = true;
}

void Repl::cmd_pointlight(const Line &l) {
	const bool replace = l.arg(0) == "replace";
	const size_t at = replace ? 1 : 0;
	if (l.args() != at + 8 || l.arg(at) != "pos" || l.arg(at + 4) != "col") refuse("expected: pointlight [replace] pos x y z col r g b");
	const vec3 pos = l.triple(at + 1, "pos"), col = l.triple(at + 5, "col");
	// the renderer's algorithms light with the first point light only (gi/primary-hit.cpp:41)
	if (scene.have_light && !replace) {
		std::cout << "note: only the first point light is used (as by the reference's algorithms); use 'pointlight replace ...'" << std::endl;
		return;
	}
	scene.light_pos = pos, scene.light_col = col, scene.have_light = true;
}

void Repl::cmd_omp(const Line &l) {
	if (l.arg(0) != "on" && l.arg(0) != "off") refuse("expected: omp on|off");
	// accepted for the scripts' sake: no host thread traces rays here
}

void Repl::cmd_stats(const Line &l) {
	if (l.arg(0) == "clear")
		timers.clear();
	else if (l.arg(0) == "print")
		print_timers();
	else
		refuse("expected: stats print|clear");
}

void Repl::cmd_echo(const Line &l) { std::cout << l.tail(0) << std::endl; }

void Repl::cmd_history(const Line &) {
	for (size_t k = 0; k + 1 < journal.size(); ++k) std::cout << journal[k] << std::endl;  // (all but this very line)
}

void Repl::cmd_leave(const Line &) { leave = true; }

// `bvh ...` and `hip ...`: settings of the frame loop are kept here, everything else is the tracer's own business
// (ray_tracer::interprete, the hook of driver/interaction.cpp:399-400).
void Repl::cmd_backend(const Line &l) {
	const std::string &what = l.arg(0), &value = l.arg(1);
	if (l.word[0] == "hip" && (what == "sharding" || what == "tile" || what == "seed" || what == "jitter")) {
		if (l.args() != 2) refuse("expected: hip " + what + " <value>");
		if (what == "sharding") {
			if (value != "tiles" && value != "samples") refuse("expected: hip sharding tiles|samples");
			sharding = value == "tiles" ? RTGI_SHARD_TILES : RTGI_SHARD_SAMPLES;
		} else if (what == "tile") {
			const long n = l.integer(1, "tile");
			if (n < 1 || n > 4096) refuse("tile edge must lie in 1..4096");
			tile = (unsigned)n;
		} else if (what == "seed")
			seed = (unsigned)l.integer(1, "seed");
		else {
			if (value != "on" && value != "off") refuse("expected: hip jitter on|off");
			jitter = value == "on";
		}
		return;
	}
	if (tracer_touched_at == 0) refuse("unknown command (no ray tracer yet that could take it)");
	if (opt.dry_run) return;
	const int rc = rtgi_hip_renderer_interprete(renderer, l.word[0].c_str(), l.tail(0).c_str());
	if (rc < 0) refuse(rtgi_hip_renderer_last_error(renderer));
	if (rc == 0) refuse("unknown command (the ray tracer does not know it either)");
}

void Repl::interpret(const std::string &text) {
	const Line l(text);
	if (l.word.empty() || l.word[0][0] == '#') return;  // blank line, comment
	try {
		const CommandEntry *entry = nullptr;
		for (const CommandEntry &c : COMMANDS)
			if (l.word[0] == c.name) entry = &c;
		if (!entry) refuse("unknown command");
		if ((int)l.args() < entry->least || (entry->most >= 0 && (int)l.args() > entry->most)) refuse(std::string("expected: ") + entry->usage);
		(this->*entry->serve)(l);
	} catch (const Refused &r) {
		std::cerr << "line " << line_no << ": " << l.word[0] << ": " << r.why << std::endl;
		++errors;
	}
}

void Repl::read_commands(std::istream &input, bool prompt) {
	std::string text;
	while (!leave) {
		if (prompt) std::cout << "rtgi > " << std::flush;
		if (!std::getline(input, text)) break;
		++line_no;
		journal.push_back(text);
		interpret(text);
	}
	if (prompt) std::cout << std::endl;
}

}  // namespace

int run(int argc, char **argv);
int main(int argc, char **argv) {
	try {
		return run(argc, argv);
	} catch (const std::exception &e) {  // (out of memory on a huge scene, a stream failure)
		std::cerr << "rtgi_hip_repl: " << e.what() << std::endl;
		return 3;
	}
}
int run(int argc, char **argv) {
	Repl repl;
	for (int i = 1; i < argc; ++i) {
		const std::string a = argv[i];
		auto value = [&]() -> std::string { return i + 1 < argc ? argv[++i] : ""; };
		if (a == "-s" || a == "--script")
			repl.opt.script = value(), repl.opt.interact = false;
		else if (a == "-l" || a == "--load")
			repl.opt.script = value();
		else if (a == "-o" || a == "--outfile")
			repl.opt.outfile = value(), repl.opt.outfile_from_cmdline = true;
		else if (a == "--dry-run")
			repl.opt.dry_run = true;
		else if (a == "--strict")
			repl.opt.strict = true;
		else {
			std::cerr << "usage: rtgi_hip_repl [-s script | -l script] [-o out.png] [--dry-run] [--strict]" << std::endl;
			return 2;
		}
	}
	if (!repl.opt.script.empty()) {
		std::ifstream script(repl.opt.script);
		if (!script) {
			std::cerr << "cannot open " << repl.opt.script << std::endl;
			return 2;
		}
		repl.read_commands(script, false);
	}
	if (repl.opt.interact && !repl.leave) repl.read_commands(std::cin, true);
	repl.print_timers();
	return repl.opt.strict && repl.errors ? 1 : 0;
}
