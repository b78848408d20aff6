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
#include <hip/hip_runtime_api.h>

#include <algorithm>
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
		rtgi_hip_tracer *h = rtgi_hip_renderer_handle(renderer, 0);
		const rtgi_frame f = frame();
		const size_t n = (size_t)scene.w * scene.h;
		rtgi_ray *d_rays = nullptr;
		rtgi_hit *d_hits = nullptr;
		if (hipMalloc((void **)&d_rays, n * sizeof(rtgi_ray)) != hipSuccess || hipMalloc((void **)&d_hits, n * sizeof(rtgi_hit)) != hipSuccess) {
			if (d_rays) (void)hipFree(d_rays);
			return "out of GPU memory";
		}
		std::string err;
		if (rtgi_hip_camera_rays_device(h, &f.cam, nullptr, d_rays, nullptr) < 0 || hipDeviceSynchronize() != hipSuccess) err = rtgi_hip_last_error(h);
		if (err.empty()) {
			const auto t0 = std::chrono::steady_clock::now();
			if (rtgi_hip_closest_hit_device(h, d_rays, (uint32_t)n, d_hits, nullptr, nullptr) < 0 || hipDeviceSynchronize() != hipSuccess)
				err = rtgi_hip_last_error(h);
			const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			note_time("rt_bench", t0);
			if (err.empty()) std::cout << "rt_bench: " << n << " rays in " << ms << " ms (" << (double)n / (ms * 1e3) << " Mrays/s)" << std::endl;
		}
		(void)hipFree(d_rays), (void)hipFree(d_hits);
		return err;
	}

	void loop(std::istream &infile, bool prompt);
};

#define error(x)                                                                                 \
	{                                                                                            \
		std::cerr << "command " << cmdid << " (" << command << "): " << x << std::endl;         \
		++errors;                                                                                \
		continue;                                                                                \
	}
#define check_in(x) \
	if (in.bad() || in.fail()) error(x)
#define check_in_complete(x) \
	if (in.bad() || in.fail() || !in.eof()) error(x)

void Repl::loop(std::istream &infile, bool prompt) {
	bool cam_has_up = false, scene_up_set = false;
	Material *mat = nullptr;
	std::string mat_name;
	std::vector<std::string> commands;
	while (!infile.eof()) {
		if (prompt) std::cout << "rtgi > " << std::flush;
		cmdid++;
		std::string line, command;
		std::getline(infile, line);
		while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
		commands.push_back(line);
		std::istringstream in(line);
		in >> command;
		vec3 tmp;
		if (command == "history") {
			commands.pop_back();
			for (auto &x : commands) std::cout << x << std::endl;
		} else if (command == "quit" || command == "exit")
			return;
		else if (command == "at" || command == "look" || command == "camup") {
			vec3 &target = command == "at" ? scene.cam_pos : command == "look" ? scene.cam_dir : scene.cam_up;
			in.get();
			if (in.eof()) {
				std::cout << target << std::endl;
				continue;
			}
			in >> tmp;
			check_in_complete("Syntax error, requires 3 numerical components");
			target = tmp;
			if (command == "camup") cam_has_up = true;
		} else if (command == "up") {
			in.get();
			if (in.eof() && scene_up_set) {
				std::cout << scene.up << std::endl;
				continue;
			}
			if (scene_up_set) error("Cannot set scene up vector twice, did you mean camup?");
			in >> tmp;
			check_in_complete("Syntax error, requires 3 numerical components");
			scene_up_set = true;
			scene.up = tmp;
			if (!cam_has_up) {
				scene.cam_up = scene.up;
				cam_has_up = true;
			}
		} else if (command == "load") {
			std::string file, name;
			in >> file;
			if (!in.eof()) in >> name;
			check_in_complete("Syntax error, requires a file name (no spaces, sorry) and (optionally) a name");
			mat = nullptr;  // (the material table may move)
			const std::string err = load_obj(scene, file, name);
			if (!err.empty()) error(err);
			if (!mat_name.empty())
				for (auto &m : scene.materials)
					if (m.name == mat_name) mat = &m;
			scene_touched_at = cmdid;
		} else if (command == "resolution") {
			int w, h;
			in >> w >> h;
			check_in_complete("Syntax error, requires 2 integral values");
			if (w <= 0 || h <= 0) error("Resolution must be positive");
			scene.w = w, scene.h = h;
		} else if (command == "algo") {
			std::string name;
			in >> name;
			if (name == "primary")
				algo = RTGI_ALGO_PRIMARY;
			else if (name == "local")
				algo = RTGI_ALGO_LOCAL;
			else if (name == "path")  // this backend's own: diffuse paths, `bounces N` segments after the camera ray
				algo = RTGI_ALGO_PATH;
			else
				error("There is no gi algorithm called '" << name << "'");
		} else if (command == "bounces") {
			int n;
			in >> n;
			check_in_complete("Syntax error, requires exactly one non-negative integral value");
			if (n < 0 || n > 64) error("Between 0 and 64 bounces");
			bounces = (unsigned)n;
		} else if (command == "outfile") {
			std::string name;
			in >> name;
			check_in_complete("Syntax error, only accepts a single file name (no spaces, sorry)");
			if (!opt.outfile_from_cmdline) opt.outfile = name;  // "cmdline options should supersede the script's"
		} else if (command == "raytracer") {
			std::string name;
			in >> name;
			std::vector<std::string> tags;
			for (std::string t; in >> t;) tags.push_back(t);
			bool flat = true, esc = false;
			std::vector<int> devices = {0};
			if (name == "seq" || name == "naive-bvh") {
				if (!tags.empty()) error("There is no such variant");
				std::cout << "note: '" << name << "' is served by the hip tracer (flat layout)" << std::endl;
			} else if (name == "bbvh" || name == "hip") {
				bool bad = false;
				for (size_t k = 0; k < tags.size(); ++k) {
					if (tags[k] == "indexed")
						flat = false;
					else if (tags[k] == "esc")
						esc = true;
					else if (tags[k] == "devices" && name == "hip" && k + 1 < tags.size()) {
						const std::string &spec = tags[++k];
						devices.clear();
						if (spec.find(',') == std::string::npos) {
							const int n = std::atoi(spec.c_str());
							for (int d = 0; d < n; ++d) devices.push_back(d);
						} else {
							std::istringstream ds(spec);
							for (std::string d; std::getline(ds, d, ',');) devices.push_back(std::atoi(d.c_str()));
						}
						if (devices.empty() || devices.size() > 64) bad = true;
					} else if (name == "hip")
						bad = true;  // (the reference ignores unknown bbvh tags)
				}
				if (bad) error("Syntax error: raytracer hip [indexed] [esc] [devices N | devices a,b,...]");
				if (flat && esc) error("This combination is technically problematic");
			} else
				error("There is no ray tracer called '" << name << "'");
			if (!opt.dry_run) {
				rtgi_hip_renderer *r = rtgi_hip_renderer_create(devices.data(), (int)devices.size(), flat ? RTGI_LAYOUT_FLAT : RTGI_LAYOUT_INDEXED, esc ? 1 : 0);
				if (!r) error(rtgi_hip_renderer_last_error(nullptr));
				if (renderer) rtgi_hip_renderer_destroy(renderer);
				renderer = r;
			}
			tracer_desc = name;
			tracer_touched_at = cmdid;
		} else if (command == "commit") {
			if (scene.vertices.empty()) error("There is no scene data to work with");
			if (tracer_touched_at == 0) error("There is no ray traversal scheme to commit the scene data to");
			if (!opt.dry_run) {
				const std::string err = commit();
				if (!err.empty()) error(err);
			}
			accel_touched_at = cmdid;
		} else if (command == "sppx") {
			int n;
			in >> n;
			check_in_complete("Syntax error, requires exactly one positive integral value");
			if (n <= 0) error("Syntax error, requires exactly one positive integral value");
			sppx = (unsigned)n;
		} else if (command == "run" || command == "rt_bench") {
			const bool bench = command == "rt_bench";
			if (scene_touched_at == 0 || tracer_touched_at == 0 || accel_touched_at == 0 || (!bench && algo < 0))
				error("We have to have a scene loaded, a ray tracer set, an acceleration structure built" << (bench ? "" : " and an algorithm set") << " prior to running");
			if (accel_touched_at < tracer_touched_at) error("The current tracer does (might?) not have an up-to-date acceleration structure");
			if (accel_touched_at < scene_touched_at) error("The current acceleration structure is out-dated");
			if (!bench && algo != RTGI_ALGO_PRIMARY && !scene.have_light) error("There is no light: pointlight pos x y z col r g b");
			if (opt.dry_run) continue;
			const std::string err = bench ? rt_bench() : run();
			if (!err.empty()) error(err);
		} else if (command == "mesh") {
			std::string name;
			in >> name;
			if (in.eof() && name == "list") {
				for (auto &obj : scene.objects) std::cout << obj.name << std::endl;
				continue;
			}
			error("Meshes can only be listed in this version");
		} else if (command == "material") {
			std::string cmd;
			in >> cmd;
			if (in.eof() && cmd == "list") {
				for (auto &mtl : scene.materials) std::cout << mtl.name << std::endl;
				continue;
			}
			check_in("Syntax error, requires material name, command and subsequent arguments");
			command = cmd;
			if (command == "select") {
				std::string name;
				in >> name;
				check_in_complete("Only a single string (no whitespace) accepted");
				Material *m = nullptr;
				for (auto &mtl : scene.materials)
					if (mtl.name == name) {
						m = &mtl;
						break;
					}
				if (!m) error("No material called '" << name << "'");
				mat = m, mat_name = name;
				continue;
			}
			if (!mat) error("No material selected");
			if (command == "albedo") {
				in >> tmp;
				check_in_complete("Expects a color triplet");
				mat->albedo = tmp;
			} else if (command == "emissive") {
				in >> tmp;
				check_in_complete("Expects a color triplet");
				mat->emissive = tmp;
			} else if (command == "roughness") {
				in >> tmp.x;
				check_in_complete("Expects a floating point value");
				mat->roughness = tmp.x;
			} else if (command == "ior") {
				in >> tmp.x;
				check_in_complete("Expects a floating point value");
				mat->ior = tmp.x;
			} else if (command == "texture") {
				in >> cmd;
				check_in_complete("Expected a single (no whitespace) string value");
				// (the textures stay around: other materials might still use them, driver/interaction.cpp:293-294)
				if (cmd == "drop")
					mat->texture = -1;
				else {
					std::string err;
					const int t = load_image3f(scene.textures, cmd, err);
					if (t < 0) error(err);
					mat->texture = t, mat->texture_path = cmd;
				}
			} else if (command == "brdf") {
				in >> cmd;
				check_in_complete("Expected a single (no whitespace) string value");
				const int k = cmd == "default" ? (int)scene.default_brdf : brdf_by_name(cmd);
				if (k < 0) error("No such brdf defined: " << cmd);
				mat->brdf = (uint32_t)k;
			} else if (command == "show") {
				check_in_complete("Does not take further arguments");
				std::cout << "albedo     " << mat->albedo << std::endl;
				std::cout << "albedo-tex " << (mat->texture < 0 ? std::string("none") : scene.textures[(size_t)mat->texture].path) << std::endl;
				std::cout << "emissive   " << mat->emissive << std::endl;
				std::cout << "roughness  " << mat->roughness << std::endl;
				std::cout << "ior        " << mat->ior << std::endl;
				std::cout << "brdf       " << BRDF_NAMES[mat->brdf] << std::endl;
			} else
				error("Unknown subcommand");
		} else if (command == "default-brdf") {
			std::string name;
			in >> name;
			check_in_complete("Expected a single (no whitespace) string value");
			if (scene.default_brdf_set && !scene.materials.empty())
				std::cout << "Replacing default brdfs that has already been applied to some materials." << std::endl;
			const int k = brdf_by_name(name);
			if (k < 0) error("No such brdf defined: " << name);
			scene.default_brdf = (uint32_t)k;  // (materials loaded from now on get it, libgi/scene.cpp:136)
			scene.default_brdf_set = true;
		} else if (command == "pointlight") {
			vec3 p, c;
			std::string cmd;
			in >> cmd;
			check_in("Command incomplete");
			bool replace = false;
			if (cmd == "replace") {
				replace = true;
				in >> cmd;
				check_in("Command incomplete");
			}
			if (cmd == "pos")
				in >> p;
			else
				error("Syntax error: poinlight [replace] pos x y z col x y z");
			check_in("Syntax error: poinlight [replace] pos x y z col x y z");
			in >> cmd;
			check_in("Command incomplete");
			if (cmd == "col")
				in >> c;
			else
				error("Syntax error: poinlight [replace] pos x y z col x y z");
			check_in_complete("Syntax error: poinlight [replace] pos x y z col x y z");
			// local_illumination reads lights[0] only (gi/primary-hit.cpp:41): a further light without `replace` is kept out
			if (scene.have_light && !replace) {
				std::cout << "note: only the first point light is used (as by the reference's algorithms); use 'pointlight replace ...'" << std::endl;
				continue;
			}
			scene.light_pos = p, scene.light_col = c, scene.have_light = true;
		} else if (command == "omp") {
			std::string sub;
			in >> sub;
			if (sub != "on" && sub != "off") error("Syntax error: expected 'on' or 'off'");
			// (no host threads trace here)
		} else if (command == "stats") {
			std::string sub;
			in >> sub;
			if (sub == "clear")
				timers.clear();
			else if (sub == "print")
				for (auto &t : timers) std::cout << t.first << ": " << t.second.first << " ms in " << t.second.second << " call(s)" << std::endl;
			else
				error("No such stats subcommand");
		} else if (command == "echo") {
			std::string text;
			in >> std::ws;  // (the reference skips the blank with `in >> c`, which also eats the first letter)
			std::getline(in, text);
			std::cout << text << std::endl;
		} else if (command == "" || command[0] == '#')
			;
		else if (command == "hip" || command == "bvh") {
			std::string rest;
			std::getline(in, rest);
			std::istringstream sub(rest);
			std::string what;
			sub >> what;
			if (command == "hip" && (what == "sharding" || what == "tile" || what == "seed" || what == "jitter")) {
				std::string arg;
				sub >> arg;
				if (what == "sharding" && (arg == "tiles" || arg == "samples"))
					sharding = arg == "tiles" ? RTGI_SHARD_TILES : RTGI_SHARD_SAMPLES;
				else if (what == "tile" && std::atoi(arg.c_str()) > 0)
					tile = (unsigned)std::atoi(arg.c_str());
				else if (what == "seed" && !arg.empty())
					seed = (unsigned)std::strtoul(arg.c_str(), nullptr, 10);
				else if (what == "jitter" && (arg == "on" || arg == "off"))
					jitter = arg == "on";
				else
					error("Syntax error: hip sharding tiles|samples, hip tile N, hip seed N, hip jitter on|off");
				continue;
			}
			if (tracer_touched_at == 0) error("Unknown command");
			if (opt.dry_run) continue;
			const int rc = rtgi_hip_renderer_interprete(renderer, command.c_str(), rest.c_str());
			if (rc < 0) error(rtgi_hip_renderer_last_error(renderer));
			if (rc == 0) error("Unknown command");
		} else
			error("Unknown command");
	}
	std::cout << std::endl;
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
		repl.loop(script, false);
	}
	if (repl.opt.interact) repl.loop(std::cin, true);
	for (auto &t : repl.timers) std::cout << t.first << ": " << t.second.first << " ms in " << t.second.second << " call(s)" << std::endl;
	return repl.opt.strict && repl.errors ? 1 : 0;
}
