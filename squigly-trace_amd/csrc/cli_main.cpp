// squigly-trace — the reference's executable (app/Main.hs:13-75) over the C-ABI.
//
//   squigly-trace [--samples N|-s N] [--dimensions W,H|-d W,H] [--savepath F|-p F] [--objpath F]
//                 [--camerapath F|-c F] [--debug] [--debugpath F] [--cast]
//
// Same flags, defaults and printouts as the Haskell program; the render itself is sq_render_rgb8, i.e. the
// foreign call that replaces src/Lib.hs:73-74.  The material file named by `mtllib` is read from ./data/
// (src/Obj.hs:52).  PNG output (role of massiv-io's writeImage, src/Lib.hs:75): 8-bit RGB, stored deflate.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/squigly_host.h"

namespace {

uint32_t crc_table[256];
void crc_init() {
    for (uint32_t n = 0; n < 256; ++n) { uint32_t c = n; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; }
}
uint32_t crc32(const uint8_t* p, size_t n, uint32_t c = 0xFFFFFFFFu) { for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8); return c; }
void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<uint8_t>& out, const char* tag, const std::vector<uint8_t>& data) {
    put32(out, (uint32_t)data.size());
    std::vector<uint8_t> body(tag, tag + 4); body.insert(body.end(), data.begin(), data.end());
    out.insert(out.end(), body.begin(), body.end());
    put32(out, crc32(body.data(), body.size()) ^ 0xFFFFFFFFu);
}
// rows x cols RGB8, row-major: the layout of `Array S Ix2 (Pixel RGB Word8)`
bool write_png(const char* path, const uint8_t* rgb, int rows, int cols) {
    crc_init();
    std::vector<uint8_t> raw; raw.reserve((size_t)rows * (cols * 3 + 1));
    for (int r = 0; r < rows; ++r) { raw.push_back(0); raw.insert(raw.end(), rgb + (size_t)r * cols * 3, rgb + (size_t)(r + 1) * cols * 3); }
    std::vector<uint8_t> z = { 0x78, 0x01 };                              // zlib header, stored blocks
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        if (raw.empty()) break;
    }
    put32(z, (b << 16) | a);
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' }, ihdr;
    put32(ihdr, (uint32_t)cols); put32(ihdr, (uint32_t)rows); ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", z); chunk(out, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

std::string show_time(std::time_t t) {           // formatTime defaultTimeLocale "%T%P UTC"  (app/Main.hs:49-50)
    char buf[64]; std::tm g; gmtime_r(&t, &g);
    std::strftime(buf, sizeof buf, "%H:%M:%S", &g);
    return std::string(buf) + (g.tm_hour < 12 ? "am" : "pm") + " UTC";
}
int fail(const char* what) { std::fprintf(stderr, "squigly-trace: %s: %s\n", what, sq_last_error()); return 1; }

}  // namespace

int main(int argc, char** argv) {
    // defaults: app/Main.hs:14-30
    int samples = 10, w = 540, h = 540; bool debug = false, cast = false;
    std::string save = "./render/result.png", objp = "./data/scene.obj", camp = "./data/camera", dbgp;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], val; bool has = false;
        const size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); has = true; }
        auto need = [&]() -> const char* { if (has) return val.c_str(); if (i + 1 < argc) return argv[++i]; std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); };
        if (a == "--samples" || a == "-s") samples = std::atoi(need());
        else if (a == "--dimensions" || a == "-d") { if (std::sscanf(need(), "%d,%d", &w, &h) != 2) { std::fprintf(stderr, "--dimensions wants W,H\n"); return 2; } }
        else if (a == "--savepath" || a == "-p") save = need();
        else if (a == "--objpath") objp = need();
        else if (a == "--camerapath" || a == "-c") camp = need();
        else if (a == "--debugpath") dbgp = need();
        else if (a == "--debug") debug = true;
        else if (a == "--cast") cast = true;
        else if (a == "--help" || a == "-?") {
            std::puts("squigly-trace was made by Ruko (https://github.com/rukokarasu/)\nA cute raytracer\n"
                      "  -s --samples=INT        How many samples per pixel to trace\n  -d --dimensions=INT,INT Dimensions of the resulting image\n"
                      "  -p --savepath=FILE      Where to save the output\n     --objpath=FILE       File to load .obj from\n"
                      "  -c --camerapath=FILE    File to load camera data from\n     --debug              Run in debug mode\n"
                      "     --debugpath=FILE     File to write debug info to\n     --cast               Raycast instead of raytracing (i.e. don't bounce rays)");
            return 0;
        } else { std::fprintf(stderr, "Unknown flag: %s\n", a.c_str()); return 2; }
    }
    sq_camera cam;
    if (sq_camera_from_file(camp.c_str(), &cam)) return fail("Failed to parse camera");        // app/Main.hs:38
    sq_mesh* mesh = nullptr;
    if (sq_mesh_from_obj(objp.c_str(), "./data", &mesh)) return fail("loading the scene");      // app/Main.hs:58-61
    if (debug) {                                                                                 // src/Obj.hs:55-57: print (head objs); print mats
        const char *first = "", *mats = "";
        sq_mesh_debug_show(mesh, &first, &mats);
        if (!*first) { std::fprintf(stderr, "squigly-trace: Prelude.head: empty list\n"); return 1; }   // `head objs` of a file without objects throws
        std::printf("%s\n%s\n", first, mats);
    }
    sq_bih* bih = nullptr;
    // app/Main.hs:66.  Both builds give the same arrays; the GPU one wins from a few 10^4 triangles up.
    const bool on_gpu = sq_mesh_num_tris(mesh) >= 50000 && sq_device_count() > 0;
    if (on_gpu ? sq_bih_build_device(mesh, 0, &bih) : sq_bih_build(mesh, &bih)) return fail("building the BIH");
    if (debug) {                                                                                 // app/Main.hs:68-74
        if (!dbgp.empty()) {
            sq_scene sc; sq_bih_scene(bih, &sc);
            if (FILE* f = std::fopen(dbgp.c_str(), "w")) {
                for (int32_t i = 0; i < sc.n_nodes; ++i)
                    std::fprintf(f, "%d kind=%d count=%d lmax=%.9g rmin=%.9g link=%d\n", i, sc.nodes[i].kind & 3, sc.nodes[i].kind >> 2, sc.nodes[i].lmax, sc.nodes[i].rmin, sc.nodes[i].link);
                std::fclose(f);
                std::printf("Wrote BIH to %s\n", dbgp.c_str());
            }
        }
        std::printf("BIH height is %d\nLength of longest leaf is %d\nNumber of leaves is %d\n", sq_bih_height(bih), sq_bih_longest_leaf(bih), sq_bih_num_leaves(bih));
    }
    std::puts("Rendering scene...");
    const auto t0 = std::chrono::system_clock::now();
    std::printf("Started at %s\n", show_time(std::chrono::system_clock::to_time_t(t0)).c_str());
    sq_scene sc; sq_bih_scene(bih, &sc);
    std::vector<uint8_t> img((size_t)(w > 0 ? w : 0) * (size_t)(h > 0 ? h : 0) * 3);
    if (sq_render_rgb8(&sc, &cam, samples, w, h, cast ? 1 : 0, img.data())) return fail("render");  // src/Lib.hs:73-74
    if (!write_png(save.c_str(), img.data(), w, h)) { std::fprintf(stderr, "squigly-trace: cannot write %s\n", save.c_str()); return 1; }
    const auto t1 = std::chrono::system_clock::now();
    std::printf("Finished at %s\n", show_time(std::chrono::system_clock::to_time_t(t1)).c_str());
    std::printf("Took %.6fs\n", std::chrono::duration<double>(t1 - t0).count());
    sq_bih_free(bih); sq_mesh_free(mesh);
    return 0;
}
