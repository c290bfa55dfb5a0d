// Host-side robustness check, built with AddressSanitizer and UBSan on the host code only (tools/hostcheck/run.sh): the OBJ
// reader on well-formed, malformed and random input, the plane generator, the PPM writers and the parameter helpers. No GPU
// is needed: entry points that need one must fail with an error code, not crash.
#include "../../include/rt_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <string>
#include <vector>

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed at line %d: %s (last error: %s)\n", __LINE__, #cond, rt_last_error()); ++failures; } } while (0)

static std::string tmp_path(const char* name) { return std::string(getenv("RT_HOSTCHECK_TMP") ? getenv("RT_HOSTCHECK_TMP") : "/tmp") + "/" + name; }

static void write_file(const std::string& path, const std::string& text) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    fwrite(text.data(), 1, text.size(), f);
    fclose(f);
}

int main() {
    const float col[3] = {255, 0, 0}, off[3] = {0, 0, 0};
    const std::string path = tmp_path("rt_hostcheck.obj");
    // hand-written cases: (text, expected triangles or a negative status)
    struct Case { const char* text; long long expect; };
    const Case cases[] = {
        {"", 0},
        {"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", 1},
        {"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1 2/1 3/1", 1},                       // no newline at the end
        {"v 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nf 1/7/9 2//4 3\r\n", 1},            // CR LF, mixed index forms
        {"# comment\nvt 0 0\nvn 0 0 1\ng group\nv 1e308 -1e308 1e-320\nv 1 2\nv\nf 1 2 3\n", 1},   // short vertex lines read as zeros
        {"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", RT_ERR_OBJ},                  // index beyond the vertices read so far
        {"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", RT_ERR_OBJ},
        {"v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 2 3\n", RT_ERR_OBJ},
        {"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 99999999999999999999 2 3\n", RT_ERR_OBJ},
        {"f 1 2 3\nv 0 0 0\nv 1 0 0\nv 0 1 0\n", RT_ERR_OBJ},                  // face before its vertices
        {"v nan inf -inf\nv 0x10 1e5000 .\nv 1 1 1\nf 1 2 3\n", 1},
        {"\n\n   \n\t\nv 0 0 0\n   v 1 0 0\n\tv 0 1 0\n  f   1   2   3   \n", 1},
        {"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2\n", RT_ERR_OBJ},                    // a face with two indices: the third reads as 0
        {"vv 0 0 0\nff 1 2 3\nvertex 1 2 3\n", 0},
    };
    for (const Case& c : cases) {
        write_file(path, c.text);
        rt_scene* s = rt_scene_create();
        const long long n = rt_scene_add_obj(s, path.c_str(), 1, col, off);
        if (n != c.expect) { fprintf(stderr, "case \"%.30s...\": got %lld, expected %lld (%s)\n", c.text, n, c.expect, rt_last_error()); ++failures; }
        CHECK(rt_scene_num_triangles(s) == (n > 0 ? n : 0));
        rt_scene_destroy(s);
    }
    // a missing file is an empty mesh, a directory an error
    {
        rt_scene* s = rt_scene_create();
        CHECK(rt_scene_add_obj(s, tmp_path("rt_hostcheck_does_not_exist.obj").c_str(), 1, col, off) == 0);
        CHECK(rt_scene_add_obj(s, "/tmp", 1, col, off) < 0);
        CHECK(rt_scene_add_obj(s, nullptr, 1, col, off) < 0);
        CHECK(rt_scene_add_obj(nullptr, path.c_str(), 1, col, off) < 0);
        rt_scene_destroy(s);
    }
    // random input: tokens of an OBJ file shuffled with junk, and plain random bytes
    std::mt19937_64 rng(12345);
    const char* tokens[] = {"v", "f", "vt", "vn", "#", "\n", "\n", "\n", " ", "\t", "\r", "1", "2", "3", "-1", "0", "4/5/6", "7//8", "1e3", "-0.5", ".", "e", "+", "-", "/",
                            "99999999999", "1e999", "nan", "inf", "0x1p3", "\0"};
    for (int it = 0; it < 3000; ++it) {
        std::string text;
        const int len = (int)(rng() % 400);
        if (it % 3 == 0) {
            for (int k = 0; k < len; ++k) text.push_back((char)(rng() & 0xff));
        } else {
            for (int k = 0; k < len; ++k) { text += tokens[rng() % (sizeof(tokens) / sizeof(tokens[0]))]; if (rng() % 3) text.push_back(' '); }
        }
        write_file(path, text);
        rt_scene* s = rt_scene_create();
        const long long before = rt_scene_num_triangles(s);
        const long long n = rt_scene_add_obj(s, path.c_str(), (int)(rng() & 1), col, off);
        CHECK(n >= 0 || n == RT_ERR_OBJ);
        CHECK(rt_scene_num_triangles(s) == before + (n > 0 ? n : 0));
        if (n > 0) {
            std::vector<float> v9((size_t)n * 9);
            CHECK(rt_scene_get_triangles(s, v9.data()) == RT_OK);
        }
        // the same file read in several stretches side by side: same status, same triangles
        {
            const int threads = 2 + (int)(rng() % 6);
            setenv("RT_OBJ_THREADS", std::to_string(threads).c_str(), 1);
            rt_scene* s2 = rt_scene_create();
            const long long n_par = rt_scene_add_obj(s2, path.c_str(), 1, col, off);
            unsetenv("RT_OBJ_THREADS");
            CHECK(n_par == n);
            if (n > 0 && n_par == n) {
                std::vector<float> a((size_t)n * 9), b((size_t)n * 9);
                rt_scene_get_triangles(s, a.data()); rt_scene_get_triangles(s2, b.data());
                CHECK(memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0);
            }
            rt_scene_destroy(s2);
        }
        // a second mesh on top, with a material and a scale
        rt_material m; memset(&m, 0, sizeof(m)); m.kd = 1; m.base_color[0] = 9;
        const long long n2 = rt_scene_add_obj_material(s, path.c_str(), 0, &m, off, 3);
        CHECK(n2 == n || (n2 < 0 && n < 0));
        rt_scene_destroy(s);
    }
    // well-formed files with vertices and faces interleaved, some with one bad face: one stretch against several
    {
        long long valid = 0, invalid = 0;
        for (int it = 0; it < 400; ++it) {
            std::string text;
            int nv = 0;
            const int lines = 20 + (int)(rng() % 400);
            const bool spoil = (rng() % 3) == 0;
            const int spoil_at = (int)(rng() % lines);
            for (int k = 0; k < lines; ++k) {
                const unsigned r = (unsigned)(rng() % 10);
                char tmp[160];
                if (r < 5 || nv < 3) {
                    snprintf(tmp, sizeof(tmp), "v %.6f %.6f %g\n", (double)(rng() % 2000) / 100.0 - 10.0, (double)(rng() % 2000) / 7.0, (double)(rng() % 1000) * 1e-3);
                    ++nv;
                } else if (r < 9) {
                    int a = 1 + (int)(rng() % nv), b = 1 + (int)(rng() % nv), c = 1 + (int)(rng() % nv);
                    if (spoil && k >= spoil_at && (rng() % 4) == 0) c = nv + 1 + (int)(rng() % 3);       // names a vertex that comes later, or none
                    if (rng() & 1) snprintf(tmp, sizeof(tmp), "f %d/1 %d/1 %d/1\n", a, b, c); else snprintf(tmp, sizeof(tmp), "f %d %d %d\n", a, b, c);
                } else {
                    snprintf(tmp, sizeof(tmp), "vt 0 0\n# %d\n\n", k);
                }
                text += tmp;
            }
            write_file(path, text);
            rt_scene* s1 = rt_scene_create();
            unsetenv("RT_OBJ_THREADS");
            const long long n1 = rt_scene_add_obj(s1, path.c_str(), 1, col, off);
            const std::string err1 = n1 < 0 ? rt_last_error() : "";
            setenv("RT_OBJ_THREADS", std::to_string(2 + (int)(rng() % 9)).c_str(), 1);
            rt_scene* s2 = rt_scene_create();
            const long long n2 = rt_scene_add_obj(s2, path.c_str(), 1, col, off);
            const std::string err2 = n2 < 0 ? rt_last_error() : "";
            unsetenv("RT_OBJ_THREADS");
            CHECK(n1 == n2);
            CHECK(err1 == err2);                                       // the same line and vertex count in the message
            if (n1 > 0 && n1 == n2) {
                std::vector<float> a((size_t)n1 * 9), b((size_t)n1 * 9);
                rt_scene_get_triangles(s1, a.data()); rt_scene_get_triangles(s2, b.data());
                CHECK(memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0);
            }
            if (n1 >= 0) ++valid; else ++invalid;
            rt_scene_destroy(s1); rt_scene_destroy(s2);
        }
        CHECK(valid > 100 && invalid > 30);
    }
    // in-memory meshes
    {
        rt_scene* s = rt_scene_create();
        std::vector<float> v9(9 * 100, 1.f);
        CHECK(rt_scene_add_triangles(s, v9.data(), 100, col) == 100);
        CHECK(rt_scene_add_triangles(s, nullptr, 0, col) == 0);
        CHECK(rt_scene_add_triangles(s, nullptr, 5, col) < 0);
        CHECK(rt_scene_add_triangles(s, v9.data(), -1, col) < 0);
        CHECK(rt_scene_num_triangles(s) == 100);
        rt_scene_destroy(s);
    }
    // plane generator: byte counts of the two sizes the reference ships or documents, and bad arguments
    {
        const std::string pp = tmp_path("rt_hostcheck_plane.obj");
        CHECK(rt_make_plane_obj(pp.c_str(), 10) == RT_OK);
        rt_scene* s = rt_scene_create();
        CHECK(rt_scene_add_obj(s, pp.c_str(), 1, col, off) == 2 * 101 * 101);
        rt_scene_destroy(s);
        CHECK(rt_make_plane_obj(pp.c_str(), 0) < 0);
        CHECK(rt_make_plane_obj(nullptr, 10) < 0);
        CHECK(rt_make_plane_obj("/nonexistent_dir/x.obj", 10) < 0);
        remove(pp.c_str());
    }
    // PPM writers: quantisation incl. NaN and out-of-range values, bad arguments
    {
        const std::string pp = tmp_path("rt_hostcheck.ppm");
        const float rgb[12] = {0.f, 255.f, 300.f, -5.f, NAN, INFINITY, -INFINITY, 127.5f, 254.999f, 1e-30f, 255.0001f, 128.f};
        CHECK(rt_write_ppm(pp.c_str(), rgb, 2, 2) == RT_OK);
        FILE* f = fopen(pp.c_str(), "rb");
        CHECK(f != nullptr);
        if (f) {
            char head[16] = {0};
            CHECK(fread(head, 1, 11, f) == 11 && memcmp(head, "P6\n2 2\n255\n", 11) == 0);
            unsigned char b[12];
            CHECK(fread(b, 1, 12, f) == 12);
            CHECK(b[0] == 0 && b[1] == 255 && b[2] == 255 && b[4] == 255 && b[5] == 255 && b[7] == 127 && b[8] == 254 && b[11] == 128);
            fclose(f);
        }
        CHECK(rt_write_ppm(pp.c_str(), rgb, 0, 2) < 0);
        CHECK(rt_write_ppm(nullptr, rgb, 2, 2) < 0);
        CHECK(rt_write_ppm_u8("/nonexistent_dir/x.ppm", (const unsigned char*)"abc", 1, 1) < 0);
        remove(pp.c_str());
    }
    // parameters and the band layout
    {
        rt_params p;
        rt_params_default(&p);
        CHECK(p.width == 512 && p.height == 512 && p.part_count == 1);
        CHECK(rt_part_rows(&p) == 512);
        for (int h : {1, 7, 8, 9, 480, 2160}) for (int parts : {1, 2, 3, 8, 64}) {
            p.height = h; p.part_count = parts;
            const int rows = rt_part_rows(&p);
            CHECK(rows >= 1 && (parts == 1 ? rows == h : rows % RT_BAND_ROWS == 0));
            CHECK((long long)rows * parts >= h || parts == 1);
        }
        CHECK(rt_part_rows(nullptr) == 0);
        rt_params_default(nullptr);
    }
    // entry points that need a GPU fail cleanly without one (or work, when the check runs on a GPU box)
    {
        rt_scene* s = rt_scene_create();
        std::vector<float> v9 = {0, 0, 0, 1, 0, 0, 0, 1, 0};
        rt_scene_add_triangles(s, v9.data(), 1, col);
        rt_params p; rt_params_default(&p); p.width = 16; p.height = 16;
        std::vector<unsigned char> img(16 * 16 * 3);
        const int rc = rt_render(s, &p, nullptr, img.data(), nullptr);
        CHECK(rc == RT_OK || rc == RT_ERR_NO_DEVICE || rc == RT_ERR_HIP);
        p.width = -1;
        CHECK(rt_render(s, &p, nullptr, img.data(), nullptr) == RT_ERR_INVALID);
        rt_params_default(&p); p.cam_pos[0] = NAN;
        CHECK(rt_render(s, &p, nullptr, img.data(), nullptr) == RT_ERR_INVALID);
        CHECK(rt_scene_set_frames_in_flight(s, 0) < 0);
        CHECK(rt_scene_set_frames_in_flight(s, 9) < 0);
        CHECK(rt_scene_set_frames_in_flight(nullptr, 2) < 0);
        void* lanes[4];
        CHECK(rt_streams_create(0, lanes) < 0);
        CHECK(rt_streams_create(2, nullptr) < 0);
        rt_comm* c = nullptr;
        CHECK(rt_comm_create(0, nullptr, &c) < 0);
        rt_scene_destroy(s);
        rt_scene_destroy(nullptr);
    }
    remove(path.c_str());
    printf("host check: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
