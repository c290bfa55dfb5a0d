// TEST INFRASTRUCTURE ONLY - never linked into, imported by or called from the product path.
//
// Compiles the reference CPU renderer itself, unmodified, straight from where it lies
// under /root/reference (nothing is copied into this repository), and wraps it in a small
// C ABI so that tests can drive it with a parameterised image size, mesh list and camera.
// The output goes to oracle/_ref/libref.so (git-ignored). It exists to (1) pin the C
// restatement in oracle/rt_oracle.c bit-for-bit and (2) generate the golden fixtures under
// tests/golden/.
//
// What is taken from the reference, verbatim through #include:
//   load_mesh            Serial/raytracer.cpp:220-287
//   GridAccel            Serial/grid.h:37-231
//   fast_trace           Serial/raytracer.cpp:71-118
//   globals              Serial/geometry.h:10-11 (global_t, global_triangle_near)
// What is restated here because the reference hard-codes it inside render():
//   the camera set-up and per-pixel ray generation, Serial/raytracer.cpp:124-161, with
//   width/height/camera turned into arguments. With the defaults (512x512, camera (3,5,3),
//   target 0, up (0,-1,0), fov 45) the PPM body produced from this harness is byte-identical
//   to the stock binary's test.ppm.
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <fstream>
#include <vector>
#include <iostream>
#include <cassert>
#include <string>
#include <sstream>
#include <unordered_map>
#include <algorithm>
#include <chrono>

// The voxel lists are private members; the headers above are already included (their include
// guards make the reference's own #includes no-ops), so this only affects the reference code.
#define private public
#define main ref_main_unused
#include "/root/reference/Serial/raytracer.cpp"
#undef main
#undef private

namespace {
struct RefScene {
    std::vector<Triangle*> tris;
    GridAccel* grid = nullptr;
    std::unordered_map<const Triangle*, int32_t> index;
};
}  // namespace

extern "C" {

void* ref_scene_create() { return new RefScene(); }

// Mirrors one load_mesh() call of Serial/raytracer.cpp:195-196.
int ref_scene_add_obj(void* h, const char* path, int format_has_vt, const float* color, const float* offset) {
    RefScene* s = (RefScene*)h;
    size_t before = s->tris.size();
    load_mesh(path, s->tris, format_has_vt != 0, Vec3f(color[0], color[1], color[2]), false,
              Vec3f(offset[0], offset[1], offset[2]));
    return (int)(s->tris.size() - before);
}

int64_t ref_scene_num_triangles(void* h) { return (int64_t)((RefScene*)h)->tris.size(); }

// Copies the loaded triangle positions out (9 floats per triangle: v0, v1, v2).
void ref_scene_get_triangles(void* h, float* out9) {
    RefScene* s = (RefScene*)h;
    for (size_t i = 0; i < s->tris.size(); ++i) {
        const Triangle* t = s->tris[i];
        float* o = out9 + 9 * i;
        o[0] = t->v0.x; o[1] = t->v0.y; o[2] = t->v0.z;
        o[3] = t->v1.x; o[4] = t->v1.y; o[5] = t->v1.z;
        o[6] = t->v2.x; o[7] = t->v2.y; o[8] = t->v2.z;
    }
}

// Builds the reference grid (Serial/grid.h:79-153) and returns its wall time in ms.
double ref_scene_build_grid(void* h) {
    RefScene* s = (RefScene*)h;
    auto t0 = std::chrono::steady_clock::now();
    s->grid = new GridAccel(s->tris);
    auto t1 = std::chrono::steady_clock::now();
    s->index.clear();
    s->index.reserve(s->tris.size() * 2);
    for (size_t i = 0; i < s->tris.size(); ++i) s->index[s->tris[i]] = (int32_t)i;
    return std::chrono::duration<double, std::milli>(t1 - t0).count();
}

// Grid description: n[3], bounds lo[3] hi[3], width[3], invWidth[3].
void ref_grid_info(void* h, int* n, float* lohi, float* width, float* inv_width) {
    GridAccel* g = ((RefScene*)h)->grid;
    for (int a = 0; a < 3; ++a) { n[a] = g->nVoxels[a]; width[a] = g->width[a]; inv_width[a] = g->invWidth[a]; }
    lohi[0] = g->bounds.lowerB.x; lohi[1] = g->bounds.lowerB.y; lohi[2] = g->bounds.lowerB.z;
    lohi[3] = g->bounds.upperB.x; lohi[4] = g->bounds.upperB.y; lohi[5] = g->bounds.upperB.z;
}

// Voxel lists as CSR. Pass cell_tris = NULL to only get the counts; returns the number of references.
int64_t ref_grid_csr(void* h, uint32_t* cell_start /*C+1*/, uint32_t* cell_tris /*R or NULL*/) {
    RefScene* s = (RefScene*)h;
    GridAccel* g = s->grid;
    int64_t C = (int64_t)g->nVoxels[0] * g->nVoxels[1] * g->nVoxels[2];
    int64_t r = 0;
    for (int64_t c = 0; c < C; ++c) {
        if (cell_start) cell_start[c] = (uint32_t)r;
        Voxel* v = g->voxels[c];
        if (!v) continue;
        for (size_t k = 0; k < v->triangleList.size(); ++k, ++r)
            if (cell_tris) cell_tris[r] = (uint32_t)s->index[v->triangleList[k]];
    }
    if (cell_start) cell_start[C] = (uint32_t)r;
    return r;
}

// Renders rows [y0, y1) of a W x H frame. rgb: float, 3 per pixel, row-major from row y0;
// ids: primary-hit triangle index or -1. Either output may be NULL. Returns wall ms of the pixel loop.
// cam = pos[3], target[3], up[3], fov. The light and shading constants are the reference's literals.
double ref_render_rows(void* h, int W, int H, int y0, int y1, const float* cam, float* rgb, int32_t* ids) {
    RefScene* s = (RefScene*)h;
    GridAccel* grid = s->grid;
    Vec3f camera_pos(cam[0], cam[1], cam[2]);
    Vec3f camera_target(cam[3], cam[4], cam[5]);
    Vec3f camera_up(cam[6], cam[7], cam[8]);
    float fov = cam[9];
    int width = W, height = H;

    // Serial/raytracer.cpp:130-138
    camera_up.normalize();
    Vec3f line_of_sight = camera_target.subtract(camera_pos);
    Vec3f w = line_of_sight.negate().normalize();
    Vec3f u = camera_up.crossProduct(w).normalize();
    Vec3f v = w.crossProduct(u).normalize();
    float focal_height = 1.0f;
    float aspectratio = float(width) / float(height);
    float focal_distance = focal_height / (2.0 * tan(fov * M_PI / (180.0 * 2.0)));

    Intersection probe;
    probe.use_eps = false;
    auto t0 = std::chrono::steady_clock::now();
    for (int y = y0; y < y1; ++y) {
        for (int x = 0; x < width; ++x) {
            // Serial/raytracer.cpp:153-161
            Vec3f dir(0);
            dir = dir.add(w.negate().scale(focal_distance));
            float xw = aspectratio * (x - width / 2.0 + 0.5) / width;
            float yw = (y - height / 2.0 + 0.5) / height;
            dir = dir.add(u.scale(xw));
            dir = dir.add(v.scale(yw));
            dir.normalize();
            Ray ray(camera_pos, dir, 0);
            size_t o = (size_t)(y - y0) * width + x;
            if (ids) {
                // The primary-hit triangle is whatever fast_trace's first Intersect leaves in
                // global_triangle_near; run that same call on its own to read it.
                global_t = INFINITY;
                bool hit = grid->Intersect(ray, &probe);
                ids[o] = hit ? s->index[global_triangle_near] : -1;
            }
            if (rgb) {
                Vec3f c = fast_trace(ray, grid);
                rgb[3 * o + 0] = c.x; rgb[3 * o + 1] = c.y; rgb[3 * o + 2] = c.z;
            }
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t1 - t0).count();
}

// The PPM quantisation of Serial/raytracer.cpp:181-183, applied to n floats.
void ref_quantize(const float* c, int64_t n, unsigned char* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (unsigned char)(std::min(float(1), c[i] / 255) * 255);
}

}  // extern "C"
