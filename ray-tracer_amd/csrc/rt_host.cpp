// Host side of the MI355X render path: the scene (what load_mesh() appends to), the OBJ reader, the
// camera set-up, device memory management, frame orchestration and the C ABI of include/rt_amd.h.
// No CPU rendering lives here: every compute entry point launches the HIP kernels of rt_kernels.hip.
#include "../../include/rt_amd.h"
#include "../../include/rt_amd_diag.h"
#include "rt_device.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return set_err(RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Entry points whose bodies allocate host memory (std::vector): no C++ exception may cross the C boundary.
#define RT_GUARD(T, call)                                                                               \
    try { return (call); }                                                                              \
    catch (const std::bad_alloc&) { return (T)set_err(RT_ERR_NOMEM, "out of host memory"); }            \
    catch (...) { return (T)set_err(RT_ERR_INVALID, "unexpected C++ exception"); }

// for the other translation units of the library
int rt_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* rt_last_error(void) { return g_err; }
extern "C" const char* rt_version(void) { return "rt_amd 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------------
// scene
// ------------------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap) return RT_OK;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return set_err(RT_ERR_HIP, "hipFree: %s", hipGetErrorString(e)); }
        hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return set_err(e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_HIP, "hipMalloc(%zu): %s", n * sizeof(T), hipGetErrorString(e)); }
        cap = n;
        return RT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Working memory of one frame in flight: everything the grid build and the render kernels write besides the caller's
// outputs, with the state of the grid it holds and the stream its last frame ran on. A scene has one of these unless
// rt_scene_set_frames_in_flight() asks for more; frames on different streams then use different sets and may overlap on
// the GPU, frames on one stream share a set and are ordered by the stream.
#define RT_MAX_IN_FLIGHT 8
struct WorkSet {
    DevBuf<RtGrid> d_grid;
    DevBuf<uint32_t> bounds_enc, cell_count, cell_start, block_sums, occupancy, ref_tri, ref_cell;
    DevBuf<uint2> row_minmax;
    DevBuf<float4> recs, blk_bounds;
    DevBuf<unsigned long long> counters;   // 24 frame counters, the work-list words (RtRenderWork::work, 32 of them), 32 partial sums of the cost estimate, the dequeue heads
    DevBuf<uint32_t> tile_list, unit_cost, tile_cost, queue_lists;
    // The cost estimate of every tile, by tile number, and the total, as the last frame with an estimate left them; est_key
    // says for which frame they hold: the bytes of its parameters (camera, light, frame size, part) and the scene's version.
    DevBuf<uint32_t> cost_by_tile;
    DevBuf<unsigned long long> kept_total;
    rt_params est_params;
    uint64_t est_version = 0;
    bool est_valid = false;
    bool zeroed = false;               // the fixed-size tables exist and were cleared for the current triangle list
    bool grid_valid = false;
    bool build_clean = false;          // the last grid build ran to its end and left its counters reset for the next one
    bool grid_sorted = false;          // how the current grid was built
    bool grid_indexed = false;         // and what its voxel lists hold: triangle indices, or the test records themselves
    hipStream_t last_stream = nullptr; // stream of the last frame or grid build on this set
    bool stream_used = false;
    uint64_t last_ticket = 0;          // when it was last used, to find the set that has been idle longest
    hipStream_t side_stream = nullptr; // work-list preparation beside the grid build (created when first needed)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void release() {
        d_grid.release(); bounds_enc.release(); cell_count.release(); cell_start.release(); block_sums.release(); occupancy.release();
        ref_tri.release(); ref_cell.release(); row_minmax.release(); recs.release(); blk_bounds.release(); counters.release();
        tile_list.release(); unit_cost.release(); tile_cost.release(); queue_lists.release();
        cost_by_tile.release(); kept_total.release(); est_valid = false;
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        ev_fork = ev_join = nullptr; side_stream = nullptr;
        zeroed = grid_valid = build_clean = grid_sorted = grid_indexed = stream_used = false;
        last_stream = nullptr;
    }
};

struct rt_scene {
    // host triangle list: what load_mesh() builds, flattened
    std::vector<float> tri;            // 9 floats per triangle: v0 v1 v2
    std::vector<uint32_t> tri_mesh;    // mesh index per triangle (one per load_mesh call)
    std::vector<float> mesh_color;     // 3 floats per mesh: Triangle::material
    std::vector<float> mesh_mat;       // 8 floats per mesh: kd, ks, spec_alpha, ka, km, reflective, 0, 0 (second shading mode)
    // device state
    int device = -1;
    bool uploaded = false;
    uint32_t known_refs = 0;           // reference count of the current triangle list once measured
    bool refs_known = false;
    DevBuf<float4> v0, v1, v2, tri_rec;
    DevBuf<float2> tri_box;
    DevBuf<uint32_t> tri_order;        // meshes above 65,536 triangles: the triangles along a space-filling curve (scene_spatial_order)
    bool use_order = false;            // the count and scatter kernels visit the triangles in that order (measured once per triangle list)
    int order_override = -1;           // RT_TRI_ORDER: 0 / 1, -1 = the library decides
    double run_area_spatial = 0;       // summed run-box area of the unsorted build in that order
    double order_adjacent[2] = {1, 1}; // share of neighbours in one lattice cell: in the file's order, along the curve
    DevBuf<float> d_mesh_color, d_mesh_mat;
    WorkSet sets[RT_MAX_IN_FLIGHT];
    int n_sets = 1;                    // rt_scene_set_frames_in_flight()
    int cur = 0;                       // set of the last frame or grid build: what the read-backs and the statistics look at
    uint64_t ticket = 0;
    int n_cu = 0;
    uint32_t known_cells = 0;              // n_cells of the current triangle list once measured
    RtGrid h_grid;                         // grid description of the current triangle list once measured
    // outputs owned by the host-buffer entry point
    DevBuf<float> o_rgb;
    DevBuf<uint8_t> o_u8;
    DevBuf<int32_t> o_id;
    DevBuf<unsigned long long> stamps;       // per ring slot: [0] frame start, [1] trace start, [2] trace end (100 MHz real-time clock,
                                             // written by the kernels), [3] unused
    bool built[RT_STATS_RING] = {};
    bool start_is_trace[RT_STATS_RING] = {};  // a frame without a build and without visible tiles starts with the trace kernel
    uint64_t frames = 0;               // frames rendered so far; frame k uses ring slot k % RT_STATS_RING
    bool count_work = false;
    // scheduling of the trace kernel: a tile is "expensive" when its estimate exceeds 1/thr0 of the even share of one
    // wavefront and "cheap" below 1/thr1 of it; unit_shifts holds log2 of the work units a tile is cut into, 4 bits per
    // queue list (expensive 0..7 from the top down, cheap 8..11, in-between 12..15)
    float sched_thr0 = 1.f, sched_thr1 = 4.f;
    uint64_t sched_shifts = 0x2222000011112222ull;
    bool sched_adapt = true;               // off when RT_SHIFTS gives the unit sizes explicitly
    float est_w[4] = {1.f, 500.f, 0.f, 700.f};    // cost estimate weights: primary triangles, primary voxels, shadow triangles, shadow voxels
    int side_stream_override = -1;           // RT_SIDE_STREAM: 0 / 1, -1 = the library decides
    int indexed_override = -1;               // RT_INDEXED_RECS: 0 / 1, -1 = the library decides
    int defer_override = -1;                 // RT_DEFER: 0 / 1, -1 = by the counts of earlier frames
    unsigned long long* feedback = nullptr;  // one word of pinned host memory the frames' kernels report to (RtBuildDev::feedback)
    uint64_t version = 1;                    // goes up whenever the triangle list changes
    bool reuse_estimate = true;              // frames with the camera, light and triangles of the set's last estimated frame skip the estimate
    uint64_t estimates_run = 0;              // frames that ran the cost estimate
    void* noalloc_base = nullptr;            // this device's copy of the fixed working memory of rt_render_device_noalloc (looked up at upload)
    rt_params last_params;
    // experiment overrides, read once when the scene is created: RT_OWN_PATH (-1 = the library decides), RT_CULL_MIN
    // (1..65535), RT_CULL_LANE_RUNS (0..64), RT_BLOCKS_PER_CU (0 = no cap)
    int own_path_override = -1;
    int cull_min = 64, cull_lane_runs = 12;
    int blocks_per_cu_cap = 0;
    size_t n_tris() const { return tri_mesh.size(); }
    // Voxel lists in ascending triangle index are only needed where the order shows: the counting build of the render kernel
    // (its shadow-ray test counts are those of the reference's order) and the read-back of the lists. RT_SORTED_BUILD=1
    // forces it, for experiments.
    int force_sorted = -1;                   // RT_SORTED_BUILD: 0 / 1 force a build mode, -1 = the library decides
    bool prefer_sorted = false;              // measured once per triangle list (scene_choose_record_order)
    double run_area[2] = {0, 0};             // what was measured: summed run-box areas, unsorted and sorted
    bool want_sorted() const { return count_work || (force_sorted >= 0 ? force_sorted != 0 : prefer_sorted); }
};

// hit trips per streamed run from which a frame takes the render kernel variant that finishes them together
#define RT_DEFER_HITS_PER_RUN 12

static int env_int(const char* name, int lo, int hi, int dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (!end || *end != 0) return dflt;
    return v < lo ? lo : (v > hi ? hi : (int)v);
}

extern "C" rt_scene* rt_scene_create(void) {
    rt_scene* s = new (std::nothrow) rt_scene();
    if (s) memset(&s->last_params, 0, sizeof(s->last_params));
    if (s) {
        if (const char* ew = getenv("RT_EST_W")) { float w[4]; if (sscanf(ew, "%f:%f:%f:%f", &w[0], &w[1], &w[2], &w[3]) == 4) for (int k = 0; k < 4; ++k) s->est_w[k] = w[k]; }
        // experiment overrides: RT_SCHED="thr0:thr1", RT_SHIFTS=16 hex digits, list 15 first
        if (const char* e = getenv("RT_SCHED")) { float a, b; if (sscanf(e, "%f:%f", &a, &b) == 2 && a > 0 && b >= a) { s->sched_thr0 = a; s->sched_thr1 = b; } }
        if (const char* e = getenv("RT_SHIFTS")) {
            char* end = nullptr;
            const unsigned long long v = strtoull(e, &end, 16);
            bool ok = end && end != e && *end == 0;          // an empty value is no override
            for (int k = 0; k < 16 && ok; ++k) ok = ((v >> (4 * k)) & 15ull) <= 4;
            if (ok) { s->sched_shifts = v; s->sched_adapt = false; }
        }
        s->own_path_override = env_int("RT_OWN_PATH", 0, 1, -1);
        s->cull_min = env_int("RT_CULL_MIN", 1, 65535, 64);
        s->cull_lane_runs = env_int("RT_CULL_LANE_RUNS", 0, 64, 12);
        s->blocks_per_cu_cap = env_int("RT_BLOCKS_PER_CU", 1, 64, 0);
        s->force_sorted = env_int("RT_SORTED_BUILD", 0, 1, -1);
        s->side_stream_override = env_int("RT_SIDE_STREAM", 0, 1, -1);
        s->reuse_estimate = env_int("RT_REUSE_ESTIMATE", 0, 1, 1) != 0;
        s->indexed_override = env_int("RT_INDEXED_RECS", 0, 1, -1);
        s->defer_override = env_int("RT_DEFER", 0, 1, -1);
        s->order_override = env_int("RT_TRI_ORDER", 0, 1, -1);
    }
    return s;
}

extern "C" void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    if (s->device >= 0) {
        (void)hipSetDevice(s->device);
        s->v0.release(); s->v1.release(); s->v2.release(); s->tri_rec.release(); s->tri_box.release(); s->tri_order.release(); s->d_mesh_color.release(); s->d_mesh_mat.release();
        for (WorkSet& w : s->sets) w.release();
        s->o_rgb.release(); s->o_u8.release(); s->o_id.release();
        s->stamps.release();
        if (s->feedback) (void)hipHostFree(s->feedback);
    }
    delete s;
}

static void scene_touch(rt_scene* s) {
    s->uploaded = false; s->refs_known = false;
    s->version++;
    for (WorkSet& w : s->sets) { w.grid_valid = false; w.build_clean = false; w.est_valid = false; }
}

// entry 1 of the Parallel tree's material table (Parallel/raytracer.cu:451) around a given colour
static rt_material default_material(const float color[3]) {
    rt_material m;
    for (int a = 0; a < 3; ++a) m.base_color[a] = color[a];
    m.kd = 10.0f; m.ks = 10.0f; m.spec_alpha = 1.25f; m.ka = 0.3f; m.km = 0.f; m.reflective = 0;
    return m;
}
static void push_mesh(rt_scene* s, const rt_material& m) {
    s->mesh_color.insert(s->mesh_color.end(), m.base_color, m.base_color + 3);
    const float mat[8] = {m.kd, m.ks, m.spec_alpha, m.ka, m.km, m.reflective ? 1.f : 0.f, 0.f, 0.f};
    s->mesh_mat.insert(s->mesh_mat.end(), mat, mat + 8);
}

static int64_t add_triangles_impl(rt_scene* s, const float* v9, int64_t n, const rt_material* mat) {
    if (!s || n < 0 || (n > 0 && !v9) || !mat) return set_err(RT_ERR_INVALID, "rt_scene_add_triangles: bad argument");
    if ((uint64_t)s->tri_mesh.size() + (uint64_t)n >= 0x7fffffffull) return set_err(RT_ERR_INVALID, "too many triangles");
    uint32_t mesh = (uint32_t)(s->mesh_color.size() / 3);
    push_mesh(s, *mat);
    s->tri.insert(s->tri.end(), v9, v9 + 9 * n);
    s->tri_mesh.insert(s->tri_mesh.end(), (size_t)n, mesh);
    scene_touch(s);
    return n;
}

extern "C" int64_t rt_scene_add_triangles(rt_scene* s, const float* v9, int64_t n, const float color[3]) {
    if (!color) return set_err(RT_ERR_INVALID, "rt_scene_add_triangles: bad argument");
    const rt_material m = default_material(color);
    RT_GUARD(int64_t, add_triangles_impl(s, v9, n, &m))
}
extern "C" int64_t rt_scene_add_triangles_material(rt_scene* s, const float* v9, int64_t n, const rt_material* material) {
    RT_GUARD(int64_t, add_triangles_impl(s, v9, n, material))
}

extern "C" int64_t rt_scene_num_triangles(const rt_scene* s) { return s ? (int64_t)s->tri_mesh.size() : 0; }

extern "C" int rt_scene_get_triangles(const rt_scene* s, float* v9) {
    if (!s || !v9) return set_err(RT_ERR_INVALID, "rt_scene_get_triangles: bad argument");
    memcpy(v9, s->tri.data(), s->tri.size() * sizeof(float));
    return RT_OK;
}

// ---- OBJ reader -------------------------------------------------------------------------------
// Behaviour of load_mesh (Serial/raytracer.cpp:220-287): the first whitespace-separated token of a line
// selects the record; "v" reads three doubles, adds the offset in double and rounds to float; "f" takes
// the integer in front of the first '/' of each of its first three tokens as a 1-based vertex index;
// "vt" and every other record do not reach the render path. Triangles hold copies of the positions.

static inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }

// One stretch of the file, from the start of a line to the end of a line: the vertices it holds and its faces, every face
// with the number of vertices of the stretch that precede it (a face may only name vertices read before it).
struct ObjFace { int32_t idx[3]; uint32_t nv_before; uint32_t line; };
struct ObjPart {
    std::vector<float> verts;
    std::vector<ObjFace> faces;
    uint32_t lines = 0;
    bool failed = false;              // out of memory inside the worker
};

static void parse_obj_part(char* p, char* const end, const double off[3], double scale, ObjPart& out) {
    try {
        while (p < end) {
            char* nl = (char*)memchr(p, '\n', (size_t)(end - p));
            char* le = nl ? nl : end;
            ++out.lines;
            char saved = *le;
            *le = 0;   // strtod must not run into the next line (the byte belongs to this stretch, or is the buffer's terminator)
            char* q = p;
            while (q < le && is_ws(*q)) ++q;
            char* t = q;
            while (t < le && !is_ws(*t)) ++t;
            if (t - q == 1 && *q == 'v') {
                double c[3] = {0, 0, 0};
                char* r = t;
                for (int k = 0; k < 3; ++k) {
                    char* e = nullptr;
                    double d = strtod(r, &e);
                    if (e == r) break;   // operator>> stores 0 and the stream stays failed
                    c[k] = d;
                    r = e;
                }
                out.verts.push_back((float)(scale * (c[0] + off[0])));      // times 1.0 is exact: the Serial loader's value
                out.verts.push_back((float)(scale * (c[1] + off[1])));
                out.verts.push_back((float)(scale * (c[2] + off[2])));
            } else if (t - q == 1 && *q == 'f') {
                ObjFace fc;
                char* r = t;
                for (int k = 0; k < 3; ++k) {
                    while (r < le && is_ws(*r)) ++r;
                    bool neg = false;
                    if (r < le && (*r == '+' || *r == '-')) { neg = (*r == '-'); ++r; }
                    long v = 0;
                    while (r < le && *r >= '0' && *r <= '9') { v = v * 10 + (*r - '0'); if (v > 0x7fffffff) v = 0x7fffffff; ++r; }
                    fc.idx[k] = (int32_t)(neg ? -v : v);
                    while (r < le && !is_ws(*r)) ++r;   // "/vt" and anything else glued to the token
                }
                fc.nv_before = (uint32_t)(out.verts.size() / 3);
                fc.line = out.lines;
                out.faces.push_back(fc);
            }
            *le = saved;
            p = nl ? nl + 1 : end;
        }
    } catch (...) { out.failed = true; }
}

// fn(0) .. fn(n-1), each on a thread of its own. A thread that cannot be started (std::system_error: thread limit) does
// not end the process: the parts without a thread run on the calling thread, and every thread that was started is joined
// before anything leaves this function, also when fn throws here.
template <class F>
static void run_parts(int n, F fn) {
    struct Joiner {
        std::vector<std::thread> th;
        ~Joiner() { for (std::thread& t : th) if (t.joinable()) t.join(); }
    } workers;
    workers.th.reserve((size_t)n);
    int started = 0;
    try {
        for (; started < n; ++started) workers.th.emplace_back(fn, started);
    } catch (const std::system_error&) {
    }
    for (int k = started; k < n; ++k) fn(k);
}

static int64_t add_obj_impl(rt_scene* s, const char* path, int format_has_vt, const rt_material* mat, const float offset[3], int32_t scale_i) {
    (void)format_has_vt;   // the texture index is parsed by the reference but never used by the path
    if (!s || !path || !mat || !offset) return set_err(RT_ERR_INVALID, "rt_scene_add_obj: bad argument");
    const double scale = (double)scale_i;      // Parallel/raytracer.cu:824: scale * (a + offset.x), an int times a double; 1 for the Serial loader
    FILE* f = fopen(path, "rb");
    if (!f) return 0;   // like the reference: a missing file is an empty mesh
    // a directory, a pipe or any other path that cannot be sized and read whole is an I/O error, not an empty mesh
    long sz = -1;
    if (fseek(f, 0, SEEK_END) == 0) sz = ftell(f);
    if (sz < 0 || sz > (long)0x7fffffff00l || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return set_err(RT_ERR_IO, "%s: cannot determine the file size", path); }
    std::vector<char> buf;
    try { buf.resize((size_t)sz + 1); } catch (...) { fclose(f); throw; }
    const size_t got = fread(buf.data(), 1, (size_t)sz, f);
    const bool read_err = ferror(f) != 0;
    fclose(f);
    if (read_err || got > (size_t)sz) return set_err(RT_ERR_IO, "%s: read error", path);
    buf[got] = 0;

    // The file is cut into stretches of whole lines that are parsed side by side (files of a few MB: one stretch); a face's
    // indices are checked afterwards against the vertices read before it in file order, and the triangles are emitted in
    // file order, so the result is that of reading the file line by line.
    const double off[3] = {(double)offset[0], (double)offset[1], (double)offset[2]};
    int n_parts = 1;
    if (got >= ((size_t)4 << 20)) {
        const unsigned hw = std::thread::hardware_concurrency();
        n_parts = (int)(hw < 1 ? 1 : (hw > 16 ? 16 : hw));
    }
    if (const char* e = getenv("RT_OBJ_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n_parts = v; }     // experiments and tests
    std::vector<size_t> cut((size_t)n_parts + 1, got);
    cut[0] = 0;
    for (int k = 1; k < n_parts; ++k) {
        size_t c = got / (size_t)n_parts * (size_t)k;
        if (c < cut[k - 1]) c = cut[k - 1];
        const char* nl = (const char*)memchr(buf.data() + c, '\n', got - c);
        cut[k] = nl ? (size_t)(nl - buf.data()) + 1 : got;
    }
    std::vector<ObjPart> parts((size_t)n_parts);
    if (n_parts == 1) {
        parts[0].verts.reserve(1 << 16);
        parse_obj_part(buf.data(), buf.data() + got, off, scale, parts[0]);
    } else {
        run_parts(n_parts, [&](int k) { parse_obj_part(buf.data() + cut[k], buf.data() + cut[k + 1], off, scale, parts[k]); });
    }
    for (const ObjPart& pt : parts)
        if (pt.failed) throw std::bad_alloc();
    // vertices and lines before each stretch
    std::vector<size_t> v_before((size_t)n_parts + 1, 0), l_before((size_t)n_parts + 1, 0), f_before((size_t)n_parts + 1, 0);
    for (int k = 0; k < n_parts; ++k) {
        v_before[k + 1] = v_before[k] + parts[k].verts.size() / 3;
        l_before[k + 1] = l_before[k] + parts[k].lines;
        f_before[k + 1] = f_before[k] + parts[k].faces.size();
    }
    if (s->tri_mesh.size() + f_before[n_parts] >= 0x7fffffffull) return set_err(RT_ERR_INVALID, "too many triangles");
    // the first face in file order whose index lies outside the vertices read so far ends the call (the reference reads out of bounds)
    for (int k = 0; k < n_parts; ++k)
        for (const ObjFace& fc : parts[k].faces) {
            const long nv = (long)(v_before[k] + fc.nv_before);
            if (fc.idx[0] < 1 || fc.idx[1] < 1 || fc.idx[2] < 1 || fc.idx[0] > nv || fc.idx[1] > nv || fc.idx[2] > nv)
                return set_err(RT_ERR_OBJ, "%s:%lld: face index outside 1..%ld", path, (long long)(l_before[k] + fc.line), nv);
        }
    const uint32_t mesh = (uint32_t)(s->mesh_color.size() / 3);
    const size_t tri_before = s->tri_mesh.size(), n_new = f_before[n_parts];
    s->tri.resize((tri_before + n_new) * 9);
    s->tri_mesh.resize(tri_before + n_new, mesh);
    // a vertex by its index in the file: the stretch that holds it, then its place there
    auto emit_part = [&](int k) {
        float* dst = s->tri.data() + (tri_before + f_before[k]) * 9;
        for (const ObjFace& fc : parts[k].faces) {
            for (int c = 0; c < 3; ++c) {
                const size_t g = (size_t)(fc.idx[c] - 1);
                int owner = k;                                     // vertices read before the face: this stretch or an earlier one
                while (g < v_before[owner]) --owner;
                const float* v = &parts[owner].verts[3 * (g - v_before[owner])];
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
                dst += 3;
            }
        }
    };
    if (n_parts == 1) emit_part(0);
    else run_parts(n_parts, emit_part);
    push_mesh(s, *mat);
    scene_touch(s);
    return (int64_t)n_new;
}

static int64_t add_obj_guarded(rt_scene* s, const char* path, int format_has_vt, const rt_material* mat, const float offset[3], int32_t scale) {
    const size_t tris_before = s ? s->tri_mesh.size() : 0, meshes_before = s ? s->mesh_color.size() / 3 : 0;
    try { return add_obj_impl(s, path, format_has_vt, mat, offset, scale); }
    catch (...) {
        // out of memory half way through the file: the triangle list is left as it was
        if (s) { s->tri.resize(tris_before * 9); s->tri_mesh.resize(tris_before); s->mesh_color.resize(meshes_before * 3); s->mesh_mat.resize(meshes_before * 8); }
        return set_err(RT_ERR_NOMEM, "out of host memory while reading %s", path ? path : "(null)");
    }
}
extern "C" int64_t rt_scene_add_obj(rt_scene* s, const char* path, int format_has_vt, const float color[3], const float offset[3]) {
    if (!color) return set_err(RT_ERR_INVALID, "rt_scene_add_obj: bad argument");
    const rt_material m = default_material(color);
    return add_obj_guarded(s, path, format_has_vt, &m, offset, 1);
}
extern "C" int64_t rt_scene_add_obj_material(rt_scene* s, const char* path, int format_has_vt, const rt_material* material, const float offset[3], int32_t scale) {
    return add_obj_guarded(s, path, format_has_vt, material, offset, scale);
}

// ------------------------------------------------------------------------------------------------
// parameters and camera
// ------------------------------------------------------------------------------------------------
extern "C" void rt_params_default(rt_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->width = 512; p->height = 512;
    p->cam_pos[0] = 3; p->cam_pos[1] = 5; p->cam_pos[2] = 3;
    p->cam_up[1] = -1;
    p->fov = 45;
    p->light_pos[0] = 5; p->light_pos[1] = -5; p->light_pos[2] = 2;
    p->kd = 2.0f; p->ks = 500000000000.0f; p->ka = 0.2f;
    p->spec_alpha = 4; p->light_intensity = 255; p->shadow_scale = 0.1f;
    p->eps = 1e-1f;
    p->walk_mode = RT_WALK_EARLY;
    p->part_index = 0; p->part_count = 1;
    p->shading_mode = RT_SHADE_SERIAL; p->reflect_depth = 3; p->shadow_bias = 0.02f;
}

extern "C" int32_t rt_part_rows(const rt_params* p) {
    if (!p || p->height <= 0) return 0;
    if (p->part_count <= 1) return p->height;
    int bands = (p->height + RT_BAND_ROWS - 1) / RT_BAND_ROWS;
    int per = (bands + p->part_count - 1) / p->part_count;
    return per * RT_BAND_ROWS;
}

// Vec3::normalize of the reference (Serial/geometry.h:23-30) on the host, for the camera basis.
static void h_normalize(float v[3]) {
    float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (n2 > 0) {
        float inv = (float)(1 / sqrt((double)n2));
        v[0] *= inv; v[1] *= inv; v[2] *= inv;
    }
}
static void h_cross(const float a[3], const float b[3], float o[3]) {
    float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

// Camera basis and focal distance, Serial/raytracer.cpp:130-138. tan() comes from the host libm, as in the reference.
static RtCamera make_camera(const rt_params* p) {
    RtCamera c;
    float up[3] = {p->cam_up[0], p->cam_up[1], p->cam_up[2]};
    h_normalize(up);
    for (int a = 0; a < 3; ++a) { c.pos[a] = p->cam_pos[a]; c.w[a] = -(p->cam_target[a] - p->cam_pos[a]); }
    h_normalize(c.w);
    h_cross(up, c.w, c.u); h_normalize(c.u);
    h_cross(c.w, c.u, c.v); h_normalize(c.v);
    c.aspect = (float)p->width / (float)p->height;
    c.focal_distance = (float)(1.0f / (2.0 * tan(p->fov * M_PI / (180.0 * 2.0))));
    return c;
}

// ------------------------------------------------------------------------------------------------
// device plumbing
// ------------------------------------------------------------------------------------------------
static int scene_bind_device(rt_scene* s) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return set_err(RT_ERR_NO_DEVICE, "no HIP device: %s", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (s->device >= 0 && s->device != dev) return set_err(RT_ERR_INVALID, "scene is bound to device %d, current device is %d", s->device, dev);
    if (s->device < 0) {
        s->device = dev;
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        s->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return RT_OK;
}

// Working memory of a frame as the frame code sees it: plain pointers, either into a WorkSet of the scene (allocated on
// demand, reused by later frames) or, for rt_render_device_noalloc(), into a fixed device array.
struct FrameMem {
    RtGrid* grid = nullptr;
    uint32_t *bounds_enc = nullptr, *cell_count = nullptr, *cell_start = nullptr, *block_sums = nullptr, *occupancy = nullptr;
    uint32_t *ref_tri = nullptr, *ref_cell = nullptr;
    uint2* row_minmax = nullptr;
    float4 *recs = nullptr, *blk_bounds = nullptr;
    size_t ref_cap = 0;                      // references recs and blk_bounds can hold
    bool sorted = false;                     // build the voxel lists in ascending triangle index (needs ref_tri, ref_cell, row_minmax)
    bool indexed = false;                    // the voxel lists hold triangle indices instead of copies of the 48-byte test records
    unsigned long long* counters = nullptr;  // RT_COUNTER_WORDS
    uint32_t *tile_list = nullptr, *tile_cost = nullptr, *queue_lists = nullptr;
    size_t tile_cap = 0;                     // tiles the three lists above can hold (queue_lists: RT_QUEUE_LISTS times as many)
    uint32_t* unit_cost = nullptr;           // counting frames only: 112 words per tile
    uint32_t* cost_by_tile = nullptr;        // the tiles' cost estimates by tile number, kept from frame to frame (may be NULL)
    unsigned long long* kept_total = nullptr;
    bool report = true;                      // the frame's counters feed the scene's choice of kernel variant (not when the memory is
                                             // shared with other scenes: the counters there may be another scene's)
};

static FrameMem scene_mem(const rt_scene* s, const WorkSet& w, bool indexed = false) {
    FrameMem m;
    m.indexed = indexed;
    m.grid = w.d_grid.p; m.bounds_enc = w.bounds_enc.p; m.cell_count = w.cell_count.p; m.cell_start = w.cell_start.p;
    m.block_sums = w.block_sums.p; m.occupancy = w.occupancy.p;
    m.ref_tri = w.ref_tri.p; m.ref_cell = w.ref_cell.p; m.row_minmax = w.row_minmax.p; m.recs = w.recs.p; m.blk_bounds = w.blk_bounds.p;
    m.ref_cap = w.recs.cap / 3;
    m.sorted = s->want_sorted();
    m.counters = w.counters.p;
    m.tile_list = w.tile_list.p; m.tile_cost = w.tile_cost.p; m.queue_lists = w.queue_lists.p; m.tile_cap = w.tile_list.cap;
    m.cost_by_tile = w.cost_by_tile.cap >= w.tile_list.cap ? w.cost_by_tile.p : nullptr; m.kept_total = w.kept_total.p;
    m.unit_cost = w.unit_cost.p;
    return m;
}

// The fixed-size tables of a set, cleared once per triangle list (an empty scene never builds a grid and reads them as
// they are).
static int workset_prepare(WorkSet& w) {
    if (w.zeroed) return RT_OK;
    int rc;
    if ((rc = w.d_grid.ensure(1)) || (rc = w.bounds_enc.ensure(8)) || (rc = w.cell_count.ensure(RT_MAX_CELLS)) ||
        (rc = w.cell_start.ensure(RT_MAX_CELLS + 1)) || (rc = w.block_sums.ensure(RT_MAX_SCAN_BLOCKS)) ||
        (rc = w.occupancy.ensure(RT_MAX_CELLS / 32 + 2)) || (rc = w.counters.ensure(RT_COUNTER_WORDS)))
        return rc;
    HIP_TRY(hipMemset(w.d_grid.p, 0, sizeof(RtGrid)));
    HIP_TRY(hipMemset(w.cell_start.p, 0, (RT_MAX_CELLS + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(w.occupancy.p, 0, (RT_MAX_CELLS / 32 + 2) * sizeof(uint32_t)));
    HIP_TRY(hipStreamSynchronize(nullptr));       // the frames run on other streams than these memsets
    w.zeroed = true;
    w.grid_valid = false;
    w.build_clean = false;
    return RT_OK;
}

static RtBuildDev build_args(rt_scene* s, const FrameMem& m) {
    RtBuildDev b;
    b.v0 = s->v0.p; b.v1 = s->v1.p; b.v2 = s->v2.p; b.tri_rec = s->tri_rec.p; b.tri_box = s->tri_box.p;
    b.n_tris = (uint32_t)s->n_tris();
    b.order = (s->use_order && s->n_tris() > 65536) ? s->tri_order.p : nullptr;
    b.cube_root = 3.f * powf((float)s->n_tris(), 1.f / 3.f);   // Serial/grid.h:94, host libm
    b.grid = m.grid;
    b.bounds_enc = m.bounds_enc;
    b.cell_count = m.cell_count; b.cell_start = m.cell_start; b.block_sums = m.block_sums; b.occupancy = m.occupancy;
    b.ref_tri = m.ref_tri; b.ref_cell = m.ref_cell; b.row_minmax = m.row_minmax; b.recs = m.recs; b.blk_bounds = m.blk_bounds;
    b.ref_cap = (uint32_t)m.ref_cap;
    b.counters = m.counters;
    b.stamp = nullptr;
    b.sorted = m.sorted ? 1 : 0;
    b.indexed = m.indexed ? 1 : 0;
    b.feedback = m.report ? s->feedback : nullptr;
    return b;
}

// The build state of a working set (counters the build kernels leave reset, the work list, the fork/join events) is shared
// between the frames that use it, so those frames have to be ordered. On one stream they are: a frame takes the set its
// stream used last. A stream the scene has not seen takes a set that is still free, and once all are taken the one that
// has been idle longest, after the host has waited for the work enqueued on that set's old stream (a rare case; an event
// per frame just to be able to order streams on the device would cost every frame about 6 us of idle GPU). With one set
// this is: frames on one stream are ordered by it, a change of stream waits.
static int scene_enter_stream(rt_scene* s, hipStream_t st, WorkSet** out) {
    int pick = -1;
    for (int i = 0; i < s->n_sets && pick < 0; ++i)
        if (s->sets[i].stream_used && s->sets[i].last_stream == st) pick = i;
    for (int i = 0; i < s->n_sets && pick < 0; ++i)
        if (!s->sets[i].stream_used) pick = i;
    if (pick < 0) {
        pick = 0;
        for (int i = 1; i < s->n_sets; ++i)
            if (s->sets[i].last_ticket < s->sets[pick].last_ticket) pick = i;
        HIP_TRY(hipStreamSynchronize(s->sets[pick].last_stream));
    }
    int rc = workset_prepare(s->sets[pick]);
    if (rc) return rc;
    *out = &s->sets[pick];
    return RT_OK;
}
static int scene_leave_stream(rt_scene* s, WorkSet* w, hipStream_t st) {
    w->last_stream = st;
    w->stream_used = true;
    w->last_ticket = ++s->ticket;
    s->cur = (int)(w - s->sets);
    return RT_OK;
}
static int scene_sync_streams(rt_scene* s) {
    for (int i = 0; i < s->n_sets; ++i)
        if (s->sets[i].stream_used) HIP_TRY(hipStreamSynchronize(s->sets[i].last_stream));
    return RT_OK;
}

// First half of a grid build on the scene's own memory, synchronously: how many (voxel, triangle) references the current
// triangle list has and what its grid looks like. Known before any frame, so that frames never wait for the device.
static int scene_measure_grid(rt_scene* s) {
    if (s->refs_known) return RT_OK;
    WorkSet& w = s->sets[0];
    HIP_TRY(rt_launch_build_phase1(build_args(s, scene_mem(s, w)), true, nullptr));
    RtGrid g;
    HIP_TRY(hipMemcpy(&g, w.d_grid.p, sizeof(g), hipMemcpyDeviceToHost));
    s->known_refs = g.n_refs;
    s->known_cells = (uint32_t)g.n_cells;
    s->h_grid = g;
    s->refs_known = true;
    w.build_clean = false;      // the count was not consumed by a scatter
    w.grid_valid = false;
    return RT_OK;
}

static int build_grid_async(rt_scene* s, const FrameMem& m, bool first, hipStream_t st, hipEvent_t after_phase1, unsigned long long* stamp);

// The triangles along a Morton curve through their centroids (10 bits per axis inside the centroids' bounds). The count
// and scatter kernels of the grid build take one triangle per thread; when thread t takes triangle order[t], the lanes
// of a wavefront hold spatial neighbours whatever the order of the file: they name the same voxels (one pooled atomic
// instead of 64), their records land side by side in the voxel lists, and the runs of 64 consecutive records that the
// render kernel culls by their bounds are compact patches. A file in arbitrary triangle order otherwise builds five
// times and renders two times slower (measured on the 3.6 M-triangle stand-in, shuffled). Triangle indices, and with them
// every result, stay what they are.
// adjacent[0], adjacent[1]: the share of neighbours in the file's order and along the curve whose centroids lie in the same
// cell of a 64^3 lattice: how coherent each order is for the wavefronts of the build (lanes naming the same voxel pool
// their atomics).
static void scene_spatial_order(const std::vector<float>& tri, size_t T, std::vector<uint32_t>& order, double adjacent[2]) {
    std::vector<uint32_t> key(T), tmp_key(T), tmp_idx(T);
    order.resize(T);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < T; ++i)
        for (int a = 0; a < 3; ++a) {
            const double c = ((double)tri[9 * i + a] + (double)tri[9 * i + 3 + a] + (double)tri[9 * i + 6 + a]) / 3.0;
            if (c < lo[a]) lo[a] = c;
            if (c > hi[a]) hi[a] = c;
        }
    double scale[3];
    for (int a = 0; a < 3; ++a) scale[a] = (hi[a] > lo[a] && hi[a] - lo[a] < 1e300) ? 1023.0 / (hi[a] - lo[a]) : 0.0;
    auto spread = [](uint32_t v) {            // 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000ffu;
        v = (v | (v << 8)) & 0x0300f00fu;
        v = (v | (v << 4)) & 0x030c30c3u;
        v = (v | (v << 2)) & 0x09249249u;
        return v;
    };
    for (size_t i = 0; i < T; ++i) {
        uint32_t q[3];
        for (int a = 0; a < 3; ++a) {
            const double c = ((double)tri[9 * i + a] + (double)tri[9 * i + 3 + a] + (double)tri[9 * i + 6 + a]) / 3.0;
            const double u = (c - lo[a]) * scale[a];
            q[a] = (u >= 0.0 && u <= 1023.0) ? (uint32_t)u : 0u;      // NaN and out-of-range centroids: first cell
        }
        key[i] = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
        order[i] = (uint32_t)i;
    }
    size_t same_file = 0, same_curve = 0;
    for (size_t i = 0; i + 1 < T; ++i) same_file += (key[i] >> 12) == (key[i + 1] >> 12);
    // stable LSD radix sort, three passes of 10 bits: equal keys keep ascending triangle index
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = 10 * pass;
        size_t count[1025] = {0};
        for (size_t i = 0; i < T; ++i) ++count[((key[i] >> shift) & 1023u) + 1];
        for (int k = 0; k < 1024; ++k) count[k + 1] += count[k];
        for (size_t i = 0; i < T; ++i) {
            const size_t d = count[(key[i] >> shift) & 1023u]++;
            tmp_key[d] = key[i]; tmp_idx[d] = order[i];
        }
        key.swap(tmp_key); order.swap(tmp_idx);
    }
    for (size_t i = 0; i + 1 < T; ++i) same_curve += (key[i] >> 12) == (key[i + 1] >> 12);
    adjacent[0] = T > 1 ? (double)same_file / (double)(T - 1) : 1.0;
    adjacent[1] = T > 1 ? (double)same_curve / (double)(T - 1) : 1.0;
}

// Which order of the records inside a voxel: the default build leaves them in the order the scatter's atomics hand the
// places out, which costs nothing; ascending triangle index costs a sort and a ranking pass per build. The order never
// changes the image, but the render kernel culls whole runs of 64 consecutive records by their bounds, and how tight those
// bounds are depends on which triangles share a run. For a mesh whose voxels collect their triangles from many separate
// index ranges (the generated plane: 17 strips per voxel, two thousand indices apart) the atomics shuffle those ranges
// and the runs get several times larger than in index order; for others (subdivided meshes) it makes no difference. So
// both builds run once per triangle list, the summed surface area of the run boxes is compared (the number of runs a ray
// meets is proportional to it), and the sort is kept only where it buys tighter runs. Synchronous, part of the upload.
// The same measurement decides in which order the count and scatter kernels visit the triangles: the order of the file, or
// the spatial order made at upload (scene_spatial_order) where that gives runs at least 10 % tighter; a file whose order is
// already coherent keeps its order (no index load in front of the triangle's box).
static int scene_choose_record_order(rt_scene* s) {
    s->prefer_sorted = false;
    s->use_order = false;
    s->run_area[0] = s->run_area[1] = 0;
    s->run_area_spatial = 0;
    if (s->n_tris() <= 65536 || s->known_refs == 0) return RT_OK;      // small meshes: voxels too small for run culling
    if (s->force_sorted >= 0 && s->order_override >= 0) { s->use_order = s->order_override != 0; return RT_OK; }
    int rc;
    WorkSet& w = s->sets[0];
    const size_t need = s->known_refs;
    if ((rc = w.recs.ensure(3 * need)) || (rc = w.blk_bounds.ensure(3 * (need / 64 + 2))) || (rc = w.ref_tri.ensure(need)) ||
        (rc = w.ref_cell.ensure(need)) || (rc = w.row_minmax.ensure(need / 64 + 2)))
        return rc;
    DevBuf<double> d_area;
    if ((rc = d_area.ensure(3))) return rc;
    hipError_t e = hipMemset(d_area.p, 0, 3 * sizeof(double));
    // [1] sorted (the result does not depend on the visiting order), [0] unsorted in file order, [2] unsorted in spatial order
    for (int mode = 2; mode >= 0 && e == hipSuccess; --mode) {
        FrameMem m = scene_mem(s, w);
        m.sorted = mode == 1;
        s->use_order = mode == 2;
        if (build_grid_async(s, m, true, nullptr, nullptr, nullptr)) { d_area.release(); s->use_order = false; return RT_ERR_HIP; }
        e = rt_launch_run_area(build_args(s, m), d_area.p + mode, nullptr);
    }
    s->use_order = false;
    double area[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(area, d_area.p, 3 * sizeof(double), hipMemcpyDeviceToHost);
    d_area.release();
    if (e != hipSuccess) return set_err(RT_ERR_HIP, "record order measurement: %s", hipGetErrorString(e));
    s->run_area[0] = area[0]; s->run_area[1] = area[1]; s->run_area_spatial = area[2];
    // spatial order where it gives tighter runs, or where the file's own order is incoherent (neighbours in the file rarely
    // share a lattice cell while neighbours on the curve mostly do): the build's atomics pool per voxel only for coherent lanes
    const bool file_incoherent = s->order_adjacent[1] > 0.25 && s->order_adjacent[0] < 0.5 * s->order_adjacent[1];
    s->use_order = s->order_override >= 0 ? s->order_override != 0 : (area[2] < 0.90 * area[0] || (file_incoherent && area[2] < 1.10 * area[0]));
    s->prefer_sorted = (s->use_order ? area[2] : area[0]) > 1.10 * area[1];
    if (!s->prefer_sorted) { w.ref_tri.release(); w.ref_cell.release(); w.row_minmax.release(); }
    w.grid_valid = false;
    w.build_clean = false;
    return RT_OK;
}

extern "C" int rt_scene_upload(rt_scene* s) {
    if (!s) return set_err(RT_ERR_INVALID, "rt_scene_upload: NULL scene");
    int rc = scene_bind_device(s);
    if (rc) return rc;
    if (s->uploaded) return RT_OK;
    const size_t T = s->n_tris();
    const size_t Tn = T ? T : 1;
    if ((rc = s->v0.ensure(Tn)) || (rc = s->v1.ensure(Tn)) || (rc = s->v2.ensure(Tn)) || (rc = s->tri_rec.ensure(3 * Tn)) || (rc = s->tri_box.ensure(3 * Tn))) return rc;
    if ((rc = s->d_mesh_color.ensure(s->mesh_color.size() ? s->mesh_color.size() : 3))) return rc;
    if ((rc = s->d_mesh_mat.ensure(s->mesh_mat.size() ? s->mesh_mat.size() : 8))) return rc;
    if (!s->feedback) {
        HIP_TRY(hipHostMalloc((void**)&s->feedback, sizeof(unsigned long long), hipHostMallocDefault));
        *s->feedback = 0ull;
    }
    // The fixed working memory of rt_render_device_noalloc: its address on this device is looked up here (module load, a
    // lock), so that the entry point itself only reads a pointer and may be called under stream capture right after upload.
    if (!s->noalloc_base) HIP_TRY(rt_noalloc_scratch(&s->noalloc_base));
    if (!s->stamps.p) {
        if ((rc = s->stamps.ensure(RT_STATS_RING * 4))) return rc;
        HIP_TRY(hipMemset(s->stamps.p, 0, RT_STATS_RING * 4 * sizeof(unsigned long long)));
    }
    HIP_TRY(hipDeviceSynchronize());         // frames of the previous triangle list may still be reading these buffers
    for (WorkSet& w : s->sets) w.zeroed = false;
    if ((rc = workset_prepare(s->sets[0]))) return rc;
    if (T) {
        // stage the 36-byte host records and the mesh indices, then split them into float4 streams on the device
        DevBuf<float> stage;
        DevBuf<uint32_t> stage_mesh;
        if ((rc = stage.ensure(9 * T)) || (rc = stage_mesh.ensure(T))) { stage.release(); stage_mesh.release(); return rc; }
        hipError_t e = hipMemcpy(stage.p, s->tri.data(), 9 * T * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(stage_mesh.p, s->tri_mesh.data(), T * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = rt_launch_upload_convert(stage.p, stage_mesh.p, build_args(s, scene_mem(s, s->sets[0])), nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        stage.release(); stage_mesh.release();
        if (e != hipSuccess) return set_err(RT_ERR_HIP, "scene upload: %s", hipGetErrorString(e));
    }
    if (!s->mesh_color.empty()) {
        HIP_TRY(hipMemcpy(s->d_mesh_color.p, s->mesh_color.data(), s->mesh_color.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_mesh_mat.p, s->mesh_mat.data(), s->mesh_mat.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    s->use_order = false;
    if (T > 65536) {
        std::vector<uint32_t> order;
        try { scene_spatial_order(s->tri, T, order, s->order_adjacent); }
        catch (...) { return set_err(RT_ERR_NOMEM, "out of host memory while ordering %zu triangles", T); }     // no C++ exception crosses the C boundary
        if ((rc = s->tri_order.ensure(T))) return rc;
        HIP_TRY(hipMemcpy(s->tri_order.p, order.data(), T * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    s->uploaded = true;
    s->refs_known = false;
    if ((rc = scene_measure_grid(s))) return rc;
    return scene_choose_record_order(s);
}

// Grid build on the working memory `m`. `first`: the build counters in `m` have not been left reset by a completed
// build (k_build_init runs). `after_phase1`, when given, is recorded once cell_start and the occupancy bitmap are final
// (the records are not yet).
static int build_grid_async(rt_scene* s, const FrameMem& m, bool first, hipStream_t st, hipEvent_t after_phase1, unsigned long long* stamp) {
    if ((size_t)s->known_refs > m.ref_cap) return set_err(RT_ERR_INVALID, "grid build: %u references exceed the capacity %zu", s->known_refs, m.ref_cap);
    RtBuildDev b1 = build_args(s, m);
    b1.stamp = stamp;
    HIP_TRY(rt_launch_build_phase1(b1, first, st));
    if (after_phase1) HIP_TRY(hipEventRecord(after_phase1, st));
    HIP_TRY(rt_launch_build_phase2(build_args(s, m), st));
    return RT_OK;
}

// The scene's own reference arrays, sized from the measured reference count.
static int scene_ensure_refs(rt_scene* s, WorkSet& w) {
    int rc;
    const size_t need = s->known_refs ? s->known_refs : 1;
    if ((rc = w.recs.ensure(3 * need)) || (rc = w.blk_bounds.ensure(3 * (need / 64 + 2)))) return rc;
    if (s->want_sorted() && ((rc = w.ref_tri.ensure(need)) || (rc = w.ref_cell.ensure(need)) || (rc = w.row_minmax.ensure(need / 64 + 2)))) return rc;
    return RT_OK;
}

// What the voxel lists of a frame hold: triangle indices (4 bytes per reference, the render kernel gathers the records) or
// copies of the 48-byte test records. Indices make the build cheaper (0.21 -> 0.18 ms on the 3.6 M-triangle stand-in) and
// the trace 2 % slower. A rank's share of a frame on several GPUs has the whole build and a fraction of the trace, so it
// takes indices; for whole frames of large meshes the two balance out slightly in favour of indices (stand-in 0.916 ->
// 0.902 ms one frame at a time, 0.584 -> 0.576 ms per frame with four in flight), for small meshes there is nothing in it.
static bool want_indexed(const rt_scene* s, int part_count) {
    if (s->indexed_override >= 0) return s->indexed_override != 0;
    return part_count > 1 || s->n_tris() > 65536;
}

static int scene_build_grid_async(rt_scene* s, WorkSet& w, hipStream_t st) {
    int rc = scene_ensure_refs(s, w);
    if (rc) return rc;
    const bool first = !w.build_clean;
    w.build_clean = false;
    const bool indexed = want_indexed(s, 1);          // what a whole frame rendered from this grid will ask for
    if ((rc = build_grid_async(s, scene_mem(s, w, indexed), first, st, nullptr, nullptr))) return rc;
    w.grid_valid = true;
    w.build_clean = true;
    w.grid_sorted = s->want_sorted();
    w.grid_indexed = indexed;
    return RT_OK;
}

extern "C" int rt_scene_build_grid(rt_scene* s, void* stream) {
    if (!s) return set_err(RT_ERR_INVALID, "rt_scene_build_grid: NULL scene");
    int rc = rt_scene_upload(s);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    WorkSet* w = nullptr;
    if ((rc = scene_enter_stream(s, st, &w))) return rc;
    if ((rc = scene_build_grid_async(s, *w, st))) return rc;
    return scene_leave_stream(s, w, st);
}

extern "C" int rt_scene_get_grid(rt_scene* s, int32_t n[3], float lohi[6], float width[3], float inv_width[3], uint32_t* n_cells, uint32_t* n_refs) {
    if (!s || !s->sets[s->cur].grid_valid) return set_err(RT_ERR_INVALID, "rt_scene_get_grid: grid not built");
    HIP_TRY(hipDeviceSynchronize());
    RtGrid g;
    HIP_TRY(hipMemcpy(&g, s->sets[s->cur].d_grid.p, sizeof(g), hipMemcpyDeviceToHost));
    if (g.overflow) return set_err(RT_ERR_HIP, "grid reference capacity exceeded");
    for (int a = 0; a < 3; ++a) {
        if (n) n[a] = g.n[a];
        if (lohi) { lohi[a] = g.lo[a]; lohi[3 + a] = g.hi[a]; }
        if (width) width[a] = g.width[a];
        if (inv_width) inv_width[a] = g.inv_width[a];
    }
    if (n_cells) *n_cells = (uint32_t)(g.n[0] * g.n[1] * g.n[2]);      // voxels (the device tables have a layer of border cells around them)
    if (n_refs) *n_refs = g.n_refs;
    return RT_OK;
}

extern "C" int rt_scene_get_csr(rt_scene* s, uint32_t* cell_start, uint32_t* cell_tris) {
    if (!s || !s->sets[s->cur].grid_valid) return set_err(RT_ERR_INVALID, "rt_scene_get_csr: grid not built");
    HIP_TRY(hipDeviceSynchronize());
    WorkSet& w = s->sets[s->cur];
    if (!w.grid_sorted) {
        // the lists are read back in the reference's order: build them that way once more
        const int forced = s->force_sorted;
        s->force_sorted = 1;
        int rc = scene_build_grid_async(s, w, nullptr);
        s->force_sorted = forced;
        if (rc) return rc;
        HIP_TRY(hipDeviceSynchronize());      // (a later frame rebuilds in its own mode if that differs)
    }
    RtGrid g;
    HIP_TRY(hipMemcpy(&g, w.d_grid.p, sizeof(g), hipMemcpyDeviceToHost));
    if (cell_start) {
        // the device table is laid out with a layer of (empty) border cells around the grid: the voxels' entries, in voxel order
        std::vector<uint32_t> bordered((size_t)g.n_cells + 1);
        HIP_TRY(hipMemcpy(bordered.data(), w.cell_start.p, bordered.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const size_t row = (size_t)g.n[0] + 2, layer = row * ((size_t)g.n[1] + 2);
        size_t c = 0;
        for (int z = 0; z < g.n[2]; ++z)
            for (int y = 0; y < g.n[1]; ++y)
                for (int x = 0; x < g.n[0]; ++x) cell_start[c++] = bordered[(size_t)(x + 1) + row * (size_t)(y + 1) + layer * (size_t)(z + 1)];
        cell_start[c] = bordered[(size_t)g.n_cells];
    }
    if (cell_tris && g.n_refs) {
        DevBuf<uint32_t> tmp;
        int rc = tmp.ensure(g.n_refs);
        if (rc) return rc;
        hipError_t e = rt_launch_extract_cell_tris(w.recs.p, tmp.p, g.n_refs, w.grid_indexed ? 1 : 0, nullptr);
        if (e == hipSuccess) e = hipMemcpy(cell_tris, tmp.p, (size_t)g.n_refs * sizeof(uint32_t), hipMemcpyDeviceToHost);
        tmp.release();
        if (e != hipSuccess) return set_err(RT_ERR_HIP, "csr read-back: %s", hipGetErrorString(e));
    }
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------
// render
// ------------------------------------------------------------------------------------------------
static bool finite3(const float v[3]) { return isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]); }

static int check_params(const rt_params* p) {
    if (!p) return set_err(RT_ERR_INVALID, "NULL params");
    if (p->width <= 0 || p->height <= 0 || p->width > 65536 || p->height > 65536) return set_err(RT_ERR_INVALID, "bad frame size %dx%d", p->width, p->height);
    if (p->part_count < 1 || p->part_index < 0 || p->part_index >= p->part_count) return set_err(RT_ERR_INVALID, "bad partition %d/%d", p->part_index, p->part_count);
    if (p->walk_mode != RT_WALK_EARLY && p->walk_mode != RT_WALK_FULL) return set_err(RT_ERR_INVALID, "bad walk_mode %d", p->walk_mode);
    if (p->shading_mode != RT_SHADE_SERIAL && p->shading_mode != RT_SHADE_PARALLEL) return set_err(RT_ERR_INVALID, "bad shading_mode %d", p->shading_mode);
    if (p->shading_mode == RT_SHADE_PARALLEL && (p->reflect_depth < 0 || p->reflect_depth > 3 || !isfinite(p->shadow_bias)))
        return set_err(RT_ERR_INVALID, "bad reflect_depth %d (0..3) or shadow_bias", p->reflect_depth);
    // The reference has no such check (its parameters are literals); a non-finite camera, light or constant gives NaN rays
    // and NaN hit distances, for which its own result is undefined (it reads a stale triangle pointer).
    if (!finite3(p->cam_pos) || !finite3(p->cam_target) || !finite3(p->cam_up) || !finite3(p->light_pos) || !isfinite(p->fov) || !isfinite(p->kd) ||
        !isfinite(p->ks) || !isfinite(p->ka) || !isfinite(p->spec_alpha) || !isfinite(p->light_intensity) || !isfinite(p->shadow_scale) || !isfinite(p->eps))
        return set_err(RT_ERR_INVALID, "non-finite camera, light or shading parameter");
    return RT_OK;
}

// Conservative screen rectangle (in 8x8 tiles of the full frame) of the pixels whose primary rays can enter the grid
// bounds: the bounding rectangle of the 8 projected corners, widened by 2 pixels. The per-pixel bounds test still
// decides exactly; this only limits where it has to run. Falls back to the whole frame when a corner is not
// safely in front of the camera.
static void grid_tile_rect(const RtCamera& c, const RtGrid& g, int W, int H, int& tx0, int& tx1, int& ty0, int& ty1) {
    const int tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
    tx0 = 0; tx1 = tiles_x - 1; ty0 = 0; ty1 = tiles_y - 1;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int k = 0; k < 8; ++k) {
        const double P[3] = {(k & 1) ? g.hi[0] : g.lo[0], (k & 2) ? g.hi[1] : g.lo[1], (k & 4) ? g.hi[2] : g.lo[2]};
        const double q[3] = {P[0] - c.pos[0], P[1] - c.pos[1], P[2] - c.pos[2]};
        const double depth = -(q[0] * c.w[0] + q[1] * c.w[1] + q[2] * c.w[2]);
        const double len = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        if (!(depth > 1e-3 * len) || !(len < 1e30)) return;          // behind / beside the camera, or not finite
        const double a = (q[0] * c.u[0] + q[1] * c.u[1] + q[2] * c.u[2]) / depth * c.focal_distance;
        const double b = (q[0] * c.v[0] + q[1] * c.v[1] + q[2] * c.v[2]) / depth * c.focal_distance;
        const double x = a * W / c.aspect + W / 2.0 - 0.5, y = b * H + H / 2.0 - 0.5;
        if (!(x == x) || !(y == y)) return;
        xmin = x < xmin ? x : xmin; xmax = x > xmax ? x : xmax;
        ymin = y < ymin ? y : ymin; ymax = y > ymax ? y : ymax;
    }
    xmin = floor(xmin) - 2; xmax = ceil(xmax) + 2; ymin = floor(ymin) - 2; ymax = ceil(ymax) + 2;
    if (xmax < 0 || ymax < 0 || xmin > W - 1 || ymin > H - 1) { tx1 = -1; ty1 = -1; return; }     // off screen
    if (xmin < 0) xmin = 0;
    if (ymin < 0) ymin = 0;
    if (xmax > W - 1) xmax = W - 1;
    if (ymax > H - 1) ymax = H - 1;
    tx0 = (int)xmin / 8; tx1 = (int)xmax / 8; ty0 = (int)ymin / 8; ty1 = (int)ymax / 8;
}

static size_t frame_tiles(const rt_params* p) {
    const size_t t = (size_t)((p->width + 7) / 8) * (size_t)((rt_part_rows(p) + 7) / 8);
    return t ? t : 1;
}

// One frame on the working memory `m`: clears the outputs, builds the grid when asked to, prepares the work list and
// traces. `ws`: the working set `m` points into, whose side stream and events may be used; without it (NULL) everything is
// enqueued on `st` in order and nothing but kernels and memsets is issued.
// `build`: 0 = the grid in `m` is current, 1 = rebuild, the build state in `m` was left clean, 2 = rebuild from scratch.
// `stamps` (may be NULL): four device words the kernels write the frame's times into, see rt_scene::stamps.
// Do two frames see the scene the same way, as far as the tiles' cost estimates are concerned: camera, frame size, part of
// the frame, and the light and eps the stand-in shadow rays of the estimate start from?
static bool same_view(const rt_params& a, const rt_params& b) {
    return a.width == b.width && a.height == b.height && a.part_index == b.part_index && a.part_count == b.part_count && a.fov == b.fov && a.eps == b.eps &&
           a.walk_mode == b.walk_mode && a.shading_mode == b.shading_mode &&
           memcmp(a.cam_pos, b.cam_pos, sizeof(a.cam_pos)) == 0 && memcmp(a.cam_target, b.cam_target, sizeof(a.cam_target)) == 0 &&
           memcmp(a.cam_up, b.cam_up, sizeof(a.cam_up)) == 0 && memcmp(a.light_pos, b.light_pos, sizeof(a.light_pos)) == 0;
}

static int render_frame(rt_scene* s, const rt_params* p, const FrameMem& m, int build, WorkSet* ws, void* d_rgb_f32, void* d_rgb_u8, void* d_hit_id,
                        hipStream_t st, unsigned long long* stamps, bool* start_is_trace) {
    int rc;
    // Large meshes with a grid build in the frame: everything that only needs the grid's cell table (clearing the outputs,
    // the tile list and its cost classes, about 0.14 ms at 4K) runs on a side stream beside the second half of the build and
    // joins before the trace kernel. The fork and the join cost about 6 us and 12 us of idle stream each, which is more
    // than small meshes (up to 65,536 triangles: the whole preparation takes 45 us) would gain: they stay on one stream.
    // With several frames in flight the other frames fill what a build leaves idle, and the second stream would only compete
    // with them for the few hardware queues.
    bool use_side = ws && build != 0 && s->n_tris() > 65536 && s->n_sets == 1;
    if (ws && build != 0 && s->side_stream_override >= 0) use_side = s->side_stream_override != 0;
    hipStream_t side = st;
    if (use_side && !ws->side_stream) {
        HIP_TRY(hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ws->ev_join, hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&ws->side_stream, hipStreamNonBlocking));
    }
    if (build != 0) {
        if ((rc = build_grid_async(s, m, build == 2, st, use_side ? ws->ev_fork : nullptr, stamps))) return rc;
        if (use_side) {
            side = ws->side_stream;
            HIP_TRY(hipStreamWaitEvent(side, ws->ev_fork, 0));
        }
    }
    const bool counting = s->count_work && m.unit_cost;
    // frames with a grid build: the build has cleared the counters (k_scan_sums)
    if (build == 0) HIP_TRY(rt_launch_counters_reset(m.counters, m.report ? s->feedback : nullptr, side));
    if (counting) HIP_TRY(hipMemsetAsync(m.counters + 22, 0xff, sizeof(unsigned long long), side));   // slot of a minimum
    {
        // background everywhere first: black, no hit
        const size_t out_px = (size_t)p->width * (size_t)rt_part_rows(p);
        if (d_rgb_f32) HIP_TRY(hipMemsetAsync(d_rgb_f32, 0, out_px * 3 * sizeof(float), side));
        if (d_rgb_u8) HIP_TRY(hipMemsetAsync(d_rgb_u8, 0, out_px * 3, side));
        if (d_hit_id) HIP_TRY(hipMemsetAsync(d_hit_id, 0xff, out_px * sizeof(int32_t), side));
    }

    RtSceneDev sd;
    sd.v0 = s->v0.p; sd.v1 = s->v1.p; sd.v2 = s->v2.p;
    sd.tri_rec = s->tri_rec.p;
    sd.mesh_color = s->d_mesh_color.p;
    sd.mesh_mat = s->d_mesh_mat.p;
    sd.n_tris = (uint32_t)s->n_tris();
    sd.grid = m.grid; sd.cell_start = m.cell_start; sd.occupancy = m.occupancy; sd.recs = m.recs; sd.blk_bounds = m.blk_bounds;
    sd.indexed = m.indexed ? 1u : 0u;
    RtShade sh;
    for (int a = 0; a < 3; ++a) sh.light[a] = p->light_pos[a];
    sh.kd = p->kd; sh.ks = p->ks; sh.ka = p->ka; sh.spec_alpha = p->spec_alpha;
    sh.light_intensity = p->light_intensity; sh.shadow_scale = p->shadow_scale; sh.eps = p->eps;
    sh.reflect_depth = p->reflect_depth; sh.shadow_bias = p->shadow_bias;
    RtFrame f;
    f.width = p->width; f.height = p->height;
    f.part_index = p->part_index; f.part_count = p->part_count;
    f.out_rows = rt_part_rows(p);
    f.walk_full = (p->walk_mode == RT_WALK_FULL);
    f.early_slack = 0.f;
    f.count_work = counting ? 1 : 0;
    f.shading_mode = p->shading_mode;
    if (counting && p->shading_mode != RT_SHADE_SERIAL) return set_err(RT_ERR_INVALID, "the counting build of the render kernel exists for RT_SHADE_SERIAL only");
    // A grid has at most RT_MAX_VOXELS voxels: from two references per voxel on average it is crowded for certain, and the
    // trace kernel without the path for voxels of a few records is used.
    f.own_path = (s->known_refs >= 2u * (uint32_t)RT_MAX_VOXELS) ? 0 : 1;
    if (s->own_path_override >= 0) f.own_path = s->own_path_override;
    // Frames in which most (run, ray) trips find a triangle (every ray of a tile on one flat patch: 60 hit trips per
    // streamed run on the generated plane, against 1 on a mesh seen from afar) go to the kernel variant that finishes a
    // run's hit trips together, lane = ray: 18-21 % faster there, 7-9 % slower elsewhere (its extra live values are spilled
    // registers in the loops). What an earlier frame of this scene counted decides; the first frames take the plain kernel.
    f.defer = 0;
    if (!f.own_path && !f.count_work && f.shading_mode == 0) {
        if (s->defer_override >= 0) f.defer = s->defer_override;
        else if (s->feedback) {
            const unsigned long long w = __atomic_load_n(s->feedback, __ATOMIC_RELAXED);
            const unsigned long long runs = w & 0xffffffffull, hit_trips = w >> 32;
            f.defer = (runs >= 1024ull && hit_trips >= (unsigned long long)RT_DEFER_HITS_PER_RUN * runs) ? 1 : 0;
        }
    }
    f.unit_shifts = s->sched_shifts;
    const bool in_flight = ws && s->n_sets > 1;
    f.adapt_units = s->sched_adapt ? (in_flight ? 2 : 1) : 0;
    f.cull_min = s->cull_min | (s->cull_lane_runs << 16);
    RtOutputs o;
    o.rgb_f32 = (float*)d_rgb_f32; o.rgb_u8 = (uint8_t*)d_rgb_u8; o.hit_id = (int32_t*)d_hit_id;
    o.counters = m.counters;
    o.stamps = stamps;
    o.unit_cost = nullptr; o.tile_cycles = nullptr; o.unit_info = nullptr;
    const RtCamera cam = make_camera(p);
    RtRenderWork wk;
    wk.est_by_tile = nullptr;
    {
        // tiles that can see the grid: full-frame tile rows ty0..ty1 -> local bands of this part
        int tx0, tx1, ty0, ty1;
        grid_tile_rect(cam, s->h_grid, p->width, p->height, tx0, tx1, ty0, ty1);
        const int pc = p->part_count, pi = p->part_index;
        int lb0 = (ty0 - pi + pc - 1) / pc;
        if (ty0 < pi) lb0 = 0;
        const int lb1 = (ty1 >= pi) ? (ty1 - pi) / pc : -1;
        wk.tx0 = tx0; wk.ntx = tx1 - tx0 + 1; wk.tb0 = lb0; wk.ntb = lb1 - lb0 + 1;
        if (wk.ntx < 0) wk.ntx = 0;
        if (wk.ntb < 0) wk.ntb = 0;
        const size_t tiles = frame_tiles(p);
        if (tiles > m.tile_cap) return set_err(RT_ERR_INVALID, "frame of %zu tiles exceeds the work-list capacity %zu", tiles, m.tile_cap);
        if (f.count_work) {
            HIP_TRY(hipMemsetAsync(m.unit_cost, 0, tiles * 112 * sizeof(uint32_t), side));
            // layout: 16 x tiles unit slots, the four estimate features and the measured cycles of every tile of the frame,
            // then 16 x tiles unit descriptions
            o.unit_cost = m.unit_cost; o.tile_cycles = m.unit_cost + 20 * tiles; o.unit_info = m.unit_cost + 21 * tiles; wk.est_by_tile = m.unit_cost + 16 * tiles;
        }
        wk.work = reinterpret_cast<uint32_t*>(m.counters + 24);
        wk.cost_sum = m.counters + 56;          // 32 words behind the work-list words in use (work[0..63])
        wk.cost = m.tile_cost;
        wk.lists = m.queue_lists; wk.list_cap = (uint32_t)tiles;
        wk.thr0 = s->sched_thr0; wk.thr1 = s->sched_thr1;
        wk.n_tiles = (int32_t)tiles;
        for (int k = 0; k < 4; ++k) wk.est_w[k] = s->est_w[k];
        wk.tile_list = m.tile_list;
        // The cost estimate only orders and sizes the work units; no pixel depends on it. A frame with the camera, the light,
        // the frame size and the triangles of the last frame that was estimated on this working set takes that frame's numbers.
        wk.cost_by_tile = m.cost_by_tile; wk.kept_total = m.kept_total;
        wk.reuse_estimate = 0;
        if (ws && m.cost_by_tile && m.kept_total && !f.count_work) {
            if (s->reuse_estimate && ws->est_valid && ws->est_version == s->version && same_view(ws->est_params, *p)) wk.reuse_estimate = 1;
            else { ws->est_params = *p; ws->est_version = s->version; ws->est_valid = true; }
        } else if (ws) ws->est_valid = false;
        // LDS for the occupancy bitmap: the grid of this triangle list was measured at upload
        wk.bitmap_words = (int32_t)((s->known_cells + 31) / 32);
        if (wk.bitmap_words < 1) wk.bitmap_words = 1;
        wk.n_cu = s->n_cu;
        const size_t lds = rt_trace_lds_bytes(wk.bitmap_words);
        int per_cu = (int)((160 * 1024) / lds);
        wk.blocks_per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
        // With several frames in flight a trace kernel that claims every wavefront slot of the GPU only makes the next
        // frame's kernel wait for its tail; at half the slots (a quarter for the sparse grids of small meshes, whose
        // units are short) two to four frames' kernels are resident together and the slots one frame's tail leaves idle
        // go to the next (measured, four frames in flight: 4K stand-in 0.72 -> 0.62 ms per frame, the 1080p frame of
        // configs[1] 0.133 -> 0.093 ms).
        if (in_flight) {
            const int cap = (s->blocks_per_cu_cap >= 1) ? s->blocks_per_cu_cap : (f.own_path ? 1 : 2);
            if (wk.blocks_per_cu > cap) wk.blocks_per_cu = cap;
        } else if (s->blocks_per_cu_cap >= 1 && s->blocks_per_cu_cap < wk.blocks_per_cu) wk.blocks_per_cu = s->blocks_per_cu_cap;
    }
    // without a build the first kernel of the frame is the tile classification, if any tile can see the grid at all
    const bool prep_runs = wk.ntx * wk.ntb > 0 && sd.n_tris != 0;
    wk.stamp_start = (build == 0 && prep_runs) ? stamps : nullptr;
    if (start_is_trace) *start_is_trace = build == 0 && !prep_runs;
    if (prep_runs && !wk.reuse_estimate) s->estimates_run++;
    HIP_TRY(rt_launch_render_prep(sd, cam, sh, f, wk, side));
    if (side != st) {
        HIP_TRY(hipEventRecord(ws->ev_join, side));
        HIP_TRY(hipStreamWaitEvent(st, ws->ev_join, 0));
    }
    HIP_TRY(rt_launch_trace(sd, cam, sh, f, o, wk, st));
    return RT_OK;
}

extern "C" int rt_render_device(rt_scene* s, const rt_params* p, int rebuild_grid, void* d_rgb_f32, void* d_rgb_u8, void* d_hit_id, void* stream) {
    if (!s) return set_err(RT_ERR_INVALID, "rt_render_device: NULL scene");
    int rc = check_params(p);
    if (rc) return rc;
    if ((rc = rt_scene_upload(s))) return rc;
    hipStream_t st = (hipStream_t)stream;
    WorkSet* w = nullptr;
    if ((rc = scene_enter_stream(s, st, &w))) return rc;
    const bool indexed = want_indexed(s, p->part_count);
    const bool rebuild = rebuild_grid || !w->grid_valid || w->grid_sorted != s->want_sorted() || w->grid_indexed != indexed;
    // working memory of the frame, grown when the triangle list or the frame size has grown
    const size_t tiles = frame_tiles(p);
    if ((rc = w->tile_list.ensure(tiles)) || (rc = w->tile_cost.ensure(tiles)) || (rc = w->queue_lists.ensure(tiles * RT_QUEUE_LISTS))) return rc;
    if (w->cost_by_tile.cap < w->tile_list.cap) { w->est_valid = false; if ((rc = w->cost_by_tile.ensure(w->tile_list.cap))) return rc; }
    if ((rc = w->kept_total.ensure(1))) return rc;
    if (s->count_work && (rc = w->unit_cost.ensure(tiles * 112))) return rc;
    if (rebuild && (rc = scene_ensure_refs(s, *w))) return rc;
    const int slot = (int)(s->frames % RT_STATS_RING);
    s->built[slot] = rebuild;
    int build = 0;
    if (rebuild) { build = w->build_clean ? 1 : 2; w->build_clean = false; }
    if ((rc = render_frame(s, p, scene_mem(s, *w, indexed), build, w, d_rgb_f32, d_rgb_u8, d_hit_id, st, s->stamps.p + 4 * slot, &s->start_is_trace[slot]))) return rc;
    if (rebuild) { w->grid_valid = true; w->build_clean = true; w->grid_sorted = s->want_sorted(); w->grid_indexed = indexed; }
    s->frames++;
    s->last_params = *p;
    return scene_leave_stream(s, w, st);
}

// Variant of rt_render_device() that allocates nothing: the working memory of the frame (grid, voxel tables, reference
// lists, triangle records in voxel order, run bounds, work lists, counters) is a fixed device array inside the library
// instead of buffers the scene allocates. The triangle arrays of the scene must already be on the device
// (rt_scene_upload, which also measures the reference count the limit is checked against). The grid is rebuilt on
// every call, because the array keeps whatever the last call on this GPU left there, of this scene or another one.
extern "C" int rt_render_device_noalloc(rt_scene* s, const rt_params* p, void* d_rgb_f32, void* d_rgb_u8, void* d_hit_id, void* stream) {
    if (!s) return set_err(RT_ERR_INVALID, "rt_render_device_noalloc: NULL scene");
    int rc = check_params(p);
    if (rc) return rc;
    if (!s->uploaded || s->device < 0 || !s->refs_known)
        return set_err(RT_ERR_INVALID, "rt_render_device_noalloc: call rt_scene_upload() first (this entry point allocates nothing)");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != s->device) return set_err(RT_ERR_INVALID, "scene is bound to device %d, current device is %d", s->device, dev);
    if (s->known_refs > RT_NOALLOC_MAX_REFS)
        return set_err(RT_ERR_INVALID, "rt_render_device_noalloc: the grid of this scene has %u (voxel, triangle) references, the fixed working memory holds %u",
                       s->known_refs, (unsigned)RT_NOALLOC_MAX_REFS);
    const size_t tiles = frame_tiles(p);
    if (tiles > RT_NOALLOC_MAX_TILES)
        return set_err(RT_ERR_INVALID, "rt_render_device_noalloc: a frame of %d x %d pixels has %zu tiles of 8x8, the fixed working memory holds %u",
                       p->width, rt_part_rows(p), tiles, (unsigned)RT_NOALLOC_MAX_TILES);
    if (!s->noalloc_base) return set_err(RT_ERR_INVALID, "rt_render_device_noalloc: call rt_scene_upload() first (this entry point allocates nothing)");
    unsigned char* base = (unsigned char*)s->noalloc_base;
    constexpr RtNoallocLayout L = rt_noalloc_layout();
    FrameMem m;
    m.grid = (RtGrid*)(base + L.grid); m.bounds_enc = (uint32_t*)(base + L.bounds_enc);
    m.cell_count = (uint32_t*)(base + L.cell_count); m.cell_start = (uint32_t*)(base + L.cell_start);
    m.block_sums = (uint32_t*)(base + L.block_sums); m.occupancy = (uint32_t*)(base + L.occupancy);
    m.recs = (float4*)(base + L.recs); m.blk_bounds = (float4*)(base + L.blk_bounds);
    m.ref_cap = RT_NOALLOC_MAX_REFS;
    m.sorted = false;
    m.counters = (unsigned long long*)(base + L.counters);
    m.tile_list = (uint32_t*)(base + L.tile_list); m.tile_cost = (uint32_t*)(base + L.tile_cost); m.queue_lists = (uint32_t*)(base + L.queue_lists);
    m.tile_cap = RT_NOALLOC_MAX_TILES;
    m.unit_cost = nullptr;
    m.report = false;
    // one stream, no events, no side stream: kernels and memsets only
    return render_frame(s, p, m, 2, nullptr, d_rgb_f32, d_rgb_u8, d_hit_id, (hipStream_t)stream, nullptr, nullptr);
}

// ---- several frames in flight ------------------------------------------------------------------
extern "C" int rt_scene_set_frames_in_flight(rt_scene* s, int32_t n) {
    if (!s || n < 1 || n > RT_MAX_IN_FLIGHT) return set_err(RT_ERR_INVALID, "rt_scene_set_frames_in_flight: 1..%d working sets", RT_MAX_IN_FLIGHT);
    if (s->device >= 0) {
        int dev = -1;
        HIP_TRY(hipGetDevice(&dev));
        if (dev != s->device) return set_err(RT_ERR_INVALID, "scene is bound to device %d, current device is %d", s->device, dev);
        HIP_TRY(hipDeviceSynchronize());
        for (int i = n; i < RT_MAX_IN_FLIGHT; ++i) s->sets[i].release();
    }
    // the streams seen so far are forgotten: the next frame on any stream takes the first set
    for (WorkSet& w : s->sets) { w.stream_used = false; w.last_stream = nullptr; }
    s->n_sets = n;
    if (s->cur >= n) s->cur = 0;
    return RT_OK;
}

extern "C" int32_t rt_scene_frames_in_flight(const rt_scene* s) { return s ? s->n_sets : 0; }

// Streams that run side by side. The HIP runtime spreads the streams of a process over a few hardware queues (four by
// default) and work of two streams on one queue runs in order, as if it were one stream; which queue a new stream lands
// on depends on what the process created before. So candidates are created one at a time and each is kept only if a short
// wait kernel on it overlaps in time with the same kernel on every stream kept so far.
extern "C" int32_t rt_streams_create(int32_t n, void** streams) {
    if (n < 1 || n > RT_MAX_IN_FLIGHT || !streams) return set_err(RT_ERR_INVALID, "rt_streams_create: 1..%d streams", RT_MAX_IN_FLIGHT);
    {
        int n_dev = 0;
        hipError_t e0 = hipGetDeviceCount(&n_dev);
        if (e0 != hipSuccess || n_dev <= 0) return set_err(RT_ERR_NO_DEVICE, "no HIP device: %s", e0 == hipSuccess ? "count is 0" : hipGetErrorString(e0));
    }
    const int max_candidates = 4 * RT_MAX_IN_FLIGHT;
    const unsigned ticks = 20000;                 // 0.2 ms: long against the time between the two launches
    hipStream_t kept[RT_MAX_IN_FLIGHT], rejected[4 * RT_MAX_IN_FLIGHT];
    int n_kept = 0, n_rej = 0;
    unsigned long long* d_stamps = nullptr;
    hipError_t e = hipMalloc((void**)&d_stamps, 4 * sizeof(unsigned long long));
    for (int c = 0; c < max_candidates && n_kept < n && e == hipSuccess; ++c) {
        hipStream_t cand = nullptr;
        if ((e = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking)) != hipSuccess) break;
        bool beside_all = true;
        for (int k = 0; k < n_kept && beside_all && e == hipSuccess; ++k) {
            unsigned long long t[4] = {0, 0, 0, 0};
            e = rt_launch_probe_wait(d_stamps, ticks, kept[k]);
            if (e == hipSuccess) e = rt_launch_probe_wait(d_stamps + 2, ticks, cand);
            if (e == hipSuccess) e = hipStreamSynchronize(kept[k]);
            if (e == hipSuccess) e = hipStreamSynchronize(cand);
            if (e == hipSuccess) e = hipMemcpy(t, d_stamps, sizeof(t), hipMemcpyDeviceToHost);
            beside_all = e == hipSuccess && t[2] < t[1] && t[0] < t[3];       // the two intervals intersect
        }
        if (beside_all && e == hipSuccess) kept[n_kept++] = cand;
        else rejected[n_rej++] = cand;           // destroyed at the end: freeing a queue slot now would send the next candidate to it
    }
    for (int k = 0; k < n_rej; ++k) (void)hipStreamDestroy(rejected[k]);
    if (d_stamps) (void)hipFree(d_stamps);
    if (e != hipSuccess) {
        for (int k = 0; k < n_kept; ++k) (void)hipStreamDestroy(kept[k]);
        return set_err(RT_ERR_HIP, "rt_streams_create: %s", hipGetErrorString(e));
    }
    for (int k = 0; k < n_kept; ++k) streams[k] = (void*)kept[k];
    return n_kept;
}

extern "C" int rt_streams_destroy(int32_t n, void* const* streams) {
    if (n < 0 || (n > 0 && !streams)) return set_err(RT_ERR_INVALID, "rt_streams_destroy: bad argument");
    for (int k = 0; k < n; ++k)
        if (streams[k]) HIP_TRY(hipStreamDestroy((hipStream_t)streams[k]));
    return RT_OK;
}

extern "C" int rt_render(rt_scene* s, const rt_params* p, float* rgb_f32, uint8_t* rgb_u8, int32_t* hit_id) {
    if (!s) return set_err(RT_ERR_INVALID, "rt_render: NULL scene");
    int rc = check_params(p);
    if (rc) return rc;
    if ((rc = rt_scene_upload(s))) return rc;
    const size_t px = (size_t)p->width * (size_t)rt_part_rows(p);
    if (rgb_f32 && (rc = s->o_rgb.ensure(3 * px))) return rc;
    if (rgb_u8 && (rc = s->o_u8.ensure(3 * px))) return rc;
    if (hit_id && (rc = s->o_id.ensure(px))) return rc;
    rc = rt_render_device(s, p, 0, rgb_f32 ? s->o_rgb.p : nullptr, rgb_u8 ? s->o_u8.p : nullptr, hit_id ? s->o_id.p : nullptr, nullptr);
    if (rc) return rc;
    if (rgb_f32) HIP_TRY(hipMemcpy(rgb_f32, s->o_rgb.p, 3 * px * sizeof(float), hipMemcpyDeviceToHost));
    if (rgb_u8) HIP_TRY(hipMemcpy(rgb_u8, s->o_u8.p, 3 * px, hipMemcpyDeviceToHost));
    if (hit_id) HIP_TRY(hipMemcpy(hit_id, s->o_id.p, px * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipDeviceSynchronize());
    RtGrid g;
    HIP_TRY(hipMemcpy(&g, s->sets[s->cur].d_grid.p, sizeof(g), hipMemcpyDeviceToHost));
    if (g.overflow) return set_err(RT_ERR_HIP, "grid reference capacity exceeded");
    return RT_OK;
}

extern "C" int64_t rt_selftest_division(const double* num, const double* den, int64_t n) {
    if (n < 0 || (n > 0 && (!num || !den)) || n > (int64_t)1 << 30) return set_err(RT_ERR_INVALID, "rt_selftest_division: bad argument");
    {
        int n_dev = 0;
        hipError_t e0 = hipGetDeviceCount(&n_dev);
        if (e0 != hipSuccess || n_dev <= 0) return set_err(RT_ERR_NO_DEVICE, "no HIP device: %s", e0 == hipSuccess ? "count is 0" : hipGetErrorString(e0));
    }
    if (n == 0) return 0;
    double *d_num = nullptr, *d_den = nullptr;
    uint32_t* d_bad = nullptr;
    uint32_t bad = 0;
    hipError_t e = hipMalloc(&d_num, n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_den, n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_bad, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(d_num, num, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_den, den, n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_bad, 0, sizeof(uint32_t));
    if (e == hipSuccess) e = rt_launch_selftest_div(d_num, d_den, (uint32_t)n, d_bad, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_num); (void)hipFree(d_den); (void)hipFree(d_bad);
    if (e != hipSuccess) return set_err(RT_ERR_HIP, "rt_selftest_division: %s", hipGetErrorString(e));
    return (int64_t)bad;
}

extern "C" int rt_scene_set_estimate_reuse(rt_scene* s, int32_t enable) {
    if (!s) return set_err(RT_ERR_INVALID, "rt_scene_set_estimate_reuse: NULL scene");
    s->reuse_estimate = enable != 0;
    for (WorkSet& w : s->sets) w.est_valid = false;
    return RT_OK;
}

extern "C" int64_t rt_scene_estimates_run(const rt_scene* s) { return s ? (int64_t)s->estimates_run : -1; }

extern "C" int rt_scene_set_counting(rt_scene* s, int32_t enable) {
    if (!s) return set_err(RT_ERR_INVALID, "rt_scene_set_counting: NULL scene");
    s->count_work = enable != 0;
    return RT_OK;
}

extern "C" int rt_scene_last_stats(rt_scene* s, rt_frame_stats* out) { return rt_scene_stats_at(s, 0, out); }

extern "C" int rt_scene_test_feedback(rt_scene* s, uint64_t* runs, uint64_t* hit_trips, int* deferred_variant) {
    if (!s) return set_err(RT_ERR_INVALID, "rt_scene_test_feedback: NULL scene");
    const unsigned long long w = s->feedback ? __atomic_load_n(s->feedback, __ATOMIC_RELAXED) : 0ull;
    if (runs) *runs = w & 0xffffffffull;
    if (hit_trips) *hit_trips = w >> 32;
    if (deferred_variant) {
        *deferred_variant = s->defer_override >= 0 ? s->defer_override
                                                   : (((w & 0xffffffffull) >= 1024ull && (w >> 32) >= (unsigned long long)RT_DEFER_HITS_PER_RUN * (w & 0xffffffffull)) ? 1 : 0);
    }
    return RT_OK;
}

extern "C" int rt_scene_unit_costs(rt_scene* s, uint32_t* out, int64_t n) {
    if (!s || !out || n < 0) return set_err(RT_ERR_INVALID, "rt_scene_unit_costs: bad argument");
    const WorkSet& w = s->sets[s->cur];
    if (!w.unit_cost.p) return set_err(RT_ERR_INVALID, "no counting frame rendered yet");
    HIP_TRY(hipDeviceSynchronize());
    const size_t k = (size_t)n < w.unit_cost.cap ? (size_t)n : w.unit_cost.cap;
    HIP_TRY(hipMemcpy(out, w.unit_cost.p, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return (int)k;
}

extern "C" int rt_scene_stats_at(rt_scene* s, int32_t frames_back, rt_frame_stats* out) {
    if (!s || !out) return set_err(RT_ERR_INVALID, "rt_scene_stats_at: bad argument");
    if (s->frames == 0) return set_err(RT_ERR_INVALID, "no frame rendered yet");
    if (frames_back < 0 || frames_back >= RT_STATS_RING || (uint64_t)frames_back >= s->frames) return set_err(RT_ERR_INVALID, "frame %d is not in the stats ring", frames_back);
    const int slot = (int)((s->frames - 1 - (uint64_t)frames_back) % RT_STATS_RING);
    memset(out, 0, sizeof(*out));
    {
        int rc = scene_sync_streams(s);
        if (rc) return rc;
    }
    {
        // ticks of the 100 MHz real-time clock, written by the frame's kernels
        unsigned long long t[4];
        HIP_TRY(hipMemcpy(t, s->stamps.p + 4 * slot, sizeof(t), hipMemcpyDeviceToHost));
        if (s->start_is_trace[slot]) t[0] = t[1];
        const double ms_per_tick = 1e-5;
        if (t[1] >= t[0]) out->ms_build = (float)((double)(t[1] - t[0]) * ms_per_tick);
        if (t[2] >= t[1]) out->ms_render = (float)((double)(t[2] - t[1]) * ms_per_tick);
        if (t[2] >= t[0]) out->ms_total = (float)((double)(t[2] - t[0]) * ms_per_tick);
    }
    if (!s->built[slot]) out->ms_build = 0.f;
    RtGrid g;
    unsigned long long c[24];
    HIP_TRY(hipMemcpy(&g, s->sets[s->cur].d_grid.p, sizeof(g), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c, s->sets[s->cur].counters.p, sizeof(c), hipMemcpyDeviceToHost));
    for (int a = 0; a < 3; ++a) out->grid_n[a] = g.n[a];
    out->n_cells = (uint32_t)(g.n[0] * g.n[1] * g.n[2]);
    out->n_refs = g.n_refs;
    out->n_triangles = s->tri_mesh.size();
    // rows of this part that lie inside the frame
    const rt_params& p = s->last_params;
    uint64_t rows = 0;
    const int bands = (p.height + RT_BAND_ROWS - 1) / RT_BAND_ROWS;
    for (int b = p.part_index; b < bands; b += p.part_count) {
        int r0 = b * RT_BAND_ROWS, r1 = r0 + RT_BAND_ROWS;
        if (r1 > p.height) r1 = p.height;
        rows += (uint64_t)(r1 - r0);
    }
    out->primary_rays = rows * (uint64_t)p.width;
    out->hits = c[0];
    out->shadowed = c[1];
    return RT_OK;
}

extern "C" int rt_scene_record_order(rt_scene* s, double area[2], int32_t* sorted) {
    if (!s || !area || !sorted) return set_err(RT_ERR_INVALID, "rt_scene_record_order: bad argument");
    area[0] = s->run_area[0]; area[1] = s->run_area[1];
    *sorted = s->want_sorted() ? 1 : 0;
    return RT_OK;
}

extern "C" int rt_scene_triangle_order(rt_scene* s, double* area_spatial, int32_t* spatial) {
    if (!s || !area_spatial || !spatial) return set_err(RT_ERR_INVALID, "rt_scene_triangle_order: bad argument");
    *area_spatial = s->run_area_spatial;
    *spatial = (s->use_order && s->n_tris() > 65536) ? 1 : 0;
    return RT_OK;
}

extern "C" int rt_scene_last_counts(rt_scene* s, rt_frame_counts* out) {
    if (!s || !out) return set_err(RT_ERR_INVALID, "rt_scene_last_counts: bad argument");
    if (s->frames == 0) return set_err(RT_ERR_INVALID, "no frame rendered yet");
    memset(out, 0, sizeof(*out));
    {
        int rc = scene_sync_streams(s);
        if (rc) return rc;
    }
    unsigned long long c[24];
    HIP_TRY(hipMemcpy(c, s->sets[s->cur].counters.p, sizeof(c), hipMemcpyDeviceToHost));
    out->primary_entering = c[2]; out->steps_primary = c[3]; out->tests_primary = c[4];
    out->shadow_entering = c[5]; out->steps_shadow = c[6]; out->tests_shadow = c[7];
    out->sched_rounds = c[8]; out->sched_streams = c[9]; out->sched_lanes = c[10];
    out->tile_cycles_max = c[11]; out->tile_cycles_sum = c[12]; out->tiles_traced = c[13]; out->tile_rounds_max = c[14]; out->cycles_test = c[15]; out->tests_executed = c[16];
    out->runs_streamed = c[17]; out->run_ray_trips = c[18]; out->cycles_cull = c[19];
    out->wave_cycles_max = c[21]; out->span_cycles = (c[23] > c[22]) ? c[23] - c[22] : 0;
    return RT_OK;
}

extern "C" int rt_loop_counts(uint64_t* out, int32_t n, int32_t reset) {
    if (!out || n < 0) return set_err(RT_ERR_INVALID, "rt_loop_counts: bad argument");
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long tmp[64] = {};
    HIP_TRY(rt_read_loop_counts(tmp, 64, reset));
    for (int k = 0; k < n && k < 64; ++k) out[k] = tmp[k];
    return RT_OK;
}

extern "C" int rt_deinterleave_device(const rt_params* p, const void* d_src, void* d_dst, int32_t elem_bytes, void* stream) {
    int rc = check_params(p);
    if (rc) return rc;
    if (!d_src || !d_dst || elem_bytes <= 0) return set_err(RT_ERR_INVALID, "rt_deinterleave_device: bad argument");
    HIP_TRY(rt_launch_deinterleave(d_src, d_dst, p->width, p->height, p->part_count, rt_part_rows(p), elem_bytes, (hipStream_t)stream));
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------
// image output, Serial/raytracer.cpp:178-185
// ------------------------------------------------------------------------------------------------
extern "C" int rt_write_ppm_u8(const char* path, const uint8_t* rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) return set_err(RT_ERR_INVALID, "rt_write_ppm_u8: bad argument");
    FILE* f = fopen(path, "wb");
    if (!f) return set_err(RT_ERR_IO, "cannot open %s for writing", path);
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    size_t n = (size_t)w * h * 3;
    bool ok = fwrite(rgb, 1, n, f) == n;
    ok = (fclose(f) == 0) && ok;
    return ok ? RT_OK : set_err(RT_ERR_IO, "short write to %s", path);
}

static int write_ppm_impl(const char* path, const float* rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) return set_err(RT_ERR_INVALID, "rt_write_ppm: bad argument");
    size_t n = (size_t)w * h * 3;
    std::vector<uint8_t> b(n);
    for (size_t i = 0; i < n; ++i) {
        float q = rgb[i] / 255;
        float m = (q < 1.f) ? q : 1.f;          // std::min(float(1), q): a NaN gives 1
        // (unsigned char)(float) of a negative value is undefined in C++; what the reference's x86 code does is convert to
        // a 32-bit integer (out of range: 0x80000000) and keep the low byte
        const float v = m * 255;
        const int32_t iv = (v > -2147483648.f) ? (int32_t)v : INT32_MIN;
        b[i] = (uint8_t)(uint32_t)iv;
    }
    return rt_write_ppm_u8(path, b.data(), w, h);
}

extern "C" int rt_write_ppm(const char* path, const float* rgb, int32_t w, int32_t h) { RT_GUARD(int, write_ppm_impl(path, rgb, w, h)) }

static int render_ppm_impl(rt_scene* s, const rt_params* p, const char* path) {
    if (!path) return set_err(RT_ERR_INVALID, "rt_render_ppm: NULL path");
    int rc = check_params(p);
    if (rc) return rc;
    if (p->part_count != 1) return set_err(RT_ERR_INVALID, "rt_render_ppm renders whole frames only");
    std::vector<uint8_t> img((size_t)p->width * p->height * 3);
    if ((rc = rt_render(s, p, nullptr, img.data(), nullptr))) return rc;
    return rt_write_ppm_u8(path, img.data(), p->width, p->height);
}

extern "C" int rt_render_ppm(rt_scene* s, const rt_params* p, const char* path) { RT_GUARD(int, render_ppm_impl(s, p, path)) }

// ------------------------------------------------------------------------------------------------
// plane mesh generator: same OBJ text as plane_mesh_creator.py:23-80
// ------------------------------------------------------------------------------------------------
// round(x, ndigits) of the generator's stepper: correctly rounded decimal rounding of the exact binary value.
static double py_round(double x, int ndigits) {
    char tmp[64];
    snprintf(tmp, sizeof(tmp), "%.*f", ndigits, x);
    return strtod(tmp, nullptr);
}

static int make_plane_obj_impl(const char* path, int32_t squares_per_unit) {
    if (!path || squares_per_unit < 1) return set_err(RT_ERR_INVALID, "rt_make_plane_obj: bad argument");
    const int round_amount = (int)log10((double)squares_per_unit);        // plane_mesh_creator.py:20
    const double step = 1.0 / squares_per_unit;                           // :21
    // float_stepper(-5 + step, 5 + step, step), :23-27, with the plane corner -5 in front (:35,39)
    std::vector<double> sl;
    sl.push_back(-5.0);
    for (double v = -5.0 + step; v <= 5.0 + step;) {
        v = py_round(v, round_amount);
        sl.push_back(v);
        v += step;
    }
    const size_t n = sl.size();          // slices per axis; (n-1)^2 squares
    for (size_t i = 1; i < n; ++i)
        if (!(sl[i] > sl[i - 1])) return set_err(RT_ERR_INVALID, "squares_per_unit %d does not give distinct slices", squares_per_unit);
    FILE* f = fopen(path, "wb");
    if (!f) return set_err(RT_ERR_IO, "cannot open %s for writing", path);
    std::vector<char> iobuf(1 << 22);
    setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
    // vertices in order of first appearance over the squares (x-major, z-minor, corners A B C D), :52-58
    std::vector<int32_t> index(n * n, -1);
    int32_t count = 0;
    auto touch = [&](size_t ix, size_t iz) {
        if (index[ix * n + iz] < 0) {
            index[ix * n + iz] = count++;
            fprintf(f, "v %f %f %f\n", sl[ix], -1.0, sl[iz]);
        }
    };
    for (size_t ix = 0; ix + 1 < n; ++ix)
        for (size_t iz = 0; iz + 1 < n; ++iz) {
            touch(ix, iz); touch(ix + 1, iz); touch(ix + 1, iz + 1); touch(ix, iz + 1);
        }
    fprintf(f, "vt 0 0\n");
    // faces (A,C,B) and (A,D,C), :60-69, 1-based, texture index 1
    for (size_t ix = 0; ix + 1 < n; ++ix)
        for (size_t iz = 0; iz + 1 < n; ++iz) {
            int a = index[ix * n + iz] + 1, b = index[(ix + 1) * n + iz] + 1, c = index[(ix + 1) * n + iz + 1] + 1, d = index[ix * n + iz + 1] + 1;
            fprintf(f, "f %d/1 %d/1 %d/1\n", a, c, b);
            fprintf(f, "f %d/1 %d/1 %d/1\n", a, d, c);
        }
    bool ok = !ferror(f);
    ok = (fclose(f) == 0) && ok;
    return ok ? RT_OK : set_err(RT_ERR_IO, "write to %s failed", path);
}

extern "C" int rt_make_plane_obj(const char* path, int32_t squares_per_unit) { RT_GUARD(int, make_plane_obj_impl(path, squares_per_unit)) }
