/* rt_amd.h - C ABI of the MI355X-native per-pixel render path.
 *
 * Drop-in boundary for the two free functions of the reference CPU renderer,
 *     void load_mesh(const char*, std::vector<Triangle*>&, bool format_has_vt, Vec3f mesh_color,
 *                    bool reflective, Vec3f offset)        (Serial/raytracer.cpp:189,220-287)
 *     void render(const std::vector<Triangle*>&)           (Serial/raytracer.cpp:120-187)
 * and everything below them: GridAccel build (Serial/grid.h:79-153), fast_trace
 * (Serial/raytracer.cpp:71-118), GridAccel::Intersect (Serial/grid.h:167-231),
 * Triangle::Intersect (Serial/geometry.h:131-177), the PPM writer (Serial/raytracer.cpp:178-185).
 *
 * Plain C types only. All entry points return RT_OK (0) or a negative rt_status; none of them
 * falls back to a CPU implementation: without a usable HIP device every compute call fails with
 * RT_ERR_NO_DEVICE / RT_ERR_HIP.
 */
#ifndef RT_AMD_H
#define RT_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* NULL handle, bad size, bad enum */
    RT_ERR_IO = -2,          /* cannot create / write an output file */
    RT_ERR_OBJ = -3,         /* face index outside the vertices read so far (the reference reads out of bounds) */
    RT_ERR_HIP = -4,         /* a HIP runtime call failed; rt_last_error() has the text */
    RT_ERR_NO_DEVICE = -5,   /* no HIP device visible */
    RT_ERR_NOMEM = -6
} rt_status;

enum { RT_WALK_EARLY = 0, RT_WALK_FULL = 1 };
/* Which fast_trace: the Serial tree's (Serial/raytracer.cpp:71-118; the parity target, pinned against the compiled
 * reference) or the Parallel tree's (Parallel/raytracer.cu:445-524: per-mesh material table, nearest hit with t > eps,
 * shadow ray towards the light, mirror reflection to depth 3). The Parallel tree is CUDA and cannot be compiled where this
 * library is tested, so the second mode is checked against a CPU restatement only. */
enum { RT_SHADE_SERIAL = 0, RT_SHADE_PARALLEL = 1 };

/* Opaque scene: the triangle list that load_mesh() appends to, plus its device copy and grid. */
typedef struct rt_scene rt_scene;

/* Everything render() hard-codes (Serial/raytracer.cpp:83-89,124-128; eps Serial/geometry.h:2).
 * rt_params_default() fills in exactly those literals. */
typedef struct rt_params {
    int32_t width, height;            /* 512 x 512 */
    float cam_pos[3];                 /* (3,5,3) */
    float cam_target[3];              /* (0,0,0) */
    float cam_up[3];                  /* (0,-1,0) */
    float fov;                        /* 45 */
    float light_pos[3];               /* (5,-5,2) */
    float kd, ks, ka;                 /* 2, 5e11, 0.2 */
    float spec_alpha;                 /* 4; only 4 is bit-pinned to the reference (pow in double) */
    float light_intensity;            /* 255 */
    float shadow_scale;               /* 0.1 */
    float eps;                        /* 0.1f: shadow-ray start and the use_eps gate */
    int32_t walk_mode;                /* RT_WALK_EARLY (default): the primary walk stops once the nearest hit
                                         lies before the next voxel crossing; RT_WALK_FULL: walk to the grid
                                         exit like the reference. Same image apart from rounding-level ties. */
    /* Screen partition for multi-GPU runs: the frame is cut into bands of RT_BAND_ROWS rows and band b
     * belongs to part (b % part_count). The call renders only the bands of part_index and writes them
     * stacked (local band j = global band part_index + j*part_count) into the output buffers, which
     * hold rt_part_rows() rows. part_count = 1 renders the whole frame. */
    int32_t part_index, part_count;
    /* RT_SHADE_PARALLEL only (ignored otherwise). With it, kd / ks / ka / spec_alpha / light_intensity above are unused (the
     * materials carry them), shadow_scale is the factor of a shadowed colour (0.5 at Parallel/raytracer.cu:505) and eps the
     * hit threshold (1e-4 at Parallel/geometry.cuh:3). rt_params_default() leaves the Serial values in those two. */
    int32_t shading_mode;             /* RT_SHADE_SERIAL */
    int32_t reflect_depth;            /* 3: REFLECT_DEPTH, Parallel/raytracer.cu:16 (at most 3) */
    float shadow_bias;                /* 0.02f: the shadow ray starts at eps + 0.02f, Parallel/raytracer.cu:494 */
} rt_params;

/* One entry of the material table of the Parallel tree's fast_trace (Parallel/geometry.cuh:284-303, raytracer.cu:450-453);
 * here a material travels with the mesh it is loaded for instead of being compiled into the shader. */
typedef struct rt_material {
    float base_color[3];
    float kd, ks, spec_alpha, ka, km;
    int32_t reflective;
} rt_material;

#define RT_BAND_ROWS 8
#define RT_STATS_RING 128

/* Device-side timings (the frame's kernels write the GPU's 100 MHz real-time clock at their start and end) and the ray
 * counts of the last frame. */
typedef struct rt_frame_stats {
    float ms_build;                   /* grid build: bounds, count, scan, scatter, sort/record emit */
    float ms_render;                  /* render kernel incl. framebuffer writes */
    float ms_total;                   /* first build kernel to end of render */
    int32_t grid_n[3];
    uint32_t n_cells, n_refs;
    uint64_t n_triangles;
    uint64_t primary_rays, hits, shadowed;
} rt_frame_stats;

const char* rt_last_error(void);
const char* rt_version(void);

void rt_params_default(rt_params* p);
/* Rows held by the output buffers: height when part_count == 1, otherwise RT_BAND_ROWS times the bands
 * of the largest part (the same for every part, so that a gather can use equal counts). */
int32_t rt_part_rows(const rt_params* p);

/* ---- scene: replaces the std::vector<Triangle*> that load_mesh() fills ---- */
rt_scene* rt_scene_create(void);
void rt_scene_destroy(rt_scene* s);
/* load_mesh (Serial/raytracer.cpp:220-287): same record handling, same double->float rounding of
 * (coordinate + offset), 1-based face indices, triangle index = order of the f lines over all calls.
 * A missing file adds nothing and returns 0 like the reference. Returns triangles added or <0. */
int64_t rt_scene_add_obj(rt_scene* s, const char* path, int format_has_vt, const float color[3], const float offset[3]);
/* load_mesh of the Parallel tree (Parallel/raytracer.cu:805-873): vertex = scale * (coordinate + offset), evaluated in double
 * and rounded to float, and a material instead of a colour. With scale 1 the vertices are those of rt_scene_add_obj.
 * Meshes added without a material get the table's entry 1 (kd 10, ks 10, spec_alpha 1.25, ka 0.3, not reflective) around
 * their colour. */
int64_t rt_scene_add_obj_material(rt_scene* s, const char* path, int format_has_vt, const rt_material* material, const float offset[3], int32_t scale);
int64_t rt_scene_add_triangles_material(rt_scene* s, const float* v9, int64_t n, const rt_material* material);
/* Same, for a mesh already in memory: n triangles, 9 floats each (v0 v1 v2). */
int64_t rt_scene_add_triangles(rt_scene* s, const float* v9, int64_t n, const float color[3]);
int64_t rt_scene_num_triangles(const rt_scene* s);
/* Copies the host triangle list out (9 floats per triangle). */
int rt_scene_get_triangles(const rt_scene* s, float* v9);

/* Uploads the triangle list to the current HIP device as SoA float4 arrays (the scene upload of
 * render()'s caller side; replaces the per-Triangle heap objects of Serial/raytracer.cpp:273-282). */
int rt_scene_upload(rt_scene* s);
/* GridAccel::GridAccel (Serial/grid.h:79-153) on the GPU: bounds, resolution, count, prefix scan,
 * scatter, per-cell ascending order. Runs asynchronously on `stream` (a hipStream_t, may be NULL). */
int rt_scene_build_grid(rt_scene* s, void* stream);
/* Grid read-back for verification: n[3], lo/hi[6], width[3], inv_width[3]. */
int rt_scene_get_grid(rt_scene* s, int32_t n[3], float lohi[6], float width[3], float inv_width[3], uint32_t* n_cells, uint32_t* n_refs);
/* CSR read-back: cell_start[n_cells+1], cell_tris[n_refs] (ascending triangle index inside each cell). */
int rt_scene_get_csr(rt_scene* s, uint32_t* cell_start, uint32_t* cell_tris);

/* ---- render(): Serial/raytracer.cpp:120-187 ---- */
/* Device-resident variant: grid build (unless already built for this triangle list and rebuild == 0),
 * render kernel and framebuffer writes, all asynchronous on `stream`. Any of the three device
 * pointers may be NULL. d_rgb_f32: 3 floats per pixel (the Vec3f image of :141), d_rgb_u8: the PPM
 * body bytes of :181-183, d_hit_id: primary-hit triangle index or -1. Row-major, rt_part_rows() rows. */
int rt_render_device(rt_scene* s, const rt_params* p, int rebuild_grid, void* d_rgb_f32, void* d_rgb_u8, void* d_hit_id, void* stream);
/* rt_render_device() for callers that cannot allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc. No counterpart in the reference. The call allocates nothing, neither on the device nor
 * through any allocating library routine: the working memory of the frame (grid description, voxel tables, reference
 * lists, triangle records in voxel order, run bounds, work lists, counters) is a fixed array of about 32 MB that every GPU
 * the library runs on holds until the process ends. It issues kernels and memsets on `stream` only (no events, no second
 * stream, no host synchronisation) and rebuilds the grid on every call. The scene's triangles must already be on the
 * device: call rt_scene_upload() once, outside the code that cannot allocate and outside any stream capture;
 * rt_scene_upload() also looks up the device address of the fixed array (a module load under a lock), so that this entry
 * point only reads a stored pointer and its very first call on a device may already be under capture. Limits, rejected with RT_ERR_INVALID and a
 * message: 524,288 (voxel, triangle) references in the scene's grid; 32,640 tiles of 8x8 pixels in the output buffers
 * (1920 x 1088 pixels). All calls on one GPU share the one array, which keeps the last call's contents: calls that overlap
 * in time on one GPU (two streams, two threads) race; calls on one stream, or ordered by the caller, are safe, also for
 * different scenes. No frame statistics are recorded for these frames. */
int rt_render_device_noalloc(rt_scene* s, const rt_params* p, void* d_rgb_f32, void* d_rgb_u8, void* d_hit_id, void* stream);
/* ---- several frames in flight (no counterpart in the reference, which renders one frame and exits) ----
 * A frame is a chain of about a dozen kernels (grid build, tile lists, trace) of which the first ones are short and narrow
 * and the last one ends in a tail of a few long wavefronts: one frame at a time leaves a good part of the GPU idle. With n
 * working sets (grid tables, reference lists, work lists; the triangles are shared) a scene keeps up to n frames in flight:
 * rt_render_device() calls on different streams use different sets and overlap on the GPU, calls on one stream share a set
 * and are ordered by the stream as before. Default 1 (the call also waits for the device and forgets which stream used
 * which set). A stream handed to a scene has to stay valid until the scene is destroyed or this call is made again.
 * Output buffers belong to the caller: frames in flight need one each. */
int rt_scene_set_frames_in_flight(rt_scene* s, int32_t n);        /* 1..8 */
/* Before it traces a frame the library estimates the cost of every 8x8 tile that sees the grid (sample rays walked
 * through the grid, no triangle test) to order and size the work units of the render kernel; no pixel depends on the
 * estimate. A frame with the camera, light, frame size, part and triangle list of the last estimated frame on its
 * working set reuses that frame's numbers instead of estimating again (enable = 1, the default; RT_REUSE_ESTIMATE=0 in
 * the environment of rt_scene_create() switches it off as well). enable = 0: every frame estimates. No counterpart in the
 * reference. */
int rt_scene_set_estimate_reuse(rt_scene* s, int32_t enable);
int32_t rt_scene_frames_in_flight(const rt_scene* s);
/* Up to n streams (hipStream_t, non-blocking) that really run side by side: the HIP runtime maps the streams of a process
 * onto a few hardware queues (four by default), and two streams on one queue execute in order. Each stream returned has
 * been checked, with a short kernel, to overlap with every other one. Returns how many were found (at least 1, at most n)
 * or a negative rt_status. */
int32_t rt_streams_create(int32_t n, void** streams);
int rt_streams_destroy(int32_t n, void* const* streams);
/* Host-buffer variant (uploads if needed, renders, copies back, synchronises). Buffers may be NULL. */
int rt_render(rt_scene* s, const rt_params* p, float* rgb_f32, uint8_t* rgb_u8, int32_t* hit_id);
/* render() including its file output: writes "P6\nW H\n255\n" + body to `path` ("./test.ppm" in the reference). */
int rt_render_ppm(rt_scene* s, const rt_params* p, const char* path);
/* Stats of the last rt_render_device()/rt_render() on this scene; waits for the scene's stream. Work counters and
 * scheduling diagnostics of the counting build of the kernels are declared in rt_amd_diag.h. */
int rt_scene_last_stats(rt_scene* s, rt_frame_stats* out);
/* Timings of the frame rendered `frames_back` frames before the last one (0 = last; a ring of RT_STATS_RING
 * frames is kept, so a benchmark can time a whole run without synchronising inside it). Ray counts are
 * those of the last frame. */
int rt_scene_stats_at(rt_scene* s, int32_t frames_back, rt_frame_stats* out);
/* Self-test of the arithmetic of the triangle test (no counterpart in the reference): the kernels divide three
 * numerators by one denominator with a shared reciprocal refinement where the operands' exponents are within 2^+-300;
 * this runs that form and the plain division on the n pairs (num[i], den[i]) on the GPU and returns the number of
 * pairs whose results differ in any bit (pairs outside the exponent window are skipped), or a negative rt_status. */
int64_t rt_selftest_division(const double* num, const double* den, int64_t n);
/* Band de-interleave after a gather: src holds part_count stacked partial frames of rt_part_rows() rows
 * each (rank-major), dst is the full row-major frame. elem_bytes = bytes per pixel (12, 3 or 4). */
int rt_deinterleave_device(const rt_params* p, const void* d_src, void* d_dst, int32_t elem_bytes, void* stream);

/* ---- multi-GPU exchange step (no counterpart in the reference, which runs on one thread): one process drives several
 * GPUs, every GPU renders the bands of its part (rt_render_device with part_index = its place in `devices`, part_count =
 * n_devices, on a scene uploaded to that GPU), and one RCCL gather brings the bands to the first GPU, each peer over its
 * own xGMI link. RCCL is loaded at run time; without it rt_comm_create fails with RT_ERR_HIP. */
typedef struct rt_comm rt_comm;
/* One RCCL communicator per listed device (ncclCommInitAll); devices == NULL means 0 .. n_devices-1. */
int rt_comm_create(int32_t n_devices, const int32_t* devices, rt_comm** out);
void rt_comm_destroy(rt_comm* c);
int32_t rt_comm_size(const rt_comm* c);
/* d_parts[r]: the band buffer of part r on device r (rt_part_rows() rows of width * elem_bytes bytes); d_stacked
 * (n_devices band buffers) and d_frame (the whole frame) live on the first device. The gather is enqueued on streams[r]
 * of every device (NULL = their default streams) and rt_deinterleave_device() behind it on the first device's stream;
 * the call does not wait for the GPUs. p->part_count must equal the communicator's size. */
int rt_gather_bands(rt_comm* c, const rt_params* p, void* const* d_parts, void* d_stacked, void* d_frame, int32_t elem_bytes, void* const* streams);

/* PPM writer of Serial/raytracer.cpp:178-185 from a float image. */
int rt_write_ppm(const char* path, const float* rgb, int32_t w, int32_t h);
/* Same from already quantised bytes. */
int rt_write_ppm_u8(const char* path, const uint8_t* rgb, int32_t w, int32_t h);

/* Counterpart of plane_mesh_creator.py:23-80 (the y = -1 plane from -5 to +5.1, two triangles per
 * square): writes the same OBJ text for a given SQUARES_PER_UNIT (a power of ten). */
int rt_make_plane_obj(const char* path, int32_t squares_per_unit);

#ifdef __cplusplus
}
#endif
#endif
