// Structures shared by the host code (rt_host.cpp) and the HIP kernels (rt_kernels.hip), and the
// launch wrappers the host calls. Everything here is internal; the public surface is include/rt_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_MAX_AXIS_CELLS 64                       // resolution clamp of Serial/grid.h:100
#define RT_MAX_VOXELS (64 * 64 * 64)                // voxels of the largest grid
// Cells. Every per-voxel table (counts, list starts, occupancy bits) is laid out for the grid with ONE LAYER OF BORDER CELLS
// around it: cell (x + 1) + (n[0] + 2) * ((y + 1) + (n[1] + 2) * (z + 1)) for voxel (x, y, z). Border cells hold no triangle
// and have their occupancy bit SET: a ray that leaves the grid stops at one like at an occupied voxel and is found to be
// done there, so the walk loop carries no per-axis count of the voxels left (rt_kernels.hip, step_through_empty).
#define RT_MAX_CELLS (66 * 66 * 66)
#define RT_SCAN_CHUNK 1280                         // cells per scan block (a multiple of 256)
#define RT_MAX_SCAN_BLOCKS ((RT_MAX_CELLS + RT_SCAN_CHUNK - 1) / RT_SCAN_CHUNK)      // at most 256: k_scan_apply sums them with one block
#define RT_WORK_LIST_COUNTS 32                     // first work[] word of the per-list tile counts (128 bytes after the head)
#define RT_COST_SUM_PARTS 32u                       // partial sums of the cost estimate (power of two)
#define RT_COUNTER_WORDS 216                       // 24 frame counters, 32 work-list words (64 x u32), 32 partial sums of the cost estimate, 8 x 16 for the dequeue heads (a cache line each)
#define RT_COUNTER_FEEDBACK 20                     // frame counter the render kernel adds its streamed runs (low half) and hit trips (high half) to
#ifndef RT_QUEUE_HEADS
#define RT_QUEUE_HEADS 8u                          // dequeue heads of the trace kernel's queue, one per XCD (a power of two)
#endif
#define RT_WORK_HEADS 128                          // first work[] word of the dequeue heads (behind the partial sums), one every
#define RT_WORK_HEAD_STRIDE 32                     // 128 bytes: atomics on one cache line are served one after the other
#define RT_GETREG_XCC_ID ((3 << 11) | 20)          // s_getreg_b32 operand: hardware register 20 (XCC_ID), bits 0..3
#define RT_QUEUE_LISTS 16                          // tile lists the trace kernel serves in order (cost classes)

// Grid description, resident in device memory: k_grid_setup writes the geometry, k_scan_apply writes n_refs.
struct RtGrid {
    float lo[3];
    float hi[3];
    float width[3];
    float inv_width[3];
    int32_t n[3];
    int32_t n_cells;        // cells of the bordered layout: (n[0] + 2) * (n[1] + 2) * (n[2] + 2)
    uint32_t n_refs;        // total (cell, triangle) references = cell_start[n_cells]
    uint32_t overflow;      // set when n_refs exceeds the capacity the scatter was given
    uint32_t pad[2];
};

// Camera basis computed on the host exactly as Serial/raytracer.cpp:130-138 does.
struct RtCamera {
    float pos[3], w[3], u[3], v[3];
    float aspect, focal_distance;
};

struct RtShade {
    float light[3];
    float kd, ks, ka, spec_alpha, light_intensity, shadow_scale, eps;
    // second shading mode (fast_trace of the Parallel tree): per-mesh materials, shadow ray towards the light starting at
    // eps + shadow_bias, mirror reflection down to reflect_depth
    int32_t reflect_depth;
    float shadow_bias;
};

struct RtFrame {
    int32_t width, height;        // full frame
    int32_t part_index, part_count;
    int32_t out_rows;             // rows held by the output buffers
    int32_t walk_full;            // 1 = walk every primary ray to the grid exit
    float early_slack;
    int32_t count_work;           // 1 = the kernel also counts voxel steps and triangle tests
    int32_t shading_mode;         // 0 = fast_trace of the Serial tree, 1 = fast_trace of the Parallel tree
    uint64_t unit_shifts;         // 4 bits per queue list (list 0 lowest): log2 of the work units a tile of the list is cut into
    int32_t adapt_units;          // 1: k_bucket enlarges the units when the frame lists many tiles per wavefront
    int32_t cull_min;             // voxels with at least this many triangles get the per-run box test
    int32_t own_path;             // 1 = trace kernel with the lane = ray path for voxels of a few records (0 for grids known to be
                                  // crowded: the plain kernel needs fewer registers)
    int32_t defer;                // 1 = trace kernel variant that finishes the hit trips of a run together, lane = ray (frames in
                                  // which most trips find a triangle; only with own_path == 0)
};

// Device pointers of one scene.
struct RtSceneDev {
    const float4* v0;             // xyz + mesh index (int bits) in w
    const float4* v1;
    const float4* v2;
    const float* mesh_color;      // 3 floats per mesh
    const float* mesh_mat;        // 8 floats per mesh: kd, ks, spec_alpha, ka, km, reflective (0 / 1), two unused (second shading mode)
    uint32_t n_tris;
    const RtGrid* grid;
    const uint32_t* cell_start;   // n_cells + 1
    const uint32_t* occupancy;    // 1 bit per cell (border cells: set)
    const float4* tri_rec;        // 3 float4 per triangle, triangle order
    const float4* recs;           // cell-major voxel lists: 3 float4 per reference, or (indexed) one uint32 triangle index per reference
    uint32_t indexed;
    const float4* blk_bounds;     // 3 float4 per aligned run of 64 references: {lo.xyz, max |coord|}, {hi.xyz, slab min}, {slab normal, slab max}
};

struct RtOutputs {
    float* rgb_f32;
    uint8_t* rgb_u8;
    int32_t* hit_id;
    uint32_t* unit_cost;          // counting build only: shader cycles of every work unit, in queue order (may be NULL)
    uint32_t* unit_info;          // counting build only: tile | unit shift << 24 | queue list << 28 of every work unit (may be NULL)
    uint32_t* tile_cycles;        // counting build only: shader cycles per tile of the frame, summed over its units (may be NULL)
    unsigned long long* counters; // [0] hits, [1] shadowed, [2..4] primary entered/steps/tests, [5..7] shadow entered/steps/tests
    unsigned long long* stamps;   // when not NULL: [1] start and [2] end of the trace kernel on the 100 MHz real-time clock
};

// Per-frame work list of the render kernels.
struct RtRenderWork {
    uint32_t* work;               // [0] listed tiles, [RT_WORK_HEADS + RT_WORK_HEAD_STRIDE * q] dequeue head q, [RT_WORK_LIST_COUNTS..+15] tiles per queue list, [+16..+31] unit shift per list; zeroed before each frame
    uint32_t* tile_list;          // capacity = tiles of the frame
    uint32_t* cost;               // estimated cost per listed tile
    uint32_t* est_by_tile;        // diagnostics: the four estimate features, each indexed by tile of the frame (may be NULL)
    int32_t n_tiles;              // tiles of the frame
    float est_w[4];               // estimate weights: primary triangles, primary voxels, shadow triangles, shadow voxels
    unsigned long long* cost_sum; // RT_COST_SUM_PARTS partial sums of the tile estimates, zeroed before each frame
    uint32_t* lists;              // RT_QUEUE_LISTS tile lists in queue order, list_cap entries each
    uint32_t list_cap;            // = tiles of the frame
    float thr0, thr1;             // class limits: 1/thr0 and 1/thr1 of a wavefront's even share of the frame's estimate
    int32_t bitmap_words;         // LDS words reserved for the occupancy bitmap (>= ceil(n_cells / 32))
    int32_t n_cu, blocks_per_cu;
    int32_t tx0, ntx, tb0, ntb;   // tile columns x local bands that may see the grid (conservative); the rest stays background
    unsigned long long* stamp_start;   // when not NULL, k_classify writes the real-time clock here (frame start of a frame without a build)
    // Cost estimates kept from frame to frame: the estimate of every tile by tile number (may be NULL) and the total of the
    // listed tiles' estimates. reuse_estimate = 1: this frame has the camera, the light and the triangles of the frame that
    // left them there, so the estimate kernel is left out and the queue lists are made from the kept numbers.
    uint32_t* cost_by_tile;
    unsigned long long* kept_total;
    int32_t reuse_estimate;
};

// Build-time scratch.
struct RtBuildDev {
    float4* v0; float4* v1; float4* v2;
    float2* tri_box;              // 3 float2 per triangle, made at upload: {lo.x, lo.y} {lo.z, hi.x} {hi.y, hi.z}
    float4* tri_rec;              // 3 float4 per triangle: the test record of the triangle (see emit_record), made at upload
    uint32_t n_tris;
    const uint32_t* order;        // when not NULL (meshes above RT_WIDE_MAX_TRIS only): thread t of the count and scatter kernels takes
                                  // triangle order[t], a permutation that puts spatial neighbours side by side (made at upload)
    float cube_root;              // 3.f * powf((float)T, 1.f/3.f), evaluated on the host (libm)
    RtGrid* grid;
    uint32_t* bounds_enc;         // 6 order-preserving encoded floats: lo xyz, hi xyz
    uint32_t* cell_count;         // RT_MAX_CELLS
    uint32_t* cell_start;         // RT_MAX_CELLS + 1
    uint32_t* block_sums;         // RT_MAX_SCAN_BLOCKS
    uint32_t* occupancy;          // RT_MAX_CELLS / 32
    uint32_t* ref_tri;            // ref_cap, unsorted
    uint32_t* ref_cell;           // ref_cap: voxel of each unsorted reference
    uint2* row_minmax;            // per row of 64 references: smallest and largest triangle index after k_row_sort
    float4* recs;                 // 3 * ref_cap
    float4* blk_bounds;           // 3 * (ref_cap / 64 + 2)
    uint32_t ref_cap;
    unsigned long long* counters; // RT_COUNTER_WORDS words the render kernels of the frame count in; the build clears them (may be NULL)
    unsigned long long* feedback; // pinned host memory: the build copies counters[RT_COUNTER_FEEDBACK] of the frame before here (may be NULL)
    unsigned long long* stamp;    // when not NULL, the first kernel of the build writes the 100 MHz real-time clock here (frame start)
    int32_t indexed;              // 1: recs holds one uint32 triangle index per reference instead of the 48-byte record
    int32_t sorted;               // 1: voxel lists in ascending triangle index (scatter, row sort, ranking; needs ref_tri, ref_cell,
                                  // row_minmax); 0: records scattered straight into their voxels, in no particular order
};

// Fixed working memory of rt_render_device_noalloc(): one device array per GPU (file scope in rt_kernels.hip), carved
// into the buffers of a frame at 256-byte boundaries. Limits: references of the grid, 8x8 tiles of the frame.
#define RT_NOALLOC_MAX_REFS 524288u
#define RT_NOALLOC_MAX_TILES 32640u                // 1920 x 1088 pixels
struct RtNoallocLayout {
    size_t grid, bounds_enc, cell_count, cell_start, block_sums, occupancy, recs, blk_bounds, counters, tile_list,
        tile_cost, queue_lists, total;
};
constexpr size_t rt_align256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr RtNoallocLayout rt_noalloc_layout() {
    RtNoallocLayout L{};
    size_t o = 0;
    L.grid = o; o += rt_align256(sizeof(RtGrid));
    L.bounds_enc = o; o += rt_align256(8 * sizeof(uint32_t));
    L.cell_count = o; o += rt_align256((size_t)RT_MAX_CELLS * sizeof(uint32_t));
    L.cell_start = o; o += rt_align256(((size_t)RT_MAX_CELLS + 1) * sizeof(uint32_t));
    L.block_sums = o; o += rt_align256((size_t)RT_MAX_SCAN_BLOCKS * sizeof(uint32_t));
    L.occupancy = o; o += rt_align256(((size_t)RT_MAX_CELLS / 32 + 2) * sizeof(uint32_t));
    L.recs = o; o += rt_align256((size_t)RT_NOALLOC_MAX_REFS * 3 * sizeof(float4));
    L.blk_bounds = o; o += rt_align256(((size_t)RT_NOALLOC_MAX_REFS / 64 + 2) * 3 * sizeof(float4));
    L.counters = o; o += rt_align256(RT_COUNTER_WORDS * sizeof(unsigned long long));
    L.tile_list = o; o += rt_align256((size_t)RT_NOALLOC_MAX_TILES * sizeof(uint32_t));
    L.tile_cost = o; o += rt_align256((size_t)RT_NOALLOC_MAX_TILES * sizeof(uint32_t));
    L.queue_lists = o; o += rt_align256((size_t)RT_NOALLOC_MAX_TILES * RT_QUEUE_LISTS * sizeof(uint32_t));
    L.total = o;
    return L;
}
// Device address of that array on the current device (looked up once per device).
hipError_t rt_noalloc_scratch(void** ptr);

// Launch wrappers (rt_kernels.hip). All asynchronous on `st`; they return the hipGetLastError() of their launches.
hipError_t rt_launch_upload_convert(const float* d_tri9, const uint32_t* d_mesh, RtBuildDev b, hipStream_t st);
// bounds + set-up, count, scan -> grid.n_refs known. `first`: the scene's build state has not been left behind clean by a
// completed build (k_build_init runs)
hipError_t rt_launch_build_phase1(RtBuildDev b, bool first, hipStream_t st);
hipError_t rt_launch_counters_reset(unsigned long long* counters, unsigned long long* feedback, hipStream_t st);
hipError_t rt_launch_run_area(RtBuildDev b, double* out, hipStream_t st);   // adds the summed surface area of the run boxes to *out
hipError_t rt_launch_build_phase2(RtBuildDev b, hipStream_t st);   // scatter, per-cell order, record emit
hipError_t rt_launch_render_prep(RtSceneDev s, RtCamera cam, RtShade sh, RtFrame f, RtRenderWork wk, hipStream_t st);   // classify, estimate, cost classes
hipError_t rt_launch_trace(RtSceneDev s, RtCamera cam, RtShade sh, RtFrame f, RtOutputs o, RtRenderWork wk, hipStream_t st);
size_t rt_trace_lds_bytes(int bitmap_words);
hipError_t rt_launch_extract_cell_tris(const float4* recs, uint32_t* out, uint32_t n_refs, int indexed, hipStream_t st);
hipError_t rt_launch_probe_wait(unsigned long long* stamps, unsigned int ticks, hipStream_t st);   // stamps[0..1]: start and end on the 100 MHz clock
hipError_t rt_read_loop_counts(unsigned long long* out, int n, int reset);   // loop counts of the counting frames since the last reset, current device
hipError_t rt_launch_selftest_div(const double* num, const double* den, uint32_t n, uint32_t* mismatches, hipStream_t st);
hipError_t rt_launch_deinterleave(const void* src, void* dst, int32_t width, int32_t height, int32_t part_count,
                                  int32_t part_rows, int32_t elem_bytes, hipStream_t st);
