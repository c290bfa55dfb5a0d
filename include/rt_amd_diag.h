/* rt_amd_diag.h - diagnostics of the MI355X render path: work counters and scheduling measurements of the
 * counting build of the kernels. None of this has a counterpart in the reference; tools/ and bench.py use it to
 * report the algorithmic work of a frame (voxel steps, triangle tests) and to study the schedule of the trace kernel.
 */
#ifndef RT_AMD_DIAG_H
#define RT_AMD_DIAG_H
#include "rt_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Counters of the last frame rendered while rt_scene_set_counting(s, 1) was in effect (the counting kernel is slower). */
typedef struct rt_frame_counts {
    uint64_t primary_entering, steps_primary, tests_primary;   /* rays that enter the grid, voxels visited, triangle tests */
    uint64_t shadow_entering, steps_shadow, tests_shadow;
    /* scheduling diagnostics of the counting kernel: rounds of the wavefront walk, voxel streams and lanes
     * served over all rounds, shader-clock cycles per traced tile (max, sum), traced tiles, most rounds in one tile */
    uint64_t sched_rounds, sched_streams, sched_lanes /* cycles in the step phase */, tile_cycles_max, tile_cycles_sum, tiles_traced, tile_rounds_max,
        cycles_test /* cycles in the test phase */,
        tests_executed /* triangle tests actually run: runs whose box the ray's line misses are skipped */,
        runs_streamed /* 64-record runs read by wavefronts */,
        run_ray_trips /* (run, ray) iterations of the test loop */,
        cycles_cull /* cycles in the run culling */,
        wave_cycles_max /* busiest wavefront: cycles inside work units */,
        span_cycles /* first wavefront start to last wavefront end */;
} rt_frame_counts;

/* Work accounting: when enabled, frames also count voxel steps and triangle tests per ray class. */
int rt_scene_set_counting(rt_scene* s, int32_t enable);
/* What rt_scene_upload() measured to choose the order of the records inside a voxel: the summed surface area of the
 * bounds of all runs of 64 records with the records as the scatter leaves them (area[0]) and in ascending triangle index
 * (area[1]; both 0 when nothing was measured: small meshes, RT_SORTED_BUILD set), and whether frames sort (1) or not (0). */
int rt_scene_record_order(rt_scene* s, double area[2], int32_t* sorted);
/* The same measurement for the order in which the grid build visits the triangles of a large mesh (more than 65,536): the
 * summed run area of the unsorted build when the triangles are taken along a space-filling curve through their centroids
 * (*area_spatial; 0 when not measured), and whether frames build that way (1) or in the order of the file (0). A file in
 * arbitrary triangle order builds and renders several times slower in its own order; no result depends on the choice.
 * RT_TRI_ORDER=0/1 in the environment of rt_scene_create() pins it. */
int rt_scene_triangle_order(rt_scene* s, double* area_spatial, int32_t* spatial);
/* Counters of the last frame (zeros unless it was a counting frame); synchronises the device. */
int rt_scene_last_counts(rt_scene* s, rt_frame_counts* out);
/* Diagnostics of the last counting frame, as 32-bit words; with t = tiles of the frame (8x8 pixels each):
 * [0, 16t) shader cycles of every work unit in queue order (0 = unused slot), [16t, 20t) the four numbers the cost
 * estimate of every tile is made of, [20t, 21t) shader cycles per tile, [21t, 37t) per work unit: tile | unit shift << 24
 * | queue list << 28, [37t, 53t) per work unit: (run, ray) test iterations | record runs streamed << 18 (both saturate),
 * [53t, 69t) per work unit: cycles / 256 spent stepping | cycles / 256 spent culling runs << 16, [69t, 85t) per work
 * unit: cycles / 256 spent testing | rounds << 24, [85t, 101t) per work unit: its start on the 100 MHz
 * real-time clock (low 20 bits) | its duration on that clock / 64 << 20.
 * 112t words in all. Returns the number of words copied (at most n).
 * No counterpart in the reference. */
int rt_scene_unit_costs(rt_scene* s, uint32_t* out, int64_t n);

/* Loop counts of the counting frames rendered on the current device since the last reset: how often the wavefronts of the
 * trace kernel went through each loop of the walk, 19 numbers for the primary walks, then the same 19 for the shadow walks,
 * in this order: rounds with a ray at an occupied voxel, rounds whose first voxel step ran, iterations of the empty-voxel
 * loop, rounds with lane = ray run culling, its iterations, the (ray, run) box tests in them, voxels culled lane = run,
 * rays broadcast there, item generator calls, its loop iterations, voxels it opened, items tested, items with a live ray,
 * (run, ray) trips, trips with a candidate triangle, trips with a passing triangle, passing triangles applied, walks, trips left
 * after the single-precision pre-filter.
 * Multiplied with the instructions of each loop body (tools/isa_attrib.py) they say where the kernel's instructions go.
 * Synchronises the device. No counterpart in the reference. */
int rt_loop_counts(uint64_t* out, int32_t n, int32_t reset);

/* Frames of the scene so far that ran the tile cost estimate (the others reused an earlier frame's numbers, see
 * rt_scene_set_estimate_reuse in rt_amd.h), or -1 for a NULL scene. No counterpart in the reference. */
int64_t rt_scene_estimates_run(const rt_scene* s);

/* What the render kernel of an earlier frame of the scene reported about its triangle tests: record runs streamed and
 * (run, ray) trips that found a candidate triangle, and whether the next frame would take the kernel variant that finishes
 * a run's hit trips together (it does from 12 such trips per run on; RT_DEFER=0/1 in the environment forces it). The
 * report of a frame reaches the host when the next frame on the same working set starts its grid build; nothing waits
 * for it. No counterpart in the reference (Serial/raytracer.cpp has one code path). */
int rt_scene_test_feedback(rt_scene* s, uint64_t* runs, uint64_t* hit_trips, int* deferred_variant);

#ifdef __cplusplus
}
#endif
#endif
