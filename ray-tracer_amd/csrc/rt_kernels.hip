// HIP kernels for gfx950 (MI355X, 64-lane wavefronts): scene conversion, uniform-grid build and the
// per-pixel render path. Compile with -ffp-contract=off: the reference CPU renderer rounds after every
// multiply and every add (its x86-64 build has no fused multiply-adds), and hit ids are only bit-exact
// when this code does the same.
#include "rt_device.h"

#include <limits.h>
#include <math.h>

#include <mutex>

#define RT_TILE 8                     // a wavefront renders one 8x8 pixel tile
#define RT_BLOCK_THREADS 256

// ------------------------------------------------------------------------------------------------
// helpers: the reference's operation order, restated for the device
// ------------------------------------------------------------------------------------------------

// (int)float as the reference's x86-64 build evaluates it: out-of-range values become INT_MIN
// (v_cvt_i32_f32 would saturate to INT_MAX). NaN and large negative values give the same cell after
// the clamp either way.
__device__ __forceinline__ int f2i_x86(float x) { return (x >= 2147483648.0f) ? INT_MIN : (int)x; }
__device__ __forceinline__ int clampi(int w, int lo, int hi) { return w < lo ? lo : (w > hi ? hi : w); }

// std::max(a,b) = (a<b)?b:a and std::min(a,b) = (b<a)?b:a
__device__ __forceinline__ float max_std(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float min_std(float a, float b) { return (b < a) ? b : a; }

// Vec3::normalize, Serial/geometry.h:23-30: float length2, sqrt and 1/x in double, float scale.
__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
    float n2 = x * x + y * y + z * z;
    if (n2 > 0) {
        float inv = (float)(1.0 / sqrt((double)n2));
        x *= inv; y *= inv; z *= inv;
    }
}

// posToVoxel, Serial/grid.h:59-66
__device__ __forceinline__ int pos_to_voxel(float p, float lo, float inv_w, int n) {
    return clampi(f2i_x86((p - lo) * inv_w), 0, n - 1);
}

// order-preserving float <-> uint map, for atomicMin/atomicMax on floats
__device__ __forceinline__ uint32_t enc_f32(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// ------------------------------------------------------------------------------------------------
// scene upload: host triangle list (9 floats per triangle) -> three float4 streams in HBM
// ------------------------------------------------------------------------------------------------
// Also makes every triangle's test record: {v0.xyz, id}, {v0-v1, lo32(m)}, {v0-v2, hi32(m)} with m the double
// (e1.y*e2.z - e2.y*e1.z), the one 2x2 minor of Triangle::Intersect (Serial/geometry.h:131-177) that does not depend on
// the ray. The grid build copies these records into voxel order.
__global__ void k_upload_convert(const float* __restrict__ tri9, const uint32_t* __restrict__ mesh, float4* v0, float4* v1,
                                 float4* v2, float4* tri_rec, float2* tri_box, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* t = tri9 + 9ull * i;
    v0[i] = make_float4(t[0], t[1], t[2], __uint_as_float(mesh[i]));
    v1[i] = make_float4(t[3], t[4], t[5], 0.f);
    v2[i] = make_float4(t[6], t[7], t[8], 0.f);
    const float e1x = t[0] - t[3], e1y = t[1] - t[4], e1z = t[2] - t[5];
    const float e2x = t[0] - t[6], e2y = t[1] - t[7], e2z = t[2] - t[8];
    const double m = (double)e1y * (double)e2z - (double)e2y * (double)e1z;
    const unsigned long long mb = (unsigned long long)__double_as_longlong(m);
    // the triangle's axis-aligned box (exact: minima and maxima of floats), all the grid build needs of the vertices
    float2* bx = tri_box + 3ull * i;
    bx[0] = make_float2(fminf(t[0], fminf(t[3], t[6])), fminf(t[1], fminf(t[4], t[7])));
    bx[1] = make_float2(fminf(t[2], fminf(t[5], t[8])), fmaxf(t[0], fmaxf(t[3], t[6])));
    bx[2] = make_float2(fmaxf(t[1], fmaxf(t[4], t[7])), fmaxf(t[2], fmaxf(t[5], t[8])));
    float4* o = tri_rec + 3ull * i;
    o[0] = make_float4(t[0], t[1], t[2], __uint_as_float(i));
    o[1] = make_float4(e1x, e1y, e1z, __uint_as_float((uint32_t)mb));
    o[2] = make_float4(e2x, e2y, e2z, __uint_as_float((uint32_t)(mb >> 32)));
}

hipError_t rt_launch_upload_convert(const float* d_tri9, const uint32_t* d_mesh, RtBuildDev b, hipStream_t st) {
    if (b.n_tris == 0) return hipSuccess;
    k_upload_convert<<<(b.n_tris + 255) / 256, 256, 0, st>>>(d_tri9, d_mesh, b.v0, b.v1, b.v2, b.tri_rec, b.tri_box, b.n_tris);
    return hipGetLastError();
}

// Where the record of reference p lives. Copied records: the voxel lists hold the 48-byte test records themselves, a run
// of 64 is one contiguous 3 KB read. Indexed records: the lists hold triangle indices (4 bytes per reference) and the
// records stay in triangle order; the build writes and reads a twelfth of the bytes, the render kernel pays a dependent
// load and a gather per run. The host picks per frame (RtBuildDev::indexed, RtSceneDev::indexed).
__device__ __forceinline__ const float4* rec_ptr(const RtSceneDev& s, uint32_t p) {
    return s.indexed ? s.tri_rec + 3ull * reinterpret_cast<const uint32_t*>(s.recs)[p] : s.recs + 3ull * p;
}
#define RT_REC_PTR(s, p) rec_ptr((s), (p))

// ------------------------------------------------------------------------------------------------
// grid build, Serial/grid.h:79-153, as count / prefix-scan / scatter
// ------------------------------------------------------------------------------------------------

// Frame timings come from the 100 MHz real-time clock, written by the first and the last kernel of a frame themselves:
// an event on the stream costs about 6 us of idle GPU per record, several of them per frame.
__device__ __forceinline__ void stamp_now(unsigned long long* stamp) {
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) *stamp = __builtin_amdgcn_s_memrealtime();
}

__global__ void k_build_init(RtBuildDev b) {
    stamp_now(b.stamp);
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t c = i; c < RT_MAX_CELLS; c += stride) b.cell_count[c] = 0;
    if (i < 3) b.bounds_enc[i] = 0xff800000u;        // running minimum starts at enc(+inf)
    else if (i < 6) b.bounds_enc[i] = 0x007fffffu;   // running maximum starts at enc(-inf)
    if (i == 6) { b.grid->n_refs = 0; b.grid->overflow = 0; b.bounds_enc[6] = 0; }
}

// State the build kernels leave behind for the next build of the scene, so that k_build_init only has to run before the
// first one: k_scatter counts cell_count back down to zero (and clears it when it has to give up), grid_setup resets the
// running bounds, the block counter of k_bounds and the overflow flag, and k_scan_apply always writes n_refs.

__device__ __forceinline__ void grid_setup(const RtBuildDev& b);

// Scene bounds (Serial/grid.h:82-86): min/max over all vertices, taken over the triangles' boxes. Exact and
// order-independent.
// FUSED (small meshes, where every launch counts): the block that finishes last also sets the grid up. For large meshes
// the fences this needs cost more than the launch of k_grid_setup saves.
template <bool FUSED>
__global__ void __launch_bounds__(256) k_bounds(RtBuildDev b) {
    __shared__ float red[4][6];
    stamp_now(b.stamp);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t stride = gridDim.x * blockDim.x;
    // four triangles per trip: twelve loads in flight per lane
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < b.n_tris; i0 += 4 * stride) {
        float2 a[4], c[4], d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + k * stride < b.n_tris ? i0 + k * stride : i0;
            a[k] = b.tri_box[3ull * i]; c[k] = b.tri_box[3ull * i + 1]; d[k] = b.tri_box[3ull * i + 2];   // lo.xy, lo.z hi.x, hi.yz
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[0] = fminf(lo[0], a[k].x); lo[1] = fminf(lo[1], a[k].y); lo[2] = fminf(lo[2], c[k].x);
            hi[0] = fmaxf(hi[0], c[k].y); hi[1] = fmaxf(hi[1], d[k].x); hi[2] = fmaxf(hi[2], d[k].y);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[wv][a] = lo[a]; red[wv][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = red[0][a];
        for (int k = 1; k < 4; ++k) v = (a < 3) ? fminf(v, red[k][a]) : fmaxf(v, red[k][a]);
        if (a < 3) atomicMin(&b.bounds_enc[a], enc_f32(v));
        else atomicMax(&b.bounds_enc[a], enc_f32(v));
    }
    if (FUSED) {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            if (atomicAdd(&b.bounds_enc[6], 1u) == gridDim.x - 1u) {
                __threadfence();
                grid_setup(b);
            }
        }
    }
}

// Resolution and voxel widths, Serial/grid.h:88-107. One thread: a few dozen scalar float operations.
__device__ __forceinline__ void grid_setup(const RtBuildDev& b) {
    RtGrid* g = b.grid;
    float delta[3];
    for (int a = 0; a < 3; ++a) {
        g->lo[a] = dec_f32(__hip_atomic_load(&b.bounds_enc[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        g->hi[a] = dec_f32(__hip_atomic_load(&b.bounds_enc[3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        delta[a] = g->hi[a] - g->lo[a];
    }
    // ready for the next build
    for (int a = 0; a < 3; ++a) { b.bounds_enc[a] = 0xff800000u; b.bounds_enc[3 + a] = 0x007fffffu; }
    b.bounds_enc[6] = 0;
    g->overflow = 0;
    // boundingBox::maxAxis, Serial/geometry.h:276-285: strict '>' so ties go to the later axis
    int ax = (delta[0] > delta[1]) ? 0 : 1;
    if (ax) ax = (delta[1] > delta[2]) ? 1 : 2; else ax = (delta[0] > delta[2]) ? 0 : 2;
    float max_inv = 1.f / delta[ax];
    float vpud = b.cube_root * max_inv;
    int cells = 1;
    for (int a = 0; a < 3; ++a) {
        int nv = clampi(f2i_x86(delta[a] * vpud + 1), 1, RT_MAX_AXIS_CELLS);
        g->n[a] = nv;
        g->width[a] = delta[a] / (float)nv;
        g->inv_width[a] = (g->width[a] == 0.f) ? 0.f : 1.f / g->width[a];
        cells *= nv + 2;                   // one layer of border cells around the grid (rt_device.h)
    }
    g->n_cells = cells;
}

// Scene without triangles: no k_bounds, the set-up alone.
__global__ void k_grid_setup(RtBuildDev b) {
    stamp_now(b.stamp);
    if (threadIdx.x == 0 && blockIdx.x == 0) grid_setup(b);
}

// Voxel range of one triangle's bounding box, Serial/grid.h:121-131, as coordinates in the bordered cell layout (voxel + 1).
__device__ __forceinline__ void tri_voxel_range(const RtBuildDev& b, const RtGrid& g, uint32_t i, int vmin[3], int vmax[3]) {
    const float2 p = b.tri_box[3ull * i], q = b.tri_box[3ull * i + 1], r = b.tri_box[3ull * i + 2];
    float lo[3] = {p.x, p.y, q.x};
    float hi[3] = {q.y, r.x, r.y};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        vmin[a] = pos_to_voxel(lo[a], g.lo[a], g.inv_width[a], g.n[a]) + 1;
        vmax[a] = pos_to_voxel(hi[a], g.lo[a], g.inv_width[a], g.n[a]) + 1;
    }
}

// Groups of lanes that name the same voxel `c`: head = first lane of my group, rank = my place in it, size = its lanes.
// The first RT_GROUP_TRIPS distinct voxels of the wavefront are grouped exactly (one trip each); with coherent meshes
// that covers the wavefront. Lanes left over when every lane names another voxel (large triangles) stay on their own.
#define RT_GROUP_TRIPS 8
// thread-per-triangle build kernels: a triangle whose box covers more voxels than this is handled by the whole wavefront
#define RT_BIG_RANGE 32
__device__ __forceinline__ void lane_groups(bool act, int c, int lane, int& head, uint32_t& rank, uint32_t& size) {
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    head = lane; rank = 0; size = 1;
    unsigned long long rem = __ballot(act);
    const int key = act ? c : -1;              // voxels are numbered from 0: a lane without one matches nobody (one compare per trip)
    for (int trip = 0; trip < RT_GROUP_TRIPS && rem; ++trip) {
        const int l = __ffsll((long long)rem) - 1;
        const int c0 = __builtin_amdgcn_readlane(key, l);
        const bool mine = key == c0;
        const unsigned long long m = __ballot(mine);
        if (mine) { head = l; rank = (uint32_t)__popcll(m & lt_mask); size = (uint32_t)__popcll(m); }
        rem &= ~m;
    }
}

// COUNT: one thread per triangle. Atomics are the expensive part (they are carried out at the memory side), so the lanes
// of a wavefront pool them: in round r every triangle names the r-th voxel of its range, lanes naming the same voxel
// form a group (lane_groups), and one lane adds the group's size. Neighbouring triangle indices tend to share voxels, also the ones
// straddling a voxel border, so a wavefront issues a few atomics per round instead of one per reference.
__global__ void __launch_bounds__(256) k_count(RtBuildDev b) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = t < b.n_tris;
    const uint32_t i = (valid && b.order) ? b.order[t] : t;
    int vmin[3] = {0, 0, 0}, vmax[3] = {0, 0, 0};
    const RtGrid g = *b.grid;                  // same address in every lane; in flight together with the vertex loads
    if (valid) tri_voxel_range(b, g, i, vmin, vmax);
    const int nx = g.n[0] + 2, nxy = nx * (g.n[1] + 2);      // row and layer of the bordered cell layout
    const int ex = vmax[0] - vmin[0] + 1, ey = vmax[1] - vmin[1] + 1;
    const int n_all = valid ? ex * ey * (vmax[2] - vmin[2] + 1) : 0;
    // a triangle with a large voxel range (a ground quad under a detailed mesh: thousands of voxels) would keep its
    // wavefront in the loop below for as many rounds; those are left out here and shared out among the lanes afterwards
    const bool big = n_all > RT_BIG_RANGE;
    const int n = big ? 0 : n_all;
    int x = 0, y = 0, z = 0;
    for (int r = 0;; ++r) {
        const bool act = r < n;
        if (!__ballot(act)) break;
        const int c = (vmin[2] + z) * nxy + (vmin[1] + y) * nx + (vmin[0] + x);
        if (++x == ex) { x = 0; if (++y == ey) { y = 0; ++z; } }
        int head; uint32_t rank, size;
        lane_groups(act, c, lane, head, rank, size);
        if (act && lane == head) atomicAdd(&b.cell_count[c], size);
    }
    for (unsigned long long bm = __ballot(big); bm; bm &= bm - 1) {
        const int l = __ffsll((long long)bm) - 1;
        const int x0 = __builtin_amdgcn_readlane(vmin[0], l), y0 = __builtin_amdgcn_readlane(vmin[1], l), z0 = __builtin_amdgcn_readlane(vmin[2], l);
        const int bex = __builtin_amdgcn_readlane(ex, l), bexy = bex * __builtin_amdgcn_readlane(ey, l), bn = __builtin_amdgcn_readlane(n_all, l);
        for (int r = lane; r < bn; r += 64) {
            const int zz = r / bexy, q = r - zz * bexy, yy = q / bex, xx = q - yy * bex;
            atomicAdd(&b.cell_count[(z0 + zz) * nxy + (y0 + yy) * nx + (x0 + xx)], 1u);
        }
    }
}

// COUNT for small meshes: one WAVEFRONT per triangle, the lanes share out the voxels of its range. With a few thousand
// triangles the thread-per-triangle kernel above is a handful of wavefronts whose run time is the largest triangle's
// voxel count times a round of pooling; here the number of atomics is small anyway and a large triangle costs a few trips.
#define RT_WIDE_MAX_TRIS 65536u
__global__ void __launch_bounds__(256) k_count_wide(RtBuildDev b) {
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= b.n_tris) return;
    const RtGrid g = *b.grid;
    int vmin[3], vmax[3];
    tri_voxel_range(b, g, i, vmin, vmax);
    const int nx = g.n[0] + 2, nxy = nx * (g.n[1] + 2);      // row and layer of the bordered cell layout
    const int ex = vmax[0] - vmin[0] + 1, ey = vmax[1] - vmin[1] + 1, exy = ex * ey;
    const int n = exy * (vmax[2] - vmin[2] + 1);
    for (int r = lane; r < n; r += 64) {
        const int z = r / exy, q = r - z * exy, y = q / ex, x = q - y * ex;
        atomicAdd(&b.cell_count[(vmin[2] + z) * nxy + (vmin[1] + y) * nx + (vmin[0] + x)], 1u);
    }
}

// SCAN part 1: per-chunk totals. A chunk is RT_SCAN_CHUNK consecutive cells, one 256-thread block each.
__global__ void k_scan_sums(RtBuildDev b) {
    __shared__ uint32_t wsum[4];
    // the frame's counters and work-list words start at zero: cleared here, by a kernel the frame launches anyway before
    // anything uses them, instead of by a memset of its own (one launch less in the frame's chain)
    // The word the previous frame's render kernel counted its streamed runs and hit trips in goes to the host first (a
    // location in pinned host memory the host looks at, without waiting, when it picks the next frame's kernel variant).
    if (b.counters && blockIdx.x == 0) {
        if (threadIdx.x == 0 && b.feedback) {
            // written whatever its value: a scene whose frames stop finding triangles leaves the deferred variant again
            __atomic_store_n(b.feedback, b.counters[RT_COUNTER_FEEDBACK], __ATOMIC_RELAXED);
        }
        __syncthreads();
        if (threadIdx.x < RT_COUNTER_WORDS) b.counters[threadIdx.x] = 0ull;
    }
    const int n_cells = b.grid->n_cells;
    uint32_t base = blockIdx.x * RT_SCAN_CHUNK;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < RT_SCAN_CHUNK / 256; ++j) {
        uint32_t c = base + j * 256 + threadIdx.x;
        s += (c < (uint32_t)n_cells) ? b.cell_count[c] : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) b.block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// SCAN part 2: exclusive prefix inside every chunk plus the total of the chunks before it; also the
// occupancy bitmap (1 bit per cell, set for occupied voxels and for border cells) the render kernel walks by.
__global__ void k_scan_apply(RtBuildDev b) {
    __shared__ uint32_t red[4];
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t carry_s;
    const uint32_t n_cells = (uint32_t)b.grid->n_cells;
    const uint32_t base = blockIdx.x * RT_SCAN_CHUNK;
    if (base >= n_cells && !(blockIdx.x == 0)) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // total of all chunks in front of this one (at most 256 values)
    uint32_t pre = (threadIdx.x < blockIdx.x) ? b.block_sums[threadIdx.x] : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off);
    if (lane == 0) red[wv] = pre;
    __syncthreads();
    uint32_t carry = red[0] + red[1] + red[2] + red[3];
    const uint32_t px = (uint32_t)b.grid->n[0] + 2u, py = (uint32_t)b.grid->n[1] + 2u, pz = (uint32_t)b.grid->n[2] + 2u;
    for (int j = 0; j < RT_SCAN_CHUNK / 256; ++j) {
        uint32_t c = base + j * 256 + threadIdx.x;
        uint32_t v = (c < n_cells) ? b.cell_count[c] : 0u;
        // the bit of a border cell is set (nothing is ever counted into one)
        const uint32_t cz = c / (px * py), cr = c - cz * (px * py), cy = cr / px, cx = cr - cy * px;
        const bool border = c < n_cells && (cx == 0u || cx == px - 1u || cy == 0u || cy == py - 1u || cz == 0u || cz == pz - 1u);
        unsigned long long occ = __ballot(v != 0 || border);
        if (lane == 0 && (c < n_cells)) {
            b.occupancy[c >> 5] = (uint32_t)occ;
            b.occupancy[(c >> 5) + 1] = (uint32_t)(occ >> 32);
        }
        // inclusive scan inside the wavefront
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wtot[wv] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < wv; ++k) woff += wtot[k];
        uint32_t excl = carry + woff + inc - v;
        if (c < n_cells) b.cell_start[c] = excl;
        if (c == n_cells - 1) {
            uint32_t total = excl + v;
            b.cell_start[n_cells] = total;
            b.grid->n_refs = total;
        }
        if (threadIdx.x == 255) carry_s = carry + woff + inc;
        __syncthreads();
        carry = carry_s;
    }
}

// SCATTER: claims slots in the voxel's segment by counting cell_count back down to zero, with the same per-wavefront
// pooling as COUNT: per round the groups are found first (no memory traffic), then every group's first lane claims the
// group's slots with ONE atomic, all groups at once, so the wavefront waits for a single round trip per round. The
// order inside a voxel is whatever the atomics produce; k_row_sort and k_order_emit restore ascending triangle index.
__global__ void __launch_bounds__(256) k_scatter(RtBuildDev b) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const RtGrid g = *b.grid;
    if (g.n_refs > b.ref_cap) {              // cannot scatter: flag it and leave cell_count clear for the next build
        if (t == 0) b.grid->overflow = 1;
        for (uint32_t c = t; c < RT_MAX_CELLS; c += gridDim.x * blockDim.x) b.cell_count[c] = 0;
        return;
    }
    const bool valid = t < b.n_tris;
    const uint32_t i = (valid && b.order) ? b.order[t] : t;
    int vmin[3] = {0, 0, 0}, vmax[3] = {0, 0, 0};
    if (valid) tri_voxel_range(b, g, i, vmin, vmax);
    const int nx = g.n[0] + 2, nxy = nx * (g.n[1] + 2);      // row and layer of the bordered cell layout
    const int ex = vmax[0] - vmin[0] + 1, ey = vmax[1] - vmin[1] + 1;
    const int n_all = valid ? ex * ey * (vmax[2] - vmin[2] + 1) : 0;
    const bool big = n_all > RT_BIG_RANGE;              // as in k_count
    const int n = big ? 0 : n_all;
    int x = 0, y = 0, z = 0;
    for (int r = 0;; ++r) {
        const bool act = r < n;
        if (!__ballot(act)) break;
        const int c = (vmin[2] + z) * nxy + (vmin[1] + y) * nx + (vmin[0] + x);
        if (++x == ex) { x = 0; if (++y == ey) { y = 0; ++z; } }
        uint32_t start = 0;
        if (act) start = b.cell_start[c];
        int leader; uint32_t rank, group;
        lane_groups(act, c, lane, leader, rank, group);
        uint32_t old = 0;
        if (act && lane == leader) old = atomicSub(&b.cell_count[c], group);
        old = (uint32_t)__shfl((int)old, leader);
        if (act) {
            const uint32_t pos = start + old - 1u - rank;
            b.ref_tri[pos] = i;
            b.ref_cell[pos] = (uint32_t)c;
        }
    }
    for (unsigned long long bm = __ballot(big); bm; bm &= bm - 1) {
        const int l = __ffsll((long long)bm) - 1;
        const uint32_t bi = (uint32_t)__builtin_amdgcn_readlane((int)i, l);
        const int x0 = __builtin_amdgcn_readlane(vmin[0], l), y0 = __builtin_amdgcn_readlane(vmin[1], l), z0 = __builtin_amdgcn_readlane(vmin[2], l);
        const int bex = __builtin_amdgcn_readlane(ex, l), bexy = bex * __builtin_amdgcn_readlane(ey, l), bn = __builtin_amdgcn_readlane(n_all, l);
        for (int r = lane; r < bn; r += 64) {
            const int zz = r / bexy, q = r - zz * bexy, yy = q / bex, xx = q - yy * bex;
            const int c = (z0 + zz) * nxy + (y0 + yy) * nx + (x0 + xx);
            const uint32_t pos = b.cell_start[c] + atomicSub(&b.cell_count[c], 1u) - 1u;
            b.ref_tri[pos] = bi;
            b.ref_cell[pos] = (uint32_t)c;
        }
    }
}

// SCATTER + EMIT: the default second half of the build. As k_scatter, but what goes into the claimed slot is the
// triangle's 48-byte test record itself, so the voxel lists are final after this kernel: no reference arrays, no sort, no
// ranking. The records of a voxel end up in the order the atomics hand the slots out (consecutive triangles of a
// wavefront stay together, in descending index); the render kernel's nearest-hit update does not depend on it.
__global__ void __launch_bounds__(256) k_scatter_emit(RtBuildDev b) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const RtGrid g = *b.grid;
    if (g.n_refs > b.ref_cap) {              // cannot scatter: flag it and leave cell_count clear for the next build
        if (t == 0) b.grid->overflow = 1;
        for (uint32_t c = t; c < RT_MAX_CELLS; c += gridDim.x * blockDim.x) b.cell_count[c] = 0;
        return;
    }
    const bool valid = t < b.n_tris;
    const uint32_t i = (valid && b.order) ? b.order[t] : t;
    int vmin[3] = {0, 0, 0}, vmax[3] = {0, 0, 0};
    float4 rec0 = make_float4(0, 0, 0, 0), rec1 = rec0, rec2 = rec0;
    if (valid) {
        tri_voxel_range(b, g, i, vmin, vmax);
        if (!b.indexed) {
            const float4* r = b.tri_rec + 3ull * i;
            rec0 = r[0]; rec1 = r[1]; rec2 = r[2];
        }
    }
    const int nx = g.n[0] + 2, nxy = nx * (g.n[1] + 2);      // row and layer of the bordered cell layout
    const int ex = vmax[0] - vmin[0] + 1, ey = vmax[1] - vmin[1] + 1;
    const int n_all = valid ? ex * ey * (vmax[2] - vmin[2] + 1) : 0;
    const bool big = n_all > RT_BIG_RANGE;              // as in k_count
    const int n = big ? 0 : n_all;
    int x = 0, y = 0, z = 0;
    for (int r = 0;; ++r) {
        const bool act = r < n;
        if (!__ballot(act)) break;
        const int c = (vmin[2] + z) * nxy + (vmin[1] + y) * nx + (vmin[0] + x);
        if (++x == ex) { x = 0; if (++y == ey) { y = 0; ++z; } }
        uint32_t start = 0;
        if (act) start = b.cell_start[c];
        int leader; uint32_t rank, group;
        lane_groups(act, c, lane, leader, rank, group);
        uint32_t old = 0;
        if (act && lane == leader) old = atomicSub(&b.cell_count[c], group);
        old = (uint32_t)__shfl((int)old, leader);
        if (act) {
            if (b.indexed) {
                reinterpret_cast<uint32_t*>(b.recs)[start + old - 1u - rank] = i;
            } else {
                float4* o = b.recs + 3ull * (start + old - 1u - rank);
                o[0] = rec0; o[1] = rec1; o[2] = rec2;
            }
        }
    }
    for (unsigned long long bm = __ballot(big); bm; bm &= bm - 1) {
        const int l = __ffsll((long long)bm) - 1;
        const uint32_t bi = (uint32_t)__builtin_amdgcn_readlane((int)i, l);
        const int x0 = __builtin_amdgcn_readlane(vmin[0], l), y0 = __builtin_amdgcn_readlane(vmin[1], l), z0 = __builtin_amdgcn_readlane(vmin[2], l);
        const int bex = __builtin_amdgcn_readlane(ex, l), bexy = bex * __builtin_amdgcn_readlane(ey, l), bn = __builtin_amdgcn_readlane(n_all, l);
        float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0;
        if (!b.indexed) { const float4* r = b.tri_rec + 3ull * bi; q0 = r[0]; q1 = r[1]; q2 = r[2]; }
        for (int r = lane; r < bn; r += 64) {
            const int zz = r / bexy, q = r - zz * bexy, yy = q / bex, xx = q - yy * bex;
            const int c = (z0 + zz) * nxy + (y0 + yy) * nx + (x0 + xx);
            const uint32_t pos = b.cell_start[c] + atomicSub(&b.cell_count[c], 1u) - 1u;
            if (b.indexed) {
                reinterpret_cast<uint32_t*>(b.recs)[pos] = bi;
            } else {
                float4* o = b.recs + 3ull * pos;
                o[0] = q0; o[1] = q1; o[2] = q2;
            }
        }
    }
}

// The same for small meshes, one wavefront per triangle like k_count_wide.
__global__ void __launch_bounds__(256) k_scatter_emit_wide(RtBuildDev b) {
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const RtGrid g = *b.grid;
    if (g.n_refs > b.ref_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) b.grid->overflow = 1;
        for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < RT_MAX_CELLS; c += gridDim.x * blockDim.x) b.cell_count[c] = 0;
        return;
    }
    if (i >= b.n_tris) return;
    int vmin[3], vmax[3];
    tri_voxel_range(b, g, i, vmin, vmax);
    const float4* rr = b.tri_rec + 3ull * i;
    float4 rec0 = make_float4(0, 0, 0, 0), rec1 = rec0, rec2 = rec0;
    if (!b.indexed) { rec0 = rr[0]; rec1 = rr[1]; rec2 = rr[2]; }
    const int nx = g.n[0] + 2, nxy = nx * (g.n[1] + 2);      // row and layer of the bordered cell layout
    const int ex = vmax[0] - vmin[0] + 1, ey = vmax[1] - vmin[1] + 1, exy = ex * ey;
    const int n = exy * (vmax[2] - vmin[2] + 1);
    for (int r = lane; r < n; r += 64) {
        const int z = r / exy, q = r - z * exy, y = q / ex, x = q - y * ex;
        const int c = (vmin[2] + z) * nxy + (vmin[1] + y) * nx + (vmin[0] + x);
        const uint32_t pos = b.cell_start[c] + atomicSub(&b.cell_count[c], 1u) - 1u;
        if (b.indexed) {
            reinterpret_cast<uint32_t*>(b.recs)[pos] = i;
        } else {
            float4* o = b.recs + 3ull * pos;
            o[0] = rec0; o[1] = rec1; o[2] = rec2;
        }
    }
}

// SCATTER for small meshes, one wavefront per triangle like k_count_wide.
__global__ void __launch_bounds__(256) k_scatter_wide(RtBuildDev b) {
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const RtGrid g = *b.grid;
    if (g.n_refs > b.ref_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) b.grid->overflow = 1;
        for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < RT_MAX_CELLS; c += gridDim.x * blockDim.x) b.cell_count[c] = 0;
        return;
    }
    if (i >= b.n_tris) return;
    int vmin[3], vmax[3];
    tri_voxel_range(b, g, i, vmin, vmax);
    const int nx = g.n[0] + 2, nxy = nx * (g.n[1] + 2);      // row and layer of the bordered cell layout
    const int ex = vmax[0] - vmin[0] + 1, ey = vmax[1] - vmin[1] + 1, exy = ex * ey;
    const int n = exy * (vmax[2] - vmin[2] + 1);
    for (int r = lane; r < n; r += 64) {
        const int z = r / exy, q = r - z * exy, y = q / ex, x = q - y * ex;
        const int c = (vmin[2] + z) * nxy + (vmin[1] + y) * nx + (vmin[0] + x);
        const uint32_t pos = b.cell_start[c] + atomicSub(&b.cell_count[c], 1u) - 1u;
        b.ref_tri[pos] = i;
        b.ref_cell[pos] = (uint32_t)c;
    }
}

// ROW SORT: every aligned row of 64 references is sorted by (voxel, triangle index) inside its wavefront (bitonic
// network over the lanes). References are already grouped by voxel, so a voxel's references keep their places in the
// row and only their order inside the voxel's part of the row changes.
__global__ void __launch_bounds__(256) k_row_sort(RtBuildDev b) {
    const uint32_t n_refs = b.grid->n_refs;
    if (n_refs > b.ref_cap) return;
    const int lane = threadIdx.x & 63;
    const uint32_t n_waves = gridDim.x * 4, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t n_rows = (n_refs + 63u) / 64u;
    for (uint32_t row = wave; row < n_rows; row += n_waves) {
        const uint32_t p = row * 64u + lane;
        uint32_t cell = 0xffffffffu, tri = 0xffffffffu;
        if (p < n_refs) { cell = b.ref_cell[p]; tri = b.ref_tri[p]; }
        unsigned long long key;
        if (__builtin_amdgcn_readlane((int)cell, 0) == __builtin_amdgcn_readlane((int)cell, 63)) {
            // the whole row lies in one voxel (the usual case): the triangle index alone is the key
            uint32_t k32 = tri;
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const uint32_t other = __shfl_xor(k32, j);
                    const bool up = (lane & k) == 0, lower = (lane & j) == 0;
                    const uint32_t mn = k32 < other ? k32 : other, mx = k32 < other ? other : k32;
                    k32 = (lower == up) ? mn : mx;
                }
            }
            key = k32;
        } else {
            key = ((unsigned long long)cell << 32) | tri;
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const unsigned long long other = __shfl_xor(key, j);
                    const bool up = (lane & k) == 0, lower = (lane & j) == 0;
                    const unsigned long long mn = key < other ? key : other, mx = key < other ? other : key;
                    key = (lower == up) ? mn : mx;
                }
            }
        }
        if (p < n_refs) b.ref_tri[p] = (uint32_t)key;
        // smallest and largest index of the row: what k_order_emit looks at first for rows that lie inside one voxel
        const uint32_t first_id = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, 0);
        const uint32_t last_id = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, 63);
        if (lane == 0) b.row_minmax[row] = make_uint2(first_id, last_id);
    }
}

// ORDER + EMIT: the reference appends triangles to a voxel in ascending index (Serial/grid.h:118-150) and the
// first-tested triangle wins ties, so every voxel's list is put back into ascending order: each reference finds its
// rank = the number of smaller indices in its voxel (indices are distinct) and its triangle's 48-byte record is copied
// there. A wavefront takes one row of 64 references. The rows are sorted (k_row_sort), so for a row that lies inside
// the voxel its smallest and largest index (row_minmax) usually settle the count: a row whose largest index is below
// mine counts whole, one whose smallest index is not below mine counts nothing. The atomics of k_scatter hand out
// places almost in index order, so only a few rows straddle an index; those are read (64 indices, one coalesced load),
// parked in LDS and searched with 7 probes. The up to two rows a voxel shares with its neighbours are searched the same
// way, with the neighbours' places masked out. No wavefront waits for another, and a crowded voxel is shared out over
// as many wavefronts as it has rows.
#define RT_ORDER_BATCH 4               // straddling rows read together

// number of entries below `key` in 64 ascending entries
__device__ __forceinline__ uint32_t probe64(const uint32_t* s0, uint32_t key) {
    uint32_t n0 = 0;
#pragma unroll
    for (uint32_t st = 32; st > 0; st >>= 1) n0 += (s0[n0 + st - 1u] < key) ? st : 0u;
    n0 += (s0[n0] < key) ? 1u : 0u;
    return n0;
}

__global__ void __launch_bounds__(256) k_order_emit(RtBuildDev b) {
    __shared__ __attribute__((aligned(16))) uint32_t park_all[4][RT_ORDER_BATCH * 64];
    const uint32_t n_refs = b.grid->n_refs;
    if (n_refs > b.ref_cap) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t* park = park_all[wv];
    const uint32_t n_waves = gridDim.x * 4, wave = blockIdx.x * 4 + wv;
    const uint32_t n_rows = (n_refs + 63u) / 64u;
    // a row's inputs are fetched one row ahead: two round trips off the dependent chain
    struct RowIn { uint32_t c, me, beg, end; };
    auto fetch = [&](uint32_t row, RowIn& q) {
        q.c = 0xffffffffu; q.me = 0; q.beg = 0; q.end = 0;
        const uint32_t p = row * 64u + lane;
        if (row < n_rows && p < n_refs) { q.c = b.ref_cell[p]; q.me = b.ref_tri[p]; q.beg = b.cell_start[q.c]; q.end = b.cell_start[q.c + 1]; }
    };
    RowIn nxt;
    fetch(wave, nxt);
    for (uint32_t row = wave; row < n_rows; row += n_waves) {
        const RowIn cur = nxt;
        fetch(row + n_waves, nxt);
        const uint32_t p = row * 64u + lane;
        const bool valid = p < n_refs;
        const uint32_t c = cur.c, me = cur.me, beg = cur.beg, end = cur.end;
        // the triangle's record is requested now and written once the rank is known
        float4 rec0 = make_float4(0, 0, 0, 0), rec1 = rec0, rec2 = rec0;
        if (valid && !b.indexed) { const float4* r = b.tri_rec + 3ull * me; rec0 = r[0]; rec1 = r[1]; rec2 = r[2]; }
        uint32_t rank = 0;
        unsigned long long rem = __ballot(valid);
        while (rem) {
            const int l = __ffsll((long long)rem) - 1;
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c, l);
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)beg, l);
            const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)end, l);
            const bool mem = valid && c == c0;
            rem &= ~__ballot(mem);
            // rows of the voxel: [in_lo, in_hi) lie inside it; the row before in_lo and the row in_hi may be shared
            const uint32_t in_lo = (b0 + 63u) >> 6, in_hi = e0 >> 6;
            // ---- shared rows: the voxel's part of the row, neighbours masked (0 below, 0xffffffff above; indices + 1)
            uint32_t edge[2]; int n_edge = 0;
            if ((b0 & 63u) != 0u) edge[n_edge++] = b0 >> 6;
            if ((e0 & 63u) != 0u && (n_edge == 0 || (e0 >> 6) != edge[0])) edge[n_edge++] = e0 >> 6;
            for (int k = 0; k < n_edge; ++k) {
                const uint32_t q = edge[k] * 64u + lane;
                park[k * 64 + lane] = (q >= b0 && q < e0) ? b.ref_tri[q] + 1u : ((q < b0) ? 0u : 0xffffffffu);
            }
            __builtin_amdgcn_wave_barrier();
            for (int k = 0; k < n_edge; ++k) {
                if (mem) {
                    uint32_t n0 = probe64(park + k * 64, me + 1u);
                    const uint32_t row0 = edge[k] * 64u;
                    if (row0 < b0) n0 -= b0 - row0;              // the masked places below the voxel were counted
                    rank += n0;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- rows inside the voxel, 64 at a time: lane j looks after row base + j
            for (uint32_t base = in_lo; base < in_hi; base += 64u) {
                const uint32_t nb = (in_hi - base) < 64u ? (in_hi - base) : 64u;
                uint2 mm = make_uint2(0xffffffffu, 0xffffffffu);
                if ((uint32_t)lane < nb) mm = b.row_minmax[base + lane];
                unsigned long long straddle = 0;
                for (uint32_t j = 0; j < nb; ++j) {
                    const uint32_t lo_id = (uint32_t)__builtin_amdgcn_readlane((int)mm.x, (int)j);
                    const uint32_t hi_id = (uint32_t)__builtin_amdgcn_readlane((int)mm.y, (int)j);
                    const bool whole = hi_id < me, none = !(lo_id < me);
                    if (mem && whole) rank += 64u;
                    if (__ballot(mem && !whole && !none)) straddle |= 1ull << j;
                }
                while (straddle) {
                    uint32_t rows[RT_ORDER_BATCH]; int nr = 0;
                    while (straddle && nr < RT_ORDER_BATCH) { rows[nr++] = base + (uint32_t)(__ffsll((long long)straddle) - 1); straddle &= straddle - 1; }
                    uint32_t v[RT_ORDER_BATCH];
#pragma unroll
                    for (int k = 0; k < RT_ORDER_BATCH; ++k) v[k] = b.ref_tri[(size_t)rows[k < nr ? k : 0] * 64u + lane];
#pragma unroll
                    for (int k = 0; k < RT_ORDER_BATCH; ++k) park[k * 64 + lane] = v[k];
                    __builtin_amdgcn_wave_barrier();
                    for (int k = 0; k < nr; ++k) {
                        const uint32_t* s0 = park + k * 64;
                        if (mem && s0[0] < me && !(s0[63] < me)) rank += probe64(s0, me);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (valid) {
            if (b.indexed) reinterpret_cast<uint32_t*>(b.recs)[beg + rank] = me;
            else { float4* o = b.recs + 3ull * (beg + rank); o[0] = rec0; o[1] = rec1; o[2] = rec2; }
        }
    }
}

__device__ __forceinline__ float bcast_f32(float v, int src_lane);

// Wavefront reductions without LDS traffic, all on the data-parallel-primitive path of the vector ALU: four steps inside
// every row of 16 lanes (lane pairs 1 apart, 2 apart, mirrored in 8, mirrored in 16), then the last lane of row 0 goes to
// row 1 and that of row 2 to row 3, then the last lane of row 1 to rows 2 and 3. The result, ((r0 . r1) . (r2 . r3)) of the
// four row results, stands in lanes 48..63 only.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_or_zero(float v) {          // rows a step does not write read zero
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_or_zero<0xB1, 0xf>(v);         // quad_perm [1,0,3,2]
    v += dpp_or_zero<0x4E, 0xf>(v);         // quad_perm [2,3,0,1]
    v += dpp_or_zero<0x141, 0xf>(v);        // row_half_mirror
    v += dpp_or_zero<0x140, 0xf>(v);        // row_mirror
    v += dpp_or_zero<0x142, 0xa>(v);        // row_bcast:15 into rows 1 and 3
    return v + dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
}
// The same six steps for minima and maxima, written out: fminf() on a value the compiler cannot see through (a lane
// permutation) costs a canonicalising instruction per operand and the permutation is not folded into it, four
// instructions a step instead of one. The operands here are never NaN (they come out of fminf/fmaxf against an infinity).
// A permuting instruction must not read a register the two instructions before it wrote; the steps of several
// independent reductions are interleaved for that, with a one-cycle filler where there are only two.
#define RT_DPP_STEP3(op, a, b, c, ctrl) \
    "v_" op "_f32_dpp " a ", " a ", " a " " ctrl "\n v_" op "_f32_dpp " b ", " b ", " b " " ctrl "\n v_" op "_f32_dpp " c ", " c ", " c " " ctrl "\n"
#define RT_DPP_ALL_STEPS3(op, a, b, c) \
    "s_nop 1\n" \
    RT_DPP_STEP3(op, a, b, c, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    RT_DPP_STEP3(op, a, b, c, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") \
    RT_DPP_STEP3(op, a, b, c, "row_half_mirror row_mask:0xf bank_mask:0xf") \
    RT_DPP_STEP3(op, a, b, c, "row_mirror row_mask:0xf bank_mask:0xf") \
    RT_DPP_STEP3(op, a, b, c, "row_bcast:15 row_mask:0xa bank_mask:0xf") \
    RT_DPP_STEP3(op, a, b, c, "row_bcast:31 row_mask:0xc bank_mask:0xf")
__device__ __forceinline__ void wave_min3_f32(float& a, float& b, float& c) { asm(RT_DPP_ALL_STEPS3("min", "%0", "%1", "%2") : "+v"(a), "+v"(b), "+v"(c)); }
__device__ __forceinline__ void wave_max3_f32(float& a, float& b, float& c) { asm(RT_DPP_ALL_STEPS3("max", "%0", "%1", "%2") : "+v"(a), "+v"(b), "+v"(c)); }
#define RT_DPP_STEP_MINMAX(lo, hi, ctrl) "v_min_f32_dpp " lo ", " lo ", " lo " " ctrl "\n v_max_f32_dpp " hi ", " hi ", " hi " " ctrl "\n s_nop 0\n"
__device__ __forceinline__ void wave_min_max_f32(float& lo, float& hi) {
    asm("s_nop 1\n"
        RT_DPP_STEP_MINMAX("%0", "%1", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        RT_DPP_STEP_MINMAX("%0", "%1", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        RT_DPP_STEP_MINMAX("%0", "%1", "row_half_mirror row_mask:0xf bank_mask:0xf")
        RT_DPP_STEP_MINMAX("%0", "%1", "row_mirror row_mask:0xf bank_mask:0xf")
        RT_DPP_STEP_MINMAX("%0", "%1", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        RT_DPP_STEP_MINMAX("%0", "%1", "row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(lo), "+v"(hi));
}

// BLOCK BOUNDS of every aligned run of 64 records: the axis-aligned box of its triangles (lo.xyz + largest absolute
// coordinate, hi.xyz) and a slab: a unit vector n (the run's summed triangle normals; consecutive triangles of a mesh
// form a nearly flat patch) with the smallest and largest n.v over the run's vertices. All triangles of the run lie
// inside the box AND between the slab's two planes, so a ray can only hit one of them where its line is inside both
// at once. The render kernel skips a run when that clearly never happens; the reference would test those triangles
// and find that none passes. Any n gives a valid slab (a folded patch just gives a thick one).
__global__ void __launch_bounds__(256) k_block_bounds(RtBuildDev b) {
    const uint32_t n_refs = b.grid->n_refs;
    if (n_refs > b.ref_cap) return;
    const int lane = threadIdx.x & 63;
    const uint32_t n_waves = gridDim.x * 4, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t n_blk = (n_refs + 63u) / 64u;
    for (uint32_t blk = wave; blk < n_blk; blk += n_waves) {
        const uint32_t p = blk * 64u + lane;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float vx[3] = {0, 0, 0}, vy[3] = {0, 0, 0}, vz[3] = {0, 0, 0};      // the triangle's vertices
        float nx = 0.f, ny = 0.f, nz = 0.f;
        const bool have = p < n_refs;
        if (have) {
            // the vertices as the record gives them: v0, v0 - e1, v0 - e2. Rebuilding v1 and v2 from the rounded edge
            // vectors is off by at most one unit in the last place of the larger operand, two orders of magnitude below
            // the margin the box and slab tests add.
            const float4* rp = b.indexed ? b.tri_rec + 3ull * reinterpret_cast<const uint32_t*>(b.recs)[p] : b.recs + 3ull * p;
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
            vx[0] = r0.x; vy[0] = r0.y; vz[0] = r0.z;
            vx[1] = r0.x - r1.x; vy[1] = r0.y - r1.y; vz[1] = r0.z - r1.z;
            vx[2] = r0.x - r2.x; vy[2] = r0.y - r2.y; vz[2] = r0.z - r2.z;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[0] = fminf(lo[0], vx[k]); hi[0] = fmaxf(hi[0], vx[k]);
                lo[1] = fminf(lo[1], vy[k]); hi[1] = fmaxf(hi[1], vy[k]);
                lo[2] = fminf(lo[2], vz[k]); hi[2] = fmaxf(hi[2], vz[k]);
            }
            nx = r1.y * r2.z - r1.z * r2.y; ny = r1.z * r2.x - r1.x * r2.z; nz = r1.x * r2.y - r1.y * r2.x;    // e1 x e2
        }
        // normals that point against the first triangle's are turned round before they are summed
        {
            const float fx = bcast_f32(nx, 0), fy = bcast_f32(ny, 0), fz = bcast_f32(nz, 0);
            if (nx * fx + ny * fy + nz * fz < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
        }
        wave_min3_f32(lo[0], lo[1], lo[2]);
        wave_max3_f32(hi[0], hi[1], hi[2]);
        nx = bcast_f32(wave_sum_f32(nx), 63); ny = bcast_f32(wave_sum_f32(ny), 63); nz = bcast_f32(wave_sum_f32(nz), 63);
        {
            const float len = sqrtf(nx * nx + ny * ny + nz * nz);
            // one division: any vector near unit length will do, the slab's two distances below are measured along it
            if (len > 1e-30f && len < 1e30f) { const float inv = 1.f / len; nx *= inv; ny *= inv; nz *= inv; } else { nx = 0.f; ny = 0.f; nz = 0.f; }
        }
        float dmin = INFINITY, dmax = -INFINITY;
        if (have) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d = nx * vx[k] + ny * vy[k] + nz * vz[k];
                dmin = fminf(dmin, d); dmax = fmaxf(dmax, d);
            }
        }
        wave_min_max_f32(dmin, dmax);
        if (lane == 63) {          // the reductions end in the last row
            float m = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) m = fmaxf(m, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
            b.blk_bounds[3ull * blk] = make_float4(lo[0], lo[1], lo[2], m);
            b.blk_bounds[3ull * blk + 1] = make_float4(hi[0], hi[1], hi[2], dmin);
            b.blk_bounds[3ull * blk + 2] = make_float4(nx, ny, nz, dmax);
        }
    }
}

// Sum of the surface areas of the run boxes: the chance that a line crosses a convex body is proportional to its
// surface area, so this sum says how many runs a ray has to test on average. The host compares it between the two orders
// of the records inside a voxel, once per triangle list.
__global__ void __launch_bounds__(256) k_run_area(RtBuildDev b, double* out) {
    __shared__ double part[4];
    const uint32_t n_refs = b.grid->n_refs;
    const uint32_t n_blk = (n_refs > b.ref_cap) ? 0u : (n_refs + 63u) / 64u;
    double a = 0.0;
    for (uint32_t blk = blockIdx.x * blockDim.x + threadIdx.x; blk < n_blk; blk += gridDim.x * blockDim.x) {
        const float4 lo = b.blk_bounds[3ull * blk], hi = b.blk_bounds[3ull * blk + 1];
        const double dx = (double)hi.x - lo.x, dy = (double)hi.y - lo.y, dz = (double)hi.z - lo.z;
        const double v = dx * dy + dy * dz + dz * dx;
        if (v == v && v < 1e300) a += v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// Frames without a grid build: the frame counters start at zero, after the word the previous frame's render kernel reported
// in has gone to the host (frames with a build do both in k_scan_sums).
__global__ void k_counters_reset(unsigned long long* counters, unsigned long long* feedback) {
    if (threadIdx.x == 0 && feedback) {
        __atomic_store_n(feedback, counters[RT_COUNTER_FEEDBACK], __ATOMIC_RELAXED);
    }
    __syncthreads();
    if (threadIdx.x < RT_COUNTER_WORDS) counters[threadIdx.x] = 0ull;
}

hipError_t rt_launch_counters_reset(unsigned long long* counters, unsigned long long* feedback, hipStream_t st) {
    k_counters_reset<<<1, 256, 0, st>>>(counters, feedback);
    return hipGetLastError();
}

hipError_t rt_launch_run_area(RtBuildDev b, double* out, hipStream_t st) {
    k_run_area<<<256, 256, 0, st>>>(b, out);
    return hipGetLastError();
}

hipError_t rt_launch_build_phase1(RtBuildDev b, bool first, hipStream_t st) {
    // only the first kernel launched stamps the start of the build
    RtBuildDev b0 = b;
    b.stamp = nullptr;
    if (first) { k_build_init<<<256, 256, 0, st>>>(b0); b0.stamp = nullptr; }
    if (b.n_tris && b.n_tris <= RT_WIDE_MAX_TRIS) {
        const uint32_t nb = (b.n_tris + 255) / 256;
        k_bounds<true><<<nb < 256 ? nb : 256, 256, 0, st>>>(b0);    // its last block sets the grid up
    } else {
        if (b.n_tris) { k_bounds<false><<<256, 256, 0, st>>>(b0); b0.stamp = nullptr; }
        k_grid_setup<<<1, 64, 0, st>>>(b0);
    }
    if (b.n_tris && b.n_tris <= RT_WIDE_MAX_TRIS) k_count_wide<<<(b.n_tris + 3) / 4, 256, 0, st>>>(b);
    else if (b.n_tris) k_count<<<(b.n_tris + 255) / 256, 256, 0, st>>>(b);
    k_scan_sums<<<RT_MAX_SCAN_BLOCKS, 256, 0, st>>>(b);
    k_scan_apply<<<RT_MAX_SCAN_BLOCKS, 256, 0, st>>>(b);
    return hipGetLastError();
}

hipError_t rt_launch_build_phase2(RtBuildDev b, hipStream_t st) {
    if (b.n_tris == 0) return hipSuccess;
    if (!b.sorted) {
        if (b.n_tris <= RT_WIDE_MAX_TRIS) k_scatter_emit_wide<<<(b.n_tris + 3) / 4, 256, 0, st>>>(b);
        else k_scatter_emit<<<(b.n_tris + 255) / 256, 256, 0, st>>>(b);
    } else {
        // voxel lists in ascending triangle index, the reference's order: for the read-back of the lists and for the
        // counting build of the render kernel, whose shadow-ray test counts are those of that order
        if (b.n_tris <= RT_WIDE_MAX_TRIS) k_scatter_wide<<<(b.n_tris + 3) / 4, 256, 0, st>>>(b);
        else k_scatter<<<(b.n_tris + 255) / 256, 256, 0, st>>>(b);
        k_row_sort<<<1024, 256, 0, st>>>(b);
        k_order_emit<<<2048, 256, 0, st>>>(b);
    }
    // one wavefront per run where the capacity allows that many runs (a wavefront's runs are a chain of memory round trips:
    // 16 runs per wavefront take 63 us on the 3.6 M-triangle stand-in, 4 runs 52 us)
    const uint32_t bb_blocks = (b.ref_cap / 64u + 4u) / 4u;
    k_block_bounds<<<bb_blocks < 1u ? 1u : (bb_blocks > 4096u ? 4096u : bb_blocks), 256, 0, st>>>(b);
    return hipGetLastError();
}

__global__ void k_extract_cell_tris(const float4* __restrict__ recs, uint32_t* out, uint32_t n, int indexed) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = indexed ? reinterpret_cast<const uint32_t*>(recs)[i] : __float_as_uint(recs[3ull * i].w);
}

hipError_t rt_launch_extract_cell_tris(const float4* recs, uint32_t* out, uint32_t n_refs, int indexed, hipStream_t st) {
    if (n_refs == 0) return hipSuccess;
    k_extract_cell_tris<<<(n_refs + 255) / 256, 256, 0, st>>>(recs, out, n_refs, indexed);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// render: ray generation, 3D-DDA walk, ray-triangle tests, shading, shadow ray, framebuffer writes
//
// Two kernels per frame.
//   k_classify  one wavefront per 8x8 pixel tile: generates the tile's rays, runs the grid-bounds test of
//               GridAccel::Intersect, writes the background for tiles no ray of which enters the grid and
//               appends every other tile to a work list (one atomic per listed tile).
//   k_trace     persistent wavefronts pull tiles from that list. A wavefront walks its 64 rays through the
//               grid together: every lane skips ahead through empty voxels on its own (occupancy bitmap in
//               LDS), then the wavefront repeatedly picks one occupied voxel that some of its lanes have
//               reached, stages that voxel's triangle records into LDS with coalesced loads (converting the
//               edge vectors to double once per triangle instead of once per ray), and the lanes standing in
//               that voxel test the staged triangles in ascending index order. Lanes that wait in another
//               voxel are picked in a later round, so rays of one tile meet at the voxels they share.
// ------------------------------------------------------------------------------------------------

#define RT_WAVES_PER_BLOCK 4
#define RT_CULL_LANE_RUNS 12                        // voxels with more 64-record runs than this are culled run-parallel
#define RT_MAX_BITMAP_WORDS ((RT_MAX_CELLS + 31) / 32)

struct GridU {               // grid description held in scalar registers
    float lo[3], hi[3], width[3], inv_width[3];
    int n[3];
};

__device__ __forceinline__ GridU load_grid(const RtGrid* gp) {
    GridU g;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g.lo[a] = gp->lo[a]; g.hi[a] = gp->hi[a]; g.width[a] = gp->width[a]; g.inv_width[a] = gp->inv_width[a];
        g.n[a] = gp->n[a];
    }
    return g;
}

// The `@phase NAME` comments from here on are read by tools/isa_attrib.py: the instructions generated from the lines after a
// marker (up to the next one) are charged to NAME in the per-phase instruction counts of the trace kernel.
// @phase other
struct Nearest {             // the reference's two globals, per ray (Serial/geometry.h:10-11)
    float t;
    int id;
};

// Loop counts of wave_walk in the counting build, per wavefront: executions of a loop body or branch by the wavefront as a
// whole. Multiplied with the instructions each body holds they say where the kernel's instructions go.
enum {
    RT_LOOP_ROUNDS_AT_VOXEL,  // rounds in which some ray stands at an occupied voxel (the others only step)
    RT_LOOP_ADVANCE,          // rounds whose first, unconditional voxel step ran
    RT_LOOP_STEP_ITER,        // iterations of the empty-voxel loop
    RT_LOOP_CULL_A_ROUNDS,    // rounds with lane = ray culling
    RT_LOOP_CULL_A_ITER,      // its iterations (runs of the most crowded voxel among the lanes)
    RT_LOOP_CULL_A_LANES,     // (ray, run) box tests those iterations held
    RT_LOOP_CULL_B_VOXELS,    // voxels culled lane = run
    RT_LOOP_CULL_B_RAYS,      // rays broadcast there
    RT_LOOP_ITEMGEN_CALLS,    // calls of the item generator
    RT_LOOP_ITEMGEN_ITER,     // iterations of its loop
    RT_LOOP_ITEMGEN_VOXELS,   // voxels it opened
    RT_LOOP_ITEMS,            // items tested
    RT_LOOP_ITEMS_WITH_RAYS,  // items with at least one live ray
    RT_LOOP_TRIPS,            // (run, ray) trips
    RT_LOOP_TRIPS_CAND,       // trips with a candidate triangle (divisions)
    RT_LOOP_TRIPS_PASS,       // trips with a passing triangle
    RT_LOOP_PASS_LANES,       // passing triangles applied to a ray's summary
    RT_LOOP_WALKS,            // calls of wave_walk with a live ray
    RT_LOOP_FILTER_KEPT,      // trips left after the single-precision pre-filter (exact trips run)
    RT_LOOP_WORDS
};
struct WalkCount {           // work counters of one ray (only maintained by the counting build of the kernel)
    uint32_t entered, steps, tests;
    uint32_t rounds, slots, lanes;   // per wavefront (identical in every lane): scheduling rounds, voxel visits, triangle tests executed
    uint32_t cyc_step, cyc_test;     // shader cycles spent in the two phases
    uint32_t items, trips, cyc_cull; // record runs streamed, (run, ray) test iterations, cycles of the run culling
    uint32_t fb_items, fb_hits;      // kept by every build of the kernel: record runs streamed, (run, ray) trips with a candidate triangle
    // counting build: how often the wavefront went through each loop of wave_walk (RT_LOOP_* below), whatever its lanes did
    uint32_t loops[RT_LOOP_WORDS];
};

// @phase raygen
// Primary ray of pixel (x, y), Serial/raytracer.cpp:153-161; xw and yw are evaluated in double.
__device__ __forceinline__ void primary_ray(const RtCamera& cam, const RtFrame& f, int x, int y, float& dx, float& dy, float& dz) {
    dx = 0.f + (-cam.w[0]) * cam.focal_distance;
    dy = 0.f + (-cam.w[1]) * cam.focal_distance;
    dz = 0.f + (-cam.w[2]) * cam.focal_distance;
    const float xw = (float)((double)cam.aspect * ((double)x - (double)f.width / 2.0 + 0.5) / (double)f.width);
    const float yw = (float)(((double)y - (double)f.height / 2.0 + 0.5) / (double)f.height);
    dx = dx + cam.u[0] * xw; dy = dy + cam.u[1] * xw; dz = dz + cam.u[2] * xw;
    dx = dx + cam.v[0] * yw; dy = dy + cam.v[1] * yw; dz = dz + cam.v[2] * yw;
    normalize3(dx, dy, dz);
}

// @phase walk_setup
// Entry of GridAccel::Intersect, Serial/grid.h:172-175: rayT = mint when ray(mint) is inside the bounds
// (boundingBox::Inside, Serial/geometry.h:287-289), else the slab test's t0 (boundingBox::Intersect,
// Serial/geometry.h:291-316), else the ray misses the grid.
__device__ __forceinline__ bool grid_entry(const GridU& g, float ox, float oy, float oz, float dx, float dy, float dz, float mint,
                                           float& rayT) {
    const float o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
    const float px = ox + dx * mint, py = oy + dy * mint, pz = oz + dz * mint;
    const bool inside = px >= g.lo[0] && px <= g.hi[0] && py >= g.lo[1] && py <= g.hi[1] && pz >= g.lo[2] && pz <= g.hi[2];
    if (inside) { rayT = mint; return true; }
    float t0 = mint, t1 = INFINITY;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float inv = 1.f / d[i];
        float tn = (g.lo[i] - o[i]) * inv;
        float tf = (g.hi[i] - o[i]) * inv;
        if (tn > tf) { float x = tn; tn = tf; tf = x; }
        t0 = tn > t0 ? tn : t0;
        t1 = tf < t1 ? tf : t1;
        if (t0 > t1) ok = false;      // the reference returns here; later slabs cannot undo it
    }
    rayT = t0;
    return ok;
}

// @phase raygen
// Pixel of lane `lane` in tile `tile` of this part: x, the row inside the part's buffers, the row in the frame.
__device__ __forceinline__ void tile_pixel(const RtFrame& f, int tile, int lane, int& x, int& y_local, int& y) {
    const int tiles_x = (f.width + RT_TILE - 1) / RT_TILE;
    const int tx = tile % tiles_x, band = tile / tiles_x;
    x = tx * RT_TILE + (lane & 7);
    y_local = band * RT_TILE + (lane >> 3);
    y = (band * f.part_count + f.part_index) * RT_TILE + (lane >> 3);
}

// @phase write
__device__ __forceinline__ void write_pixel(const RtFrame& f, const RtOutputs& o, int x, int y_local, float cr, float cg, float cb, int id) {
    if (x < f.width && y_local < f.out_rows) {
        const size_t pix = (size_t)y_local * f.width + x;
        if (o.rgb_f32) { o.rgb_f32[3 * pix] = cr; o.rgb_f32[3 * pix + 1] = cg; o.rgb_f32[3 * pix + 2] = cb; }
        if (o.rgb_u8) {
            // PPM quantisation, Serial/raytracer.cpp:181-183: (unsigned char)(min(1, c/255)*255); a NaN gives 255
            // through a 32-bit integer, low byte kept: what the reference's x86 code does with a negative value as well
            o.rgb_u8[3 * pix] = (uint8_t)(uint32_t)(int32_t)(min_std(1.f, cr / 255) * 255);
            o.rgb_u8[3 * pix + 1] = (uint8_t)(uint32_t)(int32_t)(min_std(1.f, cg / 255) * 255);
            o.rgb_u8[3 * pix + 2] = (uint8_t)(uint32_t)(int32_t)(min_std(1.f, cb / 255) * 255);
        }
        if (o.hit_id) o.hit_id[pix] = id;
    }
}

// @phase other
// work[0] = number of listed tiles, work[RT_WORK_HEADS + RT_WORK_HEAD_STRIDE * q] = dequeue head q; zeroed before each frame.
// Runs over the tile rectangle [tx0, tx0+ntx) x [tb0, tb0+ntb) (tile columns x local bands) that can contain rays
// entering the grid; the host has cleared the outputs to the background, so tiles that are not listed need no write.
// tiles per wavefront: 16 on large frames (fewer atomics), 4 on small ones (shorter chains per wavefront)
template <int RT_CLS_TILES_PER_WAVE>
__global__ void __launch_bounds__(256) k_classify(RtSceneDev s, RtCamera cam, RtFrame f, uint32_t* work, uint32_t* tile_list, int tx0,
                                                  int ntx, int tb0, int ntb, unsigned long long* stamp) {
    __shared__ uint32_t wave_cnt[4];
    stamp_now(stamp);
    __shared__ uint32_t block_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles_x = (f.width + RT_TILE - 1) / RT_TILE;
    const GridU g = load_grid(s.grid);
    // Every wavefront looks at RT_CLS_TILES_PER_WAVE tiles; the tiles to list are appended with ONE atomic per block (a
    // single counter takes about a hundred atomics per microsecond: with few tiles per block the kernel is that long).
    // First the four corner pixels of all its tiles at once (lane = tile * 4 + corner): one ray that enters settles
    // a tile, and on frames the geometry fills that is nearly every tile. Only tiles none of whose corners enters get the
    // test of all 64 rays.
    const int first = (blockIdx.x * 4 + wv) * RT_CLS_TILES_PER_WAVE;
    const int n_cls = ntx * ntb;
    auto tile_of = [&](int t) { return (tb0 + t / ntx) * tiles_x + tx0 + t % ntx; };
    unsigned long long corner_hits;
    {
        const int t = first + (lane >> 2), corner = lane & 3;
        bool enters = false;
        if (lane < 4 * RT_CLS_TILES_PER_WAVE && t < n_cls) {
            int x, y_local, y;
            tile_pixel(f, tile_of(t), (corner & 1 ? 7 : 0) + (corner & 2 ? 56 : 0), x, y_local, y);
            // partial tiles at the frame's edges: the last pixel inside the frame is their corner
            if (x >= f.width) x = f.width - 1;
            if (y >= f.height) y = (y - (corner & 2 ? 7 : 0) < f.height) ? f.height - 1 : -1;     // -1: the tile lies below the frame
            if (y >= 0) {
                float dx, dy, dz, rayT;
                primary_ray(cam, f, x, y, dx, dy, dz);
                enters = grid_entry(g, cam.pos[0], cam.pos[1], cam.pos[2], dx, dy, dz, 0.f, rayT);
            }
        }
        corner_hits = __ballot(enters);
    }
    uint32_t listed = 0;                       // bit k: tile first + k is listed (the same in every lane)
    for (int k = 0; k < RT_CLS_TILES_PER_WAVE && first + k < n_cls; ++k) {
        bool any = ((corner_hits >> (4 * k)) & 0xfull) != 0ull;
        if (!any) {
            int x, y_local, y;
            tile_pixel(f, tile_of(first + k), lane, x, y_local, y);
            bool enters = false;
            if (x < f.width && y < f.height) {
                float dx, dy, dz, rayT;
                primary_ray(cam, f, x, y, dx, dy, dz);
                enters = grid_entry(g, cam.pos[0], cam.pos[1], cam.pos[2], dx, dy, dz, 0.f, rayT);
            }
            any = __ballot(enters) != 0ull;
        }
        if (any) listed |= 1u << k;
    }
    const uint32_t n_mine = (uint32_t)__popc(listed);
    if (lane == 0) wave_cnt[wv] = n_mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        block_base = total ? atomicAdd(&work[0], total) : 0u;
    }
    __syncthreads();
    if (lane < RT_CLS_TILES_PER_WAVE && ((listed >> lane) & 1u)) {
        uint32_t pos = block_base;
        for (int k = 0; k < wv; ++k) pos += wave_cnt[k];
        tile_list[pos + (uint32_t)__popc(listed & ((1u << lane) - 1u))] = (uint32_t)tile_of(first + lane);
    }
}

// @phase walk_setup
// DDA state of one ray, Serial/grid.h:182-203. Pos/Step/Out of the reference are kept in an equivalent form:
// the linear voxel index, its increment per axis and the number of steps left before the ray leaves the grid
// along each axis (Pos == Out after left+1 steps).
struct Dda {
    // scalars on purpose: with arrays the compiler turns the per-axis selects below into indexed scratch loads
    float next0, next1, next2, delta0, delta1, delta2;
    int left0, left1, left2;
    int cstep0, cstep1, cstep2;
    int cell;
};

__device__ __forceinline__ void dda_axis(const GridU& g, int a, int stride, float o, float d, float rayT, float& next, float& delta,
                                         int& left, int& cstep, int& cell) {
    const float gi = o + d * rayT;
    const int pos = pos_to_voxel(gi, g.lo[a], g.inv_width[a], g.n[a]);
    if (d >= 0) {
        next = rayT + ((g.lo[a] + (float)(pos + 1) * g.width[a]) - gi) / d;
        delta = g.width[a] / d;
        left = g.n[a] - 1 - pos;
        cstep = stride;
    } else {
        next = rayT + ((g.lo[a] + (float)pos * g.width[a]) - gi) / d;
        delta = -g.width[a] / d;
        left = pos;
        cstep = -stride;
    }
    cell += pos * stride;
}

__device__ __forceinline__ void dda_setup(Dda& w, const GridU& g, float ox, float oy, float oz, float dx, float dy, float dz, float rayT) {
    // cells of the bordered layout: voxel (0, 0, 0) is cell 1 + row + layer
    const int row = g.n[0] + 2, layer = row * (g.n[1] + 2);
    w.cell = 1 + row + layer;
    dda_axis(g, 0, 1, ox, dx, rayT, w.next0, w.delta0, w.left0, w.cstep0, w.cell);
    dda_axis(g, 1, row, oy, dy, rayT, w.next1, w.delta1, w.left1, w.cstep1, w.cell);
    dda_axis(g, 2, layer, oz, dz, rayT, w.next2, w.delta2, w.left2, w.cstep2, w.cell);
}

// @phase step
// One step of the walk loop, Serial/grid.h:216-227, plus the early-exit rule. Returns false when the ray is done.
// stepAxis comes from the 3-bit compare table {2,1,2,1,2,2,0,0}: axis 0 iff next0 < next1 and next0 < next2,
// axis 1 iff not (next0 < next1) and next1 < next2, else axis 2 (NaNs compare false in both forms). Written without
// branches on the axis.
__device__ __forceinline__ bool dda_advance(Dda& w, bool early_hit, float best_t, float slack, float* crossed_t = nullptr) {
    // The optimiser has to treat the state as register values: without this it rewrites the selects below as one load
    // from a selected address, which pins the whole state in scratch memory. (In place: no copies are made.)
    asm("" : "+v"(w.next0), "+v"(w.next1), "+v"(w.next2), "+v"(w.delta0), "+v"(w.delta1), "+v"(w.delta2));
    asm("" : "+v"(w.left0), "+v"(w.left1), "+v"(w.left2), "+v"(w.cstep0), "+v"(w.cstep1), "+v"(w.cstep2));
    const float n0 = w.next0, n1 = w.next1, n2 = w.next2;
    // two levels: the nearer of axes 0 and 1 (axis 1 on a tie), then that one against axis 2 (axis 2 on a tie); the same
    // choice as the table's, with one compare less and lane masks that combine in scalar registers
    const bool c01 = n0 < n1;
    const float na = c01 ? n0 : n1;
    const bool is01 = na < n2;                      // else axis 2
    const bool is0 = is01 && c01;
    const bool is1 = is01 && !c01;
    const float nx_t = is01 ? na : n2;
    const int la = c01 ? w.left0 : w.left1;
    const int lf = is01 ? la : w.left2;
    // the ray is done when it has a hit nearer than the next voxel (early exit) or stands in the last voxel of the
    // axis it would leave through (Pos[stepAxis] == Out[stepAxis]); one branch for both
    const float lim = best_t + fabsf(best_t) * slack;
    const bool go = !((early_hit && lim < nx_t) || lf == 0);
    if (go) {
        const float da = c01 ? w.delta0 : w.delta1;
        const float dl = is01 ? da : w.delta2;
        const int ca = c01 ? w.cstep0 : w.cstep1;
        const int cs = is01 ? ca : w.cstep2;
        if (crossed_t) *crossed_t = nx_t;
        const float nn = nx_t + dl;
        w.next0 = is0 ? nn : n0;
        w.next1 = is1 ? nn : n1;
        w.next2 = is01 ? n2 : nn;
        w.left0 = is0 ? lf - 1 : w.left0;
        w.left1 = is1 ? lf - 1 : w.left1;
        w.left2 = is01 ? w.left2 : lf - 1;
        w.cell += cs;
    }
    return go;
}

// empty voxels a ray may cross in one round of wave_walk before the round goes on without it
#ifndef RT_STEPS_PER_ROUND
#define RT_STEPS_PER_ROUND 8
#endif
// The empty-voxel loop of the walk for a whole wavefront, scheduled by hand: every lane in `moving` whose cell is empty
// takes dda_advance steps (same selections, same additions, so the same crossings to the bit) until it stands at a cell
// whose bit is set, has taken RT_STEPS_PER_ROUND steps, or has a hit nearer than `lim` where `no_early` does not hold its
// bit; the lanes that stopped for a hit come back as a mask. There is no count of the voxels left along each axis: a ray
// that leaves the grid steps into a border cell, whose bit is set (rt_device.h), and the caller finds it done there.
// Why by hand: written as a loop over dda_advance the compiler spends 33 scalar instructions per voxel on saving, narrowing
// and restoring the execution mask around the loop's two exit conditions and the axis selects; here the three axes are
// updated under three execution masks (an addition and an index step each) and the loop keeps one mask of the lanes still
// stepping: 15 vector and 15 scalar instructions per voxel.
// A compare that writes scalar registers is followed by two other instructions before a vector instruction reads them.
__device__ __forceinline__ unsigned long long step_through_empty(Dda& w, const uint32_t* bitmap, float lim, unsigned long long no_early,
                                                                 unsigned long long moving) {
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)bitmap;
    unsigned long long done, saved, c01, go, tm, ts;
    uint32_t t0, t1;
    int cnt;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_mov_b64 %[done], 0\n\t"
        "s_and_b64 exec, exec, %[moving]\n\t"
        "s_movk_i32 %[cnt], %[steps]\n\t"
        "s_cbranch_execz .Lstep_end%=\n"
        ".Lstep_loop%=:\n\t"
        "v_cmp_lt_f32_e64 %[c01], %[n0], %[n1]\n\t"
        "v_lshrrev_b32_e32 %[t0], 5, %[cell]\n\t"
        "v_lshl_add_u32 %[t0], %[t0], 2, %[base]\n\t"
        "v_cndmask_b32_e64 %[t1], %[n1], %[n0], %[c01]\n\t"        // the nearer crossing of axes 0 and 1 (axis 1 on a tie)
        "ds_read_b32 %[t0], %[t0]\n\t"
        "v_cmp_lt_f32_e32 vcc, %[t1], %[n2]\n\t"                    // ... is nearer than that of axis 2 (axis 2 on a tie)
        "s_nop 1\n\t"
        "v_cndmask_b32_e32 %[t1], %[n2], %[t1], vcc\n\t"             // t1: crossing into the next cell
        "v_cmp_nlt_f32_e64 %[go], %[lim], %[t1]\n\t"                // no hit nearer than the crossing
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[t0], %[t0], %[cell], 1\n\t"                    // the cell's bit (offset = the low five bits of the index)
        "s_or_b64 %[go], %[go], %[noearly]\n\t"
        "v_cmp_eq_u32_e64 %[tm], 0, %[t0]\n\t"                      // lanes at an empty cell
        "s_andn2_b64 %[ts], %[tm], %[go]\n\t"                       // of them: done
        "s_and_b64 %[go], %[tm], %[go]\n\t"                         //          stepping on
        "s_or_b64 %[done], %[done], %[ts]\n\t"
        "s_and_b64 %[ts], %[c01], vcc\n\t"                          // along axis 0
        "s_and_b64 exec, %[ts], %[go]\n\t"
        "v_add_f32_e32 %[n0], %[n0], %[d0]\n\t"
        "v_add_u32_e32 %[cell], %[cell], %[c0]\n\t"
        "s_andn2_b64 %[ts], vcc, %[c01]\n\t"                        // along axis 1
        "s_and_b64 exec, %[ts], %[go]\n\t"
        "v_add_f32_e32 %[n1], %[n1], %[d1]\n\t"
        "v_add_u32_e32 %[cell], %[cell], %[c1]\n\t"
        "s_andn2_b64 exec, %[go], vcc\n\t"                          // along axis 2
        "v_add_f32_e32 %[n2], %[n2], %[d2]\n\t"
        "v_add_u32_e32 %[cell], %[cell], %[c2]\n\t"
        "s_mov_b64 exec, %[go]\n\t"
        "s_cbranch_execz .Lstep_end%=\n\t"
        "s_add_i32 %[cnt], %[cnt], -1\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 .Lstep_loop%=\n"
        ".Lstep_end%=:\n\t"
        "s_mov_b64 exec, %[saved]"
        : [cell] "+v"(w.cell), [n0] "+v"(w.next0), [n1] "+v"(w.next1), [n2] "+v"(w.next2),
          [done] "=&s"(done), [saved] "=&s"(saved), [c01] "=&s"(c01), [go] "=&s"(go), [tm] "=&s"(tm), [ts] "=&s"(ts),
          [cnt] "=&s"(cnt), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [d0] "v"(w.delta0), [d1] "v"(w.delta1), [d2] "v"(w.delta2), [c0] "v"(w.cstep0), [c1] "v"(w.cstep1), [c2] "v"(w.cstep2),
          [lim] "v"(lim), [base] "s"(base), [noearly] "s"(no_early), [moving] "s"(moving), [steps] "n"(RT_STEPS_PER_ROUND)
        : "vcc", "scc");
    return done;
}

// The same loop for the case that no moving lane has a hit yet (most rounds: a ray that has one stops within a step or two,
// and a shadow ray stops with its first): nothing can end a lane's walk but a set bit, so the crossing itself is not needed,
// only its axis. 13 vector instructions per voxel.
__device__ __forceinline__ void step_through_empty_no_hit(Dda& w, const uint32_t* bitmap, unsigned long long moving) {
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t*)bitmap;
    unsigned long long saved, c01, go, ts;
    uint32_t t0, t1;
    int cnt;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_and_b64 exec, exec, %[moving]\n\t"
        "s_movk_i32 %[cnt], %[steps]\n\t"
        "s_cbranch_execz .Lstepn_end%=\n"
        ".Lstepn_loop%=:\n\t"
        "v_cmp_lt_f32_e64 %[c01], %[n0], %[n1]\n\t"
        "v_lshrrev_b32_e32 %[t0], 5, %[cell]\n\t"
        "v_lshl_add_u32 %[t0], %[t0], 2, %[base]\n\t"
        "v_cndmask_b32_e64 %[t1], %[n1], %[n0], %[c01]\n\t"        // the nearer crossing of axes 0 and 1 (axis 1 on a tie)
        "ds_read_b32 %[t0], %[t0]\n\t"
        "v_cmp_lt_f32_e32 vcc, %[t1], %[n2]\n\t"                    // ... is nearer than that of axis 2 (axis 2 on a tie)
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[t0], %[t0], %[cell], 1\n\t"
        "v_cmp_eq_u32_e64 %[go], 0, %[t0]\n\t"                      // lanes at an empty cell step on
        "s_and_b64 %[ts], %[c01], vcc\n\t"                          // along axis 0
        "s_and_b64 exec, %[ts], %[go]\n\t"
        "v_add_f32_e32 %[n0], %[n0], %[d0]\n\t"
        "v_add_u32_e32 %[cell], %[cell], %[c0]\n\t"
        "s_andn2_b64 %[ts], vcc, %[c01]\n\t"                        // along axis 1
        "s_and_b64 exec, %[ts], %[go]\n\t"
        "v_add_f32_e32 %[n1], %[n1], %[d1]\n\t"
        "v_add_u32_e32 %[cell], %[cell], %[c1]\n\t"
        "s_andn2_b64 exec, %[go], vcc\n\t"                          // along axis 2
        "v_add_f32_e32 %[n2], %[n2], %[d2]\n\t"
        "v_add_u32_e32 %[cell], %[cell], %[c2]\n\t"
        "s_mov_b64 exec, %[go]\n\t"
        "s_cbranch_execz .Lstepn_end%=\n\t"
        "s_add_i32 %[cnt], %[cnt], -1\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 .Lstepn_loop%=\n"
        ".Lstepn_end%=:\n\t"
        "s_mov_b64 exec, %[saved]"
        : [cell] "+v"(w.cell), [n0] "+v"(w.next0), [n1] "+v"(w.next1), [n2] "+v"(w.next2),
          [saved] "=&s"(saved), [c01] "=&s"(c01), [go] "=&s"(go), [ts] "=&s"(ts), [cnt] "=&s"(cnt), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [d0] "v"(w.delta0), [d1] "v"(w.delta1), [d2] "v"(w.delta2), [c0] "v"(w.cstep0), [c1] "v"(w.cstep1), [c2] "v"(w.cstep2),
          [base] "s"(base), [moving] "s"(moving), [steps] "n"(RT_STEPS_PER_ROUND)
        : "vcc", "scc");
}

// One dda_advance step for the lanes in `moving`, in the same form (a lane in the last voxel of an axis steps into the
// border cell); returns the lanes that stopped for a hit instead.
__device__ __forceinline__ unsigned long long step_once(Dda& w, float lim, unsigned long long no_early, unsigned long long moving) {
    unsigned long long done, saved, c01, go, ts;
    uint32_t t1;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_and_b64 exec, exec, %[moving]\n\t"
        "v_cmp_lt_f32_e64 %[c01], %[n0], %[n1]\n\t"
        "s_mov_b64 %[done], exec\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_e64 %[t1], %[n1], %[n0], %[c01]\n\t"
        "v_cmp_lt_f32_e32 vcc, %[t1], %[n2]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e32 %[t1], %[n2], %[t1], vcc\n\t"
        "v_cmp_nlt_f32_e64 %[go], %[lim], %[t1]\n\t"
        "s_or_b64 %[go], %[go], %[noearly]\n\t"
        "s_and_b64 %[go], %[go], %[done]\n\t"
        "s_andn2_b64 %[done], %[done], %[go]\n\t"
        "s_and_b64 %[ts], %[c01], vcc\n\t"
        "s_and_b64 exec, %[ts], %[go]\n\t"
        "v_add_f32_e32 %[n0], %[n0], %[d0]\n\t"
        "v_add_u32_e32 %[cell], %[cell], %[c0]\n\t"
        "s_andn2_b64 %[ts], vcc, %[c01]\n\t"
        "s_and_b64 exec, %[ts], %[go]\n\t"
        "v_add_f32_e32 %[n1], %[n1], %[d1]\n\t"
        "v_add_u32_e32 %[cell], %[cell], %[c1]\n\t"
        "s_andn2_b64 exec, %[go], vcc\n\t"
        "v_add_f32_e32 %[n2], %[n2], %[d2]\n\t"
        "v_add_u32_e32 %[cell], %[cell], %[c2]\n\t"
        "s_mov_b64 exec, %[saved]"
        : [cell] "+v"(w.cell), [n0] "+v"(w.next0), [n1] "+v"(w.next1), [n2] "+v"(w.next2),
          [done] "=&s"(done), [saved] "=&s"(saved), [c01] "=&s"(c01), [go] "=&s"(go), [ts] "=&s"(ts), [t1] "=&v"(t1)
        : [d0] "v"(w.delta0), [d1] "v"(w.delta1), [d2] "v"(w.delta2), [c0] "v"(w.cstep0), [c1] "v"(w.cstep1), [c2] "v"(w.cstep2),
          [lim] "v"(lim), [noearly] "s"(no_early), [moving] "s"(moving)
        : "vcc", "scc");
    return done;
}

// @phase kernel_setup
// Occupancy bitmap of the grid into LDS, all loads of a thread in flight together (16 bytes each).
__device__ __forceinline__ void bitmap_to_lds(uint32_t* bitmap, const uint32_t* occupancy, int words) {
    const int quads = words >> 2;
    const uint4* src = reinterpret_cast<const uint4*>(occupancy);
    uint4* dst = reinterpret_cast<uint4*>(bitmap);
    for (int i = threadIdx.x; i < quads; i += 4 * blockDim.x) {
        uint4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = i + j * (int)blockDim.x; q[j] = src[k < quads ? k : i]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = i + j * (int)blockDim.x; if (k < quads) dst[k] = q[j]; }
    }
    for (int i = (quads << 2) + threadIdx.x; i < words; i += blockDim.x) bitmap[i] = occupancy[i];
    __syncthreads();
}

// @phase other
// COST ESTIMATE per listed tile, from the rays of four of its pixels and without any triangle test: each ray
// is walked through the whole grid and the occupied voxels it meets and their triangle counts are summed; a stand-in
// shadow ray, started where the primary ray enters its first occupied voxel, is walked the same way. The estimate is a
// weighted sum of the four numbers. Tiles whose rays graze the surface or cross crowded voxels get a large estimate;
// they are queued first and cut into smaller work units (k_bucket), so the longest-running units start early and the
// frame does not end on a single wavefront.
// work[] words: [0] listed tiles, [RT_WORK_HEADS + RT_WORK_HEAD_STRIDE * q] dequeue head q; tiles per queue list from RT_WORK_LIST_COUNTS on, in another cache
// line than the dequeue head: every dequeue reads them, and loads from the line all wavefronts hit with atomics queue up.
#define RT_EST_CELLS 12          // occupied voxels noted per estimate ray (LDS: 4 bytes x 256 lanes each)

// Walks one ray through the grid. The walk itself only touches LDS: the occupied voxels are noted in a per-lane list
// (the first RT_EST_CELLS of them) and their triangle counts are then fetched with independent loads, several in
// flight, instead of one dependent memory round trip per voxel; longer walks are scaled up.
__device__ __forceinline__ void estimate_walk(const RtSceneDev& s, const GridU& g, const uint32_t* bitmap, uint32_t* cells, float ox, float oy,
                                              float oz, float dx, float dy, float dz, float mint, uint32_t& tris, uint32_t& voxels, float& t_first) {
    tris = 0; voxels = 0; t_first = -1.f;
    float rayT;
    if (!grid_entry(g, ox, oy, oz, dx, dy, dz, mint, rayT)) return;
    Dda w;
    dda_setup(w, g, ox, oy, oz, dx, dy, dz, rayT);
    bool alive = true;
    float t_in = rayT;
    uint32_t met = 0;
    while (alive) {
        if ((bitmap[w.cell >> 5] >> (w.cell & 31)) & 1u) {
            if (met == 0) t_first = t_in;
            if (met < RT_EST_CELLS) cells[met * 256 + threadIdx.x] = (uint32_t)w.cell;
            ++met;
        }
        alive = dda_advance(w, false, 0.f, 0.f, &t_in);
    }
    const uint32_t kept = met < RT_EST_CELLS ? met : RT_EST_CELLS;
    uint32_t c = 0;
    for (uint32_t k = 0; k < kept; k += 8) {           // sixteen loads in flight: the kernel is a chain of round trips
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t cell = cells[((k + j < kept) ? k + j : k) * 256 + threadIdx.x];
            lo[j] = s.cell_start[cell]; hi[j] = s.cell_start[cell + 1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k + j < kept) c += hi[j] - lo[j];
    }
    if (met > kept) c = (uint32_t)((unsigned long long)c * met / kept);
    tris = c; voxels = met;
}

// SPLIT (small meshes: few tiles, and the frame waits for this kernel): eight lanes per tile, four in the first two
// wavefronts of the block walk the primary sample rays and four in the other two the stand-in shadow rays; a shadow lane
// first repeats its primary ray up to the first occupied voxel. A wavefront then has one walk behind it instead of two.
template <bool SPLIT>
__global__ void __launch_bounds__(256) k_estimate(RtSceneDev s, RtCamera cam, RtShade sh, RtFrame f, const uint32_t* work, const uint32_t* tile_list,
                                                  uint32_t* cost, unsigned long long* cost_sum, uint32_t* feat_by_tile, uint32_t n_tiles, float4 wgt,
                                                  uint32_t* cost_by_tile) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(lds_raw);
    const uint32_t n = work[0];
    constexpr int LANES_LOG = SPLIT ? 3 : 2;
    if (blockIdx.x * (blockDim.x >> LANES_LOG) >= n) return;
    bitmap_to_lds(bitmap, s.occupancy, (s.grid->n_cells + 31) >> 5);
    uint32_t* cells = bitmap + ((s.grid->n_cells + 31) >> 5);
    // four lanes per tile and ray kind, one sample pixel each: (1,1), (6,1), (1,6), (6,6)
    const uint32_t i = SPLIT ? blockIdx.x * 32u + ((threadIdx.x & 127u) >> 2) : (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int smp = threadIdx.x & 3;
    const bool shadow_lane = SPLIT && (threadIdx.x & 128);
    __shared__ uint32_t shadow_sums[2][32];
    uint32_t c = 0;
    uint32_t tp = 0, vp = 0, ts = 0, vs = 0;
    int tile = 0;
    if (i < n) {
        tile = (int)tile_list[i];
        int x, y_local, y;
        tile_pixel(f, tile, ((smp & 2) ? 48 : 8) + ((smp & 1) ? 6 : 1), x, y_local, y);
        if (x >= f.width || y >= f.height) tile_pixel(f, tile, 0, x, y_local, y);
        const GridU g = load_grid(s.grid);
        float dx, dy, dz;
        primary_ray(cam, f, x, y, dx, dy, dz);
        float t_first = -1.f, unused;
        if (!shadow_lane) {
            estimate_walk(s, g, bitmap, cells, cam.pos[0], cam.pos[1], cam.pos[2], dx, dy, dz, 0.f, tp, vp, t_first);
        } else {
            float rayT;
            if (grid_entry(g, cam.pos[0], cam.pos[1], cam.pos[2], dx, dy, dz, 0.f, rayT)) {
                Dda w;
                dda_setup(w, g, cam.pos[0], cam.pos[1], cam.pos[2], dx, dy, dz, rayT);
                bool alive = true;
                float t_in = rayT;
                while (alive) {
                    if ((bitmap[w.cell >> 5] >> (w.cell & 31)) & 1u) { t_first = t_in; break; }
                    alive = dda_advance(w, false, 0.f, 0.f, &t_in);
                }
            }
        }
        if (t_first >= 0.f && (shadow_lane || !SPLIT)) {
            const float px = cam.pos[0] + dx * t_first, py = cam.pos[1] + dy * t_first, pz = cam.pos[2] + dz * t_first;
            float sx = px - sh.light[0], sy = py - sh.light[1], sz = pz - sh.light[2];
            normalize3(sx, sy, sz);
            estimate_walk(s, g, bitmap, cells, px, py, pz, sx, sy, sz, sh.eps, ts, vs, unused);
        }
    }
    // the tile's numbers: sums over its four samples (neighbouring lanes)
    tp += __shfl_xor(tp, 1); vp += __shfl_xor(vp, 1); ts += __shfl_xor(ts, 1); vs += __shfl_xor(vs, 1);
    tp += __shfl_xor(tp, 2); vp += __shfl_xor(vp, 2); ts += __shfl_xor(ts, 2); vs += __shfl_xor(vs, 2);
    if (SPLIT) {
        const int slot = (threadIdx.x & 127) >> 2;
        if (shadow_lane && smp == 0) { shadow_sums[0][slot] = ts; shadow_sums[1][slot] = vs; }
        __syncthreads();
        if (!shadow_lane) { ts = shadow_sums[0][slot]; vs = shadow_sums[1][slot]; }
    }
    if (i < n && smp == 0 && !shadow_lane) {
        const float e = wgt.x * (float)tp + wgt.y * (float)vp + wgt.z * (float)ts + wgt.w * (float)vs;
        c = e < 4.0e9f ? (uint32_t)e : 4000000000u;
        cost[i] = c;
        if (cost_by_tile) cost_by_tile[tile] = c;      // kept for the frames after this one (same camera, same triangles)
        if (feat_by_tile) {
            feat_by_tile[tile] = tp; feat_by_tile[n_tiles + tile] = vp;
            feat_by_tile[2 * n_tiles + tile] = ts; feat_by_tile[3 * n_tiles + tile] = vs;
        }
    }
    unsigned long long t = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    // 32 partial sums: one address takes only about a hundred atomics per microsecond, and a 4K frame has 8,000 wavefronts here
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(&cost_sum[(blockIdx.x * 4u + (threadIdx.x >> 6)) & (RT_COST_SUM_PARTS - 1u)], t);
}

// QUEUE ORDER. The listed tiles are spread over RT_QUEUE_LISTS lists that the trace kernel serves one after the other:
//   lists 0..7   expensive tiles (estimate >= lim0 = 1/div0 of the even share of one wavefront), most expensive first,
//                one list per factor sqrt(2);
//   lists 8..11  cheap tiles (below lim1 = share/div1), one list per factor 2;
//   lists 12..15 the tiles in between, one list per quarter of the log range, served last: they are cut into small units
//                and level out what the estimates got wrong.
// Every list has its own unit size (RtFrame::unit_shifts).
__global__ void __launch_bounds__(256) k_bucket(uint32_t* work, const uint32_t* tile_list, const uint32_t* cost,
                                                const unsigned long long* cost_sum, uint32_t* lists, uint32_t cap,
                                                float div0, float div1, int n_waves, unsigned long long unit_shifts, int adapt,
                                                const RtGrid* grid, const uint32_t* cost_by_tile, unsigned long long* kept_total, int reuse) {
    // reuse: no estimate ran for this frame; the tiles' costs are the ones an earlier frame with the same camera and the same
    // triangles left in cost_by_tile, their total is in *kept_total
    const uint32_t n = work[0];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (blockIdx.x * blockDim.x >= n) return;
    // Unit sizes of the lists, for k_trace. The given sizes suit frames with up to about two listed tiles per wavefront (a
    // rank's share of a frame on several GPUs): there the frame time is the run time of its longest units, and the
    // expensive and the in-between tiles are cut into units of 16 and 32 rays. With more tiles per wavefront there are
    // enough units to balance the load anyway, and larger units are cheaper (the stepping, culling and shading of a unit
    // cost the same for 16 rays as for 64): the in-between lists go one size up from 2.5 tiles per wavefront and two from
    // 10, the expensive lists follow one step later. In a sparse grid (fewer than two references per voxel on average)
    // most voxels hold a handful of triangles that the rays test themselves, lane = ray, at a cost that does not depend on
    // the rays of the unit at all: there the expensive lists go up together with the others.
    if (blockIdx.x == 0 && threadIdx.x < RT_QUEUE_LISTS) {
        uint32_t sh = (uint32_t)((unit_shifts >> (4 * threadIdx.x)) & 15ull);
        if (adapt) {
            // adapt == 2: other frames are in flight beside this one and fill the GPU while its last units run, so the
            // largest units pay from fewer tiles per wavefront on
            uint32_t up = (2u * n >= 5u * (uint32_t)n_waves ? 1u : 0u) + (n >= (adapt == 2 ? 8u : 10u) * (uint32_t)n_waves ? 1u : 0u);
            if (threadIdx.x < 8 && !(grid->n_refs < 2u * (uint32_t)(grid->n[0] * grid->n[1] * grid->n[2]))) up = up > 0u ? up - 1u : 0u;
            sh = sh > up ? sh - up : 0u;
        }
        work[RT_WORK_LIST_COUNTS + RT_QUEUE_LISTS + threadIdx.x] = sh;
    }
    unsigned long long cost_total = 0;
    if (reuse) cost_total = *kept_total;
    else {
#pragma unroll 8
        for (uint32_t k = 0; k < RT_COST_SUM_PARTS; ++k) cost_total += cost_sum[k];
        if (kept_total && blockIdx.x == 0 && threadIdx.x == 0) *kept_total = cost_total;
    }
    const float share = (float)((double)cost_total / (double)n_waves);
    const float lim0 = fmaxf(share / div0, 1.f), lim1 = fmaxf(share / div1, 0.5f);
    int cls = -1;
    uint32_t tile = 0;
    if (i < n) {
        tile = tile_list[i];
        const float c = fmaxf((float)(reuse ? cost_by_tile[tile] : cost[i]), 0.25f);
        if (c >= lim0) {
            const int h = (int)(2.f * __log2f(c / lim0));
            cls = 7 - (h < 0 ? 0 : (h > 7 ? 7 : h));
        } else if (c >= lim1) {
            const int q = (int)(4.f * __log2f(c / lim1) / fmaxf(__log2f(lim0 / lim1), 1e-3f));
            cls = 12 + (3 - (q < 0 ? 0 : (q > 3 ? 3 : q)));
        } else {
            const int q = (int)__log2f(lim1 / c);
            cls = 8 + (q < 0 ? 0 : (q > 3 ? 3 : q));
        }
    }
    // places in the lists: counted per wavefront (ballots), summed per block in LDS, and ONE atomic per list and block
    // claims the block's places, the sixteen lists at once
    __shared__ uint32_t wave_cnt[4][RT_QUEUE_LISTS];
    __shared__ uint32_t block_base[RT_QUEUE_LISTS];
    const int wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t my_cnt = 0, my_rank = 0;          // lane k < 16: this wavefront's tiles for list k; every lane: its place among them
    for (int k = 0; k < RT_QUEUE_LISTS; ++k) {
        const unsigned long long m = __ballot(cls == k);
        if (lane == k) my_cnt = (uint32_t)__popcll(m);
        if (cls == k) my_rank = (uint32_t)__popcll(m & lt_mask);
    }
    if (lane < RT_QUEUE_LISTS) wave_cnt[wv][lane] = my_cnt;
    __syncthreads();
    if (threadIdx.x < RT_QUEUE_LISTS) {
        const uint32_t total = wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x] + wave_cnt[2][threadIdx.x] + wave_cnt[3][threadIdx.x];
        block_base[threadIdx.x] = total ? atomicAdd(&work[RT_WORK_LIST_COUNTS + threadIdx.x], total) : 0u;
    }
    __syncthreads();
    if (cls >= 0) {
        uint32_t pos = block_base[cls] + my_rank;
        for (int k = 0; k < wv; ++k) pos += wave_cnt[k][cls];
        lists[(size_t)cls * cap + pos] = tile;
    }
}

// @phase bcast
__device__ __forceinline__ float bcast_f32(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
__device__ __forceinline__ unsigned long long bcast_u64(unsigned long long v, int src_lane) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)v, src_lane);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), src_lane);
    return ((unsigned long long)hi << 32) | lo;
}
// Bitwise or of v over the 64 lanes, as a wavefront-uniform value: four shifts inside each row of 16 lanes, two row
// broadcasts, then the last lane holds everything (data-parallel primitives of the vector unit: no memory, no LDS).
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);      // row_shr:8
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);      // row_bcast:15
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);      // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Bit i of a wavefront-uniform 64-bit mask cleared or set in one scalar instruction (`m &= m - 1` is three, `m |= 1ull << i`
// two); the loops over the set bits of a lane mask below run millions of times per frame.
__device__ __forceinline__ void mask_clear_bit(unsigned long long& m, int i) { asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(i)); }
__device__ __forceinline__ void mask_set_bit(unsigned long long& m, int i) { asm("s_bitset1_b64 %0, %1" : "+s"(m) : "s"(i)); }
__device__ __forceinline__ double bcast_f64(double v, int src_lane) {
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)b, src_lane);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// @phase trip
// a*b - c*d for doubles that all hold float values: both products are exact in double (24 + 24 significant bits,
// exponents far inside the range), so the reference's mul, mul, sub rounds once, in the subtraction, and a multiply
// followed by one fused multiply-add rounds the same exact value once: identical bits, one instruction less.
__device__ __forceinline__ double minor_f32(double a, double b, double c, double d) {
    return __builtin_fma(a, b, -(c * d));
}

// @phase resolve
// Division with a shared reciprocal. The f64 division of this target is the sequence: scale both operands, v_rcp_f64
// of the denominator, two Newton steps, quotient estimate, residual, one fused correction (+ post-scaling), fix-up of
// special operands. When the biased exponents of numerator and denominator lie within 1023 +- 300 nothing is scaled and
// nothing is fixed up, so the steps up to the refined reciprocal depend on the denominator alone: div_rcp() is that
// part, div_with_rcp() the rest. Together they execute the same instructions on the same values as `num / den`.
__device__ __forceinline__ double div_rcp(double den) {
    double y = __builtin_amdgcn_rcp(den);
    y = __builtin_fma(y, __builtin_fma(-den, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-den, y, 1.0), y);
    return y;
}
__device__ __forceinline__ double div_with_rcp(double num, double den, double y) {
    const double q = num * y;
    return __builtin_fma(__builtin_fma(-den, q, num), y, q);
}
__device__ __forceinline__ uint32_t exponent_field(double x) { return ((uint32_t)__double2hiint(x) >> 20) & 0x7ffu; }
#define RT_DIV_EXP_LO (1023u - 300u)
#define RT_DIV_EXP_HI (1023u + 300u)

// @phase other
// Self-test of the shared-reciprocal division: count the pairs for which it differs from the plain division.
__global__ void k_selftest_div(const double* __restrict__ num, const double* __restrict__ den, uint32_t n, uint32_t* mismatches) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = num[i], b = den[i];
    const uint32_t ea = exponent_field(a), eb = exponent_field(b);
    if (ea < RT_DIV_EXP_LO || ea > RT_DIV_EXP_HI || eb < RT_DIV_EXP_LO || eb > RT_DIV_EXP_HI) return;
    const double fast = div_with_rcp(a, b, div_rcp(b)), plain = a / b;
    if (__double_as_longlong(fast) != __double_as_longlong(plain)) atomicAdd(mismatches, 1u);
}
hipError_t rt_launch_selftest_div(const double* num, const double* den, uint32_t n, uint32_t* mismatches, hipStream_t st) {
    if (n == 0) return hipSuccess;
    k_selftest_div<<<(n + 255) / 256, 256, 0, st>>>(num, den, n, mismatches);
    return hipGetLastError();
}

// @phase resolve
// Order-free form of the reference's nearest-hit update. Triangle::Intersect (Serial/geometry.h:162-175) compares the
// double hit distance t_c of every passing triangle with the FLOAT global_t and stores (float)t_c, so the outcome of a
// voxel's tests depends on their order (ascending triangle index, Serial/grid.h:24-29) through that rounding. The
// dependence has a closed form. With fl() the rounding to float, for the passing triangles c of ONE voxel visit let
//     f  = min over c of fl(t_c),    j0 = the smallest index with fl(t_c) == f,
//     L  = the largest index with fl(t_c) == f and t_c < f (compared as doubles), if there is one.
// Going through the candidates in ascending index from a state (t, id): the first candidate that rounds to f sets
// t = f (fl is monotonic, so t_c < t exactly when fl(t_c) < t or fl(t_c) == t and t_c < t), nothing that rounds to more
// than f can follow it, and every later candidate that rounds to f but lies below it as a double takes the index over.
// Hence after the visit
//     t > f :  (t, id) = (f, L if it exists, else j0);     t == f :  id = L if it exists;     t < f :  unchanged.
// (f, j0, L) can be accumulated in any order, so a voxel's records need no particular order in memory and its runs
// can be tested in any order. tests/test_merge_rule.py checks the rule against the sequential loop.
struct VoxelBest {
    float f;
    uint32_t j0;
    int L;
};
__device__ __forceinline__ void vb_reset(VoxelBest& v) { v.f = INFINITY; v.j0 = 0xffffffffu; v.L = -1; }
__device__ __forceinline__ void vb_add(VoxelBest& v, double t, int id) {
    const float fj = (float)t;                           // NaN: both comparisons fail, like t_c < global_t
    const int below = (t < (double)fj) ? id : -1;
    if (fj < v.f) { v.f = fj; v.j0 = (uint32_t)id; v.L = below; }
    else if (fj == v.f) { v.j0 = min(v.j0, (uint32_t)id); v.L = max(v.L, below); }
}
// @phase apply
__device__ __forceinline__ void vb_apply(const VoxelBest& v, Nearest& st) {
    if (v.f < st.t) { st.t = v.f; st.id = (v.L >= 0) ? v.L : (int)v.j0; }
    else if (v.f == st.t && v.L >= 0) st.id = v.L;
}

// @phase item
// One triangle record as a lane holds it during the test phase.
struct TriRec {
    float v0x, v0y, v0z;
    uint32_t id;
    double e1x, e1y, e1z;    // v0 - v1 (float subtraction, widened)
    double e2x, e2y, e2z;    // v0 - v2
    double m12;              // e1.y*e2.z - e2.y*e1.z
    float f1x, f1y, f1z;     // v0 - v1 and v0 - v2 as stored (floats), for the single-precision pre-filter
    float f2x, f2y, f2z;
    float w1, w2;            // fourth words of the record's second and third quarter (the halves of m12), as loaded
};

__device__ __forceinline__ TriRec unpack_rec(const float4 r0, const float4 r1, const float4 r2) {
    TriRec t;
    t.v0x = r0.x; t.v0y = r0.y; t.v0z = r0.z; t.id = __float_as_uint(r0.w);
    t.e1x = r1.x; t.e1y = r1.y; t.e1z = r1.z;
    t.e2x = r2.x; t.e2y = r2.y; t.e2z = r2.z;
    t.f1x = r1.x; t.f1y = r1.y; t.f1z = r1.z;
    t.f2x = r2.x; t.f2y = r2.y; t.f2z = r2.z;
    t.w1 = r1.w; t.w2 = r2.w;
    t.m12 = __longlong_as_double((long long)(((unsigned long long)__float_as_uint(r2.w) << 32) | __float_as_uint(r1.w)));
    return t;
}

// @phase filter
// Single-precision pre-filter of the (run, ray) trips of rays with a common origin o, two rays per instruction.
// With s = v0 - o, e1 = v0 - v1, e2 = v0 - v2 (the floats Triangle::Intersect works on) its determinants are dot products
// of the ray direction d with vectors that do not depend on the ray:
//     bnum = det[s e2 d] = d.(s x e2),   gnum = det[e1 s d] = d.(e1 x s),   A = det[e1 e2 d] = d.(e1 x e2),
// and a triangle passes (beta > 0, gamma > 0, beta + gamma < 1) only if bnum, gnum and cnum = A - bnum - gnum all have the
// sign of A. The three normals are made once per triangle and run; every ray then costs three dot products in float, and
// a triangle is dropped for a ray only when two of the three numbers have DEFINITELY opposite signs: one above +E, one
// below -E, with E a bound of everything single precision can have lost,
//     |float result - exact determinant of the float inputs| <= 8 u (|e1|1 |e2|1 + |s|1 (|e1|1 + |e2|1)),   u = 2^-24, |d_i| <= 1
// (two roundings per cross product component, two for the difference of the three normals, three per dot product; the
// double-precision test of the reference differs from the exact determinants by 2^-29 of that). E is taken at 12 u times
// the sum, plus 1e-30 for products that underflow; sums that overflow make E infinite and a NaN never compares as above
// or below, so nothing is dropped then. A ray keeps its exact trip (the double-precision test below, unchanged) whenever
// any triangle of the run survives, so results are those of the exact test alone.
typedef float rt_f2 __attribute__((ext_vector_type(2)));
// Packed single-precision instructions (two results per lane) in which the first operand is ONE element of a register
// pair, used for both results (H = 0: the low register, 1: the high one; the op_sel bits of the instruction), and the
// second operand is a pair of scalar registers: the values of two rays. This way a triangle's numbers are used where
// they already are (the halves of its record as loaded, pairs of normals) without duplicating them into pairs.
template <int H>
__device__ __forceinline__ rt_f2 pk_mul_h(rt_f2 a, rt_f2 rays) {
    rt_f2 r;
    if constexpr (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "s"(rays));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "s"(rays));
    return r;
}
// a[H] * rays + c, or with NEG a[H] * rays - c
template <int H, bool NEG = false>
__device__ __forceinline__ rt_f2 pk_fma_h(rt_f2 a, rt_f2 rays, rt_f2 c) {
    rt_f2 r;
    if constexpr (H == 0 && !NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "s"(rays), "v"(c));
    else if constexpr (H == 1 && !NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "s"(rays), "v"(c));
    else if constexpr (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "s"(rays), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "s"(rays), "v"(c));
    return r;
}
// a[H] - rays
template <int H>
__device__ __forceinline__ rt_f2 pk_sub_h(rt_f2 a, rt_f2 rays) {
    rt_f2 r;
    if constexpr (H == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "s"(rays));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "s"(rays));
    return r;
}
// lanes in which the three numbers of ray 0 (.x) and of ray 1 (.y) do not have definitely mixed signs: the largest above +E
// and the smallest below -E is min(max3, -min3) > E (an ordered compare: a NaN keeps the lane)
__device__ __forceinline__ void prefilter_keep(rt_f2 b, rt_f2 g, rt_f2 c, float E, unsigned long long& keep_i, unsigned long long& keep_j) {
    const float mi = __builtin_fminf(__builtin_fmaxf(__builtin_fmaxf(b.x, g.x), c.x), -__builtin_fminf(__builtin_fminf(b.x, g.x), c.x));
    const float mj = __builtin_fminf(__builtin_fmaxf(__builtin_fmaxf(b.y, g.y), c.y), -__builtin_fminf(__builtin_fminf(b.y, g.y), c.y));
    keep_i = ~__builtin_amdgcn_fcmpf(mi, E, 2);
    keep_j = ~__builtin_amdgcn_fcmpf(mj, E, 2);
}
// The x and y components of a x b in two packed instructions, from the pairs a01 = (a.x, a.y), a2 = (a.z, anything),
// b01 = (b.x, b.y), b2 = (b.z, anything): (a.y b.z, -a.x b.z), then a.z (-b.y, b.x) added to it. Two roundings per component.
__device__ __forceinline__ rt_f2 pk_cross_xy(rt_f2 a01, rt_f2 a2, rt_f2 b01, rt_f2 b2) {
    rt_f2 t, c;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(t) : "v"(a01), "v"(b2));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]" : "=v"(c) : "v"(a2), "v"(b01), "v"(t));
    return c;
}
// Rays with a common origin: the three normals nb = s x e2, ng = e1 x s, nc = e1 x e2 - nb - ng and E, as five register pairs.
struct PreFilter {
    rt_f2 n0, n1, n2, n3, n4;      // (nb.x, nb.y) (ng.x, ng.y) (nc.x, nc.y) (nb.z, ng.z) (nc.z, E)
};
__device__ __forceinline__ PreFilter prefilter_setup(const TriRec& t, float ox, float oy, float oz) {
    PreFilter f;
    // the record's halves as the pairs they were loaded as; s the same way (the second half of s2 is not a number anybody reads)
    const rt_f2 e1a = {t.f1x, t.f1y}, e1b = {t.f1z, t.w1}, e2a = {t.f2x, t.f2y}, e2b = {t.f2z, t.w2};
    const rt_f2 s01 = (rt_f2){t.v0x, t.v0y} - (rt_f2){ox, oy}, s2 = (rt_f2){t.v0z, __uint_as_float(t.id)} - (rt_f2){oz, 0.f};
    const float sx = s01.x, sy = s01.y, sz = s2.x;
    f.n0 = pk_cross_xy(s01, s2, e2a, e2b);
    f.n1 = pk_cross_xy(e1a, e1b, s01, s2);
    const rt_f2 na = pk_cross_xy(e1a, e1b, e2a, e2b);
    f.n2 = (na - f.n0) - f.n1;
    const float nbz = __builtin_fmaf(sx, t.f2y, -(sy * t.f2x)), ngz = __builtin_fmaf(t.f1x, sy, -(t.f1y * sx)), naz = __builtin_fmaf(t.f1x, t.f2y, -(t.f1y * t.f2x));
    const float ls = fabsf(sx) + fabsf(sy) + fabsf(sz), l1 = fabsf(t.f1x) + fabsf(t.f1y) + fabsf(t.f1z), l2 = fabsf(t.f2x) + fabsf(t.f2y) + fabsf(t.f2z);
    f.n3 = (rt_f2){nbz, ngz};
    f.n4 = (rt_f2){(naz - nbz) - ngz, (l1 * l2 + ls * (l1 + l2)) * 7.2e-7f + 1e-30f};
    return f;
}
// Lanes (triangles) that rays i and j cannot be shown to miss; (rx, ry, rz) hold the two rays' directions.
__device__ __forceinline__ void prefilter_pair(const PreFilter& f, rt_f2 rx, rt_f2 ry, rt_f2 rz, unsigned long long& keep_i, unsigned long long& keep_j) {
    const rt_f2 b = pk_fma_h<0>(f.n3, rz, pk_fma_h<1>(f.n0, ry, pk_mul_h<0>(f.n0, rx)));
    const rt_f2 g = pk_fma_h<1>(f.n3, rz, pk_fma_h<1>(f.n1, ry, pk_mul_h<0>(f.n1, rx)));
    const rt_f2 c = pk_fma_h<0>(f.n4, rz, pk_fma_h<1>(f.n2, ry, pk_mul_h<0>(f.n2, rx)));
    prefilter_keep(b, g, c, f.n4.y, keep_i, keep_j);
}
// Rays with origins of their own (shadow rays): s = v0 - o differs from ray to ray, so the determinants are taken as
//     bnum = s.(e2 x d),   gnum = s.(d x e1),   A = d.(e1 x e2),   cnum = A - bnum - gnum,
// two rays per instruction, straight from the halves of the triangle's record. The same bound holds with |s|1 replaced
// by |v0|1 + omax, omax = the largest |o|1 among the wavefront's rays. na = (e1 x e2).xy, nb = ((e1 x e2).z, E).
struct PreFilterOwn {
    rt_f2 v01, v2i, e1a, e1b, e2a, e2b, na, nb;
};
__device__ __forceinline__ PreFilterOwn prefilter_setup_own(const TriRec& t, float omax) {
    PreFilterOwn f;
    // the pairs are the record's halves in the registers they were loaded into
    f.v01 = (rt_f2){t.v0x, t.v0y}; f.v2i = (rt_f2){t.v0z, __uint_as_float(t.id)};
    f.e1a = (rt_f2){t.f1x, t.f1y}; f.e1b = (rt_f2){t.f1z, t.w1};
    f.e2a = (rt_f2){t.f2x, t.f2y}; f.e2b = (rt_f2){t.f2z, t.w2};
    const float nax = __builtin_fmaf(t.f1y, t.f2z, -(t.f1z * t.f2y)), nay = __builtin_fmaf(t.f1z, t.f2x, -(t.f1x * t.f2z)),
                naz = __builtin_fmaf(t.f1x, t.f2y, -(t.f1y * t.f2x));
    const float ls = (fabsf(t.v0x) + fabsf(t.v0y) + fabsf(t.v0z)) + omax, l1 = fabsf(t.f1x) + fabsf(t.f1y) + fabsf(t.f1z),
                l2 = fabsf(t.f2x) + fabsf(t.f2y) + fabsf(t.f2z);
    f.na = (rt_f2){nax, nay};
    f.nb = (rt_f2){naz, (l1 * l2 + ls * (l1 + l2)) * 7.2e-7f + 1e-30f};
    return f;
}
__device__ __forceinline__ void prefilter_pair_own(const PreFilterOwn& f, rt_f2 px, rt_f2 py, rt_f2 pz, rt_f2 rx, rt_f2 ry, rt_f2 rz,
                                                   unsigned long long& keep_i, unsigned long long& keep_j) {
    const rt_f2 sx = pk_sub_h<0>(f.v01, px), sy = pk_sub_h<1>(f.v01, py), sz = pk_sub_h<0>(f.v2i, pz);
    // q = e2 x d, r = d x e1
    const rt_f2 qx = pk_fma_h<1, true>(f.e2a, rz, pk_mul_h<0>(f.e2b, ry));       // e2.y d.z - e2.z d.y
    const rt_f2 qy = pk_fma_h<0, true>(f.e2b, rx, pk_mul_h<0>(f.e2a, rz));       // e2.z d.x - e2.x d.z
    const rt_f2 qz = pk_fma_h<0, true>(f.e2a, ry, pk_mul_h<1>(f.e2a, rx));       // e2.x d.y - e2.y d.x
    const rt_f2 ux = pk_fma_h<0, true>(f.e1b, ry, pk_mul_h<1>(f.e1a, rz));       // d.y e1.z - d.z e1.y
    const rt_f2 uy = pk_fma_h<0, true>(f.e1a, rz, pk_mul_h<0>(f.e1b, rx));       // d.z e1.x - d.x e1.z
    const rt_f2 uz = pk_fma_h<1, true>(f.e1a, rx, pk_mul_h<0>(f.e1a, ry));       // d.x e1.y - d.y e1.x
    const rt_f2 b = __builtin_elementwise_fma(sz, qz, __builtin_elementwise_fma(sy, qy, sx * qx));
    const rt_f2 g = __builtin_elementwise_fma(sz, uz, __builtin_elementwise_fma(sy, uy, sx * ux));
    const rt_f2 a = pk_fma_h<0>(f.nb, rz, pk_fma_h<1>(f.na, ry, pk_mul_h<0>(f.na, rx)));
    const rt_f2 c = (a - b) - g;
    prefilter_keep(b, g, c, f.nb.y, keep_i, keep_j);
}

// @phase other
// primary rays resolve their single-candidate hit trips together at the end of the round (wave_walk); 0: on the spot
#ifndef RT_ROUND_DEFER
#define RT_ROUND_DEFER 1
#endif
// kernel variant DEFER: runs tested by at least this many rays resolve single-candidate trips together after the ray loop
#ifndef RT_DEFER_MIN_RAYS
#define RT_DEFER_MIN_RAYS 16
#endif
// the empty-voxel loop of the walk as hand-scheduled instructions (step_through_empty); 0: the loop the compiler makes of dda_advance
#ifndef RT_STEP_BY_HAND
#define RT_STEP_BY_HAND 1
#endif
// single-precision pre-filter in front of the exact (run, ray) trips of primary rays; 0 switches it off
#ifndef RT_PREFILTER
#define RT_PREFILTER 1
#endif
// the same for rays with origins of their own (shadow rays): removes four of five of their exact trips and 3 % of the
// kernel's vector instructions (46 per pair of rays instead of 47 per ray) and adds as many scalar ones. With frames in
// flight the GPU is bound by vector issue, and the frame rate follows the vector count (1.5 %); one frame at a time
// nothing changes.
#ifndef RT_PREFILTER_OWN
#define RT_PREFILTER_OWN 1
#endif
// voxels of up to this many records are tested apart from the streamed runs, by the lanes dealt to the rays standing in
// them, when at least RT_OWN_MIN_RAYS rays of the wavefront stand in such voxels; 0 switches the path off
#ifndef RT_OWN_MAX_RECORDS
#define RT_OWN_MAX_RECORDS 8
#endif
// the same limit for units of 32 rays or fewer (never more than 32 such rays: at least two lanes each)
#ifndef RT_OWN_MAX_RECORDS_SHARED
#define RT_OWN_MAX_RECORDS_SHARED 24
#endif
#ifndef RT_OWN_MIN_RAYS
#define RT_OWN_MIN_RAYS 4
#endif

// GridAccel::Intersect (Serial/grid.h:167-231) for the 64 rays of a wavefront at once. Every lane must call it.
//   ANY_HIT: the shadow query. The reference ORs every voxel's result over the whole walk and ignores t,
//   so leaving at the first passing triangle returns the same value.
//   SHARED_ORIGIN: all rays start at (ox,oy,oz) of lane 0 (primary rays); the origin-dependent part of the
//   triangle test is then evaluated once per triangle instead of once per ray.
//   early (primary rays): leave once the nearest hit so far lies before the next voxel crossing.
//
// The walk goes in rounds of three phases until every ray is done.
//   STEP  every live lane advances its own DDA through empty voxels (occupancy bitmap in LDS) until it stands
//         in an occupied voxel or has left the grid.
//   CULL  crowded voxels only: which 64-record runs of its voxel can a ray's line touch at all (box and slab of
//         the run, see the comment at box_hit)?
//   TEST  the live lanes are grouped by the voxel they stand in. The round's (voxel, needed run) pairs are streamed
//         one after the other, two in flight: the run's triangle records are read ONE PER LANE (64 consecutive
//         48-byte records, a coalesced stream into registers), and the rays that need the run are broadcast one
//         after the other through scalar registers and tested against the 64 triangles at once. A record is
//         therefore loaded once per voxel visit of the work unit, not once per ray, and lane utilisation is the
//         run's fill, whatever the number of rays sharing it.
//         The reference tests a voxel's triangles in ascending index and updates (global_t, nearest triangle)
//         after each one; the float rounding of global_t between those updates makes the outcome depend on the
//         order. Here every ray collects the passing triangles of its voxel in an order-free summary (VoxelBest)
//         and applies it to its state when the voxel is done, which gives the reference's outcome whatever the
//         order of the records and of the runs.
//
// Triangle::Intersect (Serial/geometry.h:131-177): the four 3x3 determinants share their 2x2 minors over the
// y,z rows; evaluating each minor once changes no bit because negating a minor is exact and the t1 - t2 + t3
// association is kept. With P(u,v) = u.y*v.z - v.y*u.z:
//   A     = e1.x*P(e2,d) - e2.x*P(e1,d) + d.x*P(e1,e2)
//   tnum  = e1.x*P(e2,s) - e2.x*P(e1,s) + s.x*P(e1,e2)
//   bnum  =  s.x*P(e2,d) - e2.x*P(s,d)  - d.x*P(e2,s)
//   gnum  = e1.x*P(s,d)  -  s.x*P(e1,d) + d.x*P(e1,s)
// The divisions are only executed when the signs allow beta > 0 and gamma > 0 and the numerators allow
// beta + gamma < 1 with a 1e-12 relative margin; inside that margin the reference's own expression decides.
// RULE selects the hit rule. 0: Triangle::Intersect of the Serial tree as described above. 1: the one of the Parallel tree
// (Parallel/geometry.cuh:123-170): a passing triangle counts only if eps < t < t_min, with t_min a double that is never
// rounded, so the nearest hit is the plain minimum (ties: earlier voxel, then lower index) and a shadow query is true only
// for passing triangles beyond eps. With RULE 1 `td` carries the nearest distance; st.t is kept at td rounded up to float
// for the early-exit comparison (never stops a ray the exact comparison would not stop), st.id is the nearest triangle.
// IDX says what the voxel lists hold, where the kernel is built for one answer: 0 records, 1 triangle indices, 2 either
// (RtSceneDev::indexed decides at run time).
// DEFER: the kernel variant for frames in which most trips find a triangle (see process()).
template <bool ANY_HIT, bool SHARED_ORIGIN, bool COUNT, bool OWN, int RULE = 0, int IDX = 2, bool DEFER = false>
__device__ __forceinline__ bool wave_walk(const RtSceneDev& s, const uint32_t* bitmap, int lane, bool active, float ox,
                                          float oy, float oz, float dx, float dy, float dz, float mint, bool use_eps, float eps,
                                          bool early, float slack, uint32_t cull_min, int ray_lanes, Nearest& st, WalkCount& wc,
                                          double* td = nullptr) {
    static_assert(RULE == 0 || !OWN, "the Parallel tree's hit rule is only built into the streamed test path");
    // @phase walk_setup
    Dda w;
    bool alive = false;
    if (__ballot(active)) {
        if (COUNT) wc.loops[RT_LOOP_WALKS] += 1;
        const GridU g = load_grid(s.grid);          // only needed for the set-up of the walk
        if (active) {
        float rayT;
        alive = grid_entry(g, ox, oy, oz, dx, dy, dz, mint, rayT);
        if (alive) {
            if (COUNT) { wc.entered = 1; wc.steps = 1; }
            dda_setup(w, g, ox, oy, oz, dx, dy, dz, rayT);
        }
        }
    }
    bool hit = false;
    bool first = true;
    // largest |o.x| + |o.y| + |o.z| among the wavefront's rays (error bound of the pre-filter for rays with origins of their own)
    float omax = active ? fabsf(ox) + fabsf(oy) + fabsf(oz) : 0.f;
    if (!SHARED_ORIGIN && !DEFER && RT_PREFILTER_OWN) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) omax = fmaxf(omax, __shfl_xor(omax, off));
    }
    const double ddx = dx, ddy = dy, ddz = dz;
    // reciprocals for the per-run box test (1 ulp: the test's margins allow for it; a zero component gives inf like a division)
    const float ifx = __builtin_amdgcn_rcpf(dx), ify = __builtin_amdgcn_rcpf(dy), ifz = __builtin_amdgcn_rcpf(dz);
    // @phase step
    for (;;) {
        // ---- STEP: on to the next occupied voxel
        const long long t0 = COUNT ? (long long)__builtin_amdgcn_s_memtime() : 0;
        if (COUNT && __ballot(alive && !first)) wc.loops[RT_LOOP_ADVANCE] += 1;
        const float lim = st.t + fabsf(st.t) * slack;
        const unsigned long long no_early = __ballot(!(early && hit));
        if (COUNT || !RT_STEP_BY_HAND) {
            if (alive && !first) {
                alive = dda_advance(w, early && hit, st.t, slack);
                if (COUNT && alive) ++wc.steps;
            }
        } else {
            const unsigned long long adv = __ballot(alive && !first);
            if (adv) alive = alive && !__builtin_amdgcn_inverse_ballot_w64(step_once(w, lim, no_early, adv));
        }
        // at most RT_STEPS_PER_ROUND empty voxels per round: a ray on a long empty stretch goes on in the next round
        // instead of keeping the others waiting (it takes part in this round with an empty voxel)
        int steps_left = RT_STEPS_PER_ROUND;
        if (COUNT) {          // iterations the wavefront makes: those of the lane that steps the most
            uint32_t it = 0;
            Dda c = w;
            bool a2 = alive;
            int sl = RT_STEPS_PER_ROUND;
            while (a2 && !((bitmap[c.cell >> 5] >> (c.cell & 31)) & 1u) && sl > 0) { a2 = dda_advance(c, early && hit, st.t, slack); --sl; ++it; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) it = max(it, (uint32_t)__shfl_xor((int)it, off));
            wc.loops[RT_LOOP_STEP_ITER] += it;
        }
        if (COUNT || !RT_STEP_BY_HAND) {
            while (alive && !((bitmap[w.cell >> 5] >> (w.cell & 31)) & 1u) && steps_left > 0) {
                alive = dda_advance(w, early && hit, st.t, slack);
                if (COUNT && alive) ++wc.steps;
                --steps_left;
            }
        } else {
            const unsigned long long moving = __ballot(alive);
            if ((moving & ~no_early) == 0ull) step_through_empty_no_hit(w, bitmap, moving);
            else alive = alive && !__builtin_amdgcn_inverse_ballot_w64(step_through_empty(w, bitmap, lim, no_early, moving));
        }
        if (!__ballot(alive)) break;
        bool at_voxel = alive && ((bitmap[w.cell >> 5] >> (w.cell & 31)) & 1u);
        uint32_t cbeg = 0, cend = 0;
        if (at_voxel) { cbeg = s.cell_start[w.cell]; cend = s.cell_start[w.cell + 1]; }
        // A cell whose bit is set and that holds nothing is a border cell: the ray has left the grid (the loop the compiler
        // makes of dda_advance never gets there, it counts the voxels down).
        if (!COUNT && RT_STEP_BY_HAND && at_voxel && cend == cbeg) { alive = false; at_voxel = false; }
        first = !at_voxel;                               // a ray still under way has not looked at its voxel yet
        unsigned long long un = __ballot(at_voxel);
        const long long t1 = COUNT ? (long long)__builtin_amdgcn_s_memtime() : 0;
        if (COUNT) { wc.rounds += 1; wc.cyc_step += (uint32_t)(t1 - t0); if (alive) wc.tests += cend - cbeg; }
        // most rounds of a unit only move a few rays on through empty space: nothing to cull or to test
        if (!un) continue;
        if (COUNT) wc.loops[RT_LOOP_ROUNDS_AT_VOXEL] += 1;
        // @phase cull
        // ---- CULL (crowded voxels only): every ray against the boxes of the 64-record runs of ITS voxel, lane = ray.
        // A voxel's records are visited in runs of 64. Crowded voxels use runs aligned to multiples of 64 in the record
        // array, which is what the per-run boxes describe; a run is skipped for a ray when the ray's whole line misses
        // the run's box grown by a margin far above the rounding of this test and of the reference's double-precision
        // triangle test (NaNs keep the run). Small voxels start at their first record and usually fit one run.
        // need0 bit k = this ray has to test run k of its voxel; runs from 64 on (voxels of more than 4096 records) are always
        // tested: a second word of bits costs the item generator, which every run of every voxel goes through, more than the
        // culling of such voxels' far runs could save.
        const bool cull = alive && (cend - cbeg) >= (cull_min & 0xffffu);
        const uint32_t org = cull ? (cbeg & ~63u) : cbeg;
        const uint32_t nrun_me = (cend - org + 63u) >> 6;
        unsigned long long need0 = ~0ull, any0 = ~0ull;
        VoxelBest vb;                                    // passing triangles of this round's voxel (primary rays)
        vb_reset(vb);
        double pb_t = INFINITY;                          // RULE 1: nearest passing triangle beyond eps of this round's voxel
        int pb_id = -1;
        // FP32 test of a ray's whole line against a run's box and slab (k_block_bounds): the parameter interval in which
        // the line is inside the box must overlap the one in which it is between the slab's planes. Box faces and slab
        // planes are moved outwards by mg = 4e-6 (8e-6 for the slab) * (largest |coordinate| of the box + |origin|), more
        // than 20 times the rounding error of the expressions below and of the stored bounds, so a line that touches a
        // triangle of the run is always kept; fminf/fmaxf drop NaNs (0 * inf, inf - inf), which keeps the run as well, and
        // so does a line parallel to the slab whose origin lies inside it (-inf .. +inf).
        // A face at q is crossed at (q - o) / d, evaluated as fma(q, 1/d, -(o * 1/d)) with o * 1/d made once per ray:
        // with 1/d good to one unit in the last place this rounds within 2^-22 * (|q| + |o|) / |d| of the exact value, a fifteenth of
        // what mg moves the face by.
        // roa = |o.x| + |o.y| + |o.z|; (rpx, rpy, rpz) = o * (1/d) per component.
        auto box_hit = [](const float4 q0, const float4 q1, const float4 q2, float rox, float roy, float roz, float roa, float rpx, float rpy, float rpz,
                          float rix, float riy, float riz, float rdx, float rdy, float rdz) -> bool {
            const float mg = __builtin_fmaf(4e-6f, q0.w + roa, 1e-30f);
            const float ax0 = __builtin_fmaf(q0.x - mg, rix, -rpx), ax1 = __builtin_fmaf(q1.x + mg, rix, -rpx);
            const float ay0 = __builtin_fmaf(q0.y - mg, riy, -rpy), ay1 = __builtin_fmaf(q1.y + mg, riy, -rpy);
            const float az0 = __builtin_fmaf(q0.z - mg, riz, -rpz), az1 = __builtin_fmaf(q1.z + mg, riz, -rpz);
            const float n_o = __builtin_fmaf(q2.z, roz, __builtin_fmaf(q2.y, roy, q2.x * rox));
            const float n_d = __builtin_fmaf(q2.z, rdz, __builtin_fmaf(q2.y, rdy, q2.x * rdx));
            const float inv = __builtin_amdgcn_rcpf(n_d);      // 1 ulp is far inside the margins; 0 gives inf like a division
            const float mg2 = mg + mg;
            const float s0 = ((q1.w - mg2) - n_o) * inv, s1 = ((q2.w + mg2) - n_o) * inv;
            const float t0 = fmaxf(fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fminf(az0, az1)), fminf(s0, s1));
            const float t1 = fminf(fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1)), fmaxf(s0, s1));
            return !(t0 > t1);
        };
        // what box_hit needs of a ray beside its origin and direction (a handful of operations per round, lane = ray)
        const float oabs = fabsf(ox) + fabsf(oy) + fabsf(oz), opx = ox * ifx, opy = oy * ify, opz = oz * ifz;
        // (a) voxels with up to RT_CULL_LANE_RUNS runs: lane = ray, a short loop over the runs of its own voxel,
        //     all voxels of the round at once
        if (COUNT) {
            const bool ca = cull && nrun_me <= (cull_min >> 16);
            if (__ballot(ca)) {
                uint32_t mx = ca ? nrun_me : 0u, sm = mx;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, off)); sm += (uint32_t)__shfl_xor((int)sm, off); }
                wc.loops[RT_LOOP_CULL_A_ROUNDS] += 1; wc.loops[RT_LOOP_CULL_A_ITER] += mx; wc.loops[RT_LOOP_CULL_A_LANES] += sm;
            }
        }
        //     A round usually has a dozen such rays with half a dozen runs each: with lane = ray three quarters of the
        //     wavefront would idle through every iteration. So the M rays are numbered and the 64 lanes dealt among them,
        //     G = 64 >> ceil(log2 M) lanes per ray (as in the small-voxel path below): lane g * 2^lg + j tests runs g, g + G,
        //     ... of ray j's voxel, the G partial masks are or-ed together and handed back to the ray's own lane.
        {
            const bool ca = cull && nrun_me <= (cull_min >> 16);
            const unsigned long long cam = __ballot(ca);
            const int M = __popcll(cam);
            if (M > 32) {
                if (ca) {
                    const float4* bb = s.blk_bounds + 3ull * (org >> 6);
                    need0 = 0;
#pragma unroll 2
                    for (uint32_t k = 0; k < nrun_me; ++k)
                        if (box_hit(bb[3 * k], bb[3 * k + 1], bb[3 * k + 2], ox, oy, oz, oabs, opx, opy, opz, ifx, ify, ifz, dx, dy, dz)) need0 |= 1ull << k;
                }
            } else if (M > 0) {
                const int lg = M <= 2 ? 1 : (M <= 4 ? 2 : (M <= 8 ? 3 : (M <= 16 ? 4 : 5)));
                const uint32_t G = 64u >> lg;
                const unsigned long long below = (1ull << lane) - 1ull;
                // place of this lane when the culling rays come first: a permutation of the 64 lanes
                const int place = ca ? __popcll(cam & below) : M + __popcll(~cam & below);
                const int by_place = __builtin_amdgcn_ds_permute(place << 2, lane);     // lane `place` receives this lane's number
                const int j = lane & ((1 << lg) - 1);
                const uint32_t g = (uint32_t)lane >> lg;
                const int r = __shfl(by_place, j);                                      // the lane that holds ray j
                const bool mine = j < M;
                // (every lane takes part in every exchange: a lane that sat one out could not be read by the others)
                const uint32_t h_org = (uint32_t)__shfl((int)org, r), h_nrun_r = (uint32_t)__shfl((int)nrun_me, r);
                const uint32_t h_nrun = mine ? h_nrun_r : 0u;
                const float hox = SHARED_ORIGIN ? ox : __shfl(ox, r), hoy = SHARED_ORIGIN ? oy : __shfl(oy, r), hoz = SHARED_ORIGIN ? oz : __shfl(oz, r);
                const float hoa = SHARED_ORIGIN ? oabs : __shfl(oabs, r);
                const float hpx = __shfl(opx, r), hpy = __shfl(opy, r), hpz = __shfl(opz, r);
                const float hix = __shfl(ifx, r), hiy = __shfl(ify, r), hiz = __shfl(ifz, r);
                const float hdx = __shfl(dx, r), hdy = __shfl(dy, r), hdz = __shfl(dz, r);
                const float4* bb = s.blk_bounds + 3ull * (h_org >> 6);
                uint32_t bits = 0;
                for (uint32_t k = g; __ballot(k < h_nrun); k += G)
                    if (k < h_nrun && box_hit(bb[3 * k], bb[3 * k + 1], bb[3 * k + 2], hox, hoy, hoz, hoa, hpx, hpy, hpz, hix, hiy, hiz, hdx, hdy, hdz)) bits |= 1u << k;
                // or over the G lanes of a ray (lanes j, j + 2^lg, ...), then back to the ray's own lane from lane `place`
                for (int off = 1 << lg; off < 64; off <<= 1) bits |= (uint32_t)__shfl_xor((int)bits, off);
                const uint32_t mybits = (uint32_t)__shfl((int)bits, place);
                if (ca) need0 = (unsigned long long)mybits;
            }
        }
        // (b) more crowded voxels, one after the other: lane = run (the first 64 of them), the rays standing in the voxel
        //     are broadcast one by one
        {
            unsigned long long dn = __ballot(cull && nrun_me > (cull_min >> 16));
            while (dn) {
                const int leader = __ffsll((long long)dn) - 1;
                const uint32_t vb = (uint32_t)__builtin_amdgcn_readlane((int)cbeg, leader);
                const uint32_t vn = (uint32_t)__builtin_amdgcn_readlane((int)nrun_me, leader);
                const uint32_t vo = (uint32_t)__builtin_amdgcn_readlane((int)org, leader) >> 6;
                const unsigned long long mem = __ballot(cull && cbeg == vb);
                dn &= ~mem;
                if (COUNT) { wc.loops[RT_LOOP_CULL_B_VOXELS] += 1; wc.loops[RT_LOOP_CULL_B_RAYS] += (uint32_t)__popcll(mem); }
                float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, a2 = a0;
                const bool h0 = (uint32_t)lane < vn;
                if (h0) { const float4* q = s.blk_bounds + 3ull * (vo + lane); a0 = q[0]; a1 = q[1]; a2 = q[2]; }
                unsigned long long lm = mem, u0 = 0;
                while (lm) {
                    const int i = __ffsll((long long)lm) - 1;
                    mask_clear_bit(lm, i);
                    // with a shared origin everything of the test that depends on the origin alone is the same for all
                    // rays and leaves this loop
                    const float rox = SHARED_ORIGIN ? bcast_f32(ox, 0) : bcast_f32(ox, i), roy = SHARED_ORIGIN ? bcast_f32(oy, 0) : bcast_f32(oy, i),
                                roz = SHARED_ORIGIN ? bcast_f32(oz, 0) : bcast_f32(oz, i);
                    const float roa = SHARED_ORIGIN ? bcast_f32(oabs, 0) : bcast_f32(oabs, i);
                    const float rpx = bcast_f32(opx, i), rpy = bcast_f32(opy, i), rpz = bcast_f32(opz, i);
                    const float rix = bcast_f32(ifx, i), riy = bcast_f32(ify, i), riz = bcast_f32(ifz, i);
                    const float rdx = bcast_f32(dx, i), rdy = bcast_f32(dy, i), rdz = bcast_f32(dz, i);
                    const unsigned long long m0 = __ballot(h0 && box_hit(a0, a1, a2, rox, roy, roz, roa, rpx, rpy, rpz, rix, riy, riz, rdx, rdy, rdz));
                    if (lane == i) need0 = m0;
                    u0 |= m0;
                }
                if (lane == leader) any0 = u0;                     // runs some ray of this voxel needs, kept in its first lane
            }
        }
        // @phase own
        // ---- OWN: voxels with a handful of records, lane = ray. In a sparse scene the rays of a unit stand in dozens of
        // different voxels of a few triangles each; streamed as (voxel, run) items every one of them would cost a trip of
        // the whole wavefront for a handful of busy lanes. Here the records of every such ray's voxel are tested by the
        // lanes dealt to that ray and applied in ascending order (the reference's order), with the expressions of TEST.
        if (OWN && RT_OWN_MAX_RECORDS > 0) {
            const bool own = at_voxel && !cull && (cend - cbeg) <= (uint32_t)(ray_lanes <= 32 ? RT_OWN_MAX_RECORDS_SHARED : RT_OWN_MAX_RECORDS);
            const unsigned long long om = __ballot(own);
            if (__popcll(om) >= RT_OWN_MIN_RAYS) {
                un &= ~om;
                uint32_t own_tests = 0;
                {
                    // The M rays of the round that stand in such voxels are numbered (ray j = the j-th of them) and share
                    // the 64 lanes: lane g * 2^lg + j helps ray j and takes every G-th record of its voxel, G = 64 >> lg
                    // lanes per ray with 2^lg the power of two from M on. The passing triangles of a step are then applied
                    // by the ray's own lane in ascending record order, which is the reference's order.
                    const int M = __popcll(om);
                    const int lg = M <= 4 ? 2 : (M <= 8 ? 3 : (M <= 16 ? 4 : (M <= 32 ? 5 : 6)));
                    const int G = 64 >> lg;
                    const unsigned long long below = (1ull << lane) - 1ull;
                    // place of this lane when the rays in small voxels come first: a permutation of the 64 lanes
                    const int place = own ? __popcll(om & below) : M + __popcll(~om & below);
                    const int by_place = __builtin_amdgcn_ds_permute(place << 2, lane);     // lane `place` receives this lane's number
                    const int j = lane & ((1 << lg) - 1), g = lane >> lg;
                    const int r = __shfl(by_place, j);                                      // the lane that holds ray j
                    bool own_r = j < M;
                    const uint32_t cb = (uint32_t)__shfl((int)cbeg, r), ce = (uint32_t)__shfl((int)cend, r);
                    const float rox = SHARED_ORIGIN ? ox : __shfl(ox, r), roy = SHARED_ORIGIN ? oy : __shfl(oy, r), roz = SHARED_ORIGIN ? oz : __shfl(oz, r);
                    const double rdx = (double)__shfl(dx, r), rdy = (double)__shfl(dy, r), rdz = (double)__shfl(dz, r);
                    for (uint32_t k = 0; __ballot(own_r && cb + k < ce); k += (uint32_t)G) {
                        const uint32_t p = cb + k + (uint32_t)g;
                        int pass_id = -1;                       // record index of a passing triangle
                        double t_p = 0.0;
                        if (own_r && p < ce) {
                            const float4* rr = RT_REC_PTR(s, p);
                            const TriRec t = unpack_rec(rr[0], rr[1], rr[2]);
                            if (COUNT) ++own_tests;
                            const double sx = t.v0x - rox, sy = t.v0y - roy, sz = t.v0z - roz;       // float subtractions, then widened
                            const double p_e2s = minor_f32(t.e2y, sz, sy, t.e2z);
                            const double p_e1s = minor_f32(t.e1y, sz, sy, t.e1z);
                            const double p_e2d = minor_f32(t.e2y, rdz, rdy, t.e2z);
                            const double p_e1d = minor_f32(t.e1y, rdz, rdy, t.e1z);
                            const double p_sd = minor_f32(sy, rdz, rdy, sz);
                            const double A = t.e1x * p_e2d - t.e2x * p_e1d + rdx * t.m12;
                            const double bnum = sx * p_e2d - t.e2x * p_sd - rdx * p_e2s;
                            const double gnum = t.e1x * p_sd - sx * p_e1d + rdx * p_e1s;
                            const int ha = __double2hiint(A), hb = __double2hiint(bnum), hg = __double2hiint(gnum);
                            if (((ha ^ hb) | (ha ^ hg)) >= 0 && !(fabs(bnum + gnum) > fabs(A) * (1.0 + 1e-12))) {
                                const double beta = bnum / A, gamma = gnum / A;
                                if (beta > 0.0 && gamma > 0.0 && beta + gamma < 1.0) {
                                    pass_id = (int)t.id;
                                    if (!ANY_HIT) {
                                        const double tnum = t.e1x * p_e2s - t.e2x * p_e1s + sx * t.m12;
                                        t_p = tnum / A;
                                    }
                                }
                            }
                        }
                        if (__ballot(pass_id >= 0)) {
                            bool stop = false;                  // own lane of a ray: a shadow ray has met a triangle
                            for (int gg = 0; gg < G; ++gg) {
                                const int src = ((gg << lg) + place) & 63;          // own lanes: place < M <= 2^lg
                                const int id_g = __shfl(pass_id, src);
                                if (ANY_HIT) {
                                    if (id_g >= 0 && !stop && own) {
                                        stop = true;
                                        hit = true; alive = false;
                                        if (COUNT) wc.tests -= cend - (cbeg + k + (uint32_t)gg + 1u);   // tests the reference order skips
                                    }
                                } else {
                                    const double t_g = __shfl(t_p, src);
                                    if (id_g >= 0 && own) {
                                        hit = true;
                                        if (!use_eps || t_g > (double)eps) vb_add(vb, t_g, id_g);
                                    }
                                }
                            }
                            if (ANY_HIT && __shfl((int)stop, r) != 0) own_r = false;        // its helpers stop with it
                        }
                    }
                }
                if (COUNT) {                     // tests executed: a wavefront-wide number like the streamed runs' lanes
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) own_tests += __shfl_xor(own_tests, off);
                    wc.lanes += own_tests;
                }
            }
        }
        const long long t2 = COUNT ? (long long)__builtin_amdgcn_s_memtime() : 0;
        if (COUNT) wc.cyc_cull += (uint32_t)(t2 - t1);
        // @phase itemgen
        // ---- TEST: the round's (voxel, needed run) items in order: voxels by their lowest lane, runs ascending. The
        // generator runs one item ahead of the tests so that the next run's records are in flight, also across voxels.
        uint32_t sb = 0, se = 0, sorg = 0, sextra = 0, sextra_end = 0;
        unsigned long long smem = 0, srest = 0;                 // rays of the open voxel; its runs below 64 still to visit
        struct Item { uint32_t base, b, e; unsigned long long rays; };
        auto next_item = [&](Item& it) -> bool {
            if (COUNT) wc.loops[RT_LOOP_ITEMGEN_CALLS] += 1;
            for (;;) {
                if (COUNT) wc.loops[RT_LOOP_ITEMGEN_ITER] += 1;
                uint32_t k;
                unsigned long long rays = smem;
                if (srest) {
                    // straight to the next run that some ray needs
                    k = (uint32_t)__ffsll((long long)srest) - 1u;
                    mask_clear_bit(srest, (int)k);
                    rays &= __ballot((need0 >> k) & 1ull);
                } else if (sextra < sextra_end) {
                    k = sextra++;                               // runs from 64 on (voxels of more than 4096 records): every ray
                } else {
                    if (!un) return false;
                    const int leader = __ffsll((long long)un) - 1;
                    sb = (uint32_t)__builtin_amdgcn_readlane((int)cbeg, leader);
                    se = (uint32_t)__builtin_amdgcn_readlane((int)cend, leader);
                    smem = __ballot(at_voxel && cbeg == sb);
                    un &= ~smem;
                    const bool scull = (se - sb) >= (cull_min & 0xffffu);
                    sorg = scull ? (sb & ~63u) : sb;
                    const uint32_t snrun = (se - sorg + 63u) >> 6;
                    unsigned long long sany0 = ~0ull;
                    if (scull && snrun > (cull_min >> 16)) sany0 = bcast_u64(any0, leader);
                    // voxels culled lane = ray: the runs that any ray standing in the voxel needs, so that the generator goes
                    // from needed run to needed run instead of asking the lanes about every run
                    else if (scull) sany0 = (unsigned long long)wave_or_u32((at_voxel && cbeg == sb) ? (uint32_t)need0 : 0u);
                    srest = snrun < 64u ? sany0 & ((1ull << snrun) - 1ull) : sany0;
                    sextra = 64u; sextra_end = snrun;
                    if (COUNT) { wc.slots += 1; wc.loops[RT_LOOP_ITEMGEN_VOXELS] += 1; }
                    continue;
                }
                if (ANY_HIT) rays &= __ballot(alive);           // shadow query: rays already hit need no more runs
                if (rays) { it.base = sorg + 64u * k; it.b = sb; it.e = se; it.rays = rays; return true; }
            }
        };
        // @phase load
        auto load_item = [&](const Item& it, float4& r0, float4& r1, float4& r2) {
            const uint32_t p = it.base + lane;
            const uint32_t q = (p >= it.b && p < it.e) ? p : it.b;
            const float4* r = IDX == 0 ? s.recs + 3ull * q
                            : IDX == 1 ? s.tri_rec + 3ull * reinterpret_cast<const uint32_t*>(s.recs)[q] : RT_REC_PTR(s, q);
            r0 = r[0]; r1 = r[1]; r2 = r[2];
        };
        // Two runs are kept in flight: the records of item i+2 are requested before item i is tested, so a load has the
        // time of two items' tests to arrive (slots A and B take turns). Every load below is unconditional -- a slot
        // without a next item loads the item it has again -- because loads behind a branch make the compiler merge the
        // two slots into one and wait for all outstanding loads before each test.
        // @phase item
        // Hit trips put off to the end of the round (primary rays): a trip that leaves ONE candidate triangle only notes its
        // index in the ray's lane; when the round's items are through, all rays with a note fetch their triangle's record and
        // go through the determinants, the divisions and the exact test together, lane = ray, instead of one busy lane at a
        // time in 64 (same expressions on the same values: same quotients to the bit). A ray that already holds a note, or
        // finds several candidates in a run, resolves on the spot as before.
        constexpr bool ROUND_DEFER = RT_ROUND_DEFER && SHARED_ORIGIN && !ANY_HIT && RULE == 0 && !DEFER && !COUNT;
        int note_id = -1;
        unsigned long long noted_m = 0ull;
        auto process = [&](const Item& cur, const TriRec& t) {
            const uint32_t base = cur.base, e = cur.e;
            const bool have = base + lane - cur.b < e - cur.b;           // cur.b <= base + lane < e in one unsigned compare (b <= e)
            const unsigned long long have_m = __ballot(have);
            unsigned long long lm = cur.rays;
            if (ANY_HIT) lm &= __ballot(alive);                          // rays that were hit meanwhile (shadow query) drop out
            if (COUNT) { wc.lanes += (uint32_t)__popcll(have_m) * (uint32_t)__popcll(lm); wc.items += 1; wc.trips += (uint32_t)__popcll(lm); }
            wc.fb_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wc.fb_items + 1u));      // a scalar register, not one per lane
            if (COUNT) { wc.loops[RT_LOOP_ITEMS] += 1; wc.loops[RT_LOOP_TRIPS] += (uint32_t)__popcll(lm); if (lm) wc.loops[RT_LOOP_ITEMS_WITH_RAYS] += 1; }
            // Rays with a common origin first go through the single-precision pre-filter, two at a time; only those it cannot
            // show to miss every triangle of the run take the exact trip below. Not in the variant for frames full of hits.
            if (SHARED_ORIGIN && !DEFER && RT_PREFILTER && lm) {
                const PreFilter pf = prefilter_setup(t, ox, oy, oz);
                unsigned long long todo = 0ull, rest = lm;
                while (rest) {
                    const int i = __ffsll((long long)rest) - 1;
                    mask_clear_bit(rest, i);
                    const int j = rest ? __ffsll((long long)rest) - 1 : i;
                    mask_clear_bit(rest, j);
                    const rt_f2 rx = {bcast_f32(dx, i), bcast_f32(dx, j)}, ry = {bcast_f32(dy, i), bcast_f32(dy, j)}, rz = {bcast_f32(dz, i), bcast_f32(dz, j)};
                    unsigned long long ki, kj;
                    prefilter_pair(pf, rx, ry, rz, ki, kj);
                    if (ki & have_m) mask_set_bit(todo, i);
                    if (kj & have_m) mask_set_bit(todo, j);
                }
                if (COUNT) wc.loops[RT_LOOP_FILTER_KEPT] += (uint32_t)__popcll(todo);
                lm = todo;
            }
            if (!SHARED_ORIGIN && !DEFER && RT_PREFILTER_OWN && lm) {
                const PreFilterOwn pf = prefilter_setup_own(t, omax);
                unsigned long long todo = 0ull, rest = lm;
                while (rest) {
                    const int i = __ffsll((long long)rest) - 1;
                    mask_clear_bit(rest, i);
                    const int j = rest ? __ffsll((long long)rest) - 1 : i;
                    mask_clear_bit(rest, j);
                    const rt_f2 px = {bcast_f32(ox, i), bcast_f32(ox, j)}, py = {bcast_f32(oy, i), bcast_f32(oy, j)}, pz = {bcast_f32(oz, i), bcast_f32(oz, j)};
                    const rt_f2 rx = {bcast_f32(dx, i), bcast_f32(dx, j)}, ry = {bcast_f32(dy, i), bcast_f32(dy, j)}, rz = {bcast_f32(dz, i), bcast_f32(dz, j)};
                    unsigned long long ki, kj;
                    prefilter_pair_own(pf, px, py, pz, rx, ry, rz, ki, kj);
                    if (ki & have_m) mask_set_bit(todo, i);
                    if (kj & have_m) mask_set_bit(todo, j);
                }
                if (COUNT) wc.loops[RT_LOOP_FILTER_KEPT] += (uint32_t)__popcll(todo);
                lm = todo;
            }
            if (lm) {
                // origin-dependent terms, hoisted out of the ray loop when every ray has the same origin
                double sx = 0, sy = 0, sz = 0, p_e2s = 0, p_e1s = 0;
                if (SHARED_ORIGIN) {
                    sx = t.v0x - ox; sy = t.v0y - oy; sz = t.v0z - oz;        // float subtractions, then widened
                    p_e2s = minor_f32(t.e2y, sz, sy, t.e2z);
                    p_e1s = minor_f32(t.e1y, sz, sy, t.e1z);
                }
                // DEFER: a ray with ONE candidate triangle in this run does not go through the divisions in its trip (a few
                // dozen instructions for one busy lane): the ray's lane notes the triangle's lane, and after the loop all
                // such rays finish their tests together, lane = ray (below). One or two such rays take the loop once more
                // instead, and runs with few rays do not defer at all.
                int dc = -1;
                bool at_once = !DEFER || __popcll(lm) < RT_DEFER_MIN_RAYS;
                // @phase trip
                for (;;) {
                while (lm) {
                    const int i = __ffsll((long long)lm) - 1;
                    mask_clear_bit(lm, i);
                    const double rdx = (double)bcast_f32(dx, i), rdy = (double)bcast_f32(dy, i), rdz = (double)bcast_f32(dz, i);   // widening is exact
                    if (!SHARED_ORIGIN) {
                        const float rox = bcast_f32(ox, i), roy = bcast_f32(oy, i), roz = bcast_f32(oz, i);
                        sx = t.v0x - rox; sy = t.v0y - roy; sz = t.v0z - roz;
                        p_e2s = minor_f32(t.e2y, sz, sy, t.e2z);
                        p_e1s = minor_f32(t.e1y, sz, sy, t.e1z);
                    }
                    const double p_e2d = minor_f32(t.e2y, rdz, rdy, t.e2z);
                    const double p_e1d = minor_f32(t.e1y, rdz, rdy, t.e1z);
                    const double p_sd = minor_f32(sy, rdz, rdy, sz);
                    const double A = t.e1x * p_e2d - t.e2x * p_e1d + rdx * t.m12;
                    const double bnum = sx * p_e2d - t.e2x * p_sd - rdx * p_e2s;
                    const double gnum = t.e1x * p_sd - sx * p_e1d + rdx * p_e1s;
                    // beta > 0 and gamma > 0 need A, bnum and gnum of one sign: three integer operations on the sign bits
                    // instead of six double compares (zeros and NaNs get through here and fail the exact test below)
                    // Lane masks straight from the compares (predicates as in llvm::CmpInst: 38 signed >, 13 unordered or <=,
                    // 2 ordered >, 4 ordered <). beta > 0 and gamma > 0 need A, bnum and gnum of one sign: one three-input
                    // bit operation on the upper halves, (a ^ b) | (a ^ c), sign bit clear (zeros and NaNs get through here
                    // and fail the exact test below).
                    uint32_t sgn;
                    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x7e" : "=v"(sgn) : "v"(__double2hiint(A)), "v"(__double2hiint(bnum)), "v"(__double2hiint(gnum)));
                    const unsigned long long cand = have_m & __builtin_amdgcn_sicmp((int)sgn, -1, 38) &
                                                    __builtin_amdgcn_fcmp(fabs(bnum + gnum), fabs(A) * (1.0 + 1e-12), 13);
                    unsigned long long pm = 0;
                    double tt = 0.0;
                    if (cand) wc.fb_hits = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wc.fb_hits + 1u));
                    if (COUNT && cand) wc.loops[RT_LOOP_TRIPS_CAND] += 1;
                    if (ROUND_DEFER && cand && (cand & (cand - 1ull)) == 0ull && !((noted_m >> i) & 1ull)) {
                        const int id_c = __builtin_amdgcn_readlane((int)t.id, __ffsll((long long)cand) - 1);
                        if (lane == i) note_id = id_c;
                        mask_set_bit(noted_m, i);
                    } else if (DEFER && !at_once && (cand & (cand - 1ull)) == 0ull) {
                        if (cand && lane == i) dc = __ffsll((long long)cand) - 1;
                    } else if (cand) {
                        // @phase resolve
                        // The three quotients have one denominator: where the exponents of all operands allow it (see
                        // div_rcp) the reciprocal part of the division is evaluated once. Any candidate lane outside the
                        // window (zeros, infinities and NaNs included) sends the wavefront down the plain path.
                        const uint32_t ea = exponent_field(A), eb = exponent_field(bnum), eg = exponent_field(gnum);
                        const uint32_t e_lo = min(min(ea, eb), eg), e_hi = max(max(ea, eb), eg);
                        const bool fast = (cand & ~(__builtin_amdgcn_uicmp(e_lo, RT_DIV_EXP_LO, 35) & __builtin_amdgcn_uicmp(e_hi, RT_DIV_EXP_HI, 37))) == 0ull;
                        double beta, gamma, rcp_a = 0.0;
                        if (fast) {
                            rcp_a = div_rcp(A);
                            beta = div_with_rcp(bnum, A, rcp_a);
                            gamma = div_with_rcp(gnum, A, rcp_a);
                        } else {
                            beta = bnum / A;
                            gamma = gnum / A;
                        }
                        pm = cand & __builtin_amdgcn_fcmp(beta, 0.0, 2) & __builtin_amdgcn_fcmp(gamma, 0.0, 2) & __builtin_amdgcn_fcmp(beta + gamma, 1.0, 4);
                        if ((!ANY_HIT || RULE == 1) && pm) {
                            const double tnum = t.e1x * p_e2s - t.e2x * p_e1s + sx * t.m12;
                            const uint32_t et = exponent_field(tnum);
                            if (fast && (pm & ~(__builtin_amdgcn_uicmp(et, RT_DIV_EXP_LO, 35) & __builtin_amdgcn_uicmp(et, RT_DIV_EXP_HI, 37))) == 0ull) {
                                tt = div_with_rcp(tnum, A, rcp_a);
                            } else {
                                tt = tnum / A;
                            }
                        }
                    }
                    if (RULE == 1 && ANY_HIT) pm &= __builtin_amdgcn_fcmp(tt, (double)eps, 2) & __builtin_amdgcn_fcmp(tt, (double)INFINITY, 4);
                    if (COUNT && pm) { wc.loops[RT_LOOP_TRIPS_PASS] += 1; wc.loops[RT_LOOP_PASS_LANES] += (uint32_t)__popcll(pm); }
                    if (pm) {
                        if (ANY_HIT) {
                            if (lane == i) {
                                hit = true; alive = false;
                                if (COUNT) wc.tests -= e - (base + (uint32_t)__ffsll((long long)pm));   // tests the reference order skips
                            }
                        } else {
                            if (RULE == 0 && lane == i) hit = true;
                            // the passing triangles go into the ray's summary of this voxel visit (any order)
                            while (pm) {
                                const int pl = __ffsll((long long)pm) - 1;
                                mask_clear_bit(pm, pl);
                                const double t_p = bcast_f64(tt, pl);
                                const int id_p = __builtin_amdgcn_readlane((int)t.id, pl);
                                if (RULE == 0) {
                                    if (lane == i && (!use_eps || t_p > (double)eps)) vb_add(vb, t_p, id_p);
                                } else if (lane == i && t_p > (double)eps && (t_p < pb_t || (t_p == pb_t && id_p < pb_id))) {
                                    pb_t = t_p; pb_id = id_p;
                                }
                            }
                        }
                    }
                }
                // @phase defer
                if (!DEFER) break;
                // the rays that noted a candidate
                const unsigned long long dm = __ballot(dc >= 0);
                if (dm == 0ull) break;
                if (__popcll(dm) < 3) {                    // one or two rays: their trips again, with the divisions
                    lm = dm; dc = -1; at_once = true;
                    wc.fb_hits = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wc.fb_hits - (uint32_t)__popcll(dm)));   // counted once
                    continue;
                }
                // lane = ray: the ray fetches its candidate's record from that lane and repeats the trip's expressions
                // on the same values (same results), then the exact test, every ray at once
                {
                    const int src = dc >= 0 ? dc : lane;
                    const float c_v0x = __shfl(t.v0x, src), c_v0y = __shfl(t.v0y, src), c_v0z = __shfl(t.v0z, src);
                    const float c_e1x = __shfl(t.e1x, src), c_e1y = __shfl(t.e1y, src), c_e1z = __shfl(t.e1z, src);
                    const float c_e2x = __shfl(t.e2x, src), c_e2y = __shfl(t.e2y, src), c_e2z = __shfl(t.e2z, src);
                    const double c_m12 = __shfl(t.m12, src);
                    const int c_id = __shfl((int)t.id, src);
                    const double c_sx = c_v0x - ox, c_sy = c_v0y - oy, c_sz = c_v0z - oz;       // float subtractions, then widened
                    const double q_e2s = minor_f32(c_e2y, c_sz, c_sy, c_e2z);
                    const double q_e1s = minor_f32(c_e1y, c_sz, c_sy, c_e1z);
                    const double q_e2d = minor_f32(c_e2y, ddz, ddy, c_e2z);
                    const double q_e1d = minor_f32(c_e1y, ddz, ddy, c_e1z);
                    const double q_sd = minor_f32(c_sy, ddz, ddy, c_sz);
                    const double A = c_e1x * q_e2d - c_e2x * q_e1d + ddx * c_m12;
                    const double bnum = c_sx * q_e2d - c_e2x * q_sd - ddx * q_e2s;
                    const double gnum = c_e1x * q_sd - c_sx * q_e1d + ddx * q_e1s;
                    const double tnum = c_e1x * q_e2s - c_e2x * q_e1s + c_sx * c_m12;
                    const uint32_t ea = exponent_field(A), eb = exponent_field(bnum), eg = exponent_field(gnum),
                                   et = (!ANY_HIT || RULE == 1) ? exponent_field(tnum) : ea;
                    const uint32_t e_lo = min(min(ea, eb), min(eg, et)), e_hi = max(max(ea, eb), max(eg, et));
                    double beta, gamma, tq;
                    if ((dm & ~(__builtin_amdgcn_uicmp(e_lo, RT_DIV_EXP_LO, 35) & __builtin_amdgcn_uicmp(e_hi, RT_DIV_EXP_HI, 37))) == 0ull) {
                        const double rcp_a = div_rcp(A);
                        beta = div_with_rcp(bnum, A, rcp_a); gamma = div_with_rcp(gnum, A, rcp_a); tq = div_with_rcp(tnum, A, rcp_a);
                    } else {
                        beta = bnum / A; gamma = gnum / A; tq = tnum / A;
                    }
                    bool pass = dc >= 0 && beta > 0.0 && gamma > 0.0 && beta + gamma < 1.0;
                    if (RULE == 1 && ANY_HIT) pass = pass && tq > (double)eps && tq < (double)INFINITY;
                    if (pass) {
                        if (ANY_HIT) {
                            hit = true; alive = false;
                            if (COUNT) wc.tests -= e - (base + (uint32_t)dc + 1u);      // tests the reference order skips
                        } else if (RULE == 0) {
                            hit = true;
                            if (!use_eps || tq > (double)eps) vb_add(vb, tq, c_id);
                        } else if (tq > (double)eps && (tq < pb_t || (tq == pb_t && c_id < pb_id))) {
                            pb_t = tq; pb_id = c_id;
                        }
                    }
                }
                break;
                }
            }
        };
        // @phase pipe
        Item itA, itB;
        float4 a0, a1, a2, b0, b1, b2;
        bool hA = next_item(itA);
        if (IDX == 1 && hA) {
            // Lists of triangle indices: a third stage in front. The indices of item i+3 are requested a turn before the
            // records of that item, so the gather never waits for an index load it has just issued. P is the item
            // whose indices are on their way; once the items run out it keeps naming the last one. Two register sets
            // here: with three in rotation (as below) a rank's share of a frame was 4-6 % slower.
            auto load_idx = [&](const Item& it) -> uint32_t {
                const uint32_t p = it.base + lane;
                return reinterpret_cast<const uint32_t*>(s.recs)[(p >= it.b && p < it.e) ? p : it.b];
            };
            auto gather = [&](uint32_t idx, float4& r0, float4& r1, float4& r2) {
                const float4* r = s.tri_rec + 3ull * idx;
                r0 = r[0]; r1 = r[1]; r2 = r[2];
            };
            gather(load_idx(itA), a0, a1, a2);
            itB = itA;
            bool hB = next_item(itB);
            gather(load_idx(itB), b0, b1, b2);
            Item P = itB;
            bool hP = hB && next_item(P);
            uint32_t idxP = load_idx(P);
            for (;;) {
                {
                    const TriRec t = unpack_rec(a0, a1, a2);
                    const Item cur = itA;
                    Item N = P;
                    const bool hN = hP && next_item(N);
                    const uint32_t idxN = load_idx(N);
                    process(cur, t);
                    gather(idxP, a0, a1, a2);          // into the registers the test has just left: nothing is copied aside
                    itA = P; hA = hP; P = N; hP = hN; idxP = idxN;
                }
                if (!hB) break;
                {
                    const TriRec t = unpack_rec(b0, b1, b2);
                    const Item cur = itB;
                    Item N = P;
                    const bool hN = hP && next_item(N);
                    const uint32_t idxN = load_idx(N);
                    process(cur, t);
                    gather(idxP, b0, b1, b2);
                    itB = P; hB = hP; P = N; hP = hN; idxP = idxN;
                }
                if (!hA) break;
            }
        } else if (OWN && hA) {
            // kernel variants for sparse grids, where few voxels are streamed at all: two sets, the smaller code
            // (measured: the three-set loop below costs these frames 2-7 %)
            load_item(itA, a0, a1, a2);
            itB = itA;
            bool hB = next_item(itB);
            load_item(itB, b0, b1, b2);
            for (;;) {
                {
                    const TriRec t = unpack_rec(a0, a1, a2);
                    const Item cur = itA;
                    hA = hB && next_item(itA);
                    load_item(itA, a0, a1, a2);
                    process(cur, t);
                }
                if (!hB) break;
                {
                    const TriRec t = unpack_rec(b0, b1, b2);
                    const Item cur = itB;
                    hB = hA && next_item(itB);
                    load_item(itB, b0, b1, b2);
                    process(cur, t);
                }
                if (!hA) break;
            }
        } else if (hA) {
            // Three register sets take turns: an item is tested straight out of the registers its records arrived in,
            // and only then is the set given to the item three places on, so nothing is copied aside and the loads of
            // the next two items are in flight during every test. An item is generated after the test before it, when
            // the rays that test has finished (shadow query) are known.
            Item itC;
            float4 c0, c1, c2;
            bool more = true;
            auto gen = [&](Item& it) -> bool { if (more) more = next_item(it); return more; };   // `it` is left alone at the end
            load_item(itA, a0, a1, a2);
            itB = itA;
            bool hB = gen(itB);
            load_item(itB, b0, b1, b2);
            itC = itA;
            bool hC = gen(itC);
            load_item(itC, c0, c1, c2);
            for (;;) {
                process(itA, unpack_rec(a0, a1, a2));
                if (!hB) break;
                hA = gen(itA);
                load_item(itA, a0, a1, a2);
                process(itB, unpack_rec(b0, b1, b2));
                if (!hC) break;
                hB = gen(itB);
                load_item(itB, b0, b1, b2);
                process(itC, unpack_rec(c0, c1, c2));
                if (!hA) break;
                hC = gen(itC);
                load_item(itC, c0, c1, c2);
            }
        }
        // @phase resolve
        if (ROUND_DEFER && noted_m) {
            const bool has = note_id >= 0;
            float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
            if (has) { const float4* r = s.tri_rec + 3ull * (uint32_t)note_id; r0 = r[0]; r1 = r[1]; r2 = r[2]; }
            const TriRec c = unpack_rec(r0, r1, r2);
            const double rdx = dx, rdy = dy, rdz = dz;                               // widening is exact
            const double c_sx = c.v0x - ox, c_sy = c.v0y - oy, c_sz = c.v0z - oz;       // float subtractions, then widened
            const double q_e2s = minor_f32(c.e2y, c_sz, c_sy, c.e2z);
            const double q_e1s = minor_f32(c.e1y, c_sz, c_sy, c.e1z);
            const double q_e2d = minor_f32(c.e2y, rdz, rdy, c.e2z);
            const double q_e1d = minor_f32(c.e1y, rdz, rdy, c.e1z);
            const double q_sd = minor_f32(c_sy, rdz, rdy, c_sz);
            const double A = c.e1x * q_e2d - c.e2x * q_e1d + rdx * c.m12;
            const double bnum = c_sx * q_e2d - c.e2x * q_sd - rdx * q_e2s;
            const double gnum = c.e1x * q_sd - c_sx * q_e1d + rdx * q_e1s;
            const double tnum = c.e1x * q_e2s - c.e2x * q_e1s + c_sx * c.m12;
            const uint32_t ea = exponent_field(A), eb = exponent_field(bnum), eg = exponent_field(gnum), et = exponent_field(tnum);
            const uint32_t e_lo = min(min(ea, eb), min(eg, et)), e_hi = max(max(ea, eb), max(eg, et));
            double beta, gamma, tq;
            if ((noted_m & ~(__builtin_amdgcn_uicmp(e_lo, RT_DIV_EXP_LO, 35) & __builtin_amdgcn_uicmp(e_hi, RT_DIV_EXP_HI, 37))) == 0ull) {
                const double rcp_a = div_rcp(A);
                beta = div_with_rcp(bnum, A, rcp_a); gamma = div_with_rcp(gnum, A, rcp_a); tq = div_with_rcp(tnum, A, rcp_a);
            } else {
                beta = bnum / A; gamma = gnum / A; tq = tnum / A;
            }
            if (has && beta > 0.0 && gamma > 0.0 && beta + gamma < 1.0) {
                hit = true;
                if (!use_eps || tq > (double)eps) vb_add(vb, tq, note_id);
            }
        }
        // @phase apply
        if (!ANY_HIT && RULE == 0) vb_apply(vb, st);
        if (!ANY_HIT && RULE == 1 && pb_t < *td) {       // strict: an equal distance found in an earlier voxel stays
            *td = pb_t; st.id = pb_id; hit = true;
            st.t = __double2float_ru(pb_t);
        }
        if (COUNT) wc.cyc_test += (uint32_t)((long long)__builtin_amdgcn_s_memtime() - t2);
    }
    return hit;
}

// @phase other
// pow() for exponents other than the reference's 4. Kept out of line: inlined, its two dozen polynomial constants are
// hoisted into registers for the whole persistent loop of the trace kernel.
__device__ __attribute__((noinline)) double pow_general(double x, double a) { return pow(x, a); }

// Loop counts of the counting build's frames since the last read (RT_LOOP_*: primary walks first, then shadow walks).
__device__ unsigned long long rt_loop_counts[2 * RT_LOOP_WORDS];
hipError_t rt_read_loop_counts(unsigned long long* out, int n, int reset) {
    unsigned long long host[2 * RT_LOOP_WORDS] = {};
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(rt_loop_counts), sizeof(host));
    if (e != hipSuccess) return e;
    for (int k = 0; k < n && k < 2 * RT_LOOP_WORDS; ++k) out[k] = host[k];
    if (reset) {
        unsigned long long zero[2 * RT_LOOP_WORDS] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(rt_loop_counts), zero, sizeof(zero));
    }
    return e;
}

struct RtTraceArgs {
    RtSceneDev s;
    RtCamera cam;
    RtShade sh;
    RtFrame f;
    RtOutputs o;
    uint32_t* work;
    const uint32_t* lists;        // RT_QUEUE_LISTS tile lists of list_cap entries each, in queue order
    uint32_t list_cap;
};

// fast_trace of the Parallel tree (Parallel/raytracer.cu:445-524) for the rays of one work unit: nearest hit by the
// Parallel tree's rule, Blinn-Phong with the hit mesh's material and the unnormalised normal (v2-v1)x(v0-v1), a shadow ray
// TOWARDS the light from eps + bias that halves (shadow_scale) the whole colour, and for reflective materials a mirror ray
// from the hit point, down to reflect_depth levels: colour = (colour * base) * (1 - km) + reflected * km. The recursion is
// run as a loop; the levels' terms are kept and folded from the deepest level outwards, which is the order in which the
// recursion adds them. Vector normalisation as everywhere in this file (float length, double sqrt).
// The Parallel tree cannot be compiled here (CUDA), so this mode is checked against a CPU restatement only.
template <bool COUNT>
__device__ __forceinline__ void trace_unit_parallel(const RtSceneDev& s, const RtCamera& cam, const RtShade& sh, const RtFrame& f, const uint32_t* bitmap,
                                                    int lane, bool live, float dx, float dy, float dz, int per_unit, float& out_r, float& out_g,
                                                    float& out_b, int& out_id, bool& out_hit, bool& out_shadowed) {
    constexpr int MAXD = 4;                      // primary ray + at most three reflections (REFLECT_DEPTH, Parallel/raytracer.cu:16)
    float Ar[MAXD], Ag[MAXD], Ab[MAXD], Km[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; ++k) { Ar[k] = 0.f; Ag[k] = 0.f; Ab[k] = 0.f; Km[k] = 0.f; }
    const int depth = f.shading_mode == 1 ? (sh.reflect_depth < 0 ? 0 : (sh.reflect_depth > MAXD - 1 ? MAXD - 1 : sh.reflect_depth)) : 0;
    float ox = cam.pos[0], oy = cam.pos[1], oz = cam.pos[2];
    Nearest st = {INFINITY, -1};
    double td = INFINITY;
    WalkCount wc = {};
    bool cur = wave_walk<false, true, COUNT, false, 1>(s, bitmap, lane, live, ox, oy, oz, dx, dy, dz, 0.f, true, sh.eps, f.walk_full == 0,
                                                        f.early_slack, (uint32_t)f.cull_min, per_unit, st, wc, &td);
    cur = cur && st.id >= 0;
    out_id = cur ? st.id : -1;
    out_hit = cur;
    out_shadowed = false;
#pragma nounroll
    for (int d = 0; d < MAXD; ++d) {        // one copy of the walks in the code: the levels' terms live in scratch memory
        if (d > depth || !__ballot(cur)) break;
        float colr = 0.f, colg = 0.f, colb = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, km = 0.f;
        float poix = 0.f, poiy = 0.f, poiz = 0.f, lx = 0.f, ly = 0.f, lz = 1.f, nx = 0.f, ny = 0.f, nz = 0.f;
        bool reflective = false;
        if (cur) {
            const float4 p0 = s.v0[st.id], p1 = s.v1[st.id], p2 = s.v2[st.id];
            const uint32_t mesh = __float_as_uint(p0.w);
            colr = s.mesh_color[3 * mesh]; colg = s.mesh_color[3 * mesh + 1]; colb = s.mesh_color[3 * mesh + 2];
            const float* m = s.mesh_mat + 8 * mesh;
            const float kd = m[0], ks = m[1], alpha = m[2], ka = m[3];
            km = m[4];
            reflective = m[5] != 0.f && d < depth;
            const float tn = (float)td;                              // Vec3f::scale takes a float
            poix = ox + dx * tn; poiy = oy + dy * tn; poiz = oz + dz * tn;
            float vx = -dx, vy = -dy, vz = -dz;
            normalize3(vx, vy, vz);
            lx = sh.light[0] - poix; ly = sh.light[1] - poiy; lz = sh.light[2] - poiz;
            normalize3(lx, ly, lz);
            float hx = vx + lx, hy = vy + ly, hz = vz + lz;
            normalize3(hx, hy, hz);
            // normal = (v2 - v1) x (v0 - v1), Parallel/geometry.cuh:160
            const float ax = p2.x - p1.x, ay = p2.y - p1.y, az = p2.z - p1.z;
            const float bx = p0.x - p1.x, by = p0.y - p1.y, bz = p0.z - p1.z;
            nx = ay * bz - az * by; ny = az * bx - ax * bz; nz = ax * by - ay * bx;
            const float fd = max_std(0.f, nx * lx + ny * ly + nz * lz);
            const float fs = powf(max_std(0.f, nx * hx + ny * hy + nz * hz), alpha);
            cr = ((colr * fd) * kd + (colr * fs) * ks) + colr * ka;
            cg = ((colg * fd) * kd + (colg * fs) * ks) + colg * ka;
            cb = ((colb * fd) * kd + (colb * fs) * ks) + colb * ka;
        }
        // shadow ray towards the light, Parallel/raytracer.cu:492-506 (the light's direction is l again)
        Nearest sst = {INFINITY, -1};
        const bool shadowed = wave_walk<true, false, COUNT, false, 1>(s, bitmap, lane, cur, poix, poiy, poiz, lx, ly, lz, sh.eps + sh.shadow_bias, true,
                                                                   sh.eps, true, 0.f, (uint32_t)f.cull_min, per_unit, sst, wc);
        if (d == 0) out_shadowed = shadowed;
        float rx = 0.f, ry = 0.f, rz = 1.f;
        if (cur) {
            if (shadowed) { cr = cr * sh.shadow_scale; cg = cg * sh.shadow_scale; cb = cb * sh.shadow_scale; }
            if (reflective) {
                Ar[d] = (cr * colr) * (1 - km); Ag[d] = (cg * colg) * (1 - km); Ab[d] = (cb * colb) * (1 - km);
                Km[d] = km;
                // reflect(raydir.normalize(), normal.normalize()).normalize(), Parallel/raytracer.cu:509,875-878
                float ix = dx, iy = dy, iz = dz;
                normalize3(ix, iy, iz);
                normalize3(nx, ny, nz);
                const float factor = (ix * nx + iy * ny + iz * nz) * 2;
                rx = ix - nx * factor; ry = iy - ny * factor; rz = iz - nz * factor;
                normalize3(rx, ry, rz);
            } else {
                Ar[d] = cr; Ag[d] = cg; Ab[d] = cb;
            }
        }
        const bool next = cur && reflective;
        if (d + 1 >= MAXD || d + 1 > depth) break;
        if (next) { ox = poix; oy = poiy; oz = poiz; dx = rx; dy = ry; dz = rz; }
        st.t = INFINITY; st.id = -1; td = INFINITY;
        const bool h = wave_walk<false, false, COUNT, false, 1>(s, bitmap, lane, next, ox, oy, oz, dx, dy, dz, sh.eps, true, sh.eps, f.walk_full == 0,
                                                               f.early_slack, (uint32_t)f.cull_min, per_unit, st, wc, &td);
        cur = next && h && st.id >= 0;              // a reflected ray that hits nothing contributes Vec3f(0): A stays 0
    }
    // fold: level d's colour = A[d] + (level d+1's colour) * km[d]
    float Rr = Ar[MAXD - 1], Rg = Ag[MAXD - 1], Rb = Ab[MAXD - 1];
#pragma unroll
    for (int k = MAXD - 2; k >= 0; --k) { Rr = Ar[k] + Rr * Km[k]; Rg = Ag[k] + Rg * Km[k]; Rb = Ab[k] + Rb * Km[k]; }
    out_r = Rr; out_g = Rg; out_b = Rb;
}

// First tile of part q (0..RT_QUEUE_HEADS) when a list of n tiles is cut into RT_QUEUE_HEADS contiguous parts.
__device__ __forceinline__ uint32_t queue_part(uint32_t n, uint32_t q) {
    return (n * q) / RT_QUEUE_HEADS;                  // n < 2^21 tiles (RT_MAX_TILES), q <= 8
}

// @phase kernel_setup
template <bool COUNT, bool OWN, int MODE = 0, int IDX = 2, bool DEFER = false>
// Four blocks per CU (128 VGPRs, a few of them spilled) for the streamed path of crowded grids; three (144 VGPRs, nothing
// spilled) for the variants that also hold the small-voxel loop, whose frames are bound by the longest chain of one wavefront
// rather than by the number of wavefronts (measured: 1080p frame of a 5.8 k triangle mesh 0.176 -> 0.166 ms; the streamed
// path at three blocks is 4-11 % slower).
__global__ void __launch_bounds__(RT_WAVES_PER_BLOCK * 64, OWN ? 3 : 4) k_trace(RtTraceArgs args_in_kernarg_segment) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(lds_raw);
    // The ~80 dwords of frame constants are read from the kernel-argument segment where they are needed (scalar
    // loads, cached) instead of living in scalar registers for the whole persistent loop, where they would be
    // spilled; the empty asm at the top of the loop keeps the compiler from hoisting those loads back out.
    (void)args_in_kernarg_segment;
    const RtTraceArgs* A = reinterpret_cast<const RtTraceArgs*>((const void*)__builtin_amdgcn_kernarg_segment_ptr());
    uint32_t* const work = A->work;

    if (A->o.stamps && blockIdx.x == 0 && threadIdx.x == 0) A->o.stamps[1] = __builtin_amdgcn_s_memrealtime();
    const uint32_t n_listed = work[0];
    if (n_listed == 0) {
        if (A->o.stamps && blockIdx.x == 0 && threadIdx.x == 0) A->o.stamps[2] = __builtin_amdgcn_s_memrealtime();
        return;
    }
    {
        const RtSceneDev& s = A->s;
        // occupancy bitmap of the whole grid into LDS, once per block
        bitmap_to_lds(bitmap, s.occupancy, (s.grid->n_cells + 31) >> 5);
    }

    // Work unit of a wavefront: a 2^-k part of an 8x8 tile (64, 32, 16 or 8 rays, Z-order blocks). The tiles come in three
    // cost classes (k_bucket), most expensive first, and the expensive classes are cut finer, so that the long-running
    // units start early and are spread over many wavefronts while the bulk of the frame runs as large, efficient units.
    // frame counters are summed per wavefront and added to memory once, after the last unit
    uint32_t acc_hits = 0, acc_shadowed = 0;
    unsigned long long acc_fb = 0;        // record runs streamed (low half) and trips with a candidate triangle (high half)
    unsigned long long acc_ray[6] = {0, 0, 0, 0, 0, 0}, acc_w[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, max_cyc = 0, max_rounds = 0;
    const long long wave_t0 = COUNT ? (long long)__builtin_amdgcn_s_memtime() : 0;
    uint32_t acc_loops[2 * RT_LOOP_WORDS] = {};          // counting build: loop counts of the primary and of the shadow walks
    uint32_t slot_next = 0;
    // The queue has one head per XCD, each in a cache line of its own: atomics on one line are served one after the
    // other, about 7 ns each, and with a single head for the 4096 wavefronts of a launch that queue of atomics, not the
    // rays, set the pace of frames with short units. Every list is cut into eight contiguous parts (whole tiles); the
    // wavefronts of an XCD serve their part of every list, list after list. A wavefront whose head has run past its end
    // looks at all eight heads at once (plain loads) and goes on to the next one that still has units, or leaves. A list
    // is in screen order, so the tiles an XCD works on at one time are neighbours that stream the same record runs through
    // that XCD's L2. Which XCD a wavefront runs on is read from the hardware; any value 0..7 gives the same frame.
    // The head a wavefront draws from lives in LDS, not in a register: the loop below has no scalar register to spare.
    __shared__ uint32_t wave_head[RT_WAVES_PER_BLOCK];
    if (!DEFER && lane == 0) wave_head[threadIdx.x >> 6] = ((uint32_t)__builtin_amdgcn_s_getreg(RT_GETREG_XCC_ID) & 7u) % RT_QUEUE_HEADS;
    // @phase dequeue
    for (;;) {
        asm volatile("" ::: "memory");
        const RtSceneDev& s = A->s;
        const RtCamera& cam = A->cam;
        const RtShade& sh = A->sh;
        const RtFrame& f = A->f;
        const RtOutputs& o = A->o;
        // The variant for frames full of hits keeps to one head: its loops are the shortest of registers, the code of the
        // eight heads costs them spills (measured: the 4K plane frame 6 % slower with four frames in flight), and its units
        // are long, so its atomics are far apart anyway.
        const uint32_t head = DEFER ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_head[threadIdx.x >> 6]);
        // units per list (lane k < 16 looks after list k), requested before the atomic so that both round trips overlap
        const uint32_t my_shift = (lane < RT_QUEUE_LISTS) ? work[RT_WORK_LIST_COUNTS + RT_QUEUE_LISTS + lane] : 0u;
        const uint32_t my_tiles = (lane < RT_QUEUE_LISTS) ? work[RT_WORK_LIST_COUNTS + lane] : 0u;
        if (lane == 0) slot_next = atomicAdd(&work[RT_WORK_HEADS + RT_WORK_HEAD_STRIDE * head], 1u);
        const uint32_t slot = __builtin_amdgcn_readfirstlane(slot_next);
        const uint32_t my_t0 = DEFER ? 0u : queue_part(my_tiles, head);  // this head's tiles of the list: [my_t0, my_t1)
        const uint32_t my_units = ((DEFER ? my_tiles : queue_part(my_tiles, head + 1u)) - my_t0) << my_shift;
        uint32_t my_end = my_units;
#pragma unroll
        for (int off = 1; off < RT_QUEUE_LISTS; off <<= 1) {
            const uint32_t v = __shfl_up(my_end, off);
            if (lane >= off) my_end += v;
        }
        const unsigned long long owner = __ballot(lane < RT_QUEUE_LISTS && slot < my_end);
        if (!owner) {                                       // this head is past its end
            if constexpr (DEFER) break;
            // lane = (head q, lists j and j + 8): the units of head q, summed over the eight lanes of q, against its counter
            const uint32_t q = (uint32_t)lane >> 3, j = (uint32_t)lane & 7u;
            const uint32_t ta = work[RT_WORK_LIST_COUNTS + j], tb = work[RT_WORK_LIST_COUNTS + 8u + j];
            const uint32_t sa = work[RT_WORK_LIST_COUNTS + RT_QUEUE_LISTS + j], sb = work[RT_WORK_LIST_COUNTS + RT_QUEUE_LISTS + 8u + j];
            uint32_t u = ((queue_part(ta, q + 1u) - queue_part(ta, q)) << sa) + ((queue_part(tb, q + 1u) - queue_part(tb, q)) << sb);
            u += __shfl_xor(u, 1); u += __shfl_xor(u, 2); u += __shfl_xor(u, 4);
            const uint32_t drawn = __hip_atomic_load(&work[RT_WORK_HEADS + RT_WORK_HEAD_STRIDE * (q % RT_QUEUE_HEADS)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long has = __ballot(j == 0u && q < RT_QUEUE_HEADS && drawn < u);          // bit 8 q
            uint32_t live = 0;
#pragma unroll
            for (uint32_t k = 0; k < RT_QUEUE_HEADS; ++k) live |= (uint32_t)((has >> (8u * k)) & 1ull) << k;
            // the load above comes after this wavefront's own atomic on `head` and reads the same location: it shows `head`
            // as used up, and so every head a wavefront has drawn from in vain; the bit is cleared all the same
            live &= ~(1u << head);
            if (!live) break;                               // the queue is empty
            const uint32_t ahead = ((live | (live << RT_QUEUE_HEADS)) >> (head + 1u)) & ((1u << RT_QUEUE_HEADS) - 1u);
            if (lane == 0) wave_head[threadIdx.x >> 6] = (head + 1u + (uint32_t)__builtin_ctz(ahead)) % RT_QUEUE_HEADS;
            continue;
        }
        const int li = __ffsll((long long)owner) - 1;
        const int ushift = __builtin_amdgcn_readlane((int)my_shift, li);
        const uint32_t rel = slot - (uint32_t)__builtin_amdgcn_readlane((int)(my_end - my_units), li);
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)my_t0, li);
        const int tile = (int)A->lists[(size_t)li * A->list_cap + t0 + (rel >> ushift)];
        const int sub = (int)(rel & ((1u << ushift) - 1u));
        uint32_t uslot = 0;                                 // counting build: the unit's number in the order of the lists
        if constexpr (COUNT) {
            const uint32_t all_units = my_tiles << my_shift;
            uint32_t all_end = all_units;
#pragma unroll
            for (int off = 1; off < RT_QUEUE_LISTS; off <<= 1) {
                const uint32_t v = __shfl_up(all_end, off);
                if (lane >= off) all_end += v;
            }
            uslot = (uint32_t)__builtin_amdgcn_readlane((int)(all_end - all_units), li) + (t0 << ushift) + rel;
        }
        // @phase raygen
        // lane -> pixel of the tile: the tile's 64 pixels are numbered along a Z-order curve and unit `sub` owns
        // 64 >> ushift consecutive ones (a compact 8x8, 8x4, 4x4, 4x2 ... block); the other lanes carry no ray
        const int per_unit = 64 >> ushift;
        const bool in_unit = lane < per_unit;
        const int m = sub * per_unit + (lane & (per_unit - 1));
        const int mx = (m & 1) | ((m >> 1) & 2) | ((m >> 2) & 4);
        const int my = ((m >> 1) & 1) | ((m >> 2) & 2) | ((m >> 3) & 4);
        const int tl = my * 8 + mx;
        int x, y_local, y;
        tile_pixel(f, tile, tl, x, y_local, y);
        const bool live = in_unit && (x < f.width) && (y < f.height);

        float dx = 0.f, dy = 0.f, dz = 1.f;
        if (live) primary_ray(cam, f, x, y, dx, dy, dz);

        float cr = 0.f, cg = 0.f, cb = 0.f;
        int id = -1;
        if constexpr (MODE == 1) {
            bool hit = false, shadowed = false;
            trace_unit_parallel<COUNT>(s, cam, sh, f, bitmap, lane, live, dx, dy, dz, per_unit, cr, cg, cb, id, hit, shadowed);
            acc_hits += (uint32_t)__popcll(__ballot(hit));
            acc_shadowed += (uint32_t)__popcll(__ballot(shadowed));
        } else {
        // fast_trace, Serial/raytracer.cpp:71-118
        Nearest st = {INFINITY, -1};
        WalkCount wp = {}, ws = {};
        const long long t_start = COUNT ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const unsigned long long rt_start = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const bool hit_any = wave_walk<false, true, COUNT, OWN, 0, IDX, DEFER>(s, bitmap, lane, live, cam.pos[0], cam.pos[1], cam.pos[2], dx, dy, dz, 0.f, false,
                                                       sh.eps, f.walk_full == 0, f.early_slack, (uint32_t)f.cull_min, per_unit, st, wp);
        // @phase shade
        float colr = 0.f, colg = 0.f, colb = 0.f;
        float poix = 0.f, poiy = 0.f, poiz = 0.f, sx = 0.f, sy = 0.f, sz = 1.f;
        // A triangle can pass the barycentric test with a hit distance that is NaN or +inf (overflowing coordinates): the
        // reference then shades whatever triangle an earlier pixel left in its global; here such a pixel stays background.
        const bool hit = hit_any && st.id >= 0;
        if (hit) {
            id = st.id;
            const float4 p0 = s.v0[id], p1 = s.v1[id], p2 = s.v2[id];
            const uint32_t mesh = __float_as_uint(p0.w);
            colr = s.mesh_color[3 * mesh]; colg = s.mesh_color[3 * mesh + 1]; colb = s.mesh_color[3 * mesh + 2];
            poix = cam.pos[0] + dx * st.t; poiy = cam.pos[1] + dy * st.t; poiz = cam.pos[2] + dz * st.t;
            float vx = -dx, vy = -dy, vz = -dz;
            normalize3(vx, vy, vz);
            float lx = sh.light[0] - poix, ly = sh.light[1] - poiy, lz = sh.light[2] - poiz;
            normalize3(lx, ly, lz);
            const float hx = vx + lx, hy = vy + ly, hz = vz + lz;
            // getNormalMod, Serial/geometry.h:234-240: (v0-v1) x (v2-v0), not normalised
            const float abx = p0.x - p1.x, aby = p0.y - p1.y, abz = p0.z - p1.z;
            const float cax = p2.x - p0.x, cay = p2.y - p0.y, caz = p2.z - p0.z;
            const float nx = aby * caz - abz * cay;
            const float ny = abz * cax - abx * caz;
            const float nz = abx * cay - aby * cax;
            const float ndl = nx * lx + ny * ly + nz * lz;
            const float ndh = nx * hx + ny * hy + nz * hz;
            const float fd = sh.kd * max_std(0.f, ndl);
            // pow(x, 4.0) in double: x*x is exact for a float x, so (x*x)*(x*x) is the correctly rounded x^4
            const double xs = (double)max_std(0.f, ndh);
            double pw;
            if (sh.spec_alpha == 4.0f) { const double x2 = xs * xs; pw = x2 * x2; }
            else pw = pow_general(xs, (double)sh.spec_alpha);
            const float fs = (float)((double)sh.ks * pw);
            cr = (colr * fs) * sh.light_intensity + (colr * fd) * sh.light_intensity;
            cg = (colg * fs) * sh.light_intensity + (colg * fd) * sh.light_intensity;
            cb = (colb * fs) * sh.light_intensity + (colb * fd) * sh.light_intensity;
            // shadow ray, Serial/raytracer.cpp:106-113: away from the light, starting at eps
            sx = -(sh.light[0] - poix); sy = -(sh.light[1] - poiy); sz = -(sh.light[2] - poiz);
            normalize3(sx, sy, sz);
        }
        const bool shadowed = wave_walk<true, false, COUNT, OWN, 0, IDX, DEFER>(s, bitmap, lane, hit, poix, poiy, poiz, sx, sy, sz, sh.eps, true, sh.eps,
                                                            true, 0.f, (uint32_t)f.cull_min, per_unit, st, ws);
        if (hit) {
            if (shadowed) { cr = cr * sh.shadow_scale; cg = cg * sh.shadow_scale; cb = cb * sh.shadow_scale; }
            cr = cr + colr * sh.ka; cg = cg + colg * sh.ka; cb = cb + colb * sh.ka;
        }
        {
            const unsigned long long hm = __ballot(hit), sm = __ballot(shadowed);
            acc_hits += (uint32_t)__popcll(hm);
            acc_shadowed += (uint32_t)__popcll(sm);
            acc_fb += (unsigned long long)(wp.fb_items + ws.fb_items) | ((unsigned long long)(wp.fb_hits + ws.fb_hits) << 32);
            if (COUNT) {
                const unsigned long long cyc = (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - t_start);
                acc_ray[0] += wp.entered; acc_ray[1] += wp.steps; acc_ray[2] += wp.tests;
                acc_ray[3] += ws.entered; acc_ray[4] += ws.steps; acc_ray[5] += ws.tests;
                acc_w[0] += wp.rounds + ws.rounds; acc_w[1] += wp.slots + ws.slots; acc_w[2] += wp.cyc_step + ws.cyc_step;
                acc_w[3] += wp.cyc_test + ws.cyc_test; acc_w[4] += cyc; acc_w[5] += 1;
                acc_w[6] += (unsigned long long)wp.lanes + ws.lanes; acc_w[7] += wp.items + ws.items; acc_w[8] += wp.trips + ws.trips;
                acc_w[9] += wp.cyc_cull + ws.cyc_cull;
#pragma unroll
                for (int k = 0; k < RT_LOOP_WORDS; ++k) { acc_loops[k] += wp.loops[k]; acc_loops[RT_LOOP_WORDS + k] += ws.loops[k]; }
                if (cyc > max_cyc) max_cyc = cyc;
                if (wp.rounds + ws.rounds > max_rounds) max_rounds = wp.rounds + ws.rounds;
                if (lane == 0 && o.unit_cost) {
                    o.unit_cost[uslot] = (uint32_t)cyc; atomicAdd(&o.tile_cycles[tile], (uint32_t)cyc);
                    o.unit_info[uslot] = (uint32_t)tile | ((uint32_t)ushift << 24) | ((uint32_t)li << 28);
                    // (run, ray) test iterations and record runs streamed by the unit, 18 + 14 bits, saturating
                    const uint32_t ut = wp.trips + ws.trips, ui = wp.items + ws.items;
                    const size_t n16 = (size_t)16 * ((f.width + RT_TILE - 1) / RT_TILE) * ((f.out_rows + RT_TILE - 1) / RT_TILE);
                    o.unit_info[n16 + uslot] = (ut > 0x3ffffu ? 0x3ffffu : ut) | ((ui > 0x3fffu ? 0x3fffu : ui) << 18);
                    // cycles of the three phases in units of 256, and the rounds
                    const uint32_t c_s = (wp.cyc_step + ws.cyc_step) >> 8, c_c = (wp.cyc_cull + ws.cyc_cull) >> 8, c_t = (wp.cyc_test + ws.cyc_test) >> 8;
                    o.unit_info[2 * n16 + uslot] = (c_s > 0xffffu ? 0xffffu : c_s) | ((c_c > 0xffffu ? 0xffffu : c_c) << 16);
                    // start and duration on the 100 MHz real-time clock (the shader clock is not the same on all dies):
                    // start in the low 20 bits, duration / 64 above, saturating
                    const uint32_t rt_dur = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt_start) >> 6;
                    o.unit_info[4 * n16 + uslot] = ((uint32_t)rt_start & 0xfffffu) | ((rt_dur > 0xfffu ? 0xfffu : rt_dur) << 20);
                    o.unit_info[3 * n16 + uslot] = (c_t > 0xffffffu ? 0xffffffu : c_t) | ((wp.rounds + ws.rounds > 255u ? 255u : wp.rounds + ws.rounds) << 24);
                }
            }
        }
        }
        // @phase write
        if (in_unit) write_pixel(f, o, x, y_local, cr, cg, cb, id);
    }
    // end of the kernel = the latest exit of a wavefront (the clock only goes up, so the maximum also beats whatever an
    // earlier frame left in the slot)
    {
        unsigned long long* const stamps_end = A->o.stamps;      // read again: not worth a register pair through the loop
        if (stamps_end && lane == 0) atomicMax(&stamps_end[2], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
    unsigned long long* const counters = A->o.counters;
    if (counters) {
        if (lane == 0) {
            if (acc_hits) atomicAdd(&counters[0], (unsigned long long)acc_hits);
            if (acc_shadowed) atomicAdd(&counters[1], (unsigned long long)acc_shadowed);
            if (acc_fb) atomicAdd(&counters[RT_COUNTER_FEEDBACK], acc_fb);
        }
        if (COUNT) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                unsigned long long t = acc_ray[k];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
                if (lane == 0 && t) atomicAdd(&counters[2 + k], t);
            }
            if (lane == 0) {
                atomicAdd(&counters[8], acc_w[0]);
                atomicAdd(&counters[9], acc_w[1]);
                atomicAdd(&counters[10], acc_w[2]);
                atomicAdd(&counters[15], acc_w[3]);
                atomicMax(&counters[11], max_cyc);
                atomicAdd(&counters[12], acc_w[4]);
                atomicAdd(&counters[13], acc_w[5]);
                atomicMax(&counters[14], max_rounds);
                atomicAdd(&counters[16], acc_w[6]);
                atomicAdd(&counters[17], acc_w[7]);
                atomicAdd(&counters[18], acc_w[8]);
                atomicAdd(&counters[19], acc_w[9]);
                atomicMax(&counters[21], acc_w[4]);                                           // busiest wavefront
                atomicMin(&counters[22], (unsigned long long)wave_t0);                         // first start
                atomicMax(&counters[23], (unsigned long long)__builtin_amdgcn_s_memtime());    // last end
#pragma unroll
                for (int k = 0; k < 2 * RT_LOOP_WORDS; ++k)
                    if (acc_loops[k]) atomicAdd(&rt_loop_counts[k], (unsigned long long)acc_loops[k]);
            }
        }
    }
}

static int trace_blocks(const RtFrame& f, const RtRenderWork& wk) {
    const int tiles_x = (f.width + RT_TILE - 1) / RT_TILE;
    const int tiles_y = (f.out_rows + RT_TILE - 1) / RT_TILE;
    const int tiles = tiles_x * tiles_y;
    int blocks = wk.n_cu * (f.own_path && wk.blocks_per_cu > 3 ? 3 : wk.blocks_per_cu);      // k_trace's launch bounds
    const int max_useful = (tiles + RT_WAVES_PER_BLOCK - 1) / RT_WAVES_PER_BLOCK;
    return blocks > max_useful ? max_useful : blocks;
}

// Work list of a frame: tiles that see the grid, their cost estimates and cost classes. Needs the grid's cell_start and
// occupancy, not the triangle records, so it can run beside the second half of a grid build.
hipError_t rt_launch_render_prep(RtSceneDev s, RtCamera cam, RtShade sh, RtFrame f, RtRenderWork wk, hipStream_t st) {
    const int n_cls = wk.ntx * wk.ntb;
    if (n_cls <= 0 || s.n_tris == 0) return hipSuccess;
    if (n_cls > 16384) k_classify<16><<<(n_cls + 63) / 64, 256, 0, st>>>(s, cam, f, wk.work, wk.tile_list, wk.tx0, wk.ntx, wk.tb0, wk.ntb, wk.stamp_start);
    else k_classify<4><<<(n_cls + 15) / 16, 256, 0, st>>>(s, cam, f, wk.work, wk.tile_list, wk.tx0, wk.ntx, wk.tb0, wk.ntb, wk.stamp_start);
    const int nb = (n_cls + 255) / 256;
    const float4 est_w = make_float4(wk.est_w[0], wk.est_w[1], wk.est_w[2], wk.est_w[3]);
    const size_t est_lds = (size_t)wk.bitmap_words * 4 + RT_EST_CELLS * 256 * 4;
    const int reuse = (wk.reuse_estimate && wk.cost_by_tile && wk.kept_total) ? 1 : 0;
    if (reuse) {
        // the tiles' costs of an earlier frame with this camera and these triangles stand: no estimate
    } else if (s.n_tris <= RT_WIDE_MAX_TRIS)
        k_estimate<true><<<(n_cls + 31) / 32, 256, est_lds, st>>>(s, cam, sh, f, wk.work, wk.tile_list, wk.cost, wk.cost_sum, wk.est_by_tile, (uint32_t)wk.n_tiles, est_w,
                                                                   wk.cost_by_tile);
    else
        k_estimate<false><<<(n_cls + 63) / 64, 256, est_lds, st>>>(s, cam, sh, f, wk.work, wk.tile_list, wk.cost, wk.cost_sum, wk.est_by_tile, (uint32_t)wk.n_tiles, est_w,
                                                                    wk.cost_by_tile);
    k_bucket<<<nb, 256, 0, st>>>(wk.work, wk.tile_list, wk.cost, wk.cost_sum, wk.lists, wk.list_cap, wk.thr0, wk.thr1, trace_blocks(f, wk) * RT_WAVES_PER_BLOCK,
                                (unsigned long long)f.unit_shifts, f.adapt_units, s.grid, wk.cost_by_tile, wk.kept_total, reuse);
    return hipGetLastError();
}

hipError_t rt_launch_trace(RtSceneDev s, RtCamera cam, RtShade sh, RtFrame f, RtOutputs o, RtRenderWork wk, hipStream_t st) {
    const int blocks = trace_blocks(f, wk);
    if (blocks <= 0) return hipSuccess;
    const size_t lds = rt_trace_lds_bytes(wk.bitmap_words);
    RtTraceArgs ta;
    ta.s = s; ta.cam = cam; ta.sh = sh; ta.f = f; ta.o = o; ta.work = wk.work;
    ta.lists = wk.lists; ta.list_cap = wk.list_cap;
    if (f.shading_mode == 1) {
        k_trace<false, false, 1><<<blocks, RT_WAVES_PER_BLOCK * 64, lds, st>>>(ta);
    } else if (f.count_work) {
        if (f.own_path) k_trace<true, true><<<blocks, RT_WAVES_PER_BLOCK * 64, lds, st>>>(ta);
        else k_trace<true, false><<<blocks, RT_WAVES_PER_BLOCK * 64, lds, st>>>(ta);
    } else {
        if (f.own_path) k_trace<false, true><<<blocks, RT_WAVES_PER_BLOCK * 64, lds, st>>>(ta);
        else if (f.defer && s.indexed) k_trace<false, false, 0, 1, true><<<blocks, RT_WAVES_PER_BLOCK * 64, lds, st>>>(ta);
        else if (f.defer) k_trace<false, false, 0, 0, true><<<blocks, RT_WAVES_PER_BLOCK * 64, lds, st>>>(ta);
        else if (s.indexed) k_trace<false, false, 0, 1><<<blocks, RT_WAVES_PER_BLOCK * 64, lds, st>>>(ta);
        else k_trace<false, false, 0, 0><<<blocks, RT_WAVES_PER_BLOCK * 64, lds, st>>>(ta);
    }
    return hipGetLastError();
}

size_t rt_trace_lds_bytes(int bitmap_words) {
    return (size_t)bitmap_words * 4;
}

// ------------------------------------------------------------------------------------------------
// fixed working memory of rt_render_device_noalloc()
// ------------------------------------------------------------------------------------------------
// Every GPU the library runs on holds its own copy of this array (about 32 MB) until the process ends. In host code the
// array's name is only a handle for hipGetSymbolAddress, not an address a kernel could use: the device address is looked
// up once per device, with that device current, and that pointer is what kernels and memsets get.
static __device__ __align__(256) unsigned char scratch[rt_noalloc_layout().total];

#define RT_MAX_DEVICES 64
hipError_t rt_noalloc_scratch(void** ptr) {
    static std::mutex mu;
    static void* per_device[RT_MAX_DEVICES] = {};
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= RT_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!per_device[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        per_device[dev] = p;
    }
    *ptr = per_device[dev];
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// probe for rt_streams_create(): do two streams run side by side?
// ------------------------------------------------------------------------------------------------
// One wavefront that notes the 100 MHz clock, waits `ticks` of it (bounded by an iteration count as well, so that it ends
// whatever the clock does) and notes it again. Two of these on two streams overlap in time exactly when the runtime has
// put the streams on different hardware queues.
__global__ void k_probe_wait(unsigned long long* stamps, unsigned int ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t = t0;
    for (unsigned int it = 0; it < (1u << 22) && t - t0 < ticks; ++it) {
        __builtin_amdgcn_s_sleep(32);
        t = __builtin_amdgcn_s_memrealtime();
    }
    stamps[0] = t0;
    stamps[1] = t;
}

hipError_t rt_launch_probe_wait(unsigned long long* stamps, unsigned int ticks, hipStream_t st) {
    k_probe_wait<<<1, 64, 0, st>>>(stamps, ticks);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// band de-interleave after a gather to rank 0
// ------------------------------------------------------------------------------------------------
__global__ void k_deinterleave(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int width, int height, int part_count,
                               int part_rows, int elem_bytes) {
    // one thread per 4 bytes of a destination row where possible; rows are elem_bytes*width long
    const size_t row_bytes = (size_t)width * elem_bytes;
    const int y = blockIdx.y;
    if (y >= height) return;
    const int band = y / RT_TILE, part = band % part_count, local_band = band / part_count;
    const uint8_t* srow = src + ((size_t)part * part_rows + (size_t)local_band * RT_TILE + (y % RT_TILE)) * row_bytes;
    uint8_t* drow = dst + (size_t)y * row_bytes;
    if ((row_bytes & 15) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const uint4* s16 = (const uint4*)srow;
        uint4* d16 = (uint4*)drow;
        for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes / 16; i += (size_t)gridDim.x * blockDim.x) d16[i] = s16[i];
    } else if ((row_bytes & 3) == 0) {
        const uint32_t* s4 = (const uint32_t*)srow;
        uint32_t* d4 = (uint32_t*)drow;
        for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes / 4; i += (size_t)gridDim.x * blockDim.x) d4[i] = s4[i];
    } else {
        for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += (size_t)gridDim.x * blockDim.x) drow[i] = srow[i];
    }
}

hipError_t rt_launch_deinterleave(const void* src, void* dst, int32_t width, int32_t height, int32_t part_count, int32_t part_rows,
                                  int32_t elem_bytes, hipStream_t st) {
    if (width <= 0 || height <= 0) return hipSuccess;
    const size_t row_bytes = (size_t)width * elem_bytes;
    const unsigned bx = (unsigned)((row_bytes / 16 + 255) / 256);
    dim3 grid(bx < 1 ? 1 : (bx > 8 ? 8 : bx), height);
    k_deinterleave<<<grid, 256, 0, st>>>((const uint8_t*)src, (uint8_t*)dst, width, height, part_count, part_rows, elem_bytes);
    return hipGetLastError();
}
