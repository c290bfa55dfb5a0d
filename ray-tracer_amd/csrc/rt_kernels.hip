// HIP kernels for gfx950 (MI355X, 64-lane wavefronts): scene conversion, uniform-grid build and the
// per-pixel render path. Compile with -ffp-contract=off: the reference CPU renderer rounds after every
// multiply and every add (its x86-64 build has no fused multiply-adds), and hit ids are only bit-exact
// when this code does the same.
#include "rt_device.h"

#include <limits.h>
#include <math.h>

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
__global__ void k_upload_convert(const float* __restrict__ tri9, const uint32_t* __restrict__ mesh, float4* v0, float4* v1,
                                 float4* v2, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* t = tri9 + 9ull * i;
    v0[i] = make_float4(t[0], t[1], t[2], __uint_as_float(mesh[i]));
    v1[i] = make_float4(t[3], t[4], t[5], 0.f);
    v2[i] = make_float4(t[6], t[7], t[8], 0.f);
}

hipError_t rt_launch_upload_convert(const float* d_tri9, const uint32_t* d_mesh, RtBuildDev b, hipStream_t st) {
    if (b.n_tris == 0) return hipSuccess;
    k_upload_convert<<<(b.n_tris + 255) / 256, 256, 0, st>>>(d_tri9, d_mesh, b.v0, b.v1, b.v2, b.n_tris);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// grid build, Serial/grid.h:79-153, as count / prefix-scan / scatter
// ------------------------------------------------------------------------------------------------

__global__ void k_build_init(RtBuildDev b) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t c = i; c < RT_MAX_CELLS; c += stride) b.cell_count[c] = 0;
    if (i < 3) b.bounds_enc[i] = 0xff800000u;        // running minimum starts at enc(+inf)
    else if (i < 6) b.bounds_enc[i] = 0x007fffffu;   // running maximum starts at enc(-inf)
    if (i == 6) { b.grid->n_refs = 0; b.grid->overflow = 0; }
}

// Scene bounds (Serial/grid.h:82-86): min/max over all vertices. Exact and order-independent.
__global__ void k_bounds(RtBuildDev b) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < b.n_tris; i += stride) {
        float4 a = b.v0[i], c = b.v1[i], d = b.v2[i];
        lo[0] = fminf(lo[0], fminf(a.x, fminf(c.x, d.x))); hi[0] = fmaxf(hi[0], fmaxf(a.x, fmaxf(c.x, d.x)));
        lo[1] = fminf(lo[1], fminf(a.y, fminf(c.y, d.y))); hi[1] = fmaxf(hi[1], fmaxf(a.y, fmaxf(c.y, d.y)));
        lo[2] = fminf(lo[2], fminf(a.z, fminf(c.z, d.z))); hi[2] = fmaxf(hi[2], fmaxf(a.z, fmaxf(c.z, d.z)));
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&b.bounds_enc[a], enc_f32(lo[a]));
            atomicMax(&b.bounds_enc[3 + a], enc_f32(hi[a]));
        }
    }
}

// Resolution and voxel widths, Serial/grid.h:88-107. One thread: a few dozen scalar float operations.
__global__ void k_grid_setup(RtBuildDev b) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    RtGrid* g = b.grid;
    float delta[3];
    for (int a = 0; a < 3; ++a) {
        g->lo[a] = dec_f32(b.bounds_enc[a]);
        g->hi[a] = dec_f32(b.bounds_enc[3 + a]);
        delta[a] = g->hi[a] - g->lo[a];
    }
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
        cells *= nv;
    }
    g->n_cells = cells;
}

// Voxel range of one triangle's bounding box, Serial/grid.h:121-131.
__device__ __forceinline__ void tri_voxel_range(const RtBuildDev& b, const RtGrid& g, uint32_t i, int vmin[3], int vmax[3]) {
    float4 p = b.v0[i], q = b.v1[i], r = b.v2[i];
    float lo[3] = {fminf(p.x, fminf(q.x, r.x)), fminf(p.y, fminf(q.y, r.y)), fminf(p.z, fminf(q.z, r.z))};
    float hi[3] = {fmaxf(p.x, fmaxf(q.x, r.x)), fmaxf(p.y, fmaxf(q.y, r.y)), fmaxf(p.z, fmaxf(q.z, r.z))};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        vmin[a] = pos_to_voxel(lo[a], g.lo[a], g.inv_width[a], g.n[a]);
        vmax[a] = pos_to_voxel(hi[a], g.lo[a], g.inv_width[a], g.n[a]);
    }
}

// COUNT: one thread per triangle, one atomic per overlapped voxel.
__global__ void k_count(RtBuildDev b) {
    __shared__ RtGrid g;
    if (threadIdx.x == 0) g = *b.grid;
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n_tris) return;
    int vmin[3], vmax[3];
    tri_voxel_range(b, g, i, vmin, vmax);
    const int nx = g.n[0], nxy = g.n[0] * g.n[1];
    for (int z = vmin[2]; z <= vmax[2]; ++z)
        for (int y = vmin[1]; y <= vmax[1]; ++y)
            for (int x = vmin[0]; x <= vmax[0]; ++x) atomicAdd(&b.cell_count[z * nxy + y * nx + x], 1u);
}

// SCAN part 1: per-chunk totals. A chunk is RT_SCAN_CHUNK consecutive cells, one 256-thread block each.
__global__ void k_scan_sums(RtBuildDev b) {
    __shared__ uint32_t wsum[4];
    const int n_cells = b.grid->n_cells;
    uint32_t base = blockIdx.x * RT_SCAN_CHUNK;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
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
// occupancy bitmap (1 bit per cell) the render kernel uses to skip empty voxels.
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
    for (int j = 0; j < 4; ++j) {
        uint32_t c = base + j * 256 + threadIdx.x;
        uint32_t v = (c < n_cells) ? b.cell_count[c] : 0u;
        unsigned long long occ = __ballot(v != 0);
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

// SCATTER: claims a slot in the voxel's segment by counting cell_count back down to zero.
__global__ void k_scatter(RtBuildDev b) {
    __shared__ RtGrid g;
    if (threadIdx.x == 0) g = *b.grid;
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n_tris) return;
    if (g.n_refs > b.ref_cap) { if (i == 0) b.grid->overflow = 1; return; }
    int vmin[3], vmax[3];
    tri_voxel_range(b, g, i, vmin, vmax);
    const int nx = g.n[0], nxy = g.n[0] * g.n[1];
    for (int z = vmin[2]; z <= vmax[2]; ++z)
        for (int y = vmin[1]; y <= vmax[1]; ++y)
            for (int x = vmin[0]; x <= vmax[0]; ++x) {
                uint32_t c = z * nxy + y * nx + x;
                uint32_t slot = atomicSub(&b.cell_count[c], 1u) - 1u;
                uint32_t pos = b.cell_start[c] + slot;
                b.ref_tri[pos] = i;
                b.ref_cell[pos] = c;
            }
}

// ORDER + EMIT: the reference appends triangles to a voxel in ascending index (Serial/grid.h:118-150)
// and the first-tested triangle wins ties, so every voxel's list is put back into ascending order: each
// reference counts the smaller indices in its voxel (indices are distinct) and writes its 48-byte record
// at that rank. Record = {v0.xyz, id}, {v0-v1, lo32(m)}, {v0-v2, hi32(m)} with m the double
// (e1.y*e2.z - e2.y*e1.z), the one 2x2 minor of Triangle::Intersect that does not depend on the ray.
__global__ void k_order_emit(RtBuildDev b) {
    const uint32_t n_refs = b.grid->n_refs;
    if (n_refs > b.ref_cap) return;
    uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_refs; r += stride) {
        uint32_t c = b.ref_cell[r];
        uint32_t beg = b.cell_start[c], end = b.cell_start[c + 1];
        uint32_t me = b.ref_tri[r];
        uint32_t rank = 0;
        for (uint32_t k = beg; k < end; ++k) rank += (b.ref_tri[k] < me) ? 1u : 0u;
        float4 p = b.v0[me], q = b.v1[me], s = b.v2[me];
        float e1x = p.x - q.x, e1y = p.y - q.y, e1z = p.z - q.z;
        float e2x = p.x - s.x, e2y = p.y - s.y, e2z = p.z - s.z;
        double m = (double)e1y * (double)e2z - (double)e2y * (double)e1z;
        unsigned long long mb = (unsigned long long)__double_as_longlong(m);
        float4* o = b.recs + 3ull * (beg + rank);
        o[0] = make_float4(p.x, p.y, p.z, __uint_as_float(me));
        o[1] = make_float4(e1x, e1y, e1z, __uint_as_float((uint32_t)mb));
        o[2] = make_float4(e2x, e2y, e2z, __uint_as_float((uint32_t)(mb >> 32)));
    }
}

hipError_t rt_launch_build_phase1(RtBuildDev b, hipStream_t st) {
    k_build_init<<<256, 256, 0, st>>>(b);
    if (b.n_tris) {
        uint32_t nb = (b.n_tris + 255) / 256;
        k_bounds<<<nb < 1024 ? nb : 1024, 256, 0, st>>>(b);
    }
    k_grid_setup<<<1, 64, 0, st>>>(b);
    if (b.n_tris) k_count<<<(b.n_tris + 255) / 256, 256, 0, st>>>(b);
    k_scan_sums<<<RT_MAX_SCAN_BLOCKS, 256, 0, st>>>(b);
    k_scan_apply<<<RT_MAX_SCAN_BLOCKS, 256, 0, st>>>(b);
    return hipGetLastError();
}

hipError_t rt_launch_build_phase2(RtBuildDev b, hipStream_t st) {
    if (b.n_tris == 0) return hipSuccess;
    k_scatter<<<(b.n_tris + 255) / 256, 256, 0, st>>>(b);
    k_order_emit<<<2048, 256, 0, st>>>(b);
    return hipGetLastError();
}

__global__ void k_extract_cell_tris(const float4* __restrict__ recs, uint32_t* out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __float_as_uint(recs[3ull * i].w);
}

hipError_t rt_launch_extract_cell_tris(const float4* recs, uint32_t* out, uint32_t n_refs, hipStream_t st) {
    if (n_refs == 0) return hipSuccess;
    k_extract_cell_tris<<<(n_refs + 255) / 256, 256, 0, st>>>(recs, out, n_refs);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// render: ray generation, 3D-DDA walk, ray-triangle tests, shading, shadow ray, framebuffer writes
// ------------------------------------------------------------------------------------------------

struct GridU {               // grid description held in scalar registers
    float lo[3], hi[3], width[3], inv_width[3];
    int n[3];
};

struct Nearest {             // the reference's two globals, per ray (Serial/geometry.h:10-11)
    float t;
    int id;
};

// Triangle::Intersect (Serial/geometry.h:131-177) on one 48-byte record. The four 3x3 determinants share
// their 2x2 minors over the y,z rows; evaluating each minor once changes no bit because negating a minor
// is exact and the t1 - t2 + t3 association is kept. P(u,v) = u.y*v.z - v.y*u.z.
//   A     = e1.x*P(e2,d) - e2.x*P(e1,d) + d.x*P(e1,e2)
//   tnum  = e1.x*P(e2,s) - e2.x*P(e1,s) + s.x*P(e1,e2)
//   bnum  =  s.x*P(e2,d) - e2.x*P(s,d)  - d.x*P(e2,s)
//   gnum  = e1.x*P(s,d)  -  s.x*P(e1,d) + d.x*P(e1,s)
// The divisions are only executed when the signs allow beta > 0 and gamma > 0 and the numerators allow
// beta + gamma < 1 with a 1e-12 relative margin; inside that margin the reference's own expression decides.
template <bool ANY_HIT>
__device__ __forceinline__ bool tri_test(float4 r0, float4 r1, float4 r2, float ox, float oy, float oz, double dx, double dy,
                                         double dz, bool use_eps, float eps, Nearest& st) {
    const double e1x = r1.x, e1y = r1.y, e1z = r1.z;
    const double e2x = r2.x, e2y = r2.y, e2z = r2.z;
    const double sx = r0.x - ox, sy = r0.y - oy, sz = r0.z - oz;       // float subtractions, then widened
    const double m12 = __longlong_as_double((long long)(((unsigned long long)__float_as_uint(r2.w) << 32) | __float_as_uint(r1.w)));
    const double p_e2d = e2y * dz - dy * e2z;
    const double p_e1d = e1y * dz - dy * e1z;
    const double p_sd = sy * dz - dy * sz;
    const double A = e1x * p_e2d - e2x * p_e1d + dx * m12;
    const double p_e2s = e2y * sz - sy * e2z;
    const double p_e1s = e1y * sz - sy * e1z;
    const double bnum = sx * p_e2d - e2x * p_sd - dx * p_e2s;
    const double gnum = e1x * p_sd - sx * p_e1d + dx * p_e1s;
    const bool pos = (A > 0.0) && (bnum > 0.0) && (gnum > 0.0);
    const bool neg = (A < 0.0) && (bnum < 0.0) && (gnum < 0.0);
    if (!(pos || neg)) return false;
    const double sum = bnum + gnum;
    if (fabs(sum) > fabs(A) * (1.0 + 1e-12)) return false;
    const double beta = bnum / A;
    const double gamma = gnum / A;
    if (!(beta > 0 && gamma > 0 && beta + gamma < 1)) return false;
    if (ANY_HIT) return true;
    const double tnum = e1x * p_e2s - e2x * p_e1s + sx * m12;
    const double t = tnum / A;
    if (t < (double)st.t) {
        if (!use_eps || t > (double)eps) {
            st.t = (float)t;
            st.id = (int)__float_as_uint(r0.w);
        }
    }
    return true;
}

// GridAccel::Intersect, Serial/grid.h:167-231, one ray per lane.
//   ANY_HIT: the shadow query. The reference ORs every voxel's result over the whole walk and ignores t,
//   so leaving at the first passing triangle returns the same value.
//   EARLY (primary rays): leave once the nearest hit so far lies before the next voxel crossing.
struct WalkCount {           // work counters of one ray (only maintained by the counting build of the kernel)
    uint32_t entered, steps, tests;
};

template <bool ANY_HIT, bool COUNT>
__device__ __forceinline__ bool grid_walk(const GridU& g, const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ occupancy,
                                          const float4* __restrict__ recs, float ox, float oy, float oz, float dx, float dy, float dz,
                                          float mint, bool use_eps, float eps, bool early, float slack, Nearest& st, WalkCount& wc) {
    const float o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
    // bounds.Inside(ray(mint)), Serial/geometry.h:287-289
    float rayT;
    const float px = ox + dx * mint, py = oy + dy * mint, pz = oz + dz * mint;
    const bool inside = px >= g.lo[0] && px <= g.hi[0] && py >= g.lo[1] && py <= g.hi[1] && pz >= g.lo[2] && pz <= g.hi[2];
    if (inside) {
        rayT = mint;
    } else {
        // boundingBox::Intersect, Serial/geometry.h:291-316
        float t0 = mint, t1 = INFINITY;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float inv = 1.f / d[i];
            float tn = (g.lo[i] - o[i]) * inv;
            float tf = (g.hi[i] - o[i]) * inv;
            if (tn > tf) { float x = tn; tn = tf; tf = x; }
            t0 = tn > t0 ? tn : t0;
            t1 = tf < t1 ? tf : t1;
            if (t0 > t1) return false;
        }
        rayT = t0;
    }
    if (COUNT) wc.entered = 1;
    const float gi[3] = {ox + dx * rayT, oy + dy * rayT, oz + dz * rayT};
    float next[3], delta[3];
    int pos[3], step[3], out[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        pos[a] = pos_to_voxel(gi[a], g.lo[a], g.inv_width[a], g.n[a]);
        if (d[a] >= 0) {
            next[a] = rayT + ((g.lo[a] + (float)(pos[a] + 1) * g.width[a]) - gi[a]) / d[a];
            delta[a] = g.width[a] / d[a];
            step[a] = 1; out[a] = g.n[a];
        } else {
            next[a] = rayT + ((g.lo[a] + (float)pos[a] * g.width[a]) - gi[a]) / d[a];
            delta[a] = -g.width[a] / d[a];
            step[a] = -1; out[a] = -1;
        }
    }
    const double ddx = dx, ddy = dy, ddz = dz;
    const int nx = g.n[0], nxy = g.n[0] * g.n[1];
    bool hit = false;
    for (;;) {
        const int c = pos[2] * nxy + pos[1] * nx + pos[0];
        if (COUNT) ++wc.steps;
        if ((occupancy[c >> 5] >> (c & 31)) & 1u) {
            const uint32_t beg = cell_start[c], end = cell_start[c + 1];
            for (uint32_t k = beg; k < end; ++k) {
                const float4* r = recs + 3ull * k;
                if (COUNT) ++wc.tests;
                if (tri_test<ANY_HIT>(r[0], r[1], r[2], ox, oy, oz, ddx, ddy, ddz, use_eps, eps, st)) {
                    if (ANY_HIT) return true;
                    hit = true;
                }
            }
        }
        // stepAxis through the 3-bit compare table {2,1,2,1,2,2,0,0}, Serial/grid.h:217-221
        const int bits = ((next[0] < next[1]) << 2) + ((next[0] < next[2]) << 1) + (next[1] < next[2]);
        const int ax = (0x0A66 >> (2 * bits)) & 3;     // 2-bit entries of the table, entry 0 in the low bits
        const float nx_t = ax == 0 ? next[0] : (ax == 1 ? next[1] : next[2]);
        if (early && hit) {
            const float lim = st.t + fabsf(st.t) * slack;
            if (lim < nx_t) break;
        }
        if (ax == 0) { pos[0] += step[0]; if (pos[0] == out[0]) break; next[0] += delta[0]; }
        else if (ax == 1) { pos[1] += step[1]; if (pos[1] == out[1]) break; next[1] += delta[1]; }
        else { pos[2] += step[2]; if (pos[2] == out[2]) break; next[2] += delta[2]; }
    }
    return hit;
}

template <bool COUNT>
__global__ void __launch_bounds__(RT_BLOCK_THREADS) k_render(RtSceneDev s, RtCamera cam, RtShade sh, RtFrame f, RtOutputs o) {
    // pixel of this lane: 4 wavefronts per block, one 8x8 tile each, tiles laid out left to right
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles_x = (f.width + RT_TILE - 1) / RT_TILE;
    const int tile = blockIdx.x * 4 + wv;
    const int tx = tile % tiles_x, ty_local = tile / tiles_x;                 // ty_local = band of this part
    const int x = tx * RT_TILE + (lane & 7);
    const int y_local = ty_local * RT_TILE + (lane >> 3);
    const int y = (ty_local * f.part_count + f.part_index) * RT_TILE + (lane >> 3);   // row in the full frame
    if (ty_local * RT_TILE >= f.out_rows) return;
    const bool live = (x < f.width) && (y < f.height);

    GridU g;
    {
        const RtGrid* gp = s.grid;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            g.lo[a] = gp->lo[a]; g.hi[a] = gp->hi[a]; g.width[a] = gp->width[a]; g.inv_width[a] = gp->inv_width[a];
            g.n[a] = gp->n[a];
        }
    }

    float cr = 0.f, cg = 0.f, cb = 0.f;
    int id = -1;
    bool shadowed = false;
    WalkCount wp = {0, 0, 0}, ws = {0, 0, 0};
    if (live && s.n_tris > 0) {
        // ray generation, Serial/raytracer.cpp:153-161; xw and yw are evaluated in double
        float dx = 0.f + (-cam.w[0]) * cam.focal_distance;
        float dy = 0.f + (-cam.w[1]) * cam.focal_distance;
        float dz = 0.f + (-cam.w[2]) * cam.focal_distance;
        const float xw = (float)((double)cam.aspect * ((double)x - (double)f.width / 2.0 + 0.5) / (double)f.width);
        const float yw = (float)(((double)y - (double)f.height / 2.0 + 0.5) / (double)f.height);
        dx = dx + cam.u[0] * xw; dy = dy + cam.u[1] * xw; dz = dz + cam.u[2] * xw;
        dx = dx + cam.v[0] * yw; dy = dy + cam.v[1] * yw; dz = dz + cam.v[2] * yw;
        normalize3(dx, dy, dz);

        // fast_trace, Serial/raytracer.cpp:71-118
        Nearest st = {INFINITY, -1};
        const bool hit = grid_walk<false, COUNT>(g, s.cell_start, s.occupancy, s.recs, cam.pos[0], cam.pos[1], cam.pos[2], dx, dy, dz,
                                                 0.f, false, sh.eps, f.walk_full == 0, f.early_slack, st, wp);
        if (hit) {
            id = st.id;
            const float4 p0 = s.v0[id], p1 = s.v1[id], p2 = s.v2[id];
            const uint32_t mesh = __float_as_uint(p0.w);
            const float colr = s.mesh_color[3 * mesh], colg = s.mesh_color[3 * mesh + 1], colb = s.mesh_color[3 * mesh + 2];
            const float poix = cam.pos[0] + dx * st.t, poiy = cam.pos[1] + dy * st.t, poiz = cam.pos[2] + dz * st.t;
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
            else pw = pow(xs, (double)sh.spec_alpha);
            const float fs = (float)((double)sh.ks * pw);
            cr = (colr * fs) * sh.light_intensity + (colr * fd) * sh.light_intensity;
            cg = (colg * fs) * sh.light_intensity + (colg * fd) * sh.light_intensity;
            cb = (colb * fs) * sh.light_intensity + (colb * fd) * sh.light_intensity;
            // shadow ray, Serial/raytracer.cpp:106-113: away from the light, starting at eps
            float sx = -(sh.light[0] - poix), sy = -(sh.light[1] - poiy), sz = -(sh.light[2] - poiz);
            normalize3(sx, sy, sz);
            shadowed = grid_walk<true, COUNT>(g, s.cell_start, s.occupancy, s.recs, poix, poiy, poiz, sx, sy, sz, sh.eps, true, sh.eps,
                                              true, 0.f, st, ws);
            if (shadowed) { cr = cr * sh.shadow_scale; cg = cg * sh.shadow_scale; cb = cb * sh.shadow_scale; }
            cr = cr + colr * sh.ka; cg = cg + colg * sh.ka; cb = cb + colb * sh.ka;
        }
    }
    if (o.counters) {
        const unsigned long long hm = __ballot(id >= 0), sm = __ballot(shadowed);
        if (lane == 0) {
            if (hm) atomicAdd(&o.counters[0], (unsigned long long)__popcll(hm));
            if (sm) atomicAdd(&o.counters[1], (unsigned long long)__popcll(sm));
        }
        if (COUNT) {
            uint32_t v[6] = {wp.entered, wp.steps, wp.tests, ws.entered, ws.steps, ws.tests};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                unsigned long long t = v[k];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
                if (lane == 0 && t) atomicAdd(&o.counters[2 + k], t);
            }
        }
    }
    if (x < f.width && y_local < f.out_rows) {
        const size_t pix = (size_t)y_local * f.width + x;
        if (o.rgb_f32) { o.rgb_f32[3 * pix] = cr; o.rgb_f32[3 * pix + 1] = cg; o.rgb_f32[3 * pix + 2] = cb; }
        if (o.rgb_u8) {
            // PPM quantisation, Serial/raytracer.cpp:181-183: (unsigned char)(min(1, c/255)*255); a NaN gives 255
            o.rgb_u8[3 * pix] = (uint8_t)(min_std(1.f, cr / 255) * 255);
            o.rgb_u8[3 * pix + 1] = (uint8_t)(min_std(1.f, cg / 255) * 255);
            o.rgb_u8[3 * pix + 2] = (uint8_t)(min_std(1.f, cb / 255) * 255);
        }
        if (o.hit_id) o.hit_id[pix] = id;
    }
}

hipError_t rt_launch_render(RtSceneDev s, RtCamera cam, RtShade sh, RtFrame f, RtOutputs o, hipStream_t st) {
    const int tiles_x = (f.width + RT_TILE - 1) / RT_TILE;
    const int tiles_y = (f.out_rows + RT_TILE - 1) / RT_TILE;
    const int tiles = tiles_x * tiles_y;
    if (tiles <= 0) return hipSuccess;
    if (f.count_work) k_render<true><<<(tiles + 3) / 4, RT_BLOCK_THREADS, 0, st>>>(s, cam, sh, f, o);
    else k_render<false><<<(tiles + 3) / 4, RT_BLOCK_THREADS, 0, st>>>(s, cam, sh, f, o);
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
    if ((row_bytes & 3) == 0) {
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
    dim3 grid(8, height);
    k_deinterleave<<<grid, 256, 0, st>>>((const uint8_t*)src, (uint8_t*)dst, width, height, part_count, part_rows, elem_bytes);
    return hipGetLastError();
}
