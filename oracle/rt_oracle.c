/* TEST INFRASTRUCTURE ONLY - see rt_oracle.h. Plain-C restatement of the reference's per-pixel
 * render path, re-entrant (the reference's two globals become per-ray state) and parameterised.
 * Build with -ffp-contract=off: every float/double operation below must round exactly once,
 * as in the reference's x86-64 build, which contains no fused multiply-adds.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <time.h>
#include <limits.h>

struct rto_scene {
    int64_t ntri, cap;
    float* v;        /* 9 per triangle: v0 v1 v2 */
    float* color;    /* 3 per triangle (Triangle::material, Serial/geometry.h:113) */
    float* mat;      /* 6 per triangle: kd, ks, spec_alpha, ka, km, reflective (material table of the Parallel tree) */
    /* grid */
    int built;
    int32_t n[3];
    float lo[3], hi[3], width[3], inv_width[3];
    uint32_t* cell_start;
    uint32_t* cell_tris;
    int64_t nref;
};

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

void rto_params_default(rto_params* p) {
    memset(p, 0, sizeof(*p));
    p->width = 512; p->height = 512;                       /* Serial/raytracer.cpp:128 */
    p->cam_pos[0] = 3; p->cam_pos[1] = 5; p->cam_pos[2] = 3; /* :124 */
    p->cam_up[1] = -1;                                     /* :126 */
    p->fov = 45;                                           /* :127 */
    p->light_pos[0] = 5; p->light_pos[1] = -5; p->light_pos[2] = 2; /* :87 */
    p->kd = 2.0f; p->ks = 500000000000.0f; p->ka = 0.2f;   /* :83-85 */
    p->spec_alpha = 4; p->light_intensity = 255;           /* :86,88 */
    p->shadow_scale = 0.1f;                                /* :89 */
    p->eps = 1e-1f;                                        /* Serial/geometry.h:2 */
    p->walk_mode = 0;
    p->early_slack = 0.f;
    p->reflect_depth = 3;
    p->shadow_bias = 0.02f;
}

rto_scene* rto_scene_create(void) { return (rto_scene*)calloc(1, sizeof(rto_scene)); }

void rto_scene_destroy(rto_scene* s) {
    if (!s) return;
    free(s->v); free(s->color); free(s->mat); free(s->cell_start); free(s->cell_tris); free(s);
}

static void scene_reserve(rto_scene* s, int64_t extra) {
    if (s->ntri + extra <= s->cap) return;
    int64_t nc = s->cap ? s->cap : 1024;
    while (nc < s->ntri + extra) nc *= 2;
    s->v = (float*)realloc(s->v, (size_t)nc * 9 * sizeof(float));
    s->color = (float*)realloc(s->color, (size_t)nc * 3 * sizeof(float));
    s->mat = (float*)realloc(s->mat, (size_t)nc * 6 * sizeof(float));
    s->cap = nc;
}

/* entry 1 of the Parallel tree's table (Parallel/raytracer.cu:451) around a colour */
static rto_material default_material(const float color[3]) {
    rto_material m;
    memcpy(m.base_color, color, 3 * sizeof(float));
    m.kd = 10.0f; m.ks = 10.0f; m.spec_alpha = 1.25f; m.ka = 0.3f; m.km = 0.f; m.reflective = 0;
    return m;
}
static void set_material(rto_scene* s, int64_t i, const rto_material* m) {
    memcpy(s->color + 3 * i, m->base_color, 3 * sizeof(float));
    float* q = s->mat + 6 * i;
    q[0] = m->kd; q[1] = m->ks; q[2] = m->spec_alpha; q[3] = m->ka; q[4] = m->km; q[5] = m->reflective ? 1.f : 0.f;
}

int64_t rto_scene_add_triangles_material(rto_scene* s, const float* v9, int64_t n, const rto_material* m) {
    scene_reserve(s, n);
    memcpy(s->v + 9 * s->ntri, v9, (size_t)n * 9 * sizeof(float));
    for (int64_t i = 0; i < n; ++i) set_material(s, s->ntri + i, m);
    s->ntri += n;
    s->built = 0;
    return n;
}

int64_t rto_scene_add_triangles(rto_scene* s, const float* v9, int64_t n, const float color[3]) {
    rto_material m = default_material(color);
    return rto_scene_add_triangles_material(s, v9, n, &m);
}

int64_t rto_scene_num_triangles(const rto_scene* s) { return s->ntri; }
const float* rto_scene_triangles(const rto_scene* s) { return s->v; }

/* ---- OBJ reader: what load_mesh's getline / operator>> calls do (Serial/raytracer.cpp:229-284) ---- */

static const char* skip_ws(const char* p, const char* e) { while (p < e && isspace((unsigned char)*p)) ++p; return p; }
static const char* skip_tok(const char* p, const char* e) { while (p < e && !isspace((unsigned char)*p)) ++p; return p; }

/* operator>>(double): leading whitespace skipped, 0 stored when nothing converts. */
static double read_double(const char** pp, const char* e, int* ok) {
    const char* p = skip_ws(*pp, e);
    if (!*ok || p >= e) { *ok = 0; return 0.0; }
    char* end = NULL;
    double d = strtod(p, &end);          /* the line buffer is NUL-terminated at e */
    if (end == p) { *ok = 0; return 0.0; }
    *pp = end;
    return d;
}

/* operator>>(int) on a token: optional sign, then digits; 0 when there are none. */
static int read_int(const char** pp, const char* e) {
    const char* p = *pp;
    int neg = 0;
    long long v = 0;
    if (p < e && (*p == '+' || *p == '-')) { neg = (*p == '-'); ++p; }
    const char* d0 = p;
    while (p < e && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); if (v > INT_MAX) v = INT_MAX; ++p; }
    if (p == d0) return 0;
    *pp = p;
    return (int)(neg ? -v : v);
}

int64_t rto_scene_add_obj(rto_scene* s, const char* path, int format_has_vt, const float color[3], const float offset[3]) {
    rto_material m = default_material(color);
    return rto_scene_add_obj_material(s, path, format_has_vt, &m, offset, 1);
}

int64_t rto_scene_add_obj_material(rto_scene* s, const char* path, int format_has_vt, const rto_material* mat, const float offset[3], int scale_i) {
    const double scale = (double)scale_i;   /* Parallel/raytracer.cu:824: int * double; 1 for the Serial loader, where the product is exact */
    FILE* f = fopen(path, "rb");
    if (!f) return 0;                    /* a missing file gives an empty mesh, no error */
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* buf = (char*)malloc((size_t)sz + 1);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return 0; }
    fclose(f);
    buf[sz] = 0;

    int64_t nv = 0, vcap = 1024, added = 0;
    float* verts = (float*)malloc((size_t)vcap * 3 * sizeof(float));
    int64_t rc = 0;
    char* line = buf;
    char* bend = buf + sz;
    while (line < bend) {
        char* nl = (char*)memchr(line, '\n', (size_t)(bend - line));
        char* le = nl ? nl : bend;
        char saved = *le;
        *le = 0;
        const char* p = skip_ws(line, le);
        const char* t = skip_tok(p, le);
        size_t tl = (size_t)(t - p);
        if (tl == 1 && p[0] == 'v') {
            int ok = 1;
            const char* q = t;
            double a = read_double(&q, le, &ok), b = read_double(&q, le, &ok), c = read_double(&q, le, &ok);
            if (nv == vcap) { vcap *= 2; verts = (float*)realloc(verts, (size_t)vcap * 3 * sizeof(float)); }
            /* double + (double)float, then rounded to float by the Vec3f constructor (:239) */
            verts[3 * nv + 0] = (float)(scale * (a + (double)offset[0]));
            verts[3 * nv + 1] = (float)(scale * (b + (double)offset[1]));
            verts[3 * nv + 2] = (float)(scale * (c + (double)offset[2]));
            ++nv;
        } else if (tl == 1 && p[0] == 'f') {
            int idx[3];
            const char* q = t;
            for (int k = 0; k < 3; ++k) {
                q = skip_ws(q, le);
                const char* te = skip_tok(q, le);
                const char* r = q;
                idx[k] = read_int(&r, te);
                /* with format_has_vt the reference then skips one character and reads the vt index,
                 * which never reaches the render path */
                q = te;
            }
            if (idx[0] < 1 || idx[1] < 1 || idx[2] < 1 || idx[0] > nv || idx[1] > nv || idx[2] > nv) { rc = -1; *le = saved; break; }
            scene_reserve(s, 1);
            float* o = s->v + 9 * s->ntri;
            for (int k = 0; k < 3; ++k) memcpy(o + 3 * k, verts + 3 * (int64_t)(idx[k] - 1), 3 * sizeof(float));
            set_material(s, s->ntri, mat);
            ++s->ntri; ++added;
        }
        /* "vt" records are parsed by the reference but unused; everything else is skipped */
        *le = saved;
        line = nl ? nl + 1 : bend;
    }
    (void)format_has_vt;
    free(verts); free(buf);
    s->built = 0;
    return rc < 0 ? rc : added;
}

/* ---- small helpers with the reference's exact operation order ---- */

/* (int)float as the x86-64 build does it (cvttss2si): out-of-range and NaN give INT_MIN. */
static inline int f2i(float x) {
    if (!(x > -2147483904.0f && x < 2147483648.0f)) return INT_MIN;
    return (int)x;
}
static inline int clampi(int w, int lo, int hi) { return w < lo ? lo : (w > hi ? hi : w); }
/* std::max(a,b) = (a<b)?b:a ; std::min(a,b) = (b<a)?b:a */
static inline float maxf_std(float a, float b) { return (a < b) ? b : a; }
static inline float minf_std(float a, float b) { return (b < a) ? b : a; }

/* Vec3::normalize, Serial/geometry.h:23-30: float length2, double sqrt, double divide, float scale */
static inline void normalize3(float v[3]) {
    float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (n2 > 0) {
        float inv = (float)(1 / sqrt((double)n2));
        v[0] *= inv; v[1] *= inv; v[2] *= inv;
    }
}
static inline void cross3(const float a[3], const float b[3], float o[3]) {
    float x = a[1] * b[2] - a[2] * b[1];
    float y = a[2] * b[0] - a[0] * b[2];
    float z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* det, Serial/raytracer.cpp:203-211 */
static inline double det3(double a1, double a2, double a3, double b1, double b2, double b3, double c1, double c2, double c3) {
    double t1 = a1 * (b2 * c3 - b3 * c2);
    double t2 = a2 * (b1 * c3 - b3 * c1);
    double t3 = a3 * (b1 * c2 - b2 * c1);
    return t1 - t2 + t3;
}

/* posToVoxel / voxelToPos, Serial/grid.h:59-71 */
static inline int pos_to_voxel(const rto_scene* s, float p, int a) {
    return clampi(f2i((p - s->lo[a]) * s->inv_width[a]), 0, s->n[a] - 1);
}
static inline float voxel_to_pos(const rto_scene* s, int p, int a) { return s->lo[a] + (float)p * s->width[a]; }

/* ---- grid build, Serial/grid.h:79-153, emitted as CSR ---- */

double rto_scene_build_grid(rto_scene* s) {
    double t0 = now_ms();
    free(s->cell_start); free(s->cell_tris);
    s->cell_start = NULL; s->cell_tris = NULL; s->nref = 0;
    for (int a = 0; a < 3; ++a) { s->lo[a] = INFINITY; s->hi[a] = -INFINITY; }
    for (int64_t i = 0; i < s->ntri; ++i)
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a) {
                float x = s->v[9 * i + 3 * k + a];
                s->hi[a] = maxf_std(x, s->hi[a]);
                s->lo[a] = minf_std(x, s->lo[a]);
            }
    float delta[3];
    for (int a = 0; a < 3; ++a) delta[a] = s->hi[a] - s->lo[a];
    /* boundingBox::maxAxis, Serial/geometry.h:276-285 */
    int ax = (delta[0] > delta[1]) ? 0 : 1;
    if (ax) ax = (delta[1] > delta[2]) ? 1 : 2; else ax = (delta[0] > delta[2]) ? 0 : 2;
    float max_inv = 1.f / delta[ax];
    float cube_root = 3.f * powf((float)s->ntri, 1.f / 3.f);
    float vpud = cube_root * max_inv;
    for (int a = 0; a < 3; ++a) {
        int nv = f2i(delta[a] * vpud + 1);
        s->n[a] = clampi(nv, 1, 64);
        s->width[a] = delta[a] / (float)s->n[a];
        s->inv_width[a] = (s->width[a] == 0.f) ? 0.f : 1.f / s->width[a];
    }
    int64_t C = (int64_t)s->n[0] * s->n[1] * s->n[2];
    s->cell_start = (uint32_t*)calloc((size_t)C + 1, sizeof(uint32_t));
    for (int pass = 0; pass < 2; ++pass) {
        uint32_t* fill = NULL;
        if (pass == 1) {
            uint32_t run = 0;
            for (int64_t c = 0; c < C; ++c) { uint32_t k = s->cell_start[c]; s->cell_start[c] = run; run += k; }
            s->cell_start[C] = run;
            s->nref = run;
            s->cell_tris = (uint32_t*)malloc(((size_t)run + 1) * sizeof(uint32_t));
            fill = (uint32_t*)malloc((size_t)C * sizeof(uint32_t));
            memcpy(fill, s->cell_start, (size_t)C * sizeof(uint32_t));
        }
        for (int64_t i = 0; i < s->ntri; ++i) {
            float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int k = 0; k < 3; ++k)
                for (int a = 0; a < 3; ++a) {
                    float x = s->v[9 * i + 3 * k + a];
                    bhi[a] = maxf_std(x, bhi[a]);
                    blo[a] = minf_std(x, blo[a]);
                }
            int vmin[3], vmax[3];
            for (int a = 0; a < 3; ++a) { vmin[a] = pos_to_voxel(s, blo[a], a); vmax[a] = pos_to_voxel(s, bhi[a], a); }
            for (int x = vmin[0]; x <= vmax[0]; ++x)
                for (int y = vmin[1]; y <= vmax[1]; ++y)
                    for (int z = vmin[2]; z <= vmax[2]; ++z) {
                        int64_t c = (int64_t)z * s->n[0] * s->n[1] + (int64_t)y * s->n[0] + x;
                        if (pass == 0) s->cell_start[c]++;
                        else s->cell_tris[fill[c]++] = (uint32_t)i;   /* ascending i within a cell */
                    }
        }
        free(fill);
    }
    s->built = 1;
    return now_ms() - t0;
}

void rto_grid_info(const rto_scene* s, int32_t n[3], float lohi[6], float width[3], float inv_width[3]) {
    for (int a = 0; a < 3; ++a) {
        n[a] = s->n[a]; lohi[a] = s->lo[a]; lohi[3 + a] = s->hi[a];
        width[a] = s->width[a]; inv_width[a] = s->inv_width[a];
    }
}

int64_t rto_grid_csr(const rto_scene* s, const uint32_t** cell_start, const uint32_t** cell_tris) {
    if (cell_start) *cell_start = s->cell_start;
    if (cell_tris) *cell_tris = s->cell_tris;
    return s->nref;
}

/* ---- traversal ---- */

typedef struct {
    float t;       /* global_t, Serial/geometry.h:10 */
    int32_t id;    /* global_triangle_near, :11 */
} near_state;

/* Triangle::Intersect, Serial/geometry.h:131-177, four determinants as written. */
static inline int tri_test(const float* tv, const float o[3], const float d[3], int use_eps, float eps, int32_t id, near_state* st) {
    float e1x = tv[0] - tv[3], e1y = tv[1] - tv[4], e1z = tv[2] - tv[5];
    float e2x = tv[0] - tv[6], e2y = tv[1] - tv[7], e2z = tv[2] - tv[8];
    float sx = tv[0] - o[0], sy = tv[1] - o[1], sz = tv[2] - o[2];
    double A = det3(e1x, e2x, d[0], e1y, e2y, d[1], e1z, e2z, d[2]);
    double t = det3(e1x, e2x, sx, e1y, e2y, sy, e1z, e2z, sz);
    t = t / A;
    double beta = det3(sx, e2x, d[0], sy, e2y, d[1], sz, e2z, d[2]);
    beta = beta / A;
    double gamma = det3(e1x, sx, d[0], e1y, sy, d[1], e1z, sz, d[2]);
    gamma = gamma / A;
    if (beta > 0 && gamma > 0 && beta + gamma < 1) {
        if (t < (double)st->t) {
            if (!use_eps || t > (double)eps) { st->t = (float)t; st->id = id; }
        }
        return 1;
    }
    return 0;
}

/* boundingBox::Intersect, Serial/geometry.h:291-316 (only hitt0 is consumed by the caller) */
static inline int slab_test(const rto_scene* s, const float o[3], const float d[3], float mint, float* hitt0) {
    float t0 = mint, t1 = INFINITY;
    for (int i = 0; i < 3; ++i) {
        float inv = 1.f / d[i];
        float tn = (s->lo[i] - o[i]) * inv;
        float tf = (s->hi[i] - o[i]) * inv;
        if (tn > tf) { float x = tn; tn = tf; tf = x; }
        t0 = tn > t0 ? tn : t0;
        t1 = tf < t1 ? tf : t1;
        if (t0 > t1) return 0;
    }
    *hitt0 = t0;
    return 1;
}

enum { WALK_FULL = 0, WALK_NEAREST_EARLY = 1, WALK_ANY_EARLY = 2 };

/* GridAccel::Intersect, Serial/grid.h:167-231. Returns hitSomething. entered/steps/tests are counters. */
static int grid_walk(const rto_scene* s, const float o[3], const float d[3], float mint, int use_eps, float eps,
                     int mode, float slack, near_state* st, uint64_t* entered, uint64_t* steps, uint64_t* tests) {
    float rayT;
    float p0[3] = {o[0] + d[0] * mint, o[1] + d[1] * mint, o[2] + d[2] * mint};
    int inside = p0[0] >= s->lo[0] && p0[0] <= s->hi[0] && p0[1] >= s->lo[1] && p0[1] <= s->hi[1] &&
                 p0[2] >= s->lo[2] && p0[2] <= s->hi[2];
    if (inside) rayT = mint;
    else if (!slab_test(s, o, d, mint, &rayT)) return 0;
    ++*entered;
    float gi[3] = {o[0] + d[0] * rayT, o[1] + d[1] * rayT, o[2] + d[2] * rayT};
    float next[3], delta[3];
    int step[3], out[3], pos[3];
    for (int a = 0; a < 3; ++a) {
        pos[a] = pos_to_voxel(s, gi[a], a);
        if (d[a] >= 0) {
            next[a] = rayT + (voxel_to_pos(s, pos[a] + 1, a) - gi[a]) / d[a];
            delta[a] = s->width[a] / d[a];
            step[a] = 1; out[a] = s->n[a];
        } else {
            next[a] = rayT + (voxel_to_pos(s, pos[a], a) - gi[a]) / d[a];
            delta[a] = -s->width[a] / d[a];
            step[a] = -1; out[a] = -1;
        }
    }
    static const int cmp_to_axis[8] = {2, 1, 2, 1, 2, 2, 0, 0};
    int hit = 0;
    for (;;) {
        int64_t c = (int64_t)pos[2] * s->n[0] * s->n[1] + (int64_t)pos[1] * s->n[0] + pos[0];
        uint32_t b = s->cell_start[c], e = s->cell_start[c + 1];
        ++*steps;
        for (uint32_t k = b; k < e; ++k) {
            uint32_t id = s->cell_tris[k];
            ++*tests;
            if (tri_test(s->v + 9 * (size_t)id, o, d, use_eps, eps, (int32_t)id, st)) {
                hit = 1;
                if (mode == WALK_ANY_EARLY) return 1;
            }
        }
        int bits = ((next[0] < next[1]) << 2) + ((next[0] < next[2]) << 1) + (next[1] < next[2]);
        int ax = cmp_to_axis[bits];
        /* ray.maxt is INFINITY for every ray the path creates, so the reference's maxt test never fires */
        if (mode == WALK_NEAREST_EARLY && hit) {
            float lim = st->t + fabsf(st->t) * slack;
            if (lim < next[ax]) break;
        }
        pos[ax] += step[ax];
        if (pos[ax] == out[ax]) break;
        next[ax] += delta[ax];
    }
    return hit;
}

typedef struct {
    float pos[3], w[3], u[3], v[3];
    float aspect, focal_distance;
} camera_t;

/* Serial/raytracer.cpp:130-138 */
static void camera_setup(const rto_params* p, camera_t* c) {
    float up[3] = {p->cam_up[0], p->cam_up[1], p->cam_up[2]};
    normalize3(up);
    float los[3];
    for (int a = 0; a < 3; ++a) { c->pos[a] = p->cam_pos[a]; los[a] = p->cam_target[a] - p->cam_pos[a]; }
    for (int a = 0; a < 3; ++a) c->w[a] = -los[a];
    normalize3(c->w);
    cross3(up, c->w, c->u); normalize3(c->u);
    cross3(c->w, c->u, c->v); normalize3(c->v);
    c->aspect = (float)p->width / (float)p->height;
    c->focal_distance = (float)(1.0f / (2.0 * tan(p->fov * M_PI / (180.0 * 2.0))));
}

double rto_render_rows(const rto_scene* s, const rto_params* p, int y0, int y1, float* rgb, int32_t* ids, rto_counters* cn) {
    camera_t cam;
    camera_setup(p, &cam);
    rto_counters local;
    memset(&local, 0, sizeof(local));
    const int W = p->width, H = p->height;
    const int pmode = p->walk_mode ? WALK_NEAREST_EARLY : WALK_FULL;
    const int smode = p->walk_mode ? WALK_ANY_EARLY : WALK_FULL;
    double t0 = now_ms();
    for (int y = y0; y < y1; ++y) {
        for (int x = 0; x < W; ++x) {
            size_t o = (size_t)(y - y0) * W + x;
            /* ray generation, Serial/raytracer.cpp:153-161 */
            float dir[3];
            for (int a = 0; a < 3; ++a) dir[a] = 0.f + (-cam.w[a]) * cam.focal_distance;
            float xw = (float)((double)cam.aspect * (x - W / 2.0 + 0.5) / W);
            float yw = (float)((y - H / 2.0 + 0.5) / H);
            for (int a = 0; a < 3; ++a) dir[a] = dir[a] + cam.u[a] * xw;
            for (int a = 0; a < 3; ++a) dir[a] = dir[a] + cam.v[a] * yw;
            normalize3(dir);
            ++local.primary_rays;

            /* fast_trace, Serial/raytracer.cpp:71-118 */
            near_state st = {INFINITY, -1};
            int hit = 0;
            if (s->ntri > 0)
                hit = grid_walk(s, cam.pos, dir, 0.f, 0, p->eps, pmode, p->early_slack, &st,
                                &local.primary_entering, &local.steps_primary, &local.tests_primary);
            if (ids) ids[o] = hit ? st.id : -1;
            if (!hit) {
                if (rgb) { rgb[3 * o] = 0.f; rgb[3 * o + 1] = 0.f; rgb[3 * o + 2] = 0.f; }
                continue;
            }
            ++local.hits;
            const float* tv = s->v + 9 * (size_t)st.id;
            const float* col = s->color + 3 * (size_t)st.id;
            float poi[3], vv[3], l[3], h[3], nrm[3];
            for (int a = 0; a < 3; ++a) poi[a] = cam.pos[a] + dir[a] * st.t;
            for (int a = 0; a < 3; ++a) vv[a] = -dir[a];
            normalize3(vv);
            for (int a = 0; a < 3; ++a) l[a] = p->light_pos[a] - poi[a];
            normalize3(l);
            for (int a = 0; a < 3; ++a) h[a] = vv[a] + l[a];
            float ab[3] = {tv[0] - tv[3], tv[1] - tv[4], tv[2] - tv[5]};   /* getNormalMod, geometry.h:234-240 */
            float ca[3] = {tv[6] - tv[0], tv[7] - tv[1], tv[8] - tv[2]};
            cross3(ab, ca, nrm);
            float fd = p->kd * maxf_std(0.f, dot3(nrm, l));
            float fs = (float)((double)p->ks * pow((double)maxf_std(0.f, dot3(nrm, h)), (double)p->spec_alpha));
            float c3[3];
            for (int a = 0; a < 3; ++a) {
                float diffuse = (col[a] * fd) * p->light_intensity;
                float specular = (col[a] * fs) * p->light_intensity;
                c3[a] = specular + diffuse;
            }
            /* shadow ray, Serial/raytracer.cpp:106-113: direction away from the light, start at eps,
             * the nearest-hit state is carried over, shadowed iff anything on the walk passes */
            float sd[3];
            for (int a = 0; a < 3; ++a) sd[a] = -(p->light_pos[a] - poi[a]);
            normalize3(sd);
            ++local.shadow_rays;
            int sh = grid_walk(s, poi, sd, p->eps, 1, p->eps, smode, 0.f, &st,
                               &local.shadow_entering, &local.steps_shadow, &local.tests_shadow);
            if (sh) { ++local.shadowed; for (int a = 0; a < 3; ++a) c3[a] = c3[a] * p->shadow_scale; }
            if (rgb) for (int a = 0; a < 3; ++a) rgb[3 * o + a] = c3[a] + col[a] * p->ka;
        }
    }
    double ms = now_ms() - t0;
    if (cn) {
        cn->primary_rays += local.primary_rays; cn->primary_entering += local.primary_entering; cn->hits += local.hits;
        cn->shadow_rays += local.shadow_rays; cn->shadow_entering += local.shadow_entering; cn->shadowed += local.shadowed;
        cn->steps_primary += local.steps_primary; cn->tests_primary += local.tests_primary;
        cn->steps_shadow += local.steps_shadow; cn->tests_shadow += local.tests_shadow;
    }
    return ms;
}

/* ---- the Parallel tree's fast_trace (PARITY UNPINNED, see rt_oracle.h) ---- */

typedef struct {
    double t;      /* t_min / tnear, a double that is never rounded (Parallel/raytracer.cu:459) */
    int32_t id;    /* triangle_near */
} near_par;

/* Triangle::Intersect, Parallel/geometry.cuh:123-170: true only when eps < t < t_min (and then t_min = t). */
static inline int tri_test_par(const float* tv, const float o[3], const float d[3], float eps, int32_t id, near_par* st) {
    float e1x = tv[0] - tv[3], e1y = tv[1] - tv[4], e1z = tv[2] - tv[5];
    float e2x = tv[0] - tv[6], e2y = tv[1] - tv[7], e2z = tv[2] - tv[8];
    float sx = tv[0] - o[0], sy = tv[1] - o[1], sz = tv[2] - o[2];
    double A = det3(e1x, e2x, d[0], e1y, e2y, d[1], e1z, e2z, d[2]);
    double beta = det3(sx, e2x, d[0], sy, e2y, d[1], sz, e2z, d[2]) / A;
    double gamma = det3(e1x, sx, d[0], e1y, sy, d[1], e1z, sz, d[2]) / A;
    if (beta > 0 && gamma > 0 && beta + gamma < 1) {
        double t = det3(e1x, e2x, sx, e1y, e2y, sy, e1z, e2z, sz) / A;
        if (t < st->t && t > eps) { st->t = t; st->id = id; return 1; }
    }
    return 0;
}

/* GridAccel::Intersect, Parallel/grid.cuh:224-290 (the same walk as the Serial tree's). mode: WALK_FULL as the reference,
 * WALK_NEAREST_EARLY stops once the nearest hit lies before the next crossing, WALK_ANY_EARLY at the first hit. */
static int grid_walk_par(const rto_scene* s, const float o[3], const float d[3], float mint, float eps, int mode, near_par* st) {
    float rayT;
    float p0[3] = {o[0] + d[0] * mint, o[1] + d[1] * mint, o[2] + d[2] * mint};
    int inside = p0[0] >= s->lo[0] && p0[0] <= s->hi[0] && p0[1] >= s->lo[1] && p0[1] <= s->hi[1] &&
                 p0[2] >= s->lo[2] && p0[2] <= s->hi[2];
    if (inside) rayT = mint;
    else if (!slab_test(s, o, d, mint, &rayT)) return 0;
    float gi[3] = {o[0] + d[0] * rayT, o[1] + d[1] * rayT, o[2] + d[2] * rayT};
    float next[3], delta[3];
    int step[3], out[3], pos[3];
    for (int a = 0; a < 3; ++a) {
        pos[a] = pos_to_voxel(s, gi[a], a);
        if (d[a] >= 0) {
            next[a] = rayT + (voxel_to_pos(s, pos[a] + 1, a) - gi[a]) / d[a];
            delta[a] = s->width[a] / d[a];
            step[a] = 1; out[a] = s->n[a];
        } else {
            next[a] = rayT + (voxel_to_pos(s, pos[a], a) - gi[a]) / d[a];
            delta[a] = -s->width[a] / d[a];
            step[a] = -1; out[a] = -1;
        }
    }
    static const int cmp_to_axis[8] = {2, 1, 2, 1, 2, 2, 0, 0};
    int hit = 0;
    for (;;) {
        int64_t c = (int64_t)pos[2] * s->n[0] * s->n[1] + (int64_t)pos[1] * s->n[0] + pos[0];
        uint32_t b = s->cell_start[c], e = s->cell_start[c + 1];
        for (uint32_t k = b; k < e; ++k) {
            uint32_t id = s->cell_tris[k];
            if (tri_test_par(s->v + 9 * (size_t)id, o, d, eps, (int32_t)id, st)) {
                hit = 1;
                if (mode == WALK_ANY_EARLY) return 1;
            }
        }
        int bits = ((next[0] < next[1]) << 2) + ((next[0] < next[2]) << 1) + (next[1] < next[2]);
        int ax = cmp_to_axis[bits];
        if (mode == WALK_NEAREST_EARLY && hit && st->t < (double)next[ax]) break;
        pos[ax] += step[ax];
        if (pos[ax] == out[ax]) break;
        next[ax] += delta[ax];
    }
    return hit;
}

/* fast_trace, Parallel/raytracer.cu:445-524, recursive like the reference. */
static void fast_trace_par(const rto_scene* s, const rto_params* p, const float orig[3], const float dir_in[3], float mint, int depth,
                           float out[3], int32_t* first_id) {
    out[0] = out[1] = out[2] = 0.f;
    near_par st = {INFINITY, -1};
    const int pmode = p->walk_mode ? WALK_NEAREST_EARLY : WALK_FULL;
    const int smode = p->walk_mode ? WALK_ANY_EARLY : WALK_FULL;
    int hit = s->ntri > 0 ? grid_walk_par(s, orig, dir_in, mint, p->eps, pmode, &st) : 0;
    if (first_id) *first_id = hit ? st.id : -1;
    if (!hit) return;
    const float* tv = s->v + 9 * (size_t)st.id;
    const float* base = s->color + 3 * (size_t)st.id;
    const float* m = s->mat + 6 * (size_t)st.id;
    const float kd = m[0], ks = m[1], alpha = m[2], ka = m[3], km = m[4];
    const int reflective = m[5] != 0.f;
    const float tn = (float)st.t;                          /* Vec3f::scale takes a float */
    float poi[3], v[3], l[3], h[3], n[3], color[3];
    for (int a = 0; a < 3; ++a) poi[a] = orig[a] + dir_in[a] * tn;
    for (int a = 0; a < 3; ++a) v[a] = -dir_in[a];
    normalize3(v);
    for (int a = 0; a < 3; ++a) l[a] = p->light_pos[a] - poi[a];
    normalize3(l);
    for (int a = 0; a < 3; ++a) h[a] = v[a] + l[a];
    normalize3(h);
    float a21[3] = {tv[6] - tv[3], tv[7] - tv[4], tv[8] - tv[5]};    /* v2 - v1 */
    float a01[3] = {tv[0] - tv[3], tv[1] - tv[4], tv[2] - tv[5]};    /* v0 - v1 */
    cross3(a21, a01, n);                                              /* Parallel/geometry.cuh:160 */
    float fd = maxf_std(0.f, dot3(n, l));
    float fs = powf(maxf_std(0.f, dot3(n, h)), alpha);
    for (int a = 0; a < 3; ++a) color[a] = ((base[a] * fd) * kd + (base[a] * fs) * ks) + base[a] * ka;
    /* shadow ray towards the light, :492-506 */
    near_par sst = {INFINITY, -1};
    if (grid_walk_par(s, poi, l, p->eps + p->shadow_bias, p->eps, smode, &sst))
        for (int a = 0; a < 3; ++a) color[a] = color[a] * p->shadow_scale;
    if (reflective && depth < p->reflect_depth) {
        float I[3] = {dir_in[0], dir_in[1], dir_in[2]}, r[3], rec[3];
        normalize3(I);
        normalize3(n);
        float factor = dot3(I, n) * 2;
        for (int a = 0; a < 3; ++a) r[a] = I[a] - n[a] * factor;
        normalize3(r);
        fast_trace_par(s, p, poi, r, p->eps, depth + 1, rec, NULL);
        for (int a = 0; a < 3; ++a) out[a] = (color[a] * base[a]) * (1 - km) + rec[a] * km;
        return;
    }
    for (int a = 0; a < 3; ++a) out[a] = color[a];
}

double rto_render_rows_parallel(const rto_scene* s, const rto_params* p, int y0, int y1, float* rgb, int32_t* ids) {
    camera_t cam;
    camera_setup(p, &cam);
    const int W = p->width, H = p->height;
    double t0 = now_ms();
    for (int y = y0; y < y1; ++y) {
        for (int x = 0; x < W; ++x) {
            size_t o = (size_t)(y - y0) * W + x;
            float dir[3], c3[3];
            for (int a = 0; a < 3; ++a) dir[a] = 0.f + (-cam.w[a]) * cam.focal_distance;     /* Parallel/raytracer.cu:352-359 */
            float xw = (float)((double)cam.aspect * (x - W / 2.0 + 0.5) / W);
            float yw = (float)((y - H / 2.0 + 0.5) / H);
            for (int a = 0; a < 3; ++a) dir[a] = dir[a] + cam.u[a] * xw;
            for (int a = 0; a < 3; ++a) dir[a] = dir[a] + cam.v[a] * yw;
            normalize3(dir);
            int32_t id = -1;
            fast_trace_par(s, p, cam.pos, dir, 0.f, 0, c3, &id);
            if (ids) ids[o] = id;
            if (rgb) { rgb[3 * o] = c3[0]; rgb[3 * o + 1] = c3[1]; rgb[3 * o + 2] = c3[2]; }
        }
    }
    return now_ms() - t0;
}

void rto_quantize(const float* c, int64_t n, unsigned char* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (unsigned char)(minf_std(1.f, c[i] / 255) * 255);
}

int rto_write_ppm(const char* path, const float* rgb, int w, int h) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    size_t n = (size_t)w * h * 3;
    unsigned char* b = (unsigned char*)malloc(n);
    rto_quantize(rgb, (int64_t)n, b);
    int ok = fwrite(b, 1, n, f) == n;
    free(b);
    fclose(f);
    return ok ? 0 : -1;
}

/* Indexed triangles as OBJ text ("v x y z" with 9 significant digits, which a float survives; "f a b c", 1-based, no
   texture index): the input the reference's load_mesh (Serial/raytracer.cpp:220-287, format_has_vt = false) reads back into
   the same floats. For timing the compiled reference on meshes that exist only in memory. */
int rto_write_obj(const char* path, const float* verts, int64_t nv, const int32_t* idx, int64_t nf) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    static char iobuf[1 << 22];
    setvbuf(f, iobuf, _IOFBF, sizeof(iobuf));
    for (int64_t i = 0; i < nv; ++i) fprintf(f, "v %.9g %.9g %.9g\n", (double)verts[3 * i], (double)verts[3 * i + 1], (double)verts[3 * i + 2]);
    for (int64_t i = 0; i < nf; ++i) fprintf(f, "f %d %d %d\n", idx[3 * i] + 1, idx[3 * i + 1] + 1, idx[3 * i + 2] + 1);
    int ok = !ferror(f);
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : -1;
}
