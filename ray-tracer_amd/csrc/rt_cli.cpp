// Command-line driver. With no arguments it does what the reference's main() does
// (Serial/raytracer.cpp:191-200): load spot_triangulated.obj, load blub_triangulated.obj offset by
// (1.5,0,0), both coloured (255,0,0), render 512x512 and write ./test.ppm.
#include "../../include/rt_amd.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

struct MeshArg { std::string path; int has_vt; float color[3]; float offset[3]; int scale; rt_material mat; bool has_mat; };

static void usage() {
    fprintf(stderr,
            "usage: rt [--obj PATH [--no-vt] [--color R G B] [--offset X Y Z]]... [--size W H] [--out FILE]\n"
            "          [--cam X Y Z] [--target X Y Z] [--up X Y Z] [--fov DEG] [--full-walk] [--stats]\n"
            "          [--frames N [--in-flight D]]   (render the frame N times, D frames in flight, and print the frame rate)\n"
            "          [--gpus G]   (the frame shared out over the first G GPUs in bands, gathered to GPU 0 with RCCL; with --frames: the rate)\n"
            "          [--parallel-shading [--light X Y Z] [--eps E] [--shadow-scale S] [--reflect-depth D]]\n"
            "          per mesh: [--scale N] [--material KD KS ALPHA KA KM REFLECTIVE]   (fast_trace of the Parallel tree)\n"
            "       with no --obj: spot_triangulated.obj and blub_triangulated.obj (+1.5 in x) from the working directory\n");
}

int main(int argc, char** argv) {
    rt_params p;
    rt_params_default(&p);
    std::vector<MeshArg> meshes;
    std::string out = "./test.ppm";
    bool stats = false;
    int frames = 0, in_flight = 4, gpus = 0;
    auto need = [&](int i, int n) { if (i + n >= argc) { usage(); exit(2); } };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--obj") { need(i, 1); MeshArg m{argv[++i], 1, {255, 0, 0}, {0, 0, 0}, 1, {}, false}; meshes.push_back(m); }
        else if (a == "--scale" && !meshes.empty()) { need(i, 1); meshes.back().scale = atoi(argv[++i]); }
        else if (a == "--material" && !meshes.empty()) {
            need(i, 6);
            rt_material& m = meshes.back().mat;
            m.kd = (float)atof(argv[++i]); m.ks = (float)atof(argv[++i]); m.spec_alpha = (float)atof(argv[++i]); m.ka = (float)atof(argv[++i]);
            m.km = (float)atof(argv[++i]); m.reflective = atoi(argv[++i]);
            meshes.back().has_mat = true;
        }
        else if (a == "--parallel-shading") { p.shading_mode = RT_SHADE_PARALLEL; p.eps = 1e-4f; p.shadow_scale = 0.5f; }
        else if (a == "--light") { need(i, 3); for (int k = 0; k < 3; ++k) p.light_pos[k] = (float)atof(argv[++i]); }
        else if (a == "--eps") { need(i, 1); p.eps = (float)atof(argv[++i]); }
        else if (a == "--shadow-scale") { need(i, 1); p.shadow_scale = (float)atof(argv[++i]); }
        else if (a == "--reflect-depth") { need(i, 1); p.reflect_depth = atoi(argv[++i]); }
        else if (a == "--no-vt" && !meshes.empty()) meshes.back().has_vt = 0;
        else if (a == "--color" && !meshes.empty()) { need(i, 3); for (int k = 0; k < 3; ++k) meshes.back().color[k] = (float)atof(argv[++i]); }
        else if (a == "--offset" && !meshes.empty()) { need(i, 3); for (int k = 0; k < 3; ++k) meshes.back().offset[k] = (float)atof(argv[++i]); }
        else if (a == "--size") { need(i, 2); p.width = atoi(argv[++i]); p.height = atoi(argv[++i]); }
        else if (a == "--out") { need(i, 1); out = argv[++i]; }
        else if (a == "--cam") { need(i, 3); for (int k = 0; k < 3; ++k) p.cam_pos[k] = (float)atof(argv[++i]); }
        else if (a == "--target") { need(i, 3); for (int k = 0; k < 3; ++k) p.cam_target[k] = (float)atof(argv[++i]); }
        else if (a == "--up") { need(i, 3); for (int k = 0; k < 3; ++k) p.cam_up[k] = (float)atof(argv[++i]); }
        else if (a == "--fov") { need(i, 1); p.fov = (float)atof(argv[++i]); }
        else if (a == "--full-walk") p.walk_mode = RT_WALK_FULL;
        else if (a == "--stats") stats = true;
        else if (a == "--frames") { need(i, 1); frames = atoi(argv[++i]); }
        else if (a == "--in-flight") { need(i, 1); in_flight = atoi(argv[++i]); }
        else if (a == "--gpus") { need(i, 1); gpus = atoi(argv[++i]); }
        else { usage(); return 2; }
    }
    if (meshes.empty()) {
        meshes.push_back(MeshArg{"spot_triangulated.obj", 1, {255, 0, 0}, {0, 0, 0}, 1, {}, false});
        meshes.push_back(MeshArg{"blub_triangulated.obj", 1, {255, 0, 0}, {1.5f, 0, 0}, 1, {}, false});
    }
    auto load_scene = [&]() -> rt_scene* {
        rt_scene* sc = rt_scene_create();
        for (auto& m : meshes) {
            long long n;
            if (m.has_mat || m.scale != 1) {
                if (!m.has_mat) { m.mat.kd = 10; m.mat.ks = 10; m.mat.spec_alpha = 1.25f; m.mat.ka = 0.3f; m.mat.km = 0; m.mat.reflective = 0; }
                for (int k = 0; k < 3; ++k) m.mat.base_color[k] = m.color[k];
                n = rt_scene_add_obj_material(sc, m.path.c_str(), m.has_vt, &m.mat, m.offset, m.scale);
            } else {
                n = rt_scene_add_obj(sc, m.path.c_str(), m.has_vt, m.color, m.offset);
            }
            if (n < 0) { fprintf(stderr, "rt: %s\n", rt_last_error()); rt_scene_destroy(sc); return nullptr; }
        }
        return sc;
    };
    if (gpus > 0) {
        // The frame on several GPUs of this process: every GPU holds the scene and renders the bands of its part, one RCCL
        // gather brings them to the first GPU, which puts them in frame order.
        if (gpus > 64) { usage(); return 2; }
        auto fail = [&](const char* what) { fprintf(stderr, "rt: %s: %s\n", what, rt_last_error()); return 1; };
        rt_comm* comm = nullptr;
        if (rt_comm_create(gpus, nullptr, &comm)) return fail("rt_comm_create");
        std::vector<rt_scene*> sc((size_t)gpus, nullptr);
        std::vector<void*> d_part((size_t)gpus, nullptr);
        p.part_count = gpus; p.part_index = 0;
        const size_t part_bytes = (size_t)rt_part_rows(&p) * p.width * 3, frame_bytes = (size_t)p.width * p.height * 3;
        void *d_stacked = nullptr, *d_frame = nullptr;
        for (int r = 0; r < gpus; ++r) {
            if (hipSetDevice(r) != hipSuccess) { fprintf(stderr, "rt: cannot select GPU %d\n", r); return 1; }
            if (!(sc[r] = load_scene())) return 1;
            if (rt_scene_upload(sc[r])) return fail("rt_scene_upload");
            if (hipMalloc(&d_part[r], part_bytes) != hipSuccess) { fprintf(stderr, "rt: out of device memory\n"); return 1; }
        }
        (void)hipSetDevice(0);
        if (hipMalloc(&d_stacked, part_bytes * gpus) != hipSuccess || hipMalloc(&d_frame, frame_bytes) != hipSuccess) { fprintf(stderr, "rt: out of device memory\n"); return 1; }
        const int count = frames > 0 ? frames : 1;
        double ms = 0;
        for (int pass = 0; pass < (frames > 0 ? 2 : 1); ++pass) {        // with --frames the first pass warms up
            const int n_pass = (frames > 0 && pass == 0) ? 3 : count;
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < n_pass; ++k) {
                for (int r = 0; r < gpus; ++r) {
                    (void)hipSetDevice(r);
                    rt_params q = p;
                    q.part_index = r;
                    if (rt_render_device(sc[r], &q, 1, nullptr, d_part[r], nullptr, nullptr)) return fail("rt_render_device");
                }
                if (rt_gather_bands(comm, &p, d_part.data(), d_stacked, d_frame, 3, nullptr)) return fail("rt_gather_bands");
            }
            for (int r = 0; r < gpus; ++r) { (void)hipSetDevice(r); if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "rt: device error\n"); return 1; } }
            ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n_pass;
        }
        if (frames > 0) fprintf(stderr, "%d frames of %dx%d on %d GPUs: %.4f ms per frame\n", frames, p.width, p.height, gpus, ms);
        (void)hipSetDevice(0);
        std::vector<unsigned char> img(frame_bytes);
        if (hipMemcpy(img.data(), d_frame, frame_bytes, hipMemcpyDeviceToHost) != hipSuccess || rt_write_ppm_u8(out.c_str(), img.data(), p.width, p.height)) {
            fprintf(stderr, "rt: cannot write %s\n", out.c_str()); return 1;
        }
        (void)hipFree(d_stacked); (void)hipFree(d_frame);
        for (int r = 0; r < gpus; ++r) { (void)hipSetDevice(r); (void)hipFree(d_part[r]); rt_scene_destroy(sc[r]); }
        rt_comm_destroy(comm);
        return 0;
    }
    rt_scene* s = load_scene();
    if (!s) return 1;
    if (frames > 0) {
        // a stream of frames: frame k on lane k % D, each with its own grid build and its own RGB8 buffer on the device
        void* lane[8] = {};
        void* d_u8[8] = {};
        if (rt_scene_upload(s) || rt_scene_set_frames_in_flight(s, in_flight)) { fprintf(stderr, "rt: %s\n", rt_last_error()); return 1; }
        const int n = rt_streams_create(in_flight, lane);
        if (n < 1) { fprintf(stderr, "rt: %s\n", rt_last_error()); return 1; }
        const size_t bytes = (size_t)p.width * p.height * 3;
        for (int k = 0; k < n; ++k)
            if (hipMalloc(&d_u8[k], bytes) != hipSuccess) { fprintf(stderr, "rt: out of device memory\n"); return 1; }
        double ms = 0;
        for (int pass = 0; pass < 2; ++pass) {           // the first pass warms up
            const int count = pass ? frames : 2 * n;
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < count; ++k)
                if (rt_render_device(s, &p, 1, nullptr, d_u8[k % n], nullptr, lane[k % n])) { fprintf(stderr, "rt: %s\n", rt_last_error()); return 1; }
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "rt: device error\n"); return 1; }
            ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / count;
        }
        fprintf(stderr, "%d frames of %dx%d, %d in flight: %.4f ms per frame\n", frames, p.width, p.height, n, ms);
        std::vector<unsigned char> img(bytes);
        if (hipMemcpy(img.data(), d_u8[(frames - 1) % n], bytes, hipMemcpyDeviceToHost) != hipSuccess || rt_write_ppm_u8(out.c_str(), img.data(), p.width, p.height)) {
            fprintf(stderr, "rt: cannot write %s\n", out.c_str()); return 1;
        }
        for (int k = 0; k < n; ++k) (void)hipFree(d_u8[k]);
        (void)rt_streams_destroy(n, lane);
    } else {
        int rc = rt_render_ppm(s, &p, out.c_str());
        if (rc) { fprintf(stderr, "rt: %s\n", rt_last_error()); return 1; }
    }
    if (stats) {
        rt_frame_stats st;
        if (rt_scene_last_stats(s, &st) == RT_OK)
            fprintf(stderr, "triangles %llu grid %dx%dx%d refs %u build %.3f ms render %.3f ms hits %llu shadowed %llu\n",
                    (unsigned long long)st.n_triangles, st.grid_n[0], st.grid_n[1], st.grid_n[2], st.n_refs, st.ms_build, st.ms_render,
                    (unsigned long long)st.hits, (unsigned long long)st.shadowed);
    }
    rt_scene_destroy(s);
    return 0;
}
