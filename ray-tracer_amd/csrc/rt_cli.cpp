// Command-line driver. With no arguments it does what the reference's main() does
// (Serial/raytracer.cpp:191-200): load spot_triangulated.obj, load blub_triangulated.obj offset by
// (1.5,0,0), both coloured (255,0,0), render 512x512 and write ./test.ppm.
#include "../../include/rt_amd.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
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
    auto print_stats = [&](rt_scene* sc) {
        rt_frame_stats st;
        if (rt_scene_last_stats(sc, &st) == RT_OK)
            fprintf(stderr, "triangles %llu grid %dx%dx%d refs %u build %.3f ms render %.3f ms hits %llu shadowed %llu\n",
                    (unsigned long long)st.n_triangles, st.grid_n[0], st.grid_n[1], st.grid_n[2], st.n_refs, st.ms_build, st.ms_render,
                    (unsigned long long)st.hits, (unsigned long long)st.shadowed);
        else
            fprintf(stderr, "rt: no stats: %s\n", rt_last_error());
    };
    if (gpus > 0) {
        // The frame on several GPUs of this process: every GPU holds the scene and renders the bands of its part, one RCCL
        // gather brings them to the first GPU, which puts them in frame order. Every GPU has a host thread of its own that
        // issues its frames (frame k on lane k % D of that GPU, D frames in flight), so the host cost of issuing a frame is
        // paid side by side; the main thread posts the gather of frame k, on the lanes of frame k, once every GPU has
        // issued that frame, and a GPU may run at most D frames ahead of the gathers (its band buffers are per lane).
        if (gpus > 64) { usage(); return 2; }
        auto fail = [&](const char* what) { fprintf(stderr, "rt: %s: %s\n", what, rt_last_error()); return 1; };
        rt_comm* comm = nullptr;
        if (rt_comm_create(gpus, nullptr, &comm)) return fail("rt_comm_create");
        struct Dev { rt_scene* sc = nullptr; void* lane[8] = {}; void* d_part[8] = {}; double enqueue_ms = 0; std::string err; };
        std::vector<Dev> dev((size_t)gpus);
        p.part_count = gpus; p.part_index = 0;
        const size_t part_bytes = (size_t)rt_part_rows(&p) * p.width * 3, frame_bytes = (size_t)p.width * p.height * 3;
        int depth = frames > 0 ? (in_flight < 1 ? 1 : (in_flight > 8 ? 8 : in_flight)) : 1;
        for (int r = 0; r < gpus; ++r) {
            if (hipSetDevice(r) != hipSuccess) { fprintf(stderr, "rt: cannot select GPU %d\n", r); return 1; }
            if (!(dev[r].sc = load_scene())) return 1;
            if (rt_scene_upload(dev[r].sc) || rt_scene_set_frames_in_flight(dev[r].sc, depth)) return fail("rt_scene_upload");
            const int n = rt_streams_create(depth, dev[r].lane);
            if (n < 1) return fail("rt_streams_create");
            if (n < depth) {                       // fewer streams that run side by side than asked for: every GPU uses that many
                for (int q = 0; q <= r; ++q) { (void)hipSetDevice(q); (void)rt_scene_set_frames_in_flight(dev[q].sc, n); }
                depth = n;
            }
        }
        void *d_stacked[8] = {}, *d_frame[8] = {};
        for (int r = 0; r < gpus; ++r) {
            (void)hipSetDevice(r);
            for (int k = 0; k < depth; ++k)
                if (hipMalloc(&dev[r].d_part[k], part_bytes) != hipSuccess) { fprintf(stderr, "rt: out of device memory\n"); return 1; }
        }
        (void)hipSetDevice(0);
        for (int k = 0; k < depth; ++k)
            if (hipMalloc(&d_stacked[k], part_bytes * gpus) != hipSuccess || hipMalloc(&d_frame[k], frame_bytes) != hipSuccess) { fprintf(stderr, "rt: out of device memory\n"); return 1; }
        const int count = frames > 0 ? frames : 1;
        double ms = 0;
        for (int pass = 0; pass < (frames > 0 ? 2 : 1); ++pass) {        // with --frames the first pass warms up
            const int n_pass = (frames > 0 && pass == 0) ? 2 * depth : count;
            std::mutex mu;
            std::condition_variable cv;
            std::vector<int> issued((size_t)gpus, 0);
            int gathered = 0;
            bool stop = false;
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int r = 0; r < gpus; ++r) {
                th.emplace_back([&, r] {
                    Dev& d = dev[r];
                    d.enqueue_ms = 0;
                    if (hipSetDevice(r) != hipSuccess) d.err = "cannot select the GPU";
                    rt_params q = p;
                    q.part_index = r;
                    for (int k = 0; k < n_pass && d.err.empty(); ++k) {
                        {
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [&] { return stop || k < gathered + depth; });
                            if (stop) break;
                        }
                        const auto e0 = std::chrono::steady_clock::now();
                        if (rt_render_device(d.sc, &q, 1, nullptr, d.d_part[k % depth], nullptr, d.lane[k % depth])) d.err = rt_last_error();
                        d.enqueue_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - e0).count();
                        {
                            std::lock_guard<std::mutex> lk(mu);
                            issued[r] = k + 1;
                            if (!d.err.empty()) stop = true;
                        }
                        cv.notify_all();
                    }
                    if (!d.err.empty()) { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_all(); }
                });
            }
            int rc = 0;
            for (int k = 0; k < n_pass && !rc; ++k) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { if (stop) return true; for (int v : issued) if (v < k + 1) return false; return true; });
                    if (stop) break;
                }
                void* parts[64]; void* lanes[64];
                for (int r = 0; r < gpus; ++r) { parts[r] = dev[r].d_part[k % depth]; lanes[r] = dev[r].lane[k % depth]; }
                if (rt_gather_bands(comm, &p, parts, d_stacked[k % depth], d_frame[k % depth], 3, lanes)) rc = fail("rt_gather_bands");
                { std::lock_guard<std::mutex> lk(mu); gathered = k + 1; if (rc) stop = true; }
                cv.notify_all();
            }
            for (auto& t : th) t.join();
            for (int r = 0; r < gpus; ++r)
                if (!dev[r].err.empty()) { fprintf(stderr, "rt: GPU %d: %s\n", r, dev[r].err.c_str()); rc = 1; }
            if (rc) return 1;
            for (int r = 0; r < gpus; ++r) { (void)hipSetDevice(r); if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "rt: device error\n"); return 1; } }
            ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n_pass;
        }
        if (frames > 0) {
            double enq = 0;
            for (int r = 0; r < gpus; ++r) enq += dev[r].enqueue_ms / count;
            fprintf(stderr, "%d frames of %dx%d on %d GPUs, %d in flight: %.4f ms per frame; host enqueue %.4f ms per frame and GPU (one host thread per GPU)\n",
                    frames, p.width, p.height, gpus, depth, ms, enq / gpus);
        }
        (void)hipSetDevice(0);
        std::vector<unsigned char> img(frame_bytes);
        if (hipMemcpy(img.data(), d_frame[(count - 1) % depth], frame_bytes, hipMemcpyDeviceToHost) != hipSuccess || rt_write_ppm_u8(out.c_str(), img.data(), p.width, p.height)) {
            fprintf(stderr, "rt: cannot write %s\n", out.c_str()); return 1;
        }
        if (stats) print_stats(dev[0].sc);
        for (int k = 0; k < depth; ++k) { (void)hipFree(d_stacked[k]); (void)hipFree(d_frame[k]); }
        for (int r = 0; r < gpus; ++r) {
            (void)hipSetDevice(r);
            for (int k = 0; k < depth; ++k) (void)hipFree(dev[r].d_part[k]);
            rt_scene_destroy(dev[r].sc);             // before its streams go: a scene keeps the streams of its last frames
            (void)rt_streams_destroy(depth, dev[r].lane);
        }
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
        // the scene still refers to the lanes of its last frames: stats first, then the scene, then the streams
        if (stats) print_stats(s);
        rt_scene_destroy(s);
        (void)rt_streams_destroy(n, lane);
        return 0;
    }
    int rc = rt_render_ppm(s, &p, out.c_str());
    if (rc) { fprintf(stderr, "rt: %s\n", rt_last_error()); return 1; }
    if (stats) print_stats(s);
    rt_scene_destroy(s);
    return 0;
}
