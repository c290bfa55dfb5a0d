"""Python host-side mirror of the reference's load_mesh()/render() interface, bound to the C ABI of
include/rt_amd.h (librt_amd.so, hand-written HIP for gfx950). This module is plumbing for tests and the
benchmark driver: it contains no rendering code and has no CPU fallback. If the shared library is missing
the import fails; if no HIP device is visible every compute call raises RtError.

The directory name contains a hyphen, so import it with
    rt = importlib.import_module("ray-tracer_amd")
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB", os.path.join(_HERE, "librt_amd.so"))   # override: A/B runs of two builds

RT_WALK_EARLY, RT_WALK_FULL = 0, 1
RT_SHADE_SERIAL, RT_SHADE_PARALLEL = 0, 1
RT_BAND_ROWS = 8


class RtError(RuntimeError):
    pass


class Params(C.Structure):
    """rt_params of include/rt_amd.h; the defaults are the literals of Serial/raytracer.cpp:83-89,124-128."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("cam_pos", C.c_float * 3), ("cam_target", C.c_float * 3), ("cam_up", C.c_float * 3),
        ("fov", C.c_float),
        ("light_pos", C.c_float * 3),
        ("kd", C.c_float), ("ks", C.c_float), ("ka", C.c_float),
        ("spec_alpha", C.c_float), ("light_intensity", C.c_float), ("shadow_scale", C.c_float), ("eps", C.c_float),
        ("walk_mode", C.c_int32),
        ("part_index", C.c_int32), ("part_count", C.c_int32),
        ("shading_mode", C.c_int32), ("reflect_depth", C.c_int32), ("shadow_bias", C.c_float),
    ]


class Material(C.Structure):
    """rt_material: one entry of the Parallel tree's material table (Parallel/geometry.cuh:284-303)."""
    _fields_ = [("base_color", C.c_float * 3), ("kd", C.c_float), ("ks", C.c_float), ("spec_alpha", C.c_float), ("ka", C.c_float),
                ("km", C.c_float), ("reflective", C.c_int32)]

    def __init__(self, base_color=(255, 0, 0), kd=10.0, ks=10.0, spec_alpha=1.25, ka=0.3, km=0.0, reflective=False):
        super().__init__((C.c_float * 3)(*[float(x) for x in base_color]), kd, ks, spec_alpha, ka, km, int(bool(reflective)))


def _as_dict(st):
    d = {}
    for n, _ in st._fields_:
        v = getattr(st, n)
        d[n] = list(v) if hasattr(v, "__len__") else v
    return d


class FrameStats(C.Structure):
    """rt_frame_stats of include/rt_amd.h."""
    _fields_ = [
        ("ms_build", C.c_float), ("ms_render", C.c_float), ("ms_total", C.c_float),
        ("grid_n", C.c_int32 * 3), ("n_cells", C.c_uint32), ("n_refs", C.c_uint32),
        ("n_triangles", C.c_uint64), ("primary_rays", C.c_uint64), ("hits", C.c_uint64), ("shadowed", C.c_uint64),
    ]


class FrameCounts(C.Structure):
    """rt_frame_counts of include/rt_amd_diag.h."""
    _fields_ = [
        ("primary_entering", C.c_uint64), ("steps_primary", C.c_uint64), ("tests_primary", C.c_uint64),
        ("shadow_entering", C.c_uint64), ("steps_shadow", C.c_uint64), ("tests_shadow", C.c_uint64),
        ("sched_rounds", C.c_uint64), ("sched_streams", C.c_uint64), ("sched_lanes", C.c_uint64),
        ("tile_cycles_max", C.c_uint64), ("tile_cycles_sum", C.c_uint64), ("tiles_traced", C.c_uint64),
        ("tile_rounds_max", C.c_uint64), ("cycles_test", C.c_uint64), ("tests_executed", C.c_uint64),
        ("runs_streamed", C.c_uint64), ("run_ray_trips", C.c_uint64), ("cycles_cull", C.c_uint64),
        ("wave_cycles_max", C.c_uint64), ("span_cycles", C.c_uint64),
    ]


# every symbol include/rt_amd.h and include/rt_amd_diag.h declare
EXPORTS = [
    "rt_last_error", "rt_version", "rt_params_default", "rt_part_rows",
    "rt_scene_create", "rt_scene_destroy", "rt_scene_add_obj", "rt_scene_add_triangles", "rt_scene_add_obj_material", "rt_scene_add_triangles_material", "rt_scene_num_triangles",
    "rt_scene_get_triangles", "rt_scene_upload", "rt_scene_build_grid", "rt_scene_get_grid", "rt_scene_get_csr",
    "rt_render_device", "rt_render_device_noalloc", "rt_scene_set_frames_in_flight", "rt_scene_frames_in_flight", "rt_streams_create", "rt_streams_destroy", "rt_render", "rt_render_ppm", "rt_scene_last_stats", "rt_scene_stats_at", "rt_scene_set_counting", "rt_scene_last_counts", "rt_scene_record_order", "rt_scene_triangle_order", "rt_scene_unit_costs", "rt_scene_test_feedback", "rt_loop_counts", "rt_scene_set_estimate_reuse", "rt_scene_estimates_run",
    "rt_deinterleave_device", "rt_selftest_division", "rt_comm_create", "rt_comm_destroy", "rt_comm_size", "rt_gather_bands",
    "rt_write_ppm", "rt_write_ppm_u8", "rt_make_plane_obj",
]

_lib = None


def lib():
    """Loads librt_amd.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    # A process must hold ONE HIP runtime. PyTorch wheels bundle their own libamdhip64.so.7; importing torch
    # first makes the loader resolve this library's dependency on that SONAME to the copy torch already
    # loaded, so tensors and this library share a device context. Without torch the system ROCm one is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing: build it with `make -C ray-tracer_amd/csrc` (or __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_uint32)
    L.rt_last_error.restype = C.c_char_p
    L.rt_version.restype = C.c_char_p
    L.rt_params_default.argtypes = [vp]
    L.rt_part_rows.argtypes = [vp]
    L.rt_part_rows.restype = i32
    L.rt_scene_create.restype = vp
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_add_obj.argtypes = [vp, C.c_char_p, C.c_int, vp, vp]
    L.rt_scene_add_obj.restype = i64
    L.rt_scene_add_triangles.argtypes = [vp, vp, i64, vp]
    L.rt_scene_add_triangles.restype = i64
    L.rt_scene_add_obj_material.argtypes = [vp, C.c_char_p, C.c_int, vp, vp, i32]
    L.rt_scene_add_obj_material.restype = i64
    L.rt_scene_add_triangles_material.argtypes = [vp, vp, i64, vp]
    L.rt_scene_add_triangles_material.restype = i64
    L.rt_scene_num_triangles.argtypes = [vp]
    L.rt_scene_num_triangles.restype = i64
    L.rt_scene_get_triangles.argtypes = [vp, vp]
    L.rt_scene_upload.argtypes = [vp]
    L.rt_scene_build_grid.argtypes = [vp, vp]
    L.rt_scene_get_grid.argtypes = [vp, vp, vp, vp, vp, u32p, u32p]
    L.rt_scene_get_csr.argtypes = [vp, vp, vp]
    L.rt_render_device.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
    L.rt_render.argtypes = [vp, vp, vp, vp, vp]
    L.rt_render_ppm.argtypes = [vp, vp, C.c_char_p]
    L.rt_scene_last_stats.argtypes = [vp, vp]
    L.rt_scene_stats_at.argtypes = [vp, i32, vp]
    L.rt_scene_set_counting.argtypes = [vp, i32]
    L.rt_scene_last_counts.argtypes = [vp, vp]
    L.rt_scene_record_order.argtypes = [vp, vp, vp]
    L.rt_scene_triangle_order.argtypes = [vp, vp, vp]
    L.rt_scene_unit_costs.argtypes = [vp, vp, i64]
    L.rt_scene_test_feedback.argtypes = [vp, vp, vp, vp]
    L.rt_loop_counts.argtypes = [vp, i32, i32]
    L.rt_scene_set_estimate_reuse.argtypes = [vp, i32]
    L.rt_scene_estimates_run.argtypes = [vp]
    L.rt_scene_estimates_run.restype = i64
    L.rt_render_device_noalloc.argtypes = [vp, vp, vp, vp, vp, vp]
    L.rt_scene_set_frames_in_flight.argtypes = [vp, i32]
    L.rt_scene_frames_in_flight.argtypes = [vp]
    L.rt_scene_frames_in_flight.restype = i32
    L.rt_streams_create.argtypes = [i32, vp]
    L.rt_streams_create.restype = i32
    L.rt_streams_destroy.argtypes = [i32, vp]
    L.rt_deinterleave_device.argtypes = [vp, vp, vp, i32, vp]
    L.rt_comm_create.argtypes = [i32, vp, C.POINTER(vp)]
    L.rt_comm_destroy.argtypes = [vp]
    L.rt_comm_size.argtypes = [vp]
    L.rt_comm_size.restype = i32
    L.rt_gather_bands.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    L.rt_selftest_division.argtypes = [vp, vp, i64]
    L.rt_selftest_division.restype = i64
    L.rt_write_ppm.argtypes = [C.c_char_p, vp, i32, i32]
    L.rt_write_ppm_u8.argtypes = [C.c_char_p, vp, i32, i32]
    L.rt_make_plane_obj.argtypes = [C.c_char_p, i32]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise RtError(f"rt_amd error {rc}: {lib().rt_last_error().decode(errors='replace')}")
    return rc


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def default_params(width=512, height=512, **kw):
    p = Params()
    lib().rt_params_default(C.byref(p))
    p.width, p.height = width, height
    for k, v in kw.items():
        setattr(p, k, _f3(v) if isinstance(v, (tuple, list)) else v)
    return p


def part_rows(p):
    return lib().rt_part_rows(C.byref(p))


class Scene:
    """The triangle list load_mesh() appends to (Serial/raytracer.cpp:193-196) and render() consumes."""

    def __init__(self):
        self.h = C.c_void_p(lib().rt_scene_create())
        if not self.h:
            raise RtError("rt_scene_create failed")

    def close(self):
        if getattr(self, "h", None):
            lib().rt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown: the module's globals may be gone already
            pass

    # load_mesh(filename, triangle_list, format_has_vt, mesh_color, reflective=false, offset=(0,0,0))
    def load_mesh(self, filename, format_has_vt=True, mesh_color=(255, 0, 0), offset=(0, 0, 0)):
        return _check(lib().rt_scene_add_obj(self.h, os.fsencode(filename), int(format_has_vt), _f3(mesh_color), _f3(offset)))

    def add_triangles(self, v9, mesh_color=(255, 0, 0)):
        v9 = np.ascontiguousarray(v9, dtype=np.float32).reshape(-1, 9)
        return _check(lib().rt_scene_add_triangles(self.h, v9.ctypes.data, v9.shape[0], _f3(mesh_color)))

    # load_mesh(filename, triangle_list, format_has_vt, material_index, offset, scale) of the Parallel tree
    def load_mesh_material(self, filename, format_has_vt, material, offset=(0, 0, 0), scale=1):
        return _check(lib().rt_scene_add_obj_material(self.h, os.fsencode(filename), int(format_has_vt), C.byref(material), _f3(offset), int(scale)))

    def add_triangles_material(self, v9, material):
        v9 = np.ascontiguousarray(v9, dtype=np.float32).reshape(-1, 9)
        return _check(lib().rt_scene_add_triangles_material(self.h, v9.ctypes.data, v9.shape[0], C.byref(material)))

    @property
    def num_triangles(self):
        return lib().rt_scene_num_triangles(self.h)

    def triangles(self):
        out = np.empty((self.num_triangles, 9), np.float32)
        if out.size:
            _check(lib().rt_scene_get_triangles(self.h, out.ctypes.data))
        return out

    def upload(self):
        _check(lib().rt_scene_upload(self.h))

    def build_grid(self, stream=None):
        _check(lib().rt_scene_build_grid(self.h, stream))

    def grid_info(self):
        n = (C.c_int32 * 3)()
        lohi = (C.c_float * 6)()
        w = (C.c_float * 3)()
        iw = (C.c_float * 3)()
        nc, nr = C.c_uint32(), C.c_uint32()
        _check(lib().rt_scene_get_grid(self.h, n, lohi, w, iw, C.byref(nc), C.byref(nr)))
        return dict(n=np.array(n[:], np.int32), lohi=np.array(lohi[:], np.float32), width=np.array(w[:], np.float32),
                    inv_width=np.array(iw[:], np.float32), n_cells=nc.value, n_refs=nr.value)

    def grid_csr(self):
        gi = self.grid_info()
        start = np.empty(gi["n_cells"] + 1, np.uint32)
        tris = np.empty(max(gi["n_refs"], 1), np.uint32)
        _check(lib().rt_scene_get_csr(self.h, start.ctypes.data, tris.ctypes.data))
        return start, tris[:gi["n_refs"]]

    # render(triangle_list): host-buffer variant
    def render(self, params, want_rgb=True, want_u8=False, want_ids=True):
        rows, w = part_rows(params), params.width
        rgb = np.empty((rows, w, 3), np.float32) if want_rgb else None
        u8 = np.empty((rows, w, 3), np.uint8) if want_u8 else None
        ids = np.empty((rows, w), np.int32) if want_ids else None
        _check(lib().rt_render(self.h, C.byref(params), rgb.ctypes.data if want_rgb else None,
                               u8.ctypes.data if want_u8 else None, ids.ctypes.data if want_ids else None))
        return rgb, u8, ids

    def render_device(self, params, rebuild_grid=False, d_rgb=None, d_u8=None, d_ids=None, stream=None):
        """Device pointers are plain integers (e.g. torch.Tensor.data_ptr())."""
        _check(lib().rt_render_device(self.h, C.byref(params), int(rebuild_grid), d_rgb, d_u8, d_ids, stream))

    def render_ppm(self, params, path):
        _check(lib().rt_render_ppm(self.h, C.byref(params), os.fsencode(path)))

    def last_stats(self):
        """Timings and ray counts of the last frame plus the work counters of the counting build (zeros otherwise)."""
        st, cn = FrameStats(), FrameCounts()
        _check(lib().rt_scene_last_stats(self.h, C.byref(st)))
        _check(lib().rt_scene_last_counts(self.h, C.byref(cn)))
        d = _as_dict(st)
        d.update(_as_dict(cn))
        return d

    def stats_at(self, frames_back):
        st = FrameStats()
        _check(lib().rt_scene_stats_at(self.h, frames_back, C.byref(st)))
        return _as_dict(st)

    def render_device_noalloc(self, params, d_rgb=None, d_u8=None, d_ids=None, stream=None):
        """rt_render_device_noalloc: the frame's working memory is the library's fixed device array."""
        _check(lib().rt_render_device_noalloc(self.h, C.byref(params), d_rgb, d_u8, d_ids, stream))

    def set_frames_in_flight(self, n):
        """Working sets of the scene: frames on up to n different streams overlap on the GPU (rt_scene_set_frames_in_flight)."""
        _check(lib().rt_scene_set_frames_in_flight(self.h, n))

    def frames_in_flight(self):
        return lib().rt_scene_frames_in_flight(self.h)

    def record_order(self):
        """(area unsorted, area sorted, frames sort?) as measured by upload(); see rt_scene_record_order."""
        a, srt = (C.c_double * 2)(), C.c_int32()
        _check(lib().rt_scene_record_order(self.h, a, C.byref(srt)))
        return a[0], a[1], bool(srt.value)

    def triangle_order(self):
        """(area of the unsorted build in spatial order, frames build in spatial order?); see rt_scene_triangle_order."""
        a, sp = C.c_double(), C.c_int32()
        _check(lib().rt_scene_triangle_order(self.h, C.byref(a), C.byref(sp)))
        return a.value, bool(sp.value)

    def unit_costs(self, n):
        out = np.zeros(n, np.uint32)
        k = _check(lib().rt_scene_unit_costs(self.h, out.ctypes.data, n))
        return out[:k]

    def test_feedback(self):
        """(record runs streamed, trips with a candidate triangle, deferred kernel variant next?) as reported by an earlier
        frame; see rt_scene_test_feedback."""
        r, h, d = C.c_uint64(), C.c_uint64(), C.c_int()
        _check(lib().rt_scene_test_feedback(self.h, C.byref(r), C.byref(h), C.byref(d)))
        return r.value, h.value, bool(d.value)

    def set_counting(self, enable):
        _check(lib().rt_scene_set_counting(self.h, int(bool(enable))))

    def estimates_run(self):
        return int(lib().rt_scene_estimates_run(self.h))

    def set_estimate_reuse(self, enable):
        """Frames with the view and triangles of the last estimated frame skip the tile cost estimate (default on)."""
        _check(lib().rt_scene_set_estimate_reuse(self.h, int(bool(enable))))


def create_streams(n):
    """Up to n stream handles (integers) that run side by side on the GPU, see rt_streams_create; release them with
    destroy_streams(). torch.cuda.ExternalStream(handle) makes one usable from torch."""
    arr = (C.c_void_p * n)()
    k = _check(lib().rt_streams_create(n, arr))
    return [int(arr[i]) for i in range(k)]


def destroy_streams(handles):
    arr = (C.c_void_p * len(handles))(*handles)
    _check(lib().rt_streams_destroy(len(handles), arr))


def part_row_map(params):
    """Global frame row of every local row of part params.part_index (-1 for padding rows below the frame):
    band b of RT_BAND_ROWS rows belongs to part b % part_count, local band j of a part is global band
    part_index + j * part_count. Host-side statement of the layout rt_render_device() writes."""
    rows = part_rows(params)
    out = np.full(rows, -1, np.int64)
    if params.part_count <= 1:
        out[:] = np.arange(rows)
        return out
    for local in range(rows):
        band = (local // RT_BAND_ROWS) * params.part_count + params.part_index
        y = band * RT_BAND_ROWS + local % RT_BAND_ROWS
        if y < params.height:
            out[local] = y
    return out


def deinterleave_host(params, stacked):
    """Host counterpart of rt_deinterleave_device: stacked = [part_count, part_rows, W, ...] -> [H, W, ...]."""
    stacked = np.asarray(stacked)
    out = np.zeros((params.height,) + stacked.shape[2:], stacked.dtype)
    for k in range(params.part_count):
        q = Params.from_buffer_copy(params)
        q.part_index = k
        m = part_row_map(q)
        out[m[m >= 0]] = stacked[k][m >= 0]
    return out


def deinterleave_device(params, d_src, d_dst, elem_bytes, stream=None):
    _check(lib().rt_deinterleave_device(C.byref(params), d_src, d_dst, elem_bytes, stream))


class Comm:
    """rt_comm: one process, several GPUs, RCCL gather of the rendered bands to the first of them."""

    def __init__(self, devices):
        devices = list(devices)
        arr = (C.c_int32 * len(devices))(*devices)
        self.h = C.c_void_p()
        _check(lib().rt_comm_create(len(devices), arr, C.byref(self.h)))
        self.devices = devices

    def close(self):
        if getattr(self, "h", None):
            lib().rt_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def gather_bands(self, params, d_parts, d_stacked, d_frame, elem_bytes, streams=None):
        n = len(self.devices)
        parts = (C.c_void_p * n)(*d_parts)
        st = (C.c_void_p * n)(*streams) if streams is not None else None
        _check(lib().rt_gather_bands(self.h, C.byref(params), parts, d_stacked, d_frame, elem_bytes, st))


def selftest_division(num, den):
    """rt_selftest_division: pairs for which the shared-reciprocal division differs from the plain one (GPU)."""
    num = np.ascontiguousarray(num, dtype=np.float64); den = np.ascontiguousarray(den, dtype=np.float64)
    assert num.shape == den.shape and num.ndim == 1
    r = lib().rt_selftest_division(num.ctypes.data, den.ctypes.data, num.size)
    if r < 0:
        _check(int(r))
    return int(r)


def write_ppm(path, rgb_f32):
    rgb = np.ascontiguousarray(rgb_f32, dtype=np.float32)
    h, w = rgb.shape[:2]
    _check(lib().rt_write_ppm(os.fsencode(path), rgb.ctypes.data, w, h))


def make_plane_obj(path, squares_per_unit=10):
    _check(lib().rt_make_plane_obj(os.fsencode(path), squares_per_unit))


def subdivide(v9, k):
    """Midpoint subdivision of a triangle soup, k times (each triangle -> 4). The midpoint of an edge is
    (a + b) * 0.5 in float, symmetric in a and b, so shared edges stay watertight without an index.
    Used for the labelled stand-in of the absent nefertiti mesh."""
    t = np.ascontiguousarray(v9, dtype=np.float32).reshape(-1, 3, 3)
    for _ in range(k):
        a, b, c = t[:, 0], t[:, 1], t[:, 2]
        ab = ((a + b) * np.float32(0.5)).astype(np.float32)
        bc = ((b + c) * np.float32(0.5)).astype(np.float32)
        ca = ((c + a) * np.float32(0.5)).astype(np.float32)
        t = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1),
                      np.stack([ab, bc, ca], 1)], 1).reshape(-1, 3, 3)
    return np.ascontiguousarray(t.reshape(-1, 9))
