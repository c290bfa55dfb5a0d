"""TEST INFRASTRUCTURE ONLY: ctypes bindings for oracle/liboracle.so (the C restatement) and
oracle/_ref/libref.so (the reference compiled from its own sources). Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by the product package."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("cam_pos", C.c_float * 3), ("cam_target", C.c_float * 3), ("cam_up", C.c_float * 3),
        ("fov", C.c_float),
        ("light_pos", C.c_float * 3),
        ("kd", C.c_float), ("ks", C.c_float), ("ka", C.c_float), ("spec_alpha", C.c_float),
        ("light_intensity", C.c_float), ("shadow_scale", C.c_float), ("eps", C.c_float),
        ("walk_mode", C.c_int32), ("early_slack", C.c_float),
        ("reflect_depth", C.c_int32), ("shadow_bias", C.c_float),
    ]


class Material(C.Structure):
    _fields_ = [("base_color", C.c_float * 3), ("kd", C.c_float), ("ks", C.c_float), ("spec_alpha", C.c_float), ("ka", C.c_float),
                ("km", C.c_float), ("reflective", C.c_int32)]

    def __init__(self, base_color=(255, 0, 0), kd=10.0, ks=10.0, spec_alpha=1.25, ka=0.3, km=0.0, reflective=False):
        super().__init__((C.c_float * 3)(*[float(x) for x in base_color]), kd, ks, spec_alpha, ka, km, int(bool(reflective)))


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "primary_rays", "primary_entering", "hits", "shadow_rays", "shadow_entering", "shadowed",
        "steps_primary", "tests_primary", "steps_shadow", "tests_shadow")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.rto_scene_create.restype = C.c_void_p
        L.rto_scene_destroy.argtypes = [C.c_void_p]
        L.rto_scene_add_obj.restype = C.c_int64
        L.rto_scene_add_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rto_scene_add_triangles.restype = C.c_int64
        L.rto_scene_add_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.rto_scene_add_obj_material.restype = C.c_int64
        L.rto_scene_add_obj_material.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.rto_scene_add_triangles_material.restype = C.c_int64
        L.rto_scene_add_triangles_material.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.rto_render_rows_parallel.restype = C.c_double
        L.rto_render_rows_parallel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rto_scene_num_triangles.restype = C.c_int64
        L.rto_scene_num_triangles.argtypes = [C.c_void_p]
        L.rto_scene_triangles.restype = C.POINTER(C.c_float)
        L.rto_scene_triangles.argtypes = [C.c_void_p]
        L.rto_scene_build_grid.restype = C.c_double
        L.rto_scene_build_grid.argtypes = [C.c_void_p]
        L.rto_grid_info.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.rto_grid_csr.restype = C.c_int64
        L.rto_grid_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rto_render_rows.restype = C.c_double
        L.rto_render_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rto_quantize.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.rto_write_obj.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.rto_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.rto_params_default.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def default_params(width=512, height=512, walk_mode=0, **kw):
    p = Params()
    lib().rto_params_default(C.byref(p))
    p.width, p.height, p.walk_mode = width, height, walk_mode
    for k, v in kw.items():
        if isinstance(v, (tuple, list)):
            setattr(p, k, _f3(v))
        else:
            setattr(p, k, v)
    return p


class OracleScene:
    """The C restatement."""

    def __init__(self):
        self.h = C.c_void_p(lib().rto_scene_create())
        self.grid_ms = None

    def close(self):
        if self.h:
            lib().rto_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: the module globals may be gone already
            pass

    def add_obj(self, path, has_vt=True, color=(255, 0, 0), offset=(0, 0, 0)):
        return lib().rto_scene_add_obj(self.h, os.fsencode(path), int(has_vt), _f3(color), _f3(offset))

    def add_triangles(self, v9, color=(255, 0, 0)):
        v9 = np.ascontiguousarray(v9, dtype=np.float32).reshape(-1, 9)
        return lib().rto_scene_add_triangles(self.h, v9.ctypes.data, v9.shape[0], _f3(color))

    def add_obj_material(self, path, has_vt, material, offset=(0, 0, 0), scale=1):
        return lib().rto_scene_add_obj_material(self.h, os.fsencode(path), int(has_vt), C.byref(material), _f3(offset), int(scale))

    def add_triangles_material(self, v9, material):
        v9 = np.ascontiguousarray(v9, dtype=np.float32).reshape(-1, 9)
        return lib().rto_scene_add_triangles_material(self.h, v9.ctypes.data, v9.shape[0], C.byref(material))

    def render_parallel(self, params, y0=0, y1=None):
        """fast_trace of the Parallel tree (parity unpinned): returns (rgb float32 [rows,W,3], ids int32 [rows,W], ms)."""
        if self.grid_ms is None:
            self.build_grid()
        y1 = params.height if y1 is None else y1
        rows, w = y1 - y0, params.width
        rgb = np.empty((rows, w, 3), np.float32)
        ids = np.empty((rows, w), np.int32)
        ms = lib().rto_render_rows_parallel(self.h, C.byref(params), y0, y1, rgb.ctypes.data, ids.ctypes.data)
        return rgb, ids, ms

    @property
    def num_triangles(self):
        return lib().rto_scene_num_triangles(self.h)

    def triangles(self):
        n = self.num_triangles
        return np.ctypeslib.as_array(lib().rto_scene_triangles(self.h), shape=(n, 9)).copy()

    def build_grid(self):
        self.grid_ms = lib().rto_scene_build_grid(self.h)
        return self.grid_ms

    def grid_info(self):
        n = (C.c_int32 * 3)()
        lohi = (C.c_float * 6)()
        w = (C.c_float * 3)()
        iw = (C.c_float * 3)()
        lib().rto_grid_info(self.h, n, lohi, w, iw)
        return dict(n=np.array(n[:], dtype=np.int32), lohi=np.array(lohi[:], dtype=np.float32),
                    width=np.array(w[:], dtype=np.float32), inv_width=np.array(iw[:], dtype=np.float32))

    def grid_csr(self):
        cs = C.POINTER(C.c_uint32)()
        ct = C.POINTER(C.c_uint32)()
        r = lib().rto_grid_csr(self.h, C.byref(cs), C.byref(ct))
        n = self.grid_info()["n"]
        ncell = int(n[0]) * int(n[1]) * int(n[2])
        start = np.ctypeslib.as_array(cs, shape=(ncell + 1,)).copy()
        tris = np.ctypeslib.as_array(ct, shape=(max(r, 1),))[:r].copy()
        return start, tris

    def render(self, params, y0=0, y1=None, want_rgb=True, want_ids=True, counters=None):
        """Returns (rgb float32 [rows,W,3] or None, ids int32 [rows,W] or None, ms)."""
        if self.grid_ms is None:
            self.build_grid()
        y1 = params.height if y1 is None else y1
        rows, w = y1 - y0, params.width
        rgb = np.empty((rows, w, 3), np.float32) if want_rgb else None
        ids = np.empty((rows, w), np.int32) if want_ids else None
        ms = lib().rto_render_rows(self.h, C.byref(params), y0, y1,
                                   rgb.ctypes.data if want_rgb else None,
                                   ids.ctypes.data if want_ids else None,
                                   C.byref(counters) if counters is not None else None)
        return rgb, ids, ms


def quantize(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.empty(rgb.shape, np.uint8)
    lib().rto_quantize(rgb.ctypes.data, rgb.size, out.ctypes.data)
    return out


def write_obj(path, v9):
    """Writes a triangle soup (n x 9 floats) as an indexed OBJ without texture indices; load_mesh(path, format_has_vt=False)
    of the reference, or add_obj(path, False) here, reads the same floats back. Returns (vertices, faces)."""
    v9 = np.ascontiguousarray(v9, dtype=np.float32)
    pts = v9.reshape(-1, 3)
    key = np.where(pts == 0, np.float32(0), pts)          # -0.0 and +0.0 print alike: one vertex
    key = np.ascontiguousarray(key).view(np.dtype((np.void, 12))).ravel()
    u, inv = np.unique(key, return_inverse=True)
    verts = np.ascontiguousarray(u.view(np.float32).reshape(-1, 3))
    idx = np.ascontiguousarray(inv.reshape(-1, 3), dtype=np.int32)
    if lib().rto_write_obj(os.fsencode(path), verts.ctypes.data, verts.shape[0], idx.ctypes.data, idx.shape[0]) != 0:
        raise OSError("cannot write " + path)
    return verts.shape[0], idx.shape[0]


# ---------------------------------------------------------------------------------------------
_ref = None


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref.so"))


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref.so"))
        L.ref_scene_create.restype = C.c_void_p
        L.ref_scene_add_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_scene_num_triangles.restype = C.c_int64
        L.ref_scene_num_triangles.argtypes = [C.c_void_p]
        L.ref_scene_get_triangles.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_scene_build_grid.restype = C.c_double
        L.ref_scene_build_grid.argtypes = [C.c_void_p]
        L.ref_grid_info.argtypes = [C.c_void_p] * 5
        L.ref_grid_csr.restype = C.c_int64
        L.ref_grid_csr.argtypes = [C.c_void_p] * 3
        L.ref_render_rows.restype = C.c_double
        L.ref_render_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_quantize.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        _ref = L
    return _ref


class RefScene:
    """The reference itself (compiled from /root/reference/Serial by oracle/Makefile). It leaks by
    design (the reference never frees), so keep frames small."""

    def __init__(self):
        self.h = C.c_void_p(ref_lib().ref_scene_create())
        self.grid_ms = None

    def add_obj(self, path, has_vt=True, color=(255, 0, 0), offset=(0, 0, 0)):
        return ref_lib().ref_scene_add_obj(self.h, os.fsencode(path), int(has_vt), _f3(color), _f3(offset))

    @property
    def num_triangles(self):
        return ref_lib().ref_scene_num_triangles(self.h)

    def triangles(self):
        out = np.empty((self.num_triangles, 9), np.float32)
        ref_lib().ref_scene_get_triangles(self.h, out.ctypes.data)
        return out

    def build_grid(self):
        self.grid_ms = ref_lib().ref_scene_build_grid(self.h)
        return self.grid_ms

    def grid_info(self):
        n = (C.c_int32 * 3)()
        lohi = (C.c_float * 6)()
        w = (C.c_float * 3)()
        iw = (C.c_float * 3)()
        ref_lib().ref_grid_info(self.h, n, lohi, w, iw)
        return dict(n=np.array(n[:], dtype=np.int32), lohi=np.array(lohi[:], dtype=np.float32),
                    width=np.array(w[:], dtype=np.float32), inv_width=np.array(iw[:], dtype=np.float32))

    def grid_csr(self):
        n = self.grid_info()["n"]
        ncell = int(n[0]) * int(n[1]) * int(n[2])
        start = np.empty(ncell + 1, np.uint32)
        r = ref_lib().ref_grid_csr(self.h, start.ctypes.data, None)
        tris = np.empty(max(r, 1), np.uint32)
        ref_lib().ref_grid_csr(self.h, start.ctypes.data, tris.ctypes.data)
        return start, tris[:r]

    def render(self, width, height, y0=0, y1=None, cam=(3, 5, 3, 0, 0, 0, 0, -1, 0, 45), want_rgb=True, want_ids=True):
        if self.grid_ms is None:
            self.build_grid()
        y1 = height if y1 is None else y1
        rows = y1 - y0
        rgb = np.empty((rows, width, 3), np.float32) if want_rgb else None
        ids = np.empty((rows, width), np.int32) if want_ids else None
        camv = (C.c_float * 10)(*[float(x) for x in cam])
        ms = ref_lib().ref_render_rows(self.h, width, height, y0, y1, camv,
                                       rgb.ctypes.data if want_rgb else None,
                                       ids.ctypes.data if want_ids else None)
        return rgb, ids, ms
