"""Time of the OBJ reader (rt_scene_add_obj, the load_mesh counterpart) on the 88.6 MB plane file of plane_mesh_creator at
SQUARES_PER_UNIT=100, beside the oracle's reader (a restatement of the reference's iostream loop in C, not the iostream code
itself) and the scene upload. CPU + GPU: python tools/objtime.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
rt = importlib.import_module("ray-tracer_amd")
import oracle_py as O
path = "/tmp/rt_plane100_%d.obj" % os.getpid()
t = time.time(); rt.make_plane_obj(path, 100); t_gen = time.time() - t
size = os.path.getsize(path)
s = rt.Scene(); t = time.time(); n = s.load_mesh(path); t_load = time.time() - t
o = O.OracleScene(); t = time.time(); n2 = o.add_obj(path); t_oracle = time.time() - t
print("plane SPU=100: %d bytes, %d triangles; rt_make_plane_obj %.2f s; rt_scene_add_obj %.2f s (%.0f MB/s); oracle reader %.2f s" % (size, n, t_gen, t_load, size / t_load / 1e6, t_oracle))
assert n == n2 == 2004002
try:
    t = time.time(); s.upload(); print("rt_scene_upload (H2D, record conversion, grid measurement, record-order choice): %.3f s" % (time.time() - t))
except rt.RtError as e:
    print("no GPU:", e)
os.remove(path)
