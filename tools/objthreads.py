"""Time of the OBJ reader on the 88.6 MB plane file per number of parsing threads (RT_OBJ_THREADS). CPU only."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
path = "/tmp/rt_plane100_%d.obj" % os.getpid()
if not os.path.exists(path): rt.make_plane_obj(path, 100)
for th in ("1", "2", "4", "8", "16"):
    os.environ["RT_OBJ_THREADS"] = th
    best = 9
    for _ in range(3):
        s = rt.Scene(); t = time.time(); n = s.load_mesh(path); best = min(best, time.time() - t); s.close()
    print("threads", th, "%.3f s" % best, n)
os.remove(path)
