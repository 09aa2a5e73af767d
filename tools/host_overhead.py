import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import Tracer
from pathtracer_amd.scenes import SceneData
for timing in (True, False):
    sc = SceneData("world")
    t = Tracer(0)
    sc.upload(t)
    t.target_create(sc.width, sc.height)
    t.enable_timing(timing)
    for s in range(1, 50):
        t.dispatch(s, 1, 1)
    t.finish()
    t0 = time.perf_counter()
    for s in range(50, 2050):
        t.dispatch(s, 1, 1)
    t1 = time.perf_counter()
    t.finish()
    t2 = time.perf_counter()
    print(f"timing={timing}: issue {1e6*(t1-t0)/2000:.1f} us per dispatch, total {1e6*(t2-t0)/2000:.1f} us per frame")
    t0 = time.perf_counter()
    for s in range(2050, 3050):
        t.dispatch(s, 1, 1)
        t.finish()
    t2 = time.perf_counter()
    print(f"timing={timing}: with finish per frame {1e6*(t2-t0)/1000:.1f} us per frame")
    t.close()
