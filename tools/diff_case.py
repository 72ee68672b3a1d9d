import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import support as T
from terminalraytracer_amd import hip
name = sys.argv[1]
case = next(c for c in T.golden_cases(("small","medium","large")) if c["name"] == name)
scene = T.golden_scene(case)
w,h,b,s = case["width"], case["height"], case["bounce_limit"], case["rays_per_pixel"]
with hip.Context(0) as ctx:
    ctx.set_scene(scene)
    ctx.set_kernel(1); ref = ctx.render_host(scene.camera, hip.RowSet.whole(w,h), b, s)
    ctx.set_kernel(0); got = ctx.render_host(scene.camera, hip.RowSet.whole(w,h), b, s)
bad = (ref.view(np.uint64) != got.view(np.uint64)).any(axis=2)
print(name, "pixels differing:", int(bad.sum()), "of", w*h, "max abs diff", float(np.abs(ref-got).max()))
ys, xs = np.nonzero(bad)
print("first:", list(zip(ys[:10].tolist(), xs[:10].tolist())))
if bad.sum():
    print("rows hist (16 bins):", np.histogram(ys, bins=16, range=(0,h))[0].tolist())
    y,x = ys[0], xs[0]; print("ref", ref[y,x], "got", got[y,x])
