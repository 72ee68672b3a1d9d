"""Stage shares (s_memtime stamps, -DTRT_STAMP=1 build selected through TRT_HIP_LIB) for an arbitrary config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TRT_PRINT_STAMPS"] = "1"
from terminalraytracer_amd import hip, scenes as S
w, h, n, b = (int(x) for x in sys.argv[1:5])
scene = S.synth_scene(n, S.synth_sky(256), S.orbit_camera(1.0, w, h))
with hip.Context(0) as ctx:
    ctx.set_scene(scene)
    ctx.enable_counters(True)
    ctx.render_host(scene.camera, hip.RowSet.whole(w, h), b, 10)
    print("counters", ctx.read_counters(), ctx.read_diagnostics())
