"""How long the eye's two family tables take to build (they are rebuilt whenever the camera moves): tiny frames from a moving
camera, to be run under rocprofv3 --kernel-trace --stats (tools/archive/gpu_eye_build.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from terminalraytracer_amd import hip, scenes as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene = S.synth_scene(n, S.synth_sky(64), S.orbit_camera(1.0, 32, 18))
with hip.Context(0) as ctx:
    ctx.set_scene(scene)
    for f in range(40):
        ctx.render_host(S.orbit_camera(1.0 + f / 60.0, 32, 18), hip.RowSet.whole(32, 18), 1, 1)
    print("scene", ctx.scene_info())
