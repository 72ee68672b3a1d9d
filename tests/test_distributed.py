"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the row-tile shard / gather / assembly
logic; rank 0 must end up with the reference's framebuffer bit for bit."""
import os
import socket
import subprocess
import sys

import pytest

import support as T


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,tile_rows,case", [(2, 8, "demo_160x48_b4"), (3, 5, "demo_67x13_b2_s3"),
                                                  (2, 1, "demo_1x9_b4")])
def test_sharded_frame_over_gloo(world, tile_rows, case):
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(T.ROOT, "tests", "dist_worker.py"), case, str(tile_rows)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert f"DIST_OK world={world} tile_rows={tile_rows} {case}" in out.stdout
