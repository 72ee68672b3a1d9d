"""Host-side C companions (include/trt_host.h) against vectors taken from the reference's own
functions: camera orbit (TRT.c:1327-1336), read_ppm/load_skybox (TRT.c:309-427) and the
screenbuffer of buffered_draw_screen (TRT.c:1142-1172)."""
import os
import re
import subprocess
import zlib

import numpy as np
import pytest

import support as T
from terminalraytracer_amd import hip, host


def test_header_and_library_agree_on_host_symbols():
    text = open(os.path.join(T.ROOT, "include", "trt_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(trt_[a-z0-9_]+)\s*\(", text)))
    assert set(declared) == set(host.HOST_SYMBOLS), set(declared) ^ set(host.HOST_SYMBOLS)
    dll = hip.lib()
    for name in declared:
        assert hasattr(dll, name)


def test_orbit_camera_matches_reference():
    d = np.load(os.path.join(T.GOLDEN, "cameras.npz"))
    for t, want in zip(d["t"], d["camera"]):
        got = host.orbit_camera(float(t))
        # sin/cos come from the host libm: allow 1 ulp there, everything else is exact arithmetic
        ulps = np.abs(got.view(np.int64) - want.view(np.int64))
        assert ulps.max() <= 1, (t, got, want)
    # and the Python mirror used by the bench builds the same camera
    from terminalraytracer_amd import scenes as S
    py = S.orbit_camera(1.0, 480, 280, reference_aspect=True)
    assert np.abs(py.view(np.int64) - d["camera"][2].view(np.int64)).max() <= 1


def _unpack_sky(tmp_path, name):
    dst = tmp_path / name
    dst.mkdir()
    for f in T.FACES:
        (dst / (f + ".ppm")).write_bytes(T.golden_ppm_raw(name, f))
    return str(dst)


@pytest.mark.parametrize("name", ["colors", "uv_checker"])
def test_skybox_loader_matches_reference_decode(tmp_path, name):
    sky = host.load_skybox(_unpack_sky(tmp_path, name))
    meta = T.golden_meta()["skybox"][name]
    assert sky.shape == (6, meta["dim"], meta["dim"], 3)
    assert T.fnv(sky) == meta["decoded_fnv"]  # bytes the reference's read_ppm produced
    assert np.array_equal(sky, T.sky(name))


def test_skybox_loader_reports_errors_instead_of_exiting(tmp_path):
    good = _unpack_sky(tmp_path, "colors")
    with pytest.raises(OSError) as e:
        host.load_skybox(str(tmp_path / "missing"))
    assert e.value.errno == -101
    raw = T.golden_ppm_raw("colors", "+X")
    cases = {"magic": (b"P5" + raw[2:], -102), "maxval": (raw.replace(b"\n255\n", b"\n65535\n", 1), -103),
             "truncated": (raw[:-10], -107), "garbage_header": (b"P6\nabc def\n255\n", -102)}
    for label, (blob, code) in cases.items():
        d = tmp_path / label
        d.mkdir()
        for f in T.FACES:
            (d / (f + ".ppm")).write_bytes(raw)
        (d / "-Y.ppm").write_bytes(blob)
        with pytest.raises(OSError) as e:
            host.load_skybox(str(d))
        assert e.value.errno == code, label
    d = tmp_path / "shape"
    d.mkdir()
    for f in T.FACES:
        (d / (f + ".ppm")).write_bytes(raw)
    (d / "+Z.ppm").write_bytes(b"P6\n2 1\n255\n" + bytes(6))
    with pytest.raises(OSError) as e:
        host.load_skybox(str(d))
    assert e.value.errno == -105
    assert host.load_skybox(good).shape[0] == 6  # still usable afterwards


def test_emitter_bytes_match_reference_screenbuffer():
    meta = T.golden_meta()["emitter"]["demo_160x48_b4"]
    case = next(c for c in T.golden_cases() if c["name"] == "demo_160x48_b4")
    fb = T.golden_fb(case)
    em = host.Emitter(160, 48)
    em.patch(fb)
    got = em.bytes()
    want = zlib.decompress(open(os.path.join(T.GOLDEN, "emit_demo_160x48_b4.bin.z"), "rb").read())
    assert len(got) == meta["bytes"] == 8 + (25 * 160 + 1) * 48 + 1
    assert got == want
    assert T.fnv(np.frombuffer(got, dtype=np.uint8)) == meta["fnv"]
    # the same cells from already-quantised bytes (what trt_quantize_device hands over)
    em2 = host.Emitter(160, 48)
    em2.patch_rgb8(T.oracle_rgb8(fb))
    assert em2.bytes() == want


def test_emitter_480x280_hash():
    meta = T.golden_meta()["emitter"]["demo_480x280_b10"]
    case = next(c for c in T.golden_cases(("medium",)) if c["name"] == "demo_480x280_b10")
    px, _ = T.oracle_render(T.golden_scene(case), 480, 280, 10, 10)
    em = host.Emitter(480, 280)
    em.patch(px)
    got = em.bytes()
    assert len(got) == meta["bytes"] == 3360289
    assert T.fnv(np.frombuffer(got, dtype=np.uint8)) == meta["fnv"]


@pytest.mark.skipif(not os.path.exists("/root/reference/TerminalRayTracer.c"), reason="reference not present (GPU box)")
def test_reference_main_links_against_the_drop_in(tmp_path):
    """The integration claim of INTEGRATION.md, checked at link level: the reference's own file with
    ONLY the body of project_scene removed (declaration kept) links against libtrt_hip.so."""
    src = open("/root/reference/TerminalRayTracer.c").read()
    start = src.index("void project_scene(Scene *scene, Screen *screen)")
    end = src.index("//function to set all pixels in the screen to a given color")
    patched = src[:start] + "void project_scene(Scene *scene, Screen *screen); /* now provided by libtrt_hip.so */\n\n" + src[end:]
    exe = tmp_path / "trt_ref_on_gpu"
    cmd = ["gcc", "-x", "c", "-", "-O2", "-w", "-o", str(exe), "-L" + os.path.join(T.ROOT, "terminalraytracer_amd"),
           "-ltrt_hip", "-lm", "-Wl,-rpath," + os.path.join(T.ROOT, "terminalraytracer_amd")]
    subprocess.run(cmd, input=patched.encode(), check=True)
    out = subprocess.run(["nm", "-u", str(exe)], capture_output=True, text=True, check=True).stdout
    assert "project_scene" in out  # resolved from the shared library, not from the file


def test_host_c_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """csrc/host/*.c (the PPM / cubemap loader parses files it does not control) compiled with -fsanitize=address,undefined and
    driven by tests/host_sanitize.c through good, malformed and hostile inputs: every status code as documented, no out-of-bounds
    access, no leak on an error path (LeakSanitizer), no undefined arithmetic."""
    exe = str(tmp_path / "host_sanitize")
    sources = [os.path.join(T.ROOT, "tests", "host_sanitize.c")] + sorted(
        os.path.join(T.ROOT, "terminalraytracer_amd", "csrc", "host", f)
        for f in os.listdir(os.path.join(T.ROOT, "terminalraytracer_amd", "csrc", "host")) if f.endswith(".c"))
    build = subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-fno-omit-frame-pointer", "-Wno-format-truncation", "-I" + os.path.join(T.ROOT, "include"), "-o", exe] + sources + ["-lm"],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("the sanitizer runtimes are not installed: " + build.stderr.strip().splitlines()[-1])
    assert build.returncode == 0, build.stderr
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    run = subprocess.run([exe, str(scratch)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("ok "), run.stdout + run.stderr
