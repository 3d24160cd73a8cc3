"""GPU: the drop-in.  The REFERENCE's own control flow (ImageProcess::ImageProcess -> readFile -> matching: VLFeat
SIFT, kd-tree matching, RANSAC, stitch order, canvas sizing -- oracle/_ref/libref_hotpath.so, compiled from
/root/reference where it lies) runs with its per-pixel functions replaced by the product's C++ adaptor
(computervisionimagestich2_amd/adaptor/cimg_dropin.cpp -> include/stitch.h -> HIP kernels).  Loading the adaptor
with RTLD_GLOBAL ahead of the reference library makes the reference's PLT calls bind to the adaptor's definitions.
The panorama must equal the one the unmodified reference produced (tests/golden), byte for byte -- which also means
the projection fed SIFT/RANSAC bit-identical pixels.

Runs in a fresh interpreter without torch, i.e. against the system ROCm runtime alone, like a C++ user would."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DROPIN = os.path.join(ROOT, "oracle", "_ref", "libstitch_dropin.so")
REF = os.path.join(ROOT, "oracle", "_ref", "libref_hotpath.so")
EXE = os.path.join(ROOT, "oracle", "_ref", "stitch_dropin_main")


def need(*paths):
    """These are -m gpu tests: they run on the GPU box, where the artefacts built next to the reference (make -C oracle ref, in
    the container that has /root/reference) arrive with the snapshot.  A box without them must FAIL here, not report the
    drop-in proof as skipped."""
    missing = [p for p in paths if not os.path.exists(p)]
    assert not missing, f"drop-in artefacts missing (built by `make -C oracle ref` where /root/reference exists): {missing}"


SCRIPT = r'''
import ctypes as C, hashlib, json, sys
import numpy as np
dropin = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)     # first: its definitions win the symbol lookup
ref = C.CDLL(sys.argv[2])
w, h = C.c_int(), C.c_int()
buf = np.zeros(16 << 20, np.uint8)
rc = ref.ref_pipeline((sys.argv[3].rstrip("/") + "/").encode(), int(sys.argv[4]), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(w), C.byref(h))
res = buf[: w.value * h.value * 3]
print("RESULT " + json.dumps({"rc": rc, "w": w.value, "h": h.value, "sha256": hashlib.sha256(res.tobytes()).hexdigest(),
                              "mean": float(res.mean()), "calls": [dropin.stitch_dropin_call_count(i) for i in range(6)]}))
'''


@pytest.mark.parametrize("n", [2, 4])
def test_reference_control_flow_on_hip_kernels(n):
    need(DROPIN, REF)
    J = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    out = subprocess.run([sys.executable, "-c", SCRIPT, DROPIN, REF, os.path.join(HERE, "golden", "input"), str(n)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    r = json.loads(line[7:])
    run = J["runs"][str(n)]
    assert r["rc"] == 0 and [3, r["h"], r["w"]] == run["final_shape"]
    # project and gray once per frame, warp/move/blend once per stitched neighbour, equalise once
    assert r["calls"] == [n, n - 1, n - 1, n - 1, 1, n], r["calls"]
    assert r["sha256"] == run["final_sha256"], (r["mean"], run["final_mean"])


TRANSFER_SCRIPT = r'''
import ctypes as C, hashlib, json, sys
import numpy as np
sys.path.insert(0, sys.argv[2])
import oracle_lib
dropin = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)
O = oracle_lib.Oracle()
src, tem = O.synth(300, 200, 1), O.synth(97, 61, 8)
want, _ = O.transfer(src, tem)
got = src.copy()
rc = dropin.stitch_dropin_transfer_in_place(got.ctypes.data_as(C.c_void_p), 300, 200, tem.ctypes.data_as(C.c_void_p), 97, 61)
print("RESULT " + json.dumps({"rc": rc, "equal": bool(np.array_equal(got, want)), "calls": dropin.stitch_dropin_call_count(6)}))
'''


def test_transfer_class_binding():
    """The reference's `transfer` class (transfer.h; its own transfer.cpp needs the Win32 thread API) constructed exactly
    as ImageProcess.cpp:180 would -- output aliasing the source -- runs on the HIP path and equals the CPU restatement."""
    need(DROPIN)
    out = subprocess.run([sys.executable, "-c", TRANSFER_SCRIPT, DROPIN, HERE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert r == {"rc": 0, "equal": True, "calls": 1}, r


def test_literal_main_cpp_executable(tmp_path):
    """The reference's OWN program: main.cpp (main.cpp:3-11: `ImageProcess ip("../../Input/", 4)`) + ImageProcess.cpp with the
    bodies of toGrayScale, warpingImageByHomography, movingImageByOffset and blendTwoImages excised at build time, linked with
    the adaptor object, VLFeat and -lstitch_hip (make -C oracle dropin-exe).  SIFT, kd-tree matching,
    RANSAC, stitch order and the luminance mix are the reference's compiled code; every per-pixel function is a HIP kernel.
    Run from a directory two levels below a copy of the golden Input/ frames; the panorama it leaves (its two
    result.display() calls are result.save_bmp() in this head-less build) must be the recorded one, byte for byte."""
    need(EXE)
    import hashlib
    import shutil
    import numpy as np
    sys.path.insert(0, ROOT)
    from computervisionimagestich2_amd import bmp
    shutil.copytree(os.path.join(HERE, "golden", "input"), tmp_path / "Input")
    cwd = tmp_path / "build" / "bin"
    cwd.mkdir(parents=True)
    out_bmp = tmp_path / "panorama.bmp"
    env = dict(os.environ, STITCH_DROPIN_OUT=str(out_bmp))
    r = subprocess.run([EXE], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    run = json.load(open(os.path.join(HERE, "golden", "golden.json")))["runs"]["4"]
    img = bmp.load_bmp(str(out_bmp))
    assert list(img.shape) == run["final_shape"]
    assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == run["final_sha256"], float(img.mean())
