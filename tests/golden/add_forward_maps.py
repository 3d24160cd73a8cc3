"""Adds the FORWARD map of every recorded stitch step (the one that sizes the canvas, ImageProcess.cpp:206-216, and moves the
features, :226) to tests/golden/golden.json -- "p_fwd" next to the backward map "p" -- by running the reference's Input/
pipelines again under the recorder (oracle/_ref/libref_record.so, hook on updateFeaturesByHomography).  The run must be the
recorded one: every backward map, offset and canvas size is compared with what golden.json already holds.
Run where /root/reference exists:  python tests/golden/add_forward_maps.py"""
import ctypes as C, json, os, re, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
REF_DIR = os.environ.get("REF", "/root/reference")
REF_SO = oracle_lib.REF_SO
rec = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_record.so"), mode=C.RTLD_GLOBAL)
R = oracle_lib.Reference()
path = os.path.join(HERE, "golden.json")
J = json.load(open(path))
for n in (4, 2):
    logp = os.path.join(tempfile.mkdtemp(), "log.txt")
    assert rec.rec_init(REF_SO.encode(), logp.encode(), None) == 0
    R.pipeline(f"{REF_DIR}/Input/", n)
    rec.rec_close()
    steps = J["runs"][str(n)]["steps"]
    cur = {}
    for ln in open(logp).read().strip().split("\n"):
        kv = dict(re.findall(r"(\w+)=([^ ]+)", ln))
        k = int(kv["step"])
        if ln.startswith("warp"):
            assert [float(v) for v in kv["p"].split(",")] == steps[k]["p"] and int(kv["cw"]) == steps[k]["cw"] and int(kv["ch"]) == steps[k]["ch"]
            assert float(kv["offx"]) == steps[k]["offx"] and float(kv["offy"]) == steps[k]["offy"]
        elif ln.startswith("fwd"):
            steps[k]["p_fwd"] = [float(v) for v in kv["p"].split(",")]
            assert float(kv["offx"]) == steps[k]["offx"] and float(kv["offy"]) == steps[k]["offy"]
    assert all("p_fwd" in s for s in steps)
json.dump(J, open(path, "w"), indent=1)
print("forward maps added")
