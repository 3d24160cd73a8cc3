"""Adds golden vectors of the `src/ex6` variant's whole blend (src/ex6/ImageProcess.cpp:638-742) to golden.json
("blend_ex6": input recipe + SHA-256 of the bytes the variant's own compiled function returned,
oracle/_ref/libref6_hotpath.so).  Run where /root/reference exists:  python tests/golden/add_ex6_goldens.py"""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests"))
import numpy as np
import oracle_lib
O, R6 = oracle_lib.Oracle(), oracle_lib.ReferenceEx6()
path = os.path.join(HERE, "golden.json")
J = json.load(open(path))
J["blend_ex6"] = []
for (w, h, fa, fb) in [(67, 33, 5, 6), (270, 131, 1, 2), (100, 64, 7, 8), (33, 67, 9, 10), (512, 512, 3, 4), (600, 800, 13, 14), (1081, 527, 15, 16)]:
    for a_left in (True, False):
        A, B = O.synth(w, h, fa), O.synth(w, h, fb)
        if a_left:
            A[:, :, (2 * w) // 3:] = 0
            B[:, :, : w // 3] = 0
        else:
            A[:, :, : w // 3] = 0
            B[:, :, (2 * w) // 3:] = 0
        out = R6.blend(A, B)
        J["blend_ex6"].append({"w": w, "h": h, "fa": fa, "fb": fb, "a_left": a_left,
                               "out_sha256": hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest()})
json.dump(J, open(path, "w"), indent=1)
print(len(J["blend_ex6"]), "vectors")
