import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slam_helpers import Pair
pr = Pair(20, 40, batch=3)
pr.engine.set_search_variant(1)
for k in range(40):
    pr.step_both(k, save_trajectory=True)
    bad = False
    for b in range(3):
        o = pr.oracles[b]
        feats = pr.engine.features(b)
        for i, fe in enumerate(feats):
            fo = o.feature(i)
            if fe["selected"] and (fe["success"] != fo["success"] or (fo["success"] and not np.array_equal(fe["z"], fo["z"]))):
                print("frame", k, "seq", b, "feature", i, "engine", fe["success"], fe["z"], "oracle", fo["success"], fo["z"], "h", fo["h"], "S", fo.get("S"))
                bad = True
    if bad:
        break
print("done", k)
