#!/usr/bin/env python3
"""Development tool (GPU): minimum beam-ranking margins of tests/test_gpu_configs4_depth.py's scenario for candidate audio seed pairs, in both cross-attention modes.
    python tests/tools/configs4_fixture_search.py "77,78" "79,80" ...
One JSON line per pair: the oracle's minimum margins of the two windows (absorbed / K-V rows) and whether every other check of the test passed."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import test_gpu_configs4_depth as T  # noqa: E402

gen = T.rig.__wrapped__() if hasattr(T.rig, "__wrapped__") else None
if gen is None:
    raise SystemExit("pytest fixture wrapper changed: build the rig by hand")
rig = next(gen)
for spec in sys.argv[1:]:
    seeds = tuple(int(x) for x in spec.split(","))
    rec = {"seeds": seeds}
    for name, mode in (("absorbed", 1), ("kv_rows", 0)):
        try:
            m = T.run_case(rig, 5, mode, seeds)
            rec[name] = {"margins": [round(float(x), 5) for x in m], "checks": "passed"}
        except AssertionError as e:
            rec[name] = {"checks": "FAILED: " + str(e)[:200]}
    print(json.dumps(rec), flush=True)
