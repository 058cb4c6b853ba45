#!/usr/bin/env python3
"""GPU timing probe (development tool): BASELINE configs[4] alone - the 10 min audio in 20 VAD chunks with the temperature ladder forced
once, greedy and beam = 5 (bench.long_audio_config), without the rest of the bench.   python tools/time_beam.py"""
import argparse, json, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (bench.py's process set-up, see tools/time_decode.py)
import bench

box = {}
th = threading.Thread(target=lambda: box.update(out=bench.long_audio_config(argparse.Namespace(sample_length=224), 0)))
th.start(); th.join()      # worker thread, as bench.py drives its sessions
out = box["out"]
b = out["beam5_no_reference_behaviour"]
print(json.dumps({"xatt_beam_shared": os.environ.get("WH_XATT_BEAM_SHARED", "1"), "greedy_audio_s_per_s": out["value"], "greedy_seconds": out["seconds"],
                  "beam5_audio_s_per_s": b["value"], "beam5_seconds": b["seconds"], "beam5_over_greedy": round(b["value"] / out["value"], 3)}))
