#!/usr/bin/env python3
"""Development tool: several bench.py configurations of ONE model in ONE process (the synthetic weights are built once).

    python tools/bench_sweep.py large-v3 "B:F:device_batch:splits[:steps]" ...      (splits -1 = bench.py's automatic choice)

Prints one JSON line per configuration: audio-s/s, ms per step, the session geometry bench.py chose."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

model = sys.argv[1]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
for spec in sys.argv[2:]:
    f = [int(x) for x in spec.split(":")]
    B, F, db, splits = f[:4]
    steps = f[4] if len(f) > 4 else 12
    args = argparse.Namespace(cross_attention_splits=splits, sample_length=224, dist_backend="nccl", serial_reference=False, dump_records=None)
    try:
        o = bench.run_config(args, model, B, F, steps, 3, 1, 0, 0, dev, want_roofline=False, want_cpu=False, device_batch=db)
        print(json.dumps({"spec": spec, "audio_s_per_s": round(o["value"], 1), "ms_per_step": round(o["elapsed"] / steps * 1e3, 2), "slots": o["slots"],
                          "steps_per_batch": o["steps_per_batch"], "inflight": o["inflight"], "cross_attention": o["cross_attention"],
                          "encoder_ms_per_chunk": round(o["stages"]["encoder_ms_per_chunk"], 3),
                          "us_per_decoder_step": round(o["stages"]["us_per_decoder_step"], 1)}), flush=True)
    except Exception as e:   # noqa: BLE001
        print(json.dumps({"spec": spec, "error": str(e)[:300]}), flush=True)
