#!/usr/bin/env python3
"""HBM bytes per launch per kernel kind from the two rocprofv3 PMC passes (tools/profile_gpu.sh): FETCH_SIZE and WRITE_SIZE are KB
per dispatch; FETCH_SIZE is doubled (gfx950 tallies a 128-byte request of a wide coalesced read as 64 B - MI355X_MICROARCH.md,
HBM section); WRITE_SIZE is used as reported.  Writes the JSON bench.py reads for `roofline.traffic`.

    python tools/pmc_traffic.py gpurun_out/r02f_pmc_FETCH_SIZE.csv gpurun_out/r02f_pmc_WRITE_SIZE.csv large-v3 32 > profiles/r02_pmc_traffic.json
"""
import csv
import json
import os
import re
import sys

KIND = [  # (regex on the kernel name, kernel kind of bench.py)
    (r"dec_cross_attn_kernel", "dec_cross_attn"), (r"xabs_attn_kernel", "dec_cross_attn"), (r"xabs_qk_kernel", "dec_xabs_qk"),
    (r"xabs_vup_kernel", "dec_xabs_vup"), (r"dec_self_attn_kernel", "dec_self_attn"),
    (r"dec32_proj_kernel<0,", "dec_proj_qkv"), (r"dec32_proj_kernel<1,", "dec_proj_cq"),
    (r"dec32_proj_kernel<2, true", "dec_proj_resid_avg"), (r"dec32_proj_kernel<2, false", "dec_proj_fc2"),
    (r"dec32_proj_kernel<3,", "dec_proj_fc1"), (r"dec32_proj_kernel<4,", "dec_proj_logits"),
    (r"dec32_embed_kernel", "dec_embed"), (r"sampler_final_kernel", "sampler"),
    (r"gemm256_kernel<1>", "gemm_enc_fc1"), (r"gemm256_kernel<3>", "gemm_enc_qkv"), (r"gemm256_kernel<7>", "gemm_cross_kv"),
    (r"gemm256_kernel<5>", "gemm_conv2"), (r"gemm256_kernel<4>", "gemm_conv1"), (r"gemm256_kernel<2>", "gemm_enc_o+fc2"),
    (r"encoder_attention(_v2)?_kernel", "enc_attention"), (r"layernorm_kernel", "layernorm"), (r"mel_power_kernel", "mel_power"),
    (r"mel_finalize_kernel", "mel_finalize"),
]


def read(path):
    out, resid = {}, []
    for row in csv.DictReader(open(path)):
        m = re.search(r"dec32_proj_kernel<2, true.*@grid(\d+)", row["Kernel"])
        if m:       # (round 5) tools/rocpd_pmc.py keys the shared RESID instantiation by its launch grid: the largest grid is fc2 (K split 4 ways)
            resid.append((int(m.group(1)), float(row["AvgValue"]) * 1024.0, int(row["Dispatches"])))
            continue
        for pat, kind in KIND:
            if re.search(pat, row["Kernel"]):
                out[kind] = (float(row["AvgValue"]) * 1024.0, int(row["Dispatches"]))
                break
    if len(resid) >= 2:
        resid.sort()
        out["dec_proj_fc2"] = resid[-1][1:]
        n = sum(r[2] for r in resid[:-1])
        out["dec_proj_oproj"] = out["dec_proj_coproj"] = (sum(r[1] * r[2] for r in resid[:-1]) / n, n)
    elif resid:
        out["dec_proj_resid_avg"] = resid[0][1:]
    return out


def main():
    fetch, write = read(sys.argv[1]), read(sys.argv[2])
    model, B = sys.argv[3], int(sys.argv[4])
    bpl, detail = {}, {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, (0.0, 0))[0], write.get(k, (0.0, 0))[0]
        bpl[k] = int(round(2.0 * f + w))
        detail[k] = {"fetch_size_kb": round(f / 1024, 1), "write_size_kb": round(w / 1024, 1), "dispatches": fetch.get(k, write.get(k))[1]}
    # Round 3: fc2 reads an f16 hi | lo plane pair like the two out projections, so the three launches of a layer share ONE instantiation
    # and the per-kernel-name average mixes them.  The out projections are unchanged since the round-2 pass that saw them alone
    # (profiles/r02_pmc_traffic.json, optional 5th argument): they keep that figure, fc2 = 3 x average - 2 x that.
    avg = bpl.pop("dec_proj_resid_avg", None)
    if avg is not None and "dec_proj_fc2" not in bpl:
        prev = json.load(open(sys.argv[5]))["bytes_per_launch"]["dec_proj_oproj"] if len(sys.argv) > 5 else None
        if prev:
            bpl["dec_proj_oproj"] = bpl["dec_proj_coproj"] = prev
            bpl["dec_proj_fc2"] = 3 * avg - 2 * prev
        detail["dec_proj_resid_avg"]["note"] = "average over the oproj, coproj and fc2 launches (one instantiation)"
        detail["dec_proj_resid_avg"]["bytes_per_launch_avg"] = avg
    elif avg is not None:
        bpl["dec_proj_oproj"] = bpl["dec_proj_coproj"] = avg
    json.dump({"config": f"whisper-{model}, {B} chunks per step (tools/pmc_run.py, eager launches, " + os.environ.get("WH_PMC_STEPS", "16") + " decoder steps from position 0: the positions bench.py's algorithmic bytes assume)",
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (each with --kernel-trace only); KB per "
                         "dispatch averaged over all dispatches of the kernel; bytes = 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE doubled per the "
                         "gfx950 correction of MI355X_MICROARCH.md; Infinity-Cache hits are counted, so this is traffic at the L2's memory side)",
               "model": model, "chunks_per_step": B, "cross_attention_splits": int(os.environ.get("WH_XABS_SPLITS", "4")), "bytes_per_launch": bpl, "counters": detail}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
