#!/usr/bin/env python3
"""Derive profiles/pmc_traffic.json from the PMC passes of tools/pmc_walk.sh (gpurun_out/pmc_walk_<cfg>.log, or the committed
profiles/r03_<cfg>_pmc.txt): per evaluation, HBM bytes of EVERY kernel of a step (transition matrices + partials + integration;
FETCH_SIZE and WRITE_SIZE are in KiB per dispatch, FETCH_SIZE x 2 on gfx950 as MI355X_MICROARCH.md prescribes) and the matrix-core
flops issued (SQ_VALU_MFMA_BUSY_CYCLES x 64 flop per cycle: v_mfma_f32_32x32x2_f32 = 4096 flop in 64 cycles).
bench.py reads the file for roofline.traffic / frac.      usage: pmc_traffic.py <dir with pmc logs> <round tag>"""
import ast
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
out = {"_comment": "HBM bytes and issued matrix-core flops per evaluation, ALL kernels of a step, from rocprofv3 --pmc passes (tools/pmc_walk.sh; "
                   "raw per-dispatch averages in profiles/%s_*_pmc.txt).  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE x 2 on gfx950 (128-byte "
                   "requests tallied at 64 B, MI355X_MICROARCH.md); WRITE_SIZE as is.  mfma_issued_gflop = SQ_VALU_MFMA_BUSY_CYCLES x 64 flop "
                   "(v_mfma_f32_32x32x2_f32: 4096 flop per 64 busy cycles).  Deterministic per workload; written by tools/pmc_traffic.py." % tag}
for cfg in ("c2", "c3", "c4", "c5"):
    path = next((p for p in (os.path.join(src, "pmc_walk_%s.log" % cfg), os.path.join(src, "%s_%s_pmc.txt" % (tag, cfg))) if os.path.exists(p)), None)
    if path is None:
        continue
    kernels = {}
    for line in open(path):
        m = re.match(r"(.+?) (\{.*?\}) dispatches (\d+) sum", line)
        if not m:
            continue
        name = m.group(1).strip()
        vals = ast.literal_eval(m.group(2))
        kernels.setdefault(name, {}).update(vals)
    fetch = sum(v.get("FETCH_SIZE", 0.0) for v in kernels.values())
    write = sum(v.get("WRITE_SIZE", 0.0) for v in kernels.values())
    busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k, v in kernels.items() if "walkg" in k or "partials" in k)
    e = {"kernels": sorted(k[:60] for k in kernels), "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
         "traffic_bytes": int((2 * fetch + write) * 1024), "source": "profiles/%s_%s_pmc.txt" % (tag, cfg)}
    insts = sum(v.get("SQ_INSTS_MFMA", 0.0) for k, v in kernels.items() if "walkg" in k or "partials" in k)
    if busy > 0 and insts > 0 and busy / insts < 48:
        # the walk issues v_mfma_f32_32x32x16_bf16 (32 busy cycles each; six of them -- three bf16 pieces per operand -- do the work of
        # eight v_mfma_f32_32x32x2_f32): the matrix work counted as what the fp32 matrix path would issue for it, and the 16-bit pipe's own time
        e["mfma_bf16_busy_cycles"] = busy
        e["mfma_bf16_instructions"] = insts
        e["mfma_issued_gflop"] = insts * (8.0 / 6.0) * 4096 / 1e9
        e["mfma_issued_note"] = "fp32-equivalent: bf16 MFMA instructions x 8/6 x 4096 flop"
    elif busy > 0:
        e["mfma_busy_cycles"] = busy
        e["mfma_issued_gflop"] = busy * 64 / 1e9
    # where the partials kernel's wave-cycles go (pass 1 of tools/pmc_walk.sh): issuing an instruction / stalled at issue (most of it behind
    # the wave's own previous instruction: a dependent MFMA or VALU chain) / in s_waitcnt for memory or LDS
    for k, v in kernels.items():
        if ("walkg" in k or "walk4" in k) and v.get("SQ_WAVE_CYCLES", 0.0) > 0:
            wc = v["SQ_WAVE_CYCLES"]
            e["partials_kernel_wave_cycles"] = {"issuing": round(v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4), "stalled_at_issue": round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                                                "waiting_s_waitcnt": round(v.get("SQ_WAIT_ANY", 0.0) / wc, 4), "waves": v.get("SQ_WAVES", 0.0)}
    out[cfg] = e
with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({k: (v if k == "_comment" else {a: b for a, b in v.items() if a != "kernels"}) for k, v in out.items() if k != "_comment"}, indent=1))
