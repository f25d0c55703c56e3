#!/usr/bin/env python3
"""The rows of DESIGN.md section 7 / README.md from a bench line:   python tools/bench_tables.py profiles/r04_bench_default.json"""
import json
import sys


def main(path):
    with open(path) as fh:
        text = fh.read()
    d = json.loads([l for l in text.splitlines() if l.startswith("{")][-1])
    rows = [("C4 DNA 1000 x 50 000 (`value`)", d)] + [(n, a) for n, a in zip(("C2 DNA 500 x 20 000", "C3 protein 200 x 10 000", "C5 codon M3 100 x 5 000"), d["also"])]
    print("| workload | M updates/s | ms / step (wall) | all kernels, ms | partials kernel, ms | bound | frac | HBM bytes (PMC) | issued MFMA | x reference on one core |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, x in rows:
        r = x["roofline"]
        mf = r.get("mfma_issued_frac")
        print("| %s | %s | %.3f | %.3f | %.3f | %s | **%.2f** | %.2f GB -> %.1f TB/s (%.2f of the box's fill) | %s | %s |" % (
            name, "{:,.0f}".format(x["value"]).replace(",", " "), x["ms_per_step"], r.get("all_kernels_ms_per_step", float("nan")), r.get("partials_kernel_ms_per_step", float("nan")),
            r["bound"], r["frac"], r["traffic"] / 1e9, r["hbm_GBs"] / 1e3, r.get("hbm_frac_of_box_write_stream", float("nan")),
            ("%.2f of 157 TFLOP/s" % mf) if mf else "-", "{:,.0f}".format(x["value"] / x["cpu_baseline"]["value"]).replace(",", " ")))
    print()
    print("box fill GB/s:", d["roofline"].get("box_write_stream_GBs"))
    for x in d.get("double_precision", []):
        print("f64:", x.get("workload", "")[:40], x.get("ms_per_step"), x.get("kernel"))
    m = d.get("mcmc_gen_per_s", {})
    for k in ("default_moves", "fixed_topology", "codon_m3_fixed_topology"):
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in m.get(k, {}).items() if not a.endswith("_ngen")})
    print("mpi_mcmc:", json.dumps(d.get("mpi_mcmc"))[:400])
    print("devices:", d.get("instance_devices"), d.get("ranks_devices"))


if __name__ == "__main__":
    main(sys.argv[1])
