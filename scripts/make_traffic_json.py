#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the per-pass counter summaries scripts/profile_round.sh leaves in
gpurun_out/prof: per-launch HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md: FETCH_SIZE
counts 64 B for a 128-B request on gfx950 -> x2, an upper bound for narrow loads; WRITE_SIZE is exact)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def load(name):
    out = {}
    f = os.path.join(P, name + "_summary.csv")
    if not os.path.exists(f):
        return out
    for r in csv.DictReader(open(f)):
        out.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean"])
    return out


def label(k):
    """rocprof kernel signature -> the name bench.py's in-library profile uses"""
    m = re.match(r"void k_conv_halo<(\d+), (\d+), 5, (\d), \d+, (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), \d+, (\d+), (true|false)", k)
    if m:
        cin, cph, kw, th, tw, nimg, wm, wn, tm, tn, np_, pool = m.groups()
        cin = int(cph)
        layer = {(4, "false"): "conv2_1", (32, "true"): "conv2_2", (32, "false"): "conv3_1", (64, "true"): "conv3_2",
                 (64, "false"): "conv4_1", (128, "true"): "conv4_2"}[(cin, pool)]
        return "k_conv_halo<%s>" % layer
    m = re.match(r"void k_conv_halo_cc<3, 32, (\d+), (\d+), (\d+)", k)
    if m:
        return "k_conv_halo_cc<tile %sx%s x%s images>" % m.groups()
    if "k_warp_net" in k or "k_warp_quad" in k:
        return "k_warp_net"
    if "k_composite" in k:
        return "k_composite"
    return None


res = {"note": __doc__.strip().replace("\n", " "), "kernels": {}}
for wl, images in (("gtsrb", 32000), ("trhits", 8000), ("wrn", 1024)):
    fe, wr, sq = load("fetch_" + wl), load("write_" + wl), load("sq_" + wl)
    tcc, tcp, ta = load("tcc_" + wl), load("tcp_" + wl), load("ta_" + wl)
    for k in fe:
        name = label(k)
        if not name or (wl == "trhits" and name.startswith("k_conv")):
            continue
        f_kb, w_kb = fe[k].get("FETCH_SIZE", 0.0), wr.get(k, {}).get("WRITE_SIZE", 0.0)
        e = {"workload": {"gtsrb": "scripts/tune_conv.py 32000", "trhits": "scripts/prof_tr_hits.py (16 x 1000, chunks of 8192 / 7808)",
                          "wrn": "scripts/tune_wrn.py 1024"}[wl],
             "images_per_launch": images, "fetch_size_kb": f_kb, "write_size_kb": w_kb,
             "hbm_bytes_raw": (f_kb + w_kb) * 1024, "hbm_bytes_corrected": (2 * f_kb + w_kb) * 1024}
        if k in sq:
            e["mfma_busy_cycles"] = sq[k].get("SQ_VALU_MFMA_BUSY_CYCLES")
            e["sq_busy_cycles"] = sq[k].get("SQ_BUSY_CYCLES")
            if e["mfma_busy_cycles"] and e["sq_busy_cycles"]:
                e["mfma_busy_over_32_sq_busy"] = e["mfma_busy_cycles"] / (32.0 * e["sq_busy_cycles"])
        if k in tcc:
            h, m_ = tcc[k].get("TCC_HIT_sum", 0.0), tcc[k].get("TCC_MISS_sum", 0.0)
            e["l2_hit_rate"] = h / (h + m_) if h + m_ else None
        if k in tcp:
            e["tcp_tcc_read_req"] = tcp[k].get("TCP_TCC_READ_REQ_sum")
            e["l2_to_l1_bytes_at_64B_per_req"] = 64.0 * e["tcp_tcc_read_req"] if e["tcp_tcc_read_req"] else None
        if k in ta:
            e["ta_busy_sum"], e["grbm_gui_active"] = ta[k].get("TA_TA_BUSY_sum"), ta[k].get("GRBM_GUI_ACTIVE")
            e["ta_flat_read_wavefronts"] = ta[k].get("TA_FLAT_READ_WAVEFRONTS_sum")
        res["kernels"][name] = e
out = os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")
json.dump(res, open(out, "w"), indent=1)
print(out)
for k, v in res["kernels"].items():
    print("%-44s %8.1f MB corrected  mfma %s  l2hit %s" % (k, v["hbm_bytes_corrected"] / 1e6, v.get("mfma_busy_over_32_sq_busy"), v.get("l2_hit_rate")))
