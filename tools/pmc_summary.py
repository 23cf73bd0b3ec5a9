#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass, csv output) into profiles/<tag>_pmc_summary.json.

    python tools/pmc_summary.py <tag> <dir with SQ counters> <dir with FETCH_SIZE> <dir with WRITE_SIZE>

Per (kernel, grid size): launches, mean cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs -> /8), MFMA utilisation =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles), wait fractions, and HBM bytes per launch: FETCH_SIZE (KB) x 2 (gfx950
counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE (KB)."""
import collections
import csv
import json
import os
import statistics as st
import sys

tag, d_sq, d_f, d_w = sys.argv[1:5]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def agg(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    f = [x for x in os.listdir(d) if x.endswith("counter_collection.csv")][0]
    for x in csv.DictReader(open(os.path.join(d, f))):
        k = x["Kernel_Name"].split("(")[0].replace("void ", "").replace("es::", "")
        acc[(k, int(x["Grid_Size"]))][x["Counter_Name"]].append(float(x["Counter_Value"]))
    return acc


# the chain kernels are instantiations of two dispatcher templates (point_fwd.hip / point_bwd.hip): <B0, B1> = (tail body, main body)
# body ids: fwd 1 deform (value + J d), 2 sdf, 3 colour, 4 vjp, 5 sdf+vjp (tail); bwd 1 colour, 2 sdf, 3 deform, 4 tan, 5 tan+sdf (tail)
LOGICAL = {"k_point_fwd<0, 1>": "k_deform_fwd", "k_point_fwd<5, 1>": "k_deform_fwd", "k_point_fwd<0, 2>": "k_sdf_fwd",
           "k_point_fwd<0, 3>": "k_color_fwd", "k_point_fwd<2, 3>": "k_color_fwd", "k_point_bwd<2, 1>": "k_color_bwd", "k_point_fwd<0, 4>": "k_deform_vjp", "k_point_bwd<0, 1>": "k_color_bwd",
           "k_point_bwd<0, 2>": "k_sdf_bwd", "k_point_bwd<0, 3>": "k_deform_bwd", "k_point_bwd<5, 3>": "k_deform_bwd",
           "k_point_bwd<0, 4>": "k_deform_tan", "k_wgrad<0>": "k_wgrad[deform]", "k_wgrad<1>": "k_wgrad[sdf]", "k_wgrad<2>": "k_wgrad[color]",
           "k_wgrad<0, false>": "k_wgrad[deform]", "k_wgrad<1, false>": "k_wgrad[sdf]", "k_wgrad<2, false>": "k_wgrad[color]",
           "k_query_sdf_x3r<true>": "k_query_sdf_x3", "k_query_sdf_x3r<false>": "k_query_sdf_x3", "k_deform_jvp_x3r": "k_deform_fwd_x3",
           "k_deform_vjp_x3r": "k_deform_vjp_x3", "k_sdf_fwd_x3r<true, true>": "k_sdf_fwd_x3", "k_sdf_fwd_x3r<true, false>": "k_sdf_fwd_x3",
           "k_sdf_fwd_x3r<false, true>": "k_sdf_fwd_x3", "k_sdf_fwd_x3r<false, false>": "k_sdf_fwd_x3", "k_color_fwd_x3r<true>": "k_color_fwd_x3",
           "k_color_fwd_x3r<false>": "k_color_fwd_x3",
           # round 3: the training chain of the family (SAVE instantiations of the above + csrc/train_x3r.hip)
           "k_deform_jvp_x3r<true>": "k_deform_fwd_x3", "k_deform_jvp_x3r<false>": "k_deform_fwd_x3", "k_deform_vjp_x3r<true>": "k_deform_vjp_x3",
           "k_deform_vjp_x3r<false>": "k_deform_vjp_x3", "k_color_fwd_x3r<true, true>": "k_color_fwd_x3", "k_color_fwd_x3r<true, false>": "k_color_fwd_x3",
           "k_color_fwd_x3r<false, true>": "k_color_fwd_x3", "k_color_fwd_x3r<false, false>": "k_color_fwd_x3", "k_deform_tan_x3r": "k_deform_tan_x3",
           "k_deform_bwd_x3r": "k_deform_bwd_x3", "k_color_bwd_x3r<true>": "k_color_bwd_x3", "k_color_bwd_x3r<false>": "k_color_bwd_x3",
           "k_wgrad_x3<0, false>": "k_wgrad_x3[deform]", "k_wgrad_x3<1, false>": "k_wgrad_x3[sdf]", "k_wgrad_x3<2, false>": "k_wgrad_x3[color]"}
for _d in (True, False):
    for _c in (True, False):
        for _s in (True, False):
            LOGICAL["k_sdf_fwd_x3r<%s, %s, %s>" % tuple(str(v).lower() for v in (_d, _c, _s))] = "k_sdf_fwd_x3"
a, f, w = agg(d_sq), agg(d_f), agg(d_w)
out = []
for key in sorted(a, key=lambda k: -sum(a[k].get("GRBM_GUI_ACTIVE", [0]))):
    c = a[key]
    if not key[0].startswith("k_") or "GRBM_GUI_ACTIVE" not in c:
        continue
    m = lambda n: st.mean(c[n]) if n in c else float("nan")
    cyc = m("GRBM_GUI_ACTIVE") / 8
    fetch = st.mean(f.get(key, {}).get("FETCH_SIZE", [float("nan")])) * 1024 * 2
    write = st.mean(w.get(key, {}).get("WRITE_SIZE", [float("nan")])) * 1024
    out.append(dict(kernel=key[0], logical=LOGICAL.get(key[0], key[0].split("<")[0]), grid_threads=key[1], launches=len(c["GRBM_GUI_ACTIVE"]), cycles=round(cyc),
                    mfma_util=round(m("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc), 4),
                    wait_any=round(m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES"), 4), wait_inst_any=round(m("SQ_WAIT_INST_ANY") / m("SQ_WAVE_CYCLES"), 4),
                    hbm_fetch_bytes=round(fetch), hbm_write_bytes=round(write), hbm_bytes=round(fetch + write)))
path = os.path.join(REPO, "profiles", f"{tag}_pmc_summary.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
for o in out[:12]:
    print(o)
