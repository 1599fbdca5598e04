#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/profile.sh) into committed evidence under profiles/:
   profiles/<tag>_kernel_stats.csv   the rocprofv3 --kernel-trace --stats table, verbatim
   profiles/<tag>_pmc.json           per-kernel counter sums per launch + derived metrics
   profiles/<tag>_summary.md         the same, readable
and updates profiles/pmc_traffic.json (HBM bytes per launch, read by bench.py's roofline.traffic).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are reported
in KiB... rocprofv3 reports them in kilobytes; on gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced reads, so the corrected figure doubles it; this kernel's reads are 16-B-per-lane gathers
(uncalibrated width), so BOTH the raw and the doubled value are recorded and the doubled one (the upper
bound) is what pmc_traffic.json carries.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    workload_key = sys.argv[2] if len(sys.argv) > 2 else "c4_1920x1080_spp100_n1"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
    stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_stats.csv")))}

    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    meta = {}
    for f in sorted(glob.glob(os.path.join(src, "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_render_tiles" not in k:
                continue
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
            meta[k] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "lds_bytes": int(r["LDS_Block_Size"]),
                       "scratch_bytes": int(r["Scratch_Size"]), "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]),
                       "sgpr": int(r["SGPR_Count"])}
    out = {}
    for k, d in sums.items():
        per = {c: v / max(1, len(launches[k][c])) for c, v in d.items()}
        der = {}
        if "FETCH_SIZE" in per:
            der["hbm_read_bytes_raw"] = per["FETCH_SIZE"] * 1024
            der["hbm_read_bytes_x2_gfx950"] = 2 * per["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in per:
            der["hbm_write_bytes"] = per["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in per:
            der["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
        if "SQ_THREAD_CYCLES_VALU" in per and "SQ_ACTIVE_INST_VALU" in sums[k]:
            a = sums[k]["SQ_ACTIVE_INST_VALU"] / max(1, len(launches[k]["SQ_ACTIVE_INST_VALU"]))
            der["valu_lane_utilisation"] = per["SQ_THREAD_CYCLES_VALU"] / (64.0 * a)
        if "SQ_WAVE_CYCLES" in per:
            wc = per["SQ_WAVE_CYCLES"]
            der["frac_wave_cycles_waiting(SQ_WAIT_ANY)"] = per.get("SQ_WAIT_ANY", 0) / wc
            der["frac_wave_cycles_issue_stall(SQ_WAIT_INST_ANY)"] = per.get("SQ_WAIT_INST_ANY", 0) / wc
            der["frac_wave_cycles_issuing(SQ_ACTIVE_INST_ANY)"] = per.get("SQ_ACTIVE_INST_ANY", 0) / wc
            der["frac_wave_cycles_valu(SQ_ACTIVE_INST_VALU)"] = per.get("SQ_ACTIVE_INST_VALU", 0) / wc
        if "SQ_WAVES" in per and k in meta:
            der["waves_per_launch"] = per["SQ_WAVES"]
            der["resident_waves_per_simd"] = per["SQ_WAVES"] / (256 * 4)
        if "GRBM_GUI_ACTIVE" in per and k in stats:
            der["effective_clock_GHz"] = per["GRBM_GUI_ACTIVE"] / 8 / (float(stats[k]["AverageNs"]))
        if k in stats:
            der["avg_launch_ms(kernel-trace)"] = float(stats[k]["AverageNs"]) / 1e6
            der["launches(kernel-trace)"] = int(stats[k]["Calls"])
        out[k] = {"launch": meta.get(k, {}), "counters_per_launch": per, "derived": der}
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)

    # the kernel the bench line of the SAME run names (roofline.kernel = rt_kernel_name) and the digest of the sources it was
    # built from (roofline.build_id = rt_build_id): the counter record is tied to both, bench.py calls it stale otherwise
    bench_line = {}
    for ln in open(os.path.join(src, "kt.log"), errors="replace"):
        if ln.startswith("{") and '"roofline"' in ln:
            bench_line = json.loads(ln)
    timed_name = bench_line.get("roofline", {}).get("kernel")
    build_id = bench_line.get("roofline", {}).get("build_id")
    if timed_name:
        main_k = [k for k in out if timed_name + "(" in k or k.endswith(timed_name) or (timed_name in k)]
    else:
        main_k = [k for k in out if "<float, false" in k]     # (older logs) the timed variant: FP32 boxes, no STATS counters
    lines = [f"# rocprofv3 summary {tag} — `bench.py` ({workload_key})", "",
             "Source: `tools/profile.sh` on one MI355X; one rocprofv3 run per pass (kernel trace, then PMC passes).", "",
             "## kernel-trace --stats", "", "| kernel | calls | avg ms | % |", "|---|---|---|---|"]
    for n, r in stats.items():
        lines.append(f"| `{n[:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |")
    for k in main_k:
        lines += ["", f"## `{k}` (the timed kernel), per launch", "", f"launch: {out[k]['launch']}", "", "| derived | value |", "|---|---|"]
        for a, b in sorted(out[k]["derived"].items()):
            lines.append(f"| {a} | {b:.6g} |")
        lines += ["", "| counter | per launch |", "|---|---|"]
        for a, b in sorted(out[k]["counters_per_launch"].items()):
            lines.append(f"| {a} | {b:.6g} |")
    open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")

    tf = os.path.join(dst, "pmc_traffic.json")
    traffic = json.load(open(tf)) if os.path.exists(tf) else {}
    for k in main_k:
        d = out[k]["derived"]
        if "hbm_read_bytes_x2_gfx950" in d:
            traffic[workload_key] = {"hbm_bytes_per_launch": int(d["hbm_read_bytes_x2_gfx950"] + d.get("hbm_write_bytes", 0)),
                                     "source": f"profiles/{tag}_pmc.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
                                     "build_id": build_id, "kernel": timed_name}
            # the limiter that matters for this kernel: VALU issue.  SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES is the share of a
            # wave's life spent issuing VALU work; times the resident waves per SIMD = how busy each SIMD's VALU is.
            if "frac_wave_cycles_valu(SQ_ACTIVE_INST_VALU)" in d and "resident_waves_per_simd" in d:
                traffic[workload_key]["valu_busy_frac"] = round(d["frac_wave_cycles_valu(SQ_ACTIVE_INST_VALU)"] * d["resident_waves_per_simd"], 4)
            if "valu_lane_utilisation" in d:
                traffic[workload_key]["valu_lane_utilisation"] = round(d["valu_lane_utilisation"], 4)
            if "l2_hit_rate" in d:
                traffic[workload_key]["l2_hit_rate"] = round(d["l2_hit_rate"], 4)
    json.dump(traffic, open(tf, "w"), indent=1, sort_keys=True)
    print(open(os.path.join(dst, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main()
