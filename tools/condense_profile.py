"""Turns what tools/profile_round.sh left under gpurun_out/<round>/ into the files committed under profiles/<round>/:
the bench JSON line, the rocprofv3 kernel statistics (all kernels, and this library's only), and pmc_traffic.json — per profiled
kernel: HBM bytes per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as reported), VALU
lane-operations, issue utilisation, wait fraction of the wave cycles, LDS conflict share.   usage: python tools/condense_profile.py r04a r04"""
import collections, csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
I = os.path.join(ROOT, "gpurun_out", src)
O = os.path.join(ROOT, "profiles", dst)
os.makedirs(O, exist_ok=True)

line = open(os.path.join(I, "bench_default_n1.json")).read().strip().splitlines()[-1]
bench = json.loads(line)

newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)      # gpurun merges runs into one directory: take the last one
stats = newest(os.path.join(I, "stats", "*", "*_kernel_stats.csv"))
shutil.copy(stats, os.path.join(O, "bench_default_kernel_stats.csv"))
rows = list(csv.reader(open(stats)))
with open(os.path.join(O, "bench_default_kernel_stats_lslam.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        if "lslam::" in r[0]:
            w.writerow(r)


def counters(sub):
    f = newest(os.path.join(I, sub, "*", "*_counter_collection.csv"))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f.replace("counter_collection", "kernel_trace"))):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return per, dur


def pick(per, needle):
    hits = [k for k in per if needle in k]
    return hits[0] if hits else None


mean = lambda v: sum(v) / len(v)
cfg = bench["config"]
S, P, It = cfg["pairs_per_gpu"], cfg["points"], cfg["iterations"]
pair_evals = float(S) * It * P * P
c4_f32 = 2 * S * P * 8 + S * 2 * 128                      # both float32 clouds once + transforms in / out
c4_f64 = c4_f32 + 2 * S * P * 16                           # + both float64 clouds once
# (entry, pass suffix, kernel needle, pair evaluations one launch is equivalent to or None, compulsory HBM bytes per launch or None,
#  take only the launches at least this long in ms (0: all) — a workload may launch the same kernel for warm-ups on tiny inputs)
ENTRIES = [
    ("c4", "c4", "icp_batch_filter_kernel<2, 8, false>", pair_evals, c4_f32, 50),
    ("c4_default_search_strategy", "c4", "icp_queue_kernel<false, true, true>", pair_evals, c4_f32, 1),
    ("c4_float64_exact_all_pairs", "c4", "icp_batch_filter_kernel<2, 4, true>", pair_evals, c4_f64, 50),
    ("c4_float64_exact_default", "c4", "icp_queue_kernel<true, true, true>", pair_evals, c4_f64, 1),
    ("c4_sort", "c4", "icp_sort_kernel", None, None, 0),
    ("c3_float64_exact", "c3", "icp_queue_kernel<true, false, false>", None, None, 0.3),
    ("c3_float32", "c3", "icp_queue_kernel<false, false, false>", None, None, 0.3),
    ("c3_sort", "c3", "icp_sort_kernel", None, None, 0.02),
    ("c5_all_pairs", "c5", "nn_tiled_filter_kernel<3, 4, false>", 1e10, None, 0),
    ("c5_default_search_strategy", "c5", "nn_pruned_kernel<3, 1, false>", 1e10, None, 0),
    ("c5_float64_exact_default", "c5", "nn_pruned_kernel<3, 1, true>", 1e10, None, 0),
    ("ogm", "ogm", "ogm_tile_kernel<short>", None, None, 0.02),
    ("ogm_apply", "ogm", "ogm_apply_kernel<short>", None, None, 0.02),
    ("filter_lidar_dbscan_core", "filter_lidar", "dbscan_mask_kernel<0>", None, None, 0.2),
    ("filter_lidar_dbscan_keep", "filter_lidar", "dbscan_mask_kernel<1>", None, None, 0.2),
    ("c2", "c2", "icp_coop_kernel<2, 1, false>", 50 * 1081.0 * 1081, None, 0),
    ("c2_float64_exact", "c2", "icp_coop_kernel<2, 1, true>", 50 * 1081.0 * 1081, None, 0),
]
out = {}
cache = {}
for tag, suffix, needle, evals, compulsory, min_ms in ENTRIES:
    try:
        if suffix not in cache:
            cache[suffix] = (counters("pmc_fetch_" + suffix), counters("pmc_write_" + suffix), counters("pmc_sq_" + suffix))
    except (ValueError, OSError):
        continue
    (fetch, fdur), (write, wdur), (sq, dur) = cache[suffix]
    kf, kw, ks = pick(fetch, needle), pick(write, needle), pick(sq, needle)
    if not (kf and kw and ks):
        continue

    def sel(per, durs, key, counter):                       # the launches of the real workload only
        v, d = per[key][counter], durs[key]
        keep = [x for x, t in zip(v, d) if t >= min_ms] if len(v) == len(d) else v
        return keep or v

    fe, wr = mean(sel(fetch, fdur, kf, "FETCH_SIZE")), mean(sel(write, wdur, kw, "WRITE_SIZE"))
    c = {name: mean(sel(sq, dur, ks, name)) for name in sq[ks]}
    ms = mean([t for t in dur[ks] if t >= min_ms] or dur[ks])
    e = {"kernel": needle.replace(", ", ","), "launches_profiled": len(sel(sq, dur, ks, "SQ_INSTS_VALU")),
         "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "hbm_bytes_per_launch": (2 * fe + wr) * 1024,
         "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
         "sq_counters_per_launch": c, "kernel_ms_under_pmc": ms}
    e["valu_lane_ops_per_launch"] = c["SQ_INSTS_VALU"] * 64
    e["shader_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9          # the counter sums the 8 XCDs
    e["valu_instruction_issue_utilisation_at_2_cycles_per_wave64"] = c["SQ_INSTS_VALU"] * 2 / (1024 * ms * 1e-3 * e["shader_clock_GHz"] * 1e9)
    e["wait_fraction_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
    e["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None
    if evals:
        e["lane_ops_per_pair_eval"] = c["SQ_INSTS_VALU"] * 64 / evals
    if compulsory:
        e["compulsory_bytes_per_launch"] = compulsory
        e["ratio_to_compulsory"] = e["hbm_bytes_per_launch"] / compulsory
    out[tag] = e
# the bench line was printed before this round's PMC passes existed: its profile-derived fields (read from the newest committed
# pmc_traffic.json at run time) are brought in line with the passes of THIS round
if "c4" in out:
    for key in ("roofline", "roofline_hbm"):
        if key in bench:
            bench[key]["traffic"] = out["c4"]["hbm_bytes_per_launch"]
    bench["roofline"]["kernel_valu_instructions_per_pair_eval"] = out["c4"]["lane_ops_per_pair_eval"]
    bench["roofline"].pop("stale_profile", None)
    bench["roofline"]["profile"] = os.path.join("profiles", dst, "pmc_traffic.json")


def patch(obj, tag):
    if not isinstance(obj, dict) or tag not in out:
        return
    e = out[tag]
    obj.update({"traffic": e["hbm_bytes_per_launch"], "compulsory_bytes": e.get("compulsory_bytes_per_launch"),
                "traffic_ratio_to_compulsory": e.get("ratio_to_compulsory"),
                "issue_utilisation": e["valu_instruction_issue_utilisation_at_2_cycles_per_wave64"],
                "wait_fraction": e["wait_fraction_of_wave_cycles"], "stale_profile": None, "profile": os.path.join("profiles", dst, "pmc_traffic.json")})


patch(bench.get("default_search_strategy", {}).get("roofline"), "c4_default_search_strategy")
patch(bench.get("float64_exact_mode", {}).get("roofline_all_pairs"), "c4_float64_exact_all_pairs")
patch(bench.get("float64_exact_mode", {}).get("roofline_default_strategy"), "c4_float64_exact_default")
oc = bench.get("other_configs", {})
patch(oc.get("c3", {}).get("roofline"), "c3_float64_exact")
patch(oc.get("c3", {}).get("roofline_float32"), "c3_float32")
patch(oc.get("c5", {}).get("default_search_strategy", {}).get("roofline"), "c5_default_search_strategy")
if "ogm" in out and isinstance(oc.get("ogm", {}).get("roofline"), dict):
    e, r = out["ogm"], oc["ogm"]["roofline"]
    ach = e["valu_lane_ops_per_launch"] / (e["kernel_ms_under_pmc"] * 1e-3) / 1e12
    cells = oc["ogm"].get("cell_updates_per_scan", 0) * 4956
    r.update({"achieved": ach, "frac": ach / 43.0, "kernel_ms": e["kernel_ms_under_pmc"], "traffic": e["hbm_bytes_per_launch"],
              "lane_instructions_per_cell_update": e["valu_lane_ops_per_launch"] * 39 / cells if cells else None,
              "stale_profile": None, "profile": os.path.join("profiles", dst, "pmc_traffic.json")})
open(os.path.join(O, "bench_default_n1.json"), "w").write(json.dumps(bench, indent=1) + "\n")
# bench.py reports these figures only while the library is still built from the sources they were taken on
out["csrc_sha256"] = open(os.path.join(I, "csrc_sha256.txt")).read().strip()
out["command"] = ("tools/profile_round.sh: rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py [--workload W] --no-cpu-baseline … "
                  "(one counter group per pass: FETCH_SIZE | WRITE_SIZE | SQ_*, GRBM_GUI_ACTIVE)")
json.dump(out, open(os.path.join(O, "pmc_traffic.json"), "w"), indent=1)
for k, e in out.items():
    if isinstance(e, dict):
        print(f"{k:34s} {e['kernel'][:40]:40s} {e['kernel_ms_under_pmc']:9.3f} ms  hbm {e['hbm_bytes_per_launch']/1e9:8.3f} GB"
              f"  x{e.get('ratio_to_compulsory') or float('nan'):5.2f}  issue {e['valu_instruction_issue_utilisation_at_2_cycles_per_wave64']:.3f}  wait {e['wait_fraction_of_wave_cycles'] or float('nan'):.3f}")
for r in rows[1:14]:
    print(r[:6])
