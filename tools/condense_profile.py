"""Turns what tools/profile_round.sh left under gpurun_out/<round>/ into the files committed under profiles/<round>/:
the bench JSON line, the rocprofv3 kernel statistics (all kernels, and this library's only), and pmc_traffic.json
(HBM bytes per launch and VALU instructions per pair evaluation of the headline kernel; FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950).   usage: python tools/condense_profile.py r02a r02"""
import collections, csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
I = os.path.join(ROOT, "gpurun_out", src)
O = os.path.join(ROOT, "profiles", dst)
os.makedirs(O, exist_ok=True)

line = open(os.path.join(I, "bench_default_n1.json")).read().strip().splitlines()[-1]
bench = json.loads(line)
open(os.path.join(O, "bench_default_n1.json"), "w").write(json.dumps(bench, indent=1) + "\n")

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
    for k in per:
        if needle in k:
            return k
    return None


cfg = bench["config"]
S, P, It = cfg["pairs_per_gpu"], cfg["points"], cfg["iterations"]
pair_evals = float(S) * It * P * P
out = {}
fetch, _ = counters("pmc_fetch")
write, _ = counters("pmc_write")
sq, dur = counters("pmc_sq")
for tag, needle in (("c4", "icp_batch_filter_kernel<2, 8, false>"), ("c4_default_search_strategy", "icp_pruned_kernel<1, false, true>"), ("c4_float64_exact", "icp_pruned_kernel<1, true, true>"), ("c4_sort", "icp_sort_kernel")):
    kf, kw, ks = pick(fetch, needle), pick(write, needle), pick(sq, needle)
    if not (kf and kw and ks):
        continue
    mean = lambda v: sum(v) / len(v)
    fe, wr = mean(fetch[kf]["FETCH_SIZE"]), mean(write[kw]["WRITE_SIZE"])
    e = {"kernel": needle.replace(", ", ","), "launches_profiled": len(fetch[kf]["FETCH_SIZE"]),
         "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "hbm_bytes_per_launch": (2 * fe + wr) * 1024,
         "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
         "sq_counters_per_launch": {c: mean(v) for c, v in sq[ks].items()},
         "kernel_ms_under_pmc": mean(dur[ks])}
    c = e["sq_counters_per_launch"]
    e["lane_ops_per_pair_eval"] = c["SQ_INSTS_VALU"] * 64 / pair_evals
    e["shader_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (e["kernel_ms_under_pmc"] * 1e-3) / 1e9          # the counter sums the 8 XCDs
    e["valu_instruction_issue_utilisation_at_2_cycles_per_wave64"] = c["SQ_INSTS_VALU"] * 2 / (1024 * e["kernel_ms_under_pmc"] * 1e-3 * e["shader_clock_GHz"] * 1e9)
    if tag == "c4":
        e["compulsory_bytes_per_launch"] = 2 * S * P * 8 + S * 2 * 128
        e["ratio_to_compulsory"] = e["hbm_bytes_per_launch"] / e["compulsory_bytes_per_launch"]
    out[tag] = e
# the bench line was printed before this round's PMC passes existed: its `traffic` / instruction-count fields (read from
# the newest committed pmc_traffic.json at run time) are brought in line with the passes of THIS round
if "c4" in out:
    for key in ("roofline", "roofline_hbm"):
        if key in bench:
            bench[key]["traffic"] = out["c4"]["hbm_bytes_per_launch"]
    bench["roofline"]["kernel_valu_instructions_per_pair_eval"] = out["c4"]["lane_ops_per_pair_eval"]
    bench["roofline"].pop("stale_profile", None)
    bench["roofline"]["profile"] = os.path.join("profiles", dst, "pmc_traffic.json")
    open(os.path.join(O, "bench_default_n1.json"), "w").write(json.dumps(bench, indent=1) + "\n")
# bench.py reports these figures only while the library is still built from the sources they were taken on
out["csrc_sha256"] = open(os.path.join(I, "csrc_sha256.txt")).read().strip()
out["command"] = "tools/profile_round.sh: rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs (one counter group per pass: FETCH_SIZE | WRITE_SIZE | SQ_*, GRBM_GUI_ACTIVE)"
json.dump(out, open(os.path.join(O, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
for r in rows[1:12]:
    print(r[:6])
