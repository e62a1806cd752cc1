"""Turn one round of rocprofv3 output under gpurun_out/ into the tracked summaries under profiles/.

    python tools/make_profiles.py <tag> <kernel-trace dir> <pmc FETCH_SIZE dir> <pmc WRITE_SIZE dir> <bench log>

writes profiles/<tag>_rocprofv3_kernel_stats.csv (rocprofv3's own --stats table), profiles/<tag>_kernel_stats.md (per-step
view), profiles/<tag>_pmc_hbm_traffic.json and profiles/<tag>_bench.json.log (the JSON line of the profiled command)."""
import collections, csv, json, os, re, shutil, sys

tag, tdir, fdir, wdir, blog = sys.argv[1:6]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
shutil.copy(os.path.join(tdir, "r_kernel_stats.csv"), os.path.join(P, tag + "_rocprofv3_kernel_stats.csv"))
line = [l for l in open(blog) if l.startswith("{")][-1]
open(os.path.join(P, tag + "_bench.json.log"), "w").write(line)
b = json.loads(line)
steps_traced = b["steps"] + b["warmup"] + 2 + 3          # timed + warm-up + 2 capture warm-ups + 3 eager event-timing steps
rows = list(csv.DictReader(open(os.path.join(tdir, "r_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(P, tag + "_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats of `python bench.py --steps %d --warmup %d` (configs[1], 1 GPU)\n\n" % (b["steps"], b["warmup"]))
    f.write("bench line of this run: %.1f points/s, %.3f ms/step.  The trace holds %d passes over the step (timed + warm-up + capture "
            "warm-ups + the 3 eager steps bench.py uses for its HIP-event timing); 'us/step' = total / %d.  Kernels of the geometry "
            "prefetch (fps, ball_query, three_nn, build_csr, group_gather_xyz) run on the forked stream, concurrently with the rest.\n\n"
            % (b["value"], b["ms_per_step"], steps_traced, steps_traced))
    f.write("| kernel | calls | avg us | us/step | % of kernel time |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:48]:
        f.write("| `%s` | %s | %.1f | %.1f | %.2f |\n" % (short(r["Name"])[:100], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                        float(r["TotalDurationNs"]) / 1e3 / steps_traced, float(r["Percentage"])))
    f.write("\ntotal kernel time %.3f ms over %d kernels names\n" % (tot / 1e6, len(rows)))
out = {}
for c, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(d, "r_counter_collection.csv"))):
        if r["Counter_Name"] == c:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {"launches": len(v)})
        out[k]["fetch_bytes_per_launch" if c == "FETCH_SIZE" else "write_bytes_per_launch"] = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1)
keep = {k: v for k, v in out.items() if any(s in k for s in ("bwd_fused_pp", "fwd_pp", "gemm_kernel", "csr_gather", "maxpool", "fps_kernel",
                                                             "group_gather_kernel", "three_interp", "bn_bwd_partial", "axis_kernel"))}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python bench.py --steps 2 --warmup 1 --no_graph "
                   "--no_cpu_baseline` (configs[1]); counters are in KB (x1024); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies a "
                   "128-B request of a wide coalesced read as 64 B); WRITE_SIZE uncorrected.  Per-launch means over all launches of the kernel.",
           "kernels": keep}, open(os.path.join(P, tag + "_pmc_hbm_traffic.json"), "w"), indent=1)
print("wrote", [x for x in sorted(os.listdir(P)) if x.startswith(tag)])
