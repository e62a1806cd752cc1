"""Collect rocprofv3 PMC counters for EVERY kernel of one training step (run ON the GPU box), one counter group per pass.

    python tools/collect_pmc.py <tag>          # writes profiles/<tag>_pmc_step.json and profiles/<tag>_pmc_step.md

Each pass profiles `python bench.py --steps 2 --warmup 1 --no_graph --no_cpu_baseline --no_extras` (configs[1], kernels launched one by one so
that every dispatch is attributed) with `rocprofv3 --pmc ...` and nothing else but the kernel trace (MI355X_MICROARCH.md: counters
in their own runs; FETCH_SIZE and WRITE_SIZE cannot share a pass; 8 SQ slots).  Units as the guide calibrates them:
FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B);
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs.
Derived per kernel: HBM bytes per launch; MFMA issue share = 64 cycles x SQ_INSTS_MFMA... / available SIMD cycles is NOT formed here
(no calibrated clock in a profiled pass): the md table gives ratios of counters of the same pass only.
"""
import collections
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "sq_time": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
                "SQ_INSTS_MFMA", "SQ_VALU_MFMA_COEXEC_CYCLES"],
    "sq_mix": ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_VALU", "SQ_INSTS_LDS",
               "SQ_INSTS_VMEM", "SQ_INSTS_SALU"],
    "sq_lds": ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
               "SQ_ACTIVE_INST_MISC", "SQ_BUSY_CU_CYCLES"],
}
CMD = [sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no_graph", "--no_cpu_baseline", "--no_extras"]


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")[:120]


def main():
    tag = sys.argv[1]
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(PASSES)
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    env = dict(os.environ, TMPDIR="/tmp")
    for name in only:
        d = os.path.join(ROOT, "gpurun_out", "%s_pmc_%s" % (tag, name))
        cmd = ["rocprofv3", "--pmc"] + PASSES[name] + ["--kernel-trace", "-d", d, "-o", "r", "--output-format", "csv", "--"] + CMD
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
        f = os.path.join(d, "r_counter_collection.csv")
        if r.returncode != 0 or not os.path.exists(f):
            err = "\n".join(l for l in r.stderr.splitlines() if "Opened result file" not in l and "simple_timer" not in l)
            sys.stderr.write("pass %s failed (rc %d): %s\n" % (name, r.returncode, err[-1500:]))
            if not os.path.exists(f):
                continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))     # (kernel, dispatch) -> counter -> value
        for row in csv.DictReader(open(f)):
            per[(short(row["Kernel_Name"]), row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
        for (k, _), cs in per.items():
            for c, v in cs.items():
                out[k][c].append(v)
    res = {}
    for k, cs in out.items():
        e = {"launches": max(len(v) for v in cs.values())}
        for c, v in cs.items():
            e[c] = sum(v) / len(v)
        if "FETCH_SIZE" in e:
            e["fetch_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in e:
            e["write_bytes_per_launch"] = e["WRITE_SIZE"] * 1024
        res[k] = e
    note = ("rocprofv3 --pmc, one counter group per pass of `python bench.py --steps 2 --warmup 1 --no_graph --no_cpu_baseline --no_extras` (configs[1]); "
            "per-launch means over all launches of each kernel (3 steps + graph-less warm-up); FETCH_SIZE/WRITE_SIZE in KiB, fetch bytes = "
            "FETCH_SIZE x 1024 x 2 (MI355X_MICROARCH.md gfx950 correction), write bytes uncorrected")
    steps = 3
    P = os.path.join(ROOT, "profiles")
    sys.path.insert(0, ROOT)
    from point2cyl_amd.build import source_hash
    json.dump({"note": note, "steps_in_trace": steps, "source_hash": source_hash(), "kernels": res}, open(os.path.join(P, tag + "_pmc_step.json"), "w"), indent=1)
    tot_f = sum(e.get("fetch_bytes_per_launch", 0) * e["launches"] for e in res.values()) / steps
    tot_w = sum(e.get("write_bytes_per_launch", 0) * e["launches"] for e in res.values()) / steps
    with open(os.path.join(P, tag + "_pmc_step.md"), "w") as f:
        f.write("# PMC view of one training step (configs[1], eager launches), all kernels\n\n%s.\n\n" % note)
        f.write("HBM traffic of one step, all kernels: read %.2f GB + write %.2f GB = %.2f GB\n\n" % (tot_f / 1e9, tot_w / 1e9, (tot_f + tot_w) / 1e9))
        f.write("| kernel | launches/step | read MB | write MB | MFMA insts | MFMA busy / SQ busy-CU cyc | wave cyc: active / wait / inst-stall | "
                "active inst: VALU / LDS / VMEM / SALU | LDS bank-conflict / LDS active |\n|---|---:|---:|---:|---:|---:|---|---|---:|\n")
        key = lambda kv: -(kv[1].get("fetch_bytes_per_launch", 0) + kv[1].get("write_bytes_per_launch", 0)) * kv[1]["launches"]
        for k, e in sorted(res.items(), key=key)[:60]:
            g = lambda c: e.get(c, float("nan"))
            wc = g("SQ_WAVE_CYCLES")
            f.write("| `%s` | %.1f | %.1f | %.1f | %.3g | %.2f | %.2f / %.2f / %.2f | %.2f / %.2f / %.2f / %.2f | %.3f |\n" % (
                k[:80], e["launches"] / steps, g("fetch_bytes_per_launch") / 1e6, g("write_bytes_per_launch") / 1e6, g("SQ_INSTS_MFMA"),
                g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_BUSY_CU_CYCLES")) if g("SQ_BUSY_CU_CYCLES") == g("SQ_BUSY_CU_CYCLES") and g("SQ_BUSY_CU_CYCLES") else float("nan"),
                g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc,
                g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VMEM") / wc, g("SQ_ACTIVE_INST_SCA") / wc,
                g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else float("nan")))
    print("wrote profiles/%s_pmc_step.{json,md}: step traffic %.2f GB" % (tag, (tot_f + tot_w) / 1e9))


if __name__ == "__main__":
    main()
