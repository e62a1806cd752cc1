"""Per-launch view of one training step from a rocprofv3 --kernel-trace CSV (r_kernel_trace.csv).

    python tools/timeline.py gpurun_out/prof/r_kernel_trace.csv [step_index_from_end=2] > profiles/rNN_step_timeline.md

A step is delimited by consecutive launches of the SA1 farthest-point-sampling kernel (fps_kernel<8, false>: one per step, on the
forked geometry stream).  For every launch of the chosen step: start offset, duration, the shortest duration the same kernel (same
name, same position in the step) shows anywhere in the trace, and whether it ran while the FPS kernel was resident - the
persistent GEMM kernels need a whole CU's register file, so a workgroup that finds its CU taken by FPS waits for another to
finish."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    return name[:90]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id") or r.get("Queue_Id") or "?"))
    rows.sort()
    everything = "--all" in sys.argv          # every launch (also the < 8 us ones) with the idle time in front of it
    argv = [a for a in sys.argv if a != "--all"]
    back = int(argv[2]) if len(argv) > 2 else 2
    marks = [i for i, r in enumerate(rows) if (r[2].startswith("fps_kernel<8, false") or r[2].startswith("fps_pk_kernel<"))]
    if len(marks) < back + 2:
        raise SystemExit("trace holds %d steps only" % len(marks))
    # duration statistics per (kernel name, occurrence index within its step)
    stat = collections.defaultdict(list)
    for a, b in zip(marks[:-1], marks[1:]):
        seen = collections.Counter()
        for r in rows[a:b]:
            stat[(r[2], seen[r[2]])].append(r[1] - r[0])
            seen[r[2]] += 1
    a, b = marks[-back - 1], marks[-back]
    step = rows[a:b]
    t0 = step[0][0]
    fps = [r for r in step if (r[2].startswith("fps_kernel") or r[2].startswith("fps_pk_kernel"))]
    fps_lo, fps_hi = fps[0][0], max(r[1] for r in fps)
    print("# one step of `python bench.py` (graph replay), per launch; times in us; step length %.1f us" % ((rows[b][0] - t0) / 1e3))
    print("FPS resident %.1f .. %.1f us of the step\n" % ((fps_lo - t0) / 1e3, (fps_hi - t0) / 1e3))
    print("| start | dur | min dur in trace | x min | under FPS | stream | kernel |")
    print("|---:|---:|---:|---:|:--:|---|---|")
    seen = collections.Counter()
    infl = 0.0
    for r in step:
        mn = min(stat[(r[2], seen[r[2]])])
        seen[r[2]] += 1
        under = r[0] < fps_hi and r[1] > fps_lo and not (r[2].startswith("fps_kernel") or r[2].startswith("fps_pk_kernel"))
        d = r[1] - r[0]
        if under and d > 20000:
            infl += d - mn
        if d >= 8000 or everything:
            print("| %.1f | %.1f | %.1f | %.2f | %s | %s | `%s` |" % ((r[0] - t0) / 1e3, d / 1e3, mn / 1e3, d / mn, "yes" if under else "", r[3], r[2]))
    print("\nsum over the launches under FPS of (duration - shortest duration of the same launch in the trace): %.1f us" % (infl / 1e3))


if __name__ == "__main__":
    main()
