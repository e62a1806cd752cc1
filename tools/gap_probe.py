"""Busy time and idle gaps of the training step's MAIN chain from a rocprofv3 --kernel-trace CSV:   python tools/gap_probe.py r_kernel_trace.csv
A step = from one launch of the staging kernel (copy2d_batch*: first node of every step) to the next; per step the union of the main-chain
kernels' [start, end) intervals (geometry kernels - fps / ball_query / three_nn / build_csr / group_gather_xyz - excluded), the span, and the
idle time inside the span.  Median over the steps of the trace."""
import csv, re, sys, statistics
GEOM = ("fps_", "ball_query", "three_nn_kernel", "build_csr", "group_gather_xyz")
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("copy2d_batch")]
res = []
for a, b in zip(marks[:-1], marks[1:]):
    main = [(s, e) for s, e, n in rows[a:b] if not n.startswith(GEOM)]
    geom = [(s, e) for s, e, n in rows[a:b] if n.startswith(GEOM)]
    if len(main) < 50:
        continue
    busy, cur_s, cur_e = 0, main[0][0], main[0][1]
    for s, e in main[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = max(e for _, e in main) - main[0][0]
    res.append((span, busy, span - busy, len(main), sum(e - s for s, e in main), sum(e - s for s, e in geom), rows[b][0] - rows[a][0]))
med = lambda i: statistics.median(r[i] for r in res) / 1e3
print("%d steps: step period %.1f us | main chain span %.1f us = busy %.1f us + idle gaps %.1f us over %d launches (sum of kernel durations %.1f us) | geometry kernels %.1f us"
      % (len(res), med(6), med(0), med(1), med(2), int(statistics.median(r[3] for r in res)), med(4), med(5)))
