"""Does an EXTERNAL event recorded inside a captured HIP graph release a side stream before the graph's tail has run?
(the gate the data-parallel exchange needs to overlap the replay tail: ddp.FlatGradSync / graph.py)"""

import torch

dev = torch.device("cuda:0")
a = torch.zeros(1 << 20, device=dev)
big = torch.randn(8192, 8192, device=dev)
out = torch.empty_like(big)
flag = torch.zeros(1, device=dev)
side = torch.cuda.Stream()
cap = torch.cuda.Stream()
ev = torch.cuda.Event(external=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    a.add_(1.0); torch.mm(big, big, out=out)
    cap.synchronize()
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        a.add_(1.0)                       # "the gradients"
        ev.record()                       # external: an event-record NODE
        for _ in range(6):                # "the tail": ~6 x 0.4 ms
            torch.mm(big, big, out=out)
torch.cuda.synchronize()
t0, t1, t2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
for it in range(3):
    a.zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(cap):
        t0.record()
        g.replay()
        t2.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        seen = a[:4].clone()              # must see the graph's first node
        t1.record()
    torch.cuda.synchronize()
    print("replay %d: side stream done %.3f ms after the replay started, graph end at %.3f ms, side saw a = %s"
          % (it, t0.elapsed_time(t1), t0.elapsed_time(t2), seen.tolist()))
