"""Adam for the trainers (train_Point2Cyl_without_sketch.py:204: torch.optim.Adam(model.parameters(), lr) with torch's defaults).

torch's fused / foreach Adam hands every 65,536-element chunk of every tensor to one thread block: the backbone's 1.4 M
parameters in 123 tensors become ~140 blocks in two launches, 88 us per step on an MI355X whose 256 CUs could do the whole
update in a few microseconds.  Here the update of ALL tensors of a parameter group is ONE launch of
csrc/bn.hip:adam_multi_kernel with 1024-element work items (p2c_adam_multi_f32), driven by a device-resident table of
(param, grad, exp_avg, exp_avg_sq) pointers.  The table is rebuilt only when a gradient tensor moved (eager steps; under the HIP
graph the gradient tensors are static).  Same update rule and operation order as torch.optim.Adam (no weight decay, no amsgrad);
`param_groups`, `zero_grad`, `state_dict` keep torch's shapes so schedules that poke `param_groups[i]["lr"]` keep working.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tables = {}

    def _table(self, gi, group):
        ps = [p for p in group["params"] if p.grad is not None]
        # the key covers every pointer the table holds: after load_state_dict (or anything else that replaces exp_avg / exp_avg_sq) the
        # cached table would otherwise keep pointing at the old, freed moment tensors
        key = tuple((p.data_ptr(), p.grad.data_ptr(), *((st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) if (st := self.state.get(p)) else (0, 0)))
                    for p in ps)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached
        dev = ps[0].device
        rows, numel, chunks = [], [], []
        for t, p in enumerate(ps):
            _lib.require_device(p, p.grad)
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise RuntimeError("point2cyl_amd.optim.Adam takes contiguous fp32 parameters and gradients")
            st = self.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            rows.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()])
            numel.append(p.numel())
            chunks += [[t, c] for c in range((p.numel() + 1023) // 1024)]
        tab = torch.tensor(rows, dtype=torch.int64).to(dev, non_blocking=True)
        nel = torch.tensor(numel, dtype=torch.int64).to(dev, non_blocking=True)
        chk = torch.tensor(chunks, dtype=torch.int32).to(dev, non_blocking=True)
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in ps)
        cached = (key, tab, nel, chk, len(chunks), ps)
        self._tables[gi] = cached
        return cached

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if not any(p.grad is not None for p in group["params"]):
                continue
            _, tab, nel, chk, n_chunks, ps = self._table(gi, group)
            for p in ps:
                self.state[p]["step"] += 1
            steps = {int(self.state[p]["step"]) for p in ps}
            if len(steps) != 1:     # one bias correction per launch: every tensor of the group must have taken the same number of updates
                raise RuntimeError("point2cyl_amd.optim.Adam: parameters of one group have different step counts %s (a parameter that first "
                                   "received a gradient later than the others); put it in its own param group" % sorted(steps))
            b1, b2 = group["betas"]
            call("p2c_adam_multi_f32", ptr(tab), ptr(nel), ptr(chk), n_chunks, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                 int(self.state[ps[0]]["step"]), stream())
        return loss
