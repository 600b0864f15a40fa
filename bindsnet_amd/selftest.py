"""`python -m bindsnet_amd.selftest` -- does THIS machine's torch still sum the way the kernels assume?

"Bit-exact against the reference" is a statement about the order in which the reference's PyTorch-CPU operators add floats
(DESIGN.md section 2).  That order belongs to torch, not to BindsNET: a torch release that changes
aten/src/ATen/native/cpu/SumKernel.cpp or TensorIteratorReduce.cpp changes what the reference computes.  This check needs no
fixtures and no oracle:

  host   torch.sum at 1 / 8 / 9 / 16 threads against the model the package carries of when ATen leaves its serial order
         (network/host_path.py::aten_sum_leaves_serial_order), and the host path's own reductions against torch at 1 thread;
  device (when an MI355X is present) snn_prop_cascade_f32, snn_stdp_postpre's batch sum and snn_normalize against the same torch
         expressions the reference evaluates, at 1 thread, on random data -- bit for bit.

Exit status 0 = the assumptions hold here; 1 = they do not (the run prints which)."""
import sys

import numpy as np
import torch


def _bits(t):
    return t.detach().cpu().contiguous().numpy().view(np.uint32)


def host_checks(report):
    from .network.host_path import _sum, aten_sum_leaves_serial_order
    n0 = torch.get_num_threads()
    ok = True
    g = torch.Generator().manual_seed(1234)
    try:
        for (B, Nin, N) in ((1, 784, 100), (3, 784, 100), (32, 784, 400), (2, 1000, 37), (4, 100, 100)):
            x = (torch.rand(B, Nin, 1, generator=g) < 0.3).float().repeat(1, 1, N) * (torch.rand(Nin, N, generator=g) - 0.5)
            torch.set_num_threads(1)
            serial = _bits(x.sum(1))
            for t in (8, 9, 16):
                torch.set_num_threads(t)
                differs = bool((_bits(x.sum(1)) != serial).any())
                predicted = aten_sum_leaves_serial_order(B, Nin, N, t)
                if differs and not predicted:
                    ok = False
                    report(f"host: torch.sum(dim=1) of [{B},{Nin},{N}] at {t} threads leaves the serial order where the model says it does not")
                if not (_bits(_sum(x, 1)) == serial).all():
                    ok = False
                    report(f"host: the host path's reduction of [{B},{Nin},{N}] at {t} threads is not the serial order")
    finally:
        torch.set_num_threads(n0)
    return ok


def device_checks(report):
    from . import ops
    dev = torch.device("cuda")
    n0 = torch.get_num_threads()
    torch.set_num_threads(1)                       # the order the package pins: the reference's serial one
    ok = True
    g = torch.Generator().manual_seed(4321)
    try:
        for (B, Nin, N) in ((3, 784, 100), (32, 784, 400), (2, 1000, 37)):
            W = torch.rand(Nin, N, generator=g) - 0.5
            s = (torch.rand(B, Nin, generator=g) < 0.3).to(torch.uint8)
            want = (s.view(B, Nin, 1).repeat(1, 1, N) * W).sum(1)                    # topology.py:469-471 + topology_features.py:641
            out = torch.empty(B, N, device=dev)
            ops.prop_cascade(W.to(dev), s.to(dev), out)
            if not (_bits(out) == _bits(want)).all():
                ok = False
                report(f"device: snn_prop_cascade_f32 [{B},{Nin},{N}] differs from torch's serial sum(dim=1)")
            Wn = torch.rand(Nin, N, generator=g) + 0.01
            cs = Wn.sum(0).unsqueeze(0)                                             # topology_features.py:250-266
            wantn = Wn * (78.4 / cs)
            Wd = Wn.to(dev).contiguous()
            ops.normalize(Wd, 78.4, use_abs=False)
            if not (_bits(Wd) == _bits(wantn)).all():
                ok = False
                report(f"device: snn_normalize [{Nin},{N}] differs from torch's serial column sums")
        for (B, Nin, N) in ((16, 64, 32), (32, 784, 100), (48, 40, 24)):
            W = torch.rand(Nin, N, generator=g)
            s_src, s_tgt = (torch.rand(B, Nin, generator=g) < 0.3), (torch.rand(B, N, generator=g) < 0.2)
            x_src, x_tgt = torch.rand(B, Nin, generator=g), torch.rand(B, N, generator=g)
            nu0, nu1 = torch.tensor(1e-4), torch.tensor(1e-2)
            want = W.clone()                                                        # MCC_learning.py:224-302 (+ :86-110), dt = 1
            want -= torch.sum(torch.bmm(s_src.unsqueeze(2).float(), x_tgt.unsqueeze(1) * nu0), dim=0) * 1.0
            want += torch.sum(torch.bmm(x_src.unsqueeze(2), s_tgt.unsqueeze(1).float() * nu1), dim=0) * 1.0
            want.clamp_(0.0, 1.0)
            Wd = W.to(dev).contiguous()
            ops.stdp_postpre(Wd, s_src.to(dev).to(torch.uint8), x_src.to(dev), s_tgt.to(dev).to(torch.uint8), x_tgt.to(dev), 1e-4, 1e-2, True, dt=1.0,
                             wmin=0.0, wmax=1.0)
            if not (_bits(Wd) == _bits(want)).all():
                ok = False
                report(f"device: snn_stdp_postpre [{B},{Nin},{N}] differs from torch's serial batch sum")
        torch.cuda.synchronize()
    finally:
        torch.set_num_threads(n0)
    return ok


def run(verbose=True) -> bool:
    msgs = []
    ok = host_checks(msgs.append)
    if torch.cuda.is_available():
        ok = device_checks(msgs.append) and ok
    elif verbose:
        print("bindsnet_amd.selftest: no GPU here -- host checks only")
    if verbose:
        for m in msgs:
            print("bindsnet_amd.selftest: FAILED --", m)
        print(f"bindsnet_amd.selftest: torch {torch.__version__}: " + ("summation-order assumptions hold" if ok else "summation-order assumptions DO NOT hold"))
    return ok


if __name__ == "__main__":
    sys.exit(0 if run() else 1)
