import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch, cases
from oracle.torch_cpu_ref import DcTorchRef
g = cases.gold("full_cfg2_dc_n400_b32_poisson")
N, B, T = 400, 32, 250
for nt in (8, 1):
    torch.set_num_threads(nt)
    torch.manual_seed(0); ref = DcTorchRef(n_inpt=784, n_neurons=N); ref.set_batch(B); torch.manual_seed(2)
    for r in range(3 if nt == 8 else 1):
        sp = cases.fixture_input(g, r, T, B)
        t0 = time.time(); rec = ref.run(torch.from_numpy(sp))
        W = ref.W_xe.numpy()
        print(nt, r, round(time.time() - t0, 1), "ras", np.array_equal(rec["Ae"].numpy().astype(np.uint8), cases.unpack(g[f"r{r}_sE"], (T, B, N))),
              "W sha", cases.sha(W) == str(g[f"r{r}_W_sha"]), "sample maxdiff", np.abs(W.reshape(-1)[::97] - g[f"r{r}_W_sample"]).max(), flush=True)
        ref.reset()
