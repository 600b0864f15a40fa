"""PyTorch-CPU restatement of the reference's DiehlAndCook2015 `Network.run()` -- the SAME ATen operator
sequence the reference executes per timestep, written as one flat loop over plain tensors.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): it is (1) pinned bit-for-bit against the
reference-generated fixtures in tests/test_oracle_golden.py and (2) timed by bench.py's `cpu_baseline`
leg on the GPU box's host cores, where /root/reference does not exist.  Because it issues the same ATen
calls on the same shapes as the reference (repeat + broadcast multiply + sum(1) for Weight.compute, two
bmm + sum(0) for PostPre, multinomial for one_spike, a clone per monitor per step) its throughput is the
reference's CPU throughput minus Python attribute/dispatch overhead of the class hierarchy -- i.e. an
upper bound of what the reference reaches on this host.  Nothing under bindsnet_amd/ imports this.

Operator order follows (paths inside BindsNET):
  network/network.py:211-250, 380-465      run loop, `zeros + c1 + c2` current accumulation, normalise
  network/topology.py:437-479              MulticompartmentConnection.compute (repeat, sum(1))
  network/topology_features.py:633-645     Weight.compute (value * conn_spikes); :250-266 normalize
  network/nodes.py:96-107, 211-221         trace update, Input.forward
  network/nodes.py:500-529                 LIFNodes.forward
  network/nodes.py:1069-1111               DiehlAndCookNodes.forward (theta, multinomial winner)
  learning/MCC_learning.py:224-302,86-110  PostPre (bmm, sum(0), *dt) then decay / clamp
  network/monitors.py:94-111               Monitor.record (one clone per monitor per step)
"""
import torch


class DcTorchRef:
    def __init__(self, n_inpt=784, n_neurons=400, exc=22.5, inh=120.0, dt=1.0, norm=78.4, theta_plus=0.05,
                 nu=(1e-4, 1e-2), wmin=0.0, wmax=1.0, tc_theta_decay=1e7, w=None):
        """Draws the learned weights from the global generator exactly where models.py:184 does."""
        N = n_neurons
        self.Nin, self.N, self.norm, self.learning = n_inpt, N, norm, True
        self.dt = torch.tensor(dt)
        self.W_xe = (0.3 * torch.rand(n_inpt, N)) if w is None else w.clone()
        self.W_ei = exc * torch.diag(torch.ones(N))
        self.W_ie = -inh * (torch.ones(N, N) - torch.diag(torch.ones(N)))
        self.nu = torch.zeros(2, dtype=torch.float)
        self.nu[0], self.nu[1] = nu[0], nu[1]
        self.wmin, self.wmax = wmin, wmax
        f = lambda v: torch.tensor(v)                                     # noqa: E731
        self.x_trace_decay = torch.exp(-self.dt / f(20.0))
        # excitatory D&C nodes (models.py:160-171)
        self.e_rest, self.e_reset, self.e_thresh, self.e_refrac = f(-65.0), f(-60.0), f(-52.0), f(5)
        self.e_decay = torch.exp(-self.dt / f(100.0))
        self.e_trace_decay = torch.exp(-self.dt / f(20.0))
        self.theta_plus = f(theta_plus)
        self.theta_decay = torch.exp(-self.dt / f(tc_theta_decay))
        # inhibitory LIF nodes (models.py:172-181)
        g = lambda v: torch.tensor(v, dtype=torch.float)                  # noqa: E731
        self.i_rest, self.i_reset, self.i_thresh, self.i_refrac = g(-60.0), g(-45.0), g(-40.0), f(2)
        self.i_decay = torch.exp(-self.dt / g(10.0))
        self.theta = torch.zeros(N)
        self.B = None
        self.consumed = 0

    # ---------------------------------------------------------------------------------------------
    def set_batch(self, B):
        self.B = B
        self.reset()

    def reset(self):
        """Network.reset_state_variables(): theta and weights persist."""
        B, N, Nin = self.B, self.N, self.Nin
        self.sX = torch.zeros(B, Nin, dtype=torch.uint8)
        self.xX = torch.zeros(B, Nin)
        self.vE = self.e_rest * torch.ones(B, N)
        self.rE = torch.zeros(B, N)
        self.sE = torch.zeros(B, N, dtype=torch.bool)
        self.xE = torch.zeros(B, N)
        self.vI = self.i_rest * torch.ones(B, N)
        self.rI = torch.zeros(B, N)
        self.sI = torch.zeros(B, N, dtype=torch.bool)

    @staticmethod
    def _mcc(value, s, n_src, n_tgt):
        conn = s.view(s.size(0), n_src, 1).repeat(1, 1, n_tgt)
        return (value * conn).sum(1)

    def run(self, spikes, monitors=("X", "Ae", "Ai")):
        """spikes: u8 [T, B, Nin].  Returns {layer: bool/u8 [T, B, n]} for the monitored layers."""
        T, B = spikes.shape[0], spikes.shape[1]
        if B != self.B:
            self.set_batch(B)
        N, Nin = self.N, self.Nin
        rec = {m: [] for m in monitors}
        self.consumed = 0
        for t in range(T):
            # currents from the PREVIOUS step's spikes, connection insertion order X->Ae, Ae->Ai, Ai->Ae
            cur_e = torch.zeros(B, N)
            cur_e += self._mcc(self.W_xe, self.sX, Nin, N)
            cur_i = torch.zeros(B, N)
            cur_i += self._mcc(self.W_ei, self.sE, N, N)
            cur_e += self._mcc(self.W_ie, self.sI, N, N)
            # X
            self.sX = spikes[t].view(B, Nin)
            self.xX *= self.x_trace_decay
            self.xX.masked_fill_(self.sX.bool(), 1.0)
            # Ae
            self.vE = self.e_decay * (self.vE - self.e_rest) + self.e_rest
            if self.learning:
                self.theta *= self.theta_decay
            self.vE += (self.rE <= 0).float() * cur_e
            self.rE -= self.dt
            self.sE = self.vE >= self.e_thresh + self.theta
            self.rE.masked_fill_(self.sE, self.e_refrac)
            self.vE.masked_fill_(self.sE, self.e_reset)
            if self.learning:
                self.theta += self.theta_plus * self.sE.float().sum(0)
            if self.sE.any():
                rows = self.sE.view(B, -1).any(1)
                p = self.sE.float().view(B, -1)[rows]
                self.consumed += p.numel()
                ind = torch.multinomial(p, 1)
                rows = rows.nonzero()
                self.sE.zero_()
                self.sE.view(B, -1)[rows, ind] = 1
            self.xE *= self.e_trace_decay
            self.xE.masked_fill_(self.sE.bool(), 1.0)
            # Ai
            self.vI = self.i_decay * (self.vI - self.i_rest) + self.i_rest
            cur_i.masked_fill_(self.rI > 0, 0.0)
            self.rI -= self.dt
            self.vI += cur_i
            self.sI = self.vI >= self.i_thresh
            self.rI.masked_fill_(self.sI, self.i_refrac)
            self.vI.masked_fill_(self.sI, self.i_reset)
            # PostPre on X->Ae
            if self.learning:
                src_s = self.sX.view(B, -1).unsqueeze(2).float()
                tgt_x = self.xE.view(B, -1).unsqueeze(1) * self.nu[0]
                self.W_xe -= torch.sum(torch.bmm(src_s, tgt_x), dim=0) * 1.0
                tgt_s = self.sE.view(B, -1).unsqueeze(1).float() * self.nu[1]
                src_x = self.xX.view(B, -1).unsqueeze(2)
                self.W_xe += torch.sum(torch.bmm(src_x, tgt_s), dim=0) * 1.0
                self.W_xe *= 1.0
                self.W_xe.clamp_(self.wmin, self.wmax)
            for m in monitors:
                rec[m].append({"X": self.sX, "Ae": self.sE, "Ai": self.sI}[m].unsqueeze(0).clone())
        colsum = self.W_xe.sum(0).unsqueeze(0)
        colsum[colsum == 0] = 1.0
        self.W_xe *= self.norm / colsum
        return {m: torch.cat(rec[m], 0) for m in monitors}
