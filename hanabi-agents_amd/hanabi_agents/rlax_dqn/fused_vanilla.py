"""FusedVanillaLearner — the scalar double-DQN update (BASELINE config 2; spec hanabi_agents/rlax_dqn/rlax_dqn.py:170-205:
plain 2-layer MLP, double-Q TD with the bootstrap zeroed at terminal states, IS-weighted l2) on the kernels of the C51 learner.

The torch-autograd form of this update is ~50 launches at B = 256 (0.37 ms). Here it is ~15: one gather, the forward as two
GEMMs over [online | target] weights, `hb_dqn_loss_sparse` (td, IS weights, the ONE non-zero dLoss/dq per sample),
`hb_c51_backward` (dH with the ReLU mask, db1, dW2, db2 in one launch), the dW1 GEMM, and torch's fused Adam on the fp32 master
parameters (their `.grad` tensors are written directly). The scalar head is stored as a 2-atom head whose second atom is unused
(action a at column 2a), which is what lets the C51 backward kernel serve it unchanged. The autograd path stays as the reference
(`DQNLearning.loss`, tests/test_hip_policy.py::test_vanilla_double_dqn_gpu_equals_cpu_reference runs both).
"""
import torch

from hanabi_hip import _capi as K

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_DTYPES = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}


class FusedVanillaLearner:
    @staticmethod
    def supports(agent):
        p = agent.params
        return (agent.device.type == "cuda" and not agent.distributional and len(p.layers) == 1 and p.train_batch_size <= 256
                and 2 * agent.n_actions <= 64 and not p.use_priority)

    def __init__(self, agent):
        self.agent, p, dev = agent, agent.params, agent.device
        self.cd = _DTYPES[p.compute_dtype]
        self.L, self.H, self.A, self.B = agent.obs_len, p.layers[0], agent.n_actions, p.train_batch_size
        self.Kp, self.Np = -(-self.L // 64) * 64, 64
        cd, f32 = dict(dtype=self.cd, device=dev), dict(dtype=torch.float32, device=dev)
        B, H = self.B, self.H
        self.w1cat = torch.zeros(self.Kp, 2 * H, **cd)          # [online | target], pad rows stay zero
        self.b1cat = torch.zeros(2 * H, **cd)
        self.w2st = torch.zeros(2, H, self.Np, **cd)            # [online, target]; action a at column 2a
        self.b2st = torch.zeros(2, self.Np, **cd)
        self.x = torch.zeros(2 * B, self.Kp, **cd)              # pad columns stay zero
        self.act = torch.empty(B, dtype=torch.int32, device=dev)
        self.rew, self.term, self.disc = (torch.empty(B, **f32) for _ in range(3))
        self.td, self.w_is = torch.empty(B, **f32), torch.empty(B, **f32)
        self.dl = torch.zeros(B, 64, **f32)
        self.dh = torch.zeros(B, H, **cd)
        self.gw2 = torch.zeros(H, self.Np, **cd)
        self.gb2 = torch.zeros(self.Np, **f32)
        self.refresh_all()

    # ---- GEMM-dtype copies of the weights ---------------------------------------------------------------------------------
    @torch.no_grad()
    def _store(self, net, half):
        (w1, w2), (b1, b2) = net.weights, net.biases
        H, A = self.H, self.A
        self.w1cat[:self.L, half * H:(half + 1) * H].copy_(w1)
        self.b1cat[half * H:(half + 1) * H].copy_(b1)
        self.w2st[half][:, 0:2 * A:2].copy_(w2)
        self.b2st[half][0:2 * A:2].copy_(b2)

    def refresh_online(self):
        self._store(self.agent.online, 0)

    def refresh_target(self):
        self._store(self.agent.target, 1)

    def refresh_all(self):
        self.refresh_online()
        self.refresh_target()

    # ---- forward, loss, backward: gradients land in the parameters' .grad --------------------------------------------------
    def part1(self, indices, prios):
        a, L = self.agent, K.lib()
        buf, B, H, A = a.experience, self.B, self.H, self.A
        s = K.current_stream()
        gather = L.hb_replay_gather_packed if buf.packed else L.hb_replay_gather
        K.check(gather(K.dptr(buf._obs_tm1_buf), K.dptr(buf._obs_t_buf), K.dptr(buf._act_tm1_buf), K.dptr(buf._rew_t_buf),
                       K.dptr(buf._terminal_t_buf), K.dptr(indices), B, self.L, K.dptr(self.x), _DT[self.cd], self.Kp,
                       K.dptr(self.act), K.dptr(self.rew), K.dptr(self.term), K.dptr(self.disc), int(a.params.n_step),
                       float(a.params.discount), buf.capacity, int(buf.rows_per_insert or 1), K.dptr(buf._size_wp), s))
        hcat = torch._addmm_activation(self.b1cat, self.x, self.w1cat, use_gelu=False)       # [2B, 2H], bias + ReLU fused
        q = torch.bmm(hcat.view(2 * B, 2, H).transpose(0, 1), self.w2st)                     # [2, 2B, Np]
        q_on, q_t = q[0], q[1, B:]
        K.check(L.hb_dqn_loss_sparse(K.dptr(q_on), K.dptr(q_t), _DT[self.cd], K.dptr(self.act), K.dptr(self.rew), K.dptr(self.term),
                                     K.dptr(prios), K.dptr(a._beta), K.dptr(self.disc), B, A, 2, self.Np, K.dptr(self.td),
                                     K.dptr(self.w_is), K.dptr(self.dl), K.dptr(self.b2st[0]), K.dptr(self.b2st[1]), s))
        (w1, w2), (b1, b2) = a.online.weights, a.online.biases
        hb = hcat[:B, :H]
        K.check(L.hb_c51_backward(K.dptr(self.dl), K.dptr(self.act), K.dptr(hb), hb.stride(0), K.dptr(self.w2st[0]), self.Np,
                                  _DT[self.cd], B, H, A, 2, K.dptr(self.dh), K.dptr(b1.grad), K.dptr(self.gw2), self.Np,
                                  K.dptr(self.gb2), s))
        gw1 = torch.mm(self.x[:B].t(), self.dh)                                               # [Kp, H]
        w1.grad.copy_(gw1[:self.L])
        w2.grad.copy_(self.gw2[:, 0:2 * A:2])
        b2.grad.copy_(self.gb2[0:2 * A:2])
        return self.td

    def loss(self):
        """mean(w_IS * 0.5 * td^2) of the last update (rlax_dqn.py:196-203)."""
        return torch.mean(self.w_is * 0.5 * self.td * self.td)
