"""FusedLearner — one learner update as ~10 launches instead of ~250 (GPU only).

Same arithmetic as `DQNLearning.loss` + `torch.optim.Adam` on the NoisyMLP (the PyTorch-autograd
path in rlax_rainbow.py stays as the fp32 reference and is what runs on the CPU), arranged as

    PER sample (hb_per_sample)  ->  gather into the GEMM operand (hb_replay_gather)
    ->  online and target passes together: one GEMM per layer (layer 1 on the concatenated [W1 | W1_target], layer 2
        as a batched GEMM over the two networks), effective weights W = w + w_mu + w_sigma*eps already materialised in
        the GEMM dtype
    ->  hb_c51_loss_sparse: IS weights, double-Q selection, projection, cross-entropy, and dLoss/dlogits in its compact
        form (non-zero only in the K atoms of the action each sample took: [B, 64] fp32)
    ->  backward by hand: hb_c51_backward (one launch: dH masked by the ReLU, db1, dW2, db2 from the compact gradient,
        fixed summation order), then dW1 = X^T dH (GEMM)
    [-> when data-parallel: packed into ONE flat fp32 buffer and all-reduced over RCCL]
    ->  hb_noisy_adam per merged tensor: routes the gradient to (w, w_mu, w_sigma) as (g, g, g*eps), Adam on
        each, and emits the next effective weight
    ->  priority update (hb_per_update)

The reference computes the same thing through jax.grad of the loss (hanabi_agents/rlax_dqn/rlax_rainbow.py:
152-217) with the six-parameter layer of noisy_mlp.py:61-91; since dW/dw = dW/dw_mu = 1 and dW/dw_sigma = eps,
the hand-written backward is exact, not an approximation. Both halves (before / after the optional collective)
are captured into HIP graphs by the agent.
"""
import os

import torch

from hanabi_hip import _capi as K

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


class FusedLearner:
    def __init__(self, agent):
        self.agent = agent
        p = agent.params
        dev = agent.device
        self.cd = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}[p.compute_dtype]
        self.B = p.train_batch_size
        self.A, self.Kk = agent.atoms.shape
        self.L = agent.obs_len
        layers = agent.online.layers
        assert len(layers) == 2, "the fused learner covers the reference topology: one hidden layer (params.layers=[H])"
        self.layers = layers
        B, AK = self.B, self.A * self.Kk
        H = layers[0].out_features
        f32 = dict(dtype=torch.float32, device=dev)
        # merged tensors per layer: (weight, bias); effective values in the GEMM dtype, PADDED to the shapes
        # hipBLASLt runs fastest (scripts/gemm_probe.py: K 658 -> 704: 39 -> 27 us, N 1020 -> 1024: 61 -> 43 us at
        # 32 768 rows). Padding rows / columns are zero and never written, so results are unchanged.
        pad = (lambda v, m: (v + m - 1) // m * m) if self.cd != torch.float32 else (lambda v, m: v)
        self.Kp = pad(self.L, 64)           # first-layer K
        self.Np = pad(AK, 64)               # second-layer N (logits row stride)
        # Online and target operands live side by side so that ONE GEMM per layer serves both networks:
        #   layer 1: X [2B, Kp] @ [W1 | W1_target] [Kp, 2H]           (the target half of rows 0..B-1 is unused work)
        #   layer 2: batched GEMM over {online, target}: [2, 2B, H] @ [2, H, Np]; the biases are added by the loss kernel
        self.w1cat = torch.zeros(self.Kp, 2 * H, dtype=self.cd, device=dev)
        self.b1cat = torch.zeros(2 * H, dtype=self.cd, device=dev)
        self.w2st = torch.zeros(2, H, self.Np, dtype=self.cd, device=dev)
        self.b2st = torch.zeros(2, self.Np, dtype=self.cd, device=dev)
        self.eff = [(self.w1cat[:, :H], self.b1cat[:H]), (self.w2st[0], self.b2st[0])]
        self.trg = [(self.w1cat[:, H:], self.b1cat[H:]), (self.w2st[1], self.b2st[1])]
        self.H = H
        # "thin" forward (bf16): the two dense layers on hb_thin_gemm, a kernel small enough to run on the CUs WHILE the other
        # seat's actor GEMMs hold them (a library GEMM would wait for those to retire). It reads the weights TRANSPOSED
        # (k-contiguous): copies kept current by one hb_actor_pack_weights launch after every optimizer step / target sync.
        self.thin = (self.cd in (torch.bfloat16, torch.float16) and os.environ.get("HB_THIN_LEARNER", "1") != "0" and self.Kp % 32 == 0
                     and H % 64 == 0 and self.Np % 16 == 0 and (2 * B) % 32 == 0)
        if self.thin:
            self.w1catT = torch.zeros(2 * H, self.Kp, dtype=self.cd, device=dev)
            self.w2stT = torch.zeros(2, self.Np, H, dtype=self.cd, device=dev)
            self._hcat = torch.zeros(2 * B, 2 * H, dtype=self.cd, device=dev)
            self._logits = torch.zeros(2, 2 * B, self.Np, **f32)   # fp32: the accumulators + bias as they are (round 3; bf16 before)
            self._bias_sink = torch.zeros(max(2 * H, self.Np), **f32)   # (the transposer also converts a bias: unused here)
            self._t_jobs = [None, None]
        # flat gradient buffer [dW1 | db1 | dW2 | db2] (fp32), one all-reduce bucket
        sizes = [layers[0].w.numel(), H, layers[1].w.numel(), AK]
        self.flat_grad = torch.zeros(sum(sizes), **f32)
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        self.g_w1 = self.flat_grad[offs[0]:offs[1]].view_as(layers[0].w)
        self.g_b1 = self.flat_grad[offs[1]:offs[2]]
        self.g_w2 = self.flat_grad[offs[2]:offs[3]].view_as(layers[1].w)
        self.g_b2 = self.flat_grad[offs[3]:offs[4]]
        # Adam moments for the 12 parameter tensors
        self.state = {}
        for li, l in enumerate(layers):
            for name in ("w", "w_mu", "w_sigma", "b", "b_mu", "b_sigma"):
                t = getattr(l, name)
                self.state[(li, name)] = (torch.zeros_like(t), torch.zeros_like(t))
        # w and w_mu (b and b_mu) always receive the same gradient, so their Adam moments are equal for ever: they share
        # storage and the kernel applies w's step to w_mu (one fifth less Adam traffic)
        for li in range(len(layers)):
            self.state[(li, "w_mu")] = self.state[(li, "w")]
            self.state[(li, "b_mu")] = self.state[(li, "b")]
        self.step = torch.zeros((), **f32)
        # static batch buffers
        self.x = torch.zeros(2 * B, self.Kp, dtype=self.cd, device=dev)   # pad columns stay zero
        self.act = torch.empty(B, dtype=torch.int32, device=dev)
        self.rew = torch.empty(B, **f32)
        self.term = torch.empty(B, **f32)
        self.disc = torch.empty(B, **f32)
        self.td = torch.empty(B, **f32)
        self.w_is = torch.empty(B, **f32)
        self.dlogits = torch.zeros(B, self.Np, dtype=self.cd, device=dev)   # dense dLoss/dlogits (sparse_backward False only)
        self.dl = torch.zeros(B, 64, **f32)                                # compact dLoss/dlogits: the K atoms of the taken action
        self.dh = torch.zeros(B, H, dtype=self.cd, device=dev)
        self.sparse_backward = self.Kk <= 64 and self.A <= 64 and B <= 256   # (else: the dense chain below)
        self.AK = AK
        self.support = agent.atoms[0].contiguous()
        self._gb2_pad = torch.zeros(self.Np, **f32)
        self._adam_tab = None
        self._idx = self._prob = None     # outputs of the fused sample + gather launch
        self._sg_call = None              # cached argument list of that launch
        # Single rank: Adam reads the weight gradients straight from the (padded, GEMM-dtype) outputs of the two
        # backward GEMMs and the bias gradients from the column-sum outputs: no pack / convert launches. With data
        # parallelism the gradients are first packed into the flat fp32 all-reduce bucket above.
        # (decided at the first update, when the process group the run uses is certainly up)
        self.direct = None
        # the actor's forward on the hand-written MFMA kernels (csrc/actor.hip) reads transposed copies of eff
        from hanabi_hip.ops import ActorMFMA

        self.actor = None
        self.actor_stale = True
        # params.actor_lag = 1: two actor weight sets. Update number u (counted from 0) packs its result into set u % 2 on the
        # stream it ran on and records packed_ev[u % 2]; a policy call made when u updates have been launched reads set u % 2, i.e.
        # the weights of update u - 2 (the initial weights while u < 2), which the update in flight (u - 1) does not touch.
        self.lag = int(getattr(agent.params, "actor_lag", 0))
        if self.lag not in (0, 1):
            raise ValueError("actor_lag must be 0 or 1")
        self.n_packed = 0
        self._packed_in_part2 = False
        self.packed_ev = [None, None]
        if ActorMFMA.supports(self.L, H, self.Kk, self.Kp, self.cd, self.A) and getattr(agent, "use_mfma_actor", True):
            self.actor = ActorMFMA(self.L, H, self.A, self.Kk, self.Kp, dev, n_sets=2 if self.lag else 1, dtype=self.cd)
        # one pack launch per update: hb_actor_fused_pack_thin writes the thin GEMMs' transposed ONLINE weights together with the
        # one-kernel actor's copies, right behind Adam; part1() then launches no transposer (HB_PACK_THIN=0: two launches as before)
        self.pack_thin = bool(self.thin and self.actor is not None and self.actor.fused and os.environ.get("HB_PACK_THIN", "1") != "0")
        if self.lag and (self.actor is None or not agent.params.use_priority or agent.params.resample_noise):
            raise ValueError("actor_lag=1 needs the MFMA actor (bf16 GEMM dtype, one hidden layer of a multiple of 256 units), "
                             "prioritized replay and frozen noise")
        # Round 3, measured and NOT the default: Adam itself writing the copies the forward kernels read
        # (hb_noisy_adam_multi_pack: the thin GEMMs' transposed online weights and the one-kernel actor's fragment-major set 0) — two
        # launches fewer per update, but the kernel must walk the tensors in 8 x 128 tiles to own 8 consecutive k of a column, and
        # its eleven 55 MB streams then run in 512-byte pieces: 27 us against 12.5 (Adam) + 5.5 + 4.5 (the two packers); the step
        # time is the same (profiles/r03/ab_update_tail.txt). HB_ADAM_PACK=1 selects it. What IS the default: the one-kernel
        # actor's packer also writes the thin GEMMs' copies (one launch instead of two, `pack_thin`).
        self.adam_pack = None
        self._pack_tab = None
        self._gw2_out = torch.zeros(H, self.Np, dtype=self.cd, device=dev)
        self._gw1_out = torch.zeros(self.Kp, H, dtype=self.cd, device=dev)
        self.refresh_effective()
        self.refresh_target()

    # ---- effective weights ------------------------------------------------------------------------------
    @staticmethod
    def _store(dst_w, dst_b, w, b):
        dst_w[:w.shape[0], :w.shape[1]].copy_(w)
        dst_b[:b.shape[0]].copy_(b)

    def _transpose(self, which):
        """thin forward: refresh the transposed copies of the online (0) or target (1) weights: one launch."""
        jobs = self._t_jobs[which]
        if jobs is None:
            H = self.H
            (w1, b1), (w2, b2) = (self.eff, self.trg)[which]
            jobs = (K.HbPackJob * 2)()
            for j, (w, b, wt, k_rows, n_cols, kp) in enumerate(((w1, b1, self.w1catT[which * H:(which + 1) * H], self.Kp, H, self.Kp),
                                                                (w2, b2, self.w2stT[which], H, self.Np, H))):
                jobs[j].w, jobs[j].bias, jobs[j].wt, jobs[j].bias_out = w.data_ptr(), b.data_ptr(), wt.data_ptr(), self._bias_sink.data_ptr()
                jobs[j].k_rows, jobs[j].n_cols, jobs[j].w_ld, jobs[j].group_cols, jobs[j].k_pad = k_rows, n_cols, w.stride(0), 0, kp
            self._t_jobs[which] = jobs
        K.check(K.lib().hb_actor_pack_weights(jobs, 2, K.current_stream()))

    @torch.no_grad()
    def refresh_effective(self):
        for (w_e, b_e), l in zip(self.eff, self.layers):
            self._store(w_e, b_e, *l.effective())
        self.actor_stale = True
        if self.thin:
            self._transpose(0)

    def pack_actor(self):
        """Refresh the actor's transposed weight copies if the effective weights changed since the last call. Runs on
        the ACTING stream just before the actor kernels (the update that wrote `eff` has been waited for by then), so the
        two small launches stay off the learner chain, which is the critical path of a step. (actor_lag=1: only after the
        weights were replaced wholesale — construction, restore, checkpoint load; updates pack through weights_updated().)"""
        if self.actor is not None and self.actor_stale:
            (w1, b1), (w2, b2) = self.eff
            for s in range(self.actor.n_sets):
                self.actor.pack(w1, b1, w2, b2, s, lazy_two_kernel=not self.lag)
            self.actor_stale = False
            self.packed_ev = [None, None]

    def _thin_out(self):
        """The thin GEMMs' transposed online weights as extra outputs of the actor's pack launch (pack_thin), else None."""
        return (self.w1catT, self.Kp, self.w2stT[0], self.H) if self.pack_thin else None

    def weights_updated(self):
        """Called after every optimizer step, on the stream the step ran on."""
        if not self.lag:
            # part2() has already re-packed the actor's copies behind Adam, on the stream (and inside the graph) the update ran on
            self.actor_stale = self.actor is not None and not self._packed_in_part2
            if self._packed_in_part2 and self.actor.fused and self.actor.two_kernel:
                # part2() packed the one-kernel form's copies and left the two-kernel form's to the next policy call that takes
                # that form (pack(lazy_two_kernel=True)). The mark it set is a host-side effect: a REPLAYED graph does not repeat
                # it, so it is renewed here, after every optimizer step (without this the two-kernel form — batches below
                # fused_min_rows — kept acting on the weights of the update that was captured)
                self.actor._two_stale[0] = True
            return
        self.pack_actor()
        s = self.n_packed % 2
        (w1, b1), (w2, b2) = self.eff
        self.actor.pack(w1, b1, w2, b2, s, thin=self._thin_out())
        if self.packed_ev[s] is None:
            self.packed_ev[s] = K.Event()
        self.packed_ev[s].record()
        self.n_packed += 1

    def acting_set(self):
        """The weight set the next policy call reads; the calling stream is made to wait for the launch that packed it."""
        self.pack_actor()
        if not self.lag:
            return 0
        s = self.n_packed % 2
        if self.packed_ev[s] is not None:
            self.packed_ev[s].wait()
        return s

    @torch.no_grad()
    def refresh_target(self):
        for (w_t, b_t), l in zip(self.trg, self.agent.target.layers):
            self._store(w_t, b_t, *l.effective())
        if self.thin:
            self._transpose(1)

    # ---- the two halves of an update ----------------------------------------------------------------------
    def sample_and_gather(self, seed):
        """Prioritized sampling (hb_per_sample_philox) and the gather into the GEMM operand as ONE launch. Returns the sampled
        indices and probabilities (persistent buffers); part1(..., gathered=True) then skips its own gather."""
        a = self.agent
        buf, B = a.experience, self.B
        rpi = buf.rows_per_insert
        c = self._sg_call
        if c is not None and c[0] == (rpi, buf._obs_t_buf.data_ptr()) and c[1] == seed:   # only the stream can differ from last time
            K.check(c[2](*c[3], K.current_stream()))
            return self._idx, self._prob
        L = K.lib()
        if a.params.n_step > 1 and (rpi is None or rpi < 1):
            raise ValueError("n_step > 1 needs inserts of a constant row count (lock-step self-play)")
        if self._idx is None:
            self._idx = torch.empty(B, dtype=torch.int64, device=a.device)
            self._prob = torch.empty(B, dtype=torch.float64, device=a.device)
        args = (buf.sum_tree.h, int(seed), K.dptr(self.step), B, K.dptr(self._idx), K.dptr(self._prob),
                K.dptr(buf._obs_tm1_buf), K.dptr(buf._obs_t_buf), K.dptr(buf._act_tm1_buf),
                K.dptr(buf._rew_t_buf), K.dptr(buf._terminal_t_buf), self.L, 1 if buf.packed else 0,
                K.dptr(self.x), _DT[self.cd], self.Kp, K.dptr(self.act), K.dptr(self.rew), K.dptr(self.term),
                K.dptr(self.disc), int(a.params.n_step), float(a.params.discount), buf.capacity,
                int(rpi or 1), K.dptr(buf._size_wp))
        self._sg_call = ((rpi, buf._obs_t_buf.data_ptr()), seed, L.hb_per_sample_gather, args)   # (all operands are persistent buffers of this learner / ring)
        K.check(L.hb_per_sample_gather(*args, K.current_stream()))
        return self._idx, self._prob

    def part1(self, indices, prios, gathered=False):
        """Forward, loss, backward into flat_grad. indices int64 [B], prios float64 [B] (device)."""
        a, L = self.agent, K.lib()
        buf, B = a.experience, self.B
        if self.direct is None:
            self.direct = not a._collective()
        if self.adam_pack is None:
            self.adam_pack = bool(self.direct and self.thin and not self.lag and self.actor is not None and self.actor.fused
                                  and self.H == 512 and os.environ.get("HB_ADAM_PACK", "0") == "1")
        if a.params.n_step > 1 and (buf.rows_per_insert is None or buf.rows_per_insert < 1):
            raise ValueError("n_step > 1 needs inserts of a constant row count (lock-step self-play)")
        s = K.current_stream()
        if not gathered:
            gather = L.hb_replay_gather_packed if buf.packed else L.hb_replay_gather   # bit-packed rings expand to the same operand
            K.check(gather(K.dptr(buf._obs_tm1_buf), K.dptr(buf._obs_t_buf), K.dptr(buf._act_tm1_buf),
                           K.dptr(buf._rew_t_buf), K.dptr(buf._terminal_t_buf), K.dptr(indices), B, self.L,
                           K.dptr(self.x), _DT[self.cd], self.Kp, K.dptr(self.act), K.dptr(self.rew),
                           K.dptr(self.term), K.dptr(self.disc), int(a.params.n_step), float(a.params.discount),
                           buf.capacity, int(buf.rows_per_insert or 1), K.dptr(buf._size_wp), s))
        H = self.H
        w2 = self.eff[1][0]
        if self.thin:
            hcat, logits = self._hcat, self._logits
            # the transposed online weights are refreshed HERE (not right after Adam): the acting stream does not wait for
            # it, and like the two GEMMs below the small transposer runs beside the other seat's policy GEMMs
            if not (self.adam_pack or self.pack_thin):   # (else the pack behind Adam keeps the online half current)
                self._transpose(0)
            f16 = 4 if self.cd == torch.float16 else 0   # (hb_thin_gemm: bit 2 of its flags = fp16 operands)
            K.check(L.hb_thin_gemm(K.dptr(self.x), K.dptr(self.w1catT), K.dptr(self.b1cat), K.dptr(hcat), 2 * B, 2 * H, self.Kp,
                                   self.Kp, self.Kp, 2 * H, 1, 0, 0, 0, 1 | f16, s))                 # bias + ReLU, [2B, 2H]
            K.check(L.hb_thin_gemm(K.dptr(hcat), K.dptr(self.w2stT), K.dptr(self.b2st), K.dptr(logits), 2 * B, self.Np, H, 2 * H, H,
                                   self.Np, 2, H, self.Np * H, 2 * B * self.Np, 2 | f16, s))         # {online, target}: [2, 2B, Np] fp32, biases added
        else:
            hcat = torch._addmm_activation(self.b1cat, self.x, self.w1cat, use_gelu=False)   # bias + ReLU in the epilogue, [2B, 2H]
            logits = torch.bmm(hcat.view(2 * B, 2, H).transpose(0, 1), self.w2st)            # [2, 2B, Np], strided A operand: no copy
        logits_on, logits_t = logits[0], logits[1, B:]                                   # online on all 2B rows, target on obs_t
        hb, xb = hcat[:B, :H], self.x[:B]
        # thin forward: fp32 logits that already contain the output biases; library GEMMs: logits in the GEMM dtype, biases added by the loss kernel
        ldt = 0 if self.thin else _DT[self.cd]
        bon, btg = (None, None) if self.thin else (self.b2st[0], self.b2st[1])
        if self.sparse_backward:
            # dLoss/dlogits is non-zero only in the K atoms of the action each sample took: the loss kernel emits that compact
            # [B, 64] fp32 form and ONE launch turns it into dH (ReLU-masked), db1, dW2 and db2 (csrc/learner2.hip)
            K.check(L.hb_c51_loss_sparse(K.dptr(logits_on), K.dptr(logits_t), ldt, K.dptr(self.act), K.dptr(self.rew),
                                         K.dptr(self.term), K.dptr(prios), K.dptr(a._beta), K.dptr(self.disc),
                                         1 if a.params.mask_terminal else 0, K.dptr(self.support), B, self.A, self.Kk, self.Np,
                                         K.dptr(self.td), K.dptr(self.w_is), K.dptr(self.dl), K.dptr(self.step),
                                         K.dptr(bon), K.dptr(btg), s))
            K.check(L.hb_c51_backward(K.dptr(self.dl), K.dptr(self.act), K.dptr(hb), hb.stride(0), K.dptr(w2), w2.stride(0),
                                      _DT[self.cd], B, H, self.A, self.Kk, K.dptr(self.dh), K.dptr(self.g_b1),
                                      K.dptr(self._gw2_out), self.Np, K.dptr(self._gb2_pad), s))
            dh = self.dh
        else:   # the dense chain of the first fused learner: kept as a cross-check (tests) of the sparse kernels
            if self.thin:   # (this cross-check chain keeps its GEMM-dtype gradient: give it logits of that dtype; biases are already in)
                logits_on, logits_t, ldt = logits_on.to(self.cd), logits_t.to(self.cd), _DT[self.cd]
            K.check(L.hb_c51_loss_grad(K.dptr(logits_on), K.dptr(logits_t), ldt, K.dptr(self.act), K.dptr(self.rew),
                                       K.dptr(self.term), K.dptr(prios), K.dptr(a._beta), K.dptr(self.disc),
                                       1 if a.params.mask_terminal else 0, K.dptr(self.support), B, self.A, self.Kk, self.Np,
                                       K.dptr(self.td), K.dptr(self.w_is), K.dptr(self.dlogits), K.dptr(self.step),
                                       K.dptr(bon), K.dptr(btg), s))
            dl = self.dlogits
            torch.mm(hb.t(), dl, out=self._gw2_out)                       # [H, Np]; the padding columns are never read
            K.check(L.hb_colsum(K.dptr(dl), _DT[self.cd], B, self.Np, K.dptr(self._gb2_pad), s))
            dh = torch.mm(dl, w2.t())                                      # masked by the ReLU in place, with its column sums
            K.check(L.hb_relu_bwd_colsum(K.dptr(dh), K.dptr(hb), hb.stride(0), _DT[self.cd], B, dh.shape[1], K.dptr(self.g_b1), s))
        torch.mm(xb.t(), dh, out=self._gw1_out)                       # [Kp, H]; the padding rows are never read
        if not self.direct:  # pack the all-reduce bucket (fp32, unpadded)
            self.g_w2.copy_(self._gw2_out[:, :self.AK])
            self.g_b2.copy_(self._gb2_pad[:self.AK])
            self.g_w1.copy_(self._gw1_out[:self.L])
        return self.td, self.w_is

    def _adam_table(self):
        """ctypes array describing the four merged tensors (built once: all pointers are persistent)."""
        if self._adam_tab is None:
            if self.direct:   # (tensor, dtype code, row stride)
                grads = (((self._gw1_out, _DT[self.cd], self._gw1_out.shape[1]), (self.g_b1, 0, 0)),
                         ((self._gw2_out, _DT[self.cd], self.Np), (self._gb2_pad, 0, 0)))
            else:
                grads = (((self.g_w1, 0, 0), (self.g_b1, 0, 0)), ((self.g_w2, 0, 0), (self.g_b2, 0, 0)))
            tab = (K.HbAdamTensor * 4)()
            k = 0
            for li, l in enumerate(self.layers):
                for names, noise, g, eff in ((("w", "w_mu", "w_sigma"), l.eps_w, grads[li][0], self.eff[li][0]),
                                             (("b", "b_mu", "b_sigma"), l.eps_b, grads[li][1], self.eff[li][1])):
                    ps = [getattr(l, n) for n in names]
                    st = [self.state[(li, n)] for n in names]
                    d = tab[k]
                    d.w, d.w_mu, d.w_sigma = (p.data_ptr() for p in ps)
                    d.noise, d.grad, d.grad_dtype, d.grad_ld = noise.data_ptr(), g[0].data_ptr(), g[1], g[2]
                    d.m_w, d.v_w = st[0][0].data_ptr(), st[0][1].data_ptr()
                    d.m_mu, d.v_mu = st[1][0].data_ptr(), st[1][1].data_ptr()
                    d.m_sigma, d.v_sigma = st[2][0].data_ptr(), st[2][1].data_ptr()
                    d.eff, d.n, d.cols = eff.data_ptr(), ps[0].numel(), ps[0].shape[-1]
                    d.eff_ld = eff.stride(0) if eff.dim() == 2 else eff.shape[-1]   # eff may be a column block of a wider buffer
                    k += 1
            self._adam_tab = tab
        return self._adam_tab

    def _pack_table(self):
        """hb_adam_pack entries for the four merged tensors (order of _adam_table: W1, b1, W2, b2), built once."""
        if self._pack_tab is None:
            import ctypes

            n_log = self.A * self.Kk
            host = (ctypes.c_int32 * n_log)()
            K.check(K.lib().hb_actor_fused_columns(self.A, host))
            self._col_map = torch.tensor(list(host), dtype=torch.int32, device=self.agent.device)
            f = self.actor.fsets[0]          # (w1f, b1f, w2f, b2f) of weight set 0
            tab = (K.HbAdamPack * 4)()
            tab[0].wt, tab[0].wt_ld, tab[0].frag, tab[0].frag_kind = self.w1catT.data_ptr(), self.Kp, f[0].data_ptr(), 1
            tab[1].bias_f32 = f[1].data_ptr()
            tab[2].wt, tab[2].wt_ld, tab[2].frag, tab[2].frag_kind = self.w2stT[0].data_ptr(), self.H, f[2].data_ptr(), 2
            tab[2].col_map_dev = self._col_map.data_ptr()
            tab[3].bias_f32, tab[3].col_map_dev = f[3].data_ptr(), self._col_map.data_ptr()
            self._pack_tab = tab
        return self._pack_tab

    def part2(self):
        p = self.agent.params
        if self.adam_pack:
            # one launch: optimizer step + the thin GEMMs' transposed online weights + the one-kernel actor's copies (set 0)
            K.check(K.lib().hb_noisy_adam_multi_pack(self._adam_table(), self._pack_table(), 4, K.dptr(self.step), 0.0, _DT[self.cd],
                                                     float(p.learning_rate), 0.9, 0.999, 3.125e-5, K.current_stream()))
            self._packed_in_part2 = True     # (the two-kernel actor form's copies: lazily, weights_updated() marks them)
            return
        # self.step was advanced by this update's loss kernel (part1): it already is this step's number
        K.check(K.lib().hb_noisy_adam_multi(self._adam_table(), 4, K.dptr(self.step), 0.0, _DT[self.cd],
                                            float(p.learning_rate), 0.9, 0.999, 3.125e-5, K.current_stream()))
        # Synchronous agent: the actor's weight copies follow Adam right here — on the learner's stream, inside the captured
        # update — instead of on the acting stream before the next policy call (round 2: two launches and ~12 us per step there).
        # Sound because an update starts only after its agent's last policy call has finished (`acted` event / same stream) and
        # the next one waits for this update (weights_ev / the stream). actor_lag: weights_updated() packs the alternate set.
        self._packed_in_part2 = False
        if self.actor is not None and not self.lag:
            (w1, b1), (w2, b2) = self.eff
            self.actor.pack(w1, b1, w2, b2, 0, lazy_two_kernel=True, thin=self._thin_out())
            self._packed_in_part2 = True

    def loss(self):
        return torch.mean(self.td * self.w_is)
