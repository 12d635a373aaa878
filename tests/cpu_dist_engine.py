"""TEST-ONLY CPU stand-in for cmlpl_amd.distributed.DistTrainEngine: same stage protocol
(STAGES / stage_* / exchange_after / waits_before, same packed-buffer layouts), with the per-shard math written in
PyTorch-CPU from the oracle's formulas.  It lets the multi-process wiring (drive_step + TorchDistComm
over gloo) be exercised without a GPU, and doubles as an independent statement of the sharded algorithm."""
import torch
import torch.nn.functional as F

from oracle import cmlpl_oracle as O

FD = O.FEAT_DIM


class CpuDistEngine:
    STAGES = ("spectral", "spatial", "phase1", "phase2", "backward_data", "backward_weights", "update")

    def __init__(self, shape, bt_l, btu_l, hp, world, rank, params0, params1):
        self.shape, self.hp, self.world, self.rank = shape, hp, world, rank
        self.bt_l, self.btu_l, self.n_l = bt_l, btu_l, bt_l + btu_l
        W, K = world, shape.K
        self.bt_g, self.btu_g, self.n_g = bt_l * W, btu_l * W, (bt_l + btu_l) * W
        self.params = [{k: v.clone() for k, v in params0.items()}, {k: v.clone() for k, v in params1.items()}]
        self.adam = [O.AdamState({k: torch.zeros_like(p[k]) for k in O.LIVE_KEYS},
                                 {k: torch.zeros_like(p[k]) for k in O.LIVE_KEYS}) for p in self.params]
        Q = hp.bank_mult * self.bt_g * 2
        self.Q, self.ptr = Q, [0, 0]
        self.bank_feats = [torch.zeros(Q, FD), torch.zeros(Q, FD)]
        self.bank_probs = [torch.zeros(Q, K), torch.zeros(Q, K)]
        self.sizes = [self.params[0][k].numel() for k in O.LIVE_KEYS]
        self.grads = torch.zeros(2, sum(self.sizes))
        self.scalars = torch.zeros(16)
        self.rebind(bt_l, btu_l)

    def rebind(self, bt_l, btu_l):
        """(Re)allocate the buffers whose size follows the shard of the current step (a short last batch)."""
        W, K = self.world, self.shape.K
        self.bt_l, self.btu_l, self.n_l = bt_l, btu_l, bt_l + btu_l
        self.bt_g, self.btu_g, self.n_g = bt_l * W, btu_l * W, (bt_l + btu_l) * W
        n_l = self.n_l
        self.pack_len = 2 * n_l * FD + bt_l                  # this rank's block of the exchange buffer: [feat | labels]
        self.pack, self.recv = torch.zeros(self.pack_len), torch.zeros(W * self.pack_len)
        self.logits_l = torch.zeros(2, n_l, K)               # never exchanged
        self.probs_l, self.probs_g = torch.zeros(4, btu_l, K), torch.zeros(W, 4, btu_l, K)
        self.dlogits_l, self.dfeat_l = torch.zeros(2, n_l, K), torch.zeros(2, n_l, FD)
        self.dfw_part = torch.zeros(self.btu_g, FD)

    # --------------------------------------------------------------
    def stage_spectral(self, XPl, Xl, Y, XPu, Xu, epoch, batch_index, noise, dropmask, apply_update=True):
        """the embeddings from the spectral branch ALONE (tools/models.py:142-146): feat = l2norm(relu(feat_spe(x)))"""
        sg = self.hp.noise
        self.xp = [torch.cat([XPl + noise[0] * sg, XPu + noise[4] * sg]), torch.cat([XPl + noise[2] * sg, XPu + noise[6] * sg])]
        self.xs = [torch.cat([Xl + noise[1] * sg, Xu + noise[5] * sg]), torch.cat([Xl + noise[3] * sg, Xu + noise[7] * sg])]
        self.dropmask = dropmask
        self.leaf = [{k: (v.clone().requires_grad_(True) if k in O.LIVE_KEYS else v) for k, v in p.items()}
                     for p in self.params]
        n_l = self.n_l
        early = []
        for net in range(2):
            y = torch.relu(F.linear(self.xs[net], self.leaf[net]["feat_spe.weight"], self.leaf[net]["feat_spe.bias"]))
            early.append((y / y.pow(2).sum(1, keepdim=True).sqrt()).detach())
        self.feat_early = torch.stack(early)
        self.pack[:2 * n_l * FD] = self.feat_early.reshape(-1)
        self.pack[2 * n_l * FD:] = Y.float()
        self.ctx = dict(smooth=(epoch > 0) or (batch_index > self.hp.queue_batch),
                        adap=self.hp.thr * self.hp.adap_thr(epoch), apply_update=apply_update)

    def stage_spatial(self):
        """the whole forward (the logits need both branches); its embeddings must be the ones already sent"""
        self.fwd = [O.basenet2_forward(self.leaf[net], self.xp[net], self.xs[net], self.dropmask[net]) for net in range(2)]
        self.logits_l.copy_(torch.stack([f[0].detach() for f in self.fwd]))
        late = torch.stack([f[1].detach() for f in self.fwd])
        assert torch.allclose(late, self.feat_early, rtol=1e-6, atol=1e-7, equal_nan=True)

    def _unpack(self):
        W, n_l, bt_l = self.world, self.n_l, self.bt_l
        r = self.recv.view(W, self.pack_len)
        fe = r[:, :2 * n_l * FD].reshape(W, 2, n_l, FD)
        lab = r[:, 2 * n_l * FD:]
        glob = lambda t, d: torch.cat([t[:, :, :bt_l].permute(1, 0, 2, 3).reshape(2, -1, d),
                                       t[:, :, bt_l:].permute(1, 0, 2, 3).reshape(2, -1, d)], dim=1)
        self.feat_g = glob(fe, FD).contiguous()
        self.labels_g = (lab.reshape(-1) + 0.5).long()

    def stage_phase1(self):
        self._unpack()
        hp, T = self.hp, self.hp.temperature
        bt_g, btu_g, bt_l, btu_l, r = self.bt_g, self.btu_g, self.bt_l, self.btu_l, self.rank
        # only THIS rank's logits exist here (local rows [labelled ; unlabelled]); embeddings and labels are global
        lab, unl = slice(0, bt_l), slice(bt_l, bt_l + btu_l)
        gunl = slice(bt_g + r * btu_l, bt_g + (r + 1) * btu_l)
        z = [self.logits_l[net].clone().requires_grad_(True) for net in range(2)]     # [s, w]
        f = self.feat_g
        Yl = self.labels_g[r * bt_l:(r + 1) * bt_l]
        with torch.no_grad():
            p_w0, p_s0 = torch.softmax(z[1][unl], 1), torch.softmax(z[0][unl], 1)
            p_w, p_s = p_w0.clone(), p_s0.clone()
            if self.ctx["smooth"]:
                A = torch.exp(f[1][gunl] @ self.bank_feats[0].t() / T); A = A / A.sum(1, keepdim=True)
                p_w = hp.alpha * p_w + (1 - hp.alpha) * (A @ self.bank_probs[0])
                A1 = torch.exp(f[0][gunl] @ self.bank_feats[1].t() / T); A1 = A1 / A1.sum(1, keepdim=True)
                p_s = hp.alpha * p_s + (1 - hp.alpha) * (A1 @ self.bank_probs[1])
            m_w = (p_w.max(1)[0] >= torch.tensor(self.ctx["adap"], dtype=torch.float32)).float()
            m_s = (p_s.max(1)[0] >= torch.tensor(self.ctx["adap"], dtype=torch.float32)).float()
        cls = [F.cross_entropy(z[net][lab], Yl, reduction="sum") / bt_g for net in range(2)]
        con_s = (-(F.log_softmax(z[0][unl], 1) * p_w).sum(1) * m_w).sum() / btu_g
        con_w = (-(F.log_softmax(z[1][unl], 1) * p_s).sum(1) * m_s).sum() / btu_g
        g = torch.autograd.grad([cls[0] + hp.w_mutual * con_s, cls[1] + hp.w_mutual * con_w], z)
        for net in range(2):
            self.dlogits_l[net] = torch.cat([g[net][lab], g[net][unl]])
        self.probs_l.copy_(torch.stack([p_w, p_s, p_w0, p_s0]))
        acc = (z[1][lab].argmax(1) == Yl).float().sum() / bt_g
        self.p1 = dict(cls=cls, con=(con_s, con_w), acc=acc, m=(m_w.sum(), m_s.sum()))

    def stage_phase2(self):
        hp, T = self.hp, self.hp.temperature
        bt_g, btu_g, bt_l, btu_l, r = self.bt_g, self.btu_g, self.bt_l, self.btu_l, self.rank
        pg = self.probs_g.permute(1, 0, 2, 3).reshape(4, btu_g, -1)            # [4][btu_g][K] global order
        rows = slice(r * btu_l, (r + 1) * btu_l)
        fs_l = self.feat_g[0][bt_g:][rows].clone().requires_grad_(True)        # rows path (ctr_s)
        fw_g = self.feat_g[1][bt_g:].clone().requires_grad_(True)              # columns path (ctr_w)
        with torch.no_grad():
            Q0 = pg[1][rows] @ pg[0].t()
            Q0[torch.arange(btu_l), torch.arange(btu_l) + r * btu_l] = 1.0      # GLOBAL diagonal
            Qp = Q0 * (Q0 >= hp.pos_thr).float(); Qp = Qp / Qp.sum(1, keepdim=True)
            Qn = (1 - Q0) * (Q0 <= hp.neg_thr).float(); Qn = Qn / (Qn.sum(1, keepdim=True) + 1e-8)
        def ctr(fa, fb):
            S = torch.exp(fa @ fb.t() / T); P = S / S.sum(1, keepdim=True)
            return ((-(torch.log(P) * Qp).sum(1)).sum() + ((torch.log(P + 1) * Qn).sum(1)).sum()) / btu_g
        ctr_s = ctr(fs_l, fw_g.detach())
        ctr_w = ctr(fs_l.detach(), fw_g)
        (gs,) = torch.autograd.grad(hp.w_contrast * ctr_s, fs_l)
        (gw,) = torch.autograd.grad(hp.w_contrast * ctr_w, fw_g)
        self.dfeat_l.zero_()
        self.dfeat_l[0, bt_l:] = gs
        self.dfw_part.copy_(gw)
        # bank write of the GLOBAL batch (identical on every rank)
        f, Y = self.feat_g, self.labels_g
        onehot = torch.zeros(bt_g, self.shape.K).scatter(1, Y.view(-1, 1), 1)
        O.bank_write(self.bank_feats[0], self.ptr[0], torch.cat([f[1][bt_g:], f[0][:bt_g]]))
        O.bank_write(self.bank_probs[0], self.ptr[0], torch.cat([pg[2], onehot]))
        O.bank_write(self.bank_feats[1], self.ptr[1], torch.cat([f[0][bt_g:], f[1][:bt_g]]))
        O.bank_write(self.bank_probs[1], self.ptr[1], torch.cat([pg[3], onehot]))
        p1 = self.p1
        c = float(ctr_s)
        self.scalars[:9] = torch.tensor([
            c, float(p1["cls"][0]) + hp.w_contrast * c + hp.w_mutual * float(p1["con"][0]), float(p1["cls"][0]),
            float(p1["con"][0]), float(p1["acc"]),
            float(p1["cls"][1]) + hp.w_contrast * c + hp.w_mutual * float(p1["con"][1]), float(p1["cls"][1]),
            float(p1["con"][1]), c])

    def _leaf_grads(self, outs, gouts, retain):
        leaves = [self.leaf[net][k] for net in range(2) for k in O.LIVE_KEYS]
        gr = torch.autograd.grad(outs, leaves, grad_outputs=gouts, retain_graph=retain, allow_unused=True)
        nk = len(O.LIVE_KEYS)
        return torch.stack([torch.cat([(g if g is not None else torch.zeros_like(l)).reshape(-1)
                                       for g, l in zip(gr[net * nk:(net + 1) * nk], leaves[net * nk:(net + 1) * nk])])
                            for net in range(2)])

    def stage_backward_data(self):
        """everything that flows back from the logits (needs dlogits only; the reduce-scatter is still in flight)"""
        self.g_logits = self._leaf_grads([self.fwd[net][0] for net in range(2)], [self.dlogits_l[net] for net in range(2)], True)

    def stage_backward_weights(self):
        """+ what flows back from the embeddings (dfeat complete): the gradient is linear in the two"""
        g_feat = self._leaf_grads([self.fwd[net][1] for net in range(2)], [self.dfeat_l[net] for net in range(2)], False)
        self.grads.copy_(self.g_logits + g_feat)

    def stage_update(self):
        if self.ctx["apply_update"]:
            for net in range(2):
                st = self.adam[net]
                st.t += 1
                off = 0
                for k, sz in zip(O.LIVE_KEYS, self.sizes):
                    g = self.grads[net, off:off + sz].view_as(self.params[net][k])
                    O.adam_update(self.params[net][k], g, st.m[k], st.v[k], st.t, self.hp)
                    off += sz
        q = self.Q
        p0 = (self.ptr[0] + self.hp.bank_step) % q
        self.ptr = [p0, (p0 + self.hp.bank_step) % q]

    def exchange_after(self, stage):
        if stage == "spectral":
            return [("all_gather", self.recv, self.pack, "feat")]
        if stage == "phase1":
            return [("all_gather", self.probs_g, self.probs_l, None)]
        if stage == "phase2":
            return [("reduce_scatter", self.dfeat_l[1, self.bt_l:], self.dfw_part, "dfw")]
        if stage == "backward_weights":
            return [("all_reduce", self.grads, None, None)]
        return []

    def waits_before(self, stage):
        return {"phase1": ["feat"], "backward_weights": ["dfw"]}.get(stage, [])


class CpuLoopEngine:
    """TEST-ONLY: the part of DistTrainEngine's interface that train.py's loop uses (step / loss_window /
    step_count / state_dict / init_params_default), on top of CpuDistEngine + the real drive_step and communicator.
    Lets `train.py` run as two gloo ranks on CPU, so the sharding decisions of its loop (equal shards decided on the
    global batch, the short last batch, the loss_hist windows) are exercised across real processes."""

    def __init__(self, shape, bt_l, btu_l, hp, comm, hist_rows=1):
        from cmlpl_amd.distributed import drive_step
        self._drive = drive_step
        self.oshape = O.NetShape(shape.C, shape.H, shape.W, shape.bands, shape.K)
        self.hp = O.HyperParams(**{k: getattr(hp, k) for k in O.HyperParams.__dataclass_fields__})
        self.comm, self.bt_max, self.btu_max = comm, bt_l, btu_l
        self.n_max = bt_l + btu_l
        self.hist_rows, self.rows, self.step_count = hist_rows, [], 0
        self.eng = None

    def init_params_default(self, seed=1088):
        p0, p1 = O.closed_form_params(self.oshape, seed), O.closed_form_params(self.oshape, seed + 1)
        self.eng = CpuDistEngine(self.oshape, self.bt_max, self.btu_max, self.hp, self.comm.world, self.comm.rank, p0, p1)

    def step(self, XPl, Xl, Y, XPu, Xu, epoch, batch_index):
        e = self.eng
        bt_l, btu_l = XPl.shape[0], XPu.shape[0]
        if bt_l > self.bt_max or btu_l > self.btu_max:
            raise ValueError("shard larger than the engine's capacity")
        if (bt_l, btu_l) != (e.bt_l, e.btu_l):
            e.rebind(bt_l, btu_l)
        zl = [torch.zeros_like(XPl), torch.zeros_like(Xl)] * 2 + [torch.zeros_like(XPu), torch.zeros_like(Xu)] * 2
        self._drive(e, self.comm, XPl, Xl, Y, XPu, Xu, epoch, batch_index, zl, [None, None])
        row = e.scalars.clone()
        self.comm.all_reduce(row)
        self.rows.append(row)
        self.step_count += 1

    def loss_window(self, k):
        assert 1 <= k <= self.hist_rows and k <= self.step_count
        return torch.stack(self.rows[-k:])[:, :5].double().numpy()

    def state_dict(self, net):
        return {k: v.clone() for k, v in self.eng.params[net].items()}
