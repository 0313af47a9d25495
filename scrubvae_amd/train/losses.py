"""get_batch_loss for the MI355X ResVAE (reference: src/scrubvae/train/losses.py:182-324).

Same signature and dict contract as the reference:

    batch_loss = get_batch_loss(model, data, data_o, loss_scale, disentangle_config)
    batch_loss["total"].backward()

Every term is computed by HIP kernels; each kernel that evaluates a loss also writes the
seed gradient of ``loss_scale[k] * loss_k`` with respect to the network output it reads
(fused forward+backward of the loss), and ``total.backward()`` then runs the model's
reverse schedule from those seeds.  Dispatch is by key presence in ``loss_scale`` exactly
as in the reference (losses.py:186-219,311), including its quirks: JPE uses root_hat=None
(:209-214), the rotation loss is not divided by the batch (:136), the gradient-reversal
ensemble is normalised inside the loop (:279-284), the adversarial net applies
CrossEntropyLoss to its own softmax output (:304-307).
"""
from __future__ import annotations

import os

import torch

from .. import ops
from ..ops import pad16

_TAIL_FORK = os.environ.get("SVAE_TAIL_FORK", "1") != "0"  # env: schedule experiments (0 = the tail stays on the main stream)
_EARLY_SCRUB = os.environ.get("SVAE_EARLY_SCRUB", "1") != "0"  # (0 = the heads' backward stays in total.backward())

SUPPORTED = ("rotation", "prior", "jpe", "root", "total_correlation")


class _LossPack:
    """The loss terms of one get_batch_loss call as slots of ONE device buffer: every term is reduced straight into its slot, the
    running `total` of the reference (losses.py:320-322) is one `svae_loss_total` launch over the slots, and the dict handed back
    holds 0-dim views of one snapshot of the buffer -- instead of an axpy and a clone per term."""

    def __init__(self, model, loss_scale):
        from .._lib import MAX_LOSS_TERMS
        self.buf = model._buf("loss.pack", (MAX_LOSS_TERMS + 1,), zero=True)
        self.loss_scale = loss_scale
        self.keys, self.weights = [], []

    def slot(self, key):
        """1-element view for loss term `key` (None: an unnamed spare slot); its weight in the total is loss_scale[key] (0 if absent)"""
        i = len(self.keys)
        if i >= self.buf.numel() - 1:
            raise RuntimeError("too many loss terms for one get_batch_loss call")
        self.keys.append(key)
        self.weights.append(float(self.loss_scale.get(key, 0.0)) if key is not None else 0.0)
        return self.buf[i: i + 1]

    def finish(self, batch_loss):
        """total into the slot behind the last term, one snapshot, views into `batch_loss` (dict order = the order of slot() calls
        that registered a placeholder); returns the 0-dim total"""
        n = len(self.keys)
        ops.loss_total(self.buf, self.weights, self.buf[n: n + 1])
        snap = self.buf[: n + 1].clone()
        for i, k in enumerate(self.keys):
            if k is not None:
                batch_loss[k] = snap[i]
        return snap[n]


def get_batch_loss(model, data, data_o, loss_scale, disentangle_config, adv_perm=None):
    """adv_perm: optional dict feature -> LongTensor permutation replacing torch.randperm in
    AdvNetScrubber.shuffle (disentangle.py:680) -- for parity tests."""
    st = getattr(model, "_state", None)
    if st is None or "klp" not in st:
        raise RuntimeError("get_batch_loss needs the data_o of the immediately preceding model(data) call")
    B = st["B"]
    W, J, z = model.window, model.n_keypts, model.z_dim
    zp = pad16(z)
    world = model.world_size
    Bg = B * world  # losses are normalised by the GLOBAL batch so that summed grads match 1 GPU
    train = model.training and torch.is_grad_enabled()
    for k in loss_scale.keys():
        if k in SUPPORTED or k == "mcmi" or k.endswith(("_gr", "_an", "_mals", "_ma", "_qda", "_lsq", "_lin")):
            continue
        raise NotImplementedError(f"loss '{k}' is outside this build's scope (SURVEY 8a: L5/A2 rows)")
    batch_loss = {}
    pack = _LossPack(model, loss_scale)
    dev_data = {k: v for k, v in data.items()}

    def term(key):
        """device slot of loss term `key`, registered in batch_loss in call order (filled from the snapshot at the end); it enters
        the total with weight loss_scale[key] when that key exists and is non-zero, as in the reference"""
        batch_loss[key] = None
        return pack.slot(key)

    # ---- rotation (L4): seeds an extra gradient on x6d_hat that the tail adds
    ext_dx6d = None
    if "rotation" in loss_scale:
        if not getattr(model, "_tail_done", True):  # deferred tail: the rotation loss reads x6d_hat
            model._run_tail(B, dev_data, 0.0, 0.0, None, False)
        n = B * W * J
        part = model._buf("rot.part", (ops.rot_blocks(n),))
        x6d = model._prep(data["x6d"])
        x6d_hat = model._buf("out.x6d", (B, W, J, 6))
        ext_dx6d = model._buf("rot.dx", (B * W, J * 6)) if train else None
        ops.rot_loss(x6d, x6d_hat, float(loss_scale["rotation"]), part, ext_dx6d, n)
        ops.reduce_rows(part, part.numel(), 1, 1.0, term("rotation"))

    # ---- prior (L3): KL partials were produced by the heads kernel in forward
    kl_scale = 0.0
    if "prior" in loss_scale:
        ops.reduce_rows(st["klp"], st["klp"].numel(), 1, 1.0 / Bg, term("prior"))
        kl_scale = float(loss_scale["prior"]) / Bg

    # ---- jpe (L1) + root (L2): fused tail, also produces d total / d conv_out
    jpe_s = float(loss_scale.get("jpe", 0.0)) / (Bg * 3 * J)
    root_s = float(loss_scale.get("root", 0.0)) / Bg
    if "jpe" in loss_scale and ("offsets" not in data or "target_pose" not in data):
        raise KeyError("jpe loss needs data['offsets'] and data['target_pose']")
    # The tail kernel is latency-bound (one 6-wave workgroup per CU, ~0.6 ms at B = 4096: DESIGN.md 4) and nothing behind it on the
    # main chain can start before it ends -- except the scrubbing losses below, which only read the encoder's outputs.  Where the
    # pass overlaps streams the tail is forked to a side stream (a C-ABI launch; its inputs are prepared here, on the main stream)
    # and joined after the scrubbing section; the loss terms keep their place in `batch_loss` and in the running total.
    for k in ("offsets", "target_pose", "root"):
        if k in dev_data and torch.is_tensor(dev_data[k]):
            dev_data[k] = model._prep(dev_data[k])
    tail_out = {}

    def _tail():
        tail_out["r"] = model._run_tail(B, dev_data, jpe_s, root_s, ext_dx6d, train)

    fork_tail = model._ov == 2 and _TAIL_FORK
    if fork_tail:
        model._tail_prealloc(B, dev_data, train)  # first-use allocations / zero fills happen on the main stream, before the fork
        model._fork(_tail, k=1)
    else:
        _tail()
    # two adjacent slots (dict order as the serial schedule writes it); a term that is not configured keeps an unnamed spare slot
    for k in ("jpe", "root"):
        term(k) if k in loss_scale else pack.slot(None)
    tail_slots = pack.buf[len(pack.keys) - 2: len(pack.keys)]

    def join_tail():
        model._join_side(1)
        lp = tail_out["r"][2]
        if "jpe" in loss_scale or "root" in loss_scale:  # the two column sums of the tail's partials, each with its own normalisation
            ops.reduce_rows_scaled(lp, lp.shape[0], [1.0 / (Bg * 3 * J), 1.0 / Bg], tail_slots)
        return tail_out["r"][3]

    if not fork_tail:  # serial schedule: the terms enter the total in the reference's order
        dy = join_tail()

    # ---- scrubbing losses
    d_mu = model._buf("seed.d_mu", (B, zp), zero=True)
    if train:
        d_mu.zero_()
    scrub = []
    # `linear` method: every other scrubber of a feature reads the null-space projection z_null(mu, W) of that feature
    # instead of mu (losses.py:232-235); gradients then go through the projection's torch graph to mu and to its decoder
    lin = st.get("lin")
    lin_pgrads = {}

    def latent_graph(key):
        """(latent [B, z] to evaluate a torch-graph loss on, leaves to differentiate with respect to; leaves[0] is mu)"""
        if lin is not None:
            return lin["out"][key]["z_null"], lin["leaves"][key]
        t = st["mu"][:, :z].detach().clone().requires_grad_(bool(train))
        return t, [t]

    def push(leaves, grads, sc=1.0):
        d_mu[:, :z] += sc * grads[0]
        for p, g in zip(leaves[1:], grads[1:]):
            if g is not None:
                lin_pgrads[id(p)] = (p, lin_pgrads[id(p)][1] + sc * g if id(p) in lin_pgrads else sc * g)

    def push_latent_seed(key, seed):
        """seed = d total / d latent [B, z] (analytic)"""
        if lin is None:
            d_mu[:, :z] += seed
        else:
            lat, leaves = latent_graph(key)
            push(leaves, torch.autograd.grad(lat, leaves, grad_outputs=seed, retain_graph=True, allow_unused=True))

    # ---- mcmi (losses.py:221-225): KDE mutual information between mu and the conditioning variables under the
    # estimator built from the previous batch; zero (shaped like the jpe term, as in the reference) before the first refresh
    if "mcmi" in loss_scale:
        v = term("mcmi")
        if model.mi_estimator is not None:
            sc = float(loss_scale["mcmi"])
            mu_t = st["mu"][:, :z].detach().clone().requires_grad_(bool(train and sc != 0))
            with torch.enable_grad():
                val = model.mi_estimator(mu_t, data_o["var"]) / world  # mean over the GLOBAL batch once summed over ranks
            v.copy_(val.detach().reshape(1))
            if train and sc != 0:
                d_mu[:, :z] += sc * torch.autograd.grad(val, mu_t)[0]
        else:
            v.zero_()
    methods = disentangle_config["method"] if disentangle_config is not None else {}
    for method, keys in methods.items():
        nk = len(keys)
        for key in keys:
            if method == "conditional":
                continue
            if method == "grad_reversal":
                m = model.disentangle[method][key]
                runner = model._runner(method, key, m.ensemble, B)
                outs = runner.outs if runner.fused else [l[-1]["pre"] for l in runner.members]
                out_dim, out_p = m.ensemble.out_dim, pad16(m.ensemble.out_dim)
                c = 4.0 * nk * Bg
                weights = [c ** -4, c ** -3, c ** -2, c ** -1]  # normalisation inside the loop (losses.py:279-284)
                lk = key + "_gr"
                v = term(lk)
                nb_r = ops.rowloss_blocks(B)
                if runner.fused:  # the four members' losses and seed gradients in one launch, one reduction
                    part = model._buf("gr.part4", (4 * nb_r,))
                    d_outs = runner.d_outs if train else [None] * 4
                    gs = [float(loss_scale[lk]) * w for w in weights]
                    if key == "ids":
                        labels = data[key].to(model.device).ravel().int().contiguous()
                        ops.ens_loss(1, outs, d_outs, weights, gs, None, 0, labels, B, out_dim, out_p, part)
                    else:
                        tgt = model._prep(data[key])
                        ops.ens_loss(0, outs, d_outs, weights, gs, tgt, tgt.shape[-1], None, B, out_dim, out_p, part)
                    ops.reduce_rows(part, 4 * nb_r, 1, 1.0, v)
                else:
                    v.zero_()
                    d_outs = []
                    for i, (o, w) in enumerate(zip(outs, weights)):
                        part = model._buf(f"gr.part.{i}", (nb_r,))
                        dpred = model._buf(f"gr.{key}.d{i}", (B, out_p), zero=True) if train else None
                        sc = float(loss_scale[lk]) * w
                        if key == "ids":
                            labels = data[key].to(model.device).ravel().int().contiguous()
                            ops.ce_sum(o, out_p, labels, B, out_dim, sc, part, dpred)
                        else:
                            tgt = model._prep(data[key])
                            ops.mse_sum(o, out_p, tgt, tgt.shape[-1], B, out_dim, sc, part, dpred)
                        ops.reduce_rows(part, nb_r, 1, w, v, accumulate=True)
                        d_outs.append(dpred)
                if train and loss_scale[lk] != 0:
                    scrub.append(dict(kind="gr", runner=runner, d_outs=d_outs, alpha=m.alpha,
                                      lin=None if lin is None else (lin["out"][key]["z_null"], lin["leaves"][key])))
            elif method == "adversarial_net":
                m = model.disentangle[method][key]
                v_ind = model.disentangle_keys.index(key)
                var = data_o["var"]
                shuf_vals = None
                if world > 1:
                    # data parallel: ONE permutation of the GLOBAL batch (every rank draws the same one from the shared seed),
                    # so that N ranks shuffle exactly like one rank at the global batch; the shuffled column's values come
                    # from the other ranks through a [Bg] all-gather
                    perm_g = adv_perm[key] if adv_perm is not None else model.global_permutation(Bg)
                    col = model._allgather(var[:, v_ind].float().contiguous())
                    shuf_vals = col[perm_g[model.rank * B: (model.rank + 1) * B].to(model.device)].contiguous()
                    perm = None
                else:
                    perm = adv_perm[key].to(model.device) if adv_perm is not None else torch.randperm(B, device=model.device)
                mu = st["mu"]
                D = model.conditional_dim
                runner = model._runner(method, key, m.ensemble, B, halves=2)
                lk = key + "_an"
                vv = term(lk)
                nb_r = ops.rowloss_blocks(2 * B)
                w = -1.0 / (4 * Bg)
                if runner.fused:
                    # cat([mu;mu], [v;v_shuffle]) is assembled inside the kernel: one COLUMN of var is shuffled (disentangle.py:680)
                    outs = runner.forward(mu, z, src1=model._var32(var), perm=None if perm is None else perm.long().contiguous(),
                                          shuf_col=v_ind, shuf_vals=shuf_vals)
                    part = model._buf("an.part4", (4 * nb_r,))
                    d_outs = runner.d_outs if train else [None] * 4
                    ops.ens_loss(2, outs, d_outs, [w] * 4, [float(loss_scale[lk]) * w] * 4, None, 0, None, 2 * B, 2, 16, part)
                    ops.reduce_rows(part, 4 * nb_r, 1, 1.0, vv)
                else:
                    in_p = pad16(m.ensemble.in_dim)
                    x = model._buf(f"an.{key}.x", (2 * B, in_p), zero=True)
                    x[:B, :z] = mu[:, :z]
                    x[B:, :z] = mu[:, :z]
                    x[:B, z: z + D] = var
                    x[B:, z: z + D] = var
                    x[B:, z + v_ind] = var[perm, v_ind] if shuf_vals is None else shuf_vals  # one COLUMN of var is shuffled (disentangle.py:680)
                    outs = runner.forward(x)
                    vv.zero_()
                    d_outs = []
                    for i, o in enumerate(outs):
                        part = model._buf(f"an.part.{i}", (nb_r,))
                        dl = model._buf(f"an.{key}.d{i}", (2 * B, 16), zero=True) if train else None
                        ops.double_softmax_ce_sum(o, 16, 2 * B, float(loss_scale[lk]) * w, part, dl)
                        ops.reduce_rows(part, nb_r, 1, w, vv, accumulate=True)
                        d_outs.append(dl)
                if train and loss_scale[lk] != 0:
                    scrub.append(dict(kind="an", runner=runner, d_outs=d_outs))
            elif method == "moving_avg_lsq":  # losses.py:237-246
                m = model.disentangle[method][key]
                lk = key + "_mals"
                y0, y1 = data_o["disentangle"][method][key]
                tgt = model._prep(data[key])
                v = term(lk)
                v.copy_((m.evaluate_loss(y0, y1, tgt) / Bg).reshape(1))
                if lk in loss_scale:
                    if train and loss_scale[lk] != 0:  # d/d mu of scale * (l0 + l1) / (2 Bg), decoders W constant
                        lat = st["mu"][:, : m.nx_in] if lin is None else lin["out"][key]["z_null"]
                        seed = m.latent_seed(y0, y1, tgt, lat.detach(), float(loss_scale[lk]) / Bg)
                        push_latent_seed(key, seed)
            elif method == "direct_lsq":  # direct_lsq_loss, losses.py:173-179,254-257: stateless least-squares decoder
                lk = key + "_lsq"
                sc = float(loss_scale[lk])
                zm = st["mu"][:, :z] if lin is None else lin["out"][key]["z_null"].detach()
                if sc < 0:  # bias column
                    zm = torch.column_stack((zm, torch.ones(B, 1, device=zm.device)))
                tgt = model._prep(data[key])
                zz, zy = zm.T @ zm, zm.T @ tgt
                if world > 1:  # normal equations of the GLOBAL batch
                    model._allreduce(zz)
                    model._allreduce(zy)
                Wd = ops.small_solve(zz, zy) if (zz.is_cuda and zz.shape[0] <= 64 and zy.shape[1] <= 64) else torch.linalg.solve(zz, zy)
                res = zm @ Wd - tgt
                v = term(lk)
                v.copy_((res * res).sum().reshape(1))
                if train and sc != 0:
                    # d/d mu of the summed squared residual: the decoder is the minimiser, so its own dependence on mu
                    # contributes nothing (dL/dW = 0) and the gradient is 2 * res * W^T
                    push_latent_seed(key, (2.0 * sc) * (res @ Wd[:z].T))
            elif method in ("moving_avg", "qda"):  # losses.py:248-252,286-289
                m = model.disentangle[method][key]
                lk = key + ("_ma" if method == "moving_avg" else "_qda")
                sc = float(loss_scale.get(lk, 0.0))
                mu_t, leaves = latent_graph(key)
                tgt = data[key].to(model.device)
                with torch.set_grad_enabled(bool(train and sc != 0)):
                    val = m.evaluate_loss(mu_t, tgt)
                    if method == "qda":
                        val = val / Bg
                v = term(lk)
                v.copy_(val.detach().reshape(1))
                if lk in loss_scale:
                    if train and sc != 0:  # seed of the HIP backward: d(scale * loss) / d mu through the small torch graph
                        push(leaves, torch.autograd.grad(val, leaves, retain_graph=lin is not None, allow_unused=True), sc)
            elif method == "linear":  # losses.py:258-265: the projection's own decoder regresses the feature
                lk = key + "_lin"
                sc = float(loss_scale.get(lk, 0.0))
                pred = lin["out"][key]["v"]
                tgt = model._prep(data[key])
                with torch.set_grad_enabled(bool(train and sc != 0)):
                    val = ((pred - tgt) ** 2).sum() / nk / Bg
                v = term(lk)
                v.copy_(val.detach().reshape(1))
                if lk in loss_scale:
                    if train and sc != 0:
                        leaves = lin["leaves"][key]
                        push(leaves, torch.autograd.grad(val, leaves, retain_graph=True, allow_unused=True), sc)
            else:
                raise NotImplementedError(f"scrubber '{method}' is outside this build's scope (SURVEY 8a row A2)")

    # ---- total correlation (L5): O(B^2 z) estimator over the local batch, z detached
    dsigma = dlv = None
    if "total_correlation" in loss_scale:
        if model.prior == "beta":
            raise KeyError("L")  # the reference reads data_o["L"] here (losses.py:312-316): no such entry under prior="beta"
        if z > 128:
            raise NotImplementedError("total_correlation kernel supports z_dim <= 128")
        zc, zcp = st["zc"], st["zc"].shape[1]
        mu_b, sig = st["mu"], st["sigma"]
        lv = model._buf("tc.lv", (B, z))
        ops.tc_logvar(sig if model.is_diag else None, zp, None if model.is_diag else model._L, lv, B, z)
        lse_l, lse_a, lj = model._buf("tc.lse_l", (B, z)), model._buf("tc.lse_a", (B,)), model._buf("tc.loss", (B,))
        ops.tc_fwd(zc, zcp, mu_b, zp, lv, B, z, lse_l, lse_a, lj)
        ops.reduce_rows(lj, B, 1, 1.0 / B, term("total_correlation"))
        if train and loss_scale["total_correlation"] != 0:
            w = float(loss_scale["total_correlation"]) / (B * world)
            if model.is_diag:
                dsigma = model._buf("tc.dsigma", (B, zp), zero=True)
                ops.tc_bwd(zc, zcp, mu_b, zp, lv, B, z, lse_l, lse_a, w, d_mu, zp, dsigma, zp, sig, zp)
            else:
                dlv = model._buf("tc.dlv", (B, z))
                ops.tc_bwd(zc, zcp, mu_b, zp, lv, B, z, lse_l, lse_a, w, d_mu, zp, dlv, z)

    pend = None
    if train:
        pend = dict(dy=None, kl_scale=kl_scale, d_mu=d_mu, scrub=scrub, dsigma=dsigma, dlv=dlv, lin_pgrads=lin_pgrads,
                    accumulate=getattr(model, "accumulate_grads", False))
        if fork_tail and model.defer_tail and scrub and _EARLY_SCRUB:
            # fast path: the heads' backward (it reads only what this section produced) also runs beside the tail, instead of
            # as the first thing of total.backward()
            model._assign_grad_views()
            model._scrub_backward(pend)
    if fork_tail:
        dy = join_tail()
    total = pack.finish(batch_loss)  # one launch for the running total, one snapshot of all terms
    if train:
        pend["dy"] = dy
        model._pending = pend
        batch_loss["total"] = model.make_total(total)
    else:
        batch_loss["total"] = total
    return batch_loss
