"""Training loop drop-in (reference: src/scrubvae/train/trainer.py:26-212,322-516).

``train_test_epoch`` keeps the reference's signature and per-batch order of operations
(forward, losses, grad=None, backward, clip_grad_norm_(1e6), optimizer step, scheduler step
with fractional epoch, device-side metric accumulation, one host read per epoch).  The
optimizer returned by ``get_optimizer_and_lr_scheduler`` is a fused single-launch
Adam/AdamW over the model's flat parameter buffer (``FusedAdam``); any torch.optim
optimizer built on ``model.parameters()`` works too.
"""
from __future__ import annotations

import math
import time
from pathlib import Path

import torch

from .. import ops
from .losses import get_batch_loss

__all__ = ["CyclicalBetaAnnealing", "get_beta_schedule", "get_optimizer_and_lr_scheduler", "predict_batch",
           "train_test_epoch", "train_epoch", "train", "FusedAdam", "clip_grad_norm_", "GraphedStep"]


class CyclicalBetaAnnealing:
    """trainer.py:26-40."""

    def __init__(self, beta_max=1, len_cycle=100, R=0.5):
        self.beta_max, self.len_cycle, self.R = beta_max, len_cycle, R
        self.len_increasing = int(len_cycle * R)

    def get(self, epoch):
        remainder = (epoch - 1) % self.len_cycle
        if remainder >= self.len_increasing:
            return self.beta_max
        return self.beta_max * remainder / self.len_increasing


def get_beta_schedule(schedule, beta):
    """trainer.py:43-51 (note: the reference passes loss.prior as `schedule`, :336-339)."""
    if schedule == "cyclical":
        print("Initializing cyclical beta annealing")
        return CyclicalBetaAnnealing(beta_max=beta)
    print("No beta annealing selected")
    return None


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW semantics (defaults: betas (0.9,0.999), eps 1e-8, weight decay
    0 / 0.01; trainer.py:60-65) as ONE HIP launch per contiguous span of trainable parameters in the
    model's flat parameter, gradient and moment buffers (one span, or two around a frozen
    AdvNetScrubber -- torch's optimizers skip parameters without a gradient, so must the weight
    decay).  ``param_groups[0]["lr"]`` is honoured, so torch LR schedulers work."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled))
        self.exp_avg = torch.zeros_like(model.flat_params)
        self.exp_avg_sq = torch.zeros_like(model.flat_params)
        self.step_count = 0
        self.grad_scale = 1.0
        # device copy of the step-dependent scalars for hipGraph replays: {lr, lr/bc1, 1/sqrt(bc2), step}.  The step
        # counter lives on the DEVICE and is advanced by a captured one-thread kernel; the host only writes lr, with a
        # stream-ordered fill, when it changes -- replays queued ahead of the GPU cannot race a host buffer.
        self.hyper = torch.zeros(4, device=model.device)
        self._dev_lr = None
        self._dev_step = None

    def _spans(self):
        m = self.model
        return [(m.flat_params[a:b], m.flat_grads[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b]) for a, b in m.trainable_spans]

    def refresh_hyper(self):
        """Bookkeeping for one captured step (outside any graph): host step counter, and -- only when they changed --
        the device copies of lr and of the step counter (after load_state_dict / eager steps in between)."""
        g = self.param_groups[0]
        lr = float(torch.tensor(float(g["lr"]), dtype=torch.float32))  # the eager path passes lr as a C float
        if self._dev_lr != lr:
            self.hyper[0:1].fill_(lr)
            self._dev_lr = lr
        if self._dev_step != self.step_count:
            self.hyper[3:4].fill_(float(self.step_count))
        self.step_count += 1
        self._dev_step = self.step_count

    @torch.no_grad()
    def step_captured(self):
        """The launches that go INTO a graph: no host-side scalars."""
        g = self.param_groups[0]
        ops.adam_advance(self.hyper, g["betas"][0], g["betas"][1])
        for p, gr, m, v in self._spans():
            ops.adam_step_dev(p, gr, m, v, self.hyper, g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], g["decoupled"],
                              self.grad_scale)

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self.step_count += 1
        self.model._assign_grad_views()
        for p, gr, m, v in self._spans():
            ops.adam_step(p, gr, m, v, float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count,
                          g["decoupled"], self.grad_scale)

    def zero_grad(self, set_to_none=True):
        pass  # gradients are overwritten by every backward

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count,
                # position in the data-parallel adversarial-shuffle stream (ResVAE.global_permutation): a resumed run continues it
                "shuffle": [int(self.model.shuffle_seed), int(self.model._shuffle_draws)],
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        if "shuffle" in sd:
            self.model.shuffle_seed, self.model._shuffle_draws = int(sd["shuffle"][0]), int(sd["shuffle"][1])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


def clip_grad_norm_(model, max_norm=1e6):
    """torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) on the flat gradient
    buffer (trainer.py:164): gradients *= min(1, max_norm / (norm + 1e-6)), decided on the device
    (`svae_clip_grads` reads the squared norm and returns without touching the gradients when the
    clip does not bite, which is always with the reference's max_norm=1e6 unless a step diverges).
    Returns the total norm as a 0-dim device tensor -- a view of a per-model buffer that the next call overwrites (clone it to
    keep it across steps).  Frozen parameters have no gradient (their span of the flat buffer stays zero)."""
    n = model.flat_grads.numel()
    part = model._buf("gn.part", (ops.sumsq_blocks(n),))
    out = model._buf("gn.out", (2,))  # [sum of squares, norm]
    ops.sumsq_partial(model.flat_grads, part)
    ops.reduce_rows(part, part.numel(), 1, 1.0, out)
    ops.clip_grads(model.flat_grads, out, max_norm, out[1:])
    return out[1]


def get_optimizer_and_lr_scheduler(model, train_config, load_path=None, start_epoch=None):
    """trainer.py:54-89."""
    name = train_config["optimizer"]
    if name == "adam":
        print("Initializing Adam optimizer ...")
        optimizer = FusedAdam(model, lr=train_config["lr"])
    elif name == "adamw":
        print("Initializing AdamW optimizer ...")
        optimizer = FusedAdam(model, lr=train_config["lr"], weight_decay=0.01, decoupled=True)
    elif name == "sgd":
        print("Initializing SGD optimizer ...")
        optimizer = torch.optim.SGD(model.parameters(), lr=train_config["lr"], momentum=0.2, nesterov=True)
    else:
        raise ValueError("No valid optimizer selected")
    scheduler = None
    if train_config.get("lr_schedule") == "cawr":
        print("Initializing cosine annealing w/warm restarts learning rate scheduler")
        scheduler = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(optimizer, T_0=50)
    else:
        print("No learning rate scheduler selected")
    if load_path is not None:
        ck = Path("{}/checkpoints/epoch_{}.pth".format(load_path, start_epoch))
        if ck.exists():
            checkpoint = torch.load(ck, map_location=model.device, weights_only=True)
            optimizer.load_state_dict(checkpoint["optimizer"])
            if scheduler is not None and "lr_scheduler" in checkpoint:
                scheduler.load_state_dict(checkpoint["lr_scheduler"])
    return optimizer, scheduler


class DevicePrefetcher:
    """Iterates `loader` one batch ahead: batch i+1 is copied to the GPU on a copy stream (from pinned memory) while batch i
    is being computed, so the PCIe transfer of the pose tensors (72 MB per 1024 windows) leaves the step's critical path.
    The reference loop copies in line (`trainer.py:128`); the tensors yielded here are the same `{k: v.to(device)}` dict.
    CPU devices pass through unchanged."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        if self.device.type != "cuda":
            for batch in self.loader:
                yield {k: v.to(self.device) for k, v in batch.items()}
            return
        copy = torch.cuda.Stream(device=self.device)
        pending = None
        for batch in self.loader:
            with torch.cuda.stream(copy):
                dev = {k: (v if (v.is_cuda or v.is_pinned()) else v.pin_memory()).to(self.device, non_blocking=True)
                       for k, v in batch.items()}
                ev = torch.cuda.Event()
                ev.record(copy)
            if pending is not None:
                yield self._hand_over(*pending)
            pending = (dev, ev)
        if pending is not None:
            yield self._hand_over(*pending)

    @staticmethod
    def _hand_over(dev, ev):
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for t in dev.values():  # allocated on the copy stream, consumed on the compute stream
            t.record_stream(cur)
        return dev


_COMPUTE_STREAMS = {}


class on_compute_stream:
    """Runs the enclosed steps on a high-priority HIP stream (one per device, created once): the step's main chain then
    wins the dispatch over its own weight-gradient / skip-branch streams, which only fill the CUs it leaves idle (-1 % step
    time at B=1024).  Ordered after everything already queued on the caller's stream at entry, and the caller's stream
    waits for it at exit, so code around the block needs no extra synchronisation.  No-op for CPU devices."""

    def __init__(self, device):
        d = torch.device(device)
        self.stream = None
        if d.type == "cuda" and torch.cuda.is_available():
            idx = d.index if d.index is not None else torch.cuda.current_device()
            if idx not in _COMPUTE_STREAMS:
                _COMPUTE_STREAMS[idx] = torch.cuda.Stream(device=idx, priority=-1)
            self.stream = _COMPUTE_STREAMS[idx]

    def __enter__(self):
        if self.stream is not None:
            self.prev = torch.cuda.current_stream(self.stream.device)
            self.stream.wait_stream(self.prev)
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.stream is not None:
            self.ctx.__exit__(*exc)
            self.prev.wait_stream(self.stream)
        return False


def predict_batch(model, data, disentangle_keys=None):
    """trainer.py:92-99 (plus the fused tail's inputs and optional injected noise)."""
    keep = ["x6d", "root", "var", "offsets", "target_pose", "eps"]
    data_i = {k: v for k, v in data.items() if (k in (disentangle_keys or [])) or (k in keep)}
    return model(data_i)


def train_test_epoch(config, model, loader, device, epoch, optimizer=None, scheduler=None, mode="train"):
    """trainer.py:102-212."""
    if mode == "train":
        model.train()
        grad_env = torch.enable_grad
    elif mode == "test":
        model.eval()
        grad_env = torch.no_grad
    else:
        raise ValueError("This mode is not recognized.")
    with grad_env(), on_compute_stream(device):
        model.mi_estimator = None
        epoch_metrics = {k: 0 for k in ["total"] + list(config["loss"].keys())}
        n_batches = 0
        for batch_idx, data in enumerate(DevicePrefetcher(loader, device)):
            data_o = predict_batch(model, data, model.disentangle_keys)
            # (the reference's adversarial `fit` branch compares mode to "Train" and never runs,
            #  trainer.py:133 -- the discriminator stays at its initialisation)
            batch_loss = get_batch_loss(model, data, data_o, config["loss"], config["disentangle"])
            if mode == "train":
                for param in model.parameters():
                    param.grad = None
                batch_loss["total"].backward()
                clip_grad_norm_(model, max_norm=1e6)
                optimizer.step()
                if scheduler is not None:
                    scheduler.step(epoch + batch_idx / len(loader))
                # update the exponential-moving-average scrubbers (trainer.py:170-180)
                for method in model.disentangle.keys():
                    if method in ["moving_avg_lsq", "moving_avg", "qda"]:
                        for k in model.disentangle[method].keys():
                            model.disentangle[method][k].update(data_o["mu"].detach().clone(), data[k].detach().clone())
            epoch_metrics = {k: v + batch_loss[k].detach() for k, v in epoch_metrics.items()}
            n_batches += 1
            if "mcmi" in config["loss"].keys():
                # refresh the mutual-information estimator from this batch, re-encoded with the updated weights
                # (trainer.py:184-199); under data parallelism the centres of all ranks are gathered
                updated = model.encode(data)
                model.mi_estimator = make_mi_estimator(model, config, updated["mu"].detach().clone(), data_o["var"].clone(),
                                                       updated["L"].detach().clone() if "L" in updated else None)
        for k, v in epoch_metrics.items():
            epoch_metrics[k] = (v.item() if torch.is_tensor(v) else float(v)) / max(n_batches, 1)
            print("====> Epoch: {} Average {} loss: {:.4f}".format(epoch, k, epoch_metrics[k]))
    return epoch_metrics


def make_mi_estimator(model, config, mu, var, L):
    from ..model.disentangle import MutInfoEstimator, _rank_cat
    group = getattr(model, "process_group", None)
    mu, var = _rank_cat(mu, group), _rank_cat(var, group)
    L = _rank_cat(L, group) if L is not None else None
    return MutInfoEstimator(x_s=mu, y_s=var, bandwidth=config["disentangle"]["bandwidth"],
                            var_mode=config["disentangle"]["var_mode"], model_var=L, device=mu.device)


def test_epoch(config, model, loader, device="cuda", epoch=0):
    """trainer.py:215-303: eval-mode pass over `loader` -> (epoch_metrics incl. r2_gen_restrict_<key>, mu [N,z] on
    the CPU).  With an `mcmi` loss the mutual-information estimator is first rebuilt from a strided sample of the
    dataset (trainer.py:228-252)."""
    from sklearn.metrics import r2_score

    from ..eval import generative_restrictiveness
    print("Running test epoch")
    model.eval()
    with torch.no_grad():
        z = []
        if "mcmi" in config["loss"].keys():
            sample = loader.dataset[:: int(len(loader.dataset) / config["data"]["batch_size"])]
            enc = model.encode({k: v.to(device) for k, v in sample.items() if k in ["x6d", "root"]})
            var = torch.cat([sample[k] for k in model.conditional_keys], dim=-1).to(device)
            model.mi_estimator = make_mi_estimator(model, config, enc["mu"].detach().clone(), var,
                                                   enc["L"].detach().clone() if "L" in enc else None)
        epoch_metrics = {k: 0 for k in ["total"] + list(config["loss"].keys())}
        gen_res = {k1: {k2: [] for k2 in ["pred", "target"]} for k1 in model.disentangle_keys if k1 != "ids"}
        n_batches = 0
        for batch_idx, data in enumerate(loader):
            data = {k: v.to(device) for k, v in data.items()}
            data_o = predict_batch(model, data, model.disentangle_keys)
            z += [data_o["mu"].clone().detach()]
            batch_metrics = get_batch_loss(model, data, data_o, config["loss"], config["disentangle"])
            for key in gen_res.keys():
                key_pred, key_target = generative_restrictiveness(model, data_o["mu"], data, key, loader.dataset.kinematic_tree)
                gen_res[key]["pred"] += [key_pred.detach().cpu()]
                gen_res[key]["target"] += [key_target.detach().cpu()]
            epoch_metrics = {k: v + batch_metrics[k].detach() for k, v in epoch_metrics.items()}
            n_batches += 1
    for k, v in epoch_metrics.items():
        epoch_metrics[k] = (v.item() if torch.is_tensor(v) else float(v)) / max(n_batches, 1)
        print("====> Epoch: {} Average {} loss: {:.4f}".format(epoch, k, epoch_metrics[k]))
    for key in gen_res.keys():
        epoch_metrics["r2_gen_restrict_{}".format(key)] = r2_score(torch.cat(gen_res[key]["target"], dim=0),
                                                                    torch.cat(gen_res[key]["pred"], dim=0))
    return epoch_metrics, torch.cat(z, dim=0).cpu()


def train_epoch(config, model, loader, optimizer, scheduler, device="cuda", epoch=0):
    return train_test_epoch(config, model, loader, device, epoch, optimizer, scheduler, mode="train")


def train(config, model, loader_dict, run=None):
    """trainer.py:322-516: optimizer / scheduler set-up, beta schedule, per-epoch scrubber re-initialisation, tuned
    forgetting factors in the metrics, weights every 5 epochs, optimizer state every 20, `test_epoch` on
    loader_dict["val"] from epoch 50 on.  The sklearn decodability / clustering metrics (trainer.py:414-507) are
    outside the path.  Metrics go to `run.log(metrics, epoch)` when a W&B-like `run` is given and are
    always appended as JSON lines to <out_path>/metrics.jsonl (the W&B-free log of SURVEY 8f N3)."""
    import json
    optimizer, scheduler = get_optimizer_and_lr_scheduler(
        model, config["train"], config["model"].get("load_model"), config["model"].get("start_epoch"))
    beta_scheduler = (get_beta_schedule(config["loss"]["prior"], config["train"].get("beta_anneal"))
                      if "prior" in config["loss"].keys() else None)
    start_epoch = config["model"].get("start_epoch") or 0
    out_path = config.get("out_path")
    for epoch in range(start_epoch + 1, config["train"]["num_epochs"] + 1):
        if beta_scheduler is not None:
            config["loss"]["prior"] = beta_scheduler.get(epoch)
            print("Beta schedule: {:.3f}".format(config["loss"]["prior"]))
        t0 = time.time()
        train_metrics = train_epoch(config, model, loader_dict["train"], optimizer, scheduler, model.device, epoch)
        metrics = {"{}_train".format(k): v for k, v in train_metrics.items()}
        if "grad_reversal" in model.disentangle.keys():
            for k in model.disentangle["grad_reversal"].keys():
                model.disentangle["grad_reversal"][k].reset_parameters()
        if "moving_avg_lsq" in model.disentangle.keys():  # automatically tuned smoothing factors (trainer.py:373-377)
            for k in model.disentangle["moving_avg_lsq"].keys():
                metrics["lambda_mals_{}".format(k)] = float(model.disentangle["moving_avg_lsq"][k].lam1.detach().cpu())
        metrics["time"] = time.time() - t0
        if epoch % 5 == 0:
            if out_path:
                print("Saving model to folder: {}".format(out_path))
                Path(out_path + "weights/").mkdir(parents=True, exist_ok=True)
                torch.save({k: v.cpu() for k, v in model.state_dict().items()}, "{}/weights/epoch_{}.pth".format(out_path, epoch))
                if epoch % 20 == 0:
                    Path(out_path + "checkpoints/").mkdir(parents=True, exist_ok=True)
                    ck = {"optimizer": optimizer.state_dict()}
                    if scheduler is not None:
                        ck["lr_scheduler"] = scheduler.state_dict()
                    torch.save(ck, "{}/checkpoints/epoch_{}.pth".format(out_path, epoch))
            if epoch >= 50 and loader_dict.get("val") is not None:
                test_metrics, _ = test_epoch(config, model, loader_dict["val"], model.device, epoch)
                metrics.update({"{}_test".format(k): float(v) for k, v in test_metrics.items()})
        if run is not None:
            run.log(metrics, epoch)
        if out_path:
            Path(out_path).mkdir(parents=True, exist_ok=True)
            with open(str(Path(out_path) / "metrics.jsonl"), "a") as f:
                f.write(json.dumps({"epoch": epoch, **{k: float(v) for k, v in metrics.items()}}) + "\n")
    return model


class GraphedStep:
    """One whole optimizer step (forward, losses, reverse schedule, grad-norm, fused Adam) captured
    into a hipGraph and replayed: removes the ~250 per-step launch calls from the host, which is
    what bounds small batches (B=32: 4.0 ms eager).  Single-rank only; the batch shape, loss
    scales and disentangle config are frozen at capture (re-create after changing them).

        step = GraphedStep(model, optimizer, config["loss"], config["disentangle"], example_batch)
        losses = step(batch)            # dict of device scalars (static tensors: clone to keep)
    """

    def __init__(self, model, optimizer, loss_scale, disentangle_config, example, warmup=3, capture=True):
        if model.world_size > 1:
            raise NotImplementedError("GraphedStep is single-rank (collectives are not captured)")
        if not isinstance(optimizer, FusedAdam):
            raise TypeError("GraphedStep needs the FusedAdam optimizer")
        self.model, self.opt = model, optimizer
        self.loss_scale, self.dis = dict(loss_scale), disentangle_config
        self.static = {k: v.to(model.device).clone() for k, v in example.items() if torch.is_tensor(v)}
        model.train()
        model.defer_tail = True
        self.grad_norm = None
        for _ in range(warmup):  # eager: allocates every workspace, runs the tile autotuner
            optimizer.refresh_hyper()
            self._body()
        torch.cuda.synchronize()
        self.graph = None
        if not capture:  # same schedule launched eagerly (debugging / parity reference)
            return
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):  # capture does not execute: neither the host nor the device counter moves
            self.losses = self._body()

    def _body(self):
        m = self.model
        data_o = m(self.static)
        bl = get_batch_loss(m, self.static, data_o, self.loss_scale, self.dis)
        m.backward_from_seeds()
        self.grad_norm = clip_grad_norm_(m, 1e6)
        self.opt.step_captured()
        return {k: v.detach() for k, v in bl.items()}

    def __call__(self, batch=None):
        if batch is not None:
            for k, v in batch.items():
                if k in self.static and v.data_ptr() != self.static[k].data_ptr():
                    self.static[k].copy_(v, non_blocking=True)
        self.opt.refresh_hyper()
        if self.graph is None:
            self.losses = self._body()
        else:
            self.graph.replay()
        return self.losses
