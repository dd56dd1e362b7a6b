"""Training / evaluation loop with the reference Trainer's constructor, checkpoint format and printed
metrics (reference future_od/trainer.py), driving the HIP-backed model.  wandb and the PNG
visualisation are optional imports: absent packages disable those features instead of failing."""
import os

import torch
import torch.distributed as distrib

from future_od.graph import CaptureError
from future_od.models.set_criterion import join_matchers
from future_od.utils.distributed import EXIT, gather_distrib_od_map_stuffs, reduce_distrib_loss
from future_od.utils.od_map import aggregate_mean_average_precision
from future_od.utils.prefetch import DevicePrefetcher
from future_od.utils.recursive_functions import recursive_detach_cpu, recursive_to
from future_od.utils.stats import AverageMeter
from future_od.utils.wandb import WandBConfig

try:
    import wandb
except ImportError:            # logging backend is not part of the hot path
    wandb = None


def _unwrap(model):
    return model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model


class Trainer:
    def __init__(self, model, optimizer, lr_sched, train_loader, val_loaders, checkpoint_path, visualization_path,
                 save_name, device, print_interval, visualization_epochs, visualization_iterations, category_dict,
                 checkpoint_epochs=None, gradient_clip_value=None, distributed=False, is_master=False,
                 wandb_config=WandBConfig(), max_norm=0.0):
        self._model = model
        self._model_no_ddp = _unwrap(model)
        self._optimizer, self._lr_sched = optimizer, lr_sched
        self._train_loader = train_loader
        self._val_loaders = ({f"val{i}": l for i, l in enumerate(val_loaders)} if isinstance(val_loaders, list)
                             else val_loaders)
        assert len(train_loader) and min(len(l) for l in self._val_loaders.values()) > 0, "All loaders must be non-empty"
        assert gradient_clip_value is None or isinstance(gradient_clip_value, (int, float))
        self._gradient_clip_value = gradient_clip_value
        self._save_checkpoints = bool(checkpoint_epochs)
        self._checkpoint_path, self._visualization_path = checkpoint_path, visualization_path
        self._save_name, self._device = save_name, device
        self._print_interval = print_interval
        self._visualization_epochs, self._visualization_iterations = visualization_epochs, visualization_iterations
        self._category_dict = category_dict
        self._distributed, self._is_master = distributed, is_master
        self._stats = {f"{mode} {key} loss": AverageMeter()
                       for mode in ["train"] + list(self._val_loaders.keys())
                       for key in self._model_no_ddp.get_stat_idfs()}
        self._epoch = 0
        self._training_iterations = 0
        self._wandb_config = wandb_config
        self._max_norm = max_norm
        # an optimizer that clips inside its fused update (future_od.optim.FusedAdamW) makes the separate
        # clip_grad_norm_ pass unnecessary
        self._optimizer_clips = getattr(optimizer, "max_norm", 0.0) > 0

    # ------------------------------------------------------------------ public API
    def train(self, max_epochs):
        self._setup_wandb(["training"])
        print(f"Training epochs {self._epoch + 1} to {max_epochs}. Moving model to {self._device}.")
        if not self._distributed:
            self._model.to(self._device)
        for epoch in range(self._epoch + 1, max_epochs + 1):
            self._epoch = epoch
            if self._distributed and hasattr(self._train_loader, "sampler") and hasattr(self._train_loader.sampler, "set_epoch"):
                self._train_loader.sampler.set_epoch(epoch)
            print(f"Starting epoch {epoch} with lr={self._lr_sched.get_last_lr()}")
            self._model.train(True)
            if hasattr(self._model_no_ddp._model, "drop_mode"):
                self._model_no_ddp._model.drop_mode = "train"
            self._run_epoch("train", self._train_loader)
            self._run_eval()
            for meter in self._stats.values():
                meter.new_epoch()
            self._lr_sched.step()
            if EXIT.is_set():
                return
            if self._save_checkpoints:
                print("Saving Checkpoint")
                self.save_checkpoint(is_final=(epoch == max_epochs))
        print("Finished training!")

    def eval(self):
        self._setup_wandb(["eval"])
        print(f"Running eval. Moving model to {self._device}.")
        if not self._distributed:
            self._model.to(self._device)
        self._run_eval()

    # ------------------------------------------------------------------ internals
    def _setup_wandb(self, tags):
        c = self._wandb_config
        if self._is_master and c.enabled and wandb is not None:
            wandb.init(project=c.project, entity=c.entity, config=c.hyperparams, name=c.name, notes=c.notes,
                       resume="must" if c.resume_id else None, id=c.resume_id, tags=tags)

    def _run_eval(self):
        self._model.train(False)
        with torch.no_grad():
            for name, loader in self._val_loaders.items():
                if hasattr(self._model_no_ddp._model, "drop_mode"):
                    self._model_no_ddp._model.drop_mode = name
                self._run_epoch(name, loader)

    def _graphed_step(self):
        """The training step as one captured hipGraph (future_od/graph.py) when the setup allows it: one device, the
        fused optimizer (its step count and clipping live on the device), the device-side matcher.  The reference's loop
        (trainer.py:171-189) costs ~1400 launches behind ~16 us of Python each -- every configuration but the largest is
        bound by the launching thread; a replay costs one call.  Dropout stays active (train mode): the kernels draw new
        masks per replay from a device-side counter.  FOD_GRAPH_TRAIN=0 keeps the eager loop."""
        g = getattr(self, "_graphed", None)
        if g is False:
            return None
        if g is None:
            import os
            from future_od.models.set_criterion import device_matching_enabled
            dev = torch.device(self._device)
            # data parallel: with the FodDataParallel wrapper built without torch's reducer (its default) the step is two
            # graphs around an eagerly launched gradient all-reduce (graph.py); any other wrapper keeps the eager loop
            dp_ok = not self._distributed or getattr(self._model, "light", False)
            ok = (os.environ.get("FOD_GRAPH_TRAIN", "1") != "0" and dp_ok and dev.type == "cuda"
                  and hasattr(self._optimizer, "enable_device_step")
                  and (self._optimizer_clips or not self._max_norm) and device_matching_enabled(dev))
            if not ok:
                self._graphed = False
                return None
            from future_od.graph import GraphedStep
            g = self._graphed = GraphedStep(self._model, self._optimizer, warmup=2, rollback_warmup=True)
        return g

    def _graphed_forward(self):
        """The evaluation pass as one captured hipGraph (future_od/graph.py: GraphedForward), same conditions as the
        captured training step; FOD_GRAPH_EVAL=0 keeps the eager loop."""
        g = getattr(self, "_graphed_eval", None)
        if g is False:
            return None
        if g is None:
            import os
            from future_od.models.set_criterion import device_matching_enabled
            dev = torch.device(self._device)
            ok = (os.environ.get("FOD_GRAPH_EVAL", "1") != "0" and not self._distributed and dev.type == "cuda"
                  and device_matching_enabled(dev) and not torch.is_grad_enabled())
            if not ok:
                self._graphed_eval = False
                return None
            from future_od.graph import GraphedForward
            g = self._graphed_eval = GraphedForward(self._model)
        return g

    @staticmethod
    def _fetch_async(tensors):
        """Device tensors -> pinned host copies queued on the stream, plus the event that says when they are there.
        The bookkeeping below consumes an iteration's numbers one iteration LATER, after the next step has been
        queued: reading them back at once (as the reference does) stalls the host on the whole backward, and the
        GPU then idles while the next forward is being queued (~10 % of the step)."""
        if not tensors or not tensors[0].is_cuda:
            return [t.detach().cpu() for t in tensors], None
        host = []
        for t in tensors:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t.detach(), non_blocking=True)
            host.append(h)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    def _run_epoch(self, mode, data_loader):
        od_lists = [[], [], [], []]
        world = distrib.get_world_size() if self._distributed else 1
        batch_size = getattr(data_loader, "batch_size", 1) or 1
        stat_keys = []
        pending = None                     # (keys, host stats, host od tensors or None, event) of the previous iteration

        def consume(item):
            keys, (hstats, hod, ev) = item
            if ev is not None:
                ev.synchronize()
            for k, v in zip(keys, hstats[0].tolist()):
                self._stats[f"{mode} {k} loss"].update(v, 1)
            if hod is not None:
                n = len(hod) // 4
                for j in range(4):
                    od_lists[j].extend(hod[j * n:(j + 1) * n])

        graphed = self._graphed_step() if mode == "train" else None
        graphed_eval = self._graphed_forward() if mode != "train" else None
        for i, data in enumerate(DevicePrefetcher(data_loader, self._device)):      # next batch staged on a side stream
            if EXIT.is_set():
                return
            if graphed is not None:
                try:
                    out, loss, stats, od = graphed(data)      # zero_grad + forward + backward + clip + AdamW: one launch
                except CaptureError as e:                     # not capturable here (parameters rolled back, all ranks
                    print(f"[fod] captured training step unavailable ({e}); launching eagerly")   # agreed): eager from now on
                    self._graphed, graphed = False, None
            if graphed is not None:
                self._training_iterations += 1
                if self._epoch == 1 and i == 0:
                    print("\nFirst iteration. Checking whether all layers gradients.")
                    for name, p in self._model.named_parameters():
                        if p.requires_grad and p.grad is None:
                            print(name, "got no gradient")
            else:
                if mode == "train":
                    self._optimizer.zero_grad()
                done = False
                if graphed_eval is not None:
                    try:
                        out, loss, stats, od = graphed_eval(data)     # the evaluation pass: one launch
                        done = True
                    except CaptureError as e:
                        print(f"[fod] captured evaluation pass unavailable ({e}); launching eagerly")
                        self._graphed_eval, graphed_eval = False, None
                if not done:
                    out, _state, loss, stats, od = self._model(data=data, visualize=False, epoch=self._epoch,
                                                               distributed=self._distributed)
            if mode == "train" and graphed is None:
                loss.backward()
                if self._epoch == 1 and i == 0:
                    print("\nFirst iteration. Checking whether all layers gradients.")
                    for name, p in self._model.named_parameters():
                        if p.requires_grad and p.grad is None:
                            print(name, "got no gradient")
                if self._max_norm > 0 and not self._optimizer_clips:
                    torch.nn.utils.clip_grad_norm_(self._model.parameters(), self._max_norm)
                self._optimizer.step()
                self._training_iterations += 1
            if self._distributed:
                stats = reduce_distrib_loss(stats, average=True)
            keys = list(stats.keys())
            fetch = [torch.stack([stats[k].detach().float().reshape(()) for k in keys])]
            n_od = 0
            if i * batch_size * world < 10000:          # cap the AP bookkeeping at ~10k images
                od = gather_distrib_od_map_stuffs(od) if self._distributed else [[t] for t in od]
                n_od = len(od[0])
                for j in range(4):
                    fetch.extend(od[j])
            host, ev = self._fetch_async(fetch)
            if pending is not None:
                consume(pending)
            pending = (keys, (host[:1], host[1:] if n_od else None, ev))
            stat_keys = keys
            if (i + 1) % self._print_interval == 0:
                consume(pending)
                pending = None
                msg = "  ".join(f"{self._stats[f'{mode} {k} loss'].avg:.5f} ({k})" for k in keys)
                print(f"[{mode}: {self._epoch}, {i + 1:4d}/{len(data_loader)}] Loss: {msg}.")
        if pending is not None:
            consume(pending)
        join_matchers()                    # a matcher failure of the last iteration is raised here, not lost
        msg = "  ".join(f"{self._stats[f'{mode} {k} loss'].avg:.5f} ({k})" for k in stat_keys)
        print(f"[{mode}: {self._epoch}] Loss: {msg}")
        if not od_lists[0]:
            return
        ap = aggregate_mean_average_precision(torch.cat(od_lists[0], dim=2), torch.cat(od_lists[1], dim=2),
                                              torch.cat(od_lists[2], dim=2), torch.stack(od_lists[3], dim=2),
                                              self._device)
        fmt = lambda xs: " ".join(f"{float(e):.3f}" for e in xs)
        print("AP50 for epoch is:", fmt(ap["all"][0, :, 0]))
        print("MAP for epoch is:", fmt(ap["threshavg"][:, 0]))
        print("MAP for small objects is:", fmt(ap["threshavg"][:, 1]))
        print("MAP for medium objects is:", fmt(ap["threshavg"][:, 2]))
        print("MAP for large objects is:", fmt(ap["threshavg"][:, 3]))
        self.last_ap = ap

    # ------------------------------------------------------------------ checkpoints (reference :282-328)
    def save_checkpoint(self, is_final: bool = False):
        if not self._is_master:
            return
        state = {"epoch": self._epoch, "net_type": type(self._model_no_ddp).__name__,
                 "net": self._model_no_ddp.state_dict(), "optimizer": self._optimizer.state_dict(),
                 "lr_schedule": self._lr_sched.state_dict(), "stats": self._stats, "device": self._device}
        torch.save(state, f"{self._checkpoint_path}/{self._save_name}.pth.tar")
        if is_final:
            torch.save({"net": state["net"]}, f"{self._checkpoint_path}/{self._save_name}_final.pth.tar")

    def load_checkpoint(self, checkpoint: str = None, load_only_net=False):
        print(f"Loading checkpoint: {checkpoint}")
        if checkpoint is None:
            path = f"{self._checkpoint_path}/{self._save_name}.pth.tar"
        elif isinstance(checkpoint, str):
            path = os.path.expanduser(checkpoint)
        else:
            raise TypeError("Checkpoint must be string or None")
        if not os.path.isfile(path):
            print(f"WARNING: Attempted to load checkpoint {path}, but it does not exist. Continuing without loading.")
            return
        ck = torch.load(path, map_location="cpu", weights_only=False)
        assert ck.get("net_type", type(self._model_no_ddp).__name__) == type(self._model_no_ddp).__name__, \
            "Network is not of correct type"
        self._model_no_ddp.load_state_dict(ck["net"])
        if not load_only_net and "optimizer" in ck:
            self._epoch = ck["epoch"]
            self._optimizer.load_state_dict(ck["optimizer"])
            self._stats = ck["stats"]
            self._lr_sched.load_state_dict(ck["lr_schedule"])
        print(f"Loaded: {path}")
