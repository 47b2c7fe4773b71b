"""Training / evaluation driver object with the interface main.py and the reference's callers use
(reference model.py:11-81: `Model(net, phase, opt)`, `set_input`, `forward`, `optimize`,
`compute_chamfer_loss`, `test_model`, attributes `net`, `optimizer`, `chamfer_criteria`,
`error_log`, `step`, `predicted`, `gt`).

The semantics of one optimisation step are the reference's (:53-77): Adam(lr_init, betas
(0.9, 0.999)), Chamfer loss scaled by log(max_up_ratio / up_ratio, step_ratio) -- which is 0 at the
maximum ratio, reproduced --, gradient values clipped to [-1, 1], running mean of the loss per ratio.

What is built differently:

  * the step never synchronises the host.  The reference reads `loss.item()` in every step to
    update its running mean (:74-76), which stalls the launch queue of a step that is already
    launch-bound; here the running means live in a small device tensor (`LossLog`) that is only
    copied to the host when somebody reads `error_log`;
  * with `opt.graph_steps` (or TPU3_GRAPH_STEPS=1) a step of a given (ratio, shapes) is captured
    once into a hipGraph and replayed: the 1000-2000 eager launches of a step become one graph
    launch.  Input and label are then copied into static buffers, the random patch seeds come from
    the graph-safe device generator, Adam runs in its capturable form.
"""
import os
from math import log

import torch

from .network.model_loss import ChamferLoss
from .utils.pytorch_utils import load_network, save_network  # noqa: F401  (re-exported like the reference)


class LossLog(object):
    """Mapping `name -> running mean` whose values stay on the device until they are read.

    update(name, value, count): mean += (value - mean) / count with a 0-d device tensor `value`
    (the reference's rule, :74-76, where count = step + 1)."""

    ROWS = 64                    # fixed: captured graphs hold views into the table, so it must never move

    def __init__(self):
        self._slot = {}          # name -> row in the device table
        self._table = None       # (ROWS,) float32 on the device of the first update

    def _row(self, name, like):
        if name not in self._slot:
            if len(self._slot) >= self.ROWS:
                raise RuntimeError("LossLog: more than %d distinct keys" % self.ROWS)
            if self._table is None:
                self._table = torch.zeros((self.ROWS,), dtype=torch.float32, device=like.device)
            self._slot[name] = len(self._slot)
        return self._slot[name]

    def update(self, name, value, count):
        r = self._row(name, value)
        cell = self._table[r:r + 1]
        # in place on the device: no .item(), nothing for the host to wait for
        cell.add_((value.detach().reshape(1).to(torch.float32) - cell) / float(count))

    def cell(self, name, like):
        """The 1-element device view a captured graph updates in place."""
        r = self._row(name, like)
        return self._table[r:r + 1]

    # ---- read side (synchronises) --------------------------------------------------------------
    def __contains__(self, name):
        return name in self._slot

    def __getitem__(self, name):
        if name not in self._slot:
            return 0                                   # defaultdict(int) behaviour of the reference
        return float(self._table[self._slot[name]].item())

    def keys(self):
        return list(self._slot)

    def items(self):
        if not self._slot:
            return []
        host = self._table.detach().cpu()
        return [(k, float(host[r])) for k, r in self._slot.items()]

    def __len__(self):
        return len(self._slot)

    def __iter__(self):
        return iter(self.keys())


class _CapturedStep(object):
    """One training step of fixed shapes as a hipGraph."""

    def __init__(self, model, input_pc, up_ratio, label_pc):
        self.inp = input_pc.clone()
        self.lab = label_pc.clone()
        self.count = torch.ones((), dtype=torch.float32, device=input_pc.device)
        self.key = "cd_loss_x{}".format(up_ratio)
        cell = model.error_log.cell(self.key, input_pc)
        net, optim = model.net, model.optimizer
        weight = model.loss_weight(up_ratio)

        def body():
            optim.zero_grad(set_to_none=False)
            pred, gt = net(self.inp, ratio=up_ratio, gt=self.lab)
            loss = model.chamfer_criteria(pred.transpose(1, 2).contiguous(),
                                          gt.transpose(1, 2).contiguous()) * weight
            loss.backward()
            torch.nn.utils.clip_grad_value_(net.parameters(), 1)
            optim.step()
            cell.add_((loss.detach().reshape(1) - cell) / self.count)
            return pred, gt

        # warm-up on a side stream (allocator pools, lazily created optimizer state, rocBLAS handles)
        side = torch.cuda.Stream(device=input_pc.device)
        side.wait_stream(torch.cuda.current_stream())
        state = {n: p.detach().clone() for n, p in net.named_parameters()}
        moments = {p: {k: v.clone() for k, v in st.items() if torch.is_tensor(v)}
                   for p, st in optim.state.items()}
        saved_cell = cell.clone()
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream().wait_stream(side)
        # the warm-up step must not count: restore parameters, Adam's moments / step counters (zero
        # where the warm-up created them) and the log
        with torch.no_grad():
            for n, p in net.named_parameters():
                p.copy_(state[n])
            cell.copy_(saved_cell)
            for p, st in optim.state.items():
                before = moments.get(p, {})
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if k in before:
                            v.copy_(before[k])
                        else:
                            v.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.pred, self.gt = body()

    def run(self, input_pc, label_pc, count):
        self.inp.copy_(input_pc)
        self.lab.copy_(label_pc)
        self.count.fill_(float(count))
        self.graph.replay()
        return self.pred, self.gt


class Model(object):
    def __init__(self, net, phase, opt):
        self.net = net
        self.phase = phase
        self.input = self.gt = self.predicted = None
        self.up_ratio = None
        if phase == 'train':
            self.error_log = LossLog()
            self.chamfer_criteria = ChamferLoss()
            self.lr = self.old_lr = opt.lr_init
            self.graph_steps = bool(getattr(opt, "graph_steps", False)
                                    or os.environ.get("TPU3_GRAPH_STEPS", "").strip().lower() not in ("", "0", "false", "no"))
            self.optimizer = torch.optim.Adam(self.net.parameters(), lr=opt.lr_init, betas=(0.9, 0.999),
                                              capturable=self.graph_steps)
            self._captured = {}
        ckpt = getattr(opt, "ckpt", None)
        # "random" = main.py's spelling of "no checkpoint, random-init weights"
        self.step = load_network(self.net, ckpt) if ckpt not in (None, "random") else 0

    # ---- reference interface ---------------------------------------------------------------------
    def set_input(self, input_pc, up_ratio, label_pc=None):
        """input_pc (B,3,N), up_ratio int, label_pc (B,3,N') or None."""
        self.input = input_pc.detach()
        self.up_ratio = up_ratio
        self.gt = None if label_pc is None else label_pc.detach()

    def forward(self):
        if self.gt is None:
            self.predicted = self.net(self.input, ratio=self.up_ratio)
        else:
            self.predicted, self.gt = self.net(self.input, ratio=self.up_ratio, gt=self.gt)

    def loss_weight(self, up_ratio):
        """log_{step_ratio}(max_up_ratio / up_ratio): the number of levels NOT trained by this ratio."""
        return log(self.net.max_up_ratio / up_ratio, self.net.step_ratio)

    def compute_chamfer_loss(self, pc, pc_label):
        """pc, pc_label (B,3,n) -> weighted Chamfer loss (0-d tensor); logs its running mean."""
        loss = self.chamfer_criteria(pc.transpose(1, 2).contiguous(), pc_label.transpose(1, 2).contiguous())
        loss = loss * self.loss_weight(self.up_ratio)
        self.error_log.update("cd_loss_x{}".format(self.up_ratio), loss, self.step + 1)
        return loss

    def optimize(self, epoch=None):
        """One optimisation step on the tensors given to set_input."""
        self.net.train()
        if self.graph_steps and self.input.is_cuda:
            # threshold and forward weight are scalar launch arguments, i.e. frozen into a captured graph:
            # main.py's curriculum toggles the threshold mid-training (set_threshold / unset_threshold), so
            # they are part of the key -- a toggle captures (once) a second graph instead of being ignored
            crit = self.chamfer_criteria
            key = (self.up_ratio, tuple(self.input.shape), tuple(self.gt.shape),
                   crit._threshold, float(crit.forward_weight))
            step = self._captured.get(key)
            if step is None:
                step = self._captured[key] = _CapturedStep(self, self.input, self.up_ratio, self.gt)
            self.predicted, self.gt = step.run(self.input, self.gt, self.step + 1)
        else:
            self.optimizer.zero_grad()
            self.forward()
            self.compute_chamfer_loss(self.predicted, self.gt).backward()
            torch.nn.utils.clip_grad_value_(self.net.parameters(), 1)
            self.optimizer.step()
        self.step += 1

    def test_model(self):
        self.net.eval()
        self.forward()
