"""Training wrapper -- counterpart of the reference's model.py (Model :11-81): Adam(lr_init,
betas (0.9, 0.999)), gradient value clipping at 1, Chamfer loss weighted by
log(max_up_ratio / up_ratio, step_ratio) (which is 0 at the maximum ratio -- reproduced, :72),
running-mean error log.  Checkpoint resume (`opt.ckpt`) goes through utils-style loading of a
``{'states': state_dict, 'step': ...}`` file (pytorch_utils.py:18-51)."""
from collections import defaultdict
from math import log

import torch

from .network.model_loss import ChamferLoss
from .utils.pytorch_utils import load_network, save_network  # noqa: F401


class Model(object):
    def __init__(self, net, phase, opt):
        self.net = net
        self.phase = phase
        if phase == 'train':
            self.error_log = defaultdict(int)
            self.chamfer_criteria = ChamferLoss()
            self.old_lr = opt.lr_init
            self.lr = opt.lr_init
            self.optimizer = torch.optim.Adam(self.net.parameters(), lr=opt.lr_init, betas=(0.9, 0.999))
        if getattr(opt, "ckpt", None) not in (None, "random"):
            self.step = load_network(self.net, opt.ckpt)
        else:
            self.step = 0

    def set_input(self, input_pc, up_ratio, label_pc=None):
        """input_pc Bx3xN, up_ratio int, label_pc Bx3xN'"""
        self.input = input_pc.detach()
        self.up_ratio = up_ratio
        self.gt = label_pc.detach() if label_pc is not None else None

    def forward(self):
        if self.gt is not None:
            self.predicted, self.gt = self.net(self.input, ratio=self.up_ratio, gt=self.gt)
        else:
            self.predicted = self.net(self.input, ratio=self.up_ratio)

    def optimize(self, epoch=None):
        """run forward and backward, apply gradients (reference :53-66)"""
        self.optimizer.zero_grad()
        self.net.train()
        self.forward()
        loss = self.compute_chamfer_loss(self.predicted, self.gt)
        loss.backward()
        torch.nn.utils.clip_grad_value_(self.net.parameters(), 1)
        self.optimizer.step()
        self.step += 1

    def compute_chamfer_loss(self, pc, pc_label):
        loss_chamfer = self.chamfer_criteria(pc.transpose(1, 2).contiguous(),
                                             pc_label.transpose(1, 2).contiguous())
        weight = log(self.net.max_up_ratio / self.up_ratio, self.net.step_ratio)
        loss_chamfer = loss_chamfer * weight
        key = "cd_loss_x{}".format(self.up_ratio)
        prev_err = self.error_log[key]
        self.error_log[key] = prev_err + (loss_chamfer.item() - prev_err) / (self.step + 1)
        return loss_chamfer

    def test_model(self):
        self.net.eval()
        self.forward()
