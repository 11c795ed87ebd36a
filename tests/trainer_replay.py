"""Test harness: the call sequence NetworkTrainer drives a network through, restated (the reference file cannot travel to the
GPU box).  Follows NetworkTrainer/network_trainer.py: set_GPU_device 92-105 (.to(device)), set_optimizer 107-125
(optim.Adam(..., weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-8, amsgrad=True)), set_lr_scheduler 127-140 (cosine), train
215-258 (network.train(); per batch: zero_grad -> forward -> loss -> backward -> step -> moving loss -> update_lr), val 260-272
(network.eval(); online evaluation), run 274-330 (per-epoch bookkeeping, save_trainer for each status) and save_trainer 340-357
(checkpoint keys).  Pinned by tests/golden/g6_trainer.npz, which the REAL trainer produced (make_golden.py g6)."""
import os
import time

import torch
from torch import optim


class ReplayLog:
    def __init__(self):
        self.iter = -1
        self.epoch = -1
        self.moving_train_loss = None
        self.average_train_loss = 99999999.
        self.best_average_train_loss = 99999999.
        self.average_val_index = -99999999.
        self.best_average_val_index = -99999999.
        self.list_average_train_loss_associate_iter = []
        self.list_average_val_index_associate_iter = []
        self.list_lr_associate_iter = []
        self.save_status = []


class TrainerReplay:
    def __init__(self, network, device, loss_function, val_function, train_loader, output_dir, optimizer_cls=optim.Adam, lr=1e-3,
                 sched_args=None, eps_train_loss=0.01, max_epoch=1, update_on_iter=True, max_iter=99999999):
        self.log = ReplayLog()
        self.network = network
        self.device = device
        self.network.to(device)                                                          # set_GPU_device
        self.optimizer = optimizer_cls(self.network.parameters(), lr=lr, weight_decay=3e-5, betas=(0.9, 0.999), eps=1e-08,
                                       amsgrad=True)                                     # set_optimizer (single-group branch)
        a = sched_args or {"T_max": 10, "eta_min": 1e-7, "last_epoch": -1}
        self.lr_scheduler = optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=a["T_max"], eta_min=a["eta_min"],
                                                                 last_epoch=a["last_epoch"])
        self.loss_function, self.val_function = loss_function, val_function
        self.train_loader, self.output_dir = train_loader, output_dir
        self.eps_train_loss, self.max_epoch, self.update_on_iter = eps_train_loss, max_epoch, update_on_iter
        self.losses = []
        self.max_iter = max_iter

    def _log(self, txt, mode):
        with open(os.path.join(self.output_dir, "log.txt"), mode) as f:
            f.write(txt)

    def train(self):
        self.network.train()
        total, count = 0., 0
        for batch in self.train_loader:
            if self.log.iter >= self.max_iter - 1:
                break
            self.log.iter += 1
            input_ = batch["Input"].float().to(self.device)
            target = batch["GT"].to(self.device)
            self.optimizer.zero_grad()                                                   # forward(phase='train')
            output = self.network(input_.to(self.device))
            for i in range(len(target)):                                                 # backward(): per-sample .to(device)
                target[i] = target[i].to(self.device)
            loss = self.loss_function(output, target)
            loss.backward()
            self.optimizer.step()
            li = loss.item()
            self.losses.append(li)
            total += li
            count += 1
            self.log.moving_train_loss = li if self.log.moving_train_loss is None else \
                (1 - self.eps_train_loss) * self.log.moving_train_loss + self.eps_train_loss * li
            self.lr_scheduler.step()                                                     # update_lr()
            if self.log.epoch == 0 and self.log.iter % 10 == 0:
                self._log('                Iter %12d       %12.5f\n' % (self.log.iter, self.log.moving_train_loss), 'a')
        if count:
            avg = total / count
            self.log.average_train_loss = avg
            if avg < self.log.best_average_train_loss:
                self.log.best_average_train_loss = avg
                self.log.save_status.append('best_train_loss')
            self.log.list_average_train_loss_associate_iter.append([avg, self.log.iter])

    def val(self):
        self.network.eval()
        v = self.val_function(self)
        self.log.average_val_index = v
        if v > self.log.best_average_val_index:
            self.log.best_average_val_index = v
            self.log.save_status.append('best_val_evaluation_index')
        self.log.list_average_val_index_associate_iter.append([v, self.log.iter])

    def save_trainer(self, status):
        ckpt = {'network_state_dict': self.network.state_dict(), 'lr_scheduler_state_dict': self.lr_scheduler.state_dict(),
                'optimizer_state_dict': self.optimizer.state_dict(), 'log': self.log}
        torch.save(ckpt, os.path.join(self.output_dir, status + '.pkl'))
        self._log('        ==> Saving ' + status + ' model successfully !\n', 'a')

    def run(self):
        self._log('Start training !\n', 'w')
        self._log(time.strftime('Local time: %H:%M:%S\n', time.localtime(time.time())), 'a')
        while self.log.epoch < self.max_epoch - 1 and self.log.iter < self.max_iter - 1:
            self.log.epoch += 1
            self._log('Epoch: %d, iter: %d\n' % (self.log.epoch, self.log.iter), 'a')
            lr0 = self.optimizer.param_groups[0]['lr']
            self._log('    Begin lr is %12.12f, %12.12f\n' % (lr0, self.optimizer.param_groups[-1]['lr']), 'a')
            self.log.list_lr_associate_iter.append([lr0, self.log.iter])
            self.train()
            self.val()
            if not self.update_on_iter:
                self.lr_scheduler.step()
            self.log.save_status.append('latest')
            for status in self.log.save_status:
                self.save_trainer(status)
            self.log.save_status = []
            self._log('            Average train loss is             %12.12f,     best is           %12.12f\n' %
                      (self.log.average_train_loss, self.log.best_average_train_loss), 'a')
            self._log('            Average val evaluation index is   %12.12f,     best is           %12.12f\n' %
                      (self.log.average_val_index, self.log.best_average_val_index), 'a')
            self._log('    End lr is %12.12f, %12.12f\n' % (self.optimizer.param_groups[0]['lr'], self.optimizer.param_groups[-1]['lr']), 'a')
        self._log('===============================> End successfully\n', 'a')
