"""Training loop (reference: caspr/utils/train_utils.py:82-232 `run_one_epoch`, caspr/train.py:135-190).

Same loss and bookkeeping as the reference: loss = cnf_loss_weight * mean_{b,t}(sum_n nll) + tnocs_loss_weight *
mean(L1 tnocs) (train_utils.py:151-165), Adam, periodic checkpoints `time_model_<epoch>.pth` and
`BEST_time_model.pth` on the best validation loss (train.py:160-188).

Multi-GPU: instead of nn.DataParallel (one process, scatter/gather, train.py:131-132) each rank owns a
contiguous block of the batch's sequences; the only collective is ONE all-reduce (RCCL over xGMI on the GPU box) of a
flat bucket holding every gradient (and one touched-flag per parameter), after which each rank applies the same Adam step.  The bucket carries each rank's
gradient of its LOCAL mean weighted by its shard size (plus the shard size itself in one extra slot), so the reduced
gradient is the gradient of the reference's mean over the GLOBAL batch (train_utils.py:154,163) for any split --
including a rank whose shard is empty (a last batch shorter than the world size): it contributes zeros, still enters
the collective and applies the same step, so replicas never diverge or deadlock.  MovingBatchNorm running statistics are
rank-local (the reference's DataParallel keeps replica 0's); rank 0's are the ones checkpointed.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from ..utils.sharding import shard_range


def training_loss(losses, cnf_loss_weight, tnocs_loss_weight):
    """(loss, cnf_loss, tnocs_loss) from CaSPR.forward's tuple (train_utils.py:131-165)."""
    if len(losses) == 1:
        per_point_nll, per_point_tnocs = None, losses[0]
    elif len(losses) == 2:
        per_point_nll, per_point_tnocs = losses
    else:
        raise ValueError("unexpected number of losses returned: %d" % len(losses))
    ref = per_point_tnocs if per_point_tnocs is not None else per_point_nll
    loss = torch.zeros(1, device=ref.device)
    cnf_loss = tnocs_loss = torch.zeros(1, device=ref.device)
    if per_point_nll is not None:
        cnf_loss = cnf_loss_weight * per_point_nll.sum(2).mean()
        loss = loss + cnf_loss
    if per_point_tnocs is not None:
        tnocs_loss = tnocs_loss_weight * per_point_tnocs[:, :, :, :4].mean()
        loss = loss + tnocs_loss
    return loss, cnf_loss, tnocs_loss


class GradBucket:
    """All gradients of a model in one flat f32 buffer (16,262,189 elements = 65 MB for the full model): a single all-reduce
    per step instead of one per tensor -- and no copies around it: every parameter's `.grad` IS a view of the bucket
    (autograd accumulates into an existing .grad in place), so the collective runs on the gradients where they are.
    `zero()` replaces optimizer.zero_grad() (whose default, set_to_none, would detach the views; `attach()` re-creates any
    view a caller has dropped that way)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        self.flat = None
        self.views = None
        # Which parameters the loss reaches.  With .grad bound to a zeroed view EVERY parameter has a gradient tensor, also the
        # ones no loss term touches; the reference (optimizer.zero_grad(): set_to_none) leaves those at None and Adam skips them
        # -- no state, no weight decay.  A post-accumulate hook marks the parameters autograd wrote to THIS step; the marks ride in
        # the gradient all-reduce itself (one flag per parameter behind the gradients: SUM > 0 = some rank's loss reached it), so
        # the decision is collective by construction -- every rank enters the same ONE collective every step, whatever its local
        # marks are (a rank without data, a loss-weight schedule that switches a term on later, layers unfrozen mid-run) -- and it
        # is taken per step, like the reference's zero_grad(): a parameter touched once is NOT kept alive afterwards.
        self.timers = None         # a list: all_reduce_mean appends the duration of its collective (HIP event pairs on the current stream under
                                   # RCCL -- read with collective_ms(); seconds of wall clock under gloo, whose copy is on the host)
        self._touched = set()
        self._mask = None          # the reduced flags of the last all_reduce_mean (None: no collective ran -> the local marks decide)
        self._hooks = [p.register_post_accumulate_grad_hook(lambda t, i=i: self._touched.add(i)) for i, p in enumerate(self.params)]

    def close(self):
        """Remove the hooks (a second GradBucket on the same model would otherwise stack its own on top of these)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def attach(self):
        """(Re)bind every .grad to its slice of the flat buffer; gradient values already present are kept.
        Layout: [gradients (sum of sizes) | one touched-flag per parameter | this rank's weight]."""
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(sum(self.sizes) + len(self.params) + 1, device=dev, dtype=torch.float32)
            self.views, off = [], 0
            for p, n in zip(self.params, self.sizes):
                self.views.append(self.flat[off:off + n].view_as(p))
                off += n
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
                p.grad = v
            elif g.data_ptr() != v.data_ptr() or g.device != v.device:
                v.copy_(g)
                p.grad = v
        return self.flat

    def zero(self):
        self.attach()
        self.flat.zero_()
        self._touched.clear()
        self._mask = None

    def drop_untouched(self):
        """After backward (and the all-reduce): parameters no rank's loss reached THIS step get .grad = None, as in the reference, so
        that the optimizer skips them (zero() re-binds the views).  No collective here: with several ranks the flags were reduced
        inside all_reduce_mean; alone, the local marks are the answer."""
        if self._mask is not None:
            keep = self._mask
        else:
            keep = [i in self._touched for i in range(len(self.params))]
        for p, k in zip(self.params, keep):
            if not k:
                p.grad = None

    def all_reduce_mean(self, weight=1.0):
        """Weighted average of the gradients over the ranks: sum_r weight_r * grad_r / sum_r weight_r, with weight = the
        number of sequences behind this rank's (mean-reduced) loss; weight 0 = a rank without data (its gradients are
        taken as zero whatever .grad holds).  ONE collective, in place on the gradients, carrying the touched-flags too (every
        rank enters it every step).  No-op without an initialised process group."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        flat = self.attach()
        ng, npar = sum(self.sizes), len(self.params)
        if weight == 0:
            flat.zero_()
        else:
            flat[:ng].mul_(float(weight))
            flags = torch.zeros(npar, dtype=torch.float32)
            if self._touched:
                flags[sorted(self._touched)] = 1.0
            flat[ng:ng + npar].copy_(flags, non_blocking=True)
            flat[-1] = float(weight)
        if self.timers is None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        elif dist.get_backend() == "gloo":
            import time
            torch.cuda.synchronize(flat.device) if flat.is_cuda else None
            t0 = time.perf_counter()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize(flat.device) if flat.is_cuda else None
            self.timers.append(time.perf_counter() - t0)
        else:
            # RCCL: the collective runs on the library's own stream, which waits for the current stream at entry and which the current stream
            # waits for at exit (async_op=False) -- an event pair on the current stream brackets exactly the collective
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            b.record()
            self.timers.append((a, b))
        flat[:ng].div_(flat[-1].clamp_min(1e-30))
        self._mask = (flat[ng:ng + npar] > 0).cpu().tolist()


def collective_ms(timers):
    """Milliseconds of each record of GradBucket.timers (synchronises: call after the timed region)."""
    out = []
    for t in timers or []:
        if isinstance(t, tuple):
            t[1].synchronize()
            out.append(t[0].elapsed_time(t[1]))
        else:
            out.append(1e3 * t)
    return out


def broadcast_model(model, src=0):
    """Every replica starts from rank `src`'s parameters AND buffers (MovingBatchNorm running statistics, step counters): the
    reference's nn.DataParallel replicates module 0 on every forward (train.py:131-132); with one process per GPU the replicas
    agree only if they start equal and apply the same averaged gradient -- whatever seed each rank initialised with.  One
    flat broadcast per dtype.  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
    seen, uniq = set(), []
    for t in tensors:                      # latent_ode.solver.ode_func.* aliases latent_ode.ode_func.*
        if t.data_ptr() not in seen and t.numel():
            seen.add(t.data_ptr())
            uniq.append(t)
    for dt in sorted({t.dtype for t in uniq}, key=str):
        group = [t for t in uniq if t.dtype == dt]
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src)
        off = 0
        for t in group:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def train_step(model, optimizer, pcl_in, nocs_out, cnf_loss_weight=0.01, tnocs_loss_weight=100.0, bucket=None, e=None):
    """One optimisation step on this rank's sequences (train_utils.py:120-176).  Returns the python floats
    (loss, cnf_loss, tnocs_loss) of the local shard."""
    model.train()
    if bucket is not None:
        bucket.zero()              # the gradients live in the bucket (views): zero them there, keep the views
    else:
        optimizer.zero_grad()
    n_local = int(pcl_in.shape[0])
    if n_local == 0:
        # no sequence of this batch landed on this rank: zero gradient, but the collective and the (identical) Adam step
        # still happen -- skipping them would leave the other ranks blocked in the all-reduce and this replica behind
        if bucket is None:
            return float('nan'), float('nan'), float('nan')
        bucket.all_reduce_mean(weight=0)
        bucket.drop_untouched()
        optimizer.step()
        return float('nan'), float('nan'), float('nan')
    losses = model(pcl_in, nocs_out, e=e) if e is not None else model(pcl_in, nocs_out)
    loss, cnf_loss, tnocs_loss = training_loss(losses, cnf_loss_weight, tnocs_loss_weight)
    loss.backward()
    if bucket is not None:
        bucket.all_reduce_mean(weight=n_local)
        bucket.drop_untouched()
    optimizer.step()
    return float(loss.detach()), float(cnf_loss.detach()), float(tnocs_loss.detach())


def shard_batch(pcl_in, nocs_out, rank=None, world=None):
    """This rank's contiguous block of the global batch (frames of one sequence stay together)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    lo, hi = shard_range(pcl_in.shape[0], rank, world)
    return pcl_in[lo:hi], nocs_out[lo:hi]


def run_one_epoch(model, data_loader, device, optimizer, cnf_loss_weight, tnocs_loss_weight, epoch, log=print, mode='train',
                  print_stats_every=10, bucket=None):
    """train_utils.py:82-232 without the plotting: iterates `data_loader` (items: ((pcl_in, nocs_out), ...)), returns the
    list of per-batch losses (train) or the mean loss (val / test)."""
    if mode not in ['train', 'val', 'test']:
        raise ValueError('mode must be train, val or test')
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    out, wsum, nsum, bsum, bnum = [], 0.0, 0, 0.0, 0
    for i, data in enumerate(data_loader):
        pcl_in, nocs_out = data[0]
        pcl_in, nocs_out = shard_batch(pcl_in.to(device), nocs_out.to(device))
        n_local = int(pcl_in.shape[0])
        if mode == 'train':
            if n_local == 0 and not distributed:
                continue
            loss, cnf_l, tnocs_l = train_step(model, optimizer, pcl_in, nocs_out, cnf_loss_weight, tnocs_loss_weight, bucket)
        else:
            if n_local == 0:
                continue            # evaluation has no collective inside the loop: the reduction happens once, below
            model.eval()
            with torch.no_grad():
                loss, cnf_l, tnocs_l = (float(v) for v in training_loss(model(pcl_in, nocs_out), cnf_loss_weight, tnocs_loss_weight))
            wsum, nsum = wsum + loss * n_local, nsum + n_local
            bsum, bnum = bsum + loss, bnum + 1
        if n_local:
            out.append(loss)
            if i % print_stats_every == 0:
                log('%s epoch %d batch %d/%d: loss %.6f (cnf %.6f, tnocs %.6f)' % (mode, epoch, i, len(data_loader), loss, cnf_l, tnocs_l))
    if mode == 'train':
        return out
    # One process: the reference's number exactly -- the mean of the per-batch means (train_utils.py:226), ragged last batch
    # included -- so that the BEST-checkpoint decision is the reference's.  Several ranks: the mean over every SEQUENCE of every
    # rank (a rank's "batches" are shards, their per-batch means are not the reference's either; the sequence-weighted mean is the
    # one that does not depend on the number of GPUs, and equals the reference's whenever every batch is full).
    if not distributed:
        return bsum / bnum if bnum else float('nan')
    if distributed:
        acc = torch.tensor([wsum, float(nsum)], dtype=torch.float64, device=device)
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        wsum, nsum = float(acc[0]), int(acc[1])
    return wsum / nsum if nsum else float('nan')


def train(model, train_loader, val_loader, device, out_dir, num_epochs, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
          cnf_loss_weight=0.01, tnocs_loss_weight=100.0, val_every=1, save_every=1, log=print, resume=None):
    """train.py:135-190: Adam, validation every `val_every` epochs with BEST checkpointing, periodic checkpoints.
    `resume` = path of a checkpoint written by this function (model + optimizer + epoch)."""
    optimizer = torch.optim.Adam(model.parameters(), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    bucket = GradBucket(model.parameters())
    start, val_losses = 0, []
    if resume:
        ck = torch.load(resume, map_location=device)
        model.load_state_dict(ck["model"])
        optimizer.load_state_dict(ck["optimizer"])
        start, val_losses = ck["epoch"] + 1, ck.get("val_losses", [])
    broadcast_model(model, 0)      # replicas start from rank 0's parameters and buffers, whatever each rank's seed was
    is_rank0 = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
    for epoch in range(start, num_epochs):
        run_one_epoch(model, train_loader, device, optimizer, cnf_loss_weight, tnocs_loss_weight, epoch, log, 'train', bucket=bucket)
        if val_loader is not None and epoch % val_every == 0:
            val = run_one_epoch(model, val_loader, device, None, cnf_loss_weight, tnocs_loss_weight, epoch, log, 'val')
            if not math.isnan(val):
                best = len(val_losses) == 0 or val < min(val_losses)
                val_losses.append(val)
                if best and is_rank0:
                    log('BEST Val loss so far! Saving checkpoint...')
                    torch.save(model.state_dict(), os.path.join(out_dir, 'BEST_time_model.pth'))
        if epoch % save_every == 0 and is_rank0:
            torch.save(model.state_dict(), os.path.join(out_dir, 'time_model_%d.pth' % epoch))          # reference-format weights
            torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch, "val_losses": val_losses},
                       os.path.join(out_dir, 'resume_%d.pth' % epoch))
    return val_losses
