"""Minimal training step for the reference's configuration (BASELINE.json configs[4]):
forward in training mode, the configured losses (`loss: [render, depth, sdf, vgn]`), backward, ONE flat gradient
all-reduce over RCCL, Adam with the exponential-decay schedule.
ref: src/nr/train/trainer.py:142-158 (step), train/lr_common_manager.py:19-29 (ExpDecayLR), network/loss.py.

Scenes are independent, so data parallelism is "every rank takes its scenes, gradients are summed": the only
collective is a sum all-reduce of one flat fp32 buffer (4.66 M parameters = 18.6 MB), divided by the global scene count.
Parameters that received no gradient on a rank contribute zeros (a rank with an empty shard, or fix_s > 0 keeping the NeuS
variance frozen for the first steps, neus.py:17-18), so the buffer layout is static.  On the GPU the volumetric path runs in
HIP in both directions (renderer.py autograd.Functions over csrc/gnr_bwd.inc); PyTorch autograd connects them with the 2D
backbones, the grasp head and the losses."""
import torch
import torch.distributed as dist

from . import losses


def train_losses(out, data, cfg=None):
    """Every loss / metric term of the configured losses for one scene's outputs (ref: loss.py, name2loss)."""
    ref = data['ref_imgs_info']
    terms = {}
    terms.update(losses.render_loss(out, use_ray_mask='ray_mask' in out))   # renderer cfg use_ray_mask false drops the key (renderer.py:129-132)
    terms.update(losses.depth_loss(out, ref['true_depth'], ref['depth_range']))
    terms.update(losses.sdf_loss(out, ref['sdf_gt']))
    terms.update(losses.vgn_loss(out['vgn_pred'], data['grasp_info']))
    return terms


def train_losses_stacked(st, datas):
    """`train_losses` of B scenes at once on the scene-major stacks of forward_scenes(..., stacked=True): the same terms,
    each a [B] vector with one entry per scene (losses.py `scenes=B`)."""
    B = len(datas)
    refs = [d['ref_imgs_info'] for d in datas]
    terms = {}
    terms.update(losses.render_loss(st, use_ray_mask='ray_mask' in st))    # the reference's qn axis is the scene axis here
    terms.update(losses.depth_loss(st, torch.cat([r['true_depth'] for r in refs]), torch.cat([r['depth_range'] for r in refs]), scenes=B))
    terms.update(losses.sdf_loss(st, torch.stack([r['sdf_gt'] for r in refs]), scenes=B))
    terms.update(losses.vgn_loss(st['vgn_pred'], tuple(torch.stack(x) for x in zip(*[d['grasp_info'] for d in datas])), scenes=B))
    return terms


def exp_decay_lr(step, lr_init=1e-4, decay_step=100000, decay_rate=0.5, lr_min=1e-5):
    return max(lr_init * (decay_rate ** (step // decay_step)), lr_min)


class Trainer:
    def __init__(self, net, lr_cfg=None, batched=True, log_every=1, flat_exchange='auto', reproducible_feature_grads=False):
        """reproducible_feature_grads: the path's feature-map gradients (the gradients it hands the 2D backbones) through 64-bit
        fixed-point adds instead of float sums in arrival order (include/gnr.h GNR_OPT_FEATURE_GRAD_FIXED): with it every gradient the HIP
        path produces is the same bits from run to run, as on the reference's CPU path.  A per-call option of THIS net's calls (set either
        way here: another Trainer in the process is not affected, and a later Trainer with False is not left in the fixed-point mode).
        A backward whose status words carry GNR_STATUS_LOST_PARTNER (include/gnr.h: a wavefront of the view kernels gave up waiting for its
        partner -- the gradients are garbage) does not reach the parameters: the flag is tested ON THE DEVICE (no host wait), joins the
        gradient exchange, and the fused Adam step skips itself on it (`found_inf`, the mechanism of torch's GradScaler);
        `skipped_steps()` counts them.
        log_every: the loss terms leave the device every log_every-th step only (the reference writes its log every
        `train_log_step` = 20 steps, trainer.py:31,159; what it reads back EVERY step is the loss shown in its progress bar, :190).  With
        log_every = 1 every step() returns floats -- and ends in a device-to-host copy the host waits for, so the queue of the next step
        starts empty: the GPU idles while the host launches its first kernels, every step.  In between step() returns only `lr`; the terms
        of the latest step stay on the device until last_log() asks for them."""
        self.net = net
        self.batched = batched
        nr = getattr(net, 'nr_net', None)
        if hasattr(nr, 'set_hot_option'):
            nr.set_hot_option('feature_grad_fixed', bool(reproducible_feature_grads))
        self._bad = None                                   # device bool: a backward of the current step flagged GNR_STATUS_LOST_PARTNER
        self._skipped = None                               # device counter of skipped optimiser steps
        self.log_every = max(int(log_every), 1)
        self._pending = None
        self.flat_exchange = flat_exchange
        self._flat = self._views = None
        self.lr_cfg = lr_cfg or {}
        self.params = [p for p in net.parameters()]
        # lr_common_manager.py:9-13.  Fused on the GPU: one multi-tensor kernel, and the only Adam that takes the device-side skip flag
        self._fused = bool(self.params) and all(p.is_cuda for p in self.params)
        self.optimizer = torch.optim.Adam(self.params, lr=1e-3, **({'fused': True} if self._fused else {}))
        self.step_id = 0

    def _mode_modules(self):
        ms = self.__dict__.get('_mode_mods')
        if ms is None:
            names = ('nr_net', 'vgn_net')
            ms = self.__dict__['_mode_mods'] = [getattr(self.net, n) for n in names if isinstance(getattr(self.net, n, None), torch.nn.Module)]
            nr = getattr(self.net, 'nr_net', None)
            for n in ('agg_net', 'fine_agg_net', 'image_encoder', 'init_net', 'vis_encoder'):
                if isinstance(getattr(nr, n, None), torch.nn.Module):
                    ms.append(getattr(nr, n))
        return ms

    def _note_backward_status(self):
        """After a backward: OR the lost-partner bit of the HIP path's status words into the step's flag -- on the device, nothing waits."""
        hot = getattr(getattr(self.net, 'nr_net', None), '_hot', None)
        if hot is None or getattr(hot, '_prepared', None) is None:
            return
        bad = (hot.status_words() & 16).ne(0).any()
        self._bad = bad if self._bad is None else (self._bad | bad)

    def skipped_steps(self):
        """Optimiser steps skipped because a backward reported GNR_STATUS_LOST_PARTNER (one device-to-host copy)."""
        return 0 if self._skipped is None else int(self._skipped.item())

    def _flat_views(self):
        """One persistent flat fp32 buffer for the gradient exchange (all parameters + a bad-step flag + one scene counter) and the per-parameter views
        into it, made once: a step then moves the gradients in with ONE multi-tensor copy and out with one (no 346-way torch.cat,
        no per-tensor reshape / split on the host every step)."""
        if self._flat is None or self._flat.device != self.params[0].device:
            n = sum(p.numel() for p in self.params)
            self._flat = torch.zeros(n + 2, dtype=torch.float32, device=self.params[0].device)
            self._views = [v.view(p.shape) for v, p in zip(torch.split(self._flat[:-2], [p.numel() for p in self.params]), self.params)]
        return self._flat, self._views

    def _allreduce_grads(self, n_local):
        """Sum of per-scene gradients over all ranks / global scene count, through one flat buffer: one multi-tensor copy in,
        one all-reduce, one multi-tensor copy back (parameters without a gradient on this rank count as zeros).
        flat_exchange: 'auto' = only when there is more than one rank; True = always (one rank: the collective is RCCL's
        single-rank all-reduce -- how bench.py prices the N > 1 path of a step on one GPU)."""
        ready = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if ready else 1
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        grads = [p.grad for p in self.params]
        if not (world > 1 or (self.flat_exchange is True and ready)):
            torch._foreach_div_(grads, float(max(n_local, 1)))
            return world
        flat, views = self._flat_views()
        torch._foreach_copy_(views, grads)
        flat[-1:].fill_(float(n_local))                     # (a kernel launch; `flat[-1] = x` is a pageable host-to-device copy the HOST waits for:
                                                            #  measured, tools/dbg/allreduce_blocking.py -- it cost the step its run-ahead, 3 - 4.6 ms)
        if self._bad is None:
            flat[-2:-1].zero_()
        else:
            flat[-2:-1].copy_(self._bad.to(torch.float32).reshape(1))      # a rank's garbage gradients must not reach ANY rank's parameters
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat[:-2].div_(flat[-1:].clamp(min=1.0))           # (a one-element tensor, broadcast on the device)
        if self._bad is not None or self._fused:
            self._bad = flat[-2:-1].gt(0).reshape(())
        torch._foreach_copy_(grads, views)
        return world

    def step(self, scenes):
        """scenes: list of `data` dicts (this rank's share of the global batch).  Gradients of the per-scene total
        losses are accumulated, all-reduced, averaged over the global scene count; one Adam update.
        -> dict of loss terms averaged over the local scenes, lr."""
        # (train() walks every sub-module: not every step.)  The root flag alone is not enough: `net.nr_net.eval()` for a render-only
        # validation leaves the root in training mode, and the next steps would silently train with the renderer in eval mode (eval
        # sampling, no NeuS step bookkeeping) -- every module whose behaviour depends on the mode is checked.
        if not (self.net.training and all(m.training for m in self._mode_modules())):
            self.net.train()
        lr = exp_decay_lr(self.step_id, **self.lr_cfg)
        for g in self.optimizer.param_groups:
            g['lr'] = lr
        self.optimizer.zero_grad(set_to_none=True)
        datas = [dict(d, step=self.step_id) for d in scenes]
        outs = st = None
        if self.batched and hasattr(self.net, 'forward_scenes') and len(datas) > 1:
            # all scenes of the rank in one forward (None: cannot batch): first choice scene-major stacks and ONE set of
            # loss launches for the batch, second the per-scene dicts and the per-scene losses
            st = self.net.forward_scenes(datas, stacked=True)
            outs = self.net.forward_scenes(datas) if st is None else None
        self._bad = None
        if st is not None:
            terms = train_losses_stacked(st, datas)
            losses.total_loss(terms, scenes=len(datas)).backward()
            self._note_backward_status()
            all_terms = terms
        elif outs is not None:
            all_terms = [train_losses(o, d) for o, d in zip(outs, datas)]
            sum(losses.total_loss(t) for t in all_terms).backward()
            self._note_backward_status()
        else:
            all_terms = []
            for data in datas:
                terms = train_losses(self.net(data), data)
                losses.total_loss(terms).backward()                       # accumulates into .grad
                self._note_backward_status()                              # (the next scene's prepare zeroes the status words)
                all_terms.append(terms)
        self._allreduce_grads(len(scenes))
        if self._bad is not None and self._fused:
            # the step skips itself on the device when a backward lost a partner wavefront: no host wait, no garbage in the parameters
            found = self._bad.to(torch.float32).reshape(())              # (0-dim, as torch's GradScaler hands it over)
            self._skipped = found.clone() if self._skipped is None else self._skipped + found
            self.optimizer.grad_scale, self.optimizer.found_inf = None, found
            try:
                self.optimizer.step()
            finally:
                del self.optimizer.grad_scale, self.optimizer.found_inf
        else:
            self.optimizer.step()
        if self._fused:
            # torch's fused Adam writes the parameters without bumping their version counters (2.10: checked), and the packed copies of
            # the hot path / the grasp head are refreshed when a version moves (renderer._hot_versions): bump them (no data is touched)
            torch.autograd.graph.increment_version(self.params)
        self.step_id += 1
        # loss terms leave the device in ONE copy after the whole step is queued (a float() per term and scene would
        # stall the host 80 times in front of the all-reduce and the optimiser)
        if not all_terms:                                  # empty shard (global batch < world): took part in the all-reduce only
            return {'lr': lr}
        if isinstance(all_terms, dict):                    # [B] vectors: the mean over the local scenes is the mean of each
            keys = list(all_terms)
            means = torch.stack([all_terms[k].detach().float().mean() for k in keys])
        else:
            keys = list(all_terms[0])
            means = torch.stack([torch.stack([t[k].detach().float().mean() for k in keys]) for t in all_terms]).mean(0)
        self._pending = (keys, means, lr)
        if self.step_id % self.log_every:                  # not a logging step: nothing leaves the device, the host runs on
            return {'lr': lr}
        return self.last_log()

    def last_log(self):
        """Loss terms of the latest step as floats (one device-to-host copy; the only synchronisation of a step)."""
        if self._pending is None:
            return {}
        keys, means, lr = self._pending
        log = dict(zip(keys, means.tolist()))
        log['lr'] = lr
        return log
