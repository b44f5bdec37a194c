"""Data-parallel gradient exchange for the hot path: one process per GPU, RCCL over xGMI (torch.distributed backend
"nccl" IS RCCL on ROCm), bucketed all-reduce overlapped with backward.

What it replaces: the reference wraps the detector in ``MMDistributedDataParallel`` (mmcv/mmcv/parallel/
distributed.py:13-87 -> torch DDP's C++ reducer, 25 MiB buckets, mmrotate/apis/train.py:53-57) and then issues
~15 blocking scalar all-reduces per step for the log vars.  Here:

* ``zero_grad()`` sets every ``p.grad`` to None, so autograd's AccumulateGrad ADOPTS the freshly produced gradient
  tensor instead of launching one ``grad += new`` kernel per parameter into pre-zeroed storage (that was ~250
  launches and ~2 GB of HBM traffic per step for the 178 M-parameter backbone);
* world_size > 1: gradients are packed into a few large flat buckets sized for xGMI -- the 8-GPU mesh is
  point-to-point (7 links x ~153 GB/s per GPU), so a ring collective is per-link bound and small messages are
  latency-bound; default 64 MiB buckets (~1.7 ms each at one-link ring speed) give ~11 collectives instead of DDP's
  ~28.  Buckets are filled in reverse parameter order (= the order backward produces gradients); the moment the
  last gradient of a bucket exists it is packed (one multi-tensor copy) and all-reduced asynchronously on RCCL's own
  stream, overlapping the remaining backward kernels;
* the collectives AVERAGE in flight (``ReduceOp.AVG`` on RCCL; backends without it -- gloo in the CPU tests -- sum and
  ``finalize()`` divides), so no extra 0.56 GB pass over the buckets follows them; ``finalize()`` waits for the
  outstanding collectives and re-points every ``p.grad`` at its slice of the reduced bucket (no copy back), so any
  optimizer sees the mean;
* world_size == 1: nothing is allocated or launched; ``p.grad`` stays the tensor autograd produced;
* ``allreduce_scalars()`` fuses any number of logging scalars into ONE small all-reduce.

The class is backend-agnostic (it only uses ``torch.distributed`` collectives), so the world_size-2 CPU tests run it
over ``gloo``.  MoE routing is rank-local (all experts replicated, no all-to-all), as in the reference.
"""
import torch
import torch.distributed as dist


class BucketedGradReducer:
    def __init__(self, params, bucket_mb=64.0, process_group=None, groups=None, force_comm=False):
        """params: ANY trainable parameters whose gradients are to be averaged -- the backbone's, or the whole detector's
        (backbone + neck + heads), exactly what the reference hands to DDP.
        groups: optional list of parameter lists (a partition of ``params``, in the order their gradients become
        available); buckets never straddle two groups, so a whole group can be packed / all-reduced as soon as the
        backward segment that produces it is done (``pack_group`` / ``allreduce_group_async``).
        force_comm: run the bucket + collective path even in a 1-rank group (the collectives are then identities); used
        to exercise RCCL initialisation, stream hand-over and hipGraph coexistence on a single-GPU box."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.comm = self.world > 1 or (bool(force_comm) and dist.is_initialized())
        # average inside the collective where the backend can (RCCL); otherwise sum and divide in finalize()
        self._avg = bool(self.comm and dist.get_backend(process_group) == 'nccl')
        self._op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        cap = int(bucket_mb * 1024 * 1024)
        if groups is None:
            groups = [list(reversed(self.params))]  # reverse order: the last layers' gradients are ready first
        else:
            groups = [[p for p in g if p.requires_grad] for g in groups]
            if sorted(id(p) for g in groups for p in g) != sorted(id(p) for p in self.params):
                raise ValueError('groups must partition the trainable parameters')
        self.buckets = []  # dicts: flat (None when world == 1), params, views, pending, handle, group
        for gi, plist in enumerate(groups):
            cur, cur_bytes = [], 0
            for p in plist:
                nbytes = p.numel() * p.element_size()
                if cur and (cur_bytes + nbytes > cap or cur[0].dtype != p.dtype or cur[0].device != p.device):
                    self._close(cur, gi)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nbytes
            if cur:
                self._close(cur, gi)
        self.pack_stats = dict(copied_bytes=0, in_place_bytes=0)  # cumulative over the packs issued from Python
        self._index = {}
        for bi, b in enumerate(self.buckets):
            for p in b['params']:
                self._index[p] = bi
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._armed = False
        # overlap=True: each bucket is packed and all-reduced from the autograd hook of its last gradient (eager mode).
        # overlap=False: hooks only count; pack_all() / finalize() do the work back to back (used when forward+backward
        # are replayed from a captured hipGraph, where Python hooks do not run).
        self.overlap = True

    def _close(self, plist, group=0):
        flat, views = None, None
        if self.comm:
            total = sum(p.numel() for p in plist)
            flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
            views, off = [], 0
            for p in plist:
                n = p.numel()
                views.append(flat[off:off + n].view_as(p))
                # backward kernels that know their parameter write the gradient straight into the bucket slice
                # (backbone_ops._bucket_out: the FFN / expert weight gradients, 95 % of the bytes) -> _pack copies nothing
                p._sm3_grad_view = views[-1]
                off += n
        self.buckets.append(dict(flat=flat, params=plist, views=views, pending=len(plist), handle=None,
                                 packed=False, group=group))

    # ------------------------------------------------------------------------------------------------ step API
    def zero_grad(self):
        """Call instead of optimizer.zero_grad(): drops the gradients (set_to_none) and re-arms the buckets."""
        for p in self.params:
            p.grad = None
        for b in self.buckets:
            b['pending'] = len(b['params'])
            b['handle'] = None
            b['packed'] = False
        self._armed = True

    def _pack(self, b):
        """gradients of one bucket -> its flat buffer (parameters without a gradient contribute zeros)."""
        if b['packed'] or not self.comm:
            return
        src, dst = [], []
        for p, v in zip(b['params'], b['views']):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
                self.pack_stats['copied_bytes'] += p.numel() * p.element_size()
            else:  # the backward wrote this gradient into the bucket slice itself
                self.pack_stats['in_place_bytes'] += p.numel() * p.element_size()
        if src:
            torch._foreach_copy_(dst, src)
        b['packed'] = True

    def pack_all(self):
        """Pack every bucket now (the tail of a captured forward+backward graph; finalize() then only reduces)."""
        for b in self.buckets:
            self._pack(b)

    def pack_group(self, g):
        for b in self.buckets:
            if b['group'] == g:
                self._pack(b)

    def allreduce_group_async(self, g):
        """issue the all-reduces of one group's (already packed) buckets; finalize() waits for them"""
        if self.comm:
            for b in self.buckets:
                if b['group'] == g and b['handle'] is None:
                    b['handle'] = dist.all_reduce(b['flat'], op=self._op, group=self.group, async_op=True)

    def _on_grad(self, p):
        if not self._armed:
            return
        b = self.buckets[self._index[p]]
        b['pending'] -= 1
        if b['pending'] == 0 and self.comm and self.overlap:
            # parameter gradients whose column reductions are batched at the end of the backward pass (backbone_ops) must
            # hold their values before this bucket is packed in the middle of it
            from . import backbone_ops
            backbone_ops.flush_deferred_reductions()
            self._pack(b)
            b['handle'] = dist.all_reduce(b['flat'], op=self._op, group=self.group, async_op=True)

    def finalize(self, repack=True):
        """Wait for the in-flight collectives (packing + launching any bucket whose last gradient never arrived), turn
        the sums into means and point every ``p.grad`` at its reduced slice.  ``repack=False``: the flats were filled by
        a replayed graph that ended in ``pack_all()`` -- only reduce."""
        if self.comm:
            for b in self.buckets:
                if b['handle'] is None:
                    if repack:
                        self._pack(b)
                    b['handle'] = dist.all_reduce(b['flat'], op=self._op, group=self.group, async_op=True)
            for b in self.buckets:
                b['handle'].wait()
                if not self._avg and self.world > 1:
                    b['flat'].div_(self.world)
                b['handle'] = None
                for p, v in zip(b['params'], b['views']):
                    p.grad = v
        self._armed = False

    def allreduce_scalars(self, values):
        """mean of a list of 0-d tensors over the ranks with ONE collective (replaces the reference's one
        all-reduce + .item() per log var)."""
        t = torch.stack([v.detach().float().reshape(()) for v in values])
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t /= self.world
        return t

    def broadcast_parameters(self, src=0, module=None, extra=()):
        """rank-src state to everyone (torch DDP does this at construction): the trainable parameters, plus -- when
        ``module`` is given -- every parameter (frozen stages included) and buffer of it, plus any ``extra`` tensors."""
        if self.world > 1:
            seen, todo = set(), []
            sources = [self.params, extra]
            if module is not None:
                sources += [module.parameters(), module.buffers()]
            for src_list in sources:
                for t in src_list:
                    if id(t) not in seen:
                        seen.add(id(t))
                        todo.append(t)
            for t in todo:
                dist.broadcast(t.data, src=src, group=self.group)

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    @property
    def num_buckets(self):
        return len(self.buckets)
