"""CPU tier: the N>1 path (sm3det_amd.data_parallel.BucketedGradReducer) with world_size 2 over gloo.
Checks: bucketed, overlapped all-reduce == mean of per-rank gradients == single-process gradient of the mean loss;
several buckets; parameters that receive no gradient; fused scalar all-reduce; parameter broadcast."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.body = nn.Sequential(nn.Linear(16, 64), nn.GELU(), nn.Linear(64, 64), nn.GELU(), nn.Linear(64, 8))
        self.unused = nn.Linear(4, 4)  # never used in forward: gets no gradient

    def forward(self, x):
        return self.body(x)


def _model(seed):
    torch.manual_seed(seed)
    return _Net()


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sm3det_amd.data_parallel import BucketedGradReducer
        model = _model(seed=100 + rank)  # different init per rank on purpose
        red = BucketedGradReducer(model.parameters(), bucket_mb=0.01)  # ~10 KB buckets -> several buckets
        red.broadcast_parameters(0)
        ref = _model(seed=100)
        for a, b in zip(model.parameters(), ref.parameters()):
            assert torch.equal(a, b)
        g = torch.Generator().manual_seed(7)
        X = torch.randn(8, 16, generator=g)
        Y = torch.randn(8, 8, generator=g)
        xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
        for step in range(2):  # two steps: zero_grad must re-arm the buckets
            red.zero_grad()
            loss = ((model(xs) - ys) ** 2).mean()
            loss.backward()
            red.finalize()
        # single-process reference: mean over both shards == mean of the two per-rank means
        ref.zero_grad()
        (((ref(X) - Y) ** 2).mean()).backward()
        err = 0.0
        for (n, a), b in zip(model.named_parameters(), ref.parameters()):
            if b.grad is None:
                assert torch.count_nonzero(a.grad) == 0, n
                continue
            err = max(err, float((a.grad - b.grad).abs().max()))
        s = red.allreduce_scalars([loss, torch.tensor(float(rank))])
        q.put((rank, red.num_buckets, err, float(s[1])))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nb, err, mean_rank in res:
        assert nb >= 3
        assert err < 1e-6
        assert abs(mean_rank - 0.5) < 1e-6


def _worker_groups(rank, world, port, q):
    """two parameter groups, each packed + all-reduced as a unit (the two-segment backward of bench.py for N>1)"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sm3det_amd.data_parallel import BucketedGradReducer
        model = _model(seed=100)
        named = list(model.named_parameters())
        late = [p for n, p in reversed(named) if n.startswith(('body.4', 'body.2'))]
        early = [p for n, p in reversed(named) if not n.startswith(('body.4', 'body.2'))]
        red = BucketedGradReducer(model.parameters(), bucket_mb=0.01, groups=[late, early])
        red.overlap = False
        assert all(b['group'] in (0, 1) for b in red.buckets)
        assert {id(p) for b in red.buckets if b['group'] == 0 for p in b['params']} == {id(p) for p in late}
        g = torch.Generator().manual_seed(7)
        X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
        xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
        red.zero_grad()
        h = model.body[:2](xs)  # boundary activation
        loss = ((model.body[2:](h) - ys) ** 2).mean()
        grads = torch.autograd.grad(loss, [h] + late)  # segment 1: everything above h
        for p, gr in zip(late, grads[1:]):
            p.grad = gr
        red.pack_group(0)
        red.allreduce_group_async(0)
        torch.autograd.backward(h, grad_tensors=grads[0], inputs=[p for p in early if p is not model.unused.weight
                                                                  and p is not model.unused.bias])  # segment 2
        red.pack_group(1)
        red.allreduce_group_async(1)
        red.finalize(repack=False)
        ref = _model(seed=100)
        (((ref(X) - Y) ** 2).mean()).backward()
        err = 0.0
        for (n, a), b in zip(model.named_parameters(), ref.parameters()):
            if b.grad is None:
                assert torch.count_nonzero(a.grad) == 0, n
                continue
            err = max(err, float((a.grad - b.grad).abs().max()))
        q.put((rank, err))
    finally:
        dist.destroy_process_group()


def test_grouped_two_segment_allreduce_world2_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_groups, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err in res:
        assert err < 1e-6


def test_groups_must_partition_parameters():
    from sm3det_amd.data_parallel import BucketedGradReducer
    m = _model(0)
    ps = list(m.parameters())
    with pytest.raises(ValueError):
        BucketedGradReducer(ps, groups=[ps[:2], ps[3:]])


def test_single_process_reducer_is_a_noop_mean():
    from sm3det_amd.data_parallel import BucketedGradReducer
    m = _model(0)
    red = BucketedGradReducer(m.parameters(), bucket_mb=64)
    assert red.num_buckets == 1 and red.world == 1
    red.zero_grad()
    m(torch.randn(3, 16)).sum().backward()
    red.finalize()
    assert red.buckets[0]['flat'] is None  # world 1: no flat buffers, autograd's own gradient tensors are kept
    assert float(m.body[0].weight.grad.abs().sum()) > 0
    assert m.unused.weight.grad is None
    red.zero_grad()
    assert all(p.grad is None for p in m.parameters())
