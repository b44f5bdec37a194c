"""GPU tier: gradients of a REPLAYED linear hipGraph equal the eager ones.

Regression test for the round-2 finding: ROCm 7.2 replays linear hipGraphs through a
packet-capture fast path that does not keep memset nodes ordered with the kernels around them.  While the library zeroed
its atomic-accumulation buffers (depthwise weight / bias gradients, bias column sums) with hipMemsetAsync, a captured
training step without the weight-gradient side stream returned unzeroed gradients from the second replay on, although
every eager test passed.  The library now zero-fills with a kernel, so the replay must be right WITH the runtime's
default settings: the check runs in a subprocess with DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 (the harness default is 0 because
of torch's own reduction semaphores, which only affect loss read-outs) and with the side stream both off and on."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
torch.manual_seed(0)
net = ConvNeXt_moe_MultiInput(arch=dict(depths=[1, 2, 2, 1], channels=[32, 64, 96, 128]),
                              MoE_Block_inds=[[], [0], [0, 1], [0]], num_experts=4, top_k=2, drop_path_rate=0.0,
                              noisy_gating=False).cuda().train()  # no RNG in the step: eager and replay are comparable
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith('gamma'):
            p.fill_(1.0)
x = torch.randn(2, 3, 128, 128, device='cuda')
proj = None
def fwd_bwd():
    global proj
    for p in net.parameters():
        p.grad = None
    outs, gl = net(x, ['single'])
    if proj is None:
        proj = [torch.randn_like(o) for o in outs]
    l = gl
    for o, r in zip(outs, proj):
        l = l + (o * r).sum() * 1e-2
    l.backward()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        fwd_bwd()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
eager = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd_bwd()
held = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
worst = 0.0
for rep in range(4):
    g.replay()
    torch.cuda.synchronize()
    for n, ref in eager.items():
        e = float((held[n] - ref).abs().max()) / max(float(ref.abs().max()), 1e-12)
        if not (e < 1e-4):
            print('MISMATCH replay', rep, n, e)
            sys.exit(3)
        worst = max(worst, e)
print('OK', len(eager), 'gradients, worst relative difference', worst)
'''


@pytest.mark.parametrize('side_stream', ['0', '1'])
def test_replayed_graph_gradients_equal_eager_with_default_runtime_settings(side_stream):
    env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE='1', SM3_WGRAD_STREAM=side_stream)
    r = subprocess.run([sys.executable, '-c', CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert 'OK' in r.stdout, r.stdout[-500:]


# ------------------------------------------------------------------------------------------------ concurrent backward partners
PAIR_CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from sm3det_amd import backbone_ops
from sm3det_amd.convnext_moe import ConvNeXt_moe_MultiInput
torch.manual_seed(0)
net = ConvNeXt_moe_MultiInput(arch=dict(depths=[1, 2, 2, 1], channels=[32, 64, 96, 128]),
                              MoE_Block_inds=[[], [0], [0, 1], [0]], num_experts=4, top_k=2, drop_path_rate=0.0,
                              noisy_gating=False).cuda().train()
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith('gamma'):
            p.fill_(1.0)
x = torch.randn(2, 3, 128, 128, device='cuda')
proj = None
def fwd_bwd(level, accumulate=False):
    global proj
    if not accumulate:
        for p in net.parameters():
            p.grad = None
    with backbone_ops.pairing(level):
        outs, gl = net(x, ['single'])
    if proj is None:
        proj = [torch.randn_like(o) for o in outs]
    l = gl
    for o, r in zip(outs, proj):
        l = l + (o * r).sum() * 1e-2
    l.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
def same(a, b, what):
    for n in a:
        # the depthwise weight / bias gradients are built with fp32 atomics (summation order varies run to run); everything
        # else is the same kernel on the same operands and must be bit-identical whatever stream it ran on
        if '.depthwise_conv.' in n or n.endswith('dwconv.weight') or n.endswith('dwconv.bias'):
            e = float((a[n] - b[n]).abs().max()) / max(float(a[n].abs().max()), 1e-12)
            ok = e < 1e-5
        else:
            ok = torch.equal(a[n], b[n])
        if not ok:
            print('MISMATCH', what, n, float((a[n] - b[n]).abs().max()))
            sys.exit(3)
s0 = torch.cuda.Stream()  # never the legacy default stream: autograd's AccumulateGrad nodes remember their stream, and a
s0.wait_stream(torch.cuda.current_stream())  # wait on the NULL stream inside a later capture is illegal
with torch.cuda.stream(s0):
    ref = fwd_bwd(0)
    for level in (2, 4):
        for rep in range(3):
            same(ref, fwd_bwd(level), 'eager level %%d' %% level)
    # gradient accumulation: the second pass ADDS to existing gradients -> the side stream must be joined before autograd adds
    twice = fwd_bwd(4, accumulate=True)
    for n in ref:
        e = float((twice[n] - 2 * ref[n]).abs().max()) / max(float(ref[n].abs().max()), 1e-12)
        if not e < 1e-5:
            print('MISMATCH accumulate', n, e)
            sys.exit(3)
    fwd_bwd(4)
torch.cuda.current_stream().wait_stream(s0)
torch.cuda.synchronize()
# replayed
g = torch.cuda.CUDAGraph()
for p in net.parameters():
    p.grad = None
with torch.cuda.graph(g):
    with backbone_ops.pairing(4):
        outs, gl = net(x, ['single'])
    l = gl
    for o, r in zip(outs, proj):
        l = l + (o * r).sum() * 1e-2
    l.backward()
held = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
for rep in range(3):
    g.replay()
    torch.cuda.synchronize()
    same(ref, held, 'replay')
print('OK', len(ref), 'gradients')
'''


def test_paired_backward_schedule_gives_the_serial_schedules_gradients():
    """SM3_PAIR_DGRAD / SM3_DEFER_JOIN (backbone_ops._paired): input-gradient GEMMs on the main stream, weight gradients on the
    side stream, joined once at the end of the pass -- the same kernels on the same operands: gradients bit-identical to
    the serial schedule (levels 2 and 4, eager x3 and replayed x3), and correct under gradient accumulation (the one case in
    which the join cannot be deferred)."""
    env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE='0')
    r = subprocess.run([sys.executable, '-c', PAIR_CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert 'OK' in r.stdout, r.stdout[-500:]
