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
