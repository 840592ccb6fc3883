"""Whole train step (fwd + bwd + Adam) captured in a hipGraph vs eager: python tools/graph_probe.py [tier] [variant] [amp]"""
import sys, time, torch
sys.path.insert(0, ".")
from rubiksnet_amd import RubiksNet, dp
tier = sys.argv[1] if len(sys.argv) > 1 else "large"
variant = sys.argv[2] if len(sys.argv) > 2 else "rubiks3d"
amp = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else None
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = RubiksNet(tier, 174, variant=variant, verbose=False).to(dev).train()
params = [p for p in net.parameters()]
opt = torch.optim.Adam(params, lr=1e-3, capturable=True, fused=True)
clips = torch.randn(32, 8, 3, 224, 224, device=dev); labels = torch.randint(0, 174, (32,), device=dev)
def step():
    with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
        opt.zero_grad(set_to_none=False)
        loss = torch.nn.functional.cross_entropy(net(clips), labels)
    loss.backward()
    opt.step()
    return loss
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(20): step()
print("eager: %.2f ms/step" % timeit(step))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = step()
    print("graph: %.2f ms/step" % timeit(g.replay))
    l0 = float(loss); g.replay(); torch.cuda.synchronize(); print("loss after replays", l0, float(loss))
except Exception as e:
    print("capture failed:", repr(e)[:600])
