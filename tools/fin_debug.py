"""debug: where do the NaNs of the fused finalizers come from?"""
import numpy as np, torch, sys
sys.path.insert(0, '.')
from rubiksnet_amd import _native, rubiksnet_cuda as rc
L = _native.lib()
dev = 'cuda:0'
print("spins default", L.rk_debug_set_finalize_spins(0))
C, P = 5, 70
rng = np.random.default_rng(0)
vals = rng.uniform(-1, 1, (C, 3, P)).astype(np.float32)
tag = int(L.rk_debug_peek_launch_tag())
tag2 = (tag * 2654435761 ^ 0x9e3779b9) & 0xffffffff
bits = vals.view(np.uint32)
gran = np.empty((C, 3, P, 4), dtype=np.uint32)
gran[..., 0] = bits; gran[..., 1] = tag; gran[..., 2] = ~bits; gran[..., 3] = tag2
ws = torch.from_numpy(gran.reshape(-1).view(np.uint8)).to(dev)
gs = torch.zeros(3, C, device=dev)
print("rc", L.rk3d_debug_finalize_only_f32(ws.data_ptr(), ws.numel(), C, P, gs.data_ptr(), 1, 1.0, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("finalize-only on valid pairs:", gs)
x = torch.rand(2, 8, 16, 56, 56, device=dev) * 2 - 1
gy = torch.rand_like(x); shift = torch.rand(3, 16, device=dev) * 2 - 1
gx = torch.empty_like(x); g = torch.empty_like(shift)
rc.rubiks_shift_3d_backward_float(x, shift, gy, [1, 1, 1], [0, 0, 0], gx, g, True, 1.0, False)
torch.cuda.synchronize()
print("fused backward d(shift):", g[:, :4], "gx finite", torch.isfinite(gx).all().item())
# raw call with my own workspace
N, T, Cc, H, W = x.shape
nb = L.rk3d_backward_workspace_bytes(N, T, Cc, H, W, 1, 1, 1, 0, 0, 0, 4)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
tag = int(L.rk_debug_peek_launch_tag())
print("ws bytes", nb, "tag %08x" % tag, "tag2 %08x" % ((tag * 2654435761 ^ 0x9e3779b9) & 0xffffffff))
r = L.rk3d_backward_f32(x.data_ptr(), shift.data_ptr(), gy.data_ptr(), gx.data_ptr(), g.data_ptr(), N, T, Cc, H, W, 1, 1, 1, 0, 0, 0, 1, 1.0, 0,
                        ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
w32 = ws.view(torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 4)
nz = np.nonzero(w32.any(axis=1))[0]
print("rc", r, "g", g[:, :3], "nonzero pairs", len(nz), "of", len(w32))
for i in nz[:6]:
    print(i, ["%08x" % v for v in w32[i]])
import time
for spins in (0, 1000, 100000):
    L.rk_debug_set_finalize_spins(spins)
    ws.zero_(); torch.cuda.synchronize(); t0 = time.time()
    r = L.rk3d_backward_f32(x.data_ptr(), shift.data_ptr(), gy.data_ptr(), gx.data_ptr(), g.data_ptr(), N, T, Cc, H, W, 1, 1, 1, 0, 0, 0, 1, 1.0, 0,
                            ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("spins", spins, "elapsed %.4f s" % (time.time() - t0), "nan", torch.isnan(g).sum().item())
L.rk_debug_set_finalize_spins(0)
# two-phase
import ctypes
pc = ctypes.c_int(0)
r = L.rk3d_backward_partials_f32(x.data_ptr(), shift.data_ptr(), gy.data_ptr(), gx.data_ptr(), N, T, Cc, H, W, 1, 1, 1, 0, 0, 0, 0, ws.data_ptr(), nb, ctypes.byref(pc), torch.cuda.current_stream().cuda_stream)
r2 = L.rk3d_backward_finalize_f32(ws.data_ptr(), Cc, pc.value, g.data_ptr(), 1, 1.0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("two-phase", r, r2, pc.value, g[:, :3])
L.rk_debug_set_finalize_spins(1)
ws.zero_(); torch.cuda.synchronize()
tag = int(L.rk_debug_peek_launch_tag())
r = L.rk3d_backward_f32(x.data_ptr(), shift.data_ptr(), gy.data_ptr(), gx.data_ptr(), g.data_ptr(), N, T, Cc, H, W, 1, 1, 1, 0, 0, 0, 1, 1.0, 0,
                        ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
L.rk_debug_set_finalize_spins(0)
w32 = ws.view(torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 4)
nz = np.nonzero(w32.any(axis=1))[0]
print("spins=1: tag %08x tag2 %08x nonzero pairs" % (tag, (tag * 2654435761 ^ 0x9e3779b9) & 0xffffffff), len(nz), "of", len(w32), "expected", Cc * 3 * 4)
for i in nz[:8]:
    print(i, ["%08x" % v for v in w32[i]], np.array([w32[i][0]], dtype=np.uint32).view(np.float32))
