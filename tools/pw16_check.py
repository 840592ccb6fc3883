import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rubiksnet_amd import _native
L = _native.lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
def run(Fr, K, M, H, W, res, bwd=False):
    P = H * W
    g = torch.Generator(device="cpu").manual_seed(K * 7 + M)
    x = torch.randn(Fr, K, P, generator=g).bfloat16().to(dev)
    r = torch.randn(Fr, M, P, generator=g).bfloat16().to(dev) if res else None
    w = (torch.randn(M, K, generator=g) / K ** 0.5).to(dev)        # the GEMM's A [M][K]
    nb = int(L.rk_pw_packed_bytes(M, K))
    pk = torch.empty(nb, dtype=torch.uint8, device=dev)
    if bwd:   # pack from W^T storage: weight [Cout=K][Cin=M], A = W^T
        wt = w.t().contiguous()
        _native.check(L.rk_pw_pack_bf16(wt.data_ptr(), K, M, None, pk.data_ptr(), st), "pack")
    else:
        _native.check(L.rk_pw_pack_bf16(w.data_ptr(), M, K, pk.data_ptr(), None, st), "pack")
    y = torch.full((Fr, M, P), float("nan"), device=dev, dtype=torch.bfloat16)
    _native.check(L.rk_pw_gemm_packed_bf16(pk.data_ptr(), x.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), Fr, K, M, P, st), "gemm")
    torch.cuda.synchronize()
    ref = torch.einsum("mk,fkp->fmp", w.bfloat16().double(), x.double())
    if res: ref = ref + r.double()
    err = (y.double() - ref).abs().max().item(); sc = ref.abs().max().item()
    print((Fr, K, M, H, W), "res" if res else "   ", "bwd" if bwd else "fwd", f"max err {err:.3e} scale {sc:.2f} rel {err/sc:.2e}", "OK" if err <= sc * 2 ** -7 else "FAIL", flush=True)
for shp in [(8, 288, 288, 14, 14), (5, 72, 72, 12, 12), (3, 144, 144, 28, 28), (2, 70, 50, 6, 10), (4, 288, 576, 14, 14), (3, 576, 288, 14, 14), (1, 32, 16, 2, 4), (3, 40, 24, 3, 4), (2, 64, 64, 4, 5), (7, 100, 330, 6, 6)]:
    for res in (False, True):
        run(*shp, res)
    run(*shp, False, True)

def wrun(Fr, K, M, H, W):
    P = H * W
    g = torch.Generator(device="cpu").manual_seed(K * 3 + M)
    x = torch.randn(Fr, K, P, generator=g).bfloat16().to(dev)
    dy = torch.randn(Fr, M, P, generator=g).bfloat16().to(dev)
    nb = int(L.rk_pw_wgrad16_workspace_bytes(Fr, K, M, P))
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    dw = torch.full((M, K), float("nan"), device=dev)
    _native.check(L.rk_pw_wgrad16_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Fr, K, M, P, ws.data_ptr(), nb, st), "wgrad16")
    torch.cuda.synchronize()
    ref = torch.einsum("fmp,fkp->mk", dy.double(), x.double())
    err = (dw.double() - ref).abs().max().item(); sc = ref.abs().max().item()
    print("wgrad", (Fr, K, M, H, W), f"max err {err:.3e} scale {sc:.2f} rel {err/sc:.2e}", "OK" if err <= sc * 1e-5 else "FAIL", flush=True)
for shp in [(8, 288, 288, 14, 14), (5, 72, 72, 12, 12), (3, 144, 144, 28, 28), (2, 70, 50, 6, 10), (4, 288, 576, 14, 14), (3, 576, 288, 14, 14),
            (1, 32, 16, 2, 4), (3, 40, 24, 3, 4), (2, 64, 64, 4, 5), (7, 100, 330, 6, 6), (256, 288, 288, 14, 14), (32, 72, 144, 56, 56)]:
    wrun(*shp)
