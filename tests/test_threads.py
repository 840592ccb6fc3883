"""Thread safety of the C-ABI library (SURVEY 8b "Threading": the reference's multi-GPU mechanism is single-process
nn.DataParallel, scripts/test_models.py:153 -- one forward thread per replica and one autograd thread per device, all inside
one process and one copy of this library).  The host side keeps per-device atomics only (rk_common.hpp: device_cus,
raise_dynamic_lds; rk_dma.hpp: launch tags): the CPU half hammers the entry points that read them from many threads, the
GPU half runs the kernels that need the raised dynamic-LDS ceiling from two threads at once and wraps the model in
nn.DataParallel."""
import threading

import pytest
import torch


def _hammer(fn, threads=8, rounds=400):
    out, errs = [[] for _ in range(threads)], []

    def work(i):
        try:
            for _ in range(rounds):
                out[i].append(fn())
        except Exception as e:     # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    return out


def test_planners_and_hooks_are_reentrant_from_many_threads():
    """No GPU needed: the workspace planners read the cached CU count (per-device atomic table), the debug hooks read and
    write the launch-tag counter and the poll budget -- eight threads at once, every answer equal to the single-threaded
    one (ctypes releases the GIL around each call, so the calls really overlap)."""
    from rubiksnet_amd import _native
    L = _native.lib()
    want = (L.rk_pw2_wgrad_workspace_bytes(256, 288, 288, 196), L.rk_pw_wgrad_workspace_bytes(256, 288, 288, 196),
            L.rk3d_backward_workspace_bytes(32, 8, 64, 56, 56, 1, 1, 1, 0, 0, 0, 4), L.rk_out_len(56, 2, 0))

    def once():
        L.rk_debug_peek_launch_tag()
        return (L.rk_pw2_wgrad_workspace_bytes(256, 288, 288, 196), L.rk_pw_wgrad_workspace_bytes(256, 288, 288, 196),
                L.rk3d_backward_workspace_bytes(32, 8, 64, 56, 56, 1, 1, 1, 0, 0, 0, 4), L.rk_out_len(56, 2, 0))

    for per_thread in _hammer(once):
        assert all(r == want for r in per_thread)
    default = L.rk_debug_set_finalize_spins(0)
    seen = _hammer(lambda: L.rk_debug_set_finalize_spins(12345), threads=4, rounds=50)
    assert {v for per in seen for v in per} <= {default, 12345}
    assert L.rk_debug_set_finalize_spins(0) == 12345 and L.rk_debug_set_finalize_spins(0) == default


@pytest.mark.gpu
def test_large_lds_kernels_launched_from_two_threads_at_once():
    """The kernels that take more than 64 KB of dynamic LDS (rk_pw3 / rk_pw4 / rk_pw2 wgrad / rk_pw16) raise the function
    attribute on first use -- per device and atomically.  Two threads, each on its own stream, run a 1x1 convolution
    forward + backward of the shapes that take those kernels, concurrently and repeatedly: every result bit-equal to the
    single-threaded one."""
    from rubiksnet_amd import pointwise

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cases = []
    for cin, cout, hw, dtype in [(288, 288, 14, torch.float32), (54, 54, 56, torch.float32), (288, 288, 14, torch.bfloat16),
                                 (144, 144, 28, torch.float32), (576, 576, 7, torch.bfloat16)]:
        conv = torch.nn.Conv2d(cin, cout, 1, bias=False).to(dev)
        x = torch.randn(64, cin, hw, hw, device=dev).to(dtype)
        gy = torch.randn(64, cout, hw, hw, device=dev).to(dtype)
        cases.append((conv, x, gy))

    def run(conv, x, gy):
        xin = x.clone().requires_grad_(True)
        w = conv.weight
        w.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=x.dtype == torch.bfloat16):
            y = pointwise.conv1x1(conv, xin)
        gx, gw = torch.autograd.grad(y, (xin, w), gy)
        return y.detach(), gx, gw

    ref = [run(*c) for c in cases]
    torch.cuda.synchronize()
    errs = []

    def work(order):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                for _ in range(6):
                    for i in order:
                        got = run(*cases[i])
                        torch.cuda.current_stream().synchronize()
                        for a, b in zip(got, ref[i]):
                            assert torch.equal(a, b), "case %d differs when launched from two threads" % i
        except Exception as e:     # noqa: BLE001
            errs.append(e)

    # (each thread owns its convs' .grad-free path: autograd.grad returns the gradients, nothing is accumulated in place)
    ts = [threading.Thread(target=work, args=(o,)) for o in ([0, 1, 2, 3, 4], [4, 3, 2, 1, 0])]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["rubiks3d", "rubiks3d-aq"])
def test_data_parallel_wrapper_gives_the_bare_models_logits(variant):
    """The reference's own multi-GPU mechanism (scripts/test_models.py:153: nn.DataParallel around the network, eval
    mode) on the devices this box has: replicas are fresh module objects on every forward and run on worker threads when
    there is more than one device -- logits bit-equal to the bare model's, forward after forward; and a training
    forward + backward through the wrapper leaves the bare model's gradients."""
    from rubiksnet_amd import RubiksNet

    torch.manual_seed(1)
    net = RubiksNet("tiny", num_classes=23, num_frames=8, variant=variant, verbose=False).to("cuda:0").eval()
    ids = list(range(torch.cuda.device_count()))
    wrapped = torch.nn.DataParallel(net, device_ids=ids)
    clips = torch.randn(2 * len(ids) + 2, 8, 3, 224, 224, device="cuda:0")
    with torch.no_grad():
        want = net(clips)
        for _ in range(3):
            got = wrapped(clips)
            if len(ids) == 1:
                assert torch.equal(got, want)
            else:   # the scatter changes the per-launch batch: the GEMMs' split of the pixel axis may differ in the last bits
                torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
    # training through the wrapper (one replica per device, BN statistics per replica as under the reference)
    net.train()
    labels = torch.randint(0, 23, (clips.shape[0],), device="cuda:0")
    if len(ids) == 1:
        import copy
        twin = copy.deepcopy(net)
        torch.nn.functional.cross_entropy(torch.nn.DataParallel(net, device_ids=ids)(clips), labels).backward()
        torch.nn.functional.cross_entropy(twin(clips), labels).backward()
        for (n, p), q in zip(net.named_parameters(), twin.parameters()):
            assert (p.grad is None) == (q.grad is None), n
            if p.grad is not None:
                assert torch.equal(p.grad, q.grad), n
