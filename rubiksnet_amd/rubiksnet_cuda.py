"""Drop-in for the reference's pybind11 extension module `rubiksnet_cuda`
(cuda_src/rubiks.cpp:384-396), bound to librubiks_hip.so through ctypes.

Same six callables, same argument order / keyword names, same `return 0` convention:

    rubiks_shift_3d_forward_{float,double}(input, shift, strides, paddings, quantize, output)
    rubiks_shift_3d_backward_{float,double}(input, shift, output_grad, strides, paddings,
                                            input_grad, shift_grad, normalize_grad,
                                            normalize_t_factor, quantize)
    rubiks2d_forward(input, shift, strides, paddings, quantize, output)
    rubiks2d_backward(upstream_grad, input, shift, strides, paddings, normalize_grad,
                      enable_shift_grad, quantize, input_grad, shift_grad)

so `sys.modules["rubiksnet_cuda"] = rubiksnet_amd.rubiksnet_cuda` makes the reference's own
rubiksnet/shiftlib run on an MI355X unchanged.  Differences from the
reference binding, all deliberate: kernels go to PyTorch's CURRENT stream of the tensor's
device (the reference uses the legacy default stream of whatever device is current,
rubiks3d_kernels.cu:1002-1038); scratch comes from PyTorch's caching allocator, not
torch::zeros per call (rubiks.cpp:295-299); failures raise instead of exit()ing.
"""
import torch

from . import _native

__all__ = [
    "rubiks2d_forward",
    "rubiks2d_backward",
    "rubiks_shift_3d_forward_float",
    "rubiks_shift_3d_forward_double",
    "rubiks_shift_3d_backward_float",
    "rubiks_shift_3d_backward_double",
]


def _require(t, name, dtype=None):
    # cuda_src/utils.h:306-308 (TX_CHECK_TENSOR) and rubiks.cpp:216-222 (CHECK_CONTIGUOUS)
    if not torch.is_tensor(t):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (HIP) tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))


def _same_device(*ts):
    dev = ts[0].device
    for t in ts[1:]:
        if t.device != dev:
            raise RuntimeError("all tensors must live on one device (%s vs %s)" % (dev, t.device))
    return dev


def _stream_ptr(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _ints(v, n, what):
    v = [int(e) for e in v]
    if len(v) != n:
        raise ValueError("%s must have %d entries, got %r" % (what, n, v))
    return v


def _forward3d(dtype, sfx, input, shift, strides, paddings, quantize, output):
    _require(input, "input", dtype); _require(shift, "shift", dtype); _require(output, "output", dtype)
    dev = _same_device(input, shift, output)
    s, p = _ints(strides, 3, "strides"), _ints(paddings, 3, "paddings")
    if input.dim() != 5:
        raise RuntimeError("input must be [N,T,C,H,W]")
    N, T, C, H, W = input.shape
    if tuple(shift.shape) != (3, C):
        raise RuntimeError("shift must be [3, %d], got %s" % (C, tuple(shift.shape)))
    L = _native.lib()
    want = (N, L.rk_out_len(T, s[0], p[0]), C, L.rk_out_len(H, s[1], p[1]), L.rk_out_len(W, s[2], p[2]))
    if tuple(output.shape) != want:
        raise RuntimeError("output has shape %s, expected %s" % (tuple(output.shape), want))
    if input.numel() == 0 or output.numel() == 0:
        return 0          # empty batch: the reference launches over zero elements (rubiks3d_kernels.cu:34-36)
    with torch.cuda.device(dev):
        rc = getattr(L, "rk3d_forward_" + sfx)(
            input.data_ptr(), shift.data_ptr(), output.data_ptr(), N, T, C, H, W, *s, *p,
            int(bool(quantize)), _stream_ptr(dev))
    _native.check(rc, "rk3d_forward_" + sfx)
    return 0


def _backward3d(dtype, sfx, input, shift, output_grad, strides, paddings, input_grad, shift_grad,
                normalize_grad, normalize_t_factor, quantize):
    _require(input, "input", dtype); _require(shift, "shift", dtype)
    _require(output_grad, "output_grad", dtype)
    if input_grad is not None:
        _require(input_grad, "input_grad", dtype)
    if shift_grad is not None:
        _require(shift_grad, "shift_grad", dtype)
    dev = _same_device(input, shift, output_grad)
    s, p = _ints(strides, 3, "strides"), _ints(paddings, 3, "paddings")
    N, T, C, H, W = input.shape
    L = _native.lib()
    want = (N, L.rk_out_len(T, s[0], p[0]), C, L.rk_out_len(H, s[1], p[1]), L.rk_out_len(W, s[2], p[2]))
    if tuple(output_grad.shape) != want:
        raise RuntimeError("output_grad has shape %s, expected %s" % (tuple(output_grad.shape), want))
    if input_grad is not None and input_grad.shape != input.shape:
        raise RuntimeError("input_grad must have the shape of input")
    if shift_grad is not None and tuple(shift_grad.shape) != (3, C):
        raise RuntimeError("shift_grad must be [3, C]")
    if input.numel() == 0 or output_grad.numel() == 0:
        if shift_grad is not None:
            shift_grad.zero_()   # addmv_ over an empty scratch gives zeros (rubiks.cpp:344-345)
        return 0
    with torch.cuda.device(dev):
        ws_bytes = L.rk3d_backward_workspace_bytes(N, T, C, H, W, *s, *p, input.element_size())
        ws = torch.empty(max(int(ws_bytes), 1), dtype=torch.uint8, device=dev) if shift_grad is not None else None
        rc = getattr(L, "rk3d_backward_" + sfx)(
            input.data_ptr(), shift.data_ptr(), output_grad.data_ptr(),
            input_grad.data_ptr() if input_grad is not None else None,
            shift_grad.data_ptr() if shift_grad is not None else None,
            N, T, C, H, W, *s, *p, int(bool(normalize_grad)), float(normalize_t_factor),
            int(bool(quantize)), ws.data_ptr() if ws is not None else None, int(ws_bytes), _stream_ptr(dev))
    _native.check(rc, "rk3d_backward_" + sfx)
    return 0


def rubiks_shift_3d_forward_float(input, shift, strides, paddings, quantize, output):
    """cuda_src/rubiks.cpp:181-253 at T=float."""
    return _forward3d(torch.float32, "f32", input, shift, strides, paddings, quantize, output)


def rubiks_shift_3d_forward_double(input, shift, strides, paddings, quantize, output):
    """cuda_src/rubiks.cpp:181-253 at T=double."""
    return _forward3d(torch.float64, "f64", input, shift, strides, paddings, quantize, output)


def rubiks_shift_3d_backward_float(input, shift, output_grad, strides, paddings, input_grad, shift_grad,
                                   normalize_grad, normalize_t_factor, quantize):
    """cuda_src/rubiks.cpp:256-379 at T=float.  input_grad / shift_grad may be None to skip that half."""
    return _backward3d(torch.float32, "f32", input, shift, output_grad, strides, paddings, input_grad,
                       shift_grad, normalize_grad, normalize_t_factor, quantize)


def rubiks_shift_3d_backward_double(input, shift, output_grad, strides, paddings, input_grad, shift_grad,
                                    normalize_grad, normalize_t_factor, quantize):
    """cuda_src/rubiks.cpp:256-379 at T=double."""
    return _backward3d(torch.float64, "f64", input, shift, output_grad, strides, paddings, input_grad,
                       shift_grad, normalize_grad, normalize_t_factor, quantize)


def _sfx2d(t, shift):
    """Entry-point suffix and the dtype the shift table / d(shift) must have.  The reference runs K6-K9 at the
    tensor's own scalar type, shift included (rubiks2d_kernels.cu:113-114); next to 16-bit activations an fp32
    shift is accepted as well and then stays fp32 inside the kernels (rk2d_*_sf32)."""
    sfx = _native.dtype_suffix(t.dtype)
    if sfx is None:
        # AT_DISPATCH_FLOATING_TYPES_AND_HALF (rubiks2d_kernels.cu:422) raises for other dtypes
        raise RuntimeError("rubiks2d not implemented for dtype %s" % t.dtype)
    if torch.is_tensor(shift) and shift.dtype == torch.float32 and t.dtype in (torch.float16, torch.bfloat16):
        return sfx + "_sf32", torch.float32
    return sfx, t.dtype


def rubiks2d_forward(input, shift, strides, paddings, quantize, output):
    """cuda_src/rubiks.cpp:44-67.  With quantize, out-of-range outputs are left untouched
    (rubiks2d_kernels.cu:116-121): pass a zero-filled `output`, as rubiksnet/utils.py:26 does."""
    _require(input, "input")
    sfx, shift_dtype = _sfx2d(input, shift)
    _require(shift, "shift", shift_dtype); _require(output, "output", input.dtype)
    dev = _same_device(input, shift, output)
    s, p = _ints(strides, 2, "strides"), _ints(paddings, 2, "paddings")
    N, C, H, W = input.shape
    if tuple(shift.shape) != (2, C):                      # rubiks.cpp:61-63 (ShapeException)
        raise RuntimeError("rubiks shift: expected shape (2, %d), got %s" % (C, tuple(shift.shape)))
    L = _native.lib()
    want = (N, C, L.rk_out_len(H, s[0], p[0]), L.rk_out_len(W, s[1], p[1]))
    if tuple(output.shape) != want:
        raise RuntimeError("output has shape %s, expected %s" % (tuple(output.shape), want))
    if input.numel() == 0 or output.numel() == 0:
        return 0
    with torch.cuda.device(dev):
        rc = getattr(L, "rk2d_forward_" + sfx)(
            input.data_ptr(), shift.data_ptr(), output.data_ptr(), N, C, H, W, *s, *p,
            int(bool(quantize)), _stream_ptr(dev))
    _native.check(rc, "rk2d_forward_" + sfx)
    return 0


def rubiks2d_backward(upstream_grad, input, shift, strides, paddings, normalize_grad, enable_shift_grad,
                      quantize, input_grad, shift_grad):
    """cuda_src/rubiks.cpp:94-155.  input_grad must be zero-filled when quantize is set
    (rubiks.cpp:106-107, rubiks2d_kernels.cu:294-309)."""
    _require(upstream_grad, "upstream_grad", input.dtype); _require(input, "input")
    sfx, shift_dtype = _sfx2d(input, shift)
    _require(shift, "shift", shift_dtype); _require(input_grad, "input_grad", input.dtype)
    _require(shift_grad, "shift_grad", shift_dtype)
    dev = _same_device(upstream_grad, input, shift, input_grad, shift_grad)
    s, p = _ints(strides, 2, "strides"), _ints(paddings, 2, "paddings")
    N, C, H, W = input.shape
    L = _native.lib()
    want = (N, C, L.rk_out_len(H, s[0], p[0]), L.rk_out_len(W, s[1], p[1]))
    if tuple(upstream_grad.shape) != want:
        raise RuntimeError("upstream_grad has shape %s, expected %s" % (tuple(upstream_grad.shape), want))
    if input.numel() == 0 or upstream_grad.numel() == 0:
        if enable_shift_grad:
            shift_grad.zero_()
        return 0
    with torch.cuda.device(dev):
        ws_bytes = L.rk2d_backward_workspace_bytes(N, C, H, W, *s, *p, input.element_size())
        ws = torch.empty(max(int(ws_bytes), 1), dtype=torch.uint8, device=dev)
        rc = getattr(L, "rk2d_backward_" + sfx)(
            upstream_grad.data_ptr(), input.data_ptr(), shift.data_ptr(), input_grad.data_ptr(),
            shift_grad.data_ptr(), N, C, H, W, *s, *p, int(bool(normalize_grad)),
            int(bool(enable_shift_grad)), int(bool(quantize)), ws.data_ptr(), int(ws_bytes), _stream_ptr(dev))
    _native.check(rc, "rk2d_backward_" + sfx)
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# Training fusion (train_block.py): the 3-D shift applied to relu(bn(z)) without the activation being stored.  No
# counterpart in the reference's extension module (its block runs BatchNorm, ReLU and the shift as separate layers,
# rubiksnet/backbone.py:131-132); same calling style as the six callables above.  Both return UNSUPPORTED -- nothing
# launched -- when no fused kernel covers the configuration.
UNSUPPORTED = -7


def rubiks_shift_3d_forward_bn_float(input, abmi, shift, strides, paddings, quantize, output):
    """output = RubiksShift3D(relu(a * input + b)); abmi [C, 4] = (a, b, mean, invstd) per channel."""
    _require(input, "input", torch.float32); _require(shift, "shift", torch.float32)
    _require(output, "output", torch.float32); _require(abmi, "abmi", torch.float32)
    dev = _same_device(input, shift, output, abmi)
    s, p = _ints(strides, 3, "strides"), _ints(paddings, 3, "paddings")
    N, T, C, H, W = input.shape
    if tuple(shift.shape) != (3, C) or tuple(abmi.shape) != (C, 4):
        raise RuntimeError("shift must be [3, C] and abmi [C, 4]")
    L = _native.lib()
    want = (N, L.rk_out_len(T, s[0], p[0]), C, L.rk_out_len(H, s[1], p[1]), L.rk_out_len(W, s[2], p[2]))
    if tuple(output.shape) != want:
        raise RuntimeError("output has shape %s, expected %s" % (tuple(output.shape), want))
    if input.numel() == 0:
        return UNSUPPORTED
    with torch.cuda.device(dev):
        rc = L.rk3d_forward_bn_f32(input.data_ptr(), abmi.data_ptr(), shift.data_ptr(), output.data_ptr(), N, T, C, H, W,
                                   *s, *p, int(bool(quantize)), _stream_ptr(dev))
    if rc == UNSUPPORTED:
        return rc
    _native.check(rc, "rk3d_forward_bn_f32")
    return 0


def rubiks_shift_3d_backward_bn_float(input, abmi, shift, output_grad, strides, paddings, input_grad, shift_grad, k12,
                                      dgamma, dbeta, normalize_grad, normalize_t_factor, quantize):
    """Backward of the above: input_grad = d(relu(bn(input))) masked by the ReLU, shift_grad as the plain operator, and
    the BatchNorm backward's constants k12 [2, C] = (sum dz, sum dz zhat) / count, dgamma, dbeta [C]."""
    for t, name in ((input, "input"), (abmi, "abmi"), (shift, "shift"), (output_grad, "output_grad"),
                    (input_grad, "input_grad"), (shift_grad, "shift_grad"), (k12, "k12"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _require(t, name, torch.float32)
    dev = _same_device(input, abmi, shift, output_grad, input_grad, shift_grad, k12, dgamma, dbeta)
    s, p = _ints(strides, 3, "strides"), _ints(paddings, 3, "paddings")
    N, T, C, H, W = input.shape
    L = _native.lib()
    want = (N, L.rk_out_len(T, s[0], p[0]), C, L.rk_out_len(H, s[1], p[1]), L.rk_out_len(W, s[2], p[2]))
    if tuple(output_grad.shape) != want or input_grad.shape != input.shape:
        raise RuntimeError("output_grad / input_grad have the wrong shape")
    if tuple(shift_grad.shape) != (3, C) or tuple(k12.shape) != (2, C) or dgamma.numel() != C or dbeta.numel() != C:
        raise RuntimeError("shift_grad [3, C], k12 [2, C], dgamma / dbeta [C] expected")
    if input.numel() == 0:
        return UNSUPPORTED
    with torch.cuda.device(dev):
        ws_bytes = int(L.rk3d_backward_bn_workspace_bytes(N, T, C, H, W, *s, *p))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        rc = L.rk3d_backward_bn_f32(input.data_ptr(), abmi.data_ptr(), shift.data_ptr(), output_grad.data_ptr(),
                                    input_grad.data_ptr(), shift_grad.data_ptr(), k12.data_ptr(), dgamma.data_ptr(),
                                    dbeta.data_ptr(), N, T, C, H, W, *s, *p, int(bool(normalize_grad)),
                                    float(normalize_t_factor), int(bool(quantize)), ws.data_ptr(), ws_bytes, _stream_ptr(dev))
    if rc == UNSUPPORTED:
        return rc
    _native.check(rc, "rk3d_backward_bn_f32")
    return 0
