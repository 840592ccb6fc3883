"""Small host helpers shared by the shift operators (counterpart of rubiksnet/utils.py:4-45)."""
import torch

__all__ = ["make_tuple", "allocate_output"]


def make_tuple(elem, repeats):
    """3 -> [3, 3]; a sequence is validated and int-cast (rubiksnet/utils.py:4-12)."""
    if isinstance(elem, int):
        return [elem] * repeats
    assert len(elem) == repeats
    return [int(x) for x in elem]


def allocate_output(output, tensor_like, desired_shape, zero=True):
    """Return the buffer an operator writes into (rubiksnet/utils.py:15-45).

    `output is None` -> a fresh tensor with `tensor_like`'s dtype/device; otherwise the
    caller's tensor is validated (shape, dtype, device) and used as is.  The reference always
    zero-fills fresh buffers; callers here pass zero=False when the kernel is known to write
    every element (all 3D kernels, 2D without quantize), saving one full HBM write pass.
    """
    desired_shape = torch.Size(desired_shape)
    if output is None:
        if zero:
            return tensor_like.new_zeros(desired_shape)
        return tensor_like.new_empty(desired_shape)
    assert torch.is_tensor(output)
    assert output.size() == desired_shape, "output tensor has wrong shape {}, which should be {}".format(
        output.size(), desired_shape)
    assert output.dtype == tensor_like.dtype, "output tensor has wrong dtype {}, which should be {}".format(
        output.dtype, tensor_like.dtype)
    assert output.device == tensor_like.device, "output tensor has wrong device {}, which should be {}".format(
        output.device, tensor_like.device)
    return output
