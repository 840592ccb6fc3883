// rk3d_stream.hpp -- LDS-tiled streaming kernels for RubiksShift3D (placeholder: the
// generic kernels serve every shape until the streaming path lands).
#pragma once
#include "rk3d_generic.hpp"

namespace rk {
namespace stream3d {

template <typename T> bool forward_supported(const Dims3&, int) { return false; }
template <typename T> bool backward_supported(const Dims3&, int) { return false; }

template <typename T>
int launch_forward(const T*, const T*, T*, const Dims3&, hipStream_t) { return RK_ERR_LAUNCH; }

template <typename T>
int launch_backward(const T*, const T*, const T*, T*, T*, const Dims3&, int, T, T*, hipStream_t) {
    return RK_ERR_LAUNCH;
}

}  // namespace stream3d
}  // namespace rk
