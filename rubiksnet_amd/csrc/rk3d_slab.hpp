// rk3d_slab.hpp -- host entry points of the small-plane RubiksShift3D kernels (rk3d_slab.hip) for rk3d.hip's dispatch.
#pragma once
#include "rk3d_generic.hpp"

namespace rk {
namespace slab3d {

// 14x14 planes here instead of rk3d_tile.hpp (forward by default; RK_SLAB14 = 1 / 0: both / neither, see rk3d_slab.hip)
bool slab14_on(bool backward);
// forward (negate = false: src = x, dst = y) / d(x) alone (negate = true: src = gy, dst = gx); false = not handled here
bool launch_interp(bool negate, const float* src, const float* shift, float* dst, const Dims3& d, hipStream_t stream);
// d(shift) (+ d(x) when gx != nullptr); gshift != nullptr: row-sum + K5 inside the launch (ws = granule pairs), else plain
// partials ws[C][3][P].  Returns P (0 = not handled here)
int launch_bwd(const float* x, const float* shift, const float* gy, float* gx, float* gshift, float* ws, const Dims3& d,
               int normalize, float t_factor, hipStream_t stream);

// the same for stride (1,2,2) / pad 0 on even planes up to 56 wide (the layers rk3d_stride2.hpp does not take: 28 -> 14, 14 -> 7)
bool launch_fwd_s2(const float* x, const float* shift, float* y, const Dims3& d, hipStream_t stream);
int launch_bwd_s2(const float* x, const float* shift, const float* gy, float* gx, float* gshift, float* ws, const Dims3& d,
                  int normalize, float t_factor, hipStream_t stream);

}  // namespace slab3d
}  // namespace rk
