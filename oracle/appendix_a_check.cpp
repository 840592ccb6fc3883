// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
//
// Re-runs the experiment SURVEY.md Appendix A recorded from the reference's own
// device code (cuda_src/rubiks3d_kernels.cu K1, K2, K4 at <float>):
//   std::mt19937(0); uniform_real(-1,1) fills x, shift, gy in that order;
//   [N=2,T=8,C=16,H=14,W=14], stride 1, pad 0, quantize off
//   -> sum(y) = -151.035112, sum(scratch) = 23.565896, sum(gx) = -154.699262
// and prints the same three sums from this repo's oracle, one per line, for each
// plausible reading of "uniform_real" (float / double distribution).
// tests/test_oracle_pins.py asserts one reading reproduces all three numbers.
#include <cstdio>
#include <random>
#include <vector>

extern "C" {
void oracle_rk3d_forward_f32(const float*, const float*, float*, int, int, int, int, int, int, int, int,
                             int, int, int, int, int, int, int);
void oracle_rk3d_backward_shift_partials_f32(const float*, const float*, const float*, float*, int, int,
                                             int, int, int, int, int, int, int, int, int, int, int, int);
void oracle_rk3d_backward_input_f32(const float*, const float*, float*, int, int, int, int, int, int,
                                    int, int, int, int, int, int, int, int, int);
}

template <class Dist>
static void run(const char* tag) {
    const int N = 2, T = 8, C = 16, H = 14, W = 14;
    const size_t n = (size_t)N * T * C * H * W;
    std::mt19937 gen(0);
    Dist d(-1, 1);
    std::vector<float> x(n), sh(3 * C), gy(n), y(n), gx(n), scratch(3 * (size_t)C * H * W, 0.f);
    for (auto& v : x) v = (float)d(gen);
    for (auto& v : sh) v = (float)d(gen);
    for (auto& v : gy) v = (float)d(gen);
    oracle_rk3d_forward_f32(x.data(), sh.data(), y.data(), N, T, C, H, W, T, H, W, 1, 1, 1, 0, 0, 0, 0);
    oracle_rk3d_backward_shift_partials_f32(x.data(), sh.data(), gy.data(), scratch.data(), N, T, C, H, W,
                                            T, H, W, 1, 1, 1, 0, 0, 0);
    oracle_rk3d_backward_input_f32(sh.data(), gy.data(), gx.data(), N, T, C, H, W, T, H, W, 1, 1, 1, 0, 0,
                                   0, 0);
    double sy = 0, ss = 0, sg = 0;
    for (float v : y) sy += v;
    for (float v : scratch) ss += v;
    for (float v : gx) sg += v;
    std::printf("%s %.6f %.6f %.6f\n", tag, sy, ss, sg);
}

int main() {
    run<std::uniform_real_distribution<float>>("float");
    run<std::uniform_real_distribution<double>>("double");
    return 0;
}
