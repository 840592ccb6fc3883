"""Process-wide switches of the MI355X path, read ONCE (at import) into one frozen object.

    RK_FUSED_BN    1 | 0          relu(bn(x)) pairs through the fused HIP operator (fused_bn.py) / stock modules
    RK_PW          auto | 0 | all 1x1 convolutions on the HIP MFMA GEMM wherever it can run / never / (same as auto)
    RK_FUSED_EVAL  1 | 0          inference blocks with BN + residual folded into the two GEMMs / layer by layer
    RK_FUSED_TRAIN 1 | 0          training blocks as one autograd node with the BatchNorm statistics / normalisation / backward
                                  reduction folded into the 1x1 GEMMs (train_block.py), -aq blocks' bn1 + ReLU inside the
                                  temporal filter (fused_bn.bn_relu_tshift_skip) / layer by layer
    RK_WGRAD_OVERLAP 1 | 0        the d(weight) kernels of a fused training block on a second HIP stream, next to the
                                  streaming kernels of the same backward / on the current stream
    RK_BN_SHIFT2D  1 | 0          -aq training blocks: bn2 + ReLU inside the 2-D shift kernels (fused_bn.bn_relu_shift2d; 14 x 14
                                  planes) / bn2 + ReLU as their own passes
    RK_PW16_STATS  0 | 1          bf16 training: separate statistics pass / conv2's GEMM epilogue leaves the tile statistics
                                  of its output for bn2 (rk_pw_gemm_packed_stats_bf16).  Off by default: measured on the
                                  Large-AQ step the epilogue costs what the pass it replaces cost (33.1 vs 32.8 ms), DESIGN 3.5b

    RK_PRESOFT     1 | 0          train step: the [C, 3] tap softmax of every AttentionShift layer in one launch each way
                                  (attention_shift.presoftened) / one launch per layer and direction

Everything else that used to be tunable from the environment (tile shapes, channel limits, prefetch depths)
is a constant next to the code it tunes.  The native library has one switch of its own, RK_SHIFT_KERNELS
(include/rubiks_hip.h), and RK_PW2 = 1 | 0 | 2 (second-generation fp32 1x1 kernels where they are ahead /
never / wherever they can run).  Tests flip switches with `config.reload()` after changing os.environ.
"""
import dataclasses
import os

__all__ = ["Switches", "switches", "reload"]


@dataclasses.dataclass(frozen=True)
class Switches:
    fused_bn: bool = True
    pointwise: str = "auto"          # "auto" | "0" | "all"
    fused_eval: bool = True
    fused_train: bool = True
    wgrad_overlap: bool = True
    pw16_stats: bool = False
    bn_shift2d: bool = True
    prepack: bool = True
    bn_tshift_fork: bool = True
    presoft: bool = True

    @staticmethod
    def from_env(env=None):
        env = os.environ if env is None else env
        pw = env.get("RK_PW", "auto")
        if pw not in ("auto", "0", "all"):
            raise ValueError("RK_PW must be auto, 0 or all (got %r)" % pw)
        return Switches(fused_bn=env.get("RK_FUSED_BN", "1") != "0", pointwise=pw,
                        fused_eval=env.get("RK_FUSED_EVAL", "1") != "0",
                        fused_train=env.get("RK_FUSED_TRAIN", "1") != "0",
                        wgrad_overlap=env.get("RK_WGRAD_OVERLAP", "1") != "0",
                        pw16_stats=env.get("RK_PW16_STATS", "0") == "1",
                        bn_shift2d=env.get("RK_BN_SHIFT2D", "1") != "0",
                        prepack=env.get("RK_PREPACK", "1") != "0",
                        bn_tshift_fork=env.get("RK_BN_TSHIFT_FORK", "1") != "0",
                        presoft=env.get("RK_PRESOFT", "1") != "0")


_current = Switches.from_env()


def switches():
    return _current


def reload(env=None):
    """Re-read the environment (tests only: production code reads the switches once, at import)."""
    global _current
    _current = Switches.from_env(env)
    return _current
