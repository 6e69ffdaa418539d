"""Model geometry and the flat parameter layout (state_dict order).

The parameter inventory and its order are the compatibility contract with checkpoints of
the reference (QPNet.__init__, reference src/nets/qpnet.py:174-237; key list in
SURVEY.md §8a row a6).  The flat fp32 vector handed to the C ABI (`qpn_set_weights`) is the
concatenation of all tensors in exactly this order, each in its PyTorch-native layout.
"""
from dataclasses import dataclass, asdict
from typing import List, Tuple


@dataclass(frozen=True)
class QPNetConfig:
    n_quantize: int = 256
    n_aux: int = 39
    n_resch: int = 512
    n_skipch: int = 256
    dilationF_depth: int = 4
    dilationF_repeat: int = 3
    dilationA_depth: int = 4
    dilationA_repeat: int = 1
    kernel_size: int = 2
    upsampling_factor: int = 110

    # ---- derived (reference qpnet.py:186-199)
    @property
    def dilationsF(self) -> List[int]:
        return [2 ** i for i in range(self.dilationF_depth)] * self.dilationF_repeat

    @property
    def dilationsA(self) -> List[int]:
        return [2 ** i for i in range(self.dilationA_depth)] * self.dilationA_repeat

    @property
    def receptiveCausal_field(self) -> int:
        return self.kernel_size - 1

    @property
    def receptiveF_field(self) -> int:
        return (self.kernel_size - 1) * sum(self.dilationsF)

    @property
    def receptiveA_field(self) -> int:
        return (self.kernel_size - 1) * sum(self.dilationsA)

    def receptive_field(self, maxd: int) -> int:
        """qpnet.py:254-261 / 351-355: RF = RF_A*ceil(max d) + RF_F + RF_causal."""
        return self.receptiveA_field * int(maxd) + self.receptiveF_field + self.receptiveCausal_field

    def as_tuple(self) -> Tuple[int, ...]:
        return (self.n_quantize, self.n_aux, self.n_resch, self.n_skipch,
                self.dilationF_depth, self.dilationF_repeat,
                self.dilationA_depth, self.dilationA_repeat,
                self.kernel_size, self.upsampling_factor)

    def kwargs(self) -> dict:
        return asdict(self)

    # ---- flat parameter layout
    def param_layout(self) -> List[Tuple[str, Tuple[int, ...]]]:
        C, S, Q, A, K = self.n_resch, self.n_skipch, self.n_quantize, self.n_aux, self.kernel_size
        LF, LA = len(self.dilationsF), len(self.dilationsA)
        out: List[Tuple[str, Tuple[int, ...]]] = []
        out += [("causal.conv.weight", (C, Q, K)), ("causal.conv.bias", (C,))]
        if self.upsampling_factor > 0:
            out += [("upsampling.conv.weight", (1, 1, 1, self.upsampling_factor)),
                    ("upsampling.conv.bias", (1,))]
        for name in ("dilF_sigmoid", "dilF_tanh"):
            for i in range(LF):
                out += [(f"{name}.{i}.conv.weight", (C, C, K)), (f"{name}.{i}.conv.bias", (C,))]
        for name in ("auxF_1x1_sigmoid", "auxF_1x1_tanh"):
            for i in range(LF):
                out += [(f"{name}.{i}.weight", (C, A, 1)), (f"{name}.{i}.bias", (C,))]
        for i in range(LF):
            out += [(f"skipF_1x1.{i}.weight", (S, C, 1)), (f"skipF_1x1.{i}.bias", (S,))]
        for i in range(LF):
            out += [(f"resF_1x1.{i}.weight", (C, C, 1)), (f"resF_1x1.{i}.bias", (C,))]
        for name in ("dilA_sigmoid", "dilA_tanh"):
            for i in range(LA):
                out += [(f"{name}.{i}.convC.weight", (C, C, 1)), (f"{name}.{i}.convC.bias", (C,)),
                        (f"{name}.{i}.convP.weight", (C, C, 1)), (f"{name}.{i}.convP.bias", (C,))]
        for name in ("auxA_1x1_sigmoid", "auxA_1x1_tanh"):
            for i in range(LA):
                out += [(f"{name}.{i}.weight", (C, A, 1)), (f"{name}.{i}.bias", (C,))]
        for i in range(LA):
            out += [(f"skipA_1x1.{i}.weight", (S, C, 1)), (f"skipA_1x1.{i}.bias", (S,))]
        for i in range(LA):
            out += [(f"resA_1x1.{i}.weight", (C, C, 1)), (f"resA_1x1.{i}.bias", (C,))]
        out += [("conv_post_1.weight", (S, S, 1)), ("conv_post_1.bias", (S,)),
                ("conv_post_2.weight", (Q, S, 1)), ("conv_post_2.bias", (Q,))]
        return out

    def param_offsets(self):
        """{key: (offset, shape)} into the flat vector, plus total size."""
        offs, o = {}, 0
        for k, shp in self.param_layout():
            n = 1
            for s in shp:
                n *= s
            offs[k] = (o, shp)
            o += n
        return offs, o

    @property
    def n_params(self) -> int:
        return self.param_offsets()[1]


# the three geometries named in SURVEY.md §8
TINY = QPNetConfig(n_resch=32, n_skipch=32, dilationF_depth=2, dilationF_repeat=1,
                   dilationA_depth=1, dilationA_repeat=1)
PAPER = QPNetConfig(n_resch=64, n_skipch=256, dilationF_depth=4, dilationF_repeat=1,
                    dilationA_depth=4, dilationA_repeat=1)
DEFAULT = QPNetConfig()
# the reference's second shipped network, 'Rd10Rr3Ed4Er1' (src/utils/param_model.py:66-72: 30 fixed layers with dilations up to 512 + 4 adaptive,
# max_length 22500), at the repo's channel widths ...
DEEP = QPNetConfig(dilationF_depth=10, dilationF_repeat=3, dilationA_depth=4, dilationA_repeat=1)
# ... and at the paper-size widths the reference-made fixtures use (tests/golden/*_deep.npz)
DEEP64 = QPNetConfig(n_resch=64, n_skipch=256, dilationF_depth=10, dilationF_repeat=3, dilationA_depth=4, dilationA_repeat=1)
