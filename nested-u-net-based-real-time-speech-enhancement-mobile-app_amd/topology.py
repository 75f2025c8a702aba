"""Static description of the streaming NUNet-TLS-LSTM step: stage list, layer
names and the 130 recurrent-state tensors of the reference's signature.

Follows ``/root/reference/dnn_model/converter_proposed.py:26-187`` (input
signature = state names/shapes), ``:188-867`` (wiring) and SURVEY.md A.2/A.8.
Pure data -- no arithmetic lives here.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

N_BINS = 256      # network input bins (DC dropped, interpreter_proposed.py:212-213)
LSTM_UNITS = 21   # proposed.py:21
MID_CH = 32
OUT_CH = 64


@dataclass(frozen=True)
class Stage:
    prefix: str            # Keras layer prefix, e.g. "msfe4_en2"
    depth: int             # D
    f0: int                # frequency bins at the stage input
    conv_tag: str          # state tag of the strided-conv inputs, e.g. "msfe4_ee2"
    spconv_tag: str        # state tag of the sub-pixel-conv inputs, e.g. "msfe4_ed2"
    resample: str          # down-sampling (encoder) / up-sampling (decoder) layer name
    pair: Optional[str]    # decoder only: prefix of the paired encoder stage

    @property
    def is_decoder(self) -> bool:
        return self.pair is not None

    @property
    def fd(self) -> int:   # bottleneck bins F_D
        return self.f0 >> self.depth

    @property
    def lstm_dim(self) -> int:
        return self.fd * MID_CH


ENCODER: Tuple[Stage, ...] = (
    Stage("msfe6_en", 6, 256, "msfe6_ee", "msfe6_ed", "msfe6_down_sampling", None),
    Stage("msfe5_en", 5, 128, "msfe5_ee", "msfe5_ed", "msfe5_down_sampling", None),
    Stage("msfe4_en", 4, 64, "msfe4_ee", "msfe4_ed", "msfe4_down_sampling", None),
    Stage("msfe4_en2", 4, 32, "msfe4_ee2", "msfe4_ed2", "msfe4_down_sampling2", None),
    Stage("msfe4_en3", 4, 16, "msfe4_ee3", "msfe4_ed3", "msfe4_down_sampling3", None),
    Stage("msfe3_en", 3, 8, "msfe3_ee", "msfe3_ed", "msfe3_down_sampling", None),
)
# decoder order and encoder pairing: converter_proposed.py:464,500,542,585,627,675
DECODER: Tuple[Stage, ...] = (
    Stage("msfe3_de", 3, 8, "msfe3_de", "msfe3_dd", "msfe3_upsampling", "msfe3_en"),
    Stage("msfe4_de", 4, 16, "msfe4_de", "msfe4_dd", "msfe4_upsampling", "msfe4_en3"),
    Stage("msfe4_de2", 4, 32, "msfe4_de2", "msfe4_dd2", "msfe4_upsampling2", "msfe4_en2"),
    Stage("msfe4_de3", 4, 64, "msfe4_de3", "msfe4_dd3", "msfe4_upsampling3", "msfe4_en"),
    Stage("msfe5_de", 5, 128, "msfe5_de", "msfe5_dd", "msfe5_upsampling", "msfe5_en"),
    Stage("msfe6_de", 6, 256, "msfe6_de", "msfe6_dd", "msfe6_upsampling", "msfe6_en"),
)
STAGES: Tuple[Stage, ...] = ENCODER + DECODER
STAGE_BY_PREFIX: Dict[str, Stage] = {s.prefix: s for s in STAGES}
CENTRAL_F, CENTRAL_C = 4, 64   # central LSTM sees [4, 64] = 256 features


def conv_state_shape(st: Stage, i: int) -> Tuple[int, int]:
    """(F, C) of the input of strided conv ``i`` (1-based) of a stage."""
    if i == 1:
        return st.f0, (2 * OUT_CH if st.is_decoder else OUT_CH)
    return st.f0 >> (i - 1), (2 * MID_CH if st.is_decoder else MID_CH)


def spconv_state_shape(st: Stage, j: int) -> Tuple[int, int]:
    """(F, C) of the input of sub-pixel conv ``j`` (1-based)."""
    return st.fd << (j - 1), 2 * MID_CH


# ---- dilated-dense bottleneck of the baseline variant (models/nunet_tls.py:190-359) ------------
DDB_BLOCKS = 6                     # dilation 1, 2, 4, 8, 16, 32 in time AND frequency


def bottlenecks() -> List[Tuple[str, int, int]]:
    """[(prefix, F, C)] of the 13 bottlenecks in network order: 6 encoder stages, the central one
    (prefix "ddb" / LSTM "lstm"), 6 decoder stages.  C = 32 in the stages, 64 centrally."""
    out = [(st.prefix, st.fd, MID_CH) for st in ENCODER]
    out.append(("", CENTRAL_F, CENTRAL_C))
    out += [(st.prefix, st.fd, MID_CH) for st in DECODER]
    return out


def ddb_state_specs(prefix: str, f: int, c: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """State tensors of one dilated-dense block, converter_nunet_tls.py:173-180 (stage) / :228-235
    (central): prev_in [1,F,C], prev_k [d,F,k*C/2] for k = 1..6 (d = 2^(k-1) past frames, oldest
    first), prev_out [1,F,C/2]."""
    g = c // 2
    tag = (prefix + "_ddb") if prefix else "ddb"
    specs = [(tag + "_{}_in", (1, f, c))]
    for k in range(1, DDB_BLOCKS + 1):
        specs.append((tag + "_{}%d" % k, (1 << (k - 1), f, k * g)))
    specs.append((tag + "_{}_out", (1, f, g)))
    return specs


def state_specs(variant: str = "lstm") -> List[Tuple[str, Tuple[int, ...]]]:
    """[(base name, per-stream shape)] of the state tensors: 130 for the LSTM variant
    (converter_proposed.py:27-186), 208 for the dilated-dense baseline
    (converter_nunet_tls.py:41-249).  ``base`` carries ``{}`` where the signature says ``prev``
    (inputs) / ``cur`` (outputs); LSTM states have the same name on both sides."""
    specs: List[Tuple[str, Tuple[int, ...]]] = []
    for st in STAGES:
        for i in range(1, st.depth + 1):
            f, c = conv_state_shape(st, i)
            specs.append(("%s_{}%d" % (st.conv_tag, i), (1, f, c)))
        for j in range(1, st.depth + 1):
            f, c = spconv_state_shape(st, j)
            specs.append(("%s_{}%d" % (st.spconv_tag, j), (1, f, c)))
    if variant == "baseline":
        for prefix, f, c in bottlenecks():
            specs += ddb_state_specs(prefix, f, c)
        return specs
    if variant != "lstm":
        raise ValueError("variant must be 'lstm' or 'baseline'")
    for st in ENCODER:
        specs += [(st.prefix + "_h", (LSTM_UNITS,)), (st.prefix + "_c", (LSTM_UNITS,))]
    specs += [("state_h", (LSTM_UNITS,)), ("state_c", (LSTM_UNITS,))]
    for st in DECODER:
        specs += [(st.prefix + "_h", (LSTM_UNITS,)), (st.prefix + "_c", (LSTM_UNITS,))]
    return specs


def input_names(variant: str = "lstm") -> List[str]:
    return ["input"] + [b.format("prev") for b, _ in state_specs(variant)]


def output_names(variant: str = "lstm") -> List[str]:
    return [b.format("cur") for b, _ in state_specs(variant)] + ["model_out"]


def state_floats_per_stream(variant: str = "lstm") -> int:
    n = 0
    for _, shp in state_specs(variant):
        k = 1
        for d in shp:
            k *= d
        n += k
    return n
