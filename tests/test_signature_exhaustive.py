"""The full named-tensor signature of both variants, compared name by name and shape by shape with the reference's own text
(build container only: reads /root/reference with `re`, imports nothing from it).

  * inputs:  the tf.TensorSpec list of the @tf.function that becomes the TF-Lite signature
             (dnn_model/converter_proposed.py:26-187 'nutls_lstm', dnn_model/converter_nunet_tls.py:38-250 'nutls');
  * outputs: the keys of the dict that function returns (converter_proposed.py:729-867, converter_nunet_tls.py:1307-1540);
  * the keyword arguments the streaming loop passes per frame and the outputs it echoes back
    (dnn_model/interpreter_proposed.py:215-350, dnn_model/interpreter_nunet_tls.py:306-545).

topology.py generates the 131 / 209 names from 12 Stage records instead of listing them; this test is what ties the generator to the
reference's literal lists."""
import os
import re

import pytest

from nunet_amd import topology as T

REF = "/root/reference/dnn_model"
pytestmark = [pytest.mark.reference, pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference (build container only)")]

SPEC = re.compile(r"tf\.TensorSpec\(\s*shape=\[([^\]]*)\]\s*,\s*dtype=tf\.float32\s*,\s*name='([^']+)'\s*\)")


def _tensor_specs(path):
    """[(name, shape without the batch dimension)] in file order."""
    out = []
    for shape, name in SPEC.findall(open(path).read()):
        dims = [d.strip() for d in shape.split(",")]
        # conv states / the input: [None, 1, F, C]; the LSTM states are declared with a fixed batch of one: [1, 21]
        assert dims[0] == "None" or (dims[0] == "1" and len(dims) == 2), (name, dims)
        out.append((name, tuple(int(d) for d in dims[1:])))
    return out


def _returned_keys(path):
    """Keys of the signature function's `return { ... }` dict, in file order."""
    text = open(path).read()
    start = text.index("return {")
    body = text[start:text.index("}", start)]
    return re.findall(r"[\"']([A-Za-z0-9_]+)[\"']\s*:", body)


def _runner_call(path, runner):
    """(keyword names, echoed output names) of the per-frame call `tflite_out = <runner>(input=..., a=tflite_out['b'], ...)`."""
    text = open(path).read()
    start = text.index("tflite_out = %s(input=" % runner)
    depth, i = 0, text.index("(", start)
    j = i
    while True:
        depth += text[j] == "("
        depth -= text[j] == ")"
        if depth == 0:
            break
        j += 1
    call = text[i + 1:j]
    kw = re.findall(r"(\w+)\s*=", call)
    echoed = re.findall(r"(\w+)\s*=\s*tflite_out\['(\w+)'\]", call)
    return kw, echoed


@pytest.mark.parametrize("variant,conv,interp,runner", [
    ("lstm", "converter_proposed.py", "interpreter_proposed.py", "nutls_lstm_sm"),
    ("baseline", "converter_nunet_tls.py", "interpreter_nunet_tls.py", "nutls"),
])
def test_every_signature_tensor_matches_the_reference_text(variant, conv, interp, runner):
    specs = _tensor_specs(os.path.join(REF, conv))
    want_n = 131 if variant == "lstm" else 209
    assert len(specs) == want_n
    # The signature is keyword-based on both sides (runner(**kwargs) -> dict), so the ORDER of the lists is not part of the contract: the
    # reference lists all `ee` tensors, then `ed`, ..., topology.py walks stage by stage.  Names as sets, shapes name by name.
    names = [n for n, _ in specs]
    assert len(set(names)) == len(names)
    assert sorted(names) == sorted(T.input_names(variant))
    ours = dict(zip(T.input_names(variant)[1:], (shape for _, shape in T.state_specs(variant))))
    assert specs[0] == ("input", (1, 256, 1))
    for name, shape in specs[1:]:
        assert ours[name] == shape, (name, ours[name], shape)
    # outputs: the returned dict's keys are topology's output names, one per state + model_out
    keys = _returned_keys(os.path.join(REF, conv))
    assert len(set(keys)) == len(keys) == want_n
    assert sorted(keys) == sorted(T.output_names(variant))
    # the streaming loop passes exactly the signature's inputs and echoes output X_cur<k> (or the LSTM state of the same name) into X_prev<k>
    kw, echoed = _runner_call(os.path.join(REF, interp), runner)
    assert sorted(kw) == sorted(T.input_names(variant))
    pairs = dict(zip(T.input_names(variant)[1:], T.output_names(variant)[:-1]))
    assert len(echoed) == want_n - 1
    for a, b in echoed:
        assert pairs[a] == b, (a, b, pairs[a])
