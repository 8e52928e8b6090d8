"""The host side of the path pass's shortcut for one-row sections inside an intron (csrc/c4_engine_launch.inc Engine::init_host ->
KParams::loop_tr, csrc/c4_viterbi_kernel.h viterbi_kernel): which states the parameters prove it for.  No device needed
(c4gpu_loop_sections); the device side is tests/test_gpu_parity.py::test_one_row_sections_inside_an_intron_answered_without_a_dp."""
import ctypes as C

import exonerate_amd as ex
from exonerate_amd import _abi


def _loops(model):
    lib = _abi.load()
    out = (C.c_int32 * model.c.n_states)()
    n = lib.c4gpu_loop_sections(model.c, model.params, out, model.c.n_states)
    return n, list(out)


def _name(x):
    return bytes(x).split(b"\0")[0].decode()


def test_default_est2genome_has_its_two_intron_states():
    """13 + 15 - 30 < 0 (the best 5' site sums to 13.09, the best 3' site to 15.50 -- just under the half that would round it up):
    an intron's two sites never pay for opening it, on either strand."""
    model = ex.Model("est2genome")
    n, loops = _loops(model)
    states = [_name(model.c.state_names[s]) for s in range(model.c.n_states)]
    got = {states[s]: loops[s] for s in range(model.c.n_states) if loops[s] >= 0}
    assert n == len(got) == 2 and all("intron" in k for k in got), got
    for s, k in enumerate(loops):
        if k >= 0:                                    # the loop: from the state to itself, one target column, no calc
            t = model.c.transitions[k]
            assert t.input == t.output == s and (t.advance_query, t.advance_target) == (0, 1) and t.calc < 0


def test_parameters_that_let_an_intron_pay_switch_it_off():
    for change in ({"intron_open_penalty": -20}, {"intron_open_penalty": -28}):
        params = ex.default_params()
        for k, v in change.items():
            setattr(params, k, v)
        n, loops = _loops(ex.Model("est2genome", params=params))
        # -28: 13 + 15 - 28 = 0 is a tie with the loop, and ties are the candidates' order's to decide: not proven
        assert n == 0 and all(x < 0 for x in loops), (change, loops)
    params = ex.default_params()
    params.intron_open_penalty = -29
    assert _loops(ex.Model("est2genome", params=params))[0] == 2


def test_other_models_have_none():
    for name in ("affine:local", "protein2dna", "protein2genome", "ungapped"):
        n, loops = _loops(ex.Model(name))
        assert n == 0 and all(x < 0 for x in loops), (name, loops)
