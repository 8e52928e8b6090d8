"""Closed model tables produced by this repo's builder == tables dumped from the reference build.

Golden: tests/golden/model_tables.json  (oracle/_ref/refdump --cmd tables; reference functions
Model_Type_get_model -> C4_Model_close, src/c4/c4.c:1669).  The transition id order is the evaluation and
tie-break order of the Viterbi recurrence, so this is the first parity pin.
"""
import json, os
import ctypes as C
import pytest
from exonerate_amd import _abi

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "model_tables.json")))
IN_SCOPE = {
    "affine:global:protein": ("affine:global", 1, 1), "affine:bestfit:protein": ("affine:bestfit", 1, 1),
    "affine:local:protein": ("affine:local", 1, 1), "affine:overlap:protein": ("affine:overlap", 1, 1),
    "affine:global:dna": ("affine:global", 0, 0), "affine:bestfit:dna": ("affine:bestfit", 0, 0),
    "affine:local:dna": ("affine:local", 0, 0), "affine:overlap:dna": ("affine:overlap", 0, 0),
    "ungapped:dna": ("ungapped", 0, 0), "ungapped:protein": ("ungapped", 1, 1),
    "est2genome": ("est2genome", 0, 0),
    "protein2dna": ("protein2dna", 1, 0), "protein2dna:bestfit": ("protein2dna:bestfit", 1, 0),
    "protein2genome": ("protein2genome", 1, 0), "protein2genome:bestfit": ("protein2genome:bestfit", 1, 0),
}


def get_model(lib, params, key):
    name, qa, ta = IN_SCOPE[key]
    m = _abi.Model()
    assert lib.c4gpu_model_get(name.encode(), qa, ta, params, m) == 0
    return m


@pytest.mark.parametrize("gold", [g for g in GOLD if g["key"] in IN_SCOPE], ids=lambda g: g["key"])
def test_closed_tables_match_reference(lib, params, gold):
    m = get_model(lib, params, gold["key"])
    assert m.name.decode() == gold["name"]
    assert m.n_states == len(gold["states"])
    assert [m.state_names[i].value.decode() for i in range(m.n_states)] == gold["states"]
    assert (m.start_scope, m.end_scope) == (gold["start_scope"], gold["end_scope"])
    assert (m.start_state, m.end_state) == (gold["start_state"], gold["end_state"])
    assert (m.max_query_advance, m.max_target_advance) == (gold["max_query_advance"], gold["max_target_advance"])
    assert m.total_shadow_designations == gold["shadow_designations"]
    assert m.n_transitions == len(gold["transitions"])
    for i, g in enumerate(gold["transitions"]):
        t = m.transitions[i]
        assert g["id"] == i
        got = (t.name.decode(), t.input, t.output, t.advance_query, t.advance_target, t.calc, t.label)
        exp = (g["name"], g["in"], g["out"], g["aq"], g["at"], g["calc"], g["label"])
        assert got == exp, (i, got, exp)
        mask = 0
        for s in g["dst_shadows"]:
            mask |= 1 << s
        assert t.dst_shadow_mask == mask
    assert m.n_calcs == len(gold["calcs"])
    for i, g in enumerate(gold["calcs"]):
        c = m.calcs[i]
        assert (c.name.decode(), c.max_score, c.protect) == (g["name"], g["max_score"], g["protect"]), i
    assert m.n_shadows == len(gold["shadows"])
    for i, g in enumerate(gold["shadows"]):
        s = m.shadows[i]
        assert s.name.decode() == g["name"] and s.designation == g["designation"]
        assert s.src_state_mask == sum(1 << x for x in g["src_states"])
        assert s.dst_transition_mask == sum(1 << x for x in g["dst_transitions"])


def test_plugin_names_are_the_bootstrapper_keys(lib, params):
    # Codegen_clean_path_component("optimal:est2genome find score"), SURVEY.md section 8b
    m = get_model(lib, params, "est2genome")
    buf = C.create_string_buffer(256)
    lib.c4gpu_model_plugin_name(m, _abi.MODE_FIND_SCORE, 0, buf, 256)
    assert buf.value.decode() == "optimal_58_est2genome_32_find_32_score"
    lib.c4gpu_model_plugin_name(m, _abi.MODE_FIND_PATH, 1, buf, 256)
    assert buf.value.decode() == "optimal_58_est2genome_32_find_32_path_32_continuation"
    lib.c4gpu_model_plugin_name(m, _abi.MODE_FIND_CHECKPOINTS, 1, buf, 256)
    assert buf.value.decode() == "optimal_58_est2genome_32_find_32_checkpoint"


def test_builder_api_reproduces_affine_by_hand(lib, params):
    """Drive the c4m_* builder the way src/model/affine.c drives C4_Model_* and get the same table."""
    m = lib.c4m_model_create(b"by-hand")
    match = lib.c4m_add_state(m, b"match")
    calc = lib.c4m_add_calc(m, b"match", _abi.CALC_MATCH_DNA, 0, 0, 5, 0)
    lib.c4m_add_transition(m, b"start to match", -1, match, 0, 0, -1, _abi.LABEL_NONE)
    lib.c4m_add_transition(m, b"match to end", match, -1, 0, 0, -1, _abi.LABEL_NONE)
    lib.c4m_add_transition(m, b"match", match, match, 1, 1, calc, _abi.LABEL_MATCH)
    assert lib.c4m_model_close(m) == 0
    flat = _abi.Model()
    assert lib.c4m_flatten(m, flat) == 0
    gold = [g for g in GOLD if g["key"] == "ungapped:dna"][0]
    assert [flat.transitions[i].name.decode() for i in range(3)] == [t["name"] for t in gold["transitions"]]
    lib.c4m_model_destroy(m)


def test_invalid_models_are_rejected(lib):
    m = lib.c4m_model_create(b"orphan")
    lib.c4m_add_state(m, b"nowhere")
    assert lib.c4m_model_close(m) == -1      # C4_Model_is_valid, c4.c:1385
    lib.c4m_model_destroy(m)


DERIVED = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "derived_tables.json")))


@pytest.mark.parametrize("gold", DERIVED, ids=lambda g: g["derived_key"])
def test_derived_tables_match_reference(lib, params, gold):
    """C4_DerivedModel_create (c4.c:2292) for BSDP's join / terminal models (heuristic.c:242-330): the closed
    tables and the transition map of c4gpu_model_get_derived against the reference's (refdump --cmd derived)."""
    key, src, dst, ss, es = gold["derived_key"].split("|")
    name, qa, ta = IN_SCOPE[key]
    m = _abi.Model()
    tmap = (C.c_int32 * _abi.MAX_TRANSITIONS)()
    g = gold["table"]
    if len(g["transitions"]) > _abi.MAX_TRANSITIONS:
        pytest.skip("derived model larger than the flattened table")
    assert lib.c4gpu_model_get_derived(name.encode(), qa, ta, params, int(src), int(dst), int(ss), int(es), m, tmap) == 0
    assert m.name.decode() == g["name"][:_abi.NAME_LEN - 1]        # the flattened name field is 48 bytes
    assert [m.state_names[i].value.decode() for i in range(m.n_states)] == g["states"]
    assert (m.start_scope, m.end_scope) == (g["start_scope"], g["end_scope"])
    assert (m.max_query_advance, m.max_target_advance) == (g["max_query_advance"], g["max_target_advance"])
    assert m.total_shadow_designations == g["shadow_designations"]
    assert m.n_transitions == len(g["transitions"])
    for i, t_ in enumerate(g["transitions"]):
        t = m.transitions[i]
        got = (t.name.decode(), t.input, t.output, t.advance_query, t.advance_target, t.calc, t.label)
        exp = (t_["name"], t_["in"], t_["out"], t_["aq"], t_["at"], t_["calc"], t_["label"])
        assert got == exp, (i, got, exp)
        mask = 0
        for s in t_["dst_shadows"]:
            mask |= 1 << s
        assert t.dst_shadow_mask == mask
    assert [tmap[i] for i in range(m.n_transitions)] == gold["transition_map"]
    assert [(m.calcs[i].name.decode(), m.calcs[i].max_score, m.calcs[i].protect) for i in range(m.n_calcs)] == \
           [(c["name"], c["max_score"], c["protect"]) for c in g["calcs"]]
    assert m.n_shadows == len(g["shadows"])
    for i, s_ in enumerate(g["shadows"]):
        s = m.shadows[i]
        assert s.name.decode() == s_["name"] and s.designation == s_["designation"]
