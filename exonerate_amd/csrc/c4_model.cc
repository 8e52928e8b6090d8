// c4_model.cc — host-side C4 model builder + the in-scope model constructors (plain C++, no GLib).
//
// Re-implements the *semantics* of the reference's model API so that closing a model yields the same
// transition id order (= evaluation and tie-break order of the Viterbi recurrence), the same shadow
// designations and the same scopes as src/c4/c4.c.  tests/test_models.py compares every flattened
// table with tables dumped from the reference build (tests/golden/model_tables.json).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#include "c4gpu.h"
#include "c4m.h"

namespace {

struct Calc {
    std::string name;
    int kind, value, param, max_score, protect;
};
struct Transition {
    std::string name;
    int input, output;          // state indices
    int aq, at;
    int calc;                   // index into calcs or -1
    int label;
    std::vector<int> dst_shadows;
    int id;                     // set by close()
};
struct State {
    std::string name;
    std::vector<int> in_tr, out_tr;     // transition handles, insertion order
    std::vector<int> src_shadows;
};
struct Shadow {
    std::string name;
    std::vector<int> src_states;
    std::vector<int> dst_transitions;   // handles
    int on_target;
    int designation;
};

}  // namespace

struct c4m_model {
    std::string name;
    bool open = true;
    std::vector<State> states;          // [0] START, [1] END
    std::vector<Transition> tr;         // arena, indexed by handle
    std::vector<int> order;             // transition_list: handles in list order
    std::vector<Calc> calcs;
    std::vector<Shadow> shadows;
    int start_scope = C4GPU_SCOPE_ANYWHERE, end_scope = C4GPU_SCOPE_ANYWHERE;
    int max_aq = 0, max_at = 0, total_designations = 0;
    int query_alphabet = C4GPU_ALPHABET_DNA, target_alphabet = C4GPU_ALPHABET_DNA;
};

namespace {

bool is_silent(const Transition &t) { return t.aq == 0 && t.at == 0; }

// C4_Model_topological_sort, c4.c:1418-1486
bool topological_sort(c4m_model *m) {
    const int n = (int)m->order.size();
    std::vector<int> dependent(n, 0), ordered;
    // ids = position in the current list (C4_Model_set_ids, c4.c:1349)
    for (int i = 0; i < n; i++) m->tr[m->order[i]].id = i;
    for (int i = 0; i < n; i++) {
        const Transition &t = m->tr[m->order[i]];
        if (!is_silent(t)) continue;
        for (int h : m->states[t.input].in_tr)
            if (is_silent(m->tr[h])) dependent[m->tr[h].id]++;
    }
    bool removed;
    do {
        removed = false;
        for (int i = 0; i < n; i++) {
            if (dependent[i] != 0) continue;
            const Transition &t = m->tr[m->order[i]];
            if (!is_silent(t)) continue;
            removed = true;
            dependent[i] = -1;
            ordered.push_back(m->order[i]);
            for (int h : m->states[t.input].in_tr) dependent[m->tr[h].id]--;
        }
    } while (removed);
    for (int i = 0; i < n; i++)
        if (!is_silent(m->tr[m->order[i]])) ordered.push_back(m->order[i]);
    if ((int)ordered.size() != n) return false;      // cycle of silent transitions
    std::reverse(ordered.begin(), ordered.end());
    m->order = ordered;
    for (int i = 0; i < n; i++) m->tr[m->order[i]].id = i;
    return true;
}

// C4_Shadow_designate_recur / C4_Shadow_get_designation, c4.c:1539-1581
void designate_recur(const c4m_model *m, int shadow, int handle, std::vector<char> &des,
                     std::vector<char> &visited) {
    const Transition &t = m->tr[handle];
    int state = t.input;
    if (visited[state]) return;
    visited[state] = 1;
    for (int s : t.dst_shadows)
        if (s == shadow) return;
    for (int h : m->states[state].in_tr) {
        des[m->tr[h].id] = 1;
        designate_recur(m, shadow, h, des, visited);
    }
}

std::vector<char> get_designation(const c4m_model *m, int shadow) {
    std::vector<char> des(m->order.size(), 0), visited(m->states.size(), 0);
    for (int h : m->shadows[shadow].dst_transitions) {
        des[m->tr[h].id] = 1;
        designate_recur(m, shadow, h, des, visited);
    }
    return des;
}

// C4_Shadow_designation_fits, c4.c:1583-1624
bool designation_fits(const c4m_model *m, const std::vector<char> &a, const std::vector<char> &b) {
    const int n = (int)m->order.size();
    for (int i = 0; i < n; i++)
        if (a[i] && b[i]) return false;
    std::vector<char> used(m->states.size(), 0);
    for (int i = 0; i < n; i++)
        if (a[i]) used[m->tr[m->order[i]].output] = 1;
    for (int i = 0; i < n; i++)
        if (b[i] && used[m->tr[m->order[i]].input]) return false;
    std::fill(used.begin(), used.end(), 0);
    for (int i = 0; i < n; i++)
        if (b[i]) used[m->tr[m->order[i]].output] = 1;
    for (int i = 0; i < n; i++)
        if (a[i] && used[m->tr[m->order[i]].input]) return false;
    return true;
}

// C4_Model_designate_shadows, c4.c:1638-1667
void designate_shadows(c4m_model *m) {
    std::vector<std::vector<char>> groups;
    for (size_t s = 0; s < m->shadows.size(); s++) {
        std::vector<char> cur = get_designation(m, (int)s);
        m->shadows[s].designation = -1;
        for (size_t g = 0; g < groups.size(); g++) {
            if (designation_fits(m, groups[g], cur)) {
                for (size_t i = 0; i < cur.size(); i++)
                    if (cur[i]) groups[g][i] = 1;
                m->shadows[s].designation = (int)g;
                break;
            }
        }
        if (m->shadows[s].designation == -1) {
            m->shadows[s].designation = (int)groups.size();
            groups.push_back(cur);
        }
    }
    m->total_designations = (int)groups.size();
}

bool path_possible(const c4m_model *m, int src, int dst, std::vector<char> &visited) {
    visited[src] = 1;
    for (int h : m->states[src].out_tr) {
        int next = m->tr[h].output;
        if (next == dst) return true;
        if (!visited[next] && path_possible(m, next, dst, visited)) return true;
    }
    return false;
}

void copy_name(char *dst, const std::string &s) {
    snprintf(dst, C4GPU_NAME_LEN, "%s", s.c_str());
}

const char *match_name(int qa, int ta) {
    if (qa == C4GPU_ALPHABET_DNA && ta == C4GPU_ALPHABET_DNA) return "dna2dna";
    if (qa == C4GPU_ALPHABET_PROTEIN && ta == C4GPU_ALPHABET_PROTEIN) return "protein2protein";
    if (qa == C4GPU_ALPHABET_PROTEIN && ta == C4GPU_ALPHABET_DNA) return "protein2dna";
    return "dna2protein";
}

int submat_max(const int32_t m[24][24]) {   // Submat_max_score, submat.c
    int mx = m[0][0];
    for (int i = 0; i < 24; i++)
        for (int j = 0; j < 24; j++) mx = std::max(mx, (int)m[i][j]);
    return mx;
}

}  // namespace

extern "C" {

c4m_model *c4m_model_create(const char *name) {
    c4m_model *m = new c4m_model;
    m->name = name;
    m->states.push_back(State{"START", {}, {}, {}});
    m->states.push_back(State{"END", {}, {}, {}});
    return m;
}
void c4m_model_destroy(c4m_model *m) { delete m; }
void c4m_model_rename(c4m_model *m, const char *name) { m->name = name; }
void c4m_model_open(c4m_model *m) { m->open = true; }
int c4m_model_is_open(const c4m_model *m) { return m->open ? 1 : 0; }
void c4m_model_set_alphabets(c4m_model *m, int qa, int ta) { m->query_alphabet = qa; m->target_alphabet = ta; }

int c4m_add_state(c4m_model *m, const char *name) {
    m->states.push_back(State{name, {}, {}, {}});
    return (int)m->states.size() - 1;
}

int c4m_add_calc(c4m_model *m, const char *name, int kind, int value, int param, int max_score, int protect) {
    m->calcs.push_back(Calc{name, kind, value, param, max_score, protect});
    return (int)m->calcs.size() - 1;
}

int c4m_add_transition(c4m_model *m, const char *name, int input, int output, int aq, int at,
                       int calc, int label) {
    if (input < 0) input = 0;
    if (output < 0) output = 1;
    Transition t{name, input, output, aq, at, calc, label, {}, -1};
    int h = (int)m->tr.size();
    m->tr.push_back(t);
    m->states[input].out_tr.push_back(h);
    m->states[output].in_tr.push_back(h);
    m->order.push_back(h);
    return h;
}

int c4m_add_shadow(c4m_model *m, const char *name, int src_state, int dst_transition, int on_target) {
    if (src_state < 0) src_state = 0;
    Shadow s{name, {}, {}, on_target, -1};
    int id = (int)m->shadows.size();
    m->shadows.push_back(s);
    c4m_shadow_add_src_state(m, id, src_state);
    if (dst_transition >= 0) {
        c4m_shadow_add_dst_transition(m, id, dst_transition);
    } else {
        std::vector<int> into_end = m->states[1].in_tr;
        for (int h : into_end) c4m_shadow_add_dst_transition(m, id, h);
    }
    return id;
}
void c4m_shadow_add_src_state(c4m_model *m, int shadow, int state) {
    m->shadows[shadow].src_states.push_back(state);
    m->states[state].src_shadows.push_back(shadow);
}
void c4m_shadow_add_dst_transition(c4m_model *m, int shadow, int transition) {
    m->shadows[shadow].dst_transitions.push_back(transition);
    m->tr[transition].dst_shadows.push_back(shadow);
}

void c4m_configure_start_state(c4m_model *m, int scope) { m->start_scope = scope; }
void c4m_configure_end_state(c4m_model *m, int scope) { m->end_scope = scope; }

int c4m_model_close(c4m_model *m) {
    // C4_Model_is_valid, c4.c:1385
    for (size_t s = 0; s < m->states.size(); s++) {
        if (s == 0) { if (!m->states[s].in_tr.empty()) return -1; }
        else if (m->states[s].in_tr.empty()) return -1;
        if (s == 1) { if (!m->states[s].out_tr.empty()) return -1; }
        else if (m->states[s].out_tr.empty()) return -1;
    }
    std::vector<char> visited(m->states.size(), 0);
    if (!path_possible(m, 0, 1, visited)) return -1;
    if (!topological_sort(m)) return -1;
    designate_shadows(m);
    m->max_aq = m->max_at = 0;                       // C4_Model_finalise, c4.c:1513
    for (int h : m->order) {
        m->max_aq = std::max(m->max_aq, m->tr[h].aq);
        m->max_at = std::max(m->max_at, m->tr[h].at);
    }
    m->open = false;
    return 0;
}

// C4_Model_make_stereo, c4.c:681-770
void c4m_make_stereo(c4m_model *m, const char *suffix_a, const char *suffix_b) {
    const int prev_states = (int)m->states.size();
    const std::vector<int> prev_order = m->order;
    const int prev_shadows = (int)m->shadows.size();
    std::vector<int> state_map(prev_states, -1);
    std::vector<int> tr_map(m->tr.size(), -1);
    for (int s = 2; s < prev_states; s++)
        state_map[s] = c4m_add_state(m, (m->states[s].name + " " + suffix_b).c_str());
    for (int h : prev_order) {
        Transition t = m->tr[h];
        tr_map[h] = c4m_add_transition(m, (t.name + " " + suffix_b).c_str(),
                                       state_map[t.input], state_map[t.output], t.aq, t.at, t.calc, t.label);
    }
    for (int s = 0; s < prev_shadows; s++) {
        Shadow sh = m->shadows[s];
        int ns = c4m_add_shadow(m, (sh.name + " " + suffix_b).c_str(), state_map[sh.src_states[0]],
                                tr_map[sh.dst_transitions[0]], sh.on_target);
        // the reference copies the *remaining* src states / dst transitions unmapped (c4.c:732-741)
        for (size_t j = 1; j < sh.src_states.size(); j++) c4m_shadow_add_src_state(m, ns, sh.src_states[j]);
        for (size_t j = 1; j < sh.dst_transitions.size(); j++)
            c4m_shadow_add_dst_transition(m, ns, sh.dst_transitions[j]);
    }
    for (int s = 2; s < prev_states; s++) m->states[s].name += std::string(" ") + suffix_a;
    for (int h : prev_order) m->tr[h].name += std::string(" ") + suffix_a;
    for (int s = 0; s < prev_shadows; s++) m->shadows[s].name += std::string(" ") + suffix_a;
}

// C4_Model_insert, c4.c:963-996 (portals/spans/codegen strings are not part of the DP and are not kept)
int c4m_insert(c4m_model *target, const c4m_model *insert, int src, int dst) {
    if (!target->open || insert->open) return -1;
    if (src < 0) src = 0;
    if (dst < 0) dst = 1;
    std::vector<int> calc_map(insert->calcs.size(), -1);
    for (size_t c = 0; c < insert->calcs.size(); c++) {          // C4_Model_insert_calcs / C4_Calc_diff
        const Calc &ic = insert->calcs[c];
        int found = -1;
        for (size_t k = 0; k < target->calcs.size(); k++) {
            const Calc &tc = target->calcs[k];
            if (tc.max_score == ic.max_score && tc.kind == ic.kind && tc.value == ic.value &&
                tc.param == ic.param && tc.protect == ic.protect) { found = (int)k; break; }
        }
        if (found < 0)
            found = c4m_add_calc(target, ic.name.c_str(), ic.kind, ic.value, ic.param, ic.max_score, ic.protect);
        calc_map[c] = found;
    }
    std::vector<int> state_map(insert->states.size(), -1);
    for (size_t s = 2; s < insert->states.size(); s++)
        state_map[s] = c4m_add_state(target, insert->states[s].name.c_str());
    state_map[0] = src;
    state_map[1] = dst;
    std::vector<int> tr_map(insert->tr.size(), -1);
    for (int h : insert->order) {
        const Transition &t = insert->tr[h];
        tr_map[h] = c4m_add_transition(target, t.name.c_str(), state_map[t.input], state_map[t.output],
                                       t.aq, t.at, t.calc >= 0 ? calc_map[t.calc] : -1, t.label);
    }
    for (const Shadow &sh : insert->shadows) {
        int ns = c4m_add_shadow(target, sh.name.c_str(), state_map[sh.src_states[0]],
                                tr_map[sh.dst_transitions[0]], sh.on_target);
        for (size_t j = 1; j < sh.src_states.size(); j++)
            c4m_shadow_add_src_state(target, ns, state_map[sh.src_states[j]]);
        for (size_t j = 1; j < sh.dst_transitions.size(); j++)
            c4m_shadow_add_dst_transition(target, ns, tr_map[sh.dst_transitions[j]]);
    }
    return 0;
}

// ---- derived models -------------------------------------------------------------------------------------------
// C4_DerivedModel_create (c4.c:2292-2337) over C4_Model_select (c4.c:2217-2285): the sub-model of every path
// from state `src` to state `dst` of a CLOSED model — BSDP's join and terminal models (heuristic.c:242-330).
// Creation order is the reference's: transitions out of src (from START), transitions into dst (to END), then
// a depth-first walk over the states reused so far; states, calcs and shadows are created on first use.
namespace {

// C4_Model_path_is_possible, c4.c:1307-1341: dst reachable from src over at least one transition
bool derive_path_possible(const c4m_model *m, int src, int dst, std::vector<char> &visited) {
    visited[src] = 1;
    for (int h : m->states[src].out_tr) {
        const int next = m->tr[h].output;
        if (next == dst) return true;
        if (!visited[next] && derive_path_possible(m, next, dst, visited)) return true;
    }
    return false;
}
bool derive_reach(const c4m_model *m, int src, int dst) {
    std::vector<char> visited(m->states.size(), 0);
    return derive_path_possible(m, src, dst, visited);
}

struct Derive {
    const c4m_model *old_model;
    c4m_model *new_model;
    std::vector<int> state_map, calc_map;                 // old index -> new index or -1
    std::vector<std::vector<int>> proto_states, proto_transitions;   // per old shadow: new src states / dst transitions
    std::vector<char> proto_used;
    std::vector<int> transition_map;                      // new handle -> old handle

    void reuse_state(int old_state) {                     // C4_Model_segment_reuse_state, c4.c:2047
        if (old_state == 0 || old_state == 1 || state_map[old_state] >= 0) return;
        const int ns = c4m_add_state(new_model, old_model->states[old_state].name.c_str());
        state_map[old_state] = ns;
        for (int sh : old_model->states[old_state].src_shadows) { proto_used[sh] = 1; proto_states[sh].push_back(ns); }
    }
    void add_transition(int old_handle, bool from_start, bool to_end) {   // C4_Model_segment_add_transition, c4.c:2072
        const Transition &t = old_model->tr[old_handle];
        if (!from_start) reuse_state(t.input);
        if (!to_end) reuse_state(t.output);
        int calc = -1;
        if (t.calc >= 0) {
            if (calc_map[t.calc] < 0) {
                const Calc &c = old_model->calcs[t.calc];
                calc_map[t.calc] = c4m_add_calc(new_model, c.name.c_str(), c.kind, c.value, c.param, c.max_score, c.protect);
            }
            calc = calc_map[t.calc];
        }
        const int nh = c4m_add_transition(new_model, t.name.c_str(), from_start ? C4M_START : state_map[t.input],
                                          to_end ? C4M_END : state_map[t.output], t.aq, t.at, calc, t.label);
        if ((int)transition_map.size() <= nh) transition_map.resize(nh + 1, -1);
        transition_map[nh] = old_handle;
        for (int sh : t.dst_shadows) { proto_used[sh] = 1; proto_transitions[sh].push_back(nh); }
    }
    void recur(int state, std::vector<char> &visited) {   // C4_Model_segment_recur, c4.c:2127
        if (state_map[state] < 0 || visited[state] || state == 0 || state == 1) return;
        visited[state] = 1;
        const std::vector<int> outs = old_model->states[state].out_tr;
        for (int h : outs) {
            if (old_model->tr[h].output == 1) continue;
            add_transition(h, false, false);
            recur(old_model->tr[h].output, visited);
        }
    }
};

}  // namespace

extern "C" c4m_model *c4m_derive(const c4m_model *model, int src_state, int dst_state, int start_scope, int end_scope,
                                 int *transition_map, int transition_map_len) {
    if (!model || model->open) return nullptr;
    if (src_state < 0) src_state = 0;
    if (dst_state < 0) dst_state = 1;
    const std::string name = "Segment(\"" + model->states[src_state].name + "\"->\"" + model->states[dst_state].name +
                             "\"):[" + model->name + "]";
    Derive d;
    d.old_model = model;
    d.new_model = c4m_model_create(name.c_str());
    c4m_model_set_alphabets(d.new_model, model->query_alphabet, model->target_alphabet);
    d.state_map.assign(model->states.size(), -1);
    d.calc_map.assign(model->calcs.size(), -1);
    d.proto_states.resize(model->shadows.size());
    d.proto_transitions.resize(model->shadows.size());
    d.proto_used.assign(model->shadows.size(), 0);
    // shadows that start at src start at the new START (c4.c:2240-2247)
    for (int sh : model->states[src_state].src_shadows) { d.proto_used[sh] = 1; d.proto_states[sh].push_back(0); }
    for (int h : model->states[src_state].out_tr)                       // transitions from start
        if (derive_reach(model, model->tr[h].output, dst_state)) d.add_transition(h, true, false);
    for (int h : model->states[dst_state].in_tr)                        // transitions to end
        if (derive_reach(model, src_state, model->tr[h].input)) d.add_transition(h, false, true);
    std::vector<char> visited(model->states.size(), 0);
    for (size_t s = 0; s < model->states.size(); s++) d.recur((int)s, visited);
    for (size_t sh = 0; sh < model->shadows.size(); sh++) {              // C4_ProtoShadow_generate, c4.c:2013
        if (!d.proto_used[sh]) continue;
        if (d.proto_states[sh].empty() || d.proto_transitions[sh].empty()) { c4m_model_destroy(d.new_model); return nullptr; }
        const int ns = c4m_add_shadow(d.new_model, model->shadows[sh].name.c_str(),
                                      d.proto_states[sh][0] == 0 ? C4M_START : d.proto_states[sh][0],
                                      d.proto_transitions[sh][0], model->shadows[sh].on_target);
        for (size_t j = 1; j < d.proto_states[sh].size(); j++) c4m_shadow_add_src_state(d.new_model, ns, d.proto_states[sh][j]);
        for (size_t j = 1; j < d.proto_transitions[sh].size(); j++)
            c4m_shadow_add_dst_transition(d.new_model, ns, d.proto_transitions[sh][j]);
    }
    if (c4m_model_close(d.new_model) != 0) { c4m_model_destroy(d.new_model); return nullptr; }
    c4m_configure_start_state(d.new_model, start_scope);
    c4m_configure_end_state(d.new_model, end_scope);
    // transition_map[derived id] = original id (C4_DerivedModel::transition_map, c4.c:2322-2333)
    if (transition_map)
        for (size_t nh = 0; nh < d.transition_map.size(); nh++) {
            const int id = d.new_model->tr[nh].id;
            if (id >= 0 && id < transition_map_len) transition_map[id] = model->tr[d.transition_map[nh]].id;
        }
    return d.new_model;
}

int c4m_select_transitions(const c4m_model *m, int label, int *handles, int max) {
    int n = 0;
    for (int h : m->order)
        if (m->tr[h].label == label) { if (n < max) handles[n] = h; n++; }
    return n;
}
int c4m_select_single_transition(const c4m_model *m, int label) {
    int h[2];
    return c4m_select_transitions(m, label, h, 2) == 1 ? h[0] : -1;
}
int c4m_transition_input(const c4m_model *m, int t) { return m->tr[t].input; }
int c4m_transition_output(const c4m_model *m, int t) { return m->tr[t].output; }
int c4m_transition_id(const c4m_model *m, int t) { return m->tr[t].id; }

int c4m_flatten(const c4m_model *m, c4gpu_model *out) {
    if (m->open) return -1;
    if (m->states.size() > C4GPU_MAX_STATES || m->order.size() > C4GPU_MAX_TRANSITIONS ||
        m->calcs.size() > C4GPU_MAX_CALCS || m->shadows.size() > C4GPU_MAX_SHADOWS) return -2;
    memset(out, 0, sizeof(*out));
    copy_name(out->name, m->name);
    out->n_states = (int)m->states.size();
    out->n_transitions = (int)m->order.size();
    out->n_calcs = (int)m->calcs.size();
    out->n_shadows = (int)m->shadows.size();
    out->start_state = 0;
    out->end_state = 1;
    out->start_scope = m->start_scope;
    out->end_scope = m->end_scope;
    out->max_query_advance = m->max_aq;
    out->max_target_advance = m->max_at;
    out->total_shadow_designations = m->total_designations;
    out->query_alphabet = m->query_alphabet;
    out->target_alphabet = m->target_alphabet;
    for (size_t s = 0; s < m->states.size(); s++) copy_name(out->state_names[s], m->states[s].name);
    for (size_t c = 0; c < m->calcs.size(); c++) {
        const Calc &k = m->calcs[c];
        copy_name(out->calcs[c].name, k.name);
        out->calcs[c].kind = k.kind;  out->calcs[c].value = k.value;  out->calcs[c].param = k.param;
        out->calcs[c].max_score = k.max_score;  out->calcs[c].protect = k.protect;
    }
    for (size_t i = 0; i < m->order.size(); i++) {
        const Transition &t = m->tr[m->order[i]];
        c4gpu_transition &o = out->transitions[i];
        copy_name(o.name, t.name);
        o.input = t.input;  o.output = t.output;
        o.advance_query = t.aq;  o.advance_target = t.at;
        o.calc = t.calc;  o.label = t.label;
        o.dst_shadow_mask = 0;
        for (int s : t.dst_shadows) o.dst_shadow_mask |= 1u << s;
    }
    for (size_t s = 0; s < m->shadows.size(); s++) {
        const Shadow &sh = m->shadows[s];
        c4gpu_shadow &o = out->shadows[s];
        copy_name(o.name, sh.name);
        o.designation = sh.designation;
        o.on_target = sh.on_target;
        for (int st : sh.src_states) o.src_state_mask |= 1u << st;
        for (int h : sh.dst_transitions) o.dst_transition_mask |= 1ull << m->tr[h].id;
    }
    return 0;
}

/* ---- model constructors -------------------------------------------------------------------------- */

// Ungapped_create, src/model/ungapped.c:122-178
c4m_model *c4m_ungapped_create(int qa, int ta, const c4gpu_params *p) {
    std::string name = std::string("ungapped:") + match_name(qa, ta);
    c4m_model *m = c4m_model_create(name.c_str());
    c4m_model_set_alphabets(m, qa, ta);
    int match_state = c4m_add_state(m, "match");
    int kind, aq = 1, at = 1, mx;
    if (qa == C4GPU_ALPHABET_DNA && ta == C4GPU_ALPHABET_DNA) {
        kind = C4GPU_CALC_MATCH_DNA;  mx = submat_max(p->dna_submat);
    } else if (qa == C4GPU_ALPHABET_PROTEIN && ta == C4GPU_ALPHABET_PROTEIN) {
        kind = C4GPU_CALC_MATCH_PROTEIN;  mx = submat_max(p->protein_submat);
    } else if (qa == C4GPU_ALPHABET_PROTEIN && ta == C4GPU_ALPHABET_DNA) {
        kind = C4GPU_CALC_MATCH_P2D;  at = 3;  mx = submat_max(p->protein_submat);
    } else {
        c4m_model_destroy(m);
        return nullptr;                       // dna2protein / codon2codon: not accelerated
    }
    int calc = c4m_add_calc(m, "match", kind, 0, 0, mx, C4GPU_PROTECT_NONE);
    c4m_add_transition(m, "start to match", C4M_START, match_state, 0, 0, -1, C4GPU_LABEL_NONE);
    c4m_add_transition(m, "match to end", match_state, C4M_END, 0, 0, -1, C4GPU_LABEL_NONE);
    c4m_add_transition(m, "match", match_state, match_state, aq, at, calc, C4GPU_LABEL_MATCH);
    c4m_model_close(m);
    return m;
}

// Affine_create, src/model/affine.c:150-255
c4m_model *c4m_affine_create(int type, int qa, int ta, const c4gpu_params *p) {
    static const char *type_name[] = {"global", "bestfit", "local", "overlap"};
    static const int type_scope[] = {C4GPU_SCOPE_CORNER, C4GPU_SCOPE_QUERY, C4GPU_SCOPE_ANYWHERE,
                                     C4GPU_SCOPE_EDGE};
    if (type < 0 || type > 3) return nullptr;
    c4m_model *m = c4m_ungapped_create(qa, ta, p);
    if (!m) return nullptr;
    std::string name = std::string("affine:") + type_name[type] + ":" + match_name(qa, ta);
    c4m_model_rename(m, name.c_str());
    c4m_configure_start_state(m, type_scope[type]);
    c4m_configure_end_state(m, type_scope[type]);
    c4m_model_open(m);
    int ins = c4m_add_state(m, "insert");
    int del = c4m_add_state(m, "delete");
    int match_tr = c4m_select_single_transition(m, C4GPU_LABEL_MATCH);
    const Transition mt = m->tr[match_tr];
    const bool codon = std::max(mt.aq, mt.at) == 3;
    int open_v = codon ? p->codon_gap_open : p->gap_open;
    int ext_v = codon ? p->codon_gap_extend : p->gap_extend;
    // the calc's max_score is always the plain gap penalty (affine.c:210-217); the codon variants
    // only differ in the value the calc function returns
    int open_c = c4m_add_calc(m, "gap open", C4GPU_CALC_CONST, open_v, codon ? 1 : 0, p->gap_open, 0);
    int ext_c = c4m_add_calc(m, "gap extend", C4GPU_CALC_CONST, ext_v, codon ? 1 : 0, p->gap_extend, 0);
    c4m_add_transition(m, "match to insert", mt.input, ins, mt.aq, 0, open_c, C4GPU_LABEL_GAP);
    c4m_add_transition(m, "match to delete", mt.input, del, 0, mt.at, open_c, C4GPU_LABEL_GAP);
    c4m_add_transition(m, "insert", ins, ins, mt.aq, 0, ext_c, C4GPU_LABEL_GAP);
    c4m_add_transition(m, "insert to match", ins, mt.output, 0, 0, -1, C4GPU_LABEL_NONE);
    c4m_add_transition(m, "delete", del, del, 0, mt.at, ext_c, C4GPU_LABEL_GAP);
    c4m_add_transition(m, "delete to match", del, mt.output, 0, 0, -1, C4GPU_LABEL_NONE);
    c4m_model_close(m);
    return m;
}

// Intron_create(suffix, on_query=FALSE, on_target=TRUE, is_forward), src/model/intron.c:497-697
c4m_model *c4m_intron_create(const char *suffix, int is_forward, const c4gpu_params *p) {
    std::string sfx = suffix;
    c4m_model *m = c4m_model_create(("intron " + sfx).c_str());
    const char *pre_name = is_forward ? "5'ss forward" : "3'ss reverse";
    const char *post_name = is_forward ? "3'ss forward" : "5'ss reverse";
    int pre_label = is_forward ? C4GPU_LABEL_5SS : C4GPU_LABEL_3SS;
    int post_label = is_forward ? C4GPU_LABEL_3SS : C4GPU_LABEL_5SS;
    int pre_ss = is_forward ? C4GPU_SS5_FORWARD : C4GPU_SS3_REVERSE;
    int post_ss = is_forward ? C4GPU_SS3_FORWARD : C4GPU_SS5_REVERSE;
    // bound = (int)((float)bound + max_score)   (intron.c:510,564: `bound += gfloat`)
    int pre_bound = p->intron_open_penalty;
    pre_bound = (int)((float)pre_bound + c4gpu_splice_max_score(&p->splice[pre_ss]));
    int post_bound = 0;
    post_bound = (int)((float)post_bound + c4gpu_splice_max_score(&p->splice[post_ss]));
    int pre_calc = c4m_add_calc(m, (std::string(pre_name) + " " + sfx).c_str(), C4GPU_CALC_SPLICE_PRE,
                                p->intron_open_penalty, pre_ss, pre_bound, C4GPU_PROTECT_UNDERFLOW);
    int post_calc = c4m_add_calc(m, (std::string(post_name) + " " + sfx).c_str(), C4GPU_CALC_SPLICE_POST,
                                 0, post_ss, post_bound, C4GPU_PROTECT_UNDERFLOW);
    std::string iname = "intron " + sfx;
    int istate = c4m_add_state(m, iname.c_str());
    c4m_add_transition(m, ("(START) to " + iname).c_str(), C4M_START, istate, 0, 2, pre_calc, pre_label);
    c4m_add_transition(m, ("target intron loop " + sfx).c_str(), istate, istate, 0, 1, -1, C4GPU_LABEL_INTRON);
    c4m_add_transition(m, (iname + " to (END)").c_str(), istate, C4M_END, 0, 2, post_calc, post_label);
    c4m_add_shadow(m, ("target intron " + sfx).c_str(), C4M_START, -1, 1);
    c4m_model_close(m);
    return m;
}

// EST2Genome_create, src/model/est2genome.c:58-94
c4m_model *c4m_est2genome_create(const c4gpu_params *p) {
    c4m_model *m = c4m_affine_create(C4M_AFFINE_LOCAL, C4GPU_ALPHABET_DNA, C4GPU_ALPHABET_DNA, p);
    c4m_model_rename(m, "est2genome");
    c4m_model_open(m);
    c4m_make_stereo(m, "forward", "reverse");
    int match[2];
    c4m_select_transitions(m, C4GPU_LABEL_MATCH, match, 2);
    c4m_model *fwd = c4m_intron_create("forward", 1, p);
    c4m_model *rev = c4m_intron_create("reverse", 0, p);
    c4m_insert(m, fwd, m->tr[match[0]].input, m->tr[match[0]].input);
    c4m_insert(m, rev, m->tr[match[1]].input, m->tr[match[1]].input);
    c4m_model_destroy(fwd);
    c4m_model_destroy(rev);
    c4m_model_close(m);
    return m;
}

// Frameshift_add(model, match_state, suffix, apply_to_query=FALSE), src/model/frameshift.c:74-131
static void frameshift_add(c4m_model *m, int match_state, const char *suffix, const c4gpu_params *p) {
    std::string sfx = suffix;
    int fs = c4m_add_state(m, ("frameshift " + sfx).c_str());
    int calc = -1;
    for (size_t c = 0; c < m->calcs.size(); c++)
        if (m->calcs[c].kind == C4GPU_CALC_CONST && m->calcs[c].param == 2) calc = (int)c;
    if (calc < 0)
        calc = c4m_add_calc(m, "frameshift", C4GPU_CALC_CONST, p->frameshift_penalty, 2,
                            p->frameshift_penalty, C4GPU_PROTECT_NONE);
    c4m_add_transition(m, ("frameshift open 1 " + sfx).c_str(), match_state, fs, 0, 1, calc, C4GPU_LABEL_FRAMESHIFT);
    c4m_add_transition(m, ("frameshift open 2 " + sfx).c_str(), match_state, fs, 0, 2, calc, C4GPU_LABEL_FRAMESHIFT);
    c4m_add_transition(m, ("frameshift close 0 " + sfx).c_str(), fs, match_state, 0, 0, -1, C4GPU_LABEL_NONE);
    c4m_add_transition(m, ("frameshift close 3 " + sfx).c_str(), fs, match_state, 0, 3, -1, C4GPU_LABEL_FRAMESHIFT);
}

// Protein2DNA_create, src/model/protein2dna.c:56-74
c4m_model *c4m_protein2dna_create(int type, const c4gpu_params *p) {
    static const char *type_name[] = {"global", "bestfit", "local", "overlap"};
    c4m_model *m = c4m_affine_create(type, C4GPU_ALPHABET_PROTEIN, C4GPU_ALPHABET_DNA, p);
    if (!m) return nullptr;
    c4m_model_rename(m, (std::string("protein2dna:") + type_name[type]).c_str());
    c4m_model_open(m);
    int match_tr = c4m_select_single_transition(m, C4GPU_LABEL_MATCH);
    frameshift_add(m, m->tr[match_tr].input, "p2d", p);
    c4m_model_close(m);
    return m;
}

// Phase_create(NULL, match = protein2dna, on_query = FALSE, on_target = TRUE), src/model/phase.c:354-548:
// three target introns (phase 0:0, 1:2, 2:1) around split-codon states; the post-intron transitions score
// the codon re-assembled across the intron from the shadow (phase.c:188-208).
c4m_model *c4m_phase_create(const c4gpu_params *p) {
    const std::string sfx = "phase-T";
    c4m_model *m = c4m_model_create(sfx.c_str());
    c4m_model_set_alphabets(m, C4GPU_ALPHABET_PROTEIN, C4GPU_ALPHABET_DNA);
    c4m_model *i00 = c4m_intron_create(("0:0 " + sfx).c_str(), 1, p);
    c4m_model *i12 = c4m_intron_create(("1:2 " + sfx).c_str(), 1, p);
    c4m_model *i21 = c4m_intron_create(("2:1 " + sfx).c_str(), 1, p);
    const int mx = submat_max(p->protein_submat);               // Match_max_score, match.c:885
    int c1 = c4m_add_calc(m, ("phase1post to dst " + sfx).c_str(), C4GPU_CALC_PHASE_POST, 0, 1, mx, 0);
    int c2 = c4m_add_calc(m, ("phase2post to dst " + sfx).c_str(), C4GPU_CALC_PHASE_POST, 0, 2, mx, 0);
    int pre1 = c4m_add_state(m, ("phase1pre " + sfx).c_str());
    int post1 = c4m_add_state(m, ("phase1post " + sfx).c_str());
    int pre2 = c4m_add_state(m, ("phase2pre " + sfx).c_str());
    int post2 = c4m_add_state(m, ("phase2post " + sfx).c_str());
    // against a peptide, introns on the target: pre 0/1 and 0/2, post 1/2 and 1/1 (phase.c:400-411)
    c4m_add_transition(m, ("(START) to phase1pre " + sfx).c_str(), C4M_START, pre1, 0, 1, -1, C4GPU_LABEL_SPLIT_CODON);
    c4m_add_transition(m, ("(START) to phase2pre " + sfx).c_str(), C4M_START, pre2, 0, 2, -1, C4GPU_LABEL_SPLIT_CODON);
    int t1 = c4m_add_transition(m, ("phase1post " + sfx + " to (END)").c_str(), post1, C4M_END, 1, 2, c1, C4GPU_LABEL_SPLIT_CODON);
    int t2 = c4m_add_transition(m, ("phase2post " + sfx + " to (END)").c_str(), post2, C4M_END, 1, 1, c2, C4GPU_LABEL_SPLIT_CODON);
    c4m_insert(m, i00, C4M_START, C4M_END);
    c4m_insert(m, i12, pre1, post1);
    c4m_insert(m, i21, pre2, post2);
    c4m_shadow_add_dst_transition(m, 1, t1);                    // phase.c:528-532
    c4m_shadow_add_dst_transition(m, 2, t2);
    c4m_model_destroy(i00); c4m_model_destroy(i12); c4m_model_destroy(i21);
    c4m_model_close(m);
    return m;
}

// Protein2Genome_create, src/model/protein2genome.c:44-68
c4m_model *c4m_protein2genome_create(int type, const c4gpu_params *p) {
    static const char *type_name[] = {"global", "bestfit", "local", "overlap"};
    c4m_model *m = c4m_protein2dna_create(type, p);
    if (!m) return nullptr;
    c4m_model_rename(m, (std::string("protein2genome:") + type_name[type]).c_str());
    c4m_model_open(m);
    int match_tr = c4m_select_single_transition(m, C4GPU_LABEL_MATCH);
    c4m_model *phase = c4m_phase_create(p);
    c4m_insert(m, phase, m->tr[match_tr].input, m->tr[match_tr].output);
    c4m_model_destroy(phase);
    c4m_model_close(m);
    return m;
}

static c4m_model *model_of_type(const char *type, int qa, int ta, const c4gpu_params *params);

// Model_Type_get_model, src/model/modeltype.c
int c4gpu_model_get(const char *type, int qa, int ta, const c4gpu_params *params, c4gpu_model *out) {
    c4gpu_params defaults;
    if (!params) { c4gpu_params_default(&defaults); params = &defaults; }
    c4m_model *m = model_of_type(type, qa, ta, params);
    if (!m) return -1;
    int rc = c4m_flatten(m, out);
    c4m_model_destroy(m);
    return rc;
}

// C4_DerivedModel_create on a model type: BSDP's join / terminal models (heuristic.c:242-330)
int c4gpu_model_get_derived(const char *type, int qa, int ta, const c4gpu_params *params, int src_state,
                            int dst_state, int start_scope, int end_scope, c4gpu_model *out,
                            int32_t *transition_map) {
    c4gpu_params defaults;
    if (!params) { c4gpu_params_default(&defaults); params = &defaults; }
    c4m_model *m = model_of_type(type, qa, ta, params);
    if (!m) return -1;
    int map[C4GPU_MAX_TRANSITIONS];
    for (int i = 0; i < C4GPU_MAX_TRANSITIONS; i++) map[i] = -1;
    c4m_model *d = c4m_derive(m, src_state, dst_state, start_scope, end_scope, map, C4GPU_MAX_TRANSITIONS);
    c4m_model_destroy(m);
    if (!d) return -1;
    int rc = c4m_flatten(d, out);
    c4m_model_destroy(d);
    if (transition_map && rc == 0)
        for (int i = 0; i < out->n_transitions; i++) transition_map[i] = map[i];
    return rc;
}

static c4m_model *model_of_type(const char *type, int qa, int ta, const c4gpu_params *params) {
    c4m_model *m = nullptr;
    std::string t = type;
    if (t == "ungapped" || t == "u") m = c4m_ungapped_create(qa, ta, params);
    else if (t == "affine:global" || t == "a:g") m = c4m_affine_create(C4M_AFFINE_GLOBAL, qa, ta, params);
    else if (t == "affine:bestfit" || t == "a:b") m = c4m_affine_create(C4M_AFFINE_BESTFIT, qa, ta, params);
    else if (t == "affine:local" || t == "a:l") m = c4m_affine_create(C4M_AFFINE_LOCAL, qa, ta, params);
    else if (t == "affine:overlap" || t == "a:o") m = c4m_affine_create(C4M_AFFINE_OVERLAP, qa, ta, params);
    else if (t == "est2genome" || t == "e2g") m = c4m_est2genome_create(params);
    else if (t == "protein2dna" || t == "p2d") m = c4m_protein2dna_create(C4M_AFFINE_LOCAL, params);
    else if (t == "protein2dna:bestfit" || t == "p2d:b") m = c4m_protein2dna_create(C4M_AFFINE_BESTFIT, params);
    else if (t == "protein2genome" || t == "p2g") m = c4m_protein2genome_create(C4M_AFFINE_LOCAL, params);
    else if (t == "protein2genome:bestfit" || t == "p2g:b") m = c4m_protein2genome_create(C4M_AFFINE_BESTFIT, params);
    return m;
}

// Viterbi_create with use_continuation (src/c4/viterbi.c:68-76): C4_Model_copy + CORNER/CORNER.
// C4_Model_copy (c4.c:1713) keeps transition ids (no re-sort), so the tables are unchanged.
void c4gpu_model_make_continuation(const c4gpu_model *model, c4gpu_model *out) {
    if (out != model) *out = *model;
    out->start_scope = C4GPU_SCOPE_CORNER;
    out->end_scope = C4GPU_SCOPE_CORNER;
}

// Codegen_clean_path_component (codegen.c:39-55) of "optimal:<model> find <what>" (optimal.c:31-67)
int c4gpu_model_plugin_name(const c4gpu_model *model, int mode, int use_continuation, char *buf, size_t len) {
    const char *what = "score";
    if (mode == C4GPU_MODE_FIND_PATH) what = use_continuation ? "path continuation" : "path";
    else if (mode == C4GPU_MODE_FIND_REGION) what = "region";
    else if (mode == C4GPU_MODE_FIND_CHECKPOINTS) what = "checkpoint";
    std::string raw = std::string("optimal:") + model->name + " find " + what, clean;
    for (unsigned char c : raw) {
        if (isalnum(c) || c == '_') clean += (char)c;
        else { char tmp[16]; snprintf(tmp, sizeof tmp, "_%d_", (int)c); clean += tmp; }
    }
    snprintf(buf, len, "%s", clean.c_str());
    return (int)clean.size();
}

int c4gpu_model_is_accelerated(const c4gpu_model *model) {
    for (int c = 0; c < model->n_calcs; c++)
        if (model->calcs[c].kind > C4GPU_CALC_PHASE_POST) return 0;
    for (int s = 0; s < model->n_shadows; s++)
        if (!model->shadows[s].on_target) return 0;
    if (model->total_shadow_designations > 1) return 0;
    return 1;
}

}  // extern "C"
