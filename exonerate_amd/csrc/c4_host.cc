// c4_host.cc — host-side pieces of the C ABI that need no device: the reference's memory decisions
// (which decide WHICH Viterbi passes run, hence results) and the sugar/cigar/vulgar printers.
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "c4gpu.h"
#include "c4_internal.h"
#include "c4_memrule.h"

namespace c4h {

// the reference's memory rule itself is in c4_memrule.h (shared with the device code, which lists the sub-alignment
// jobs of a checkpoint pass without the host)
static MemRule rule_of(const c4gpu_model *m) {
    return MemRule{m->max_query_advance, m->max_target_advance, m->n_states, m->total_shadow_designations};
}

bool use_reduced_space(const c4gpu_model *m, const c4gpu_region *r, int dpmemory_mb) {
    return use_reduced_space(rule_of(m), r->query_length, r->target_length, dpmemory_mb);
}

int checkpoint_rows(const c4gpu_model *m, const c4gpu_region *r, int dpmemory_mb) {
    return checkpoint_rows(rule_of(m), r->query_length, r->target_length, dpmemory_mb);
}

// Alignment_add (alignment.c:75-102): run-length merge of equal consecutive transitions
void alignment_add(c4gpu_alignment *a, int *cap, int transition, int length) {
    if (a->n_ops && a->op_transition[a->n_ops - 1] == transition) {
        a->op_length[a->n_ops - 1] += length;
        if (a->op_length[a->n_ops - 1] == 0) a->n_ops--;
        return;
    }
    if (a->n_ops == *cap) {
        *cap = *cap ? *cap * 2 : 32;
        a->op_transition = (int32_t *)realloc(a->op_transition, sizeof(int32_t) * *cap);
        a->op_length = (int32_t *)realloc(a->op_length, sizeof(int32_t) * *cap);
    }
    a->op_transition[a->n_ops] = transition;
    a->op_length[a->n_ops++] = length;
}

void subopt_region_points(const c4gpu_subopt *so, const c4gpu_region &r,
                          std::vector<std::pair<int32_t, int32_t>> &out) {
    out.clear();
    if (!so) return;
    auto it = std::lower_bound(so->points.begin(), so->points.end(), std::make_pair(r.target_start, INT32_MIN));
    for (; it != so->points.end() && it->first <= r.target_start + r.target_length; ++it)
        if (it->second >= r.query_start && it->second <= r.query_start + r.query_length)
            out.emplace_back(it->first - r.target_start, it->second - r.query_start);
}

}  // namespace c4h

void c4gpu_subopt::merge(std::vector<std::pair<int32_t, int32_t>> &fresh) {
    if (fresh.empty()) return;
    if (!std::is_sorted(fresh.begin(), fresh.end())) std::sort(fresh.begin(), fresh.end());
    const size_t old = points.size();
    points.insert(points.end(), fresh.begin(), fresh.end());
    std::inplace_merge(points.begin(), points.begin() + old, points.end());
    points.erase(std::unique(points.begin(), points.end()), points.end());
}

extern "C" {

c4gpu_subopt *c4gpu_subopt_create(int32_t query_length, int32_t target_length) {     // subopt.c:24-33
    c4gpu_subopt *so = new c4gpu_subopt;
    so->query_length = query_length; so->target_length = target_length; so->path_count = 0;
    return so;
}
void c4gpu_subopt_destroy(c4gpu_subopt *so) { delete so; }

int c4gpu_subopt_add_point(c4gpu_subopt *so, int32_t query_pos, int32_t target_pos) {
    if (!so) return -1;
    std::vector<std::pair<int32_t, int32_t>> one(1, std::make_pair(target_pos, query_pos));
    so->merge(one);
    return 0;
}

int32_t c4gpu_subopt_points(const c4gpu_subopt *so, int32_t *query_pos, int32_t *target_pos, int32_t max) {
    int32_t k = 0;
    for (const auto &p : so->points) {
        if (k >= max) break;
        query_pos[k] = p.second; target_pos[k] = p.first; k++;
    }
    return (int32_t)so->points.size();
}

// SubOpt_add_alignment (subopt.c:131-148) over SubOpt_add_AlignmentOperation (subopt.c:64-128).  A match
// operation (aq, at) of length n starting at (q, t) blocks, per step, the cell the step leaves from and
// the intermediate cells of a multi-residue step (stride (aq, at) / gcd); then the lead-in cells in front
// of its first step, when they are inside the sequences.  The reference tests membership before every
// insertion; a set needs no such test.
int c4gpu_subopt_add_alignment(c4gpu_subopt *so, const c4gpu_model *model, const c4gpu_alignment *a) {
    if (!so || !model || !a) return -1;
    int32_t q = a->region.query_start, t = a->region.target_start;
    std::vector<std::pair<int32_t, int32_t>> fresh;           // an alignment's cells come out in ascending order
    for (int32_t k = 0; k < a->n_ops; k++) {
        const int tr = a->op_transition[k];
        if (tr < 0 || tr >= model->n_transitions) { c4h::set_error("alignment operation outside the model"); return -1; }
        const c4gpu_transition &x = model->transitions[tr];
        const int32_t len = a->op_length[k];
        if (x.label == C4GPU_LABEL_MATCH) {
            int g = x.advance_query, h = x.advance_target;
            while (h) { const int r = g % h; g = h; h = r; }
            if (g > 0) {
                const int dq = x.advance_query / g, dt = x.advance_target / g;
                for (int32_t step = 0; step < len; step++)
                    for (int sub = 0; sub * dq < x.advance_query; sub++)
                        fresh.emplace_back(t + step * x.advance_target + sub * dt, q + step * x.advance_query + sub * dq);
                for (int sub = 1; sub * dq < x.advance_query; sub++) {
                    const int32_t lq = q - x.advance_query + sub * dq, lt = t - x.advance_target + sub * dt;
                    if (lq >= 0 && lt >= 0) fresh.emplace_back(lt, lq);
                }
            }
        }
        q += x.advance_query * len;
        t += x.advance_target * len;
    }
    so->merge(fresh);
    so->path_count++;
    return 0;
}

int c4gpu_use_reduced_space(const c4gpu_model *model, const c4gpu_region *region, int dpmemory_mb) {
    return c4h::use_reduced_space(model, region, dpmemory_mb) ? 1 : 0;
}
int c4gpu_checkpoint_rows(const c4gpu_model *model, const c4gpu_region *region, int dpmemory_mb) {
    return c4h::checkpoint_rows(model, region, dpmemory_mb);
}

void c4gpu_alignment_clear(c4gpu_alignment *a) {
    free(a->op_transition);
    free(a->op_length);
    memset(a, 0, sizeof(*a));
}

// Alignment_display_{sugar,cigar,vulgar} (alignment.c:2671-2706) on top of the *_block printers
// (alignment.c:1622-1779).
int c4gpu_alignment_format(const c4gpu_model *m, const c4gpu_alignment *a, int what, const char *qid,
                           int32_t qlen, char qstrand, const char *tid, int32_t tlen, char tstrand,
                           int forward_coords, char *buf, size_t buf_len) {
    if (what < 0 || what > 2 || !buf_len) return -1;
    auto coord = [&](bool on_query, bool start) {      // Alignment_get_coordinate, alignment.c:177-205
        int pos = on_query ? (start ? a->region.query_start : a->region.query_start + a->region.query_length)
                           : (start ? a->region.target_start : a->region.target_start + a->region.target_length);
        if (forward_coords && (on_query ? qstrand : tstrand) == '-') pos = (on_query ? qlen : tlen) - pos;
        return pos;
    };
    static const char *prefix[] = {"sugar: ", "cigar: ", "vulgar: "};
    std::string out = prefix[what];
    char tmp[128];
    out += qid;
    snprintf(tmp, sizeof tmp, " %d %d %c ", coord(true, true), coord(true, false), qstrand);
    out += tmp;
    out += tid;
    snprintf(tmp, sizeof tmp, " %d %d %c %d", coord(false, true), coord(false, false), tstrand, a->score);
    out += tmp;
    if (what == 1 && a->n_ops > 0) {
        out += " ";
        const char *gap = "";
        char type = 0;
        int move = 0;
        for (int i = 0; i < a->n_ops; i++) {
            const c4gpu_transition &t = m->transitions[a->op_transition[i]];
            char ntype;
            int nmove;
            if (!t.advance_query) { ntype = 'D'; nmove = t.advance_target * a->op_length[i]; }
            else if (!t.advance_target) { ntype = 'I'; nmove = t.advance_query * a->op_length[i]; }
            else { ntype = 'M'; nmove = (t.advance_query > t.advance_target ? t.advance_query : t.advance_target) * a->op_length[i]; }
            if (i == 0) { type = ntype; move = nmove; continue; }
            if (ntype == type) { move += nmove; continue; }
            if (move) { snprintf(tmp, sizeof tmp, "%s%c %d", gap, type, move); out += tmp; }
            move = nmove; type = ntype; gap = " ";
        }
        if (move) { snprintf(tmp, sizeof tmp, "%s%c %d", gap, type, move); out += tmp; }
    } else if (what == 2 && a->n_ops > 0) {
        out += " ";
        static const char label_char[] = {0, 'M', 'G', 'N', '5', '3', 'I', 'S', 'F'};
        const char *gap = "";
        const c4gpu_transition *t = &m->transitions[a->op_transition[0]];
        int label = t->label, aq = t->advance_query * a->op_length[0], at = t->advance_target * a->op_length[0];
        bool codon = false;
        for (int i = 1; i < a->n_ops; i++) {
            t = &m->transitions[a->op_transition[i]];
            const bool tcodon = t->advance_query == 3 && t->advance_target == 3;
            if (t->label == label && (aq || !t->advance_query) && (at || !t->advance_target) && codon == tcodon) {
                aq += t->advance_query * a->op_length[i];
                at += t->advance_target * a->op_length[i];
                continue;
            }
            if (label != C4GPU_LABEL_NONE) {
                char c = (label == C4GPU_LABEL_MATCH && codon) ? 'C' : label_char[label];
                snprintf(tmp, sizeof tmp, "%s%c %d %d", gap, c, aq, at);
                out += tmp;
                gap = " ";
            }
            label = t->label; codon = tcodon;
            aq = t->advance_query * a->op_length[i];
            at = t->advance_target * a->op_length[i];
        }
        // the run still pending here is never printed (alignment.c:1697-1777 has no epilogue)
    } else if (what != 0) {
        out += " ";
    }
    if (out.size() + 1 > buf_len) return -1;
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

}  // extern "C"
