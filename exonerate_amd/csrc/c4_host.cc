// c4_host.cc — host-side pieces of the C ABI that need no device: the reference's memory decisions
// (which decide WHICH Viterbi passes run, hence results) and the sugar/cigar/vulgar printers.
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <cctype>
#include <ctime>
#include <cstdarg>

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

// ---- Alignment_display_gff (alignment.c:2710-3236) ----------------------------------------------------------------
namespace {

struct GffCtx {
    const c4gpu_model *m;
    const c4gpu_params *p;
    const c4gpu_alignment *a;
    const c4gpu_gff_request *r;
    std::string out;
    int match_kind = -1;                                   // C4GPU_CALC_MATCH_* of the model's MATCH transitions

    const c4gpu_transition &tr(int i) const { return m->transitions[a->op_transition[i]]; }
    // Alignment_match_get_symbol (alignment.c:104-124)
    char symbol(bool on_query, int pos, int advance) const {
        const uint8_t *s = on_query ? r->query : r->target;
        if (advance == 1) return (char)s[pos];
        return (char)p->aa[p->trans[p->nt2d[s[pos]] | (p->nt2d[s[pos + 1]] << 4) | (p->nt2d[s[pos + 2]] << 8)]];   // Translate_base
    }
    // C4_Calc_score of a MATCH transition's calc (match.c:271,287,347)
    int match_score(int qpos, int tpos) const {
        const uint8_t q = r->query[qpos];
        if (match_kind == C4GPU_CALC_MATCH_DNA) return p->dna_submat[p->submat_index[q]][p->submat_index[r->target[tpos]]];
        if (match_kind == C4GPU_CALC_MATCH_PROTEIN) return p->protein_submat[p->submat_index[q]][p->submat_index[r->target[tpos]]];
        const uint8_t aa = (uint8_t)symbol(false, tpos, 3);
        return p->protein_submat[p->submat_index[q]][p->submat_index[aa]];
    }
    // Alignment_get_equivalenced_matching[_region] / _total[_region] (alignment.c:1382-1522); region < 0: whole alignment
    void equivalenced(bool report_id, int exon_query_start, int exon_query_end, bool region, int *match_out, int *total_out) const {
        int match = 0, total = 0, qp = a->region.query_start, tp = a->region.target_start;
        for (int i = 0; i < a->n_ops; i++) {
            const c4gpu_transition &t = tr(i);
            if (t.label == C4GPU_LABEL_MATCH) {
                for (int j = 0; j < a->op_length[i]; j++) {
                    if (region && qp > exon_query_end) { *match_out = match; *total_out = total; return; }
                    if (!region || qp >= exon_query_start) {
                        total++;
                        if (report_id) {
                            const int qs = toupper((unsigned char)symbol(true, qp, t.advance_query));
                            const int ts = toupper((unsigned char)symbol(false, tp, t.advance_target));
                            if (qs == ts) match++;
                        } else if (match_score(qp, tp) > 0) match++;
                    }
                    qp += t.advance_query; tp += t.advance_target;
                }
            } else {
                qp += t.advance_query * a->op_length[i];
                tp += t.advance_target * a->op_length[i];
            }
        }
        *match_out = match; *total_out = total;
    }
    // Alignment_get_percent_score[_region] (alignment.c:1524-1561): gfloat arithmetic
    float percent(bool report_id, bool region, int exon_query_start, int exon_query_end) const {
        int mt, tt;
        equivalenced(report_id, exon_query_start, exon_query_end, region, &mt, &tt);
        if (region) {
            // the two region passes stop at different operations (the total's pass returns when it runs past the end, the
            // matching pass likewise): both counted above in one pass with the same stop rule
        }
        return (((float)mt) / ((float)tt)) * 100;
    }
    void printf_out(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        // two passes: sequence identifiers of any length (the reference prints through g_strdup_printf)
        va_list ap, ap2;
        va_start(ap, fmt);
        va_copy(ap2, ap);
        const int n = vsnprintf(nullptr, 0, fmt, ap);
        va_end(ap);
        std::string tmp((size_t)(n < 0 ? 0 : n) + 1, '\0');
        vsnprintf(&tmp[0], tmp.size(), fmt, ap2);
        va_end(ap2);
        tmp.resize((size_t)(n < 0 ? 0 : n));
        out += tmp;
    }
    // Alignment_display_gff_line (alignment.c:2732-2794)
    void line(const char *feature, int query_start, int target_start, int query_end, int target_end, bool show_score,
              int score, const std::vector<std::string> *attributes) {
        const bool on_query = r->report_on_query != 0;
        const char *id = on_query ? r->query_id : r->target_id;
        const int len = on_query ? r->query_len : r->target_len;
        const char strand = on_query ? r->query_strand : r->target_strand;
        int start = on_query ? query_start : target_start, end = on_query ? query_end : target_end;
        if (strand == '-') { const int s2 = len - end, e2 = len - start; start = s2; end = e2; }
        out += id;
        printf_out("\t%s:%s\t%s\t%d\t%d\t", "exonerate", m->name, feature, start + 1, end);
        if (show_score) printf_out("%d", score); else out += ".";
        printf_out("\t%c\t", strand);
        out += ".";                                        // no caller shows a frame
        out += "\t";
        if (attributes)
            for (size_t i = 0; i < attributes->size(); i++) {
                out += (*attributes)[i];
                if (i + 1 < attributes->size()) out += " ; ";
            }
        out += "\n";
    }
    static std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2))) {
        va_list ap, ap2;
        va_start(ap, f);
        va_copy(ap2, ap);
        const int n = vsnprintf(nullptr, 0, f, ap);
        va_end(ap);
        std::string tmp((size_t)(n < 0 ? 0 : n) + 1, '\0');
        vsnprintf(&tmp[0], tmp.size(), f, ap2);
        va_end(ap2);
        tmp.resize((size_t)(n < 0 ? 0 : n));
        return tmp;
    }
    // Alignment_display_gff_exon (alignment.c:2804-2858)
    void exon(int query_pos, int target_pos, int eqs, int ets, int eqgap, int etgap, int eqfs, int etfs) {
        const bool on_query = r->report_on_query != 0;
        std::vector<std::string> at;
        at.push_back(fmt("insertions %d", on_query ? eqgap : etgap));
        at.push_back(fmt("deletions %d", on_query ? etgap : eqgap));
        at.push_back(fmt("identity %2.2f", percent(true, true, eqs, query_pos)));
        at.push_back(fmt("similarity %2.2f", percent(false, true, eqs, query_pos)));
        if (on_query) {
            if (eqfs) at.push_back(fmt("frameshifts %d", eqfs));
        } else {
            at.push_back(fmt("Target %s %d %d", r->query_id, eqs + 1, query_pos));
            if (etfs) at.push_back(fmt("frameshifts %d", etfs));
        }
        line("exon", eqs, ets, query_pos, target_pos, false, 0, &at);
    }
    // Alignment_display_gff_utr (alignment.c:2860-2895)
    void utr(bool post_cds, int cqs, int cts, int cqe, int cte, int eqs, int ets, int query_pos, int target_pos) {
        if (post_cds) line("utr3", std::max(eqs, cqe), std::max(ets, cte), query_pos, target_pos, false, 0, nullptr);
        else if (cqs == -1) line("utr5", eqs, ets, query_pos, target_pos, false, 0, nullptr);
        else line("cds", std::max(cqs, eqs), std::max(cts, ets), query_pos, target_pos, false, 0, nullptr);
    }
    char report_symbol(int query_pos, int target_pos, int k) const {
        return (char)(r->report_on_query ? r->query[query_pos + k] : r->target[target_pos + k]);
    }
    // Alignment_display_gff_gene (alignment.c:2897-3142)
    int gene() {
        int query_pos = a->region.query_start, target_pos = a->region.target_start;
        int intron_id = 0, intron_length = 0, eqs = 0, ets = 0, eqgap = 0, etgap = 0, eqfs = 0, etfs = 0;
        int cqs = -1, cts = -1, cqe = -1, cte = -1;
        bool in_exon = false, post_cds = false;
        char orientation = '.';                            // Alignment_get_gene_orientation (alignment.c:164-175)
        for (int i = 0; i < a->n_ops; i++) {
            if (tr(i).label == C4GPU_LABEL_5SS) { orientation = '+'; break; }
            if (tr(i).label == C4GPU_LABEL_3SS) { orientation = '-'; break; }
        }
        {
            std::vector<std::string> at;
            at.push_back(fmt("gene_id %d", r->result_id));
            at.push_back(fmt("sequence %s", r->report_on_query ? r->target_id : r->query_id));
            at.push_back(fmt("gene_orientation %c", orientation));
            at.push_back(fmt("identity %2.2f", percent(true, false, 0, 0)));
            at.push_back(fmt("similarity %2.2f", percent(false, false, 0, 0)));
            line("gene", a->region.query_start, a->region.target_start, a->region.query_start + a->region.query_length,
                 a->region.target_start + a->region.target_length, true, a->score, &at);
        }
        for (int i = 1; i < a->n_ops; i++) {
            const c4gpu_transition &t = tr(i);
            const int len = a->op_length[i];
            switch (t.label) {
                case C4GPU_LABEL_MATCH:
                    if (t.advance_query == 1 && t.advance_target == 1) {
                        if (cqs != -1 && !post_cds) {
                            line("cds", eqs, ets, query_pos, target_pos, false, 0, nullptr);
                            post_cds = true;
                        }
                    } else {
                        if (cqs == -1) {                   // first coding exon
                            if (in_exon) line("utr5", eqs, ets, query_pos, target_pos, false, 0, nullptr);
                            cqs = query_pos; cts = target_pos;
                        }
                        cqe = query_pos + t.advance_query * len;
                        cte = target_pos + t.advance_target * len;
                    }
                    /* fallthrough */
                case C4GPU_LABEL_SPLIT_CODON:
                    if (!in_exon) {
                        eqs = query_pos; ets = target_pos;
                        eqgap = etgap = eqfs = etfs = 0;
                        in_exon = true;
                    }
                    break;
                case C4GPU_LABEL_NONE:
                    break;
                case C4GPU_LABEL_GAP:
                    eqgap += t.advance_query * len;
                    etgap += t.advance_target * len;
                    break;
                case C4GPU_LABEL_5SS:
                case C4GPU_LABEL_3SS: {
                    if (in_exon) {
                        utr(post_cds, cqs, cts, cqe, cte, eqs, ets, query_pos, target_pos);
                        exon(query_pos, target_pos, eqs, ets, eqgap, etgap, eqfs, etfs);
                        in_exon = false;
                    }
                    std::vector<std::string> at;
                    if (t.label == C4GPU_LABEL_5SS) {
                        at.push_back(fmt("intron_id %d", intron_id + 1));
                        at.push_back(fmt("splice_site \"%c%c\"", report_symbol(query_pos, target_pos, 0), report_symbol(query_pos, target_pos, 1)));
                        line("splice5", query_pos, target_pos, query_pos + 2, target_pos + 2, false, 0, &at);
                    } else {
                        if (orientation == '+') {
                            std::vector<std::string> ia;
                            ia.push_back(fmt("intron_id %d", ++intron_id));
                            line("intron", query_pos - intron_length - 2, target_pos - intron_length - 2, query_pos + 2,
                                 target_pos + 2, false, 0, &ia);
                        }
                        at.push_back(fmt("intron_id %d", intron_id - 1));
                        at.push_back(fmt("splice_site \"%c%c\"", report_symbol(query_pos, target_pos, 0), report_symbol(query_pos, target_pos, 1)));
                        line("splice3", query_pos, target_pos, query_pos + 2, target_pos + 2, false, 0, &at);
                    }
                    intron_length = 0;
                    break;
                }
                case C4GPU_LABEL_INTRON:
                    intron_length += len;
                    break;
                case C4GPU_LABEL_FRAMESHIFT:
                    eqfs += t.advance_query * len;
                    etfs += t.advance_target * len;
                    break;
                default:
                    return -1;                             // NER: "Unexpected NER for gff gene output"
            }
            query_pos += t.advance_query * len;
            target_pos += t.advance_target * len;
        }
        if (in_exon) {
            if (cqe != -1) {
                if (cqe != query_pos) line("utr3b", std::max(eqs, cqe), std::max(ets, cte), query_pos, target_pos, false, 0, nullptr);
                else line("cds", eqs, ets, query_pos, target_pos, false, 0, nullptr);
            }
            exon(query_pos, target_pos, eqs, ets, eqgap, etgap, eqfs, etfs);
        }
        return 0;
    }
    // Alignment_display_gff_similarity (alignment.c:3144-3208)
    void similarity() {
        int query_pos = a->region.query_start, target_pos = a->region.target_start;
        std::vector<std::string> at;
        at.push_back(fmt("alignment_id %d", r->result_id));
        at.push_back(r->report_on_query ? fmt("Target %s", r->target_id) : fmt("Query %s", r->query_id));
        for (int i = 1; i < a->n_ops; i++) {
            const c4gpu_transition &t = tr(i);
            if (t.label == C4GPU_LABEL_MATCH) {
                int qp = query_pos, tp = target_pos;
                if (r->query_strand == '-') qp = r->query_len - qp;
                if (r->target_strand == '-') tp = r->target_len - tp;
                if (r->report_on_query) at.push_back(fmt("Align %d %d %d", qp + 1, tp + 1, a->op_length[i] * t.advance_query));
                else at.push_back(fmt("Align %d %d %d", tp + 1, qp + 1, a->op_length[i] * t.advance_target));
            }
            query_pos += t.advance_query * a->op_length[i];
            target_pos += t.advance_target * a->op_length[i];
        }
        line("similarity", a->region.query_start, a->region.target_start, a->region.query_start + a->region.query_length,
             a->region.target_start + a->region.target_length, true, a->score, &at);
    }
};

}  // namespace

extern "C" int c4gpu_alignment_format_gff(const c4gpu_model *m, const c4gpu_params *p, const c4gpu_alignment *a,
                                          const c4gpu_gff_request *req, char *buf, size_t buf_len) {
    GffCtx g;
    g.m = m; g.p = p; g.a = a; g.r = req;
    bool query_protein = false, target_protein = false;
    for (int k = 0; k < m->n_transitions; k++) {
        const c4gpu_transition &t = m->transitions[k];
        if (t.label != C4GPU_LABEL_MATCH || t.calc < 0) continue;
        const int kind = m->calcs[t.calc].kind;
        if (kind < C4GPU_CALC_MATCH_DNA || kind > C4GPU_CALC_MATCH_P2D) return INT32_MIN;
        g.match_kind = kind;
    }
    if (g.match_kind < 0) return INT32_MIN;
    query_protein = g.match_kind != C4GPU_CALC_MATCH_DNA;
    target_protein = g.match_kind == C4GPU_CALC_MATCH_PROTEIN;
    char today[16];
    const char *date = req->date;
    if (!date) {
        time_t now = time(nullptr);
        strftime(today, sizeof today, "%Y-%m-%d", localtime(&now));
        date = today;
    }
    g.out = "# --- START OF GFF DUMP ---\n#\n";
    // Alignment_display_gff_header (alignment.c:2710-2730)
    g.printf_out("#\n##gff-version 2\n##source-version %s:%s %s\n##date %s\n##type %s\n#\n", "exonerate", m->name,
                 req->version ? req->version : "2.4.0", date,
                 (req->report_on_query ? query_protein : target_protein) ? "Protein" : "DNA");
    g.out += "#\n# seqname source feature start end score strand frame attributes\n#\n";
    if (req->report_on_genomic && g.gene()) return INT32_MIN;
    g.similarity();
    g.out += "# --- END OF GFF DUMP ---\n#\n";
    if (g.out.size() + 1 > buf_len) return -(int)(g.out.size() + 1);
    memcpy(buf, g.out.c_str(), g.out.size() + 1);
    return (int)g.out.size();
}

// ---- Alignment_display (alignment.c:234-1380) ---------------------------------------------------------------------
namespace {

struct ViewCtx {
    const c4gpu_model *m;
    const c4gpu_params *p;
    const c4gpu_alignment *a;
    const c4gpu_display_request *r;
    int match_kind = -1;
    bool q_prot = false, t_prot = false;
    // AlignmentView (alignment.c:260-283)
    std::string oq, iq, mid, it, ot;
    bool has_iq = false, has_it = false;
    std::vector<std::pair<int, int>> row_marker;
    int max_pos_len = 0, width = 0, limit = 0;
    int query_intron_count = 0, target_intron_count = 0, joint_intron_count = 0, intron_aq = 0, intron_at = 0;
    char orientation = '.';
    int split_count = 0;
    std::vector<std::pair<int, int>> split_sep;           // (query, target) separation of each split-codon pair
    bool failed = false;

    const c4gpu_transition &tr(int i) const { return m->transitions[a->op_transition[i]]; }
    char qsym(int pos) const { return (char)r->query[pos]; }
    char tsym(int pos) const { return (char)r->target[pos]; }
    int submat(bool protein, char x, char y) const {      // Submat_lookup (submat.h:54-56)
        const int i = p->submat_index[(unsigned char)x], j = p->submat_index[(unsigned char)y];
        return protein ? p->protein_submat[i][j] : p->dna_submat[i][j];
    }
    char translate3(const char *c) const {                // Translate_codon / Translate_base (translate.h:70-78)
        return (char)p->aa[p->trans[p->nt2d[(unsigned char)c[0]] | (p->nt2d[(unsigned char)c[1]] << 4) | (p->nt2d[(unsigned char)c[2]] << 8)]];
    }
    const char *tla(char aa) {                            // Alphabet_aa2tla (alphabet.c:330-375)
        static const char *tla_names[25] = {"Ala", "Arg", "Asn", "Asp", "Cys", "Gln", "Glu", "Gly", "His", "Ile", "Leu", "Lys", "Met",
                                            "Phe", "Pro", "Ser", "Thr", "Trp", "Tyr", "Val", "Asx", "Zed", "Unk", "***", "Sec"};
        static const char *short_names[25] = {"^A^", "^R^", "^N^", "^D^", "^C^", "^Q^", "^E^", "^G^", "^H^", "^I^", "^L^", "^K^", "^M^",
                                              "^F^", "^P^", "^S^", "^T^", "^W^", "^Y^", "^V^", "^B^", "^Z^", "^X^", "^*^", "^U^"};
        static const char *letters = "ARNDCQEGHILKMFPSTWYVBZX*U";
        const char *at = aa ? strchr(letters, toupper((unsigned char)aa)) : nullptr;
        if (!at) { failed = true; return "???"; }         // "Unknown amino acid"
        return (r->use_aa_tla ? tla_names : short_names)[at - letters];
    }
    // Match_get_display_symbol (match.c:224-236)
    char display_symbol(bool protein, char q, char t) const {
        if (toupper((unsigned char)q) == toupper((unsigned char)t)) return '|';
        const int score = submat(protein, q, t);
        if (score == 0) return '.';
        if (score > 0) return ':';
        return ' ';
    }
    // Alignment_get_equiv_symbol (alignment.c:431-451) with a substitution matrix
    char equiv_symbol(char x, char y) const {
        const int score = submat(true, x, y);
        if (score == 0) return '.';
        if (score > 0) return toupper((unsigned char)x) == toupper((unsigned char)y) ? '|' : ':';
        return ' ';
    }
    // Translate_reverse over one residue (translate.c:296-343): '!' wherever some codon of `aa` has the base `codon` has there
    void reverse_translate_marks(char aa, const char *codon, char *marks) const {
        static const char bases[] = "ACGT";
        bool any = false;
        for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) for (int z = 0; z < 4; z++) {
            const char c[3] = {bases[x], bases[y], bases[z]};
            if (translate3(c) != aa) continue;
            any = true;
            for (int k = 0; k < 3; k++) if (c[k] == codon[k]) marks[k] = '!';
        }
        if (!any) for (int k = 0; k < 3; k++) if (codon[k] == 'N') marks[k] = '!';    // no codon list: "NNN"
    }
    // Match_1_3_split_display_func (match.c:385-417)
    void p2d_display(char q, const char *codon, char *out3) const {
        const char t = translate3(codon);
        const char d = display_symbol(true, q, t);
        out3[0] = out3[1] = out3[2] = d; out3[3] = 0;
        if (q != t) reverse_translate_marks(q, codon, out3);
    }
    // Alignment_get_codon_match_string (alignment.c:453-474)
    void codon_match_string(const char *codon, char aa, char *out3) const {
        const char codon_aa = translate3(codon);
        const char s = equiv_symbol(codon_aa, aa);
        out3[0] = out3[1] = out3[2] = s; out3[3] = 0;
        if (s != '|') reverse_translate_marks(aa, codon, out3);
    }
    // AlignmentView_add (alignment.c:372-414)
    void add(const char *qs, const char *iqs, const char *ms, const char *its, const char *ts, int qpos, int tpos) {
        const size_t n = strlen(ms);
        if (has_iq) { if (iqs) iq += iqs; else iq.append(n, ' '); }
        if (has_it) { if (its) it += its; else it.append(n, ' '); }
        oq += qs; mid += ms; ot += ts;
        if ((int)oq.size() >= limit) {
            row_marker.emplace_back(qpos, tpos);
            limit += width;
        }
    }
    // Alignment_match_get_string (alignment.c:126-160)
    std::string match_string(bool on_query, int pos, int advance, int max) {
        const uint8_t *s = on_query ? r->query : r->target;
        if (max == 1) return std::string(1, (char)s[pos]);
        if (advance == 1) return tla((char)s[pos]);
        return std::string(reinterpret_cast<const char *>(s) + pos, 3);
    }
    void add_match(const c4gpu_transition &t, int total, int qpos, int tpos) {          // alignment.c:476-533
        for (int i = 0; i < total; i++) {
            const int max_adv = std::max(t.advance_query, t.advance_target);
            const std::string qs = match_string(true, qpos, t.advance_query, max_adv);
            const std::string ts = match_string(false, tpos, t.advance_target, max_adv);
            const char *iqs = nullptr, *its = nullptr;
            char ms[4];
            if (t.advance_target == 3) {
                const char codon[3] = {tsym(tpos), tsym(tpos + 1), tsym(tpos + 2)};
                its = tla(translate3(codon));
                p2d_display(qsym(qpos), codon, ms);                                     // Match_1_3_display_func
            } else {
                ms[0] = display_symbol(match_kind != C4GPU_CALC_MATCH_DNA, qsym(qpos), tsym(tpos));   // Match_1_1_display_func
                ms[1] = 0;
            }
            add(qs.c_str(), iqs, ms, its, ts.c_str(), qpos, tpos);
            qpos += t.advance_query; tpos += t.advance_target;
        }
    }
    void add_gap(int aq, int at, int total, int qpos, int tpos) {                        // alignment.c:535-608
        const bool translating = (q_prot != t_prot) || ((aq | at) == 3);
        const bool emitted_protein = aq ? q_prot : t_prot;
        for (int i = 0; i < total; i++) {
            char seq[4] = {0, 0, 0, 0}, ms[4] = {0, 0, 0, 0}, gap[4] = {0, 0, 0, 0};
            const int n = aq | at;
            for (int j = 0; j < n; j++) { seq[j] = aq ? qsym(qpos + j) : tsym(tpos + j); ms[j] = ' '; gap[j] = '-'; }
            const char *codon_name = nullptr;
            if (translating) {
                if (emitted_protein) {
                    memcpy(seq, tla(seq[0]), 3);
                    ms[0] = ms[1] = ms[2] = ' '; gap[0] = gap[1] = gap[2] = '-';
                    seq[3] = ms[3] = gap[3] = 0;
                }
                if (n == 3) {
                    gap[0] = '<'; gap[1] = '-'; gap[2] = '>'; gap[3] = 0;
                    codon_name = tla(translate3(seq));
                }
            }
            if (aq) add(seq, codon_name, ms, translating ? gap : nullptr, gap, qpos, tpos);
            else add(gap, translating ? gap : nullptr, ms, codon_name, seq, qpos, tpos);
            qpos += aq; tpos += at;
        }
    }
    void consensus(bool is_5_prime, const char *site, char *cons) const {                // alignment.c:612-642
        char ca, cb;
        if (orientation == '+') { ca = is_5_prime ? 'G' : 'A'; cb = is_5_prime ? 'T' : 'G'; }
        else { ca = is_5_prime ? 'A' : 'C'; cb = is_5_prime ? 'C' : 'T'; }
        cons[0] = toupper((unsigned char)site[0]) == ca ? '+' : '-';
        cons[1] = toupper((unsigned char)site[1]) == cb ? '+' : '-';
    }
    void add_splice_site(int aq, int at, int qpos, int tpos, bool is_5_prime, const c4gpu_transition *last_match) {   // :644-703
        if (aq != 0 || at != 2 || !last_match) { failed = true; return; }               // target introns only (the accelerated models)
        char seq[3] = {tsym(tpos), tsym(tpos + 1), 0}, cons[3] = {' ', ' ', 0};
        consensus(is_5_prime, seq, cons);
        seq[0] = (char)tolower((unsigned char)seq[0]); seq[1] = (char)tolower((unsigned char)seq[1]);
        if (last_match->advance_target == 3) add("  ", "  ", "  ", cons, seq, qpos, tpos);
        else add("  ", nullptr, cons, nullptr, seq, qpos, tpos);
    }
    void add_intron(int aq, int at, int qpos, int tpos, const c4gpu_transition *last_match) {                        // :705-772
        if (aq != 0 || !last_match) { failed = true; return; }
        const char *dir = orientation == '+' ? ">>>>" : orientation == '-' ? "<<<<" : "????";
        char label[64], name[128];
        const int count = ++target_intron_count;
        snprintf(label, sizeof label, "%d bp", at + 4);
        snprintf(name, sizeof name, "%s %s Intron %d %s", dir, "Target", count, dir);
        const int fill = (int)(strlen(name) - strlen(label)) + 1;
        char middle[256];
        snprintf(middle, sizeof middle, "%*c%s%*c", ((fill | 1) >> 1), ' ', label, ((fill - 1) >> 1), ' ');
        const std::string gap(strlen(name), '.'), pad(strlen(name), '^');
        if (last_match->advance_target == 3) add(name, pad.c_str(), middle, pad.c_str(), gap.c_str(), qpos, tpos);
        else add(name, nullptr, middle, nullptr, gap.c_str(), qpos, tpos);
    }
    void add_split_codon(int aq, int at, int qpos, int tpos) {                            // alignment.c:817-1038, the p,d branch
        if (!q_prot || t_prot || (size_t)(split_count >> 1) >= split_sep.size()) { failed = true; return; }
        const int sep = split_sep[split_count >> 1].second;
        int start = -1, tp0 = 0, tp1 = 0, tp2 = 0;
        const char q_aa = qsym(qpos);
        if (aq == 0 && at == 1) { start = 0; tp0 = tpos; tp1 = tpos + sep; tp2 = tpos + sep + 1; }
        else if (aq == 0 && at == 2) { start = 0; tp0 = tpos; tp1 = tpos + 1; tp2 = tpos + sep; }
        else if (aq == 1 && at == 2) { start = 1; tp0 = tpos - sep; tp1 = tpos; tp2 = tpos + 1; }
        else if (aq == 1 && at == 1) { start = 2; tp0 = tpos - sep; tp1 = tpos - sep + 1; tp2 = tpos; }
        else { failed = true; return; }
        char codon[4] = {tsym(tp0), tsym(tp1), tsym(tp2), 0};
        split_count++;
        const char *q_name = tla(q_aa);
        const int n = std::max(aq, at);
        char qs[16], ts[16], its[16], ms[16], cm[4];
        snprintf(qs, sizeof qs, "{%.*s}", n, q_name + start);
        snprintf(ts, sizeof ts, "{%.*s}", n, codon + start);
        for (int k = 0; k < 3; k++) codon[k] = (char)toupper((unsigned char)codon[k]);      // strup
        const char t_aa = translate3(codon);
        snprintf(its, sizeof its, "{%.*s}", n, tla(t_aa) + start);
        codon_match_string(codon, q_aa, cm);
        snprintf(ms, sizeof ms, "{%.*s}", n, cm + start);
        add(qs, nullptr, ms, its, ts, qpos, tpos);
    }
    void add_frameshift(int aq, int at, int total, int qpos, int tpos) {                   // alignment.c:1040-1090
        const bool emitted_protein = aq ? q_prot : t_prot;
        for (int i = 0; i < total; i++) {
            char seq[4] = {0, 0, 0, 0}, ms[4] = {0, 0, 0, 0}, gap[4] = {0, 0, 0, 0};
            const int n = aq | at;
            for (int j = 0; j < n && j < 3; j++) { seq[j] = aq ? qsym(qpos + j) : tsym(tpos + j); ms[j] = '#'; gap[j] = '-'; }
            if (emitted_protein) {
                memcpy(seq, tla(seq[0]), 3);
                ms[0] = ms[1] = ms[2] = '#'; gap[0] = gap[1] = gap[2] = '-';
                seq[3] = ms[3] = gap[3] = 0;
            }
            if (aq) add(seq, ms, ms, gap, gap, qpos, tpos);
            else add(gap, gap, ms, ms, seq, qpos, tpos);
            qpos += aq; tpos += at;
        }
    }
    // AlignmentView_add_label_operation (alignment.c:1092-1181)
    void add_label_operation(const c4gpu_transition &t, int total, int qpos, int tpos, bool next_same_label,
                             const c4gpu_transition **last_match) {
        switch (t.label) {
            case C4GPU_LABEL_NONE: break;
            case C4GPU_LABEL_MATCH: *last_match = &t; add_match(t, total, qpos, tpos); break;
            case C4GPU_LABEL_GAP: add_gap(t.advance_query, t.advance_target, total, qpos, tpos); break;
            case C4GPU_LABEL_5SS: add_splice_site(t.advance_query, t.advance_target, qpos, tpos, true, *last_match); break;
            case C4GPU_LABEL_3SS: add_splice_site(t.advance_query, t.advance_target, qpos, tpos, false, *last_match); break;
            case C4GPU_LABEL_INTRON:
                intron_aq += t.advance_query * total; intron_at += t.advance_target * total;
                if (!next_same_label) { add_intron(intron_aq, intron_at, qpos, tpos, *last_match); intron_aq = intron_at = 0; }
                break;
            case C4GPU_LABEL_SPLIT_CODON: add_split_codon(t.advance_query, t.advance_target, qpos, tpos); break;
            case C4GPU_LABEL_FRAMESHIFT: add_frameshift(t.advance_query, t.advance_target, total, qpos, tpos); break;
            default: failed = true; break;                                                  // NER
        }
    }
    int coordinate(bool on_query, bool start) const {                                       // Alignment_get_coordinate, :177-205
        int pos = on_query ? (start ? a->region.query_start : a->region.query_start + a->region.query_length)
                           : (start ? a->region.target_start : a->region.target_start + a->region.target_length);
        if (r->forward_coords && (on_query ? r->query_strand : r->target_strand) == '-')
            pos = (on_query ? r->query_len : r->target_len) - pos;
        return pos;
    }
    static void prepare_seq(std::string &outer, std::string &inner, int pos, int w) {        // :1242-1257
        for (int i = 0; i < w; i++) {
            if (inner[pos + i] == ' ') { std::swap(inner[pos + i], outer[pos + i]); continue; }
            if (inner[pos + i] == '^') inner[pos + i] = ' ';
        }
    }
    static bool is_empty(const std::string &s, int pos, int w) {
        for (int i = 0; i < w; i++) if (s[pos + i] != ' ') return false;
        return true;
    }
    static void replace_padding(std::string &s, int pos, int w) {
        for (int i = 0; i < w; i++) if (s[pos + i] == '^') s[pos + i] = ' ';
    }
    static void outf(std::string &out, const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        va_list ap;
        va_start(ap, fmt);
        va_list ap2;
        va_copy(ap2, ap);
        const int n = vsnprintf(nullptr, 0, fmt, ap);
        va_end(ap);
        std::string tmp((size_t)n + 1, '\0');
        vsnprintf(&tmp[0], tmp.size(), fmt, ap2);
        va_end(ap2);
        tmp.resize((size_t)n);
        out += tmp;
    }
    void display_row(std::string &out, int row, int pos, int w) {                            // :1268-1321
        int p1q = row_marker[row].first + 1, p2q = row_marker[row + 1].first + 1;
        int p1t = row_marker[row].second + 1, p2t = row_marker[row + 1].second + 1;
        if (r->forward_coords) {
            if (r->query_strand == '-') { p1q = r->query_len - p1q - 1; p2q = r->query_len - p2q + 1; }
            if (r->target_strand == '-') { p1t = r->target_len - p1t - 1; p2t = r->target_len - p2t + 1; }
        }
        bool show_iq = false, show_it = false;
        if (has_iq && !is_empty(iq, pos, w)) { show_iq = true; prepare_seq(oq, iq, pos, w); }
        if (has_it && !is_empty(it, pos, w)) { show_it = true; prepare_seq(ot, it, pos, w); }
        replace_padding(oq, pos, w);
        replace_padding(ot, pos, w);
        outf(out, " %*d : %.*s : %*d\n", max_pos_len, p1q + 1, w, oq.c_str() + pos, max_pos_len, p2q);
        if (show_iq) outf(out, " %*s   %.*s\n", max_pos_len, " ", w, iq.c_str() + pos);
        outf(out, " %*s   %.*s\n", max_pos_len, " ", w, mid.c_str() + pos);
        if (show_it) outf(out, " %*s   %.*s\n", max_pos_len, " ", w, it.c_str() + pos);
        outf(out, " %*d : %.*s : %*d\n", max_pos_len, p1t + 1, w, ot.c_str() + pos, max_pos_len, p2t);
    }
};

}  // namespace

extern "C" int c4gpu_alignment_display(const c4gpu_model *m, const c4gpu_params *p, const c4gpu_alignment *a,
                                       const c4gpu_display_request *req, char *buf, size_t buf_len) {
    ViewCtx v;
    v.m = m; v.p = p; v.a = a; v.r = req;
    for (int k = 0; k < m->n_transitions; k++) {
        const c4gpu_transition &t = m->transitions[k];
        if (t.label != C4GPU_LABEL_MATCH || t.calc < 0) continue;
        const int kind = m->calcs[t.calc].kind;
        if (kind < C4GPU_CALC_MATCH_DNA || kind > C4GPU_CALC_MATCH_P2D) return INT32_MIN;
        v.match_kind = kind;
    }
    if (v.match_kind < 0 || a->n_ops < 1) return INT32_MIN;
    v.q_prot = v.match_kind != C4GPU_CALC_MATCH_DNA;
    v.t_prot = v.match_kind == C4GPU_CALC_MATCH_PROTEIN;
    // AlignmentView_create (alignment.c:285-349)
    v.has_iq = m->max_query_advance == 3;
    v.has_it = m->max_target_advance == 3;
    {
        const int qmax = std::max(v.coordinate(true, true), v.coordinate(true, false));
        const int tmax = std::max(v.coordinate(false, true), v.coordinate(false, false));
        char tmp[32];
        v.max_pos_len = snprintf(tmp, sizeof tmp, "%d", std::max(qmax, tmax));
    }
    v.width = (req->width > 0 ? req->width : 80) - ((v.max_pos_len + 5) << 1);
    if (v.width <= 0) return INT32_MIN;
    v.limit = v.width;
    for (int i = 0; i < a->n_ops; i++) {
        if (v.tr(i).label == C4GPU_LABEL_5SS) { v.orientation = '+'; break; }
        if (v.tr(i).label == C4GPU_LABEL_3SS) { v.orientation = '-'; break; }
    }
    {
        bool open = false;
        std::pair<int, int> cur(0, 0);
        for (int i = 0; i < a->n_ops; i++) {
            const c4gpu_transition &t = v.tr(i);
            const int dq = a->op_length[i] * t.advance_query, dt = a->op_length[i] * t.advance_target;
            if (open) {
                if (t.label == C4GPU_LABEL_SPLIT_CODON) { v.split_sep.push_back(cur); open = false; }
                else { cur.first += dq; cur.second += dt; }
            } else if (t.label == C4GPU_LABEL_SPLIT_CODON) { cur = std::make_pair(dq, dt); open = true; }
        }
        if (open) return INT32_MIN;
    }
    std::string out;
    ViewCtx::outf(out, "\nC4 Alignment:\n------------\n         Query: %s%s%s\n        Target: %s%s%s\n         Model: %s\n"
                  "     Raw score: %d\n   Query range: %d -> %d\n  Target range: %d -> %d\n\n",
                  req->query_id, req->query_def ? " " : "", req->query_def ? req->query_def : "",
                  req->target_id, req->target_def ? " " : "", req->target_def ? req->target_def : "", m->name, a->score,
                  v.coordinate(true, true), v.coordinate(true, false), v.coordinate(false, true), v.coordinate(false, false));
    // AlignmentView_prepare (alignment.c:1183-1231): equal transitions that follow each other are one operation
    {
        int qpos = a->region.query_start, tpos = a->region.target_start;
        const c4gpu_transition *last_match = nullptr;
        v.row_marker.emplace_back(a->region.query_start - 1, a->region.target_start - 1);
        int prev = 0, total = a->op_length[0];
        for (int i = 1; i < a->n_ops; i++) {
            if (a->op_transition[prev] == a->op_transition[i]) { total += a->op_length[i]; continue; }
            const c4gpu_transition &pt = v.tr(prev);
            v.add_label_operation(pt, total, qpos, tpos, pt.label == v.tr(i).label, &last_match);
            qpos += pt.advance_query * total; tpos += pt.advance_target * total;
            prev = i; total = a->op_length[i];
        }
        v.add_label_operation(v.tr(prev), total, qpos, tpos, false, &last_match);
        v.row_marker.emplace_back(a->region.query_start + a->region.query_length - 1, a->region.target_start + a->region.target_length - 1);
    }
    if (v.failed) return INT32_MIN;
    // AlignmentView_display (alignment.c:1323-1339)
    {
        int pos = 0, row = 0;
        const int pause = (int)v.oq.size() - v.width;
        while (pos < pause) {
            if ((size_t)row + 1 >= v.row_marker.size()) return INT32_MIN;
            v.display_row(out, row, pos, v.width);
            pos += v.width; row++;
            out += "\n";
        }
        if ((size_t)row + 1 >= v.row_marker.size()) return INT32_MIN;
        v.display_row(out, row, pos, (int)v.oq.size() - pos);
        out += "\n";
    }
    if (out.size() + 1 > buf_len) return -(int)(out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

// ---- Alignment_display_ryo (alignment.c:1781-2669) -------------------------------------------------------------------
namespace {

// SplicePredictor_predict (splice.c:299-352) at one position: what the device's splice_kernel computes per target position
int host_splice_score(const c4gpu_splice_model *sp, const uint8_t *s, int n, int pos) {
    int seq_start = pos - sp->splice_after, model_start = 0, calc_length = sp->model_length;
    if (seq_start < 0) { model_start = -seq_start; seq_start = 0; calc_length -= model_start; }
    if (seq_start + calc_length > n) calc_length = n - seq_start;
    float score = 0.0f;
    for (int i = 0; i < calc_length; i++) score = score + sp->data[model_start + i][sp->index[s[seq_start + i]]];
    if (sp->gtag_only) {
        const int b1 = s[pos], b2 = pos + 1 < n ? s[pos + 1] : 0;
        if (toupper(b1) != sp->expect_one || toupper(b2) != sp->expect_two) score = -987654321.0f;
    }
    const double r = score < 0 ? (double)score - 0.5 : (double)score + 0.5;
    return (int)r;
}

void fasta_block(std::string &out, const uint8_t *s, int len) {        // Sequence_print_fasta_block (sequence.c:287-303)
    if (!len) return;
    int pos = 0;
    const int width = 70, pause = len - width;
    while (pos < pause) { out.append(reinterpret_cast<const char *>(s) + pos, width); out += "\n"; pos += width; }
    out.append(reinterpret_cast<const char *>(s) + pos, len - pos);
    out += "\n";
}

}  // namespace

extern "C" int c4gpu_alignment_format_ryo(const c4gpu_model *m, const c4gpu_params *p, const c4gpu_alignment *a,
                                          const c4gpu_ryo_request *req, char *buf, size_t buf_len) {
    c4gpu_gff_request gr;
    memset(&gr, 0, sizeof gr);
    gr.query_id = req->query_id; gr.target_id = req->target_id; gr.query = req->query; gr.target = req->target;
    gr.query_len = req->query_len; gr.target_len = req->target_len; gr.query_strand = req->query_strand; gr.target_strand = req->target_strand;
    GffCtx g;
    g.m = m; g.p = p; g.a = a; g.r = &gr;
    for (int k = 0; k < m->n_transitions; k++) {
        const c4gpu_transition &t = m->transitions[k];
        if (t.label != C4GPU_LABEL_MATCH || t.calc < 0) continue;
        const int kind = m->calcs[t.calc].kind;
        if (kind < C4GPU_CALC_MATCH_DNA || kind > C4GPU_CALC_MATCH_P2D) return INT32_MIN;
        g.match_kind = kind;
    }
    if (g.match_kind < 0 || a->n_ops < 1 || !req->format) return INT32_MIN;
    const bool q_prot = g.match_kind != C4GPU_CALC_MATCH_DNA, t_prot = g.match_kind == C4GPU_CALC_MATCH_PROTEIN;
    auto coord = [&](bool on_query, int pos) {             // Alignment_convert_coordinate (alignment.c:209-232)
        if (req->forward_coords && (on_query ? req->query_strand : req->target_strand) == '-')
            pos = (on_query ? req->query_len : req->target_len) - pos;
        return pos;
    };
    std::string &out = g.out;
    // Alignment_Coding_create (alignment.c:2330-2392): the codons (and split-codon bases, codon gaps) of one side
    struct Coding { bool made = false; int begin = 0, end = 0; std::string seq; } coding[2];
    auto make_coding = [&](bool on_query) -> Coding & {
        Coding &c = coding[on_query ? 0 : 1];
        if (c.made) return c;
        c.made = true;
        const uint8_t *src = on_query ? req->query : req->target;
        int qp = a->region.query_start, tp = a->region.target_start;
        bool first = true;
        for (int i = 0; i < a->n_ops; i++) {
            const c4gpu_transition &t = g.tr(i);
            for (int j = 0; j < a->op_length[i]; j++) {
                // Alignment_Position_next moves before anything is looked at: the very first transition step is skipped
                if (first) { first = false; qp += t.advance_query; tp += t.advance_target; continue; }
                const int advance = on_query ? t.advance_query : t.advance_target, pos = on_query ? qp : tp;
                if (t.label == C4GPU_LABEL_MATCH && advance == 3) {
                    if (c.seq.empty()) c.begin = pos;
                    c.seq.append(reinterpret_cast<const char *>(src) + pos, 3);
                    c.end = pos;
                } else if (t.label == C4GPU_LABEL_SPLIT_CODON) {
                    c.seq.append(reinterpret_cast<const char *>(src) + pos, advance);
                } else if (t.label == C4GPU_LABEL_GAP && advance == 3) {
                    c.seq.append(reinterpret_cast<const char *>(src) + pos, 3);
                }
                qp += t.advance_query; tp += t.advance_target;
            }
        }
        return c;
    };
    // the walk of the per-transition section (Alignment_Position, alignment.c:2226-2328)
    struct Walk { int op = 0, op_pos = 0, qp = 0, tp = 0; int cell[C4GPU_MAX_SHADOWS + 1]; int curr_intron_start = 0; } w;
    auto set_shadows = [&]() {                             // Alignment_Position_set_shadows
        const c4gpu_transition &t = g.tr(w.op);
        for (int h = 0; h < m->n_shadows; h++)
            if (m->shadows[h].src_state_mask >> t.input & 1u)
                w.cell[1 + m->shadows[h].designation] = m->shadows[h].on_target ? w.tp : w.qp;
        for (int h = 0; h < m->n_shadows; h++)
            if (t.dst_shadow_mask >> h & 1u) w.curr_intron_start = w.cell[1 + m->shadows[h].designation];   // intron.c:468
    };
    auto walk_next = [&]() -> bool {                       // Alignment_Position_next
        const c4gpu_transition &t = g.tr(w.op);
        w.qp += t.advance_query; w.tp += t.advance_target;
        if (++w.op_pos < a->op_length[w.op]) { set_shadows(); return true; }
        if (++w.op < a->n_ops) { w.op_pos = 0; set_shadows(); return true; }
        return false;
    };
    auto calc_score = [&](const c4gpu_transition &t, int qpos, int tpos, bool *ok) -> int {   // C4_Calc_score (c4.c:1700)
        if (t.calc < 0) return 0;
        const c4gpu_calc &c = m->calcs[t.calc];
        switch (c.kind) {
            case C4GPU_CALC_CONST: return c.value;
            case C4GPU_CALC_MATCH_DNA: case C4GPU_CALC_MATCH_PROTEIN: case C4GPU_CALC_MATCH_P2D: return g.match_score(qpos, tpos);
            case C4GPU_CALC_SPLICE_PRE:
                return c.value + host_splice_score(&p->splice[c.param], req->target, req->target_len, tpos);
            case C4GPU_CALC_SPLICE_POST: {                 // Intron_calc_*, intron.c:138-161
                const int intron_length = tpos - w.curr_intron_start + 2;
                if (intron_length < p->min_intron || intron_length > p->max_intron) return -987654321;
                return host_splice_score(&p->splice[c.param], req->target, req->target_len, tpos);
            }
            case C4GPU_CALC_PHASE_POST: {                  // Phase_{1,2}_PROTEIN2DNA_FALSE_TRUE_calc_func, phase.c:188-213
                const int phase = c.param, cis = w.curr_intron_start;
                if (cis < phase) return -987654321;
                int tp1, tp2, tp3;
                if (phase == 1) { tp1 = cis - 1; tp2 = tpos; tp3 = tpos + 1; } else { tp1 = cis - 2; tp2 = cis - 1; tp3 = tpos; }
                const uint8_t aa = p->aa[p->trans[p->nt2d[req->target[tp1]] | (p->nt2d[req->target[tp2]] << 4) | (p->nt2d[req->target[tp3]] << 8)]];
                return p->protein_submat[p->submat_index[req->query[qpos]]][p->submat_index[aa]];
            }
            default: *ok = false; return 0;
        }
    };
    static const char *label_name[] = {"none", "match", "gap", "ner", "5'ss", "3'ss", "intron", "split codon", "frameshift"};
    // Alignment_RYO_tokenise + Alignment_RYO_token_list_print in one pass over the format (the section { } loops by index)
    const char *f = req->format;
    const int flen = (int)strlen(f);
    int pto_start = -1;
    char tmp[64];
    auto block = [&](int what) -> bool {                   // Alignment_print_{sugar,cigar,vulgar}_block: the line without its tag
        std::vector<char> b(64 + 32 * ((size_t)a->n_ops + 4) + strlen(req->query_id) + strlen(req->target_id));
        const int n = c4gpu_alignment_format(m, a, what, req->query_id, req->query_len, req->query_strand, req->target_id,
                                             req->target_len, req->target_strand, req->forward_coords, b.data(), b.size());
        if (n < 0) return false;
        const std::string line(strchr(b.data(), ':') + 2);
        if (what == 0) { out += line; return true; }
        // the line is the sugar block, a space, then the block (alignment.c:2681-2708)
        std::vector<char> sb(b.size());
        const int sn = c4gpu_alignment_format(m, a, 0, req->query_id, req->query_len, req->query_strand, req->target_id,
                                              req->target_len, req->target_strand, req->forward_coords, sb.data(), sb.size());
        if (sn < 0) return false;
        const size_t sugar_len = strlen(strchr(sb.data(), ':') + 2);
        std::string rest = line.substr(sugar_len);
        if (!rest.empty() && rest[0] == ' ') rest.erase(0, 1);
        out += rest;
        return true;
    };
    for (int i = 0; i < flen; i++) {
        const char ch = f[i];
        if (ch == '\\') {
            const char n = i + 1 < flen ? f[i + 1] : 0;
            if (n == '\\') out += "\\"; else if (n == 'n') out += "\n"; else if (n == 't') out += "\t";
            else if (n == '{') out += "{"; else if (n == '}') out += "}"; else return INT32_MIN;
            i++;
        } else if (ch == '{') {
            if (pto_start != -1) return INT32_MIN;        // "Cannot nest PTO brackets"
            pto_start = i;
            w = Walk();
            w.qp = a->region.query_start; w.tp = a->region.target_start;
            memset(w.cell, 0, sizeof w.cell);
        } else if (ch == '}') {
            if (pto_start == -1) return INT32_MIN;
            if (walk_next()) i = pto_start; else pto_start = -1;
        } else if (ch != '%') {
            out += ch;
        } else {
            const char c1 = i + 1 < flen ? f[i + 1] : 0, c2 = i + 2 < flen ? f[i + 2] : 0, c3 = i + 3 < flen ? f[i + 3] : 0;
            int used = 1;                                   // characters after the '%'
            if (c1 == '%') out += "%";
            else if (c1 == 'q' || c1 == 't') {
                const bool oq = c1 == 'q';
                const uint8_t *seq = oq ? req->query : req->target;
                const int len = oq ? req->query_len : req->target_len;
                used = 2;
                switch (c2) {
                    case 'i': out += oq ? req->query_id : req->target_id; break;
                    case 'd': { const char *d = oq ? req->query_def : req->target_def; if (d) out += d; break; }
                    case 'l': snprintf(tmp, sizeof tmp, "%d", len); out += tmp; break;
                    case 's': fasta_block(out, seq, len); break;
                    case 'S': out += oq ? req->query_strand : req->target_strand; break;
                    case 't': out += (oq ? q_prot : t_prot) ? "Protein" : "DNA"; break;
                    case 'a': {
                        used = 3;
                        const int start = oq ? a->region.query_start : a->region.target_start;
                        const int alen = oq ? a->region.query_length : a->region.target_length;
                        if (c3 == 'b') { snprintf(tmp, sizeof tmp, "%d", coord(oq, start)); out += tmp; }
                        else if (c3 == 'e') { snprintf(tmp, sizeof tmp, "%d", coord(oq, start + alen)); out += tmp; }
                        else if (c3 == 'l') { snprintf(tmp, sizeof tmp, "%d", alen); out += tmp; }
                        else if (c3 == 's') fasta_block(out, seq + start, alen);
                        else return INT32_MIN;
                        break;
                    }
                    case 'c': {
                        used = 3;
                        if (oq ? q_prot : t_prot) return INT32_MIN;        // g_assert: the coding side is DNA
                        Coding &c = make_coding(oq);
                        if (c3 == 'b') { snprintf(tmp, sizeof tmp, "%d", coord(oq, c.begin)); out += tmp; }
                        else if (c3 == 'e') { snprintf(tmp, sizeof tmp, "%d", coord(oq, c.end)); out += tmp; }
                        else if (c3 == 'l') { snprintf(tmp, sizeof tmp, "%d", (int)c.seq.size()); out += tmp; }
                        else if (c3 == 's') fasta_block(out, reinterpret_cast<const uint8_t *>(c.seq.data()), (int)c.seq.size());
                        else return INT32_MIN;
                        break;
                    }
                    default: return INT32_MIN;
                }
            } else if (c1 == 's') { snprintf(tmp, sizeof tmp, "%d", a->score); out += tmp; }
            else if (c1 == 'm') out += m->name;
            else if (c1 == 'r') { if (req->rank == -1) out += "%_EXONERATE_BESTN_RANK_%"; else { snprintf(tmp, sizeof tmp, "%d", req->rank); out += tmp; } }
            else if (c1 == 'p' || c1 == 'e') {
                used = 2;
                int idm, idt, sim, simt, gaps = 0;
                g.equivalenced(true, 0, 0, false, &idm, &idt);
                g.equivalenced(false, 0, 0, false, &sim, &simt);
                for (int k = 0; k < a->n_ops; k++) if (g.tr(k).label == C4GPU_LABEL_GAP) gaps += a->op_length[k];
                float v = 0;
                bool is_float = true;
                int iv = 0;
                if (c1 == 'p') {
                    if (c2 == 'c') v = ((float)idt / (float)req->query_len) * 100;
                    else if (c2 == 'I') v = ((float)idm / ((float)idt + (float)gaps)) * 100;
                    else if (c2 == 'i') v = (((float)idm) / ((float)idt)) * 100;
                    else if (c2 == 's') v = (((float)sim) / ((float)simt)) * 100;
                    else if (c2 == 'S') {                   // Alignment_get_percent_self (alignment.c:1563-1618)
                        if (g.match_kind == C4GPU_CALC_MATCH_P2D) return INT32_MIN;
                        int score = 0, self = 0, qp = a->region.query_start, tp = a->region.target_start;
                        const bool prot = g.match_kind == C4GPU_CALC_MATCH_PROTEIN;
                        for (int k = 0; k < a->n_ops; k++) {
                            const c4gpu_transition &t = g.tr(k);
                            for (int j = 0; j < a->op_length[k]; j++) {
                                if (t.label == C4GPU_LABEL_MATCH) {
                                    score += g.match_score(qp, tp);
                                    const int x = p->submat_index[req->query[qp]];
                                    self += prot ? p->protein_submat[x][x] : p->dna_submat[x][x];
                                }
                                qp += t.advance_query; tp += t.advance_target;
                            }
                        }
                        v = (((float)score) / ((float)self)) * 100;
                    } else return INT32_MIN;
                } else {
                    is_float = false;
                    if (c2 == 't') iv = idt; else if (c2 == 'i') iv = idm; else if (c2 == 's') iv = sim; else if (c2 == 'm') iv = idt - idm;
                    else return INT32_MIN;
                }
                if (is_float) snprintf(tmp, sizeof tmp, "%2.2f", v); else snprintf(tmp, sizeof tmp, "%d", iv);
                out += tmp;
            } else if (c1 == 'g') {
                char o = '.';
                for (int k = 0; k < a->n_ops; k++) {
                    if (g.tr(k).label == C4GPU_LABEL_5SS) { o = '+'; break; }
                    if (g.tr(k).label == C4GPU_LABEL_3SS) { o = '-'; break; }
                }
                out += o;
            } else if (c1 == 'S') { if (!block(0)) return INT32_MIN; }
            else if (c1 == 'C') { if (!block(1)) return INT32_MIN; }
            else if (c1 == 'V') { if (!block(2)) return INT32_MIN; }
            else if (c1 == 'P') {
                if (pto_start == -1) return INT32_MIN;     // g_assert(pto_start != -1)
                const c4gpu_transition &t = g.tr(w.op);
                used = 2;
                if (c2 == 'q' || c2 == 't') {
                    used = 3;
                    const bool oq = c2 == 'q';
                    const int adv = oq ? t.advance_query : t.advance_target, pos = oq ? w.qp : w.tp;
                    if (c3 == 's') { if (adv) out.append(reinterpret_cast<const char *>(oq ? req->query : req->target) + pos, adv); else out += "-"; }
                    else if (c3 == 'a') { snprintf(tmp, sizeof tmp, "%d", adv); out += tmp; }
                    else if (c3 == 'b') { snprintf(tmp, sizeof tmp, "%d", coord(oq, pos)); out += tmp; }
                    else if (c3 == 'e') { snprintf(tmp, sizeof tmp, "%d", coord(oq, pos + adv)); out += tmp; }
                    else return INT32_MIN;
                } else if (c2 == 'n') out += t.name;
                else if (c2 == 's') {
                    bool ok = true;
                    snprintf(tmp, sizeof tmp, "%d", calc_score(t, w.qp, w.tp, &ok));
                    if (!ok) return INT32_MIN;
                    out += tmp;
                } else if (c2 == 'l') out += label_name[t.label];
                else return INT32_MIN;
            } else return INT32_MIN;
            i += used;
        }
    }
    if (pto_start != -1) return INT32_MIN;                  // "No closing PTO bracket in --ryo string"
    if (out.size() + 1 > buf_len) return -(int)(out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}
