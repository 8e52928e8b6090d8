// c4_sdp_host.h — the host side of c4gpu_sdp_batch around the sparse wavefront passes (c4_sdp_wave.h): the seed list of
// every pair (SDP_Pair_create_seed_list, sdp.c:438-477), the job / seed / stream descriptors the passes take, and — once
// the passes and the walks are back — the reference's single-pass loop over the seeds (SDP_Pair_next_path, sdp.c:743-815).
// Shared by the product (c4_engine.hip: the passes are HIP kernels) and by tests/sdp_sim.hip (the same per-lane code
// driven by CPU loops, test infrastructure); nothing here touches a device.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "c4_sdp_wave.h"
#include "c4_internal.h"

namespace c4sdp {

// per model family: which flavour SDP_create picks (sdp.c:322-366) and how its passes lay out their streams
struct SdpFamilyInfo {
    bool ok = false, bnd = false;
    int rpc_log[2], rec_bytes[2], cpc_log[2], cent_bytes[2];       // [0] reverse pass, [1] forward pass
    SdpWalkTab walk_rev, walk_fwd;
};
template <class M, bool BND>
inline SdpFamilyInfo make_family_info() {
    SdpFamilyInfo f;
    f.ok = true; f.bnd = BND;
    using LR = Layout<M, false, BND>;
    using LF = Layout<M, true, BND>;
    f.rpc_log[0] = LR::RPC_LOG; f.rec_bytes[0] = LR::REC_BYTES; f.cpc_log[0] = LR::CPC_LOG; f.cent_bytes[0] = LR::CENT_BYTES;
    f.rpc_log[1] = LF::RPC_LOG; f.rec_bytes[1] = LF::REC_BYTES; f.cpc_log[1] = LF::CPC_LOG; f.cent_bytes[1] = LF::CENT_BYTES;
    f.walk_rev = make_walk_tab<M, false, BND>();
    f.walk_fwd = make_walk_tab<M, true, BND>();
    return f;
}

// the launch constants of a model (calc values, intron window, substitution matrix, codon -> matrix row): what Engine::init
// stages for the Viterbi kernels, shared with the SDP passes
inline void make_kparams(const c4gpu_model *m, const c4gpu_params *params, KParams *kp) {
    memset(kp, 0, sizeof *kp);
    for (int i = 0; i < m->n_calcs; i++) kp->calc_value[i] = m->calcs[i].value;
    kp->min_intron = params->min_intron; kp->max_intron = params->max_intron;
    kp->start_scope = m->start_scope; kp->end_scope = m->end_scope;
    bool protein = false;
    for (int i = 0; i < m->n_calcs; i++)
        if (m->calcs[i].kind == C4GPU_CALC_MATCH_PROTEIN || m->calcs[i].kind == C4GPU_CALC_MATCH_P2D) protein = true;
    memcpy(kp->submat, protein ? &params->protein_submat[0][0] : &params->dna_submat[0][0], sizeof(int) * 24 * 24);
    for (int c = 0; c < 4096; c++) {
        const uint8_t row = params->submat_index[params->aa[params->trans[c]]];
        kp->codon_row[c] = row < 24 ? row : 0;
    }
}

struct SdpTerminal {                                   // SDP_Terminal, sdp.h:48
    c4gpu_score score = C4GPU_IMPOSSIBLY_LOW_SCORE;
    int q = 0, t = 0;
    std::vector<unsigned> runs;                        // (transition, count) pairs in walk order (end -> start)
};
struct SdpHostSeed { int seed_id; const c4gpu_hsp *hsp; int qcobs, tcobs; SdpTerminal max_start, max_end; };   // SDP_Seed

inline int sdp_hsp_cmp(const void *a, const void *b) {                                  // sdp.c:425-436
    const SdpHostSeed *x = *(SdpHostSeed *const *)a, *y = *(SdpHostSeed *const *)b;
    const int td = x->tcobs - y->tcobs;
    return td ? td : (x->qcobs - y->qcobs);
}
inline int sdp_score_cmp(const void *a, const void *b) {                                // sdp.c:736-741
    const SdpHostSeed *x = *(SdpHostSeed *const *)a, *y = *(SdpHostSeed *const *)b;
    return y->max_end.score - x->max_end.score;
}

// SDP_Pair_create_seed_list (sdp.c:438-477): the HSPs sorted on their cobs point in DP order, one seed per point
inline void sdp_seed_list(const c4gpu_hsp *hsps, int n, int query_advance, int target_advance, std::vector<SdpHostSeed> &seeds) {
    seeds.clear();
    if (n <= 0) return;
    std::vector<SdpHostSeed> all(n);
    std::vector<SdpHostSeed *> sorted(n);
    for (int k = 0; k < n; k++) {
        all[k].hsp = &hsps[k];
        all[k].qcobs = hsps[k].query_start + hsps[k].cobs * query_advance;               // HSP_query_cobs, hspset.h:93
        all[k].tcobs = hsps[k].target_start + hsps[k].cobs * target_advance;
        sorted[k] = &all[k];
    }
    qsort(sorted.data(), n, sizeof(SdpHostSeed *), sdp_hsp_cmp);
    for (int k = 0; k < n; k++)
        if (!k || sorted[k]->qcobs != sorted[k - 1]->qcobs || sorted[k]->tcobs != sorted[k - 1]->tcobs) {
            seeds.push_back(*sorted[k]);
            seeds.back().seed_id = (int)seeds.size() - 1;
        }
}

// everything the passes of one launch take, host copy
struct SdpHostPlan {
    std::vector<SdpJob> jobs;
    std::vector<int> job_pair;                         // job -> pair of the batch
    std::vector<SdpDSeed> rev_seeds, fwd_seeds;        // per job at [seed_off, seed_off + n_seeds), sorted by (strip, c)
    std::vector<int> seed_job;                         // per seed slot
    long long tabs_total = 0, dirs_total = 0;
    double work = 0;                                   // a guess of the steps the passes will execute (arena sizing)
};

// jobs in descending order of expected work (one wave each: the long ones start first)
inline void sdp_make_plan(const SdpFamilyInfo &fi, const c4gpu_pair *pairs, int n_pairs, const std::vector<long long> &q_off,
                          const std::vector<long long> &t_off, const std::vector<std::vector<SdpHostSeed>> &seeds,
                          const std::vector<char> &active, unsigned arena_chunks, SdpHostPlan &plan) {
    plan = SdpHostPlan();
    std::vector<int> order;
    std::vector<double> work(n_pairs, 0.0);
    for (int i = 0; i < n_pairs; i++) {
        if (seeds[i].empty() || !active[i]) continue;
        order.push_back(i);
        for (const SdpHostSeed &h : seeds[i]) work[i] += h.hsp->length + 160.0;
        work[i] *= 1.0 + pairs[i].query_len / 64.0 / std::max<size_t>(1, seeds[i].size());
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return work[a] > work[b]; });
    for (int i : order) {
        SdpJob j;
        memset(&j, 0, sizeof j);
        j.q_off = q_off[i]; j.t_off = t_off[i]; j.Q = pairs[i].query_len; j.T = pairs[i].target_len;
        j.n_strips = (j.Q + 1 + 63) / 64;
        j.n_seeds = (int)seeds[i].size();
        j.seed_off = (int)plan.seed_job.size();
        j.dir_off = (int)plan.dirs_total;
        plan.dirs_total += (long long)ST_COUNT * (j.n_strips + 1);
        for (int st = 0; st < ST_COUNT; st++) {
            const int pass = (st == ST_FWDREC || st == ST_FWDCARRY) ? 1 : 0;
            const bool rec = st == ST_REVREC || st == ST_FWDREC;
            const long long dense = (long long)j.n_strips * ((long long)j.T + 64);
            long long cap = (dense >> (rec ? fi.rpc_log[pass] : fi.cpc_log[pass])) + j.n_strips + 2;
            cap = std::min<long long>(cap, (long long)arena_chunks + 1);
            j.tab_off[st] = plan.tabs_total; j.tab_cap[st] = (int)cap;
            plan.tabs_total += cap;
        }
        std::vector<SdpDSeed> rv, fw;
        for (int k = 0; k < j.n_seeds; k++) {
            const SdpHostSeed &h = seeds[i][k];
            SdpDSeed d;
            int u = j.Q - h.qcobs, v = j.T - h.tcobs;                  // reverse pass: Scheduler_Seed_List_get_reverse, sdp.c:95-108
            d.strip = u >> 6; d.lane = u & 63; d.c = v + d.lane; d.val = h.hsp->score >> 1; d.sid = k;
            rv.push_back(d);
            u = h.qcobs; v = h.tcobs;                                  // forward pass of the seeded flavour: sdp.c:79-93
            d.strip = u >> 6; d.lane = u & 63; d.c = v + d.lane;
            fw.push_back(d);
        }
        auto by_step = [](const SdpDSeed &a, const SdpDSeed &b) { return a.strip != b.strip ? a.strip < b.strip : (a.c != b.c ? a.c < b.c : a.lane < b.lane); };
        std::sort(rv.begin(), rv.end(), by_step);
        std::sort(fw.begin(), fw.end(), by_step);
        plan.rev_seeds.insert(plan.rev_seeds.end(), rv.begin(), rv.end());
        plan.fwd_seeds.insert(plan.fwd_seeds.end(), fw.begin(), fw.end());
        for (int k = 0; k < j.n_seeds; k++) plan.seed_job.push_back((int)plan.jobs.size());
        plan.work += work[i];
        plan.jobs.push_back(j); plan.job_pair.push_back(i);
    }
}

// what comes back from the passes and the walks
struct SdpBackendOut {
    std::vector<SdpWalkOut> rev, fwd;                  // per seed slot (rev: seeded flavour only)
    std::vector<unsigned> runs;
    std::vector<int> status;                           // per job
};

// seed slot results -> the pair's seeds; false when the pair was not served (arena exhausted, walk lost)
inline bool sdp_collect(const SdpFamilyInfo &fi, const c4gpu_model *m, const SdpHostPlan &plan, const SdpBackendOut &out, int jx,
                        std::vector<SdpHostSeed> &seeds) {
    const SdpJob &j = plan.jobs[jx];
    if (out.status[jx] != SDP_OK) return false;
    for (int k = 0; k < j.n_seeds; k++) {
        const int x = j.seed_off + k;
        SdpHostSeed &h = seeds[k];
        const SdpWalkOut &f = out.fwd[x];
        if (f.status != SDP_OK) return false;
        h.max_end.score = f.score; h.max_end.q = f.q; h.max_end.t = f.t;
        h.max_end.runs.assign(out.runs.begin() + 2 * f.runs_off, out.runs.begin() + 2 * (f.runs_off + f.n_runs));
        if (!fi.bnd) {
            const SdpWalkOut &r = out.rev[x];
            if (r.status != SDP_OK) return false;
            h.max_start.score = r.score; h.max_start.q = r.q; h.max_start.t = r.t;
            h.max_start.runs.assign(out.runs.begin() + 2 * r.runs_off, out.runs.begin() + 2 * (r.runs_off + r.n_runs));
        } else {
            // SDP_Seed_find_start (sdp.c:640-659): the end minus every step of the path
            h.max_start.q = h.max_end.q; h.max_start.t = h.max_end.t;
            for (size_t r = 0; r + 1 < h.max_end.runs.size(); r += 2) {
                const c4gpu_transition &tr = m->transitions[h.max_end.runs[r]];
                h.max_start.q -= tr.advance_query * (int)h.max_end.runs[r + 1];
                h.max_start.t -= tr.advance_target * (int)h.max_end.runs[r + 1];
            }
        }
    }
    return true;
}

// SDP_Pair_next_path's single-pass loop (sdp.c:743-815) in the loop of GAM_Result_SDP_create (gam.c:868-881): seeds by end
// score, the first whose path does not cross an earlier alignment of the pair; returns the number of alignments
inline int sdp_single_pass(const SdpFamilyInfo &fi, const c4gpu_model *model, const c4gpu_pair &pair, std::vector<SdpHostSeed> &seeds,
                           c4gpu_score threshold, int max_alignments, c4gpu_alignment *out) {
    const int n = (int)seeds.size();
    std::vector<SdpHostSeed *> by_score(n);
    for (int k = 0; k < n; k++) by_score[k] = &seeds[k];
    qsort(by_score.data(), n, sizeof(SdpHostSeed *), sdp_score_cmp);
    c4gpu_subopt *so = c4gpu_subopt_create(pair.query_len, pair.target_len);
    int pos = 0, n_out = 0;
    while (n_out < max_alignments) {
        c4gpu_alignment *a = nullptr;
        while (pos < n) {
            SdpHostSeed *b = by_score[pos++];
            if (b->max_end.score < threshold) { pos = n; break; }
            // SDP_Pair_find_path + SDP_Pair_add_traceback, sdp.c:640-734
            c4gpu_alignment cand;
            memset(&cand, 0, sizeof cand);
            int cap = 0;
            cand.score = b->max_end.score;
            cand.region.query_start = b->max_start.q; cand.region.target_start = b->max_start.t;
            cand.region.query_length = b->max_end.q - b->max_start.q;
            cand.region.target_length = b->max_end.t - b->max_start.t;
            cand.valid = 1;
            const std::vector<unsigned> &fr = b->max_end.runs;
            if (!fi.bnd) {
                // the reverse path from the cell that leaves START towards the seed, without its last operation (into END)
                const std::vector<unsigned> &rr = b->max_start.runs;
                for (size_t r = 0; r + 1 < rr.size(); r += 2) {
                    const int cnt = (int)rr[r + 1] - (r + 2 == rr.size() ? 1 : 0);
                    if (cnt > 0) c4h::alignment_add(&cand, &cap, (int)rr[r], cnt);
                }
                // the forward path from the seed to the end, without its first operation (out of START): last in walk order
                for (long long r = (long long)fr.size() - 2; r >= 0; r -= 2) {
                    const int cnt = (int)fr[r + 1] - (r + 2 == (long long)fr.size() ? 1 : 0);
                    if (cnt > 0) c4h::alignment_add(&cand, &cap, (int)fr[r], cnt);
                }
            } else {                                                            // boundary: the forward path, whole
                for (long long r = (long long)fr.size() - 2; r >= 0; r -= 2) c4h::alignment_add(&cand, &cap, (int)fr[r], (int)fr[r + 1]);
            }
            // SubOpt_overlaps_alignment, subopt.c:177-203
            bool overlaps = false;
            int qp = cand.region.query_start, tp = cand.region.target_start;
            for (int k = 0; k < cand.n_ops && !overlaps; k++) {
                const c4gpu_transition &tr = model->transitions[cand.op_transition[k]];
                if (tr.label == C4GPU_LABEL_MATCH) {
                    for (int j = 0; j < cand.op_length[k] && !overlaps; j++) {
                        for (const auto &pt : so->points)
                            if (pt.second >= qp && pt.second < qp + tr.advance_query && pt.first >= tp &&
                                pt.first < tp + tr.advance_target) { overlaps = true; break; }
                        qp += tr.advance_query; tp += tr.advance_target;
                    }
                } else {
                    qp += tr.advance_query * cand.op_length[k]; tp += tr.advance_target * cand.op_length[k];
                }
            }
            if (overlaps) { c4gpu_alignment_clear(&cand); continue; }
            a = &out[n_out];
            *a = cand;
            break;
        }
        if (!a) break;
        c4gpu_subopt_add_alignment(so, model, a);                               // GAM_Result_add_alignment, gam.c:673
        n_out++;
    }
    c4gpu_subopt_destroy(so);
    return n_out;
}

}  // namespace c4sdp
