// sdp_sim.hip — TEST INFRASTRUCTURE: the sparse SDP wavefront (exonerate_amd/csrc/c4_sdp_wave.h) driven on the CPU.
//
// The per-lane cell function (Eval::cell), the stream / record layout, the walk (walk_path) and the host side
// (c4_sdp_host.h) are the product's own code, compiled for the host; only the wave-level driver is restated here with plain
// loops over 64 lanes (what the kernel does with DPP, ballots and loads issued ahead).  tests/test_sdp_sim.py runs it
// against the pinned oracle (oracle/c4_oracle_sdp.c) on the reference's SDP vector sets and on seeded random pairs, in
// the build container, where there is no GPU: it checks the ALGORITHM (strips, events, carried rows, mirrored boundary
// records, span stores, order keys, the walk); the -m gpu tests check the kernels.  Nothing in the product links this file.
#include <cstdio>
#include <vector>
#include "../exonerate_amd/csrc/c4_sdp_host.h"
#include "../exonerate_amd/csrc/c4_launch.h"

using namespace c4sdp;

namespace {

struct SimArena {
    std::vector<uint8_t> mem;
    unsigned next = 0, n_chunks = 0;
};

template <class M, bool FWD, bool BND>
int sim_pass(const SdpLaunch &A, SimArena &ar, int jx, long long *steps) {
    using P = Plan<M, FWD>;
    using L = Layout<M, FWD, BND>;
    using LR = Layout<M, false, BND>;
    using E = Eval<M, FWD, BND>;
    using C = SCell<M::NS>;
    constexpr int NS = M::NS;
    const SdpJob &job = A.jobs[jx];
    const int Q = job.Q, T = job.T, n_strips = job.n_strips;
    const KParams *kp = A.kp;
    SdpBest *best = (FWD ? A.best_fwd : A.best_rev) + job.seed_off;
    if (L::CBK) for (int i = 0; i < job.n_seeds; i++) best[i] = SdpBest{LOW, 0xffffffffu, 0xffffffffu, 0};
    int cv[M::NC];
    for (int i = 0; i < M::NC; i++) cv[i] = kp->calc_value[i];
    const uint8_t *qc = A.qcode + job.q_off, *tc = A.tcode + job.t_off;
    const int *ssb = A.ss ? A.ss + job.t_off : nullptr;
    const uint16_t *tn4 = A.tn4 ? A.tn4 + job.t_off : nullptr;
    int *tab_rec = A.tabs + job.tab_off[L::ST_REC], *tab_carry = A.tabs + job.tab_off[L::ST_CARRY];
    int *dir_rec = A.dirs + job.dir_off + L::ST_REC * (n_strips + 1), *dir_carry = A.dirs + job.dir_off + L::ST_CARRY * (n_strips + 1);
    const int *tab_rrec = A.tabs + job.tab_off[ST_REVREC];
    const int *dir_rrec = A.dirs + job.dir_off + ST_REVREC * (n_strips + 1);
    int rec_count = 0, carry_count = 0, prev_carry_first = 0;
    const SdpDSeed *seeds = A.seeds + job.seed_off;
    int sp = 0;
    const int ubase0 = strip_ubase0(Q, n_strips, FWD && BND);
    const int tlast = T > 0 ? T - 1 : 0;
    auto new_chunk = [&](int *tab, int cap, int slot) -> bool {
        const unsigned ch = ar.next++;
        if (ch >= ar.n_chunks || slot >= cap) return false;
        tab[slot] = (int)ch;
        return true;
    };
    auto dead_cell = [&](C &x) { for (int s = 0; s < NS; s++) { x.sc[s] = LOW; x.mx[s] = 0; x.sd[s] = 0; x.sh[s] = 0; } };
    for (int strip = 0; strip < n_strips; strip++) {
        dir_rec[strip] = rec_count; dir_carry[strip] = carry_count;
        static thread_local C cur[64], own[64][M::MAXAT + 1], up[64][M::MAXAT + 2], nup[64];
        static thread_local SpanCache cache[64][E::NSPA];
        for (int l = 0; l < 64; l++) {
            dead_cell(cur[l]);
            for (int a = 0; a <= M::MAXAT; a++) dead_cell(own[l][a]);
            for (int a = 0; a <= M::MAXAT + 1; a++) dead_cell(up[l][a]);
            for (int i = 0; i < E::NSPA; i++) cache[l][i] = SpanCache{0, 0, 0, 0, 0, 0, 0};
        }
        while (sp < job.n_seeds && seeds[sp].strip < strip) sp++;
        auto seed_c = [&]() -> int { return (!(FWD && BND) && sp < job.n_seeds && seeds[sp].strip == strip) ? seeds[sp].c : SDP_NO_EVENT; };
        int ci = 0, ci_end = 0;
        if (strip > 0) { ci = prev_carry_first; ci_end = carry_count; }
        prev_carry_first = carry_count;
        auto carry_entry = [&](int idx) -> const int * { return reinterpret_cast<const int *>(stream_entry(A.arena, tab_carry, idx, L::CPC_LOG, L::CENT_BYTES)); };
        auto carry_v = [&]() -> int { return ci < ci_end ? carry_entry(ci)[0] : SDP_NO_EVENT; };
        int bi = -1, bi_lo = 0;
        if (FWD && BND) { bi_lo = dir_rrec[n_strips - 1 - strip]; bi = dir_rrec[n_strips - strip] - 1; }
        auto bnd_rec = [&]() -> const uint8_t * { return stream_entry(A.arena, tab_rrec, bi, LR::RPC_LOG, LR::REC_BYTES); };
        auto bnd_cf = [&]() -> int { return (FWD && BND && bi >= bi_lo) ? T + 63 - *reinterpret_cast<const int *>(bnd_rec()) : SDP_NO_EVENT; };
        auto next_event = [&]() -> int { return std::min(seed_c(), std::min(carry_v(), bnd_cf())); };
        int c = next_event();
        int dead_run = M::MAXAT + 2;
        while (c <= T + 63) {
            (*steps)++;
            // rings
            const bool has_cin = carry_v() == c;
            for (int l = 0; l < 64; l++) {
                if (l > 0) nup[l] = cur[l - 1];
                else {
                    dead_cell(nup[0]);
                    if (has_cin) {
                        const int *e = carry_entry(ci);
                        for (int s = 0; s < NS; s++)
                            if (P::exported(s)) {
                                const int X = 2 + P::exported_index(s) * L::CW;
                                nup[0].sc[s] = e[X]; nup[0].mx[s] = e[X + 1]; nup[0].sd[s] = e[X + 2];
                                if (L::SH) nup[0].sh[s] = e[X + 3];
                            }
                    }
                }
            }
            if (has_cin) { ci++; dead_run = 0; }
            for (int l = 0; l < 64; l++) {
                for (int a = M::MAXAT + 1; a >= 2; a--) up[l][a] = up[l][a - 1];
                up[l][1] = nup[l];
                // only exported states cross lanes on the device: the others must never be read from the ring above
                for (int s = 0; s < NS; s++) if (!P::exported(s)) { up[l][1].sc[s] = 0x7f7f7f7f; up[l][1].mx[s] = 0x7f7f7f7f; }
                for (int a = M::MAXAT; a >= 2; a--) own[l][a] = own[l][a - 1];
                own[l][1] = cur[l];
            }
            // seeds
            int seed_lane_score[64], seed_lane_id[64];
            bool seed_lane[64];
            for (int l = 0; l < 64; l++) seed_lane[l] = false;
            while (seed_c() == c) {
                const SdpDSeed &s = seeds[sp];
                int val = s.val;
                if (FWD && !BND) val = A.best_rev[job.seed_off + s.sid].score - s.val;
                seed_lane[s.lane] = true; seed_lane_score[s.lane] = val; seed_lane_id[s.lane] = s.sid;
                sp++;
            }
            if (FWD && BND && bnd_cf() == c) {
                const int *bv = reinterpret_cast<const int *>(bnd_rec() + REC_HEAD);
                for (int l = 0; l < 64; l++)
                    if (bv[63 - l]) { seed_lane[l] = true; seed_lane_score[l] = 0; seed_lane_id[l] = bv[63 - l] - 1; }
                bi--;
            }
            unsigned long long alive = 0;
            typename E::Out outs[64];
            for (int l = 0; l < 64; l++) {
                const int u = ubase0 + 64 * strip + l, v = c - l;
                typename E::In in;
                in.u = u; in.v = v; in.Q = Q; in.T = T;
                in.inside = u >= 0 && u <= Q && v >= 0 && v <= T;
                in.seed_here = seed_lane[l]; in.seed_score = seed_lane[l] ? seed_lane_score[l] : 0; in.seed_id = seed_lane[l] ? seed_lane_id[l] : 0;
                int qpos = FWD ? u - 1 : Q - u;
                qpos = qpos < 0 ? 0 : (qpos > Q - 1 ? (Q > 0 ? Q - 1 : 0) : qpos);
                in.qrow24 = 24 * (int)qc[qpos];
                int mpos = FWD ? v - P::match_at() : T - v, spos = FWD ? v - P::splice_at() : T - v;
                mpos = mpos < 0 ? 0 : (mpos > tlast ? tlast : mpos);
                spos = spos < 0 ? 0 : (spos > tlast ? tlast : spos);
                in.tv.mcode = P::has_match() ? (int)tc[mpos] : 0;
                for (int k = 0; k < 4; k++) in.tv.ss[k] = P::uses_ss(k) ? ssb[(long long)k * A.ss_stride + spos] : 0;
                in.mscore = P::has_match() ? kp->submat[in.qrow24 + in.tv.mcode] : 0;
                in.dropoff = A.dropoff; in.min_intron = kp->min_intron; in.max_intron = kp->max_intron; in.span_max_target = kp->max_intron;
                in.tn4 = tn4; in.cv = cv;
                E::cell(cur[l], own[l], up[l], cache[l], in, outs[l], kp, [&](bool fire, int sid, int score, auto K_) {
                    constexpr int k = K_;
                    if (!fire) return;
                    const unsigned long long order = ((unsigned long long)(unsigned)v * (unsigned)(Q + 1) + (unsigned)u) * (unsigned)M::NT + (unsigned)(M::NT - 1 - k);
                    const unsigned hi = (unsigned)(order >> 32), lo = (unsigned)order;
                    SdpBest &e = best[sid];
                    if (score > e.score || (score == e.score && (hi < e.ohi || (hi == e.ohi && lo < e.olo)))) e = SdpBest{score, hi, lo, 0};
                });
                if (outs[l].alive) alive |= 1ull << l;
            }
            if (alive) {
                if ((rec_count & ((1 << L::RPC_LOG) - 1)) == 0 && !new_chunk(tab_rec, job.tab_cap[L::ST_REC], rec_count >> L::RPC_LOG)) return SDP_FAIL_ARENA;
                uint8_t *r = stream_entry(A.arena, tab_rec, rec_count, L::RPC_LOG, L::REC_BYTES);
                *reinterpret_cast<int *>(r) = c;
                unsigned *ln = reinterpret_cast<unsigned *>(r + REC_HEAD);
                for (int l = 0; l < 64; l++) {
                    if (L::TB) {
                        for (int w = 0; w < L::PTW; w++) ln[w * 64 + l] = outs[l].ptw[w];
                        for (int i = 0; i < L::NSP; i++) ln[(L::PTW + i) * 64 + l] = (unsigned)outs[l].tf[i];
                    } else ln[l] = (unsigned)outs[l].bnd;
                }
                rec_count++;
                if (P::n_exported() > 0 && strip + 1 < n_strips && (alive >> 63)) {
                    if ((carry_count & ((1 << L::CPC_LOG) - 1)) == 0 && !new_chunk(tab_carry, job.tab_cap[L::ST_CARRY], carry_count >> L::CPC_LOG)) return SDP_FAIL_ARENA;
                    int *e = reinterpret_cast<int *>(stream_entry(A.arena, tab_carry, carry_count, L::CPC_LOG, L::CENT_BYTES));
                    e[0] = c - 63; e[1] = 0;
                    for (int s = 0; s < NS; s++)
                        if (P::exported(s)) {
                            const int X = 2 + P::exported_index(s) * L::CW;
                            e[X] = cur[63].sc[s]; e[X + 1] = cur[63].mx[s]; e[X + 2] = cur[63].sd[s];
                            if (L::SH) e[X + 3] = cur[63].sh[s];
                        }
                    carry_count++;
                }
                dead_run = 0;
            } else dead_run++;
            c++;
            if (dead_run > M::MAXAT + 1) {
                const int e = next_event();
                if (e == SDP_NO_EVENT) break;
                if (e > c) c = e;
            }
        }
    }
    dir_rec[n_strips] = rec_count; dir_carry[n_strips] = carry_count;
    return SDP_OK;
}

template <class M, bool BND>
int sim_pair(const SdpFamilyInfo &fi, const c4gpu_model *model, const KParams &kp, const uint8_t *qcode, const uint8_t *tcode,
             const int *ss, long long ss_stride, const uint16_t *tn4, const c4gpu_pair &pair, const c4gpu_hsp *hsps, int n_hsps, int qa, int ta,
             int dropoff, int threshold, int max_alignments, c4gpu_alignment *out, long long *steps, int arena_chunks) {
    std::vector<std::vector<SdpHostSeed>> seeds(1);
    sdp_seed_list(hsps, n_hsps, qa, ta, seeds[0]);
    if (seeds[0].empty()) return 0;
    SdpHostPlan plan;
    std::vector<long long> qo(1, 0), to(1, 0);
    std::vector<char> active(1, 1);
    sdp_make_plan(fi, &pair, 1, qo, to, seeds, active, (unsigned)arena_chunks, plan);
    SimArena ar;
    ar.n_chunks = (unsigned)arena_chunks;
    ar.mem.assign((size_t)arena_chunks << CHUNK_LOG, 0xAB);
    std::vector<int> tabs(plan.tabs_total, -1), dirs(plan.dirs_total, 0), status(1, 0);
    const int ns = (int)plan.seed_job.size();
    std::vector<SdpBest> best_rev(ns), best_fwd(ns);
    SdpLaunch A;
    memset(&A, 0, sizeof A);
    A.kp = &kp; A.jobs = plan.jobs.data(); A.qcode = qcode; A.tcode = tcode; A.ss = ss; A.ss_stride = ss_stride; A.tn4 = tn4;
    A.arena = ar.mem.data(); A.tabs = tabs.data(); A.dirs = dirs.data(); A.best_rev = best_rev.data(); A.best_fwd = best_fwd.data();
    A.status = status.data(); A.dropoff = dropoff; A.n_chunks = (unsigned)arena_chunks;
    A.seeds = plan.rev_seeds.data();
    int st = sim_pass<M, false, BND>(A, ar, 0, steps);
    if (st != SDP_OK) return -10 - st;
    A.seeds = plan.fwd_seeds.data();
    st = sim_pass<M, true, BND>(A, ar, 0, steps);
    if (st != SDP_OK) return -10 - st;
    SdpBackendOut bo;
    bo.status.assign(1, SDP_OK);
    bo.fwd.resize(ns); bo.rev.resize(ns);
    auto walk = [&](const SdpWalkTab &W, const SdpBest &b, SdpWalkOut &w) {
        w = SdpWalkOut{LOW, 0, 0, 0, SDP_OK, 0, 0};
        if (b.score == LOW) return;
        w.score = b.score;
        w.runs_off = (long long)bo.runs.size() / 2;
        int last = -1;
        w.status = walk_path(W, plan.jobs[0], ar.mem.data(), tabs.data(), dirs.data(), b, &w.q, &w.t, [&](int id, int cnt) {
            if (id != last) { bo.runs.push_back((unsigned)id); bo.runs.push_back(0); last = id; w.n_runs++; }
            bo.runs.back() += (unsigned)cnt;
        });
    };
    for (int x = 0; x < ns; x++) {
        if (best_fwd[x].score < threshold) { bo.fwd[x] = SdpWalkOut{best_fwd[x].score, 0, 0, 0, SDP_OK, 0, 0}; bo.rev[x] = bo.fwd[x]; continue; }
        walk(fi.walk_fwd, best_fwd[x], bo.fwd[x]);
        if (!BND) walk(fi.walk_rev, best_rev[x], bo.rev[x]);
    }
    if (!sdp_collect(fi, model, plan, bo, 0, seeds[0])) return -2;
    return sdp_single_pass(fi, model, pair, seeds[0], threshold, max_alignments, out);
}

}  // namespace

// One pair through the simulated passes: the loop of GAM_Result_SDP_create on its HSPs (same contract as oracle_sdp in
// oracle/c4_oracle_sdp.c).  qcode / tcode / ss / tn4: the coded arrays the device builds (ResidentSeqs), here made by the
// caller.  Returns the number of alignments, < 0 on failure; *steps: wave steps executed (both passes).
extern "C" int sdpsim_pair(const c4gpu_model *model, const c4gpu_params *params, int family, const uint8_t *qcode, const uint8_t *tcode,
                           const int32_t *ss, const uint16_t *tn4, const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                           const c4gpu_hsp *hsps, int32_t n_hsps, int32_t qa, int32_t ta, int32_t dropoff, int32_t threshold,
                           int32_t max_alignments, c4gpu_alignment *out, int64_t *steps, int32_t arena_chunks) {
    KParams kp;
    make_kparams(model, params, &kp);
    c4gpu_pair pair;
    memset(&pair, 0, sizeof pair);
    pair.query = query; pair.query_len = qlen; pair.target = target; pair.target_len = tlen;
    long long st = 0;
    int n = -1;
    const long long stride = (long long)tlen + 64;
    switch (family) {
        case c4k::FAM_AFFINE: { static const SdpFamilyInfo fi = make_family_info<AffineDesc, false>();
            n = sim_pair<AffineDesc, false>(fi, model, kp, qcode, tcode, ss, stride, tn4, pair, hsps, n_hsps, qa, ta, dropoff, threshold, max_alignments, out, &st, arena_chunks); break; }
        case c4k::FAM_PROTEIN2DNA: { static const SdpFamilyInfo fi = make_family_info<Protein2DnaDesc, false>();
            n = sim_pair<Protein2DnaDesc, false>(fi, model, kp, qcode, tcode, ss, stride, tn4, pair, hsps, n_hsps, qa, ta, dropoff, threshold, max_alignments, out, &st, arena_chunks); break; }
        case c4k::FAM_EST2GENOME: { static const SdpFamilyInfo fi = make_family_info<Est2GenomeDesc, true>();
            n = sim_pair<Est2GenomeDesc, true>(fi, model, kp, qcode, tcode, ss, stride, tn4, pair, hsps, n_hsps, qa, ta, dropoff, threshold, max_alignments, out, &st, arena_chunks); break; }
        case c4k::FAM_PROTEIN2GENOME: { static const SdpFamilyInfo fi = make_family_info<Protein2GenomeDesc, true>();
            n = sim_pair<Protein2GenomeDesc, true>(fi, model, kp, qcode, tcode, ss, stride, tn4, pair, hsps, n_hsps, qa, ta, dropoff, threshold, max_alignments, out, &st, arena_chunks); break; }
        default: return -3;
    }
    if (steps) *steps = st;
    return n;
}
