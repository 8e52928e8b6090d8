/* c4_oracle_seed.c — CPU restatement of the seeder's automaton walk.  TEST INFRASTRUCTURE ONLY (see c4_oracle.c's header): the
 * checker of c4gpu_seed_scan, never linked into or called from the product.  Compiled into libc4oracle.so by being included
 * from c4_oracle.c.
 *
 * What it restates (exonerate 2.4.0, src/comparison/seeder.c):
 *   Seeder_add_target (:852-915, untranslated branch :887-897): the masked target string is walked once through the
 *     automaton the queries built -- FSM_traverse (src/struct/fsm.c:186-198: one state change per symbol, the callback wherever
 *     the new node carries data) or Seeder_VFSM_traverse_single (:698-720: a symbol outside the alphabet resets the state, a
 *     leaf state with a word calls the same function) --
 *   Seeder_FSM_traverse_func (:649-695) at every position where a word ends: first the word's OWN seeds in list order
 *     (:675-680), then for each NEIGHBOUR word in list order that word's seeds in list order (:681-692), each one handed to
 *     Seeder_WordInfo_seed (:624-647) -> HSPset_seed_hsp(query position, target position) with
 *     target position = position of the word's last symbol - tpos_modifier (wordlen - 1, :293).
 * Every word of one seeder has the same length, so "the automaton is in a state that carries word w" is "the last `wordlen`
 * symbols are inside the alphabet and spell w" (the trie's failure links only ever lead to the longest proper suffix that is
 * a prefix of a word; a symbol outside the alphabet leads to the root): the walk below keeps the code of the last `wordlen`
 * symbols and the length of the run of in-alphabet symbols, and looks the code up.  Not restated: --saturatethreshold (off by
 * default, :663-673), translated targets (three walks of the same kind, :871-886), --wordambiguity > 1 (:722-777).
 * Pinned by tests/golden/seeds_*.jsonl: word tables read off the reference's own automaton and the calls its own walk made
 * (oracle/refdump.c --cmd seeds), tests/test_oracle_seed.py. */
#include <stdint.h>
#include <stdlib.h>

/* words: n_words codes in ASCENDING order (base-`width` numbers of the `wordlen` columns, most significant first; column 0 =
 * outside the alphabet never occurs in a code); seed_first[w] .. seed_first[w + 1]: the word's own seeds in list order,
 * (query, query position) pairs in seeds[2 * k], seeds[2 * k + 1]; nbr_first[w] .. nbr_first[w + 1]: its neighbour words
 * (indices into the word list) in list order.  symbols: the target as automaton columns.  out: (query, query position,
 * target position) triples in the order the reference's walk makes its HSPset_seed_hsp calls; returns their number (all
 * counted, only the first `cap` written). */
int64_t oracle_seed_walk(int32_t width, int32_t wordlen, int32_t n_words, const uint64_t *codes, const int32_t *seed_first,
                         const int32_t *seeds, const int32_t *nbr_first, const int32_t *nbrs, const uint8_t *symbols,
                         int32_t n_symbols, int32_t tpos_modifier, int32_t *out, int64_t cap) {
    uint64_t modulus = 1, code = 0;
    int64_t n_out = 0;
    int32_t run = 0;
    for (int32_t k = 0; k < wordlen; k++) modulus *= (uint64_t)width;
    for (int32_t i = 0; i < n_symbols; i++) {
        const int32_t c = symbols[i];
        if (!c) { run = 0; code = 0; continue; }                  /* seeder.c:706-709 / the FSM's edge back to the root */
        code = (code * (uint64_t)width + (uint64_t)c) % modulus;
        if (++run < wordlen) continue;
        int32_t lo = 0, hi = n_words - 1, w = -1;
        while (lo <= hi) {
            const int32_t mid = lo + (hi - lo) / 2;
            if (codes[mid] == code) { w = mid; break; }
            if (codes[mid] < code) lo = mid + 1; else hi = mid - 1;
        }
        if (w < 0) continue;
        const int32_t tpos = i - tpos_modifier;                   /* seeder.c:676 */
        for (int32_t s = seed_first[w]; s < seed_first[w + 1]; s++) {        /* the word's own seeds, :675-680 */
            if (n_out < cap) { out[3 * n_out] = seeds[2 * s]; out[3 * n_out + 1] = seeds[2 * s + 1]; out[3 * n_out + 2] = tpos; }
            n_out++;
        }
        for (int32_t b = nbr_first[w]; b < nbr_first[w + 1]; b++) {          /* then each neighbour's seeds, :681-692 */
            const int32_t v = nbrs[b];
            for (int32_t s = seed_first[v]; s < seed_first[v + 1]; s++) {
                if (n_out < cap) { out[3 * n_out] = seeds[2 * s]; out[3 * n_out + 1] = seeds[2 * s + 1]; out[3 * n_out + 2] = tpos; }
                n_out++;
            }
        }
    }
    return n_out;
}
