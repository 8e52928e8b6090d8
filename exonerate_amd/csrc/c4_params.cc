// c4_params.cc — default scoring data of the engine (host side, plain C++).
//
// Mirrors what the reference keeps in static ArgumentSets + built-in tables:
//   Match_ArgumentSet   (src/comparison/match.c:36-66)  : nucleic / blosum62, Translate_create(FALSE)
//   Submat index        (src/sequence/submat.c:27-61)   : "ARNDCQEGHILKMFPSTWYVBZX*", U->C
//   Translate tables    (src/sequence/translate.c:58-110): 4-bit IUPAC masks, standard code
//   Affine / Intron / Frameshift defaults (affine.c:24-35, intron.c:24-32, frameshift.c)
//   SplicePredictor_create (src/sequence/splice.c:242-295): log-odds PSSMs
// The matrices themselves are data (c4_params_data.inc, see tools/gen_params_data.py).
#include <cmath>
#include <cstring>
#include <cctype>

#include "c4gpu.h"
#include "c4_params_data.inc"

namespace {

void build_submat_index(uint8_t *index) {
    for (int i = 0; i < 256; i++) index[i] = 24;
    for (int k = 0; kSubmatOrder[k]; k++) {
        unsigned char c = (unsigned char)kSubmatOrder[k];
        index[c] = (uint8_t)k;
        index[(unsigned char)tolower(c)] = (uint8_t)k;
    }
    // selenocysteine is scored as cysteine (submat.c: "U is just treated as a cysteine")
    index[(unsigned char)'U'] = index[(unsigned char)'u'] = index[(unsigned char)'C'];
}

void build_dna_submat(const uint8_t *index, int32_t m[24][24]) {
    memset(m, 0, sizeof(int32_t) * 24 * 24);
    // NUC.4.4 over the IUPAC letters; X is scored as N.
    const char *extra = "X";
    for (int a = 0; a < 16; a++) {
        char ca = a < 15 ? kNuc44Order[a] : extra[0];
        int ra = a < 15 ? a : 14;
        for (int b = 0; b < 16; b++) {
            char cb = b < 15 ? kNuc44Order[b] : extra[0];
            int rb = b < 15 ? b : 14;
            m[index[(unsigned char)ca]][index[(unsigned char)cb]] = kNuc44[ra][rb];
        }
    }
}

// translate.c:58-66 — residues to 4-bit masks over "-GARTKWDCSMVYBHN" (bit0 G, bit1 A, bit2 T, bit3 C)
void build_nt2d(uint8_t *nt2d) {
    static const char nt[] = "-GARTKWDCSMVYBHN";
    memset(nt2d, 0, 256);
    for (int i = 0; i < 16; i++) {
        nt2d[(unsigned char)nt[i]] = (uint8_t)i;
        nt2d[(unsigned char)tolower(nt[i])] = (uint8_t)i;
    }
    nt2d[(unsigned char)'X'] = nt2d[(unsigned char)'x'] = nt2d[(unsigned char)'N'];
    nt2d[(unsigned char)'U'] = nt2d[(unsigned char)'u'] = nt2d[(unsigned char)'T'];
}

// translate.c:86-110 with Translate_create(FALSE): an ambiguous codon whose expansions agree gives that
// residue, disagreeing expansions give 'X', an empty mask gives '-'.  `trans` holds an index into `aa`;
// only aa[trans[.]] is observable (Translate_base, translate.h:73-76).
void build_translation(uint8_t *trans, uint8_t *aa) {
    static const char aa_set[] = "-ARNDCQEGHILKMFPSTWYV*";   // 22 symbols, then 18 x 'X'
    memset(aa, 'X', 40);
    memcpy(aa, aa_set, 22);
    // NCBI standard code in TCAG order -> internal order where base index = mask bit (G,A,T,C)
    static const char ncbi[] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    static const int to_ncbi[4] = {3, 2, 0, 1};
    char code[64];
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++)
            for (int c = 0; c < 4; c++)
                code[(a << 4) | (b << 2) | c] = ncbi[(to_ncbi[a] << 4) | (to_ncbi[b] << 2) | to_ncbi[c]];
    for (int x = 0; x < 16; x++)
        for (int y = 0; y < 16; y++)
            for (int z = 0; z < 16; z++) {
                int found = -1;   // -1 none yet, -2 ambiguous
                for (int a = 0; a < 4; a++) {
                    if (!(x & (1 << a))) continue;
                    for (int b = 0; b < 4; b++) {
                        if (!(y & (1 << b))) continue;
                        for (int c = 0; c < 4; c++) {
                            if (!(z & (1 << c))) continue;
                            int idx = (int)(strchr(aa_set, code[(a << 4) | (b << 2) | c]) - aa_set);
                            if (found == -1) found = idx;
                            else if (found != idx) found = -2;
                        }
                    }
                }
                trans[x | (y << 4) | (z << 8)] = (uint8_t)(found == -1 ? 0 : found == -2 ? 39 : found);
            }
}

// splice.c:242-295
void build_splice(c4gpu_splice_model *sp, int type) {
    const bool is5 = (type == C4GPU_SS5_FORWARD || type == C4GPU_SS5_REVERSE);
    const bool rev = (type == C4GPU_SS5_REVERSE || type == C4GPU_SS3_REVERSE);
    memset(sp, 0, sizeof(*sp));
    float raw[C4GPU_SPLICE_MAX_LEN][4];
    if (is5) {
        sp->model_length = kSplice5Length;
        sp->splice_after = 3;
        for (int i = 0; i < sp->model_length; i++)
            for (int j = 0; j < 4; j++) raw[i][j] = (float)kSplice5[i][j];
    } else {
        sp->model_length = kSplice3Length;
        sp->splice_after = 14 - 2;                       // splice.c:203-210
        for (int i = 0; i < sp->model_length; i++)
            for (int j = 0; j < 4; j++) raw[i][j] = (float)kSplice3[i][j];
    }
    if (rev) {                                           // splice.c:255-267
        for (int a = 0, z = sp->model_length - 1; a < z; a++, z--)
            for (int j = 0; j < 4; j++) { float s = raw[a][j]; raw[a][j] = raw[z][j]; raw[z][j] = s; }
        sp->splice_after = sp->model_length - sp->splice_after - 2;
    }
    for (int i = 0; i < 256; i++) sp->index[i] = 4;
    const char *fw = rev ? "TGCA" : "ACGT";              // splice.c:270-281
    for (int k = 0; k < 4; k++) {
        sp->index[(unsigned char)fw[k]] = (uint8_t)k;
        sp->index[(unsigned char)tolower(fw[k])] = (uint8_t)k;
    }
    for (int i = 0; i < sp->model_length; i++) {         // splice.c:282-289 (two float roundings)
        for (int j = 0; j < 4; j++) {
            volatile float p = (float)(((double)(float)(1 + raw[i][j])) / (25.0 + 1.0));
            volatile float l = (float)(log((double)p) * 1.5);
            sp->data[i][j] = l;
        }
        sp->data[i][4] = 0.0f;
    }
}

}  // namespace

extern "C" void c4gpu_params_default(c4gpu_params *out) {
    memset(out, 0, sizeof(*out));
    build_submat_index(out->submat_index);
    build_dna_submat(out->submat_index, out->dna_submat);
    for (int i = 0; i < 24; i++)
        for (int j = 0; j < 24; j++) out->protein_submat[i][j] = kBlosum62[i][j];
    build_nt2d(out->nt2d);
    build_translation(out->trans, out->aa);
    out->gap_open = -12;  out->gap_extend = -4;
    out->codon_gap_open = -18;  out->codon_gap_extend = -8;
    out->min_intron = 30;  out->max_intron = 200000;  out->intron_open_penalty = -30;
    out->frameshift_penalty = -28;
    for (int t = 0; t < 4; t++) build_splice(&out->splice[t], t);
}

// SplicePredictor_GTAGonly_create, splice.c:213-240
extern "C" void c4gpu_params_set_forcegtag(c4gpu_params *p, int on) {
    static const char pair[4][2] = {{'G', 'T'}, {'A', 'G'}, {'C', 'T'}, {'A', 'C'}};   // C4GPU_SS5_FORWARD, SS3_FORWARD, SS3_REVERSE, SS5_REVERSE
    for (int t = 0; t < 4; t++) {
        p->splice[t].gtag_only = on ? 1 : 0;
        p->splice[t].expect_one = (uint8_t)pair[t][0];
        p->splice[t].expect_two = (uint8_t)pair[t][1];
    }
}

// SplicePredictor_get_max_score, splice.c:399-410 (float accumulation, row order)
extern "C" float c4gpu_splice_max_score(const c4gpu_splice_model *sp) {
    float score = 0.0f;
    for (int i = 0; i < sp->model_length; i++) {
        float pos = sp->data[i][0];
        for (int j = 1; j < 4; j++)
            if (pos < sp->data[i][j]) pos = sp->data[i][j];
        score += pos;
    }
    return score;
}
