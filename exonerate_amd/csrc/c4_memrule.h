// c4_memrule.h — the reference's memory decisions as plain arithmetic that compiles for the host AND the device:
// which Viterbi passes a region gets (quadratic traceback or checkpoints) is part of the parity contract, and the
// sub-alignment jobs of a checkpoint pass are now listed on the device (c4_engine.hip, fused_reduced_paths), so the
// same rule has to be evaluated there, bit for bit.  Integer arithmetic apart from the reference's floating-point
// overflow probe, which is kept unfused (the probe compares a double sum with an unsigned one).
#pragma once
#include <stddef.h>

#if defined(__HIPCC__)
#define C4_HD __host__ __device__
#else
#define C4_HD
#endif

namespace c4h {

struct MemRule {                 // the four facts of a C4 model the rule reads (c4gpu_model fields of the same names)
    int max_query_advance, max_target_advance, n_states, total_shadow_designations;
};

// Matrix3d_size / Matrix4d_size (exonerate src/struct/matrix.c:74-100,137-171): index blocks + data with
// the "+= size % sizeof(pointer)" padding rule and the floating-point overflow probe.
C4_HD inline size_t matrix3d_bytes(int a, int b, int c, size_t cell) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const size_t P = sizeof(void *);
    unsigned long block = b * P + (unsigned long)b * (c * cell);
    double dblock = (double)(b * P) + (double)b * (double)(c * cell);
    block += block % P;
    dblock += (double)(block % P);
    unsigned long total = a * P + (unsigned long)a * block;
    double dtotal = (double)(a * P) + (double)a * dblock;
    return (dtotal - (double)total) > 1 ? 0 : total;
}

C4_HD inline size_t matrix4d_bytes(int a, int b, int c, int d, size_t cell) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const size_t P = sizeof(void *);
    unsigned long block = c * P + (unsigned long)c * (d * cell);
    double dblock = (double)(c * P) + (double)c * (double)(d * cell);
    block += block % P;
    dblock += (double)(block % P);
    unsigned long sheet = b * P + (unsigned long)b * block;
    double dsheet = (double)(b * P) + (double)b * dblock;
    sheet += sheet % P;
    dsheet += (double)(sheet % P);
    unsigned long total = a * P + (unsigned long)a * sheet;
    double dtotal = (double)(a * P) + (double)a * dsheet;
    return (dtotal - (double)total) > 1 ? 0 : total;
}

// Viterbi_get_row_size (viterbi.c:108-118): the matrix size passes through a gint
C4_HD inline size_t viterbi_row_bytes(const MemRule &m, int query_length, int cell_size) {
    const int mat = (int)matrix4d_bytes(m.max_target_advance + 1, query_length + 1, m.n_states, cell_size, 4 /* sizeof(C4_Score) */);
    if (!mat) return 0;
    return (size_t)24 /* sizeof(Viterbi_Row) */ + (size_t)mat;
}

// Viterbi_use_reduced_space (viterbi.c:128-151)
C4_HD inline bool use_reduced_space(const MemRule &m, int query_length, int target_length, int dpmemory_mb) {
    if (query_length <= m.max_query_advance * 6) return false;
    if (target_length <= m.max_target_advance * 6) return false;
    const size_t rows = viterbi_row_bytes(m, query_length, 1 + m.total_shadow_designations);
    const size_t traceback = matrix3d_bytes(query_length + 1, target_length + 1, m.n_states, sizeof(void *));
    const size_t limit = (size_t)(dpmemory_mb << 20);
    if (!rows || !traceback) return true;
    return rows + traceback > limit;
}

// Viterbi_checkpoint_rows (viterbi.c:207-218)
C4_HD inline int checkpoint_rows(const MemRule &m, int query_length, int target_length, int dpmemory_mb) {
    const size_t rows = viterbi_row_bytes(m, query_length, 1 + m.total_shadow_designations + 1);
    const int avail = (int)(((size_t)(dpmemory_mb << 20)) / rows - 1);
    const int max_rows = target_length / (m.max_target_advance << 1) - 2;
    if (avail < 1) return 1;
    return avail < max_rows ? avail : max_rows;
}

}  // namespace c4h
