// c4_internal.h — declarations shared by the host translation units of libc4gpu.so
#pragma once
#include <set>
#include <string>
#include <utility>
#include <vector>
#include "c4gpu.h"

// SubOpt (src/c4/subopt.h:33-48): the reference keeps the points in a RangeTree; all it ever asks of it is
// membership and "every point inside a rectangle", which a sorted, duplicate-free array keyed (target, query)
// answers in the order SubOpt_Index_create sorts them (subopt.c:239-248,268).
struct c4gpu_subopt {
    int32_t query_length, target_length;
    std::vector<std::pair<int32_t, int32_t>> points;   // (target_pos, query_pos), sequence coordinates, sorted
    int32_t path_count;
    // merge a batch of new points (any order, duplicates allowed) into the sorted array
    void merge(std::vector<std::pair<int32_t, int32_t>> &fresh);
};

namespace c4h {
bool use_reduced_space(const c4gpu_model *m, const c4gpu_region *r, int dpmemory_mb);
int  checkpoint_rows(const c4gpu_model *m, const c4gpu_region *r, int dpmemory_mb);
void alignment_add(c4gpu_alignment *a, int *cap, int transition, int length);
void set_error(const std::string &msg);
// points of `so` inside `r` (both ends inclusive: RangeTree_find is called with the lengths + 1,
// subopt.c:258-261), in region coordinates, as (target, query) in ascending order
void subopt_region_points(const c4gpu_subopt *so, const c4gpu_region &r,
                          std::vector<std::pair<int32_t, int32_t>> &out);
}  // namespace c4h
