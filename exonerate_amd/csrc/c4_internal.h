// c4_internal.h — declarations shared by the host translation units of libc4gpu.so
#pragma once
#include <string>
#include "c4gpu.h"

namespace c4h {
bool use_reduced_space(const c4gpu_model *m, const c4gpu_region *r, int dpmemory_mb);
int  checkpoint_rows(const c4gpu_model *m, const c4gpu_region *r, int dpmemory_mb);
void alignment_add(c4gpu_alignment *a, int *cap, int transition, int length);
void set_error(const std::string &msg);
}  // namespace c4h
