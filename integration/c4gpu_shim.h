/* c4gpu_shim.h — what the two files of the exonerate-gpu shim share (c4gpu_shim.c: the Viterbi plug-in table and
 * the exhaustive batching seam; c4gpu_bsdp.c: the heuristic / BSDP seam). */
#ifndef INCLUDED_C4GPU_SHIM_H
#define INCLUDED_C4GPU_SHIM_H

#include "c4.h"
#include "ungapped.h"
#include "optimal.h"
#include "gam.h"
#include "comparison.h"
#include "hspset.h"
#include "c4gpu.h"

c4gpu_ctx *shim_get_ctx(void);
const gchar *shim_env(const gchar *name);   /* a C4GPU_* variable as it was when the process started: never getenv (c4gpu_shim.c) */
gint shim_device_ordinal(void);        /* --gpudevice / C4GPU_DEVICE: for a second context of the same device */
c4gpu_ctx *shim_ctx_nowait(void);       /* NULL while the device is still being opened (small work does not wait) */
void shim_mark(const gchar *what);
gint shim_batch_size(void);
/* closed C4_Model -> c4gpu_model; allow_span: BSDP's span models (cell_start_func / cell_end_func become matrices) */
gboolean shim_flatten_any(C4_Model *m, Ungapped_Data *ud, c4gpu_model *out, gboolean allow_span);
void shim_params(Ungapped_Data *ud, c4gpu_params *p);
/* scoring data for calls that have no model at hand (HSP seeding): Match_ArgumentSet's matrices and translation */
void shim_hsp_params(c4gpu_params *p);
/* c4gpu_hsp.c */
void shim_hsp_report(void);
/* c4gpu_bsdp.c */
void shim_bsdp_flush(void);
void shim_bsdp_report(void);
Alignment *shim_bsdp_find_path(Optimal *optimal, Region *region, SubOpt *subopt);
/* c4gpu_seed.c */
gboolean shim_seed_recording(HSPset *hsp_set, guint query_start, guint target_start);
void shim_seed_report(void);
/* c4gpu_sdp.c */
gboolean shim_sdp_collect(GAM *gam, Comparison *comparison);
gboolean shim_sdp_replaying(void);
gboolean shim_sdp_busy(void);
void shim_sdp_flush(void);
void shim_sdp_report(void);

#endif /* INCLUDED_C4GPU_SHIM_H */
