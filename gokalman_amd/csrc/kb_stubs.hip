// kb_stubs.hip -- entry points whose kernels are not built yet (replaced file by file).
#include "kb_internal.h"
namespace kb {
int launch_materialise(const Batch &, const void *, bool, void *) { set_error("getter kernels not built"); return KB_ERR_UNSUPPORTED; }
int launch_within_nsigma(const Batch &, const void *, double, uint8_t *) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int launch_init(Batch &b, int *not_pd) { *not_pd = 0; if (b.kind == KB_VANILLA || b.kind == KB_VANILLA_PREDICT) return KB_OK; set_error("kind %d not built", b.kind); return KB_ERR_UNSUPPORTED; }
int launch_refresh(Batch &, int, int *not_pd) { *not_pd = 0; return KB_OK; }
int launch_squareroot(const Batch &, const StepArgs &) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int launch_information(const Batch &, const StepArgs &) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int launch_srif(const Batch &, const StepArgs &) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int launch_hybrid(const Batch &, const StepArgs &) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int launch_mc(const Batch &, const StepArgs &, const void *, int, double *) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
}
using namespace kb;
extern "C" {
int kb_prepare(kb_batch *, const double *, const double *, int64_t, int) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int kb_prepare_dev(kb_batch *, const void *, const void *, int64_t) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int kb_prepare_pnt(kb_batch *, const double *, int64_t, int) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int kb_update_nl(kb_batch *, const double *, int, const double *, int) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int kb_update_nl_dev(kb_batch *, const void *, const void *, int64_t) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int kb_predict_nl(kb_batch *) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int kb_set_noise_kind(kb_batch *, int, uint64_t) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int kb_noise_sample(kb_batch *, int64_t, int64_t, int64_t, int, double *) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int kb_mc_run(kb_batch *, int, const double *, int, int64_t, double *) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
int kb_mc_stats(const double *, int, int, int64_t, double *, double *) { set_error("not built"); return KB_ERR_UNSUPPORTED; }
}
