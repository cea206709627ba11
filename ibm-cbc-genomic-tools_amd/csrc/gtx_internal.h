// gtx_internal.h -- hooks of gtx_capi.hip that the multi-GPU layer (gtx_group.hip) builds on; not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "gtx.h"

extern "C" {
int gtxi_count_finish(gtx_ctx *c, void **d_out);       // close the open count stream: result in the context's HBM vector (enqueued)
int gtxi_coverage_finish(gtx_ctx *c, void **d_out);
void gtxi_fetch_info(gtx_ctx *c, gtx_count_info *info); // after a wait for the stream
int gtxi_scan_enqueue(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, const int32_t *class_len, int32_t n_classes,
                      int32_t step, int32_t size, char prep, uint32_t flags, const int64_t *class_offsets, void **d_out, int64_t *extent);
hipStream_t gtxi_stream(gtx_ctx *c);
int gtxi_device(gtx_ctx *c);
void gtxi_set_error(gtx_ctx *c, const char *msg);
}
