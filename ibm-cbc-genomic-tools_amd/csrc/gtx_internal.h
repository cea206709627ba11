// gtx_internal.h -- hooks of gtx_capi.hip that the multi-GPU layer (gtx_group.hip) builds on; not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "gtx.h"

// device calls of a group whose exchange may be on its way at once: compact vectors and events are dealt in turn (call k takes slot
// k % GTXI_SHARE_SLOTS); call k + GTXI_SHARE_SLOTS waits for the exchange of call k
#define GTXI_SHARE_SLOTS 4
// the streams of a group's own that the device calls of one member take in turn, each with a histogram set of its own
#define GTXI_SHARE_STREAMS 4

extern "C" {
int gtxi_count_finish(gtx_ctx *c, void **d_out, int share);   // close the open count stream: result in the context's HBM vector (enqueued);
                                                               // share: only the member's classes, *d_out = its piece of the compact vector
int gtxi_coverage_finish(gtx_ctx *c, void **d_out);
void gtxi_fetch_info(gtx_ctx *c, gtx_count_info *info); // after a wait for the stream
int gtxi_scan_enqueue(gtx_ctx *c, const int32_t *reads, const int32_t *weights, int64_t n, const int32_t *class_len, int32_t n_classes,
                      int32_t step, int32_t size, char prep, uint32_t flags, const int64_t *class_offsets, void **d_out, int64_t *extent);
int gtxi_set_share(gtx_ctx *c, const uint8_t *owned, int32_t nClasses, const int32_t *regions, int64_t nRegions, int64_t offset);
int gtxi_count_device_share(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n, uint32_t flags, int slot, void *direct_out, void **d_piece, int64_t *pieceLen);
                                                        // direct_out (may be null): a result vector in file order that receives the member's regions at their places, instead of the compact piece
int gtxi_count_device_share_async(gtx_ctx *c, const void *d_reads, const void *d_weights, int64_t n, uint32_t flags, int slot, int set, hipStream_t run,
                                  void *direct_out, void **d_piece, int64_t *pieceLen);   // kernel + finalize on `run` with histogram set `set` (0 | 1), into the piece of compact vector `slot`
int gtxi_last_share_info(gtx_ctx *c, gtx_count_info *info);
void *gtxi_out_buffer(gtx_ctx *c);                      // the context's result vector in HBM (n_refs uint64)
int gtxi_ensure_out(gtx_ctx *c, int64_t n);             // ... with room for n uint64
int gtxi_scratch(gtx_ctx *c, size_t bytes, void **p);   // a second device buffer of the context
int gtxi_wait_direct(gtx_ctx *c);                       // the previous host-buffer call's DMA out of page-locked caller memory is done
hipStream_t gtxi_stream(gtx_ctx *c);
int gtxi_device(gtx_ctx *c);
int gtxi_pairs_on(gtx_ctx *c);                           // multi-interval regions declared, or multi-interval queries added to the open call
void gtxi_set_error(gtx_ctx *c, const char *msg);
}
