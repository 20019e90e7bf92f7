/* The calls INTEGRATION.md shows (sections 4 and 6), as a C translation unit: tests/test_abi.py compiles it with
 * -Wall -Werror against include/mvk.h and links it against libmvk.so, so the document's examples cannot drift from the
 * ABI.  Never run. */
#include <stddef.h>
#include <stdint.h>

#include "mvk.h"

int pipelined_batches(const mvk_model* model, const mvk_box* boxes, size_t n_boxes, const mvk_params* params, const double* const* d_soa,
                      uint64_t* const* d_mask, size_t n, size_t ld, int n_batches, void* upload_stream, void* reduce_stream) {
  mvk_pipe* pipe;
  if (mvk_pipe_create(model, boxes, n_boxes, params, /*device*/ 0, /*depth*/ 2, &pipe)) return -1;
  for (int k = 0; k < n_batches; ++k) {
    int ticket;
    if (mvk_pipe_is_valid_mask(pipe, d_soa[k], n, ld, MVK_LAYOUT_POSE7, d_mask[k], /*producer stream*/ upload_stream, &ticket)) return -1;
    if (mvk_pipe_wait(pipe, ticket, /*consumer stream*/ reduce_stream)) return -1;
  }
  if (mvk_pipe_sweep_mask(pipe, (const double[7]){0, 0, 0, 0, 0, 0, 1}, 0, n, 1u, d_mask[0], NULL)) return -1;
  if (mvk_pipe_eval_batch(pipe, d_soa[0], n, ld, MVK_LAYOUT_POSE7, NULL, NULL, NULL, NULL, MVK_PIPE_NO_PRODUCER, NULL)) return -1;
  if (mvk_pipe_wait(pipe, 0, MVK_PIPE_HOST)) return -1;
  (void)mvk_pipe_ctx(pipe, 0);
  (void)mvk_pipe_stream(pipe, 1);
  mvk_pipe_drain(pipe);
  mvk_pipe_destroy(pipe);
  return 0;
}

int sharded_step(mvk_ctx* ctx, int rank, int world, int local_rank, const double* d_soa, size_t n_total, size_t ld, int layout,
                 uint64_t* d_shard_mask, uint64_t* d_all_masks, uint64_t* d_count, void* stream) {
  unsigned char id[MVK_COMM_ID_BYTES];
  if (rank == 0 && mvk_comm_unique_id(id)) return -1; /* hand the bytes to the other ranks with whatever you have */
  mvk_comm* comm;
  if (mvk_comm_create(id, rank, world, /*device*/ local_rank, &comm)) return -1;
  size_t lo, hi, words;
  mvk_shard_range(n_total, rank, world, &lo, &hi, &words);
  mvk_is_valid_mask_dev(ctx, d_soa + lo, hi - lo, ld, layout, d_shard_mask, stream);
  mvk_gather_mask_dev(comm, d_shard_mask, words, d_all_masks /* world * words */, stream);
  mvk_mask_count_dev(d_all_masks, (size_t)world * words * 64, d_count, stream);
  mvk_allreduce_count_dev(comm, d_count, 1, stream);
  mvk_stats local, global;
  mvk_reduce_stats(ctx, &local);
  mvk_allreduce_stats(comm, &local, /*index offset*/ lo, &global, stream);
  mvk_exchange_masks_dev(comm, 0, d_shard_mask, words, d_all_masks, d_count, stream, stream);
  mvk_exchange_wait(comm, 0);
  mvk_comm_destroy(comm);
  return (int)global.valid_count;
}

int fp32_outputs(mvk_ctx* ctx, const double* d_soa, size_t n, size_t ld, uint8_t* d_flags, float* d_min_dist, float* d_margin,
                 float* d_margin_err, uint8_t* d_band, void* stream) {
  const double tol = mvk_f32_distance_tolerance(ctx, /*largest |root coordinate| of the batch*/ 5.0);
  (void)tol;
  return mvk_eval_batch_f32_dev(ctx, d_soa, n, ld, MVK_LAYOUT_POSE7, d_flags, d_min_dist, d_margin, d_margin_err, d_band, stream);
}
