// parquet_kernels.hpp -- launchers of kernels_parquet.hip (enqueue on plx::stream(), return immediately).
#pragma once
#include "core.hpp"
#include "parquet_device.hpp"
#include "parquet_zstd.hpp"

namespace plx {
namespace k {
void pq_snappy(const pq::DecompJob* jobs, uint32_t n_jobs, uint64_t bytes_out, uint32_t* err);
void pq_zstd(pq::ZstdBlock* blocks, const uint32_t* order, uint32_t n_compressed, uint32_t n_huf_only, const pq::ZstdHufDesc* hufs, const pq::ZstdFseDesc* fses, const pq::ZstdStream* streams,
             uint32_t n_streams, uint64_t bytes_in, uint64_t bytes_out, uint32_t* err);
void pq_page_prepare(pq::PageDesc* pages, uint32_t n_pages, uint32_t* err);
void pq_count_runs(const pq::PageDesc* pages, uint32_t n_pages, bool levels, uint32_t* counts, uint32_t* err);
void pq_fill_runs(const pq::PageDesc* pages, uint32_t n_pages, const uint64_t* offs, pq::RunEntry* runs);
void pq_validity(const pq::PageDesc* pages, uint32_t n_pages, const pq::RunEntry* runs, const uint64_t* offs, uint64_t n_rows, uint64_t* validity, uint32_t* popc,
                 uint32_t* err);
void pq_page_valid0(pq::PageDesc* pages, uint32_t n_pages, const uint64_t* validity, const uint64_t* word_prefix);
void pq_decode(const pq::ColumnDecode& c, void* out, uint32_t out_width, uint64_t encoded_bytes, uint32_t* err);
}  // namespace k
}  // namespace plx
