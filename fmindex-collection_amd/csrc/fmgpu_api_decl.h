// fmgpu_api_decl.h — the per-width implementations of the C-ABI (include/fmgpu.h), declared inside `namespace fmgpu32::api` / `fmgpu64::api`.
// Included (without include guard, on purpose) once per namespace; fmgpu_abi.hip routes every extern "C" entry point to one of the two.
int fmgpu_index_create(const fmgpu_index_desc* desc, fmgpu_index_t* out);
int fmgpu_index_destroy(fmgpu_index_t h);
int fmgpu_index_info(fmgpu_index_t h, uint64_t* n, int32_t* sigma, int32_t* layout, int32_t* bidirectional, uint64_t* device_bytes);
int fmgpu_index_formats(fmgpu_index_t h, uint32_t* mask);
int fmgpu_index_accelerate(fmgpu_index_t h, int32_t kstep);
int fmgpu_index_accelerate_exact(fmgpu_index_t h, int32_t kstep, int32_t lut_len, int32_t walk);
int fmgpu_index_accelerate_search(fmgpu_index_t h, int32_t prefix_len, int32_t walk);
int fmgpu_index_accelerate_locate(fmgpu_index_t h, int32_t enable);
int fmgpu_index_accelerate_lf(fmgpu_index_t h, int32_t enable);
int fmgpu_string_query(fmgpu_index_t h, int which, const uint64_t* idx, const uint8_t* symb, const uint8_t* what, uint64_t count, uint64_t* out, void* stream);
int fmgpu_search_exact(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint64_t* out_lb, uint64_t* out_len, fmgpu_stats* stats, void* stream);
int fmgpu_search_exact_packed(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint64_t* out_interval, fmgpu_stats* stats, void* stream);
int fmgpu_search_exact_depth(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint32_t* out_depth, void* stream);
int fmgpu_search_scheme(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_scheme* scheme, uint64_t max_hits_per_query,
                        fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, void* stream);
int fmgpu_search_ng21(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_expanded_scheme* scheme, uint64_t max_hits_per_query,
                      fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, void* stream);
int fmgpu_search_backtracking(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint64_t max_errors, fmgpu_hit* out, uint64_t capacity,
                              uint64_t* out_count, fmgpu_stats* stats, void* stream);
int fmgpu_locate(fmgpu_index_t h, const uint64_t* rows, uint64_t count, uint64_t* out_seq, uint64_t* out_pos, uint64_t* out_steps, fmgpu_stats* stats, void* stream);
int fmgpu_cursor_extend(fmgpu_index_t h, int32_t direction, uint64_t count, const uint64_t* lb, const uint64_t* lb_rev, const uint64_t* len, const uint8_t* symb,
                        uint64_t* out_lb, uint64_t* out_lb_rev, uint64_t* out_len, void* stream);
int fmgpu_build_index(const uint8_t* seqs, const uint64_t* seq_off, uint64_t nseq, int32_t sigma, int32_t layout, uint64_t sampling_rate, int32_t bidirectional,
                      int32_t keep_host, fmgpu_index_t* out, fmgpu_built_t* built);
int fmgpu_index_save(fmgpu_index_t h, const char* path, int32_t include_tables);
int fmgpu_index_clone(fmgpu_index_t h, fmgpu_index_t* out);
int index_load(FILE* f, const void* file_header, fmgpu_index_t* out);      // (fmgpu_index_load has read and checked the 64-byte header: it names the row width)
