/*
 * fpcc_host.h -- C ABI of libfpcc_host.so: the host-side entropy coders of the FastPCC hot path.
 *
 * These entry points are what a maintainer of the reference would bind in place of its two pybind11 extensions
 *   rans_ext_cpp          (/root/reference/lib/entropy_models/rans_coder/rans_wrapper.cpp:430-451)
 *   simple_rans_ext_cpp   (/root/reference/models/convolutional/lossy_coord_v3/rans_coder/simple_rans_wrapper.cpp:272-286)
 * Plain pointers and sizes only; no Python, torch or pybind types.  All functions are re-entrant; coder objects are
 * not shared between threads.  A negative return value is an error (see fpcc_host_strerror).
 *
 * Stream layout (identical to the reference, rans_byte.h:66-165): 32-bit state, L = 2^23, byte renormalisation,
 * 16-bit probabilities; symbols are pushed last-to-first and the stream is emitted as
 * [final state, 4 bytes LE][renormalisation bytes ...] so the decoder reads symbols first-to-last.
 */
#ifndef FPCC_HOST_H_
#define FPCC_HOST_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
    FPCC_HOST_OK = 0,
    FPCC_HOST_E_BUFFER = -1,   /* output buffer too small */
    FPCC_HOST_E_ARG = -2,      /* invalid argument (null pointer, bad size, zero-frequency symbol, ...) */
    FPCC_HOST_E_CDF = -3,      /* CDF cannot be made strictly increasing */
    FPCC_HOST_E_TIMEOUT = -4   /* a background job's input flag never became ready */
};
const char *fpcc_host_strerror(int64_t code);

/* Replaces batched_pmf_to_quantized_cdf / pmf_to_quantized_cdf<OVERFLOW> (cdf_ops.cpp:4-109,136-143), one row.
 * pmf[n] is read only.  cdf_out must hold n + 2 entries.  *offset is advanced by the number of trimmed leading bins
 * when overflow != 0.  Returns the CDF length (>= 3 in overflow mode, n + 1 otherwise). */
int64_t fpcc_pmf_to_quantized_cdf(const double *pmf, int64_t n, int overflow, int32_t *offset, uint32_t *cdf_out);

/* Replaces IndexedRansCoder::{encode, encode_with_indexes, decode, decode_with_indexes} for ONE batch unit
 * (rans_wrapper.cpp:89-185,206-279).  Tables are flattened: table t = cdf[cdf_start[t] .. cdf_start[t]+cdf_len[t]),
 * cdf[0] == 0, last == 65536.  index == NULL selects table (i % n_tables) as the reference's non-indexed mode does.
 * Encode writes the stream at the END of out[0..cap) and returns its length. */
int64_t fpcc_rans_indexed_encode(const int32_t *symbols, const int32_t *index, int64_t n,
                                 const uint32_t *cdf, const int64_t *cdf_start, const int64_t *cdf_len,
                                 const int32_t *offsets, int64_t n_tables, int overflow,
                                 uint8_t *out, int64_t cap);
int64_t fpcc_rans_indexed_decode(const uint8_t *stream, int64_t stream_len, const int32_t *index, int64_t n,
                                 const uint32_t *cdf, const int64_t *cdf_start, const int64_t *cdf_len,
                                 const int32_t *offsets, int64_t n_tables, int overflow, int32_t *symbols_out);

/* Replaces BinaryRansCoder::{encode, decode} for one batch unit (rans_wrapper.cpp:326-382,385-428).
 * prob1[i] = P(bit i == 1) * 65536, in [1, 65535]. */
int64_t fpcc_rans_binary_encode(const uint8_t *bits, const uint16_t *prob1, int64_t n, uint8_t *out, int64_t cap);
int64_t fpcc_rans_binary_decode(const uint8_t *stream, int64_t stream_len, const uint16_t *prob1, int64_t n,
                                uint8_t *bits_out);

/* Several independent binary streams at once, one host thread each (the reference's OpenMP-over-batch loop,
 * rans_wrapper.cpp:340-341, used here across the occupancy levels of one frame).  Stream s covers
 * bits[start[s] .. start[s+1]).  Its bytes are written at the END of out + s*cap_each .. +cap_each; len_out[s] gets the
 * length or a negative error code.  Returns 0 or the first error. */
int64_t fpcc_rans_binary_encode_multi(const uint8_t *bits, const uint16_t *prob1, const int64_t *start, int64_t n_streams,
                                      uint8_t *out, int64_t cap_each, int64_t *len_out, int n_threads);

/* Background coder pool: the same coders, run on the library's own threads so that entropy coding overlaps the GPU work
 * of later pyramid levels.  This has no counterpart in the reference, whose coders run inline between blocking .cpu()
 * calls (geo_lossl_em.py:95-114,59-74); the streams produced are the same bytes.
 * A job starts when *flag == ready (flag == NULL: at once).  The caller makes `flag` the last of a stream-ordered group of
 * device->host copies into pinned memory, so the job's inputs are complete when it fires; the library never calls HIP.
 * Results land in caller-owned buffers (*len_out: stream length at the END of out[0..cap), or a negative code) and
 * become valid after fpcc_pool_wait, which returns 0 or the first error of the jobs waited for. */
typedef struct fpcc_pool fpcc_pool;
fpcc_pool *fpcc_pool_new(int n_threads);
void fpcc_pool_free(fpcc_pool *p);
int64_t fpcc_pool_binary_encode(fpcc_pool *p, const volatile uint32_t *flag, uint32_t ready, const uint8_t *bits,
                                const uint16_t *prob1, int64_t n, uint8_t *out, int64_t cap, int64_t *len_out);
/* rans_encode_with_cdf (geo_lossl_em.py:59-74) as one job: offset = min(symbols) (or *offset_io when fixed_offset),
 * histogram -> quantised CDF (no overflow bin) -> one rANS stream.  cdf_out[cdf_cap] receives *cdf_len_out entries. */
int64_t fpcc_pool_histogram_encode(fpcc_pool *p, const volatile uint32_t *flag, uint32_t ready, const int32_t *symbols,
                                   int64_t n, int fixed_offset, int32_t *offset_io, uint32_t *cdf_out, int64_t cdf_cap,
                                   int64_t *cdf_len_out, uint8_t *out, int64_t cap, int64_t *len_out);
/* Single-table decode (rans_decode_with_cdf, geo_lossl_em.py:76-93) in the background.  *progress counts the symbols that
 * are final (first published after first_chunk symbols, then every 4096); fpcc_progress_wait blocks until it reaches
 * `needed` and returns it, or a negative code. */
int64_t fpcc_pool_table_decode(fpcc_pool *p, const uint8_t *stream, int64_t stream_len, int64_t n, const uint32_t *cdf,
                               int64_t cdf_len, int32_t offset, int32_t *symbols_out, int64_t first_chunk,
                               int64_t *progress);
/* fpcc_rans_binary_decode in the background: the occupancy levels of SEVERAL clouds coded in one traversal are decoded side by
 * side (one job per cloud; the reference decodes the clouds of a list one after the other, lossy_coord_v2/model.py:277-288).
 * *done becomes 1 when bits_out[0..n) is final, or the job's negative code; wait for it with fpcc_progress_wait(done, 1) -- other
 * jobs of the pool (a residual stream still being decoded) are not waited for. */
int64_t fpcc_pool_binary_decode(fpcc_pool *p, const uint8_t *stream, int64_t stream_len, const uint16_t *prob1, int64_t n,
                                uint8_t *bits_out, int64_t *done);
int64_t fpcc_progress_wait(const int64_t *progress, int64_t needed);
int64_t fpcc_pool_wait(fpcc_pool *p);

/* Replaces RansEncoder / RansDecoder of simple_rans_ext_cpp: one persistent stream, blocks pushed LIFO.
 * rows: uint16 [n_rows, width], entry s = upper edge of symbol s, last edge implicitly 65536; n_rows == n or 1. */
typedef struct fpcc_simple_enc fpcc_simple_enc;
typedef struct fpcc_simple_dec fpcc_simple_dec;
fpcc_simple_enc *fpcc_simple_enc_new(int64_t buf_bytes);
void fpcc_simple_enc_free(fpcc_simple_enc *);
int64_t fpcc_simple_enc_push(fpcc_simple_enc *, const uint16_t *rows, int64_t n_rows, int64_t width,
                             const uint16_t *symbols, int64_t n);                       /* RansEncoder::encode */
int64_t fpcc_simple_enc_push_bin(fpcc_simple_enc *, const uint16_t *edge, int64_t n_rows, const uint8_t *bits, int64_t n);
int64_t fpcc_simple_enc_push_ranges(fpcc_simple_enc *, const uint16_t *start, const uint16_t *freq_m1, int64_t n);
int64_t fpcc_simple_enc_finish(fpcc_simple_enc *, uint8_t *out, int64_t cap);           /* RansEncoder::flush */
/* The decoder copies nothing: `stream` must stay alive until the decoder is freed (the reference keeps a raw pointer
 * into the Python bytes object, simple_rans_wrapper.cpp:139-145). */
fpcc_simple_dec *fpcc_simple_dec_new(const uint8_t *stream, int64_t stream_len);        /* RansDecoder::flush */
void fpcc_simple_dec_free(fpcc_simple_dec *);
int64_t fpcc_simple_dec_pop(fpcc_simple_dec *, const uint16_t *rows, int64_t n_rows, int64_t width,
                            uint16_t *symbols_out, int64_t n);                           /* RansDecoder::decode */
/* Where the decoder stands: the 32-bit rANS state and the offset of the next unread byte.  A device decoder
 * (fpcc_simple_dec_pop_dev of libfpcc_hip) continues the same stream from there. */
int64_t fpcc_simple_dec_tell(const fpcc_simple_dec *d, uint32_t *state_out, int64_t *position_out);
int64_t fpcc_simple_dec_pop_bin(fpcc_simple_dec *, const uint16_t *edge, int64_t n_rows, uint8_t *bits_out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
