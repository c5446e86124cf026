/* oracle/stubs_full/htslib/hts.h -- TEST INFRASTRUCTURE.
 * Minimal stand-in for htslib 1.15.1 (the reference's pinned submodule, README.md:51-55; absent from this image), just
 * enough for the reference's read-level call-methylation translation units to COMPILE in place.  Types are this file's
 * own (no real htslib is ever linked, so no ABI has to match); the handful of accessors that actually execute on the
 * oracle path are restated in sam.h from the SAM/BAM specification; every other function is only declared and is
 * resolved at link time by an aborting stub (oracle/gen_abort_stubs.py). */
#ifndef NP_STUBFULL_HTS_H
#define NP_STUBFULL_HTS_H
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct htsFile { int dummy; } htsFile;
typedef struct hts_idx_t hts_idx_t;
typedef struct hts_itr_t { int dummy; } hts_itr_t;
typedef int64_t hts_pos_t;
typedef struct htsThreadPool { void* pool; int qsize; } htsThreadPool;
#define HTS_IDX_NOCOOR (-2)
#define HTS_IDX_START  (-3)
#define HTS_FMT_BAI 1
htsFile* hts_open(const char* fn, const char* mode);
int hts_close(htsFile* fp);
int hts_set_threads(htsFile* fp, int n);
void hts_idx_destroy(hts_idx_t* idx);
void hts_itr_destroy(hts_itr_t* iter);
const char* hts_parse_reg(const char* str, int* beg, int* end);
#ifdef __cplusplus
}
#endif
#endif
