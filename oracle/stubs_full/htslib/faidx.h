/* oracle/stubs_full/htslib/faidx.h -- TEST INFRASTRUCTURE (see hts.h).  The harness defines faidx_t and the two
 * functions the oracle path executes (an in-memory contig table). */
#ifndef NP_STUBFULL_FAIDX_H
#define NP_STUBFULL_FAIDX_H
#include "hts.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct faidx_t faidx_t;
faidx_t* fai_load(const char* fn);
faidx_t* fai_load3(const char* fn, const char* fnfai, const char* fngzi, int flags);
void fai_destroy(faidx_t* fai);
char* fai_fetch(const faidx_t* fai, const char* reg, int* len);
char* faidx_fetch_seq(const faidx_t* fai, const char* c_name, int p_beg_i, int p_end_i, int* len);
int faidx_has_seq(const faidx_t* fai, const char* seq);
int faidx_nseq(const faidx_t* fai);
const char* faidx_iseq(const faidx_t* fai, int i);
int faidx_seq_len(const faidx_t* fai, const char* seq);
#ifdef __cplusplus
}
#endif
#endif
