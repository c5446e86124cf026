/* Opaque forward declarations only (see faidx.h). */
#ifndef NP_STUB_HTS_H
#define NP_STUB_HTS_H
#include <stdint.h>
typedef struct htsFile htsFile;
typedef struct hts_idx_t hts_idx_t;
typedef struct hts_itr_t hts_itr_t;
#endif
