/* Opaque forward declarations only (see faidx.h). */
#ifndef NP_STUB_SAM_H
#define NP_STUB_SAM_H
#include "hts.h"
typedef struct bam_hdr_t bam_hdr_t;
typedef struct bam1_t bam1_t;
typedef htsFile samFile;
#endif
