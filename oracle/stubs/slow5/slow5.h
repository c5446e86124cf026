/* Opaque forward declarations only (see ../htslib/faidx.h). */
#ifndef NP_STUB_SLOW5_H
#define NP_STUB_SLOW5_H
typedef struct slow5_file slow5_file_t;
typedef struct slow5_rec slow5_rec_t;
#endif
