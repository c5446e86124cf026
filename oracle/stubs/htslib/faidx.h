/* Opaque forward declaration only: lets the reference's hot-path headers parse
 * without htslib (nothing from htslib is called on the HMM / event-align path). */
#ifndef NP_STUB_FAIDX_H
#define NP_STUB_FAIDX_H
typedef struct faidx_t faidx_t;
#endif
