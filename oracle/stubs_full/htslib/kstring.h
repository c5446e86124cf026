/* TEST INFRASTRUCTURE (see hts.h): declarations only. */
#ifndef NP_STUBFULL_KSTRING_H
#define NP_STUBFULL_KSTRING_H
#include <stddef.h>
typedef struct kstring_t { size_t l, m; char* s; } kstring_t;
#endif
