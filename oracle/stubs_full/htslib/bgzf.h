/* TEST INFRASTRUCTURE (see hts.h): declarations only. */
#ifndef NP_STUBFULL_BGZF_H
#define NP_STUBFULL_BGZF_H
#include "hts.h"
typedef struct BGZF BGZF;
#endif
