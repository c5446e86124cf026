/* oracle/stubs_full/slow5/slow5.h -- TEST INFRASTRUCTURE: stand-in for slow5lib (absent); SquiggleRead's constructor
 * only tests slow5_file_t::index (src/nanopolish_squiggle_read.cpp:83), and the oracle never takes that branch. */
#ifndef NP_STUBFULL_SLOW5_H
#define NP_STUBFULL_SLOW5_H
typedef struct slow5_file { void* index; } slow5_file_t;
typedef struct slow5_rec slow5_rec_t;
#endif
